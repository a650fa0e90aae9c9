#!/bin/bash
# Round 6, visit I: the split runtime units - whole GPU suite, smoke, the default bench line; the locked literal caller with parked / direct releases, fold on / off
O=gpurun_out/r06i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed|FAILED|ERROR" $O/pytest.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
for args in "--locked" "--locked --direct-free"; do for fold in 1 0; do
  CN_FOLD_ZERO=$fold python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 5 $args > $O/replay.txt 2> $O/replay.err
  python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln)
    if 'padded' in d['caller']: print('$args fold $fold:', d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'), d.get('launches_per_batch'))"
done; done
( time python bench.py > $O/bench_default_flags.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python -c "
import json
d=json.loads(open('$O/bench_default_flags.json').read().strip().splitlines()[-1])
print('default flags:', d['value'], d['steps'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['square']['ms_per_chain'])
print('literal', d['literal_call_sequence'])
u=d['unchanged_caller']; print('unchanged', u['frac_of_batched'], u.get('at_visible_cpu_count'), u['windows_ms'], u.get('locked'), u.get('skipped_taps'))
print('lola', {k: d['lola'].get(k) for k in ('ms_per_image','verified','unchanged_caller_ms','unchanged_frac_of_batched','batched_from_the_same_host_ms','child_wall_s')})
print('cifar', {k: d['cifar'].get(k) for k in ('s_per_image','verified','child_wall_s')})
"
