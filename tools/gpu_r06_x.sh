#!/bin/bash
# Round 6, visit X: the default bench line with 20-batch windows for the unchanged caller
O=gpurun_out/r06x; mkdir -p $O
( time python bench.py > $O/bench_default_flags.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python -c "
import json
d=json.loads(open('$O/bench_default_flags.json').read().strip().splitlines()[-1])
print('default flags:', d['value'], d['steps'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'])
print('literal', d['literal_call_sequence'])
u=d['unchanged_caller']; print('unchanged', u['frac_of_batched'], u.get('at_visible_cpu_count'), u['windows_ms'], u.get('locked'), u.get('skipped_taps'), u['timing'])
print('lola', {k: d['lola'].get(k) for k in ('ms_per_image','verified','unchanged_caller_ms','unchanged_frac_of_batched')})
print('cifar', {k: d['cifar'].get(k) for k in ('s_per_image','verified')})
"
