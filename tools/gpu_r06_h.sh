#!/bin/bash
# Round 6, visit H: (1) the default bench line on the round-6 tree (lock-free unchanged caller, zero fold); (2) LoLa: one launch chain for all plaintext primes - the bound
# (tools/sumslots_merge_probe.py, tools/chain_concurrency_probe.py); (3) the unchanged LoLa caller: flush host times per queue level (CN_DEFER_TRACE=2)
O=gpurun_out/r06h; mkdir -p $O
( time python bench.py > $O/bench_default_flags.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python -c "
import json
d=json.loads(open('$O/bench_default_flags.json').read().strip().splitlines()[-1])
print('default flags:', d['value'], d['steps'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['square']['ms_per_chain'])
print('literal', d['literal_call_sequence'])
u=d['unchanged_caller']; print('unchanged', u['frac_of_batched'], u.get('at_visible_cpu_count'), u['windows_ms'], u.get('locked'), u.get('skipped_taps'))
print('lola', {k: d['lola'].get(k) for k in ('ms_per_image','verified','unchanged_caller_ms','unchanged_frac_of_batched','batched_from_the_same_host_ms','child_wall_s')})
print('cifar', {k: d['cifar'].get(k) for k in ('s_per_image','ms_per_image','verified','child_wall_s')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
timeout 300 python tools/sumslots_merge_probe.py 2>&1 | tail -8 | tee $O/sumslots_merge.txt
timeout 300 python tools/chain_concurrency_probe.py LoLa 2>&1 | tail -10 | tee $O/chain_concurrency.txt
CN_DEFER_TRACE=2 timeout 300 python tools/lola_unchanged_caller.py LoLa --reps 3 > $O/lola.txt 2> $O/lola.err
ctx=$(grep "flush of" $O/lola.err | tail -1 | awk '{print $2}')
grep "$ctx" $O/lola.err | tail -120 | grep "flush of" | awk '{n++; s+=$(NF-4)} END {print n, "flushes of the last lines,", s, "us"}'
grep "$ctx" $O/lola.err | tail -90 | cut -c1-160
tail -6 $O/lola.txt | cut -c1-300
