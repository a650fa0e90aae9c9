#!/bin/bash
# Round-5 closing visit: GPU suite + smoke + the default bench line (incl. the lola / cifar children) + a serialised kernel trace of the batch
O=gpurun_out/r05final; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
( time python bench.py > $O/bench_default_flags.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python -c "
import json
d=json.loads(open('$O/bench_default_flags.json').read().strip().splitlines()[-1])
print('default flags:', d['value'], d['steps'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['square']['ms_per_chain'])
print('unchanged', d['unchanged_caller']['frac_of_batched'], d['unchanged_caller'].get('at_visible_cpu_count'), d['unchanged_caller']['windows_ms'])
print('lola', {k: d['lola'].get(k) for k in ('ms_per_image','ms_per_image_samples','verified','unchanged_caller_ms','unchanged_frac_of_batched','child_wall_s')})
print('cifar', {k: d['cifar'].get(k) for k in ('s_per_image','ms_per_image','verified','child_wall_s')})
print('cifar ks', json.dumps(d['cifar'].get('key_switch'))[:900])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
python bench.py --steps 20 --warmup 3 > $O/bench.json 2>> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['unchanged_caller']['frac_of_batched'])"
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > /dev/null 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/bench_kernel_trace_summary.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
head -16 $O/bench_kernel_trace_summary.txt | cut -c1-130
