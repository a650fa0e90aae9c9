#!/bin/bash
# Round 6, closing record visit: GPU suite + smoke + the default bench line (incl. the lola / cifar children) + a 20-step line + a serialised kernel trace of the batch +
# the HBM traffic passes of the NTT launch (roofline.traffic)
O=gpurun_out/r06final; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
bash tools/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1; tail -12 $O/pmc_traffic.txt | cut -c1-200
cp gpurun_out/pmc_traffic/ntt_hbm_traffic.json $O/ 2>/dev/null; cp gpurun_out/pmc_traffic/fetch_size_counter_collection.csv $O/ 2>/dev/null; cp gpurun_out/pmc_traffic/write_size_counter_collection.csv $O/ 2>/dev/null
mkdir -p profiles; cp $O/ntt_hbm_traffic.json profiles/r06_ntt_hbm_traffic.json 2>/dev/null     # the bench line reads the latest committed traffic file (here: the one of this visit)
( time python bench.py > $O/bench_default_flags.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
python -c "
import json
d=json.loads(open('$O/bench_default_flags.json').read().strip().splitlines()[-1])
print('default flags:', d['value'], d['steps'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['roofline']['traffic_source'], d['key_switch']['ms_per_launch'], d['square']['ms_per_chain'])
print('literal', d['literal_call_sequence'])
u=d['unchanged_caller']; print('unchanged', u['frac_of_batched'], u.get('at_visible_cpu_count'), u['windows_ms'], u.get('locked'), u.get('skipped_taps'))
print('lola', {k: d['lola'].get(k) for k in ('ms_per_image','verified','unchanged_caller_ms','unchanged_frac_of_batched','batched_from_the_same_host_ms','child_wall_s')})
print('cifar', {k: d['cifar'].get(k) for k in ('s_per_image','verified','child_wall_s')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
python bench.py --steps 20 --warmup 3 --no-single-image > $O/bench.json 2>> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['unchanged_caller']['frac_of_batched'])"
python bench.py --workload lola --steps 20 --warmup 3 > $O/bench_lola.json 2>> $O/bench.err; python bench.py --workload cifar --steps 3 --warmup 2 > $O/bench_cifar.json 2>> $O/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > /dev/null 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/bench_kernel_trace_summary.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
head -16 $O/bench_kernel_trace_summary.txt | cut -c1-130
# the literal unchanged caller, traced (what the device runs for the reference's own call sequence)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_literal -- python $R/tools/replay_reference_calls.py --trained --threads 1 --literal-threads 16 --steps 6 > $R/$O/prof_literal.txt 2> $R/$O/prof_literal.err)
KT=$(find $O/prof_literal -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py $KT 0.3 8 > $O/literal_caller_trace.txt 2>&1; find $O/prof_literal -name "*kernel_trace.csv" -delete
head -30 $O/literal_caller_trace.txt | cut -c1-150
