#!/bin/bash
# visit AG: kernel trace of the LoLa-MNIST bench line (eager) on the final tree + chain probe
O=gpurun_out/r03ag; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/bench.py --workload lola --steps 10 --warmup 2 --no-unchanged-caller > $R/$O/bench_lola_prof.json 2>/dev/null)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/lola_trace_summary.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
head -30 $O/lola_trace_summary.txt | cut -c1-130
python tools/chain_concurrency_probe.py LoLa > $O/chain_probe.txt 2>&1; grep -v "^    " $O/chain_probe.txt | tail -10
