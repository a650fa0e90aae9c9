#!/bin/bash
mkdir -p gpurun_out/rj
python tools/ks_probe.py > gpurun_out/rj/ks.txt 2>&1; cat gpurun_out/rj/ks.txt
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rj/bench.json 2> gpurun_out/rj/bench.err
cut -c1-220 gpurun_out/rj/bench.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/rj/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/gpurun_out/rj/prof_bench.json 2> $R/gpurun_out/rj/prof.err)
KT=$(find gpurun_out/rj/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/rj/trace_summary.txt 2>&1
find gpurun_out/rj/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rj/kernel_stats.csv \;
find gpurun_out/rj/prof -name "*kernel_trace.csv" -delete
head -14 gpurun_out/rj/trace_summary.txt
