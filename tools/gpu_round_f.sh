#!/bin/bash
mkdir -p gpurun_out/rf
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rf/bench.json 2> gpurun_out/rf/bench.err
cut -c1-220 gpurun_out/rf/bench.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/rf/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/gpurun_out/rf/prof_bench.json 2> $R/gpurun_out/rf/prof.err)
KT=$(find gpurun_out/rf/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/rf/trace_summary.txt 2>&1
find gpurun_out/rf/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rf/kernel_stats.csv \;
find gpurun_out/rf/prof -name "*kernel_trace.csv" -delete
head -16 gpurun_out/rf/trace_summary.txt
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_layers.py -m gpu -x -q 2>&1 | tail -3
