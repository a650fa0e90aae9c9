#!/bin/bash
mkdir -p gpurun_out/rh
timeout 900 python -m pytest tests/test_cryptonets_mnist.py tests/test_gpu_evaluator.py -m gpu -x -q > gpurun_out/rh/pytest.txt 2>&1
tail -4 gpurun_out/rh/pytest.txt
python bench.py --steps 5 --warmup 2 > gpurun_out/rh/bench.json 2> gpurun_out/rh/bench.err
cut -c1-220 gpurun_out/rh/bench.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/rh/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/gpurun_out/rh/prof_bench.json 2> $R/gpurun_out/rh/prof.err)
KT=$(find gpurun_out/rh/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/rh/trace_summary.txt 2>&1
find gpurun_out/rh/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rh/kernel_stats.csv \;
find gpurun_out/rh/prof -name "*kernel_trace.csv" -delete
head -14 gpurun_out/rh/trace_summary.txt
