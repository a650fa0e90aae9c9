#!/bin/bash
mkdir -p gpurun_out/rk
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/rk/pytest.txt 2>&1
tail -6 gpurun_out/rk/pytest.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rk/bench.json 2> gpurun_out/rk/bench.err
cut -c1-220 gpurun_out/rk/bench.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/rk/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/gpurun_out/rk/prof_bench.json 2> $R/gpurun_out/rk/prof.err)
KT=$(find gpurun_out/rk/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/rk/trace_summary.txt 2>&1
find gpurun_out/rk/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rk/kernel_stats.csv \;
find gpurun_out/rk/prof -name "*kernel_trace.csv" -delete
head -14 gpurun_out/rk/trace_summary.txt
