#!/bin/bash
mkdir -p gpurun_out/rg
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/rg/pytest.txt 2>&1
tail -6 gpurun_out/rg/pytest.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rg/bench.json 2> gpurun_out/rg/bench.err
cut -c1-220 gpurun_out/rg/bench.json
timeout 600 python tools/lola_latency.py > gpurun_out/rg/lola.txt 2>&1
tail -5 gpurun_out/rg/lola.txt | cut -c1-300
timeout 900 python tools/cifar_latency.py > gpurun_out/rg/cifar.txt 2>&1
tail -4 gpurun_out/rg/cifar.txt | cut -c1-300
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/rg/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/gpurun_out/rg/prof_bench.json 2> $R/gpurun_out/rg/prof.err)
KT=$(find gpurun_out/rg/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/rg/trace_summary.txt 2>&1
find gpurun_out/rg/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rg/kernel_stats.csv \;
find gpurun_out/rg/prof -name "*kernel_trace.csv" -delete
head -14 gpurun_out/rg/trace_summary.txt
