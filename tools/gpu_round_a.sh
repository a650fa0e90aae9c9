#!/bin/bash
# one GPU visit: stdout purity under RCCL init, key-switch fence A/B, bench + kernel trace, GPU tests
set -x
mkdir -p gpurun_out/ra
BENCH_FORCE_DIST=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ra/dist_stdout.txt 2> gpurun_out/ra/dist_stderr.txt
wc -l gpurun_out/ra/dist_stdout.txt
python tools/ks_probe.py > gpurun_out/ra/ks_default.txt 2>&1
CNHIP_LIB=$PWD/gpurun_variants/libcnhip_nofence.so python tools/ks_probe.py > gpurun_out/ra/ks_nofence.txt 2>&1
python bench.py --steps 5 --warmup 2 > gpurun_out/ra/bench.json 2> gpurun_out/ra/bench.err
cat gpurun_out/ra/bench.json
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/ra/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > $R/gpurun_out/ra/prof_bench.json 2> $R/gpurun_out/ra/prof.err)
KT=$(find gpurun_out/ra/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/ra/trace_summary.txt 2>&1
find gpurun_out/ra/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/ra/kernel_stats.csv \;
find gpurun_out/ra/prof -name "*kernel_trace.csv" -delete
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/ra/pytest.txt 2>&1
tail -5 gpurun_out/ra/pytest.txt
head -30 gpurun_out/ra/ks_default.txt gpurun_out/ra/ks_nofence.txt
