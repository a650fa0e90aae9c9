#!/bin/bash
# Round-2 visit F: full GPU suite (k+2 auxiliary base, upload checks, issue probes), default bench line, kernel trace of the bench,
# UNPROFILED LoLa-MNIST / LoLa-CIFAR latencies, key-switch micro-benchmark, unchanged-caller replay table
OUT=gpurun_out/r02f
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cut -c1-2500 $OUT/bench.json; tail -2 $OUT/bench.err | cut -c1-200
timeout 300 ./tools/ubench_ks > $OUT/ubench_ks.txt 2>&1
cat $OUT/ubench_ks.txt | cut -c1-120
timeout 300 python tools/lola_latency.py LoLa --graph > $OUT/lola_mnist_latency.txt 2>&1; tail -8 $OUT/lola_mnist_latency.txt | cut -c1-200
timeout 600 python tools/cifar_latency.py > $OUT/lola_cifar_latency.txt 2>&1; tail -4 $OUT/lola_cifar_latency.txt | cut -c1-200
CN_AUX_EXTRA=0 timeout 600 python tools/cifar_latency.py > $OUT/lola_cifar_latency_seal_aux.txt 2>&1; tail -2 $OUT/lola_cifar_latency_seal_aux.txt | cut -c1-200
timeout 600 python tools/replay_reference_calls.py --threads 1,4,8,16,32,64 --immediate --trained > $OUT/unchanged_caller_replay.txt 2>&1; tail -12 $OUT/unchanged_caller_replay.txt | cut -c1-160
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/benchtrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $R/$OUT/bench_traced.json 2> $R/$OUT/bench_traced.err)
KT=$(find $OUT/benchtrace -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/bench_kernel_trace_summary.txt 2>&1
KS=$(find $OUT/benchtrace -name "*kernel_stats.csv" | head -1); cp $KS $OUT/bench_kernel_stats.csv
head -16 $OUT/bench_kernel_trace_summary.txt | cut -c1-140
find $OUT -name "*kernel_trace.csv" -delete
