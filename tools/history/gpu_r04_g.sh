#!/bin/bash
# round 4, visit g: where does the unchanged CryptoNets caller (literal padded taps, 16 threads) spend its extra 3 ms per batch?  kernel trace of the replay + CN_DEFER_TRACE;
# LoLa-MNIST per-layer latency with and without the recorded graph; LoLa-CIFAR per-layer latency
OUT=gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 4 > $R/$OUT/replay.txt 2> $R/$OUT/replay.err)
cat $OUT/replay.txt | cut -c1-330
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/replay_trace_summary.txt 2>&1
rm -rf $OUT/prof
head -40 $OUT/replay_trace_summary.txt
CN_DEFER_TRACE=1 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 1 2> $OUT/defer_trace.txt > /dev/null
grep -c "defer" $OUT/defer_trace.txt; tail -40 $OUT/defer_trace.txt | cut -c1-200
python tools/lola_latency.py LoLa --graph > $OUT/lola_latency.txt 2>&1; tail -12 $OUT/lola_latency.txt | cut -c1-300
python tools/cifar_latency.py > $OUT/cifar_latency.txt 2>&1; tail -6 $OUT/cifar_latency.txt | cut -c1-400
