#!/bin/bash
# kernel trace of the LoLa-CIFAR layer shapes (tools/cifar_latency.py)
OUT=gpurun_out/cifarprof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/tools/cifar_latency.py > $R/$OUT/cifar.txt 2>&1)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -delete
head -24 $OUT/trace_summary.txt | cut -c1-130
tail -3 $OUT/cifar.txt | cut -c1-250
