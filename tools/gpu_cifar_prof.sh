#!/bin/bash
OUT=gpurun_out/cifarprof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/tools/cifar_latency.py > $R/$OUT/cifar.txt 2>&1)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -delete
head -14 $OUT/trace_summary.txt
