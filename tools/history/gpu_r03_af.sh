#!/bin/bash
# visit AF: the serialised kernel trace of the record visit again, without the relinearize-late section (it runs both channels at once and inflates the averages)
OUT=gpurun_out/r03z; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-relinearize-late --serialize --stagger 0 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof2 -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; ; find $OUT/prof2 -name "*kernel_trace.csv" -delete
head -16 $OUT/trace_summary.txt | cut -c1-130
