#!/bin/bash
# round 4, visit l: per-layer DEVICE time of the CryptoNets batch on the final tree (roctx ranges that synchronise; kernel + marker trace, no counters)
OUT=gpurun_out/r04l
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && CN_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize --stagger 0 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
find $OUT/prof -name "*marker*stats*.csv" -exec cp {} $OUT/marker_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*marker_api_trace.csv" -delete
cat $OUT/marker_stats.csv
