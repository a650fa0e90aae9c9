#!/bin/bash
OUT=gpurun_out/lolaprof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/tools/lola_latency.py > $R/$OUT/lola.txt 2>&1)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -delete
head -36 $OUT/trace_summary.txt
tail -5 $OUT/lola.txt | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-200
