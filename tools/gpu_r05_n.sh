#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- python $R/tools/ks14_probe.py 5488 ks_pair14=0 > $R/$O/probe.txt 2>&1)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT | head -8 | cut -c1-140; rm -f $KT
cat $O/probe.txt | tail -3
CN_KS_XCD=0 timeout 300 python tools/ks14_probe.py 5488 ks_pair14=0,ks_xcd=0 2>&1 | tail -2
