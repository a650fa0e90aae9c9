#!/bin/bash
# Round 6, visit F: why the squaring overlap does not pay - the chain alone on the device, HIP-event timed and traced
O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
python tools/square_overlap_probe.py 2>&1 | tee $O/probe.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $R/$O/prof -- python $R/tools/square_overlap_probe.py > $R/$O/probe_traced.txt 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/square_overlap_timeline.py $KT | tee $O/timeline.txt; find $O/prof -name "*kernel_trace.csv" -delete
