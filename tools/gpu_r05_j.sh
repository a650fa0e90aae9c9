#!/bin/bash
# Round 5, visit J: per-kernel GPU time of the unchanged caller (literal taps) in steady state, 64 caller threads
O=gpurun_out/r05j; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $R/$O/prof -- python $R/tools/replay_reference_calls.py --trained --threads 4 --literal-threads 64 --steps 12 > $R/$O/replay.txt 2> $R/$O/replay.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1)
tail -1 $O/replay.txt | cut -c1-200
python tools/trace_gaps.py $KT 0.25 6 | tee $O/gaps.txt | cut -c1-200
rm -f $KT
