#!/bin/bash
# Round 5, visit I: GPU idle gaps of the unchanged CryptoNets caller (literal taps, merged calls) at 16 and 256 caller threads, and of the batched path
O=gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for t in 16 256; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $R/$O/prof$t -- python $R/tools/replay_reference_calls.py --trained --threads 4 --literal-threads $t --steps 8 > $R/$O/replay$t.txt 2> $R/$O/replay$t.err)
  KT=$(find $O/prof$t -name "*kernel_trace.csv" | head -1)
  echo "== literal, $t threads"; tail -1 $O/replay$t.txt | cut -c1-200
  python tools/trace_gaps.py $KT 0.12 22 | tee $O/gaps$t.txt | cut -c1-220
  rm -f $KT
done
