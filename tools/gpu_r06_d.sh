#!/bin/bash
# Round 6, visit D: kernel traces of the unchanged caller on the lock-free tree (skipped taps / literal, 16 threads) with the idle gaps; flush host times
O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
CN_DEFER_TRACE=2 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 2 > $O/trace16.txt 2> $O/trace16.err
grep "flush of" $O/trace16.err | tail -24 | cut -c1-160
for mode in skipped literal batched; do
  if [ $mode = skipped ]; then A="--threads 16 --steps 6"; elif [ $mode = literal ]; then A="--threads 1 --literal-threads 16 --steps 6"; else A="--threads 1 --steps 30"; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $R/$O/prof_$mode -- python $R/tools/replay_reference_calls.py --trained $A > $R/$O/prof_$mode.txt 2> $R/$O/prof_$mode.err)
  KT=$(find $O/prof_$mode -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $KT ${FRAC:-0.3} 12 > $O/gaps_$mode.txt 2>&1
  find $O/prof_$mode -name "*kernel_trace.csv" -delete
  echo "== $mode"; head -40 $O/gaps_$mode.txt | cut -c1-170
done
