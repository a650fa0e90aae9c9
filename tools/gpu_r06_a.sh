#!/bin/bash
# Round 6, visit A: where the unchanged CryptoNets caller loses its 2-3 ms against the batched path on the round-5 tree: timings with and without the literal
# padded taps, flush host times (CN_DEFER_TRACE=2), kernel traces of both with the idle gaps (tools/trace_gaps.py)
O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
python tools/replay_reference_calls.py --trained --threads 4,16 --literal-threads 16,256 --steps 5 > $O/replay.txt 2> $O/replay.err; cut -c1-330 $O/replay.txt
CN_DEFER_TRACE=2 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 2 > $O/trace16.txt 2> $O/trace16.err
grep "flush of" $O/trace16.err | tail -40 | cut -c1-160
for mode in skipped literal; do
  if [ $mode = skipped ]; then A="--threads 16 --steps 4"; else A="--threads 1 --literal-threads 16 --steps 4"; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $R/$O/prof_$mode -- python $R/tools/replay_reference_calls.py --trained $A > $R/$O/prof_$mode.txt 2> $R/$O/prof_$mode.err)
  KT=$(find $O/prof_$mode -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $KT 0.25 30 > $O/gaps_$mode.txt 2>&1
  find $O/prof_$mode -name "*kernel_trace.csv" -delete
  echo "== $mode"; head -45 $O/gaps_$mode.txt | cut -c1-200
done
