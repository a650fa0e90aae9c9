#!/bin/bash
# Round 6, visit Q: LoLa-MNIST, the unchanged per-call sequence (deferred) against the mirror's batched conveniences: per-kernel device time per image from two kernel traces
O=gpurun_out/r06q; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
for pat in batched literal; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_$pat -- python $R/tools/lola_unchanged_caller.py LoLa --reps 100 --only $pat > $R/$O/run_$pat.txt 2> $R/$O/prof_$pat.err)
  KT=$(find $O/prof_$pat -name "*kernel_trace.csv" | head -1)
  python tools/trace_gaps.py $KT 0.5 4 > $O/gaps_$pat.txt 2>&1; find $O/prof_$pat -name "*kernel_trace.csv" -delete
  echo "== $pat"; tail -1 $O/run_$pat.txt | cut -c1-250; head -44 $O/gaps_$pat.txt | cut -c1-130
done
