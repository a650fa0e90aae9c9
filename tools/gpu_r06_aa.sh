#!/bin/bash
# Round 6, visit AA: LoLa-MNIST unchanged per-call sequence (deferred): where each plaintext-prime chain waits (per-queue gaps of a kernel trace)
O=gpurun_out/r06aa; mkdir -p $O
export TMPDIR=/tmp; R=$PWD
for pat in literal batched; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $R/$O/prof_$pat -- python $R/tools/lola_unchanged_caller.py LoLa --reps 100 --only $pat > $R/$O/run_$pat.txt 2> $R/$O/prof_$pat.err)
  KT=$(find $O/prof_$pat -name "*kernel_trace.csv" | head -1)
  echo "== $pat"; python tools/trace_queue_gaps.py $KT 0.4 10 > $O/queue_gaps_$pat.txt 2>&1; find $O/prof_$pat -name "*kernel_trace.csv" -delete
  head -26 $O/queue_gaps_$pat.txt | cut -c1-150
done
