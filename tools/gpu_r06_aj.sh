#!/bin/bash
# Round 6, visit AJ: pairing of the gather lists inside the deferred flush (CN_DEFER_PAIR=1) again, now over 20-batch windows (visit of profiles/r06_defer_pair_ab.txt: 5-batch windows)
R=$(pwd); O=$R/gpurun_out/r06aj; mkdir -p $O
for rep in 1 2 3; do for pair in 0 1; do
  CN_DEFER_PAIR=$pair REPLAY_HOST_TIMES=1 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 2> $O/host_$pair.txt | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('pair $pair rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'])" | tee -a $O/ab.txt
  grep "callers returned" $O/host_$pair.txt | sed 's/.*, \([0-9.]*\)\]; device/last caller return \1; device/' | tee -a $O/ab.txt
done; done
