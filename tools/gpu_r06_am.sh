#!/bin/bash
# Round 6, visit AM: is the lock step of the two prime queues an attractor?  ONE device-side offset (REPLAY_OFFSET fronts of filler on prime 0, prime 1 waits for it once) at the start of a 20-batch window
R=$(pwd); O=$R/gpurun_out/r06am; mkdir -p $O
for rep in 1 2 3; do for off in 0 1 2; do
  REPLAY_OFFSET=$off python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('offset $off rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'])" | tee -a $O/ab.txt
done; done
