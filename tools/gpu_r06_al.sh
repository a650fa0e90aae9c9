#!/bin/bash
# Round 6, visit AL: defer_stagger (form 3) again now that the upload ring no longer laps inside a window (a lap = a wait for the stream = that context's queue runs dry and the phase between the primes is lost)
R=$(pwd); O=$R/gpurun_out/r06al; mkdir -p $O
for rep in 1 2 3; do for st in 0 1; do
  CN_DEFER_STAGGER=$st python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps ${AL_STEPS:-20} 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('stagger $st rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'], (r.get('host') or {}).get('pin_ring_laps'))" | tee -a $O/ab.txt
done; done
