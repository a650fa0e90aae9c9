#!/bin/bash
# Round 6, visit AF: does the lap of the pinned upload ring (8 MiB, hipStreamSynchronize per lap) cost the unchanged caller time?  CN_PIN_RING_MIB = 8 / 64, alternating, 20-batch windows
R=$(pwd); O=$R/gpurun_out/r06af; mkdir -p $O
for rep in 1 2 3; do for mib in 8 64; do
  CN_PIN_RING_MIB=$mib python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln)
    print('ring $mib MiB rep $rep:', r['caller'][:48], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'], (r.get('host') or {}).get('pin_ring_laps'))
" | tee -a $O/pin_ring_ab.txt
done; done
