#!/bin/bash
# Round 6, visit AR: does sq_halves hurt the unchanged caller because a context's second stream shares a hardware queue with the OTHER context's stream?  GPU_MAX_HW_QUEUES=8, halves 0 / 1
R=$(pwd); O=$R/gpurun_out/r06ar; mkdir -p $O
for rep in 1 2 3; do for hv in 0 1; do
  GPU_MAX_HW_QUEUES=8 CN_SQ_HALVES=$hv python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('8 queues, halves $hv rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'])" | tee -a $O/ab.txt
done; done
