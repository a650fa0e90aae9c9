#!/bin/bash
# Round 6, visit AX: three parts inside the deferred flush as well (CN_SQ_HALVES=2 CN_SQ_PARTS=3) for the unchanged caller
R=$(pwd); O=$R/gpurun_out/r06ax; mkdir -p $O
for rep in 1 2 3; do for hv in 1 2; do
  CN_SQ_HALVES=$hv CN_SQ_PARTS=3 python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('sq_halves $hv parts 3 rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'])" | tee -a $O/ab.txt
done; done
