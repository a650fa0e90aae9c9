#!/bin/bash
# Round 6, visit AQ: sq_halves for the unchanged caller, more repetitions (tools/replay_reference_calls.py, 20-batch windows)
R=$(pwd); O=$R/gpurun_out/r06aq; mkdir -p $O
for rep in 1 2 3 4; do for hv in 0 1; do
  CN_SQ_HALVES=$hv python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16,256 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('halves $hv rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'])" | tee -a $O/ab.txt
done; done
