#!/bin/bash
# Round 6, visit AO: pairing + defer_stagger TOGETHER (the staggered pattern needs the key-switch phase and the rest of a batch balanced: the unchanged caller's rest is longer by its slower scalar products)
R=$(pwd); O=$R/gpurun_out/r06ao; mkdir -p $O
for rep in 1 2 3; do for both in 0 1; do
  CN_DEFER_PAIR=$both CN_DEFER_STAGGER=$both python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('pair+stagger $both rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'), r['words_identical'])" | tee -a $O/ab.txt
done; done
