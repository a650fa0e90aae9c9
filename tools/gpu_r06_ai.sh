#!/bin/bash
# Round 6, visit AI: the unchanged caller measured by tools/replay_reference_calls.py and by bench.py on ONE box, alternating (is there a systematic difference between the two harnesses?)
R=$(pwd); O=$R/gpurun_out/r06ai; mkdir -p $O
for rep in 1 2; do
  python tools/replay_reference_calls.py --trained --threads 4 --literal-threads 16 --steps 20 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    r = json.loads(ln); print('tool  rep $rep:', r['caller'][:40], r['threads'], r['ms_per_batch'], r.get('frac_of_batched'))" | tee -a $O/ab.txt
  python bench.py --no-cpu-baseline --no-single-image --no-relinearize-late 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); u = d['unchanged_caller']
print('bench rep $rep: batched', d['ms_per_step'], 'literal', u['ms_per_step'], u['windows_ms']['unchanged'], 'skipped', u['skipped_taps']['ms_per_step'], 'batched windows', u['windows_ms']['batched'])" | tee -a $O/ab.txt
  python bench.py --stagger 0 --no-cpu-baseline --no-single-image --no-relinearize-late 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); u = d['unchanged_caller']
print('bench --stagger 0 rep $rep: batched', d['ms_per_step'], 'literal', u['ms_per_step'], u['windows_ms']['unchanged'], 'skipped', u['skipped_taps']['ms_per_step'], 'batched windows', u['windows_ms']['batched'])" | tee -a $O/ab.txt
done
