#!/bin/bash
# Round 6, visit AT: CN_SQ_PARTS 2 / 3 / 5 (unstaggered batched loop) against the staggered loop, three alternating rounds
R=$(pwd); O=$R/gpurun_out/r06at; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "1 2" "0 2" "0 3" "0 5"; do set -- $cfg
    CN_SQ_PARTS=$2 python bench.py --stagger $1 --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger $1 parts $2 rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
