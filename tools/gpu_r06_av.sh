#!/bin/bash
# Round 6, visit AV: staggered loop against the plain loop with cn_mul_relin in three parts (280 / 640 per mille), three alternating rounds
R=$(pwd); O=$R/gpurun_out/r06av; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "1 2 500" "0 3 280,640" "0 3 300,700"; do set -- $cfg
    CN_SQ_PARTS=$2 CN_SQ_SPLIT=$3 python bench.py --stagger $1 --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger $1 parts $2 split $3 rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
