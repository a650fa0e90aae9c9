#!/bin/bash
# Round 6, visit AU: three parts of unequal size (CN_SQ_SPLIT: cut points in per mille), unstaggered batched loop
R=$(pwd); O=$R/gpurun_out/r06au; mkdir -p $O
for rep in 1 2; do
  for split in "333,667" "250,625" "200,600" "300,700" "400,750" "280,640"; do
    CN_SQ_PARTS=3 CN_SQ_SPLIT=$split python bench.py --stagger 0 --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('split $split rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
