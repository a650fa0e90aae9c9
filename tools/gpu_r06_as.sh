#!/bin/bash
# Round 6, visit AS: cn_mul_relin pipelined in P parts (CN_SQ_PARTS) - the unstaggered batched loop against the staggered one
R=$(pwd); O=$R/gpurun_out/r06as; mkdir -p $O
CN_SQ_PARTS=4 timeout 600 python -m pytest tests/test_gpu_evaluator.py -q -k "two_pipelined_halves" 2>&1 | tail -2 | tee $O/test.txt
CN_SQ_PARTS=3 timeout 600 python -m pytest tests/test_gpu_evaluator.py -q -k "two_pipelined_halves" 2>&1 | tail -2 | tee -a $O/test.txt
for rep in 1 2; do
  for cfg in "1 2" "0 2" "0 3" "0 4" "0 6"; do set -- $cfg
    CN_SQ_PARTS=$2 python bench.py --stagger $1 --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger $1 parts $2 rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
