#!/bin/bash
# Round 6, visit BA: scheme 2 of the pipelined cn_mul_relin (all key switches on the context's stream, the later Multiplies on the second stream), parts 2-5, against scheme 1 with three parts
R=$(pwd); O=$R/gpurun_out/r06ba; mkdir -p $O
CN_SQ_SCHEME=2 CN_SQ_PARTS=4 timeout 600 python -m pytest tests/test_gpu_evaluator.py -q -k "two_pipelined_halves" 2>&1 | tail -2 | tee $O/test.txt
for rep in 1 2; do
  for cfg in "1 3" "2 2" "2 3" "2 4" "2 5" "2 6"; do set -- $cfg
    CN_SQ_SCHEME=$1 CN_SQ_PARTS=$2 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('scheme $1 parts $2 rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
