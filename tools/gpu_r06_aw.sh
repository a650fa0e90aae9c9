#!/bin/bash
# Round 6, visit AW: --stagger 2 (the primes' squaring layers one after the other, each pipelined in parts by cn_mul_relin) against --stagger 0 / 1
R=$(pwd); O=$R/gpurun_out/r06aw; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "1 3" "0 3" "2 3" "2 4" "2 5"; do set -- $cfg
    CN_SQ_PARTS=$2 python bench.py --stagger $1 --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stagger $1 parts $2 rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'])" | tee -a $O/ab.txt
  done
done
