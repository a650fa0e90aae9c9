#!/bin/bash
# Round 6, visit AY: the 100-ciphertext squaring layer pipelined as well (CN_SQ_MIN=64) - plain loop, three alternating rounds
R=$(pwd); O=$R/gpurun_out/r06ay; mkdir -p $O
for rep in 1 2 3; do
  for mn in 512 64; do
    CN_SQ_MIN=$mn python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late --no-unchanged-caller 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('min $mn rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['config']['program'][:40])" | tee -a $O/ab.txt
  done
done
