#!/bin/bash
# Round 5, visit W: LoLa-MNIST (four concurrent prime chains): up to how many (ciphertext, limb) blocks should a key switch run as two launches?
O=gpurun_out/r05w; mkdir -p $O
for w in 160 64 49 24 0; do
  CN_KS_WIDE_MAX=$w python bench.py --workload lola --steps 20 --warmup 3 --no-unchanged-caller 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CN_KS_WIDE_MAX=$w', d['ms_per_step'], d['ms_per_image']['min'], d['verified_against_integer_model'])" | tee -a $O/sweep.txt
done
