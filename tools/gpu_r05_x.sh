#!/bin/bash
# Round 5, visit X: CN_KS_WIDE_MAX 160 vs 64 vs 80, alternating, LoLa-MNIST and LoLa-CIFAR lines
O=gpurun_out/r05x; mkdir -p $O
for rep in 1 2 3; do for w in 160 64 80; do
  CN_KS_WIDE_MAX=$w python bench.py --workload lola --steps 30 --warmup 3 --no-unchanged-caller 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lola  CN_KS_WIDE_MAX=$w', d['ms_per_step'], d['ms_per_image']['min'], d['ms_per_image']['median'], d['verified_against_integer_model'])" | tee -a $O/sweep.txt
done; done
for w in 160 64; do
  CN_KS_WIDE_MAX=$w python bench.py --workload cifar --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cifar CN_KS_WIDE_MAX=$w', d['ms_per_step'], d['ms_per_image']['min'], d['verified_against_integer_model'])" | tee -a $O/sweep.txt
done
