#!/bin/bash
# Round 5, visit R: lazy accumulators of k_keyswitch_pair14 recentred once per half instead of every other digit
O=gpurun_out/r05r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py tests/test_deferred.py -m gpu -q -x -k "n16384 or key_switch or fused_rotate_and_add or c5_shapes or cifar or extreme" > $O/pytest_ks.txt 2>&1; tail -2 $O/pytest_ks.txt
timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1 2>&1 | tee $O/ks14_probe.txt
python bench.py --workload cifar --steps 3 --warmup 2 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image'], d['verified_against_integer_model'])"
