#!/bin/bash
# Round 5, visit S: large cn_rotate_rows_many batches in table-driven pieces (LoLa-CIFAR's ConvertToColumnVector)
O=gpurun_out/r05s; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py tests/test_lola.py tests/test_layers.py tests/test_basic_operations.py tests/test_deferred.py -m gpu -q -x -k "rotate or cifar or lola or Stack or Interleave or layers or deferred_rotations or random_programs" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python bench.py --workload cifar --steps 3 --warmup 2 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image'], d['verified_against_integer_model'])"
python bench.py --workload lola --steps 20 --warmup 3 --no-unchanged-caller > $O/lola.json 2> $O/lola.err; tail -1 $O/lola.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image']['min'], d['verified_against_integer_model'])"
python tools/cifar_latency.py 2>&1 | tail -4 | cut -c1-300
