#!/bin/bash
# Round 5, visit L: one ciphertext x many plaintexts in one launch (k_mul_plain_bcast) - parity, CIFAR / LoLa lines; unchanged caller with the cgroup accounting
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py tests/test_lola.py tests/test_gpu_multi_context.py -m gpu -q -x -k "multiply_plain or rowdot or cifar or lola or convention" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --workload cifar --steps 3 --warmup 2 > $O/cifar.json 2> $O/cifar.err; tail -1 $O/cifar.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image'], d['verified_against_integer_model'])"
python bench.py --workload lola --steps 20 --warmup 3 > $O/lola.json 2> $O/lola.err; tail -1 $O/lola.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_image']['min'], d['verified_against_integer_model'], (d.get('unchanged_caller') or {}).get('ms_per_image'), (d.get('unchanged_caller') or {}).get('frac_of_batched'))"
python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16,64,256 --steps 8 2>/dev/null | cut -c1-420 | tee $O/replay.txt
