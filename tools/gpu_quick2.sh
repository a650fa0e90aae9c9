#!/bin/bash
python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py -q -x -m gpu -k "key_switch or rotat or c5 or cifar" 2>&1 | grep -E "passed|failed" | tail -1
python bench.py --workload cifar --steps 2 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cifar', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
python bench.py --steps 20 --warmup 3 --no-unchanged-caller --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['key_switch']['ms_per_launch'], d['key_switch'].get('frac_valu_in_situ'))"
