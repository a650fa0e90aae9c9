#!/bin/bash
python -m pytest tests/test_gpu_evaluator.py tests/test_lola.py tests/test_deferred.py -q -x -m gpu -k "rotat or key_switch or lola or deferred or sum_slots" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
python bench.py --workload lola --steps 20 --warmup 2 > gpurun_out/bench_lola_q.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_lola_q.json')); u=d['unchanged_caller']; print('bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'], {k:u.get(k) for k in u if k not in ('all_rows','pattern')})"
