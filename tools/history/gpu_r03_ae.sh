#!/bin/bash
# visit AE: small tables read from the pinned ring (no copy dispatch): A/B
O=gpurun_out/r03ae; mkdir -p $O
python -m pytest tests/test_gpu_evaluator.py tests/test_deferred.py tests/test_lola.py -q -x -m gpu -k "rotat or copy_many or deferred or lola" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for z in 1 0 1 0; do
CN_TABLES_ZERO_COPY=$z python tools/chain_concurrency_probe.py LoLa 2>&1 | grep "contexts \[0\] \|contexts \[0, 1, 2, 3\]" | sed "s/^/zero_copy=$z /"
done
for z in 1 0; do CN_TABLES_ZERO_COPY=$z python bench.py --workload lola --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('zero_copy=$z bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"; done
