#!/bin/bash
# visit N: size-3 scalar GEMMs + the relinearize-late program variant
O=gpurun_out/r03n; mkdir -p $O
python -m pytest tests/test_gpu_evaluator.py -q -x -m gpu -k "unrelinearized or scalar_gemm" 2>&1 | tail -4
python -m pytest tests/test_cryptonets_mnist.py -q -x -m gpu 2>&1 | tail -4
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03n/bench.json'))
print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'])
print(json.dumps(d['relinearize_late']))
print(json.dumps(d['unchanged_caller'].get('at_visible_cpu_count')), d['unchanged_caller']['frac_of_batched'])
PY
tail -3 $O/bench.err
