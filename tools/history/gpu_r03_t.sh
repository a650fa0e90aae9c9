#!/bin/bash
# visit T2: cn_copy_many + direct rotations in the mirror: tests, LoLa latencies
O=gpurun_out/r03t; mkdir -p $O
python -m pytest tests/test_gpu_evaluator.py -q -x -m gpu -k "copy_many" 2>&1 | tail -3
python -m pytest tests/test_deferred.py tests/test_lola.py tests/test_call_trace.py tests/test_layers.py tests/test_basic_operations.py -q -x -m gpu 2>&1 | tail -3
python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/lola_unchanged_caller.txt 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r03t/lola_unchanged_caller.txt"):
    r=json.loads(l); print("  %-60s %-62s %6.2f ms %s %s" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"], r.get("launches_per_prime",""), r["logits_exact"]))
PY
python tools/lola_latency.py LoLa --graph 2>/dev/null | tail -4
python bench.py --workload lola --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
