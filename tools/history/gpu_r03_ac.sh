#!/bin/bash
# visit AC: cn_rotate_rows_many
O=gpurun_out/r03ac; mkdir -p $O
python -m pytest tests/test_gpu_evaluator.py -q -x -m gpu -k "rotate_rows_many or rotation" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -6
python -m pytest tests/test_deferred.py tests/test_lola.py tests/test_call_trace.py tests/test_layers.py tests/test_basic_operations.py -q -x -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -3
python tools/chain_concurrency_probe.py LoLa 2>&1 | grep "calls of one\|contexts \[0\] \|contexts \[0, 1, 2, 3\]\|contexts \[0, 1\]"
python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/lola_unchanged_caller.txt 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r03ac/lola_unchanged_caller.txt"):
    r=json.loads(l); print("  %-60s %-62s %6.2f ms %s %s" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"], r.get("launches_per_prime",""), r["logits_exact"]))
PY
python bench.py --workload lola --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
