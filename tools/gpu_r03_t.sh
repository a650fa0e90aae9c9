#!/bin/bash
# visit T: staged deferral without gather / scatter for equally spaced operands; stream probe; LoLa unchanged caller again
O=gpurun_out/r03t; mkdir -p $O
python -m pytest tests/test_deferred.py tests/test_lola.py tests/test_call_trace.py -q -x -m gpu 2>&1 | tail -3
python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/lola_unchanged_caller.txt 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r03t/lola_unchanged_caller.txt"):
    r=json.loads(l); print("  %-60s %-62s %6.2f ms %s %s" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"], r.get("launches_per_prime",""), r["logits_exact"]))
PY
