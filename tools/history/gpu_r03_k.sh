#!/bin/bash
# Round-3 visit K: combining context lock (CnMutex::run): tests, then the unchanged-caller thread sweep with CN_LOCK_COMBINE = 1 / 0, twice each
OUT=gpurun_out/r03k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_deferred.py tests/test_gpu_evaluator.py tests/test_call_trace.py tests/test_gpu_multi_context.py -m gpu -x -q 2>&1 | grep -n "passed\|failed\|rror\|assert" | head
for rep in 1 2; do for C in 1 0; do
  CN_LOCK_COMBINE=$C timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,16,64,256 --literal-threads 1,4,16,64,256 --steps 5 > $OUT/replay_combine${C}_$rep.txt 2>&1
  echo "== combine $C rep $rep"; python - <<PY
import json
for ln in open("$OUT/replay_combine${C}_$rep.txt"):
    try: d = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print("%-40s thr %3d  %6.2f ms  %.3f  %s" % (d["caller"][:40], d["threads"], d["ms_per_batch"], d.get("frac_of_batched", 1.0), d["words_identical"]))
PY
done; done
