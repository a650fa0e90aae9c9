#!/bin/bash
# Round-3 visit G: owner-biased context lock (CN_LOCK_GRACE_NS 0 vs 400) on the unchanged-caller table with launch counts; LoLa unchanged caller
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout 600 python -m pytest tests/test_call_trace.py tests/test_deferred.py -m gpu -x -q 2>&1 | grep -n "passed\|failed\|rror" | head
for G in 0 400 1000; do
  CN_LOCK_GRACE_NS=$G timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,16,256 --literal-threads 1,4,16,256 --steps 5 > $OUT/unchanged_caller_replay_grace$G.txt 2>&1
  echo "== grace $G"; python - <<PY
import json
for ln in open("$OUT/unchanged_caller_replay_grace$G.txt"):
    try: d = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print("%-28s thr %3d  %6.2f ms  %.3f  launches %s %s" % (d["caller"][:28], d["threads"], d["ms_per_batch"], d.get("frac_of_batched", 1.0), d.get("launches_per_batch"), d["words_identical"]))
PY
done
timeout 900 python tools/lola_unchanged_caller.py LoLa --reps 20 > $OUT/lola_unchanged_caller.txt 2>&1; cut -c1-260 $OUT/lola_unchanged_caller.txt | tail -10
