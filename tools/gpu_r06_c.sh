#!/bin/bash
# Round 6, visit C: zero-encryption fold (k_encrypt_fold): the deferred suite, the whole GPU suite, then the literal unchanged caller with the fold on / off
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_deferred.py -m gpu -x -q > $O/pytest_deferred.txt 2>&1; tail -5 $O/pytest_deferred.txt
for fold in 1 0; do
  echo "== replay CN_FOLD_ZERO=$fold"
  CN_FOLD_ZERO=$fold python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 4,16,256 --steps 5 > $O/replay_fold$fold.txt 2> $O/replay_fold$fold.err
  python - <<PY
import json
for ln in open("$O/replay_fold$fold.txt"):
    d = json.loads(ln); print(d["caller"][:48], d["threads"], d["ms_per_batch"], d.get("frac_of_batched"), d.get("words_identical"), d.get("launches_per_batch"), (d.get("host") or {}).get("cpu_s_per_wall_s"))
PY
  tail -3 $O/replay_fold$fold.err
done
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.txt 2>&1; tail -5 $O/pytest_all.txt
