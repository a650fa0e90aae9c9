#!/bin/bash
# Round 6, visit B: lock-free submission ("defer" = 2): parity of the deferred suite, then the unchanged CryptoNets caller through the ring against the
# locked queue (defer = 1) at 4 / 16 / 256 caller threads, with and without the literal padded taps
O=gpurun_out/r06b; mkdir -p $O
timeout 1200 python -m pytest tests/test_deferred.py -m gpu -x -q > $O/pytest_deferred.txt 2>&1; tail -3 $O/pytest_deferred.txt
for mode in "" "--locked"; do
  echo "== replay $mode"
  python tools/replay_reference_calls.py --trained --threads 4,16,256 --literal-threads 4,16,256 --steps 5 $mode > $O/replay$mode.txt 2> $O/replay$mode.err
  python - <<PY
import json
for ln in open("$O/replay$mode.txt"):
    d = json.loads(ln); print(d["caller"][:48], d["threads"], d["ms_per_batch"], d.get("frac_of_batched"), d.get("words_identical"), d.get("launches_per_batch"), (d.get("host") or {}).get("cpu_s_per_wall_s"))
PY
  tail -3 $O/replay$mode.err
done
