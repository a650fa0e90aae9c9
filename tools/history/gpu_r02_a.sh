#!/bin/bash
# Round-2 visit A: the refactored library + deferred submission: GPU test-suite, bench line (with unchanged_caller), the replay tool at
# several thread counts, the key-switch micro-benchmark with the software-pipelined variant.
OUT=gpurun_out/r02a
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-260 $OUT/bench.json; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r02a/bench.json"))
    print("unchanged:", json.dumps(d.get("unchanged_caller"))[:400])
    print("key_switch:", json.dumps(d.get("key_switch"))[:400])
    print("cpu:", json.dumps(d.get("cpu_baseline"))[:300])
except Exception as e:
    print("bench parse:", e)
PY
timeout 600 python tools/replay_reference_calls.py --threads 1,4,16,64 --steps 5 --immediate --trained > $OUT/replay.txt 2>&1
tail -8 $OUT/replay.txt | cut -c1-250
timeout 300 ./tools/ubench_ks > $OUT/ubench_ks.txt 2>&1
head -20 $OUT/ubench_ks.txt
