#!/bin/bash
# round 4, visit c2: where does the LoLa child of the default bench line hang?  (short parent, child limit 60 s, faulthandler stacks on timeout)
OUT=gpurun_out/r04c
mkdir -p $OUT
BENCH_CHILD_LIMIT=60 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-relinearize-late > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04c/bench.json").read().strip().splitlines()[-1])
l=d["lola"]
print({k:v for k,v in l.items() if k!="where"})
print(l.get("where"))
PY
