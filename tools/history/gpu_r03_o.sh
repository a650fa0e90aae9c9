#!/bin/bash
# visit O: why does the LoLa tool measure 7.7 ms per image inside bench.py and 10.4 ms on its own?  (a) plain, (b) HIP_FORCE_DEV_KERNARG=1,
# (c) torch imported and its HIP context initialised first (what bench.py does)
O=gpurun_out/r03o; mkdir -p $O
short() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: r=json.loads(l)
    except Exception: continue
    print("  %-60s %-62s %6.2f ms %s" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"], r.get("launches_per_prime","")))
PY
}
echo "(a) plain"; python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/a.txt 2>$O/a.err; short $O/a.txt
echo "(b) HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/b.txt 2>$O/b.err; short $O/b.txt
echo "(c) torch.cuda initialised first"; python - > $O/c.txt 2>$O/c.err <<'PY'
import sys, json, os
sys.path.insert(0, "tools")
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("env after torch:", {k: v for k, v in os.environ.items() if "HIP" in k or "HSA" in k or "AMD" in k or "ROC" in k}, file=sys.stderr)
import lola_unchanged_caller
for r in lola_unchanged_caller.measure("LoLa", 20):
    print(json.dumps(r))
PY
short $O/c.txt; tail -2 $O/c.err
echo "(d) bench lola eager, plain / KERNARG"; python bench.py --workload lola --no-unchanged-caller > $O/d1.json 2>/dev/null; HIP_FORCE_DEV_KERNARG=1 python bench.py --workload lola --no-unchanged-caller > $O/d2.json 2>/dev/null
python - <<'PY'
import json
for f in ("d1","d2"):
    d=json.load(open("gpurun_out/r03o/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["verified_against_integer_model"])
PY
