#!/bin/bash
# visit Q: does the ORDER in which a process creates its streams decide the LoLa latency?  k dummy streams created before the contexts
O=gpurun_out/r03q; mkdir -p $O
for k in 0 1 2 3 4 5; do
python - $k > $O/k$k.txt 2>/dev/null <<'PY'
import sys, json, ctypes
sys.path.insert(0, "tools")
k = int(sys.argv[1])
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
keep = []
for i in range(k):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    keep.append(s)
import lola_unchanged_caller
for r in lola_unchanged_caller.measure("LoLa", 10):
    print(json.dumps(r))
PY
python - $k <<'PY'
import json, sys
rows=[json.loads(l) for l in open("gpurun_out/r03q/k%s.txt" % sys.argv[1])]
print("dummy streams %s:" % sys.argv[1], " | ".join("%s %.2f" % (("bat" if r["pattern"].startswith("batched") else ("def" if r["pattern"].endswith("submission") else "imm")) + ("/py" if r["host"].startswith("python") else "/c1" if "one host" in r["host"] else "/cj" if "joined" in r["host"] else "/cf"), r["ms_per_image"]) for r in rows))
PY
done
