#!/bin/bash
# visit R: queue-aware stream selection (pick_stream): LoLa latency for k = 0..3 dummy streams created first, with and without the probe
O=gpurun_out/r03r; mkdir -p $O
run() { # $1 = k, $2 = probe
CN_STREAM_PROBE=$2 python - $1 > $O/k$1_p$2.txt 2>/dev/null <<'PY'
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
python - $1 $2 <<'PY'
import json, sys
rows=[json.loads(l) for l in open("gpurun_out/r03r/k%s_p%s.txt" % (sys.argv[1], sys.argv[2]))]
print("dummy streams %s probe %s:" % (sys.argv[1], sys.argv[2]), " | ".join("%s %.2f" % (("bat" if r["pattern"].startswith("batched") else ("def" if r["pattern"].endswith("submission") else "imm")) + ("/py" if r["host"].startswith("python") else "/c1" if "one host" in r["host"] else "/cj" if "joined" in r["host"] else "/cf"), r["ms_per_image"]) for r in rows))
PY
}
for k in 0 1 2 3; do run $k 1; done
run 0 0
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
from cryptonets_amd._native import Context
from cryptonets_amd import cryptonets_mnist as cm
t0 = time.perf_counter()
ctxs = []
for i in range(6):
    t = time.perf_counter()
    g = Context(4096, 65537, dbc=10, gdbc=20, device=0)
    ctxs.append(g)
    print("context %d: %.1f ms, stream_tries %d" % (i, 1e3 * (time.perf_counter() - t), g.get_option("stream_tries")))
PY
python bench.py --workload lola --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
python bench.py --steps 20 --warmup 3 --no-unchanged-caller 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['relinearize_late']['ms_per_step'])"
