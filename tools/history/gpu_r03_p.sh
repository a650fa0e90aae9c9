#!/bin/bash
# visit P: stream -> hardware-queue mapping.  HIP deals streams round-robin onto GPU_MAX_HW_QUEUES (default 4) queues; 4 plaintext-prime
# streams + the null stream = two chains on one queue?
O=gpurun_out/r03p; mkdir -p $O
for q in 2 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --workload lola --no-unchanged-caller > $O/lola_q$q.json 2>/dev/null
done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 3 --no-unchanged-caller > $O/cn_q$q.json 2>/dev/null
done
GPU_MAX_HW_QUEUES=8 python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/tool_q8.txt 2>/dev/null
python - <<'PY'
import json
for f in ("lola_q2","lola_q4","lola_q8","cn_q4","cn_q8"):
    d=json.load(open("gpurun_out/r03p/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["verified_against_integer_model"], (d.get("relinearize_late") or {}).get("ms_per_step"))
for l in open("gpurun_out/r03p/tool_q8.txt"):
    r=json.loads(l); print("  %-60s %-62s %6.2f ms" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"]))
PY
