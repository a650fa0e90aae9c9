#!/bin/bash
# round 4, visit k: host time of the queue flushes of the unchanged LoLa caller (CN_DEFER_TRACE=2) - where do its 1.6 ms over the batched form go?
OUT=gpurun_out/r04k
mkdir -p $OUT
CN_DEFER_TRACE=2 python tools/lola_unchanged_caller.py LoLa --reps 3 2> $OUT/flush_times.txt > $OUT/rows.txt
python - <<'PY'
import json, re, collections
for l in open("gpurun_out/r04k/rows.txt"):
    r = json.loads(l); print("  %-60s %-62s %6.2f ms %s" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"], r.get("launches_per_prime", "")))
t = [(int(m.group(2)), float(m.group(3))) for m in (re.search(r"defer (\S+) flush of (\d+) calls: (\d+) us", l) for l in open("gpurun_out/r04k/flush_times.txt")) if m]
print(len(t), "flushes; total host us", sum(x[1] for x in t))
c = collections.Counter((n) for n, _ in t)
by = collections.defaultdict(list)
for n, us in t: by[n].append(us)
for n in sorted(by): print("  flush of %4d calls: x%d, mean %.0f us" % (n, len(by[n]), sum(by[n]) / len(by[n])))
PY
