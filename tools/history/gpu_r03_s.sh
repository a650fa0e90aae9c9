#!/bin/bash
# visit S: what the deferred LoLa replay launches per level (CN_DEFER_TRACE)
O=gpurun_out/r03s; mkdir -p $O
CN_DEFER_TRACE=1 python tools/lola_unchanged_caller.py LoLa --reps 1 > $O/rows.txt 2> $O/trace.txt
python - <<'PY'
import re, collections
lines=[l for l in open("gpurun_out/r03s/trace.txt") if l.startswith("defer ")]
ctxs=collections.Counter(l.split()[1] for l in lines)
print(len(lines), "trace lines;", ctxs.most_common(8))
# last context created = most lines near the end; take the last 400 lines of one context and print the tail (one deferred inference)
c=lines[-1].split()[1]
mine=[l for l in lines if l.split()[1]==c]
# find the last inference: print the last 120 lines
tot=0
for l in mine[-130:]:
    print(l.rstrip().replace("defer "+c+" ",""))
PY
