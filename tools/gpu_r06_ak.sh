#!/bin/bash
# Round 6, visit AK: kernel trace of the STAGGERED batched program in steady state (12 batches, last third): the interleaving the unchanged caller's flush pattern would have to reproduce
R=$(pwd); O=$R/gpurun_out/r06ak; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late > $O/bench.json 2> $O/bench.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$KT" > $O/sequence.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?"), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
rows.sort()
# the timed steps: the 12 launches of the 845-ciphertext key switch per queue
ks = [r for r in rows if r[2].startswith("k_keyswitch_rr") and r[4] > 1000000]
lo = ks[len(ks) // 2][0] - 8000000; hi = ks[-3][1]
base = None
for s, e, k, q, g in rows:
    if s < lo or s > hi: continue
    if base is None: base = s
    print("%10.1f us  +%8.1f  q%-3s %s" % ((s - base) / 1e3, (e - s) / 1e3, q, k))
PY
find $O/prof -name "*.csv" -delete
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
head -70 $O/sequence.txt
