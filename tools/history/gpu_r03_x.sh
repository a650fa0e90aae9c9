#!/bin/bash
# visit X: where does a LoLa chain step spend its ~35 us?  kernel durations vs gaps between consecutive kernels of one hardware queue
O=gpurun_out/r03x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o lola -- python $GRAFT_REPO_ROOT/bench.py --workload lola --steps 6 --warmup 2 --no-unchanged-caller > $GRAFT_REPO_ROOT/$O/bench_lola_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); echo $f; head -2 $f
python - $f <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernels; columns:", list(rows[0].keys()))
qk=[k for k in rows[0] if "queue" in k.lower()][0]
byq=collections.defaultdict(list)
for r in rows: byq[r[qk]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in byq.items():
    ks.sort()
    # steady state: the last 60 % of the kernels
    ks=ks[int(len(ks)*0.4):]
    dur=[(e-s)/1e3 for s,e,_ in ks]
    gaps=[(ks[i+1][0]-ks[i][1])/1e3 for i in range(len(ks)-1)]
    small=[g for g in gaps if g < 200]
    import statistics as st
    print("queue", q, "kernels", len(ks), "mean dur %.1f us, median %.1f | gaps < 200 us: n %d mean %.1f median %.1f p90 %.1f | span %.1f ms, busy %.1f ms" % (st.mean(dur), st.median(dur), len(small), st.mean(small), st.median(small), sorted(small)[int(0.9*len(small))], (ks[-1][1]-ks[0][0])/1e6, sum(dur)/1e3))
PY
