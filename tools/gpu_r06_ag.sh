#!/bin/bash
# Round 6, visit AG: per-queue idle time of the literal unchanged caller in steady state (24 batches traced, last half analysed): which layer boundary does a stream wait for its host at?
R=$(pwd); O=$R/gpurun_out/r06ag${AG_TAG}; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/tools/replay_reference_calls.py --trained --threads ${AG_THREADS:-1} --literal-threads ${AG_LIT:-16} --steps 24 > $O/run.txt 2> $O/run.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/trace_queue_gaps.py $KT 0.45 14 > $O/queue_gaps.txt 2>&1
python tools/trace_gaps.py $KT 0.45 12 > $O/gaps.txt 2>&1
python - "$KT" > $O/sequence.txt <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?")))
rows.sort()
t1 = max(r[1] for r in rows); t0 = rows[0][0]
lo = t1 - int((t1 - t0) * 0.12)
base = None
for s, e, k, q in rows:
    if s < lo: continue
    if base is None: base = s
    print("%10.1f us  +%8.1f  q%-3s %s" % ((s - base) / 1e3, (e - s) / 1e3, q, k))
PY
find $O/prof -name "*.csv" -delete
cat $O/queue_gaps.txt | cut -c1-200; tail -3 $O/run.txt | cut -c1-300
