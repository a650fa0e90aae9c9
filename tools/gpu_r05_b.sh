#!/bin/bash
# Round 5, visit B: where does k_keyswitch_pair14 spend its time?  Timing-experiment builds (tools/build_ks14_dbg.py: pieces switched off), then the counter passes.
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_evaluator.py -m gpu -q -x -k "n16384" > $O/pytest_ks.txt 2>&1; tail -3 $O/pytest_ks.txt
for m in "" _dbg7 _dbg15 _dbg16 _dbg32 _dbg48 _dbg63; do
  echo "== libcnhip$m.so"
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$m.so timeout 300 python tools/ks14_probe.py 5488 ks_pair14=1,ks_chain=1 ks_pair14=1,ks_chain=1,ks_xcd=1 2>&1 | grep -v "^N =" | tee -a $O/ks14_dbg.txt
done
P="python $R/tools/ks14_probe.py 5488 ks_pair14=1,ks_chain=1"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $R/$O/p1 -- $P > $R/$O/run1.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -f csv -d $R/$O/p2 -- $P > $R/$O/run2.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -f csv -d $R/$O/p3 -- $P > $R/$O/run3.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/$O/p4 -- $P > $R/$O/run4.txt 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $R/$O/p5 -- $P > $R/$O/run5.txt 2>&1)
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3", "p4", "p5"):
    f = glob.glob("gpurun_out/r05b/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Grid_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    names = sorted({c for v in acc.values() for c in v})
    with open("gpurun_out/r05b/%s_summary.txt" % p, "w") as o:
        o.write("%-46s %10s " % ("kernel", "grid") + " ".join("%16s" % n for n in names) + "\n")
        rows = sorted(acc.items(), key=lambda kv: -max(kv[1].values()))[:8]
        for k, v in rows:
            o.write("%-46s %10s " % k + " ".join("%16.4g" % (v[n] / max(1, cnt[(k, n)])) for n in names) + "\n")
    print(open("gpurun_out/r05b/%s_summary.txt" % p).read())
PY
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
