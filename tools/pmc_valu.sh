#!/bin/bash
# VALU utilisation per kernel of one CryptoNets batch (counters only: --pmc with --kernel-trace, no other trace domains).
OUT=gpurun_out/pmc_valu
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $R/$OUT/p1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/run1.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -f csv -d $R/$OUT/p2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/run2.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -f csv -d $R/$OUT/p3 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/run3.txt 2>&1)
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    f = glob.glob("gpurun_out/pmc_valu/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f:
        print(p, "no counter file"); continue
    f.sort(key=lambda x: -__import__("os").path.getsize(x))          # the batch process, not a helper's
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Grid_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    names = sorted({c for v in acc.values() for c in v})
    with open("gpurun_out/pmc_valu/%s_summary.txt" % p, "w") as o:
        o.write("%-46s %10s " % ("kernel", "grid") + " ".join("%16s" % n for n in names) + "\n")
        rows = sorted(acc.items(), key=lambda kv: -max(kv[1].values()))[:24]
        for k, v in rows:
            o.write("%-46s %10s " % k + " ".join("%16.4g" % (v[n] / max(1, cnt[(k, n)])) for n in names) + "\n")
    print(open("gpurun_out/pmc_valu/%s_summary.txt" % p).read())
PY
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
