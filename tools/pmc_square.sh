#!/bin/bash
# HBM traffic of the fused squaring kernel (are the parked NTT-form operands served from L2?): separate FETCH_SIZE / WRITE_SIZE passes over
# one serialised CryptoNets batch, per-launch averages of k_square_fused next to its algorithmic bytes.
OUT=gpurun_out/pmc_square
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 100 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$OUT/$c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --serialize > /dev/null 2> $R/$OUT/$c.err)
done
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/pmc_square/%s/**/*counter_collection.csv" % c, recursive=True):
        rows = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("k_square_fused" in r["Kernel_Name"] or "k_behz_floor_f64" in r["Kernel_Name"] or "k_addsub" in r["Kernel_Name"]):
                rows[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
        for k, v in rows.items():
            res[k][c] = sum(v) / len(v)
for k, v in sorted(res.items()):
    print(k, {c: round(x) for c, x in v.items()}, "KiB per launch")
PY
find $OUT -name "*kernel_trace.csv" -delete
