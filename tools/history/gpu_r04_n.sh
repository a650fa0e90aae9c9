#!/bin/bash
# round 4, visit n: HBM traffic of the squaring chain on the final tree (k_square_pipe, extend, floor) and of the fused key switch: separate FETCH_SIZE / WRITE_SIZE passes
# over one serialised CryptoNets batch (gfx950: FETCH_SIZE x 2 = bytes fetched, calibrated on k_addsub in profiles/r04_ntt_hbm_traffic.json)
OUT=gpurun_out/r04n
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$OUT/$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > /dev/null 2> $R/$OUT/$c.err)
done
python - > $OUT/chain_traffic.txt <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/r04n/%s/**/*counter_collection.csv" % c, recursive=True):
        rows = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if r["Counter_Name"] == c and any(x in n for x in ("k_square_pipe", "k_square_fused", "k_behz_floor_f64", "k_behz_extend_f64", "k_keyswitch_rr", "k_scalar_gemm")):
                rows[(n.split("(")[0].replace("void ", "")[:46], r["Grid_Size"])].append(float(r["Counter_Value"]))
        for k, v in rows.items():
            res[k][c] = sum(v) / len(v)
print("%-48s %10s %14s %14s   (KiB per launch; FETCH_SIZE under-reports by 2 x on gfx950)" % ("kernel", "grid", "FETCH_SIZE", "WRITE_SIZE"))
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("WRITE_SIZE", 0)):
    print("%-48s %10s %14.0f %14.0f   fetched %.2f GiB, written %.2f GiB" % (k[0], k[1], v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), 2 * v.get("FETCH_SIZE", 0) / 2**20, v.get("WRITE_SIZE", 0) / 2**20))
PY
cat $OUT/chain_traffic.txt
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
