#!/bin/bash
# HBM-side traffic of the fused key switch under the three workgroup orders (CN_KS_XCD = 0 / 1 / 2): separate FETCH_SIZE / WRITE_SIZE passes over one
# serialised CryptoNets batch (gfx950: FETCH_SIZE under-reports by 2 x, profiles/r01_ntt_hbm_traffic.json), plus HIP-event times of the same launch
OUT=gpurun_out/pmc_ks
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for m in 0 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && CN_KS_XCD=$m timeout 120 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$OUT/m$m/$c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unchanged-caller --serialize --stagger 0 > /dev/null 2> $R/$OUT/m$m_$c.err)
  done
done
python - <<'PY'
import csv, glob, collections
for m in (0, 1, 2):
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/pmc_ks/m%d/%s/**/*counter_collection.csv" % (m, c), recursive=True):
            rows = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "k_keyswitch_rr" in r["Kernel_Name"]:
                    rows[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
            for k, v in rows.items():
                res[k][c] = sum(v) / len(v)
    for k, v in sorted(res.items()):
        print("CN_KS_XCD=%d" % m, k, {c: round(x) for c, x in v.items()}, "KiB per launch (FETCH x 2 = bytes fetched)")
PY
find $OUT -name "*kernel_trace.csv" -delete
for m in 0 1 2; do
  CN_KS_XCD=$m timeout 300 python bench.py --no-cpu-baseline --no-unchanged-caller --steps 10 --warmup 2 > $OUT/bench_m$m.json 2>/dev/null
  python -c "import json; d=json.load(open('$OUT/bench_m$m.json')); print('CN_KS_XCD=$m bench', d['value'], d['ms_per_step'], 'key switch', d['key_switch']['ms_per_launch'], d['verified_against_integer_model'])"
done
