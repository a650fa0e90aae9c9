#!/bin/bash
# Round 6, visit G: parity of the shifted-range rotations / broadcast change; which memory-side counters exist on this box (DRAM vs fabric requests); N = 16384 key switch: L2-level vs DRAM-level reads
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_evaluator.py tests/test_gpu_multi_context.py -m gpu -x -q -k "n16384 or rotat or broadcast or multi_context or galois" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
(cd /tmp && rocprofv3 -L > $R/$O/counters.txt 2>&1)
grep -i -E "dram|mall|hbm|EA0_RDREQ|EA0_WRREQ|TCC_REQ|TCC_HIT|TCC_MISS" $O/counters.txt | cut -c1-200 | sort -u | head -60
P="python $R/tools/ks14_probe.py 5488 ks_pair14=1"
for C in "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $R/$O/p_$tag -- $P > $R/$O/run_$tag.txt 2>&1)
  F=$(find $O/p_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - <<PY
import csv, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open("$F")):
    if "pair14" in r["Kernel_Name"]:
        t = tot[r["Counter_Name"]]; t[0] += 1; t[1] += float(r["Counter_Value"])
for k, (n, v) in tot.items():
    print("%-28s per launch %.4g  (x64 B = %.2f GB, x32 B = %.2f GB; %d rows)" % (k, v / max(1, n) , v / max(1, n) * 64 / 1e9, v / max(1, n) * 32 / 1e9, n))
PY
  tail -2 $O/run_$tag.txt | cut -c1-200
  find $O/p_$tag -name "*.csv" -size +5M -delete
done
