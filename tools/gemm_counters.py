"""Counters of the scalar-GEMM kernels from rocprofv3 --pmc passes over tools/gemm_probe.py (tools/gpu_r05_gemm_pmc.sh): per kernel the average per launch and
per wave (instructions issued, cycles issuing / stalled / parked, L2-level bytes).

    python tools/gemm_counters.py <dir with p*/...counter_collection.csv> out.json
"""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_scalar_gemm" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"launches_averaged": max(len(v) for v in cs.values()), "counters_per_launch": avg}
    w = avg.get("SQ_WAVES")
    if w:
        d["per_wave"] = {c: avg[c] / w for c in avg if c.startswith("SQ_") and c != "SQ_WAVES"}
    if "FETCH_SIZE" in avg:
        d["fetch_bytes_per_launch"] = 2.0 * avg["FETCH_SIZE"] * 1024        # KiB units, x 2 on gfx950 (profiles/r0N_ntt_hbm_traffic.json)
    if "WRITE_SIZE" in avg:
        d["write_bytes_per_launch"] = avg["WRITE_SIZE"] * 1024
    res[k] = d
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
