"""profiles/r05_ks14_counters.json from the rocprofv3 --pmc passes over tools/ks14_probe.py (tools/gpu_r05_f.sh): DYNAMIC instruction counts per wave and
HBM-level bytes per launch of k_keyswitch_pair14 at the LoLa-CIFAR geometry (5488 ciphertexts x 8 limbs = 43 904 workgroups of 8 waves).  bench.py
--workload cifar prices its key_switch block with these figures (a process cannot collect its own counters: rocprofv3 wraps a command).

    python tools/ks14_counters.py <dir with p*/...counter_collection.csv> out.json [ms per link of rounds 1-4]
"""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_keyswitch_pair14" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
waves = avg.get("SQ_WAVES")
res = {"kernel": "k_keyswitch_pair14<ArF64T<1>, false>", "geometry": "5488 ciphertexts x 8 output limbs, 512 threads per workgroup", "launches_averaged": {k: len(v) for k, v in acc.items()},
       "counters_per_launch": avg, "waves_per_launch": waves}
f64 = [k for k in avg if k.startswith("SQ_INSTS_VALU_") and k.endswith("_F64")]
if waves and f64:
    res["fp64_per_wave"] = sum(avg[k] for k in f64) / waves
    res["fp64_counters"] = sorted(f64)
if waves and "SQ_INSTS_VALU" in avg:
    res["valu_per_wave"] = avg["SQ_INSTS_VALU"] / waves
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    # KiB units; FETCH_SIZE x 2 on gfx950 (calibrated on k_addsub every round: profiles/r0N_ntt_hbm_traffic.json)
    res["hbm_bytes_per_launch"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024
    res["fetch_bytes_per_launch"] = 2.0 * avg["FETCH_SIZE"] * 1024
    res["write_bytes_per_launch"] = avg["WRITE_SIZE"] * 1024
if len(sys.argv) > 3:
    res["rounds_1_to_4_ms_per_link"] = float(sys.argv[3])
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
