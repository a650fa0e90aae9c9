"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units) of tools/pmc_traffic.py into
profiles/<name>.json: calibration of FETCH_SIZE on the known-traffic cn_add launch, corrected HBM traffic of the NTT launches."""
import csv, json, sys, collections
fetch_csv, write_csv, out = sys.argv[1:4]
def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return rows
F, W = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
n, cts, k = 8192, 845, 5
ct_kib = 2 * k * n * 8 / 1024.0
add_read_kib, add_write_kib = 2 * cts * ct_kib, cts * ct_kib
res = {"units": "KiB per launch", "calibration_kernel": "k_addsub (2 x 845 ciphertext reads + 1 write, 8 B/lane coalesced)"}
fa = sum(F["k_addsub"]) / len(F["k_addsub"]); wa = sum(W["k_addsub"]) / len(W["k_addsub"])
res["k_addsub"] = {"FETCH_SIZE": fa, "WRITE_SIZE": wa, "true_read": add_read_kib, "true_write": add_write_kib,
                   "fetch_correction": add_read_kib / fa, "write_correction": add_write_kib / wa}
for name in F:
    if "k_ntt" in name:                                # one instantiation per direction: k_ntt_rr<L, AR, false|true>
        tag = "inverse" if name.rstrip(" >").endswith("true") else "forward"
        f_ = sum(F[name]) / len(F[name]); w_ = sum(W[name]) / len(W[name])
        corr = f_ * res["k_addsub"]["fetch_correction"] + w_ * res["k_addsub"]["write_correction"]
        res["%s %s" % (name, tag)] = {"FETCH_SIZE": f_, "WRITE_SIZE": w_, "hbm_traffic_corrected_KiB": corr,
                                      "hbm_traffic_corrected_bytes": corr * 1024, "algorithmic_bytes": cts * 10 * n * 16,
                                      "traffic_over_algorithmic": corr * 1024 / (cts * 10 * n * 16)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
