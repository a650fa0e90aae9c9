"""Timeline of the LAST squaring chains in a rocprofv3 kernel trace of tools/square_overlap_probe.py: start / end of every kernel relative to the chain's first kernel, per queue."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?"), r.get("Grid_Size", "")))
rows.sort()
rows = [r for r in rows if any(k in r[2] for k in ("behz", "square"))]
# chains = groups of 4 kernels: extend, pipe q, pipe Bsk, floor; print the last 3 of each mode (the probe runs 1 + 5 chains per setting, settings 0 1 0 1)
chains = [rows[i:i + 4] for i in range(0, len(rows) - 3, 4)]
for label, idx in (("serial (third setting, last chain)", 17), ("overlapped (fourth setting, last chain)", 23), ("overlapped (fourth setting, chain before)", 22)):
    if idx >= len(chains):
        continue
    c = chains[idx]; t0 = min(r[0] for r in c)
    print(label)
    for r in sorted(c):
        print("   q%-3s %-40s grid %-9s start %8.1f us  end %8.1f us  (%.1f us)" % (r[3], r[2], r[4], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3))
    print("   chain: %.1f us" % ((max(r[1] for r in c) - t0) / 1e3))
