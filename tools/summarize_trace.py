"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / avg / total duration."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("kernel_name")
        name = name.split("(")[0]
        grid = r.get("Grid_Size_X") or r.get("grid_size_x") or r.get("Grid_Size") or "?"
        wg = r.get("Workgroup_Size_X") or r.get("workgroup_size_x") or "?"
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows[(name, grid, wg)].append(dur)
tot = sum(sum(v) for v in rows.values())
print("%-46s %10s %6s %7s %12s %12s %7s" % ("kernel", "grid_x", "wg_x", "calls", "avg_us", "total_us", "pct"))
for (name, grid, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("%-46s %10s %6s %7d %12.1f %12.1f %6.1f%%" % (name[:46], grid, wg, len(v), sum(v) / len(v), sum(v), 100 * sum(v) / tot))
