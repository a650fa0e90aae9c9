"""GPU idle time inside a rocprofv3 --kernel-trace: the union of the kernel intervals over the last `frac` of the trace, the largest gaps in it with the
kernels on either side, and per queue the busy time.

    python tools/trace_gaps.py <kernel_trace.csv> [frac=0.5] [top=25]
"""
import csv, sys, collections
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5; top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Queue_Id", "?"), r.get("Grid_Size", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - int((t1 - t0) * frac)
sel = [r for r in rows if r[0] >= lo]
span = sel[-1][1] - sel[0][0]
busy, end, gaps, last = 0, sel[0][0], [], sel[0]
for r in sel:
    if r[0] > end:
        gaps.append((r[0] - end, last, r)); busy += 0
        cur_start = r[0]
    busy += max(0, r[1] - max(end, r[0]))
    if r[1] > end:
        end = r[1]; last = r
print("window %.2f ms, %d kernels, GPU busy (union) %.2f ms = %.1f %%, idle %.2f ms in %d gaps" % (span / 1e6, len(sel), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
perq = collections.Counter()
for r in sel:
    perq[r[3]] += r[1] - r[0]
print("busy per queue (ms):", {q: round(v / 1e6, 2) for q, v in perq.items()})
for g, a, b in sorted(gaps, reverse=True)[:top]:
    print("%8.1f us idle after %-48s (q%s, %s) before %-48s (q%s, %s) at +%.2f ms" % (g / 1e3, a[2], a[3], a[4], b[2], b[3], b[4], (b[0] - sel[0][0]) / 1e6))

# per-kernel totals inside the window
tot = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = (r[2], r[4]); tot[k][0] += 1; tot[k][1] += r[1] - r[0]
print("%-50s %10s %6s %10s %10s" % ("kernel", "grid", "calls", "total_ms", "avg_us"))
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-50s %10s %6d %10.2f %10.1f" % (k[0], k[1], c, t / 1e6, t / 1e3 / c))
