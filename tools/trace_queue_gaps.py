"""Per hardware queue of a rocprofv3 --kernel-trace: busy / idle time inside the last `frac` of the trace, the gaps grouped by the pair (kernel before, kernel after) - where a launch chain waits for its host.

    python tools/trace_queue_gaps.py <kernel_trace.csv> [frac=0.5] [top=12]
"""
import csv, sys, collections
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5; top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:36], r.get("Queue_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - int((t1 - t0) * frac)
byq = collections.defaultdict(list)
for r in rows:
    if r[0] >= lo:
        byq[r[3]].append(r)
for q, rs in sorted(byq.items()):
    span = rs[-1][1] - rs[0][0]
    busy = sum(r[1] - r[0] for r in rs)
    pairs = collections.defaultdict(lambda: [0, 0])
    for a, b in zip(rs, rs[1:]):
        g = b[0] - a[1]
        if g > 0:
            p = pairs[(a[2], b[2])]; p[0] += 1; p[1] += g
    print("queue %s: %d kernels, span %.1f ms, busy %.1f ms (%.0f %%), idle %.1f ms" % (q, len(rs), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
    for (a, b), (n, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:top]:
        print("   %8.2f ms in %5d gaps (%7.1f us each)  after %-36s before %s" % (g / 1e6, n, g / 1e3 / n, a, b))
