#!/usr/bin/env python
"""NTT micro-benchmark grid of SURVEY 8(d): B x k limbs of N u64 residues, uniform in [0, q_j) from numpy.random.default_rng(20250925);
B in {1, 64, 1690 (= 845 x 2 polys), 8192}; k = 2 (C2) and k = 5 (C3) at N = 8192, and N = 16384 / k = 8 (C5).  Forward and inverse
batched transform (`k_ntt_rr`), HIP events on the context stream (cn_ntt_time, 20 launches after 3 warm-up launches).
Algorithmic bytes = 2 * N * 8 per limb (read once + written once); peak 8 TB/s.  Prints a table and one JSON line per cell."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cryptonets_amd._native import Context  # noqa: E402
from cryptonets_amd import _native  # noqa: E402

C5_Q = [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001]
CASES = [("C2", 8192, 549764251649, [0x7fffffd8001, 0x7fffffc8001]), ("C3", 8192, 549764251649, None), ("C5", 16384, 957181001729, C5_Q)]


def main():
    rng = np.random.default_rng(20250925)
    rows = []
    print("%-4s %6s %2s %6s %9s | %10s %9s %6s | %10s %9s %6s" % ("cfg", "N", "k", "B", "limbs", "fwd us", "GB/s", "frac", "inv us", "GB/s", "frac"))
    for name, n, t, q in CASES:
        g = Context(n, t, q=q, dbc=60 if n == 16384 else 10, gdbc=60 if n == 16384 else 20, device=0)
        k = g.k
        for B in (1, 64, 1690, 8192):
            limbs = B * k
            cts = -(-limbs // (2 * k))
            h = g.ct_alloc(cts)
            block = np.stack([np.concatenate([rng.integers(0, qq, size=n, dtype=np.uint64) for _ in range(2) for qq in g.q]) for _ in range(min(cts, 8))])
            for i in range(0, cts, len(block)):
                g.ct_upload(h, i, block[:min(len(block), cts - i)])
            ptr, _ = g.device_ptr(h)
            cell = dict(config=name, n=n, k=k, B=B, limbs=limbs, bytes=limbs * n * 16)
            for inverse in (False, True):
                g.ntt_time(ptr, limbs, 0, inverse, 3)
                ms = g.ntt_time(ptr, limbs, 0, inverse, 20)
                gbs = limbs * n * 16 / (ms * 1e-3) / 1e9
                cell["inv" if inverse else "fwd"] = dict(us=round(ms * 1e3, 2), GBps=round(gbs, 1), frac_of_8TBps=round(gbs / 8000.0, 4))
            g.free(h)
            rows.append(cell)
            f, i_ = cell["fwd"], cell["inv"]
            print("%-4s %6d %2d %6d %9d | %10.2f %9.1f %6.3f | %10.2f %9.1f %6.3f" % (name, n, k, B, limbs, f["us"], f["GBps"], f["frac_of_8TBps"], i_["us"], i_["GBps"], i_["frac_of_8TBps"]))
        g.close()
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
