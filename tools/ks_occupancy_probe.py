#!/usr/bin/env python
"""Does the fused key switch run closer to its VALU-issue floor when the two waves a SIMD holds belong to DIFFERENT workgroups?
N = 8192: one 512-thread workgroup per CU (image + LDS twiddles = 130 KiB): the two waves of a SIMD meet at the same barriers.
N = 4096: 256-thread workgroups (one wave per SIMD), two per CU (65 KiB each): the two waves of a SIMD are independent.
For both: relinearisation of a batch that fills the chip, HIP-event timed, against the floor from the ISA counts of the built kernel priced at
the issue rates measured in the same process (cn_valu_issue_time)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cryptonets_amd._native import Context  # noqa: E402
import ks_isa_counts  # noqa: E402

for n, t, logn, cnt in ((8192, 549764251649, 13, 845), (4096, 40961, 12, 1690)):
    g = Context(n, t)
    rng = np.random.default_rng(1)
    kw = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(g.key_words() // g.ctw) for _ in range(2) for q in g.q])
    g.set_relin_key(kw)
    h3, h2 = g.ct_alloc(cnt, 3), g.ct_alloc(cnt)
    one = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(3) for q in g.q])
    blk = np.repeat(one[None, :], 65, axis=0)
    for i in range(0, cnt, 65):
        g.ct_upload(h3, i, blk[:min(65, cnt - i)])
    g.relinearize(h3, 0, h2, 0, cnt); g.sync()
    g.time_begin()
    for _ in range(5):
        g.relinearize(h3, 0, h2, 0, cnt)
    ms = g.time_end() / 5
    f64ns, v32ns = g.fp64_issue_ns(iters=4096, launches=6), g.valu32_issue_ns(iters=16384, launches=6)
    per_limb = [-(-int(q).bit_length() // 10) for q in g.q]
    bits = max(int(q).bit_length() for q in g.q)
    pol = "0" if bits <= 44 else "1"
    obj = os.path.join(ROOT, "cryptonets_amd", "lib", "obj", "cn_l_ks_f64l.o" if pol == "0" else "cn_l_ks_f64.o")
    kern = "_Z14k_keyswitch_rrILi%dE6ArF64TILi%sEELi1ELb1ELb0EE" % (logn, pol)
    fp, isa = ks_isa_counts.fp64_per_thread(g.k, per_limb, obj, kern)
    vo = ks_isa_counts.valu_per_thread(per_limb, isa)
    waves = (n // 16) // 64
    floor = cnt * g.k * waves * (fp * f64ns + vo * v32ns) / 1024 * 1e-6
    print("N=%5d k=%d digits=%2d cts=%4d: %.3f ms | fp64/thread %d, other VALU %d | fp64 %.2f ns, valu32 %.2f ns | VALU-issue floor %.3f ms | frac %.3f"
          % (n, g.k, sum(per_limb), cnt, ms, fp, vo, f64ns, v32ns, floor, floor / ms))
    g.free(h3); g.free(h2); g.close()
