#!/usr/bin/env python
"""Would ONE launch chain for the SumAllSlots steps of all four plaintext-prime channels beat four chains side by side?  (a) four contexts, 13
ciphertexts each, `cn_sum_slots(length 1024)` issued to all four and awaited together - what a LoLa dense layer does now; (b) one context with
52 ciphertexts in one call (the key-switch work of a merged launch; CN_KS_WIDE_MAX must admit 52 x k blocks to the two-launch kernels).

    CN_KS_WIDE_MAX=400 python tools/sumslots_merge_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cryptonets_amd._native import Context


def main(rows=13, primes=4, length=1024, reps=30):
    ts = [557057, 638977, 737281, 786433][:primes]
    ctxs = []
    for t in ts:
        g = Context(8192, t, dbc=10, gdbc=20, device=0)
        g.keygen(1234 ^ t, galois=True)
        ctxs.append(g)
    rng = np.random.default_rng(3)

    def fill(g, n):
        ph, h = g.pt_alloc(n), g.ct_alloc(n)
        g.encode_batch(rng.integers(0, g.t, size=(n, g.n), dtype=np.uint64), ph, 0)
        g.encrypt(ph, 0, h, 0, n, seed=5)
        g.free(ph)
        return h

    hs = [fill(g, rows) for g in ctxs]
    big = fill(ctxs[0], rows * primes)

    def sync():
        for g in ctxs:
            g.sync()

    def timed(fn):
        for _ in range(3):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            sync()
        return 1e3 * (time.perf_counter() - t0) / reps

    one = timed(lambda: ctxs[0].sum_slots(hs[0], 0, rows, length))
    four = timed(lambda: [g.sum_slots(h, 0, rows, length) for g, h in zip(ctxs, hs)])
    merged = timed(lambda: ctxs[0].sum_slots(big, 0, rows * primes, length))
    print("SumAllSlots(%d) of %d ciphertexts on one context: %.3f ms | on %d contexts at once: %.3f ms | %d ciphertexts in one call: %.3f ms (ks_wide max %s)" % (
        length, rows, one, primes, four, rows * primes, merged, os.environ.get("CN_KS_WIDE_MAX", "160")))


if __name__ == "__main__":
    main()
