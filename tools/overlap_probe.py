#!/usr/bin/env python
"""Can the HBM-bound kernels of one plaintext-prime channel hide under the FP64-bound key switch of the other?  Two contexts (two streams) on
one MI355X: A runs the 845-ciphertext relinearisation (k_keyswitch_rr: 1 workgroup per CU, 2 x 216 VGPRs per SIMD, 132 KiB LDS - leaves
80 VGPRs and 6 wave slots per SIMD free), B runs one of the other kernels of the batch.  Wall time alone, together, and the sum."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cryptonets_amd import cryptonets_mnist as cm           # noqa: E402
from cryptonets_amd._native import Context                  # noqa: E402


def main():
    layers = cm.layer_tables(*cm.reference_weights())
    ctxs = []
    for p in cm.PLAIN_PRIMES:
        g = Context(cm.N, p, dbc=10, gdbc=20, device=0)
        g.keygen(7 ^ p, galois=False)
        ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
        ph = g.pt_alloc(784)
        g.encode_batch(np.zeros((784, 8), dtype=np.uint64) + 3, ph, 0)
        g.encrypt(ph, 0, ch.h_in, 0, 784, seed=5)
        g.free(ph)
        ch.forward()
        ch.front()
        g.sync()
        ctxs.append(ch)
    A, B = ctxs

    def timed(fa, fb, reps=4):
        for ch in ctxs:
            ch.g.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            if fa:
                fa()
            if fb:
                fb()
        for ch in ctxs:
            ch.g.sync()
        return (time.perf_counter() - t0) * 1e3 / reps

    ks = lambda: A.g.relinearize(A.t3, 0, A.h2, 0, 845)
    others = {
        "conv GEMM (784 -> 845)": lambda: B.g.gemm_apply(B.layers[0]["plan"], B.h_in, B.h1, 0),
        "dense GEMM on the matrix cores (845 -> 100)": lambda: B.g.gemm_apply(B.layers[1]["plan"], B.h2, B.h3, 0),
        "add (845 cts, pure streaming)": lambda: B.g.add(B.h1, 0, B.h2, 0, B.h2, 0, 845),
        "BEHZ multiply (extend, fused squaring x2, floor)": lambda: B.g.multiply(B.h1, 0, B.h1, 0, B.t3, 0, 845),
        "key switch (the twin)": lambda: B.g.relinearize(B.t3, 0, B.h2, 0, 845),
    }
    timed(ks, None, 2)
    t_ks = timed(ks, None)
    print("key switch of 845 ciphertexts alone: %.3f ms" % t_ks)
    for name, fb in others.items():
        timed(None, fb, 2)
        # B's kernel is short: repeat it so that it spans the key switch
        t_b = timed(None, fb)
        rep = max(1, int(round(t_ks / t_b))) if t_b < t_ks else 1
        fbr = lambda fb=fb, rep=rep: [fb() for _ in range(rep)]
        t_b = timed(None, fbr)
        t_ab = timed(ks, fbr)
        print("%-50s x%-2d alone %.3f ms | together %.3f ms | sum %.3f | hidden %.0f %% of the shorter" % (
            name, rep, t_b, t_ab, t_ks + t_b, 100.0 * (t_ks + t_b - t_ab) / min(t_ks, t_b)))


if __name__ == "__main__":
    main()
