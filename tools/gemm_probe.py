"""Timing probe of the three scalar GEMMs of a CryptoNets batch (N = 8192, 5 limbs, first plaintext prime): the planned convolution
(784 -> 845), dense 845 -> 100 and dense 100 -> 10 launches on random ciphertext words, HIP-event timed on the context stream.
Timing only (exactness: tests/test_gpu_evaluator.py::test_scalar_gemm*).  CNHIP_LIB selects a kernel build, BENCH_CONV_TILE the
gather-list tiling of the convolution.

    python tools/gemm_probe.py [reps=20]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context
from cryptonets_amd import cryptonets_mnist as cm

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tile = os.environ.get("BENCH_CONV_TILE", "1")
tile = tuple(int(x) for x in tile.split("x")) if "x" in tile else int(tile)
layers = cm.layer_tables(*cm.synthetic_weights(1), conv_tile=tile)
g = Context(cm.N, cm.PLAIN_PRIMES[0], dbc=10, gdbc=20)
if os.environ.get("PROBE_GEMM_MFMA"):
    g.set_option("gemm_mfma", int(os.environ["PROBE_GEMM_MFMA"]))      # 0: tiled convolutions (BENCH_CONV_TILE) stay on the VALU kernel
ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
rng = np.random.default_rng(3)
n, k = g.n, g.k
Q = [int(q) for q in g.coeff_modulus()] if hasattr(g, "coeff_modulus") else None
def fill(h, cnt):
    blk = rng.integers(0, 1 << 36, size=(64, 2 * k * n), dtype=np.uint64)        # below every q_j
    for i in range(0, cnt, 64):
        g.ct_upload(h, i, blk[:min(64, cnt - i)])
fill(ch.h_in, 784); fill(ch.h2, 845); fill(ch.h4, 100)
L = ch.layers if hasattr(ch, "layers") else ch.L
jobs = [("conv 784 -> 845", L[0]["plan"], ch.h_in, ch.h1, (784 + 845)), ("dense 845 -> 100", L[1]["plan"], ch.h2, ch.h3, (845 + 100)),
        ("dense 100 -> 10", L[2]["plan"], ch.h4, ch.h5, (100 + 10))]
ctb = 2 * k * n * 8
for rnd in range(2):
    for name, plan, src, dst, cts in jobs:
        g.gemm_apply(plan, src, dst, 0); g.sync()
        g.time_begin()
        for _ in range(reps):
            g.gemm_apply(plan, src, dst, 0)
        ms = g.time_end() / reps
        print("%-18s %8.1f us  %6.2f TB/s of algorithmic bytes (%.1f MiB)" % (name, ms * 1e3, cts * ctb / ms / 1e9, cts * ctb / 2**20), flush=True)
