"""Squaring chain of 845 ciphertexts (Evaluator.Multiply(a, a): k_behz_extend, k_square_pipe q / Bsk, k_behz_floor) with the q-side kernel on a second stream
("sq_overlap") against the serial chain: HIP-event time per chain, alone on the device.  Under rocprofv3 --kernel-trace, tools/square_overlap_timeline.py prints
where every kernel of one chain ran."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context
g = Context(8192, 549764251649)
rng = np.random.default_rng(1)
cnt = 845
h2, h3 = g.ct_alloc(cnt), g.ct_alloc(cnt, 3)
one = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(2) for q in g.q])
blk = np.stack([np.roll(one, i) % np.concatenate([np.full(g.n, q, dtype=np.uint64) for _ in range(2) for q in g.q]) for i in range(13)])
for i in range(0, cnt, 13):
    g.ct_upload(h2, i, blk[: min(13, cnt - i)])
for ov in (0, 1, 0, 1):
    g.set_option("sq_overlap", ov)
    g.multiply(h2, 0, h2, 0, h3, 0, cnt); g.sync()
    g.time_begin()
    for _ in range(5):
        g.multiply(h2, 0, h2, 0, h3, 0, cnt)
    print("sq_overlap=%d: %.3f ms per 845-ciphertext squaring chain" % (ov, g.time_end() / 5), flush=True)
