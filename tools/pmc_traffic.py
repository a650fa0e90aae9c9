"""Workload for the HBM-traffic PMC passes: (1) cn_add over two 845-ciphertext arrays = a KNOWN byte count with the same
8 B/lane coalesced access width as the NTT kernel's loads (calibrates FETCH_SIZE, which under-reports on gfx950 -
MI355X_MICROARCH.md HBM section), (2) the forward NTT over 8450 limbs (the bench's roofline kernel), (3) the inverse."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context
g = Context(8192, 549764251649)
cts = 845
a, b, c = g.ct_alloc(cts), g.ct_alloc(cts), g.ct_alloc(cts)
rng = np.random.default_rng(1)
data = np.stack([np.concatenate([rng.integers(0, qq, size=8192, dtype=np.uint64) for _ in range(2) for qq in g.q]) for _ in range(5)])
for i in range(0, cts, 5):
    g.ct_upload(a, i, data); g.ct_upload(b, i, data[::-1].copy())
for _ in range(3):
    g.add(a, 0, b, 0, c, 0, cts)
g.sync()
ptr, _ = g.device_ptr(a)
g.ntt_time(ptr, cts * 10, 0, False, 3)
g.ntt_time(ptr, cts * 10, 0, True, 3)
