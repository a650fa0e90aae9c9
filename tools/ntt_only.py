import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context
g = Context(8192, 549764251649)
cts = 845
h = g.ct_alloc(cts)
rng = np.random.default_rng(1)
data = np.stack([np.concatenate([rng.integers(0, qq, size=8192, dtype=np.uint64) for _ in range(2) for qq in g.q]) for _ in range(5)])
for i in range(0, cts, 5):
    g.ct_upload(h, i, data)
ptr, _ = g.device_ptr(h)
for inv in (False, True):
    g.ntt_time(ptr, cts * 10, 0, inv, 3)
