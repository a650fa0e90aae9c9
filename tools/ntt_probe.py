"""Quick NTT kernel timing probe (HIP events on the context stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context

CONFIGS = ((8192, 5, 845), (8192, 2, 845), (16384, 8, 200), (4096, 3, 845))
for n, k, cts in (CONFIGS[:1] if os.environ.get("NTT_PROBE_ONLY") else CONFIGS):
    q = None
    g = Context(n, 549764251649 if n >= 8192 else 40961, q=None if k in (3, 5) else ([0x7fffffd8001, 0x7fffffc8001] if k == 2 else
                [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001]))
    h = g.ct_alloc(cts)
    rng = np.random.default_rng(1)
    data = np.stack([np.concatenate([rng.integers(0, qq, size=n, dtype=np.uint64) for _ in range(2) for qq in g.q]) for _ in range(8)])
    for i in range(0, cts, 8):
        g.ct_upload(h, i, data[: min(8, cts - i)])
    ptr, nbytes = g.device_ptr(h)
    limbs = cts * 2 * g.k
    for inv in (False, True):
        g.ntt_time(ptr, limbs, 0, inv, 2)
        ms = g.ntt_time(ptr, limbs, 0, inv, 10)
        gb = limbs * n * 16 / 1e9
        print("N=%d k=%d limbs=%d %s: %.3f ms/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (n, k, limbs, "inv" if inv else "fwd", ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80))
    g.free(h); g.close()
