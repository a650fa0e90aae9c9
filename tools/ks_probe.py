"""A/B probe of the key-switch kernel variants (HIP-event timed on the ctx stream): relinearise 845 size-3 ciphertexts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context
g = Context(8192, 549764251649)
rng = np.random.default_rng(1)
kw = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(g.key_words() // g.ctw) for _ in range(2) for q in g.q])
g.set_relin_key(kw)
cnt = 845
h3, h2 = g.ct_alloc(cnt, 3), g.ct_alloc(cnt)
one = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(3) for q in g.q])
for i in range(cnt):
    g.ct_upload(h3, i, one[None, :])
for variant in ("ks_tight=0", "ks_tight=1", "ks_tight=0", "ks_tight=1"):
    name, val = variant.split("=")
    g.set_option(name, int(val))
    g.relinearize(h3, 0, h2, 0, cnt); g.sync()
    g.time_begin()
    for _ in range(3):
        g.relinearize(h3, 0, h2, 0, cnt)
    ms = g.time_end() / 3
    print("%s: %.3f ms per 845-ct relinearize (%.1f ns per limb-NTT equivalent)" % (variant, ms, ms * 1e6 / (cnt * 5 * 27)))
