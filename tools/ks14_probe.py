"""A/B probe of the N = 16384 batch key switch (BASELINE config 5: k = 8, one 60-bit digit per limb): `cnt` ciphertexts through the
SumAllSlots chain (14 rotate-and-add links) and through one in-place rotate-and-add, HIP-event timed on the context stream, for the
kernel variants of round 5.  Random words and random key words (timing only; exactness is tests/test_gpu_evaluator.py's job).

    python tools/ks14_probe.py [cnt=5488] [variants...]      variant = name=value[,name=value...] of cn_set_option
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context

args = sys.argv[1:]
cnt = int(args[0]) if args and args[0].isdigit() else 5488
variants = [a for a in args if "=" in a] or ["ks_pair14=0", "ks_pair14=1,ks_chain=0,ks_xcd=0", "ks_pair14=1,ks_chain=1,ks_xcd=0", "ks_pair14=1,ks_chain=1,ks_xcd=1"]
Q = [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001]
g = Context(16384, 957181001729, q=Q, dbc=60, gdbc=60)
rng = np.random.default_rng(1)
n, k = g.n, g.k
kw = np.concatenate([rng.integers(0, q, size=n, dtype=np.uint64) for _ in range(g.key_words(1) // (2 * k * n)) for _ in range(2) for q in Q])
elts = [2 * n - 1] + [g.galois_elt_from_step(-(1 << i)) for i in range(13)]
for e in elts:
    g.set_galois_key(e, kw)
h = g.ct_alloc(cnt)
one = np.concatenate([rng.integers(0, q, size=n, dtype=np.uint64) for _ in range(2) for q in Q])
blk = np.repeat(one[None, :], 64, axis=0)
def fill():
    for i in range(0, cnt, 64):
        g.ct_upload(h, i, blk[:min(64, cnt - i)])
defaults = dict(ks_pair14=1, ks_chain=1, ks_xcd=1)
print("N = 16384, k = 8, %d ciphertexts (%.1f GiB)" % (cnt, cnt * 2 * k * n * 8 / 2**30))
for var in variants * 2:
    opts = dict(defaults); opts.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in var.split(",")})
    for name, v in opts.items():
        g.set_option(name, v)
    fill()
    g.rotate_rows_add(h, 0, -1, h, 0, h, 0, cnt); g.sync()          # warm-up (arenas)
    g.time_begin()
    for _ in range(2):
        g.rotate_rows_add(h, 0, -1, h, 0, h, 0, cnt)
    one_ms = g.time_end() / 2
    g.sum_slots(h, 0, cnt, 0); g.sync()
    g.time_begin()
    g.sum_slots(h, 0, cnt, 0)
    chain_ms = g.time_end()
    print("%-52s rotate-and-add %7.2f ms | SumAllSlots (14 links) %8.2f ms = %6.2f ms per link" % (var, one_ms, chain_ms, chain_ms / 14), flush=True)
