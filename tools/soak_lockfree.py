"""Soak of the per-ciphertext caller on the deferred queue (`cn_set_option("defer", 1 | 2)`): random small CryptoNets-shaped networks (a 1-d convolution
with padded border taps, SquareActivation, a dense layer) on the "tiny" ring, driven by tools/replay_reference_calls.cpp from 1 ... 200 caller threads on
one or two contexts at once, with every switch of the path drawn at random (lock-free ring or locked queue, literal zero encryptions folded or
materialised, one-call zero vectors, releases at once or parked).  Every run is checked three ways: the decrypted slots against the integer model, the
ciphertext WORDS against the batched entry points of the same context (runs without fresh zero encryptions), the handle count against its value before
the run (nothing leaked, nothing released twice).  Prints the seed of every configuration: a failure is reproduced with --seed S --iters 1.

    python tools/soak_lockfree.py --seconds 300 [--seed 1]
"""
import argparse, faulthandler, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_reference_calls as rp
from cryptonets_amd._native import Context

P = dict(n=1024, t=12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001], dbc=10, gdbc=20)          # tests/conftest.py "tiny"
SLOTS = 8


def network(rng, n_in, t):
    K, stride, maps = int(rng.integers(2, 8)), int(rng.integers(1, 4)), int(rng.integers(1, 5))
    corners = list(range(0, n_in - 1, stride))
    maps = max(1, min(maps, 64 // len(corners)))                       # fan-in of the dense layer as in tests/test_deferred.py: the tiny ring's noise budget
    idx0 = np.array([[c + k if c + k < n_in else -1 for k in range(K)] for _ in range(maps) for c in corners], dtype=np.int32)
    W0 = rng.integers(-20, 21, size=idx0.shape).astype(np.int64)
    W0[rng.random(W0.shape) < 0.05] = 0
    W0[~((W0 != 0) & (idx0 >= 0)).any(axis=1), 0] = 7                  # (an output without a non-zero term is an error, as in the reference: AddMany of nothing)
    O0, O1 = idx0.shape[0], int(rng.integers(1, 7))
    idx1 = np.tile(np.arange(O0, dtype=np.int32), (O1, 1))
    W1 = rng.integers(-30, 31, size=(O1, O0)).astype(np.int64)
    W1[~(W1 != 0).any(axis=1), 0] = 5
    return [dict(idx=idx0, W=np.mod(W0, t).astype(np.uint64), bias_idx=(np.arange(O0) % 2).astype(np.int32), square=True),
            dict(idx=idx1, W=np.mod(W1, t).astype(np.uint64), bias_idx=np.zeros(O1, dtype=np.int32), square=False)]


def model(layers, x, t, bias=(3, 11)):
    cols = x.astype(np.int64)                                          # [n_in, SLOTS]
    for L in layers:
        out = np.zeros((L["idx"].shape[0], cols.shape[1]), dtype=np.int64)
        for o in range(L["idx"].shape[0]):
            acc = np.full(cols.shape[1], bias[int(L["bias_idx"][o])], dtype=np.int64)
            for kk, c in enumerate(L["idx"][o]):
                if c >= 0:
                    acc = (acc + int(L["W"][o, kk]) * cols[int(c)]) % t
            out[o] = acc
        cols = out * out % t if L["square"] else out
    return cols.astype(np.uint64)


class Chan:
    def __init__(self, seed):
        self.g = g = Context(P["n"], P["t"], q=P["q"], dbc=P["dbc"], gdbc=P["gdbc"], device=0)
        g.keygen(seed, galois=False)
        self.bias = g.pt_alloc(2)
        g.encode_batch(np.array([[3] * P["n"], [11] * P["n"]], dtype=np.uint64), self.bias, 0)

    def load(self, x, seed):
        g = self.g
        n_in = x.shape[0]
        ph, self.hin = g.pt_alloc(n_in), g.ct_alloc(n_in)
        g.encode_batch(x, ph, 0)
        g.encrypt(ph, 0, self.hin, 0, n_in, seed=seed)
        g.free(ph)
        self.ins = rp.split_columns(g, self.hin, n_in)

    def batched_words(self, layers):
        g = self.g
        O0, O1 = layers[0]["idx"].shape[0], layers[1]["idx"].shape[0]
        h1, h2, h3 = g.ct_alloc(O0), g.ct_alloc(O0), g.ct_alloc(O1)
        g.scalar_gemm(self.hin, layers[0]["W"], h1, 0, idx=layers[0]["idx"], bias_pt=self.bias, bias_idx=layers[0]["bias_idx"])
        g.mul_relin(h1, 0, h1, 0, h2, 0, O0)
        g.scalar_gemm(h2, layers[1]["W"], h3, 0, idx=layers[1]["idx"], bias_pt=self.bias, bias_idx=layers[1]["bias_idx"])
        w = g.ct_download(h3, 0, O1)
        for h in (h1, h2, h3):
            g.free(h)
        return w

    def unload(self):
        for h in list(self.ins) + [self.hin]:
            self.g.free(int(h))

    def slots(self, handles):
        g = self.g
        dh = g.pt_alloc(len(handles))
        for i, h in enumerate(handles):
            g.decrypt(int(h), 0, 1, dh, i)
        s = g.decode_batch(dh, 0, len(handles))[:, :SLOTS]
        g.free(dh)
        return s


def one(chans, seed, verbose):
    rng = np.random.default_rng(seed)
    n_in = int(rng.integers(5, 49))
    layers = network(rng, n_in, P["t"])
    nctx = int(rng.integers(1, len(chans) + 1))
    use = chans[:nctx]
    xs = [rng.integers(0, 12, size=(n_in, SLOTS), dtype=np.uint64) for _ in use]
    xfull = [np.zeros((n_in, P["n"]), dtype=np.uint64) for _ in use]
    for xf, x in zip(xfull, xs):
        xf[:, :SLOTS] = x
    live0 = [c.g.live_handles() for c in use]
    for c, xf in zip(use, xfull):
        c.load(xf, int(rng.integers(1, 1 << 30)))
    want = [model(layers, x, P["t"]) for x in xs]
    ref = [c.batched_words(layers) for c in use]
    net = rp.Replay([c.g for c in use], [dict(idx=L["idx"], W=[L["W"]] * nctx, bias_pt=[c.bias for c in use], bias_idx=L["bias_idx"], square=L["square"]) for L in layers])
    ins = np.stack([c.ins for c in use])
    runs = int(rng.integers(1, 5))
    cfgs = []
    for r in range(runs):
        defer = int(rng.choice([2, 2, 2, 1, 0]))
        cfg = dict(defer=defer, threads=int(rng.choice([1, 2, 3, 6, 16, 64, 200])), literal=bool(rng.integers(0, 2)), merged=bool(rng.integers(0, 2)),
                   direct_free=bool(rng.integers(0, 2)), fold=int(rng.integers(0, 2)), keep=bool(rng.integers(0, 2)))
        cfgs.append(cfg)
        if verbose > 1:
            print("  seed %d run %d: %r" % (seed, r, cfg), flush=True)
        for c in use:
            c.g.set_option("fold_zero", cfg["fold"])
            c.g.set_option("defer", defer)
        try:
            out = net.run(ins, cfg["threads"], literal_taps=cfg["literal"], nonce0=int(rng.integers(1, 1 << 40)), merged=cfg["merged"], direct_free=cfg["direct_free"])
            if not cfg["keep"]:                                   # half of the runs read their results while the queue mode is still on (the download synchronises)
                for c in use:
                    c.g.set_option("defer", 0)
            for p, c in enumerate(use):
                got = np.stack([c.g.ct_download(int(h), 0, 1)[0] for h in out[p]])
                s = c.slots(out[p])
                for h in out[p]:
                    c.g.free(int(h))
                if not np.array_equal(s, want[p]):
                    raise AssertionError("slots differ from the integer model: seed %d run %d ctx %d %r" % (seed, r, p, cfg))
                has_pad = bool((layers[0]["idx"] < 0).any())
                if not (cfg["literal"] and has_pad) and not np.array_equal(got, ref[p]):
                    raise AssertionError("words differ from the batched entry points: seed %d run %d ctx %d %r" % (seed, r, p, cfg))
        finally:
            for c in use:
                c.g.set_option("defer", 0)
    for c in use:
        c.unload()
        c.g.sync()
    live1 = [c.g.live_handles() for c in use]
    if live1 != live0:
        raise AssertionError("handle count %r -> %r: seed %d %r" % (live0, live1, seed, cfgs))
    if verbose:
        print("seed %d ok: n_in %d, outputs %d / %d, contexts %d, %s" % (seed, n_in, layers[0]["idx"].shape[0], layers[1]["idx"].shape[0], nctx,
              " | ".join("defer %(defer)d thr %(threads)d lit %(literal)d mrg %(merged)d dfree %(direct_free)d fold %(fold)d" % c for c in cfgs)), flush=True)
    return runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--trace", action="store_true", help="print every run's switches before it starts")
    ap.add_argument("--same-seed", action="store_true", help="repeat --seed instead of counting up (a race that needs many attempts)")
    a = ap.parse_args()
    faulthandler.enable()
    chans = [Chan(42), Chan(43)]
    t0, n, runs, seed = time.time(), 0, 0, a.seed
    while (a.iters and n < a.iters) or (not a.iters and time.time() - t0 < a.seconds):
        runs += one(chans, seed, 0 if a.quiet else 2 if a.trace else 1)
        seed += 0 if a.same_seed else 1
        n += 1
    folded = sum(c.g.get_option("folded_zero_encryptions") for c in chans)
    print("soak ok: %d networks, %d runs, %.0f s, %d zero encryptions folded, seeds %d..%d" % (n, runs, time.time() - t0, folded, a.seed, seed - 1))


if __name__ == "__main__":
    main()
