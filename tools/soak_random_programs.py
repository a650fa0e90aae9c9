"""Soak of the deferred queue's hazard logic: the random single-thread programs of tests/test_deferred.py (`_random_program`: every deferrable call kind over a
pool of single-ciphertext handles, reads and writes colliding at random, handles released and re-allocated while calls are pending) for many more seeds than
the test runs - queued ("defer" = 1 and 2) against launched one by one, word for word.

    python tools/soak_random_programs.py --seconds 300 [--seed 100] [--params tiny]
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import get_gpu, get_oracle
import test_deferred as td


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=100)
    ap.add_argument("--params", default="tiny")
    ap.add_argument("--length", type=int, default=300)
    a = ap.parse_args()
    o, g = get_oracle(a.params, galois=True), get_gpu(a.params, galois=True)
    rng = np.random.default_rng(a.seed)
    cts = td._fresh(o, rng, 7)
    pts = np.stack([o.encode(rng.integers(1, 5, size=o.n, dtype=np.uint64)) for _ in range(3)])
    t0, seed, pend = time.time(), a.seed, 0
    live0 = g.live_handles()
    while time.time() - t0 < a.seconds:
        length = int(np.random.default_rng(seed).integers(20, a.length + 1))
        now, _ = td._random_program(g, o, cts, pts, seed, defer=False, length=length)
        for mode in (1, 2):
            later, pending = td._random_program(g, o, cts, pts, seed, defer=mode, length=length)
            pend = max(pend, pending)
            for i, (x, y) in enumerate(zip(now, later)):
                if not np.array_equal(x, y):
                    raise AssertionError("seed %d length %d defer %d: ciphertext %d differs" % (seed, length, mode, i))
        g.sync()
        if g.live_handles() != live0:
            raise AssertionError("seed %d: handle count %d -> %d" % (seed, live0, g.live_handles()))
        seed += 1
    print("soak ok: %s, seeds %d..%d (%d programs x 3 modes), %.0f s, deepest queue %d calls" % (a.params, a.seed, seed - 1, seed - a.seed, time.time() - t0, pend))


if __name__ == "__main__":
    main()
