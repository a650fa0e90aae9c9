"""Regenerates tests/golden/oracle_digests.json: SHA-256 digests of the oracle's words for seeded keys / inputs at three parameter sets.
They pin the oracle (and, through the parity suite, the HIP path) against drift between rounds - a refactoring that changes a single
ciphertext word changes a digest.  They are NOT known answers of SEAL 3.2 (no SEAL binary exists here: "parity unpinned" stays).

    python tests/golden/make_oracle_digests.py            # rewrites the JSON next to this file
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.cno import COEFF_MODULUS_128, Oracle  # noqa: E402

SETS = {
    "tiny": dict(n=1024, t=12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001], dbc=10, gdbc=20),
    "default4096": dict(n=4096, t=40961, q=None, dbc=10, gdbc=20),
    "c3": dict(n=8192, t=549764251649, q=None, dbc=10, gdbc=20),
}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def digests(name):
    """name or name + "-xi": the same parameter set with the other key-switch decomposition convention (oracle ks_xi, round 4) - keys and every
    key-switching operation then have other words, everything else must keep the digests of the plain name"""
    xi = name.endswith("-xi")
    p = SETS[name[:-3] if xi else name]
    o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], ks_xi=xi)
    o.keygen(20250926, galois=True)
    o.seed(7)
    r = np.random.default_rng(5)
    va, vb = r.integers(0, o.t, size=o.n, dtype=np.uint64), r.integers(0, o.t, size=o.n, dtype=np.uint64)
    pa, pb = o.encode(va), o.encode(vb)
    a, b = o.encrypt(pa), o.encrypt(pb)
    m3 = o.multiply(a, b)
    out = {
        "relin_key": digest(o.relin_key()), "galois_key_0": digest(o.galois_key(0)), "public_key": digest(o.public_key()),
        "encode": digest(pa), "encrypt": digest(a), "add": digest(o.add(a, b)), "sub": digest(o.sub(a, b)),
        "add_plain": digest(o.add_plain(a, pb)), "multiply_plain": digest(o.multiply_plain(a, pb)),
        "multiply": digest(m3), "square": digest(o.multiply(a, a)), "relinearize": digest(o.relinearize(m3)),
        "rotate_rows_3": digest(o.rotate_rows(a, 3)), "rotate_rows_-5": digest(o.rotate_rows(a, -5)), "rotate_columns": digest(o.rotate_columns(a)),
        "decrypt": digest(o.decrypt(o.relinearize(m3))),
        "scalar_gemm": digest(o.scalar_gemm(np.stack([a, b]), np.array([[3, o.t - 2], [1, 0]], dtype=np.uint64))),
    }
    return out


if __name__ == "__main__":
    res = {name: digests(name) for base in SETS for name in (base, base + "-xi")}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_digests.json")
    json.dump(res, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path)
