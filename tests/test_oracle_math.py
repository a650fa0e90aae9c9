"""The oracle checked against independent Python big-int models (no GPU): negacyclic NTT products vs schoolbook convolution,
encrypt/decrypt round trips, exact round(t*x/q) decryption, every evaluator op at the slot level, NAF rotations, the parameter
tables (prime rules of SURVEY 9.1/9.4)."""
import numpy as np
import pytest

from oracle.cno import COEFF_MODULUS_128, Oracle

CASES = [
    (64, 257, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 10, 20),
    (64, 257, [0xffffee001, 0xffffc4001], 16, 60),
    (128, 12289, COEFF_MODULUS_128[8192], 10, 20),
    (32, 193, COEFF_MODULUS_128[16384][:8], 60, 60),
]


def is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def test_default_coeff_modulus_follows_seal_rules():
    """DefaultParams.CoeffModulus128: NTT-friendly primes with the bit lengths of SURVEY 9.1"""
    bits = {2048: [54], 4096: [36, 36, 37], 8192: [43, 43, 44, 44, 44], 16384: [48, 48, 48, 49, 49, 49, 49, 49, 49]}
    for n, qs in COEFF_MODULUS_128.items():
        assert [q.bit_length() for q in qs] == bits[n]
        for q in qs:
            assert is_prime(q) and q % (2 * n) == 1


def test_behz_auxiliary_primes_follow_seal_rule():
    """m_sk, gamma, aux base = the largest 61-bit primes == 1 mod 2^18 in decreasing order"""
    o = Oracle(64, 257, q=CASES[0][2])
    found, x = [], (1 << 61) - (1 << 18) + 1
    while len(found) < 6:
        if is_prime(x):
            found.append(x)
        x -= 1 << 18
    bsk = o.bsk_moduli()
    assert bsk[-1] == found[0] and bsk[:3] == found[2:5]


@pytest.mark.parametrize("n,t,q,dbc,gdbc", CASES)
def test_oracle_against_bigint_models(n, t, q, dbc, gdbc):
    rng = np.random.default_rng(n + len(q))
    o = Oracle(n, t, q=q, dbc=dbc, gdbc=gdbc)
    o.keygen(7)
    for j in range(o.k):                                           # NTT = negacyclic convolution, minimal root
        a = rng.integers(0, q[j], n, dtype=np.uint64)
        b = rng.integers(0, q[j], n, dtype=np.uint64)
        fa, fb = o.ntt_fwd(j, a), o.ntt_fwd(j, b)
        c = o.ntt_inv(j, np.array([int(x) * int(y) % q[j] for x, y in zip(fa, fb)], dtype=np.uint64))
        ref = [0] * n
        for i in range(n):
            for l in range(n):
                v, idx = int(a[i]) * int(b[l]), i + l
                if idx >= n:
                    idx, v = idx - n, -v
                ref[idx] = (ref[idx] + v) % q[j]
        assert [int(x) for x in c] == ref
        psi = o.psi(j)
        assert pow(psi, n, q[j]) == q[j] - 1
        assert all(pow(psi, 2 * e + 1, q[j]) >= psi for e in range(n))          # minimal primitive 2n-th root
    v1 = rng.integers(0, t, n, dtype=np.uint64)
    v2 = rng.integers(0, t, n, dtype=np.uint64)
    p1, p2 = o.encode(v1), o.encode(v2)
    assert np.array_equal(o.decode(p1), v1)
    c1, c2 = o.encrypt(p1), o.encrypt(p2)
    dec = lambda c: o.decode(o.decrypt(c))
    mul = np.array([int(x) * int(y) % t for x, y in zip(v1, v2)], dtype=np.uint64)
    assert np.array_equal(dec(c1), v1)
    assert np.array_equal(dec(o.add(c1, c2)), (v1 + v2) % t)
    assert np.array_equal(dec(o.sub(c1, c2)), (v1 + t - v2) % t)
    assert np.array_equal(dec(o.negate(c1)), (t - v1) % t)
    assert np.array_equal(dec(o.add_plain(c1, p2)), (v1 + v2) % t)
    assert np.array_equal(dec(o.add_plain(c1, p2, True)), (v1 + t - v2) % t)
    assert np.array_equal(dec(o.multiply_plain(c1, p2)), mul)
    w = np.array([t - 5], dtype=np.uint64)                          # constant (sparse-format) plaintext, negative weight
    assert np.array_equal(dec(o.multiply_plain(c1, w)), np.array([int(x) * (t - 5) % t for x in v1], dtype=np.uint64))
    with pytest.raises(ValueError):
        o.multiply_plain(c1, np.zeros(n, dtype=np.uint64))          # SEAL: plain cannot be zero
    m3 = o.multiply(c1, c2)
    assert np.array_equal(dec(m3), mul)
    m2 = o.relinearize(m3)
    assert np.array_equal(dec(m2), mul)
    h = n // 2
    for s in [1, 2, 3, -1, -3, 5, h - 1, -(h - 1)]:                 # direct keys and NAF-decomposed steps
        assert np.array_equal(dec(o.rotate_rows(c1, s)), np.concatenate([np.roll(v1[:h], -s), np.roll(v1[h:], -s)])), s
    assert np.array_equal(dec(o.rotate_columns(c1)), np.concatenate([v1[h:], v1[:h]]))
    Q = 1
    for x in q:
        Q *= x
    x = o.dot_with_secret(m2).reshape(o.k, n)                       # decryption = exact round(t*x/q)
    d = o.decrypt(m2)
    for i in range(0, n, 5):
        X = 0
        for j in range(o.k):
            Qj = Q // q[j]
            X += int(x[j, i]) * Qj * pow(Qj, -1, q[j])
        X %= Q
        assert ((X * t * 2 + Q) // (2 * Q)) % t == int(d[i])


@pytest.mark.parametrize("name", ["tiny", "default4096", "c3", "tiny-xi", "default4096-xi", "c3-xi"])
def test_oracle_words_match_the_committed_digests(name):
    """Regression pin: SHA-256 of the oracle's words (keys, fresh encryption under a seeded stream, every evaluator op) for seeded inputs
    equal tests/golden/oracle_digests.json (written by tests/golden/make_oracle_digests.py).  A drift of the oracle - and with it of the
    words the HIP path is held to - shows up here on the CPU.  (Not SEAL known answers: ciphertext-word parity with SEAL 3.2 stays unpinned.)"""
    import importlib.util
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_digests", os.path.join(here, "make_oracle_digests.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "oracle_digests.json")))[name]
    got = mod.digests(name)
    assert got == want, sorted(k for k in want if got.get(k) != want[k])
    if name.endswith("-xi"):                 # the other key-switch convention changes the keys and what is computed WITH them - nothing else
        plain = json.load(open(os.path.join(here, "oracle_digests.json")))[name[:-3]]
        differ = {k for k in want if want[k] != plain[k]}
        assert differ == {"relin_key", "galois_key_0", "relinearize", "rotate_rows_3", "rotate_rows_-5", "rotate_columns"}, differ
