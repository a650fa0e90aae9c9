"""Client-side operations on the device: cn_keygen / cn_encrypt / cn_decrypt.

Fresh encryption is randomised (and SEAL's RNG is not reproducible), so the checks are:
  * cross-implementation round trips: oracle decrypts what the GPU encrypted and vice versa, under device-generated keys;
  * key STRUCTURE: every component of the device-made public / relinearisation / Galois keys is (-(a*s + e) + f*s', a) with a
    small error polynomial |e| <= 19 (clipped normal, sigma 3.2) - checked with the oracle's transforms;
  * distribution sanity of the Philox-driven samplers;
  * the reference's known-answer tests run with the device client (no oracle in the loop) in test_basic_operations-style.
"""
import numpy as np
import pytest

from conftest import PARAMS, get_gpu, get_oracle

pytestmark = pytest.mark.gpu


def make(name, seed=77, galois=True):
    from cryptonets_amd._native import Context
    from oracle.cno import Oracle
    p = PARAMS[name]
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    g.keygen(seed, galois=galois)
    o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"])
    o.import_keys(g.get_key(3), g.get_key(2))
    return g, o


def centered(x, q):
    x = x.astype(np.int64)
    return np.where(x > q // 2, x - q, x)


@pytest.mark.parametrize("name", ["tiny", "c3", "c5"])
def test_encrypt_decrypt_cross_implementation(name, rng):
    g, o = make(name, galois=False)
    vals = rng.integers(0, o.t, size=(3, o.n), dtype=np.uint64)
    plains = np.stack([o.encode(v) for v in vals])
    ph, ch, dh = g.pt_alloc(3), g.ct_alloc(3), g.pt_alloc(3)
    g.pt_upload(ph, 0, plains)
    g.encrypt(ph, 0, ch, 0, 3, seed=5)
    cts = g.ct_download(ch, 0, 3)
    for i in range(3):                                            # GPU encrypt -> oracle decrypt
        assert np.array_equal(o.decode(o.decrypt(cts[i])), vals[i])
    assert not np.array_equal(cts[0][:o.n], cts[1][:o.n])
    g.encrypt(ph, 0, ch, 0, 3, seed=5)                            # a second call continues the stream: fresh randomness
    assert not np.array_equal(g.ct_download(ch, 0, 1)[0], cts[0])
    g.decrypt(ch, 0, 3, dh, 0)                                    # GPU decrypt of GPU ciphertexts
    assert np.array_equal(g.pt_download(dh, 0, 3), plains)
    oc = np.stack([o.encrypt(p) for p in plains])                 # oracle encrypt -> GPU decrypt
    g.ct_upload(ch, 0, oc)
    g.decrypt(ch, 0, 3, dh, 0)
    assert np.array_equal(g.pt_download(dh, 0, 3), plains)
    # decode on the device as well
    assert np.array_equal(g.decode(dh, 1), vals[1])
    # fresh noise is small: c0 + c1 s - Delta m has |.| <= a few hundred
    x = o.dot_with_secret(cts[0]).reshape(o.k, o.n)
    ref = o.dot_with_secret(o.add_plain(np.zeros_like(cts[0]), plains[0])).reshape(o.k, o.n)
    for j in range(o.k):
        e = centered((x[j] + o.q[j] - ref[j]) % o.q[j], o.q[j])
        assert np.abs(e).max() < 64 * np.sqrt(o.n), np.abs(e).max()
    # encryption of zero (PoolLayer padded taps, AtomicSealBfvVector.cs:566)
    g.encrypt(0, 0, ch, 0, 1, seed=9)
    assert not o.decrypt(g.ct_download(ch, 0, 1)[0]).any()
    for h in (ph, ch, dh):
        g.free(h)


@pytest.mark.parametrize("name", ["tiny", "c4", "c5"])
def test_device_keygen_under_the_xi_convention(name, rng):
    """cn_set_option("ks_xi", 1) before cn_keygen: key (l, d) carries (q/q_l) 2^(dbc d) s' in EVERY limb.  The device-made keys, imported
    into an oracle of the same convention, give the device's words there too (cross-implementation, both directions of trust) and the
    results decrypt to the product / the rotated slots."""
    from cryptonets_amd._native import Context
    from oracle.cno import Oracle
    p = PARAMS[name]
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    g.set_option("ks_xi", 1)
    g.keygen(31, galois=True)
    o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], ks_xi=True)
    o.import_keys(g.get_key(3), g.get_key(2))
    o.import_relin_key(g.get_key(0))
    e1, em1 = g.galois_elt_from_step(1), g.galois_elt_from_step(-2)
    for e in (e1, em1, 2 * o.n - 1):
        o.import_galois_key(e, g.get_key(1, e))
    vals = rng.integers(0, o.t, size=(2, o.n), dtype=np.uint64)
    cts = np.stack([o.encrypt(o.encode(v)) for v in vals])
    h, out, dh = g.ct_alloc(2), g.ct_alloc(3), g.pt_alloc(3)
    g.ct_upload(h, 0, cts)
    g.mul_relin(h, 0, h, 1, out, 0, 1)
    g.rotate_rows(h, 0, 1, out, 1, 1)
    g.rotate_columns(h, 1, out, 2, 1)
    got = g.ct_download(out, 0, 3)
    assert np.array_equal(got[0], o.relinearize(o.multiply(cts[0], cts[1])))
    assert np.array_equal(got[1], o.apply_galois(cts[0], e1))
    assert np.array_equal(got[2], o.rotate_columns(cts[1]))
    g.decrypt(out, 0, 3, dh, 0)
    dec = g.decode_batch(dh, 0, 3)
    half = o.n // 2
    assert np.array_equal(dec[0], np.array([int(a) * int(b) % o.t for a, b in zip(vals[0], vals[1])], dtype=np.uint64))
    assert np.array_equal(dec[1], np.concatenate([np.roll(vals[0][:half], -1), np.roll(vals[0][half:], -1)]))
    assert np.array_equal(dec[2], np.concatenate([vals[1][half:], vals[1][:half]]))
    # structure: the message term sits in a limb OTHER than l too
    n, k = o.n, o.k
    sk = g.get_key(3).reshape(k, n)
    rl = g.get_key(0).reshape(-1, 2, k, n)
    Q = 1
    for qj in o.q:
        Q *= int(qj)
    l, d, j = 0, 0, 1                                    # key component 0 = (limb 0, digit 0), looked at in limb 1
    msg = (Q // int(o.q[l])) * pow(2, o.dbc * d) % int(o.q[j])
    v = (rl[d, 0, j].astype(object) + rl[d, 1, j].astype(object) * sk[j].astype(object) - sk[j].astype(object) ** 2 * msg) % int(o.q[j])
    assert np.abs(centered(o.ntt_inv(j, np.array(v, dtype=np.uint64)), o.q[j])).max() <= 19
    for x in (h, out, dh):
        g.free(x)
    g.close()


def test_decrypt_size3(rng):
    g, o = make("c3", galois=False)
    vals = rng.integers(0, o.t, size=(2, o.n), dtype=np.uint64)
    cts = np.stack([o.encrypt(o.encode(v)) for v in vals])
    h, h3, dh = g.ct_alloc(2), g.ct_alloc(1, 3), g.pt_alloc(1)
    g.ct_upload(h, 0, cts)
    g.multiply(h, 0, h, 1, h3, 0, 1)
    g.decrypt(h3, 0, 1, dh, 0)
    exp = np.array([int(a) * int(b) % o.t for a, b in zip(vals[0], vals[1])], dtype=np.uint64)
    assert np.array_equal(g.decode(dh, 0), exp)


@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_device_keys_have_the_right_structure(name):
    g, o = make(name, galois=True)
    n, k = o.n, o.k
    sk = g.get_key(3).reshape(k, n)
    pk = g.get_key(2).reshape(2, k, n)

    def err(b, a, j, extra=None):
        """INTT(b + a*s (- extra)) centred: must be the small error polynomial (negated)"""
        v = (b.astype(object) + a.astype(object) * sk[j].astype(object)) % o.q[j]
        if extra is not None:
            v = (v - extra) % o.q[j]
        return centered(o.ntt_inv(j, np.array(v, dtype=np.uint64)), o.q[j])
    # secret key is ternary
    for j in range(k):
        s = centered(o.ntt_inv(j, sk[j]), o.q[j])
        assert set(np.unique(s)) <= {-1, 0, 1}
        if j == 0:
            counts = [(s == v).mean() for v in (-1, 0, 1)]
            assert all(0.25 < c < 0.42 for c in counts), counts
    for j in range(k):
        e = err(pk[0, j], pk[1, j], j)
        assert np.abs(e).max() <= 19 and 2.0 < e.std() < 4.5
    # relinearisation key: message 2^(dbc d) s^2 on limb l only
    rl = g.get_key(0).reshape(-1, 2, k, n)
    dig = [len(range(0, int(q).bit_length(), o.dbc)) for q in o.q]
    comp = 0
    for l in range(k):
        for d in range(dig[l]):
            for j in (l, (l + 1) % k):
                extra = None
                if j == l:
                    f = pow(2, o.dbc * d, o.q[l])
                    extra = (sk[l].astype(object) ** 2 * f) % o.q[l]
                e = err(rl[comp, 0, j], rl[comp, 1, j], j, extra)
                assert np.abs(e).max() <= 19, (l, d, j)
            comp += 1
    assert comp == rl.shape[0]
    # Galois key for 2N-1 (column swap): message sigma(s) with the first digit on limb 0
    elt = 2 * n - 1
    gk = g.get_key(1, elt).reshape(-1, 2, k, n)
    s0 = o.ntt_inv(0, sk[0])
    sig = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        raw = i * elt
        sig[raw & (n - 1)] = (o.q[0] - s0[i]) % o.q[0] if (raw >> int(np.log2(n))) & 1 else s0[i]
    e = err(gk[0, 0, 0], gk[0, 1, 0], 0, o.ntt_fwd(0, sig).astype(object))
    assert np.abs(e).max() <= 19
    # and the keys work: rotate + multiply on the device, decrypt on the device
    vals = np.arange(n, dtype=np.uint64) % o.t
    ph, ch, dh = g.pt_alloc(1), g.ct_alloc(2), g.pt_alloc(1)
    g.encode(vals, ph, 0)
    g.encrypt(ph, 0, ch, 0, 1, seed=3)
    g.rotate_rows(ch, 0, -3, ch, 1, 1)
    g.mul_relin(ch, 0, ch, 1, ch, 1, 1)
    g.decrypt(ch, 1, 1, dh, 0)
    half = n // 2
    rot = np.concatenate([np.roll(vals[:half], 3), np.roll(vals[half:], 3)])
    assert np.array_equal(g.decode(dh, 0), np.array([int(a) * int(b) % o.t for a, b in zip(vals, rot)], dtype=np.uint64))


def test_reference_kats_with_the_device_client():
    """HE Wrapper Tests/BasicOperations.cs values through the default factory with keygen / encrypt / decrypt on the GPU - no
    oracle anywhere in the loop."""
    from cryptonets_amd.hewrapper import EMatrixFormat, EncryptedSealBfvFactory, EVectorFormat
    F = EncryptedSealBfvFactory()                                  # default: N=4096, 5 plaintext primes, DeviceClient
    env = F.AllocateComputationEnv()
    v1 = np.array([-1, 9, 3, 20, 1000, -6945], dtype=float)
    v2 = np.array([8, -22, 5, 4, 254, -12], dtype=float)
    e1, e2 = F.GetEncryptedVector(v1, EVectorFormat.dense, 12.0), F.GetEncryptedVector(v2, EVectorFormat.dense, 12.0)
    p2 = F.GetPlainVector(v2, EVectorFormat.dense, 12.0)
    assert np.array_equal(e1.Decrypt(env), v1)
    assert np.array_equal(e1.Add(e2, env).Decrypt(env), v1 + v2)
    assert np.array_equal(e1.PointwiseMultiply(e2, env).Decrypt(env), v1 * v2)
    assert np.array_equal(e1.PointwiseMultiply(p2, env).Decrypt(env), v1 * v2)
    assert e1.DotProduct(e2, env).Decrypt(env)[0] == float(v1 @ v2)
    assert e1.DotProduct(e2, env, length=4).Decrypt(env)[3] == float(v1[:4] @ v2[:4])
    m = np.array([[1, -2, 3, -44, 5, 7], [99, 12, -88, 22, 16, 13]], dtype=float)
    mat = F.GetEncryptedMatrix(m, EMatrixFormat.ColumnMajor, 12.0)
    sp = F.GetEncryptedVector(v1, EVectorFormat.sparse, 12.0)
    assert np.array_equal(mat.Mul(sp, env).Decrypt(env), m @ v1)
    dup = e1.Duplicate(10, env)
    exp = np.zeros(8)
    exp[:6] = v1
    assert np.array_equal(dup.Decrypt(env), np.tile(exp, 10))
    data = np.array([[0, 0, 1, 0, 0, 2], [0, 0, 3, 0, 0, 4], [0, 0, 5, 0, 0, 6]], dtype=float).T
    mm = F.GetEncryptedMatrix(data, EMatrixFormat.ColumnMajor, 10)
    assert list(mm.Interleave(-1, env).Decrypt(env)) == [5, 3, 1, 6, 4, 2]


def test_basic_example_on_the_default_encrypted_factory():
    """BASELINE config 0 as the reference ships it (`Basic Example/Program.cs:18`: `new EncryptedSealBfvFactory()`, N = 4096,
    five plaintext primes): the same three results as on RawFactory (tests/test_raw_operations.py::test_basic_example)."""
    import importlib.util
    import os
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    from cryptonets_amd.raw import RawFactory
    spec = importlib.util.spec_from_file_location("basic_example", os.path.join(os.path.dirname(__file__), "..", "examples", "basic_example.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    enc = mod.run(EncryptedSealBfvFactory())
    raw = mod.run(RawFactory(4096))
    assert enc["norm_squared"][0] == raw["norm_squared"][0] == 14.0
    assert enc["sum"][0] == raw["sum"][0] == 6.0
    assert enc["elementwise"][:3] == raw["elementwise"] == [-1.0, 10.0, -12.0]


@pytest.mark.gpu
def test_default_client_draws_its_randomness_from_the_os():
    """The default DeviceClient seeds key generation and EVERY encryption from os.urandom (plus a per-context salt): two factories never
    share keys, two encryptions of one plaintext never share (u, e1, e2); an explicit seed (tests) reproduces keys and ciphertexts."""
    from cryptonets_amd._native import Context
    from cryptonets_amd.client import DeviceClient
    n, t = 4096, 40961
    keys, cts = [], []
    for _ in range(2):
        g = Context(n, t)
        c = DeviceClient(g)
        c.generate_keys(with_galois=False)
        keys.append(g.get_key(3))
        p = np.zeros(n, dtype=np.uint64)
        p[0] = 7
        cts.append([c.encrypt(p), c.encrypt(p)])
        assert np.array_equal(c.decrypt(cts[-1][0]), p) and np.array_equal(c.decrypt(cts[-1][1]), p)
        g.close()
    assert not np.array_equal(keys[0], keys[1])
    assert not np.array_equal(cts[0][0], cts[0][1]) and not np.array_equal(cts[0][0], cts[1][0])
    fixed = []
    for _ in range(2):
        g = Context(n, t)
        c = DeviceClient(g, seed=1234)
        c.generate_keys(with_galois=False)
        fixed.append((g.get_key(3), c.encrypt(np.arange(n, dtype=np.uint64) % t)))
        g.close()
    assert np.array_equal(fixed[0][0], fixed[1][0]) and np.array_equal(fixed[0][1], fixed[1][1])


@pytest.mark.gpu
def test_rng_salt_changes_every_stream():
    from cryptonets_amd._native import Context
    outs = []
    for salt in (0, 0, 0x1234567890ABCDEF):
        g = Context(1024, 12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001])
        g.set_rng_salt(salt)
        g.keygen(99, galois=False)
        outs.append((g.get_key(3), g.get_key(2)))
        g.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert not np.array_equal(outs[0][0], outs[2][0]) and not np.array_equal(outs[0][1], outs[2][1])


@pytest.mark.gpu
def test_recording_rejects_encryption_and_key_changes():
    """a replayed graph would reuse the randomness of a recorded cn_encrypt; key changes synchronise and reallocate: both are refused
    while a graph is recorded, and the recording can still be closed afterwards"""
    from cryptonets_amd._native import Context, CnError
    g = Context(1024, 12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001])
    g.keygen(5, galois=False)
    rk = g.get_key(0)
    pt, ct = g.pt_alloc(1), g.ct_alloc(1)
    g.pt_upload(pt, 0, np.ones((1, 1024), dtype=np.uint64))
    g.encrypt(pt, 0, ct, 0, 1, seed=3)
    g.graph_begin()
    with pytest.raises(CnError):
        g.encrypt(pt, 0, ct, 0, 1, seed=3)
    with pytest.raises(CnError):
        g.set_relin_key(rk)
    g.add(ct, 0, ct, 0, ct, 0)
    graph = g.graph_end()
    g.graph_launch(graph)
    g.sync()
    g.free(graph)
    g.close()


@pytest.mark.gpu
def test_multiply_plain_rejects_partially_overlapping_ranges():
    from cryptonets_amd._native import Context, CnError
    g = Context(1024, 12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001])
    g.keygen(5, galois=False)
    pt, ct = g.pt_alloc(1), g.ct_alloc(4)
    g.pt_upload(pt, 0, np.ones((1, 1024), dtype=np.uint64))
    g.encrypt(pt, 0, ct, 0, 4, seed=3, pt_stride=0)
    g.mul_plain(ct, 0, pt, 0, ct, 0, 2, pt_stride=0)               # exactly in place: fine
    g.mul_plain(ct, 0, pt, 0, ct, 2, 2, pt_stride=0)               # disjoint: fine
    with pytest.raises(CnError):
        g.mul_plain(ct, 0, pt, 0, ct, 1, 2, pt_stride=0)           # [0,2) -> [1,3): a block would read what another has overwritten
    g.close()


@pytest.mark.gpu
def test_ct_upload_checks_the_row_width():
    from cryptonets_amd._native import Context
    g = Context(1024, 12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001])
    h3 = g.ct_alloc(2, 3)
    with pytest.raises(ValueError):
        g.ct_upload(h3, 0, np.zeros((2, 2 * 3 * 1024), dtype=np.uint64))      # size-2 rows into a size-3 handle
    with pytest.raises(ValueError):
        g.ct_upload(h3, 1, np.zeros((2, 3 * 3 * 1024), dtype=np.uint64))      # past the end
    g.ct_upload(h3, 0, np.zeros((2, 3 * 3 * 1024), dtype=np.uint64))
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "c4"])
def test_encode_decode_batch_equal_the_oracle(name, rng):
    """cn_encode_batch / cn_decode_batch (one upload, one scatter, one batched transform mod t) against BatchEncoder of the oracle, with
    ragged rows (fewer values than slots), an all-zero row (IsZero bookkeeping: MultiplyPlain must refuse it) and the single-plaintext calls"""
    from cryptonets_amd._native import CnError
    o, g = get_oracle(name), get_gpu(name)
    vals = rng.integers(0, o.t, size=(7, o.n), dtype=np.uint64)
    vals[3] = 0
    ph = g.pt_alloc(9)
    g.encode_batch(vals, ph, 1)
    got = g.pt_download(ph, 1, 7)
    for r in range(7):
        assert np.array_equal(got[r], o.encode(vals[r]))
    assert np.array_equal(g.decode_batch(ph, 1, 7), vals)
    short = rng.integers(0, o.t, size=(2, 5), dtype=np.uint64)
    g.encode_batch(short, ph, 0)
    assert np.array_equal(g.pt_download(ph, 0, 1)[0], o.encode(short[0]))
    assert np.array_equal(g.pt_download(ph, 1, 1)[0], o.encode(short[1]))
    assert np.array_equal(g.decode(ph, 1)[:5], short[1]) and not g.decode(ph, 1)[5:].any()
    g.encode(vals[6], ph, 8)
    assert np.array_equal(g.pt_download(ph, 8, 1)[0], o.encode(vals[6]))
    ct = g.ct_alloc(1)
    g.ct_upload(ct, 0, o.encrypt(o.encode(vals[0]))[None, :])
    with pytest.raises(CnError):
        g.mul_plain(ct, 0, ph, 4, ct, 0)                  # row 3 of the batch: the zero plaintext
    with pytest.raises(CnError):
        g.encode_batch(np.full((1, 4), o.t, dtype=np.uint64), ph, 0)
    with pytest.raises(CnError):
        g.encode_batch(vals, ph, 3)                       # 7 plaintexts do not fit behind index 3
    g.free(ph), g.free(ct)


@pytest.mark.gpu
def test_sampler_generator_is_chacha20_rfc7539():
    """the device sampler's generator against the RFC 7539 section 2.3.2 block-function test vector (key 00..1f, block counter 1, nonce
    00:00:00:09:00:00:00:4a:00:00:00:00) and against a host model of the same 20-round function on a second input"""
    from cryptonets_amd._native import Context
    p = PARAMS["tiny"]
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    key = bytes(range(32))
    got = g.rng_block(key, (0x09000000 << 32) | 1, 0x4a000000)
    want = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
            0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert [int(x) for x in got] == want

    def model(key, counter, nonce):
        rot = lambda x, r: ((x << r) | (x >> (32 - r))) & 0xffffffff
        st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)] + \
             [counter & 0xffffffff, counter >> 32, nonce & 0xffffffff, nonce >> 32]
        w = list(st)

        def qr(a, b, c, d):
            w[a] = (w[a] + w[b]) & 0xffffffff; w[d] = rot(w[d] ^ w[a], 16)
            w[c] = (w[c] + w[d]) & 0xffffffff; w[b] = rot(w[b] ^ w[c], 12)
            w[a] = (w[a] + w[b]) & 0xffffffff; w[d] = rot(w[d] ^ w[a], 8)
            w[c] = (w[c] + w[d]) & 0xffffffff; w[b] = rot(w[b] ^ w[c], 7)
        for _ in range(10):
            qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15)
            qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14)
        return [(a + b) & 0xffffffff for a, b in zip(w, st)]
    assert model(key, (0x09000000 << 32) | 1, 0x4a000000) == want
    key2 = bytes((7 * i + 3) & 255 for i in range(32))
    assert [int(x) for x in g.rng_block(key2, 0x0123456789abcdef, 0xfedcba9876543210)] == model(key2, 0x0123456789abcdef, 0xfedcba9876543210)
    g.close()


@pytest.mark.gpu
def test_encryptions_depend_on_key_nonce_and_item(rng):
    """fresh encryptions of the same plaintext: equal words for equal (key, nonce, item), different words when any of the three changes; the
    default client draws a 256-bit key from the OS"""
    from cryptonets_amd._native import Context
    p = PARAMS["tiny"]

    def enc(key, nonce, skip):
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        g.keygen(99, galois=False)                       # (all-zero sampler key: the same keys every time)
        if key is not None:
            g.set_rng_key(key)
        ph, ct = g.pt_alloc(1), g.ct_alloc(2)
        g.encode(np.arange(8, dtype=np.uint64), ph, 0)
        g.encrypt(ph, 0, ct, 0, 2, seed=nonce, pt_stride=0)          # two ciphertexts of one plaintext: items i, i + 1
        w = g.ct_download(ct, skip, 1)[0]
        g.close()
        return w
    base = enc(bytes(32), 5, 0)
    assert np.array_equal(base, enc(bytes(32), 5, 0))
    assert not np.array_equal(base, enc(bytes(32), 6, 0))
    assert not np.array_equal(base, enc(bytes([1] + [0] * 31), 5, 0))
    assert not np.array_equal(base, enc(bytes(32), 5, 1))


def test_encrypt_zero_new_and_free_many(rng):
    """cn_encrypt_zero_new = cn_ct_alloc + cn_encrypt(pt = 0) in one call (PoolLayer.ElementAt per padded tap, PoolLayer.cs:67-80): same words -
    immediately and under deferred submission - on two contexts with the same keys and the same history; cn_free_many releases n handles with one call,
    all or nothing."""
    from cryptonets_amd._native import CnError
    words = []
    for merged in (False, True):
        for defer in (0, 1):
            g, o = make("tiny", seed=5, galois=False)
            g.set_option("defer", defer)
            hs = []
            for i in range(6):
                if merged:
                    hs.append(g.encrypt_zero_new(seed=100 + i))
                else:
                    h = g.ct_alloc(1)
                    g.encrypt(0, 0, h, 0, 1, seed=100 + i)
                    hs.append(h)
            w = np.stack([g.ct_download(h, 0, 1)[0] for h in hs])
            assert all(not o.decrypt(c).any() for c in w) and len({c.tobytes() for c in w}) == 6
            words.append(w)
            live = g.live_handles()
            with pytest.raises(CnError):
                g.free_many(hs[:3] + [hs[1]])                 # repeated handle: nothing is released
            with pytest.raises(CnError):
                g.free_many(hs[:3] + [0xdeadbeef])            # invalid handle: nothing is released
            assert g.live_handles() == live
            t = g.ct_alloc(1)
            g.add(hs[0], 0, hs[1], 0, t, 0)                   # (deferred: a pending reader of the arrays that are released next)
            g.free_many(hs)
            assert g.live_handles() == live - 6 + 1
            assert np.array_equal(g.ct_download(t, 0, 1)[0], o.add(w[0], w[1]))
            g.free_many([])
            g.free_many([t])
            assert g.live_handles() == live - 6
            g.close()
    for w in words[1:]:
        assert np.array_equal(w, words[0])


@pytest.mark.parametrize("name,f64", [("tiny", True), ("default4096", True), ("c3", True), ("c3", False), ("n16k7", True)])
def test_fused_encryption_kernel_is_the_three_launch_chain(name, f64, rng):
    """k_encrypt_fused (u from the sampler's int8 polynomial -> one transform -> both components, N <= 8192) against expand + batched transform +
    k_encrypt_tail: the SAME words for the same key, nonce and sampler items - dense plaintexts, encryptions of zero, a broadcast plaintext, and the
    deferred per-ciphertext form (one-call zero vectors); at N = 16384 the switch changes nothing (the chain stays)."""
    from cryptonets_amd._native import Context
    p = PARAMS[name]
    words = {}
    for fused in (1, 0, 2):                                      # 2: a block per (ciphertext, component, limb) - k_encrypt_split (round 6)
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        if not f64:
            g.set_option("f64", 0)
        g.set_option("enc_fused", fused)
        assert g.get_option("enc_fused") == fused
        g.keygen(41, galois=False)
        r = np.random.default_rng(8)
        plains = r.integers(0, g.t, size=(5, g.n), dtype=np.uint64)
        ph, ch = g.pt_alloc(5), g.ct_alloc(12)
        g.pt_upload(ph, 0, plains)
        g.encrypt(ph, 0, ch, 0, 5, seed=3)                       # five dense plaintexts
        g.encrypt(0, 0, ch, 5, 3, seed=4)                        # three encryptions of zero
        g.encrypt(ph, 2, ch, 8, 2, seed=5, pt_stride=0)          # one plaintext twice (fresh randomness each)
        g.set_option("defer", 1)
        z = [g.encrypt_zero_new(seed=60 + i) for i in range(3)]
        g.encrypt(ph, 1, ch, 10, 2, seed=7)                      # queued beside them
        g.set_option("defer", 0)
        w = [g.ct_download(ch, 0, 12)] + [g.ct_download(h, 0, 1) for h in z]
        words[fused] = np.concatenate(w)
        if fused == 1:
            o = get_oracle(name, galois=False)
            from oracle.cno import Oracle
            oo = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"])
            oo.import_keys(g.get_key(3), g.get_key(2))
            dec = [oo.decrypt(c) for c in words[1]]
            want = [plains[i] for i in range(5)] + [np.zeros(g.n, dtype=np.uint64)] * 3 + [plains[2]] * 2 + [plains[1], plains[2]] + [np.zeros(g.n, dtype=np.uint64)] * 3
            assert all(np.array_equal(a, b) for a, b in zip(dec, want))
            assert not np.array_equal(words[1][8], words[1][9])
        g.close()
    assert np.array_equal(words[0], words[1]) and np.array_equal(words[2], words[1])
