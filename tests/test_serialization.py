"""Wire formats (cryptonets_amd/serialization.py).

The SEAL 3.2 object layouts are restated (no SEAL binary or SEAL-written file exists here: byte parity with real SEAL is
UNPINNED); what is pinned: the wrapper's text framing against the reference source, self-consistency of every object stream,
and the validity checks of Load (foreign parameters, truncated or padded streams, residues out of range).
"""
import hashlib
import io
import struct
import zipfile

import numpy as np
import pytest

from cryptonets_amd import serialization as S
from conftest import PARAMS


def parms(name="tiny"):
    from oracle.cno import COEFF_MODULUS_128
    p = PARAMS[name]
    return S.Parameters(p["n"], p["q"] or COEFF_MODULUS_128[p["n"]], p["t"])


def random_ct(P, rng, size=2):
    return np.concatenate([rng.integers(0, q, size=P.n, dtype=np.uint64) for _ in range(size) for q in P.q])


def test_parameters_and_parms_id():
    P = parms("c3")
    buf = io.BytesIO()
    P.save(buf)
    raw = buf.getvalue()
    assert len(raw) == 1 + 8 + 8 + 8 * P.k + 8 + 8 and raw[0] == 1
    assert struct.unpack_from("<Q", raw, 1)[0] == 8192 and struct.unpack_from("<d", raw, len(raw) - 8)[0] == 3.2
    assert S.Parameters.load(io.BytesIO(raw)) == P
    # parms_id = SHA3-256 of the little-endian u64 words [scheme, N, q..., t, bits(sigma)]
    words = [1, P.n] + P.q + [P.t, struct.unpack("<Q", struct.pack("<d", 3.2))[0]]
    assert P.parms_id() == hashlib.sha3_256(b"".join(struct.pack("<Q", w) for w in words)).digest()
    ids = {P.parms_id(), S.Parameters(P.n, P.q[:4], P.t).parms_id(), S.Parameters(P.n, P.q, P.t + 2).parms_id(),
           S.Parameters(P.n, P.q, P.t, 3.19).parms_id(), S.Parameters(4096, P.q, P.t).parms_id()}
    assert len(ids) == 5
    with pytest.raises(S.BadStream):
        S.Parameters.load(io.BytesIO(raw[:-3]))
    with pytest.raises(S.BadStream):
        S.Parameters.load(io.BytesIO(b"\x02" + raw[1:]))           # CKKS stream


def test_ciphertext_plaintext_streams(rng):
    P = parms("tiny")
    ct3 = random_ct(P, rng, 3)
    buf = io.BytesIO()
    S.save_ciphertext(buf, ct3, P, size=3)
    raw = buf.getvalue()
    assert len(raw) == 32 + 1 + 8 * 3 + 8 + 8 + 8 * ct3.size
    got, size = S.load_ciphertext(io.BytesIO(raw), P)
    assert size == 3 and np.array_equal(got, ct3)
    with pytest.raises(S.BadStream):                                # another environment's ciphertext
        S.load_ciphertext(io.BytesIO(raw), S.Parameters(P.n, P.q, P.t + 2))
    with pytest.raises(S.BadStream):
        S.load_ciphertext(io.BytesIO(raw[:-8]), P)
    with pytest.raises(S.BadStream):
        S.load_ciphertext(io.BytesIO(raw), P, want_ntt_form=True)
    bad = ct3.copy()
    bad[5] = P.q[0]                                                 # residue not reduced
    buf = io.BytesIO()
    S.save_ciphertext(buf, bad, P, size=3)
    with pytest.raises(S.BadStream):
        S.load_ciphertext(io.BytesIO(buf.getvalue()), P)
    with pytest.raises(ValueError):
        S.save_ciphertext(io.BytesIO(), ct3[:-1], P, size=3)
    # plaintexts: full, constant (CoeffCount 1), empty
    for coeffs in (rng.integers(0, P.t, size=P.n, dtype=np.uint64), np.array([7], dtype=np.uint64), np.zeros(0, dtype=np.uint64)):
        buf = io.BytesIO()
        S.save_plaintext(buf, coeffs)
        assert len(buf.getvalue()) == 32 + 8 + 8 + 8 * coeffs.size
        assert np.array_equal(S.load_plaintext(io.BytesIO(buf.getvalue()), P), coeffs)
    buf = io.BytesIO()
    S.save_plaintext(buf, np.array([P.t], dtype=np.uint64))
    with pytest.raises(S.BadStream):
        S.load_plaintext(io.BytesIO(buf.getvalue()), P)


def test_key_switching_key_streams(rng):
    P = parms("tiny")
    dig = S.digit_count(P, 10)
    assert dig == sum(-(-q.bit_length() // 10) for q in P.q)
    rl = np.stack([random_ct(P, rng) for _ in range(dig)])
    buf = io.BytesIO()
    S.save_kswitch_keys(buf, P, 10, [rl])
    dbc, got = S.load_kswitch_keys(io.BytesIO(buf.getvalue()), P)
    assert dbc == 10 and len(got) == 1 and np.array_equal(got[0], rl)
    gk = [None] * P.n
    gk[1], gk[P.n - 1] = rl[:3], rl[1:4]                            # galois elements 3 and 2N-1
    buf = io.BytesIO()
    S.save_kswitch_keys(buf, P, 20, gk)
    dbc, got = S.load_kswitch_keys(io.BytesIO(buf.getvalue()), P, expect_dbc=20)
    assert [i for i, e in enumerate(got) if e is not None] == [1, P.n - 1] and np.array_equal(got[P.n - 1], rl[1:4])
    with pytest.raises(S.BadStream):
        S.load_kswitch_keys(io.BytesIO(buf.getvalue()), P, expect_dbc=10)
    with pytest.raises(S.BadStream):
        S.load_kswitch_keys(io.BytesIO(buf.getvalue()[:-100]), P)


@pytest.fixture(scope="module")
def cpu_factory():
    from oracle_backend import make_factory
    return make_factory("cpu", primes=[40961, 65537], n=4096)


def test_vector_and_matrix_framing(cpu_factory):
    """EncryptedSealBfvVector / Matrix Write + Read: the text structure of the reference and a value-preserving round trip"""
    from cryptonets_amd.hewrapper import EMatrixFormat, EVectorFormat
    F = cpu_factory
    env = F.AllocateComputationEnv()
    v = np.array([-1, 9, 3, 20, 1000, -6945], dtype=float)
    for encrypted in (True, False):
        for fmt in (EVectorFormat.dense, EVectorFormat.sparse):
            vec = (F.GetEncryptedVector if encrypted else F.GetPlainVector)(v, fmt, 12.0)
            text = io.StringIO()
            vec.Write(text, env)
            lines = text.getvalue().split("\n")
            assert lines[0] == "<Start LargeEncryptedVector>" and lines[1] == "12" and lines[2] == "2"
            assert lines[3] == "<Start EncryptedVector>" and lines[4] == "1" and lines[5] == "False"      # under the CRT layer atoms are unsigned residues with scale 1
            assert lines[6] == fmt.name and lines[7] == "6" and lines[8] == ("Encrypted" if encrypted else "Plain")
            assert lines[9] == ("1" if fmt == EVectorFormat.dense else "6") and lines[11] == "<End EncryptedVector>"
            assert lines[-2] == "<End LargeEncryptedVector>" and lines[-1] == ""
            back = F.LoadVector(io.StringIO(text.getvalue()))
            assert back.Scale == 12.0 and back.Format == fmt and back.IsEncrypted == encrypted
            if encrypted:
                assert np.array_equal(back.Decrypt(env)[:6], v)
            else:                                                   # a plain operand: use it
                e = F.GetEncryptedVector(np.ones(6), EVectorFormat.dense, 1.0)
                if fmt == EVectorFormat.dense:
                    assert np.array_equal(e.PointwiseMultiply(back, env).Decrypt(env)[:6], v)
                else:
                    assert back.eVectors[0].plainSparse == vec.eVectors[0].plainSparse
    m = np.array([[1, -2, 3], [99, 12, -88]], dtype=float)
    mat = F.GetEncryptedMatrix(m, EMatrixFormat.ColumnMajor, 4.0)
    text = io.StringIO()
    mat.Write(text, env)
    head = text.getvalue().split("\n")[:3]
    assert head == ["<Start LargeEncryptedMatrix>", "ColumnMajor", "3"]
    back = F.LoadMatrix(io.StringIO(text.getvalue()))
    assert back.Format == EMatrixFormat.ColumnMajor and np.array_equal(np.asarray(back.Decrypt(env))[:2, :3], m)
    # malformed text
    broken = text.getvalue().replace("<End EncryptedVector>", "<End Vector>", 1)
    with pytest.raises(S.BadStream):
        F.LoadMatrix(io.StringIO(broken))
    with pytest.raises(S.BadStream):
        F.LoadVector(io.StringIO(text.getvalue()))                  # a matrix is not a vector
    with pytest.raises(S.BadStream):
        F.LoadMatrix(io.StringIO(text.getvalue()[:2000]))
    assert S._fmt_double(1e15) == "1E+15" and S._fmt_double(0.03125) == "0.03125" and S._fmt_double(1e-7) == "1E-07"


class RecordingContext:
    """stands in for a libcnhip context on the CPU: records what load_environment installs"""

    def __init__(self, n, t, q, dbc, gdbc):
        self.n, self.t, self.q, self.k, self.dbc, self.gdbc = n, t, list(q), len(q), dbc, gdbc
        self.pk = self.sk = self.rl = None
        self.gk = {}

    def set_public_key(self, w):
        self.pk = np.array(w)

    def set_secret_key(self, w):
        self.sk = np.array(w)

    def set_relin_key(self, w):
        self.rl = np.array(w)

    def set_galois_key(self, elt, w):
        self.gk[elt] = np.array(w)


def test_key_container(cpu_factory):
    """IFactory.Save -> zip of environmentNNN entries (stored, one per plaintext prime) -> load: every key arrives intact;
    without private keys the secret key is absent"""
    F = cpu_factory
    envs = F.AllocateComputationEnv().Environments
    for private in (True, False):
        blob = io.BytesIO()
        F.Save(blob, withPrivateKeys=private)
        with zipfile.ZipFile(io.BytesIO(blob.getvalue())) as z:
            assert z.namelist() == ["environment000", "environment001"]
            assert all(i.compress_type == zipfile.ZIP_STORED for i in z.infolist())
        made = []

        def context_factory(n, t, q, dbc, gdbc):
            made.append(RecordingContext(n, t, q, dbc, gdbc))
            return made[-1]
        loaded = S.load_environments(io.BytesIO(blob.getvalue()), context_factory, client_factory=lambda ctx: object.__new__(type("C", (), {})))
        assert len(loaded) == 2
        for e, c in zip(envs, made):
            o = e.ctx.o
            assert (c.n, c.t, c.q, c.dbc, c.gdbc) == (o.n, o.t, list(o.q), o.dbc, o.gdbc)
            assert np.array_equal(c.pk, o.public_key()) and np.array_equal(c.rl, o.relin_key())
            assert sorted(c.gk) == sorted(set(o.galois_elts()))     # 3^(N/4) is its own inverse: listed twice by the default rule
            for elt in set(o.galois_elts()):
                assert np.array_equal(c.gk[elt], o.galois_key(o.galois_elts().index(elt)))
            if private:
                assert np.array_equal(c.sk, o.secret_key())
            else:
                assert c.sk is None
    # a corrupted entry is rejected
    raw = bytearray(blob.getvalue())
    raw[200] ^= 0xFF
    with pytest.raises((S.BadStream, zipfile.BadZipFile)):
        S.load_environments(io.BytesIO(bytes(raw)), context_factory)
