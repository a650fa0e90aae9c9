"""Wire formats: SEAL 3.2 object streams, the wrapper's text + base64 framing, the zip key container.

Two layers, as in the reference:

* wrapper framing - exact, taken from the reference source: `AtomicSealBfvEncryptedVector.Write/Read`
  (AtomicSealBfvVector.cs:1273-1345), `EncryptedSealBfvVector.Write/Read` (EncryptedSealBfvVector.cs:414-439),
  `EncryptedSealBfvMatrix.Write/Read` (EncryptedSealBfvMatrix.cs:182-208), `AtomicSealBfvEncryptedEnvironment.SaveToStream /
  LoadFromStream` (AtomicSealBfvVector.cs:93-130: parameters, public key, relin keys, Galois keys, secret key - an EMPTY secret
  key when saved without private keys) and the zip of `environmentNNN` entries, stored uncompressed
  (EncryptedSealBfvVector.cs:104-134).

* SEAL object streams - the payload inside that framing is produced by SEAL 3.2's own `Save` methods, which are NOT in the
  reference repository (SEALNet 3.2.0 is an un-vendored NuGet dependency, `HE Wrapper/packages.config:6`).  The layouts below
  restate the published SEAL 3.2 `save()` members (all little-endian, no compression in 3.2):

      EncryptionParameters  u8 scheme (1 = BFV) | u64 poly_modulus_degree | u64 coeff_mod_count | u64 q[...] | u64 plain_modulus
                            | f64 noise_standard_deviation
      parms_id              SHA3-256 over the u64 words [scheme, N, q..., t, bits(noise_standard_deviation)] -> 4 x u64
      IntArray<u64>         u64 count | count x u64
      Plaintext             parms_id | f64 scale | IntArray coefficients      (BFV plaintexts carry the all-zero parms_id)
      Ciphertext            parms_id | u8 is_ntt_form | u64 size | u64 poly_modulus_degree | u64 coeff_mod_count | f64 scale
                            | IntArray data, laid out [poly][limb][coefficient]
      PublicKey             the Ciphertext (c0, c1) in NTT form;   SecretKey: a Plaintext of k*N words (NTT form, key parms_id)
      RelinKeys/GaloisKeys  parms_id | i32 decomposition_bit_count | u64 dim1 | dim1 x ( u64 dim2 | dim2 x Ciphertext )
                            RelinKeys: dim1 = 1 (key for s^2), dim2 = number of (limb, digit) pairs, each a size-2 NTT-form
                            ciphertext - the layout libcnhip keeps in HBM (include/cnhip.h).  GaloisKeys: dim1 = N, entry
                            (galois_elt - 1) / 2 holds that element's digits, absent elements have dim2 = 0.

  PARITY UNPINNED: no SEAL binary or SEAL-written file exists in this environment or in the reference repository, so byte
  compatibility with real SEAL 3.2 streams is restated, not tested.  What the tests pin is self-consistency (round trips of
  every object, rejection of foreign parameters / truncated streams) and the wrapper framing against the reference source.
"""
import base64
import hashlib
import io
import struct
import zipfile

import numpy as np

SCHEME_BFV = 1
DEFAULT_NOISE_STANDARD_DEVIATION = 3.20          # SEAL 3.2 util::global_variables::default_noise_standard_deviation
PARMS_ID_ZERO = b"\0" * 32


class BadStream(Exception):
    """the reference throws Exception("Bad stream format.")"""


# ------------------------------------------------------------------------------------------------ primitives
def _rd(f, nbytes):
    b = f.read(nbytes)
    if len(b) != nbytes:
        raise BadStream("Bad stream format. (truncated)")
    return b


def _w_u64(f, v):
    f.write(struct.pack("<Q", int(v)))


def _r_u64(f):
    return struct.unpack("<Q", _rd(f, 8))[0]


def _w_array(f, words):
    a = np.ascontiguousarray(words, dtype="<u8").reshape(-1)
    _w_u64(f, a.size)
    f.write(a.tobytes())


def _r_array(f, limit=1 << 31):
    count = _r_u64(f)
    if count > limit:
        raise BadStream("Bad stream format. (array of %d words)" % count)
    return np.frombuffer(_rd(f, 8 * count), dtype="<u8").astype(np.uint64)


class Parameters:
    """EncryptionParameters of one environment (scheme BFV)."""

    def __init__(self, n, q, t, noise_standard_deviation=DEFAULT_NOISE_STANDARD_DEVIATION):
        self.n, self.q, self.t, self.sigma = int(n), [int(x) for x in q], int(t), float(noise_standard_deviation)

    @property
    def k(self):
        return len(self.q)

    def parms_id(self):
        words = [SCHEME_BFV, self.n, *self.q, self.t, struct.unpack("<Q", struct.pack("<d", self.sigma))[0]]
        return hashlib.sha3_256(struct.pack("<%dQ" % len(words), *words)).digest()

    def save(self, f):
        f.write(struct.pack("<B", SCHEME_BFV))
        _w_u64(f, self.n)
        _w_u64(f, self.k)
        for q in self.q:
            _w_u64(f, q)
        _w_u64(f, self.t)
        f.write(struct.pack("<d", self.sigma))

    @classmethod
    def load(cls, f):
        scheme = struct.unpack("<B", _rd(f, 1))[0]
        if scheme != SCHEME_BFV:
            raise BadStream("unsupported scheme %d (BFV expected)" % scheme)
        n, k = _r_u64(f), _r_u64(f)
        if n < 2 or n & (n - 1) or n > 32768 or not 1 <= k <= 64:
            raise BadStream("Bad stream format. (parameters)")
        q = [_r_u64(f) for _ in range(k)]
        t = _r_u64(f)
        sigma = struct.unpack("<d", _rd(f, 8))[0]
        return cls(n, q, t, sigma)

    def __eq__(self, o):
        return (self.n, self.q, self.t, self.sigma) == (o.n, o.q, o.t, o.sigma)


# ------------------------------------------------------------------------------------------------ SEAL objects
def save_plaintext(f, coeffs, parms_id=PARMS_ID_ZERO, scale=1.0):
    f.write(parms_id)
    f.write(struct.pack("<d", scale))
    _w_array(f, coeffs)


def load_plaintext(f, parms=None, expect_parms_id=PARMS_ID_ZERO):
    pid = _rd(f, 32)
    if expect_parms_id is not None and pid != expect_parms_id:
        raise BadStream("plaintext is not valid for the encryption parameters")
    struct.unpack("<d", _rd(f, 8))
    data = _r_array(f)
    if parms is not None and expect_parms_id == PARMS_ID_ZERO and (data.size > parms.n or (data.size and int(data.max()) >= parms.t)):
        raise BadStream("plaintext is not valid for the encryption parameters")
    return data


def save_ciphertext(f, words, parms, size=2, is_ntt_form=False, scale=1.0):
    w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    if w.size != size * parms.k * parms.n:
        raise ValueError("ciphertext has %d words, expected %d" % (w.size, size * parms.k * parms.n))
    f.write(parms.parms_id())
    f.write(struct.pack("<B", 1 if is_ntt_form else 0))
    _w_u64(f, size)
    _w_u64(f, parms.n)
    _w_u64(f, parms.k)
    f.write(struct.pack("<d", scale))
    _w_array(f, w)


def load_ciphertext(f, parms, want_ntt_form=None):
    """-> (words, size); Ciphertext.Load(context, stream) validity checks: parms_id of THIS context, consistent sizes,
    residues below their moduli"""
    if _rd(f, 32) != parms.parms_id():
        raise BadStream("ciphertext is not valid for the encryption parameters")
    ntt = struct.unpack("<B", _rd(f, 1))[0]
    size, n, k = _r_u64(f), _r_u64(f), _r_u64(f)
    struct.unpack("<d", _rd(f, 8))
    if ntt > 1 or n != parms.n or k != parms.k or not 2 <= size <= 16 or (want_ntt_form is not None and bool(ntt) != want_ntt_form):
        raise BadStream("ciphertext is not valid for the encryption parameters")
    data = _r_array(f)
    if data.size != size * k * n:
        raise BadStream("Bad stream format. (ciphertext data)")
    limbs = data.reshape(size, k, n)
    for j, q in enumerate(parms.q):
        if int(limbs[:, j].max()) >= q:
            raise BadStream("ciphertext is not valid for the encryption parameters")
    return data, int(size)


def save_kswitch_keys(f, parms, dbc, entries):
    """entries: list (dim1) of arrays [digits, 2*k*N] (or None / empty for an absent key)"""
    f.write(parms.parms_id())
    f.write(struct.pack("<i", int(dbc)))
    _w_u64(f, len(entries))
    for e in entries:
        digits = [] if e is None else list(np.asarray(e, dtype=np.uint64).reshape(-1, 2 * parms.k * parms.n))
        _w_u64(f, len(digits))
        for d in digits:
            save_ciphertext(f, d, parms, size=2, is_ntt_form=True)


def load_kswitch_keys(f, parms, expect_dbc=None):
    """-> (decomposition_bit_count, list of arrays [digits, 2*k*N] or None)"""
    if _rd(f, 32) != parms.parms_id():
        raise BadStream("keys are not valid for the encryption parameters")
    dbc = struct.unpack("<i", _rd(f, 4))[0]
    if not 1 <= dbc <= 60 or (expect_dbc is not None and dbc != expect_dbc):
        raise BadStream("keys are not valid for the encryption parameters (decomposition bit count %d)" % dbc)
    dim1 = _r_u64(f)
    if dim1 > parms.n:
        raise BadStream("Bad stream format. (key count)")
    out = []
    for _ in range(dim1):
        dim2 = _r_u64(f)
        if dim2 > 64 * parms.k:
            raise BadStream("Bad stream format. (digit count)")
        out.append(np.stack([load_ciphertext(f, parms, want_ntt_form=True)[0] for _ in range(dim2)]) if dim2 else None)
    return dbc, out


def digit_count(parms, dbc):
    """(limb, digit) pairs of a key-switching key: sum over limbs of ceil(bits(q_l) / dbc)"""
    return sum(-(-q.bit_length() // dbc) for q in parms.q)


# ------------------------------------------------------------------------------------------------ one environment
def _galois_elements(ctx):
    """elements this context holds keys for: KeyGenerator.GaloisKeys(dbc) default set = 3^(+-2^i) and 2N-1"""
    m, n = 2 * ctx.n, ctx.n
    cand, g = {m - 1}, 3
    for _ in range(max(1, n.bit_length() - 2)):
        cand.add(g)
        cand.add(pow(g, -1, m))
        g = g * g % m
    return sorted(e for e in cand if ctx.has_galois_key(e))


def save_environment(stream, env, withPrivateKeys=True, noise_standard_deviation=DEFAULT_NOISE_STANDARD_DEVIATION):
    """AtomicSealBfvEncryptedEnvironment.SaveToStream (AtomicSealBfvVector.cs:93-104).  Keys are read back from HBM
    (cn_get_key); the secret key only exists when the client ran on this context (DeviceClient)."""
    ctx = env.ctx
    parms = Parameters(ctx.n, ctx.q, ctx.t, noise_standard_deviation)
    parms.save(stream)
    save_ciphertext(stream, ctx.get_key(2), parms, size=2, is_ntt_form=True)                       # PublicKey
    save_kswitch_keys(stream, parms, ctx.dbc, [ctx.get_key(0).reshape(-1, 2 * parms.k * parms.n)])   # RelinKeys
    elts = _galois_elements(ctx)
    entries = [None] * parms.n if elts else []
    for e in elts:
        entries[(e - 1) // 2] = ctx.get_key(1, e).reshape(-1, 2 * parms.k * parms.n)
    save_kswitch_keys(stream, parms, ctx.gdbc, entries)                                            # GaloisKeys
    if withPrivateKeys:
        save_plaintext(stream, ctx.get_key(3), parms.parms_id())                                   # SecretKey
    else:
        save_plaintext(stream, np.zeros(0, dtype=np.uint64), PARMS_ID_ZERO)                        # new SecretKey()


def load_environment(stream, context_factory, client_factory=None):
    """AtomicSealBfvEncryptedEnvironment.LoadFromStream (AtomicSealBfvVector.cs:106-130): parameters -> context, then the
    keys.  `context_factory(n, t, q, dbc, gdbc)` builds the libcnhip context; the decomposition bit counts come from the key
    streams.  Without a secret key the environment can evaluate but not decrypt (the reference prints a warning)."""
    from .hewrapper import AtomicSealBfvEncryptedEnvironment
    blob = io.BytesIO(stream.read())
    parms = Parameters.load(blob)
    pk, size = load_ciphertext(blob, parms, want_ntt_form=True)
    if size != 2:
        raise BadStream("public key is not valid for the encryption parameters")
    dbc, rl = load_kswitch_keys(blob, parms)
    gdbc, gk = load_kswitch_keys(blob, parms)
    sk = load_plaintext(blob, parms, expect_parms_id=None)
    if len(rl) != 1 or rl[0] is None or rl[0].shape[0] != digit_count(parms, dbc):
        raise BadStream("relinearization keys are not valid for the encryption parameters")
    ctx = context_factory(parms.n, parms.t, parms.q, dbc, gdbc)
    ctx.set_public_key(pk)
    ctx.set_relin_key(rl[0].reshape(-1))
    for i, e in enumerate(gk):
        if e is not None:
            if e.shape[0] != digit_count(parms, gdbc):
                raise BadStream("Galois keys are not valid for the encryption parameters")
            ctx.set_galois_key(2 * i + 1, e.reshape(-1))
    if sk.size:
        if sk.size != parms.k * parms.n:
            raise BadStream("secret key is not valid for the encryption parameters")
        ctx.set_secret_key(sk)
    client = client_factory(ctx) if client_factory is not None else None
    if client is None:
        from .client import DeviceClient
        client = DeviceClient(ctx)
    client.has_secret_key = bool(sk.size)
    return AtomicSealBfvEncryptedEnvironment(ctx, client)


def save_environments(stream, envs, withPrivateKeys):
    """EncryptedSealBfvEnvironment.Save (EncryptedSealBfvVector.cs:104-126): zip, one stored entry per plaintext prime"""
    with zipfile.ZipFile(stream, "w", compression=zipfile.ZIP_STORED) as z:
        for i, e in enumerate(envs):
            mem = io.BytesIO()
            save_environment(mem, e, withPrivateKeys)
            z.writestr("environment%03d" % i, mem.getvalue())
    return stream


def load_environments(stream, context_factory, client_factory=None):
    """EncryptedSealBfvEnvironment(Stream) (EncryptedSealBfvVector.cs:49-68): entries in name order"""
    envs = []
    with zipfile.ZipFile(stream, "r") as z:
        for name in sorted(z.namelist()):
            if not name.startswith("environment"):
                raise BadStream("Bad stream format. (unexpected entry %s)" % name)
            envs.append(load_environment(io.BytesIO(z.read(name)), context_factory, client_factory))
    if not envs:
        raise BadStream("Bad stream format. (no environments)")
    return envs


# ------------------------------------------------------------------------------------------------ vectors / matrices
def _fmt_double(x):
    """Double.ToString() of .NET Framework (15 significant digits, 'E+XX' exponents)"""
    s = "%.15g" % float(x)
    if "e" in s:
        m, e = s.split("e")
        s = "%sE%s%02d" % (m, e[0], int(e[1:]))
    return s


def _readline(s):
    line = s.readline()
    if line == "":
        raise BadStream("Bad stream format.")
    return line.rstrip("\r\n")


def _expect(s, text):
    if _readline(s) != text:
        raise BadStream("Bad stream format.")


def write_atomic_vector(s, vec, env):
    """AtomicSealBfvEncryptedVector.Write (AtomicSealBfvVector.cs:1273-1302); `s` is a text stream"""
    ctx = env.ctx
    parms = Parameters(ctx.n, ctx.q, ctx.t)
    s.write("<Start EncryptedVector>\n")
    s.write(_fmt_double(vec.Scale) + "\n")
    s.write(("True" if vec.IsSigned else "False") + "\n")
    s.write(vec.Format.name + "\n")
    s.write("%d\n" % vec.Dim)
    mem = io.BytesIO()
    if vec.encData is not None:
        cts = ctx.ct_download(vec.encData.h, vec.encData.first, vec.encData.count)
        s.write("Encrypted\n%d\n" % len(cts))
        for c in cts:
            save_ciphertext(mem, c, parms, size=2)
    else:
        if vec.plainDense is not None:
            plains = list(ctx.pt_download(vec.plainDense.h, vec.plainDense.first, vec.plainDense.count))
        else:
            plains = [np.array([x], dtype=np.uint64) for x in vec.plainSparse]     # constant polynomial, CoeffCount 1
        s.write("Plain\n%d\n" % len(plains))
        for p in plains:
            save_plaintext(mem, p)
    s.write(base64.b64encode(mem.getvalue()).decode("ascii") + "\n")
    s.write("<End EncryptedVector>\n")
    s.flush()


def read_atomic_vector(s, env):
    """AtomicSealBfvEncryptedVector.Read (AtomicSealBfvVector.cs:1304-1345)"""
    from .hewrapper import AtomicSealBfvEncryptedVector, EVectorFormat, _Buf
    ctx = env.ctx
    parms = Parameters(ctx.n, ctx.q, ctx.t)
    _expect(s, "<Start EncryptedVector>")
    try:
        scale = float(_readline(s))
        signed = {"True": True, "False": False}[_readline(s)]
        fmt = EVectorFormat[_readline(s)]
        dim = int(_readline(s))
        mode = _readline(s)
        length = int(_readline(s))
        mem = io.BytesIO(base64.b64decode(_readline(s), validate=True))
    except (KeyError, ValueError) as e:
        raise BadStream("Bad stream format. (%s)" % e)
    vec = AtomicSealBfvEncryptedVector._new(Scale=scale, IsSigned=signed, Format=fmt, Dim=dim)
    if length <= 0:
        raise BadStream("Bad stream format. (empty vector)")
    if mode == "Encrypted":
        cts = []
        for _ in range(length):
            w, size = load_ciphertext(mem, parms, want_ntt_form=False)
            if size != 2:
                raise BadStream("Bad stream format. (ciphertext size %d)" % size)
            cts.append(w)
        vec.encData = _Buf(ctx, "ct", length).view()
        ctx.ct_upload(vec.encData.h, 0, np.stack(cts))
    elif mode == "Plain":
        plains = [load_plaintext(mem, parms) for _ in range(length)]
        if fmt == EVectorFormat.sparse:
            if any(p.size > 1 for p in plains):
                raise BadStream("Bad stream format. (sparse plaintext with more than one coefficient)")
            vec.plainSparse = [int(p[0]) if p.size else 0 for p in plains]
        else:
            full = np.zeros((length, ctx.n), dtype=np.uint64)
            for i, p in enumerate(plains):
                full[i, :p.size] = p
            vec.plainDense = _Buf(ctx, "pt", length).view()
            ctx.pt_upload(vec.plainDense.h, 0, full)
            vec.plainZero = [not p.any() for p in plains]
    else:
        raise BadStream("unknown format")
    if mem.read(1):
        raise BadStream("Bad stream format. (trailing bytes)")
    _expect(s, "<End EncryptedVector>")
    return vec


def write_vector(s, vec, env):
    """EncryptedSealBfvVector.Write (EncryptedSealBfvVector.cs:430-439)"""
    s.write("<Start LargeEncryptedVector>\n")
    s.write(_fmt_double(vec.Scale) + "\n")
    s.write("%d\n" % len(vec.eVectors))
    for a, e in zip(vec.eVectors, env.Environments):
        write_atomic_vector(s, a, e)
    s.write("<End LargeEncryptedVector>\n")
    s.flush()


def read_vector(s, env):
    """EncryptedSealBfvVector.Read (EncryptedSealBfvVector.cs:414-429)"""
    from .hewrapper import EncryptedSealBfvVector
    _expect(s, "<Start LargeEncryptedVector>")
    try:
        scale = float(_readline(s))
        count = int(_readline(s))
    except ValueError as e:
        raise BadStream("Bad stream format. (%s)" % e)
    if count != len(env.Environments):
        raise BadStream("Bad stream format. (%d plaintext primes, environment has %d)" % (count, len(env.Environments)))
    atoms = [read_atomic_vector(s, e) for e in env.Environments]
    _expect(s, "<End LargeEncryptedVector>")
    return EncryptedSealBfvVector._of(atoms, scale)


def write_matrix(s, mat, env):
    """EncryptedSealBfvMatrix.Write (EncryptedSealBfvMatrix.cs:199-208)"""
    s.write("<Start LargeEncryptedMatrix>\n")
    s.write(mat.Format.name + "\n")
    s.write("%d\n" % len(mat.leVectors))
    for v in mat.leVectors:
        write_vector(s, v, env)
    s.write("<End LargeEncryptedMatrix>\n")
    s.flush()


def read_matrix(s, env):
    """EncryptedSealBfvMatrix.Read (EncryptedSealBfvMatrix.cs:182-197)"""
    from .hewrapper import EMatrixFormat, EncryptedSealBfvMatrix
    _expect(s, "<Start LargeEncryptedMatrix>")
    try:
        fmt = EMatrixFormat[_readline(s)]
        count = int(_readline(s))
    except (KeyError, ValueError) as e:
        raise BadStream("Bad stream format. (%s)" % e)
    vecs = [read_vector(s, env) for _ in range(count)]
    _expect(s, "<End LargeEncryptedMatrix>")
    return EncryptedSealBfvMatrix(vecs, env, CopyVectors=False, Format=fmt)
