"""Host-side mirror of the reference's HE wrapper (`HE Wrapper/*.cs`) on top of libcnhip.

Same names, argument meaning and error behaviour as the C# classes so the parity tests read like the reference's
MSTest suite:
  AtomicSealBfvEncryptedEnvironment / AtomicSealBfvEncryptedVector   <- AtomicSealBfvVector.cs
  EncryptedSealBfvEnvironment / EncryptedSealBfvVector               <- EncryptedSealBfvVector.cs (CRT over plaintext primes)
  EncryptedSealBfvMatrix                                             <- EncryptedSealBfvMatrix.cs
  EncryptedSealBfvFactory                                            <- IFactory.cs:240-410
Every `epenv.evaluator.*` call of the reference becomes a (batched) libcnhip call on device-resident ciphertext
arrays.  What stays on the client in the reference (KeyGenerator, Encryptor, Decryptor - SEAL calls at
AtomicSealBfvVector.cs:62-74,1042,1211) is behind the small `ClientCrypto` interface: the evaluator never sees a
secret key.  BatchEncoder (public math) runs on the device (cn_encode / cn_decode).
"""
import enum
import os
import math

import numpy as np


class EVectorFormat(enum.Enum):
    dense = 0
    sparse = 1


class EMatrixFormat(enum.Enum):
    ColumnMajor = 0
    RowMajor = 1



# LITERAL = True: the matrix- and layer-level conveniences of this mirror (RowsDotProduct, MulColumnsByPlain, the batched LLPoolLayer /
# MulManySparse forms) are switched off and every layer takes the per-vector path the reference's UNCHANGED files take - one
# AtomicSealBfvEncryptedVector method call per row / column / map (EncryptedSealBfvMatrix.cs:79-120, LLPoolLayer.cs, LLInterleaveLayer.cs,
# LLDenseLayer.cs).  That is the call sequence the C# twin receives; tools/call_trace.py records it at the C ABI and
# tools/replay_call_trace.cpp replays it from native code (bench.py --workload lola: `unchanged_caller`).  Same ciphertext words.
# What a vector method does INSIDE is the twin's own business on both sides of the switch: AtomicSealBfvEncryptedVector is the file the twin
# replaces (integration/GpuAtomicSealBfvEncryptedVector.cs), and like it this class gathers with cn_copy_many and rotates the vectors of an
# Interleave / the copies of a Duplicate with one cn_rotate_rows_many.
LITERAL = bool(int(os.environ.get("CN_LITERAL_CALLS", "0")))


def set_literal(on):
    global LITERAL
    LITERAL = bool(on)

class ClientCrypto:
    """The client-side SEAL objects of AtomicSealBfvEncryptedEnvironment.SetKeys (AtomicSealBfvVector.cs:62-74):
    KeyGenerator / Encryptor / Decryptor for ONE plaintext modulus."""

    def generate_keys(self, with_galois=True):
        raise NotImplementedError

    def relin_key(self):            # u64 words, layout of include/cnhip.h
        raise NotImplementedError

    def galois_keys(self):          # dict galois_elt -> u64 words
        raise NotImplementedError

    def encrypt(self, plain):       # N plaintext coefficients -> size-2 ciphertext words
        raise NotImplementedError

    def decrypt(self, ct):          # ciphertext words -> N plaintext coefficients
        raise NotImplementedError

    def reference_evaluator(self):
        """The client's OWN evaluator on host words, or None.  AtomicSealBfvEncryptedEnvironment keeps a live SEAL `Evaluator` next to
        the keys (AtomicSealBfvVector.cs:22,66); the start-up self-test (AtomicSealBfvEncryptedEnvironment.SelfTest) runs a handful of
        operations through it and through the device and compares ciphertext WORDS.  Methods, all on flat u64 word arrays:
        multiply_plain(ct, plain_coeffs) (a length-1 plain = constant plaintext), add_plain(ct, plain_coeffs), multiply(a, b) -> size 3,
        relinearize(c3), rotate_rows(ct, steps), rotate_columns(ct)."""
        return None

    def relin_key_coeff_form(self):   # the relinearisation key with every polynomial in COEFFICIENT form (Evaluator.TransformFromNTTInplace), or None
        return None

    def galois_keys_coeff_form(self):
        return None


# ------------------------------------------------------------------------------------------------ device buffers
class _Buf:
    """Ref-counted device array of ciphertexts (kind 'ct') or dense plaintexts (kind 'pt')."""

    def __init__(self, ctx, kind, count, size=2):
        self.ctx, self.kind, self.count, self.size, self.refs = ctx, kind, count, size, 0
        self.h = ctx.ct_alloc(count, size) if kind == "ct" else ctx.pt_alloc(count)

    def view(self, first=0, count=None):
        return _View(self, first, self.count - first if count is None else count)


class _View:
    def __init__(self, buf, first, count):
        self.buf, self.first, self.count = buf, first, count
        buf.refs += 1
        self.live = True

    def release(self):
        if self.live:
            self.live = False
            self.buf.refs -= 1
            if self.buf.refs == 0 and self.buf.h is not None:
                self.buf.ctx.free(self.buf.h)
                self.buf.h = None

    def sub(self, i, count=1):
        return _View(self.buf, self.first + i, count)

    @property
    def h(self):
        return self.buf.h


def _gather(ctx, views):
    """Return (buffer handle, [index per ciphertext], temp view or None) with every ciphertext of `views` in ONE device
    array: in place when they already share a buffer, otherwise packed into a temporary (device-to-device copies)."""
    b0 = views[0].buf
    if all(v.buf is b0 for v in views):
        return b0.h, [v.first + i for v in views for i in range(v.count)], None
    total = sum(v.count for v in views)
    tmp = _Buf(ctx, "ct", total).view()
    if hasattr(ctx, "copy_many"):                          # one launch for the whole gather (cn_copy_many) instead of one per vector
        ctx.copy_many([v.h for v in views for _ in range(v.count)], [v.first + i for v in views for i in range(v.count)], tmp.h, 0)
        return tmp.h, list(range(total)), tmp
    pos, idx = 0, []
    for v in views:
        ctx.copy(v.h, v.first, tmp.h, pos, v.count)
        idx.extend(range(pos, pos + v.count))
        pos += v.count
    return tmp.h, idx, tmp


# ------------------------------------------------------------------------------------------------ CRT fan-out
PARALLEL_PRIMES = os.environ.get("CN_PARALLEL_PRIMES", "0") != "0"
_crt_pool = None


def _fan_out(envs, fn):
    """ForEveryEncryptedVector (EncryptedSealBfvVector.cs:225-236): the reference issues every operation as one Task per
    plaintext prime.  Here each prime is a device context with its own HIP stream and every libcnhip call only ENQUEUES work, so
    one host thread walking the primes in turn already keeps the streams busy; the Task-per-prime form (CN_PARALLEL_PRIMES=1,
    one host thread per prime, ctypes drops the GIL inside libcnhip) was measured SLOWER for single-image LoLa - 27 ms vs 17.7 ms
    per image - the four threads spend their time handing the GIL to each other between ~25 us calls."""
    global _crt_pool
    if len(envs) == 1 or not PARALLEL_PRIMES:
        return [fn(i, e) for i, e in enumerate(envs)]
    if _crt_pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _crt_pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="crt")
    futures = [_crt_pool.submit(fn, i, e) for i, e in enumerate(envs)]
    results, first_error = [], None
    for f in futures:                       # wait for every prime before raising: no task may outlive the call
        try:
            results.append(f.result())
        except Exception as ex:             # noqa: BLE001 - re-raised below
            first_error = first_error or ex
            results.append(None)
    if first_error is not None:
        raise first_error
    return results


# ------------------------------------------------------------------------------------------------ atomic layer
class AtomicSealBfvEncryptedEnvironment:
    """One plaintext modulus = one device context + its evaluation keys (AtomicSealBfvVector.cs:19-206)."""

    def __init__(self, ctx, client=None):
        self.ctx, self.client = ctx, client
        self.plainmodulusValue = ctx.t
        self.ParentFactory = None

    @property
    def SlotCount(self):
        return self.ctx.n

    @property
    def PlaintextCapacity(self):
        return self.ctx.n

    @property
    def Primes(self):
        return [self.plainmodulusValue]

    def GenerateEncryptionKeys(self, with_galois=True):
        """KeyGenerator + SetKeys (AtomicSealBfvVector.cs:62-74,163-173): keys are made by the client, the public
        evaluation keys are uploaded to HBM."""
        self.client.generate_keys(with_galois)
        rk = self.client.relin_key()
        if rk is not None:                       # a device-side client has already installed its keys in the context
            self.ctx.set_relin_key(rk)
        if with_galois:
            for elt, words in self.client.galois_keys().items():
                self.ctx.set_galois_key(elt, words)
        self.SelfTest(with_galois)

    # what the last SelfTest found: None (not run: the client has no evaluator of its own), or a dict
    #   {"ks_xi": 0 | 1, "key_form": "ntt" | "coeff", "ops": [names compared], "tried": [(ks_xi, key_form, first failing op or None), ...],
    #    "warnings": [key-less operations whose words differ from the client's evaluator while their decrypted slots agree]}
    self_test_report = None

    def SelfTest(self, with_galois=True):
        """Start-up agreement check of the drop-in with the evaluator it replaces (the C# twin does the same at the end of
        CreateDevice, integration/GpuAtomicSealBfvEncryptedVector.cs).  Two fresh ciphertexts go through the client's own evaluator and
        through the device: MultiplyPlain (dense and constant), AddPlain, Multiply - then the key-switching operations Relinearize,
        RotateRows(+1), RotateRows(-1), RotateColumns.  WORDS are compared.  Two things about real SEAL 3.2 keys cannot be read off
        the reference (SEAL is an un-vendored dependency): which decomposition convention the key switch uses (cn_set_option "ks_xi")
        and whether the NTT-form key words are in this library's transform order.  So when a key-switching operation disagrees the test
        retries with the other convention, then with coefficient-form keys (cn_load_key form 1: the device transforms them itself),
        and keeps the first combination that reproduces the client's words.  Nothing matches: an exception that names the operation -
        never rc 0 and garbage."""
        ev = self.client.reference_evaluator() if hasattr(self.client, "reference_evaluator") else None
        if ev is None:
            return None
        ctx, n, t = self.ctx, self.ctx.n, self.ctx.t
        # deterministic operands (the encryption randomness is the client's): slot-like values over the whole range of t
        v0 = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(t)
        v1 = (np.arange(n, dtype=np.uint64) * np.uint64(40503) + np.uint64(7)) % np.uint64(t)
        ca, cb = self.client.encrypt(v0), self.client.encrypt(v1)
        h2, h3, o2, pt = ctx.ct_alloc(2), ctx.ct_alloc(1, 3), ctx.ct_alloc(1), ctx.pt_alloc(1)
        try:
            ctx.ct_upload(h2, 0, np.stack([ca, cb]))
            ctx.pt_upload(pt, 0, v1)
            const = int(v1[1]) or 1

            def dev(op):
                if op == "MultiplyPlain":
                    ctx.mul_plain(h2, 0, pt, 0, o2, 0)
                elif op == "MultiplyPlain(constant)":
                    ctx.mul_scalar(h2, 0, [const], o2, 0)
                elif op == "AddPlain":
                    ctx.add_plain(h2, 0, pt, 0, o2, 0)
                elif op == "Multiply":
                    ctx.multiply(h2, 0, h2, 1, h3, 0)
                    return ctx.ct_download(h3, 0, 1, size=3)[0]
                elif op == "Relinearize":
                    ctx.multiply(h2, 0, h2, 1, h3, 0)
                    ctx.relinearize(h3, 0, o2, 0)
                elif op == "RotateRows(1)":
                    ctx.rotate_rows(h2, 0, 1, o2, 0)
                elif op == "RotateRows(-1)":
                    ctx.rotate_rows(h2, 0, -1, o2, 0)
                elif op == "RotateColumns":
                    ctx.rotate_columns(h2, 0, o2, 0)
                return ctx.ct_download(o2, 0, 1)[0]

            c3 = ev.multiply(ca, cb)
            want = {"MultiplyPlain": lambda: ev.multiply_plain(ca, v1), "MultiplyPlain(constant)": lambda: ev.multiply_plain(ca, np.array([const], dtype=np.uint64)),
                    "AddPlain": lambda: ev.add_plain(ca, v1), "Multiply": lambda: c3, "Relinearize": lambda: ev.relinearize(c3),
                    "RotateRows(1)": lambda: ev.rotate_rows(ca, 1), "RotateRows(-1)": lambda: ev.rotate_rows(ca, -1),
                    "RotateColumns": lambda: ev.rotate_columns(ca)}
            want = {k: np.asarray(f(), dtype=np.uint64).reshape(-1) for k, f in want.items()
                    if with_galois or not k.startswith("Rotate")}
            # Operations without keys.  Equal WORDS is what the restatement of SEAL 3.2 predicts; a product whose words differ from the client's but which
            # DECRYPTS to the same slots is still a valid drop-in for these operations (an evaluator may pick another valid BEHZ auxiliary base, another
            # representative in a lift: nothing downstream depends on it - only the key-switch convention is an interoperability property): recorded as a
            # warning in self_test_report, the start-up goes on.  Different slots: fatal, as before.
            warnings = []
            for op in ("MultiplyPlain", "MultiplyPlain(constant)", "AddPlain", "Multiply"):
                got = dev(op)
                if np.array_equal(got, want[op]):
                    continue
                try:
                    same = np.array_equal(np.asarray(self.client.decrypt(got)), np.asarray(self.client.decrypt(want[op])))
                except Exception:
                    same = False
                if not same:
                    raise Exception("libcnhip self-test: %s differs from the client's evaluator (plaintext modulus %d), words AND decrypted slots - the device "
                                    "does not implement this evaluator's arithmetic; no key convention can repair that" % (op, t))
                warnings.append("%s: ciphertext words differ from the client's evaluator, decrypted slots are equal (another valid representative; "
                                "not an interoperability property)" % op)
                # never silent: the operator sees at start-up that this device and the client's evaluator disagree on words
                import sys as _sys
                print("libcnhip self-test warning (plaintext modulus %d): %s" % (t, warnings[-1]), file=_sys.stderr)
            ks_ops = [op for op in ("Relinearize", "RotateRows(1)", "RotateRows(-1)", "RotateColumns") if op in want]

            # Key-switching operations: WORDS again - unless the product itself already differs in words (then Relinearize inherits the difference and only
            # its decrypted slots can be compared; the rotations still compare words)
            by_slots = {"Relinearize"} if any(w.startswith("Multiply:") for w in warnings) else set()

            def first_failure():
                for op in ks_ops:
                    got = dev(op)
                    if op in by_slots:
                        if not np.array_equal(np.asarray(self.client.decrypt(got)), np.asarray(self.client.decrypt(want[op]))):
                            return op
                    elif not np.array_equal(got, want[op]):
                        return op
                return None

            xi0 = ctx.get_option("ks_xi")
            tried = []
            for form in ("ntt", "coeff"):
                if form == "coeff":
                    rk = self.client.relin_key_coeff_form() if hasattr(self.client, "relin_key_coeff_form") else None
                    if rk is None:
                        break
                    ctx.load_key(0, rk, coeff_form=True)
                    for elt, words in (self.client.galois_keys_coeff_form() or {}).items() if with_galois else ():
                        ctx.load_key(1, words, elt=elt, coeff_form=True)
                for xi in (xi0, 1 - xi0):
                    ctx.set_option("ks_xi", xi)
                    bad = first_failure()
                    tried.append((xi, form, bad))
                    if bad is None:
                        self.self_test_report = {"ks_xi": xi, "key_form": form, "ops": list(want), "tried": tried, "warnings": warnings}
                        return self.self_test_report
            ctx.set_option("ks_xi", xi0)
            raise Exception("libcnhip self-test: no key-switch convention reproduces the client's evaluator (plaintext modulus %d); tried (ks_xi, key form, "
                            "first differing operation): %s" % (t, tried))
        finally:
            for h in (h2, h3, o2, pt):
                ctx.free(h)

    def encode(self, values):
        """BatchEncoder.Encode into a fresh device plaintext; returns a pt view."""
        v = _Buf(self.ctx, "pt", 1).view()
        self.ctx.encode(np.asarray(values, dtype=np.uint64), v.h, 0)
        return v

    def mask_plain(self, kind, count):
        """Selection masks of the packing ops (ones over the first `count` slots: Interleave's row-boundary split,
        AtomicSealBfvVector.cs:628-688; a single one in slot `count`: ForceOutputInColumn, :936-945), encoded once per environment and
        kept in HBM: the reference re-encodes them on every call; here that would also put a host upload + synchronisation into the
        middle of an inference (and a recorded evaluation could not contain it).  The caller must NOT release the view."""
        cache = self.__dict__.setdefault("_mask_plains", {})
        key = (kind, int(count))
        if key not in cache:
            if kind == "ones":
                vals = np.ones(count, dtype=np.uint64)
            else:
                vals = np.zeros(count + 1, dtype=np.uint64)
                vals[count] = 1
            cache[key] = self.encode(vals)
        return cache[key]


class AtomicSealBfvEncryptedVector:
    """A vector under ONE plaintext modulus: ceil(Dim/N) ciphertexts (dense) or Dim ciphertexts holding one constant
    polynomial each (sparse) - or the same as plaintexts (AtomicSealBfvVector.cs:303-326)."""

    def __init__(self, v=None, env=None, Scale=1.0, SignedNumbers=True, EncryptData=True, Format=EVectorFormat.dense):
        self.encData = None          # _View over a ciphertext array
        self.plainDense = None       # _View over a device plaintext array (dense format)
        self.plainZero = None        # per dense plaintext: IsZero
        self.plainSparse = None      # list of ints, one constant polynomial each (sparse format)
        self.Scale, self.IsSigned, self.Format, self.Dim = Scale, SignedNumbers, Format, 0
        if v is not None:
            if EncryptData:
                self._Encrypt(v, Format, env)
            else:
                self._Plain(v, Format, env)

    # -- construction -----------------------------------------------------------------------------------------
    @classmethod
    def _new(cls, **kw):
        r = cls()
        for k, val in kw.items():
            setattr(r, k, val)
        return r

    @classmethod
    def Copy(cls, v, env):
        """copy constructor (AtomicSealBfvVector.cs:365-374): deep copy"""
        r = cls._new(Scale=v.Scale, Dim=v.Dim, IsSigned=v.IsSigned, Format=v.Format)
        if v.encData is not None:
            r.encData = _Buf(env.ctx, "ct", v.encData.count).view()
            env.ctx.copy(v.encData.h, v.encData.first, r.encData.h, 0, v.encData.count)
        if v.plainDense is not None:
            r.plainDense = _Buf(env.ctx, "pt", v.plainDense.count).view()
            env.ctx.copy(v.plainDense.h, v.plainDense.first, r.plainDense.h, 0, v.plainDense.count)
            r.plainZero = list(v.plainZero)
        if v.plainSparse is not None:
            r.plainSparse = list(v.plainSparse)
        return r

    def _values(self, v, env):
        """VectorToPlaintext value mapping (AtomicSealBfvVector.cs:1120-1122): round(v*Scale), negatives -> t + x.
        Returns a uint64 numpy array (vectorised: weight matrices have 10^8 entries)."""
        t = env.plainmodulusValue
        a = np.asarray(v)
        if a.dtype == np.uint64:
            return np.ascontiguousarray(a)
        if self.Scale == 0:
            self.Scale = 1
        r = np.rint(np.asarray(v, dtype=np.float64) * self.Scale)
        if r.size and float(np.max(np.abs(r))) >= 2.0 ** 62:
            raise Exception("value does not fit the plaintext modulus")
        w = r.astype(np.int64)
        if self.IsSigned:
            if w.size and (int(w.min()) <= -t or int(w.max()) >= t):
                raise Exception("value does not fit the plaintext modulus")
            w = np.where(w < 0, w + t, w)
        elif w.size and int(w.min()) < 0:
            raise Exception("value does not fit the plaintext modulus")
        return w.astype(np.uint64)

    def _to_plaintexts(self, v, env):
        values = self._values(v, env)
        t = env.plainmodulusValue
        if values.size and int(values.max()) >= t:
            raise Exception("value does not fit the plaintext modulus")
        self.Dim = len(values)
        if self.Format == EVectorFormat.dense:
            slots = env.SlotCount
            blocks = -(-len(values) // slots)
            if blocks == 0:
                raise Exception("empty vector")
            pv = _Buf(env.ctx, "pt", blocks).view()
            padded = np.zeros(blocks * slots, dtype=np.uint64)          # all blocks in ONE BatchEncoder call (cn_encode_batch)
            padded[:len(values)] = values
            padded = padded.reshape(blocks, slots)
            env.ctx.encode_batch(padded, pv.h, 0)
            zero = [not bool(padded[b].any()) for b in range(blocks)]
            return pv, zero, None
        return None, None, [int(x) for x in values]

    def _Plain(self, v, Format, env):
        self.Format = Format
        self.plainDense, self.plainZero, self.plainSparse = self._to_plaintexts(v, env)
        self.encData = None

    def _Encrypt(self, v, Format, env):
        """Encrypt (AtomicSealBfvVector.cs:1202-1232): one Encryptor.Encrypt per plaintext, done by the client."""
        self.Format = Format
        pv, _, sparse = self._to_plaintexts(v, env)
        ctx = env.ctx
        if sparse is not None:
            polys = np.zeros((len(sparse), ctx.n), dtype=np.uint64)
            polys[:, 0] = np.array(sparse, dtype=np.uint64)                # Plaintext(hex) = constant polynomial
            pv = _Buf(ctx, "pt", len(sparse)).view()
            ctx.pt_upload(pv.h, 0, polys)
        self.encData = _Buf(ctx, "ct", pv.count).view()
        if hasattr(env.client, "encrypt_device"):                          # Encryptor on the device: no host round trip
            env.client.encrypt_device(pv.h, 0, pv.count, self.encData.h, 0)
        else:
            plains = ctx.pt_download(pv.h, 0, pv.count)
            ctx.ct_upload(self.encData.h, 0, np.stack([env.client.encrypt(p) for p in plains]))
        pv.release()

    # -- persistence (AtomicSealBfvVector.cs:1273-1345) ------------------------------------------------------------
    def Write(self, stream, env):
        from . import serialization
        serialization.write_atomic_vector(stream, self, env)

    @staticmethod
    def Read(stream, env):
        from . import serialization
        return serialization.read_atomic_vector(stream, env)

    # -- properties -------------------------------------------------------------------------------------------
    @property
    def IsEncrypted(self):
        return self.encData is not None

    def _blocks(self):
        if self.encData is not None:
            return self.encData.count
        return self.plainDense.count if self.plainDense is not None else len(self.plainSparse)

    def RegisterDim(self, dim):
        self.Dim = dim

    def RegisterScale(self, scale):
        self.Scale = scale

    def Dispose(self):
        cached = self.__dict__.pop("_sparse_as_dense", None)
        for v in (self.encData, self.plainDense, cached[1] if cached else None):
            if v is not None:
                v.release()
        self.encData = self.plainDense = self.plainSparse = None

    def _plain_is_zero(self, i):
        return self.plainZero[i] if self.plainDense is not None else self.plainSparse[i] == 0

    # -- HOT LOOP A -------------------------------------------------------------------------------------------
    @staticmethod
    def DenseMatrixBySparseVectorMultiply(denses, sparse, env):
        """AtomicSealBfvVector.cs:434-521: out_block[i] = sum_k denses[k].block[i] * sparse[k]."""
        if len(denses) != sparse.Dim:
            raise Exception("dimensions do not match")
        if sparse.Format != EVectorFormat.sparse:
            raise Exception("expecting a sparse vector")
        if not denses[0].IsEncrypted and not sparse.IsEncrypted:
            raise Exception("at least one parameter has to be encrypted")
        if denses[0].IsSigned != sparse.IsSigned:
            raise Exception("can't mix signed and unsigned messages")
        ctx = env.ctx
        l = denses[0]._blocks()
        K = len(denses)
        res = _Buf(ctx, "ct", l).view()
        if denses[0].IsEncrypted and sparse.IsEncrypted:
            # Multiply + Relinearize per term, then AddMany (:459-465,502)
            terms = _Buf(ctx, "ct", K * l).view()
            for k in range(K):
                ctx.mul_relin(denses[k].encData.h, denses[k].encData.first, sparse.encData.h, sparse.encData.first + k,
                              terms.h, k * l, l, a_stride=1, b_stride=0)
            for i in range(l):
                ctx.add_many(terms.h, [k * l + i for k in range(K)], res.h, i)
            terms.release()
        elif denses[0].IsEncrypted:
            # ct blocks x constant plaintexts.  The twin's form (integration/GpuAtomicSealBfvEncryptedVector.cs): ONE cn_scalar_dot per output block
            # over the K ciphertexts where they lie - no gather copies
            if hasattr(ctx, "scalar_dot"):
                w = np.array(sparse.plainSparse, dtype=np.uint64)
                for i in range(l):
                    ctx.scalar_dot([d.encData.h for d in denses], [d.encData.first + i for d in denses], w, res.h, i)
                return AtomicSealBfvEncryptedVector._new(Format=EVectorFormat.dense, Scale=denses[0].Scale * sparse.Scale, IsSigned=sparse.IsSigned,
                                                         encData=res, Dim=denses[0].Dim)
            # (backends without cn_scalar_dot - the CPU test harness): one scalar GEMM with l outputs over a gathered array
            h, idx, tmp = _gather(ctx, [d.encData for d in denses])
            W = np.tile(np.array(sparse.plainSparse, dtype=np.uint64), (l, 1))
            gidx = np.array([[idx[k * l + i] for k in range(K)] for i in range(l)], dtype=np.int32)
            try:
                ctx.scalar_gemm(h, W, res.h, 0, idx=gidx)
            finally:
                if tmp is not None:
                    tmp.release()
        else:
            # plain dense columns x encrypted sparse entries (:476-485): MultiplyPlain(sparse.enc[k], denses[k].plain[i])
            terms = _Buf(ctx, "ct", K).view()
            for i in range(l):
                used = []
                for k in range(K):
                    if denses[k]._plain_is_zero(i):
                        continue
                    ctx.mul_plain(sparse.encData.h, sparse.encData.first + k, denses[k].plainDense.h, denses[k].plainDense.first + i,
                                  terms.h, k, 1)
                    used.append(k)
                if not used:
                    raise Exception("AddMany of an empty list")
                ctx.add_many(terms.h, used, res.h, i)
            terms.release()
        return AtomicSealBfvEncryptedVector._new(Format=EVectorFormat.dense, Scale=denses[0].Scale * sparse.Scale, IsSigned=sparse.IsSigned,
                                                 encData=res, Dim=denses[0].Dim)

    # -- packing ops (slot movement) ----------------------------------------------------------------------------
    @staticmethod
    def _Inteleave(vecs, shift, outputBlockCount, env):
        """AtomicSealBfvVector.cs:600-722.  vecs: list of single-ciphertext views."""
        ctx = env.ctx
        blockSize = env.SlotCount
        absShift = -shift if shift < 0 else shift
        if shift < 0 and outputBlockCount > 1:
            raise Exception("Negative shifts with multiple output blocks are not implemented yet")
        if absShift > blockSize // 2 and outputBlockCount > 1:
            raise Exception("Shifts of more than half block size with multiple output blocks are not implemented yet")
        if absShift * len(vecs) > blockSize * outputBlockCount:
            raise Exception("not enough room for interleaving")
        n = len(vecs)
        work = _Buf(ctx, "ct", 2 * n).view()               # slot k: v, slot n + k: v2 (the split-off part)
        lower = [[] for _ in range(outputBlockCount)]
        upper = [[] for _ in range(outputBlockCount)]
        half = blockSize // 2

        def ones_mask(count):
            return env.mask_plain("ones", count)

        def geometry(k):
            thisShift = shift * k
            if thisShift < 0:
                thisShift = half + thisShift
            inBlockShift = thisShift % blockSize
            return thisShift, inBlockShift, thisShift // blockSize, (thisShift + absShift) // blockSize

        def steps_of(k):
            """the RotateRows step count of vector k (0: none)"""
            thisShift, inBlockShift, _, _ = geometry(k)
            if inBlockShift == 0:
                return 0
            if inBlockShift + absShift < half:
                return -thisShift
            return -(inBlockShift - half) if inBlockShift >= half else -inBlockShift

        # work[k] = RotateRows(vecs[k], steps_of(k)).  The reference copies and rotates vector by vector (:628-660); here - as in the C# twin - all
        # rotations of the interleave are ONE library call (cn_rotate_rows_many: the hops of the 13 differently rotated vectors run in rounds,
        # the same hops and words), straight from the source array when the vectors share one, behind one gather launch otherwise
        steps = [steps_of(k) for k in range(n)]
        if hasattr(ctx, "rotate_rows_many") and n > 1:
            if all(x.buf is vecs[0].buf for x in vecs):
                ctx.rotate_rows_many(vecs[0].h, [x.first for x in vecs], steps, work.h, list(range(n)))
            else:
                ctx.copy_many([x.h for x in vecs], [x.first for x in vecs], work.h, 0)
                ctx.rotate_rows_many(work.h, list(range(n)), steps, work.h, list(range(n)))
        else:
            for k, src in enumerate(vecs):
                if steps[k] == 0:
                    ctx.copy(src.h, src.first, work.h, k, 1)
                else:
                    ctx.rotate_rows(src.h, src.first, steps[k], work.h, k, 1)
        for k in range(n):
            thisShift, inBlockShift, startBlock, endBlock = geometry(k)
            v, v2 = k, n + k
            if inBlockShift == 0:
                lower[startBlock].append(v)
            elif inBlockShift + absShift < half:
                lower[startBlock].append(v)
            elif inBlockShift >= half:
                if startBlock == endBlock:
                    upper[startBlock].append(v)
                else:
                    upperPartSize = inBlockShift + absShift - blockSize
                    ctx.copy(work.h, v, work.h, v2, 1)
                    p = ones_mask(upperPartSize)
                    ctx.mul_plain(work.h, v, p.h, 0, work.h, v, 1)
                    ctx.sub(work.h, v2, work.h, v, work.h, v2, 1)
                    upper[startBlock].append(v2)
                    lower[endBlock].append(v)
            else:
                upperPartSize = inBlockShift + absShift - half
                if upperPartSize > 0:
                    ctx.copy(work.h, v, work.h, v2, 1)
                    p = ones_mask(upperPartSize)
                    ctx.mul_plain(work.h, v, p.h, 0, work.h, v, 1)
                    ctx.sub(work.h, v2, work.h, v, work.h, v2, 1)
                    upper[startBlock].append(v)
                    lower[startBlock].append(v2)
                else:
                    lower[startBlock].append(v)
        res = _Buf(ctx, "ct", outputBlockCount).view()
        tmp = _Buf(ctx, "ct", 1).view()
        for i in range(outputBlockCount):
            if not lower[i]:
                raise Exception("AddMany of an empty list")
            ctx.add_many(work.h, lower[i], res.h, i)
            if upper[i]:
                ctx.add_many(work.h, upper[i], tmp.h, 0)
                ctx.rotate_columns(tmp.h, 0, tmp.h, 0, 1)
                ctx.add(res.h, i, tmp.h, 0, res.h, i, 1)
        tmp.release()
        work.release()
        return res

    @staticmethod
    def Interleave(vecs, shift, env):
        """AtomicSealBfvVector.cs:729-750"""
        if vecs[0].Format != EVectorFormat.dense:
            raise Exception("Expecting dense vector")
        blockSize = env.SlotCount
        outputBlocks = 1
        if shift > 0:
            outputBlocks = int(math.ceil(vecs[0].Dim * len(vecs) / float(blockSize)))
        subs = [v.encData.sub(0) for v in vecs]
        try:
            enc = AtomicSealBfvEncryptedVector._Inteleave(subs, shift, outputBlocks, env)
        finally:
            for sv in subs:                              # (these views used to stay alive: every Interleave / Stack kept its inputs' arrays)
                sv.release()
        return AtomicSealBfvEncryptedVector._new(encData=enc, Dim=vecs[0].Dim, Scale=vecs[0].Scale, IsSigned=vecs[0].IsSigned,
                                                 Format=EVectorFormat.dense)

    @staticmethod
    def Stack(vecs, env):
        """AtomicSealBfvVector.cs:756-761"""
        res = AtomicSealBfvEncryptedVector.Interleave(vecs, int(vecs[0].Dim), env)
        res.Dim = vecs[0].Dim * len(vecs)
        return res

    # -- HOT LOOP B -------------------------------------------------------------------------------------------
    def _PointwiseMultiplySparseDimOne(self, ev, env):
        """AtomicSealBfvVector.cs:774-810: multiply every block by one constant"""
        ctx = env.ctx
        t = AtomicSealBfvEncryptedVector._new(Scale=self.Scale * ev.Scale, Dim=self.Dim, Format=self.Format, IsSigned=self.IsSigned)
        if self.encData is not None and ev.encData is not None:
            t.encData = _Buf(ctx, "ct", self.encData.count).view()
            ctx.mul_relin(ev.encData.h, ev.encData.first, self.encData.h, self.encData.first, t.encData.h, 0, self.encData.count,
                          a_stride=0, b_stride=1)
            return t
        if self.encData is not None:                 # enc blocks x plain constant
            t.encData = _Buf(ctx, "ct", self.encData.count).view()
            w = ev.plainSparse[0]
            if w == 0:
                raise Exception("plain cannot be zero")
            ctx.mul_scalar(self.encData.h, self.encData.first, np.array([w], dtype=np.uint64), t.encData.h, 0, self.encData.count, broadcast=True)
        else:                                        # plain blocks x encrypted constant
            cnt = self.plainDense.count
            t.encData = _Buf(ctx, "ct", cnt).view()
            for i in range(cnt):
                ctx.mul_plain(ev.encData.h, ev.encData.first, self.plainDense.h, self.plainDense.first + i, t.encData.h, i, 1)
        return t

    def _encrypt_zero_into(self, env, buf_h, index):
        """encryptor.Encrypt(PlainZero) (AtomicSealBfvVector.cs:566,588): a fresh encryption of zero from the client"""
        ctx = env.ctx
        if hasattr(env.client, "encrypt_device"):
            z = _Buf(ctx, "pt", 1).view()
            ctx.pt_upload(z.h, 0, np.zeros((1, ctx.n), dtype=np.uint64))
            env.client.encrypt_device(z.h, 0, 1, buf_h, index)
            z.release()
        else:
            ctx.ct_upload(buf_h, index, np.asarray(env.client.encrypt(np.zeros(ctx.n, dtype=np.uint64)))[None, :])

    def SparseMultiply(self, v, colIndex, env):
        """AtomicSealBfvVector.cs:529-598: every block of this (dense) vector times ELEMENT colIndex of the sparse vector v.
        (No caller in the reference's networks - kept for interface parity)"""
        ev = v
        if colIndex >= ev.Dim:
            raise Exception("index exceeds dimension")
        if ev.Format != EVectorFormat.sparse:
            raise Exception("expecting sparse format")
        if ev.encData is None and self.encData is None:
            raise Exception("at least one argument is expected to be encrypted")
        if self.IsSigned != ev.IsSigned:
            raise Exception("can't mix signed and unsigned numbers.")
        ctx = env.ctx
        t = AtomicSealBfvEncryptedVector._new(Scale=self.Scale * ev.Scale, Dim=self.Dim, Format=EVectorFormat.dense, IsSigned=self.IsSigned)
        if self.encData is not None and ev.encData is not None:
            n = self.encData.count
            t.encData = _Buf(ctx, "ct", n).view()
            ctx.mul_relin(ev.encData.h, ev.encData.first + colIndex, self.encData.h, self.encData.first, t.encData.h, 0, n, a_stride=0, b_stride=1)
            return t
        if self.encData is None:                       # plain dense blocks x one encrypted element
            n = self.plainDense.count
            t.encData = _Buf(ctx, "ct", n).view()
            for i in range(n):
                if self.plainZero[i]:
                    self._encrypt_zero_into(env, t.encData.h, i)
                else:
                    ctx.mul_plain(ev.encData.h, ev.encData.first + colIndex, self.plainDense.h, self.plainDense.first + i, t.encData.h, i, 1)
            return t
        n = self.encData.count                         # encrypted blocks x one plain constant
        t.encData = _Buf(ctx, "ct", n).view()
        w = ev.plainSparse[colIndex]
        if w == 0:
            for i in range(n):
                self._encrypt_zero_into(env, t.encData.h, i)
        else:
            ctx.mul_scalar(self.encData.h, self.encData.first, np.array([w], dtype=np.uint64), t.encData.h, 0, n, broadcast=True)
        return t

    def PointwiseMultiply(self, v, env):
        """AtomicSealBfvVector.cs:813-860"""
        ev = v
        if self.IsSigned != ev.IsSigned:
            raise Exception("Can't mix signed and unsigned numbers.")
        if not self.IsEncrypted and not ev.IsEncrypted:
            raise Exception("multiplying two plaintexts is not implemented")
        if self.Dim == 1 and self.Format == EVectorFormat.sparse:
            return ev._PointwiseMultiplySparseDimOne(self, env)
        if ev.Dim == 1 and ev.Format == EVectorFormat.sparse:
            return self._PointwiseMultiplySparseDimOne(ev, env)
        if self.Dim != v.Dim:
            raise Exception("Dimensions do not match")
        if self.Format != ev.Format:
            raise Exception("Format mismatch")
        ctx = env.ctx
        t = AtomicSealBfvEncryptedVector._new(Scale=self.Scale * ev.Scale, Dim=self.Dim, Format=self.Format, IsSigned=self.IsSigned)
        if self.encData is not None and ev.encData is not None:
            n = ev.encData.count
            t.encData = _Buf(ctx, "ct", n).view()
            ctx.mul_relin(ev.encData.h, ev.encData.first, self.encData.h, self.encData.first, t.encData.h, 0, n)
            return t
        enc = self.encData if self.encData is not None else ev.encData
        pl = self if self.encData is None else ev
        t.encData = _Buf(ctx, "ct", enc.count).view()
        if pl.plainDense is not None:
            ctx.mul_plain(enc.h, enc.first, pl.plainDense.h, pl.plainDense.first, t.encData.h, 0, enc.count)
        else:
            if any(w == 0 for w in pl.plainSparse):
                raise Exception("plain cannot be zero")
            ctx.mul_scalar(enc.h, enc.first, np.array(pl.plainSparse, dtype=np.uint64), t.encData.h, 0, enc.count)
        return t

    # -- HOT LOOP C -------------------------------------------------------------------------------------------
    @staticmethod
    def _RotateRowsAndAdd(ctx, c_h, c_i, steps, agg_h, agg_i, tmp_h, tmp_i):
        """AtomicSealBfvVector.cs:862-868: agg += RotateRows(c, -steps)"""
        ctx.rotate_rows_add(c_h, c_i, -steps, agg_h, agg_i, agg_h, agg_i, 1)      # rotation + AddInplace in one launch chain

    def SumAllSlots(self, env, length=None, ForceOutputInColumn=None, _consume=False):
        """AtomicSealBfvVector.cs:877-955 (length None = Int32.MaxValue = full sum).  _consume: the caller owns this vector as a temporary (the
        product inside DotProduct) - the sum is built in its array instead of in a copy, and this object must not be used or disposed afterwards"""
        INT_MAX = 2 ** 31 - 1
        if length is None:
            length = INT_MAX
        if self.Format != EVectorFormat.dense:
            raise Exception("Expecting dense vector format")
        if length != INT_MAX and ForceOutputInColumn is not None:
            raise Exception("forcing output in a column works only when doing complete sum")
        if self.encData is None:
            raise Exception("SumAllSlots can be applied to encrypted data only")
        if length <= 0:
            raise Exception("Can't sum over less then one element")
        if length == 1:
            return self
        ctx, slots = env.ctx, env.SlotCount
        if _consume and self.encData.count == 1 and self.encData.first == 0 and self.encData.buf.refs == 1:
            res, self.encData = self.encData, None
        else:
            res = _Buf(ctx, "ct", 1).view()        # the sum is built where it is returned (as in the twin: one copy, not three)
            if self.encData.count > 1:
                ctx.add_many(self.encData.h, [self.encData.first + i for i in range(self.encData.count)], res.h, 0)
            else:
                ctx.copy(self.encData.h, self.encData.first, res.h, 0, 1)
            if _consume:
                self.Dispose()
        ctx.sum_slots(res.h, 0, 1, 0 if length >= slots else length)      # column swap + log2 rotate-and-add steps in one call
        if length >= slots // 2:
            length = slots // 2
        if ForceOutputInColumn is not None:
            p = env.mask_plain("slot", ForceOutputInColumn)
            ctx.mul_plain(res.h, 0, p.h, 0, res.h, 0, 1)
            length = 1
        return AtomicSealBfvEncryptedVector._new(IsSigned=self.IsSigned, Scale=self.Scale, Dim=1 if length >= slots // 2 else self.Dim, encData=res,
                                                 Format=EVectorFormat.sparse if length >= slots else EVectorFormat.dense)

    def DotProduct(self, v, env, length=None, ForceOutputInColumn=None):
        """AtomicSealBfvVector.cs:963-977"""
        mul = self.PointwiseMultiply(v, env)
        if (length is None or length > 1) and mul.encData is not None and mul.Format == EVectorFormat.dense:
            return mul.SumAllSlots(env, length, ForceOutputInColumn, _consume=True)       # the product is a temporary: summed in its own array
        res = mul.SumAllSlots(env, length, ForceOutputInColumn)
        if res is not mul:
            mul.Dispose()
        return res

    # -- linear ---------------------------------------------------------------------------------------------
    def Add(self, v, env):
        """AtomicSealBfvVector.cs:983-1024"""
        if self.Scale == 0:
            return v
        if v.Scale == 0:
            return self
        if self.Scale != v.Scale:
            raise Exception("Scales do not match.")
        if self.Dim != v.Dim:
            raise Exception("Dimensions do not match")
        if self.Format != v.Format:
            raise Exception("Format mismatch")
        if self.IsSigned != v.IsSigned:
            raise Exception("can't mix signed and unsigned numbers.")
        if not self.IsEncrypted and not v.IsEncrypted:
            raise Exception("adding two plaintexts is not supported")
        ctx = env.ctx
        t = AtomicSealBfvEncryptedVector._new(Scale=self.Scale, Dim=self.Dim, Format=self.Format, IsSigned=self.IsSigned)
        if self.IsEncrypted and v.IsEncrypted:
            n = self.encData.count
            t.encData = _Buf(ctx, "ct", n).view()
            ctx.add(v.encData.h, v.encData.first, self.encData.h, self.encData.first, t.encData.h, 0, n)
            return t
        enc, pl = (self.encData, v) if self.IsEncrypted else (v.encData, self)
        t.encData = _Buf(ctx, "ct", enc.count).view()
        pv, tmp = pl._dense_plain_view(env)
        ctx.add_plain(enc.h, enc.first, pv.h, pv.first, t.encData.h, 0, pv.count)
        if tmp:
            pv.release()
        return t

    def Subtract(self, v, env):
        """AtomicSealBfvVector.cs:1238-1271"""
        if v.Scale == 0:
            return self
        if self.Scale != v.Scale:
            raise Exception("Scales do not match.")
        if self.Dim != v.Dim:
            raise Exception("Dimensions do not match")
        if self.Format != v.Format:
            raise Exception("Format mismatch")
        if self.IsSigned != v.IsSigned:
            raise Exception("Can't mix signed and unsigned numbers.")
        if not self.IsEncrypted:
            raise Exception("the first argument for subtraction must be encrypted")
        ctx = env.ctx
        n = self.encData.count
        t = AtomicSealBfvEncryptedVector._new(Scale=self.Scale, Dim=self.Dim, Format=self.Format, IsSigned=self.IsSigned)
        t.encData = _Buf(ctx, "ct", n).view()
        if v.IsEncrypted:
            ctx.sub(self.encData.h, self.encData.first, v.encData.h, v.encData.first, t.encData.h, 0, n)
        else:
            pv, tmp = v._dense_plain_view(env)
            ctx.add_plain(self.encData.h, self.encData.first, pv.h, pv.first, t.encData.h, 0, n, subtract=True)
            if tmp:
                pv.release()
        return t

    def _dense_plain_view(self, env):
        """Plaintexts as device polynomials: dense plaintexts as they are; sparse ones as constant polynomials."""
        if self.plainDense is not None:
            return self.plainDense, False
        # sparse plaintexts (biases) are constants of the network: their polynomial form is uploaded once and kept with the vector - no
        # host upload + synchronisation in the middle of every inference (a recorded evaluation could not contain one either)
        key = tuple(int(x) for x in self.plainSparse)
        cached = self.__dict__.get("_sparse_as_dense")
        if cached is None or cached[0] != key or not cached[1].live:
            if cached is not None:
                cached[1].release()
            pv = _Buf(env.ctx, "pt", len(self.plainSparse)).view()
            polys = np.zeros((len(self.plainSparse), env.ctx.n), dtype=np.uint64)
            polys[:, 0] = np.array(self.plainSparse, dtype=np.uint64)
            env.ctx.pt_upload(pv.h, 0, polys)
            self._sparse_as_dense = cached = (key, pv)
        return cached[1], False

    # -- decryption (client side) -------------------------------------------------------------------------------
    def _decrypt_ints(self, env):
        """Unsigned residues per element, exactly as Decrypt / DecryptFullPrecision read them (AtomicSealBfvVector.cs:1030-1110)."""
        ctx = env.ctx
        res = []
        n_items = self._blocks()
        if self.encData is not None:
            if hasattr(env.client, "decrypt_device"):
                plains = env.client.decrypt_device(self.encData.h, self.encData.first, self.encData.count)
            else:
                cts = ctx.ct_download(self.encData.h, self.encData.first, self.encData.count)
                plains = np.stack([env.client.decrypt(c) for c in cts])
        elif self.plainDense is not None:
            plains = ctx.pt_download(self.plainDense.h, self.plainDense.first, self.plainDense.count)
        else:
            plains = None
        if self.Format == EVectorFormat.dense:
            tmp = _Buf(ctx, "pt", n_items).view()
            ctx.pt_upload(tmp.h, 0, np.asarray(plains)[:n_items])
            slots = ctx.decode_batch(tmp.h, 0, n_items)                               # every block in one BatchEncoder.Decode call
            for i in range(n_items):
                left = int(self.Dim) - len(res)
                res.extend(int(x) for x in slots[i][:min(left, slots.shape[1])])
            tmp.release()
        else:
            for i in range(n_items):
                res.append(int(plains[i][0]) if plains is not None else int(self.plainSparse[i]))
        return res

    def DecryptFullPrecision(self, env):
        t = env.plainmodulusValue
        return [(v - t) if (self.IsSigned and v * 2 > t) else v for v in self._decrypt_ints(env)]

    def Decrypt(self, env):
        mod = float(env.plainmodulusValue)
        t = env.plainmodulusValue
        return np.array([((float(v) - mod) if (self.IsSigned and v * 2 > t) else float(v)) / self.Scale for v in self._decrypt_ints(env)])

    # -- misc -------------------------------------------------------------------------------------------------
    @staticmethod
    def GenerateSparseOfArray(encryptedVector, env):
        """AtomicSealBfvVector.cs:1347-1359: first block of every vector becomes one entry of a sparse vector"""
        ctx = env.ctx
        res = _Buf(ctx, "ct", len(encryptedVector)).view()
        if hasattr(ctx, "copy_many"):
            ctx.copy_many([e.encData.h for e in encryptedVector], [e.encData.first for e in encryptedVector], res.h, 0)
        else:
            for i, e in enumerate(encryptedVector):
                ctx.copy(e.encData.h, e.encData.first, res.h, i, 1)
        return AtomicSealBfvEncryptedVector._new(Scale=encryptedVector[0].Scale, Dim=len(encryptedVector), Format=EVectorFormat.sparse,
                                                 IsSigned=encryptedVector[0].IsSigned, encData=res)

    def Duplicate(self, count, env):
        """AtomicSealBfvVector.cs:1370-1408"""
        shift = 1
        while shift < self.Dim:
            shift *= 2
        if self.encData is None:
            raise Exception("Duplicate operates only on encrypted data")
        if self.Format == EVectorFormat.sparse:
            raise Exception("Duplicate operates only on dense vectors")
        slots, ctx = env.SlotCount, env.ctx
        if shift * count > slots:
            raise Exception("Packed vector must fit in a single ciphertext")
        if hasattr(ctx, "rotate_rows_many") and int(count) > 2:
            # every copy is a rotation of the SAME ciphertext (or of its column-swapped form) and they are only added up: the count - 1 rotations
            # are one library call (their hops share rounds), the sum one AddMany - the words of the reference's rotate-and-add chain (modular
            # addition is exact in any order), 6 dispatches instead of 2 per copy
            n = int(count) - 1
            work = _Buf(ctx, "ct", 2 + n).view()    # 0: the vector, 1: its column-swapped form, 2 ..: the rotated copies
            ctx.copy(self.encData.h, self.encData.first, work.h, 0, 1)
            srcs, steps = [], []
            for i in range(1, int(count)):
                target = i * shift
                if target * 2 >= slots:
                    srcs.append(1)
                    target -= slots // 2
                else:
                    srcs.append(0)
                steps.append(-target)
            if 1 in srcs:
                ctx.rotate_columns(self.encData.h, self.encData.first, work.h, 1, 1)
            ctx.rotate_rows_many(work.h, srcs, steps, work.h, [2 + i for i in range(n)])
            res = _Buf(ctx, "ct", 1).view()
            ctx.add_many(work.h, [0] + [2 + i for i in range(n)], res.h, 0)
            work.release()
            return AtomicSealBfvEncryptedVector._new(IsSigned=self.IsSigned, Scale=self.Scale, Dim=count * shift, encData=res, Format=EVectorFormat.dense)
        work = _Buf(ctx, "ct", 3).view()            # 0: res, 1: rotator, 2: tmp
        ctx.copy(self.encData.h, self.encData.first, work.h, 0, 1)
        ctx.copy(self.encData.h, self.encData.first, work.h, 1, 1)
        columnRotated = False
        for i in range(1, int(count)):
            target = i * shift
            if target * 2 >= slots:
                if not columnRotated:
                    columnRotated = True
                    ctx.rotate_columns(self.encData.h, self.encData.first, work.h, 1, 1)
                target -= slots // 2
            self._RotateRowsAndAdd(ctx, work.h, 1, target, work.h, 0, work.h, 2)
        res = _Buf(ctx, "ct", 1).view()
        ctx.copy(work.h, 0, res.h, 0, 1)
        work.release()
        return AtomicSealBfvEncryptedVector._new(IsSigned=self.IsSigned, Scale=self.Scale, Dim=count * shift, encData=res, Format=EVectorFormat.dense)

    def Rotate(self, amount, env):
        """AtomicSealBfvVector.cs:1414-1430"""
        if self.encData is None:
            raise Exception("Rotate operates only on encrypted data")
        if self.Format == EVectorFormat.sparse:
            raise Exception("Rotate operates only on dense vectors")
        res = _Buf(env.ctx, "ct", 1).view()
        env.ctx.rotate_rows(self.encData.h, self.encData.first, amount, res.h, 0, 1)
        return AtomicSealBfvEncryptedVector._new(IsSigned=self.IsSigned, Scale=self.Scale, Dim=self.Dim, encData=res, Format=EVectorFormat.dense)

    def Permute(self, selections, shifts, outputDim, env):
        """AtomicSealBfvVector.cs:1436-1475: sum_i rotL(x * sel_i, shifts[i])"""
        if self.Format != EVectorFormat.dense:
            raise Exception("Permute works only on dense vectors")
        if len(selections) != len(shifts):
            raise Exception("number of selection vectors and number of shifts does not match")
        if self.encData is None:
            raise Exception("can permute only encrypted vectors")
        if self.encData.count > 1:
            raise Exception("can permute only a single block")
        ctx = env.ctx
        work = _Buf(ctx, "ct", 3).view()            # 0: res, 1: t, 2: t3->relin tmp
        first = -1
        for i, s in enumerate(selections):
            if s is None:
                continue
            if first < 0:
                first = i
            if s.Dim != self.Dim:
                raise Exception("dimension of selection vector does not match dimension of data vector")
            if s.Scale != selections[first].Scale:
                raise Exception("scales of all selection vectors should be the same")
            if s.plainDense is not None:
                ctx.mul_plain(self.encData.h, self.encData.first, s.plainDense.h, s.plainDense.first, work.h, 1, 1)
            else:
                # the reference multiplies without relinearising here (:1457) and would then fail to rotate a size-3
                # ciphertext; encrypted selections are therefore relinearised
                ctx.mul_relin(self.encData.h, self.encData.first, s.encData.h, s.encData.first, work.h, 1, 1)
            if i == first:
                ctx.rotate_rows(work.h, 1, shifts[i], work.h, 0, 1)
            else:
                ctx.rotate_rows_add(work.h, 1, shifts[i], work.h, 0, work.h, 0, 1)     # res += rot(x * sel_i)
        if first < 0:
            raise Exception("permuting with no selected values is illigal")
        res = _Buf(ctx, "ct", 1).view()
        ctx.copy(work.h, 0, res.h, 0, 1)
        work.release()
        return AtomicSealBfvEncryptedVector._new(IsSigned=self.IsSigned, Scale=self.Scale * selections[first].Scale, Dim=outputDim, encData=res,
                                                 Format=EVectorFormat.dense)


# ------------------------------------------------------------------------------------------------ CRT layer
class EncryptedSealBfvEnvironment:
    """All plaintext-prime channels + CRT coefficients (EncryptedSealBfvVector.cs:17-149)."""

    def __init__(self, environments, ParentFactory=None):
        self.Environments = environments
        self.ParentFactory = ParentFactory
        primes = [e.plainmodulusValue for e in environments]
        self.bigFactor = 1
        for p in primes:
            self.bigFactor *= p
        # PreCompute (:79-90): coef_i = (M/p_i) * ((M/p_i)^-1 mod p_i)
        self.preComputedCoefficients = [(self.bigFactor // p) * pow((self.bigFactor // p) % p, -1, p) for p in primes]

    @property
    def Primes(self):
        return [e.plainmodulusValue for e in self.Environments]


class EncryptedSealBfvVector:
    """Values split over the plaintext primes; every op fans out to eVectors[i] (EncryptedSealBfvVector.cs:150-573)."""

    def __init__(self, v=None, env=None, Scale=1.0, EncryptData=True, Format=EVectorFormat.dense, integers=None):
        self.eVectors = None
        self.Scale = Scale
        self.IsSigned = True
        if v is not None or integers is not None:
            primes = [e.plainmodulusValue for e in env.Environments]
            res = None
            if integers is None:
                r = np.rint(np.asarray(v, dtype=np.float64) * Scale)              # SplitBigNumbers (:352-365)
                if r.size == 0 or float(np.max(np.abs(r))) < 2.0 ** 62:
                    w = r.astype(np.int64)                                        # vectorised: (X mod M) mod p = X mod p
                    res = [np.mod(w, p).astype(np.uint64) for p in primes]
                else:
                    integers = [int(x) for x in r]
            if res is None:
                z = [int(x) + env.bigFactor if int(x) < 0 else int(x) for x in integers]
                res = [np.array([x % p for x in z], dtype=np.uint64) for p in primes]
            self.eVectors = [AtomicSealBfvEncryptedVector(res[i], e, Scale=1, SignedNumbers=False, EncryptData=EncryptData, Format=Format)
                             for i, e in enumerate(env.Environments)]

    @classmethod
    def _of(cls, vecs, Scale=1.0):
        r = cls()
        r.eVectors, r.Scale = vecs, Scale
        return r

    @classmethod
    def Copy(cls, v, env):
        return cls._of([AtomicSealBfvEncryptedVector.Copy(x, e) for x, e in zip(v.eVectors, env.Environments)], v.Scale)

    Dim = property(lambda self: 0 if self.eVectors is None else self.eVectors[0].Dim)
    IsEncrypted = property(lambda self: False if self.eVectors is None else self.eVectors[0].IsEncrypted)
    Format = property(lambda self: self.eVectors[0].Format)
    BlockSize = property(lambda self: None)

    def Dispose(self):
        if self.eVectors is not None:
            for v in self.eVectors:
                if v is not None:
                    v.Dispose()
        self.eVectors = None

    def RegisterScale(self, scale):
        self.Scale = scale

    def RegisterDim(self, dim):
        for v in self.eVectors:
            v.RegisterDim(dim)

    def _each(self, fn, env):
        return _fan_out(env.Environments, fn)

    @staticmethod
    def Interleave(vecs, shift, env):
        return EncryptedSealBfvVector._of([AtomicSealBfvEncryptedVector.Interleave([v.eVectors[i] for v in vecs], shift, e)
                                           for i, e in enumerate(env.Environments)], vecs[0].Scale)

    @staticmethod
    def Stack(vecs, env):
        return EncryptedSealBfvVector._of([AtomicSealBfvEncryptedVector.Stack([v.eVectors[i] for v in vecs], e)
                                           for i, e in enumerate(env.Environments)], vecs[0].Scale)

    def Add(self, v, env):
        if self.Scale == 0:
            return v
        if v.Scale == 0:
            return self
        if self.Scale != v.Scale:
            raise Exception("Scales do not match.")
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].Add(v.eVectors[i], e), env), self.Scale)

    def Subtract(self, v, env):
        if v.Scale == 0:
            return self
        if self.Scale != v.Scale:
            raise Exception("Scales do not match.")
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].Subtract(v.eVectors[i], e), env), self.Scale)

    @staticmethod
    def GenerateSpareOfArray(SparseVectors, env):
        return EncryptedSealBfvVector._of([AtomicSealBfvEncryptedVector.GenerateSparseOfArray([v.eVectors[i] for v in SparseVectors], e)
                                           for i, e in enumerate(env.Environments)], SparseVectors[0].Scale)

    @staticmethod
    def DenseMatrixBySparseVectorMultiply(denses, sparse, env):
        return EncryptedSealBfvVector._of([AtomicSealBfvEncryptedVector.DenseMatrixBySparseVectorMultiply([d.eVectors[i] for d in denses],
                                                                                                           sparse.eVectors[i], e)
                                           for i, e in enumerate(env.Environments)], denses[0].Scale * sparse.Scale)

    def PointwiseMultiply(self, v, env):
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].PointwiseMultiply(v.eVectors[i], e), env), self.Scale * v.Scale)

    def SumAllSlots(self, env, length=None):
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].SumAllSlots(e, length), env), self.Scale)

    def DotProduct(self, v, env, length=None, ForceOutputInColumn=None):
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].DotProduct(v.eVectors[i], e, length, ForceOutputInColumn), env),
                                          self.Scale * v.Scale)

    def Duplicate(self, count, env):
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].Duplicate(count, e), env), self.Scale)

    def Rotate(self, amount, env):
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].Rotate(amount, e), env), self.Scale)

    def Permute(self, selections, shifts, outputDim, env):
        I = [i for i, s in enumerate(selections) if s is not None]
        sel, sh = [selections[i] for i in I], [shifts[i] for i in I]
        return EncryptedSealBfvVector._of(self._each(lambda i, e: self.eVectors[i].Permute([x.eVectors[i] for x in sel], sh, outputDim, e), env), self.Scale)

    def Write(self, stream, env):
        """EncryptedSealBfvVector.cs:430-439"""
        from . import serialization
        serialization.write_vector(stream, self, env)

    @staticmethod
    def Read(stream, env):
        """EncryptedSealBfvVector.cs:414-429"""
        from . import serialization
        return serialization.read_vector(stream, env)

    def _join(self, split, env, signed=True):
        """JoinSplitNumbers (EncryptedSealBfvVector.cs:381-411)"""
        out = []
        for j in range(len(split[0])):
            x = sum(int(split[i][j]) * env.preComputedCoefficients[i] for i in range(len(split))) % env.bigFactor
            if signed and x * 2 > env.bigFactor:
                x -= env.bigFactor
            out.append(x)
        return out

    def DecryptFullPrecision(self, env):
        return self._join([v.DecryptFullPrecision(e) for v, e in zip(self.eVectors, env.Environments)], env, self.IsSigned)

    def Decrypt(self, env):
        ints = self._join([v._decrypt_ints(e) for v, e in zip(self.eVectors, env.Environments)], env)
        return np.array([float(x) / self.Scale for x in ints])


# ------------------------------------------------------------------------------------------------ matrix
class EncryptedSealBfvMatrix:
    """Array of CRT vectors, column- or row-major (EncryptedSealBfvMatrix.cs)."""

    def __init__(self, columns=None, env=None, CopyVectors=True, Format=EMatrixFormat.ColumnMajor):
        self.Format = Format
        self.DataDisposedExternaly = False
        self.leVectors = None
        if columns is not None:
            if any(c.Dim != columns[0].Dim for c in columns):
                raise Exception("all columns of a matrix should have the same size")
            self.leVectors = [EncryptedSealBfvVector.Copy(c, env) for c in columns] if CopyVectors else list(columns)

    RowCount = property(lambda self: len(self.leVectors) if self.Format == EMatrixFormat.RowMajor else self.leVectors[0].Dim)
    ColumnCount = property(lambda self: len(self.leVectors) if self.Format == EMatrixFormat.ColumnMajor else self.leVectors[0].Dim)
    Scale = property(lambda self: self.leVectors[0].Scale)
    IsEncrypted = property(lambda self: all(v.IsEncrypted for v in self.leVectors))

    def Dispose(self):
        for cache in ("_rowpt", "_rowmask"):
            for buf in self.__dict__.pop(cache, {}).values():
                buf.release()
        if self.leVectors is not None and not self.DataDisposedExternaly:
            for v in self.leVectors:
                if v is not None:
                    v.Dispose()
        self.leVectors = None

    def Decrypt(self, env):
        vecs = [v.Decrypt(env) for v in self.leVectors]
        return np.stack(vecs, axis=0) if self.Format == EMatrixFormat.RowMajor else np.stack(vecs, axis=1)

    def Mul(self, v, env, ForceDenseFormat=False):
        """EncryptedSealBfvMatrix.cs:70-121"""
        if self.Format == EMatrixFormat.ColumnMajor:
            if ForceDenseFormat:
                raise Exception("Forcing dense format is available only in RowMajor mode")
            return EncryptedSealBfvVector.DenseMatrixBySparseVectorMultiply(self.leVectors, v, env)
        if self._can_batch_rows(v):
            if ForceDenseFormat:
                return self.RowsDotProduct(v, env, ForceOutputInColumns=True)
            temp = self.RowsDotProduct(v, env)
            res = EncryptedSealBfvVector.GenerateSpareOfArray(temp, env)
            for t in temp:
                t.Dispose()
            return res
        if not ForceDenseFormat:
            temp = [row.DotProduct(v, env) for row in self.leVectors]
            res = EncryptedSealBfvVector.GenerateSpareOfArray(temp, env)
            for t in temp:
                t.Dispose()
            return res
        total = None
        for colIndex, row in enumerate(self.leVectors):
            t = row.DotProduct(v, env, ForceOutputInColumn=colIndex)
            if total is None:
                total = t
            else:
                s = total.Add(t, env)
                total.Dispose()
                t.Dispose()
                total = s
        total.RegisterDim(len(self.leVectors))
        if total.Format != EVectorFormat.dense:
            raise Exception("Internal probloem: expecting the output to be dense")
        return total

    def _check(self, m):
        if m.Format != self.Format:
            raise Exception("Format mismatch")
        if m.RowCount != self.RowCount:
            raise Exception("Row count mismatch")
        if m.ColumnCount != self.ColumnCount:
            raise Exception("Column count mismatch")

    def Add(self, m, env):
        self._check(m)
        r = EncryptedSealBfvMatrix(Format=self.Format)
        r.leVectors = [a.Add(b, env) for a, b in zip(self.leVectors, m.leVectors)]
        return r

    def ElementWiseMultiply(self, m, env):
        """EncryptedSealBfvMatrix.cs:140-154; on the GPU all columns of one prime go through ONE cn_mul_relin launch chain."""
        self._check(m)
        r = EncryptedSealBfvMatrix(Format=self.Format)
        cols = len(self.leVectors)
        batched = all(a.IsEncrypted and b.IsEncrypted and a.Format == b.Format and a.Dim == b.Dim for a, b in zip(self.leVectors, m.leVectors))
        if not batched:
            r.leVectors = [a.PointwiseMultiply(b, env) for a, b in zip(self.leVectors, m.leVectors)]
            return r
        out = [[None] * len(env.Environments) for _ in range(cols)]

        def one_prime(i, e):
            ctx = e.ctx
            av = [c.eVectors[i] for c in self.leVectors]
            bv = [c.eVectors[i] for c in m.leVectors]
            for x, y in zip(av, bv):
                if x.IsSigned != y.IsSigned:
                    raise Exception("Can't mix signed and unsigned numbers.")
            ha, ia, ta = _gather(ctx, [x.encData for x in av])
            if m is self:
                hb, ib, tb = ha, ia, None
            else:
                hb, ib, tb = _gather(ctx, [y.encData for y in bv])
            total = len(ia)
            res = _Buf(ctx, "ct", total)
            contiguous = ia == list(range(ia[0], ia[0] + total)) and ib == list(range(ib[0], ib[0] + total))
            if contiguous:
                ctx.mul_relin(ha, ia[0], hb, ib[0], res.h, 0, total)
            else:
                for j in range(total):
                    ctx.mul_relin(ha, ia[j], hb, ib[j], res.h, j, 1)
            pos = 0
            for c, x in enumerate(av):
                cnt = x.encData.count
                out[c][i] = AtomicSealBfvEncryptedVector._new(Scale=x.Scale * bv[c].Scale, Dim=x.Dim, Format=x.Format, IsSigned=x.IsSigned,
                                                              encData=res.view(pos, cnt))
                pos += cnt
            for t in (ta, tb):
                if t is not None:
                    t.release()
        _fan_out(env.Environments, one_prime)
        r.leVectors = [EncryptedSealBfvVector._of(out[c], self.leVectors[c].Scale * m.leVectors[c].Scale) for c in range(cols)]
        return r

    def GetColumn(self, columnNumber):
        if columnNumber >= len(self.leVectors):
            raise Exception("Column does not exist")
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Columns can be extracted only from a column major matrix")
        return self.leVectors[columnNumber]

    def GetRow(self, rowNumber):
        if rowNumber >= len(self.leVectors):
            raise Exception("Row does not exist")
        if self.Format != EMatrixFormat.RowMajor:
            raise Exception("Rows can be extracted only from a row major matrix")
        return self.leVectors[rowNumber]

    def SetColumn(self, columnNumber, vector):
        if columnNumber >= len(self.leVectors):
            raise Exception("Column does not exist")
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Columns can be set only from a column major matrix")
        old = self.leVectors[columnNumber]
        if vector.Dim != old.Dim:
            raise Exception("dimension of vector does not match the dimension of the vector it is replacing")
        if vector.Scale != old.Scale:
            raise Exception("Scale of vector does not match the scale of the vector it is replacing")
        if vector.IsEncrypted != old.IsEncrypted:
            raise Exception("can't exchange encrypted and not encrypted vectors")
        self.leVectors[columnNumber] = vector

    def RegisterScale(self, scale):
        for v in self.leVectors:
            v.RegisterScale(scale)

    def Write(self, stream, env):
        """EncryptedSealBfvMatrix.cs:199-208"""
        from . import serialization
        serialization.write_matrix(stream, self, env)

    @staticmethod
    def Read(stream, env):
        """EncryptedSealBfvMatrix.cs:182-197"""
        from . import serialization
        return serialization.read_matrix(stream, env)

    def ConvertToColumnVector(self, env):
        return EncryptedSealBfvVector.Stack(self.leVectors, env)

    def Interleave(self, shift, env):
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Expecting ColumnMajor matrix")
        return EncryptedSealBfvVector.Interleave(self.leVectors, shift, env)

    def MulColumnsByPlain(self, plain, env):
        """Every column PointwiseMultiply(plain) with ONE MultiplyPlain launch chain per plaintext prime (LLInterleaveLayer masks all
        its columns with the same selection vector, LLInterleaveLayer.cs:40-47).  Same words as the per-column loop."""
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Expecting ColumnMajor matrix")
        cols = self.leVectors
        ok = (not LITERAL) and all(c.IsEncrypted and c.Format == EVectorFormat.dense and all(a.encData.count == 1 for a in c.eVectors) and c.Dim == cols[0].Dim
                 for c in cols) and (not plain.IsEncrypted) and plain.Format == EVectorFormat.dense and plain.Dim == cols[0].Dim \
            and all(a.plainDense is not None and a.plainDense.count == 1 and not a.plainZero[0] for a in plain.eVectors)
        if not ok:
            r = EncryptedSealBfvMatrix(Format=self.Format)
            r.leVectors = [c.PointwiseMultiply(plain, env) for c in cols]
            return r

        def one_prime(i, e):
            ctx = e.ctx
            h, idx, tmp = _gather(ctx, [c.eVectors[i].encData for c in cols])
            res = _Buf(ctx, "ct", len(cols))
            pd = plain.eVectors[i].plainDense
            try:
                if idx == list(range(idx[0], idx[0] + len(idx))):
                    ctx.mul_plain(h, idx[0], pd.h, pd.first, res.h, 0, len(cols), pt_stride=0)
                else:
                    for j, src in enumerate(idx):
                        ctx.mul_plain(h, src, pd.h, pd.first, res.h, j, 1)
            finally:
                if tmp is not None:
                    tmp.release()
            return res
        per_prime = _fan_out(env.Environments, one_prime)
        vecs = []
        for j, c in enumerate(cols):
            atoms = [AtomicSealBfvEncryptedVector._new(Scale=c.eVectors[i].Scale * plain.eVectors[i].Scale, Dim=c.eVectors[i].Dim, Format=EVectorFormat.dense,
                                                       IsSigned=c.eVectors[i].IsSigned, encData=per_prime[i].view(j, 1)) for i in range(len(env.Environments))]
            vecs.append(EncryptedSealBfvVector._of(atoms, c.Scale * plain.Scale))
        r = EncryptedSealBfvMatrix(Format=self.Format)
        r.leVectors = vecs
        return r

    # ---- batched HOT LOOP C: every row of a RowMajor plaintext matrix against one packed ciphertext ---------------------
    def _row_plaintexts(self, i, env_i):
        """contiguous device array with the (single-block, dense) plaintext of every row for prime i - built once"""
        cache = self.__dict__.setdefault("_rowpt", {})
        if i not in cache:
            rows = [r.eVectors[i] for r in self.leVectors]
            buf = _Buf(env_i.ctx, "pt", len(rows)).view()
            for k, r in enumerate(rows):
                env_i.ctx.copy(r.plainDense.h, r.plainDense.first, buf.h, k, 1)
            cache[i] = buf
        return cache[i]

    def _can_batch_rows(self, v):
        return (not LITERAL and self.Format == EMatrixFormat.RowMajor and v.IsEncrypted and v.Format == EVectorFormat.dense
                and all(a.encData.count == 1 for a in v.eVectors)
                and all((not r.IsEncrypted) and r.Format == EVectorFormat.dense and r.Dim == v.Dim
                        and all(a.plainDense is not None and a.plainDense.count == 1 and not a.plainZero[0] for a in r.eVectors)
                        for r in self.leVectors))

    def RowsDotProduct(self, v, env, length=None, ForceOutputInColumns=False, bias=None):
        """[row_r . v for every row r] with ONE launch chain per plaintext prime: the packed ciphertext is replicated R times,
        multiplied by the R row plaintexts (cn_mul_plain, count R) and reduced by the rotate-and-add tree of
        AtomicSealBfvEncryptedVector.SumAllSlots (:888-955) applied to all R ciphertexts at once (cn_rotate_rows / cn_add, count R).
        Same ciphertext words as R sequential DotProduct calls.  Returns the list of R result vectors (or, with
        ForceOutputInColumns, their mask-accumulated dense sum as a single vector, EncryptedSealBfvMatrix.cs:91-120);
        `bias`: optional RowMajor plaintext matrix added row-wise (LLPackedDenseLayer.cs:72)."""
        INT_MAX = 2 ** 31 - 1
        R = len(self.leVectors)
        full = length is None

        def one_prime(i, e):
            ctx, slots = e.ctx, e.SlotCount
            src = v.eVectors[i].encData
            pts = self._row_plaintexts(i, e)
            work = _Buf(ctx, "ct", R)
            wv = work.view()
            ln = INT_MAX if full else int(length)
            if ln <= 0:
                raise Exception("Can't sum over less then one element")
            # replicate + MultiplyPlain + the SumAllSlots rotate-and-add tree for all R rows: ONE library call per prime
            ctx.rowdot_batch(src.h, src.first, pts.h, pts.first, R, 0 if ln >= slots else ln, work.h, 0)
            if ln >= slots // 2:
                ln = slots // 2
            dim = 1 if ln >= slots // 2 else v.Dim
            fmt = EVectorFormat.sparse if ln >= slots else EVectorFormat.dense
            if ForceOutputInColumns:
                mcache = self.__dict__.setdefault("_rowmask", {})
                if i not in mcache:                                  # one-hot masks e_r (SumAllSlots ForceOutputInColumn, :936-945)
                    masks = _Buf(ctx, "pt", R).view()
                    ctx.encode_batch(np.eye(R, dtype=np.uint64), masks.h, 0)          # e_0 .. e_{R-1} in one call
                    mcache[i] = masks
                masks = mcache[i]
                ctx.mul_plain(work.h, 0, masks.h, 0, work.h, 0, R)
                tot = _Buf(ctx, "ct", 1)
                ctx.add_many(work.h, list(range(R)), tot.h, 0)
                wv.release()
                work, wv = tot, tot.view()
            elif bias is not None:
                bp = bias._row_plaintexts(i, e)
                ctx.add_plain(work.h, 0, bp.h, bp.first, work.h, 0, R)
            return work, wv, dim, fmt
        done = _fan_out(env.Environments, one_prime)
        per_prime = [(w, x) for (w, x, _, _) in done]
        out_dim, out_fmt = done[0][2], done[0][3]
        signed = v.eVectors[0].IsSigned
        if ForceOutputInColumns:
            atoms = [AtomicSealBfvEncryptedVector._new(Scale=1, Dim=R, Format=EVectorFormat.dense, IsSigned=signed, encData=wv)
                     for (_, wv) in per_prime]
            return EncryptedSealBfvVector._of(atoms, self.leVectors[0].Scale * v.Scale)
        res = []
        for r in range(R):
            atoms = [AtomicSealBfvEncryptedVector._new(Scale=1, Dim=out_dim, Format=out_fmt, IsSigned=signed, encData=work.view(r, 1))
                     for (work, _) in per_prime]
            res.append(EncryptedSealBfvVector._of(atoms, self.leVectors[r].Scale * v.Scale))
        for (_, wv) in per_prime:
            wv.release()
        return res

    # ---- batched HOT LOOP A for a whole PoolLayer (PoolLayer.cs:149-229 issues one Mul per output) -----------------
    def MulManySparse(self, gather, weights, bias, out_scale, env, cache=None, bias_vectors=None):
        """out[o] = sum_k weights[o][k] * column[gather[o][k]] + bias[o]  for all outputs in one scalar GEMM per prime.
        gather: int32 [O,K] (-1 = padded tap), weights: integer rows (scaled, signed), bias: integers or None.
        Equivalent to O calls of Mul(sparse plain weight window) followed by Add(dense plain bias).  `cache`: a dict owned by the
        calling layer - the planned GEMM (weight tiles, gather tables, bias plaintexts in HBM) is kept there per plaintext prime and
        only launched on the next inference (cn_gemm_plan_create / cn_gemm_plan_apply).  `bias_vectors`: instead of integer
        constants, one dense plaintext vector per output (LLPoolLayer adds `hot * bias`, LLPoolLayer.cs:128-134)."""
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Expecting ColumnMajor matrix")
        O = len(weights)

        def one_prime(i, e):
            ctx, p = e.ctx, e.plainmodulusValue
            cols = [c.eVectors[i] for c in self.leVectors]
            if any(c.encData is None or c.encData.count != 1 for c in cols):
                raise Exception("batched PoolLayer expects single-block encrypted columns")
            h, idx, tmp = _gather(ctx, [c.encData for c in cols])
            res = _Buf(ctx, "ct", O)
            key = (i, tuple(idx), None if bias is None else tuple(int(b) for b in bias), bias_vectors is not None)   # the bias depends on the input scale
            if cache is not None and key in cache and hasattr(ctx, "gemm_apply"):
                try:
                    ctx.gemm_apply(cache[key][0], h, res.h, 0)
                finally:
                    if tmp is not None:
                        tmp.release()
                return res
            g = np.asarray(gather, dtype=np.int64)
            gidx = np.where(g >= 0, np.asarray(idx, dtype=np.int64)[np.maximum(g, 0)], -1).astype(np.int32)
            W = np.array([[int(x) % p for x in row] for row in weights], dtype=np.uint64)
            bh, bidx = 0, None
            if bias_vectors is not None:
                bp = _Buf(ctx, "pt", O).view()
                for o_, bv in enumerate(bias_vectors):
                    pd = bv.eVectors[i].plainDense
                    ctx.copy(pd.h, pd.first, bp.h, o_, 1)
                bh, bidx = bp.h, np.arange(O, dtype=np.int32)
            elif bias is not None:
                bvals = [int(b) % p for b in bias]
                uniq = sorted(set(bvals))
                pos = {v: j for j, v in enumerate(uniq)}
                bp = _Buf(ctx, "pt", len(uniq)).view()
                polys = np.zeros((len(uniq), ctx.n), dtype=np.uint64)
                polys[:, 0] = np.array(uniq, dtype=np.uint64)      # Encode(constant vector) = constant polynomial
                ctx.pt_upload(bp.h, 0, polys)
                bh, bidx = bp.h, np.array([pos[v] for v in bvals], dtype=np.int32)
            if cache is not None and hasattr(ctx, "gemm_plan"):
                try:
                    plan = ctx.gemm_plan(W, idx=gidx, bias_pt=bh, bias_idx=bidx)
                    cache[key] = (plan, bp if (bias is not None or bias_vectors is not None) else None)      # the plan references the bias plaintexts: keep both
                    ctx.gemm_apply(plan, h, res.h, 0)
                finally:
                    if tmp is not None:
                        tmp.release()
                return res
            try:
                ctx.scalar_gemm(h, W, res.h, 0, idx=gidx, bias_pt=bh, bias_idx=bidx)
            finally:
                if tmp is not None:
                    tmp.release()
                if bias is not None or bias_vectors is not None:
                    bp.release()
            return res
        per_prime = _fan_out(env.Environments, one_prime)
        dim = self.leVectors[0].Dim
        vecs = []
        for o in range(O):
            atoms = [AtomicSealBfvEncryptedVector._new(Scale=1, Dim=dim, Format=EVectorFormat.dense, IsSigned=False, encData=per_prime[i].view(o, 1))
                     for i in range(len(env.Environments))]
            vecs.append(EncryptedSealBfvVector._of(atoms, out_scale))
        r = EncryptedSealBfvMatrix(Format=EMatrixFormat.ColumnMajor)
        r.leVectors = vecs
        return r


# ------------------------------------------------------------------------------------------------ factory
class CapturedEvaluation:
    """A recorded evaluation (one HIP graph per plaintext-prime context, `cn_graph_begin/end/launch`): the launch-bound chain of a
    single-image inference - LoLa issues ~235 small launches per prime (`LoLaCryptonets.cs:236-278`) - replayed with one launch per
    prime.  `fn(inputs)` is the evaluation (layer `Apply` chain) on encrypted matrices; it must have run once before on data of the same
    shapes (temporaries then come out of the handle pools).  `run(new_inputs)` writes the new ciphertext words into the buffers the
    recording read, launches the graphs and returns the matrix the recording produced (same object every time: decrypt or copy it
    before the next run).  Everything the recording created stays alive until `Dispose`."""

    def __init__(self, env, fn, inputs):
        self.env, self.inputs = env, list(inputs)
        ctxs = [e.ctx for e in env.Environments]
        import gc
        gc.collect()                                  # temporaries of the rehearsal that only the collector frees go back to the pools first
        self.graphs, begun, failure = [], [], None
        try:
            for c in ctxs:
                c.graph_begin()
                begun.append(c)
            self.result = fn(*self.inputs)
        except BaseException as ex:                   # noqa: BLE001 - re-raised below, after every context has left capture mode
            failure = ex
        for c in begun:                               # every context that entered capture mode leaves it, whatever happened
            try:
                self.graphs.append((c, c.graph_end()))
            except Exception as ex:                   # noqa: BLE001
                failure = failure or ex
        if failure is not None:                       # a failed recording keeps nothing: its graphs would pin the scratch arenas
            for c, g in self.graphs:
                c.free(g)
            self.graphs = []
            raise failure
        self.graphs = [g for _, g in self.graphs]

    @staticmethod
    def _views(m):
        return [[a.encData for a in v.eVectors] for v in m.leVectors]

    def run(self, *new_inputs):
        for dst, src in zip(self.inputs, new_inputs):
            if src is dst:
                continue
            for dv, sv in zip(self._views(dst), self._views(src)):
                for e, d, s_ in zip(self.env.Environments, dv, sv):
                    if d.count != s_.count:
                        raise Exception("a captured evaluation takes inputs of the shapes it was recorded with")
                    e.ctx.copy(s_.h, s_.first, d.h, d.first, d.count)
        for e, g in zip(self.env.Environments, self.graphs):
            e.ctx.graph_launch(g)
        return self.result

    def Dispose(self):
        for e, g in zip(self.env.Environments, self.graphs):
            e.ctx.free(g)
        self.graphs = []


class EncryptedSealBfvFactory:
    """IFactory.cs:240-410.  `client_factory(t, n, q, dbc, gdbc)` builds the client-side SEAL objects for one plaintext
    prime; the device contexts are libcnhip contexts (`context_factory` is only overridden by the CPU test harness)."""
    DefaultDecompositionBitCount = 10
    DefaultGaloisDecompositionBitCount = 20

    def __init__(self, primes=None, n=4096, DecompositionBitCount=10, GaloisDecompositionBitCount=20, SmallModulusCount=-1,
                 client_factory=None, context_factory=None, device=0, galois=True, client_seed=None, device_client_factory=None):
        """client_seed: None (default) = keys and encryption randomness from the OS entropy source; an integer makes the default
        DeviceClient reproducible (tests only - whoever knows it can regenerate the secret key).  device_client_factory(ctx, t): a client
        that lives on the context (e.g. client.SharedKeyDeviceClient: one rank generates, the others receive the keys by RCCL broadcast)."""
        if primes is None:
            primes = [40961, 65537, 114689, 147457, 188417]
            n = 4096
        if context_factory is None:
            from ._native import Context, default_coeff_modulus

            def context_factory(n_, t_, q_, dbc_, gdbc_):
                return Context(n_, t_, q=q_, dbc=dbc_, gdbc=gdbc_, device=device)
            q = default_coeff_modulus(n)
        else:
            q = context_factory.default_coeff_modulus(n)
        if SmallModulusCount > 0:
            q = q[:SmallModulusCount]
        envs = []
        for t in primes:
            ctx = context_factory(n, t, q, DecompositionBitCount, GaloisDecompositionBitCount)
            if client_factory is not None:
                client = client_factory(t, n, q, DecompositionBitCount, GaloisDecompositionBitCount)
            elif device_client_factory is not None:
                client = device_client_factory(ctx, t)
            else:
                from .client import DeviceClient                   # keygen / encrypt / decrypt on the device, seeded from os.urandom
                client = DeviceClient(ctx, seed=None if client_seed is None else client_seed ^ t)
            e = AtomicSealBfvEncryptedEnvironment(ctx, client)
            if client is not None:
                e.GenerateEncryptionKeys(with_galois=galois)
            envs.append(e)
        self.referenceEnvironment = EncryptedSealBfvEnvironment(envs, ParentFactory=self)

    @classmethod
    def Load(cls, source, device=0, context_factory=None, client_factory=None):
        """EncryptedSealBfvFactory(string fileName) / (Stream stream) (IFactory.cs:262-276): environments from the zip key
        container.  `client_factory(ctx)` overrides the client built on each loaded context (default DeviceClient)."""
        from . import serialization
        if context_factory is None:
            from ._native import Context

            def context_factory(n_, t_, q_, dbc_, gdbc_):
                return Context(n_, t_, q=q_, dbc=dbc_, gdbc=gdbc_, device=device)
        self = cls.__new__(cls)
        if isinstance(source, (str, bytes)):
            with open(source, "rb") as f:
                envs = serialization.load_environments(f, context_factory, client_factory)
        else:
            envs = serialization.load_environments(source, context_factory, client_factory)
        self.referenceEnvironment = EncryptedSealBfvEnvironment(envs, ParentFactory=self)
        return self

    def Save(self, target, withPrivateKeys=False):
        """IFactory.cs:296-304: zip of one key stream per plaintext prime; the secret keys only on request"""
        from . import serialization
        if isinstance(target, (str, bytes)):
            with open(target, "wb") as f:
                serialization.save_environments(f, self.referenceEnvironment.Environments, withPrivateKeys)
            return None
        return serialization.save_environments(target, self.referenceEnvironment.Environments, withPrivateKeys)

    def LoadVector(self, stream):
        return EncryptedSealBfvVector.Read(stream, self.referenceEnvironment)

    def LoadMatrix(self, stream):
        return EncryptedSealBfvMatrix.Read(stream, self.referenceEnvironment)

    def AllocateComputationEnv(self):
        return self.referenceEnvironment

    def FreeComputationEnv(self, env):
        pass

    def CopyVector(self, v):
        return EncryptedSealBfvVector.Copy(v, self.referenceEnvironment)

    def GetPlainVector(self, v, format, scale=None):
        if scale is None:
            return EncryptedSealBfvVector(env=self.referenceEnvironment, EncryptData=False, Format=format, integers=v)
        return EncryptedSealBfvVector(v, self.referenceEnvironment, scale, EncryptData=False, Format=format)

    def GetEncryptedVector(self, v, format, scale=None):
        if scale is None:
            return EncryptedSealBfvVector(env=self.referenceEnvironment, EncryptData=True, Format=format, integers=v)
        return EncryptedSealBfvVector(v, self.referenceEnvironment, scale, EncryptData=True, Format=format)

    def _matrix(self, m, format, scale, encrypt):
        m = np.asarray(m, dtype=np.float64)
        rows = m.T if format == EMatrixFormat.ColumnMajor else m
        vecs = [EncryptedSealBfvVector(r, self.referenceEnvironment, scale, EncryptData=encrypt, Format=EVectorFormat.dense) for r in rows]
        return EncryptedSealBfvMatrix(vecs, self.referenceEnvironment, CopyVectors=False, Format=format)

    def GetPlainMatrix(self, m, format, scale):
        return self._matrix(m, format, scale, False)

    def GetEncryptedMatrix(self, m, format, scale):
        return self._matrix(m, format, scale, True)

    def GetMatrix(self, vectors, format, CopyVectors=True):
        return EncryptedSealBfvMatrix(list(vectors), self.referenceEnvironment, CopyVectors=CopyVectors, Format=format)

    def GetValueFromString(self, s):
        f = [int(x) for x in s.split(",")]
        env = self.referenceEnvironment
        return sum(c * x for c, x in zip(env.preComputedCoefficients, f)) % env.bigFactor

    def GetStringFromValue(self, value):
        return ",".join(str(value % e.plainmodulusValue) for e in self.referenceEnvironment.Environments)
