"""ctypes binding of the CPU oracle (oracle/seal32_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (cryptonets_amd/) never imports it.
Ciphertext-word parity with real SEAL 3.2 is UNPINNED (see the C file header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcnoracle.so")
_SRC = os.path.join(_HERE, "seal32_oracle.c")

U64P = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile the oracle with gcc (no-op when the .so is newer than the source)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcnoracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.cno_ctx_create.restype = C.c_void_p
        L.cno_ctx_create.argtypes = [C.c_uint32, U64P, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        for name in ("cno_secret_key", "cno_public_key", "cno_relin_key", "cno_galois_key"):
            getattr(L, name).restype = U64P
        for name in ("cno_psi", "cno_bsk_mod", "cno_galois_elt", "cno_galois_elt_from_step"):
            getattr(L, name).restype = C.c_uint64
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


# SEAL 3.2 DefaultParams.CoeffModulus128(n) (used at AtomicSealBfvVector.cs:146)
COEFF_MODULUS_128 = {
    2048: [0x3fffffff000001],
    4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
    8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
    16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001,
            0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001],
}


class Oracle:
    """One SEAL context + keys (AtomicSealBfvEncryptedEnvironment, AtomicSealBfvVector.cs:19-206)."""

    def __init__(self, n, t, q=None, small_modulus_count=-1, dbc=10, gdbc=20, ks_xi=False):
        q = list(COEFF_MODULUS_128[n] if q is None else q)
        if small_modulus_count > 0:
            q = q[:small_modulus_count]
        self.n, self.t, self.q, self.k, self.dbc, self.gdbc = n, t, q, len(q), dbc, gdbc
        qa = (C.c_uint64 * len(q))(*q)
        self.L = lib()
        self.h = C.c_void_p(self.L.cno_ctx_create(n, qa, len(q), t, dbc, gdbc))
        if not self.h:
            raise ValueError("invalid BFV parameters")
        self.ctw = 2 * self.k * n
        self.ks_xi = bool(ks_xi)
        if ks_xi:
            self.L.cno_set_ks_xi(self.h, 1)

    def set_ks_xi(self, on):
        """decomposition convention of the key switch (seal32_oracle.c: gen_ksk): False = digits of the raw residues, message term in limb l
        only; True = digits of [c_l (q/q_l)^-1]_{q_l}, message term the RNS image of (q/q_l) 2^(dbc d) s' (non-zero in limb l only).  Affects keys generated
        and key switches run after the call."""
        self.ks_xi = bool(on)
        self.L.cno_set_ks_xi(self.h, int(bool(on)))

    def key_to_coeff_form(self, words):
        """a key (any [..][k][N] array of NTT-form polynomials) in coefficient form - what a client produces with
        Evaluator.TransformFromNTTInplace before a coefficient-form upload (cn_load_key, form 1)"""
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, self.k, self.n).copy()
        for i in range(w.shape[0]):
            for j in range(self.k):
                self.L.cno_ntt_inv(self.h, j, _p(w[i, j]))
        return w.reshape(-1)

    def __del__(self):
        try:
            self.L.cno_ctx_destroy(self.h)
        except Exception:
            pass

    # --- keys -------------------------------------------------------------
    def keygen(self, seed=1, galois=True):
        self.L.cno_keygen(self.h, C.c_uint64(seed), int(galois))

    def _arr(self, ptr, words):
        return np.ctypeslib.as_array(ptr, shape=(words,)).copy()

    def import_keys(self, sk, pk):
        sk = np.ascontiguousarray(sk, dtype=np.uint64); pk = np.ascontiguousarray(pk, dtype=np.uint64)
        self.L.cno_import_keys(self.h, _p(sk), _p(pk))

    def import_relin_key(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        assert w.size == self.L.cno_relin_digits(self.h) * self.ctw
        self.L.cno_import_relin_key(self.h, _p(w))

    def import_galois_key(self, elt, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        assert w.size == self.L.cno_galois_digits(self.h) * self.ctw
        if self.L.cno_import_galois_key(self.h, C.c_uint64(int(elt)), _p(w)):
            raise ValueError("too many Galois keys")

    def public_key(self):
        return self._arr(self.L.cno_public_key(self.h), 2 * self.k * self.n)

    def secret_key(self):
        return self._arr(self.L.cno_secret_key(self.h), self.k * self.n)

    def relin_key(self):
        return self._arr(self.L.cno_relin_key(self.h), self.L.cno_relin_digits(self.h) * self.ctw)

    def galois_elts(self):
        return [int(self.L.cno_galois_elt(self.h, g)) for g in range(self.L.cno_galois_count(self.h))]

    def galois_key(self, g):
        return self._arr(self.L.cno_galois_key(self.h, g), self.L.cno_galois_digits(self.h) * self.ctw)

    def galois_elt_from_step(self, steps):
        return int(self.L.cno_galois_elt_from_step(self.h, int(steps)))

    # --- encode / encrypt ---------------------------------------------------
    def encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        out = np.zeros(self.n, dtype=np.uint64)
        rc = self.L.cno_encode(self.h, _p(v), len(v), _p(out))
        if rc:
            raise ValueError("encode failed rc=%d" % rc)
        return out

    def decode(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros(self.n, dtype=np.uint64)
        if self.L.cno_decode(self.h, _p(p), len(p), _p(out)):
            raise ValueError("decode failed")
        return out

    def encrypt(self, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        ct = np.zeros(self.ctw, dtype=np.uint64)
        rc = self.L.cno_encrypt(self.h, _p(p), len(p), _p(ct))
        if rc:
            raise ValueError("encrypt failed rc=%d" % rc)
        return ct

    def decrypt(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        size = len(ct) // (self.k * self.n)
        out = np.zeros(self.n, dtype=np.uint64)
        if self.L.cno_decrypt(self.h, _p(ct), size, _p(out)):
            raise ValueError("decrypt failed")
        return out

    def dot_with_secret(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        size = len(ct) // (self.k * self.n)
        out = np.zeros(self.k * self.n, dtype=np.uint64)
        self.L.cno_dot_with_secret(self.h, _p(ct), size, _p(out))
        return out

    def seed(self, s):
        self.L.cno_seed(self.h, C.c_uint64(s))

    # --- evaluator ----------------------------------------------------------
    def _size(self, ct):
        return len(ct) // (self.k * self.n)

    def add(self, a, b):
        out = np.zeros(max(len(a), len(b)), dtype=np.uint64)
        self.L.cno_add(self.h, _p(a), self._size(a), _p(b), self._size(b), _p(out))
        return out

    def sub(self, a, b):
        out = np.zeros(max(len(a), len(b)), dtype=np.uint64)
        self.L.cno_sub(self.h, _p(a), self._size(a), _p(b), self._size(b), _p(out))
        return out

    def negate(self, a):
        out = np.zeros_like(a)
        self.L.cno_negate(self.h, _p(a), self._size(a), _p(out))
        return out

    def add_plain(self, ct, plain, subtract=False):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros_like(ct)
        rc = self.L.cno_add_plain(self.h, _p(ct), self._size(ct), _p(p), len(p), int(subtract), _p(out))
        if rc:
            raise ValueError("add_plain failed rc=%d" % rc)
        return out

    def multiply_plain(self, ct, plain):
        p = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros_like(ct)
        rc = self.L.cno_multiply_plain(self.h, _p(ct), self._size(ct), _p(p), len(p), _p(out))
        if rc == -3:
            raise ValueError("plain cannot be zero")
        if rc:
            raise ValueError("multiply_plain failed rc=%d" % rc)
        return out

    def multiply(self, a, b):
        out = np.zeros(3 * self.k * self.n, dtype=np.uint64)
        self.L.cno_multiply(self.h, _p(a), _p(b), _p(out))
        return out

    def relinearize(self, c3):
        out = np.zeros(self.ctw, dtype=np.uint64)
        if self.L.cno_relinearize(self.h, _p(c3), _p(out)):
            raise ValueError("no relin keys")
        return out

    def apply_galois(self, ct, elt):
        out = np.zeros(self.ctw, dtype=np.uint64)
        if self.L.cno_apply_galois(self.h, _p(ct), C.c_uint64(elt), _p(out)):
            raise ValueError("Galois key not present")
        return out

    def rotate_rows(self, ct, steps):
        out = np.zeros(self.ctw, dtype=np.uint64)
        rc = self.L.cno_rotate_rows(self.h, _p(ct), int(steps), _p(out))
        if rc:
            raise ValueError("rotate_rows failed rc=%d" % rc)
        return out

    def rotate_columns(self, ct):
        out = np.zeros(self.ctw, dtype=np.uint64)
        if self.L.cno_rotate_columns(self.h, _p(ct), _p(out)):
            raise ValueError("Galois key not present")
        return out

    # --- raw transforms -----------------------------------------------------
    def ntt_fwd(self, limb, x, bsk=False):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        (self.L.cno_ntt_fwd_bsk if bsk else self.L.cno_ntt_fwd)(self.h, limb, _p(y))
        return y

    def ntt_inv(self, limb, x, bsk=False):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        (self.L.cno_ntt_inv_bsk if bsk else self.L.cno_ntt_inv)(self.h, limb, _p(y))
        return y

    def bsk_moduli(self):
        return [int(self.L.cno_bsk_mod(self.h, j)) for j in range(self.k + 1)]

    def psi(self, limb):
        return int(self.L.cno_psi(self.h, limb))

    # --- wrapper-level hot loops (CPU baseline) -------------------------------
    def scalar_gemm(self, cts, W, idx=None):
        """cts: [n_in, ctw]; W: [O, K] u64 residues mod t; idx: [O, K] int32 or None."""
        W = np.ascontiguousarray(W, dtype=np.uint64)
        O, K = W.shape
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        size = cts.shape[1] // (self.k * self.n)                     # 3: unrelinearized products (multiply_plain / add on size-3 ciphertexts)
        out = np.zeros((O, cts.shape[1]), dtype=np.uint64)
        ip = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            ip = idx.ctypes.data_as(C.POINTER(C.c_int32))
        rc = self.L.cno_scalar_gemm_sized(self.h, _p(cts), size, ip, _p(W), O, K, _p(out))
        if rc:
            raise ValueError("scalar_gemm: an output had no non-zero term")
        return out

    def mul_relin_batch(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros_like(a)
        self.L.cno_mul_relin_batch(self.h, _p(a), _p(b), a.shape[0], _p(out))
        return out

    def add_plain_batch(self, cts, plains):
        cts = np.ascontiguousarray(cts, dtype=np.uint64)
        plains = np.ascontiguousarray(plains, dtype=np.uint64)
        out = np.zeros_like(cts)
        rc = self.L.cno_add_plain_batch_sized(self.h, _p(cts), cts.shape[1] // (self.k * self.n), _p(plains), plains.shape[1], cts.shape[0], _p(out))
        if rc:
            raise ValueError("add_plain_batch failed")
        return out

    def ntt_fwd_batch(self, x):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        self.L.cno_ntt_fwd_batch(self.h, _p(y), y.size // self.n)
        return y

    def ntt_inv_batch(self, x):
        y = np.ascontiguousarray(x, dtype=np.uint64).copy()
        self.L.cno_ntt_inv_batch(self.h, _p(y), y.size // self.n)
        return y
