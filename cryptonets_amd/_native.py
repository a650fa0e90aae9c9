"""ctypes binding of libcnhip.so (C ABI declared in include/cnhip.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is present, the
operations raise.  The oracle under oracle/ is never imported from this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "lib", "libcnhip.so")
CSRC = os.path.join(_PKG, "csrc")
# one translation unit per kernel family: hipcc compiles them in parallel (the register-radix kernels alone are ~180 instantiations)
SOURCES = [os.path.join(CSRC, f) for f in (
    "cn_api.hip", "cn_eval.hip", "cn_client.hip", "cn_defer.hip", "cn_multi.hip", "cn_host.cpp", "cn_tables.cpp", "cn_l_gemm.hip", "cn_l_behz.hip",
    "cn_l_rr_u64.hip", "cn_l_rr_f64.hip", "cn_l_rr_f64l.hip", "cn_l_ks_u64.hip", "cn_l_ks_f64.hip", "cn_l_ks_f64l.hip")]
OBJ_DIR = os.path.join(_PKG, "lib", "obj")

U64P = C.POINTER(C.c_uint64)
I32P = C.POINTER(C.c_int32)
U32P = C.POINTER(C.c_uint32)


class CnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libcnhip error %d: %s" % (code, msg))
        self.code = code


class CnStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "Multiplication", "PlainMultiplication", "Addition", "PlainAddition", "Subtraction", "PlainSubtraction",
        "Rotation", "AddMany", "AddManyItemCount", "Relinarization",
        "ntt_forward_limbs", "ntt_inverse_limbs", "kernel_launches")]


def _deps(path, seen=None):
    """`path` and every header it includes (transitively, quoted includes only)"""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        _deps(os.path.normpath(os.path.join(os.path.dirname(path), inc)), seen)
    return seen


# Instruction-scheduling strategy of the compiler per translation unit (-mllvm -amdgpu-sched-strategy=...), measured on the MI355X against the default
# (profiles/r03_sched_strategy.txt, one box, alternating runs): the key-switch unit of the moduli up to 44 bits (CryptoNets / LoLa: N <= 8192) under
# "max-memory-clause" (fused key switch 3.37-3.41 -> 3.27-3.31 ms, the same 212 VGPRs; batch -0.7 %); under "max-ilp" the key switch is 18 % SLOWER (224
# VGPRs).  The unit of the 48-49-bit moduli (N = 16384) stays with the default: the split key switch of the LoLa-CIFAR shapes lost 29 % under
# "max-memory-clause" (1.77 against 1.37 s per image).  The transform units stay with the default: "max-ilp" makes the forward
# N=8192 kernel of the roofline line 7 % faster (0.46-0.47 -> 0.50 of the HBM peak at 115 VGPRs) but takes the inverse kernel to 162 VGPRs (one workgroup
# per CU instead of two) and makes the N=16384 kernels spill.  CN_SCHED_STRATEGY=0 builds everything with the default.
SCHED_STRATEGY = {"cn_l_ks_f64l.hip": "max-memory-clause"}


def _unit_flags(src):
    if os.environ.get("CN_SCHED_STRATEGY", "1") == "0":
        return []
    st = SCHED_STRATEGY.get(os.path.basename(src))
    return ["-mllvm", "-amdgpu-sched-strategy=" + st] if st else []


def build(force=False, verbose=False, defines=(), out=None):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU): the translation units are compiled in
    parallel into lib/obj/ (only those whose sources changed), then linked.  `defines` / `out`: A/B builds of tools/ (-D switches,
    another library name - their objects go to a directory of their own)."""
    from concurrent.futures import ThreadPoolExecutor
    if defines and out is None:
        raise ValueError("build(defines=...) needs `out`: an A/B build must not share objects or the library name with the default build")
    lib_path = out or LIB_PATH
    obj_dir = OBJ_DIR
    if out is not None:                  # objects of another flag set never mix: the directory name carries a digest of the -D switches
        import hashlib
        obj_dir = os.path.splitext(out)[0] + "_obj" + ("_" + hashlib.sha1(" ".join(sorted(defines)).encode()).hexdigest()[:8] if defines else "")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", *["-D" + d for d in defines]]
    jobs, stamps = [], {}
    for src in SOURCES:
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        newest = max(os.path.getmtime(d) for d in _deps(src))
        cmd = [hipcc, *flags, *_unit_flags(src), "-c", src, "-o", obj]
        # the command line is part of the staleness key (toggling CN_SCHED_STRATEGY used to reuse objects built with the other setting)
        stamp = obj + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest or not same_cmd:
            jobs.append(cmd)
            stamps[obj] = (stamp, " ".join(cmd))
    objs = [os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o") for src in SOURCES]
    if not jobs and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(o) for o in objs):
        return lib_path

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(run, jobs))
    for stamp, text in stamps.values():
        with open(stamp, "w") as f:
            f.write(text)
    run([hipcc, "-shared", "--offload-arch=gfx950", *objs, "-o", lib_path])
    return lib_path


# every exported symbol of include/cnhip.h: name -> (restype, argtypes)
_H = C.c_uint64
_CTX = C.c_void_p
_u32 = C.c_uint32
SIGNATURES = {
    "cn_version": (C.c_int, []),
    "cn_last_error": (C.c_char_p, []),
    "cn_device_count": (C.c_int, []),
    "cn_ctx_create": (C.c_int, [_u32, U64P, _u32, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(_CTX)]),
    "cn_ctx_destroy": (C.c_int, [_CTX]),
    "cn_sync": (C.c_int, [_CTX]),
    "cn_ctx_wait_for": (C.c_int, [_CTX, _CTX]),
    "cn_set_option": (C.c_int, [_CTX, C.c_char_p, C.c_int]),
    "cn_get_option": (C.c_int, [_CTX, C.c_char_p, C.POINTER(C.c_int)]),
    "cn_default_coeff_modulus": (C.c_int, [_u32, U64P]),
    "cn_key_words": (C.c_size_t, [_CTX, C.c_int]),
    "cn_set_relin_key": (C.c_int, [_CTX, C.c_void_p, C.c_size_t, C.c_int]),
    "cn_set_galois_key": (C.c_int, [_CTX, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]),
    "cn_ctx_broadcast_keys": (C.c_int, [C.POINTER(_CTX), C.c_int]),
    "cn_has_galois_key": (C.c_int, [_CTX, C.c_uint64]),
    "cn_load_key": (C.c_int, [_CTX, C.c_int, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "cn_galois_elt_from_step": (C.c_uint64, [_CTX, C.c_int]),
    "cn_ct_alloc": (C.c_int, [_CTX, _u32, _u32, C.POINTER(_H)]),
    "cn_pt_alloc": (C.c_int, [_CTX, _u32, C.POINTER(_H)]),
    "cn_free": (C.c_int, [_CTX, _H]),
    "cn_free_many": (C.c_int, [_CTX, C.POINTER(_H), _u32]),
    "cn_encrypt_zero_new": (C.c_int, [_CTX, C.c_uint64, C.POINTER(_H)]),
    "cn_ct_upload": (C.c_int, [_CTX, _H, _u32, _u32, U64P]),
    "cn_ct_download": (C.c_int, [_CTX, _H, _u32, _u32, U64P]),
    "cn_pt_upload": (C.c_int, [_CTX, _H, _u32, _u32, U64P]),
    "cn_pt_download": (C.c_int, [_CTX, _H, _u32, _u32, U64P]),
    "cn_encode": (C.c_int, [_CTX, U64P, _u32, _H, _u32]),
    "cn_decode": (C.c_int, [_CTX, _H, _u32, U64P]),
    "cn_encode_batch": (C.c_int, [_CTX, U64P, _u32, _u32, _H, _u32]),
    "cn_decode_batch": (C.c_int, [_CTX, _H, _u32, _u32, U64P]),
    "cn_copy": (C.c_int, [_CTX, _H, _u32, _H, _u32, _u32]),
    "cn_copy_many": (C.c_int, [_CTX, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), _u32, _H, _u32]),
    "cn_device_ptr": (C.c_int, [_CTX, _H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "cn_live_handles": (C.c_int, [_CTX]),
    "cn_add": (C.c_int, [_CTX, _H, _u32, _H, _u32, _H, _u32, _u32]),
    "cn_sub": (C.c_int, [_CTX, _H, _u32, _H, _u32, _H, _u32, _u32]),
    "cn_negate": (C.c_int, [_CTX, _H, _u32, _H, _u32, _u32]),
    "cn_add_many": (C.c_int, [_CTX, _H, U32P, _u32, _H, _u32]),
    "cn_add_plain": (C.c_int, [_CTX, _H, _u32, _H, _u32, C.c_int, _H, _u32, _u32]),
    "cn_mul_plain": (C.c_int, [_CTX, _H, _u32, _H, _u32, _u32, _H, _u32, _u32]),
    "cn_mul_scalar": (C.c_int, [_CTX, _H, _u32, U64P, _u32, _H, _u32, _u32]),
    "cn_scalar_gemm": (C.c_int, [_CTX, _H, I32P, U64P, _u32, _u32, _H, I32P, _H, _u32]),
    "cn_scalar_dot": (C.c_int, [_CTX, C.POINTER(_H), U32P, U64P, _u32, _H, _u32]),
    "cn_gemm_plan_create": (C.c_int, [_CTX, I32P, U64P, _u32, _u32, _H, I32P, C.POINTER(_H)]),
    "cn_gemm_plan_apply": (C.c_int, [_CTX, _H, _H, _H, _u32]),
    "cn_graph_begin": (C.c_int, [_CTX]),
    "cn_graph_end": (C.c_int, [_CTX, C.POINTER(_H)]),
    "cn_graph_launch": (C.c_int, [_CTX, _H]),
    "cn_multiply": (C.c_int, [_CTX, _H, _u32, _H, _u32, _H, _u32, _u32]),
    "cn_relinearize": (C.c_int, [_CTX, _H, _u32, _H, _u32, _u32]),
    "cn_mul_relin": (C.c_int, [_CTX, _H, _u32, _u32, _H, _u32, _u32, _H, _u32, _u32]),
    "cn_apply_galois": (C.c_int, [_CTX, _H, _u32, C.c_uint64, _H, _u32, _u32]),
    "cn_rotate_rows": (C.c_int, [_CTX, _H, _u32, C.c_int, _H, _u32, _u32]),
    "cn_rotate_rows_many": (C.c_int, [_CTX, _H, C.POINTER(C.c_uint32), C.POINTER(C.c_int), _u32, _H, C.POINTER(C.c_uint32)]),
    "cn_rotate_columns": (C.c_int, [_CTX, _H, _u32, _H, _u32, _u32]),
    "cn_rotate_rows_add": (C.c_int, [_CTX, _H, _u32, C.c_int, _H, _u32, _H, _u32, _u32]),
    "cn_rotate_columns_add": (C.c_int, [_CTX, _H, _u32, _H, _u32, _H, _u32, _u32]),
    "cn_sum_slots": (C.c_int, [_CTX, _H, _u32, _u32, _u32]),
    "cn_rowdot_batch": (C.c_int, [_CTX, _H, _u32, _H, _u32, _u32, _u32, _H, _u32]),
    "cn_set_rng_salt": (C.c_int, [_CTX, C.c_uint64]),
    "cn_set_rng_key": (C.c_int, [_CTX, C.c_char_p]),
    "cn_rng_selftest": (C.c_int, [_CTX, C.c_char_p, C.c_uint64, C.c_uint64, U32P]),
    "cn_keygen": (C.c_int, [_CTX, C.c_uint64, C.c_int]),
    "cn_set_public_key": (C.c_int, [_CTX, U64P, C.c_size_t]),
    "cn_set_secret_key": (C.c_int, [_CTX, U64P, C.c_size_t]),
    "cn_get_key": (C.c_int, [_CTX, C.c_int, C.c_uint64, U64P, C.c_size_t]),
    "cn_encrypt": (C.c_int, [_CTX, _H, _u32, _u32, _H, _u32, _u32, C.c_uint64]),
    "cn_decrypt": (C.c_int, [_CTX, _H, _u32, _u32, _H, _u32]),
    "cn_noise_poly": (C.c_int, [_CTX, _H, _u32, _u32, C.POINTER(C.c_uint64)]),
    "cn_ntt_forward": (C.c_int, [_CTX, C.c_void_p, _u32, C.c_int]),
    "cn_ntt_inverse": (C.c_int, [_CTX, C.c_void_p, _u32, C.c_int]),
    "cn_ct_ntt": (C.c_int, [_CTX, _H, _u32, _u32, C.c_int]),
    "cn_ntt_time": (C.c_int, [_CTX, C.c_void_p, _u32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "cn_valu_issue_time": (C.c_int, [_CTX, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "cn_stream": (C.c_void_p, [_CTX]),
    "cn_event_time_begin": (C.c_int, [_CTX]),
    "cn_event_time_end": (C.c_int, [_CTX, C.POINTER(C.c_float)]),
    "cn_stats_get": (C.c_int, [_CTX, C.POINTER(CnStats), C.c_int]),
}

_lib = None


def lib():
    """Load libcnhip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cryptonets_amd: %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(os.environ.get("CNHIP_LIB", LIB_PATH))    # CNHIP_LIB: developer override for A/B kernel builds (tools/)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the ABI and the header diverge
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p64(a):
    if a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
        raise ValueError("expected contiguous uint64 array")
    return a.ctypes.data_as(U64P)


def broadcast_keys(contexts):
    """evaluation keys of contexts[0] into the others (cn_ctx_broadcast_keys: RCCL across GPUs, device copies on one GPU)"""
    arr = (_CTX * len(contexts))(*[c._h for c in contexts])
    rc = lib().cn_ctx_broadcast_keys(arr, len(contexts))
    if rc:
        raise CnError(rc, lib().cn_last_error().decode())


def default_coeff_modulus(n):
    buf = (C.c_uint64 * 16)()
    cnt = lib().cn_default_coeff_modulus(n, buf)
    if cnt <= 0:
        raise ValueError("no default coefficient modulus for n=%d" % n)
    return [int(buf[i]) for i in range(cnt)]


class Context:
    """One BFV evaluation context resident on one MI355X (HBM tables, keys, buffers, stream)."""

    def __init__(self, n, t, q=None, dbc=10, gdbc=20, device=0):
        self.L = lib()
        if q is None:
            q = default_coeff_modulus(n)
        self.n, self.t, self.q, self.k = int(n), int(t), [int(x) for x in q], len(q)
        self.dbc, self.gdbc, self.device = dbc, gdbc, device
        self.ctw = 2 * self.k * self.n
        self._ct_size = {}
        qa = (C.c_uint64 * self.k)(*self.q)
        h = _CTX()
        self._h = None
        self._chk(self.L.cn_ctx_create(self.n, qa, self.k, self.t, dbc, gdbc, device, C.byref(h)))
        self._h = h

    def _chk(self, rc):
        if rc:
            raise CnError(rc, self.L.cn_last_error().decode())

    def set_option(self, name, value):
        self._chk(self.L.cn_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int()
        self._chk(self.L.cn_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def close(self):
        if self._h is not None:
            self.L.cn_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- keys
    def key_words(self, galois=False):
        return int(self.L.cn_key_words(self._h, int(galois)))

    def set_relin_key(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self._chk(self.L.cn_set_relin_key(self._h, w.ctypes.data, w.size, 0))

    def set_relin_key_device(self, dev_ptr, words):
        self._chk(self.L.cn_set_relin_key(self._h, dev_ptr, words, 1))

    def set_galois_key(self, elt, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self._chk(self.L.cn_set_galois_key(self._h, elt, w.ctypes.data, w.size, 0))

    def set_galois_key_device(self, elt, dev_ptr, words):
        self._chk(self.L.cn_set_galois_key(self._h, elt, dev_ptr, words, 1))

    def has_galois_key(self, elt):
        return bool(self.L.cn_has_galois_key(self._h, elt))

    def load_key(self, which, words, elt=0, coeff_form=False):
        """cn_load_key: which 0 relin, 1 galois (elt), 2 public, 3 secret; coeff_form: the polynomials are in coefficient form and the
        device transforms them with its own tables"""
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self._chk(self.L.cn_load_key(self._h, which, elt, w.ctypes.data, w.size, 0, int(bool(coeff_form))))

    def galois_elt_from_step(self, steps):
        return int(self.L.cn_galois_elt_from_step(self._h, steps))

    # ---- buffers
    def ct_alloc(self, count, size=2):
        h = _H()
        self._chk(self.L.cn_ct_alloc(self._h, count, size, C.byref(h)))
        self._ct_size[h.value] = size
        return h.value

    def pt_alloc(self, count):
        h = _H()
        self._chk(self.L.cn_pt_alloc(self._h, count, C.byref(h)))
        return h.value

    def free(self, h):
        if self._h is None:                                # context closed: its arrays went with it
            return
        self._chk(self.L.cn_free(self._h, h))
        self._ct_size.pop(h, None)

    def free_many(self, handles):
        if self._h is None or not len(handles):
            return
        a = (_H * len(handles))(*[int(h) for h in handles])
        self._chk(self.L.cn_free_many(self._h, a, len(handles)))
        for h in handles:
            self._ct_size.pop(int(h), None)

    def encrypt_zero_new(self, seed=1):
        """a new one-ciphertext array holding a fresh encryption of zero (cn_ct_alloc + cn_encrypt(pt = 0) in one call)"""
        h = _H()
        self._chk(self.L.cn_encrypt_zero_new(self._h, seed, C.byref(h)))
        self._ct_size[h.value] = 2
        return h.value

    def ct_upload(self, h, first, data):
        d = np.ascontiguousarray(data, dtype=np.uint64)
        count = d.shape[0] if d.ndim > 1 else 1
        _, nbytes = self.device_ptr(h)                     # the library reads count * item words from the host pointer: check the row width here
        row = d.size // count if count else 0
        size = self._ct_size.get(h)                        # polynomials per ciphertext of this handle, when it was allocated through this object
        if count == 0 or d.size != count * row or row % (self.k * self.n) or row // (self.k * self.n) not in (2, 3) \
                or (size is not None and row != size * self.k * self.n) or nbytes % (row * 8) or first + count > nbytes // (row * 8):
            raise ValueError("ct_upload: data of shape %s does not fit the ciphertexts of this handle" % (d.shape,))
        self._chk(self.L.cn_ct_upload(self._h, h, first, count, _p64(d)))

    def ct_download(self, h, first, count, size=2):
        out = np.empty((count, size * self.k * self.n), dtype=np.uint64)
        self._chk(self.L.cn_ct_download(self._h, h, first, count, _p64(out)))
        return out

    def pt_upload(self, h, first, data):
        d = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, self.n)
        self._chk(self.L.cn_pt_upload(self._h, h, first, d.shape[0], _p64(d)))

    def pt_download(self, h, first, count):
        out = np.empty((count, self.n), dtype=np.uint64)
        self._chk(self.L.cn_pt_download(self._h, h, first, count, _p64(out)))
        return out

    def encode(self, values, pt, pi):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        self._chk(self.L.cn_encode(self._h, _p64(v), v.size, pt, pi))

    def decode(self, pt, pi):
        out = np.empty(self.n, dtype=np.uint64)
        self._chk(self.L.cn_decode(self._h, pt, pi, _p64(out)))
        return out

    def encode_batch(self, values, pt, pi):
        """BatchEncoder.Encode of values[count, nvalues] into pt[pi .. pi + count) with one library call"""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        if v.ndim != 2:
            raise ValueError("encode_batch expects a [count, nvalues] array")
        self._chk(self.L.cn_encode_batch(self._h, _p64(v), v.shape[1], v.shape[0], pt, pi))

    def decode_batch(self, pt, pi, count):
        """slot values [count, N] of pt[pi .. pi + count)"""
        out = np.empty((count, self.n), dtype=np.uint64)
        self._chk(self.L.cn_decode_batch(self._h, pt, pi, count, _p64(out)))
        return out

    def copy(self, src, sfirst, dst, dfirst, count):
        self._chk(self.L.cn_copy(self._h, src, sfirst, dst, dfirst, count))

    def copy_many(self, srcs, sfirsts, dst, dfirst):
        """dst[dfirst + i] = srcs[i][sfirsts[i]]: single ciphertexts (or plaintexts) of many arrays into one array with one launch"""
        hs = np.ascontiguousarray(srcs, dtype=np.uint64)
        fs = np.ascontiguousarray(sfirsts, dtype=np.uint32)
        self._chk(self.L.cn_copy_many(self._h, hs.ctypes.data_as(C.POINTER(C.c_uint64)), fs.ctypes.data_as(C.POINTER(C.c_uint32)), len(hs), dst, dfirst))

    def device_ptr(self, h):
        p, b = C.c_void_p(), C.c_size_t()
        self._chk(self.L.cn_device_ptr(self._h, h, C.byref(p), C.byref(b)))
        return p.value, b.value

    def live_handles(self):
        return self.L.cn_live_handles(self._h)

    def sync(self):
        self._chk(self.L.cn_sync(self._h))

    def wait_for(self, other):
        """work submitted to this context from now on starts after everything submitted to `other` so far (device-side ordering)"""
        self._chk(self.L.cn_ctx_wait_for(self._h, other._h))

    # ---- evaluator
    def add(self, a, ai, b, bi, out, oi, count=1):
        self._chk(self.L.cn_add(self._h, a, ai, b, bi, out, oi, count))

    def sub(self, a, ai, b, bi, out, oi, count=1):
        self._chk(self.L.cn_sub(self._h, a, ai, b, bi, out, oi, count))

    def negate(self, a, ai, out, oi, count=1):
        self._chk(self.L.cn_negate(self._h, a, ai, out, oi, count))

    def add_many(self, src, idx, out, oi):
        ia = np.ascontiguousarray(idx, dtype=np.uint32)
        self._chk(self.L.cn_add_many(self._h, src, ia.ctypes.data_as(U32P), ia.size, out, oi))

    def add_plain(self, a, ai, pt, pi, out, oi, count=1, subtract=False):
        self._chk(self.L.cn_add_plain(self._h, a, ai, pt, pi, int(subtract), out, oi, count))

    def mul_plain(self, a, ai, pt, pi, out, oi, count=1, pt_stride=1):
        self._chk(self.L.cn_mul_plain(self._h, a, ai, pt, pi, pt_stride, out, oi, count))

    def mul_scalar(self, a, ai, scalars, out, oi, count=1, broadcast=False):
        s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1)
        self._chk(self.L.cn_mul_scalar(self._h, a, ai, _p64(s), 0 if broadcast else 1, out, oi, count))

    def scalar_gemm(self, src, W, out, oi, idx=None, bias_pt=0, bias_idx=None):
        W = np.ascontiguousarray(W, dtype=np.uint64)
        O, K = W.shape
        ip = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            if idx.shape != (O, K):
                raise ValueError("gather table must have the shape of the weight matrix")
            ip = idx.ctypes.data_as(I32P)
        bp = None
        if bias_pt:
            bias_idx = np.ascontiguousarray(bias_idx, dtype=np.int32)
            if bias_idx.shape != (O,):
                raise ValueError("one bias index per output expected")
            bp = bias_idx.ctypes.data_as(I32P)
        self._chk(self.L.cn_scalar_gemm(self._h, src, ip, _p64(W), O, K, bias_pt, bp, out, oi))

    def scalar_dot(self, handles, indices, weights, out, oi):
        """out[oi] = sum_k weights[k] * handles[k][indices[k]]: one output of DenseMatrixBySparseVectorMultiply whose input ciphertexts
        are separate arrays (handle 0 = padded tap)"""
        hs = np.ascontiguousarray(handles, dtype=np.uint64)
        ix = np.ascontiguousarray(indices, dtype=np.uint32)
        w = np.ascontiguousarray(weights, dtype=np.uint64)
        if not (hs.shape == ix.shape == w.shape) or hs.ndim != 1:
            raise ValueError("scalar_dot: handles, indices and weights must be 1-d arrays of one length")
        self._chk(self.L.cn_scalar_dot(self._h, hs.ctypes.data_as(C.POINTER(_H)), ix.ctypes.data_as(U32P), _p64(w), hs.size, out, oi))

    def gemm_plan(self, W, idx=None, bias_pt=0, bias_idx=None):
        """plan a scalar GEMM once (weights and gather tables stay in HBM); returns a handle for gemm_apply / free"""
        W = np.ascontiguousarray(W, dtype=np.uint64)
        O, K = W.shape
        ip = None
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            if idx.shape != (O, K):
                raise ValueError("gather table must have the shape of the weight matrix")
            ip = idx.ctypes.data_as(I32P)
        bp = None
        if bias_pt:
            bias_idx = np.ascontiguousarray(bias_idx, dtype=np.int32)
            if bias_idx.shape != (O,):
                raise ValueError("one bias index per output expected")
            bp = bias_idx.ctypes.data_as(I32P)
        h = _H()
        self._chk(self.L.cn_gemm_plan_create(self._h, ip, _p64(W), O, K, bias_pt, bp, C.byref(h)))
        return h.value

    def gemm_apply(self, plan, src, out, oi):
        self._chk(self.L.cn_gemm_plan_apply(self._h, plan, src, out, oi))

    def multiply(self, a, ai, b, bi, out3, oi, count=1):
        self._chk(self.L.cn_multiply(self._h, a, ai, b, bi, out3, oi, count))

    def relinearize(self, in3, ii, out, oi, count=1):
        self._chk(self.L.cn_relinearize(self._h, in3, ii, out, oi, count))

    def mul_relin(self, a, ai, b, bi, out, oi, count=1, a_stride=1, b_stride=1):
        self._chk(self.L.cn_mul_relin(self._h, a, ai, a_stride, b, bi, b_stride, out, oi, count))

    def apply_galois(self, src, ii, elt, out, oi, count=1):
        self._chk(self.L.cn_apply_galois(self._h, src, ii, elt, out, oi, count))

    def rotate_rows(self, src, ii, steps, out, oi, count=1):
        self._chk(self.L.cn_rotate_rows(self._h, src, ii, steps, out, oi, count))

    def rotate_rows_many(self, src, iis, steps, out, ois):
        """out[ois[i]] = RotateRows(src[iis[i]], steps[i]): n rotations by n different step counts as one launch chain"""
        a = np.ascontiguousarray(iis, dtype=np.uint32)
        st = np.ascontiguousarray(steps, dtype=np.int32)
        o = np.ascontiguousarray(ois, dtype=np.uint32)
        self._chk(self.L.cn_rotate_rows_many(self._h, src, a.ctypes.data_as(C.POINTER(C.c_uint32)), st.ctypes.data_as(C.POINTER(C.c_int)), len(a), out,
                                             o.ctypes.data_as(C.POINTER(C.c_uint32))))

    def rotate_rows_add(self, src, ii, steps, acc, ai, out, oi, count=1):
        """out = acc + RotateRows(src, steps) (fused rotate-and-add of SumAllSlots)"""
        self._chk(self.L.cn_rotate_rows_add(self._h, src, ii, steps, acc, ai, out, oi, count))

    def sum_slots(self, h, first, count, length=0):
        """in-place SumAllSlots(length) of `count` single-block ciphertexts (length 0 = every slot)"""
        self._chk(self.L.cn_sum_slots(self._h, h, first, count, length))

    def rowdot_batch(self, v, vi, pt, pi, rows, length, out, oi):
        """out[oi + r] = SumAllSlots(v[vi] * pt[pi + r], length) for r < rows"""
        self._chk(self.L.cn_rowdot_batch(self._h, v, vi, pt, pi, rows, length, out, oi))

    def rotate_columns_add(self, src, ii, acc, ai, out, oi, count=1):
        self._chk(self.L.cn_rotate_columns_add(self._h, src, ii, acc, ai, out, oi, count))

    def rotate_columns(self, src, ii, out, oi, count=1):
        self._chk(self.L.cn_rotate_columns(self._h, src, ii, out, oi, count))

    # ---- client side on the device
    def set_rng_salt(self, salt):
        self._chk(self.L.cn_set_rng_salt(self._h, salt))

    def set_rng_key(self, key32):
        """256-bit ChaCha20 key of the context's sampler (bytes of length 32)"""
        key32 = bytes(key32)
        if len(key32) != 32:
            raise ValueError("the sampler key has 32 bytes")
        self._chk(self.L.cn_set_rng_key(self._h, key32))

    def rng_block(self, key32, counter, nonce):
        """one raw block (16 words) of the device sampler's generator (known-answer tests)"""
        out = np.zeros(16, dtype=np.uint32)
        self._chk(self.L.cn_rng_selftest(self._h, bytes(key32), counter, nonce, out.ctypes.data_as(U32P)))
        return out

    def keygen(self, seed, galois=True):
        self._chk(self.L.cn_keygen(self._h, seed, int(galois)))

    def set_public_key(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self._chk(self.L.cn_set_public_key(self._h, _p64(w), w.size))

    def set_secret_key(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self._chk(self.L.cn_set_secret_key(self._h, _p64(w), w.size))

    def get_key(self, which, elt=0):
        words = {0: self.key_words(False), 1: self.key_words(True), 2: self.ctw, 3: self.ctw // 2}[which]
        out = np.empty(words, dtype=np.uint64)
        self._chk(self.L.cn_get_key(self._h, which, elt, _p64(out), words))
        return out

    def encrypt(self, pt, pi, out, oi, count=1, seed=1, pt_stride=1):
        self._chk(self.L.cn_encrypt(self._h, pt, pi, pt_stride, out, oi, count, seed))

    def decrypt(self, ct, ci, count, pt_out, pi):
        self._chk(self.L.cn_decrypt(self._h, ct, ci, count, pt_out, pi))

    def noise_poly(self, ct, ci=0, count=1):
        """residues of t*(c0 + c1 s + c2 s^2) mod q_j, uint64 [count, k, n] (needs the secret key)"""
        out = np.empty((count, self.k, self.n), dtype=np.uint64)
        self._chk(self.L.cn_noise_poly(self._h, ct, ci, count, _p64(out)))
        return out

    def invariant_noise_budget(self, ct, ci=0, count=1, exact_bits=False):
        """Decryptor.InvariantNoiseBudget (what CryptoTracker.TestBudget reads, CryptoTracker.cs:41-52) of `count` ciphertexts:
        log2(q) - log2(|| t (c0 + c1 s + c2 s^2) mod q ||_inf, centred) - 1, in bits (float); with `exact_bits` SEAL's integer
        max(0, bitcount(q) - bitcount(norm) - 1).  The limbs are composed on the host with Python integers: a debugging probe."""
        import math
        w = self.noise_poly(ct, ci, count)
        Q = 1
        for qj in self.q:
            Q *= qj
        coef = [(Q // qj) * pow((Q // qj) % qj, -1, qj) for qj in self.q]
        out = []
        for c in range(count):
            x = sum(w[c, j].astype(object) * coef[j] for j in range(self.k)) % Q
            norm = max(int(v) if 2 * int(v) <= Q else Q - int(v) for v in x)
            if exact_bits:
                out.append(max(0, Q.bit_length() - norm.bit_length() - 1))
            else:
                out.append(math.log2(Q) - (math.log2(norm) if norm else 0.0) - 1.0)
        return out

    # ---- raw transforms / timing / stats
    def ct_ntt(self, h, first, count, inverse=False):
        self._chk(self.L.cn_ct_ntt(self._h, h, first, count, int(inverse)))

    def ntt_forward(self, dev_ptr, limbs, base=0):
        self._chk(self.L.cn_ntt_forward(self._h, dev_ptr, limbs, base))

    def ntt_inverse(self, dev_ptr, limbs, base=0):
        self._chk(self.L.cn_ntt_inverse(self._h, dev_ptr, limbs, base))

    def ntt_time(self, dev_ptr, limbs, base=0, inverse=False, iters=10):
        ms = C.c_float()
        self._chk(self.L.cn_ntt_time(self._h, dev_ptr, limbs, base, int(inverse), iters, C.byref(ms)))
        return ms.value

    def fp64_issue_ns(self, iters=2048, launches=4):
        """ns per FP64 wave-instruction per SIMD right now (cn_valu_issue_time, kind 0)"""
        ns = C.c_float()
        self._chk(self.L.cn_valu_issue_time(self._h, 0, iters, launches, C.byref(ns)))
        return ns.value

    def valu32_issue_ns(self, iters=8192, launches=4):
        """ns per full-rate 32-bit VALU wave-instruction per SIMD right now (cn_valu_issue_time, kind 1)"""
        ns = C.c_float()
        self._chk(self.L.cn_valu_issue_time(self._h, 1, iters, launches, C.byref(ns)))
        return ns.value

    def stream(self):
        return self.L.cn_stream(self._h)

    # ---- captured sequences (HIP graphs): see include/cnhip.h
    def graph_begin(self):
        self._chk(self.L.cn_graph_begin(self._h))

    def graph_end(self):
        h = _H()
        self._chk(self.L.cn_graph_end(self._h, C.byref(h)))
        return h.value

    def graph_launch(self, graph):
        self._chk(self.L.cn_graph_launch(self._h, graph))

    def time_begin(self):
        self._chk(self.L.cn_event_time_begin(self._h))

    def time_end(self):
        ms = C.c_float()
        self._chk(self.L.cn_event_time_end(self._h, C.byref(ms)))
        return ms.value

    def stats(self, reset=False):
        s = CnStats()
        self._chk(self.L.cn_stats_get(self._h, C.byref(s), int(reset)))
        return {n: int(getattr(s, n)) for n, _ in CnStats._fields_}
