#!/usr/bin/env python
"""Record the C-ABI calls one evaluation makes and replay them from native code.

Why: the Python mirror of the reference's C# host pays ~5-10 us of interpreter / ctypes time per library call; a single-image LoLa
inference is several hundred small calls per plaintext prime, so a Python loop measures Python.  The C# twin pays a P/Invoke
(~100 ns).  This tool records, per context, the exact sequence of evaluator calls an evaluation issues (with `hewrapper.LITERAL` that is the
call sequence of the reference's UNCHANGED files: one AtomicSealBfvEncryptedVector method per row / column / map) and hands it to
`tools/replay_call_trace.cpp`, which issues the same calls on the same C ABI from C++ and times them - the unchanged caller of the
LoLa networks, through the boundary, without an interpreter in the loop (VERDICT r02 "missing" #4).

A trace is a flat list of records (opcode, integer arguments, u64 blob).  Handles are trace-local ids: ids below `n_ext` name handles that
existed before the recording started (weights, masks, the encrypted input) and are supplied by the driver at replay time; every
cn_ct_alloc / cn_pt_alloc of the trace defines the next id.  cn_free of an external handle is dropped (the input is reused by every
replayed inference).  Calls that synchronise (downloads, cn_sync) are not part of a trace.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tools", "replay_call_trace.cpp")
LIB = os.path.join(ROOT, "cryptonets_amd", "lib", "libcntrace.so")

OPS = ["CT_ALLOC", "PT_ALLOC", "FREE", "COPY", "ADD", "SUB", "NEGATE", "ADD_MANY", "ADD_PLAIN", "MUL_PLAIN", "MUL_SCALAR", "SCALAR_DOT", "MUL_RELIN",
       "ROTATE_ROWS", "ROTATE_COLUMNS", "ROTATE_ROWS_ADD", "ROTATE_COLUMNS_ADD", "SUM_SLOTS", "ENCODE_BATCH", "PT_UPLOAD", "GEMM_APPLY", "MULTIPLY",
       "RELINEARIZE", "APPLY_GALOIS", "ROWDOT_BATCH", "SCALAR_GEMM", "COPY_MANY", "ROTATE_ROWS_MANY"]
OP = {name: i for i, name in enumerate(OPS)}


class Recorder:
    """Wraps the methods of ONE cryptonets_amd._native.Context: while active every evaluator call is appended to `records` and executed."""

    def __init__(self, ctx):
        self.ctx, self.records, self.ids, self.ext, self.active = ctx, [], {}, [], False
        self._orig, self._allocs = {}, 0

    # ---- handle ids
    def _h(self, h):
        h = int(h)
        if h == 0:
            return ("none", 0)                             # "no handle" (padded tap, pt = 0)
        if h not in self.ids:
            self.ids[h] = ("ext", len(self.ext))
            self.ext.append(h)
        return self.ids[h]

    def _new(self, h):
        self.ids[int(h)] = ("new", self._allocs)         # ids of trace-made handles count the allocation records
        self._allocs += 1

    def _rec(self, op, ints, blob=None):
        self.records.append((OP[op], ints, None if blob is None else np.ascontiguousarray(blob, dtype=np.uint64).reshape(-1)))

    # ---- wrappers (signatures of cryptonets_amd._native.Context)
    def start(self):
        c, R = self.ctx, self
        wrap = {}

        def ct_alloc(count, size=2):
            R._rec("CT_ALLOC", [count, size])
            h = R._orig["ct_alloc"](count, size)
            R._new(h)
            return h

        def pt_alloc(count):
            R._rec("PT_ALLOC", [count])
            h = R._orig["pt_alloc"](count)
            R._new(h)
            return h

        def free(h):
            hid = R._h(h)
            if hid[0] != "new":
                return None                                # an external handle (the encrypted input): stays alive for the replays
            R._rec("FREE", [hid])
            R.ids.pop(int(h), None)
            return R._orig["free"](h)

        def simple(name, op, hpos):
            def f(*a, **k):
                args = R._normalise(name, a, k)
                R._rec(op, [R._h(v) if i in hpos else int(v) for i, v in enumerate(args)])
                return R._orig[name](*a, **k)
            return f
        wrap.update(ct_alloc=ct_alloc, pt_alloc=pt_alloc, free=free)
        for name, op, hpos in (("copy", "COPY", (0, 2)), ("add", "ADD", (0, 2, 4)), ("sub", "SUB", (0, 2, 4)), ("negate", "NEGATE", (0, 2)),
                               ("add_plain", "ADD_PLAIN", (0, 2, 4)), ("mul_plain", "MUL_PLAIN", (0, 2, 4)), ("mul_relin", "MUL_RELIN", (0, 2, 4)),
                               ("rotate_rows", "ROTATE_ROWS", (0, 3)), ("rotate_columns", "ROTATE_COLUMNS", (0, 2)),
                               ("rotate_rows_add", "ROTATE_ROWS_ADD", (0, 3, 5)), ("rotate_columns_add", "ROTATE_COLUMNS_ADD", (0, 2, 4)),
                               ("sum_slots", "SUM_SLOTS", (0,)), ("gemm_apply", "GEMM_APPLY", (0, 1, 2)), ("multiply", "MULTIPLY", (0, 2, 4)),
                               ("relinearize", "RELINEARIZE", (0, 2)), ("apply_galois", "APPLY_GALOIS", (0, 3)), ("rowdot_batch", "ROWDOT_BATCH", (0, 2, 6))):
            wrap[name] = simple(name, op, hpos)

        def copy_many(srcs, sfirsts, dst, dfirst):
            hs = [R._h(h) for h in np.asarray(srcs, dtype=np.uint64)]
            R._rec("COPY_MANY", [R._h(dst), int(dfirst), len(hs)],
                   np.concatenate([np.array([R._pack(h) for h in hs], dtype=np.uint64), np.asarray(sfirsts, dtype=np.uint64).reshape(-1)]))
            return R._orig["copy_many"](srcs, sfirsts, dst, dfirst)

        def rotate_rows_many(src, iis, steps, out, ois):
            n = len(iis)
            R._rec("ROTATE_ROWS_MANY", [R._h(src), R._h(out), n],
                   np.concatenate([np.asarray(iis, dtype=np.uint64).reshape(-1), np.asarray(steps, dtype=np.int64).astype(np.uint64).reshape(-1),
                                   np.asarray(ois, dtype=np.uint64).reshape(-1)]))
            return R._orig["rotate_rows_many"](src, iis, steps, out, ois)

        def add_many(src, idx, out, oi):
            R._rec("ADD_MANY", [R._h(src), R._h(out), int(oi)], np.asarray(idx, dtype=np.uint64))
            return R._orig["add_many"](src, idx, out, oi)

        def mul_scalar(a, ai, scalars, out, oi, count=1, broadcast=False):
            sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1)
            R._rec("MUL_SCALAR", [R._h(a), int(ai), 0 if broadcast else 1, R._h(out), int(oi), int(count)], sc)
            return R._orig["mul_scalar"](a, ai, scalars, out, oi, count, broadcast)

        def scalar_dot(handles, indices, weights, out, oi):
            hs = [R._h(h) for h in np.asarray(handles, dtype=np.uint64)]
            blob = np.concatenate([np.array([R._pack(h) for h in hs], dtype=np.uint64), np.asarray(indices, dtype=np.uint64).reshape(-1),
                                   np.asarray(weights, dtype=np.uint64).reshape(-1)])
            R._rec("SCALAR_DOT", [len(hs), R._h(out), int(oi)], blob)
            return R._orig["scalar_dot"](handles, indices, weights, out, oi)

        def encode_batch(values, pt, pi):
            v = np.ascontiguousarray(values, dtype=np.uint64)
            R._rec("ENCODE_BATCH", [R._h(pt), int(pi), v.shape[1], v.shape[0]], v)
            return R._orig["encode_batch"](values, pt, pi)

        def encode(values, pt, pi):
            v = np.ascontiguousarray(values, dtype=np.uint64).reshape(1, -1)
            R._rec("ENCODE_BATCH", [R._h(pt), int(pi), v.shape[1], 1], v)
            return R._orig["encode"](values, pt, pi)

        def pt_upload(h, first, data):
            d = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1, c.n)
            R._rec("PT_UPLOAD", [R._h(h), int(first), d.shape[0]], d)
            return R._orig["pt_upload"](h, first, data)

        def scalar_gemm(src, W, out, oi, idx=None, bias_pt=0, bias_idx=None):
            W_ = np.ascontiguousarray(W, dtype=np.uint64)
            O, K = W_.shape
            idx_ = np.tile(np.arange(K, dtype=np.int64), (O, 1)) if idx is None else np.asarray(idx, dtype=np.int64)
            bi = np.zeros(O, dtype=np.int64) if bias_idx is None else np.asarray(bias_idx, dtype=np.int64)
            R._rec("SCALAR_GEMM", [R._h(src), O, K, R._h(bias_pt), R._h(out), int(oi)], np.concatenate([idx_.reshape(-1).view(np.uint64), W_.reshape(-1), bi.view(np.uint64)]))
            return R._orig["scalar_gemm"](src, W, out, oi, idx, bias_pt, bias_idx)
        wrap.update(copy_many=copy_many, rotate_rows_many=rotate_rows_many, add_many=add_many, mul_scalar=mul_scalar, scalar_dot=scalar_dot, encode_batch=encode_batch, encode=encode, pt_upload=pt_upload, scalar_gemm=scalar_gemm)
        for name in ("sync", "ct_download", "pt_download", "decode", "decode_batch", "ct_upload", "decrypt", "encrypt", "gemm_plan", "graph_begin", "graph_end", "graph_launch",
                     "set_relin_key", "set_galois_key", "keygen"):
            def refuse(*a, _n=name, **k):
                raise RuntimeError("%s while a call trace is recorded: not an evaluator call of the inference" % _n)
            wrap[name] = refuse
        for name, f in wrap.items():
            if not hasattr(c, name):                       # (a backend that lacks an entry point - the CPU test harness - cannot call it either)
                continue
            self._orig[name] = getattr(c, name)
            setattr(c, name, f)
        self.active = True
        return self

    def _normalise(self, name, a, k):
        """positional argument list of a Context method call with the defaults filled in: the order the replay expects (replay_call_trace.cpp)"""
        import inspect
        b = inspect.signature(self._orig[name]).bind(*a, **k)
        b.apply_defaults()
        return list(b.arguments.values())

    @staticmethod
    def _pack(hid):
        """u64 word of a handle id: all ones = no handle, bit 63 set = allocated inside the trace, else an external handle"""
        return 0xFFFFFFFFFFFFFFFF if hid[0] == "none" else (((1 << 63) | hid[1]) if hid[0] == "new" else hid[1])

    def stop(self):
        for name, f in self._orig.items():
            setattr(self.ctx, name, f)
        self.active = False
        return self

    # ---- serialisation: u64 stream [n_records, n_ext, n_new] + per record [op, n_ints, blob_words, ints..., blob...]
    def serialise(self):
        out = [len(self.records), len(self.ext), self._allocs]
        for op, ints, blob in self.records:
            words = [self._pack(v) if isinstance(v, tuple) else int(v) & 0xFFFFFFFFFFFFFFFF for v in ints]
            out += [op, len(words), 0 if blob is None else len(blob)] + words
            if blob is not None:
                out += [int(x) for x in blob]
        return np.array(out, dtype=np.uint64)


def build(force=False):
    from cryptonets_amd import _native
    _native.build()
    deps = [SRC, os.path.join(ROOT, "include", "cnhip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", SRC, "-I" + os.path.join(ROOT, "include"),
                           "-L" + os.path.dirname(LIB), "-lcnhip", "-Wl,-rpath,$ORIGIN", "-pthread", "-o", LIB])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        from cryptonets_amd import _native
        _native.lib()
        L = C.CDLL(build())
        L.ct_replay.restype = C.c_int
        L.ct_replay.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint64)), C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


def replay(recorders, reps, mode, result_ids, warmup=1):
    """Replay the traces of `recorders` (one per context / plaintext prime) `warmup + reps` times from C++.
    mode 0: ONE host thread, the contexts round-robin call by call (asynchronous issue to the context streams, what the Python mirror does);
    mode 1: one thread per context (the reference's Task per plaintext prime, EncryptedSealBfvVector.cs:225-236), joined after every call;
    mode 2: one thread per context running its whole trace on its own.
    result_ids: per recorder, the trace id (("new", i)) of the handle that holds the result - kept alive after the LAST repetition and
    returned; everything else the trace allocates and does not free is released after every repetition.
    Returns (milliseconds per inference, [result handle per context])."""
    n = len(recorders)
    traces = [r.serialise() for r in recorders]
    exts = [np.array(r.ext, dtype=np.uint64) for r in recorders]
    tp = (C.POINTER(C.c_uint64) * n)(*[t.ctypes.data_as(C.POINTER(C.c_uint64)) for t in traces])
    ep = (C.POINTER(C.c_uint64) * n)(*[e.ctypes.data_as(C.POINTER(C.c_uint64)) for e in exts])
    ctxs = (C.c_void_p * n)(*[r.ctx._h for r in recorders])
    ms = C.c_double()
    res = np.array([rid[1] for rid in result_ids], dtype=np.uint64)
    msg = C.create_string_buffer(512)
    rc = lib().ct_replay(ctxs, n, tp, ep, int(warmup), int(reps), int(mode), C.byref(ms), res.ctypes.data_as(C.POINTER(C.c_uint64)), msg, 512)
    if rc:
        raise RuntimeError("trace replay failed (%d): %s" % (rc, msg.value.decode()))
    return ms.value, [int(x) for x in res]
