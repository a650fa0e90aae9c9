#!/usr/bin/env python
"""Drive libcnhip with the LITERAL call pattern of the reference's unchanged NeuralNetworks layers and compare with the batched path.

The caller itself is C++ (`tools/replay_reference_calls.cpp`: one evaluator call per ciphertext, every ciphertext its own handle, issued
from `threads` workers exactly like `Utils.ParallelProcessInEnv`, HE Wrapper/Utils.cs:46-88) - a Python loop would measure the GIL, a
C# host would look like the C++ one.  This module builds it, describes a CryptoNets-style network to it (`Replay`), and as a program
runs BASELINE config 3 both ways:

    python tools/replay_reference_calls.py [--threads 1,8,32] [--steps 5] [--immediate]

prints images/s of the batched path (bench.py's), of the unchanged caller with deferred submission (`cn_set_option("defer", 1)`), and
- `--immediate` - of the unchanged caller with every call launched on its own; the final ciphertext words of all three must be identical.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tools", "replay_reference_calls.cpp")
LIB = os.path.join(ROOT, "cryptonets_amd", "lib", "libcnreplay.so")


class RpLayer(C.Structure):
    _fields_ = [("O", C.c_uint32), ("K", C.c_uint32), ("idx", C.POINTER(C.c_int32)), ("W", C.POINTER(C.c_uint64)),
                ("bias_pt", C.POINTER(C.c_uint64)), ("bias_idx", C.POINTER(C.c_int32)), ("square", C.c_int)]


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
        _native.lib()                                  # libcnhip.so first: the replay library resolves its symbols there
        L = C.CDLL(build())
        L.rp_run.restype = C.c_int
        L.rp_run.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RpLayer), C.c_int, C.POINTER(C.c_uint64), C.c_uint32,
                             C.POINTER(C.c_uint64), C.c_int, C.c_char_p, C.c_size_t]
        L.rp_run2.restype = C.c_int
        L.rp_run2.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RpLayer), C.c_int, C.POINTER(C.c_uint64), C.c_uint32,
                              C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


class Replay:
    """A CryptoNets-style network (PoolLayers, optionally followed by SquareActivation) described to the C++ caller.
    ctxs: one cryptonets_amd._native.Context per plaintext prime; layers: dicts with idx [O,K] int32, W [primes][O,K] uint64 residues,
    bias_pt [primes] plaintext handles (or None), bias_idx [O] int32, square bool."""

    def __init__(self, ctxs, layers):
        self.ctxs, self.keep = ctxs, []
        self.arr = (RpLayer * len(layers))()
        for i, L in enumerate(layers):
            idx = np.ascontiguousarray(L["idx"], dtype=np.int32)
            W = np.ascontiguousarray(np.stack(L["W"]), dtype=np.uint64)
            O, K = idx.shape
            assert W.shape == (len(ctxs), O, K)
            self.keep += [idx, W]
            a = self.arr[i]
            a.O, a.K, a.square = O, K, int(bool(L.get("square")))
            a.idx = idx.ctypes.data_as(C.POINTER(C.c_int32))
            a.W = W.ctypes.data_as(C.POINTER(C.c_uint64))
            if L.get("bias_pt") is not None:
                bp = np.ascontiguousarray(L["bias_pt"], dtype=np.uint64)
                bi = np.ascontiguousarray(L["bias_idx"], dtype=np.int32)
                self.keep += [bp, bi]
                a.bias_pt = bp.ctypes.data_as(C.POINTER(C.c_uint64))
                a.bias_idx = bi.ctypes.data_as(C.POINTER(C.c_int32))
        self.n_out = int(self.arr[len(layers) - 1].O)
        self.hctx = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])

    def run(self, in_handles, threads, literal_taps=False, nonce0=1, merged=False, direct_free=False):
        """in_handles: uint64 [primes, n_in] (one count-1 handle per input column) -> uint64 [primes, O_last] handles (caller frees).
        literal_taps: a padded convolution tap is a fresh device encryption of zero, as PoolLayer.ElementAt makes it (the contexts need
        their public key); nonce0: first nonce of those encryptions (consecutive values are used); merged: the twin makes a zero vector with one
        library call (cn_encrypt_zero_new) and releases disposed arrays 32 at a time (cn_free_many) - same words; direct_free: ... releases every disposed array
        at once (cn_free: on a lock-free context a published record, no lock to save)"""
        ih = np.ascontiguousarray(in_handles, dtype=np.uint64)
        out = np.zeros((len(self.ctxs), self.n_out), dtype=np.uint64)
        msg = C.create_string_buffer(512)
        rc = lib().rp_run2(self.hctx, len(self.ctxs), self.arr, len(self.arr), ih.ctypes.data_as(C.POINTER(C.c_uint64)), ih.shape[1],
                           out.ctypes.data_as(C.POINTER(C.c_uint64)), int(threads), int(bool(literal_taps)) | (2 if merged else 0) | (4 if direct_free else 0), int(nonce0), msg, 512)
        if rc:
            raise RuntimeError("replay failed (%d): %s" % (rc, msg.value.decode()))
        return out


def split_columns(ctx, h, count):
    """one handle per ciphertext (the reference's individually allocated Ciphertext objects) from a batch handle"""
    out = np.zeros(count, dtype=np.uint64)
    for c in range(count):
        out[c] = ctx.ct_alloc(1)
        ctx.copy(h, c, int(out[c]), 0, 1)
    return out


def replay_layers(chans, layers):
    """rp_layer descriptions of a cryptonets_mnist network: chans = CryptoNetsChannel per prime (they own the bias plaintexts)"""
    from cryptonets_amd import cryptonets_mnist as cm
    out = []
    for li, L in enumerate(layers):
        out.append(dict(idx=L["idx"], W=[cm.residues(L["W"], ch.g.t) for ch in chans], bias_pt=[ch.layers[li]["bias_pt"] for ch in chans],
                        bias_idx=chans[0].layers[li]["bias_idx"], square=li < len(layers) - 1))
    return out


LAST_LAUNCHES = None
HOST_TIMES = []
LAST_HOST = None            # host-side accounting of the last measure(): CPU seconds the process used per wall second, CFS bandwidth throttling of its cgroup


def _cgroup_cpu_stat():
    """{nr_periods, nr_throttled, throttled_usec, usage_usec} of this process' cgroup (v2), {} where there is none: a quota of q cores with p >> q runnable
    threads is spent early in every 100 ms period and ALL threads of the group then wait for the next one - what 256 caller threads on a 16-core quota do"""
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            return {k: int(v) for k, v in (line.split() for line in f)}
    except Exception:
        return {}


def measure(chans, layers, threads, steps, warmup=1, defer=True, literal_taps=False, merged=True, per_step=False, lockfree=None, direct_free=None):
    """images/s of the unchanged caller on the inputs resident in chans[p].h_in; returns (ms per batch, output words [primes][O][...]).
    literal_taps: padded taps are fresh encryptions of zero (PoolLayer.cs:67-80) - the words then differ from the batched path's (fresh
    randomness), the DECRYPTED outputs must not (decrypt_outputs).  lockfree: cn_set_option("defer", 2) - the calls are published to the context's submission
    ring without taking its lock (round 6; default on, REPLAY_LOCKFREE=0 or lockfree=False: "defer" = 1, every call under the lock)"""
    if lockfree is None:
        lockfree = os.environ.get("REPLAY_LOCKFREE", "1") != "0"
    if direct_free is None:               # Dispose() = one cn_free at once (the round-6 twin) instead of the per-thread lists released 32 at a time (rounds 3-5: halves the lock acquisitions)
        direct_free = bool(defer and lockfree)
    ctxs = [ch.g for ch in chans]
    rp = Replay(ctxs, replay_layers(chans, layers))
    n_in = 784
    ins = np.stack([split_columns(g, ch.h_in, n_in) for g, ch in zip(ctxs, chans)])
    for g in ctxs:
        g.set_option("defer", (2 if lockfree else 1) if defer else 0)
        g.sync()
    words = None
    global LAST_LAUNCHES
    try:
        for it in range(warmup + steps):
            if it == warmup:
                for g in ctxs:
                    g.sync()
                if os.environ.get("REPLAY_OFFSET") and len(chans) > 1:      # experiment: ONE device-side offset between the primes at the start of the window, no ordering afterwards:
                    for _ in range(int(os.environ["REPLAY_OFFSET"])):       # prime 0's stream gets `REPLAY_OFFSET` batched fronts of filler work (convolution + Multiply of 845), every other prime waits for it
                        chans[0].front()
                    for ch in chans[1:]:
                        ch.g.wait_for(chans[0].g)
                launches0 = sum(g.stats()["kernel_launches"] for g in ctxs)
                cg0, cpu0 = _cgroup_cpu_stat(), time.process_time()
                t0 = time.perf_counter()
            ts = time.perf_counter()
            out = rp.run(ins, threads, literal_taps=literal_taps, nonce0=1 + it * 100000, merged=merged, direct_free=bool(direct_free))
            if it >= warmup and os.environ.get("REPLAY_HOST_TIMES"):       # when did the CALLERS finish batch `it` (no wait for the device)?  the last line of the list is the device's time
                HOST_TIMES.append(round(1e3 * (time.perf_counter() - t0), 1))
            if per_step:
                for g in ctxs:
                    g.sync()
                print("    step %d: %.1f ms" % (it, 1e3 * (time.perf_counter() - ts)), file=sys.stderr)
            if it == warmup + steps - 1:
                for g in ctxs:
                    g.sync()
                dt = time.perf_counter() - t0
                if os.environ.get("REPLAY_HOST_TIMES"):
                    print("callers returned at (ms): %s; device done at %.1f" % (HOST_TIMES, 1e3 * dt), file=sys.stderr)
                    del HOST_TIMES[:]
                cg1 = _cgroup_cpu_stat()
                global LAST_HOST
                LAST_HOST = {"cpu_s_per_wall_s": round((time.process_time() - cpu0) / dt, 2),
                             "cgroup_periods": cg1.get("nr_periods", 0) - cg0.get("nr_periods", 0), "cgroup_throttled_periods": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                             "cgroup_throttled_ms": round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e3, 1)} if cg0 else {"cpu_s_per_wall_s": round((time.process_time() - cpu0) / dt, 2)}
                LAST_LAUNCHES = (sum(g.stats()["kernel_launches"] for g in ctxs) - launches0) / steps       # how finely the queue was cut
                LAST_HOST["pin_ring_laps"] = [g.get_option("pin_laps") for g in ctxs]                        # (cumulative per context: each lap of the pinned upload ring waits for the stream)
                words = [np.stack([g.ct_download(int(h), 0, 1)[0] for h in out[p]]) for p, g in enumerate(ctxs)]
            for p, g in enumerate(ctxs):                 # Decrypt + Dispose of the result matrix
                for h in out[p]:
                    g.free(int(h))
    finally:
        for g in ctxs:
            g.set_option("defer", 0)
        for p, g in enumerate(ctxs):
            for h in ins[p]:
                g.free(int(h))
    return 1e3 * dt / steps, words


def decrypt_outputs(chans, words):
    """slot values [8192, O] per prime of output ciphertext words [primes][O][...] (device decryption: the contexts hold their secret key)"""
    res = []
    for ch, w in zip(chans, words):
        g = ch.g
        h, ph = g.ct_alloc(len(w)), g.pt_alloc(len(w))
        g.ct_upload(h, 0, w)
        g.decrypt(h, 0, len(w), ph, 0)
        res.append(np.ascontiguousarray(g.decode_batch(ph, 0, len(w)).T))
        g.free(h), g.free(ph)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,8,32")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--immediate", action="store_true", help="also time the unchanged caller with every call launched on its own")
    ap.add_argument("--trained", action="store_true", help="the reference's trained weights (tests/golden/cryptonets_weights.npz)")
    ap.add_argument("--literal-threads", default="", help="thread counts for the LITERAL caller: padded taps as fresh encryptions of zero (PoolLayer.cs:67-80)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed batches in front of every measurement (the first one or two batches behind a new set of input handles carry a "
                    "one-time ~55 ms - arenas and slabs settling - profiles/HISTORY.md round 4; bench.py warms up with 2 as well)")
    ap.add_argument("--per-step", action="store_true", help="print the wall time of every timed batch of the literal measurements (a sync after each)")
    ap.add_argument("--locked", action="store_true", help="cn_set_option(defer, 1): every deferred call under the context lock (rounds 2-5) instead of the lock-free submission ring")
    ap.add_argument("--direct-free", action="store_true", help="with --locked: release every disposed array at once (the lock-free twin's calls) instead of 32 at a time")
    ap.add_argument("--no-merged", action="store_true", help="the twin's round-3 calls: cn_ct_alloc + cn_encrypt per zero vector, one cn_free per disposed array")
    args = ap.parse_args()
    from cryptonets_amd._native import Context
    from cryptonets_amd import cryptonets_mnist as cm
    if args.trained:
        w = np.load(os.path.join(ROOT, "tests", "golden", "cryptonets_weights.npz"))
        layers = cm.layer_tables(w["Weights_0"], w["Weights_1"], w["Biases_2"], w["Weights_3"], w["Biases_3"])
    else:
        layers = cm.layer_tables(*cm.synthetic_weights(1))
    images = cm.synthetic_images(cm.N, seed=1000)
    x_int = np.rint(images * cm.NORMALIZATION * cm.INPUT_SCALE).astype(np.int64)
    chans = []
    for p in cm.PLAIN_PRIMES:
        g = Context(cm.N, p, dbc=10, gdbc=20, device=0)
        g.keygen(0xC0FFEE ^ p, galois=False)
        ch = cm.CryptoNetsChannel(g, layers, cm.constant_plaintext(cm.N))
        ph = g.pt_alloc(784)
        for c in range(784):
            g.encode(np.mod(x_int[:, c], p).astype(np.uint64), ph, c)          # (one call per column: as the reference's EncryptLayer does)
        g.encrypt(ph, 0, ch.h_in, 0, 784, seed=0xFEED)
        g.free(ph)
        chans.append(ch)
    for ch in chans:
        ch.forward()
    for ch in chans:
        ch.g.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for ch in chans:
            ch.forward()
    for ch in chans:
        ch.g.sync()
    batched_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    ref = [ch.g.ct_download(ch.h5, 0, 10) for ch in chans]
    rows = [dict(caller="batched (bench.py)", threads=1, ms_per_batch=round(batched_ms, 2), images_per_s=round(8192e3 / batched_ms, 1), words_identical=True)]
    for t in [int(x) for x in args.threads.split(",")]:
        ms, words = measure(chans, layers, t, args.steps, warmup=args.warmup, merged=not args.no_merged, lockfree=not args.locked, direct_free=True if args.direct_free else None)
        same = all(np.array_equal(a, b) for a, b in zip(words, ref))
        rows.append(dict(caller="unchanged (per-ciphertext calls), deferred submission", threads=t, ms_per_batch=round(ms, 2),
                         images_per_s=round(8192e3 / ms, 1), frac_of_batched=round(batched_ms / ms, 3), words_identical=same, launches_per_batch=LAST_LAUNCHES, host=LAST_HOST))
    for t in [int(x) for x in args.literal_threads.split(",") if x]:
        ms, words = measure(chans, layers, t, args.steps, warmup=args.warmup, literal_taps=True, merged=not args.no_merged, per_step=args.per_step, lockfree=not args.locked, direct_free=True if args.direct_free else None)
        dec = decrypt_outputs(chans, words)
        same = all(np.array_equal(d, cm.model_mod_p_dense(x_int, layers, ch.g.t)) for d, ch in zip(dec, chans))
        rows.append(dict(caller="unchanged, padded taps as fresh encryptions of zero (PoolLayer.ElementAt), deferred submission", threads=t, ms_per_batch=round(ms, 2),
                         images_per_s=round(8192e3 / ms, 1), frac_of_batched=round(batched_ms / ms, 3), words_identical=same, launches_per_batch=LAST_LAUNCHES, host=LAST_HOST,
                         note="words_identical here = every decrypted slot of every output equals the integer model (fresh randomness: words cannot match)"))
    if args.immediate:
        ms, words = measure(chans, layers, 8, 1, warmup=1, defer=False)
        same = all(np.array_equal(a, b) for a, b in zip(words, ref))
        rows.append(dict(caller="unchanged (per-ciphertext calls), every call launched on its own", threads=8, ms_per_batch=round(ms, 2),
                         images_per_s=round(8192e3 / ms, 1), frac_of_batched=round(batched_ms / ms, 3), words_identical=same))
    for r in rows:
        print(json.dumps(r))
    if not all(r["words_identical"] for r in rows):
        raise SystemExit("ciphertext words differ between the callers")


if __name__ == "__main__":
    main()
