#!/usr/bin/env python
"""The UNCHANGED caller of the LoLa networks through the C ABI: the per-call sequence the reference's unchanged EncryptedSealBfvMatrix /
LL*Layer files issue (one AtomicSealBfvEncryptedVector method per row / column / map - `hewrapper.LITERAL`), recorded once
(tools/call_trace.py) and replayed from C++ (tools/replay_call_trace.cpp), next to the same for this mirror's batched conveniences.

    python tools/lola_unchanged_caller.py [LoLa|LoLaSmall] [--reps 20]

Prints one JSON object per (call pattern, host) with ms per image; every replayed result is decrypted and compared with the exact integer
logits of the recorded image.  (bench.py --workload lola carries the same figures as `unchanged_caller`.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _apply_chain(layers, enc):
    """the evaluated layers of one inference (encrypted input -> encrypted logits), intermediates disposed like BaseLayer.GetNext does"""
    m = enc
    for L in layers[2:]:
        m2 = L.Apply(m)
        if m2 is not m:
            m.Dispose()
        m = m2
    return m


def measure(name="LoLa", reps=20, device=0, image_seed=1234, only=None):
    """only = "batched" | "literal": just that call pattern, replayed from free-running C++ threads (for a kernel trace of one pattern: tools/gpu_r06_q.sh)"""
    import call_trace
    from cryptonets_amd import cryptonets_mnist as cm, hewrapper, networks
    from cryptonets_amd.distributed import crt_join_over_ranks
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    parms = dict(networks.FACTORY_PARAMETERS[name], device=device)
    primes = list(parms["primes"])
    w = dict(zip(("Weights_0", "Weights_1", "Biases_2", "Weights_3", "Biases_3"), cm.reference_weights()))
    img = cm.synthetic_images(1, seed=image_seed)[0]
    tsv = "/tmp/lola_unchanged_%d.tsv" % os.getpid()
    with open(tsv, "w") as f:
        for _ in range(64):
            f.write("7\t784\t" + "\t".join("%d:%d" % (i, int(img[i])) for i in np.nonzero(img)[0]) + "\n")
    Factory = EncryptedSealBfvFactory(**parms)
    env = Factory.AllocateComputationEnv()
    reader = networks.lola_reader(name, tsv, Factory=Factory)
    net = networks.LOLA_NETWORKS[name](Factory, reader, w)
    net.PrepareNetwork()
    layers = list(networks._chain(net))[::-1]
    ctxs = [e.ctx for e in env.Environments]
    M = 1
    for p in primes:
        M *= p
    want = [int(v) for v in cm.centred(cm.int_logits(w, img), M)]

    def fresh():
        return layers[1].Apply(layers[0].GetNext())

    def sync():
        for c in ctxs:
            c.sync()

    def logits_of_handles(handles, firsts, count, sparse):
        """the 10 logits from the result handles of every prime: a sparse-format vector is `count` ciphertexts whose plaintext is the constant
        polynomial (AtomicSealBfvVector.cs:1060-1063 reads coefficient 0), a dense one is one ciphertext read through BatchEncoder.Decode"""
        res = {}
        for i, (p, c, h) in enumerate(zip(primes, ctxs, handles)):
            ph = c.pt_alloc(count)
            c.decrypt(h, firsts[i], count, ph, 0)
            if sparse:
                res[p] = np.asarray(c.pt_download(ph, 0, count)[:, 0][:10], dtype=object)
            else:
                res[p] = np.asarray(c.decode_batch(ph, 0, 1)[0][:10], dtype=object)
            c.free(ph)
        return [int(v) for v in crt_join_over_ranks(res, primes, None)]

    def result_of(out):
        from cryptonets_amd.hewrapper import EVectorFormat
        col = out.GetColumn(0)
        return col, [a.encData.first for a in col.eVectors], col.eVectors[0].encData.count, col.Format == EVectorFormat.sparse

    rows = []
    for literal in (False, True):
        if only and (only == "literal") != literal:
            continue
        hewrapper.set_literal(literal)
        try:
            for _ in range(2):                                     # warm: masks / plaintext forms cached, arenas sized, pool filled
                _apply_chain(layers, fresh()).Dispose()
            sync()
            encs = [fresh() for _ in range(2 if only else reps)]
            sync()
            t0 = time.perf_counter()
            outs = []
            for e in encs:
                outs.append(_apply_chain(layers, e))
                sync()
            py_ms = 1e3 * (time.perf_counter() - t0) / len(encs)
            ok = True
            for o in outs[-2:]:
                col, firsts, count, sparse = result_of(o)
                ok = ok and logits_of_handles([a.encData.h for a in col.eVectors], firsts, count, sparse) == want
            for o in outs:
                o.Dispose()
            label = "unchanged per-call sequence (hewrapper.LITERAL)" if literal else "batched conveniences of the mirror (RowsDotProduct, MulColumnsByPlain, MulManySparse)"
            rows.append(dict(pattern=label, host="python mirror (ctypes)", ms_per_image=round(py_ms, 2), logits_exact=bool(ok)))
            # ---- record one inference at the C ABI, replay it from C++
            enc = fresh()
            sync()
            recs = [call_trace.Recorder(c).start() for c in ctxs]
            try:
                out = _apply_chain(layers, enc)
            finally:
                for r in recs:
                    r.stop()
            col, firsts, count, sparse = result_of(out)                  # the result: `count` ciphertexts of an array the trace allocated
            rids = [r.ids[int(a.encData.h)] for r, a in zip(recs, col.eVectors)]
            assert all(rid[0] == "new" for rid in rids), rids
            calls = [len(r.records) for r in recs]
            out.Dispose()
            for defer in ((1,) if only and literal else (0, 1) if literal else (0,)):
                # defer = 1: libcnhip's deferred submission (what the C# twin switches on): the per-row calls are queued and merged level by level
                for c in ctxs:
                    c.set_option("defer", defer)
                try:
                    for mode, host in ((0, "C++ replay, one host thread, contexts call by call"), (1, "C++ replay, one thread per plaintext prime, joined after every call"),
                                       (2, "C++ replay, one free-running thread per plaintext prime")):
                        if only and mode != 2:
                            continue
                        l0 = sum(c.stats()["kernel_launches"] for c in ctxs)
                        ms, handles = call_trace.replay(recs, reps, mode, rids, warmup=2)
                        launches = (sum(c.stats()["kernel_launches"] for c in ctxs) - l0) / (reps + 2) / len(ctxs)
                        ok = logits_of_handles(handles, firsts, count, sparse) == want
                        for c, h in zip(ctxs, handles):
                            c.free(h)
                        rows.append(dict(pattern=label + (", deferred submission" if defer else ""), host=host, ms_per_image=round(ms, 2), logits_exact=bool(ok),
                                         calls_per_prime=calls[0], launches_per_prime=round(launches, 1)))
                finally:
                    for c in ctxs:
                        c.set_option("defer", 0)
        finally:
            hewrapper.set_literal(False)
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("network", nargs="?", default="LoLa")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", choices=["batched", "literal"], default=None, help="one call pattern only, replayed from free-running C++ threads (kernel traces)")
    a = ap.parse_args()
    for r in measure(a.network, a.reps, only=a.only):
        print(json.dumps(r))
