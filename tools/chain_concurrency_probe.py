#!/usr/bin/env python
"""How well do the plaintext-prime chains of one LoLa inference run side by side?  The recorded call sequence of ONE image (batched
conveniences, tools/call_trace.py) replayed from C++ on 1, 2, 3 and 4 of the four contexts at once (one free-running host thread per
context): ms per image.  A chain is ~222 dependent launches of 5-60 us kernels.

    python tools/chain_concurrency_probe.py [LoLa]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main(name="LoLa", reps=20):
    import call_trace
    from lola_unchanged_caller import _apply_chain
    from cryptonets_amd import cryptonets_mnist as cm, networks
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    parms = dict(networks.FACTORY_PARAMETERS[name], device=0)
    w = dict(zip(("Weights_0", "Weights_1", "Biases_2", "Weights_3", "Biases_3"), cm.reference_weights()))
    img = cm.synthetic_images(1, seed=1234)[0]
    tsv = "/tmp/chain_probe_%d.tsv" % os.getpid()
    with open(tsv, "w") as f:
        for _ in range(8):
            f.write("7\t784\t" + "\t".join("%d:%d" % (i, int(img[i])) for i in np.nonzero(img)[0]) + "\n")
    Factory = EncryptedSealBfvFactory(**parms)
    env = Factory.AllocateComputationEnv()
    reader = networks.lola_reader(name, tsv, Factory=Factory)
    net = networks.LOLA_NETWORKS[name](Factory, reader, w)
    net.PrepareNetwork()
    layers = list(networks._chain(net))[::-1]
    ctxs = [e.ctx for e in env.Environments]
    for _ in range(2):
        _apply_chain(layers, layers[1].Apply(layers[0].GetNext())).Dispose()
    enc = layers[1].Apply(layers[0].GetNext())
    for c in ctxs:
        c.sync()
    recs = [call_trace.Recorder(c).start() for c in ctxs]
    try:
        out = _apply_chain(layers, enc)
    finally:
        for r in recs:
            r.stop()
    col = out.GetColumn(0)
    rids = [r.ids[int(a.encData.h)] for r, a in zip(recs, col.eVectors)]
    out.Dispose()
    print("stream tries:", [c.get_option("stream_tries") for c in ctxs])
    import collections
    names = {v: k for k, v in call_trace.OP.items()}
    hist = collections.Counter(names[r[0]] for r in recs[0].records)
    print("calls of one image on one context:", dict(hist))
    if os.environ.get("CHAIN_PROBE_DUMP"):
        for r in recs[0].records:
            print("   ", names[r[0]], r[1])
    for sel in ([0], [1], [2], [3], [0, 1], [2, 3], [0, 1, 2], [0, 1, 2, 3]):
        l0 = sum(ctxs[i].stats()["kernel_launches"] for i in sel)
        ms, handles = call_trace.replay([recs[i] for i in sel], reps, 2, [rids[i] for i in sel], warmup=2)
        launches = (sum(ctxs[i].stats()["kernel_launches"] for i in sel) - l0) / (reps + 2) / len(sel)
        for i, h in zip(sel, handles):
            ctxs[i].free(h)
        print("contexts %-14s %6.2f ms per image  (%.0f launches per chain, %.1f us per chain step)" % (sel, ms, launches, 1e3 * ms / launches))


if __name__ == "__main__":
    main(*(sys.argv[1:2]))
