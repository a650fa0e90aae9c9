#!/usr/bin/env python
"""Shader clock and socket power while ONE kind of kernel runs back to back (rocm-smi sampled from a second thread): the fused key switch of
845 ciphertexts, the batched N=8192 NTT, a streaming add - is the FP64-bound key switch running at the clock the issue-rate floors assume?

    python tools/clock_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from cryptonets_amd._native import Context


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "-c", "-P"], capture_output=True, text=True, timeout=10).stdout
        except Exception as ex:                                   # noqa: BLE001
            out.append(("error", str(ex)))
            return
        sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
        mclk = re.findall(r"mclk clock level: \S+ \((\d+)Mhz\)", txt)
        pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", txt)
        out.append((sclk[:1], mclk[:1], pw[:1], txt if not sclk else ""))
        time.sleep(0.2)


def main():
    g = Context(8192, 549764251649)
    rng = np.random.default_rng(1)
    kw = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(g.key_words() // g.ctw) for _ in range(2) for q in g.q])
    g.set_relin_key(kw)
    cnt = 845
    h3, h2 = g.ct_alloc(cnt, 3), g.ct_alloc(cnt)
    one = np.concatenate([rng.integers(0, q, size=g.n, dtype=np.uint64) for _ in range(3) for q in g.q])
    g.ct_upload(h3, 0, np.tile(one, (cnt, 1)))
    g.ct_upload(h2, 0, np.tile(one[: 2 * g.k * g.n], (cnt, 1)))
    ptr, _ = g.device_ptr(h2)
    work = {
        "idle": lambda: time.sleep(0.01),
        "fused key switch (845 ciphertexts)": lambda: g.relinearize(h3, 0, h2, 0, cnt),
        "batched forward NTT (8450 limbs)": lambda: g.ntt_time(ptr, cnt * 2 * g.k, 0, False, 10),
        "streaming add (845 ciphertexts)": lambda: g.add(h2, 0, h2, 0, h2, 0, cnt),
    }
    for name, fn in work.items():
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        t0 = time.perf_counter()
        n = 0
        fn(); g.sync()
        th.start()
        while time.perf_counter() - t0 < 3.0:
            for _ in range(20):
                fn()
            g.sync()
            n += 20
        stop.set(); th.join()
        dt = time.perf_counter() - t0
        s = [int(x[0][0]) for x in out if x[0] and x[0] != "error" and x[0][0:1]]
        p = [float(x[2][0]) for x in out if len(x) > 2 and x[2]]
        print("%-40s %6d launches in %.1f s | sclk MHz min %s median %s max %s | power W median %s | samples %d" % (
            name, n, dt, min(s) if s else None, sorted(s)[len(s) // 2] if s else None, max(s) if s else None, sorted(p)[len(p) // 2] if p else None, len(out)))
        if not s and out:
            print("   (rocm-smi output not understood) ", str(out[0])[:400])


if __name__ == "__main__":
    main()
