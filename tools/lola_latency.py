"""LoLa-MNIST single-image latency on the GPU (BASELINE config 4): time from the encrypted input to the encrypted logits
(the reference's README figure 2.0-2.2 s includes the same window, `README.md:121-130`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_backend import make_factory
import test_lola as T

Factory = make_factory("gpu", primes=T.PRIMES, n=8192, galois=True)
env = Factory.AllocateComputationEnv()
img = T.image()
if os.environ.get("LOLA_KS_WIDE"):                       # A/B of the key-switch variants: -1 auto, 0 fused, 1 per digit, 2 per source limb
    for e in env.Environments:
        e.ctx.set_option("ks_wide", int(os.environ["LOLA_KS_WIDE"]))
net = T.lola(Factory, img)
net.PrepareNetwork()
layers = []
p = net
while p is not None:
    layers.append(p); p = p.Source
layers.reverse()            # reader, encrypt, conv, ...
def sync():
    for e in env.Environments:
        e.ctx.sync()
for rep in range(3):
    layers[0].Features = img / 256.0                      # a hand-set record is used for ONE GetNext (LLConvReader.cs:150)
    m = layers[0].GetNext()
    t0 = time.perf_counter(); m = layers[1].Apply(m); sync(); t_enc = time.perf_counter() - t0
    for e in env.Environments:
        e.ctx.stats(reset=True)
    times = []
    t_all = time.perf_counter()
    for L in layers[2:]:
        t0 = time.perf_counter(); m2 = L.Apply(m); sync(); times.append((type(L).__name__, time.perf_counter() - t0))
        if m2 is not m: m.Dispose()
        m = m2
    total = time.perf_counter() - t_all
    st = env.Environments[0].ctx.stats()
    print("rep %d: encrypt %.1f ms | evaluate %.1f ms | " % (rep, 1e3 * t_enc, 1e3 * total) + ", ".join("%s %.1f" % (n, 1e3 * t) for n, t in times))
print("per-prime op counts:", {k: v for k, v in st.items() if v})
got = m.GetColumn(0).DecryptFullPrecision(env)
exp = T.int_logits(img)
M = env.bigFactor
exp = [((v % M) - M) if (v % M) * 2 > M else (v % M) for v in exp]
print("logits exact:", [int(x) for x in got] == exp)
