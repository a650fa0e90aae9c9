"""Invariant-noise budget after every layer of LoLa-CIFAR at the reference's parameters (N = 16384, 8 limbs = 389-bit q, dbc 60, t ~ 2^39.8;
LolaCifarCryptoNet.cs:35,58-131) next to the textbook BFV average-case estimates for the same operation sequence.

Measured: Decryptor.InvariantNoiseBudget (cn_noise_poly: || t (c0 + c1 s) mod q ||_inf, centred) of the ciphertexts a layer hands on - the
minimum over (up to 24 of) them - on the device contexts the bench line uses (own keys, ChaCha20 sampler with sigma 3.2 clipped at 6 sigma).
Model (average case; n = 16384, sigma = 3.2, ternary secret; "max over n coefficients of a Gaussian" ~ 4.3 standard deviations):
  fresh             v = (t/q) e - (r_t(q)/q) [m]: for a BatchEncoded (full-range) plaintext the second term dominates   budget ~ log2 q - 2 log2 t + 1
  scalar MAC        sum_k w_k c_k: noise std x sqrt(sum w_k^2)                            - 0.5 log2(sum w^2)
  dense MultiplyPlain  plaintext coefficients ~ uniform mod t (BatchEncoder): x sqrt(n) t / sqrt(12)   - log2(t) - 0.5 log2(n / 12)
  key switch        adds  (t/q) k 2^min(dbc, 49) sigma sqrt(n/12 ...) - a FLOOR near log2(q/t) - 62 bits, invisible above it
  rotate-and-add    x + rot(x): independent coefficient positions: x sqrt(2) per link            - 0.5 per link
  BFV multiply      v ~ t sqrt(n/12 ...) (|m1| v2 + |m2| v1) with |m| ~ t/sqrt(12) per coefficient   - log2(t) - 0.5 log2(n) - ~1  (squaring: one more bit)
The point of the comparison: if the measured consumption of an operation exceeded its estimate by more than the max-vs-std slack, the restatement
(sampler, key switch, BEHZ multiply) would be noisier than SEAL's; if not, the trail is what these parameters give ANY exact BFV implementation.

    python tools/cifar_noise_trail.py [limbs=8]
"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cryptonets_amd import networks
from cryptonets_amd.hewrapper import EncryptedSealBfvFactory

limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(5)
qz = lambda a, s: np.rint(a * s) / s
W = [qz(rng.normal(0, 0.05, 83 * 192), 256), qz(rng.normal(0, 0.02, 112 * 8300), 512), qz(rng.normal(0, 0.05, 10 * 5488), 512)]
B = [qz(rng.normal(0, 0.05, 83), 256), qz(rng.normal(0, 0.05, 112), 512), qz(rng.normal(0, 0.05, 10), 512)]
img = rng.integers(0, 256, size=3 * 32 * 32).astype(float)
parms = dict(networks.FACTORY_PARAMETERS["LoLaCifar"], SmallModulusCount=limbs)
Factory = EncryptedSealBfvFactory(**parms)
env = Factory.AllocateComputationEnv()
reader = networks.cifar_reader(Factory=Factory)
net = networks.LoLaCifar(Factory, reader, W, B, timing=False)
net.PrepareNetwork()
layers = list(networks._chain(net))[::-1]
ctx0 = env.Environments[0].ctx
n, t = ctx0.n, ctx0.t
logq = sum(math.log2(int(q)) for q in ctx0.q)
sigma = 3.2


def budget(m):
    """min / mean invariant-noise budget (bits) over the ciphertexts of a layer's output, every plaintext prime, up to 24 ciphertexts per vector"""
    vals = []
    for col in range(m.ColumnCount):
        v = m.GetColumn(col)
        for a, e in zip(v.eVectors, env.Environments):
            d = a.encData
            cnt = min(d.count, 24)
            idx = np.linspace(0, d.count - 1, cnt).astype(int)
            for i in idx:
                vals += e.ctx.invariant_noise_budget(d.h, d.first + int(i), 1)
    return min(vals), sum(vals) / len(vals), len(vals)


# fresh: v = (t/q) e - (r_t(q)/q) [m]  with e ~ 4.3 sigma sqrt(4n/3) and, for a BatchEncoded plaintext, centred coefficients up to t/2 and r_t(q) up to t:
# the second term dominates (t^2 / 4q against t 2^11 / q): budget ~ log2 q - 2 log2 t + 1
fresh = logq - 2 * math.log2(t) + 1
w0 = np.rint(W[0] * 256)
mac1 = 0.5 * math.log2(float(np.mean(np.sum(w0.reshape(83, 192) ** 2, axis=1))))
mulp = math.log2(t) + 0.5 * math.log2(n / 12)
mul = math.log2(t) + 0.5 * math.log2(n) + 1
model = {"EncryptLayer": ("fresh: log2 q - 2 log2 t + 1 (the r_t(q) [m] / q term of a full-range plaintext)", fresh),
         "LLPoolLayer": ("scalar MAC, 192 taps x weights of ~%.0f" % np.sqrt(np.mean(w0 ** 2)), -mac1),
         "LLVectorizeLayer": ("per map: mask (dense MultiplyPlain) + rotation, 83 maps added", -mulp - 0.5 * math.log2(83)),
         "SquareActivation": ("BFV square + relinearise: log2 t + 0.5 log2 n + ~2 ... log2 t + log2(n/2)", -mul - 1),
         "LLDenseLayer": ("per row: dense MultiplyPlain + 13-14 rotate-and-add links (+ ForceDenseFormat: one-hot mask MultiplyPlain, 5488 rows added)", None)}
dense_terms = [-(mulp + 7 + mulp + 0.5 * math.log2(5488)), -(mulp + 6.5)]          # layer 4 (ForceDenseFormat), layer 6
print("LoLa-CIFAR, N = %d, %d limbs: log2 q = %.1f, log2 t = %.1f, log2(q/t) = %.1f; key-switch floor ~ %.0f bits" % (n, limbs, logq, math.log2(t), logq - math.log2(t),
      logq - math.log2(t) - 1 - math.log2(limbs * 2.0 ** 49 * sigma * math.sqrt(n) * 4.3 / math.sqrt(12))))
print("%-22s %10s %10s %8s | %10s %10s   %s" % ("layer output", "min bits", "mean bits", "probes", "consumed", "model", "model term"))
reader.Features = img / 256.0
m = layers[1].GetNext()
prev, est = None, None
for L, name in zip([None] + layers[2:], [type(layers[1]).__name__] + [type(x).__name__ for x in layers[2:]]):
    if L is not None:
        m2 = L.Apply(m)
        if m2 is not m:
            m.Dispose()
        m = m2
    for e in env.Environments:
        e.ctx.sync()
    lo, mean, cnt = budget(m)
    term, d = model[name]
    if d is None:
        d = dense_terms.pop(0)
    est = d if prev is None else est + d
    print("%-22s %10.1f %10.1f %8d | %10s %10.1f   %s" % (name, lo, mean, cnt, "" if prev is None else "%.1f" % (prev - lo), d if prev is not None else d, term), flush=True)
    if prev is None:
        est = d
    prev = lo
    if lo < 1 and L is not None:
        print("(no budget left: stopping)")
        break
print("model trail (sum of the terms): %.0f bits left behind the last layer probed; measured %.1f (a measured budget cannot go below ~0: the noise has wrapped)" % (est, prev))
