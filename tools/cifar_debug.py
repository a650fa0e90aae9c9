import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_backend import make_factory
from cryptonets_amd.convolution import ConvolutionEngine
from cryptonets_amd.hewrapper import EVectorFormat
from cryptonets_amd.layers import EncryptLayer, LLConvReader, LLDenseLayer, LLPoolLayer, LLVectorizeLayer, SquareActivation
from test_lola_cifar import PRIMES, mulmod, dense_weights
rng = np.random.default_rng(5)
Factory = make_factory("gpu", primes=PRIMES, n=16384, dbc=60, gdbc=60, small_modulus_count=8, galois=True)
env = Factory.AllocateComputationEnv()
img = rng.integers(0, 256, size=3 * 32 * 32).astype(float)
w0 = np.rint(rng.normal(0, 0.05, 83 * 192) * 256) / 256
b0 = np.rint(rng.normal(0, 0.05, 83) * 256) / 256
conv = dict(InputShape=[3, 32, 32], KernelShape=[3, 8, 8], Upperpadding=[0, 1, 1], Lowerpadding=[0, 1, 1], Stride=[1000, 2, 2])
reader = LLConvReader(Features=img, Scale=8.0, NormalizationFactor=1.0 / 256.0, Factory=Factory, **conv)
enc = EncryptLayer(Source=reader)
c1 = LLPoolLayer(Source=enc, MapCount=[83, 1, 1], WeightsScale=256.0, Weights=w0, Bias=b0, **conv)
v2 = LLVectorizeLayer(Source=c1)
a3 = SquareActivation(Source=v2)
a3.PrepareNetwork()
x = np.rint(img / 256.0 * 8.0).astype(np.int64)
g = reader.engine.gather_table()
patches = np.where(g >= 0, x[np.maximum(g, 0)], 0)
W0i = np.rint(c1.engine.weight_windows(w0, 192) * 256).astype(np.int64)     # window order = Offsets order
B0i = np.rint(b0 * 8 * 256).astype(np.int64)
act1 = (patches @ W0i.T + B0i).T.reshape(-1)
m0 = enc.GetNext()
e0 = env.Environments[0]; p = e0.plainmodulusValue
col0 = m0.GetColumn(0).eVectors[0]._decrypt_ints(e0)
print("input col0 ok:", [int(v) for v in col0] == [int(v) % p for v in patches[:, 0]])
m1 = c1.Apply(m0)
got = [m1.GetColumn(k).eVectors[0]._decrypt_ints(e0) for k in range(83)]
exp = np.mod(act1.reshape(83, 196), p)
bad = [k for k in range(83) if [int(v) for v in got[k]] != [int(v) for v in exp[k]]]
print("conv maps wrong:", bad[:10], len(bad))
m2 = v2.Apply(m1)
gv = m2.GetColumn(0).eVectors[0]._decrypt_ints(e0)
ev = [int(v) for v in np.mod(act1, p)]
diff = [i for i in range(len(ev)) if int(gv[i]) != ev[i]]
print("vectorize dim", len(gv), "wrong slots:", len(diff), diff[:20])
m3 = a3.Apply(m2)
gs = m3.GetColumn(0).eVectors[0]._decrypt_ints(e0)
es = [int(v) * int(v) % p for v in ev]
diff = [i for i in range(len(es)) if int(gs[i]) != es[i]]
print("square wrong slots:", len(diff), diff[:20])

def budget(vec, i=0):
    e = env.Environments[i]; o = e.client.o
    ct = e.ctx.ct_download(vec.eVectors[i].encData.h, vec.eVectors[i].encData.first, 1)[0]
    xs = o.dot_with_secret(ct).reshape(o.k, o.n)
    Q = 1
    for q in o.q: Q *= q
    coef = [ (Q // q) * pow(Q // q, -1, q) for q in o.q]
    worst = 0
    for c in range(0, o.n, 37):
        X = sum(int(xs[j, c]) * coef[j] for j in range(o.k)) % Q
        v = (X * o.t) % Q
        if v > Q // 2: v -= Q
        worst = max(worst, abs(v))
    import math
    return math.log2(Q) - math.log2(2 * worst + 1)
print("budget after square:", budget(m3.GetColumn(0)))
w1 = np.rint(rng.normal(0, 0.02, 112 * 8300) * 512) / 512
b1 = np.rint(rng.normal(0, 0.05, 112) * 512) / 512
eng = ConvolutionEngine([83, 14, 14], [83, 10, 10], [83, 2, 2], Upperpadding=[0, 4, 4], Lowerpadding=[0, 4, 4], MapCount=[112, 1, 1])
W1 = dense_weights(eng, w1)
d4 = LLDenseLayer(Source=a3, WeightsScale=512.0, Weights=W1.reshape(-1), Bias=eng.GetDenseBias(b1), InputFormat=EVectorFormat.dense, ForceDenseFormat=True)
d4.Prepare()
m4 = d4.Apply(m3)
print("budget after dense4:", budget(m4.GetColumn(0)))
g4 = m4.GetColumn(0).eVectors[0]._decrypt_ints(e0)
pp = np.uint64(p)
a1 = np.array(es, dtype=np.uint64)
W1i = np.rint(W1 * 512).astype(np.int64)
W1p = np.mod(W1i, p).astype(np.uint64)
acc = np.zeros(5488, dtype=np.uint64)
for c0 in range(0, 16268, 512):
    acc = (acc + (mulmod(W1p[:, c0:c0 + 512], a1[None, c0:c0 + 512], pp) % pp).sum(axis=1) % pp) % pp
s1 = (8 * 256) ** 2
B1i = [int(round(float(b) * s1 * 512)) for b in eng.GetDenseBias(b1)]
e4 = (acc + np.array([b % p for b in B1i], dtype=np.uint64)) % pp
diff = [i for i in range(5488) if int(g4[i]) != int(e4[i])]
print("dense4 dim", len(g4), "wrong:", len(diff), diff[:10])
nb = (acc % pp)
diff2 = [i for i in range(5488) if int(g4[i]) != int(nb[i])]
print("dense4 vs model without bias wrong:", len(diff2))
