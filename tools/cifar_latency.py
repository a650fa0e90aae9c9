"""LoLa-CIFAR shapes (BASELINE config 5) single-image evaluation latency on one MI355X: encrypted input -> encrypted logits
(the reference's "Inference-Time" window, LolaCifarCryptoNet.cs:57,128; README: ~750 s on an Azure B8ms)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_backend import make_factory
from cryptonets_amd.convolution import ConvolutionEngine
from cryptonets_amd.hewrapper import EVectorFormat
from cryptonets_amd.layers import EncryptLayer, LLConvReader, LLDenseLayer, LLPoolLayer, LLVectorizeLayer, SquareActivation
from test_lola_cifar import PRIMES, dense_weights
limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(5)
t0 = time.perf_counter()
Factory = make_factory("gpu", primes=PRIMES, n=16384, dbc=60, gdbc=60, small_modulus_count=limbs, galois=True)
print("keygen (oracle client) + key upload: %.1f s" % (time.perf_counter() - t0))
env = Factory.AllocateComputationEnv()
img = rng.integers(0, 256, size=3 * 32 * 32).astype(float)
w0 = np.rint(rng.normal(0, 0.05, 83 * 192) * 256) / 256; b0 = np.rint(rng.normal(0, 0.05, 83) * 256) / 256
w1 = np.rint(rng.normal(0, 0.02, 112 * 8300) * 512) / 512; b1 = np.rint(rng.normal(0, 0.05, 112) * 512) / 512
w2 = np.rint(rng.normal(0, 0.05, 10 * 5488) * 512) / 512; b2 = np.rint(rng.normal(0, 0.05, 10) * 512) / 512
conv = dict(InputShape=[3, 32, 32], KernelShape=[3, 8, 8], Upperpadding=[0, 1, 1], Lowerpadding=[0, 1, 1], Stride=[1000, 2, 2])
reader = LLConvReader(Features=img / 256.0, Scale=8.0, Factory=Factory, **conv)
enc = EncryptLayer(Source=reader)
c1 = LLPoolLayer(Source=enc, MapCount=[83, 1, 1], WeightsScale=256.0, Weights=w0, Bias=b0, **conv)
v2 = LLVectorizeLayer(Source=c1); a3 = SquareActivation(Source=v2)
eng = ConvolutionEngine([83, 14, 14], [83, 10, 10], [83, 2, 2], Upperpadding=[0, 4, 4], Lowerpadding=[0, 4, 4], MapCount=[112, 1, 1])
d4 = LLDenseLayer(Source=a3, WeightsScale=512.0, Weights=dense_weights(eng, w1).reshape(-1), Bias=eng.GetDenseBias(b1), InputFormat=EVectorFormat.dense, ForceDenseFormat=True)
a5 = SquareActivation(Source=d4)
d6 = LLDenseLayer(Source=a5, Weights=w2, Bias=b2, WeightsScale=512.0, InputFormat=EVectorFormat.dense)
t0 = time.perf_counter(); d6.PrepareNetwork(); print("prepare (encode 5498 weight rows x 2 primes): %.1f s" % (time.perf_counter() - t0))
def sync():
    for e in env.Environments: e.ctx.sync()
layers = [c1, v2, a3, d4, a5, d6]
for rep in range(3):
    reader.Features = img / 256.0
    m = enc.GetNext(); sync()
    times = []; t_all = time.perf_counter()
    for L in layers:
        t0 = time.perf_counter(); m2 = L.Apply(m); sync(); times.append((type(L).__name__, time.perf_counter() - t0))
        m.Dispose(); m = m2
    print("rep %d (%d limbs): evaluate %.1f ms | " % (rep, limbs, 1e3 * (time.perf_counter() - t_all)) + ", ".join("%s %.1f" % (n, 1e3 * t) for n, t in times))
    m.Dispose()
print("per-prime op counts:", {k: v for k, v in env.Environments[0].ctx.stats().items() if v})
