"""LoLa-CIFAR shapes (BASELINE config 5) single-image evaluation latency on one MI355X: encrypted input -> encrypted logits
(the reference's "Inference-Time" window, LolaCifarCryptoNet.cs:57,128; README: ~750 s on an Azure B8ms).  Synthetic model of the
reference's shapes (CifarWeight.csv is a missing blob); keys / encryption on the device; exactness is the test-suite's job
(tests/test_lola_cifar.py).

    python tools/cifar_latency.py [limbs=8]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cryptonets_amd import networks
from cryptonets_amd.hewrapper import EncryptedSealBfvFactory

limbs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(5)
q = lambda a, s: np.rint(a * s) / s
W = [q(rng.normal(0, 0.05, 83 * 192), 256), q(rng.normal(0, 0.02, 112 * 8300), 512), q(rng.normal(0, 0.05, 10 * 5488), 512)]
B = [q(rng.normal(0, 0.05, 83), 256), q(rng.normal(0, 0.05, 112), 512), q(rng.normal(0, 0.05, 10), 512)]
img = rng.integers(0, 256, size=3 * 32 * 32).astype(float)
parms = dict(networks.FACTORY_PARAMETERS["LoLaCifar"], SmallModulusCount=limbs)
t0 = time.perf_counter()
Factory = EncryptedSealBfvFactory(**parms)
print("keys on the device: %.1f s" % (time.perf_counter() - t0))
env = Factory.AllocateComputationEnv()
reader = networks.cifar_reader(Factory=Factory)
net = networks.LoLaCifar(Factory, reader, W, B, timing=False)
t0 = time.perf_counter(); net.PrepareNetwork(); print("prepare (encode 5498 weight rows x 2 primes): %.1f s" % (time.perf_counter() - t0))
layers = list(networks._chain(net))[::-1]                # reader, encrypt, conv, vectorize, square, dense, square, dense
def sync():
    for e in env.Environments: e.ctx.sync()
for rep in range(3):
    reader.Features = img / 256.0
    m = layers[1].GetNext(); sync()
    times = []; t_all = time.perf_counter()
    for L in layers[2:]:
        t0 = time.perf_counter(); m2 = L.Apply(m); sync(); times.append((type(L).__name__, time.perf_counter() - t0))
        m.Dispose(); m = m2
    print("rep %d (%d limbs): evaluate %.1f ms | " % (rep, limbs, 1e3 * (time.perf_counter() - t_all)) + ", ".join("%s %.1f" % (n, 1e3 * t) for n, t in times))
    m.Dispose()
print("per-prime op counts:", {k: v for k, v in env.Environments[0].ctx.stats().items() if v})
