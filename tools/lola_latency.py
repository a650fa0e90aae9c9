"""LoLa-MNIST single-image latency on the GPU (BASELINE config 4): time from the encrypted input to the encrypted logits
(the reference's README figure 2.0-2.2 s covers the same window, `README.md:121-130`).  Keys, encryption and decryption on the
device (DeviceClient); the decrypted logits are checked against the exact integer model.

    python tools/lola_latency.py [LoLa|LoLaDense|LoLaSmall] [--graph]     LOLA_KS_WIDE=-1|0|1|2 selects the key-switch variant (A/B)
    --graph: additionally record the evaluation as one HIP graph per plaintext prime and time the replay on fresh encryptions
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cryptonets_amd import cryptonets_mnist as cm, networks
from cryptonets_amd.hewrapper import EncryptedSealBfvFactory

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "LoLa")
parms = dict(networks.FACTORY_PARAMETERS[name])
if "SmallModulusCount" in parms:
    parms["SmallModulusCount"] += 1                      # the reference's count runs out of noise budget
t0 = time.perf_counter()
Factory = EncryptedSealBfvFactory(**parms)
print("%s: keys on the device in %.2f s" % (name, time.perf_counter() - t0))
env = Factory.AllocateComputationEnv()
r = np.random.default_rng(3)
img = np.where(r.random(784) < 0.81, 0, r.integers(1, 256, size=784)).astype(float)
if os.environ.get("LOLA_KS_WIDE"):
    for e in env.Environments:
        e.ctx.set_option("ks_wide", int(os.environ["LOLA_KS_WIDE"]))
golden = os.path.join(ROOT, "tests", "golden")
w = np.load(os.path.join(golden, "small_model_weights.npz" if name == "LoLaSmall" else "cryptonets_weights.npz"))
tsv = "/tmp/lola_latency_one_image.tsv"
line = "7\t784\t" + "\t".join("%d:%d" % (i, int(img[i])) for i in np.nonzero(img)[0]) + "\n"
open(tsv, "w").write(line * 12)
reader = networks.lola_reader(name, tsv, Factory=Factory)
net = networks.LOLA_NETWORKS[name](Factory, reader, w)
net.PrepareNetwork()
layers = list(networks._chain(net))[::-1]                # reader, encrypt, conv, ...
def sync():
    for e in env.Environments:
        e.ctx.sync()
for rep in range(3):
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
if "--graph" in sys.argv:
    # the same evaluation recorded once as one HIP graph per plaintext prime and replayed on freshly encrypted inputs
    from cryptonets_amd.hewrapper import CapturedEvaluation
    def evaluate(x):
        for L in layers[2:]:
            y = L.Apply(x)
            if y is not x and x is not first:
                x.Dispose()
            x = y
        return x
    m.Dispose()
    first = layers[1].Apply(layers[0].GetNext())
    evaluate(first).Dispose(); sync()                    # the recorded flow once eagerly (input kept alive): its temporaries are in the pools now
    t0 = time.perf_counter(); cap = CapturedEvaluation(env, evaluate, [first]); t_cap = time.perf_counter() - t0
    for rep in range(4):
        fresh = layers[1].Apply(layers[0].GetNext()); sync()
        t0 = time.perf_counter(); m = cap.run(fresh); sync(); t_run = time.perf_counter() - t0
        fresh.Dispose()
        print("graph rep %d: evaluate %.2f ms (recording took %.1f ms, one launch per prime)" % (rep, 1e3 * t_run, 1e3 * t_cap))
if name != "LoLaSmall":
    got = [int(x) for x in m.GetColumn(0).DecryptFullPrecision(env)]
    print("logits exact:", got == cm.centred(cm.int_logits(w, img), env.bigFactor))
