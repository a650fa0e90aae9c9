"""Encryptor.Encrypt of 784 ciphertexts (the EncryptLayer of CryptoNets: one per pixel column) and of 645 zero vectors (the padded taps of one batch) on the device:
HIP-event time per call for the three forms of the kernel behind the samplers - cn_set_option("enc_fused", 0 | 1 | 2): expand + batched transform + k_encrypt_tail,
k_encrypt_fused (block = (ciphertext, limb): u transformed once, both components), k_encrypt_split (block = (ciphertext, component, limb): two workgroups per CU).
Under rocprofv3 --kernel-trace --stats the kernels' own durations are in the summary."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cryptonets_amd._native import Context
g = Context(8192, 549764251649)
g.keygen(5, galois=False)
rng = np.random.default_rng(1)
ph, ch = g.pt_alloc(784), g.ct_alloc(784)
g.pt_upload(ph, 0, rng.integers(0, g.t, size=(784, g.n), dtype=np.uint64))
for mode in (1, 2, 0, 1, 2):
    g.set_option("enc_fused", mode)
    g.encrypt(ph, 0, ch, 0, 784, seed=1); g.sync()
    g.time_begin()
    for i in range(5):
        g.encrypt(ph, 0, ch, 0, 784, seed=2 + i)
    t784 = g.time_end() / 5
    g.time_begin()
    for i in range(5):
        g.encrypt(0, 0, ch, 0, 645, seed=20 + i)
    t645 = g.time_end() / 5
    print("enc_fused=%d: %.3f ms per 784 encryptions, %.3f ms per 645 encryptions of zero (samplers included)" % (mode, t784, t645), flush=True)
