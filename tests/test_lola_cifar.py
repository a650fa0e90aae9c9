"""LoLa-CIFAR shapes (BASELINE config 5, `CifarCryptoNet/LolaCifarCryptoNet.cs:21-168`): N=16384, 8 RNS limbs, dbc 60/60, plaintext
primes {957181001729, 957181034497}; 3x32x32 image -> LLPoolLayer (83 maps of 3x8x8, stride 2) -> Vectorize (83x196 = 16268 slots,
both batching rows) -> Square -> LLDenseLayer(ForceDenseFormat) with the 5488 x 16268 unrolled convolution matrix (per row: dense
MultiplyPlain + RotateColumns + 13 RotateRows + mask) -> Square -> LLDenseLayer 10 x 5488.  The trained CifarWeight.csv is a
missing blob of the reference (.MISSING_LARGE_BLOBS:2): synthetic weights of the same shapes.

GPU only (77 k key switches per plaintext prime); decrypted residues must equal the Z_p integer model.

Noise: with the reference's 8 limbs (389-bit q) the measured invariant-noise budget is 190 bits after the first square and 87
bits after the 5488-row dense layer; the second square (-50) and the last dense MultiplyPlain (-46) would leave none for our
synthetic full-range plaintexts, so the 8-limb case is checked through the big dense layer (all 5488 outputs) and the complete
network is checked with all 9 primes of CoeffModulus128(16384) (438-bit q)."""
import numpy as np
import pytest

from cryptonets_amd.convolution import ConvolutionEngine
from cryptonets_amd import networks

PRIMES = (957181001729, 957181034497)


def mulmod(a, b, p):
    b = np.asarray(b, dtype=np.uint64)
    hi = (a * (b >> np.uint64(20))) % p
    return (hi * np.uint64(1 << 20) + a * (b & np.uint64(0xFFFFF))) % p


@pytest.mark.gpu
@pytest.mark.parametrize("limbs", [8, 9])
def test_lola_cifar_shapes_end_to_end(limbs):
    from oracle_backend import make_factory
    rng = np.random.default_rng(5)
    Factory = make_factory("gpu", primes=PRIMES, n=16384, dbc=60, gdbc=60, small_modulus_count=limbs, galois=True)
    env = Factory.AllocateComputationEnv()
    img = rng.integers(0, 256, size=3 * 32 * 32).astype(float)
    w0 = np.rint(rng.normal(0, 0.05, 83 * 192) * 256) / 256
    b0 = np.rint(rng.normal(0, 0.05, 83) * 256) / 256
    w1 = np.rint(rng.normal(0, 0.02, 112 * 8300) * 512) / 512
    b1 = np.rint(rng.normal(0, 0.05, 112) * 512) / 512
    w2 = np.rint(rng.normal(0, 0.05, 10 * 5488) * 512) / 512
    b2 = np.rint(rng.normal(0, 0.05, 10) * 512) / 512
    reader = networks.cifar_reader(Factory=Factory)
    reader.Features = img / 256.0
    d6 = networks.LoLaCifar(Factory, reader, [w0, w1, w2], [b0, b1, b2], timing=False)          # LolaCifarCryptoNet.cs:58-131
    a5 = d6.Source
    d4 = a5.Source
    c1 = d4.Source.Source.Source
    eng = ConvolutionEngine([83, 14, 14], [83, 10, 10], [83, 2, 2], Upperpadding=[0, 4, 4], Lowerpadding=[0, 4, 4], MapCount=[112, 1, 1])
    W1 = eng.GetDenseWeights(w1).reshape(5488, 16268)
    d6.PrepareNetwork()
    out4 = d4.GetNext()                                             # encrypt ... big dense layer
    out = None
    if limbs == 9:
        a5v = a5.Apply(out4)
        out = d6.Apply(a5v)
        a5v.Dispose()
    # integer model modulo each plaintext prime (networks.lola_cifar_dense_model: also what bench.py --workload cifar verifies with)
    W2i = np.rint(w2.reshape(10, 5488) * 512).astype(np.int64)
    s1 = (8 * 256) ** 2
    s2 = (s1 * 512) ** 2
    B2i = [int(round(float(b) * s2 * 512)) for b in b2]
    for i, e in enumerate(env.Environments):
        p = np.uint64(e.plainmodulusValue)
        a2 = networks.lola_cifar_dense_model(reader, c1, [w0, w1, w2], [b0, b1, b2], img, int(p))
        got4 = out4.GetColumn(0).eVectors[i]._decrypt_ints(e)
        assert len(got4) == 5488 and [int(v) for v in got4] == [int(v) for v in a2], "dense 5488x16268, prime %d" % int(p)
        if out is None:
            continue
        a2 = mulmod(a2, a2, p)
        W2p = np.mod(W2i, int(p)).astype(np.uint64)
        lg = np.zeros(10, dtype=np.uint64)
        for c0 in range(0, 5488, 512):
            lg = (lg + (mulmod(W2p[:, c0:c0 + 512], a2[None, c0:c0 + 512], p) % p).sum(axis=1) % p) % p
        lg = (lg + np.array([b % int(p) for b in B2i], dtype=np.uint64)) % p
        got = out.GetColumn(0).eVectors[i]._decrypt_ints(e)
        assert [int(v) for v in got] == [int(v) for v in lg], "logits, prime %d" % int(p)
    out4.Dispose()
    if out is not None:
        out.Dispose()
