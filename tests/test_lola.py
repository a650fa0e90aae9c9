"""LoLa-MNIST (BASELINE config 4, `LowLatencyCryptoNets/LoLaCryptonets.cs:203-278`): ONE image through
LLPoolLayer -> Vectorize (Stack/Interleave) -> Square -> Duplicate(8) -> LLPackedDense (13 x [dense MultiplyPlain +
SumAllSlots(1024)]) -> LLInterleave(-1) -> Square -> LLInterleavedDense, with the reference's trained weights, plaintext primes
{557057, 638977, 737281, 786433}, N=8192, dbc 10/20.  Every rotation is a Galois key switch (HOT LOOP C).

Bar: the 10 decrypted logits equal exact integer arithmetic on the scaled inputs/weights (integer equality)."""
import os

import numpy as np
import pytest

from oracle_backend import make_factory
from cryptonets_amd.layers import (EncryptLayer, LLConvReader, LLDuplicateLayer, LLInterleavedDenseLayer, LLInterleaveLayer, LLPackedDenseLayer,
                                   LLPoolLayer, LLVectorizeLayer, SquareActivation)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cryptonets_weights.npz")
PRIMES = (557057, 638977, 737281, 786433)


def lola(Factory, image):
    w = np.load(GOLD)
    w1 = w["Weights_1"]
    w1t = np.zeros_like(w1)
    for i in range(845):
        w1t[i + 845 * np.arange(100)] = w1[100 * i + np.arange(100)]
    conv = dict(InputShape=[28, 28], KernelShape=[5, 5], Upperpadding=[1, 1], Stride=[2, 2])
    reader = LLConvReader(Features=image, Scale=16.0, NormalizationFactor=1.0 / 256.0, Factory=Factory, **conv)
    enc = EncryptLayer(Source=reader)
    c1 = LLPoolLayer(Source=enc, MapCount=[5, 1], WeightsScale=32, Weights=w["Weights_0"], **conv)
    v2 = LLVectorizeLayer(Source=c1)
    a3 = SquareActivation(Source=v2)
    d4 = LLDuplicateLayer(Source=a3, Count=8)
    d5 = LLPackedDenseLayer(Source=d4, Weights=w1t, Bias=w["Biases_2"], WeightsScale=32 * 32, PackingCount=8, PackingShift=1024)
    sel = [1023 + i * 1024 for i in range(8)]
    i6 = LLInterleaveLayer(Source=d5, Shift=-1, SelectedIndices=sel)
    a7 = SquareActivation(Source=i6)
    d8 = LLInterleavedDenseLayer(Source=a7, Weights=w["Weights_3"], Bias=w["Biases_3"], WeightsScale=32, Shift=-1, SelectedIndices=sel)
    return d8


def int_logits(image):
    """exact integer model (same rounding as the wrapper: round(v*scale))"""
    from cryptonets_amd import cryptonets_mnist as cm
    w = np.load(GOLD)
    L = cm.layer_tables(w["Weights_0"], w["Weights_1"], w["Biases_2"], w["Weights_3"], w["Biases_3"])
    act = [int(v) for v in np.rint(np.asarray(image) / 256.0 * 16.0)]
    for li, T in enumerate(L):
        out = [T["bias"][o] + sum(T["W"][o][k] * act[idx] for k, idx in enumerate(T["idx"][o]) if idx >= 0) for o in range(len(T["W"]))]
        act = [v * v for v in out] if li < 2 else out
    return act


def image(seed=3):
    r = np.random.default_rng(seed)
    return np.where(r.random(784) < 0.81, 0, r.integers(1, 256, size=784)).astype(float)


@pytest.mark.parametrize("backend", [pytest.param("cpu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def test_lola_mnist_single_image(backend):
    Factory = make_factory(backend, primes=PRIMES, n=8192, galois=True)
    env = Factory.AllocateComputationEnv()
    img = image()
    net = lola(Factory, img)
    net.PrepareNetwork()
    out = net.GetNext()
    got = out.GetColumn(0).DecryptFullPrecision(env)
    exp = int_logits(img)
    M = env.bigFactor
    exp = [((v % M) - M) if (v % M) * 2 > M else (v % M) for v in exp]
    assert [int(x) for x in got] == exp
    dec = out.Decrypt(env)
    assert dec.shape == (10, 1)


def lola_dense(Factory, tsv_path):
    """LoLa-Dense (`LowLatencyCryptoNets/LoLaCryptonets.cs:118-199`): the image arrives as ONE packed ciphertext; the im2col of the
    convolution is done homomorphically by LLPreConvLayer (masks + Permute), the rest is the LoLa pipeline with 16-fold packing."""
    from cryptonets_amd.layers import LLPreConvLayer, LLSingleLineReader
    w = np.load(GOLD)
    w1 = w["Weights_1"]
    w1t = np.zeros_like(w1)
    for i in range(845):
        w1t[i + 845 * np.arange(100)] = w1[100 * i + np.arange(100)]
    conv = dict(InputShape=[28, 28], KernelShape=[5, 5], Upperpadding=[1, 1], Stride=[2, 2])
    reader = LLSingleLineReader(tsv_path, SparseFormat=True, NormalizationFactor=1.0 / 256.0, Scale=16.0, Factory=Factory)
    enc = EncryptLayer(Source=reader)
    pre = LLPreConvLayer(Source=enc, UseAxisForBlocks=[True, True], **conv)
    c2 = LLPoolLayer(Source=pre, MapCount=[5, 1], WeightsScale=32, Weights=w["Weights_0"], HotIndices=pre.HotIndices, **conv)
    v3 = LLVectorizeLayer(Source=c2)
    a4 = SquareActivation(Source=v3)
    d5 = LLDuplicateLayer(Source=a4, Count=16)
    d6 = LLPackedDenseLayer(Source=d5, Weights=pre.RearrangeWeights(w1t), Bias=w["Biases_2"], WeightsScale=32 * 32, PackingCount=16, PackingShift=1024)
    a7 = SquareActivation(Source=d6)
    sel = [1023 + i * 1024 for i in range(16)]
    i8 = LLInterleaveLayer(Source=a7, Shift=-1, SelectedIndices=sel)
    d8 = LLInterleavedDenseLayer(Source=i8, Weights=w["Weights_3"], Bias=w["Biases_3"], WeightsScale=32, Shift=-1, SelectedIndices=sel)
    return reader, d8


@pytest.mark.gpu
def test_lola_dense_single_image(tmp_path):
    """N = 16384, dbc 60/60, plaintext primes {34359771137, 34360754177} (`:123`): exact integer logits.  GPU only - the homomorphic
    im2col is ~400 key switches at N = 16384, minutes on the CPU oracle.  The reference takes 7 coefficient primes (340 bits); measured
    here the invariant noise budget is 271 bits fresh and goes 229 (masks) -> 223 -> 174 (square) -> 123 (dense) -> 79 (square) -> 39
    (interleave masks) -> 1.7 bits after the last dense layer, so single coefficients already decrypt wrongly; with 8 primes
    (389 bits) ~50 bits remain and every logit is exact - the test uses 8."""
    img = image(5)
    nz = np.nonzero(img)[0]
    tsv = tmp_path / "one_image.tsv"
    tsv.write_text("7\t784\t" + "\t".join("%d:%d" % (i, int(img[i])) for i in nz) + "\n")
    Factory = make_factory("gpu", primes=(34359771137, 34360754177), n=16384, dbc=60, gdbc=60, small_modulus_count=8, galois=True)
    env = Factory.AllocateComputationEnv()
    reader, net = lola_dense(Factory, str(tsv))
    net.PrepareNetwork()
    out = net.GetNext()
    assert list(reader.Labels) == [7]
    got = out.GetColumn(0).DecryptFullPrecision(env)
    exp = int_logits(img)
    M = env.bigFactor
    exp = [((v % M) - M) if (v % M) * 2 > M else (v % M) for v in exp]
    assert [int(x) for x in got] == exp


def test_lola_mnist_on_the_raw_factory(tmp_path):
    """The reference's `Encrypt = false` switch (`LoLaCryptonets.cs:208`, `:124`): the same layer graphs on RawFactory.  Plain doubles
    (no modular wrap, 53-bit mantissa), so the logits equal the exact integer model to double precision."""
    from cryptonets_amd.raw import RawFactory
    img = image()
    net = lola(RawFactory(8192), img)
    net.PrepareNetwork()
    out = net.GetNext()
    exp = np.array(int_logits(img), dtype=float)
    got = np.array(out.GetColumn(0).DecryptFullPrecision(None), dtype=float)
    assert got.shape == (10,) and np.allclose(got, exp, rtol=1e-12, atol=0) and int(np.argmax(got)) == int(np.argmax(exp))
    img = image(5)
    tsv = tmp_path / "one_image.tsv"
    tsv.write_text("7\t784\t" + "\t".join("%d:%d" % (i, int(img[i])) for i in np.nonzero(img)[0]) + "\n")
    reader, net = lola_dense(RawFactory(16384), str(tsv))
    net.PrepareNetwork()
    out = net.GetNext()
    exp = np.array(int_logits(img), dtype=float)
    got = np.array(out.GetColumn(0).DecryptFullPrecision(None), dtype=float)
    assert list(reader.Labels) == [7]
    assert got.shape == (10,) and np.allclose(got, exp, rtol=1e-12, atol=0)


@pytest.mark.parametrize("backend", [pytest.param("cpu"), pytest.param("gpu", marks=pytest.mark.gpu)])
def test_small_lola_single_image(backend):
    """SmallLoLa (`LoLaCryptonets.cs:280-329`, BASELINE config 4b): N = 8192, dbc 40 / 40 (two digits per limb), plaintext primes
    {2277377, 2424833}: conv -> vectorize -> square -> LLDenseLayer (10 rows x 845, dense input).  Weights of the architecture's shapes
    (the trained SmallModel is not needed for parity); bar: exact integer logits.  The reference takes THREE coefficient primes (130
    bits): measured invariant noise budget 88 bits fresh -> 82 (conv) -> 56 (vectorize: the 40-bit key-switch digits put a floor of
    ~2^49 under the noise) -> 23 (square) -> overflow by ~12 bits in the dense layer (multiply_plain + 13 rotate-and-adds), i.e. the
    logits come out a few thousand units off; with FOUR primes 31 bits remain and they are exact - the test uses 4."""
    from cryptonets_amd.layers import LLDenseLayer
    from cryptonets_amd.hewrapper import EVectorFormat
    from cryptonets_amd.convolution import ConvolutionEngine
    rng = np.random.default_rng(11)
    w0 = rng.normal(0, 0.1, 130)                                   # 5 maps x (25 + bias)
    w1 = rng.normal(0, 0.02, 8450)
    b1 = rng.normal(0, 0.1, 10)
    img = image(9)
    conv = dict(InputShape=[28, 28], KernelShape=[5, 5], Upperpadding=[1, 1], Stride=[2, 2])
    Factory = make_factory(backend, primes=(2277377, 2424833), n=8192, dbc=40, gdbc=40, small_modulus_count=4, galois=True)
    env = Factory.AllocateComputationEnv()
    reader = LLConvReader(Features=img, Scale=16.0, NormalizationFactor=1.0 / 256.0, Factory=Factory, **conv)
    c1 = LLPoolLayer(Source=EncryptLayer(Source=reader), MapCount=[5, 1], WeightsScale=64, Weights=w0, **conv)
    d4 = LLDenseLayer(Source=SquareActivation(Source=LLVectorizeLayer(Source=c1)), Bias=b1, Weights=w1, WeightsScale=64, InputFormat=EVectorFormat.dense)
    d4.PrepareNetwork()
    out = d4.GetNext()
    got = [int(x) for x in out.GetColumn(0).DecryptFullPrecision(env)]
    # exact integer model with the wrapper's rounding
    eng = ConvolutionEngine([28, 28], [5, 5], [2, 2], Upperpadding=[1, 1], MapCount=[5, 1])
    act = [int(v) for v in np.rint(img / 256.0 * 16.0)]
    g, win = eng.gather_table(), eng.weight_windows(w0, 26)
    convo = []
    for m in range(5):
        wr = [int(round(float(x) * 64)) for x in win[m]]
        b = int(round(float(w0[(m + 1) * 26 - 1]) * 16.0 * 64))
        convo += [b + sum(wr[k] * act[i] for k, i in enumerate(g[c]) if i >= 0) for c in range(len(eng.Corners))]
    sq = [v * v for v in convo]
    W = [[int(round(float(x) * 64)) for x in w1[r * 845:(r + 1) * 845]] for r in range(10)]
    s_out = (16.0 * 64) ** 2 * 64
    exp = [int(round(float(b1[r]) * s_out)) + sum(W[r][k] * sq[k] for k in range(845)) for r in range(10)]
    M = env.bigFactor
    exp = [((v % M) - M) if (v % M) * 2 > M else (v % M) for v in exp]
    assert got == exp
