"""Layer-level known-answer tests ported from `NeuralNetworksTest/LayersTest.cs` (EvenPool :54-82,
PoolLayerAsSparseToDense :155-185) plus a small end-to-end CNN, on the oracle (CPU) and on libcnhip (GPU)."""
import numpy as np
import pytest

from oracle_backend import make_factory
from cryptonets_amd.hewrapper import EMatrixFormat, EVectorFormat
from cryptonets_amd.raw import RawFactory
from cryptonets_amd.layers import EncryptLayer, FakeLayer, InputLayer, LLDenseLayer, PoolLayer, SquareActivation

# "raw": the reference's own set-up for these tests (LayersTest.cs:56,87,157 run on Defaults.RawFactory)
BACKENDS = [pytest.param("raw"), pytest.param("cpu"), pytest.param("gpu", marks=pytest.mark.gpu)]
_f = {}


def factory(backend, **kw):
    if backend == "raw":
        return RawFactory(kw.get("n", 8192))
    key = (backend, tuple(sorted(kw.items())))
    if key not in _f:
        _f[key] = make_factory(backend, **kw)
    return _f[key]


@pytest.mark.parametrize("backend", BACKENDS)
def test_EvenPool(backend):
    Factory = factory(backend)
    layer = PoolLayer(Factory=Factory, InputShape=[3, 4, 4], KernelShape=[1, 2, 2], Stride=[1, 2, 2])
    layer.Prepare()
    data = np.arange(48, dtype=float).reshape(1, 48)
    m = Factory.GetEncryptedMatrix(data, EMatrixFormat.ColumnMajor, 1)
    t = layer.Apply(m)
    res = t.Decrypt(Factory.AllocateComputationEnv())
    assert res.shape == (1, 12)
    assert list(res[0]) == [2.5, 4.5, 10.5, 12.5, 18.5, 20.5, 26.5, 28.5, 34.5, 36.5, 42.5, 44.5]
    t.Dispose()
    m.Dispose()


@pytest.mark.parametrize("backend", BACKENDS)
def test_PoolLayerAsSparseToDense(backend):
    Factory = factory(backend)
    vec = Factory.GetEncryptedVector(np.array([1.0, 2.0, 3.0]), EVectorFormat.sparse, 1)
    m = Factory.GetMatrix([vec], EMatrixFormat.ColumnMajor)
    layer = LLDenseLayer(Factory=Factory, Weights=[1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1], Bias=[0] * 6, WeightsScale=1,
                         InputFormat=EVectorFormat.sparse, Source=FakeLayer())
    layer.Prepare()
    res = layer.Apply(m)
    dec = res.Decrypt(Factory.AllocateComputationEnv())
    assert dec.shape == (6, 1)
    assert [dec[i, 0] for i in range(6)] == [1 + (i % 3) for i in range(6)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("force_dense", [False, True])
def test_LLDenseLayer_dense_input(backend, force_dense):
    """LoLa dense layer on a packed ciphertext: per-row dense MultiplyPlain + SumAllSlots (rotate-and-add), sparse
    output or mask-accumulated dense output (EncryptedSealBfvMatrix.cs:79-120)."""
    Factory = factory(backend)
    rng = np.random.default_rng(3)
    x = rng.integers(-5, 6, size=40).astype(float)
    W = rng.integers(-4, 5, size=(6, 40)).astype(float)
    b = rng.integers(-9, 10, size=6).astype(float)
    col = Factory.GetEncryptedVector(x, EVectorFormat.dense, 2.0)
    m = Factory.GetMatrix([col], EMatrixFormat.ColumnMajor, CopyVectors=False)

    class Src(FakeLayer):
        def GetOutputScale(self):
            return 2.0
    layer = LLDenseLayer(Factory=Factory, Weights=W.reshape(-1), Bias=b, WeightsScale=3.0, ForceDenseFormat=force_dense, Source=Src())
    res = layer.Apply(m)
    dec = res.Decrypt(Factory.AllocateComputationEnv())
    assert np.array_equal(dec[:, 0], W @ x + b)


@pytest.mark.parametrize("backend", BACKENDS)
def test_small_cnn_end_to_end(backend):
    """conv (with padding, several maps) -> square -> dense+bias -> square -> dense+bias through the layer classes,
    checked exactly against integer arithmetic on the scaled inputs/weights."""
    Factory = factory(backend, primes=(40961, 65537, 114689), n=4096, galois=False)
    rng = np.random.default_rng(11)
    samples = 50
    img = rng.integers(0, 4, size=(samples, 36)).astype(float)                  # 6x6 images
    w0 = rng.integers(-2, 3, size=2 * 10).astype(float)                         # 2 maps x (3x3 + bias)
    conv = dict(InputShape=[6, 6], KernelShape=[3, 3], Stride=[2, 2], Upperpadding=[1, 1], MapCount=[2, 1])
    src = InputLayer(img, Scale=1.0, Factory=Factory)
    enc = EncryptLayer(Source=src)
    l1 = PoolLayer(Source=enc, Weights=w0, WeightsScale=1.0, **conv)
    a1 = SquareActivation(Source=l1)
    n1 = l1.OutputDimension()
    w1 = rng.integers(-1, 2, size=4 * n1).astype(float)
    b1 = rng.integers(-3, 4, size=4).astype(float)
    l2 = PoolLayer(Source=a1, InputShape=[n1], KernelShape=[n1], Stride=[1000], MapCount=[4], Weights=w1, Bias=b1, WeightsScale=1.0)
    a2 = SquareActivation(Source=l2)
    w2 = rng.integers(-1, 2, size=3 * 4).astype(float)
    b2 = rng.integers(-3, 4, size=3).astype(float)
    l3 = PoolLayer(Source=a2, InputShape=[4], KernelShape=[4], Stride=[1000], MapCount=[3], Weights=w2, Bias=b2, WeightsScale=1.0)
    l3.PrepareNetwork()
    out = l3.GetNext()
    dec = out.Decrypt(Factory.AllocateComputationEnv())
    # integer model
    g = l1.engine.gather_table()
    x = img.astype(np.int64)
    y1 = np.zeros((samples, n1), dtype=np.int64)
    for mi in range(2):
        for c in range(len(l1.engine.Corners)):
            acc = np.full(samples, int(w0[(mi + 1) * 10 - 1]), dtype=np.int64)
            for k, idx in enumerate(g[c]):
                if idx >= 0:
                    acc += int(l1.weightWindows[mi][k]) * x[:, idx]
            y1[:, mi * len(l1.engine.Corners) + c] = acc
    y1 = y1 ** 2
    y2 = (y1 @ w1.reshape(4, n1).T.astype(np.int64) + b1.astype(np.int64)) ** 2
    y3 = y2 @ w2.reshape(3, 4).T.astype(np.int64) + b2.astype(np.int64)
    assert dec.shape == (samples, 3)
    assert np.array_equal(dec, y3.astype(float))
    out.Dispose()


def test_batch_reader_tsv_formats(tmp_path):
    """NeuralNetworks/BatchReader.cs:59-109: sparse `label dim idx:val ...` and dense TSV lines, MaxSlots lines per batch,
    NormalizationFactor applied to the values, RawMatrix rounding at Scale."""
    from cryptonets_amd.layers import BatchReader
    sparse = tmp_path / "mnist.tsv"
    sparse.write_text("7\t6\t1:255\t4:128\n2\t6\t0:64\n9\t6\n")
    r = BatchReader(str(sparse), MaxSlots=2, NormalizationFactor=1.0 / 256.0, Scale=16.0)
    b = r.GetNext()
    assert list(r.Labels) == [7, 2] and b.Scale == 16.0 and r.OutputDimension() == 6
    assert np.array_equal(b.Data, np.array([[0, 16, 0, 0, 8, 0], [4, 0, 0, 0, 0, 0]], dtype=float))      # round(v/256*16)
    b = r.GetNext()                                                # the remaining line
    assert list(r.Labels) == [9] and not b.Data.any()
    with pytest.raises(Exception):
        r.GetNext()
    dense = tmp_path / "dense.tsv"
    dense.write_text("1.5\t3\t-2\n0.25\t5\t4\n")
    r = BatchReader(str(dense), MaxSlots=8, SparseFormat=False, LabelColumn=1, Scale=4.0)
    b = r.GetNext()
    assert list(r.Labels) == [3, 5] and np.array_equal(b.Data, np.array([[6, -8], [1, 16]], dtype=float))
    r = BatchReader(str(dense), MaxSlots=1, SparseFormat=False, LabelColumn=7, Scale=1.0)          # no label column
    b = r.GetNext()
    assert r.Labels[0] == 2 ** 31 - 1 and b.Data.shape == (1, 3)
    r.Dispose()


@pytest.mark.parametrize("backend", BACKENDS)
def test_PreConvLayer(backend):
    """NeuralNetworksTest/LayersTest.cs:84-152, on ENCRYPTED data (the reference runs it on its Raw factory): LLPreConvLayer turns one
    packed 28x28 image into the 25 aligned offset vectors of a 5x5 / stride 2 / upper-pad 1 convolution with masks + Permute."""
    from cryptonets_amd.layers import LLPreConvLayer
    Factory = factory(backend)
    layer = LLPreConvLayer(Factory=Factory, InputShape=[28, 28], KernelShape=[5, 5], Upperpadding=[1, 1], Stride=[2, 2])
    layer.Prepare()
    inp = np.arange(1, 28 * 28 + 1, dtype=float)
    v = Factory.GetEncryptedVector(inp, EVectorFormat.dense, 1)
    res = layer.Apply(Factory.GetMatrix([v], EMatrixFormat.ColumnMajor))
    dec = np.asarray(res.Decrypt(Factory.AllocateComputationEnv()))
    assert dec.shape == (196, 25) and layer.OutputDimension() == 196
    values = set()
    for j in range(196):
        val = int(dec[j, 0])
        if val != 0:
            assert val not in values
            values.add(val)
            x, y = (val - 1) // 28, (val - 1) % 28
            assert x % 2 == 0 and y % 2 == 0 and 0 <= x < 26 and 0 <= y < 26
    assert len(values) == 13 * 13
    assert int(layer.HotIndices.sum()) == 13 * 13 and all(layer.HotIndices[j] == (dec[j, 0] != 0) for j in range(196))
    for i in range(1, 25):
        dx, dy = i // 5, i % 5
        delta = dy * 28 + dx
        for j in range(196):
            val, val0 = dec[j, i], dec[j, 0]
            if val0 == 0:
                assert val == 0
            else:
                y, x = int(val0 - 1) // 28, int(val0 - 1) % 28
                assert val == (0 if (x + dx >= 28 or y + dy >= 28) else val0 + delta)
    # RearrangeWeights: per-corner weights land on the slot of their corner
    w = np.arange(1, 2 * 169 + 1, dtype=float)
    r = layer.RearrangeWeights(w)
    assert r.shape == (2 * 196,) and sorted(r[r != 0]) == list(w) and r[layer.CornersMap[5]] == w[5] and r[196 + layer.CornersMap[7]] == w[169 + 7]


def test_single_line_reader_timing_layer_weights_reader(tmp_path):
    from cryptonets_amd.layers import LLSingleLineReader, TimingLayer, WeightsReader
    f = tmp_path / "one.tsv"
    f.write_text("4\t5\t0:128\t3:64\n1\t5\t4:255\n")
    r = LLSingleLineReader(str(f), NormalizationFactor=1.0 / 256.0, Scale=8.0)
    m = r.GetNext()
    assert list(r.Labels) == [4] and m.Data.shape == (5, 1) and list(m.Data[:, 0]) == [4, 0, 0, 2, 0] and r.OutputDimension() == 5
    m = r.GetNext()
    assert list(r.Labels) == [1] and m.Data[4, 0] == 8.0
    assert r.GetNext() is None
    TimingLayer.Reset()
    src = FakeLayer()
    start, stop = TimingLayer(Source=src, StartCounters=["t"]), TimingLayer(Source=src, StopCounters=["t", "never-started"])
    tok = object()
    assert start.Apply(tok) is tok and stop.Apply(tok) is tok
    assert TimingLayer.N == {"t": 1} and TimingLayer.GetStats().startswith("t ")
    (tmp_path / "w.csv").write_text("1,2.5,-3\n4\n")
    (tmp_path / "b.csv").write_text("0.5,0.25\n")
    wr = WeightsReader(str(tmp_path / "w.csv"), str(tmp_path / "b.csv"))
    assert [list(x) for x in wr.Weights] == [[1, 2.5, -3], [4]] and list(wr.Biases[0]) == [0.5, 0.25]
