"""The reference's application networks as layer graphs (`CryptoNets/CryptoNets.cs:14-75`, `LowLatencyCryptoNets/LoLaCryptonets.cs:
118-329`): the callers of the hot path, one builder per network plus the two evaluation loops.  A builder takes the factory (an
`EncryptedSealBfvFactory` - or a `RawFactory`, the reference's `Encrypt = false`), the reader layer and the model arrays (the
reference compiles them in as `Weights.cs` / `SmallModel.cs`; here they are passed in - `tests/golden/*.npz` holds them as data).
The factory parameters each network was designed for are in `FACTORY_PARAMETERS`.
"""
import numpy as np

from .hewrapper import EVectorFormat
from .layers import (BatchReader, EncryptLayer, LLConvReader, LLDenseLayer, LLDuplicateLayer, LLInterleavedDenseLayer, LLInterleaveLayer,
                     LLPackedDenseLayer, LLPoolLayer, LLPreConvLayer, LLSingleLineReader, LLVectorizeLayer, PoolLayer, SquareActivation,
                     TimingLayer)

# plaintext primes, N, decomposition bit counts, coefficient primes taken: CryptoNets.cs:17; LoLaCryptonets.cs:123,208,285,338;
# LolaCifarCryptoNet.cs:35
FACTORY_PARAMETERS = {
    "CryptoNets": dict(primes=(549764251649, 549764284417), n=8192),
    "LoLa": dict(primes=(557057, 638977, 737281, 786433), n=8192),
    "LoLaDense": dict(primes=(34359771137, 34360754177), n=16384, DecompositionBitCount=60, GaloisDecompositionBitCount=60, SmallModulusCount=7),
    "LoLaSmall": dict(primes=(2277377, 2424833), n=8192, DecompositionBitCount=40, GaloisDecompositionBitCount=40, SmallModulusCount=3),
    "LoLaLarge": dict(primes=(2148728833, 2148794369, 2149810177), n=16384, DecompositionBitCount=60, GaloisDecompositionBitCount=60, SmallModulusCount=7),
    "LoLaCifar": dict(primes=(957181001729, 957181034497), n=16384, DecompositionBitCount=60, GaloisDecompositionBitCount=60, SmallModulusCount=8),
}
CIFAR_CONV = dict(InputShape=[3, 32, 32], KernelShape=[3, 8, 8], Upperpadding=[0, 1, 1], Lowerpadding=[0, 1, 1], Stride=[1000, 2, 2])
LARGE_CONV = dict(InputShape=[1, 28, 28], KernelShape=[1, 8, 8], Upperpadding=[0, 1, 1], Lowerpadding=[0, 1, 1], Stride=[1000, 2, 2])
MNIST_CONV = dict(InputShape=[28, 28], KernelShape=[5, 5], Upperpadding=[1, 1], Stride=[2, 2])


def Transpose(weights, inputShapeSize, outputMaps):
    """CryptoNets.cs:112-123: [input][map] -> [map][input]"""
    w = np.asarray(weights, dtype=np.float64)
    return w[:inputShapeSize * outputMaps].reshape(inputShapeSize, outputMaps).T.reshape(-1).copy()


def mnist_reader(FileName, batchSize, Factory=None):
    return BatchReader(FileName=FileName, SparseFormat=True, MaxSlots=batchSize, NormalizationFactor=1.0 / 256.0, Scale=16.0, Factory=Factory)


def CryptoNets(Factory, reader, w, weightscale=32, timing=True):
    """CryptoNets.cs:19-75: conv 5x5 stride 2 (5 maps) -> square -> dense 845->100 -> square -> dense 100->10, one sample per slot."""
    enc = EncryptLayer(Source=reader, Factory=Factory)
    src = TimingLayer(Source=enc, StartCounters=["Batch-Time"]) if timing else enc
    c1 = PoolLayer(Source=src, MapCount=[5, 1], WeightsScale=weightscale, Weights=w["Weights_0"], **MNIST_CONV)
    a2 = SquareActivation(Source=c1)
    d3 = PoolLayer(Source=a2, InputShape=[845], KernelShape=[845], Stride=[1000], MapCount=[100], Weights=Transpose(w["Weights_1"], 845, 100),
                   Bias=w["Biases_2"], WeightsScale=weightscale * weightscale)
    a4 = SquareActivation(Source=d3)
    d5 = PoolLayer(Source=a4, InputShape=[100], KernelShape=[100], Stride=[1000], MapCount=[10], Weights=w["Weights_3"], Bias=w["Biases_3"],
                   WeightsScale=weightscale)
    return TimingLayer(Source=d5, StopCounters=["Batch-Time"]) if timing else d5


def LoLa(Factory, reader, w, weightscale=32):
    """LoLaCryptonets.cs:203-278: the reader does the im2col; 8-fold packed dense layer, interleave, interleaved dense layer."""
    enc = EncryptLayer(Source=reader, Factory=Factory)
    c1 = LLPoolLayer(Source=enc, MapCount=[5, 1], WeightsScale=weightscale, Weights=w["Weights_0"], **MNIST_CONV)
    v2 = LLVectorizeLayer(Source=c1)
    a3 = SquareActivation(Source=v2)
    d4 = LLDuplicateLayer(Source=a3, Count=8)
    d5 = LLPackedDenseLayer(Source=d4, Weights=Transpose(w["Weights_1"], 845, 100), Bias=w["Biases_2"], WeightsScale=weightscale * weightscale,
                            PackingCount=d4.Count, PackingShift=1024)
    sel = [1023 + i * 1024 for i in range(int(d4.Count))]
    i6 = LLInterleaveLayer(Source=d5, Shift=-1, SelectedIndices=sel)
    a7 = SquareActivation(Source=i6)
    return LLInterleavedDenseLayer(Source=a7, Weights=w["Weights_3"], Bias=w["Biases_3"], WeightsScale=weightscale, Shift=-1, SelectedIndices=sel)


def LoLaDense(Factory, reader, w, weightscale=32):
    """LoLaCryptonets.cs:118-199: ONE packed ciphertext in; the im2col is done homomorphically (LLPreConvLayer), 16-fold packing."""
    enc = EncryptLayer(Source=reader, Factory=Factory)
    pre = LLPreConvLayer(Source=enc, UseAxisForBlocks=[True, True], **MNIST_CONV)
    c2 = LLPoolLayer(Source=pre, MapCount=[5, 1], WeightsScale=weightscale, Weights=w["Weights_0"], HotIndices=pre.HotIndices, **MNIST_CONV)
    v3 = LLVectorizeLayer(Source=c2)
    a4 = SquareActivation(Source=v3)
    d5 = LLDuplicateLayer(Source=a4, Count=16)
    d6 = LLPackedDenseLayer(Source=d5, Weights=pre.RearrangeWeights(Transpose(w["Weights_1"], 845, 100)), Bias=w["Biases_2"],
                            WeightsScale=weightscale * weightscale, PackingCount=d5.Count, PackingShift=1024)
    a7 = SquareActivation(Source=d6)
    sel = [1023 + i * 1024 for i in range(int(d5.Count))]
    i8 = LLInterleaveLayer(Source=a7, Shift=-1, SelectedIndices=sel)
    return LLInterleavedDenseLayer(Source=i8, Weights=w["Weights_3"], Bias=w["Biases_3"], WeightsScale=weightscale, Shift=-1, SelectedIndices=sel)


def SmallLoLa(Factory, reader, w, weightscale=64):
    """LoLaCryptonets.cs:280-329 (model: SmallModel.cs): conv -> vectorize -> square -> dense 845->10 on the packed vector."""
    enc = EncryptLayer(Source=reader, Factory=Factory)
    c1 = LLPoolLayer(Source=enc, MapCount=[5, 1], WeightsScale=weightscale, Weights=w["Weights_0"], **MNIST_CONV)
    v2 = LLVectorizeLayer(Source=c1)
    a3 = SquareActivation(Source=v2)
    return LLDenseLayer(Source=a3, Bias=w["Biases_1"], Weights=w["Weights_1"], WeightsScale=weightscale, InputFormat=EVectorFormat.dense)


def LargeLoLa(Factory, reader, Weights, Biases):
    """LoLaCryptonets.cs:332-409.  `Weights` / `Biases`: the arrays of `WeightsReader("MnistLargeWeight.csv", "MnistLargeBias.csv")`:
    conv1 83 x (8x8) + 83 (the file stores the conv1 weights times 256), conv2 163 x (83x6x6) + 163 unrolled into a 2608 x 11952 dense
    layer, dense 10 x 2608 + 10."""
    from .convolution import ConvolutionEngine
    enc = EncryptLayer(Source=reader, Factory=Factory)
    c1 = LLPoolLayer(Source=enc, MapCount=[83, 1, 1], WeightsScale=4096, Weights=np.asarray(Weights[0], dtype=np.float64) / 256, Bias=Biases[0],
                     **LARGE_CONV)
    v2 = LLVectorizeLayer(Source=c1)
    a3 = SquareActivation(Source=v2)
    eng = ConvolutionEngine([83, 12, 12], [83, 6, 6], [83, 2, 2], Padding=[False, False, False], MapCount=[163, 1, 1])
    d4 = LLDenseLayer(Source=a3, WeightsScale=64, Weights=eng.GetDenseWeights(Weights[1]), Bias=eng.GetDenseBias(Biases[1]),
                      InputFormat=EVectorFormat.dense, ForceDenseFormat=True)
    a5 = SquareActivation(Source=d4)
    return LLDenseLayer(Source=a5, Weights=Weights[2], Bias=Biases[2], WeightsScale=512, InputFormat=EVectorFormat.dense)


def cifar_reader(FileName=None, Factory=None):
    """LolaCifarCryptoNet.cs:43-55: dense TSV records (label in column 0, 3072 pixel values), im2col for the 8x8 stride-2 convolution"""
    return LLConvReader(FileName=FileName, SparseFormat=False, NormalizationFactor=1.0 / 256.0, Scale=8.0, Factory=Factory, **CIFAR_CONV)


def LoLaCifar(Factory, reader, Weights, Biases, timing=True):
    """LolaCifarCryptoNet.cs:58-131.  `Weights` / `Biases`: the three arrays of `WeightsReader("CifarWeight.csv", "CifarBias.csv")`:
    conv1 83 x (3x8x8) + 83, conv2 112 x (83x10x10) + 112 (unrolled into a 5488 x 16268 dense layer), dense 10 x 5488 + 10."""
    from .convolution import ConvolutionEngine
    enc = EncryptLayer(Source=reader, Factory=Factory)
    src = TimingLayer(Source=enc, StartCounters=["Inference-Time"]) if timing else enc
    c1 = LLPoolLayer(Source=src, MapCount=[83, 1, 1], WeightsScale=256.0, Weights=Weights[0], Bias=Biases[0], **CIFAR_CONV)
    v2 = LLVectorizeLayer(Source=c1)
    a3 = SquareActivation(Source=v2)
    eng = ConvolutionEngine([83, 14, 14], [83, 10, 10], [83, 2, 2], Upperpadding=[0, 4, 4], Lowerpadding=[0, 4, 4], MapCount=[112, 1, 1])
    d4 = LLDenseLayer(Source=a3, WeightsScale=512.0, Weights=eng.GetDenseWeights(Weights[1]), Bias=eng.GetDenseBias(Biases[1]),
                      InputFormat=EVectorFormat.dense, ForceDenseFormat=True)
    a5 = SquareActivation(Source=d4)
    d6 = LLDenseLayer(Source=a5, Weights=Weights[2], Bias=Biases[2], WeightsScale=512.0, InputFormat=EVectorFormat.dense)
    return TimingLayer(Source=d6, StopCounters=["Inference-Time"]) if timing else d6


def lola_cifar_dense_model(reader, conv_layer, Weights, Biases, image, p):
    """Exact integer model mod p of LoLa-CIFAR through its big dense layer (conv 83 x (3x8x8) -> square -> dense 5488 x 16268): the 5488 values
    a decryption of that layer's output must show (tests/test_lola_cifar.py, bench.py --workload cifar).  `image`: 3072 pixel values 0..255."""
    from .convolution import ConvolutionEngine

    def mulmod(a, b, pp):
        b = np.asarray(b, dtype=np.uint64)
        hi = (a * (b >> np.uint64(20))) % pp
        return (hi * np.uint64(1 << 20) + a * (b & np.uint64(0xFFFFF))) % pp
    w0, w1 = Weights[0], Weights[1]
    b0, b1 = Biases[0], Biases[1]
    eng = ConvolutionEngine([83, 14, 14], [83, 10, 10], [83, 2, 2], Upperpadding=[0, 4, 4], Lowerpadding=[0, 4, 4], MapCount=[112, 1, 1])
    W1 = eng.GetDenseWeights(w1).reshape(5488, 16268)
    x = np.rint(np.asarray(image, dtype=np.float64) / 256.0 * 8.0).astype(np.int64)
    g = reader.engine.gather_table()                                # [196, 192]
    patches = np.where(g >= 0, x[np.maximum(g, 0)], 0)
    W0i = np.rint(conv_layer.engine.weight_windows(w0, 192) * 256).astype(np.int64)
    B0i = np.rint(np.asarray(b0) * 8 * 256).astype(np.int64)
    act1 = (patches @ W0i.T + B0i).T.reshape(-1)                    # map-major stacking: 83 x 196
    s1 = (8 * 256) ** 2
    W1i = np.rint(W1 * 512).astype(np.int64)
    B1i = [int(round(float(b) * s1 * 512)) for b in eng.GetDenseBias(b1)]
    pp = np.uint64(p)
    a1 = np.mod(act1, int(p)).astype(np.uint64)
    a1 = mulmod(a1, a1, pp)
    acc = np.zeros(5488, dtype=np.uint64)
    W1p = np.mod(W1i, int(p)).astype(np.uint64)
    for c0 in range(0, 16268, 512):
        acc = (acc + (mulmod(W1p[:, c0:c0 + 512], a1[None, c0:c0 + 512], pp) % pp).sum(axis=1) % pp) % pp
    return (acc + np.array([b % int(p) for b in B1i], dtype=np.uint64)) % pp


def lola_reader(name, FileName=None, Factory=None):
    """the input layer each LoLa variant reads MNIST with (LoLaCryptonets.cs:131-137,212-223,294-305)"""
    if name == "LoLaLarge":                                      # :346-357: pixels are NOT normalised here
        return LLConvReader(FileName=FileName, SparseFormat=True, NormalizationFactor=1.0, Scale=16.0, Factory=Factory, **LARGE_CONV)
    if name == "LoLaDense":
        return LLSingleLineReader(FileName=FileName, SparseFormat=True, NormalizationFactor=1.0 / 256.0, Scale=16.0, Factory=Factory)
    return LLConvReader(FileName=FileName, SparseFormat=True, NormalizationFactor=1.0 / 256.0, Scale=16.0, Factory=Factory, **MNIST_CONV)


LOLA_NETWORKS = {"LoLa": LoLa, "LoLaDense": LoLaDense, "LoLaSmall": SmallLoLa}


def _chain(network):
    p = network
    while p is not None:
        yield p
        p = p.Source


def evaluate_batches(network, Factory, reader, numberOfRecords, report=print):
    """CryptoNets.cs:80-109: batches until `numberOfRecords` samples are scored; prediction = arg max of the decrypted row.
    Returns (errors, count)."""
    network.PrepareNetwork()
    count = errs = 0
    while count < numberOfRecords:
        try:
            m = network.GetNext()
        except Exception as e:                                   # BatchReader at end of file
            if "end of file" in str(e):
                break
            raise
        env = Factory.AllocateComputationEnv()
        try:
            decrypted = np.asarray(m.Decrypt(env))
            rows = min(decrypted.shape[0], len(reader.Labels))
            pred = np.argmax(decrypted[:rows], axis=1)               # first maximum, like the strict '>' scan (:94-97)
            errs += int(np.sum(pred != np.asarray(reader.Labels[:rows])))
            count += rows
            if report is not None:
                report("errs %d/%d accuracy %.3f%%" % (errs, count, 100 - 100.0 * errs / count))
                report("Batch size %d %s" % (rows, TimingLayer.GetStats()))
        finally:
            Factory.FreeComputationEnv(env)
            m.Dispose()
    return errs, count


def evaluate_single_recorded(network, Factory, records, report=print):
    """The loop of `evaluate_single` with the evaluation RECORDED: the first record runs layer by layer (rehearsal), the layers after the
    EncryptLayer are then recorded once as one HIP graph per plaintext prime (`hewrapper.CapturedEvaluation`) and every further record
    is encrypted, copied into the recorded input and evaluated with one launch per prime.  "Prediction-Time" covers the same window
    as the reference's brackets (after encryption .. before decryption).  Encrypted GPU factory only.  Returns (errors, count)."""
    import time
    from .hewrapper import CapturedEvaluation
    layers = list(_chain(network))[::-1]                          # reader first
    reader = layers[0]
    k = next(i for i, p in enumerate(layers) if isinstance(p, EncryptLayer))
    enc, tail = layers[k], layers[k + 1:]
    if any(isinstance(p, TimingLayer) for p in layers):
        raise Exception("a recorded evaluation cannot contain TimingLayers (they synchronise)")
    for p in layers:
        p.Factory = Factory
    network.PrepareNetwork()
    env = Factory.AllocateComputationEnv()

    def sync():
        for e in env.Environments:
            e.ctx.sync()

    def run_tail(x, keep):
        for L in tail:
            y = L.Apply(x)
            if y is not x and x is not keep:
                x.Dispose()
            x = y
        return x
    errs = count = 0
    cap = first = None
    try:
        for i in range(records):
            m = enc.GetNext()
            if m is None:
                break
            sync()
            t0 = time.perf_counter()
            if cap is None:
                first = m
                out = run_tail(first, first)
                if i == 0 and records > 1:                        # rehearsed once: record it for the remaining records
                    sync(); dt = time.perf_counter() - t0
                    score = np.asarray(out.Decrypt(env))[:, 0]
                    out.Dispose()
                    cap = CapturedEvaluation(env, lambda x: run_tail(x, first), [first])
                else:
                    sync(); dt = time.perf_counter() - t0
                    score = np.asarray(out.Decrypt(env))[:, 0]
                    out.Dispose()
            else:
                out = cap.run(m)
                sync(); dt = time.perf_counter() - t0
                score = np.asarray(out.Decrypt(env))[:, 0]
                m.Dispose()
            pred, label = int(np.argmax(score[:10])), int(reader.Labels[0])
            errs += int(pred != label)
            count += 1
            if report is not None:
                report("errs %d/%d accuracy %.3f%% Prediction-Time %.2f ms%s prediction %d label %d"
                       % (errs, count, 100 - 100.0 * errs / count, 1e3 * dt, " (recorded)" if cap is not None and i > 0 else "", pred, label))
    finally:
        if cap is not None:
            cap.result.Dispose()
            cap.Dispose()
        if first is not None:
            first.Dispose()
        Factory.FreeComputationEnv(env)
    return errs, count


def evaluate_single(network, Factory, records, verbose=False, report=print):
    """LoLaCryptonets.cs:64-115: "Prediction-Time" brackets everything after the EncryptLayer; one record per GetNext.
    Returns (errors, count)."""
    layers = list(_chain(network))
    reader = layers[-1]
    if not any(isinstance(p, TimingLayer) for p in layers):      # the CIFAR graph brings its own "Inference-Time" brackets
        first = next(p for p in layers if isinstance(p.Source, EncryptLayer))
        start = TimingLayer(Source=first.Source, StartCounters=["Prediction-Time"])
        first.Source = start
        network = TimingLayer(Source=network, StopCounters=["Prediction-Time"])
    for p in _chain(network):
        p.Factory = Factory
        p.Verbose = verbose
    network.PrepareNetwork()
    errs = count = 0
    for i in range(records):
        m = network.GetNext()
        if m is None:
            break
        env = Factory.AllocateComputationEnv()
        try:
            dec = np.asarray(m.Decrypt(env))[:, 0]
            pred = int(np.argmax(dec[:10]))
            label = int(reader.Labels[0])
            errs += int(pred != label)
            count += 1
            if report is not None:
                report("errs %d/%d accuracy %.3f%% %s prediction %d label %d" % (errs, count, 100 - 100.0 * errs / count, TimingLayer.GetStats(), pred, label))
        finally:
            Factory.FreeComputationEnv(env)
            m.Dispose()
    return errs, count
