"""Host-side mirror of the reference's network layers that sit on the hot path (`NeuralNetworks/*.cs`):
BaseLayer chaining, EncryptLayer, PoolLayer (conv / dense / mean-pool in batch packing), SquareActivation, LLDenseLayer.

The layers only talk to the wrapper algebra (IFactory / IMatrix / IVector), exactly like the reference; the GPU batching
happens below that surface (one scalar GEMM per PoolLayer and plaintext prime, one multiply+relinearise chain per
SquareActivation and prime).
"""
import numpy as np

from .convolution import ConvolutionEngine
from .hewrapper import EMatrixFormat, EVectorFormat


class RawData:
    """What a reader layer hands to EncryptLayer: RawMatrix semantics (`HE Wrapper/RawMatrix.cs:19-26`): the stored data
    is round(m * scale)."""

    def __init__(self, m, scale):
        self.Scale = scale
        self.Data = np.rint(np.asarray(m, dtype=np.float64) * scale)

    def Dispose(self):
        self.Data = None


class BaseLayer:
    """NeuralNetworks/BaseLayer.cs: pull-based chaining (GetNext :23-49), scale bookkeeping."""

    def __init__(self, Source=None, Factory=None):
        self.Source, self._factory = Source, Factory
        self.layerPrepared = False

    @property
    def Factory(self):
        return self._factory if self._factory is not None else self.Source.Factory

    @Factory.setter
    def Factory(self, f):
        self._factory = f

    def Prepare(self):
        self.layerPrepared = True

    def PrepareNetwork(self):
        if self.Source is not None:
            self.Source.PrepareNetwork()
        self.Prepare()

    def GetOutputScale(self):
        return self.Source.GetOutputScale()

    def Apply(self, m):
        raise NotImplementedError

    def GetNext(self):
        if not self.layerPrepared:
            self.Prepare()
        m = self.Source.GetNext()
        res = self.Apply(m)
        if res is not m:
            m.Dispose()
        return res

    def Dispose(self):
        pass


class InputLayer(BaseLayer):
    """Stand-in for BatchReader (NeuralNetworks/BatchReader.cs:59-109): rows = samples, columns = features,
    values multiplied by NormalizationFactor, stored at Scale."""

    def __init__(self, data, Scale=1.0, NormalizationFactor=1.0, Factory=None):
        super().__init__(None, Factory)
        self.data, self.Scale, self.NormalizationFactor = np.asarray(data, dtype=np.float64), Scale, NormalizationFactor

    def PrepareNetwork(self):
        self.Prepare()

    def GetNext(self):
        return RawData(self.data * self.NormalizationFactor, self.Scale)

    def GetOutputScale(self):
        return self.Scale


class EncryptLayer(BaseLayer):
    """NeuralNetworks/EncryptLayer.cs:12-19: encrypt the already-scaled integers at scale 1, then register the scale."""

    def Apply(self, m):
        res = self.Factory.GetEncryptedMatrix(m.Data, EMatrixFormat.ColumnMajor, 1)
        res.RegisterScale(m.Scale)
        return res


class SquareActivation(BaseLayer):
    """NeuralNetworks/SquareActivation.cs:10-19"""

    def Apply(self, m):
        return m.ElementWiseMultiply(m, self.Factory.AllocateComputationEnv())

    def GetOutputScale(self):
        s = self.Source.GetOutputScale()
        return s * s


class PoolLayer(BaseLayer):
    """NeuralNetworks/PoolLayer.cs: convolution / dense / mean pooling over column-major ciphertext matrices (slot = sample)."""

    def __init__(self, Source=None, Factory=None, InputShape=None, KernelShape=None, Stride=None, Padding=None, Upperpadding=None,
                 Lowerpadding=None, MapCount=None, Weights=None, Bias=None, WeightsScale=1.0):
        super().__init__(Source, Factory)
        self.InputShape, self.KernelShape, self.Stride = InputShape, KernelShape, Stride
        self.Padding, self.Upperpadding, self.Lowerpadding, self.MapCount = Padding, Upperpadding, Lowerpadding, MapCount
        self.Weights, self.Bias, self.WeightsScale = Weights, Bias, WeightsScale

    def Prepare(self):
        if self.layerPrepared:
            return
        self.engine = ConvolutionEngine(self.InputShape, self.KernelShape, self.Stride, self.Padding, self.Upperpadding, self.Lowerpadding, self.MapCount)
        self.kernelSize = int(np.prod(self.KernelShape))
        if self.Bias is None:
            self.kernelSize += 1                                   # PoolLayer.cs:57: the last weight of every map is its bias
        self.gather = self.engine.gather_table()
        if self.Weights is not None:
            win = self.engine.weight_windows(self.Weights, self.kernelSize)            # PrepareWeightsWindows (:101-111)
            self.weightWindows = [[int(x) for x in np.rint(row * self.WeightsScale)] for row in win]
        self.layerPrepared = True

    def GetOutputScale(self):
        if not self.layerPrepared:
            self.Prepare()
        return (len(self.engine.Offsets) if self.Weights is None else self.WeightsScale) * self.Source.GetOutputScale()

    def OutputDimension(self):
        if not self.layerPrepared:
            self.Prepare()
        return len(self.engine.Corners) * (1 if self.Weights is None else self.engine.maps)

    def Apply(self, m):
        if not self.layerPrepared:
            self.Prepare()
        env = self.Factory.AllocateComputationEnv()
        corners, maps = len(self.engine.Corners), self.engine.maps
        if self.Weights is None:
            # pool without convolve (PoolLayer.cs:122-147): sum of the window, scale multiplied by the window size
            ones = [[1] * self.gather.shape[1] for _ in range(corners)]
            return m.MulManySparse(self.gather, ones, None, m.Scale * len(self.engine.Offsets), env)
        if self.Bias is not None:
            bias_src = [self.Bias[mi] for mi in range(maps)]
        else:
            bias_src = [self.Weights[(mi + 1) * self.kernelSize - 1] for mi in range(maps)]
        bscale = m.Scale * self.WeightsScale                       # = Source.GetOutputScale() * WeightsScale (:169,206)
        bias_int = [int(round(float(b) * float(bscale))) for b in bias_src]
        gather, weights, bias = [], [], []
        for mi in range(maps):                                     # output k = mapIndex * Corners + cornerIndex (:184-186)
            for c in range(corners):
                gather.append(self.gather[c])
                weights.append(self.weightWindows[mi])
                bias.append(bias_int[mi])
        return m.MulManySparse(np.array(gather, dtype=np.int32), weights, bias, bscale, env)


class LLDenseLayer(BaseLayer):
    """NeuralNetworks/LLDenseLayer.cs: low-latency dense layer on a single packed ciphertext column."""

    def __init__(self, Source=None, Factory=None, Weights=None, Bias=None, WeightsScale=1.0, InputFormat=EVectorFormat.dense, ForceDenseFormat=False):
        super().__init__(Source, Factory)
        self.Weights, self.Bias, self.WeightsScale = Weights, Bias, WeightsScale
        self.InputFormat, self.ForceDenseFormat = InputFormat, ForceDenseFormat
        self.WeightsMatrix = self.BiasVector = None

    def GetOutputScale(self):
        return self.WeightsScale * self.Source.GetOutputScale()

    def OutputDimension(self):
        return len(self.Bias)

    def Prepare(self):
        if self.layerPrepared:
            return
        if self.ForceDenseFormat and self.InputFormat == EVectorFormat.sparse:
            raise Exception("forcing dense format is only available when the input is dense")
        rows = len(self.Bias)
        W = np.asarray(self.Weights, dtype=np.float64).reshape(rows, -1)
        bscale = self.Source.GetOutputScale() * self.WeightsScale
        if self.InputFormat == EVectorFormat.dense:
            self.BiasVector = self.Factory.GetPlainVector(self.Bias, EVectorFormat.dense if self.ForceDenseFormat else EVectorFormat.sparse, bscale)
            self.WeightsMatrix = self.Factory.GetPlainMatrix(W, EMatrixFormat.RowMajor, self.WeightsScale)
        else:
            self.BiasVector = self.Factory.GetPlainVector(self.Bias, EVectorFormat.dense, bscale)
            self.WeightsMatrix = self.Factory.GetPlainMatrix(W, EMatrixFormat.ColumnMajor, self.WeightsScale)
        self.layerPrepared = True

    def Apply(self, m):
        if not self.layerPrepared:
            self.Prepare()
        if m.ColumnCount > 1:
            raise Exception("Expecting only one column")
        env = self.Factory.AllocateComputationEnv()
        mul = self.WeightsMatrix.Mul(m.GetColumn(0), env, self.ForceDenseFormat)
        res = mul.Add(self.BiasVector, env)
        if res is not mul:
            mul.Dispose()
        return self.Factory.GetMatrix([res], EMatrixFormat.ColumnMajor, CopyVectors=False)

    def Dispose(self):
        for x in (self.WeightsMatrix, self.BiasVector):
            if x is not None:
                x.Dispose()
        self.WeightsMatrix = self.BiasVector = None


class FakeLayer(BaseLayer):
    """NeuralNetworksTest/FakeLayer.cs: a source whose output scale is 1"""

    def GetOutputScale(self):
        return 1.0

    def Apply(self, m):
        raise NotImplementedError
