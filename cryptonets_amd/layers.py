"""Host-side mirror of the reference's network layers that sit on the hot path (`NeuralNetworks/*.cs`):
BaseLayer chaining, EncryptLayer, PoolLayer (conv / dense / mean-pool in batch packing), SquareActivation, LLDenseLayer.

The layers only talk to the wrapper algebra (IFactory / IMatrix / IVector), exactly like the reference; the GPU batching
happens below that surface (one scalar GEMM per PoolLayer and plaintext prime, one multiply+relinearise chain per
SquareActivation and prime).
"""
import numpy as np

from .convolution import ConvolutionEngine
from .hewrapper import EMatrixFormat, EVectorFormat
from .raw import Defaults, RawMatrix
from .cryptotracker import CryptoTracker, OperationsCount
from . import tracing


class RawData(RawMatrix):
    """What a reader layer hands to EncryptLayer: a RawMatrix of `Defaults.RawFactory` (`BatchReader.cs:51,126`,
    `LLConvReader.cs:73`): the stored data is round(m * scale) (`HE Wrapper/RawMatrix.cs:19-26`)."""

    def __init__(self, m, scale):
        super().__init__(np.asarray(m, dtype=np.float64), scale, EMatrixFormat.ColumnMajor, Defaults.RawFactory.BlockSize)


def _sync_factory(layer):
    """device work is asynchronous: wait for the streams of the layer's factory before reading a clock (nothing to wait for on a
    RawFactory or on a chain without a factory)"""
    try:
        env = layer.Factory.AllocateComputationEnv()
    except AttributeError:
        return
    for e in getattr(env, "Environments", ()):
        e.ctx.sync()


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

    def OutputDimension(self):
        return self.Source.OutputDimension()

    def Apply(self, m):
        raise NotImplementedError

    Verbose = False

    def GetNext(self):
        if not self.layerPrepared:
            self.Prepare()
        m = self.Source.GetNext()
        if m is None:                                            # a single-record reader at end of file
            return None
        if not self.Verbose:
            with tracing.range(type(self).__name__):             # roctx range per layer (no-op unless CN_ROCTX=1)
                res = self.Apply(m)
        else:                                                    # BaseLayer.cs:30-43: per-layer wall time and width
            import time
            OperationsCount.Reset(self.Factory)
            start = time.perf_counter()
            res = self.Apply(m)
            _sync_factory(self)
            print("Layer %s computed in %.6f seconds layer width (%d,%d)" % (type(self).__name__, time.perf_counter() - start, m.RowCount, m.ColumnCount))
            CryptoTracker.TestBudget(res.GetColumn(0), self.Factory)                 # BaseLayer.cs:37 (a no-op unless budget tests are on)
            if OperationsCount._contexts(self.Factory):
                OperationsCount.Print(self.Factory)                                  # BaseLayer.cs:39
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


class BatchReader(BaseLayer):
    """NeuralNetworks/BatchReader.cs: the reference's TSV input layer.  Sparse format (default, `DataPreprocess/GetMNIST.cs`
    output): `label <TAB> dim <TAB> index:value <TAB> ...`; dense format: one value per column, the label in `LabelColumn`
    (no label when LabelColumn >= the column count).  GetNext reads up to MaxSlots lines = one batch (one sample per slot),
    values times NormalizationFactor, stored at Scale; `Labels` holds the batch's labels (:59-109)."""

    def __init__(self, FileName=None, MaxSlots=-1, NormalizationFactor=1.0, Scale=1.0, SparseFormat=True, LabelColumn=0, Factory=None):
        super().__init__(None, Factory)
        self.MaxSlots, self.NormalizationFactor, self.Scale = MaxSlots, NormalizationFactor, Scale
        self.SparseFormat, self.LabelColumn = SparseFormat, LabelColumn
        self.Labels, self._sr, self._dim = None, None, -1
        if FileName is not None:
            self.FileName = FileName

    @property
    def FileName(self):
        return self._file_name

    @FileName.setter
    def FileName(self, value):
        self._file_name = value
        if self._sr is not None:
            self._sr.close()
        self._sr = open(value, "r")
        self._dim = -1

    def PrepareNetwork(self):
        self.Prepare()

    def Apply(self, m):
        return self.GetNext()

    def GetNext(self):
        labels, rows = [], []
        while len(labels) < self.MaxSlots:
            line = self._sr.readline()
            if line == "":
                break
            f = line.rstrip("\r\n").split("\t")
            if self.SparseFormat:
                labels.append(int(f[0]))
                self._dim = int(f[1])
                row = np.zeros(self._dim)
                for item in f[2:]:
                    c, v = item.split(":")
                    row[int(c)] = float(v) * self.NormalizationFactor
            else:
                dim = len(f)
                if self.LabelColumn >= dim:
                    labels.append(2 ** 31 - 1)
                    row = np.array([float(x) for x in f])
                else:
                    labels.append(int(f[self.LabelColumn]))
                    row = np.array([float(x) for k, x in enumerate(f) if k != self.LabelColumn])
                self._dim = len(row)
                row = row * self.NormalizationFactor
            rows.append(row)
        self.Labels = np.array(labels, dtype=np.int64)
        if not rows:
            raise Exception("BatchReader: end of file")
        return RawData(np.stack(rows), self.Scale)

    def GetOutputScale(self):
        return self.Scale

    def OutputDimension(self):
        return self._dim

    def Dispose(self):
        if self._sr is not None:
            self._sr.close()
            self._sr = None


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
            return m.MulManySparse(self.gather, ones, None, m.Scale * len(self.engine.Offsets), env, cache=self.__dict__.setdefault("_gemm_plans", {}))
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
        return m.MulManySparse(np.array(gather, dtype=np.int32), weights, bias, bscale, env, cache=self.__dict__.setdefault("_gemm_plans", {}))


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


# ------------------------------------------------------------------------------------------------ LoLa (low latency) layers
# Pure callers of the IVector / IMatrix surface (component 15): one image packed into a few ciphertexts,
# dense layers as plaintext-row x ciphertext dot products with rotate-and-sum.

def _parse_record(line, SparseFormat, LabelColumn, NormalizationFactor, normalize_dense):
    """One TSV record of the reference's readers -> (label, features).  Sparse: `label <TAB> dim <TAB> index:value ...`; dense:
    one value per column, the label in `LabelColumn` (int.MaxValue when LabelColumn is beyond the columns)."""
    f = line.rstrip("\r\n").split("\t")
    if SparseFormat:
        features = np.zeros(int(f[1]))
        for item in f[2:]:
            c, v = item.split(":")
            features[int(c)] = float(v) * NormalizationFactor
        return int(f[0]), features
    if LabelColumn >= len(f):
        label, features = 2 ** 31 - 1, np.array([float(x) for x in f])
    else:
        label, features = int(f[LabelColumn]), np.array([float(x) for k, x in enumerate(f) if k != LabelColumn])
    return label, (features * NormalizationFactor if normalize_dense else features)


class LLConvReader(BaseLayer):
    """NeuralNetworks/LLConvReader.cs: im2col of ONE record into a [corners x offsets] matrix (padding = 0, :139-147).  The record is
    the next line of `FileName` (formats as BatchReader, values times NormalizationFactor, :100-137) unless `Features` was set - a
    feature vector set by hand is taken as it is and used for one GetNext (:49-60,150).  None at end of file."""

    def __init__(self, FileName=None, Features=None, Scale=1.0, NormalizationFactor=1.0, InputShape=None, KernelShape=None, Stride=None,
                 Padding=None, Upperpadding=None, Lowerpadding=None, SparseFormat=True, LabelColumn=0, Factory=None):
        super().__init__(None, Factory)
        self.Scale, self.NormalizationFactor, self.SparseFormat, self.LabelColumn = Scale, NormalizationFactor, SparseFormat, LabelColumn
        self.engine = ConvolutionEngine(InputShape, KernelShape, Stride, Padding, Upperpadding, Lowerpadding)
        self.Labels, self._sr, self._dim, self._features = None, None, -1, None
        if FileName is not None:
            self.FileName = FileName
        self.Features = Features

    @property
    def FileName(self):
        return self._file_name

    @FileName.setter
    def FileName(self, value):
        self._file_name = value
        if self._sr is not None:
            self._sr.close()
        self._sr = open(value, "r")
        self._dim = -1

    @property
    def Features(self):
        return self._features

    @Features.setter
    def Features(self, value):
        self._features = None if value is None else np.asarray(value, dtype=np.float64)
        if self._features is not None:
            self._dim = self._features.size

    def PrepareNetwork(self):
        self.Prepare()

    def Apply(self, m):
        return self.GetNext()

    def GetNext(self):
        if self._features is None:
            line = self._sr.readline() if self._sr is not None else ""
            if line == "":
                return None
            label, self._features = _parse_record(line, self.SparseFormat, self.LabelColumn, self.NormalizationFactor, True)
            self.Labels = np.array([label], dtype=np.int64)
            self._dim = self._features.size
        g = self.engine.gather_table()
        mat = np.where(g >= 0, self._features[np.maximum(g, 0)], 0.0)
        self._features = None
        return RawData(mat, self.Scale)

    def GetOutputScale(self):
        return self.Scale

    def OutputDimension(self):
        return len(self.engine.Corners)

    def Dispose(self):
        if self._sr is not None:
            self._sr.close()
            self._sr = None


class LLSingleLineReader(BaseLayer):
    """NeuralNetworks/LLSingleLineReader.cs: one TSV line = one image per GetNext (formats as BatchReader), returned as a
    single-column matrix at Scale; None at end of file (:62-110)."""

    def __init__(self, FileName=None, NormalizationFactor=1.0, Scale=1.0, SparseFormat=True, LabelColumn=0, Factory=None):
        super().__init__(None, Factory)
        self.NormalizationFactor, self.Scale, self.SparseFormat, self.LabelColumn = NormalizationFactor, Scale, SparseFormat, LabelColumn
        self.Labels, self._sr, self._dim = None, None, -1
        if FileName is not None:
            self.FileName = FileName

    @property
    def FileName(self):
        return self._file_name

    @FileName.setter
    def FileName(self, value):
        self._file_name = value
        if self._sr is not None:
            self._sr.close()
        self._sr = open(value, "r")
        self._dim = -1

    def PrepareNetwork(self):
        self.Prepare()

    def Apply(self, m):
        return self.GetNext()

    def GetNext(self):
        line = self._sr.readline()
        if line == "":
            return None
        f = line.rstrip("\r\n").split("\t")
        if self.SparseFormat:
            self.Labels = np.array([int(f[0])], dtype=np.int64)
            self._dim = int(f[1])
            features = np.zeros(self._dim)
            for item in f[2:]:
                c, v = item.split(":")
                features[int(c)] = float(v) * self.NormalizationFactor
        else:                                                    # dense format: NormalizationFactor is NOT applied (:86-103)
            if self.LabelColumn >= len(f):
                self.Labels = np.array([2 ** 31 - 1], dtype=np.int64)
                features = np.array([float(x) for x in f])
            else:
                self.Labels = np.array([int(f[self.LabelColumn])], dtype=np.int64)
                features = np.array([float(x) for k, x in enumerate(f) if k != self.LabelColumn])
            self._dim = len(features)
        return RawData(features[:, None], self.Scale)

    def GetOutputScale(self):
        return self.Scale

    def OutputDimension(self):
        return self._dim

    def Dispose(self):
        if self._sr is not None:
            self._sr.close()
            self._sr = None


class LLPreConvLayer(BaseLayer):
    """NeuralNetworks/LLPreConvLayer.cs: the im2col of a convolution done HOMOMORPHICALLY on one packed image: for every kernel offset
    the pixels that offset selects are masked out block-wise (one block per residue of the corner row modulo the stride) and moved
    with `Permute` so that all offsets end up aligned slot for slot (:73-147).  Output: [outputDim x offsets] column-major;
    `HotIndices` marks the slots that hold a convolution window, `RearrangeWeights` puts per-corner weights into that slot order."""

    def __init__(self, Source=None, Factory=None, InputShape=None, KernelShape=None, Stride=None, Padding=None, Upperpadding=None,
                 Lowerpadding=None, UseAxisForBlocks=None):
        super().__init__(Source, Factory)
        self.engine = ConvolutionEngine(InputShape, KernelShape, Stride, Padding, Upperpadding, Lowerpadding)
        self.UseAxisForBlocks = UseAxisForBlocks
        self.outputDim, self.shifts, self.masks, self.CornersMap, self._hot = -1, None, None, None, None

    @property
    def HotIndices(self):
        if not self.layerPrepared:
            self.Prepare()
        return self._hot

    def _block_offsets(self):
        """BlockOffset (:31-60): odometer over the axes used for blocks, digit i in [0, Stride[i]), weight = row-major pitch"""
        E = self.engine
        nd = len(E.Stride)
        pitch = [1] * nd
        for i in range(1, nd):
            pitch[i] = pitch[i - 1] * E.InputShape[i - 1]
        block, offset, out = [0] * nd, 0, []
        while True:
            out.append(offset)
            go = False
            for i in range(nd):
                if not self.UseAxisForBlocks[i]:
                    continue
                block[i] += 1
                offset += pitch[i]
                if block[i] < E.Stride[i]:
                    go = True
                    break
                offset -= block[i] * pitch[i]
                block[i] = 0
            if not go:
                return out

    def Prepare(self):
        if self.layerPrepared:
            return
        E, F = self.engine, self.Factory
        if self.UseAxisForBlocks is None:
            self.UseAxisForBlocks = [True] * len(E.InputShape)
        n_off = len(E.Offsets)
        dim = int(np.prod(E.InputShape))
        block_offsets = self._block_offsets()
        nb = len(block_offsets)
        projections = sorted({c[0] for c in E.Corners}, key=[c[0] for c in E.Corners].index)
        expected = len(projections) / float(nb)
        small, large = int(np.floor(expected)), int(np.ceil(expected))
        n_large = len(projections) - nb * small
        self.CornersMap = [-1] * len(E.Corners)
        self.masks, self.shifts = [], []
        for i in range(n_off):
            selections = [[] for _ in range(nb)]
            sh = [0] * nb
            for j in range(nb):
                this_block = small if j > n_large else large
                sh[j] = (E.Location(None, E.Offsets[i], E.InputShape) if j == 0
                         else sh[j - 1] + block_offsets[j - 1] - block_offsets[j] + this_block * E.Stride[0] * dim // E.InputShape[0])
            for j, corner in enumerate(E.Corners):
                location = E.Location(corner, E.Offsets[i], E.InputShape)
                corner_id = (corner[0] - E.Corners[0][0]) // E.Stride[0]
                block = corner_id // large if corner_id < large * n_large else n_large + (corner_id - large * n_large) // small
                if location >= 0:
                    selections[block].append(location)
                    m = location - sh[block]
                    if self.CornersMap[j] >= 0 and self.CornersMap[j] != m:
                        raise Exception("Internal Error")
                    self.CornersMap[j] = m
            row = []
            for sel in selections:
                if sel:
                    v = np.zeros(dim)
                    v[sel] = 1.0
                    row.append(F.GetPlainVector(v, EVectorFormat.dense, 1))
                else:
                    row.append(None)
            self.masks.append(row)
            self.shifts.append(sh)
        per_row = dim // E.InputShape[0]
        large_max = 0 if n_large == 0 else per_row * (1 + E.Stride[0] * (large - 1)) + block_offsets[n_large - 1]
        small_max = per_row * (1 + E.Stride[0] * (small - 1)) + block_offsets[-1]
        self.outputDim = max(large_max, small_max)
        self._hot = np.zeros(self.outputDim)
        for x in self.CornersMap:
            self._hot[x] = 1.0
        self.layerPrepared = True

    def Apply(self, m):
        if m.ColumnCount != 1:
            raise Exception("Expecting only a single column")
        if not self.layerPrepared:
            self.Prepare()
        env = self.Factory.AllocateComputationEnv()
        v = m.GetColumn(0)
        res = [v.Permute(self.masks[k], self.shifts[k], self.outputDim, env) for k in range(len(self.masks))]
        return self.Factory.GetMatrix(res, EMatrixFormat.ColumnMajor, CopyVectors=False)

    def OutputDimension(self):
        if not self.layerPrepared:
            self.Prepare()
        return self.outputDim

    def RearrangeWeights(self, weights):
        """:156-169: per-corner weights [map][corner] -> [map][slot of that corner]"""
        if not self.layerPrepared:
            self.Prepare()
        weights = np.asarray(weights, dtype=np.float64)
        corners = len(self.engine.Corners)
        maps = len(weights) // corners
        out = np.zeros(maps * self.outputDim)
        for i in range(maps):
            for j in range(corners):
                out[i * self.outputDim + self.CornersMap[j]] = weights[j + i * corners]
        return out

    def Dispose(self):
        if self.masks is not None:
            for row in self.masks:
                for v in row:
                    if v is not None:
                        v.Dispose()
        self.masks = None


class TimingLayer(BaseLayer):
    """NeuralNetworks/TimingLayer.cs: a pass-through layer that starts / stops named wall-clock counters (the reference brackets the
    evaluated layers with it: "Batch-Time" CryptoNets.cs:31,74, "Prediction-Time" LoLaCryptonets.cs:66-67).  Device work is
    asynchronous here, so a stopping layer first waits for the streams of the environment."""
    TotalTimeMS, N, StartTime = {}, {}, {}

    def __init__(self, Source=None, Factory=None, StartCounters=(), StopCounters=()):
        super().__init__(Source, Factory)
        self.StartCounters, self.StopCounters = list(StartCounters), list(StopCounters)

    @classmethod
    def GetStats(cls, multiLines=False):
        return ("\n" if multiLines else "\t").join("%s %.2f" % (k, v / cls.N[k]) for k, v in cls.TotalTimeMS.items())

    @classmethod
    def Reset(cls):
        cls.TotalTimeMS.clear()
        cls.N.clear()
        cls.StartTime.clear()

    def Apply(self, m):
        import time
        if self.StopCounters or self.StartCounters:
            _sync_factory(self)
        now = time.perf_counter()
        for c in self.StartCounters:
            TimingLayer.StartTime[c] = now
        for c in self.StopCounters:
            if c in TimingLayer.StartTime:
                TimingLayer.TotalTimeMS[c] = TimingLayer.TotalTimeMS.get(c, 0.0) + 1e3 * (now - TimingLayer.StartTime[c])
                TimingLayer.N[c] = TimingLayer.N.get(c, 0) + 1
        return m


class WeightsReader:
    """NeuralNetworks/WeightsReader.cs: one array per CSV line, for the weights file and the biases file"""

    def __init__(self, weightsCsvPath, biasesCsvPath):
        self.Weights, self.Biases = self._read(weightsCsvPath), self._read(biasesCsvPath)

    @staticmethod
    def _read(path):
        with open(path, "r") as f:
            return [np.array([float(x) for x in line.split(",")]) for line in f.read().splitlines() if line != ""]


class LLPoolLayer(BaseLayer):
    """NeuralNetworks/LLPoolLayer.cs: convolution on the im2col matrix: per map one Mul(weight window) + bias (:112-137)."""

    def __init__(self, Source=None, Factory=None, InputShape=None, KernelShape=None, Stride=None, Padding=None, Upperpadding=None,
                 Lowerpadding=None, MapCount=None, Weights=None, Bias=None, WeightsScale=1.0, HotIndices=None):
        super().__init__(Source, Factory)
        self.engine = ConvolutionEngine(InputShape, KernelShape, Stride, Padding, Upperpadding, Lowerpadding, MapCount)
        self.Weights, self.Bias, self.WeightsScale, self.HotIndices = Weights, Bias, WeightsScale, HotIndices
        self.weightWindows = self.biasVectors = None

    def GetOutputScale(self):
        return (len(self.engine.Offsets) if self.Weights is None else self.WeightsScale) * self.Source.GetOutputScale()

    def OutputDimension(self):
        return len(self.engine.Corners) * (1 if self.Weights is None else self.engine.maps)

    def Prepare(self):
        if self.layerPrepared:
            return
        self.kernelSize = int(np.prod(self.engine.KernelShape)) + (1 if self.Bias is None else 0)
        if self.Weights is None:
            self.layerPrepared = True
            return
        F = self.Factory
        win = self.engine.weight_windows(self.Weights, self.kernelSize)
        self.weightWindows = [F.GetPlainVector(w, EVectorFormat.sparse, self.WeightsScale) for w in win]
        self.weightInts = [[int(round(float(x) * float(self.WeightsScale))) for x in w] for w in win]       # the same rounding as GetPlainVector
        hot = np.ones(len(self.engine.Corners)) if self.HotIndices is None else np.asarray(self.HotIndices, dtype=np.float64)
        bscale = self.Source.GetOutputScale() * self.WeightsScale
        src = self.Bias if self.Bias is not None else [self.Weights[(m + 1) * self.kernelSize - 1] for m in range(self.engine.maps)]
        self.biasVectors = [F.GetPlainVector(hot * src[m], EVectorFormat.dense, bscale) for m in range(self.engine.maps)]
        self.layerPrepared = True

    def Apply(self, m):
        if not self.layerPrepared:
            self.Prepare()
        env = self.Factory.AllocateComputationEnv()
        if self.Weights is None:
            vec = m.GetColumn(0)
            for i in range(1, m.ColumnCount):
                nxt = vec.Add(m.GetColumn(i), env)
                if i > 1:
                    vec.Dispose()
                vec = nxt
            vec.RegisterScale(vec.Scale * m.ColumnCount)
            return self.Factory.GetMatrix([vec], EMatrixFormat.ColumnMajor, CopyVectors=False)
        maps, K = len(self.biasVectors), m.ColumnCount
        from . import hewrapper as _hw
        batched = (not _hw.LITERAL and hasattr(m, "leVectors") and all(c.IsEncrypted and c.Format == EVectorFormat.dense for c in m.leVectors)
                   and all(a.encData.count == 1 for c in m.leVectors for a in c.eVectors) and all(any(w) for w in self.weightInts))
        if batched:
            # the `maps` (Mul + Add) pairs as ONE scalar GEMM with dense bias plaintexts per plaintext prime, planned once
            gather = np.tile(np.arange(K, dtype=np.int32), (maps, 1))
            return m.MulManySparse(gather, self.weightInts, None, m.Scale * self.WeightsScale, env, cache=self.__dict__.setdefault("_gemm_plans", {}),
                                   bias_vectors=self.biasVectors)
        res = []
        for k in range(maps):
            mul = m.Mul(self.weightWindows[k], env)
            res.append(mul.Add(self.biasVectors[k], env))
            mul.Dispose()
        return self.Factory.GetMatrix(res, EMatrixFormat.ColumnMajor, CopyVectors=False)


class LLVectorizeLayer(BaseLayer):
    """NeuralNetworks/LLVectorizeLayer.cs: stack the columns into one packed vector"""

    def Apply(self, m):
        vec = m.ConvertToColumnVector(self.Factory.AllocateComputationEnv())
        return self.Factory.GetMatrix([vec], EMatrixFormat.ColumnMajor, CopyVectors=False)


class LLDuplicateLayer(BaseLayer):
    """NeuralNetworks/LLDuplicateLayer.cs:11-29"""

    def __init__(self, Source=None, Factory=None, Count=1):
        super().__init__(Source, Factory)
        self.Count = Count

    def Apply(self, m):
        env = self.Factory.AllocateComputationEnv()
        return self.Factory.GetMatrix([m.GetColumn(i).Duplicate(self.Count, env) for i in range(m.ColumnCount)], m.Format, CopyVectors=False)

    def OutputDimension(self):
        shift, dim = 1, self.Source.OutputDimension()
        while shift < dim:
            shift *= 2
        return shift * int(self.Count)


class LLPackedDenseLayer(BaseLayer):
    """NeuralNetworks/LLPackedDenseLayer.cs: PackingCount weight rows share one plaintext (one per PackingShift slots); every packed
    row is one dense MultiplyPlain + SumAllSlots(PackingShift) + bias (:64-75)."""

    def __init__(self, Source=None, Factory=None, Weights=None, Bias=None, WeightsScale=1.0, PackingCount=1, PackingShift=0):
        super().__init__(Source, Factory)
        self.Weights, self.Bias, self.WeightsScale, self.PackingCount, self.PackingShift = Weights, Bias, WeightsScale, int(PackingCount), PackingShift
        self.WeightsMatrix = self.BiasMatrix = None

    def GetOutputScale(self):
        return self.WeightsScale * self.Source.GetOutputScale()

    def OutputDimension(self):
        return len(self.Bias)

    def Prepare(self):
        if self.layerPrepared:
            return
        maps = len(self.Bias)
        W = np.asarray(self.Weights, dtype=np.float64).reshape(maps, -1)
        rows = (maps + self.PackingCount - 1) // self.PackingCount
        stacked = np.zeros((rows, self.PackingCount * self.PackingShift))
        padded = np.zeros_like(stacked)
        for i in range(maps):
            col, row = i % self.PackingCount, i // self.PackingCount
            stacked[row, col * self.PackingShift: col * self.PackingShift + W.shape[1]] = W[i]
            padded[row, (col + 1) * self.PackingShift - 1] = self.Bias[i]
        self.BiasMatrix = self.Factory.GetPlainMatrix(padded, EMatrixFormat.RowMajor, self.Source.GetOutputScale() * self.WeightsScale)
        self.WeightsMatrix = self.Factory.GetPlainMatrix(stacked, EMatrixFormat.RowMajor, self.WeightsScale)
        self.layerPrepared = True

    def Apply(self, m):
        if not self.layerPrepared:
            self.Prepare()
        if m.ColumnCount > 1:
            raise Exception("Expecting only one column")
        env = self.Factory.AllocateComputationEnv()
        vector, res = m.GetColumn(0), []
        if getattr(self.WeightsMatrix, "_can_batch_rows", lambda v: False)(vector):
            # all packed rows at once: one MultiplyPlain / rotate-and-add launch chain per plaintext prime
            res = self.WeightsMatrix.RowsDotProduct(vector, env, length=self.PackingShift, bias=self.BiasMatrix)
        else:
            for k in range(self.WeightsMatrix.RowCount):
                mul = self.WeightsMatrix.GetRow(k).DotProduct(vector, env, length=self.PackingShift)
                res.append(mul.Add(self.BiasMatrix.GetRow(k), env))
                mul.Dispose()
        return self.Factory.GetMatrix(res, EMatrixFormat.ColumnMajor, CopyVectors=False)


class LLInterleaveLayer(BaseLayer):
    """NeuralNetworks/LLInterleaveLayer.cs: keep the selected slots of every column (mask multiply) and interleave the columns"""

    def __init__(self, Source=None, Factory=None, Shift=0, SelectedIndices=None, InputGrossDimension=-1):
        super().__init__(Source, Factory)
        self.Shift, self.SelectedIndices, self.InputGrossDimension = Shift, list(SelectedIndices), InputGrossDimension
        self.mask = None

    def Prepare(self):
        if self.layerPrepared:
            return
        if self.InputGrossDimension < 0:
            self.InputGrossDimension = max(self.SelectedIndices) + 1
        mv = np.zeros(self.InputGrossDimension)
        mv[self.SelectedIndices] = 1.0
        self.mask = self.Factory.GetPlainVector(mv, EVectorFormat.dense, 1)
        self.layerPrepared = True

    def OutputDimension(self):
        if not self.layerPrepared:
            self.Prepare()
        return self.InputGrossDimension

    def Apply(self, m):
        if not self.layerPrepared:
            self.Prepare()
        env = self.Factory.AllocateComputationEnv()
        cleanMat = m.MulColumnsByPlain(self.mask, env)               # all columns x the selection mask in one launch chain per prime
        interleaved = cleanMat.Interleave(self.Shift, env)
        cleanMat.Dispose()
        return self.Factory.GetMatrix([interleaved], EMatrixFormat.ColumnMajor, CopyVectors=False)


class LLInterleavedDenseLayer(BaseLayer):
    """NeuralNetworks/LLInterleavedDenseLayer.cs: dense layer whose inputs sit at the interleaved slots (:47-71)"""

    def __init__(self, Source=None, Factory=None, Weights=None, Bias=None, WeightsScale=1, Shift=0, SelectedIndices=None):
        super().__init__(Source, Factory)
        self.Weights, self.Bias, self.WeightsScale, self.Shift, self.SelectedIndices = Weights, Bias, WeightsScale, Shift, list(SelectedIndices)
        self.WeightsMatrix = self.BiasVector = None

    def GetOutputScale(self):
        return self.Source.GetOutputScale() * self.WeightsScale

    def OutputDimension(self):
        return len(self.Bias)

    def _target_indices(self, count):
        out, offset = [], 0
        while count > 0:
            for i in range(len(self.SelectedIndices)):
                if count <= 0:
                    break
                out.append(self.SelectedIndices[i] + offset)
                count -= 1
            offset += self.Shift
        return out

    def Prepare(self):
        if self.layerPrepared:
            return
        rows = len(self.Bias)
        small = np.asarray(self.Weights, dtype=np.float64).reshape(rows, -1)
        big = np.zeros((rows, self.Source.OutputDimension()))
        for i, t in enumerate(self._target_indices(small.shape[1])):
            big[:, t] = small[:, i]
        self.BiasVector = self.Factory.GetPlainVector(self.Bias, EVectorFormat.sparse, self.GetOutputScale())
        self.WeightsMatrix = self.Factory.GetPlainMatrix(big, EMatrixFormat.RowMajor, self.WeightsScale)
        self.layerPrepared = True

    def Apply(self, m):
        if not self.layerPrepared:
            self.Prepare()
        env = self.Factory.AllocateComputationEnv()
        mul = self.WeightsMatrix.Mul(m.GetColumn(0), env)
        v = mul.Add(self.BiasVector, env)
        mul.Dispose()
        return self.Factory.GetMatrix([v], EMatrixFormat.ColumnMajor, CopyVectors=False)
