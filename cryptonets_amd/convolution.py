"""Host-side geometry of the CryptoNets layers: which input ciphertext feeds which output with which weight.

Mirror of the reference's `NeuralNetworks/ConvolutionEngine.cs` (Offsets :39-59, Corners :61-79,
Location :83-96, GetDenseWeights :117-144) - pure index math, no arithmetic on ciphertexts.
"""
import numpy as np


class ConvolutionEngine:
    def __init__(self, InputShape, KernelShape, Stride, Padding=None, Upperpadding=None, Lowerpadding=None, MapCount=None):
        self.InputShape = list(InputShape)
        self.KernelShape = list(KernelShape)
        self.Stride = list(Stride)
        d = len(self.InputShape)
        self.Padding = list(Padding) if Padding is not None else [False] * d
        self.Upperpadding = list(Upperpadding) if Upperpadding is not None else [0] * d
        self.Lowerpadding = list(Lowerpadding) if Lowerpadding is not None else [0] * d
        self.MapCount = list(MapCount) if MapCount is not None else None
        self.maps = int(np.prod(self.MapCount)) if self.MapCount is not None else 1
        self.Offsets = self._offsets()
        self.Corners = self._corners()

    def _offsets(self):
        # OffsetGenerator: first dimension runs fastest (ConvolutionEngine.cs:39-54)
        ks, off, res = self.KernelShape, [0] * len(self.KernelShape), []
        while True:
            res.append(list(off))
            go = False
            for i in range(len(ks)):
                off[i] += 1
                if off[i] < ks[i]:
                    go = True
                    break
                off[i] = 0
            if not go:
                return res

    def _corners(self):
        # CornerGenerator: last dimension runs fastest (ConvolutionEngine.cs:61-79)
        ks = self.KernelShape
        mn = [-self.Lowerpadding[i] - (-(v // 2) if self.Padding[i] else 0) for i, v in enumerate(ks)]
        mx = [self.InputShape[i] + self.Upperpadding[i] - (((v + 1) // 2) if self.Padding[i] else v) for i, v in enumerate(ks)]
        off, res = list(mn), []
        while True:
            res.append(list(off))
            go = False
            for i in range(len(ks) - 1, -1, -1):
                off[i] += self.Stride[i]
                if off[i] <= mx[i]:
                    go = True
                    break
                off[i] = mn[i]
            if not go:
                return res

    @staticmethod
    def Location(Corner, offset, shape, bias=0):
        """Row-major index of corner+offset in `shape`, -1 for padding (ConvolutionEngine.cs:83-96)."""
        index = 0
        for i in range(len(offset)):
            cord = (Corner[i] + offset[i]) if Corner is not None else offset[i]
            if cord < 0 or cord >= shape[i]:
                return -1
            index = index * shape[i] + cord
        return index + bias

    def gather_table(self):
        """idx[corner][offset] = input column index or -1 (the patch matrix of PoolLayer.ConvolveOnce, PoolLayer.cs:113-121)."""
        return np.array([[self.Location(c, o, self.InputShape) for o in self.Offsets] for c in self.Corners], dtype=np.int32)

    def weight_windows(self, weights, kernel_size):
        """w[map][offset] as PoolLayer.PrepareWeightsWindows builds them (PoolLayer.cs:101-111)."""
        out = np.zeros((self.maps, len(self.Offsets)))
        for m in range(self.maps):
            for j, o in enumerate(self.Offsets):
                l = self.Location(None, o, self.KernelShape, m * kernel_size)
                out[m, j] = 0.0 if l < 0 else weights[l]
        return out

    def GetDenseBias(self, bias):
        return np.array([bias[i] for i in range(self.maps) for _ in self.Corners])

    def GetDenseWeights(self, weights):
        """the convolution unrolled into a [maps x corners, inputs] matrix, row major (ConvolutionEngine.cs:117-144)"""
        w = np.asarray(weights, dtype=np.float64)
        g = self.gather_table()                                      # [corners, offsets] -> input index or -1
        ksz, corners = int(np.prod(self.KernelShape)), len(self.Corners)
        kidx = np.array([self.Location(None, o, self.KernelShape) for o in self.Offsets])
        mat = np.zeros((self.maps * corners, int(np.prod(self.InputShape))))
        for i in range(corners):
            ok = g[i] >= 0
            rows = np.arange(self.maps) * corners + i
            mat[rows[:, None], g[i][ok][None, :]] = w[kidx[ok][None, :] + (np.arange(self.maps) * ksz)[:, None]]
        return mat.reshape(-1)
