"""The reference's plaintext factory: RawFactory / RawVector / RawMatrix (`HE Wrapper/IFactory.cs:132-238`,
`RawVector.cs`, `RawMatrix.cs`) - BASELINE config 0 ("BasicExample on the RawVector path: plumbing, no GPU, no SEAL").

This is NOT a fallback for the encrypted path: nothing selects it implicitly.  The reference's applications choose it
explicitly (`Encrypt ? new EncryptedSealBfvFactory(...) : new RawFactory(...)`, `LoLaCryptonets.cs:208`), its reader layers
hand their data over as RawMatrix (`BatchReader.cs:51`), and its layer tests run on it (`NeuralNetworksTest/LayersTest.cs:56`).
Values are doubles holding round(value * Scale), every operation re-rounds its result (the RawVector constructor rounds,
`RawVector.cs:24-31`), scales multiply under multiplication and must agree under addition - the same bookkeeping as the
encrypted classes, so a network can be debugged here and moved to `EncryptedSealBfvFactory` unchanged.
"""
import io

import numpy as np

from .hewrapper import EMatrixFormat, EVectorFormat


def _rounded(values, scale):
    a = np.asarray(values, dtype=np.float64) * float(scale)
    if a.size and not np.all(np.isfinite(a)):
        raise Exception("infinity")
    return np.rint(a)                                            # Math.Round / PointwiseRound: half to even, like rint


class RawComputationalEnvironment:
    """IFactory.cs:132-136"""

    def __init__(self, ParentFactory=None, Primes=None):
        self.ParentFactory, self.Primes = ParentFactory, Primes


class RawVector:
    """RawVector.cs:14-267"""
    Max = 0.0
    IsEncrypted = False

    def __init__(self, v=None, scale=1.0, BlockSize=0, Format=EVectorFormat.dense, integers=None):
        self.BlockSize, self.Format, self.IsSigned = int(BlockSize), Format, True
        if integers is not None:                                 # RawVector(IEnumerable<BigInteger>) :33-40
            self.Scale = 1.0
            self.v = np.rint(np.array([float(int(x)) for x in integers], dtype=np.float64))
        else:
            self.Scale = float(scale)
            self.v = None if v is None else _rounded(np.atleast_1d(np.asarray(v, dtype=np.float64)), scale)

    @classmethod
    def _of(cls, values, scale, like):
        r = cls(values, 1.0, like.BlockSize)                     # "new RawVector(res, 1, BlockSize); RegisterScale(...)"
        r.Scale = float(scale)
        return r

    def Copy(self):
        r = RawVector(None, self.Scale, self.BlockSize)          # copy constructor :50-56 (Format / IsSigned are not copied there)
        r.v = self.v.copy()
        return r

    def Dispose(self):
        self.v = None

    # ---- properties
    @property
    def Dim(self):
        return 0 if self.v is None else int(self.v.size)

    @property
    def Data(self):
        return self.v.copy()

    def RegisterScale(self, scale):
        self.Scale = float(scale)

    def _seen(self):
        if self.v.size:
            RawVector.Max = max(RawVector.Max, float(np.max(np.abs(self.v))))

    def Decrypt(self, env=None):
        self._seen()
        return self.v / self.Scale

    def DecryptFullPrecision(self, env=None):
        self._seen()
        return [int(x) if self.IsSigned else abs(int(x)) for x in self.v]

    # ---- persistence: BlockSize, Scale, then one value per line (`RawVector.cs:88-106`)
    def Write(self, stream):
        stream.write("%d\n%s\n" % (self.BlockSize, repr(float(self.Scale))))
        for x in self.v:
            stream.write(repr(float(x)) + "\n")
        stream.flush()

    @staticmethod
    def Read(stream):
        block = int(stream.readline())
        scale = float(stream.readline())
        vals = [float(line) for line in stream if line.strip()]
        r = RawVector(None, scale, block)
        r.v = np.array(vals, dtype=np.float64)
        return r

    # ---- algebra
    def _same_length(self, other):
        if self.v.size != other.v.size:
            raise Exception("Vectors dimensions do not match")

    def Add(self, v, env=None):
        if self.Scale == 0:
            return v
        if v.Scale == 0:
            return self
        if self.Scale != v.Scale:
            raise Exception("Scales do not match.")
        self._same_length(v)
        return RawVector._of(self.v + v.v, self.Scale, self)

    def Subtract(self, v, env=None):
        if v.Scale == 0:
            return self
        if self.Scale != 0 and self.Scale != v.Scale:
            raise Exception("Scales do not match.")
        self._same_length(v)
        return RawVector._of(self.v - v.v, self.Scale, self)

    def Multiply(self, x, env=None):
        return RawVector._of(self.v * float(x), self.Scale, self)

    def PointwiseMultiply(self, v, env=None):
        if self.v.size == v.v.size:
            mul = self.v * v.v
        elif self.v.size == 1 and self.Format == EVectorFormat.sparse:        # multiplying by a constant
            mul = v.v * self.v[0]
        elif v.v.size == 1 and v.Format == EVectorFormat.sparse:
            mul = self.v * v.v[0]
        else:
            raise Exception("Vectors dimensions do not match")
        return RawVector._of(mul, self.Scale * v.Scale, self)

    def SumAllSlots(self, env=None):
        return RawVector._of([np.sum(self.v)], self.Scale, self)

    def DotProduct(self, v, env=None, length=None):
        """length None: the scalar product in slot 0 (:144-153); length L: running sums over windows of the next power of two
        >= L, the total of slots [i-L+1, i] lands in slot i (:168-188) - where the encrypted rotate-and-add tree leaves it."""
        self._same_length(v)
        if length is None:
            return RawVector._of([np.dot(self.v, v.v)], self.Scale * v.Scale, self)
        res, skip = self.v * v.v, 1
        while skip < int(length):
            res = res + np.roll(res, skip)
            skip *= 2
        return RawVector._of(res, self.Scale * v.Scale, self)

    def Duplicate(self, count, env=None):
        shift = 1
        while shift < self.Dim:
            shift *= 2
        w = np.zeros(shift * int(count))
        for i in range(int(count)):
            w[i * shift: i * shift + self.Dim] = self.v
        return RawVector(w / self.Scale, self.Scale, self.BlockSize)

    def _rotated(self, vec, amount):
        """slot i takes slot (i + amount) mod BlockSize; slots beyond the vector read as zero (:220-231)"""
        src = (np.arange(vec.size) + int(amount)) % self.BlockSize
        ok = src < self.v.size
        w = np.zeros(self.v.size)
        w[:vec.size][ok] = vec[src[ok]]
        return w

    def Rotate(self, amount, env=None):
        return RawVector._of(self._rotated(self.v, amount), self.Scale, self)

    def Permute(self, selections, shifts, outputDim, env=None):
        if len(selections) != len(shifts):
            raise Exception("number of selection vectors and number of shifts does not match")
        res = np.zeros(self.Dim)
        for s, shift in zip(selections, shifts):
            if s is None:
                continue
            if s.Dim != self.Dim:
                raise Exception("dimension of selection vector does not match dimension of data vector")
            if s.Scale != selections[0].Scale:
                raise Exception("scales of all selection vectors should be the same")
            res = res + self._rotated(self.v * s.v, shift)
        return RawVector._of(res[:int(outputDim)], self.Scale * selections[0].Scale, self)


class RawMatrix:
    """RawMatrix.cs:12-174.  `m` is rows x columns; a ColumnMajor matrix hands out its columns, a RowMajor one its rows."""
    Max = 0.0
    IsEncrypted = False

    def __init__(self, m=None, scale=1.0, format=EMatrixFormat.ColumnMajor, BlockSize=0):
        self.Scale, self.Format, self.BlockSize = float(scale), format, int(BlockSize)
        self.DataDisposedExternaly = False
        self.m = None
        if m is not None:
            m = np.asarray(m, dtype=np.float64)
            if m.ndim != 2:
                raise Exception("expecting a two dimensional array")
            self.m = _rounded(m, scale)
            self._seen(m)

    @staticmethod
    def _seen(m):
        if m.size:
            RawMatrix.Max = max(RawMatrix.Max, float(np.max(np.abs(m))))

    def Dispose(self):
        self.m = None

    @property
    def RowCount(self):
        return int(self.m.shape[0])

    @property
    def ColumnCount(self):
        return int(self.m.shape[1])

    @property
    def Data(self):
        return self.m.copy()

    def RegisterScale(self, scale):
        self.Scale = float(scale)

    def Decrypt(self, env=None):
        self._seen(self.m)
        return self.m / self.Scale

    # ---- persistence: Scale, the rows tab separated, BlockSize (`RawMatrix.cs:58-63`)
    def Write(self, stream):
        stream.write(repr(float(self.Scale)) + "\n")
        for row in self.m:
            stream.write("\t".join(repr(float(x)) for x in row) + "\n")
        stream.write("%d\n" % self.BlockSize)
        stream.flush()

    @staticmethod
    def Read(stream):
        """Reads what Write wrote.  (The reference's Read, :43-54, hands the rest of the stream - BlockSize line included - to the
        delimited reader and then asks for one more line; the block size is taken from the last line here.)"""
        scale = float(stream.readline())
        lines = [ln for ln in stream.read().splitlines() if ln.strip()]
        r = RawMatrix(None, scale, EMatrixFormat.ColumnMajor, int(lines[-1]))
        r.m = np.array([[float(x) for x in ln.split("\t")] for ln in lines[:-1]], dtype=np.float64)
        RawMatrix._seen(r.m)
        return r

    # ---- algebra
    def Mul(self, v, env=None, ForceDenseFormat=False):
        if self.m.shape[1] != v.Dim:
            raise Exception("Matrix dimensions must agree")
        return RawVector._of(self.m @ v.Data, self.Scale * v.Scale, v)

    def _check(self, m):
        if m.Format != self.Format:
            raise Exception("Format mismatch")
        if m.RowCount != self.RowCount:
            raise Exception("Row count mismatch")
        if m.ColumnCount != self.ColumnCount:
            raise Exception("Column count mismatch")

    def ElementWiseMultiply(self, m, env=None):
        self._check(m)
        r = RawMatrix(self.m * m.m, 1, self.Format, m.BlockSize)
        r.Scale = self.Scale * m.Scale
        return r

    def Add(self, m, env=None):
        """`new RawMatrix(this.m.Add(mr.m), Scale, ...)` (:90-99): the constructor multiplies the (already scaled) sum by Scale
        once more - the reference's behaviour for Scale != 1, kept as is; no layer of the networks adds raw matrices."""
        self._check(m)
        if m.Scale != self.Scale:
            raise Exception("Scale mismatch")
        return RawMatrix(self.m + m.m, self.Scale, self.Format, m.BlockSize)

    def GetColumn(self, columnNumber):
        if columnNumber >= self.ColumnCount:
            raise Exception("Column does not exist")
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Columns can be extracted only from a column major matrix")
        r = RawVector(self.m[:, columnNumber], 1, self.BlockSize)
        r.Scale = self.Scale
        return r

    def GetRow(self, rowNumber):
        if rowNumber >= self.RowCount:
            raise Exception("Row does not exist")
        if self.Format != EMatrixFormat.RowMajor:
            raise Exception("Row can be extracted only from a row major matrix")
        r = RawVector(self.m[rowNumber, :], 1, self.BlockSize)
        r.Scale = self.Scale
        return r

    def SetColumn(self, columnNumber, vector):
        if columnNumber >= self.ColumnCount:
            raise Exception("Column does not exist")
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Columns can be set only from a column major matrix")
        self.m[:, columnNumber] = vector.Data
        self._seen(self.m)

    def ConvertToColumnVector(self, env=None):
        if self.ColumnCount * self.RowCount > self.BlockSize:
            raise Exception("block too long for interleaving")
        r = RawVector(self.m.T.reshape(-1), 1, self.BlockSize)   # column after column
        r.Scale = self.Scale
        return r

    def Interleave(self, shift, env=None):
        """column i moved by shift*i slots (towards the end for shift > 0, towards the start for shift < 0) and summed (:142-167)"""
        if shift == 0:
            raise Exception("number of items cannot be zero")
        n = self.RowCount
        w = self.m[:, 0].copy()
        for i in range(1, self.ColumnCount):
            s, col = shift * i, self.m[:, i]
            if abs(s) >= n:
                continue
            if s < 0:
                w[:n + s] += col[-s:]
            else:
                w[s:] += col[:n - s]
        r = RawVector(w, 1, self.BlockSize)
        r.Scale = self.Scale
        return r

    # ---- the batched entry points the layers call on the encrypted classes, restated on plain numbers so that PoolLayer /
    # LLPoolLayer / LLInterleaveLayer run on either factory (each equals the loop of Mul / Add / PointwiseMultiply it replaces)
    def MulManySparse(self, gather, weights, bias, out_scale, env=None, cache=None, bias_vectors=None):
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Expecting ColumnMajor matrix")
        g = np.asarray(gather, dtype=np.int64)
        out = np.zeros((self.RowCount, len(weights)))
        for o, row in enumerate(weights):
            for k, wgt in enumerate(row):
                if g[o, k] >= 0 and wgt:
                    out[:, o] += float(wgt) * self.m[:, g[o, k]]
            if bias_vectors is not None:
                out[:, o] += bias_vectors[o].Data
            elif bias is not None:
                out[:, o] += float(bias[o])
        r = RawMatrix(out, 1, EMatrixFormat.ColumnMajor, self.BlockSize)
        r.Scale = float(out_scale)
        return r

    def MulColumnsByPlain(self, plain, env=None):
        if self.Format != EMatrixFormat.ColumnMajor:
            raise Exception("Expecting ColumnMajor matrix")
        if plain.Dim != self.RowCount:
            raise Exception("Vectors dimensions do not match")
        r = RawMatrix(self.m * plain.Data[:, None], 1, self.Format, self.BlockSize)
        r.Scale = self.Scale * plain.Scale
        return r


class RawFactory:
    """IFactory.cs:138-238"""

    def __init__(self, BlockSize):
        self.BlockSize = int(BlockSize)
        self.Primes = None
        self._parent = None

    def GetPlainVector(self, v, format, scale=None):
        if scale is None:                                        # the IEnumerable<BigInteger> overload (:154-158)
            return RawVector(integers=v, BlockSize=self.BlockSize, Format=format)
        return RawVector(v, scale, self.BlockSize, Format=format)

    GetEncryptedVector = GetPlainVector                          # :160-169: the same object, nothing is encrypted

    def CopyVector(self, v):
        return v.Copy()

    def GetPlainMatrix(self, m, format, scale):
        return RawMatrix(m, scale, format, self.BlockSize)

    GetEncryptedMatrix = GetPlainMatrix

    def GetMatrix(self, vectors, format, CopyVectors=True):
        scale = vectors[0].Scale
        lengths = {v.Dim for v in vectors}
        if len(lengths) != 1:
            raise Exception("all vectors must have the same dimension")
        return RawMatrix(np.stack([v.Data / scale for v in vectors], axis=1), scale, format, self.BlockSize)

    def LoadVector(self, stream):
        return RawVector.Read(stream)

    def LoadMatrix(self, stream):
        return RawMatrix.Read(stream)

    def AllocateComputationEnv(self):
        if self._parent is None:
            self._parent = RawComputationalEnvironment(ParentFactory=self, Primes=self.Primes)
        return self._parent

    def FreeComputationEnv(self, env):
        pass

    def Save(self, target, withPrivateKeys=False):
        text = "<RawFactory>\n%d\n</RawFactory>\n" % self.BlockSize
        if isinstance(target, (str, bytes)):
            with open(target, "w") as f:
                f.write(text)
            return None
        target.write(text if isinstance(target, io.TextIOBase) else text.encode())
        return target

    def GetValueFromString(self, s):
        return int(s)

    def GetStringFromValue(self, value):
        return str(int(value))


class Defaults:
    """HE Wrapper/Defaults.cs:10: the factory the reader layers build their RawMatrix with"""
    RawFactory = RawFactory(8192)
