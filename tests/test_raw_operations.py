"""BASELINE config 0 - the reference's plaintext path.  Port of `HE Wrapper Tests/BasicRawOperations.cs` (same names, same
values, exact equality), the two raw tests of `BasicOperations.cs` (InterleaveRaw :249-259, PatrialSumAll :333-345) and the
`Basic Example` program (`Basic Example/Program.cs:32-47`) on `cryptonets_amd.raw`.  No GPU, no oracle: plain doubles."""
import io

import numpy as np
import pytest

from cryptonets_amd.hewrapper import EMatrixFormat, EVectorFormat
from cryptonets_amd.raw import Defaults, RawFactory, RawMatrix, RawVector

values1 = np.array([-1, 9, 3, 20, 1000, -6945], dtype=float)
values2 = np.array([8, -22, 5, 4, 254, -12], dtype=float)
scale = 17.0
values_m = np.array([[1, -2, 3, -44, 5, 7], [99, 12, -88, 22, 16, 13]], dtype=float)
Factory = Defaults.RawFactory


@pytest.fixture()
def fx():
    class F:
        env = Factory.AllocateComputationEnv()
        enc1 = Factory.GetEncryptedVector(values1, EVectorFormat.dense, scale)
        enc2 = Factory.GetEncryptedVector(values2, EVectorFormat.dense, scale)
        plain2 = Factory.GetPlainVector(values2, EVectorFormat.dense, scale)
        mat = Factory.GetEncryptedMatrix(values_m, EMatrixFormat.ColumnMajor, scale)
    return F


def Compare(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape
    assert np.array_equal(a, b), (a, b)


def test_RawDecrypt(fx):
    Compare(values1, fx.enc1.Decrypt(fx.env))


def test_RawDecryptMatrix(fx):
    Compare(values_m, fx.mat.Decrypt(fx.env))


def test_RawMatrixColumn(fx):
    Compare(values_m[:, 0], fx.mat.GetColumn(0).Decrypt(fx.env))


def test_RawMatrixVectorMultiplication(fx):
    enc_sparse = Factory.GetEncryptedVector(values1, EVectorFormat.sparse, scale)
    Compare(values_m @ values1, fx.mat.Mul(enc_sparse, fx.env).Decrypt(fx.env))


def test_RawAdd(fx):
    Compare(values1 + values2, fx.enc1.Add(fx.enc2, fx.env).Decrypt(fx.env))
    Compare(values1 + values2, fx.enc1.Add(fx.plain2, fx.env).Decrypt(fx.env))


def test_RawElementMultiply(fx):
    Compare(values1 * values2, fx.enc1.PointwiseMultiply(fx.enc2, fx.env).Decrypt(fx.env))
    Compare(values1 * values2, fx.enc1.PointwiseMultiply(fx.plain2, fx.env).Decrypt(fx.env))


def test_RawDotProduct(fx):
    assert fx.enc1.DotProduct(fx.enc2, fx.env).Decrypt(fx.env)[0] == values1 @ values2
    assert fx.enc1.DotProduct(fx.plain2, fx.env).Decrypt(fx.env)[0] == values1 @ values2


def test_RawSum(fx):
    assert fx.enc1.SumAllSlots(fx.env).Decrypt(fx.env)[0] == values1.sum()


def test_RawSubtract(fx):
    Compare(values1 - values2, fx.enc1.Subtract(fx.enc2, fx.env).Decrypt(fx.env))
    Compare(values1 - values2, fx.enc1.Subtract(fx.plain2, fx.env).Decrypt(fx.env))


def test_RawMeta(fx):
    assert fx.enc1.IsEncrypted is False and fx.plain2.IsEncrypted is False
    assert fx.enc1.Scale == scale
    enc2 = Factory.CopyVector(fx.enc1)
    enc2.RegisterScale(20)
    Compare(values1 * scale / 20, enc2.Decrypt(fx.env))
    Compare(values1, fx.enc1.Decrypt(fx.env))                      # the copy owns its data


def test_RawPermute():
    factory = RawFactory(8192)
    env = factory.AllocateComputationEnv()
    v = factory.GetEncryptedVector(np.arange(1, 11, dtype=float), EVectorFormat.dense, 1)
    S1, S2 = np.zeros(10), np.zeros(10)
    S1[[1, 4]] = 1.0
    S2[[3, 6]] = 1.0
    sel1 = factory.GetPlainVector(S1, EVectorFormat.dense, 1)
    sel2 = factory.GetPlainVector(S2, EVectorFormat.dense, 1)
    w = v.Permute([sel1, sel2], [1, 2], 5, env)
    Compare([2, 4, 0, 5, 7], w.Decrypt(env))
    factory.FreeComputationEnv(env)


def test_BigStackRaw():
    factory = RawFactory(4096)
    n = 1011
    v = [factory.GetEncryptedVector(np.arange(i * n, (i + 1) * n, dtype=float), EVectorFormat.dense, 1) for i in range(4)]
    m = factory.GetMatrix(v, EMatrixFormat.ColumnMajor)
    env = factory.AllocateComputationEnv()
    Compare(np.arange(4 * n, dtype=float), m.ConvertToColumnVector(env).Decrypt(env))
    with pytest.raises(Exception, match="block too long for interleaving"):
        RawFactory(4096).GetMatrix(v + [v[0]], EMatrixFormat.ColumnMajor).ConvertToColumnVector(env)


def test_InterleaveRaw():
    """BasicOperations.cs:249-259"""
    mat = np.array([[1, 0, 0, 2, 0, 0], [3, 0, 0, 4, 0, 0]], dtype=float).T
    m = RawFactory(4096).GetEncryptedMatrix(mat, EMatrixFormat.ColumnMajor, 10)
    Compare([1, 3, 0, 2, 4, 0], m.Interleave(1, None).Decrypt(None))
    rev = np.array([[0, 0, 1, 0, 0, 2], [0, 0, 3, 0, 0, 4], [0, 0, 5, 0, 0, 6]], dtype=float).T       # InterleaveReverse :277-290
    Compare([5, 3, 1, 6, 4, 2], RawFactory(4096).GetEncryptedMatrix(rev, EMatrixFormat.ColumnMajor, 10).Interleave(-1, None).Decrypt(None))


def test_PatrialSumAll():
    """BasicOperations.cs:333-345"""
    factory = RawFactory(8192)
    env = factory.AllocateComputationEnv()
    v = np.zeros(1280)
    v[0] = 1
    vec = factory.GetEncryptedVector(v, EVectorFormat.dense, 1)
    w = vec.DotProduct(vec, env, length=128).Decrypt(env)
    Compare((np.arange(1280) < 128).astype(float), w)


def test_basic_example():
    """`Basic Example/Program.cs:32-47` with `new RawFactory(4096)` (:16): norm squared, sum of elements, elementwise product"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("basic_example", os.path.join(os.path.dirname(__file__), "..", "examples", "basic_example.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(RawFactory(4096))
    assert out["norm_squared"] == [14.0] and out["sum"] == [6.0] and out["elementwise"] == [-1.0, 10.0, -12.0]


def test_scale_rules_and_errors(fx):
    other = Factory.GetEncryptedVector(values2, EVectorFormat.dense, 3.0)
    with pytest.raises(Exception, match="Scales do not match."):
        fx.enc1.Add(other, fx.env)
    with pytest.raises(Exception, match="Scales do not match."):
        fx.enc1.Subtract(other, fx.env)
    zero = RawVector(np.zeros(6), 0.0, 8192)
    assert fx.enc1.Add(zero, fx.env) is fx.enc1 and zero.Add(fx.enc1, fx.env) is fx.enc1 and fx.enc1.Subtract(zero, fx.env) is fx.enc1
    assert fx.enc1.PointwiseMultiply(fx.enc2, fx.env).Scale == scale * scale
    with pytest.raises(Exception, match="Vectors dimensions do not match"):
        fx.enc1.PointwiseMultiply(Factory.GetPlainVector([1.0, 2.0], EVectorFormat.dense, 1.0), fx.env)
    const = Factory.GetPlainVector([3.0], EVectorFormat.sparse, 2.0)
    Compare(values1 * 3, fx.enc1.PointwiseMultiply(const, fx.env).Decrypt(fx.env))
    with pytest.raises(Exception, match="infinity"):
        Factory.GetPlainVector([np.inf], EVectorFormat.dense, 1.0)
    Compare(np.rint(values1 * scale * 0.5) / scale, fx.enc1.Multiply(0.5, fx.env).Decrypt(fx.env))       # Multiply(double) re-rounds
    row = RawFactory(64).GetPlainMatrix(values_m, EMatrixFormat.RowMajor, 1)
    with pytest.raises(Exception, match="Columns can be extracted only from a column major matrix"):
        row.GetColumn(0)
    with pytest.raises(Exception, match="Row can be extracted only from a row major matrix"):
        fx.mat.GetRow(0)
    with pytest.raises(Exception, match="Column does not exist"):
        fx.mat.GetColumn(6)
    with pytest.raises(Exception, match="Format mismatch"):
        fx.mat.ElementWiseMultiply(row, fx.env)
    sq = fx.mat.ElementWiseMultiply(fx.mat, fx.env)
    assert sq.Scale == scale * scale
    Compare(values_m * values_m, sq.Decrypt(fx.env))


def test_rotate_duplicate_and_integers():
    f = RawFactory(16)
    v = f.GetEncryptedVector(np.arange(1, 7, dtype=float), EVectorFormat.dense, 2.0)
    Compare([3, 4, 5, 6, 0, 0], v.Rotate(2, None).Decrypt(None))          # slots beyond Dim read as zero, the wheel has BlockSize slots
    Compare([0, 0, 1, 2, 3, 4], v.Rotate(-2, None).Decrypt(None))
    d = v.Duplicate(3, None)
    assert d.Dim == 24 and d.Scale == 2.0
    Compare(np.tile(np.concatenate([np.arange(1, 7), [0, 0]]), 3), d.Decrypt(None))
    big = f.GetEncryptedVector([2 ** 40, -3], EVectorFormat.dense)         # the BigInteger overload: scale 1
    assert big.Scale == 1.0 and big.DecryptFullPrecision(None) == [2 ** 40, -3]
    big.IsSigned = False
    assert big.DecryptFullPrecision(None) == [2 ** 40, 3]
    assert f.GetValueFromString("123456789012345678901234567890") == 123456789012345678901234567890
    assert f.GetStringFromValue(-17) == "-17"


def test_persistence_roundtrip(fx):
    s = io.StringIO()
    fx.enc1.Write(s)
    s.seek(0)
    back = Factory.LoadVector(s)
    assert back.BlockSize == 8192 and back.Scale == scale
    Compare(values1, back.Decrypt(fx.env))
    s = io.StringIO()
    fx.mat.Write(s)
    s.seek(0)
    mback = Factory.LoadMatrix(s)
    assert mback.BlockSize == 8192 and mback.Scale == scale
    Compare(values_m, mback.Decrypt(fx.env))
    s = io.StringIO()
    RawFactory(4096).Save(s)
    assert s.getvalue().split() == ["<RawFactory>", "4096", "</RawFactory>"]


def test_batched_entries_equal_the_loops_they_replace():
    """RawMatrix.MulManySparse / MulColumnsByPlain (what PoolLayer / LLInterleaveLayer call) against Mul + Add / PointwiseMultiply"""
    rng = np.random.default_rng(5)
    f = RawFactory(64)
    data = rng.integers(-9, 10, size=(7, 6)).astype(float)                 # 6 columns of 7 samples
    m = f.GetEncryptedMatrix(data, EMatrixFormat.ColumnMajor, 1)
    m.RegisterScale(4.0)
    gather = np.array([[0, 2, -1], [5, 1, 3]], dtype=np.int32)
    weights = [[2, -3, 7], [1, 0, -5]]
    out = m.MulManySparse(gather, weights, [10, -20], 8.0, None)
    assert isinstance(out, RawMatrix) and out.Scale == 8.0 and out.ColumnCount == 2
    Compare(np.stack([2 * data[:, 0] - 3 * data[:, 2] + 10, data[:, 5] - 5 * data[:, 3] - 20], axis=1), out.Data)
    mask = f.GetPlainVector([1, 0, 1, 0, 1, 0, 1], EVectorFormat.dense, 1)
    masked = m.MulColumnsByPlain(mask, None)
    for j in range(6):
        Compare(m.GetColumn(j).PointwiseMultiply(mask, None).Data, masked.GetColumn(j).Data)
