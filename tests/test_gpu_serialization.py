"""Client / server split through the wire formats on the device: the data owner generates keys, saves the
PUBLIC material (IFactory.Save, withPrivateKeys=false), writes encrypted vectors; the server loads both, evaluates, writes the
result; the owner reads and decrypts it.  Also: a full save with private keys restores a working client."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_client_server_round_trip():
    from cryptonets_amd._native import CnError
    from cryptonets_amd.hewrapper import EMatrixFormat, EncryptedSealBfvFactory, EVectorFormat
    client = EncryptedSealBfvFactory([40961, 65537], 4096)
    cenv = client.AllocateComputationEnv()
    public = io.BytesIO()
    client.Save(public, withPrivateKeys=False)
    v = np.array([3, -7, 11, 20, -2, 5, 1, 0], dtype=float)
    wire = io.StringIO()
    client.GetEncryptedVector(v, EVectorFormat.dense, 2.0).Write(wire, cenv)

    server = EncryptedSealBfvFactory.Load(io.BytesIO(public.getvalue()))
    senv = server.AllocateComputationEnv()
    x = server.LoadVector(io.StringIO(wire.getvalue()))
    sq = x.PointwiseMultiply(x, senv)                              # relinearisation key from the stream
    dot = x.DotProduct(x, senv, length=8)                          # Galois keys from the stream
    out = io.StringIO()
    sq.Write(out, senv)
    dot.Write(out, senv)
    with pytest.raises(CnError):                                   # the server holds no secret key
        sq.Decrypt(senv)

    back = io.StringIO(out.getvalue())
    got_sq, got_dot = client.LoadVector(back), client.LoadVector(back)
    assert got_sq.Scale == 4.0
    assert np.array_equal(got_sq.Decrypt(cenv)[:8], v * v)
    assert got_dot.Decrypt(cenv)[7] == float(v @ v)              # a partial sum lands in slot length-1 (BasicOperations.cs:139-149)

    # full save: the restored factory can encrypt and decrypt, and its ciphertexts are interchangeable with the original's
    full = io.BytesIO()
    client.Save(full, withPrivateKeys=True)
    twin = EncryptedSealBfvFactory.Load(io.BytesIO(full.getvalue()))
    tenv = twin.AllocateComputationEnv()
    assert np.array_equal(twin.LoadVector(io.StringIO(wire.getvalue())).Decrypt(tenv)[:8], v)
    m = np.array([[1, -2, 3], [4, 5, -6]], dtype=float)
    text = io.StringIO()
    twin.GetEncryptedMatrix(m, EMatrixFormat.ColumnMajor, 1.0).Write(text, tenv)
    assert np.array_equal(np.asarray(client.LoadMatrix(io.StringIO(text.getvalue())).Decrypt(cenv))[:2, :3], m)
    # a stream from a different parameter set is refused
    other = EncryptedSealBfvFactory([40961, 114689], 4096, galois=False)
    from cryptonets_amd.serialization import BadStream
    with pytest.raises(BadStream):
        other.LoadVector(io.StringIO(wire.getvalue()))
