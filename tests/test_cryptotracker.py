"""CryptoTracker (`HE Wrapper/CryptoTracker.cs`): Decryptor.InvariantNoiseBudget probes and the MinBudgetSoFar watermark.
CPU: the wrapper logic on the oracle client.  GPU: cn_noise_poly against the oracle's c0 + c1 s (+ c2 s^2), word for word."""
import numpy as np
import pytest

from conftest import PARAMS
from oracle_backend import make_factory
from cryptonets_amd.cryptotracker import CryptoTracker, INT_MAX
from cryptonets_amd.hewrapper import EVectorFormat

BACKENDS = [pytest.param("cpu"), pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("backend", BACKENDS)
def test_watermark_follows_the_computation(backend, capsys):
    Factory = make_factory(backend, primes=[65537, 114689], n=4096, galois=True)
    env = Factory.AllocateComputationEnv()
    CryptoTracker.Reset()
    CryptoTracker.DisableBudgetTests()
    v = Factory.GetEncryptedVector(np.arange(1.0, 9.0), EVectorFormat.dense, 1)
    CryptoTracker.TestBudget(v, Factory)                              # off: nothing happens (the reference's RELEASE build)
    assert CryptoTracker.MinBudgetSoFar == INT_MAX
    CryptoTracker.EnableBudgetTests()
    try:
        CryptoTracker.TestBudget(v, Factory)
        fresh = CryptoTracker.MinBudgetSoFar
        assert 40 < fresh < 109                                       # 109-bit q, 17-bit t
        sq = v.PointwiseMultiply(v, env)
        CryptoTracker.TestBudget(sq, Factory)
        after_mul = CryptoTracker.MinBudgetSoFar
        assert 0 < after_mul < fresh - 15
        CryptoTracker.TestBudget(v, Factory)                          # a larger budget does not move the watermark
        assert CryptoTracker.MinBudgetSoFar == after_mul
        rot = sq.Rotate(1, env)
        assert CryptoTracker.TestVectorBudget(rot, env) <= after_mul
        out = capsys.readouterr().out
        assert "Warning: Current minimal budget %d" % fresh in out and "Warning: Current minimal budget %d" % after_mul in out
        CryptoTracker.Show(v, Factory, "v")
        assert "v size 8" in capsys.readouterr().out
    finally:
        CryptoTracker.DisableBudgetTests()
        CryptoTracker.Reset()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c4", "c5"])
def test_noise_poly_matches_the_oracle(name, rng):
    """size-2 and size-3 ciphertexts: every residue of t*(c0 + c1 s + c2 s^2), and the budgets derived from them"""
    from cryptonets_amd._native import Context
    from oracle.cno import Oracle
    from oracle_backend import OracleClient
    p = PARAMS[name]
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    g.keygen(123, galois=False)
    o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"])
    o.import_keys(g.get_key(3), g.get_key(2))
    oc = OracleClient(p["t"], p["n"], p["q"], p["dbc"], p["gdbc"], oracle=o)
    a = o.encrypt(o.encode(rng.integers(0, p["t"], size=p["n"], dtype=np.uint64)))
    b = o.encrypt(o.encode(rng.integers(0, p["t"], size=p["n"], dtype=np.uint64)))
    c3 = o.multiply(a, b)
    h2, h3 = g.ct_alloc(2, 2), g.ct_alloc(1, 3)
    g.ct_upload(h2, 0, np.stack([a, b]))
    g.ct_upload(h3, 0, c3[None, :])
    got2, got3 = g.noise_poly(h2, 0, 2), g.noise_poly(h3, 0, 1)
    assert np.array_equal(got2[0], oc.noise_poly(a)) and np.array_equal(got2[1], oc.noise_poly(b))
    assert np.array_equal(got3[0], oc.noise_poly(c3))
    assert g.invariant_noise_budget(h2, 0, 2, exact_bits=True) == [oc.noise_budget_words(a), oc.noise_budget_words(b)]
    b3 = g.invariant_noise_budget(h3, 0, 1, exact_bits=True)[0]
    assert b3 == oc.noise_budget_words(c3) and b3 < oc.noise_budget_words(a)      # c2 (86-bit q, 39-bit t): one multiply leaves 0
    f, i2 = g.invariant_noise_budget(h2, 0, 1)[0], oc.noise_budget_words(a)
    assert i2 - 1 <= f < i2 + 1.01                                     # the float version brackets SEAL's integer one
    g.free(h2)
    g.free(h3)


@pytest.mark.gpu
def test_device_client_probes_on_the_device(capsys):
    """no oracle in the loop: keys, encryption and the budget probe all on the MI355X (DeviceClient.noise_budget -> cn_noise_poly)"""
    from cryptonets_amd.hewrapper import EncryptedSealBfvFactory
    Factory = EncryptedSealBfvFactory([65537], 4096)
    env = Factory.AllocateComputationEnv()
    v = Factory.GetEncryptedVector(np.arange(1.0, 9.0), EVectorFormat.dense, 1)
    CryptoTracker.Reset()
    CryptoTracker.EnableBudgetTests()
    try:
        CryptoTracker.TestBudget(v, Factory)
        fresh = CryptoTracker.MinBudgetSoFar
        assert 40 < fresh < 109
        CryptoTracker.TestBudget(v.PointwiseMultiply(v, env), Factory)
        assert 0 < CryptoTracker.MinBudgetSoFar < fresh - 15
    finally:
        CryptoTracker.DisableBudgetTests()
        CryptoTracker.Reset()
