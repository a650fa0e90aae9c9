"""The start-up self-test of the drop-in (hewrapper.AtomicSealBfvEncryptedEnvironment.SelfTest; the C# twin's SelfTest is the same
procedure): the device must reproduce the WORDS of the evaluator it replaces, and where the reference leaves a choice that cannot be read
off its sources (SEAL 3.2 is an un-vendored dependency) - the decomposition convention of the key switch, the transform order of the
NTT-form key words - the test finds the combination the client obeys or raises.  VERDICT r03 weak #1 / next #1.

CPU: the "device" is a second oracle instance that receives the keys like a device does (tests/oracle_backend.py: OracleDeviceBackend).
GPU: the device is libcnhip; the client is the oracle built with either convention."""
import numpy as np
import pytest

from oracle.cno import Oracle
from oracle_backend import OracleClient, OracleDeviceBackend
from cryptonets_amd.hewrapper import AtomicSealBfvEncryptedEnvironment

PARAMS = [
    (1024, 12289, [0xffffee001, 0xffffc4001, 0x1ffffe0001], 10, 20),
    (2048, 40961, [0x7fffffd8001, 0x7fffffc8001], 16, 60),
]


def _bitrev_perm(n):
    bits = n.bit_length() - 1
    return np.array([int(format(i, "0%db" % bits)[::-1], 2) for i in range(n)])


class OtherOrderClient(OracleClient):
    """A client whose NTT-form key words are in NATURAL evaluation order (this library and the oracle use bit-reversed order): what a SEAL
    whose transform differs from SURVEY 9.2's recollection would hand over.  Its coefficient-form keys are, of course, the same polynomials."""

    def _reorder(self, words):
        w = np.asarray(words, dtype=np.uint64).reshape(-1, self.o.n)
        return w[:, _bitrev_perm(self.o.n)].reshape(-1).copy()

    def relin_key(self):
        return self._reorder(self.o.relin_key())

    def galois_keys(self):
        return {e: self._reorder(w) for e, w in super().galois_keys().items()}


def _client(cls, n, t, q, dbc, gdbc, xi):
    return cls(t, n, q, dbc, gdbc, seed=5, oracle=Oracle(n, t, q=q, dbc=dbc, gdbc=gdbc, ks_xi=xi))


def _cpu_device(n, t, q, dbc, gdbc):
    return OracleDeviceBackend(Oracle(n, t, q=q, dbc=dbc, gdbc=gdbc))


def _gpu_device(n, t, q, dbc, gdbc):
    from cryptonets_amd._native import Context
    return Context(n, t, q=q, dbc=dbc, gdbc=gdbc, device=0)


DEVICES = [pytest.param(_cpu_device, id="cpu"), pytest.param(_gpu_device, id="gpu", marks=pytest.mark.gpu)]


def _after_words_agree(env, client, galois=True):
    """after the self-test the drop-in is word-exact with the client's evaluator on operands the test did not use"""
    o, ctx = client.o, env.ctx
    rng = np.random.default_rng(99)
    a, b = (o.encrypt(o.encode(rng.integers(0, o.t, o.n, dtype=np.uint64))) for _ in range(2))
    h, out = ctx.ct_alloc(2), ctx.ct_alloc(1)
    ctx.ct_upload(h, 0, np.stack([a, b]))
    ctx.mul_relin(h, 0, h, 1, out, 0)
    assert np.array_equal(ctx.ct_download(out, 0, 1)[0], o.relinearize(o.multiply(a, b)))
    if galois:
        ctx.rotate_rows(h, 1, -3, out, 0)                    # NAF: two hops
        assert np.array_equal(ctx.ct_download(out, 0, 1)[0], o.rotate_rows(b, -3))
    ctx.free(h)
    ctx.free(out)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("n,t,q,dbc,gdbc", PARAMS, ids=["n1024k3", "n2048k2"])
@pytest.mark.parametrize("xi", [False, True], ids=["raw-digits", "xi-digits"])
@pytest.mark.parametrize("cls", [OracleClient, OtherOrderClient], ids=["same-order", "other-order"])
def test_self_test_finds_the_clients_convention(device, n, t, q, dbc, gdbc, xi, cls):
    client = _client(cls, n, t, q, dbc, gdbc, xi)
    env = AtomicSealBfvEncryptedEnvironment(device(n, t, q, dbc, gdbc), client)
    env.GenerateEncryptionKeys(with_galois=True)
    rep = env.self_test_report
    assert rep["ks_xi"] == int(xi)
    assert rep["key_form"] == ("ntt" if cls is OracleClient else "coeff")
    assert set(rep["ops"]) == {"MultiplyPlain", "MultiplyPlain(constant)", "AddPlain", "Multiply", "Relinearize", "RotateRows(1)", "RotateRows(-1)", "RotateColumns"}
    assert rep["tried"][-1][2] is None and all(bad is not None for _, _, bad in rep["tried"][:-1])
    assert len(rep["tried"]) == 1 + int(xi) + 2 * (cls is OtherOrderClient)
    assert env.ctx.get_option("ks_xi") == int(xi)
    _after_words_agree(env, client)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c3", "c5"])
@pytest.mark.parametrize("xi", [False, True], ids=["raw-digits", "xi-digits"])
def test_self_test_at_the_reference_parameter_sets(name, xi):
    """the same at BASELINE sizes: CryptoNets (N = 8192, 5 limbs, dbc 10 / 20: fused and two-launch key switches) and LoLa-CIFAR (N = 16384, 8 limbs, dbc 60:
    the two-halves key switch), clients of either convention with keys in the other transform order - the self-test ends on coefficient-form keys and the
    client's convention, and the drop-in is word-exact afterwards"""
    from conftest import PARAMS as P
    p = P[name]
    q = p["q"] if p["q"] is not None else __import__("oracle.cno", fromlist=["COEFF_MODULUS_128"]).COEFF_MODULUS_128[p["n"]]
    client = _client(OtherOrderClient, p["n"], p["t"], q, p["dbc"], p["gdbc"], xi)
    env = AtomicSealBfvEncryptedEnvironment(_gpu_device(p["n"], p["t"], q, p["dbc"], p["gdbc"]), client)
    env.GenerateEncryptionKeys(with_galois=True)
    rep = env.self_test_report
    assert rep["ks_xi"] == int(xi) and rep["key_form"] == "coeff" and len(rep["tried"]) == 3 + int(xi)
    _after_words_agree(env, client)
    env.ctx.close()


@pytest.mark.parametrize("device", DEVICES)
def test_self_test_without_galois_keys(device):
    n, t, q, dbc, gdbc = PARAMS[0]
    client = _client(OracleClient, n, t, q, dbc, gdbc, True)
    env = AtomicSealBfvEncryptedEnvironment(device(n, t, q, dbc, gdbc), client)
    env.GenerateEncryptionKeys(with_galois=False)
    assert env.self_test_report["ks_xi"] == 1 and "RotateColumns" not in env.self_test_report["ops"]
    _after_words_agree(env, client, galois=False)


@pytest.mark.parametrize("device", DEVICES)
def test_self_test_raises_when_nothing_matches(device):
    """keys that belong to another secret: every combination fails, the exception lists what was tried, the option is left as it was"""
    n, t, q, dbc, gdbc = PARAMS[0]
    client = _client(OracleClient, n, t, q, dbc, gdbc, False)

    class Wrong(OracleClient):
        def relin_key(self):
            w = self.o.relin_key()
            w[::7] ^= np.uint64(1)
            return w

        def relin_key_coeff_form(self):
            w = super().relin_key_coeff_form()
            w[::7] ^= np.uint64(1)
            return w
    client.__class__ = Wrong
    env = AtomicSealBfvEncryptedEnvironment(device(n, t, q, dbc, gdbc), client)
    with pytest.raises(Exception, match="no key-switch convention reproduces.*Relinearize"):
        env.GenerateEncryptionKeys(with_galois=True)
    assert env.ctx.get_option("ks_xi") == 0 and env.self_test_report is None


@pytest.mark.parametrize("device", DEVICES)
def test_self_test_names_an_arithmetic_disagreement(device):
    """a client evaluator whose MultiplyPlain lifts the plaintext differently: not a key convention - the test says which operation"""
    n, t, q, dbc, gdbc = PARAMS[0]
    client = _client(OracleClient, n, t, q, dbc, gdbc, False)

    class OtherLift:
        def __init__(self, o):
            self.o = o

        def __getattr__(self, name):
            return getattr(self.o, name)

        def multiply_plain(self, ct, plain):
            out = self.o.multiply_plain(ct, plain)
            out[3] = (int(out[3]) + 1) % self.o.q[0]
            return out
    client.reference_evaluator = lambda: OtherLift(client.o)
    env = AtomicSealBfvEncryptedEnvironment(device(n, t, q, dbc, gdbc), client)
    with pytest.raises(Exception, match="self-test: MultiplyPlain differs.*words AND decrypted slots"):
        env.GenerateEncryptionKeys(with_galois=True)


@pytest.mark.parametrize("device", DEVICES)
def test_self_test_tolerates_other_words_with_the_same_slots(device):
    """a client evaluator whose key-less operations return ANOTHER valid representative (here: Multiply / MultiplyPlain with an
    encryption of zero added - different words, identical decrypted slots, noise a little larger) is still served: the self-test records a warning per
    operation and continues; Relinearize is then compared by slots (it inherits the product's words), the rotations by words."""
    n, t, q, dbc, gdbc = PARAMS[0]
    client = _client(OracleClient, n, t, q, dbc, gdbc, True)

    class OtherRepresentative:
        def __init__(self, o):
            self.o = o
            self.zero = o.encrypt(o.encode(np.zeros(o.n, dtype=np.uint64)))

        def __getattr__(self, name):
            return getattr(self.o, name)

        def multiply_plain(self, ct, plain):
            return self.o.add(self.o.multiply_plain(ct, plain), self.zero)

        def multiply(self, a, b):
            out = self.o.multiply(a, b)                      # size 3: the zero encryption goes onto (c0, c1)
            two = 2 * self.o.k * self.o.n
            out[:two] = self.o.add(out[:two].copy(), self.zero)
            return out
    client.reference_evaluator = lambda: OtherRepresentative(client.o)
    env = AtomicSealBfvEncryptedEnvironment(device(n, t, q, dbc, gdbc), client)
    env.GenerateEncryptionKeys(with_galois=True)
    rep = env.self_test_report
    assert rep["ks_xi"] == 1 and rep["key_form"] == "ntt"
    assert sorted(w.split(":")[0] for w in rep["warnings"]) == ["Multiply", "MultiplyPlain", "MultiplyPlain(constant)"]
    _after_words_agree(env, client)


def test_a_client_without_an_evaluator_is_not_tested():
    n, t, q, dbc, gdbc = PARAMS[0]
    client = _client(OracleClient, n, t, q, dbc, gdbc, False)
    client.reference_evaluator = lambda: None
    env = AtomicSealBfvEncryptedEnvironment(_cpu_device(n, t, q, dbc, gdbc), client)
    env.GenerateEncryptionKeys(with_galois=False)
    assert env.self_test_report is None
