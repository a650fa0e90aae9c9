import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# BFV parameter sets used across the suite (reference call sites in comments)
PARAMS = {
    # EncryptedSealBfvFactory() default: N=4096, CoeffModulus128(4096), first default plain prime (IFactory.cs:247-253)
    "default4096": dict(n=4096, t=40961, q=None, dbc=10, gdbc=20),
    # BASELINE config 2: N=8192, 2-prime RNS (SmallModulusCount: 2), CryptoNets plain prime
    "c2": dict(n=8192, t=549764251649, q=[0x7fffffd8001, 0x7fffffc8001], dbc=10, gdbc=20),
    # BASELINE config 3: CryptoNets-MNIST (CryptoNets.cs:17)
    "c3": dict(n=8192, t=549764251649, q=None, dbc=10, gdbc=20),
    # LoLa-MNIST (LoLaCryptonets.cs:208)
    "c4": dict(n=8192, t=557057, q=None, dbc=10, gdbc=20),
    # small ring for exhaustive/edge cases
    "tiny": dict(n=1024, t=12289, q=[0xffffee001, 0xffffc4001, 0x1ffffe0001], dbc=10, gdbc=20),
    # LoLa-Dense shapes: N=16384 with 7 of the CIFAR primes (LoLaCryptonets.cs:118-199 takes SmallModulusCount 7): k+1 primes below 2^49
    # are 4 bits short of a valid BEHZ auxiliary base here, k+2 are one: the N=16384 multiplications run on the FP64 kernels (k=7 and k=8)
    "n16k7": dict(n=16384, t=957181001729, q=[0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001,
                                               0x1ffffffee8001, 0x1ffffffea0001], dbc=60, gdbc=60),
    # BASELINE config 5 shapes: LoLa-CIFAR N=16384, k=8, dbc 60/60 (LolaCifarCryptoNet.cs:35)
    "c5": dict(n=16384, t=957181001729, q=[0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001,
                                            0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001], dbc=60, gdbc=60),
}

_oracles = {}


def get_oracle(name, galois=True, seed=11):
    """Keyed oracle context (cached per session)."""
    from oracle.cno import Oracle
    key = (name, galois)
    if key not in _oracles:
        p = PARAMS[name]
        o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"])
        o.keygen(seed, galois=galois)
        _oracles[key] = o
    return _oracles[key]


_gpu = {}


def get_gpu(name, galois=True, f64=True):
    """libcnhip context with the oracle's evaluation keys uploaded (keys are public material)."""
    from cryptonets_amd._native import Context
    key = (name, galois, f64)
    if key not in _gpu:
        p = PARAMS[name]
        o = get_oracle(name, galois)
        g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
        if not f64:
            g.set_option("f64", 0)          # integer (Shoup) transforms: the path moduli >= 2^49 take
        g.set_relin_key(o.relin_key())
        if galois:
            for i, e in enumerate(o.galois_elts()):
                g.set_galois_key(e, o.galois_key(i))
        _gpu[key] = g
    return _gpu[key]


@pytest.fixture
def rng():
    return np.random.default_rng(20250925)
