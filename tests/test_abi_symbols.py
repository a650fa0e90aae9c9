"""The C-ABI library loads here (no GPU needed) and exports every symbol include/cnhip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cnhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from cryptonets_amd import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_library_exports_every_declared_symbol():
    from cryptonets_amd import _native
    _native.build()
    L = ctypes.CDLL(_native.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert _native.lib().cn_version() >= 100


def test_no_cpu_fallback_without_device():
    """Without a GPU the context constructor must fail loudly (CN_ERR_NODEV), never compute on the CPU."""
    from cryptonets_amd import _native
    if _native.lib().cn_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.CnError) as e:
        _native.Context(4096, 40961)
    assert e.value.code == -5


def test_default_coeff_modulus_table():
    from cryptonets_amd import _native
    from oracle.cno import COEFF_MODULUS_128
    for n, q in COEFF_MODULUS_128.items():
        assert _native.default_coeff_modulus(n) == q


def test_product_never_imports_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "cryptonets_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|libcnoracle|oracle/", re.M)
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                hits = [m.group(0) for m in pat.finditer(txt)]
                # the only tolerated mention is documentation saying the oracle is NOT used
                assert all("oracle/" == h for h in hits) and txt.count("oracle/") <= 2, (f, hits)
