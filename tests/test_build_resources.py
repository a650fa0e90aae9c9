"""Register budgets of the BUILT hot kernels (code-object metadata, tools/kernel_resources.py): the occupancy each kernel was designed for only
holds while the compiler stays inside the budget, and a spill in one of them is a silent 10-30 % - caught here, on CPU, at build time."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
OBJ = os.path.join(ROOT, "cryptonets_amd", "lib", "obj")

# (object file, kernel, VGPR budget, why)
BUDGETS = [
    ("cn_l_rr_f64l.o", "void k_ntt_rr<13, ArF64T<0>, false>", 128, "two 512-thread workgroups per CU (4 waves per SIMD)"),
    ("cn_l_rr_f64l.o", "void k_ntt_rr<13, ArF64T<0>, true>", 128, "two 512-thread workgroups per CU"),
    ("cn_l_rr_f64.o", "void k_ntt_rr<14, ArF64T<1>, false>", 128, "one 1024-thread workgroup per CU (4 waves per SIMD)"),
    ("cn_l_rr_f64l.o", "void k_intt_tensor<13, ArF64T<0> >", 128, "two workgroups per CU"),
    ("cn_l_rr_f64l.o", "void k_square_fused<13, ArF64T<0>, true>", 256, "one workgroup per CU: image + parked operand in LDS"),
    ("cn_l_rr_f64.o", "void k_square_fused<13, ArF64T<1>, true>", 256, "one workgroup per CU"),
    ("cn_l_rr_f64l.o", "void k_mul_plain_fused<13, ArF64T<0> >", 128, "two workgroups per CU"),
    ("cn_l_ks_f64l.o", "void k_keyswitch_rr<13, ArF64T<0>, 1, true, false>", 256, "one workgroup per CU (image + LDS twiddles), two waves per SIMD"),
    ("cn_l_ks_f64.o", "void k_keyswitch_rr<13, ArF64T<1>, 1, true, false>", 256, "one workgroup per CU"),
    ("cn_l_ks_f64.o", "void k_keyswitch_split14<ArF64T<1>, false>", 256, "N = 16384 as two 8192-point halves: 512 threads"),
    ("cn_l_gemm.o", "void k_scalar_gemm_mfma<2, false>", 256, "two 256-thread workgroups per CU (two waves per SIMD); 3 or 4 per CU spill"),
]


@pytest.fixture(scope="module")
def built():
    from cryptonets_amd import _native
    _native.build()
    import kernel_resources
    cache = {}

    def get(obj):
        if obj not in cache:
            cache[obj] = kernel_resources.resources(os.path.join(OBJ, obj))
        return cache[obj]
    return get


@pytest.mark.parametrize("obj,kernel,budget,why", BUDGETS)
def test_hot_kernel_stays_inside_its_register_budget(built, obj, kernel, budget, why):
    res = built(obj)
    assert kernel in res, "%s not found in %s (have e.g. %s)" % (kernel, obj, sorted(res)[:3])
    r = res[kernel]
    assert r["vgpr_spill"] == 0 and r["scratch"] == 0, "%s spills (%s): %s" % (kernel, why, r)
    assert r["vgpr"] + r["agpr"] <= budget, "%s: %d registers, budget %d (%s)" % (kernel, r["vgpr"] + r["agpr"], budget, why)


@pytest.mark.parametrize("obj", ["cn_l_gemm.o", "cn_l_behz.o", "cn_l_rr_u64.o", "cn_l_rr_f64.o", "cn_l_rr_f64l.o", "cn_l_ks_f64.o", "cn_l_ks_f64l.o"])
def test_hot_kernels_use_global_not_flat_memory_instructions(obj):
    """addresses that come out of tables (deferred per-ciphertext calls) are cast to the global address space in the kernels: a generic
    pointer compiles to flat_load / flat_store, which count on lgkmcnt too and made the address-table GEMMs 30-60 % slower than their
    index-table twins"""
    from cryptonets_amd import _native
    _native.build()
    import kernel_resources
    flat = kernel_resources.flat_instructions(os.path.join(OBJ, obj))
    assert not flat, flat
