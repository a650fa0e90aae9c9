"""A/B builds of the scalar-GEMM translation unit: libcnhip_gemm<tag>.so = the default objects with cn_l_gemm.hip recompiled under the
given -D switches (cn_k_gemm.hip.h).  Same results as the default library; CNHIP_LIB=<path> selects one (tools/gemm_probe.py).

    python tools/build_gemm_variants.py d1=-DGEMM_MFMA_DEPTH=1 d3=-DGEMM_MFMA_DEPTH=3
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cryptonets_amd import _native
_native.build()
objs = [os.path.join(_native.OBJ_DIR, os.path.splitext(os.path.basename(s))[0] + ".o") for s in _native.SOURCES]
src = [s for s in _native.SOURCES if s.endswith("cn_l_gemm.hip")][0]
def one(spec):
    tag, defs = spec.split("=", 1)
    o = os.path.join(_native.OBJ_DIR, "cn_l_gemm_%s.o" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", *_native._unit_flags(src),
                           *defs.split(","), "-c", src, "-o", o])
    lib = os.path.join(os.path.dirname(_native.LIB_PATH), "libcnhip_gemm%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "--offload-arch=gfx950", *[o if x.endswith("cn_l_gemm.o") else x for x in objs], "-o", lib])
    return lib
with ThreadPoolExecutor(max_workers=4) as ex:
    for lib in ex.map(one, sys.argv[1:]):
        print(lib)
