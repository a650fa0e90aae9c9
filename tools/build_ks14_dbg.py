"""Timing-experiment builds of the N = 16384 key switch: libcnhip_dbg<mask>.so = the default objects with cn_l_ks_f64.hip recompiled
under -DKS14_DBG=<mask> (cn_k_ks.hip.h).  Results of these libraries are wrong by design; tools/gpu_r05_b.sh times them.

    python tools/build_ks14_dbg.py 7 15 16 32 48 63
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cryptonets_amd import _native
_native.build()
objs = [os.path.join(_native.OBJ_DIR, os.path.splitext(os.path.basename(s))[0] + ".o") for s in _native.SOURCES]
src = [s for s in _native.SOURCES if s.endswith("cn_l_ks_f64.hip")][0]
def one(mask):
    o = os.path.join(_native.OBJ_DIR, "cn_l_ks_f64_dbg%s.o" % mask)
    extra = [d for d in sys.argv[1:] if d.startswith("-D")]
    st = os.environ.get("KS14_SCHED")          # A/B of the compiler's scheduling strategy for this translation unit: "" (default), max-ilp, max-memory-clause, iterative-...
    unit = _native._unit_flags(src) if st is None else (["-mllvm", "-amdgpu-sched-strategy=" + st] if st else [])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", *unit,
                           "-DKS14_DBG=%s" % mask, *extra, "-c", src, "-o", o])
    lib = os.path.join(os.path.dirname(_native.LIB_PATH), "libcnhip_dbg%s.so" % mask)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "--offload-arch=gfx950", *[o if x.endswith("cn_l_ks_f64.o") else x for x in objs], "-o", lib])
    return lib
with ThreadPoolExecutor(max_workers=6) as ex:
    for lib in ex.map(one, [a for a in sys.argv[1:] if not a.startswith("-D")]):
        print(lib)
