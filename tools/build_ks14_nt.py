"""A/B builds of k_keyswitch_pair14 with non-temporal hints on the closing step: libcnhip_nt<mask>.so = the default objects with cn_l_ks_f64.hip recompiled under
-DKS14_NT=<mask> (cn_k_ks.hip.h: 1 result stores, 2 addend / accumulator loads, 4 the parked half, 8 the next link's c1).  Same words (tools/gpu_r06_r.sh checks).

    python tools/build_ks14_nt.py 1 3 4 15
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
    o = os.path.join(_native.OBJ_DIR, "cn_l_ks_f64_nt%s.o" % mask)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", *_native._unit_flags(src),
                           "-DKS14_NT=%s" % mask, "-c", src, "-o", o])
    lib = os.path.join(os.path.dirname(_native.LIB_PATH), "libcnhip_nt%s.so" % mask)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "--offload-arch=gfx950", *[o if x.endswith("cn_l_ks_f64.o") else x for x in objs], "-o", lib])
    return lib
with ThreadPoolExecutor(max_workers=4) as ex:
    for lib in ex.map(one, sys.argv[1:]):
        print(lib)
