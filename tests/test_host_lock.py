"""The context lock of libcnhip under the reference's thread counts (Defaults.ThreadCount = Environment.ProcessorCount, HE Wrapper/Defaults.cs):
host-only stress test (tests/cpp/lock_stress.cpp compiled with the host pass of hipcc), CPU suite."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_context_lock_stress():
    exe = os.path.join(tempfile.mkdtemp(), "lock_stress")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "lock_stress.cpp"),
                           os.path.join(ROOT, "cryptonets_amd", "csrc", "cn_host.cpp"), "-pthread", "-o", exe], cwd=tempfile.gettempdir())
    for combine in ("0", "1"):                      # CnMutex::run through the plain lock (default) and through the combining path (kept switchable)
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, env=dict(os.environ, CN_LOCK_COMBINE=combine))
        assert r.returncode == 0, r.stdout.decode()
        lines = r.stdout.decode().splitlines()
        assert len(lines) == 15 and all("counter" in ln and " bad 0 " in ln for ln in lines)


def test_submission_ring_stress():
    """the lock-free submission structures ("defer" = 2, cryptonets_amd/csrc/cn_submit.h) under 1-256 producer threads: every record executed once, in claim
    order, hand-overs between threads respected, ready handles handed out once (tests/cpp/ring_stress.cpp)"""
    exe = os.path.join(tempfile.mkdtemp(), "ring_stress")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "ring_stress.cpp"),
                           os.path.join(ROOT, "cryptonets_amd", "csrc", "cn_host.cpp"), "-pthread", "-o", exe], cwd=tempfile.gettempdir())
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 6 and all(" bad 0" in ln for ln in lines), lines


def test_replay_pool_regions():
    """the thread pool of the C++ per-ciphertext caller (tools/replay_reference_calls.cpp, the stand-in for Utils.ParallelProcessInEnv): regions of changing
    width - every item once, no body alive after its region returned.  (Round 6: generation and width lived in two atomics; a pool thread could take part
    in one region twice and the region returned early - found by tools/soak_lockfree.py on the GPU box as an 'invalid handle' / a crash of the HARNESS.)"""
    import ctypes, sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay_reference_calls as rp
    L = ctypes.CDLL(rp.build())
    L.rp_pool_selftest.argtypes = [ctypes.c_int, ctypes.c_uint]
    assert L.rp_pool_selftest(5000, 11) == 0
