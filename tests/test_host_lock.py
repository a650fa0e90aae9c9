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
