"""`python bench.py --gpus N` must start N ranks BY ITSELF when no launcher is around it (the flag was parsed and
ignored), honour an external torchrun, and print exactly ONE JSON line with n_gpus = N.  Run here with `--stub` (gloo, no device work):
the launcher, the rendezvous on 127.0.0.1, the key broadcast helper, the barriers and the MAX-over-ranks timing are the real code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_FORCE_DIST"):
        env.pop(k, None)
    return env


def _one_line(out):
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub"], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["plumbing_ok"] is True and line["stub"] is True
    # rank 1 sleeps 20 ms per step, rank 0 10 ms: the line carries the MAX over ranks
    assert line["ms_per_step"] >= 19.0, line
    assert line["scaling"] == "weak" and line["higher_is_better"] is True


def test_external_torchrun_is_honoured():
    port = subprocess.check_output([sys.executable, "-c", "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])"]).decode().strip()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--stub"]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 2 and line["plumbing_ok"] is True


def test_default_is_one_rank():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--stub"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert _one_line(r.stdout)["n_gpus"] == 1


def test_gpus_8_stub_covers_every_shard_once():
    """the driver's 8-GPU launch, on gloo: eight ranks rendezvous on 127.0.0.1, every batch of the job runs on exactly one rank, the line carries the MAX over
    ranks (rank 7 sleeps 80 ms per step) and n_gpus = 8; the side work of the N = 1 line (CPU baseline, single-image children, unchanged caller) runs on no rank"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "0", "--stub"], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = _one_line(r.stdout)
    assert line["n_gpus"] == 8 and line["plumbing_ok"] is True and line["shard_cover_ok"] is True
    assert line["ms_per_step"] >= 79.0, line
    assert not any(line["rank0_side_work"].values())
