"""SURVEY 8(e) pieces that one GPU can execute: `cn_ctx_broadcast_keys` - the single-process key replication of a
multi-threaded host (the C# twin's `GpuSealBfvFactory.ReplicateTo`) - through its device-copy path AND, with CN_BCAST_FORCE_RCCL=1, through
librccl (ncclCommInitAll + grouped ncclBroadcast on a one-device communicator); and the one-process-per-GPU path of bench.py with the process
group forced at world 1 (BENCH_FORCE_DIST=1: RCCL broadcast of the relinearisation key, the key ADOPTED from the broadcast buffer).  Every
replica must evaluate to the oracle's words."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PARAMS, get_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(name, **opts):
    from cryptonets_amd._native import Context
    p = PARAMS[name]
    g = Context(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], device=0)
    for k, v in opts.items():
        g.set_option(k, v)
    return g


def _check_replica(g, o, rng):
    """mul_relin + rotate_rows + rotate_columns on the replica against the oracle, word for word"""
    cts = np.stack([o.encrypt(o.encode(rng.integers(0, 40, size=o.n, dtype=np.uint64))) for _ in range(3)])
    h, out = g.ct_alloc(3), g.ct_alloc(3)
    g.ct_upload(h, 0, cts)
    g.mul_relin(h, 0, h, 0, out, 0, 3)
    assert np.array_equal(g.ct_download(out, 0, 3), o.mul_relin_batch(cts, cts))
    g.rotate_rows(h, 0, 5, out, 0, 3)                    # NAF: 4 + 1, two Galois keys
    got = g.ct_download(out, 0, 3)
    for c in range(3):
        assert np.array_equal(got[c], o.rotate_rows(cts[c], 5))
    g.rotate_columns(h, 0, out, 0, 1)
    assert np.array_equal(g.ct_download(out, 0, 1)[0], o.rotate_columns(cts[0]))
    g.free(h), g.free(out)


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [0, 1])
@pytest.mark.parametrize("name", ["tiny", "c2"])
def test_broadcast_keys_to_contexts_on_one_device(name, force_rccl, rng):
    from cryptonets_amd import _native
    o = get_oracle(name, galois=True)
    root = _ctx(name)
    root.set_relin_key(o.relin_key())
    for i, e in enumerate(o.galois_elts()):
        root.set_galois_key(e, o.galois_key(i))
    same = _ctx(name)                                    # keeps its keys as the root does (FP64 images)
    other = _ctx(name, f64=0)                            # integer key-switch kernels: the FP64 image must be converted on arrival
    old = os.environ.get("CN_BCAST_FORCE_RCCL")
    os.environ["CN_BCAST_FORCE_RCCL"] = str(force_rccl)
    try:
        _native.broadcast_keys([root, same, other])
    finally:
        if old is None:
            os.environ.pop("CN_BCAST_FORCE_RCCL")
        else:
            os.environ["CN_BCAST_FORCE_RCCL"] = old
    for g in (same, other):
        assert g.has_galois_key(o.galois_elts()[0])
        assert np.array_equal(g.get_key(0), o.relin_key())            # exported as u64 residues whatever the resident form
        _check_replica(g, o, rng)
    _check_replica(root, o, rng)                          # the root's own keys are untouched
    for g in (root, same, other):
        g.close()


@pytest.mark.gpu
def test_broadcast_keys_carry_the_key_switch_convention(rng):
    """the keys of a context are of ONE decomposition convention (cn_set_option("ks_xi"), settled by the client's start-up self-test).  A replica
    that adopts the root's keys through cn_ctx_broadcast_keys adopts the convention with them - before, it kept its own default and every Relinearize /
    Rotate on it returned rc 0 and garbage."""
    from cryptonets_amd import _native
    from oracle.cno import Oracle
    p = PARAMS["tiny"]
    o = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], ks_xi=True)
    o.keygen(41, galois=True)
    root, replica = _ctx("tiny"), _ctx("tiny")
    root.set_option("ks_xi", 1)
    root.set_relin_key(o.relin_key())
    for i, e in enumerate(o.galois_elts()):
        root.set_galois_key(e, o.galois_key(i))
    assert replica.get_option("ks_xi") == 0
    _native.broadcast_keys([root, replica])
    assert replica.get_option("ks_xi") == 1
    _check_replica(replica, o, rng)
    _check_replica(root, o, rng)
    root.close(), replica.close()


@pytest.mark.gpu
def test_broadcast_drops_a_replicas_keys_of_the_other_convention(rng):
    """a replica that adopts the root's key-switch convention must not keep keys it generated / received under the other one - a Galois key the
    broadcast does not overwrite would keep returning rc 0 and garbage.  The root holds the relinearisation key and ONE Galois key (convention 1); the replica
    holds a full key set of convention 0: after the broadcast the replica rotates with the root's element and refuses every other one (no key)."""
    from cryptonets_amd import _native
    from oracle.cno import Oracle
    p = PARAMS["tiny"]
    o1 = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"], ks_xi=True)
    o1.keygen(41, galois=True)
    o0 = Oracle(p["n"], p["t"], q=p["q"], dbc=p["dbc"], gdbc=p["gdbc"])
    o0.keygen(43, galois=True)
    root, replica = _ctx("tiny"), _ctx("tiny")
    root.set_option("ks_xi", 1)
    root.set_relin_key(o1.relin_key())
    elts = list(o1.galois_elts())
    root.set_galois_key(elts[1], o1.galois_key(1))                 # one element only
    replica.set_relin_key(o0.relin_key())
    for i, e in enumerate(o0.galois_elts()):
        replica.set_galois_key(e, o0.galois_key(i))
    assert replica.has_galois_key(elts[2])
    _native.broadcast_keys([root, replica])
    assert replica.get_option("ks_xi") == 1
    assert replica.has_galois_key(elts[1]) and not replica.has_galois_key(elts[2]) and not replica.has_galois_key(elts[0])
    ct = o1.encrypt(o1.encode(rng.integers(0, 50, size=o1.n, dtype=np.uint64)))
    h, out = replica.ct_alloc(1), replica.ct_alloc(1)
    replica.ct_upload(h, 0, ct[None, :])
    replica.apply_galois(h, 0, elts[1], out, 0, 1)
    assert np.array_equal(replica.ct_download(out, 0, 1)[0], o1.apply_galois(ct, elts[1]))
    with pytest.raises(_native.CnError):
        replica.apply_galois(h, 0, elts[2], out, 0, 1)
    root.close(), replica.close()


@pytest.mark.gpu
def test_broadcast_keys_argument_errors():
    from cryptonets_amd import _native
    a, b = _ctx("tiny"), _ctx("default4096")
    with pytest.raises(_native.CnError):
        _native.broadcast_keys([a, b])                    # other encryption parameters
    c = _ctx("tiny")
    with pytest.raises(_native.CnError):
        _native.broadcast_keys([a, c])                    # the root has no evaluation keys
    for g in (a, b, c):
        g.close()


def _bench(extra_env, *args):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-unchanged-caller", "--no-single-image", *args],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_with_the_process_group_forced_equals_the_plain_run():
    """bench.py at world 1 with the RCCL process group created anyway: the relinearisation key travels through dist.broadcast and is adopted
    from the broadcast buffer (cn_set_relin_key(..., is_device_ptr=1)).  Same seeds -> the final ciphertext words must be the plain run's."""
    plain = _bench({})
    forced = _bench({"BENCH_FORCE_DIST": "1", "MASTER_PORT": "29547"})
    for line in (plain, forced):
        assert line["n_gpus"] == 1 and line["verified_against_integer_model"] is True and line["verified_slots"] == 2 * 8192 * 10
        assert line["roofline"]["frac"] > 0.2 and line["metric"].startswith("encrypted images/sec")
    assert plain["logit_words_sha256"] == forced["logit_words_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["lola", "cifar"])
def test_single_image_bench_with_one_clients_keys_broadcast(workload):
    """BASELINE configs 4 / 5 as north_star words them: ONE client's keys on every rank.  --shared-keys: rank 0 runs KeyGenerator, the relinearisation
    key, all 24 / 26 distinct Galois keys and the client keys of every plaintext prime travel through dist.broadcast (RCCL; world 1 with the process group
    forced) and are adopted in place from the broadcast buffers - the full-size key set (C5: ~0.9 GB) moves through the only collective of the
    project.  Same seeds -> the result ciphertext words must be those of the run where the rank generated its own keys."""
    own = _bench({}, "--workload", workload, "--client-seed", "1234")
    shared = _bench({"BENCH_FORCE_DIST": "1", "MASTER_PORT": "29549"}, "--workload", workload, "--client-seed", "1234", "--shared-keys")
    for line in (own, shared):
        assert line["n_gpus"] == 1 and line["verified_against_integer_model"] is True
    assert shared["process_group"] == "nccl" and own["process_group"] is None
    kb = shared["key_broadcast"]
    n, k, primes = (16384, 8, 2) if workload == "cifar" else (8192, 5, 4)
    digits = k * (1 if workload == "cifar" else 3)                      # Galois dbc 60 -> one digit per 48-49-bit limb; 20 -> three per 43-44-bit limb
    galois_key_bytes = digits * 2 * k * n * 8
    assert kb["contexts"] == primes and kb["bytes"] >= primes * 2 * (n.bit_length() - 2) * galois_key_bytes
    assert own["key_broadcast"] is None
    assert own["result_words_sha256"] == shared["result_words_sha256"]


@pytest.mark.gpu
def test_bench_gpus_flag_launches_ranks_on_the_gpu_box():
    """`bench.py --gpus 1` under its own torchrun launcher (BENCH_SELF_LAUNCH=1 forces the re-exec at N = 1): the N > 1 command line, rendezvous
    and RCCL initialisation executed for real on the device that is here"""
    line = _bench({"BENCH_SELF_LAUNCH": "1"}, "--gpus", "1")
    assert line["n_gpus"] == 1 and line["verified_against_integer_model"] is True
    assert line["launcher"] == "self (torch.distributed.run)" and line["process_group"] == "nccl"


_QUEUE_SCRIPT = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %r)
from cryptonets_amd._native import Context
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
dummies = []
for _ in range(int(sys.argv[1])):                        # an application's own streams, created first
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    dummies.append(s)
ctxs = [Context(4096, 65537, dbc=10, gdbc=20, device=0) for _ in range(4)]
tries = [g.get_option("stream_tries") for g in ctxs]
rng = np.random.default_rng(5)
for g in ctxs:
    g.keygen(99, galois=False)
    vals = rng.integers(0, g.t, size=(2, g.n), dtype=np.uint64)
    ph, ch, dh = g.pt_alloc(2), g.ct_alloc(2), g.pt_alloc(2)
    g.encode_batch(vals, ph, 0)
    g.encrypt(ph, 0, ch, 0, 2, seed=3)
    g.mul_relin(ch, 0, ch, 1, ch, 0, 1)
    g.decrypt(ch, 0, 1, dh, 0)
    assert np.array_equal(g.decode_batch(dh, 0, 1)[0], (vals[0].astype(object) * vals[1].astype(object) %% g.t).astype(np.uint64))
fifth = Context(4096, 65537, dbc=10, gdbc=20, device=0)
print("TRIES", " ".join(str(t) for t in tries), fifth.get_option("stream_tries"))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("dummies", [0, 3])
def test_contexts_of_one_process_get_a_hardware_queue_each(dummies):
    """HIP deals the streams of a process onto 4 hardware queues in creation order (null stream and transfer queue included); two contexts
    on one queue run their kernels one after the other (LoLa-MNIST: 10.7 instead of 8.0 ms per image, profiles/r03_stream_queues.txt).
    cn_ctx_create measures (two spin kernels overlap or do not) and keeps the first stream that overlaps with every live context's:
    whatever streams the process created before, the four plaintext-prime contexts of a LoLa factory report a successful choice and still
    evaluate correctly; a fifth context finds no free queue, says so (< 0) and works on the first stream it got.  (Own process: the cached
    contexts of this test session would occupy the queues.)"""
    r = subprocess.run([sys.executable, "-c", _QUEUE_SCRIPT % ROOT, str(dummies)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    tries = [int(x) for x in [l for l in r.stdout.splitlines() if l.startswith("TRIES")][0].split()[1:]]
    assert tries[0] == 1 and all(t != 0 for t in tries[:4]) and sum(t >= 1 for t in tries[:4]) >= 3, tries       # (a timing measurement: one miss on a loaded host is tolerated)
    assert tries[4] != 0, tries                          # (< 0 on every box seen: four queues, four contexts - not asserted, it is a measurement)
