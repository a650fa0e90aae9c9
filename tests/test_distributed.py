"""world_size-2 gloo test (CPU) of the multi-GPU plumbing: batch sharding is a disjoint cover, the key broadcast delivers
bit-identical words to every rank, timing is reduced with MAX."""
import os
import tempfile

import numpy as np

from cryptonets_amd.distributed import shard_batches


def _worker(rank, world, path, out):
    import torch
    import torch.distributed as dist
    from cryptonets_amd.distributed import broadcast_words, max_over_ranks, shard_batches
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    words = np.random.default_rng(99).integers(0, 2 ** 63, size=4096, dtype=np.uint64) if rank == 0 else None
    t = broadcast_words(words, 4096, 0, "cpu", dist)
    got = t.numpy().view(np.uint64)
    mine = shard_batches(7, rank, world)
    dt = max_over_ranks(1.0 + rank, "cpu", dist)
    np.savez(out % rank, words=got, mine=np.array(mine), dt=dt)
    dist.barrier()
    dist.destroy_process_group()


def test_shards_cover_disjoint():
    for world in (1, 2, 4, 8):
        seen = sorted(b for r in range(world) for b in shard_batches(13, r, world))
        assert seen == list(range(13))


def test_gloo_world2_broadcast_and_timing():
    import torch.multiprocessing as mp
    d = tempfile.mkdtemp()
    path, out = os.path.join(d, "store"), os.path.join(d, "rank%d.npz")
    mp.spawn(_worker, args=(2, path, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    exp = np.random.default_rng(99).integers(0, 2 ** 63, size=4096, dtype=np.uint64)
    assert np.array_equal(r0["words"], exp) and np.array_equal(r1["words"], exp)
    assert sorted(list(r0["mine"]) + list(r1["mine"])) == list(range(7))
    assert float(r0["dt"]) == 2.0 and float(r1["dt"]) == 2.0


def _prime_worker(rank, world, path, out):
    """one inference split by plaintext prime: every rank builds the factory for ITS primes only, evaluates, decrypts its
    residues; the CRT join over ranks recovers integers far above any single prime"""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_backend import make_factory
    from cryptonets_amd.distributed import crt_join_over_ranks, shard_primes
    from cryptonets_amd.hewrapper import EVectorFormat
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    primes = [40961, 65537, 114689, 147457]
    mine = shard_primes(primes, rank, world)
    F = make_factory("cpu", primes=mine, n=4096, galois=True)
    env = F.AllocateComputationEnv()
    v = np.array([1000, -2000, 3000, 4000], dtype=float)
    x = F.GetEncryptedVector(v, EVectorFormat.dense, 1.0)
    w = F.GetPlainVector(np.array([900, 800, -700, 600], dtype=float), EVectorFormat.dense, 1.0)
    y = x.PointwiseMultiply(x, env).DotProduct(w, env, length=4)              # sum(v^2 * w) = 6.7e9 > every prime
    res = {p: np.asarray(a._decrypt_ints(e), dtype=object)[:4] for p, a, e in zip(mine, y.eVectors, env.Environments)}
    joined = crt_join_over_ranks(res, primes, dist)
    np.savez(out % rank, joined=np.array([int(j) for j in joined], dtype=np.int64), mine=np.array(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_one_inference_split_by_plaintext_prime():
    import torch.multiprocessing as mp
    d = tempfile.mkdtemp()
    path, out = os.path.join(d, "store"), os.path.join(d, "rank%d.npz")
    mp.spawn(_prime_worker, args=(2, path, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert list(r0["mine"]) == [40961, 114689] and list(r1["mine"]) == [65537, 147457]
    v, w = np.array([1000, -2000, 3000, 4000]), np.array([900, 800, -700, 600])
    exp = int(np.sum(v * v * w))
    assert int(r0["joined"][3]) == exp and int(r1["joined"][3]) == exp


class _FakeCtx:
    """the slice of the libcnhip Context protocol BroadcastKeys uses, on host memory: keys are word arrays, an adopted "device" buffer is read
    back through its address (what cn_set_*_key(..., is_device_ptr = 1) does with a real device pointer)"""

    def __init__(self, n, k, digits, root):
        self.n, self.k, self.ctw, self.digits = n, k, 2 * k * n, digits
        self.keys, self.adopted = {}, {}
        self.options = {"ks_xi": 1 if root else 0}          # the root's client settled on the other key-switch convention: it must travel with the keys
        if root:
            from cryptonets_amd.distributed import default_galois_elements
            rng = np.random.default_rng(4)
            self.keys[(0, 0)] = rng.integers(0, 2 ** 62, size=self.key_words(False), dtype=np.uint64)
            for e in default_galois_elements(n):
                self.keys[(1, e)] = rng.integers(0, 2 ** 62, size=self.key_words(True), dtype=np.uint64)
            self.keys[(2, 0)] = rng.integers(0, 2 ** 62, size=self.ctw, dtype=np.uint64)
            self.keys[(3, 0)] = rng.integers(0, 2 ** 62, size=self.ctw // 2, dtype=np.uint64)

    def key_words(self, galois=False):
        return self.digits * self.ctw

    def get_option(self, name):
        return self.options[name]

    def set_option(self, name, value):
        self.options[name] = value

    def get_key(self, which, elt=0):
        return self.keys[(which, elt)]

    def _read(self, ptr, words):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint64 * words).from_address(ptr)).copy()

    def set_relin_key_device(self, ptr, words):
        self.adopted[(0, 0)] = self._read(ptr, words)

    def set_galois_key_device(self, elt, ptr, words):
        self.adopted[(1, elt)] = self._read(ptr, words)

    def set_public_key(self, w):
        self.adopted[(2, 0)] = np.array(w, copy=True)

    def set_secret_key(self, w):
        self.adopted[(3, 0)] = np.array(w, copy=True)


def _key_worker(rank, world, path, out):
    import torch.distributed as dist
    from cryptonets_amd.distributed import BroadcastKeys
    dist.init_process_group("gloo", init_method="file://" + path, rank=rank, world_size=world)
    ctx = _FakeCtx(64, 3, 9, root=(rank == 0))
    bk = BroadcastKeys(ctx, 0, "cpu", dist, with_galois=True, with_client_keys=True)
    server = _FakeCtx(64, 3, 9, root=(rank == 0))          # the library default: an evaluation server - evaluation keys only
    bs = BroadcastKeys(server, 0, "cpu", dist, with_galois=True)
    np.savez(out % rank, bytes=bk.bytes, n_adopted=len(ctx.adopted), ks_xi=ctx.options["ks_xi"], server_bytes=bs.bytes, server_ks_xi=server.options["ks_xi"],
             server_has_client_keys=int((2, 0) in server.adopted or (3, 0) in server.adopted), server_adopted=len(server.adopted),
             **{"k_%d_%d" % key: v for key, v in ctx.adopted.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_one_clients_keys_reach_every_rank():
    """bench.py --shared-keys / client.SharedKeyDeviceClient: rank 0's relinearisation key, every default Galois key (2 (log2 N - 1) distinct elements) and
    the client keys arrive bit-identical on rank 1, the evaluation keys through the adopt-a-device-buffer calls; rank 0 (world > 1 means a process
    group exists) adopts its own broadcast buffers too"""
    import torch.multiprocessing as mp
    from cryptonets_amd.distributed import default_galois_elements
    d = tempfile.mkdtemp()
    path, out = os.path.join(d, "store"), os.path.join(d, "rank%d.npz")
    mp.spawn(_key_worker, args=(2, path, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    elts = default_galois_elements(64)
    assert len(elts) == 2 * 5 and elts[0] == 127 and elts[1] == 3 and (elts[1] * elts[2]) % 128 == 1 and len(set(elts)) == len(elts)
    want = _FakeCtx(64, 3, 9, root=True).keys
    assert int(r1["n_adopted"]) == len(elts) + 3 and int(r0["n_adopted"]) == len(elts) + 1       # rank 0 keeps its own client keys
    for (which, elt), words in want.items():
        assert np.array_equal(r1["k_%d_%d" % (which, elt)], words)
        if which < 2:
            assert np.array_equal(r0["k_%d_%d" % (which, elt)], words)
    assert int(r0["bytes"]) == int(r1["bytes"]) == sum(w.size for w in want.values()) * 8
    # the key-switch convention of the root's client arrives with its keys (a replica with the other one returns rc 0 and garbage)
    assert int(r0["ks_xi"]) == 1 and int(r1["ks_xi"]) == 1 and int(r1["server_ks_xi"]) == 1
    # the default exchange carries NO client keys: rank 1 holds neither the public nor the secret key unless the caller asked for them
    assert int(r1["server_has_client_keys"]) == 0 and int(r1["server_adopted"]) == len(elts) + 1
    assert int(r1["server_bytes"]) == sum(w.size for (which, _), w in want.items() if which < 2) * 8
