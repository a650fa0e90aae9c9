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
