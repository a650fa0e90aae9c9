"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" on CPU for tests).

The path shards with NO data-path collective: independent ciphertext batches are dealt round-robin to ranks
(throughput), or - for the latency of ONE inference - the independent plaintext-prime channels are (SURVEY 8e (2):
`EncryptedSealBfvVector.cs:225-236` fans every op out per prime, the channels only meet in the client's CRT join after
decryption, `:381-395`).  The only collectives are one broadcast of the (public) evaluation keys at start-up and, for the
prime split, one gather of the decrypted residues on the client; timing uses max-over-ranks.
"""
import numpy as np


def shard_batches(n_batches, rank, world):
    """GPU g processes batches {g, g+G, ...}."""
    return list(range(rank, n_batches, world))


def broadcast_words(words, count, src, device, dist):
    """Broadcast `count` u64 words from rank `src`; returns a torch int64 tensor on `device` holding the same bits
    on every rank (libcnhip adopts it with cn_set_*_key(..., is_device_ptr=1))."""
    import torch
    if dist is not None and dist.get_rank() != src or words is None:
        t = torch.empty(count, dtype=torch.int64, device=device)
    else:
        w = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
        assert w.size == count
        t = torch.from_numpy(w.view(np.int64)).to(device)
    if dist is not None:
        dist.broadcast(t, src=src)
    return t


def max_over_ranks(seconds, device, dist):
    import torch
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_primes(primes, rank, world):
    """plaintext-prime channels of one inference dealt to ranks: rank r owns primes[r], primes[r + world], ..."""
    return list(primes[rank::world])


def crt_join_over_ranks(residues, primes, dist, signed=True):
    """CRT join (EncryptedSealBfvVector.cs:381-395) of residues produced on different ranks.

    residues: dict {prime: integer array} for the primes THIS rank owns; primes: the full list in factory order.  Every rank
    returns the joined python-int list (values centred in (-M/2, M/2] when signed)."""
    parts = [residues]
    if dist is not None:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, {int(p): np.asarray(v).tolist() for p, v in residues.items()})
    merged = {}
    for part in parts:
        merged.update({int(p): v for p, v in part.items()})
    if sorted(merged) != sorted(int(p) for p in primes):
        raise ValueError("CRT join: residues for primes %s, expected %s" % (sorted(merged), sorted(primes)))
    M = 1
    for p in primes:
        M *= int(p)
    coef = [(M // int(p)) * pow((M // int(p)) % int(p), -1, int(p)) for p in primes]
    n = len(next(iter(merged.values())))
    out = []
    for i in range(n):
        v = sum(c * int(merged[int(p)][i]) for c, p in zip(coef, primes)) % M
        out.append(v - M if signed and 2 * v > M else v)
    return out
