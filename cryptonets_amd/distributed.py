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


def default_galois_elements(n):
    """the element set of KeyGenerator.GaloisKeys(dbc) / cn_keygen(with_galois): 2N - 1, then 3^(2^i) and 3^(-2^i) for i < log2(N) - 1.
    The last pair coincides (3^(N/4) has order 2 modulo 2N): 2 (log2 N - 1) distinct elements - 24 at N = 8192, 26 at N = 16384."""
    m = 2 * n
    elts, p3, ip3 = [m - 1], 3, pow(3, -1, m)
    for _ in range(n.bit_length() - 2):
        for e in (p3, ip3):
            if e not in elts:
                elts.append(e)
        p3, ip3 = p3 * p3 % m, ip3 * ip3 % m
    return elts


class BroadcastKeys:
    """ONE client's keys on every rank (north_star: "RCCL broadcast of evaluation/Galois keys over xGMI and no cross-GPU reduction"): rank
    `src` has generated them (KeyGenerator on its GPU or uploaded there); every key is broadcast as a device tensor - backend "nccl" = RCCL - and ADOPTED
    in place by the receiving context (cn_set_relin_key / cn_set_galois_key with is_device_ptr = 1: no host copy; the library converts the buffer to its
    FP64 key image in place).  The tensors must stay alive as long as the context uses them: this object owns them.  The key-switch convention the
    source settled on (cn_set_option("ks_xi"): the keys are of ONE convention) travels with the keys.  `with_client_keys` (default False: an evaluation
    server only ever receives public evaluation keys): the public and the SECRET key travel too - a benchmark rank that also plays the data owner
    (encrypts the inputs, decrypts the logits to verify them) asks for it explicitly.  bytes / seconds of the whole exchange are recorded."""

    def __init__(self, ctx, src, device, dist, with_galois=True, with_client_keys=False, adopt_on_src=None):
        import time
        import torch
        self.tensors, self.bytes = [], 0
        rank = dist.get_rank() if dist is not None else 0
        adopt = (rank != src) or (dist is not None if adopt_on_src is None else adopt_on_src)   # world 1 with a forced process group: the adoption path runs too
        if device is not None and str(device) != "cpu":
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        xi = broadcast_words(np.array([ctx.get_option("ks_xi")], dtype=np.uint64) if rank == src else None, 1, src, device, dist)
        xi = int(xi.cpu().numpy().view(np.uint64)[0])
        if xi != ctx.get_option("ks_xi"):
            ctx.set_option("ks_xi", xi)
        jobs = [(0, 0, ctx.key_words(False))]
        if with_galois:
            jobs += [(1, e, ctx.key_words(True)) for e in default_galois_elements(ctx.n)]
        for which, elt, words in jobs:
            t = broadcast_words(ctx.get_key(which, elt) if rank == src else None, words, src, device, dist)
            self.bytes += words * 8
            if adopt:
                if which == 0:
                    ctx.set_relin_key_device(t.data_ptr(), t.numel())
                else:
                    ctx.set_galois_key_device(elt, t.data_ptr(), t.numel())
                self.tensors.append(t)
        if with_client_keys:
            for which, words in ((2, ctx.ctw), (3, ctx.ctw // 2)):
                t = broadcast_words(ctx.get_key(which) if rank == src else None, words, src, device, dist)
                self.bytes += words * 8
                if rank != src:
                    w = t.cpu().numpy().view(np.uint64)
                    (ctx.set_public_key if which == 2 else ctx.set_secret_key)(w)
        if device is not None and str(device) != "cpu":
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        self.seconds = time.perf_counter() - t0
