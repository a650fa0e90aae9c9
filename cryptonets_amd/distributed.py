"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" on CPU for tests).

The path shards with NO data-path collective: independent ciphertext batches are dealt round-robin to ranks.
The only collective is one broadcast of the (public) evaluation keys at start-up; timing uses max-over-ranks.
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
