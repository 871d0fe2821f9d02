"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU
tests).  The path shards by image pair with no data-path collective; the only
collective is one broadcast of the flat weight blob at start-up."""
import numpy as np


def shard_range(global_batch, rank, world):
    """rank r of `world` owns pairs [lo, hi); remainder pairs go to the lowest ranks."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob, nfloats, device, src=0):
    """blob: 1-D float32 numpy array on `src`, ignored elsewhere.  Returns a torch tensor on `device`
    holding the blob on every rank (one ncclBroadcast / RCCL over xGMI when device is a GPU)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device)
            if t.numel() != nfloats:
                raise ValueError("blob has %d floats, expected %d" % (t.numel(), nfloats))
        else:
            t = torch.empty(nfloats, dtype=torch.float32, device=device)
        dist.broadcast(t, src=src)
        return t
    return torch.from_numpy(np.ascontiguousarray(blob, np.float32)).to(device)


def max_over_ranks(value, device):
    import torch
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(value)
