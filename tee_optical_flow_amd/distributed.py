"""Multi-GPU form of the hot path: frame pairs are independent, so the pair index range is cut
into `world` contiguous shards (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI), each rank
solves its shard with no data-path exchange, and ONE all-gather of the (u,v) fields assembles the result everywhere.
The reference has no counterpart (its loop is sequential, calculate_optical_flow.py:584-597).

Sequence mode shards with a one-frame halo: rank r needs frames [lo, hi] to produce pairs [lo, hi).
"""
import numpy as np


def shard_bounds(n_items, world):
    """Contiguous shards of ceil(n/world) items (the tail shard may be short or empty)."""
    s = -(-n_items // world)
    return [(min(r * s, n_items), min((r + 1) * s, n_items)) for r in range(world)], s


def _all_gather(local, world, group):
    import torch.distributed as dist
    import torch
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def sharded_sequence_flow(frames_u8, engine, rank, world, scale=1.0, group=None, device=None):
    """frames uint8 [N,H,W] (every rank holds them or at least its shard + halo) -> float32 torch tensor [N-1,H,W,2],
    identical on every rank.  `device` = torch device for the gathered tensor (cuda:<local_rank> for RCCL, cpu for gloo)."""
    import torch
    frames_u8 = np.ascontiguousarray(frames_u8)
    n_pairs = frames_u8.shape[0] - 1
    H, W = frames_u8.shape[1:]
    bounds, s = shard_bounds(n_pairs, world)
    lo, hi = bounds[rank]
    dev = torch.device(device) if device is not None else torch.device("cpu")
    local = torch.zeros((s, H, W, 2), dtype=torch.float32, device=dev)       # padded to the common shard size
    if hi > lo:
        if dev.type == "cuda":
            fr = torch.from_numpy(frames_u8[lo:hi + 1]).to(dev)               # 1-frame halo
            engine.calc_seq_device(fr.data_ptr(), hi - lo + 1, H, W, local.data_ptr(), scale=scale)
        else:
            local[:hi - lo] = torch.from_numpy(engine.calc_batch(frames_u8[lo:hi + 1], scale=scale))
    if world == 1:
        return local[:n_pairs]
    g = _all_gather(local, world, group)
    return torch.cat([g[r * s:r * s + (b[1] - b[0])] for r, b in enumerate(bounds)])


def sharded_pairs_flow(I0s, I1s, engine, rank, world, group=None, device=None):
    """B independent pairs uint8 [B,H,W] x2 -> float32 torch tensor [B,H,W,2] on every rank (BASELINE config 3)."""
    import torch
    I0s = np.ascontiguousarray(I0s)
    I1s = np.ascontiguousarray(I1s)
    B, H, W = I0s.shape
    bounds, s = shard_bounds(B, world)
    lo, hi = bounds[rank]
    dev = torch.device(device) if device is not None else torch.device("cpu")
    local = torch.zeros((s, H, W, 2), dtype=torch.float32, device=dev)
    if hi > lo:
        if dev.type == "cuda":
            fr = torch.from_numpy(np.concatenate([I0s[lo:hi], I1s[lo:hi]])).to(dev)
            n = hi - lo
            engine.calc_pairs_device(fr.data_ptr(), fr.data_ptr() + n * H * W, n, H, W, local.data_ptr())
        else:
            local[:hi - lo] = torch.from_numpy(engine.calc_pairs(I0s[lo:hi], I1s[lo:hi]))
    if world == 1:
        return local[:B]
    g = _all_gather(local, world, group)
    return torch.cat([g[r * s:r * s + (b[1] - b[0])] for r, b in enumerate(bounds)])
