"""Multi-GPU form of the hot path: frame pairs are independent, so the pair index range is cut
into `world` contiguous shards (one process per GPU), each rank solves its shard with no data-path exchange, and ONE
all-gather of the (u,v) fields assembles the result everywhere.  The reference has no counterpart (its loop is
sequential, calculate_optical_flow.py:584-597).

The all-gather itself is the library's: `tf_allgather_flows` (ncclAllGather on librccl over xGMI, include/teeflow.h).
This module only cuts the shards and carries the 128-byte communicator id from rank 0 to the others over whatever channel
the launcher has -- torch.distributed's process group when the job was started with torchrun (`torch_id_exchange`), a
shared file otherwise (`file_id_exchange`).  On CPU tensors (gloo rehearsals, the world_size-2 tests) the gather falls
back to torch.distributed.all_gather_into_tensor, the behaviour the library path must reproduce.

Sequence mode shards with a one-frame halo: rank r needs frames [lo, hi] to produce pairs [lo, hi).
"""
import os
import time

import numpy as np


def torch_id_exchange(group=None):
    """id channel over an initialised torch.distributed group: f(id_or_None) -> id on every rank."""
    def f(payload):
        import torch.distributed as dist
        box = [payload]
        dist.broadcast_object_list(box, src=0, group=group)
        return box[0]
    return f


def file_id_exchange(path, rank, timeout_s=120.0):
    """id channel over a file every rank can see (no torch at all): rank 0 writes it atomically, the others poll."""
    def f(payload):
        if rank == 0:
            tmp = f"{path}.{os.getpid()}"
            with open(tmp, "wb") as fh:
                fh.write(payload)
            os.replace(tmp, path)
            return payload
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"communicator id never appeared at {path}")
            time.sleep(0.01)
        with open(path, "rb") as fh:
            return fh.read()
    return f


def init_engine_comm(engine, rank, world, exchange):
    """Give `engine` (a DenseFlow on this rank's GPU) its rank of a `world`-rank RCCL communicator.  `exchange` is one of
    the id channels above."""
    ident = type(engine).comm_unique_id() if rank == 0 else None
    ident = exchange(ident)
    engine.comm_init_rank(world, rank, ident)


def shard_bounds(n_items, world):
    """Contiguous shards of ceil(n/world) items (the tail shard may be short or empty)."""
    s = -(-n_items // world)
    return [(min(r * s, n_items), min((r + 1) * s, n_items)) for r in range(world)], s


def _all_gather(local, world, group, engine=None):
    """[S,...] per rank -> [world*S,...] on every rank.  Device tensors + an engine that holds a communicator: the library's
    own ncclAllGather; otherwise torch.distributed (CPU rehearsals, tests)."""
    import torch.distributed as dist
    import torch
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if engine is not None and local.is_cuda and getattr(engine, "has_comm", False):
        # The library orders its communication stream behind the ENGINE's stream only.  `local` may also carry work of torch's
        # current stream (the zero fill of a short or empty shard, anything a caller computed with torch ops): finish that first.
        torch.cuda.current_stream(local.device).synchronize()
        engine.comm_wait(engine.allgather(local.data_ptr(), local.numel(), out.data_ptr()))
        return out
    dist.all_gather_into_tensor(out, local, group=group)
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
    g = _all_gather(local, world, group, engine)
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
    g = _all_gather(local, world, group, engine)
    return torch.cat([g[r * s:r * s + (b[1] - b[0])] for r, b in enumerate(bounds)])
