"""N>1 path on the CPU: world_size-2 gloo processes shard the pairs, solve their shard (test double answering through
the oracle) and all-gather the (u,v) fields; every rank must end with the single-process result."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import oracle as O
    from tests.test_pipeline_cpu import OracleModel
    from tee_optical_flow_amd import distributed as D
    from tee_optical_flow_amd.synth import speckle_pairs, speckle_sequence
    O.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = OracleModel(O)
    if mode == "pairs":
        I0s, I1s = speckle_pairs(range(5), 40, 48)          # 5 pairs over 2 ranks: ragged tail (3 + 2)
        out = D.sharded_pairs_flow(I0s, I1s, eng, rank, world)
    else:
        fr = speckle_sequence(3, 6, 40, 48)                 # 5 pairs, 1-frame halo between the shards
        out = D.sharded_sequence_flow(fr, eng, rank, world, scale=2.0)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["pairs", "seq"])
def test_two_rank_gloo_shard_and_allgather(oracle, mode):
    import torch.multiprocessing as mp
    from tee_optical_flow_amd.synth import speckle_pairs, speckle_sequence
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "pairs" else 1)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    if mode == "pairs":
        I0s, I1s = speckle_pairs(range(5), 40, 48)
        ref = np.stack([oracle.tvl1_calc(a, b) for a, b in zip(I0s, I1s)])
    else:
        fr = speckle_sequence(3, 6, 40, 48)
        ref = np.stack([oracle.tvl1_calc(fr[i], fr[i + 1]) for i in range(5)]) * np.float32(2.0)
    for r in range(2):
        assert res[r].shape == ref.shape
        assert np.array_equal(res[r], ref), f"rank {r}"


def test_shard_bounds():
    from tee_optical_flow_amd.distributed import shard_bounds
    b, s = shard_bounds(1024, 8)
    assert s == 128 and b[0] == (0, 128) and b[7] == (896, 1024)
    b, s = shard_bounds(5, 2)
    assert s == 3 and b == [(0, 3), (3, 5)]
    b, s = shard_bounds(2, 4)
    assert s == 1 and b == [(0, 1), (1, 2), (2, 2), (2, 2)]
