"""The library's own RCCL exchange (include/teeflow.h, tf_comm_* / tf_allgather_flows) on the one GPU a test box has: a
1-rank communicator formed both ways (one process per GPU: unique id + init_rank; single process: init_all), the all-gather
ticket/wait protocol, and the result equal to the rank's own flows.  N>1 needs an 8-GPU node and stays unmeasured here; the
world_size-2 gloo tests (tests/test_distributed_cpu.py) pin the sharding logic the exchange is part of.

Runs in a child process with torch imported first: the engine and torch must share one HIP runtime, and in the pytest
process the engine's fixture has usually initialised HIP before torch is touched."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, ROOT)
import tee_optical_flow_amd as T
from tee_optical_flow_amd import _lib
from tee_optical_flow_amd.distributed import init_engine_comm, file_id_exchange, sharded_pairs_flow
from tee_optical_flow_amd.synth import speckle_pairs
dev = torch.device("cuda", 0)
I0s, I1s = speckle_pairs(range(4), 96, 128)
eng = T.DenseFlow(device_id=0, max_batch=4)
init_engine_comm(eng, 0, 1, file_id_exchange(TMP + "/id", 0))              # one process per GPU, world of 1
out = sharded_pairs_flow(I0s, I1s, eng, 0, 1, device=dev)                   # world 1: no exchange
ref = torch.from_numpy(np.array(eng.calc_pairs(I0s, I1s))).to(dev)
assert torch.equal(out, ref)
recv = [torch.zeros_like(ref) for _ in range(3)]
tickets = [eng.allgather(ref.data_ptr(), ref.numel(), r.data_ptr()) for r in recv]      # three gathers in flight
assert tickets == [0, 1, 2]
eng.comm_wait(tickets[1]); eng.comm_wait(-1)
assert all(torch.equal(r, ref) for r in recv)
try:
    eng.comm_wait(99)
    raise SystemExit("an unknown ticket must be refused")
except T.OpticalFlowCalculationError:
    pass
# a second engine produces the flows on its lanes (two sub-batches, a submitted job); the communicator handle never solves.  The order is
# the host's (include/teeflow.h): tf_wait has returned -> every lane has drained -> the all-gather may be issued at once
prod = T.DenseFlow(device_id=0, max_batch=2)
fr = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
pf = torch.zeros_like(ref); gat = torch.zeros_like(ref)
tk = prod.submit_pairs_device(fr.data_ptr(), fr.data_ptr() + 4 * 96 * 128, 4, 96, 128, pf.data_ptr())
prod.wait(tk)
eng.comm_wait(eng.allgather(pf.data_ptr(), pf.numel(), gat.data_ptr()))
assert prod.counter("queue_units_done") == 2 and torch.equal(pf, ref) and torch.equal(gat, ref)
prod.close()
# single-process form (ncclCommInitAll + grouped calls) on one device
eng2 = T.DenseFlow(device_id=0, max_batch=4)
L = _lib.load()
hs = (C.c_void_p * 1)(eng2._h)
_lib.check(L.tf_comm_init_all(hs, 1), eng2._h, "tf_comm_init_all")
r2 = torch.zeros_like(ref)
snd = (C.c_void_p * 1)(ref.data_ptr()); rcv = (C.c_void_p * 1)(r2.data_ptr())
_lib.check(L.tf_allgather_flows_all(hs, 1, snd, ref.numel(), rcv), eng2._h, "tf_allgather_flows_all")
assert torch.equal(r2, ref)
eng.close(); eng2.close()
print("comm ok")
"""


def test_library_rccl_allgather_one_rank(tmp_path):
    script = SCRIPT.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "comm ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("extra", [[], ["--steps", "8", "--warmup", "2", "--in-flight", "3"], ["--steps", "3", "--batch", "4"]],
                         ids=["default", "8-steps-3-in-flight", "one-sub-batch-per-step"])
def test_bench_gpus_2_starts_its_own_ranks(extra):
    """`python bench.py --gpus 2` from a plain shell (no torchrun, WORLD_SIZE unset) must start its ranks
    itself and print ONE line for the job.  Rehearsed on the one GPU a test box has: gloo rendezvous, both ranks on cuda:0,
    the all-gather through torch (RCCL refuses two ranks on one device) -- the launch path, the sharding, the overlap with
    the next step and the checksum of every gathered shard are the ones an 8-GPU run takes."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--batch", "10", "--sub-batch", "4", "--size", "96", "--steps", "2", "--warmup", "1", "--no-deepflow", "--no-cpu-baseline",
                        "--no-profile", "--steps-only"] + extra, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    nsteps = int(extra[extra.index("--steps") + 1]) if "--steps" in extra else 2
    assert d["n_gpus"] == 2 and d["steps"] == nsteps and d["scaling"] == "weak" and d["value"] > 0
    # every step is ONE call of the boundary (10 pairs = three sub-batches on the library's lanes, or one sub-batch); with --in-flight 3 the
    # steps are submitted without waiting and the collectives are still issued in step order, each after its step has been collected
    cur = os.environ.get("PYTEST_CURRENT_TEST", "")
    assert d["config"]["steps_in_flight"] == (3 if "3-in-flight" in cur else 1) and d["config"]["calls_per_step"] == 1
    assert d["config"]["sub_batches_per_step"] == (1 if "one-sub-batch" in cur else 3)
    assert d["config"]["library_queue_lanes"] == (0 if "one-sub-batch" in cur else 3)
    assert "all_gather" in d["collective"] and d["allgather_checksums_match"] is True
    assert d["config"]["parallelism"] == "pair-sharded x2"
