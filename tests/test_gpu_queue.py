"""The library's lane queue (calc_entry / tf_submit_* / tf_wait, include/teeflow.h "Sub-batches and lanes"): a call larger than one
sub-batch is cut into units that the handle's lanes take one at a time.  Whatever lane solves what, flows and executed iteration
counts are those of the contiguous-split form of rounds 1-4 (queue_lanes = 0) and of the oracle, in pair order; a failing
sub-batch drops the ones not yet started, lets the running ones finish and leaves nothing in flight; jobs submitted without
waiting overlap and can be collected in any order.  Reference loop: calculate_optical_flow.py:584-597 (one loop -> one call)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mixed(n, H, W, seed0=900):
    from tests.test_gpu_batches import _mixed_pairs
    return _mixed_pairs(n, H, W, seed0=seed0)


def _engine(cap, **kw):
    import tee_optical_flow_amd as T
    return T.DenseFlow(device_id=0, max_batch=cap, **kw)


@pytest.mark.parametrize("n,cap,unit", [(53, 16, 0), (48, 16, 0), (40, 16, 7), (17, 16, 0)])
def test_queue_gives_the_split_forms_flows_and_iteration_order(oracle, n, cap, unit):
    """3 x 16-pair sub-batches + a ragged one on three lanes (and a 7-pair unit size): np.array_equal flows, tf_get_iters in pair order."""
    I0s, I1s = _mixed(n, 72, 96)
    eng = _engine(cap)
    try:
        eng.set_tuning("queue_lanes", 0)                                  # rounds 1-4: contiguous parts, joined
        f0 = np.array(eng.calc_pairs(I0s, I1s)); it0 = eng.last_iters().copy()
        assert eng.counter("queue_jobs") == 0
        eng.set_tuning("queue_lanes", -1)
        eng.set_tuning("queue_unit", unit)
        f1 = np.array(eng.calc_pairs(I0s, I1s)); it1 = eng.last_iters().copy()
        # unit 0: equal units, the smallest multiple of the three lanes of them that fit a sub-batch (53 pairs, 16 per sub-batch: 6 x 9)
        u = unit or -(-n // (3 * -(-n // (3 * cap))))
        assert eng.counter("queue_jobs") == 1 and eng.counter("queue_lanes") == 3
        assert eng.counter("queue_units_done") == -(-n // u) and eng.counter("queue_outstanding") == 0
        assert eng.last_stats["n_pairs"] == n and eng.last_stats["inner_iters_total"] == int(it0[..., 0].sum())
        assert np.array_equal(it0, it1) and np.array_equal(f0, f1)
        spread = it0[..., 0].sum(axis=(1, 2))
        assert spread.max() > 1.5 * spread.min(), "the batch should mix fast and slow pairs"
        for b in (0, 3, 5, u - 1, u, n - 1):                              # first / last of a unit, a zero-flow pair, an unrelated pair
            ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], return_iters=True)
            assert np.array_equal(it1[b], ref_it[:nl]) and np.array_equal(f1[b], ref), f"pair {b}"
        for lanes in (1, 2, 5):                                           # any lane count, same bits
            eng.set_tuning("queue_lanes", lanes)
            f2 = np.array(eng.calc_pairs(I0s, I1s))
            assert eng.counter("queue_lanes") == lanes
            assert np.array_equal(f2, f0) and np.array_equal(eng.last_iters(), it0)
    finally:
        eng.close()


def test_queue_sequence_mode_units_share_their_boundary_frame(oracle):
    from tee_optical_flow_amd.synth import speckle_sequence
    frames = speckle_sequence(9, 50, 80, 112)
    eng = _engine(16)
    try:
        eng.set_tuning("queue_lanes", 0)
        f0 = np.array(eng.calc_batch(frames, scale=1.5)); it0 = eng.last_iters().copy()
        eng.set_tuning("queue_lanes", -1)
        f1 = np.array(eng.calc_batch(frames, scale=1.5))
        assert eng.counter("queue_units_done") == 6                        # 49 pairs, 16 per sub-batch, three lanes: 5 x 9 + 4
        assert np.array_equal(f0, f1) and np.array_equal(it0, eng.last_iters())
        for i in (0, 8, 9, 44, 48):
            assert np.array_equal(f1[i], oracle.tvl1_calc(frames[i], frames[i + 1]) * np.float32(1.5))
    finally:
        eng.close()


def test_a_failing_sub_batch_drains_every_lane_before_the_call_returns(oracle):
    import tee_optical_flow_amd as T
    I0s, I1s = _mixed(64, 72, 96, seed0=40)
    eng = _engine(16)
    try:
        eng.set_tuning("queue_unit", 16)                                   # four units of 16 on three lanes
        good = np.array(eng.calc_pairs(I0s, I1s)); it = eng.last_iters().copy()
        done0 = eng.counter("queue_units_done")
        eng.set_tuning("queue_test_fail_unit", 1)                          # the lane that takes unit 1 reports a failure instead of solving it
        with pytest.raises(T.OpticalFlowCalculationError, match=r"sub-batch 1 \(pairs 16\.\.31\).*injected"):
            eng.calc_pairs(I0s, I1s)
        # units 0 and 2 were running on the other lanes and finished; unit 3 was never started; nothing is queued or in flight
        assert eng.counter("queue_outstanding") == 0
        assert eng.counter("queue_units_done") - done0 + eng.counter("queue_units_skipped") == 3
        assert eng.counter("queue_units_skipped") >= 1 and eng.counter("queue_units_failed") == 1
        again = np.array(eng.calc_pairs(I0s, I1s))                         # the handle is whole: same bits as before
        assert np.array_equal(again, good) and np.array_equal(eng.last_iters(), it)
        eng.set_tuning("queue_test_fail_unit", 3)                          # the last unit fails: everything else has been solved
        with pytest.raises(T.OpticalFlowCalculationError, match="sub-batch 3"):
            eng.calc_pairs(I0s, I1s)
        assert eng.counter("queue_outstanding") == 0
        assert np.array_equal(np.array(eng.calc_pairs(I0s, I1s)), good)
    finally:
        eng.close()


def test_jobs_in_flight_are_collected_in_any_order(oracle):
    """tf_submit_pairs / tf_submit_seq: three jobs queued before the first wait; each wait hands back that job's flows, statistics
    and iteration counts; a synchronous call made meanwhile queues behind them."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_sequence
    sets = [_mixed(n, 64, 88, seed0=s) for n, s in ((20, 10), (9, 200), (33, 400))]
    frames = speckle_sequence(4, 12, 64, 88)
    eng = _engine(8)
    try:
        want = []
        for I0s, I1s in sets:
            f = np.array(eng.calc_pairs(I0s, I1s))
            want.append((f, eng.last_iters().copy()))
        want_seq = np.array(eng.calc_batch(frames, scale=3.0))
        t0 = eng.submit_pairs(*sets[0])
        t1 = eng.submit_pairs(*sets[1])
        ts = eng.submit_batch(frames, scale=3.0)
        t2 = eng.submit_pairs(*sets[2])
        sync = np.array(eng.calc_pairs(*sets[1]))                          # queues behind the four jobs
        assert np.array_equal(sync, want[1][0])
        for t, k in ((t2, 2), (t0, 0), (t1, 1)):
            got = eng.wait(t)
            assert eng.last_stats["n_pairs"] == len(sets[k][0])
            assert np.array_equal(got, want[k][0]) and np.array_equal(eng.last_iters(), want[k][1]), f"job {k}"
        assert np.array_equal(eng.wait(ts), want_seq)
        with pytest.raises(T.OpticalFlowCalculationError, match="unknown ticket"):
            eng.wait(t0)
        assert eng.counter("queue_outstanding") == 0
        ref, ref_it, nl = oracle.tvl1_calc(sets[2][0][32], sets[2][1][32], return_iters=True)
        assert np.array_equal(want[2][0][32], ref) and np.array_equal(want[2][1][32], ref_it[:nl])
    finally:
        eng.close()


DEVICE_JOBS = r"""
import sys
import numpy as np, torch
sys.path.insert(0, ROOT)
import tee_optical_flow_amd as T
from tests.test_gpu_batches import _mixed_pairs
I0s, I1s = _mixed_pairs(24, 64, 88, seed0=70)
dev = torch.device("cuda", 0)
fr = torch.from_numpy(np.concatenate([I0s, I1s])).to(dev)
p0, p1 = fr.data_ptr(), fr.data_ptr() + 24 * 64 * 88
out = [torch.zeros((24, 64, 88, 2), dtype=torch.float32, device=dev) for _ in range(4)]
eng = T.DenseFlow(device_id=0, max_batch=8)
try:
    eng.calc_pairs_device(p0, p1, 24, 64, 88, out[0].data_ptr())
    ref = out[0].cpu().numpy()
    assert np.array_equal(ref, np.array(eng.calc_pairs(I0s, I1s)))
    tk = [eng.submit_pairs_device(p0, p1, 24, 64, 88, out[k].data_ptr(), scale=float(k)) for k in (1, 2, 3)]
    for k, t in zip((1, 2, 3), tk):
        st = eng.wait(t)
        assert st["n_pairs"] == 24
        assert np.array_equal(out[k].cpu().numpy(), ref * np.float32(k))
    for k in (1, 2, 3):                                                # closing with jobs queued: the lanes finish them first
        eng.submit_pairs_device(p0, p1, 24, 64, 88, out[k].data_ptr())
finally:
    eng.close()
torch.cuda.synchronize()
for k in (1, 2, 3):
    assert np.array_equal(out[k].cpu().numpy(), ref)
print("device jobs ok")
"""


def test_device_jobs_in_flight_and_close_with_jobs_queued():
    """tf_submit_pairs_device on torch tensors (a child process with torch imported first: the engine and torch must share one HIP runtime,
    and in the pytest process the engine has usually initialised HIP before torch is touched)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", DEVICE_JOBS.replace("ROOT", repr(root))], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "device jobs ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_deepflow_through_the_queue(oracle):
    import tee_optical_flow_amd as T
    I0s, I1s = _mixed(20, 96, 160, seed0=500)
    eng = T.DenseFlow(device_id=0, max_batch=8, algo="deepflow")
    try:
        eng.set_tuning("queue_lanes", 0)
        f0 = np.array(eng.calc_pairs(I0s, I1s))
        eng.set_tuning("queue_lanes", -1)
        f1 = np.array(eng.calc_pairs(I0s, I1s))
        assert eng.counter("queue_lanes") == 1 and eng.counter("queue_units_done") == 3
        assert np.array_equal(f0, f1)
        eng.set_tuning("queue_lanes", 2)
        t = eng.submit_pairs(I0s, I1s)
        assert np.array_equal(eng.wait(t), f0)
        for b in (0, 8, 19):
            assert np.array_equal(f1[b], oracle.deepflow_calc(I0s[b], I1s[b]))
    finally:
        eng.close()


PROBE = r"""
import sys, json
import numpy as np
sys.path.insert(0, ROOT)
import tee_optical_flow_amd as T
from tests.test_gpu_batches import _mixed_pairs
I0s, I1s = _mixed_pairs(40, 64, 88, seed0=11)
eng = T.DenseFlow(device_id=0, max_batch=16)
f = np.array(eng.calc_pairs(I0s, I1s))                       # 3 units on 3 lanes
one = np.array(eng.calc_pairs(I0s[:16], I1s[:16]))           # one sub-batch: the handle and its twin
print(json.dumps({"retries": eng.counter("stream_retries"), "serialised": eng.counter("streams_serialised"), "lanes": eng.counter("queue_lanes"),
                  "same": bool(np.array_equal(f[:16], one)), "sum": float(np.abs(f).sum())}))
eng.close()
"""


@pytest.mark.parametrize("queues", ["2", "4", "8"])
def test_lanes_look_for_streams_that_run_beside_each_other(queues):
    """HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues and serialises streams that share one.  With 8 or 4 queues the three
    lanes end up on streams that run concurrently (the probe may have to drop a few); with 2 they cannot, and the library says so
    instead of assuming.  (The library leaves GPU_MAX_HW_QUEUES alone: with 8 the two-lane form of one-sub-batch calls measured 2276
    against 2673 pairs/s -- DESIGN.md section 5a.)  Flows do not depend on any of it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", PROBE.replace("ROOT", repr(root))], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "GPU_MAX_HW_QUEUES": queues})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    g = json.loads(r.stdout.strip().splitlines()[-1])
    assert g["lanes"] == 3 and g["same"]
    if queues == "2":
        assert g["serialised"] & 1 and g["retries"] >= 8
    else:
        assert g["serialised"] & 1 == 0
    if queues == "8":
        assert g["serialised"] == 0                                       # room for the lanes' copy streams too
    test_lanes_look_for_streams_that_run_beside_each_other.sums = getattr(test_lanes_look_for_streams_that_run_beside_each_other, "sums", set()) | {g["sum"]}
    assert len(test_lanes_look_for_streams_that_run_beside_each_other.sums) == 1


@pytest.mark.parametrize("algo,cap", [("TVL1", 5), ("deepflow", 4)])
def test_queue_stress_random_jobs_random_collection_order(algo, cap):
    """Forty jobs of random sizes (1 .. 3.x sub-batches), submitted in bursts with synchronous calls in between and collected in random
    order: every job's flows (and DualTVL1 iteration counts) equal those of the same pairs solved alone by a second engine."""
    import tee_optical_flow_amd as T
    rng = np.random.default_rng(20260105)
    H, W = (48, 72) if algo == "TVL1" else (64, 96)
    pool0, pool1 = _mixed(60, H, W, seed0=7000)
    eng = T.DenseFlow(device_id=0, max_batch=cap, algo=algo)
    ref = T.DenseFlow(device_id=0, max_batch=64, algo=algo)
    ref.set_tuning("queue_lanes", 0)
    try:
        want_all = np.array(ref.calc_pairs(pool0, pool1))
        it_all = ref.last_iters().copy() if algo == "TVL1" else None
        open_jobs = {}
        done = 0
        for k in range(40):
            n = int(rng.integers(1, 3 * cap + 3))
            a = int(rng.integers(0, 60 - n + 1))
            open_jobs[eng.submit_pairs(pool0[a:a + n], pool1[a:a + n])] = (a, n)
            if rng.random() < 0.25:                                        # a synchronous call queues behind what is in flight
                b, m = int(rng.integers(0, 50)), int(rng.integers(1, 2 * cap))
                assert np.array_equal(np.array(eng.calc_pairs(pool0[b:b + m], pool1[b:b + m])), want_all[b:b + m])
            while open_jobs and (len(open_jobs) > 6 or rng.random() < 0.3):
                t = list(open_jobs)[int(rng.integers(0, len(open_jobs)))]
                a, n = open_jobs.pop(t)
                got = eng.wait(t)
                assert np.array_equal(got, want_all[a:a + n]), f"job {t}: pairs {a}..{a + n - 1}"
                if it_all is not None:
                    assert np.array_equal(eng.last_iters(), it_all[a:a + n])
                done += 1
        for t, (a, n) in open_jobs.items():
            assert np.array_equal(eng.wait(t), want_all[a:a + n])
            done += 1
        assert done == 40 and eng.counter("queue_outstanding") == 0 and eng.counter("queue_units_failed") == 0
    finally:
        eng.close()
        ref.close()


def test_studies_through_the_queue_device_frames_host_flows(oracle):
    """tf_calc_seq_rgb / tf_submit_seq_rgb / tf_calc_seq_saliency_f32 on studies longer than a sub-batch: the frames are conditioned (or turned
    into saliency maps) on the device, the lanes read their units from that device buffer and copy their flows to the caller's host array."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.frames import condition_frames
    from tee_optical_flow_amd.synth import speckle_sequence
    studies = [np.ascontiguousarray(np.repeat(speckle_sequence(60 + k, n, 72, 104)[..., None], 3, axis=3)) for k, n in enumerate((30, 12, 41))]
    eng = T.DenseFlow(device_id=0, max_batch=8)
    try:
        want = []
        # the device's conditioning.  (The host twin takes the luma through numpy's matmul like skimage; where that BLAS fuses multiply and add a
        # frame's maximum can move by an ulp and with it a few hundred of the frame's rounded values by one count -- DESIGN.md section 9, row a1.)
        grays = [eng.condition_frames(rgb) for rgb in studies]
        assert np.mean([float((g != condition_frames(rgb)).mean()) for g, rgb in zip(grays, studies)]) < 5e-3
        for g in grays:
            eng.set_tuning("queue_lanes", 0)
            want.append(np.array(eng.calc_batch(g, scale=1.25)))
        eng.set_tuning("queue_lanes", -1)
        for rgb, w in zip(studies, want):
            got = np.array(eng.calc_study(rgb, scale=1.25, pad_last=True))
            assert got.shape[0] == rgb.shape[0] and np.array_equal(got[:-1], w) and np.array_equal(got[-1], got[-2])
        tickets = [eng.submit_study(rgb, scale=1.25, pad_last=(k == 1)) for k, rgb in enumerate(studies)]
        studies[0][...] = 0                                                # the frames were conditioned at submission: the caller's array is free
        for k in (2, 0, 1):
            got = np.array(eng.wait(tickets[k]))
            assert np.array_equal(got[:want[k].shape[0]], want[k]), f"study {k}"
            assert got.shape[0] == want[k].shape[0] + (1 if k == 1 else 0)
        assert np.array_equal(want[1][3], oracle.tvl1_calc(grays[1][3], grays[1][4]) * np.float32(1.25))
        sal = np.array(eng.calc_study_saliency(studies[2], scale=2.0))     # 40 pairs in sub-batches of 8, float maps
        maps = eng.saliency_frames(studies[2])
        assert np.array_equal(sal, np.array(eng.calc_pairs(maps[:-1], maps[1:])) * np.float32(2.0))
    finally:
        eng.close()


def _threads_now():
    with open("/proc/self/status") as f:
        return int(next(ln for ln in f if ln.startswith("Threads:")).split()[1])


def test_engines_with_lanes_give_back_their_device_memory_and_threads():
    """tf_destroy ends the lanes (host threads, streams, device buffers, jobs never waited for): handles made and closed in a loop
    leave the device's free memory and the process's thread count where the first round left them."""
    import ctypes
    from tee_optical_flow_amd.synth import speckle_sequence
    hip = ctypes.CDLL("libamdhip64.so")                                   # (already in the process: the library links it; torch stays out of this test)

    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value
    I0s, I1s = _mixed(20, 64, 80)
    rgb = np.repeat(speckle_sequence(5, 7, 64, 80)[..., None], 3, axis=-1)
    free, threads = [], []
    for rnd in range(7):
        for algo in ("TVL1", "deepflow"):
            eng = _engine(8, algo=algo)
            eng.calc_pairs(I0s, I1s)                                      # three units on the lanes (DeepFlow: one lane, its twin)
            assert eng.counter("queue_jobs") == 1
            t1 = eng.submit_pairs(I0s[:9], I1s[:9])
            t2 = eng.submit_pairs(I0s[9:], I1s[9:])
            eng.wait(t2)                                                  # t1 is never waited for: close() owns it
            if algo == "TVL1":
                eng.calc_study_saliency(rgb, scale=1.0)
            del t1
            eng.close()
        free.append(free_bytes())
        threads.append(_threads_now())
    assert max(threads[1:]) <= threads[1] + 1, threads                    # (round 0 starts the runtime's own threads)
    assert min(free[2:]) >= free[1] - (8 << 20), [f >> 20 for f in free]  # MiB free after each round
