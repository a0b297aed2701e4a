"""End-to-end parity of the HIP DualTVL1 path (through the C ABI) against the CPU oracle.

north_star's tolerance is <= 1e-3 mean EPE; because kernels and oracle share one arithmetic contract the
tests demand more: identical executed-iteration counts and BIT-EXACT flow.  (Parity against real OpenCV is
unpinned -- cv2 is not installable here; see oracle/tvl1_oracle.c.)"""
import numpy as np
import pytest


pytestmark = pytest.mark.gpu

EPE_TOL_MEAN = 1e-3   # BASELINE.json north_star
EPE_TOL_MAX = 1e-2


def _epe(a, b):
    return np.sqrt(((a - b) ** 2).sum(-1))


@pytest.mark.parametrize("seed,H,W", [(0, 128, 128), (1, 160, 200), (2, 256, 256), (3, 97, 131), (5, 40, 64)])
def test_pair_matches_oracle(engine, oracle, seed, H, W):
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, _ = speckle_pair(seed, H, W)
    ref, ref_it, ref_levels = oracle.tvl1_calc(I0, I1, return_iters=True)
    out = engine.calc(I0, I1, None)
    it = engine.last_iters()[0]
    assert out.dtype == np.float32 and out.shape == (H, W, 2)
    assert engine.last_stats["nscales_used"] == ref_levels
    assert np.array_equal(it, ref_it[:ref_levels]), f"iteration counts differ:\n{it[..., 0]}\n{ref_it[:ref_levels, :, 0]}"
    e = _epe(out, ref)
    assert e.mean() <= EPE_TOL_MEAN and e.max() <= EPE_TOL_MAX
    assert np.array_equal(out, ref), f"not bit-exact: {np.sum(out != ref)} values differ, max {np.abs(out - ref).max()}"


def test_full_size_512_matches_oracle(engine, oracle):
    """BASELINE.json configs[1]: one 512x512 pair, all-default DualTVL1 (lambda 0.15)."""
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, truth = speckle_pair(0, 512, 512)
    ref, ref_it, _ = oracle.tvl1_calc(I0, I1, return_iters=True)
    out = engine.calc(I0, I1, None)
    assert np.array_equal(engine.last_iters()[0], ref_it)
    e = _epe(out, ref)
    assert e.mean() <= EPE_TOL_MEAN and e.max() <= EPE_TOL_MAX
    assert np.array_equal(out, ref)
    # and the solver actually solves the problem (interior EPE vs the known synthetic flow)
    assert _epe(out, truth)[16:-16, 16:-16].mean() < 0.1


def test_identical_frames_give_exact_zero(engine):
    from tee_optical_flow_amd.synth import speckle_pair
    I0, _, _ = speckle_pair(11, 96, 96)
    out = engine.calc(I0, I0, None)
    assert np.all(out == 0.0)


def test_batch_equals_singles_and_sequence_mode(engine, oracle):
    """tf_calc_pairs / tf_calc_seq: lock-step batching must not change any pair's result."""
    from tee_optical_flow_amd.synth import speckle_pairs, speckle_sequence
    I0s, I1s = speckle_pairs(range(20, 27), 96, 120)
    flows = engine.calc_pairs(I0s, I1s)
    iters = engine.last_iters()
    for b in range(len(I0s)):
        ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], return_iters=True)
        assert np.array_equal(flows[b], ref), f"pair {b}"
        assert np.array_equal(iters[b], ref_it[:nl])
    frames = speckle_sequence(30, 6, 80, 96)
    fseq = engine.calc_batch(frames, scale=2.5)
    assert fseq.shape == (5, 80, 96, 2)
    for i in range(5):
        ref = oracle.tvl1_calc(frames[i], frames[i + 1])
        assert np.array_equal(fseq[i], ref * np.float32(2.5)), f"seq pair {i}"


def test_sub_batching_when_batch_exceeds_capacity(oracle):
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    eng = T.DenseFlow(max_batch=3)
    I0s, I1s = speckle_pairs(range(40, 47), 64, 64)
    flows = eng.calc_pairs(I0s, I1s)
    for b in range(7):
        assert np.array_equal(flows[b], oracle.tvl1_calc(I0s[b], I1s[b]))
    eng.close()


@pytest.mark.parametrize("params", [dict(lambda_=0.05), dict(median_filtering=3), dict(median_filtering=1),
                                    dict(nscales=3, warps=2), dict(inner_iterations=7, outer_iterations=3),
                                    dict(epsilon=0.05, tau=0.2, theta=0.25), dict(scale_step=0.55)])
def test_non_default_parameters_match_oracle(oracle, params):
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, _ = speckle_pair(50, 120, 136)
    eng = T.DenseFlow(**params)
    out = eng.calc(I0, I1, None)
    it = eng.last_iters()[0]
    op = oracle.default_params(**{("lambda" if k == "lambda_" else k): v for k, v in params.items()})
    ref, ref_it, nl = oracle.tvl1_calc(I0, I1, op, return_iters=True)
    assert np.array_equal(it, ref_it[:nl])
    assert np.array_equal(out, ref)
    eng.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("params", [dict(), dict(inner_iterations=7, outer_iterations=4), dict(inner_iterations=4, outer_iterations=3, epsilon=0.002),
                                    dict(median_filtering=1, epsilon=0.03), dict(inner_iterations=9, outer_iterations=5, epsilon=0.004),
                                    dict(inner_iterations=3, outer_iterations=4, median_filtering=3)])
def test_every_iteration_kernel_form_end_to_end(oracle, variant, params):
    """The three tvl1_iter forms (tiles / row strips / two iterations per launch with per-pair REPLAY of an overshoot)
    must give the oracle's flow and iteration counts exactly -- including stops on odd iterations, stages that hit the
    iteration cap (tiny epsilon), odd `inner` (falls back to one iteration per launch) and a batch whose pairs stop at
    different iterations."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(70, 76), 72, 88)
    eng = T.DenseFlow(**params)
    eng.set_tuning("iter_variant", variant)
    eng.set_tuning("min_rows_work", 0)
    flows = eng.calc_pairs(I0s, I1s)
    iters = eng.last_iters()
    op = oracle.default_params(**params)
    odd = even = 0
    for b in range(len(I0s)):
        ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], op, return_iters=True)
        assert np.array_equal(iters[b], ref_it[:nl]), f"pair {b} iteration counts"
        assert np.array_equal(flows[b], ref), f"pair {b} flow"
        odd += int((ref_it[:nl, :, 0] % 2 == 1).sum()); even += int((ref_it[:nl, :, 0] % 2 == 0).sum())
    if not params:
        assert odd > 0 and even > 0    # both the REPLAY and the clean-stop path were exercised
    eng.close()


def test_two_lane_split_of_large_batches(oracle):
    """Batches of >= 32 pairs are split over two (handle, stream, host thread) lanes; results and per-pair iteration counts
    must be those of the single-lane run (and of the oracle), in the caller's pair order; sequence mode overlaps one frame."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs, speckle_sequence
    I0s, I1s = speckle_pairs(range(200, 237), 48, 64)          # 37 pairs -> lanes of 18 and 19
    eng = T.DenseFlow()
    f2 = eng.calc_pairs(I0s, I1s)
    it2 = eng.last_iters()
    eng.set_tuning("lanes", 1)
    f1 = eng.calc_pairs(I0s, I1s)
    it1 = eng.last_iters()
    assert np.array_equal(f1, f2) and np.array_equal(it1, it2) and it2.shape[0] == 37
    for b in (0, 17, 18, 36):
        ref, ref_it, nl = oracle.tvl1_calc(I0s[b], I1s[b], return_iters=True)
        assert np.array_equal(f2[b], ref) and np.array_equal(it2[b], ref_it[:nl])
    eng.set_tuning("lanes", 2)
    fr = speckle_sequence(7, 40, 40, 48)                         # 39 pairs
    fs = eng.calc_batch(fr)
    for i in (0, 18, 19, 38):
        assert np.array_equal(fs[i], oracle.tvl1_calc(fr[i], fr[i + 1])), i
    eng.close()


@pytest.mark.parametrize("width,max_strip_width", [(1100, 2048), (1100, 1024), (2050, 2048)])
def test_wide_images_strip_kernel_up_to_2048_then_tiles(oracle, width, max_strip_width):
    """Levels up to 2048 px wide run the full-width strip kernel (512-thread blocks above 1024 px); wider ones, or a lower
    `max_strip_width`, the 64x16-tile form.  Same bits either way."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, _ = speckle_pair(90, 48, width)
    eng = T.DenseFlow(nscales=3)
    eng.set_tuning("min_rows_work", 0)
    eng.set_tuning("max_strip_width", max_strip_width)
    out = eng.calc(I0, I1, None)
    ref, ref_it, nl = oracle.tvl1_calc(I0, I1, oracle.default_params(nscales=3), return_iters=True)
    assert np.array_equal(eng.last_iters()[0], ref_it[:nl]) and np.array_equal(out, ref)
    eng.close()


def test_tiny_images(oracle):
    import tee_optical_flow_amd as T
    eng = T.DenseFlow()
    rng = np.random.default_rng(3)
    for shape in [(8, 8), (17, 5), (16, 16), (1, 33)]:
        I0 = rng.integers(0, 256, shape, dtype=np.uint8)
        I1 = rng.integers(0, 256, shape, dtype=np.uint8)
        out = eng.calc(I0, I1, None)
        ref, ref_it, nl = oracle.tvl1_calc(I0, I1, return_iters=True)
        assert eng.last_stats["nscales_used"] == nl
        assert np.array_equal(out, ref), shape
    eng.close()


def test_echo_like_sector_with_black_background(engine, oracle):
    """Real TEE frames are a bright sector on an exactly-black background (zero image gradient: the flow there is pure
    TV diffusion from the sector).  Full solve, bit patterns compared (zero signs included)."""
    from tee_optical_flow_amd.synth import speckle_pair
    H, W = 192, 256
    I0, I1, _ = speckle_pair(31, H, W)
    yy, xx = np.mgrid[0:H, 0:W]
    ang = np.arctan2(xx - W / 2, yy + 8.0)
    sector = (np.abs(ang) < 0.6) & (np.hypot(xx - W / 2, yy + 8.0) < H * 0.95)
    I0 = np.where(sector, I0, 0).astype(np.uint8)
    I1 = np.where(sector, I1, 0).astype(np.uint8)
    ref, ref_it, _ = oracle.tvl1_calc(I0, I1, return_iters=True)
    out = engine.calc(I0, I1, None)
    nl = engine.last_iters().shape[1]
    assert np.array_equal(engine.last_iters()[0], ref_it[:nl])
    bad = np.ascontiguousarray(out).view(np.uint32) != np.ascontiguousarray(ref).view(np.uint32)
    assert not bad.any(), f"{bad.sum()} bit patterns differ, e.g. {out[bad][:3]!r} vs {ref[bad][:3]!r}"


def test_symmetries(engine):
    """SURVEY.md 8c item 3 (behavioural KATs on the GPU path itself): transpose swaps (u,v)."""
    from tee_optical_flow_amd.synth import speckle_pair
    I0, I1, _ = speckle_pair(60, 128, 128)
    f = engine.calc(I0, I1, None)
    ft = engine.calc(np.ascontiguousarray(I0.T), np.ascontiguousarray(I1.T), None)
    d = _epe(ft.transpose(1, 0, 2)[..., ::-1], f)
    assert d.mean() < 5e-2


def test_error_behaviour(engine):
    from tee_optical_flow_amd import OpticalFlowCalculationError
    a = np.zeros((32, 32), np.uint8)
    with pytest.raises(OpticalFlowCalculationError):
        engine.calc(a, np.zeros((32, 33), np.uint8), None)
    with pytest.raises(OpticalFlowCalculationError):
        engine.calc(a.astype(np.float64), a.astype(np.float64), None)     # CV_64F: cv2 rejects it too (CV_32F is accepted,
                                                                          # tests/test_gpu_f32_input.py)
    with pytest.raises(OpticalFlowCalculationError):
        engine.setGamma(0.5)          # unsupported -> loud error, parameter unchanged
    assert engine.getGamma() == 0.0
    engine.setLambda(0.15)
    assert engine.getLambda() == 0.15


def test_device_frame_conditioning_matches_reference_fixture(engine):
    """Row a1/f4: img2uint8(rgb2gray(frame)) on the device == the reference's own output (fixture) and the host restatement."""
    import os
    from tee_optical_flow_amd.frames import condition_frames
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_host_side.npz"))
    assert np.array_equal(engine.condition_frames(g["cond_in_rgb"][None])[0], g["cond_out_u8"])
    assert np.array_equal(engine.condition_frames(g["cond_in_ramp"][None])[0], g["cond_out_ramp"])
    rng = np.random.default_rng(5)
    study = rng.integers(0, 256, (7, 96, 130, 3), dtype=np.uint8)
    study[3] //= 4                      # a dark frame: per-frame normalisation
    study[5, ..., :] = study[5, ..., :1]  # a grey frame
    assert np.array_equal(engine.condition_frames(study), condition_frames(study))


def test_study_from_rgb_on_device_equals_host_conditioned_path(engine, oracle):
    from tee_optical_flow_amd.frames import condition_frames
    from tee_optical_flow_amd.pipeline import process_video
    from tee_optical_flow_amd.synth import speckle_sequence
    gsq = speckle_sequence(11, 5, 64, 80)
    nparr = np.repeat(gsq[..., None], 3, axis=3)
    a = engine.calc_study(nparr, scale=1.5)
    b = engine.calc_batch(condition_frames(nparr), scale=1.5)
    assert np.array_equal(a, b)
    md = {"pixel_spacing": 0.05, "frame_rate": 30.0, "R_wave_data_present": False, "R_times": None}
    out = process_video(None, None, None, verbose=False, mode="otsu", no_saliency=True, nparr=nparr, metadata=md, flow_model=engine)
    fr = condition_frames(nparr)
    assert out.shape == (5, 64, 80, 2) and np.array_equal(out[3], out[4])
    assert np.array_equal(out[0], oracle.tvl1_calc(fr[0], fr[1]) * (0.05 * 30.0))


def test_config5_process_video_to_hdf5_on_gpu(tmp_path, oracle):
    """BASELINE config 5 (as far as the offline image allows): a synthetic 256x256 study injected at the nparr level ->
    masks (Otsu path; SAM's result would be passed as mask_dict=) -> device conditioning + DualTVL1 on the GPU -> duplicated
    last flow, unit scale -> HDF5 with the reference's keys / dtypes / shapes / attrs.  h5py only exists in the image's second
    interpreter, so the study driver runs there (the engine needs ctypes + numpy only)."""
    import json
    import os
    import subprocess
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        pytest.skip("no interpreter with h5py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "study.h5")
    script = f"""
import sys, json, numpy as np
sys.path.insert(0, {root!r})
import h5py
from scipy import ndimage
from tee_optical_flow_amd.pipeline import process_video
rng = np.random.default_rng(1000)
N, H, W = 6, 256, 256
yy, xx = np.mgrid[0:H, 0:W]
base = ndimage.gaussian_filter(rng.standard_normal((H + 40, W + 40)), 2.5)
base = (base - base.min()) / (base.max() - base.min()) * 255
blob = np.exp(-(((xx - 128) / 70.0) ** 2 + ((yy - 128) / 60.0) ** 2))
frames = np.stack([np.clip(base[20 + i:20 + i + H, 20 + 2 * i:20 + 2 * i + W] * (0.25 + 0.75 * blob), 0, 255) for i in range(N)]).astype(np.uint8)
nparr = np.repeat(frames[..., None], 3, axis=3)
md = {{"pixel_spacing": 0.04, "frame_rate": 50.0, "R_wave_data_present": False, "R_times": None}}
flow = process_video(None, {out!r}, None, verbose=False, mode="otsu", no_saliency=True, nparr=nparr, metadata=md, patient_id="SYNTH-5", heart_rate=61)
np.save({str(tmp_path / 'flow.npy')!r}, flow)
np.save({str(tmp_path / 'nparr.npy')!r}, nparr)
d = {{}}
with h5py.File({out!r}, "r") as f:
    for k in f.keys():
        ds = f[k]
        d[k] = {{"dtype": str(ds.dtype), "shape": list(ds.shape), "compression": ds.compression, "compression_opts": ds.compression_opts,
                "attrs": {{n: [type(v).__name__, str(np.asarray(v).dtype)] for n, v in ds.attrs.items()}}}}
    ok = bool(np.array_equal(f["flow"][...], flow.astype(np.float16)))
print(json.dumps({{"layout": d, "payload_ok": ok}}))
"""
    env = {**os.environ, "PYTHONDONTWRITEBYTECODE": "1"}
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"       # conda ships an older libstdc++ than libamdhip64 needs
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    r = subprocess.run([py, "-c", script], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["payload_ok"]
    ref = json.load(open(os.path.join(root, "tests", "golden", "reference_host_side.json")))["hdf5_layout"]["no_waveforms"]
    lay = got["layout"]
    assert sorted(lay) == sorted(k for k in ref if k != "RWaveTime")          # echo, flow, otsu (no R-wave data in this study)
    for k in lay:
        assert lay[k]["dtype"] == ref[k]["dtype"] and lay[k]["compression"] == "gzip" and lay[k]["compression_opts"] == 9
        assert sorted(lay[k]["attrs"]) == sorted(ref[k]["attrs"])                       # same attribute names ...
        for a_, v_ in lay[k]["attrs"].items():
            r_ = ref[k]["attrs"][a_][:2]                                                  # ... with the same Python / numpy types
            assert v_[0] == r_[0] and v_[1].rstrip("0123456789") == r_[1].rstrip("0123456789"), (k, a_)   # '<U7' ~ '<U10'

    assert lay["flow"]["shape"] == [6, 256, 256, 2] and lay["echo"]["shape"] == [6, 256, 256] and lay["otsu"]["shape"] == [6, 256, 256, 2]
    # the flow itself: first pair against the oracle on the conditioned frames, times pixel_spacing * frame_rate
    from tee_optical_flow_amd.frames import condition_frames
    flow = np.load(str(tmp_path / "flow.npy"))
    fr = condition_frames(np.load(str(tmp_path / "nparr.npy")))
    assert np.array_equal(flow[0], oracle.tvl1_calc(fr[0], fr[1]) * (0.04 * 50.0))
    assert np.array_equal(flow[-1], flow[-2])


def test_pinned_result_path_overlapped_copy_out_equals_pageable_path(oracle):
    """Results returned by DenseFlow live in pooled pinned host memory (copy-out of a sub-batch overlaps the solve of the
    next: 52 pairs through a capacity of 16); a pageable destination handed to the C ABI directly takes the in-order path.
    Same bits, and buffers are recycled."""
    import ctypes as C
    import gc
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import _lib
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(range(300, 352), 64, 96)
    eng = T.DenseFlow(max_batch=16)
    a = eng.calc_pairs(I0s, I1s)
    it_a = eng.last_iters().copy()
    out = np.empty((52, 64, 96, 2), np.float32)                  # pageable
    st = _lib.TfStats()
    _lib.check(eng._L.tf_calc_pairs(eng._h, I0s.ctypes.data, I1s.ctypes.data, 52, 64, 96, out.ctypes.data, C.byref(st)), eng._h)
    assert np.array_equal(a, out) and np.array_equal(it_a, eng.last_iters())
    for b in (0, 17, 51):
        assert np.array_equal(a[b], oracle.tvl1_calc(I0s[b], I1s[b]))
    addr = a.ctypes.data
    keep = a[3].copy()
    del a
    gc.collect()
    b2 = eng.calc_pairs(I0s, I1s)                                # the pooled buffer comes back
    assert b2.ctypes.data == addr and np.array_equal(b2[3], keep)
    eng.close()
