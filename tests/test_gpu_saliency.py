"""The no_saliency=False preprocessing on the device (tf_saliency_frames / tf_calc_seq_saliency; reference
calculate_optical_flow.py:559-560, :586) against oracle/saliency_oracle.c -- byte for byte.  The oracle itself is unpinned
against OpenCV (tests/test_saliency_cpu.py says what pins it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref(oracle, frames):
    return np.stack([oracle.saliency_fine_grained(f) for f in frames])


@pytest.mark.parametrize("shape", [(3, 40, 64, 3), (2, 33, 17, 1), (4, 128, 128, 3), (1, 1, 40, 3), (1, 37, 1, 3), (2, 2, 2, 3),
                                   (2, 24, 300, 3), (3, 257, 255, 3), (2, 512, 512, 3)])
def test_saliency_maps_equal_the_oracle(engine, oracle, shape):
    N, H, W, ch = shape
    rng = np.random.default_rng(N * 7919 + H * 31 + W)
    frames = rng.integers(0, 256, (N, H, W, ch), dtype=np.uint8)
    frames[0, : H // 2] = 255 - frames[0, : H // 2] // 8            # a bright half: the float integral image passes 2^24 at 512^2
    src = frames if ch == 3 else frames[..., 0]
    got = engine.saliency_frames(src, dtype=np.uint8)
    ref = _ref(oracle, src)
    assert got.dtype == np.uint8 and got.shape == (N, H, W)
    assert np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} of {got.size} bytes differ"
    # the CV_32F form (the default: what computeSaliency() returns in opencv-contrib 4.x) = the 8-bit map * (1/255) in float, bit for bit
    gotf = engine.saliency_frames(src)
    reff = np.stack([oracle.saliency_fine_grained(f, np.float32) for f in src])
    assert gotf.dtype == np.float32 and np.array_equal(gotf.view(np.uint32), reff.view(np.uint32))
    assert np.array_equal(gotf, got.astype(np.float32) * np.float32(1.0 / 255.0)) and gotf.max() <= 1.0


def test_saliency_of_echo_like_frames_flat_frames_and_frame_independence(engine, oracle):
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(5, 6, 192, 256)                          # uint8 [6,192,256]
    rgb = np.repeat(seq[..., None], 3, axis=3)
    rgb[2] = 90                                                     # a flat frame: 0/0 in both scalings -> all zeros
    rgb[4, 50:60, 60:90] = (255, 40, 10)                            # a coloured overlay: the channel weights matter
    got = engine.saliency_frames(rgb, dtype="u8")
    assert np.array_equal(got, _ref(oracle, rgb))
    assert got[2].max() == 0 and got[0].max() > 0
    # frames do not see each other (per-frame maxima): any sub-stack gives the same maps
    assert np.array_equal(engine.saliency_frames(rgb[3:5], dtype="u8"), got[3:5])
    # gray stacks and their RGB repeats agree (the weights sum to 2^15)
    assert np.array_equal(engine.saliency_frames(seq), engine.saliency_frames(np.repeat(seq[..., None], 3, axis=3)))


def test_saliency_frames_in_chunks(engine, oracle):
    """9 frames of 4096 x 4096 exceed the 2^27-pixel work buffers: two chunks (8 + 1).  The oracle checks the first and the last frame
    (one 16.7-Mpx frame costs it ~1.5 s); the middle ones are checked through frame independence."""
    rng = np.random.default_rng(2)
    tile = rng.integers(0, 256, (9, 64, 64), dtype=np.uint8)
    frames = np.ascontiguousarray(np.tile(tile, (1, 64, 64)))       # [9,4096,4096] gray
    frames[:, 1000:3000, 500:3500] //= 3
    got = engine.saliency_frames(frames, dtype=np.uint8)
    for f in (0, 8):
        assert np.array_equal(got[f], oracle.saliency_fine_grained(frames[f])), f"frame {f}"
    assert np.array_equal(engine.saliency_frames(frames[3:5], dtype=np.uint8), got[3:5])


@pytest.mark.parametrize("algo", ["TVL1", "deepflow"])
@pytest.mark.parametrize("map_dtype", ["f32", "u8"])
def test_flow_on_saliency_maps_equals_solving_the_maps(oracle, algo, map_dtype):
    """tf_calc_seq_saliency / _f32 = saliency maps + the sequence solve, nothing else: the same bits as handing the maps to the solver, and
    the oracle's solver on the oracle's maps -- for both hand-over types and both algorithms.  Float maps reach DualTVL1 as CV_32F in
    [0,1] (x 255 in float: close to the 8-bit map's flow, not equal) and DeepFlow as they are (a different problem: zeta and epsilon are
    not rescaled)."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(9, 4, 96, 128)
    rgb = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
    eng = T.DenseFlow(algo=algo)
    try:
        flows = np.array(eng.calc_study_saliency(rgb, map_dtype=map_dtype))
        padded = np.array(eng.calc_study_saliency(rgb, scale=2.0, pad_last=True, map_dtype=map_dtype))
        maps = eng.saliency_frames(rgb, dtype=map_dtype)
        assert flows.shape == (3, 96, 128, 2) and padded.shape == (4, 96, 128, 2)
        assert np.array_equal(padded[:3], flows * np.float32(2.0)) and np.array_equal(padded[3], padded[2])
        assert np.array_equal(flows, np.array(eng.calc_pairs(maps[:-1], maps[1:])))
        dt = np.float32 if map_dtype == "f32" else np.uint8
        ref_maps = np.stack([oracle.saliency_fine_grained(f, dt) for f in rgb])
        solve = oracle.tvl1_calc if algo == "TVL1" else oracle.deepflow_calc
        for i in (0, 2):
            assert np.array_equal(flows[i], solve(ref_maps[i], ref_maps[i + 1])), f"pair {i}"
        other = np.array(eng.calc_study_saliency(rgb, map_dtype="u8" if map_dtype == "f32" else "f32"))
        d = float(np.abs(other - flows).mean())
        assert d > 0, "the two hand-over types must not give the same flow"
        if algo == "TVL1":
            assert d < 0.05                                            # x 255 in float: nearly the integers
    finally:
        eng.close()


@pytest.mark.parametrize("algo", ["TVL1", "deepflow"])
def test_process_video_default_branch_runs_on_saliency_maps(algo):
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.pipeline import process_video
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(21, 5, 64, 80)
    nparr = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
    md = {"pixel_spacing": 0.05, "frame_rate": 40.0, "R_wave_data_present": False, "R_times": None}
    eng = T.DenseFlow(algo=algo)
    try:
        kw = dict(verbose=False, mode="otsu", nparr=nparr, metadata=md, flow_model=eng, OF_algo=algo)
        out = process_video(None, None, None, **kw)                     # no_saliency defaults to False, the map to CV_32F in [0,1]
        maps = eng.saliency_frames(nparr)
        assert maps.dtype == np.float32
        ref = np.array(eng.calc_pairs(maps[:-1], maps[1:]))
        ref = np.concatenate([ref, ref[-1:]]) * np.float32(0.05 * 40.0)
        assert np.array_equal(out, ref)
        out8 = process_video(None, None, None, saliency_map="u8", **kw)
        ref8 = np.array(eng.calc_batch(eng.saliency_frames(nparr, dtype="u8")))
        assert np.array_equal(out8, np.concatenate([ref8, ref8[-1:]]) * np.float32(0.05 * 40.0)) and not np.array_equal(out8, out)
        gray = process_video(None, None, None, no_saliency=True, **kw)
        assert not np.array_equal(out, gray)
    finally:
        eng.close()
