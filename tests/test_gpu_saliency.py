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
    got = engine.saliency_frames(frames if ch == 3 else frames[..., 0])
    ref = _ref(oracle, frames if ch == 3 else frames[..., 0])
    assert got.dtype == np.uint8 and got.shape == (N, H, W)
    assert np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} of {got.size} bytes differ"


def test_saliency_of_echo_like_frames_flat_frames_and_frame_independence(engine, oracle):
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(5, 6, 192, 256)                          # uint8 [6,192,256]
    rgb = np.repeat(seq[..., None], 3, axis=3)
    rgb[2] = 90                                                     # a flat frame: 0/0 in both scalings -> all zeros
    rgb[4, 50:60, 60:90] = (255, 40, 10)                            # a coloured overlay: the channel weights matter
    got = engine.saliency_frames(rgb)
    assert np.array_equal(got, _ref(oracle, rgb))
    assert got[2].max() == 0 and got[0].max() > 0
    # frames do not see each other (per-frame maxima): any sub-stack gives the same maps
    assert np.array_equal(engine.saliency_frames(rgb[3:5]), got[3:5])
    # gray stacks and their RGB repeats agree (the weights sum to 2^15)
    assert np.array_equal(engine.saliency_frames(seq), engine.saliency_frames(np.repeat(seq[..., None], 3, axis=3)))


def test_saliency_frames_in_chunks(engine, oracle):
    """9 frames of 4096 x 4096 exceed the 2^27-pixel work buffers: two chunks (8 + 1).  The oracle checks the first and the last frame
    (one 16.7-Mpx frame costs it ~1.5 s); the middle ones are checked through frame independence."""
    rng = np.random.default_rng(2)
    tile = rng.integers(0, 256, (9, 64, 64), dtype=np.uint8)
    frames = np.ascontiguousarray(np.tile(tile, (1, 64, 64)))       # [9,4096,4096] gray
    frames[:, 1000:3000, 500:3500] //= 3
    got = engine.saliency_frames(frames)
    for f in (0, 8):
        assert np.array_equal(got[f], oracle.saliency_fine_grained(frames[f])), f"frame {f}"
    assert np.array_equal(engine.saliency_frames(frames[3:5]), got[3:5])


def test_flow_on_saliency_maps_equals_solving_the_maps(engine, oracle):
    """tf_calc_seq_saliency = saliency maps + the sequence solve, nothing else: same bits as handing the maps to calc_batch, and the
    oracle's DualTVL1 on the oracle's maps."""
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(9, 4, 96, 128)
    rgb = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
    flows = engine.calc_study_saliency(rgb)
    maps = engine.saliency_frames(rgb)
    assert flows.shape == (3, 96, 128, 2)
    assert np.array_equal(flows, engine.calc_batch(maps))
    ref_maps = _ref(oracle, rgb)
    ref0 = oracle.tvl1_calc(ref_maps[0], ref_maps[1])
    assert np.array_equal(flows[0], ref0)


def test_process_video_default_branch_runs_on_saliency_maps(engine):
    from tee_optical_flow_amd.pipeline import process_video
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(21, 5, 64, 80)
    nparr = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
    md = {"pixel_spacing": 0.05, "frame_rate": 40.0, "R_wave_data_present": False, "R_times": None}
    out = process_video(None, None, None, verbose=False, mode="otsu", nparr=nparr, metadata=md, flow_model=engine)   # no_saliency defaults to False
    ref = engine.calc_batch(engine.saliency_frames(nparr))
    ref = np.concatenate([ref, ref[-1:]]) * (0.05 * 40.0)
    assert np.array_equal(out, ref)
    gray = process_video(None, None, None, verbose=False, mode="otsu", no_saliency=True, nparr=nparr, metadata=md, flow_model=engine)
    assert not np.array_equal(out, gray)
