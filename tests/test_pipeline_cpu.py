"""Host logic of the study driver (process_video analogue) on the CPU.  The flow model is a TEST DOUBLE that answers
through the oracle -- the product's own model has no CPU path (tests/test_boundary_cpu.py checks that)."""
import numpy as np
import pytest


class OracleModel:
    """cv2-protocol stand-in used only by tests."""

    def __init__(self, oracle):
        self.o = oracle
        self.calls = 0

    def calc(self, I0, I1, flow=None):
        self.calls += 1
        return self.o.tvl1_calc(I0, I1)

    def calc_batch(self, frames, scale=1.0):
        self.calls += 1
        return np.stack([self.o.tvl1_calc(frames[i], frames[i + 1]) for i in range(len(frames) - 1)]) * np.float32(scale)

    def calc_pairs(self, I0s, I1s):
        return np.stack([self.o.tvl1_calc(a, b) for a, b in zip(I0s, I1s)])

    def close(self):
        pass


def _study(n=5, h=48, w=56, seed=1000):
    from tee_optical_flow_amd.synth import speckle_sequence
    g = speckle_sequence(seed, n, h, w)
    return np.repeat(g[..., None], 3, axis=3)       # uint8 RGB study injected at the nparr level (BASELINE config 1)


def test_flow_for_study_follows_reference_loop(oracle):
    """reference :584-600 -- N-1 pair flows, last one duplicated, then * conversion_factor."""
    from tee_optical_flow_amd.frames import condition_frames
    from tee_optical_flow_amd.pipeline import flow_for_study, calculate_optical_flow
    nparr = _study()
    fr = condition_frames(nparr)
    m = OracleModel(oracle)
    out = flow_for_study(fr, m, None, "none", 1.5)
    assert out.shape == (5, 48, 56, 2) and out.dtype == np.float32
    assert np.array_equal(out[-1], out[-2])
    # the reference's own loop, one cv2-style call per pair
    ref = [calculate_optical_flow(fr[i - 1], fr[i], {}, m, "none", "TVL1") for i in range(1, 5)]
    ref.append(ref[-1])
    assert np.array_equal(out, np.stack(ref) * 1.5)


def test_wase_background_is_one_scalar_over_all_frames(oracle):
    from tee_optical_flow_amd.frames import condition_frames
    from tee_optical_flow_amd.pipeline import flow_for_study
    nparr = _study(4, 40, 40, 7)
    fr = condition_frames(nparr)
    rng = np.random.default_rng(0)
    bk = np.repeat((rng.random((4, 40, 40)) > 0.5)[..., None], 2, axis=3)
    m = OracleModel(oracle)
    out = flow_for_study(fr, m, {"bkgd": bk}, "WASE", 1.0)
    raw = m.calc_batch(fr)
    for i in range(3):
        masked = raw[i] * bk
        assert np.array_equal(out[i], raw[i] - np.mean(masked[masked != 0]))


def test_process_video_validation_errors():
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.pipeline import process_video
    nparr = _study(3, 32, 32)
    with pytest.raises(T.ConfigurationError):
        process_video(None, None, None, mode="otsu", bkgd_comp="WASE", no_saliency=True, nparr=nparr)
    with pytest.raises(T.ConfigurationError):
        process_video(None, None, None, mode="otsu", save_mask_subset=["rv"], no_saliency=True, nparr=nparr)
    with pytest.raises(T.DICOMReadError):
        process_video("/nonexistent.dcm", None, None, mode="otsu", no_saliency=True)
    with pytest.raises(T.ConfigurationError):
        process_video(None, None, None, mode="bogus", no_saliency=True, nparr=nparr, flow_model=object())
    with pytest.raises(T.OpticalFlowCalculationError):
        from tee_optical_flow_amd.pipeline import make_flow_model
        make_flow_model("farneback")


def test_process_video_otsu_end_to_end_on_cpu_double(oracle):
    """BASELINE config 1 plumbing (256x256 in the config; 64x64 here to stay fast): frames -> conditioned u8 -> flows
    -> duplicated last -> unit scale; masks via the Otsu path."""
    from tee_optical_flow_amd.pipeline import process_video
    nparr = _study(4, 64, 64, 1000)
    md = {"pixel_spacing": 0.05, "frame_rate": 30.0, "R_wave_data_present": False, "R_times": None}
    out = process_video(None, None, None, verbose=False, mode="otsu", no_saliency=True, nparr=nparr, metadata=md,
                        flow_model=OracleModel(oracle))
    assert out.shape == (4, 64, 64, 2)
    assert np.array_equal(out[2], out[3])
    out_flip = process_video(None, None, None, verbose=False, mode="otsu", no_saliency=True, flipLR=True, nparr=nparr,
                             metadata=md, flow_model=OracleModel(oracle))
    assert out_flip.shape == out.shape and not np.array_equal(out_flip, out)
