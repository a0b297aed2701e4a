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


class _Elem:
    def __init__(self, value):
        self.value = value


class _StubDS:
    """A pydicom-Dataset-shaped object: attribute access for keywords, [group, element] access for tags."""
    def __init__(self, tags=None, **kw):
        self._tags = tags or {}
        self.__dict__.update(kw)

    def __getitem__(self, key):
        return self._tags[key]


def test_dicom_metadata_follows_the_reference_rule_for_rule():
    """the .dcm branch must mirror _extract_dicom_metadata / the colour-space step of the reference
    (calculate_optical_flow.py:315-365, 524-526).  pydicom is absent here, so the rules run on a stub dataset."""
    from tee_optical_flow_amd.pipeline import dicom_to_study, extract_dicom_metadata
    region = [{"PhysicalDeltaX": _Elem(0.0321)}]
    # CineRate wins and is taken as stored (no rounding, no float())
    md = extract_dicom_metadata(_StubDS({(0x0018, 0x6011): region}, CineRate=30, FrameTime="33.333", RWaveTimeVector=[12.0, 845.5]))
    assert md["pixel_spacing"] == 0.0321 and md["frame_rate"] == 30 and md["R_wave_data_present"] is True
    assert isinstance(md["R_times"], np.ndarray) and md["R_times"].tolist() == [12.0, 845.5]
    # no CineRate: round(1000 / FrameTime) -- 1000 / 33.333 = 30.0003 -> 30.0; unrounded it would change every stored flow value
    md = extract_dicom_metadata(_StubDS({(0x0018, 0x6011): region}, FrameTime="33.333"))
    assert md["frame_rate"] == 30.0 and md["frame_rate"] == np.round(1000 / 33.333) and md["R_wave_data_present"] is False and md["R_times"] is None
    md = extract_dicom_metadata(_StubDS(FrameTime="21.7"))
    assert md["frame_rate"] == 46.0 and md["pixel_spacing"] is None
    # neither: round(1000 / FrameTimeVector[1])
    md = extract_dicom_metadata(_StubDS(FrameTimeVector=[0.0, 16.6, 16.6]))
    assert md["frame_rate"] == 60.0
    # FrameTime present but unusable (zero) falls through to the vector; nothing usable -> None (conversion factor 1.0 downstream)
    assert extract_dicom_metadata(_StubDS(FrameTime="0", FrameTimeVector=[0.0, 20.0]))["frame_rate"] == 50.0
    assert extract_dicom_metadata(_StubDS())["frame_rate"] is None
    # a bare float RWaveTimeVector is "not present" (the reference's own test), None likewise
    assert extract_dicom_metadata(_StubDS(RWaveTimeVector=3.5))["R_wave_data_present"] is False
    assert extract_dicom_metadata(_StubDS(RWaveTimeVector=None))["R_wave_data_present"] is False
    # the colour-space step runs before anything else sees the frames
    arr = np.zeros((2, 4, 4, 3), np.uint8)
    seen = []
    out, md, pid, hr = dicom_to_study(_StubDS(PatientID="P7", HeartRate=71, CineRate=25), arr, lambda ds, a: (seen.append(ds.PatientID), a + 1)[1])
    assert seen == ["P7"] and (out == 1).all() and pid == "P7" and hr == 71 and md["frame_rate"] == 25
    out, md, pid, hr = dicom_to_study(_StubDS(), arr)
    assert out is arr and pid == "" and hr == 0
