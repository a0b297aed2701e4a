"""CV_32FC1 frames (cv2's DualTVL1 takes float32 images in [0,1] and scales them by 255): tf_calc_pair_f32 / tf_calc_pairs_f32
against the oracle's orc_tvl1_calc_f32, bit for bit.  The reference itself always passes uint8 (calculate_optical_flow.py:588);
this closes the cv2 calc() surface.  parity vs real OpenCV: unpinned (cv2 is absent), like every DualTVL1 test here."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _f32_pairs(seeds, H, W):
    from tee_optical_flow_amd.synth import speckle_pairs
    I0s, I1s = speckle_pairs(seeds, H, W)
    rng = np.random.default_rng(seeds[0])
    # not just u8/255: add sub-level detail so the float path is really exercised
    f0 = (I0s.astype(np.float32) + rng.random(I0s.shape, dtype=np.float32)) / np.float32(256)
    f1 = (I1s.astype(np.float32) + rng.random(I1s.shape, dtype=np.float32)) / np.float32(256)
    return np.ascontiguousarray(f0), np.ascontiguousarray(f1)


@pytest.mark.parametrize("H,W,B,variant", [(64, 80, 1, "cpu"), (37, 53, 3, "cpu"), (112, 112, 40, "cpu"), (48, 64, 2, "cuda")])
def test_f32_frames_match_oracle(oracle, H, W, B, variant):
    import tee_optical_flow_amd as T
    f0, f1 = _f32_pairs(list(range(7, 7 + B)), H, W)
    eng = T.DenseFlow(variant=variant)
    try:
        flows = eng.calc_pairs(f0, f1) if B > 1 else eng.calc(f0[0], f1[0])[None]
        iters = eng.last_iters()
        op = oracle.default_params(variant=1 if variant == "cuda" else 0)
        for b in sorted({0, B - 1, B // 2}):
            ref, ref_it, nl = oracle.tvl1_calc(f0[b], f1[b], params=op, return_iters=True)
            assert np.array_equal(flows[b], ref), f"pair {b}: {np.sum(flows[b] != ref)} values differ"
            assert np.array_equal(iters[b], ref_it[:nl])
        assert np.isfinite(flows).all() and np.abs(flows).max() > 0.05
    finally:
        eng.close()


def test_f32_rejections():
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.exceptions import OpticalFlowCalculationError
    f = np.zeros((32, 32), np.float32)
    eng = T.DenseFlow()
    try:
        with pytest.raises(OpticalFlowCalculationError):
            eng.calc(f, np.zeros((32, 32), np.uint8))                # mixed depths
        with pytest.raises(OpticalFlowCalculationError):
            eng.calc(f.astype(np.float64), f.astype(np.float64))     # CV_64F is not accepted by cv2 either
        assert np.array_equal(eng.calc(f, f), np.zeros((32, 32, 2), np.float32))
        # a u8 call right after a float call takes the byte path again
        z = np.zeros((32, 32), np.uint8)
        assert np.array_equal(eng.calc(z, z), np.zeros((32, 32, 2), np.float32))
    finally:
        eng.close()


@pytest.mark.parametrize("H,W,B", [(64, 80, 1), (90, 130, 5), (200, 260, 20)])
def test_deepflow_takes_float_frames_as_they_are(oracle, H, W, B):
    """cv2's DeepFlow converts with convertTo(CV_32F) and NO factor: float frames in [0,1] stay in [0,1] (a different problem from the same
    frames in 0..255: zeta and epsilon are fixed), float frames holding 0..255 values give the uint8 frames' flow exactly."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.synth import speckle_pairs
    f0, f1 = _f32_pairs(list(range(30, 30 + B)), H, W)
    I0s, I1s = speckle_pairs(list(range(30, 30 + B)), H, W)
    eng = T.DenseFlow(algo="deepflow", max_batch=8)
    try:
        flows = np.array(eng.calc_pairs(f0, f1)) if B > 1 else np.array(eng.calc(f0[0], f1[0]))[None]
        for b in sorted({0, B - 1, B // 2}):
            assert np.array_equal(flows[b], oracle.deepflow_calc(f0[b], f1[b])), f"pair {b}"
        as_bytes = np.array(eng.calc_pairs(I0s, I1s)) if B > 1 else np.array(eng.calc(I0s[0], I1s[0]))[None]
        same_values = np.array(eng.calc_pairs(I0s.astype(np.float32), I1s.astype(np.float32))) if B > 1 else np.array(eng.calc(I0s[0].astype(np.float32), I1s[0].astype(np.float32)))[None]
        assert np.array_equal(as_bytes, same_values)
        assert not np.array_equal(flows, as_bytes) and np.isfinite(flows).all()
    finally:
        eng.close()
