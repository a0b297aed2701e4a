"""WASE background compensation on the device against numpy's own np.mean -- bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host(flows, mask):
    from tee_optical_flow_amd.pipeline import wase_background
    bg = np.array([wase_background(f, mask) for f in flows], np.float32)
    return np.stack([f - b for f, b in zip(flows, bg)]), bg


@pytest.mark.parametrize("shape", [(3, 5, 24, 40), (2, 9, 64, 64), (4, 33, 97, 131), (1, 2, 3, 5), (2, 40, 128, 128)])
def test_wase_matches_numpy_mean_bitwise(engine, shape):
    P, N, H, W = shape
    rng = np.random.default_rng(P * 1000 + N)
    flows = (rng.standard_normal((P, H, W, 2)) * 3).astype(np.float32)
    flows[:, : H // 4, : W // 3] = 0.0                       # exact zeros never count
    flows[0, -1, -1] = -0.0
    mask = rng.random((N, H, W, 2)) < 0.37
    mask[0] = False
    ref, ref_bg = _host(flows, mask)
    out, bg = engine.wase_compensate(flows, mask)
    assert bg.tobytes() == ref_bg.tobytes(), (bg, ref_bg)
    assert np.ascontiguousarray(out).view(np.uint32).tobytes() == np.ascontiguousarray(ref).view(np.uint32).tobytes()


def test_wase_scale_empty_selection_and_sizes_across_numpy_pieces(engine):
    rng = np.random.default_rng(77)
    P, N, H, W = 3, 3, 64, 65
    flows = rng.standard_normal((P, H, W, 2)).astype(np.float32)
    mask = np.zeros((N, H, W, 2), bool)
    # selections of 0, 1, 7, 8, 129, 8191, 8192, 8193 and 20000 terms exercise every branch of the summation order
    for p, k in enumerate([0, 8193, 20000]):
        m = np.zeros(N * H * W * 2, bool)
        m[rng.choice(m.size, k, replace=False)] = True
        mk = m.reshape(N, H, W, 2)
        import warnings
        with warnings.catch_warnings(), np.errstate(invalid="ignore"):
            warnings.simplefilter("ignore")
            ref, ref_bg = _host(flows[p:p + 1], mk)
        out, bg = engine.wase_compensate(flows[p:p + 1], mk, scale=2.5)
        if k == 0:                                            # np.mean of an empty selection: nan (its sign bit is the host FPU's)
            assert np.isnan(bg[0]) and np.isnan(ref_bg[0]) and np.isnan(out).all()
            continue
        assert bg.tobytes() == ref_bg.tobytes(), (k, bg, ref_bg)
        assert np.array_equal(out, ref * np.float32(2.5))
    for k in (1, 7, 8, 129, 8191, 8192):
        m = np.zeros(N * H * W * 2, bool)
        m[rng.choice(m.size, k, replace=False)] = True
        mk = m.reshape(N, H, W, 2)
        _, ref_bg = _host(flows[:1], mk)
        _, bg = engine.wase_compensate(flows[:1], mk)
        assert bg.tobytes() == ref_bg.tobytes(), (k, bg, ref_bg)


def test_flow_for_study_wase_device_equals_host_path(engine):
    from tee_optical_flow_amd.pipeline import flow_for_study, _compensate
    from tee_optical_flow_amd.synth import speckle_sequence
    fr = speckle_sequence(5, 6, 48, 64)
    rng = np.random.default_rng(3)
    mask = {"bkgd": rng.random((6, 48, 64, 2)) < 0.5}
    dev = flow_for_study(fr, engine, mask_dict=mask, bkgd_comp="WASE", conversion_factor=0.04 * 50.0)
    flows = engine.calc_batch(fr)
    host = np.stack([_compensate(f, mask, "WASE") for f in flows])
    host = np.concatenate([host, host[-1:]]) * (0.04 * 50.0)
    assert np.array_equal(dev, host)
