"""Row f1 on the GPU: projection, histogram and the percentiles' order statistics from HIP kernels == the reference's own
output (fixture) and == the host restatement on a larger random study.  float64 throughout, bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_analysis.npz")


def test_device_radlong_matches_reference_fixture(engine):
    from tee_optical_flow_amd import analysis as A
    g = np.load(G)
    cent = [tuple(c) for c in g["centroids"]]
    out = A.radlong_stats_device(engine, g["flow"], cent, return_arrays=True)
    assert np.array_equal(out["rad_arr"], g["rad"]) and np.array_equal(out["long_arr"], g["long"])
    for key, name in (("radial", "rad"), ("longitudinal", "long")):
        f, e, hi, lo = out[key]
        assert np.array_equal(f, g[name + "_freq"])
        assert np.array_equal(e, g[name + "_edges"][:-1])
        assert np.array_equal(hi, g[name + "_hi"]) and np.array_equal(lo, g[name + "_lo"])


def test_device_radlong_matches_host_on_a_study_sized_input(engine):
    from tee_optical_flow_amd import analysis as A
    rng = np.random.default_rng(9)
    N, H, W = 12, 200, 256
    flow = (rng.normal(0, 4, (N, H, W, 2)) * (rng.random((N, H, W, 1)) > 0.3)).astype(np.float16).astype(np.float32)
    flow[7] = 0
    cent = [(H / 2 + rng.normal(0, 10), W / 2 + rng.normal(0, 10)) for _ in range(N)]
    dev = A.radlong_stats_device(engine, flow, cent, perc_lo=5, perc_hi=95, nbins=500, return_arrays=True)
    rad, lon = A.calculate_comp_magnitude(flow, cent)
    assert np.array_equal(dev["rad_arr"], rad) and np.array_equal(dev["long_arr"], lon)
    for key, arr in (("radial", rad), ("longitudinal", lon)):
        f, e, hi, lo = A.calc_bidirectional_hist(arr, N, perc_lo=5, perc_hi=95, nbins=500)
        df, de, dhi, dlo = dev[key]
        assert np.array_equal(df, f) and np.array_equal(de, np.asarray(e)[:-1]) and np.array_equal(dhi, hi) and np.array_equal(dlo, lo)
