"""Row f1 on the CPU: the numpy restatement of the reference's rad/long projection and bidirectional histogram equals what
the reference's own functions produced (tests/golden/reference_analysis.npz, see make_reference_analysis_fixtures.py)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_analysis.npz")


def test_projection_and_histogram_match_reference_fixture():
    from tee_optical_flow_amd import analysis as A
    g = np.load(G)
    cent = [tuple(c) for c in g["centroids"]]
    rad, lon = A.calculate_comp_magnitude(g["flow"], cent)
    assert rad.dtype == np.float64 and np.array_equal(rad, g["rad"]) and np.array_equal(lon, g["long"])
    for name, arr in (("rad", rad), ("long", lon)):
        f, e, hi, lo = A.calc_bidirectional_hist(arr, len(cent))
        assert np.array_equal(f, g[name + "_freq"]) and np.array_equal(np.asarray(e), g[name + "_edges"])
        assert np.array_equal(hi, g[name + "_hi"]) and np.array_equal(lo, g[name + "_lo"])
    # frame 3 of the fixture holds no data: the reference repeats the previous frame's values
    assert g["rad_hi"][3] == g["rad_hi"][2] and np.array_equal(g["rad_freq"][3], g["rad_freq"][2])
