#!/opt/conda/bin/python3.9
"""Generates tests/golden/reference_analysis.npz by RUNNING the reference's own post-processing functions
(/root/reference/optical_flow/analysis.py:89-212: radial_vecgrid, calc_proj_mag, calculate_comp_magnitude,
calc_bidirectional_hist) on seeded inputs.  These are pure numpy; cv2 & co. are stubbed only so the module imports.

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_reference_analysis_fixtures.py
Fixtures are data (inputs + outputs); the reference never travels to the GPU box."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

for m in ["cv2", "pydicom", "torch", "torchvision", "torchvision.transforms", "peakutils", "polars", "tsmoothie",
          "tsmoothie.smoother", "neurokit2", "models", "models.sam", "imageio.v2", "matplotlib", "matplotlib.pyplot",
          "matplotlib.colors", "matplotlib.cm"]:
    sys.modules.setdefault(m, MagicMock())
sys.path.insert(0, "/root/reference")
import optical_flow.analysis as A  # noqa: E402

rng = np.random.default_rng(42)
N, H, W = 5, 40, 52
flow = rng.normal(0, 3, (N, H, W, 2)).astype(np.float16).astype(np.float32)     # what OpticalFlowDataset hands over
flow[:, 5:9, 7:15, :] = 0.0                                                      # masked-out (exact zero) pixels
flow[3] = 0.0                                                                     # a frame with no data at all
cent = [(H / 2 + rng.normal(0, 3), W / 2 + rng.normal(0, 3)) for _ in range(N)]
cent[1] = (12.0, 20.0)                                                            # centroid exactly on a pixel: 0/0 -> 0
rad, lon = A.calculate_comp_magnitude(flow, cent)
out = {"flow": flow, "centroids": np.asarray(cent, np.float64), "rad": rad, "long": lon}
for name, arr in (("rad", rad), ("long", lon)):
    freq, edges, hi, lo = A.calc_bidirectional_hist(arr, N, perc_lo=1, perc_hi=99, nbins=1000)
    out[f"{name}_freq"] = freq
    out[f"{name}_edges"] = np.asarray(edges)
    out[f"{name}_hi"] = hi
    out[f"{name}_lo"] = lo
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_analysis.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
