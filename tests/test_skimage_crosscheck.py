"""Independent cross-check of the DualTVL1 oracle: scikit-image's own TV-L1 solver (skimage.registration.optical_flow_tvl1,
Wedel et al. -- a different implementation with different numerics: no median, its own warping and stopping rule) on the same
pairs.  This does NOT pin parity against OpenCV (that stays unpinned: cv2 is absent, see oracle/tvl1_oracle.c) -- it shows
that the restated algorithm recovers the same motion as an implementation nobody here wrote, to a few hundredths of a pixel.
skimage lives only in the image's second interpreter; the test skips where that is missing."""
import os
import subprocess

import numpy as np
import pytest

PY = "/opt/conda/bin/python3.9"

SCRIPT = r"""
import sys, warnings
warnings.filterwarnings("ignore")
import numpy as np
from skimage.registration import optical_flow_tvl1
d = np.load(sys.argv[1])
out = {}
for k in range(int(d["n"])):
    I0 = d[f"I0_{k}"].astype(np.float32) / 255; I1 = d[f"I1_{k}"].astype(np.float32) / 255
    v, u = optical_flow_tvl1(I0, I1)                  # (row, col) displacement, moving(r + v, c + u) ~ reference(r, c)
    out[f"f{k}"] = np.stack([u, v], -1).astype(np.float32)
np.savez(sys.argv[2], **out)
"""


def test_oracle_agrees_with_skimage_tvl1(oracle, tmp_path):
    if not os.path.exists(PY) or subprocess.run([PY, "-c", "import skimage.registration"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with scikit-image here")
    from tee_optical_flow_amd.synth import speckle_pair
    pairs = [speckle_pair(seed, 128, 128) for seed in (0, 1, 2)]
    arrs = {"n": len(pairs)}
    for k, (I0, I1, _) in enumerate(pairs):
        arrs[f"I0_{k}"], arrs[f"I1_{k}"] = I0, I1
    np.savez(tmp_path / "in.npz", **arrs)
    r = subprocess.run([PY, "-c", SCRIPT, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    sk = np.load(tmp_path / "out.npz")
    inner = (slice(16, -16), slice(16, -16))
    for k, (I0, I1, truth) in enumerate(pairs):
        ours = oracle.tvl1_calc(I0, I1)
        theirs = sk[f"f{k}"]
        epe = lambda a, b: float(np.sqrt(((a - b) ** 2).sum(-1))[inner].mean())
        # measured here: oracle-vs-skimage 0.050 / 0.062 / 0.050 px, each 0.03-0.05 px from the synthetic truth (|flow| 0.4-1.9 px)
        assert epe(ours, theirs) < 0.1, (k, epe(ours, theirs))
        assert epe(ours, truth) < 0.08 and epe(theirs, truth) < 0.08
