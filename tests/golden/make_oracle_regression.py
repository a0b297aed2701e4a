#!/usr/bin/env python3
"""Writes tests/golden/oracle_regression.npz: flows + executed iteration counts of oracle/tvl1_oracle.c on small
seeded pairs.  Self-generated REGRESSION vectors (they pin the restatement against accidental change; they are not
OpenCV outputs -- cv2 cannot be installed here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tee_optical_flow_amd.synth import speckle_pair  # noqa: E402

out = {}
cases = [(100, 48, 64), (101, 72, 56), (102, 33, 47)]
for i, (seed, h, w) in enumerate(cases):
    I0, I1, _ = speckle_pair(seed, h, w)
    f, it, nl = O.tvl1_calc(I0, I1, return_iters=True)
    out[f"I0_{i}"], out[f"I1_{i}"], out[f"flow_{i}"], out[f"iters_{i}"] = I0, I1, f, it[:nl]
out["n"] = np.int64(len(cases))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz"), **out)
print("ok")
