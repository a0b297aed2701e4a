#!/usr/bin/env python3
"""Generates tests/golden/reference_waveforms.npz + .json by IMPORTING the reference's own waveform loader
(/root/reference/optical_flow/waveform_loader.py, importable as it is: numpy only) in THIS container and recording what
`load_all_waveforms` returns for
  * the one data set the reference ships (/root/reference/test_data/waveforms/stanford_RVIO_49_2_*.npy), and
  * a few synthetic studies that exercise every branch (flat PAP, PAP out of range / negative, CVP out of range either way, no ECG,
    flat ART with and without a usable ABP, ABP without ART, nothing at all).
Run (build container only; the reference never travels to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_reference_waveform_fixtures.py
The fixture holds DATA only: the input arrays (the reference's own test-data arrays included) and, per case, which keys the
reference accepted and the array it returned.  No reference source text is stored.
"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, "/root/reference")
from optical_flow.waveform_loader import load_all_waveforms  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SUFFIX = ("II", "ART", "ABP", "PAP", "CVP")
rng = np.random.default_rng(11)
t = np.arange(400) / 125.0


def wave(mean, amp):
    return (mean + amp * np.sin(2 * np.pi * 1.2 * t) + rng.normal(0, 0.01, t.size)).astype(np.float64)


cases = {}
# the reference's own data set
ref_dir = "/root/reference/test_data/waveforms"
cases["stanford_RVIO_49_2"] = {s: np.load(os.path.join(ref_dir, f"stanford_RVIO_49_2_{s}.npy")) for s in SUFFIX}
cases["all_good"] = {"II": wave(0, 1), "ART": wave(80, 20), "PAP": wave(25, 8), "CVP": wave(8, 3)}
cases["pap_flat"] = {"II": wave(0, 1), "ART": wave(80, 20), "PAP": np.full(300, 20.0), "CVP": wave(8, 3)}
cases["pap_high"] = {"II": wave(0, 1), "ART": wave(80, 20), "PAP": wave(140, 8)}
cases["pap_negative"] = {"II": wave(0, 1), "PAP": wave(-5, 3)}
cases["cvp_high"] = {"II": wave(0, 1), "CVP": wave(75, 3)}
cases["cvp_low"] = {"ART": wave(80, 20), "CVP": wave(-30, 3)}
cases["cvp_flat_is_fine"] = {"II": wave(0, 1), "CVP": np.full(300, 6.0)}
cases["no_ecg"] = {"ART": wave(80, 20), "PAP": wave(25, 8)}
cases["art_flat_abp_good"] = {"II": wave(0, 1), "ART": np.full(300, 90.0), "ABP": wave(85, 25)}
cases["art_flat_abp_flat"] = {"II": wave(0, 1), "ART": np.full(300, 90.0), "ABP": np.full(300, 70.0)}
cases["art_flat_no_abp"] = {"ART": np.full(300, 90.0), "CVP": wave(8, 3)}
cases["abp_only"] = {"II": wave(0, 1), "ABP": wave(85, 25)}
cases["abp_only_flat"] = {"ABP": np.full(300, 70.0), "PAP": wave(25, 8), "CVP": wave(8, 3)}
cases["nothing"] = {}

arrs, meta = {}, {}
with tempfile.TemporaryDirectory() as d:
    for name, files in cases.items():
        for s, a in files.items():
            np.save(os.path.join(d, f"{name}_{s}.npy"), a)
            arrs[f"in/{name}/{s}"] = a
        res = load_all_waveforms(os.path.join("/studies", name + ".dcm"), d)
        meta[name] = {}
        for k, (ok, a) in res.items():
            meta[name][k] = {"valid": bool(ok), "shape": None if a is None else list(a.shape)}
            if a is not None:
                arrs[f"out/{name}/{k}"] = np.asarray(a)
np.savez_compressed(os.path.join(OUT, "reference_waveforms.npz"), **arrs)
with open(os.path.join(OUT, "reference_waveforms.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True)
print("wrote", len(cases), "cases:", {k: {kk: vv["valid"] for kk, vv in v.items()} for k, v in meta.items()}["stanford_RVIO_49_2"])
