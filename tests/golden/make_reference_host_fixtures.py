#!/opt/conda/bin/python3.9
"""Generates tests/golden/reference_host_side.npz + .json by IMPORTING the reference's own host code
(/root/reference/optical_flow/calculate_optical_flow.py) in THIS container and recording what it returns.

Run (build container only; the reference never travels to the GPU box):
    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_reference_host_fixtures.py

The third-party modules absent here (cv2, pydicom, torch, ...) are replaced by MagicMock stubs, so only the
reference's host-side glue runs -- never the OpenCV solver (which is what this repository re-implements and
whose parity stays UNPINNED, see oracle/tvl1_oracle.c).  What gets pinned:
  * OpticalFlowCalculationConfig defaults                           (config.py:174-188)
  * img2uint8(rgb2gray(x)) frame conditioning                       (calculate_optical_flow.py:588, optical_flow_utils.py:30-31)
  * calculate_optical_flow(): WASE / none background compensation   (calculate_optical_flow.py:627-660) with a stub OF_model
  * moving_avg_mask / predict_movie_thres (Otsu mask path)          (calculate_optical_flow.py:90-111, 184-213)
  * _save_optical_flow_to_hdf5 layout: keys, dtypes, shapes, attrs  (calculate_optical_flow.py:370-475)
Fixtures are DATA (inputs + outputs); no reference source text is stored.
"""
import json
import os
import sys
import tempfile
from unittest.mock import MagicMock

import numpy as np

for m in ["cv2", "pydicom", "torch", "torchvision", "torchvision.transforms", "peakutils", "polars", "tsmoothie",
          "tsmoothie.smoother", "neurokit2", "models", "models.sam", "imageio.v2"]:
    sys.modules[m] = MagicMock()
sys.modules["cv2"].cuda.getCudaEnabledDeviceCount.return_value = 0
sys.path.insert(0, "/root/reference")
import optical_flow.calculate_optical_flow as R  # noqa: E402
from optical_flow.config import default_optical_flow_config  # noqa: E402
from dataclasses import asdict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
arrs, meta = {}, {}

# 1. config defaults
meta["config_defaults"] = asdict(default_optical_flow_config())

# 2. frame conditioning
rng = np.random.default_rng(0)
rgb = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
arrs["cond_in_rgb"] = rgb
arrs["cond_out_u8"] = R.img2uint8(R.rgb2gray(rgb))
ramp = np.repeat((np.arange(48 * 40, dtype=np.uint8).reshape(48, 40) % 200 + 20)[:, :, None], 3, axis=2)  # min > 0: exposes /max (not /(max-min))
arrs["cond_in_ramp"] = ramp
arrs["cond_out_ramp"] = R.img2uint8(R.rgb2gray(ramp))

# 3. calculate_optical_flow(): background compensation with a stub model
flow = rng.normal(0, 2, (24, 32, 2)).astype(np.float32)
flow[3:6, 4:9, :] = 0.0   # exact zeros are excluded from the WASE mean
bk = rng.random((5, 24, 32)) > 0.6
bk2 = np.repeat(bk[:, :, :, None], 2, axis=3)   # masks are bool [N,H,W,2] (channel duplicated)


class StubModel:
    def calc(self, a, b, c):
        return flow.copy()


z = np.zeros((24, 32), np.uint8)
arrs["bg_flow_in"] = flow
arrs["bg_mask"] = bk2
arrs["bg_out_wase"] = np.asarray(R.calculate_optical_flow(z, z, {"bkgd": bk2}, StubModel(), bkgd_comp="WASE", OF_algo="TVL1"))
arrs["bg_out_none"] = np.asarray(R.calculate_optical_flow(z, z, {"bkgd": bk2}, StubModel(), bkgd_comp="none", OF_algo="TVL1"))
arrs["bg_out_none_deepflow"] = np.asarray(R.calculate_optical_flow(z, z, {}, StubModel(), bkgd_comp="none", OF_algo="deepflow"))
errs = {}
for kw in (dict(bkgd_comp="bogus", OF_algo="TVL1"), dict(bkgd_comp="none", OF_algo="bogus")):
    try:
        R.calculate_optical_flow(z, z, {"bkgd": bk2}, StubModel(), **kw)
        errs[json.dumps(kw)] = None
    except Exception as e:  # noqa: BLE001
        errs[json.dumps(kw)] = type(e).__name__
meta["calc_errors"] = errs

# 4. Otsu mask path
m = rng.random((9, 20, 24)) > 0.5
arrs["mavg_in"] = m
arrs["mavg_out"] = R.moving_avg_mask(m)
yy, xx = np.mgrid[0:64, 0:72]
frames = []
for i in range(6):
    blob = 255 * np.exp(-(((xx - 36 - 2 * i) / 18.0) ** 2 + ((yy - 32) / 14.0) ** 2))
    g = np.clip(blob + rng.normal(0, 6, blob.shape), 0, 255).astype(np.uint8)
    frames.append(np.repeat(g[:, :, None], 3, axis=2))
frames = np.stack(frames)
arrs["otsu_in"] = frames
arrs["otsu_out"] = R.predict_movie_thres(frames)["otsu"]

# 5. HDF5 writer layout
import h5py  # noqa: E402


class DS:
    PatientID = "SYNTH-0001"
    HeartRate = 72


N = 6
flow_arr = rng.normal(0, 1, (N, 64, 72, 2)).astype(np.float32)
masks = {"otsu": arrs["otsu_out"]}
md = {"frame_rate": 30.0, "pixel_spacing": 0.05, "R_wave_data_present": True, "R_times": np.array([100.0, 900.0])}
wf = {"ecg": (True, rng.normal(0, 1, 50)), "art": (True, rng.normal(80, 5, 20)), "cvp": (False, None), "pap": (False, None)}
layout = {}
for tag, inc in (("with_waveforms", True), ("no_waveforms", False)):
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "x.h5")
        R._save_optical_flow_to_hdf5(pth, flow_arr, frames, masks, md, wf if inc else {}, DS(), default_optical_flow_config(),
                                     "otsu", True, inc, None, False)
        d = {}
        with h5py.File(pth, "r") as f:
            for k in f.keys():
                ds = f[k]
                d[k] = {"dtype": str(ds.dtype), "shape": list(ds.shape), "compression": ds.compression,
                        "compression_opts": ds.compression_opts,
                        "attrs": {a: [type(v).__name__, str(np.asarray(v).dtype), np.asarray(v).tolist() if np.asarray(v).size < 8 else None]
                                  for a, v in ds.attrs.items()}}
            if tag == "with_waveforms":
                arrs["h5_echo"] = f["echo"][...]
                arrs["h5_flow"] = f["flow"][...]
        layout[tag] = d
meta["hdf5_layout"] = layout
arrs["h5_flow_in"] = flow_arr

np.savez_compressed(os.path.join(OUT, "reference_host_side.npz"), **arrs)
with open(os.path.join(OUT, "reference_host_side.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True, default=str)
print("wrote", {k: v.shape for k, v in arrs.items()})
