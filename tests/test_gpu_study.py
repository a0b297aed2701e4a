"""The study driver on the GPU box (BASELINE.json configs[4]): process_folder over a 3-study temp folder
with the real HIP engine, HDF5 layout against the reference-generated fixture; and config 5's mask step through a stand-in
segmentor module (SAM's checkpoint, timm and torchvision are absent)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY_H5 = "/opt/conda/bin/python3.9"

SCRIPT = r"""
import sys, json, os, numpy as np
sys.path.insert(0, ROOT)
import h5py
from tee_optical_flow_amd.pipeline import process_folder, process_video
from tee_optical_flow_amd.synth import speckle_sequence
src, dst = os.path.join(TMP, "in"), os.path.join(TMP, "out")
os.makedirs(src)
studies = {}
for k in range(3):
    g = speckle_sequence(200 + k, 6, 128, 160)
    studies[f"st{k}"] = np.repeat(g[..., None], 3, axis=3)
    np.savez(os.path.join(src, f"st{k}.npz"), nparr=studies[f"st{k}"], pixel_spacing=0.04, frame_rate=50.0, patient_id=f"SYN{k}", heart_rate=60)
open(os.path.join(src, "bad.npz"), "wb").write(b"garbage")
errs = process_folder(src, dst, None, nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz",), OF_algo="TVL1")
# (default: studies_in_flight=2 -- a study's solve is submitted and collected after the next one's submission; 1 = study by study)
errs1 = process_folder(src, dst + "1", None, nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz",), OF_algo="TVL1", studies_in_flight=1)
out = {"errors": errs, "errors1": errs1, "files": sorted(os.listdir(dst)), "ok": True, "layout": None, "same1": True}
for name in studies:
    with h5py.File(os.path.join(dst, name + ".hdf5"), "r") as a, h5py.File(os.path.join(dst + "1", name + ".hdf5"), "r") as b:
        out["same1"] = out["same1"] and sorted(a.keys()) == sorted(b.keys()) and all(bool(np.array_equal(a[k][...], b[k][...])) for k in a.keys())
for name, nparr in studies.items():
    md = {"pixel_spacing": 0.04, "frame_rate": 50.0, "R_wave_data_present": False, "R_times": None}
    ref = process_video(None, None, None, verbose=False, mode="otsu", no_saliency=True, nparr=nparr, metadata=md)
    with h5py.File(os.path.join(dst, name + ".hdf5"), "r") as f:
        out["ok"] = out["ok"] and bool(np.array_equal(f["flow"][...], ref.astype(np.float16))) and f["flow"].attrs["ID"] == "SYN" + name[-1]
        if out["layout"] is None:
            out["layout"] = {k: {"dtype": str(f[k].dtype), "shape": list(f[k].shape), "compression": f[k].compression,
                                 "compression_opts": f[k].compression_opts, "attrs": sorted(f[k].attrs.keys())} for k in f.keys()}
print(json.dumps(out, default=str))
"""


def test_process_folder_three_studies_on_gpu(tmp_path):
    if not os.path.exists(PY_H5):
        pytest.skip("no interpreter with h5py")
    env = {**os.environ, "PYTHONDONTWRITEBYTECODE": "1"}
    sys_stdcpp = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"       # conda ships an older libstdc++ than libamdhip64 needs
    if os.path.exists(sys_stdcpp):
        env["LD_PRELOAD"] = sys_stdcpp
    script = SCRIPT.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path)))
    r = subprocess.run([PY_H5, "-c", script], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])
    assert [e[0] for e in g["errors"]] == ["bad.npz"]                      # isolated, reported, the walk went on
    assert g["files"] == ["st0.hdf5", "st1.hdf5", "st2.hdf5"] and g["ok"]
    assert g["same1"] and [e[0] for e in g["errors1"]] == ["bad.npz"]     # studies in flight or one by one: the same files
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_host_side.json")))["hdf5_layout"]["no_waveforms"]
    assert set(g["layout"]) == set(ref) - {"RWaveTime"}
    for k, v in g["layout"].items():
        assert v["dtype"] == ref[k]["dtype"] and v["compression"] == "gzip" and v["compression_opts"] == 9
        assert v["attrs"] == sorted(ref[k]["attrs"].keys())
    assert g["layout"]["flow"]["shape"] == [6, 128, 160, 2] and g["layout"]["echo"]["shape"] == [6, 128, 160]


def test_config5_with_stand_in_segmentor_on_gpu():
    """process_video(mode='RVIO_2class', bkgd_comp='WASE') with segmentor_model= a stand-in nn.Module: masks come from
    evaluate_1_slice / predict_movie / clean_mask, the flow and the WASE compensation from the HIP engine."""
    from tee_optical_flow_amd import masks
    from tee_optical_flow_amd.pipeline import process_video, wase_background
    from tee_optical_flow_amd.synth import speckle_sequence
    from tests.test_study_driver_cpu import _FakeSam
    g = speckle_sequence(9, 6, 128, 128)
    nparr = np.repeat(g[..., None], 3, axis=3)
    sam = _FakeSam()
    md = {"pixel_spacing": 0.04, "frame_rate": 50.0, "R_wave_data_present": False, "R_times": None}
    out = process_video(None, None, sam, verbose=False, mode="RVIO_2class", bkgd_comp="WASE", no_saliency=True, nparr=nparr, metadata=md)
    raw = process_video(None, None, sam, verbose=False, mode="RVIO_2class", bkgd_comp="none", no_saliency=True, nparr=nparr, metadata=md)
    mask = masks.predict_movie(nparr, sam, mode="RVIO_2class")["bkgd"]
    assert out.shape == (6, 128, 128, 2) and out.dtype == np.float32
    cf = np.float32(0.04 * 50.0)
    for i in range(5):
        f = raw[i] / cf                                                   # the engine's pixel flow (scale is applied last)
        assert np.array_equal(out[i], (f - wase_background(f, mask)) * cf)


def test_config5_full_chain_256_n32():
    """BASELINE.json configs[4] end to end at the size BASELINE.md section 3 names (synthetic study, N = 32, 256 x 256, seed 1000):
    stand-in segmentor -> masks -> device frame conditioning + DualTVL1 (one batched call) -> pad + unit scale -> the float16
    the HDF5 file stores -> radial / longitudinal projection + per-frame percentiles on the device, against the host (numpy)
    restatement of the reference's analysis."""
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd import analysis as A, masks
    from tee_optical_flow_amd.pipeline import process_video
    from tee_optical_flow_amd.synth import speckle_sequence
    from tests.test_study_driver_cpu import _FakeSam
    N, H, W = 32, 256, 256
    g = speckle_sequence(1000, N, H, W)
    nparr = np.repeat(g[..., None], 3, axis=3)
    md = {"pixel_spacing": 0.05, "frame_rate": 40.0, "R_wave_data_present": False, "R_times": None}
    sam = _FakeSam()
    flow = process_video(None, None, sam, verbose=False, mode="RVIO_2class", bkgd_comp="none", no_saliency=True, nparr=nparr, metadata=md)
    assert flow.shape == (N, H, W, 2) and flow.dtype == np.float32 and np.array_equal(flow[-1], flow[-2])
    # the solver really ran on every pair: a study of a moving texture has motion everywhere
    assert (np.abs(flow[:-1]).reshape(N - 1, -1).max(axis=1) > 0.5).all()
    stored = flow.astype(np.float16).astype(np.float32)                  # what OpticalFlowDataset reads back (optical_flow_dataset.py:57)
    av = masks.predict_movie(nparr, sam, mode="RVIO_2class")["av"][..., 0]
    cent = []
    for i in range(N):
        ys, xs = np.nonzero(av[i])
        cent.append((float(ys.mean()), float(xs.mean())) if len(ys) else (H / 2.0, W / 2.0))
    eng = T.DenseFlow(device_id=0)
    try:
        dev = A.radlong_stats_device(eng, stored, cent, return_arrays=True)
    finally:
        eng.close()
    rad, lon = A.calculate_comp_magnitude(stored, cent)
    assert np.array_equal(dev["rad_arr"], rad) and np.array_equal(dev["long_arr"], lon)
    for key, arr in (("radial", rad), ("longitudinal", lon)):
        f, e, hi, lo = A.calc_bidirectional_hist(arr, N)
        df, de, dhi, dlo = dev[key]
        assert np.array_equal(df, f) and np.array_equal(de, np.asarray(e)[:-1]) and np.array_equal(dhi, hi) and np.array_equal(dlo, lo)


def test_unit_scale_and_last_flow_on_the_device_equal_the_host_order(engine):
    """flow_for_study's device path (scale in the output kernel, last flow repeated inside the pinned buffer) gives the bits of the
    reference's order: stack, append the last flow (:599), multiply by pixel_spacing * frame_rate (:600)."""
    from tee_optical_flow_amd.pipeline import flow_for_study
    from tee_optical_flow_amd.synth import speckle_sequence
    seq = speckle_sequence(77, 6, 96, 112)
    rgb = np.ascontiguousarray(np.repeat(seq[..., None], 3, axis=3))
    for cf in (1.0, 0.0371 * 47.0, 3.0e-3):
        fast = flow_for_study(None, engine, None, "none", cf, nparr_rgb=rgb)
        flows = engine.calc_study(rgb)
        slow = np.concatenate([flows, flows[-1:]], axis=0) * cf
        assert fast.dtype == np.float32 and fast.shape == (6, 96, 112, 2)
        assert np.array_equal(fast, slow)
