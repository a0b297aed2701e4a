"""Golden fixtures (CPU only).
 * tests/golden/reference_host_side.*  -- produced by RUNNING the reference's own host code (see
   make_reference_host_fixtures.py): pins this package's restatement of rows a1, a7, a9, the Otsu mask path and the
   HDF5 layout of SURVEY.md section 8.
 * tests/golden/oracle_regression.npz  -- flows of the oracle on small seeded pairs (make_oracle_regression.py):
   a REGRESSION pin of the restatement (self-generated, therefore NOT evidence of OpenCV parity)."""
import json
import os
from dataclasses import asdict

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(G, "reference_host_side.npz")), json.load(open(os.path.join(G, "reference_host_side.json")))


def test_config_defaults_match_reference(ref):
    import tee_optical_flow_amd as T
    assert asdict(T.default_optical_flow_config()) == ref[1]["config_defaults"]


def test_exception_hierarchy_matches_reference():
    import tee_optical_flow_amd as T
    for n in ("DICOMReadError", "WaveformLoadError", "WaveformValidationError", "OpticalFlowCalculationError", "ConfigurationError"):
        assert issubclass(getattr(T, n), T.OpticalFlowError)
    assert issubclass(T.OpticalFlowError, Exception)


def test_frame_conditioning_bit_exact(ref):
    from tee_optical_flow_amd.frames import img2uint8, rgb2gray, condition_frames
    a = ref[0]
    assert np.array_equal(img2uint8(rgb2gray(a["cond_in_rgb"])), a["cond_out_u8"])
    assert np.array_equal(img2uint8(rgb2gray(a["cond_in_ramp"])), a["cond_out_ramp"])      # the /max (not /(max-min)) quirk
    assert np.array_equal(condition_frames(a["cond_in_rgb"][None])[0], a["cond_out_u8"])


def test_background_compensation_matches_reference(ref):
    from tee_optical_flow_amd import OpticalFlowCalculationError
    from tee_optical_flow_amd.pipeline import calculate_optical_flow
    a, meta = ref

    class Stub:
        def calc(self, i0, i1, f):
            return a["bg_flow_in"].copy()

    z = np.zeros((24, 32), np.uint8)
    md = {"bkgd": a["bg_mask"]}
    assert np.array_equal(calculate_optical_flow(z, z, md, Stub(), bkgd_comp="WASE"), a["bg_out_wase"])
    assert np.array_equal(calculate_optical_flow(z, z, md, Stub(), bkgd_comp="none"), a["bg_out_none"])
    assert np.array_equal(calculate_optical_flow(z, z, {}, Stub(), bkgd_comp="none", OF_algo="deepflow"), a["bg_out_none_deepflow"])
    for kw, exc in meta["calc_errors"].items():
        assert exc == "OpticalFlowCalculationError"
        with pytest.raises(OpticalFlowCalculationError):
            calculate_optical_flow(z, z, md, Stub(), **json.loads(kw))


def test_otsu_mask_path_matches_reference(ref):
    from tee_optical_flow_amd.masks import moving_avg_mask, predict_movie_thres
    a = ref[0]
    assert np.array_equal(moving_avg_mask(a["mavg_in"]), a["mavg_out"])
    out = predict_movie_thres(a["otsu_in"])
    assert list(out) == ["otsu"] and out["otsu"].dtype == bool
    assert np.array_equal(out["otsu"], a["otsu_out"])


def test_oracle_regression_vectors(oracle):
    z = np.load(os.path.join(G, "oracle_regression.npz"))
    n = int(z["n"])
    for i in range(n):
        f, it, nl = oracle.tvl1_calc(z[f"I0_{i}"], z[f"I1_{i}"], return_iters=True)
        assert np.array_equal(it[:nl], z[f"iters_{i}"])
        assert np.array_equal(f, z[f"flow_{i}"])


def test_hdf5_writer_layout_matches_reference(ref, tmp_path):
    """BASELINE config 5: same keys / dtypes / shapes / filters / attr names+types as the reference writer, and the same
    float16 payload.  h5py lives only in the image's second interpreter, so the writer runs there."""
    import subprocess
    import sys
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        try:
            import h5py  # noqa: F401
            py = sys.executable
        except ImportError:
            pytest.skip("no interpreter with h5py available")
    root = os.path.dirname(G[:-len("/golden")])
    script = f"""
import sys, json, numpy as np
sys.path.insert(0, {root!r})
import h5py
from tee_optical_flow_amd.hdf5_out import save_optical_flow_to_hdf5
from tee_optical_flow_amd.config import default_optical_flow_config
a = np.load({os.path.join(G, 'reference_host_side.npz')!r})
rng = np.random.default_rng(0)
md = {{"frame_rate": 30.0, "pixel_spacing": 0.05, "R_wave_data_present": True, "R_times": np.array([100.0, 900.0])}}
wf = {{"ecg": (True, rng.normal(0, 1, 50)), "art": (True, rng.normal(80, 5, 20)), "cvp": (False, None), "pap": (False, None)}}
out = {{}}
for tag, inc in (("with_waveforms", True), ("no_waveforms", False)):
    p = {str(tmp_path)!r} + "/" + tag + ".h5"
    save_optical_flow_to_hdf5(p, a["h5_flow_in"], a["otsu_in"], {{"otsu": a["otsu_out"]}}, md, wf if inc else {{}}, "SYNTH-0001", 72,
                              default_optical_flow_config(), "otsu", True, inc, None)
    d = {{}}
    with h5py.File(p, "r") as f:
        for k in f.keys():
            ds = f[k]
            d[k] = {{"dtype": str(ds.dtype), "shape": list(ds.shape), "compression": ds.compression, "compression_opts": ds.compression_opts,
                    "attrs": {{n: [type(v).__name__, str(np.asarray(v).dtype), np.asarray(v).tolist() if np.asarray(v).size < 8 else None]
                              for n, v in ds.attrs.items()}}}}
        if inc:
            ok = bool(np.array_equal(f["echo"][...], a["h5_echo"]) and np.array_equal(f["flow"][...], a["h5_flow"]))
    out[tag] = d
print(json.dumps({{"layout": out, "payload_equal": ok}}, default=str))
"""
    r = subprocess.run([py, "-c", script], capture_output=True, text=True, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["payload_equal"]
    assert got["layout"] == json.loads(json.dumps(ref[1]["hdf5_layout"]))
