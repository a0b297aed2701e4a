"""The threaded gzip-9 chunk writer of hdf5_out against h5py's own create_dataset: same values, dtype,
chunk shape, filter settings -- and the same file size, i.e. the same deflate streams.  h5py lives only in the image's
second interpreter, so the check runs there."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = "/opt/conda/bin/python3.9"

SCRIPT = r"""
import sys, json, os, numpy as np, h5py
sys.path.insert(0, %r)
from tee_optical_flow_amd.hdf5_out import create_gzip9
rng = np.random.default_rng(1)
cases = {"f16_4d": rng.standard_normal((21, 130, 97, 2)).astype(np.float16), "bool_4d": rng.random((21, 130, 97, 2)) < 0.3,
         "f16_3d": rng.random((21, 130, 97)).astype(np.float16), "f64_1d": rng.random(5), "f16_1d_big": rng.random(700001).astype(np.float16),
         "u8_2d": rng.integers(0, 255, (1500, 1501)).astype(np.uint8)}
d = %r
with h5py.File(d + "/ref.h5", "w") as f:
    for k, v in cases.items():
        f.create_dataset(k, data=v, compression="gzip", compression_opts=9)
with h5py.File(d + "/par.h5", "w") as f:
    for k, v in cases.items():
        create_gzip9(f, k, v, min_parallel_bytes=0 if k != "f64_1d" else 1 << 20)
res = {}
with h5py.File(d + "/ref.h5") as a, h5py.File(d + "/par.h5") as b:
    for k in cases:
        res[k] = bool(np.array_equal(a[k][...], b[k][...]) and np.array_equal(b[k][...], cases[k]) and a[k].dtype == b[k].dtype
                      and a[k].chunks == b[k].chunks and a[k].compression == b[k].compression == "gzip"
                      and a[k].compression_opts == b[k].compression_opts == 9 and a[k].shape == b[k].shape)
res["same_size"] = os.path.getsize(d + "/ref.h5") == os.path.getsize(d + "/par.h5")
print(json.dumps(res))
"""


def test_threaded_gzip9_chunks_equal_h5py_create_dataset(tmp_path):
    if not os.path.exists(PY) or subprocess.run([PY, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py available")
    r = subprocess.run([PY, "-c", SCRIPT % (ROOT, str(tmp_path))], capture_output=True, text=True,
                       env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(got.values()), got
