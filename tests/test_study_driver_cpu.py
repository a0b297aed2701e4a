"""process_folder (reference calculate_optical_flow.py:243-290) and the segmentor plumbing (:47-88, :215-241) on the CPU.
The flow model is a TEST DOUBLE (the product's model has no CPU path); h5py lives only in the image's second interpreter,
so the folder walk runs there."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY_H5 = "/opt/conda/bin/python3.9"

DRIVER = r"""
import sys, json, os, numpy as np
sys.path.insert(0, ROOT)
import h5py
from tee_optical_flow_amd.pipeline import process_folder
from tee_optical_flow_amd.synth import speckle_sequence

class FakeModel:                        # cv2-protocol stand-in, tests only: the walk's rules do not depend on the flow values
    def __init__(self): self.studies = 0
    def calc_batch(self, frames, scale=1.0):
        self.studies += 1
        d = (frames[1:].astype(np.float32) - frames[:-1].astype(np.float32)) / 64
        return np.stack([d, -0.5 * d], -1) * np.float32(scale)
    def close(self): pass

src, dst = os.path.join(TMP, "in"), os.path.join(TMP, "out")
os.makedirs(src)
for k in range(5):                       # five studies: .npz with metadata, .npy without, one corrupt, one foreign extension
    g = speckle_sequence(100 + k, 4, 40, 48)
    nparr = np.repeat(g[..., None], 3, axis=3)
    if k == 1:
        np.save(os.path.join(src, f"s{k}.npy"), nparr)
    elif k == 3:
        open(os.path.join(src, f"s{k}.npz"), "wb").write(b"not a zip file")
    else:
        np.savez(os.path.join(src, f"s{k}.npz"), nparr=nparr, pixel_spacing=0.05, frame_rate=40.0, patient_id=f"P{k}", heart_rate=60 + k)
open(os.path.join(src, "notes.txt"), "w").write("x")
kw = dict(nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz", "npy"), OF_algo="TVL1")
m = FakeModel()
err1 = process_folder(src, dst, None, flow_model=m, **kw)
files1 = sorted(os.listdir(dst))
solved1 = m.studies
err2 = process_folder(src, dst, None, flow_model=m, **kw)             # everything that succeeded is skipped now
solved2 = m.studies - solved1
err3 = process_folder(src, dst, None, flow_model=m, recalculate=True, process_subset=True, file_subset_list=["s0.npz"], **kw)
solved3 = m.studies - solved1 - solved2
# two ranks deal the slice round-robin; chunking drops the remainder like the reference
dst2 = os.path.join(TMP, "out2")
r0 = process_folder(src, dst2, None, flow_model=m, rank=0, world=2, **kw)
got_r0 = sorted(os.listdir(dst2))
r1 = process_folder(src, dst2, None, flow_model=m, rank=1, world=2, **kw)
got_all = sorted(os.listdir(dst2))
dst3 = os.path.join(TMP, "out3")
process_folder(src, dst3, None, flow_model=m, nchunks=2, chunk_index=1, mode="otsu", verbose=False, extensions=("npz", "npy"))
lay = {}
with h5py.File(os.path.join(dst, "s0.hdf5"), "r") as f:
    for k in f.keys():
        ds = f[k]
        lay[k] = {"dtype": str(ds.dtype), "shape": list(ds.shape), "compression": ds.compression, "compression_opts": ds.compression_opts,
                  "attrs": {n: [type(v).__name__, str(np.asarray(v).dtype)] for n, v in ds.attrs.items()}}
    a0 = {n: (v.tolist() if hasattr(v, "tolist") else v) for n, v in f["flow"].attrs.items()}
    dup = bool(np.array_equal(f["flow"][-1], f["flow"][-2]))
with h5py.File(os.path.join(dst, "s1.hdf5"), "r") as f:
    a1 = {n: (v.tolist() if hasattr(v, "tolist") else v) for n, v in f["flow"].attrs.items()}
print(json.dumps({"err1": err1, "files1": files1, "solved": [solved1, solved2, solved3], "err2": err2, "err3": err3,
                  "got_r0": got_r0, "got_all": got_all, "r0": r0, "r1": r1, "chunk1of2": sorted(os.listdir(dst3)),
                  "layout": lay, "attrs_s0": a0, "attrs_s1": a1, "last_duplicated": dup}, default=str))
"""


def test_process_folder_rules(tmp_path):
    if not os.path.exists(PY_H5):
        pytest.skip("no interpreter with h5py")
    script = DRIVER.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path)))
    r = subprocess.run([PY_H5, "-c", script], capture_output=True, text=True, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])
    # per-study isolation: the corrupt study is reported, the others are written; the foreign extension is skipped silently
    assert [e[0] for e in g["err1"]] == ["s3.npz"]
    assert g["files1"] == ["s0.hdf5", "s1.hdf5", "s2.hdf5", "s4.hdf5"]           # complete files only: no *.part* leftovers
    assert g["solved"] == [4, 0, 1]                                              # skip-if-exists, then recalculate on a subset
    assert [e[0] for e in g["err2"]] == ["s3.npz"] and g["err3"] == []
    # ranks: listing sorted = [notes.txt, s0, s1, s2, s3, s4]; rank 0 takes indices 0,2,4 (notes, s1, s3), rank 1 takes 1,3,5
    assert g["got_r0"] == ["s1.hdf5"] and g["got_all"] == ["s0.hdf5", "s1.hdf5", "s2.hdf5", "s4.hdf5"]
    assert [e[0] for e in g["r0"]] == ["s3.npz"] and g["r1"] == []
    # nchunks=2: split = 6 // 2 = 3 files per chunk; chunk 1 = [s2, s3, s4]
    assert g["chunk1of2"] == ["s2.hdf5", "s4.hdf5"]
    # layout of a written study = the reference writer's (keys, dtypes, gzip-9, attr names and types)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_host_side.json")))["hdf5_layout"]["no_waveforms"]
    assert set(g["layout"]) == set(ref) - {"RWaveTime"}
    for k in g["layout"]:
        assert g["layout"][k]["dtype"] == ref[k]["dtype"] and g["layout"][k]["compression"] == "gzip" and g["layout"][k]["compression_opts"] == 9
        assert {n: v[:2] for n, v in ref[k]["attrs"].items()} == {n: v for n, v in g["layout"][k]["attrs"].items()} or k == "flow"
    assert set(g["layout"]["flow"]["attrs"]) == set(ref["flow"]["attrs"])
    assert g["layout"]["flow"]["shape"] == [4, 40, 48, 2] and g["last_duplicated"]
    assert g["attrs_s0"]["units_converted"] is True and g["attrs_s0"]["frame_rate"] == 40.0 and g["attrs_s0"]["ID"] == "P0"
    # a study without metadata: same attribute set, NaN values, units_converted False (the reference raises in h5py here)
    assert g["attrs_s1"]["units_converted"] is False and np.isnan(g["attrs_s1"]["frame_rate"]) and np.isnan(g["attrs_s1"]["pixel_spacing"])


class _FakeSam:
    """Stand-in for the SAM module of the reference (checkpoint, timm, torchvision are absent): the three sub-modules
    evaluate_1_slice calls, producing class 1 where the (normalised) red channel is bright, class 2 in a fixed square."""

    def __init__(self):
        import torch

        class Enc(torch.nn.Module):
            def forward(self, x):
                return x

        class Prompt(torch.nn.Module):
            def forward(self, points=None, boxes=None, masks=None):
                return None, None

            def get_dense_pe(self):
                return None

        class Dec(torch.nn.Module):
            def forward(self, image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output):
                x = image_embeddings[:, 0]                              # [1,1024,1024]
                c1 = (x > 0.2).float()
                c2 = torch.zeros_like(x)
                c2[:, 300:700, 300:700] = 2.0
                logits = torch.stack([torch.full_like(x, 0.5), c1, c2], dim=1)
                return logits, None

        self.image_encoder, self.prompt_encoder, self.mask_decoder = Enc(), Prompt(), Dec()

    def parameters(self):
        return iter(())


def test_segmentor_plumbing_runs_through_process_video(oracle):
    """config 5's mask step without mask_dict= injection: a stand-in module goes through evaluate_1_slice / predict_movie /
    clean_mask, and its 'bkgd' mask feeds the WASE compensation."""
    from tee_optical_flow_amd import masks
    from tee_optical_flow_amd.pipeline import process_video
    from tee_optical_flow_amd.synth import speckle_sequence
    from tests.test_pipeline_cpu import OracleModel
    g = speckle_sequence(3, 5, 64, 64)
    nparr = np.repeat(g[..., None], 3, axis=3)
    sam = _FakeSam()
    cls = masks.evaluate_1_slice(nparr[0], sam)
    assert cls.shape == (64, 64) and cls.dtype == np.uint8 and set(np.unique(cls)) <= {0, 1, 2} and (cls == 2).any()
    md = masks.predict_movie(nparr, sam, mode="RVIO_2class")
    assert set(md) == {"rv", "av", "bkgd"} and all(v.shape == (5, 64, 64, 2) and v.dtype == bool for v in md.values())
    assert not (md["bkgd"] & (md["rv"] | md["av"])).any()
    cfg = type("C", (), {"min_mask_size": 10 ** 9})()                    # clean_mask honours config.min_mask_size
    assert not masks.clean_mask(np.ones((4, 16, 16), np.uint8), "RVIO_2class", config=cfg)["rv"].any()
    assert masks.clean_mask(np.zeros((3, 8, 8), np.uint8), "nonsense") is None
    m = OracleModel(oracle)
    flow = process_video(None, None, sam, verbose=False, mode="RVIO_2class", bkgd_comp="WASE", no_saliency=True, nparr=nparr, flow_model=m)
    raw = process_video(None, None, sam, verbose=False, mode="RVIO_2class", bkgd_comp="none", no_saliency=True, nparr=nparr, flow_model=m)
    assert flow.shape == (5, 64, 64, 2) and not np.array_equal(flow, raw)
    from tee_optical_flow_amd.exceptions import ConfigurationError
    with pytest.raises(ConfigurationError):
        process_video(None, None, None, mode="A4C", no_saliency=True, nparr=nparr, flow_model=m)


PROC_DRIVER = r"""
import sys, json, os, numpy as np
sys.path.insert(0, ROOT)
import h5py
from tee_optical_flow_amd.pipeline import process_folder
from tee_optical_flow_amd.synth import speckle_sequence

class FakeModel:                        # cv2-protocol stand-in, tests only
    def calc_batch(self, frames, scale=1.0):
        if frames.shape[1] == 40:
            raise RuntimeError("solver failure injected for 40-row studies")
        d = (frames[1:].astype(np.float32) - frames[:-1].astype(np.float32)) / 64
        return np.stack([d, -0.5 * d], -1) * np.float32(scale)
    def close(self): pass

if __name__ == "__main__":
    src = os.path.join(TMP, "in"); os.makedirs(src)
    for k in range(4):
        g = speckle_sequence(300 + k, 5, 48, 56)
        np.savez(os.path.join(src, f"s{k}.npz"), nparr=np.repeat(g[..., None], 3, axis=3), pixel_spacing=0.05, frame_rate=40.0, patient_id=f"P{k}", heart_rate=70)
    g = speckle_sequence(9, 4, 40, 64)                  # a study whose SOLVE fails: its shared-memory blocks must go too
    np.savez(os.path.join(src, "s8.npz"), nparr=np.repeat(g[..., None], 3, axis=3), pixel_spacing=0.05, frame_rate=40.0)
    open(os.path.join(src, "s9.npz"), "wb").write(b"broken")
    kw = dict(nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz",), flow_model=FakeModel())
    e1 = process_folder(src, os.path.join(TMP, "thr"), None, workers="thread", **kw)
    shm_before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    e2 = process_folder(src, os.path.join(TMP, "prc"), None, workers="process", n_readers=2, n_writers=2, **kw)
    shm_left = sorted((set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()) - shm_before)
    from tee_optical_flow_amd import pipeline
    same = True
    for k in range(4):
        with h5py.File(os.path.join(TMP, "thr", f"s{k}.hdf5"), "r") as a, h5py.File(os.path.join(TMP, "prc", f"s{k}.hdf5"), "r") as b:
            same &= sorted(a.keys()) == sorted(b.keys())
            for key in a.keys():
                same &= a[key].dtype == b[key].dtype and a[key].shape == b[key].shape and bool(np.array_equal(a[key][...], b[key][...]))
                same &= a[key].compression == b[key].compression and a[key].compression_opts == b[key].compression_opts and a[key].chunks == b[key].chunks
            for n, v in a["flow"].attrs.items():
                w = b["flow"].attrs[n]
                same &= bool(np.array_equal(np.asarray(v), np.asarray(w))) and type(v) is type(w)
    print(json.dumps({"e1": e1, "e2": e2, "same": bool(same), "files": sorted(os.listdir(os.path.join(TMP, "prc"))),
                      "shm_left": shm_left, "shm": pipeline._shm_stats}, default=str))
"""


def test_process_folder_worker_processes_write_the_same_files(tmp_path):
    """workers='process': the reader/mask stage and the deflate/write stage run in spawned worker processes (the echo dataset is made in
    the reader stage, float16 flow travels to the writer, everything big through shared memory); datasets, filters, chunks and attributes equal the thread form's, a broken
    study is reported the same way."""
    if not os.path.exists(PY_H5):
        pytest.skip("no interpreter with h5py")
    script = tmp_path / "drv.py"
    script.write_text(PROC_DRIVER.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path))))
    # 1 KB threshold: these small studies' frames, masks, echo and flow all travel as shared-memory blocks, as real studies' do
    r = subprocess.run([PY_H5, str(script)], capture_output=True, text=True,
                       env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1", "TEEFLOW_SHM_MIN_BYTES": "1024"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert "leaked shared_memory" not in r.stderr, r.stderr[-2000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])
    assert g["same"] is True
    assert g["files"] == ["s0.hdf5", "s1.hdf5", "s2.hdf5", "s3.hdf5"]
    assert [e[0] for e in g["e1"]] == ["s8.npz", "s9.npz"] and [e[0] for e in g["e2"]] == ["s8.npz", "s9.npz"]
    assert "injected" in g["e2"][0][1]
    assert g["shm_left"] == [], "shared-memory blocks left behind"
    # per study: frames + otsu mask + echo mapped, float16 flow created; the study whose solve fails maps its three and creates none
    assert g["shm"]["mapped"] == 5 * 3 and g["shm"]["created"] == 4


UNSAFE_DRIVER = r"""
import sys, json, os, logging, numpy as np
sys.path.insert(0, ROOT)
from tee_optical_flow_amd import pipeline
from tee_optical_flow_amd.pipeline import process_folder, read_study
from tee_optical_flow_amd.synth import speckle_sequence

class FakeModel:                        # cv2-protocol stand-in, tests only
    def calc_batch(self, frames, scale=1.0):
        d = (frames[1:].astype(np.float32) - frames[:-1].astype(np.float32)) / 64
        return np.stack([d, -0.5 * d], -1) * np.float32(scale)
    def close(self): pass

def dying_reader(path):                 # picklable, but kills every worker PROCESS that runs it; in a thread of the caller it reads
    import multiprocessing
    if multiprocessing.parent_process() is not None:
        os._exit(3)
    return read_study(path)

warnings = []
class Grab(logging.Handler):
    def emit(self, rec):
        if rec.levelno >= logging.WARNING: warnings.append(rec.getMessage())
pipeline.logger.addHandler(Grab())

def walk(tag, reader):
    src, dst = os.path.join(TMP, "in"), os.path.join(TMP, tag)
    if not os.path.isdir(src):
        os.makedirs(src)
        for k in range(3):
            g = speckle_sequence(300 + k, 4, 40, 48)
            np.savez(os.path.join(src, f"s{k}.npz"), nparr=np.repeat(g[..., None], 3, axis=3), pixel_spacing=0.05, frame_rate=40.0)
    del warnings[:]
    errs = process_folder(src, dst, None, nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz",), flow_model=FakeModel(),
                          workers="process", n_readers=2, n_writers=2, reader=reader)
    return {"errors": errs, "files": sorted(os.listdir(dst)), "warnings": list(warnings)}

MODE = sys.argv[1]
if MODE == "unguarded":
    # module-level call, no __main__ guard, a lambda reader: the documented hook used the careless way
    print(json.dumps(walk("out_u", lambda p: read_study(p))))
if __name__ == "__main__":
    if MODE == "lambda":
        print(json.dumps(walk("out_l", lambda p: read_study(p))))
    elif MODE == "dying":
        print(json.dumps(walk("out_d", dying_reader)))
    elif MODE == "fine":
        print(json.dumps(walk("out_f", read_study)))
"""


@pytest.mark.parametrize("mode,expect", [("unguarded", "does not pickle"), ("lambda", "does not pickle"), ("dying", "worker processes failed"), ("fine", None)])
def test_process_folder_falls_back_to_threads_when_worker_processes_cannot_serve(tmp_path, mode, expect):
    """worker processes are spawned, so the caller's __main__ is re-imported and `reader` is pickled.  A script without a
    __main__ guard, a lambda reader or a pool whose workers die must not cost the studies: the stages fall back to threads and say why."""
    if not os.path.exists(PY_H5):
        pytest.skip("no interpreter with h5py")
    script = tmp_path / "walk.py"
    script.write_text(UNSAFE_DRIVER.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path))))
    r = subprocess.run([PY_H5, str(script), mode], capture_output=True, text=True, timeout=300, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])
    assert g["errors"] == [] and g["files"] == ["s0.hdf5", "s1.hdf5", "s2.hdf5"], g
    if expect is None:
        assert not g["warnings"], g["warnings"]
    else:
        assert any(expect in w for w in g["warnings"]), g["warnings"]


def test_unguarded_script_with_a_picklable_reader_is_detected(tmp_path):
    """The other half of the up-front check: the reader pickles, but the call sits in module-level code of a script outside its
    `if __name__ == "__main__":` block."""
    if not os.path.exists(PY_H5):
        pytest.skip("no interpreter with h5py")
    script = tmp_path / "walk2.py"
    src = UNSAFE_DRIVER.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path)))
    src = src.replace('print(json.dumps(walk("out_u", lambda p: read_study(p))))', 'print(json.dumps(walk("out_u", read_study)))')
    script.write_text(src)
    r = subprocess.run([PY_H5, str(script), "unguarded"], capture_output=True, text=True, timeout=300, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])
    assert g["errors"] == [] and g["files"] == ["s0.hdf5", "s1.hdf5", "s2.hdf5"], g
    assert any("outside an `if __name__ == '__main__':` block" in w for w in g["warnings"]), g["warnings"]


ASYNC_DRIVER = r"""
import sys, json, os, numpy as np
sys.path.insert(0, ROOT)
import h5py
from tee_optical_flow_amd.pipeline import process_folder
from tee_optical_flow_amd.synth import speckle_sequence

class AsyncFake:                        # stands in for DenseFlow's study calls (tests only): submit_study / wait like the engine's
    device_unit_scale = True
    def __init__(self): self.log, self.jobs, self.n = [], {}, 0
    def _flow(self, rgb, scale, pad_last):
        g = rgb[..., 0].astype(np.float32)
        d = (g[1:] - g[:-1]) / 64
        f = np.stack([d, -0.5 * d], -1) * np.float32(scale)
        return np.concatenate([f, f[-1:]]) if pad_last else f
    def calc_study(self, rgb, scale=1.0, pad_last=False):
        self.log.append(("calc", rgb.shape[1]))
        if rgb.shape[1] == 40: raise RuntimeError("solver failure injected for 40-row studies")
        return self._flow(rgb, scale, pad_last)
    def submit_study(self, rgb, scale=1.0, pad_last=False):
        self.n += 1
        self.log.append(("submit", rgb.shape[1]))
        self.jobs[self.n] = (rgb.copy(), scale, pad_last)
        return self.n
    def wait(self, t):
        rgb, scale, pad_last = self.jobs.pop(t)
        self.log.append(("wait", rgb.shape[1]))
        if rgb.shape[1] == 40: raise RuntimeError("solver failure injected for 40-row studies")
        return self._flow(rgb, scale, pad_last)
    def close(self): pass

src = os.path.join(TMP, "in"); os.makedirs(src)
rows = [48, 40, 56, 64, 72]                                 # study 1 fails in its solve; every study has its own height, so the log names it
for k, h in enumerate(rows):
    g = speckle_sequence(400 + k, 5, h, 64)
    np.savez(os.path.join(src, f"s{k}.npz"), nparr=np.repeat(g[..., None], 3, axis=3), pixel_spacing=0.05, frame_rate=40.0)
out = {}
for depth in (2, 1, 3):
    m = AsyncFake()
    errs = process_folder(src, os.path.join(TMP, f"out{depth}"), None, nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz",),
                          flow_model=m, workers="thread", no_saliency=True, studies_in_flight=depth)
    out[str(depth)] = {"errors": errs, "files": sorted(os.listdir(os.path.join(TMP, f"out{depth}"))), "log": m.log, "left": len(m.jobs)}
same = True
for k in (0, 2, 3, 4):
    with h5py.File(os.path.join(TMP, "out2", f"s{k}.hdf5"), "r") as a, h5py.File(os.path.join(TMP, "out1", f"s{k}.hdf5"), "r") as b:
        same &= all(bool(np.array_equal(a[key][...], b[key][...])) for key in a.keys())
out["same"] = bool(same)
print(json.dumps(out, default=str))
"""


def test_process_folder_keeps_the_next_solve_in_flight_and_blames_the_right_study(tmp_path):
    """studies_in_flight=2: study k+1's solve is SUBMITTED before study k's flows are collected (submit, submit, wait, submit, wait, ...); a
    solve that fails when it is collected is reported under ITS file name and costs no other study; depth 1 solves study by study through
    the synchronous call; the files are the same either way; nothing stays submitted."""
    if not os.path.exists(PY_H5):
        pytest.skip("no interpreter with h5py")
    script = tmp_path / "async.py"
    script.write_text(ASYNC_DRIVER.replace("ROOT", repr(ROOT)).replace("TMP", repr(str(tmp_path))))
    r = subprocess.run([PY_H5, str(script)], capture_output=True, text=True, timeout=300, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    g = json.loads(r.stdout.strip().splitlines()[-1])
    for depth in ("2", "1", "3"):
        assert [e[0] for e in g[depth]["errors"]] == ["s1.npz"] and "injected" in g[depth]["errors"][0][1], g[depth]["errors"]
        assert g[depth]["files"] == ["s0.hdf5", "s2.hdf5", "s3.hdf5", "s4.hdf5"] and g[depth]["left"] == 0
    assert g["same"]
    assert g["2"]["log"] == [["submit", 48], ["submit", 40], ["wait", 48], ["submit", 56], ["wait", 40], ["submit", 64], ["wait", 56],
                             ["submit", 72], ["wait", 64], ["wait", 72]]
    assert g["1"]["log"] == [["calc", h] for h in (48, 40, 56, 64, 72)]
    assert g["3"]["log"][:4] == [["submit", 48], ["submit", 40], ["submit", 56], ["wait", 48]]
