#!/usr/bin/env python3
"""Row f3 as a measurement: studies per second through process_folder (reference calculate_optical_flow.py:243-290) on the
GPU box, HDF5 writing included, with the writer thread beside the next solve and without it.  h5py lives in the image's
second interpreter, so run it there:
    LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6 /opt/conda/bin/python3.9 tools/study_throughput.py [--studies 6] [--frames 65] [--size 512]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--studies", type=int, default=6)
    ap.add_argument("--frames", type=int, default=65)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--algo", default="TVL1")
    ap.add_argument("--readers", type=int, default=2)
    ap.add_argument("--writers", type=int, default=2)
    ap.add_argument("--stages", action="store_true", help="also print where the caller's thread spends a study in the worker-process walk")
    a = ap.parse_args()
    from tee_optical_flow_amd import pipeline as P
    from tee_optical_flow_amd import hdf5_out
    from tee_optical_flow_amd.synth import speckle_sequence
    tmp = tempfile.mkdtemp(prefix="teeflow_studies_")
    src = os.path.join(tmp, "in")
    os.makedirs(src)
    for k in range(a.studies):
        g = speckle_sequence(500 + k, a.frames, a.size, a.size)
        np.savez(os.path.join(src, f"study{k:02d}.npz"), nparr=np.repeat(g[..., None], 3, axis=3), pixel_spacing=0.04, frame_rate=50.0, patient_id=f"S{k}")
    # the worker processes must exist before anything in this process touches the GPU
    workers = P.StudyWorkers(a.readers, a.writers)
    model = P.make_flow_model(a.algo)
    kw = dict(nchunks=1, chunk_index=0, mode="otsu", verbose=False, extensions=("npz",), OF_algo=a.algo, flow_model=model)
    P.process_folder(src, os.path.join(tmp, "warm"), None, process_subset=True, file_subset_list=["study00.npz"], workers="thread", **kw)   # warm-up: allocations, masks code paths
    P.process_folder(src, os.path.join(tmp, "warm2"), None, process_subset=True, file_subset_list=["study00.npz", "study01.npz"], workers=workers, **kw)
    # where the caller's thread spends a study in the worker-process walk: solve (flow_for_study), hand-over to the writer (defer),
    # the rest of process_video, and everything outside it (waiting for the reader stage, reaping writers)
    acc = {"begin": 0.0, "finish": 0.0, "defer": 0.0}
    real_begin = P._process_video_begin

    def timed_begin(*args, **kwargs):
        d = kwargs.get("_defer_save")
        if d is not None:
            def timed_defer(job):
                t = time.perf_counter(); d(job); acc["defer"] += time.perf_counter() - t
            kwargs["_defer_save"] = timed_defer
        t = time.perf_counter()
        fin = real_begin(*args, **kwargs)                   # masks are ready; conditions the frames and SUBMITS the solve (studies_in_flight=2)
        acc["begin"] += time.perf_counter() - t

        def timed_finish():
            t2 = time.perf_counter()
            try:
                return fin()                                # waits for the flows, hands the study to the writer stage
            finally:
                acc["finish"] += time.perf_counter() - t2
        return timed_finish
    if a.stages:
        P._process_video_begin = timed_begin
    t0 = time.perf_counter()
    errs_p = P.process_folder(src, os.path.join(tmp, "processes"), None, workers=workers, **kw)
    t_proc = time.perf_counter() - t0
    P._process_video_begin = real_begin
    if a.stages:
        n = a.studies
        print(f"  caller's thread per study (worker-process walk, the next study's solve submitted before this one's is collected): submit (conditioning "
              f"+ queueing) {acc['begin'] / n * 1e3:.1f} ms, collect + hand-over {acc['finish'] / n * 1e3:.1f} ms (of which hand-over to the writer {acc['defer'] / n * 1e3:.1f} ms), "
              f"outside both (reader wait, reaping) {(t_proc - acc['begin'] - acc['finish']) / n * 1e3:.1f} ms")
    workers.close()
    t0 = time.perf_counter()
    errs = P.process_folder(src, os.path.join(tmp, "overlapped"), None, workers="thread", **kw)
    t_overlap = time.perf_counter() - t0
    # the same walk with the HDF5 write done in line (what the reference does): time the writer by making defer synchronous
    real = hdf5_out.save_optical_flow_to_hdf5
    t_write = [0.0]

    def timed(*args, **kwargs):
        t = time.perf_counter(); real(*args, **kwargs); t_write[0] += time.perf_counter() - t
    hdf5_out.save_optical_flow_to_hdf5 = timed
    orig_pf = P.process_folder

    def serial_folder(*args, **kwargs):
        # no writer thread: process_video writes before it returns
        files = sorted(os.listdir(args[0]))
        os.makedirs(args[1], exist_ok=True)
        for f in files:
            nparr, md, pid, hr = P.read_study(os.path.join(args[0], f))
            P.process_video(None, os.path.join(args[1], f[:-4] + ".hdf5"), None, verbose=False, mode="otsu", no_saliency=True, OF_algo=a.algo,
                            nparr=nparr, metadata=md, patient_id=pid, heart_rate=hr, flow_model=model)
    t0 = time.perf_counter()
    serial_folder(src, os.path.join(tmp, "serial"))
    t_serial = time.perf_counter() - t0
    hdf5_out.save_optical_flow_to_hdf5 = real
    model.close()
    sz = sum(os.path.getsize(os.path.join(tmp, "overlapped", f)) for f in os.listdir(os.path.join(tmp, "overlapped"))) / a.studies / 1e6
    print(f"{a.algo}: {a.studies} studies of {a.frames} frames {a.size}x{a.size} (errors: {errs})")
    print(f"  {a.readers} reader + {a.writers} writer PROCESSES      : {t_proc:6.2f} s = {t_proc / a.studies * 1e3:7.1f} ms per study, {a.studies * (a.frames - 1) / t_proc:7.1f} pairs/s end to end (errors: {errs_p})")
    print(f"  reader / writer threads            : {t_overlap:6.2f} s = {t_overlap / a.studies * 1e3:7.1f} ms per study, {a.studies * (a.frames - 1) / t_overlap:7.1f} pairs/s end to end")
    print(f"  write in line (reference order)    : {t_serial:6.2f} s = {t_serial / a.studies * 1e3:7.1f} ms per study, of which HDF5 gzip-9 write {t_write[0] / a.studies * 1e3:7.1f} ms; file {sz:.1f} MB per study")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
