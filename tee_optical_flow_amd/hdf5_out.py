"""HDF5 output in the reference's layout (/root/reference/optical_flow/calculate_optical_flow.py:370-475)
so optical_flow_dataset.py / analysis.py / peak_detection.py consume the file unchanged.  h5py is an optional import
(absent from the default interpreter of this image)."""
import itertools
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .exceptions import OpticalFlowError
from .frames import rgb2gray


def _workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def create_gzip9(f, name, data, min_parallel_bytes=1 << 20):
    """`f.create_dataset(name, data=data, compression="gzip", compression_opts=9)` (reference :405-472) with the deflate
    work spread over threads: same dataset for any reader (dtype, shape, h5py's auto chunk shape, filter pipeline, values),
    but the chunks are compressed with zlib level 9 in a thread pool (zlib releases the GIL) and handed to HDF5 with
    `write_direct_chunk`.  gzip-9 of a study's float16 flow is the slowest step of process_video once the flow itself
    takes milliseconds (63 MB: 2.4 s in h5py, 0.4 s here on 8 cores).  The effort is the reference's, always: the filter header says
    gzip 9 and the chunk bytes are level-9 deflate streams (round 4's opt-in lower effort left that contract and is gone)."""
    data = np.asarray(data)
    ds = f.create_dataset(name, shape=data.shape, dtype=data.dtype, compression="gzip", compression_opts=9)
    ch = ds.chunks
    if data.nbytes < min_parallel_bytes or ch is None or data.ndim == 0 or not hasattr(ds.id, "write_direct_chunk"):
        ds[...] = data
        return ds
    offsets = list(itertools.product(*[range(0, s, c) for s, c in zip(data.shape, ch)]))

    def deflate(batch):
        out = []
        for off in batch:
            blk = data[tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, ch, data.shape))]
            if blk.shape != ch:                                   # edge chunk: HDF5 stores full chunks
                full = np.zeros(ch, data.dtype)
                full[tuple(slice(0, n) for n in blk.shape)] = blk
                blk = full
            out.append((off, zlib.compress(np.ascontiguousarray(blk), 9)))
        return out

    group = 16
    with ThreadPoolExecutor(_workers()) as pool:
        for res in pool.map(deflate, [offsets[i:i + group] for i in range(0, len(offsets), group)]):
            for off, comp in res:
                ds.id.write_direct_chunk(off, comp)
    return ds


def save_optical_flow_to_hdf5(save_path, flow_arr, nparr, mask_dict, metadata, waveforms, patient_id, heart_rate, config,
                              mode, no_saliency, include_waveforms, save_mask_subset=None, *, echo=None, nframes=None):
    """`echo` (float16 [N,H,W] = rgb2gray(nparr).astype(float16), reference :400-402) and `nframes` may be handed over instead of
    `nparr` (process_folder's worker processes: the reader stage makes `echo`, the RGB frames need not travel to the writer)."""
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise OpticalFlowError("h5py is required to write the HDF5 output (not installed in this interpreter)") from e
    # The file appears under its final name only when it is complete: an error half-way (the reference has a latent one --
    # `attrs['frame_rate'] = None` raises in h5py when the DICOM lacks the tag, :405-407) never leaves a truncated .hdf5
    # that the skip-if-exists rule of process_folder would then take for a finished study.
    tmp_path = f"{save_path}.part{os.getpid()}"
    try:
        _write(h5py, tmp_path, flow_arr, nparr, mask_dict, metadata, waveforms, patient_id, heart_rate, config, mode, no_saliency,
               include_waveforms, save_mask_subset, echo, nframes)
        os.replace(tmp_path, save_path)
    finally:
        if os.path.exists(tmp_path):
            os.remove(tmp_path)


def _write(h5py, path, flow_arr, nparr, mask_dict, metadata, waveforms, patient_id, heart_rate, config, mode, no_saliency,
           include_waveforms, save_mask_subset, echo=None, nframes=None):
    nan = float("nan")
    with h5py.File(path, "w") as f:
        create_gzip9(f, "echo", np.asarray(echo, dtype=np.float16) if echo is not None else rgb2gray(nparr).astype(np.float16))
        fd = create_gzip9(f, "flow", np.asarray(flow_arr).astype(np.float16))
        # missing metadata: same attribute names and float64 type as a complete study, value NaN (units_converted says so)
        fd.attrs["frame_rate"] = nan if metadata["frame_rate"] is None else metadata["frame_rate"]
        fd.attrs["nframes"] = int(nframes) if nframes is not None else nparr.shape[0]
        fd.attrs["pixel_spacing"] = nan if metadata["pixel_spacing"] is None else metadata["pixel_spacing"]
        fd.attrs["ID"] = patient_id
        fd.attrs["HR"] = heart_rate if heart_rate is not None else 0
        fd.attrs["no_saliency"] = no_saliency
        fd.attrs["mode"] = mode
        fd.attrs["units_converted"] = (metadata["pixel_spacing"] is not None and metadata["frame_rate"] is not None)
        fd.attrs["waveforms_present"] = include_waveforms
        if include_waveforms:
            ex = {k: waveforms.get(k, (False, None))[0] for k in ("ecg", "art", "cvp", "pap")}
            fd.attrs["CVP_exists"] = ex["cvp"]
            fd.attrs["PAP_exists"] = ex["pap"]
            fd.attrs["R_wave_data_present"] = metadata["R_wave_data_present"]
            rates = {"art": config.art_sampling_rate, "ecg": config.ecg_sampling_rate, "cvp": config.cvp_sampling_rate,
                     "pap": config.pap_sampling_rate}
            for k in ("art", "ecg", "cvp", "pap"):
                if ex[k]:
                    d = create_gzip9(f, k, np.asarray(waveforms[k][1]).astype(np.float16))
                    d.attrs["sampling_rate"] = rates[k]
        if metadata.get("R_wave_data_present"):
            create_gzip9(f, "RWaveTime", metadata["R_times"])
        saved = []
        for k in mask_dict.keys():
            if save_mask_subset is None or k in save_mask_subset:
                create_gzip9(f, k, mask_dict[k])
                saved.append(k)
        fd.attrs["labels"] = saved
