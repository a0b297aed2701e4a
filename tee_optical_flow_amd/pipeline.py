"""Study-level driver around the flow engine: the part of the reference's process_video() / process_folder() that sits
directly on either side of the hot path.

Reference: /root/reference/optical_flow/calculate_optical_flow.py
  :564-578  model construction          -> make_flow_model
  :584-600  per-pair loop, pad, scale   -> flow_for_study   (ONE batched engine call instead of N-1 cv2 calls)
  :627-660  calculate_optical_flow()    -> calculate_optical_flow (same signature/behaviour, incl. WASE quirks)
  :478-625  process_video()             -> process_video (same signature; `nparr=`/`metadata=` inject frames where the
                                           DICOM blob / pydicom are absent)
  :243-290  process_folder()            -> process_folder (same signature and chunk / skip / per-file isolation rules;
                                           studies are dealt round-robin over ranks and the gzip-9 HDF5 write of study k
                                           runs beside the solve of study k+1)
"""
import logging
import os
import traceback
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .config import default_optical_flow_config
from .dense_flow import DenseFlow
from .exceptions import ConfigurationError, DICOMReadError, OpticalFlowCalculationError
from .frames import condition_frames

logger = logging.getLogger(__name__)


def make_flow_model(OF_algo="TVL1", config=None, device_id=0, tvl1_variant="cpu"):
    """Reference :564-578.  tvl1_variant='cpu' (default, the parity target) is the reference's non-CUDA branch:
    createOptFlow_DualTVL1() + setLambda(config.lambda_value) (:577-578).  tvl1_variant='cuda' APPROXIMATES what the
    reference runs on a CUDA box (:572-575): cv2.cuda.OpticalFlowDual_TVL1.create() with NOTHING set on it -- lambda_value is
    silently ignored there, and so it is here.  Parity of that variant is unpinned: it restates
    the CUDA class's four known differences from memory, not cuda::resize's sampling or nvcc's FMA contraction."""
    if config is None:
        config = default_optical_flow_config()
    if OF_algo == "TVL1":
        if tvl1_variant == "cuda":
            return DenseFlow(device_id=device_id, variant="cuda")
        m = DenseFlow(device_id=device_id)
        m.setLambda(config.lambda_value)
        return m
    if OF_algo == "deepflow":
        return DenseFlow(device_id=device_id, algo="deepflow")      # reference :568, all-default DeepFlow
    raise OpticalFlowCalculationError("OF_algo only supports deepflow or TVL1")


def wase_background(flow, bkgd_mask):
    """Reference :647-652 verbatim semantics: ONE scalar = mean of the non-zero entries of flow[h,w,c]*mask[n,h,w,c]
    broadcast over ALL n frames, taken over both components together (Appendix C.3).  Closed form of the O(N) numpy
    expression: sum(flow*cnt) / sum(cnt*[flow != 0]) with cnt = sum_n mask -- evaluated exactly as numpy would."""
    masked = flow * bkgd_mask
    return np.mean(masked[masked != 0])


def calculate_optical_flow(saliency_1, saliency_2, mask_dict, OF_model, bkgd_comp="none", OF_algo="TVL1"):
    """Drop-in for the reference function of the same name (:627-660)."""
    if OF_algo in ("deepflow", "TVL1"):
        flow = OF_model.calc(saliency_1, saliency_2, None)
    else:
        raise OpticalFlowCalculationError("OF_algo only supports deepflow or TVL1")
    return _compensate(flow, mask_dict, bkgd_comp)


def _compensate(flow, mask_dict, bkgd_comp):
    if bkgd_comp == "WASE":
        background = wase_background(flow, mask_dict["bkgd"])
    elif bkgd_comp == "none":
        background = 0
    else:
        raise OpticalFlowCalculationError(f"bkgd_comp value must be [WASE, none], got {bkgd_comp}!")
    return flow - background


_saliency_warned = [False]


def flow_for_study(frames_u8, OF_model, mask_dict=None, bkgd_comp="none", conversion_factor=1.0, nparr_rgb=None, saliency=False,
                   saliency_map="f32"):
    """Reference :584-600 for already-conditioned uint8 frames [N,H,W]: N-1 flows, last one duplicated, scaled.
    All N-1 pairs are solved by ONE batched call (tf_calc_seq); background compensation (per pair) and the unit
    scale are applied in the reference's order: (flow - background) * conversion_factor.
    `saliency=True` is the no_saliency=False branch (:559-560, :586): the frames' fine-grained saliency maps, computed on the
    device from `nparr_rgb`, are what the solver sees -- `saliency_map` "f32" (default): as CV_32F in [0,1], what computeSaliency()
    returns under the reference's opencv-contrib >= 4.5 (DualTVL1 then scales by 255 in float, DeepFlow takes [0,1] frames as they
    are); "u8": the 8-bit maps (opencv-contrib 3.x).  Parity of the whole branch is UNPINNED (oracle/saliency_oracle.c)."""
    if bkgd_comp not in ("WASE", "none"):
        raise OpticalFlowCalculationError(f"bkgd_comp value must be [WASE, none], got {bkgd_comp}!")
    if saliency:
        if nparr_rgb is None or not hasattr(OF_model, "calc_study_saliency"):
            raise OpticalFlowCalculationError("no_saliency=False needs the device engine (DenseFlow.calc_study_saliency) and the "
                                              "study's frames; there is no CPU saliency path")
        if not _saliency_warned[0]:
            _saliency_warned[0] = True
            logger.warning("no_saliency=False: the saliency preprocessing (StaticSaliencyFineGrained, map handed over as %s) is a restatement of "
                           "opencv-contrib that no OpenCV output pins; files written on this branch are not verified against the reference's", saliency_map)
        if bkgd_comp == "none":
            # unit scale in the output kernel, last flow repeated inside the pinned result buffer (as the no_saliency=True branch below)
            return OF_model.calc_study_saliency(nparr_rgb, scale=conversion_factor, pad_last=True, map_dtype=saliency_map)
        flows = OF_model.calc_study_saliency(nparr_rgb, map_dtype=saliency_map)          # saliency maps (:586) + all pairs on the device
    elif nparr_rgb is not None and hasattr(OF_model, "calc_study"):
        if bkgd_comp == "none" and getattr(OF_model, "device_unit_scale", False):
            # the unit scale (:600, one float32 multiply per value, the same one numpy makes) is applied by the output kernel and
            # the last flow is repeated (:599) inside the pinned result buffer: no 136-MB concatenate and multiply on the host
            return OF_model.calc_study(nparr_rgb, scale=conversion_factor, pad_last=True)
        flows = OF_model.calc_study(nparr_rgb)                   # conditioning (:588) + all pairs on the device
    else:
        flows = OF_model.calc_batch(frames_u8)                   # float32 [N-1,H,W,2]
    scaled = False
    if bkgd_comp == "WASE":
        if hasattr(OF_model, "wase_compensate"):                 # device: O(N^2 H W) products, numpy's summation order kept
            flows, _ = OF_model.wase_compensate(flows, mask_dict["bkgd"], scale=conversion_factor)   # (flow - background) * factor
            scaled = True
        else:
            flows = np.stack([_compensate(flows[i], mask_dict, "WASE") for i in range(flows.shape[0])])
    flows = np.concatenate([flows, flows[-1:]], axis=0)          # copy last optical flow (:599)
    return flows if scaled else flows * conversion_factor         # (:600)


def _prep_frames(nparr, flipLR):
    """Reference :533-548: greyscale stacks become RGB, optional left-right flip."""
    nparr = np.asarray(nparr)
    if nparr.ndim == 3 and nparr.shape[0] > 1:
        nparr = np.repeat(nparr[..., None], 3, axis=3)
    if flipLR:
        nparr = np.flip(nparr, axis=2)
    return nparr


def process_video(dcm_path, save_path, segmentor_model=None, verbose=True, mode="A4C", bkgd_comp="none", flipLR=False,
                  no_saliency=False, OF_algo="TVL1", save_mask_subset=None, include_waveforms=False, waveform_folder=None,
                  config=None, *, nparr=None, metadata=None, patient_id="", heart_rate=0, waveforms=None, flow_model=None,
                  mask_dict=None, _defer_save=None, saliency_map="f32"):
    """Same positional signature as the reference (:478-483).  Keyword-only extras let a caller inject what the
    offline image cannot provide: `nparr` (frames instead of a DICOM), `metadata`, `mask_dict` (segmentation result),
    `flow_model`.  Returns the float32 flow array [N,H,W,2] that was written."""
    return _process_video_begin(dcm_path, save_path, segmentor_model, verbose, mode, bkgd_comp, flipLR, no_saliency, OF_algo, save_mask_subset,
                                include_waveforms, waveform_folder, config, nparr=nparr, metadata=metadata, patient_id=patient_id,
                                heart_rate=heart_rate, waveforms=waveforms, flow_model=flow_model, mask_dict=mask_dict, _defer_save=_defer_save,
                                saliency_map=saliency_map)()


def _process_video_begin(dcm_path, save_path, segmentor_model=None, verbose=True, mode="A4C", bkgd_comp="none", flipLR=False,
                         no_saliency=False, OF_algo="TVL1", save_mask_subset=None, include_waveforms=False, waveform_folder=None,
                         config=None, *, nparr=None, metadata=None, patient_id="", heart_rate=0, waveforms=None, flow_model=None,
                         mask_dict=None, _defer_save=None, saliency_map="f32", _submit=False):
    """process_video in two halves: everything up to the flow solve, then a callable that collects the flows and does the rest (waveforms,
    hand-over to the HDF5 writer) and returns the flow array.  `_submit` (process_folder): where the engine offers it (gray-frame branch,
    no background compensation, a model the caller holds) the solve is only SUBMITTED here -- DenseFlow.submit_study, tf_submit_seq_rgb --
    so that the next study's solve is on the GPU while this one finishes."""
    if config is None:
        config = default_optical_flow_config()
    if mode == "otsu":
        if bkgd_comp != "none":
            raise ConfigurationError(f"bkgd_comp {bkgd_comp} is not supported in mode=otsu, can only support bkgd_comp=none")
        if save_mask_subset is not None:
            raise ConfigurationError("In mode=otsu, save_mask_subset must be None")
    if nparr is None:
        # the reference's own call shape, process_video(dcm_path, save_path, ...) (:519-531): read the study file.  `.dcm` needs
        # pydicom (DICOMReadError without it, like a failed _read_dicom_file); `.npy` / `.npz` are the offline injection formats.
        nparr, md_file, pid_file, hr_file = read_study(dcm_path)
        if metadata is None:
            metadata = md_file
        patient_id = patient_id or pid_file
        heart_rate = heart_rate or hr_file
    nparr = np.asarray(nparr)
    if metadata is None:
        metadata = {"pixel_spacing": None, "frame_rate": None, "R_wave_data_present": False, "R_times": None}
    nparr = _prep_frames(nparr, flipLR)
    ps, fr = metadata["pixel_spacing"], metadata["frame_rate"]
    conversion_factor = 1.0 if ps is None or fr is None else ps * fr
    if mask_dict is None:
        if mode == "otsu":
            from .masks import predict_movie_thres
            mask_dict = predict_movie_thres(nparr, verbose=verbose, config=config)
        elif mode in ("A4C", "RVIO_2class"):
            if segmentor_model is None:
                raise ConfigurationError(f"mode={mode} needs segmentor_model (a module with image_encoder / prompt_encoder / "
                                         "mask_decoder, reference :47-88) or a precomputed mask_dict=")
            from .masks import predict_movie
            mask_dict = predict_movie(nparr, segmentor_model, mode=mode, verbose=verbose, config=config)   # reference :549-550
        else:
            raise ConfigurationError(f"Input for mode must be [A4C, otsu, RVIO_2class], not {mode}.")
    own = flow_model is None
    model = make_flow_model(OF_algo, config) if own else flow_model
    collect = None                                # () -> the study's flow array
    try:
        rgb_u8 = nparr.ndim == 4 and nparr.shape[3] == 3 and nparr.dtype == np.uint8
        if not no_saliency:
            # the reference's default branch (:559-560, :586): cv2.saliency.StaticSaliencyFineGrained on every frame
            if not rgb_u8:
                raise OpticalFlowCalculationError(f"no_saliency=False needs uint8 RGB frames [N,H,W,3], got {nparr.dtype} {nparr.shape}")
            flow_arr = flow_for_study(None, model, mask_dict, bkgd_comp, conversion_factor,
                                      nparr_rgb=np.ascontiguousarray(nparr), saliency=True, saliency_map=saliency_map)
        else:
            on_device = hasattr(model, "calc_study") and rgb_u8
            if (_submit and not own and on_device and bkgd_comp == "none" and hasattr(model, "submit_study")
                    and getattr(model, "device_unit_scale", False)):
                # conditioning now, the solve queued on the engine's lanes: the same call as flow_for_study's device branch, not waited for
                ticket = model.submit_study(np.ascontiguousarray(nparr), scale=conversion_factor, pad_last=True)
                collect = lambda: model.wait(ticket)
            else:
                frames = None if on_device else condition_frames(nparr)
                flow_arr = flow_for_study(frames, model, mask_dict, bkgd_comp, conversion_factor,
                                          nparr_rgb=np.ascontiguousarray(nparr) if on_device else None)
    finally:
        if own:
            model.close()
    if collect is None:
        collect = lambda: flow_arr

    def finish():
        flows = collect()
        # waveforms (reference :602-620): loaded and validated by the reference's rules unless the caller injected a result dict;
        # without a valid ECG and a valid arterial waveform the whole block is dropped (waveforms_present = False)
        waveform_results, with_waveforms = {}, include_waveforms
        if with_waveforms:
            from .waveforms import load_all_waveforms, waveforms_to_write
            waveform_results = waveforms if waveforms is not None else load_all_waveforms(dcm_path, waveform_folder, config, verbose)
            if not waveforms_to_write(waveform_results):
                with_waveforms = False
        if save_path is not None:
            job = (save_path, flows, nparr, mask_dict, metadata, waveform_results, patient_id, heart_rate,
                   config, mode, no_saliency, with_waveforms, save_mask_subset)
            if _defer_save is not None:
                _defer_save(job)                  # process_folder: the writer stage takes it while the next study is solved
            else:
                from .hdf5_out import save_optical_flow_to_hdf5
                save_optical_flow_to_hdf5(*job)
        return flows
    return finish


def read_study(path):
    """Frames + metadata of one study file.  `.dcm` needs pydicom (absent in this image -> DICOMReadError, as the
    reference's _read_dicom_file failure, :520-522); `.npy` (uint8 [N,H,W] or [N,H,W,3]) and `.npz` (key `nparr`, optional
    `pixel_spacing`, `frame_rate`, `patient_id`, `heart_rate`) are the injection formats of the offline image."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path, allow_pickle=False), None, "", 0
    if ext == ".npz":
        z = np.load(path, allow_pickle=False)
        md = {"pixel_spacing": float(z["pixel_spacing"]) if "pixel_spacing" in z else None,
              "frame_rate": float(z["frame_rate"]) if "frame_rate" in z else None, "R_wave_data_present": False, "R_times": None}
        return z["nparr"], md, str(z["patient_id"]) if "patient_id" in z else "", int(z["heart_rate"]) if "heart_rate" in z else 0
    try:
        import pydicom
    except ImportError as e:
        raise DICOMReadError(f"Failed to read DICOM file: {path} (pydicom is not installed)") from e
    try:
        ds = pydicom.dcmread(path)
        arr = ds.pixel_array
    except (IOError, OSError, KeyError, AttributeError) as e:             # reference _read_dicom_file (:307-312) -> DICOMReadError (:521-522)
        raise DICOMReadError(f"Failed to read DICOM file: {path}") from e
    return dicom_to_study(ds, arr, _pydicom_color_converter(pydicom))


def _pydicom_color_converter(pydicom):
    """The reference's colour-space step (calculate_optical_flow.py:524-526) as a callable (ds, arr) -> arr."""
    def convert(ds, arr):
        handlers = pydicom.pixel_data_handlers
        if handlers.numpy_handler.should_change_PhotometricInterpretation_to_RGB(ds):
            return handlers.convert_color_space(arr, ds.PhotometricInterpretation, "RGB")
        return arr
    return convert


def extract_dicom_metadata(ds):
    """The reference's _extract_dicom_metadata (calculate_optical_flow.py:315-365) on any dataset-like object, rule for rule:
    pixel spacing = PhysicalDeltaX of the first ultrasound region (tag 0018,6011), R-wave times when RWaveTimeVector is
    present and not a bare float, frame rate = CineRate as stored, else round(1000 / FrameTime), else
    round(1000 / FrameTimeVector[1]).  The rounding matters: conversion_factor = pixel_spacing * frame_rate scales every
    stored flow value (:538-541)."""
    md = {"pixel_spacing": None, "frame_rate": None, "R_times": None, "R_wave_data_present": False}
    try:
        md["pixel_spacing"] = ds[0x0018, 0x6011][0]["PhysicalDeltaX"].value
    except (KeyError, AttributeError, IndexError, TypeError):
        pass
    try:
        if type(ds.RWaveTimeVector) != float and ds.RWaveTimeVector is not None:      # noqa: E721  (the reference's own test)
            md["R_times"] = np.asarray(ds.RWaveTimeVector)
            md["R_wave_data_present"] = True
    except (AttributeError, KeyError, TypeError):
        pass
    try:
        md["frame_rate"] = ds.CineRate
    except (AttributeError, KeyError):
        try:
            md["frame_rate"] = np.round(1000 / float(ds.FrameTime))
        except (AttributeError, KeyError, ValueError, ZeroDivisionError):
            try:
                md["frame_rate"] = np.round(1000 / float(ds.FrameTimeVector[1]))
            except (AttributeError, KeyError, IndexError, ValueError, ZeroDivisionError):
                pass
    return md


def dicom_to_study(ds, arr, convert_color=None):
    """(frames, metadata, patient id, heart rate) of a read DICOM dataset, as process_video uses them (:524-531, :405-417):
    colour space converted to RGB when the dataset asks for it, metadata per extract_dicom_metadata."""
    if convert_color is not None:
        arr = convert_color(ds, arr)
    return arr, extract_dicom_metadata(ds), str(getattr(ds, "PatientID", "")), int(getattr(ds, "HeartRate", 0) or 0)


# ---- shared-memory transport between process_folder's stages -------------------------------------------------------------
# A 65-frame 512x512 study is ~50 MB of frames, ~35 MB of masks, 17 MB of `echo` and 68 MB of float16 flow.  Through the pools'
# pipes every one of those bytes is pickled, written, read and unpickled under the caller's interpreter lock (measured: 460 of a
# study's 700 ms in the caller's thread were spent receiving the reader stage's result).  Arrays above _SHM_MIN bytes therefore
# travel as POSIX shared-memory blocks: the producer fills a block and sends its name, the consumer maps it; the process_folder
# call that owns the study unlinks its blocks when the writer stage is done with them (or on any error).  If /dev/shm is too
# small for a study (containers often give it 64 MB; writing past the limit would be a SIGBUS), that study's arrays travel
# pickled, as before.
_SHM_MIN = int(os.environ.get("TEEFLOW_SHM_MIN_BYTES", 1 << 20))      # (the environment reaches spawned workers; tests lower it)
_SHM_MARGIN = 1 << 30
_shm_stats = {"mapped": 0, "created": 0}                             # blocks this process mapped / created (tests, tools)


def _shm_room(nbytes):
    try:
        st = os.statvfs("/dev/shm")
    except OSError:
        return False
    return st.f_bavail * st.f_frsize > 2 * nbytes + _SHM_MARGIN


def _shm_put(arr):
    """ndarray -> ("shm", name, shape, dtype) with the data in a new shared-memory block (this process's mapping is closed, the block
    stays), or the array itself when it is small / there is no room."""
    from multiprocessing import shared_memory
    arr = np.ascontiguousarray(arr)
    if arr.nbytes < _SHM_MIN or not _shm_room(arr.nbytes):
        return arr
    blk = shared_memory.SharedMemory(create=True, size=arr.nbytes)
    try:
        np.ndarray(arr.shape, arr.dtype, buffer=blk.buf)[...] = arr
    except BaseException:
        blk.close(); blk.unlink()
        raise
    desc = ("shm", blk.name, arr.shape, arr.dtype.str)
    blk.close()
    return desc


def _shm_new(shape, dtype):
    """An empty shared-memory array for the caller to fill: (view, block, descriptor), or (None, None, None) without room."""
    from multiprocessing import shared_memory
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes < _SHM_MIN or not _shm_room(nbytes):
        return None, None, None
    blk = shared_memory.SharedMemory(create=True, size=nbytes)
    _shm_stats["created"] += 1
    return np.ndarray(shape, dtype, buffer=blk.buf), blk, ("shm", blk.name, tuple(shape), np.dtype(dtype).str)


def _is_shm(x):
    return isinstance(x, tuple) and len(x) == 4 and x[0] == "shm"


def _shm_get(x, blocks):
    """Descriptor -> ndarray view (its block is appended to `blocks`, which keeps the mapping alive); anything else passes through."""
    if not _is_shm(x):
        return x
    from multiprocessing import shared_memory
    blk = shared_memory.SharedMemory(name=x[1])
    blocks.append(blk)
    _shm_stats["mapped"] += 1
    return np.ndarray(x[2], np.dtype(x[3]), buffer=blk.buf)


def _shm_release(blocks, unlink):
    for blk in blocks:
        try:
            blk.close()
        except (OSError, BufferError):                       # a view is still alive somewhere: the mapping goes with it
            pass
        if unlink:
            try:
                blk.unlink()                                  # by name: works whatever is still mapped
            except OSError:
                pass
    del blocks[:]


def _shm_unlink_names(descs):
    """Unlink blocks this process never mapped (a study that failed before its arrays were used)."""
    from multiprocessing import shared_memory
    for d in descs:
        if _is_shm(d):
            try:
                blk = shared_memory.SharedMemory(name=d[1])
                blk.close(); blk.unlink()
            except OSError:
                pass


def _prepare_study_shm(reader, path, mode, flipLR, config, want_echo):
    """_prepare_study in a worker process, the big arrays returned as shared-memory descriptors."""
    nparr, md, pid, hr, masks_ahead, echo = _prepare_study(reader, path, mode, flipLR, config, want_echo)
    made = []
    try:
        nparr = _shm_put(nparr); made.append(nparr)
        if masks_ahead is not None:
            packed = {}
            for k, v in masks_ahead.items():
                packed[k] = _shm_put(v); made.append(packed[k])
            masks_ahead = packed
        if echo is not None:
            echo = _shm_put(echo); made.append(echo)
    except BaseException:
        _shm_unlink_names(made)
        raise
    return nparr, md, pid, hr, masks_ahead, echo


def _save_study_shm(job, echo, nframes):
    """Writer stage in a worker process: map what arrived as descriptors, write the file, drop the mappings (the owner unlinks)."""
    from .hdf5_out import save_optical_flow_to_hdf5
    blocks = []
    try:
        save_path, flow_arr, nparr, mask_dict, *rest = job
        flow_arr = _shm_get(flow_arr, blocks)
        nparr = _shm_get(nparr, blocks)
        mask_dict = {k: _shm_get(v, blocks) for k, v in mask_dict.items()}
        echo = _shm_get(echo, blocks)
        save_optical_flow_to_hdf5(save_path, flow_arr, nparr, mask_dict, *rest, echo=echo, nframes=nframes)
        del flow_arr, nparr, mask_dict, echo
    finally:
        _shm_release(blocks, unlink=False)


def _default_stage_workers():
    """Worker processes per stage when the caller does not say: the deflate of a study is ~9 core-seconds and the mask stage ~0.6, the
    caller's thread needs a core of its own; 3 + 3 on the 16 cores of a one-GPU box (2 + 2 measured 747 ms per study, 3 + 3 616 ms,
    5 + 6 613 ms), never fewer than 2 nor more than 4 per stage."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 4
    return max(2, min(4, cpus // 5))


class StudyWorkers:
    """Worker processes for process_folder's reader/mask and deflate/write stages, for callers that hold a flow model (or a segmentor on
    the GPU) across many calls: create this object BEFORE anything in the process touches the GPU -- starting a process from a
    GPU-initialised one is not safe on ROCm hosts -- and hand it to process_folder(workers=...).  process_folder(workers="process" /
    "auto") makes its own for one call."""

    def __init__(self, n_readers=None, n_writers=None):
        import multiprocessing as mp
        n_readers = _default_stage_workers() if n_readers is None else n_readers
        n_writers = _default_stage_workers() if n_writers is None else n_writers
        from concurrent.futures import ProcessPoolExecutor
        ctx = mp.get_context("spawn")
        self.n_readers, self.n_writers = max(1, n_readers), max(1, n_writers)
        self.readers = ProcessPoolExecutor(self.n_readers, mp_context=ctx)
        self.writers = ProcessPoolExecutor(self.n_writers, mp_context=ctx)
        # make the pools start their processes now (they are created lazily on first submit)
        for f in [self.readers.submit(os.getpid) for _ in range(self.n_readers)] + [self.writers.submit(os.getpid) for _ in range(self.n_writers)]:
            f.result()

    def close(self):
        self.readers.shutdown(wait=True)
        self.writers.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _prepare_study(reader, path, mode, flipLR, config, want_echo):
    """Reader stage of process_folder (module level: it also runs in worker processes).  Reads the study and -- for the Otsu mode,
    which is pure numpy/scipy -- computes its masks one study ahead of the GPU; with the solver at milliseconds per pair, this host
    work and the gzip-9 write are what a study costs.  `want_echo`: also the `echo` dataset (rgb2gray of the frames as float16,
    reference :400-402), so that the writer process does not need the RGB frames."""
    nparr, md, pid, hr = reader(path)
    masks_ahead = None
    prepped = _prep_frames(nparr, flipLR)
    if mode == "otsu":
        from .masks import predict_movie_thres
        masks_ahead = predict_movie_thres(prepped, verbose=False, config=config)
    echo = None
    if want_echo:
        from .frames import rgb2gray
        echo = rgb2gray(prepped).astype(np.float16)
    return nparr, md, pid, hr, masks_ahead, echo


def _spawn_unsafe_reason(reader):
    """Why worker PROCESSES cannot serve this call, or None.  They are started with the spawn method (a process must never be forked
    from one that has initialised the GPU): the child re-imports the caller's `__main__` module from its file and unpickles `reader`.
    So (1) `reader` must pickle -- a lambda or a closure does not -- and (2) if `__main__` is a script, the call must come from under
    its `if __name__ == "__main__":` guard; otherwise the child would run the script's top level again and die in its bootstrap
    (every study would then come back as BrokenProcessPool).  Both are checked here, before a pool exists."""
    import pickle
    import sys
    try:
        pickle.dumps(reader)
    except Exception as e:
        return f"reader={reader!r} does not pickle ({type(e).__name__}: {e})"
    import __main__
    main_file = getattr(__main__, "__file__", None)
    if not main_file or not os.path.exists(main_file):
        return None                                           # python -c / interactive: spawn has no __main__ file to re-run
    fr = sys._getframe()
    while fr is not None and not (fr.f_code.co_name == "<module>" and fr.f_globals.get("__name__") == "__main__"):
        fr = fr.f_back
    if fr is None:
        return None                                           # not called from __main__'s module-level code (a thread, an importing tool)
    try:
        import ast
        with open(main_file) as fh:
            tree = ast.parse(fh.read())
    except (OSError, SyntaxError, ValueError):
        return f"cannot read {main_file} to see whether the call is under its __main__ guard"

    def is_guard(test):
        if not (isinstance(test, ast.Compare) and len(test.ops) == 1 and isinstance(test.ops[0], ast.Eq)):
            return False
        sides = [test.left, test.comparators[0]]
        return any(isinstance(x, ast.Name) and x.id == "__name__" for x in sides) and \
            any(isinstance(x, ast.Constant) and x.value == "__main__" for x in sides)
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and is_guard(node.test) and node.body and node.body[0].lineno <= fr.f_lineno <= node.body[-1].end_lineno:
            return None
    return (f"the call comes from module-level code of the script {os.path.basename(main_file)} (line {fr.f_lineno}) outside an "
            "`if __name__ == '__main__':` block, which a spawned worker would run again")


def _is_pool_failure(e):
    """An exception that condemns the worker pools, not the study: a worker died or an argument / result would not pickle."""
    import pickle
    from concurrent.futures.process import BrokenProcessPool
    return isinstance(e, (BrokenProcessPool, pickle.PicklingError)) or (isinstance(e, (AttributeError, TypeError)) and "pickle" in str(e).lower())


def process_folder(dcm_folder, save_folder, segmentor_model=None, nchunks=10, chunk_index=0, mode="RVIO_2class", bkgd_comp="none",
                   flipLR=False, verbose=True, recalculate=False, no_saliency=True, OF_algo="TVL1", save_mask_subset=None,
                   include_waveforms=False, waveform_folder=None, pixel_spacing=None, frame_rate=None, process_subset=False,
                   file_subset_list=(), *, rank=0, world=1, extensions=("dcm",), reader=read_study, flow_model=None, config=None,
                   device_id=0, workers="auto", n_readers=None, n_writers=None, studies_in_flight=2):
    """Drop-in for the reference's process_folder (:243-290), same positional signature and the same rules:
      * the folder listing is cut into `nchunks` slices of len // nchunks files, this call takes slice `chunk_index`
        (the remainder files are dropped, as the reference does -- SURVEY.md Appendix C.8);
      * a study whose `<name>.hdf5` exists is skipped unless `recalculate`;
      * every study runs inside its own try/except: a failure is logged, recorded and the walk goes on (:276-284);
      * files with another extension are skipped with a warning (reference: 'dcm' only; `extensions` widens that to the
        .npy/.npz injection formats);
      * like the reference, `pixel_spacing` / `frame_rate` are accepted and ignored, and `config` is not forwarded unless given.
    Beyond the reference: the slice is dealt round-robin over `world` ranks (one process per GPU, rank r takes files
    r, r+world, ...: no exchange is needed, studies are independent), one flow model serves all studies of the call, and
    the walk is a three-stage pipeline: the reader stage loads study k+1 (and computes its Otsu masks and the `echo` dataset), the
    caller's thread solves study k on the GPU, the writer stage deflates/writes study k-1.  `workers`: "process" runs the reader and
    writer stages in `n_readers` + `n_writers` worker PROCESSES (spawned here, before this call's first GPU call; the mask stage and
    the deflate both hold the interpreter lock, which is why threads bought 3 %), "thread" in one thread each, "auto" takes
    processes when this call creates the flow model itself (no `flow_model`, no `segmentor_model`: nothing in the caller's hands has
    initialised the GPU yet as far as this function can tell) and more than one study is to do.  `studies_in_flight` (2): a study's flow solve is
    submitted to the engine's lanes (tf_submit_seq_rgb) and collected only when the NEXT study's has been submitted, so the GPU goes from one
    study's solve to the next without waiting for this thread (1 = solve and collect study by study, as rounds 2-4 did).
    Returns the list of (filename, error string)."""
    os.makedirs(save_folder, exist_ok=True)
    file_list = sorted(os.listdir(dcm_folder))                      # os.listdir order is arbitrary; sorted = same slices on every rank
    errors = []
    if process_subset:
        if len(file_subset_list) == 0:
            logger.error("ERROR! File subset list is empty!")
            return errors
        file_list = [f for f in file_list if f in file_subset_list]
    if include_waveforms and waveform_folder is None:
        logger.error("ERROR if include_waveform is selected, must define waveform_folder!")
        return errors
    split = len(file_list) // nchunks
    mine = file_list[chunk_index * split:(chunk_index + 1) * split][rank::world]
    own = flow_model is None
    model = None
    pending = []
    cfg_masks = config if config is not None else default_optical_flow_config()
    shared = workers if isinstance(workers, StudyWorkers) else None
    if shared is not None:
        n_readers, n_writers = shared.n_readers, shared.n_writers
    n_readers = _default_stage_workers() if n_readers is None else n_readers
    n_writers = _default_stage_workers() if n_writers is None else n_writers
    use_proc = workers == "process" or (workers == "auto" and flow_model is None and segmentor_model is None)
    state = {"writer": None, "reader_pool": None, "proc": False, "fallback": None, "depth": 1}
    studies = {}                # save_path -> what of a study lives in shared memory until its writer is done
    begun_ref = []
    futs = {}

    def start_pools(n_todo):
        # worker processes only make sense for more than one study, and they must exist before the first GPU call of this function
        if shared is not None:
            state["reader_pool"], state["writer"], state["proc"] = shared.readers, shared.writers, True
        elif use_proc and n_todo > 1 and (why := _spawn_unsafe_reason(reader)) is not None:
            # (the :=-bound reason is logged; the walk is the same, its reader and writer stages run in one thread each)
            logger.warning(f"process_folder: worker processes not used, threads instead: {why}")
            state["fallback"] = why
            state["reader_pool"] = ThreadPoolExecutor(1)
            state["writer"] = ThreadPoolExecutor(1)
        elif use_proc and n_todo > 1:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            ctx = mp.get_context("spawn")
            state["reader_pool"] = ProcessPoolExecutor(max(1, min(n_readers, n_todo)), mp_context=ctx)
            state["writer"] = ProcessPoolExecutor(max(1, n_writers), mp_context=ctx)
            state["proc"] = True
        else:
            state["reader_pool"] = ThreadPoolExecutor(1)
            state["writer"] = ThreadPoolExecutor(1)

    def defer(job):
        from .hdf5_out import save_optical_flow_to_hdf5
        if state["proc"]:
            # the writer process needs neither the RGB frames (the reader stage made `echo` from them) nor float32 flow (the file holds
            # float16); what is big travels as shared-memory names: the flow is cast straight into a new block, the masks and `echo`
            # stay in the blocks the reader stage filled
            save_path, flow_arr, nparr, mask_dict, *rest = job
            study = studies.get(save_path, {})
            echo = study.get("echo")
            flow_arr = np.asarray(flow_arr)
            view, blk, desc = _shm_new(flow_arr.shape, np.float16)
            if blk is not None:
                study.setdefault("blocks", []).append(blk)
                view[...] = flow_arr                                # float32 -> float16 while copying
                del view
                flow16 = desc
            else:
                flow16 = flow_arr.astype(np.float16)
            masks = study.get("mask_descs") if study.get("mask_descs") is not None and mask_dict is study.get("mask_views") else mask_dict
            echo_d = study.get("echo_desc", echo)
            job = (save_path, flow16, None if echo_d is not None else nparr, masks, *rest)
            pending.append((save_path, state["writer"].submit(_save_study_shm, job, echo_d, int(np.asarray(nparr).shape[0]))))
        else:
            pending.append((job[0], state["writer"].submit(save_optical_flow_to_hdf5, *job)))

    def drop_study(save_path):
        study = studies.pop(save_path, None)
        if study is not None:
            for k in ("nparr", "mask_views", "echo"):
                study.pop(k, None)
            _shm_release(study.get("blocks", []), unlink=True)

    def reap(block):
        # at most a few studies wait for the writer: a faster solver must not pile finished studies up in host memory
        while pending and (block or len(pending) > (n_writers + 1 if state["proc"] else 2) or pending[0][1].done()):
            path, fut = pending.pop(0)
            try:
                fut.result()
            except Exception as e:                                   # the writer's failure belongs to that study
                logger.error(f"Error processing {os.path.basename(path)}: {e}")
                errors.append((os.path.basename(path), f"{type(e).__name__}: {e}"))
            drop_study(path)

    try:
        todo = []
        for filename in mine:
            stem, ext = os.path.splitext(filename)
            save_path = os.path.join(save_folder, stem + ".hdf5")
            if os.path.exists(save_path) and not recalculate:
                if verbose:
                    logger.debug(f"File {save_path} exists! Skipping file {filename}")
                continue
            if ext.lower().lstrip(".") not in extensions:
                logger.warning(f"File extension must be one of {extensions}, found {ext}, skipping")
                continue
            todo.append((filename, stem, save_path))
        start_pools(len(todo))
        state["depth"] = n_readers if state["proc"] else 1      # studies the reader stage may be ahead of the solver
        futs = {}

        def submit(k):
            if k < len(todo) and k not in futs:
                futs[k] = state["reader_pool"].submit(_prepare_study_shm if state["proc"] else _prepare_study, reader,
                                                      os.path.join(dcm_folder, todo[k][0]), mode, flipLR, cfg_masks, state["proc"])

        def pools_to_threads(e, k):
            # The worker pools are unusable (a worker died while starting, something would not pickle): the stages go on in threads,
            # as rounds 2-3 ran them.  Studies already handed to the writer pool are reaped (and reported) as they are.
            why = f"{type(e).__name__}: {e}"
            logger.warning(f"process_folder: worker processes failed ({why}); the reader and writer stages continue in threads")
            state["fallback"] = why
            reap(block=True)
            for kk, fut in list(futs.items()):
                fut.cancel()
                try:
                    res = fut.result(timeout=0) if fut.done() and not fut.cancelled() else None
                    if res is not None:
                        _shm_unlink_names([res[0], res[5]] + list((res[4] or {}).values()))
                except Exception:
                    pass
                del futs[kk]
            for pool in (state["writer"], state["reader_pool"]):
                pool.shutdown(wait=False, cancel_futures=True)
            state["reader_pool"], state["writer"], state["proc"], state["depth"] = ThreadPoolExecutor(1), ThreadPoolExecutor(1), False, 1
            submit(k)
        from collections import deque
        begun = deque()                                               # studies whose solve is submitted and not yet collected
        begun_ref.append(begun)

        def finish_oldest():
            filename, save_path, fin = begun.popleft()
            deferred = len(pending)
            try:
                fin()
            except Exception as e:
                logger.error(f"Error processing {filename}: {e}")
                if verbose:
                    traceback.print_exc()
                errors.append((filename, f"{type(e).__name__}: {e}"))
            del fin
            if len(pending) == deferred:                              # nothing was handed to the writer stage: the study's blocks go now
                if save_path in studies and not studies[save_path]["blocks"]:
                    _shm_unlink_names(studies[save_path].get("descs", []))
                drop_study(save_path)
            reap(block=False)
        for k in range(min(state["depth"], len(todo))):
            submit(k)
        for k, (filename, stem, save_path) in enumerate(todo):
            if verbose:
                logger.info(f"Processing file: {filename}...")
            submit(k)                                                 # (already there unless the pools have just been rebuilt)
            submit(k + state["depth"])
            nparr = masks_ahead = echo = None
            try:
                try:
                    prepared = futs.pop(k).result()
                except Exception as e:
                    if not (state["proc"] and shared is None and _is_pool_failure(e)):
                        raise
                    pools_to_threads(e, k)
                    prepared = futs.pop(k).result()
                nparr, md, pid, hr, masks_ahead, echo = prepared
                if state["proc"]:
                    # map what the reader stage left in shared memory; the descriptors go on to the writer stage as they are
                    study = studies[save_path] = {"blocks": [], "descs": [nparr, echo] + list((masks_ahead or {}).values())}
                    study["echo_desc"] = echo
                    study["mask_descs"] = masks_ahead if masks_ahead is not None and any(_is_shm(v) for v in masks_ahead.values()) else None
                    nparr = _shm_get(nparr, study["blocks"])
                    if masks_ahead is not None:
                        masks_ahead = {mk: _shm_get(mv, study["blocks"]) for mk, mv in masks_ahead.items()}
                    study["mask_views"] = masks_ahead
                    study["echo"] = echo = _shm_get(echo, study["blocks"]) if _is_shm(echo) else echo
                if model is None:
                    model = flow_model if flow_model is not None else make_flow_model(OF_algo, config, device_id)
                waveforms = None                                       # process_video loads and validates them (reference :602-620)
                fin = _process_video_begin(os.path.join(dcm_folder, filename), save_path, segmentor_model, verbose=verbose, mode=mode,
                                           bkgd_comp=bkgd_comp, flipLR=flipLR, no_saliency=no_saliency, OF_algo=OF_algo,
                                           save_mask_subset=save_mask_subset, include_waveforms=include_waveforms, waveform_folder=waveform_folder,
                                           config=config, nparr=nparr, metadata=md, patient_id=pid, heart_rate=hr, waveforms=waveforms,
                                           flow_model=model, mask_dict=masks_ahead, _defer_save=defer, _submit=studies_in_flight > 1)
                begun.append((filename, save_path, fin))
                del fin
            except Exception as e:
                logger.error(f"Error processing {filename}: {e}")
                if verbose:
                    traceback.print_exc()
                errors.append((filename, f"{type(e).__name__}: {e}"))
                if save_path in studies and not studies[save_path]["blocks"]:
                    _shm_unlink_names(studies[save_path].get("descs", []))
                drop_study(save_path)
            del nparr, masks_ahead, echo
            while len(begun) >= max(1, studies_in_flight):
                finish_oldest()
        while begun:
            finish_oldest()
        reap(block=True)
    finally:
        while begun_ref and begun_ref[0]:                             # (an exception above) submitted solves are collected before the model goes
            try:
                begun_ref[0].popleft()[2]()
            except Exception:
                pass
        for _path, fut in pending:                                    # (same case) writers still at work keep their study's blocks until done
            try:
                fut.result()
            except Exception:
                pass
        for k, fut in futs.items():                                   # reader results nobody took (an exception above): free their blocks
            try:
                res = fut.result()
                _shm_unlink_names([res[0], res[5]] + list((res[4] or {}).values()))
            except Exception:
                pass
        for path in list(studies):
            drop_study(path)
        for pool in (state["writer"], state["reader_pool"]):
            if pool is not None and shared is None:
                pool.shutdown(wait=True)
        if own and model is not None:
            model.close()
    return errors
