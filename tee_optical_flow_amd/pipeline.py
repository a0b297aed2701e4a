"""Study-level driver around the flow engine: the part of the reference's process_video() that sits directly on
either side of the hot path.

Reference: /root/reference/optical_flow/calculate_optical_flow.py
  :564-578  model construction          -> make_flow_model
  :584-600  per-pair loop, pad, scale   -> flow_for_study   (ONE batched engine call instead of N-1 cv2 calls)
  :627-660  calculate_optical_flow()    -> calculate_optical_flow (same signature/behaviour, incl. WASE quirks)
  :478-625  process_video()             -> process_video (same signature; `nparr=`/`metadata=` inject frames where the
                                           DICOM blob / pydicom are absent)
"""
import logging

import numpy as np

from .config import default_optical_flow_config
from .dense_flow import DenseFlow
from .exceptions import ConfigurationError, DICOMReadError, OpticalFlowCalculationError
from .frames import condition_frames

logger = logging.getLogger(__name__)


def make_flow_model(OF_algo="TVL1", config=None, device_id=0):
    """Reference :564-578.  Both TV-L1 branches (cv2.cuda / cv2.optflow) map to the one HIP engine; unlike the
    reference's CUDA branch (Appendix C.2) lambda_value IS applied."""
    if config is None:
        config = default_optical_flow_config()
    if OF_algo == "TVL1":
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


def flow_for_study(frames_u8, OF_model, mask_dict=None, bkgd_comp="none", conversion_factor=1.0, nparr_rgb=None):
    """Reference :584-600 for already-conditioned uint8 frames [N,H,W]: N-1 flows, last one duplicated, scaled.
    All N-1 pairs are solved by ONE batched call (tf_calc_seq); background compensation (per pair) and the unit
    scale are applied in the reference's order: (flow - background) * conversion_factor."""
    if bkgd_comp not in ("WASE", "none"):
        raise OpticalFlowCalculationError(f"bkgd_comp value must be [WASE, none], got {bkgd_comp}!")
    if nparr_rgb is not None and hasattr(OF_model, "calc_study"):
        flows = OF_model.calc_study(nparr_rgb)                   # conditioning (:588) + all pairs on the device
    else:
        flows = OF_model.calc_batch(frames_u8)                   # float32 [N-1,H,W,2]
    if bkgd_comp == "WASE":
        if hasattr(OF_model, "wase_compensate"):                 # device: O(N^2 H W) products, numpy's summation order kept
            flows, _ = OF_model.wase_compensate(flows, mask_dict["bkgd"])
        else:
            flows = np.stack([_compensate(flows[i], mask_dict, "WASE") for i in range(flows.shape[0])])
    flows = np.concatenate([flows, flows[-1:]], axis=0)          # copy last optical flow (:599)
    return flows * conversion_factor                              # (:600)


def process_video(dcm_path, save_path, segmentor_model=None, verbose=True, mode="A4C", bkgd_comp="none", flipLR=False,
                  no_saliency=False, OF_algo="TVL1", save_mask_subset=None, include_waveforms=False, waveform_folder=None,
                  config=None, *, nparr=None, metadata=None, patient_id="", heart_rate=0, waveforms=None, flow_model=None,
                  mask_dict=None):
    """Same positional signature as the reference (:478-483).  Keyword-only extras let a caller inject what the
    offline image cannot provide: `nparr` (frames instead of a DICOM), `metadata`, `mask_dict` (segmentation result),
    `flow_model`.  Returns the float32 flow array [N,H,W,2] that was written."""
    if config is None:
        config = default_optical_flow_config()
    if mode == "otsu":
        if bkgd_comp != "none":
            raise ConfigurationError(f"bkgd_comp {bkgd_comp} is not supported in mode=otsu, can only support bkgd_comp=none")
        if save_mask_subset is not None:
            raise ConfigurationError("In mode=otsu, save_mask_subset must be None")
    if nparr is None:
        raise DICOMReadError(f"Failed to read DICOM file: {dcm_path} (pydicom is not available here; pass nparr=)")
    if not no_saliency:
        raise OpticalFlowCalculationError("cv2.saliency preprocessing is not available; use no_saliency=True "
                                          "(what the reference's own CLI runs, calculate_optical_flow.py:737)")
    nparr = np.asarray(nparr)
    if metadata is None:
        metadata = {"pixel_spacing": None, "frame_rate": None, "R_wave_data_present": False, "R_times": None}
    if nparr.ndim == 3 and nparr.shape[0] > 1:
        nparr = np.repeat(nparr[..., None], 3, axis=3)
    ps, fr = metadata["pixel_spacing"], metadata["frame_rate"]
    conversion_factor = 1.0 if ps is None or fr is None else ps * fr
    if flipLR:
        nparr = np.flip(nparr, axis=2)
    if mask_dict is None:
        if mode == "otsu":
            from .masks import predict_movie_thres
            mask_dict = predict_movie_thres(nparr, verbose=verbose, config=config)
        elif mode in ("A4C", "RVIO_2class"):
            raise ConfigurationError("SAM segmentation stays stock PyTorch outside this engine: pass its result as mask_dict=")
        else:
            raise ConfigurationError(f"Input for mode must be [A4C, otsu, RVIO_2class], not {mode}.")
    own = flow_model is None
    model = make_flow_model(OF_algo, config) if own else flow_model
    try:
        on_device = hasattr(model, "calc_study") and nparr.ndim == 4 and nparr.shape[3] == 3 and nparr.dtype == np.uint8
        frames = None if on_device else condition_frames(nparr)
        flow_arr = flow_for_study(frames, model, mask_dict, bkgd_comp, conversion_factor,
                                  nparr_rgb=np.ascontiguousarray(nparr) if on_device else None)
    finally:
        if own:
            model.close()
    if save_path is not None:
        from .hdf5_out import save_optical_flow_to_hdf5
        save_optical_flow_to_hdf5(save_path, flow_arr, nparr, mask_dict, metadata, waveforms or {}, patient_id, heart_rate,
                                  config, mode, no_saliency, include_waveforms and bool(waveforms), save_mask_subset)
    return flow_arr
