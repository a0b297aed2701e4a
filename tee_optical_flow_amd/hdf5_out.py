"""HDF5 output in the reference's layout (/root/reference/optical_flow/calculate_optical_flow.py:370-475)
so optical_flow_dataset.py / analysis.py / peak_detection.py consume the file unchanged.  h5py is an optional import
(absent from the default interpreter of this image)."""
import os

import numpy as np

from .exceptions import OpticalFlowError
from .frames import rgb2gray


def save_optical_flow_to_hdf5(save_path, flow_arr, nparr, mask_dict, metadata, waveforms, patient_id, heart_rate, config,
                              mode, no_saliency, include_waveforms, save_mask_subset=None):
    try:
        import h5py
    except ImportError as e:  # pragma: no cover
        raise OpticalFlowError("h5py is required to write the HDF5 output (not installed in this interpreter)") from e
    if os.path.exists(save_path):
        os.remove(save_path)
    gz = dict(compression="gzip", compression_opts=9)
    with h5py.File(save_path, "w") as f:
        f.create_dataset("echo", data=rgb2gray(nparr).astype(np.float16), **gz)
        fd = f.create_dataset("flow", data=np.asarray(flow_arr).astype(np.float16), **gz)
        fd.attrs["frame_rate"] = metadata["frame_rate"]
        fd.attrs["nframes"] = nparr.shape[0]
        fd.attrs["pixel_spacing"] = metadata["pixel_spacing"]
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
                    d = f.create_dataset(k, data=np.asarray(waveforms[k][1]).astype(np.float16), **gz)
                    d.attrs["sampling_rate"] = rates[k]
        if metadata.get("R_wave_data_present"):
            f.create_dataset("RWaveTime", data=metadata["R_times"], **gz)
        saved = []
        for k in mask_dict.keys():
            if save_mask_subset is None or k in save_mask_subset:
                f.create_dataset(k, data=mask_dict[k], **gz)
                saved.append(k)
        fd.attrs["labels"] = saved
