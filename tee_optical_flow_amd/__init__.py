"""tee_optical_flow_amd -- MI355X-native dense optical flow (DualTVL1) behind the reference's
OF_model.calc() / process_video() surface.  See DESIGN.md."""
import os as _os

# The engine's lanes need streams that run beside each other; HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues
# (4 by default) and serialises streams that share one.  The library probes for concurrent streams when it makes its lanes; a roomier
# pool makes that search trivial.  Only effective if set before the HIP runtime initialises (import this package before the first
# torch.cuda / HIP call); a value the caller has set is left alone.  (csrc/teeflow.hip: streams_concurrent, tf_hw_queue_default)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import OpticalFlowCalculationConfig, default_optical_flow_config
from .exceptions import (ConfigurationError, DICOMReadError, OpticalFlowCalculationError, OpticalFlowError,
                         WaveformLoadError, WaveformValidationError)
from .dense_flow import DenseFlow, createOptFlow_DeepFlow, createOptFlow_DualTVL1, cuda_OpticalFlowDual_TVL1_create

__all__ = ["DenseFlow", "createOptFlow_DualTVL1", "createOptFlow_DeepFlow", "cuda_OpticalFlowDual_TVL1_create", "OpticalFlowCalculationConfig",
           "default_optical_flow_config", "OpticalFlowError", "DICOMReadError", "WaveformLoadError",
           "WaveformValidationError", "OpticalFlowCalculationError", "ConfigurationError"]
