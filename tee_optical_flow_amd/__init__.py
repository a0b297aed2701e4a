"""tee_optical_flow_amd -- MI355X-native dense optical flow (DualTVL1) behind the reference's
OF_model.calc() / process_video() surface.  See DESIGN.md."""
from .config import OpticalFlowCalculationConfig, default_optical_flow_config
from .exceptions import (ConfigurationError, DICOMReadError, OpticalFlowCalculationError, OpticalFlowError,
                         WaveformLoadError, WaveformValidationError)
from .dense_flow import DenseFlow, createOptFlow_DeepFlow, createOptFlow_DualTVL1, cuda_OpticalFlowDual_TVL1_create

__all__ = ["DenseFlow", "createOptFlow_DualTVL1", "createOptFlow_DeepFlow", "cuda_OpticalFlowDual_TVL1_create", "OpticalFlowCalculationConfig",
           "default_optical_flow_config", "OpticalFlowError", "DICOMReadError", "WaveformLoadError",
           "WaveformValidationError", "OpticalFlowCalculationError", "ConfigurationError"]
