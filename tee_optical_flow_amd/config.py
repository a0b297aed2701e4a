"""Configuration types crossing the flow boundary.

`OpticalFlowCalculationConfig` mirrors /root/reference/optical_flow/config.py:174-188 field for field
(defaults pinned by tests/golden/reference_host_side.json).  `EngineConfig` is this engine's own
knob set (every cv2.DualTVL1OpticalFlow parameter plus device/batch) and never changes the old type.
"""
from dataclasses import dataclass


@dataclass
class OpticalFlowCalculationConfig:
    """Configuration for optical flow calculation and processing."""
    lambda_value: float = 0.15
    moving_avg_window: int = 4
    moving_avg_threshold: float = 0.49
    min_mask_size: int = 500
    waveform_flatness_threshold: float = 0.05
    pap_max_mean: float = 100.0
    cvp_max_mean: float = 50.0
    cvp_min_mean: float = -10.0
    ecg_sampling_rate: int = 500
    art_sampling_rate: int = 125
    cvp_sampling_rate: int = 125
    pap_sampling_rate: int = 125


def default_optical_flow_config() -> OpticalFlowCalculationConfig:
    """Create default optical flow calculation configuration (reference config.py:191-193)."""
    return OpticalFlowCalculationConfig()


@dataclass
class EngineConfig:
    """All DualTVL1 parameters (cv2.optflow.createOptFlow_DualTVL1 defaults) + engine placement."""
    tau: float = 0.25
    lambda_: float = 0.15
    theta: float = 0.3
    nscales: int = 5
    warps: int = 5
    epsilon: float = 0.01
    inner_iterations: int = 30
    outer_iterations: int = 10
    scale_step: float = 0.8
    gamma: float = 0.0
    median_filtering: int = 5
    use_initial_flow: bool = False
    device_id: int = 0
    max_batch: int = 128
