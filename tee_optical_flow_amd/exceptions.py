"""Exception hierarchy at the flow boundary.

Mirrors /root/reference/optical_flow/exceptions.py:6-33 name for name so callers written against the
reference keep their `except` clauses.
"""


class OpticalFlowError(Exception):
    """Base exception for optical flow processing errors."""


class DICOMReadError(OpticalFlowError):
    """Raised when DICOM file cannot be read."""


class WaveformLoadError(OpticalFlowError):
    """Raised when waveform file cannot be loaded."""


class WaveformValidationError(OpticalFlowError):
    """Raised when waveform validation fails."""


class OpticalFlowCalculationError(OpticalFlowError):
    """Raised when optical flow calculation fails."""


class ConfigurationError(OpticalFlowError):
    """Raised when configuration is invalid."""
