"""ctypes binding of libteeflow_hip.so (include/teeflow.h).  There is no CPU fallback: if the HIP
library is missing or no gfx950 device is visible, every entry point raises."""
import ctypes as C
import os

from .exceptions import OpticalFlowCalculationError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TEEFLOW_LIB", os.path.join(_HERE, "libteeflow_hip.so"))   # override only for A/B experiments

TF_OK = 0
COMM_ID_BYTES = 128          # TF_COMM_ID_BYTES
PARAM_KEYS = {
    "tau": 0, "lambda": 1, "theta": 2, "nscales": 3, "warps": 4, "epsilon": 5, "inner_iterations": 6,
    "outer_iterations": 7, "scale_step": 8, "gamma": 9, "median_filtering": 10, "use_initial_flow": 11,
}

# every symbol include/teeflow.h declares (tests check that the built library exports all of them)
EXPORTED_SYMBOLS = [
    "tf_abi_version", "tf_device_count", "tf_default_params", "tf_create", "tf_destroy", "tf_set_param",
    "tf_get_param", "tf_set_stream", "tf_set_profile", "tf_calc_pair", "tf_calc_seq", "tf_calc_pairs", "tf_calc_pair_f32", "tf_calc_pairs_f32",
    "tf_calc_pairs_device", "tf_calc_seq_device", "tf_submit_pairs_device", "tf_submit_seq_device", "tf_submit_pairs", "tf_submit_seq", "tf_submit_seq_rgb", "tf_wait", "tf_condition_frames", "tf_calc_seq_rgb", "tf_saliency_frames", "tf_saliency_frames_f32", "tf_calc_seq_saliency", "tf_calc_seq_saliency_f32", "tf_radlong_project", "tf_radlong_hist", "tf_radlong_select", "tf_get_iters", "tf_last_error",
    "tf_set_tuning", "tf_dbg_counter", "tf_default_deepflow_params", "tf_create_deepflow", "tf_dbg_df_refine", "tf_dbg_df_blur", "tf_dbg_launch_profile", "tf_dbg_strip_rule", "tf_wase_compensate", "tf_wase_compensate_device", "tf_host_alloc", "tf_host_free",
    "tf_dbg_pyramid", "tf_dbg_resize", "tf_dbg_warp", "tf_dbg_median", "tf_dbg_iterate",
    "tf_comm_unique_id", "tf_comm_init_rank", "tf_comm_init_all", "tf_allgather_flows", "tf_allgather_flows_all", "tf_comm_wait", "tf_comm_destroy",
]


class TfParams(C.Structure):
    _fields_ = [("tau", C.c_double), ("lambda_", C.c_double), ("theta", C.c_double), ("epsilon", C.c_double),
                ("scale_step", C.c_double), ("gamma", C.c_double), ("nscales", C.c_int), ("warps", C.c_int),
                ("inner_iterations", C.c_int), ("outer_iterations", C.c_int), ("median_filtering", C.c_int),
                ("use_initial_flow", C.c_int), ("algo", C.c_int), ("max_batch", C.c_int), ("variant", C.c_int)]


class TfDeepflowParams(C.Structure):
    _fields_ = [("sigma", C.c_float), ("min_size", C.c_int), ("downscale_factor", C.c_float), ("fixed_point_iterations", C.c_int),
                ("sor_iterations", C.c_int), ("alpha", C.c_float), ("delta", C.c_float), ("gamma", C.c_float), ("omega", C.c_float),
                ("zeta", C.c_float), ("epsilon", C.c_float), ("max_batch", C.c_int)]


class TfStats(C.Structure):
    _fields_ = [("n_pairs", C.c_int), ("nscales_used", C.c_int), ("warps", C.c_int), ("reserved0", C.c_int),
                ("ms_total", C.c_double), ("ms_h2d", C.c_double), ("ms_device", C.c_double), ("ms_d2h", C.c_double),
                ("iter_launches", C.c_ulonglong), ("iter_pair_steps", C.c_ulonglong), ("iter_ms", C.c_double),
                ("iter_bytes", C.c_double), ("total_bytes", C.c_double), ("inner_iters_total", C.c_ulonglong),
                ("outer_iters_total", C.c_ulonglong), ("ms_warp", C.c_double), ("ms_median", C.c_double),
                ("ms_misc", C.c_double), ("ms_sched", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved0"}


_lib = None


def load():
    """Load the HIP library or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OpticalFlowCalculationError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tee_optical_flow_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    # The CPU checker (oracle/libteeflow_cpu.so, test infrastructure) exports the same ABI.  It must never stand in for the
    # product: refuse anything that lives under an oracle/ directory or carries the checker's own entry points.
    real = os.path.realpath(LIB_PATH)
    if "oracle" in real.split(os.sep)[:-1]:
        raise OpticalFlowCalculationError(f"{LIB_PATH} resolves into an oracle/ directory: the CPU checker is test infrastructure, not a backend")
    L = C.CDLL(LIB_PATH)
    if hasattr(L, "orc_set_num_threads") or hasattr(L, "orc_tvl1_calc"):
        raise OpticalFlowCalculationError(f"{LIB_PATH} is the CPU checker library (it exports orc_* symbols): there is no CPU backend")
    vp, i32, f32, dbl = C.c_void_p, C.c_int, C.c_float, C.c_double
    L.tf_abi_version.restype = i32
    L.tf_device_count.restype = i32
    L.tf_default_params.argtypes = [C.POINTER(TfParams)]
    L.tf_create.argtypes = [C.POINTER(TfParams), i32, C.POINTER(vp)]
    L.tf_destroy.argtypes = [vp]
    L.tf_destroy.restype = None
    L.tf_set_param.argtypes = [vp, i32, dbl]
    L.tf_get_param.argtypes = [vp, i32, C.POINTER(dbl)]
    L.tf_set_stream.argtypes = [vp, vp, i32]
    L.tf_set_profile.argtypes = [vp, i32]
    L.tf_set_tuning.argtypes = [vp, C.c_char_p, i32]
    L.tf_dbg_counter.argtypes = [vp, C.c_char_p]
    L.tf_dbg_counter.restype = C.c_longlong
    L.tf_default_deepflow_params.argtypes = [C.POINTER(TfDeepflowParams)]
    L.tf_create_deepflow.argtypes = [C.POINTER(TfDeepflowParams), i32, C.POINTER(vp)]
    L.tf_dbg_df_refine.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    L.tf_dbg_df_blur.argtypes = [vp, vp, i32, i32, vp]
    L.tf_dbg_launch_profile.argtypes = [vp, vp, vp, vp, vp, i32]
    L.tf_dbg_strip_rule.argtypes = [i32, i32, i32, i32, vp, vp]
    L.tf_dbg_strip_rule.restype = None
    L.tf_host_alloc.argtypes = [C.c_size_t]
    L.tf_host_alloc.restype = vp
    L.tf_host_free.argtypes = [vp]
    L.tf_host_free.restype = None
    L.tf_wase_compensate.argtypes = [vp, vp, i32, vp, i32, i32, i32, C.c_float, vp]
    L.tf_wase_compensate_device.argtypes = [vp, vp, i32, vp, i32, i32, i32, C.c_float, vp]
    L.tf_calc_pair.argtypes = [vp, vp, vp, i32, i32, vp, C.POINTER(TfStats)]
    L.tf_calc_seq.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(TfStats)]
    L.tf_calc_pairs.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.POINTER(TfStats)]
    L.tf_calc_pairs_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.POINTER(TfStats)]
    L.tf_calc_pair_f32.argtypes = [vp, vp, vp, i32, i32, vp, C.POINTER(TfStats)]
    L.tf_calc_pairs_device.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, C.POINTER(TfStats)]
    L.tf_calc_seq_device.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(TfStats)]
    if "TEEFLOW_LIB" not in os.environ or hasattr(L, "tf_wait"):                # (an older A/B build may lack the round-5 entry points)
        L.tf_submit_pairs_device.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, C.POINTER(i32)]
        L.tf_submit_seq_device.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(i32)]
        L.tf_submit_pairs.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.POINTER(i32)]
        L.tf_submit_seq.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(i32)]
        L.tf_submit_seq_rgb.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(i32)]
        L.tf_wait.argtypes = [vp, i32, C.POINTER(TfStats)]
    L.tf_condition_frames.argtypes = [vp, vp, i32, i32, i32, vp]
    if "TEEFLOW_LIB" not in os.environ or hasattr(L, "tf_saliency_frames"):     # (an older A/B build may lack the round-4 entry points)
        L.tf_saliency_frames.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.tf_calc_seq_saliency.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, C.POINTER(TfStats)]
    if "TEEFLOW_LIB" not in os.environ or hasattr(L, "tf_saliency_frames_f32"):
        L.tf_saliency_frames_f32.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.tf_calc_seq_saliency_f32.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, C.POINTER(TfStats)]
    L.tf_calc_seq_rgb.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(TfStats)]
    L.tf_radlong_project.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.tf_radlong_hist.argtypes = [vp, i32, vp, i32, vp]
    L.tf_radlong_select.argtypes = [vp, i32, vp, vp]
    L.tf_get_iters.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tf_last_error.argtypes = [vp]
    L.tf_last_error.restype = C.c_char_p
    L.tf_dbg_pyramid.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    L.tf_dbg_resize.argtypes = [vp, vp, i32, i32, vp, i32, i32, dbl, dbl, f32]
    L.tf_dbg_warp.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.tf_dbg_median.argtypes = [vp, vp, i32, i32, i32, vp]
    L.tf_dbg_iterate.argtypes = [vp] + [vp] * 9 + [i32, i32, i32, i32, vp]
    L.tf_comm_unique_id.argtypes = [vp]
    L.tf_comm_init_rank.argtypes = [vp, i32, i32, vp]
    L.tf_comm_init_all.argtypes = [C.POINTER(vp), i32]
    L.tf_allgather_flows.argtypes = [vp, vp, C.c_size_t, vp, C.POINTER(i32)]
    L.tf_allgather_flows_all.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), C.c_size_t, C.POINTER(vp)]
    L.tf_comm_wait.argtypes = [vp, i32]
    L.tf_comm_destroy.argtypes = [vp]
    for name in EXPORTED_SYMBOLS:
        getattr(L, name)  # AttributeError here = header/library mismatch
    if L.tf_abi_version() != 2:
        raise OpticalFlowCalculationError("libteeflow_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, handle=None, what="teeflow"):
    if rc != TF_OK:
        msg = load().tf_last_error(handle)
        raise OpticalFlowCalculationError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
