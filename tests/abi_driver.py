"""One sequence of raw C-ABI calls (include/teeflow.h), written once and run against BOTH libraries that export it: the
product (tee_optical_flow_amd/libteeflow_hip.so, MI355X) and the checker (oracle/libteeflow_cpu.so, the CPU restatement
behind the same entry points -- SURVEY.md section 8b).  Test infrastructure."""
import ctypes as C
import os

import numpy as np

from tee_optical_flow_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "oracle", "libteeflow_cpu.so")


def bind(path):
    L = C.CDLL(path)
    if hasattr(L, "orc_set_num_threads"):
        # the checker is OpenMP code: before its first parallel region, a team no larger than the cores this process is
        # granted (a GPU box shows 256 hardware threads and grants 16 -- a default-sized team spins for minutes there)
        from oracle import oracle as O
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(min(O.effective_cpus(), 16))
    vp, i32, f32, dbl = C.c_void_p, C.c_int, C.c_float, C.c_double
    L.tf_abi_version.restype = i32
    L.tf_default_params.argtypes = [C.POINTER(_lib.TfParams)]
    L.tf_create.argtypes = [C.POINTER(_lib.TfParams), i32, C.POINTER(vp)]
    L.tf_default_deepflow_params.argtypes = [C.POINTER(_lib.TfDeepflowParams)]
    L.tf_create_deepflow.argtypes = [C.POINTER(_lib.TfDeepflowParams), i32, C.POINTER(vp)]
    L.tf_destroy.argtypes = [vp]; L.tf_destroy.restype = None
    L.tf_set_param.argtypes = [vp, i32, dbl]
    L.tf_get_param.argtypes = [vp, i32, C.POINTER(dbl)]
    L.tf_calc_pair.argtypes = [vp, vp, vp, i32, i32, vp, C.POINTER(_lib.TfStats)]
    L.tf_calc_seq.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(_lib.TfStats)]
    L.tf_calc_pairs.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.POINTER(_lib.TfStats)]
    L.tf_get_iters.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.tf_submit_pairs.argtypes = [vp, vp, vp, i32, i32, i32, vp, C.POINTER(i32)]
    L.tf_submit_seq.argtypes = [vp, vp, i32, i32, i32, f32, vp, C.POINTER(i32)]
    L.tf_wait.argtypes = [vp, i32, C.POINTER(_lib.TfStats)]
    L.tf_last_error.argtypes = [vp]; L.tf_last_error.restype = C.c_char_p
    return L


def drive(L, frames, pairs):
    """Returns a dict of everything observable through the boundary for this script of calls."""
    out = {"abi": L.tf_abi_version()}
    p = _lib.TfParams()
    assert L.tf_default_params(C.byref(p)) == 0
    out["defaults"] = [getattr(p, k) for k, _ in _lib.TfParams._fields_]
    h = C.c_void_p()
    assert L.tf_create(C.byref(p), 0, C.byref(h)) == 0
    codes = []
    codes.append(L.tf_set_param(h, _lib.PARAM_KEYS["lambda"], 0.15))                 # reference :578
    codes.append(L.tf_set_param(h, _lib.PARAM_KEYS["median_filtering"], 4.0))        # not 1/3/5 -> TF_ERR_UNSUPPORTED
    codes.append(L.tf_set_param(h, _lib.PARAM_KEYS["gamma"], 0.5))                   # unsupported
    codes.append(L.tf_set_param(h, _lib.PARAM_KEYS["scale_step"], 1.5))              # invalid
    codes.append(L.tf_set_param(h, _lib.PARAM_KEYS["scale_step"], 0.5))              # cv::resize's INTER_AREA fast path: not restated -> UNSUPPORTED
    codes.append(L.tf_set_param(h, 99, 1.0))                                         # unknown key
    codes.append(L.tf_calc_pair(h, None, None, 8, 8, None, None))                    # null pointers
    v = C.c_double()
    L.tf_get_param(h, _lib.PARAM_KEYS["median_filtering"], C.byref(v))               # a rejected set leaves the value alone
    out["codes"], out["median_after_bad_set"] = codes, v.value
    I0s, I1s = pairs
    B, H, W = I0s.shape
    st = _lib.TfStats()
    f1 = np.empty((H, W, 2), np.float32)
    assert L.tf_calc_pair(h, I0s[0].ctypes.data, I1s[0].ctypes.data, H, W, f1.ctypes.data, C.byref(st)) == 0   # reference :642
    out["pair_flow"], out["pair_stats"] = f1, (st.n_pairs, st.nscales_used, st.warps, st.inner_iters_total, st.outer_iters_total)
    fb = np.empty((B, H, W, 2), np.float32)
    assert L.tf_calc_pairs(h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, fb.ctypes.data, C.byref(st)) == 0
    n = st.n_pairs * st.nscales_used * st.warps * 2
    it = np.zeros(n, np.int32); w = C.c_size_t()
    assert L.tf_get_iters(h, it.ctypes.data_as(C.c_void_p), n, C.byref(w)) == 0 and w.value == n
    out["pairs_flow"], out["pairs_iters"] = fb, it.reshape(st.n_pairs, st.nscales_used, st.warps, 2)
    N = frames.shape[0]
    assert L.tf_set_param(h, _lib.PARAM_KEYS["warps"], 3.0) == 0 and L.tf_set_param(h, _lib.PARAM_KEYS["epsilon"], 0.03) == 0
    fs = np.empty((N - 1,) + frames.shape[1:] + (2,), np.float32)
    assert L.tf_calc_seq(h, frames.ctypes.data, N, frames.shape[1], frames.shape[2], 2.0, fs.ctypes.data, C.byref(st)) == 0   # :584-600
    out["seq_flow"] = fs
    out["seq_too_short"] = L.tf_calc_seq(h, frames.ctypes.data, 1, frames.shape[1], frames.shape[2], 1.0, fs.ctypes.data, None)
    L.tf_destroy(h)
    # sub-batches of 2 pairs: the 3-pair call is cut in two queue units (the product's lanes; the checker has none), then two jobs
    # are submitted before either is waited for and collected in reverse order
    p.max_batch = 2
    assert L.tf_create(C.byref(p), 0, C.byref(h)) == 0
    fq = np.empty((B, H, W, 2), np.float32)
    assert L.tf_calc_pairs(h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, fq.ctypes.data, C.byref(st)) == 0
    itq = np.zeros(n, np.int32)
    assert L.tf_get_iters(h, itq.ctypes.data_as(C.c_void_p), n, C.byref(w)) == 0 and w.value == n
    out["queued_flow"], out["queued_iters"], out["queued_stats"] = fq, itq, (st.n_pairs, st.nscales_used, st.warps, st.inner_iters_total, st.outer_iters_total)
    fa, fb2 = np.empty((B, H, W, 2), np.float32), np.empty((N - 1,) + frames.shape[1:] + (2,), np.float32)
    ta, tb = C.c_int(-1), C.c_int(-1)
    assert L.tf_submit_pairs(h, I0s.ctypes.data, I1s.ctypes.data, B, H, W, fa.ctypes.data, C.byref(ta)) == 0
    assert L.tf_submit_seq(h, frames.ctypes.data, N, frames.shape[1], frames.shape[2], 0.5, fb2.ctypes.data, C.byref(tb)) == 0
    assert ta.value != tb.value
    assert L.tf_wait(h, tb.value, C.byref(st)) == 0
    out["async_seq_flow"], out["async_seq_pairs"] = fb2, st.n_pairs
    assert L.tf_wait(h, ta.value, C.byref(st)) == 0
    ita = np.zeros(n, np.int32)
    assert L.tf_get_iters(h, ita.ctypes.data_as(C.c_void_p), n, C.byref(w)) == 0 and w.value == n
    out["async_pairs_flow"], out["async_pairs_iters"] = fa, ita
    out["wait_twice"] = L.tf_wait(h, ta.value, None)                                   # unknown ticket -> TF_ERR_INVALID_ARG
    out["wait_all_when_none"] = L.tf_wait(h, -1, None)
    p.max_batch = 0
    L.tf_destroy(h)
    # the CUDA-branch variant (reference :575) and DeepFlow (:568) through the same entry points
    p.variant = 1
    assert L.tf_create(C.byref(p), 0, C.byref(h)) == 0
    fv = np.empty((H, W, 2), np.float32)
    assert L.tf_calc_pair(h, I0s[1].ctypes.data, I1s[1].ctypes.data, H, W, fv.ctypes.data, C.byref(st)) == 0
    out["variant_flow"], out["variant_outer"] = fv, st.outer_iters_total
    L.tf_destroy(h)
    p.variant = 7
    out["bad_variant"] = L.tf_create(C.byref(p), 0, C.byref(h))
    dp = _lib.TfDeepflowParams()
    assert L.tf_default_deepflow_params(C.byref(dp)) == 0
    assert L.tf_create_deepflow(C.byref(dp), 0, C.byref(h)) == 0
    fd = np.empty((H, W, 2), np.float32)
    assert L.tf_calc_pair(h, I0s[0].ctypes.data, I1s[0].ctypes.data, H, W, fd.ctypes.data, C.byref(st)) == 0
    out["deepflow_flow"], out["deepflow_levels"] = fd, st.nscales_used
    out["deepflow_set_param"] = L.tf_set_param(h, 1, 0.2)                            # creation-time parameters only
    L.tf_destroy(h)
    return out
