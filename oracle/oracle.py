"""ctypes wrapper over oracle/libtvl1_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (tee_optical_flow_amd) never imports it and has no CPU fallback.

See tvl1_oracle.c for what is restated (reference call sites
/root/reference/optical_flow/calculate_optical_flow.py:577-578, 642) and for the
"PARITY UNPINNED vs real OpenCV" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtvl1_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [("tau", C.c_double), ("lambda_", C.c_double), ("theta", C.c_double),
                ("epsilon", C.c_double), ("scale_step", C.c_double), ("gamma", C.c_double),
                ("nscales", C.c_int), ("warps", C.c_int), ("inner_iterations", C.c_int),
                ("outer_iterations", C.c_int), ("median_filtering", C.c_int),
                ("use_initial_flow", C.c_int), ("err_mode", C.c_int), ("variant", C.c_int)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    libs = [_LIB_PATH, os.path.join(_HERE, "libdeepflow_oracle.so"), os.path.join(_HERE, "libsaliency_oracle.so"),
            os.path.join(_HERE, "libteeflow_cpu.so")]
    srcs = [os.path.join(_HERE, f) for f in ("tvl1_oracle.c", "deepflow_oracle.c", "saliency_oracle.c", "teeflow_cpu_abi.c")]
    if force or not all(os.path.exists(l) for l in libs) or \
            min(os.path.getmtime(l) for l in libs) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def effective_cpus():
    """CPUs this process may really use: min(affinity mask, cgroup quota). A GPU box exposes every hardware
    thread of the host but grants a small CPU share; an OpenMP team sized by nproc would spin on it."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # never spin on an oversubscribed box
        L = C.CDLL(_LIB_PATH)
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_resize_linear.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_resize_cuda.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_scaled_size.argtypes = [C.c_int, C.c_double]
        L.orc_scaled_size.restype = C.c_int
        L.orc_centered_gradient.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.orc_bicubic_tab.argtypes = [fp]
        L.orc_warp.argtypes = [fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp]
        L.orc_warp_cuda.argtypes = [fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp]
        L.orc_remap_bicubic.argtypes = [fp, C.c_int, C.c_int, fp, fp, fp]
        L.orc_median_blur.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp]
        L.orc_iterate.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, C.c_int, C.c_int,
                                  C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.orc_tvl1_calc.argtypes = [C.POINTER(OrcParams), u8p, u8p, C.c_int, C.c_int, fp, C.c_void_p]
        L.orc_tvl1_calc.restype = C.c_int
        L.orc_tvl1_calc_f32.argtypes = [C.POINTER(OrcParams), fp, fp, C.c_int, C.c_int, fp, C.c_void_p]
        L.orc_tvl1_calc_f32.restype = C.c_int
        L.orc_pyramid_level.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_set_num_threads(min(effective_cpus(), 16))
        _lib = L
    return _lib


def default_params(**over):
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, "lambda_" if k == "lambda" else k, v)
    return p


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def resize_linear(src, dw, dh, inv_scale_x=None, inv_scale_y=None):
    src = _f32(src)
    sh, sw = src.shape
    if inv_scale_x is None:
        inv_scale_x = dw / sw
    if inv_scale_y is None:
        inv_scale_y = dh / sh
    dst = np.empty((dh, dw), np.float32)
    lib().orc_resize_linear(src, sw, sh, dst, dw, dh, float(inv_scale_x), float(inv_scale_y))
    return dst


def resize_cuda(src, dw, dh, inv_scale_x=None, inv_scale_y=None):
    """cv::cuda::resize INTER_LINEAR sampling (variant 1 only, [UPSTREAM-FROM-MEMORY]): no half-pixel shift, replicate at the far edges."""
    src = _f32(src)
    sh, sw = src.shape
    if inv_scale_x is None:
        inv_scale_x = dw / sw
    if inv_scale_y is None:
        inv_scale_y = dh / sh
    dst = np.empty((dh, dw), np.float32)
    lib().orc_resize_cuda(src, sw, sh, dst, dw, dh, float(inv_scale_x), float(inv_scale_y))
    return dst


def scaled_size(s, f):
    return lib().orc_scaled_size(int(s), float(f))


def centered_gradient(src):
    src = _f32(src)
    h, w = src.shape
    dx = np.empty_like(src)
    dy = np.empty_like(src)
    lib().orc_centered_gradient(src, w, h, dx, dy)
    return dx, dy


def bicubic_tab():
    t = np.empty((32, 4), np.float32)
    lib().orc_bicubic_tab(t)
    return t


def remap_bicubic(src, mapx, mapy):
    src = _f32(src)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().orc_remap_bicubic(src, w, h, _f32(mapx), _f32(mapy), dst)
    return dst


def warp(I0, I1, u1, u2):
    """One TV-L1 warp stage: returns (I1wx, I1wy, grad, rho_c)."""
    I0, I1, u1, u2 = map(_f32, (I0, I1, u1, u2))
    h, w = I0.shape
    I1x, I1y = centered_gradient(I1)
    outs = [np.empty_like(I0) for _ in range(4)]
    lib().orc_warp(I0, I1, I1x, I1y, u1, u2, w, h, *outs)
    return tuple(outs)


def median_blur(src, ksize=5):
    src = _f32(src)
    h, w = src.shape
    dst = np.empty_like(src)
    lib().orc_median_blur(src, w, h, int(ksize), dst)
    return dst


def iterate(I1wx, I1wy, grad, rho_c, u1, u2, p11, p12, p21, p22, nsteps, lam=0.15, theta=0.3, tau=0.25):
    """Run nsteps inner iterations; returns (u1,u2,p11,p12,p21,p22, err_q[uint64 nsteps])."""
    c = [_f32(a) for a in (I1wx, I1wy, grad, rho_c)]
    s = [_f32(a).copy() for a in (u1, u2, p11, p12, p21, p22)]
    h, w = c[0].shape
    err = np.zeros(nsteps, np.uint64)
    lib().orc_iterate(*c, *s, w, h, float(lam), float(theta), float(tau), int(nsteps),
                      err.ctypes.data_as(C.c_void_p))
    return (*s, err)


def tvl1_calc(I0, I1, params=None, return_iters=False):
    """Full DualTVL1 on one uint8 (or float32 in [0,1], CV_32FC1) pair -> float32 [H,W,2]."""
    f32 = np.asarray(I0).dtype == np.float32
    I0 = np.ascontiguousarray(I0, dtype=np.float32 if f32 else np.uint8)
    I1 = np.ascontiguousarray(I1, dtype=np.float32 if f32 else np.uint8)
    assert I0.shape == I1.shape and I0.ndim == 2
    p = params if params is not None else default_params()
    h, w = I0.shape
    flow = np.empty((h, w, 2), np.float32)
    iters = np.zeros((p.nscales, max(p.warps, 1), 2), np.int32)
    if f32:
        rc = lib().orc_tvl1_calc_f32(C.byref(p), I0.reshape(-1), I1.reshape(-1), h, w, flow.reshape(-1), iters.ctypes.data_as(C.c_void_p))
    else:
        rc = lib().orc_tvl1_calc(C.byref(p), I0, I1, h, w, flow.reshape(-1), iters.ctypes.data_as(C.c_void_p))
    if rc <= 0:
        raise RuntimeError(f"orc_tvl1_calc failed rc={rc}")
    if return_iters:
        return flow, iters[:, :p.warps], rc
    return flow


def pyramid_level(img, level, scale_step=0.8):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    ow, oh = C.c_int(), C.c_int()
    lib().orc_pyramid_level(img, h, w, float(scale_step), int(level), None, C.byref(ow), C.byref(oh))
    out = np.empty((oh.value, ow.value), np.float32)
    lib().orc_pyramid_level(img, h, w, float(scale_step), int(level), out.ctypes.data_as(C.c_void_p),
                            C.byref(ow), C.byref(oh))
    return out


# ----------------------------------------------------------------------------------------------------------------------
# DeepFlow oracle (oracle/deepflow_oracle.c) -- same rules: test infrastructure only
# ----------------------------------------------------------------------------------------------------------------------
class DfoParams(C.Structure):
    _fields_ = [("sigma", C.c_float), ("min_size", C.c_int), ("downscale_factor", C.c_float),
                ("fixed_point_iterations", C.c_int), ("sor_iterations", C.c_int), ("alpha", C.c_float), ("delta", C.c_float),
                ("gamma", C.c_float), ("omega", C.c_float), ("zeta", C.c_float), ("epsilon", C.c_float)]


_dlib = None


def dlib():
    global _dlib
    if _dlib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        L = C.CDLL(os.path.join(_HERE, "libdeepflow_oracle.so"))
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.dfo_default_params.argtypes = [C.POINTER(DfoParams)]
        L.dfo_resize_linear.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.dfo_gauss3.argtypes = [C.c_float, fp]
        L.dfo_gauss_blur3.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp]
        L.dfo_warp_linear.argtypes = [fp, C.c_int, C.c_int, fp, fp, fp]
        L.dfo_derivatives.argtypes = [fp, fp, C.c_int, C.c_int, fp, fp] + [fp] * 8
        L.dfo_variational_refine.argtypes = [C.POINTER(DfoParams), C.c_float, C.c_float, C.c_float, fp, fp, C.c_int, C.c_int, fp, fp]
        L.dfo_pyramid_sizes.argtypes = [C.POINTER(DfoParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.dfo_pyramid_sizes.restype = C.c_int
        L.dfo_deepflow_calc.argtypes = [C.POINTER(DfoParams), u8p, u8p, C.c_int, C.c_int, fp]
        L.dfo_deepflow_calc_f32.argtypes = [C.POINTER(DfoParams), fp, fp, C.c_int, C.c_int, fp]
        L.dfo_deepflow_calc_f32.restype = C.c_int
        L.dfo_deepflow_calc.restype = C.c_int
        lib()  # sets the OpenMP team size once for the process (shared libgomp)
        _dlib = L
    return _dlib


def deepflow_default_params(**over):
    p = DfoParams()
    dlib().dfo_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def deepflow_pyramid_sizes(W, H, params=None):
    p = params if params is not None else deepflow_default_params()
    ws = np.zeros(256, np.int32)
    hs = np.zeros(256, np.int32)
    n = dlib().dfo_pyramid_sizes(C.byref(p), int(W), int(H), ws.ctypes.data_as(C.c_void_p), hs.ctypes.data_as(C.c_void_p), 256)
    return list(zip(ws[:n].tolist(), hs[:n].tolist()))


def deepflow_gauss_blur3(src, sigma=0.6):
    src = _f32(src)
    h, w = src.shape
    dst = np.empty_like(src)
    dlib().dfo_gauss_blur3(src, w, h, float(sigma), dst)
    return dst


def deepflow_warp_linear(I1, u, v):
    I1, u, v = map(_f32, (I1, u, v))
    h, w = I1.shape
    dst = np.empty_like(I1)
    dlib().dfo_warp_linear(I1, w, h, u, v, dst)
    return dst


def deepflow_derivatives(I0, I1, u, v):
    I0, I1, u, v = map(_f32, (I0, I1, u, v))
    h, w = I0.shape
    outs = [np.empty_like(I0) for _ in range(8)]
    dlib().dfo_derivatives(I0, I1, w, h, u, v, *outs)
    return outs   # Ix, Iy, Iz, Ixx, Ixy, Iyy, Ixz, Iyz


def deepflow_variational_refine(I0, I1, u, v, alpha=4.0, delta=0.5 / 3, gamma=5.0 / 3, params=None):
    p = params if params is not None else deepflow_default_params()
    I0, I1 = _f32(I0), _f32(I1)
    u, v = _f32(u).copy(), _f32(v).copy()
    h, w = I0.shape
    dlib().dfo_variational_refine(C.byref(p), np.float32(alpha), np.float32(delta), np.float32(gamma), I0, I1, w, h, u, v)
    return u, v


def deepflow_calc(I0, I1, params=None, return_levels=False):
    """uint8 frames, or float32 frames taken as they are (cv2: convertTo(CV_32F) without a factor)."""
    f32 = np.asarray(I0).dtype == np.float32
    I0 = np.ascontiguousarray(I0, dtype=np.float32 if f32 else np.uint8)
    I1 = np.ascontiguousarray(I1, dtype=np.float32 if f32 else np.uint8)
    p = params if params is not None else deepflow_default_params()
    h, w = I0.shape
    flow = np.empty((h, w, 2), np.float32)
    fn = dlib().dfo_deepflow_calc_f32 if f32 else dlib().dfo_deepflow_calc
    n = fn(C.byref(p), I0.reshape(-1) if f32 else I0, I1.reshape(-1) if f32 else I1, h, w, flow.reshape(-1))
    if n <= 0:
        raise RuntimeError(f"dfo_deepflow_calc failed rc={n}")
    return (flow, n) if return_levels else flow


# ----------------------------------------------------------------------------------------------------------------------
# the timed baseline build (-O3 -march=native, `make o3`): same sources, compiled on the machine that times them
# ----------------------------------------------------------------------------------------------------------------------
def build_o3():
    """Compile the -O3 -march=native variants here and now (they are machine-specific and never travel).  Returns an error string or None."""
    r = subprocess.run(["make", "-C", _HERE, "-s", "-B", "o3"], capture_output=True, text=True)
    return None if r.returncode == 0 else (r.stderr or r.stdout)[-400:]


def o3_calc(algo):
    """calc(I0, I1) -> flow through the -O3 build, or None when it has not been built on this machine."""
    path = os.path.join(_HERE, "libtvl1_oracle_o3.so" if algo == "TVL1" else "libdeepflow_oracle_o3.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    if algo == "TVL1":
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_tvl1_calc.argtypes = [C.POINTER(OrcParams), u8p, u8p, C.c_int, C.c_int, fp, C.c_void_p]
        L.orc_tvl1_calc.restype = C.c_int
        p = OrcParams()
        L.orc_default_params(C.byref(p))

        def calc(I0, I1):           # thread count: set_num_threads() (one OpenMP runtime serves every oracle library of the process)
            h, w = I0.shape
            flow = np.empty((h, w, 2), np.float32)
            if L.orc_tvl1_calc(C.byref(p), np.ascontiguousarray(I0), np.ascontiguousarray(I1), h, w, flow.reshape(-1), None) <= 0:
                raise RuntimeError("orc_tvl1_calc (o3) failed")
            return flow
        return calc
    L.dfo_default_params.argtypes = [C.POINTER(DfoParams)]
    L.dfo_deepflow_calc.argtypes = [C.POINTER(DfoParams), u8p, u8p, C.c_int, C.c_int, fp]
    L.dfo_deepflow_calc.restype = C.c_int
    p = DfoParams()
    L.dfo_default_params(C.byref(p))

    def calc(I0, I1):
        h, w = I0.shape
        flow = np.empty((h, w, 2), np.float32)
        if L.dfo_deepflow_calc(C.byref(p), np.ascontiguousarray(I0), np.ascontiguousarray(I1), h, w, flow.reshape(-1)) <= 0:
            raise RuntimeError("dfo_deepflow_calc (o3) failed")
        return flow
    return calc


# ---- fine-grained static saliency (no_saliency=False preprocessing), saliency_oracle.c ----
_slib = None


def slib():
    global _slib
    if _slib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "libsaliency_oracle.so"))
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.orc_saliency_fine_grained.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_saliency_fine_grained.restype = C.c_int
        L.orc_saliency_fine_grained_f32.argtypes = [u8p, C.c_int, C.c_int, C.c_int, fp]
        L.orc_saliency_fine_grained_f32.restype = C.c_int
        L.orc_sal_gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_sal_blur3.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.orc_sal_integral.argtypes = [u8p, C.c_int, C.c_int, fp]
        for f in (L.orc_sal_gray, L.orc_sal_blur3, L.orc_sal_integral):
            f.restype = None
        _slib = L
    return _slib


def saliency_fine_grained(img, dtype=np.uint8):
    """uint8 [H,W,3] or [H,W] -> the map of StaticSaliencyFineGrained: uint8 [H,W] (the algorithm's 8-bit map) or, dtype=np.float32,
    what computeSaliency() returns in opencv-contrib 4.x: that map * (1/255) as float32 in [0,1]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    H, W = img.shape[:2]
    f32 = np.dtype(dtype) == np.float32
    out = np.empty((H, W), np.float32 if f32 else np.uint8)
    fn = slib().orc_saliency_fine_grained_f32 if f32 else slib().orc_saliency_fine_grained
    if fn(img, H, W, ch, out.reshape(-1) if f32 else out) != 0:
        raise ValueError(f"saliency oracle rejected an image of shape {img.shape}")
    return out


def saliency_parts(img):
    """(gray, blurred twice, float integral image) of the same pipeline, for step-by-step tests."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    H, W = img.shape[:2]
    L = slib()
    g = np.empty((H, W), np.uint8); t = np.empty_like(g); b = np.empty_like(g)
    L.orc_sal_gray(img, H, W, ch, g)
    L.orc_sal_blur3(g, H, W, t)
    L.orc_sal_blur3(t, H, W, b)
    I = np.empty((H + 1, W + 1), np.float32)
    L.orc_sal_integral(b, H, W, I)
    return g, b, I
