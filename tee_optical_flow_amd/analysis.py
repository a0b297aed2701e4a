"""radial / longitudinal projection of the flow and the per-frame percentile / histogram curves.

Mirrors /root/reference/optical_flow/analysis.py:
  :122-134  calc_proj_mag,  :137-163 calculate_comp_magnitude (with radial_vecgrid :89-119)   -> calculate_comp_magnitude
  :166-212  calc_bidirectional_hist                                                          -> calc_bidirectional_hist
Two implementations with identical results: a vectorised numpy one (host) and a device one that keeps the projections in
HBM and gets the histogram and the exact order statistics np.percentile interpolates between from HIP kernels
(include/teeflow.h: tf_radlong_project / _hist / _select).  Pinned by tests/golden/reference_analysis.npz, which the
reference's own functions produced.  The AV-centroid / Savitzky-Golay step stays with the reference (1-D, CPU-trivial).
"""
import ctypes as C

import numpy as np

from . import _lib


# ---- host (numpy) -------------------------------------------------------------------------------------------------------
def calculate_comp_magnitude(OF_arr, centroid_list):
    """(rad_arr, long_arr), float64 [N,H,W].  Reference quirk kept: unitvec[...,0] is the ROW direction and multiplies
    OF[...,0], the x (column) displacement."""
    n = len(centroid_list)
    OF = np.asarray(OF_arr)[:n]
    H, W = OF.shape[1:3]
    c = np.asarray(centroid_list, dtype=np.float64)
    dh = c[:, 0, None, None] - np.arange(H)[None, :, None]
    dw = c[:, 1, None, None] - np.arange(W)[None, None, :]
    dh = np.broadcast_to(dh, (n, H, W))
    dw = np.broadcast_to(dw, (n, H, W))
    with np.errstate(invalid="ignore", divide="ignore"):
        norm = np.sqrt(dh * dh + dw * dw)
        u0 = np.nan_to_num(dh / norm, nan=0)
        u1 = np.nan_to_num(dw / norm, nan=0)
    rad = OF[..., 0] * u0 + OF[..., 1] * u1
    lon = OF[..., 0] * u1 + OF[..., 1] * (-1 * u0)
    return rad, lon


def _finish_hist(n_frames, nbins, mag_min, mag_max, counts, per_frame):
    """The reference's per-frame loop (:188-212) given, for every frame, either None (no non-zero data) or (hi, lo, freq)."""
    hi_l, lo_l, fr_l = [], [], []
    edges = []
    for i in range(n_frames):
        if per_frame[i] is None:
            if hi_l:
                hi_l.append(hi_l[-1]); lo_l.append(lo_l[-1]); fr_l.append(fr_l[-1])
            else:
                hi_l.append(mag_max); lo_l.append(mag_min); fr_l.append(np.ones(nbins))
        else:
            hi, lo, freq = per_frame[i]
            hi_l.append(hi); lo_l.append(lo); fr_l.append(freq + 1)
            edges = np.linspace(mag_min, mag_max, nbins + 1) if mag_min != mag_max else np.linspace(mag_min - 0.5, mag_max + 0.5, nbins + 1)
    return np.stack(fr_l), edges, np.asarray(hi_l), np.asarray(lo_l)


def calc_bidirectional_hist(mag_arr, nframes, perc_lo=1, perc_hi=99, nbins=1000):
    mag_arr = np.asarray(mag_arr)
    mag_max, mag_min = np.max(mag_arr), np.min(mag_arr)
    per = []
    for i in range(nframes):
        flat = np.ravel(mag_arr[i])
        nz = flat[flat != 0]
        if len(nz) == 0:
            per.append(None)
        else:
            freq, _ = np.histogram(nz, bins=nbins, range=(mag_min, mag_max))
            per.append((np.percentile(nz, perc_hi), np.percentile(nz, perc_lo), freq))
    return _finish_hist(nframes, nbins, mag_min, mag_max, None, per)


# ---- device ---------------------------------------------------------------------------------------------------------------
def _lerp(a, b, t):
    """numpy's percentile interpolation (numpy/lib/_function_base_impl.py::_lerp) for scalars."""
    d = b - a
    return b - d * (1 - t) if t >= 0.5 else a + d * t


def radlong_stats_device(engine, OF_arr, centroid_list, perc_lo=1, perc_hi=99, nbins=1000, return_arrays=False):
    """{'radial': (freq, edges[:-1], hi, lo), 'longitudinal': (...)} as calculate_3dhist_radlong (:289-327) returns after its
    centroid step, computed on the device behind `engine` (a DenseFlow).  return_arrays adds 'rad_arr' / 'long_arr'."""
    L = engine._L
    n = len(centroid_list)
    OF = np.ascontiguousarray(np.asarray(OF_arr)[:n], dtype=np.float32)
    N, H, W, _ = OF.shape
    cent = np.ascontiguousarray(centroid_list, dtype=np.float64).reshape(N, 2)
    rad = np.empty((N, H, W), np.float64) if return_arrays else None
    lon = np.empty((N, H, W), np.float64) if return_arrays else None
    mm = np.zeros(4, np.float64)
    nz = np.zeros(2 * N, np.int64)
    _lib.check(L.tf_radlong_project(engine._h, OF.ctypes.data, cent.ctypes.data, N, H, W, rad.ctypes.data if return_arrays else None,
                                    lon.ctypes.data if return_arrays else None, mm.ctypes.data, nz.ctypes.data), engine._h, "tf_radlong_project")
    out = {}
    for which, name in ((0, "radial"), (1, "longitudinal")):
        mn, mx = mm[2 * which], mm[2 * which + 1]
        cnt = nz[which::2]
        first, last = (mn, mx) if mn != mx else (mn - 0.5, mx + 0.5)          # np.histogram's degenerate-range rule
        edges = np.linspace(first, last, nbins + 1)
        freq = np.zeros((N, nbins), np.int64)
        _lib.check(L.tf_radlong_hist(engine._h, which, edges.ctypes.data, nbins, freq.ctypes.data), engine._h, "tf_radlong_hist")
        ranks = np.full((N, 4), -1, np.int64)
        frac = np.zeros((N, 2))
        for i in range(N):
            if cnt[i] > 0:
                for j, q in enumerate((perc_hi, perc_lo)):
                    vi = (cnt[i] - 1) * np.true_divide(q, 100)
                    lo_i = int(np.floor(vi))
                    ranks[i, 2 * j] = lo_i
                    ranks[i, 2 * j + 1] = min(lo_i + 1, cnt[i] - 1)
                    frac[i, j] = vi - lo_i
        vals = np.zeros((N, 4), np.float64)
        _lib.check(L.tf_radlong_select(engine._h, which, ranks.ctypes.data, vals.ctypes.data), engine._h, "tf_radlong_select")
        per = [None if cnt[i] == 0 else (_lerp(vals[i, 0], vals[i, 1], frac[i, 0]), _lerp(vals[i, 2], vals[i, 3], frac[i, 1]), freq[i])
               for i in range(N)]
        f, e, hi, lo = _finish_hist(N, nbins, mn, mx, cnt, per)
        out[name] = (f, np.asarray(e)[:-1], hi, lo)
    if return_arrays:
        out["rad_arr"], out["long_arr"] = rad, lon
    return out
