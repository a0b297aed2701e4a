"""Kernel-level parity on the GPU: each HIP kernel, called through the C ABI's tf_dbg_* hooks, against the
oracle's restatement of the same OpenCV step.  Bar: BIT-EXACT (the kernels and the oracle evaluate the same
IEEE operations in the same order, no FMA)."""
import ctypes as C

import numpy as np
import pytest


pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _rng_img(seed, h, w, lo=0.0, hi=255.0):
    return np.random.default_rng(seed).uniform(lo, hi, (h, w)).astype(np.float32)


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (210, 210), (17, 16), (512, 512)])
def test_pyramid_levels_bit_exact(engine, oracle, shape):
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    img = np.random.default_rng(1).integers(0, 256, (h, w), dtype=np.uint8)
    for level in (0, 1, 2):
        ref = oracle.pyramid_level(img, level)
        ow, oh = C.c_int(), C.c_int()
        _lib.check(L.tf_dbg_pyramid(engine._h, _ptr(img), h, w, level, None, C.byref(ow), C.byref(oh)), engine._h)
        assert (oh.value, ow.value) == ref.shape
        out = np.empty(ref.shape, np.float32)
        _lib.check(L.tf_dbg_pyramid(engine._h, _ptr(img), h, w, level, _ptr(out), C.byref(ow), C.byref(oh)), engine._h)
        assert np.array_equal(out, ref), f"level {level}: max diff {np.abs(out - ref).max()}"


@pytest.mark.parametrize("src_shape,dst_shape", [((210, 210), (262, 262)), ((33, 47), (41, 59)), ((16, 16), (20, 20)),
                                                 ((328, 328), (410, 410))])
def test_flow_upsample_resize_bit_exact(engine, oracle, src_shape, dst_shape):
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    sh, sw = src_shape
    dh, dw = dst_shape
    src = _rng_img(2, sh, sw, -5, 5)
    ref = oracle.resize_linear(src, dw, dh) * np.float32(1.25)
    out = np.empty((dh, dw), np.float32)
    _lib.check(L.tf_dbg_resize(engine._h, _ptr(src), sw, sh, _ptr(out), dw, dh, dw / sw, dh / sh, 1.25), engine._h)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("margin", [0, 2, 8])
@pytest.mark.parametrize("shape,amp", [((64, 64), 3.0), ((97, 131), 8.0), ((210, 210), 1.0), ((40, 300), 60.0),
                                       ((20, 18), 30.0)])
def test_warp_bit_exact(engine, oracle, shape, amp, margin):
    """buildFlowMap + 3x remap(INTER_CUBIC, BORDER_CONSTANT) + calcGradRho; `amp` pushes samples across and
    beyond the borders (partial-tap and fully-outside cases)."""
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    rng = np.random.default_rng(3)
    I0, I1 = _rng_img(4, h, w), _rng_img(5, h, w)
    u1 = rng.uniform(-amp, amp, (h, w)).astype(np.float32)
    u2 = rng.uniform(-amp, amp, (h, w)).astype(np.float32)
    u1[0, 0] = 1e6   # far outside (saturate_cast<short> path)
    u2[-1, -1] = -1e6
    r_wx, r_wy, r_grad, r_rho = oracle.warp(I0, I1, u1, u2)
    wx, wy, rho = (np.empty((h, w), np.float32) for _ in range(3))
    engine.set_tuning("warp_margin", margin)   # 0: global gathers; >0: LDS-staged tile + margin with global fallback
    _lib.check(L.tf_dbg_warp(engine._h, _ptr(I0), _ptr(I1), _ptr(u1), _ptr(u2), w, h, _ptr(wx), _ptr(wy), _ptr(rho)), engine._h)
    assert np.array_equal(wx, r_wx)
    assert np.array_equal(wy, r_wy)
    assert np.array_equal(rho, r_rho)
    assert np.array_equal(wx * wx + wy * wy, r_grad)   # |grad|^2 is recomputed in tvl1_iter, never stored
    engine.set_tuning("warp_margin", 0)


@pytest.mark.parametrize("ksize", [3, 5])
@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (5, 7), (1, 40), (210, 210)])
def test_median_bit_exact(engine, oracle, shape, ksize):
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = shape
    src = _rng_img(6, h, w, -3, 3)
    src[::7, ::5] = 0.0  # ties
    ref = oracle.median_blur(src, ksize)
    out = np.empty((h, w), np.float32)
    _lib.check(L.tf_dbg_median(engine._h, _ptr(src), w, h, ksize, _ptr(out)), engine._h)
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("shape", [(64, 64), (97, 131), (210, 210), (15, 60), (16, 61), (31, 121), (512, 512), (3, 1021)])
@pytest.mark.parametrize("pzero", [0, 1])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_iterate_bit_exact(engine, oracle, shape, pzero, variant):
    """k steps of the fused tvl1_iter kernel == k oracle iterations (state AND exact error sums), for all four
    kernel forms: 64x16 tiles, full-width row strips, row strips with two iterations per launch, tiles with two
    iterations per launch (variant 3 = variant 2 on a launch too small for the strips)."""
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    engine.set_tuning("iter_variant", min(variant, 2))
    engine.set_tuning("min_rows_work", 0 if variant != 3 else 1 << 30)   # strips even for this single small image | tiles
    try:
        _iterate_case(engine, oracle, L, shape, pzero)
    finally:
        engine.set_tuning("iter_variant", 2)
        engine.set_tuning("min_rows_work", 8192)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_iterate_tiny_zero_and_denormal_values_bit_exact(engine, oracle, variant):
    """The dual/primal updates where real echo frames put them: next to exactly-black regions the flow and the dual
    variable decay geometrically through 1e-30 into the denormal range.  State, warp constants and rho are scaled by
    10^-k with k up to 44 per 8-px column band, with patches of +0 and -0; results are compared as BIT PATTERNS (so
    the sign of zero and every denormal count)."""
    from tee_optical_flow_amd import _lib
    L = _lib.load()
    h, w = 96, 384
    rng = np.random.default_rng(21)
    k = (np.arange(w) // 8).astype(np.float64)                       # 0 .. 47
    sc = (10.0 ** -k)[None, :]
    def scaled(lo, hi, s):
        return (rng.uniform(lo, hi, (h, w)) * s).astype(np.float32)
    u1, u2 = scaled(-1, 1, sc), scaled(-1, 1, sc)
    p = [scaled(-0.5, 0.5, sc) for _ in range(4)]
    wx, wy = scaled(-20, 20, np.sqrt(sc)), scaled(-20, 20, np.sqrt(sc))
    rho = scaled(-3, 3, sc)
    for a in (u1, u2, *p, wx, wy, rho):
        a[10:20, :] = 0.0
        a[30:34, :] = -0.0
    wx[40:50] = rng.uniform(-20, 20, (10, w)).astype(np.float32)     # ordinary gradients over tiny flow
    wy[40:50] = rng.uniform(-20, 20, (10, w)).astype(np.float32)
    grad = wx * wx + wy * wy
    engine.set_tuning("iter_variant", min(variant, 2))
    engine.set_tuning("min_rows_work", 0 if variant != 3 else 1 << 30)
    try:
        nsteps = 6
        ref = oracle.iterate(wx, wy, grad, rho, u1, u2, *p, nsteps)
        st = [a.copy() for a in (u1, u2, *p)]
        err = np.zeros(nsteps, np.uint64)
        _lib.check(L.tf_dbg_iterate(engine._h, _ptr(wx), _ptr(wy), _ptr(rho), *[_ptr(a) for a in st], w, h, nsteps, 0,
                                    _ptr(err)), engine._h)
    finally:
        engine.set_tuning("iter_variant", 2)
        engine.set_tuning("min_rows_work", 8192)
    for n, a, r in zip(["u1", "u2", "p11", "p12", "p21", "p22"], st, ref[:6]):
        bad = a.view(np.uint32) != r.view(np.uint32)
        assert not bad.any(), f"{n}: {bad.sum()} bit patterns differ, first at {np.argwhere(bad)[0]}: {a[bad][0]!r} vs {r[bad][0]!r}"
    assert np.array_equal(err, ref[6])
    assert np.any((np.abs(ref[2]) < 1e-38) & (ref[2] != 0)), "case no longer reaches the denormal range"


def _iterate_case(engine, oracle, L, shape, pzero):
    from tee_optical_flow_amd import _lib
    h, w = shape
    rng = np.random.default_rng(7)
    I0, I1 = _rng_img(8, h, w), _rng_img(9, h, w)
    # smooth-ish images make all three threshold branches occur
    from scipy import ndimage
    I0 = ndimage.gaussian_filter(I0, 1.5).astype(np.float32)
    I1 = ndimage.gaussian_filter(I1, 1.5).astype(np.float32)
    u1 = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    u2 = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    wx, wy, grad, rho = oracle.warp(I0, I1, u1, u2)
    wx[3:6, 3:9] = 0.0  # grad <= FLT_EPSILON branch
    wy[3:6, 3:9] = 0.0
    grad = wx * wx + wy * wy
    if pzero:
        p = [np.zeros((h, w), np.float32) for _ in range(4)]
    else:
        p = [rng.uniform(-0.5, 0.5, (h, w)).astype(np.float32) for _ in range(4)]
    nsteps = 6
    ref = oracle.iterate(wx, wy, grad, rho, u1, u2, *p, nsteps)
    st = [a.copy() for a in (u1, u2, *p)]
    err = np.zeros(nsteps, np.uint64)
    _lib.check(L.tf_dbg_iterate(engine._h, _ptr(wx), _ptr(wy), _ptr(rho), *[_ptr(a) for a in st], w, h, nsteps, pzero,
                                _ptr(err)), engine._h)
    names = ["u1", "u2", "p11", "p12", "p21", "p22"]
    for n, a, r in zip(names, st, ref[:6]):
        assert np.array_equal(a, r), f"{n}: {np.sum(a != r)} px differ, max {np.abs(a - r).max()}"
    assert np.array_equal(err, ref[6])
