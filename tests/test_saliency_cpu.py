"""The no_saliency=False preprocessing (reference calculate_optical_flow.py:559-560, :586): oracle/saliency_oracle.c against an
independent numpy restatement of the same published steps and against closed forms.  PARITY UNPINNED vs OpenCV (cv2.saliency is
not importable here and the reference holds no saliency fixture): these tests pin the C oracle to the written-down algorithm,
not to cv2."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

NEIGHBOURHOODS = (12, 24, 48, 28, 56, 112)


def _u8_trunc(v):
    """(uchar)double the x86 way: not representable in int32 -> 0, else low byte of the truncated value."""
    v = np.asarray(v, np.float64)
    bad = ~np.isfinite(v) | (v >= 2147483648.0) | (v <= -2147483649.0)
    t = np.trunc(np.where(bad, 0.0, v)).astype(np.int64)
    return (t & 0xFF).astype(np.uint8)


def np_gray(img):
    if img.ndim == 2:
        return img.copy()
    c = img.astype(np.int64)
    return ((c[..., 0] * 3735 + c[..., 1] * 19235 + c[..., 2] * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def np_blur3(g):
    def pad(a, axis):
        n = a.shape[axis]
        if n == 1:
            return np.concatenate([a, a, a], axis=axis)
        return np.pad(a, [(1, 1) if ax == axis else (0, 0) for ax in range(2)], mode="reflect")
    a = pad(g.astype(np.int64), 1)
    h = a[:, :-2] + 2 * a[:, 1:-1] + a[:, 2:]
    b = pad(h, 0)
    return ((b[:-2] + 2 * b[1:-1] + b[2:] + 8) >> 4).astype(np.uint8)


def np_integral(g):
    H, W = g.shape
    I = np.zeros((H + 1, W + 1), np.float32)
    rp = np.cumsum(g.astype(np.int64), axis=1).astype(np.float32)     # exact: a row sums to < 2^24
    for y in range(H):
        I[y + 1, 1:] = I[y, 1:] + rp[y]                                 # float32 + float32, one rounding
    return I


def np_saliency(img):
    g = np_blur3(np_blur3(np_gray(img)))
    H, W = g.shape
    I = np_integral(g)
    ys, xs = np.mgrid[0:H, 0:W]
    gf = g.astype(np.float32)
    mon = np.zeros((H, W), np.int64); moff = np.zeros((H, W), np.int64)
    for nb in NEIGHBOURHOODS:
        x1 = np.clip(xs - nb + 1, 0, W); y1 = np.clip(ys - nb + 1, 0, H)
        x2 = np.clip(xs + nb + 1, 0, W); y2 = np.clip(ys + nb + 1, 0, H)
        v = ((I[y2, x2] + I[y1, x1]) - I[y2, x1]) - I[y1, x2]
        with np.errstate(divide="ignore", invalid="ignore"):
            v = (v - gf) / ((x2 - x1) * (y2 - y1) - 1).astype(np.float32)
        assert v.dtype == np.float32
        on = gf - v; off = v - gf
        mon += np.where(on > 0, _u8_trunc(on), 0); moff += np.where(off > 0, _u8_trunc(off), 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        ion = _u8_trunc(255.0 * (mon.astype(np.float32) / np.float32(mon.max())).astype(np.float64))
        ioff = _u8_trunc(255.0 * (moff.astype(np.float32) / np.float32(moff.max())).astype(np.float64))
        m = max(int(ion.max()), int(ioff.max()))
        s = (ion.astype(np.int64) + ioff.astype(np.int64)).astype(np.float32).astype(np.float64)
        return _u8_trunc(255.0 * s / np.float64(np.float32(m)))


def _cases():
    rng = np.random.default_rng(11)
    yield "noise rgb 40x64", rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)
    yield "noise gray 33x17", rng.integers(0, 256, (33, 17), dtype=np.uint8)
    yy, xx = np.mgrid[0:96, 0:130]
    blob = (255 * np.exp(-((yy - 40) ** 2 + (xx - 70) ** 2) / 300.0)).astype(np.uint8)
    yield "blob 96x130", np.repeat(blob[..., None], 3, 2)
    step = np.zeros((50, 50), np.uint8); step[:, 25:] = 200
    yield "step 50x50", step
    yield "row 1x40", rng.integers(0, 256, (1, 40), dtype=np.uint8)
    yield "column 37x1", rng.integers(0, 256, (37, 1), dtype=np.uint8)
    yield "2x2", rng.integers(0, 256, (2, 2, 3), dtype=np.uint8)
    yield "wide 24x300 (all six neighbourhoods clamp differently)", rng.integers(0, 256, (24, 300, 3), dtype=np.uint8)
    big = rng.integers(0, 256, (260, 260), dtype=np.uint8)
    big[60:200, 60:200] = 255                           # integral image passes 2^24: its float rounding is exercised
    yield "bright 260x260 (integral > 2^24)", big


@pytest.mark.parametrize("name,img", list(_cases()), ids=[n for n, _ in _cases()])
def test_oracle_equals_numpy_restatement(name, img):
    got = O.saliency_fine_grained(img)
    ref = np_saliency(img)
    assert got.dtype == np.uint8 and got.shape == img.shape[:2]
    assert np.array_equal(got, ref), f"{name}: {np.count_nonzero(got != ref)} of {got.size} bytes differ"


def test_building_blocks_closed_forms():
    rng = np.random.default_rng(3)
    # gray frames stay what they are under the BGR2GRAY weights (they sum to 2^15), whatever the channel order
    g = rng.integers(0, 256, (9, 13), dtype=np.uint8)
    gray, _, _ = O.saliency_parts(np.repeat(g[..., None], 3, 2))
    assert np.array_equal(gray, g)
    # pure channels: round(255 * 0.114) on the first channel, 0.587 on the second, 0.299 on the third
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    assert O.saliency_parts(px)[0].tolist() == [[29, 150, 76]]
    # a flat image is a fixed point of the blur; an impulse of 16 spreads as the 1-2-1 x 1-2-1 kernel (one pass shown via two: 4-tap moments)
    flat = np.full((7, 9), 131, np.uint8)
    _, b, I = O.saliency_parts(flat)
    assert np.array_equal(b, flat)
    assert I[0].max() == 0 and I[:, 0].max() == 0 and I[7, 9] == 131 * 63
    imp = np.zeros((9, 9), np.uint8); imp[4, 4] = 255
    _, b2, _ = O.saliency_parts(imp)
    once = np_blur3(imp)
    assert once[3:6, 3:6].tolist() == [[16, 32, 16], [32, 64, 32], [16, 32, 16]]      # (255 * w + 8) >> 4
    assert np.array_equal(b2, np_blur3(once))
    # the float integral image of a bright frame is NOT the exact sum once it passes 2^24, and the oracle keeps the raster-order rounding
    bright = np.full((300, 300), 255, np.uint8)
    _, _, I = O.saliency_parts(bright)
    exact = 255 * 300 * 300
    assert exact > 1 << 24 and I[300, 300] == np_integral(bright)[300, 300]


def test_flat_and_contrast_images():
    # no contrast: every surround mean equals the pixel, both sums are 0, 0/0 -> 0 everywhere
    assert O.saliency_fine_grained(np.full((30, 30, 3), 90, np.uint8)).max() == 0
    # one bright square on black: the square is "on" (brighter than its surround), its surround "off"; the map is not flat
    img = np.zeros((120, 120), np.uint8); img[50:70, 50:70] = 255
    s = O.saliency_fine_grained(img)
    assert s[60, 60] > s[5, 5] > 0          # the far corner still sees the square inside its 112-px surround: faintly "off"
    assert s[60, 60] >= s[60, 40]
    # rows and columns are treated alike (while the integral image stays exact, i.e. below 2^24): transposing commutes
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (70, 45), dtype=np.uint8)
    assert np.array_equal(O.saliency_fine_grained(np.ascontiguousarray(t.T)), O.saliency_fine_grained(t).T)


def test_pipeline_saliency_branch_needs_the_device_engine():
    import tee_optical_flow_amd as T
    from tee_optical_flow_amd.pipeline import process_video
    nparr = np.random.default_rng(0).integers(0, 256, (3, 32, 32, 3), dtype=np.uint8)

    class NoSaliency:
        def calc_batch(self, frames):
            raise AssertionError("the gray-frame path must not be taken when no_saliency=False")

    with pytest.raises(T.OpticalFlowCalculationError, match="no CPU saliency path"):
        process_video(None, None, None, verbose=False, mode="otsu", no_saliency=False, nparr=nparr, flow_model=NoSaliency())

    seen = {}

    class WithSaliency:
        def calc_study_saliency(self, rgb, scale=1.0, pad_last=False, map_dtype="f32"):
            seen.update(shape=rgb.shape, map_dtype=map_dtype, pad_last=pad_last)
            return np.zeros((rgb.shape[0] - (0 if pad_last else 1),) + rgb.shape[1:3] + (2,), np.float32)

    out = process_video(None, None, None, verbose=False, mode="otsu", no_saliency=False, nparr=nparr, flow_model=WithSaliency())
    assert seen["shape"] == (3, 32, 32, 3) and out.shape == (3, 32, 32, 2)
    assert seen["map_dtype"] == "f32" and seen["pad_last"] is True          # the default hand-over: computeSaliency()'s CV_32F map (opencv-contrib >= 4.5)
    process_video(None, None, None, verbose=False, mode="otsu", no_saliency=False, nparr=nparr, flow_model=WithSaliency(), saliency_map="u8")
    assert seen["map_dtype"] == "u8"
    with pytest.raises(T.OpticalFlowCalculationError, match="uint8 RGB frames"):
        process_video(None, None, None, verbose=False, mode="otsu", no_saliency=False, nparr=nparr.astype(np.float32), flow_model=WithSaliency())
