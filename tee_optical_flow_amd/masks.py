"""Otsu mask path of the reference (mode='otsu'), restated with numpy/scipy only so the study driver runs where
skimage is absent.  Mirrors /root/reference/optical_flow/calculate_optical_flow.py:90-111 (moving_avg_mask) and
:184-213 (predict_movie_thres); pinned by tests/golden/reference_host_side.npz.  Host-side glue, not a kernel."""
import numpy as np

from .config import default_optical_flow_config
from .frames import rgb2gray


def threshold_otsu(image, nbins=256):
    """skimage.filters.threshold_otsu for a float image (histogram over [min, max], bin centres)."""
    image = np.asarray(image, dtype=np.float64)
    hist, edges = np.histogram(image.ravel(), bins=nbins, range=(image.min(), image.max()))
    centers = (edges[:-1] + edges[1:]) / 2.0
    hist = hist.astype(np.float64)
    w1 = np.cumsum(hist)
    w2 = np.cumsum(hist[::-1])[::-1]
    m1 = np.cumsum(hist * centers) / w1
    m2 = (np.cumsum((hist * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[:-1][np.argmax(var12)]


def remove_small_objects(mask, min_size):
    """skimage.morphology.remove_small_objects on a bool image (1-connectivity)."""
    from scipy import ndimage
    lab, n = ndimage.label(mask)
    if n == 0:
        return mask.copy()
    sizes = np.bincount(lab.ravel())
    small = sizes < min_size
    small[0] = False
    out = mask.copy()
    out[small[lab]] = False
    return out


def moving_avg_mask(arr, n=4, threshold=0.49, config=None):
    if config is not None:
        n, threshold = config.moving_avg_window, config.moving_avg_threshold
    arr2 = np.vstack((arr[:1], arr, arr[-1:], arr[-1:]))
    s = np.cumsum(arr2.astype(float), axis=0)
    s[n:] = s[n:] - s[:-n]
    return s[n - 1:] / n > threshold


def predict_movie_thres(nparr, verbose=False, config=None):
    """{'otsu': bool [N,H,W,2]} -- NB the reference calls moving_avg_mask WITHOUT config (Appendix C.5)."""
    from scipy.ndimage import binary_fill_holes
    if config is None:
        config = default_optical_flow_config()
    masks = []
    for i in range(nparr.shape[0]):
        g = rgb2gray(np.squeeze(nparr[i]))
        m = g > threshold_otsu(g)
        masks.append(remove_small_objects(binary_fill_holes(m), config.min_mask_size))
    arr = moving_avg_mask(np.squeeze(np.stack(masks)))
    return {"otsu": np.repeat(arr[:, :, :, None], 2, axis=3)}
