"""Mask glue of the reference, restated with numpy/scipy (+ torch/PIL for the segmentor call) so the study driver runs
where skimage / torchvision are absent.  Mirrors /root/reference/optical_flow/calculate_optical_flow.py:
  :47-88    evaluate_1_slice   (one frame through a SAM-style module: image_encoder / prompt_encoder / mask_decoder)
  :90-111   moving_avg_mask
  :113-182  clean_mask
  :184-213  predict_movie_thres (mode='otsu')
  :215-241  predict_movie      (modes 'A4C', 'RVIO_2class')
moving_avg_mask / predict_movie_thres are pinned by tests/golden/reference_host_side.npz.  Host-side glue, not a kernel:
the segmentor stays stock PyTorch(-ROCm), as north_star says."""
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


_MODE_LABELS = {
    "A4C": {"lv_inner": 1, "lv": 2, "la_inner": 3, "la": 4, "rv_inner": 5, "ra_inner": 6, "rv": 7, "ra": 8},
    "RVIO_2class": {"rv": 1, "av": 2},
    "MouseRV_A4C": {"rv": 1, "rv_inner": 2},
}


def clean_mask(arr, mode="A4C", verbose=False, config=None):
    """Reference :113-182: class map [N,H,W] -> {label: bool [N,H,W,2]} + 'bkgd'; None for an unknown mode."""
    from scipy.ndimage import binary_fill_holes
    if config is None:
        config = default_optical_flow_config()
    if mode not in _MODE_LABELS:
        return None
    arr = np.asarray(arr)
    out = {}
    aggregate = np.zeros(arr.shape, dtype=bool)
    for k, cls in _MODE_LABELS[mode].items():
        m = moving_avg_mask(np.squeeze(arr == cls))                 # NB: called without config, as the reference does
        clean = np.stack([remove_small_objects(binary_fill_holes(m[i]), config.min_mask_size) for i in range(m.shape[0])])
        aggregate = np.logical_or(clean, aggregate)
        out[k] = np.repeat(clean[:, :, :, None], 2, axis=3)
    out["bkgd"] = np.repeat(np.logical_not(aggregate)[:, :, :, None], 2, axis=3)
    return out


def evaluate_1_slice(frame, model):
    """Reference :47-88 without torchvision: RGB frame uint8 [H,W,3] -> class map uint8 [H,W].  PIL bilinear resize to
    1024 x 1024 (what transforms.Resize does on a PIL image), ToTensor, ImageNet normalisation, the three SAM sub-modules,
    argmax over classes, NEAREST resize back.  The tensor goes to the model's own device (the reference hard-codes .cuda())."""
    import torch
    from PIL import Image
    img = Image.fromarray(np.asarray(frame)).convert("RGB")
    orig_size = img.size
    img = img.resize((1024, 1024), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    x = ((x - mean) / std).unsqueeze(0)
    try:
        dev = next(model.parameters()).device
    except (StopIteration, AttributeError):
        dev = torch.device("cpu")
    x = x.to(dev)
    with torch.no_grad():
        emb = model.image_encoder(x)
        sparse, dense = model.prompt_encoder(points=None, boxes=None, masks=None)
        pred, _ = model.mask_decoder(image_embeddings=emb, image_pe=model.prompt_encoder.get_dense_pe(),
                                     sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, multimask_output=True)
        pred = pred.argmax(dim=1).cpu().float()
    pil_mask = Image.fromarray(pred[0].numpy().astype(np.uint8), "L").resize(orig_size, resample=Image.NEAREST)
    return np.asarray(pil_mask, dtype=np.uint8)


def predict_movie(nparr, model, mode="A4C", verbose=False, config=None):
    """Reference :215-241: every frame through the segmentor, then clean_mask."""
    if config is None:
        config = default_optical_flow_config()
    preds = [evaluate_1_slice(nparr[i], model) for i in range(nparr.shape[0])]
    return clean_mask(np.stack(preds), mode, verbose, config=config)
