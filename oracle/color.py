"""Oracle (CPU, numpy fp32): the reference's training-time input transform chain
``ToTensor -> Resize((256, 256)) -> ColorJitter(0.5, 0.5, 0.5, 0.5) -> Normalize`` (src/tools/dataset.py:134-146).

TEST INFRASTRUCTURE - see oracle/__init__.py.

The arithmetic of ``ColorJitter`` lives in torchvision (a dependency of the reference, not vendored and ABSENT from
this image: reference pins torchvision 0.12, requirements.yaml), so this file restates torchvision's PUBLISHED tensor
algorithm (torchvision/transforms/functional_tensor.py: ``_blend``, ``rgb_to_grayscale``, ``adjust_brightness /
contrast / saturation / hue``, ``_rgb2hsv``, ``_hsv2rgb``) -- parity UNPINNED by torchvision itself; pinned by the
closed-form known answers of tests/test_oracle_golden.py (identity factors, grey world, pure-colour hue rotations) and, since round 6,
cross-checked against Pillow's ImageEnhance / 8-bit HSV shift -- what torchvision's OTHER (PIL) backend calls for the same four ops --
to 8-bit quantisation (test_color_jitter_ops_agree_with_pillow_image_enhance): an independent implementation, still not torchvision.
The random draw (``ColorJitter.get_params``: factor ~ U[max(0, 1 - v), 1 + v], hue ~ U[-h, h], op order = randperm(4))
stays on the host: the device kernel takes per-image factors and an op order.
"""
import numpy as np

GRAY = np.array([0.2989, 0.587, 0.114], np.float32)      # rgb_to_grayscale weights


def _blend(a, b, ratio):
    return np.clip(np.float32(ratio) * a + np.float32(1.0 - ratio) * b, 0.0, 1.0).astype(np.float32)


def _gray(img):                       # img [3, H, W] float32 in [0, 1]
    return (GRAY[0] * img[0] + GRAY[1] * img[1] + GRAY[2] * img[2]).astype(np.float32)


def adjust_brightness(img, f):
    return _blend(img, np.zeros_like(img), f)


def adjust_contrast(img, f):
    mean = np.float32(_gray(img).astype(np.float64).mean())       # one scalar per image
    return _blend(img, np.full_like(img, mean), f)


def adjust_saturation(img, f):
    return _blend(img, np.broadcast_to(_gray(img)[None], img.shape), f)


def rgb2hsv(img):
    r, g, b = img
    maxc, minc = img.max(0), img.min(0)
    eqc = maxc == minc
    cr = maxc - minc
    ones = np.ones_like(maxc)
    s = cr / np.where(eqc, ones, maxc)
    div = np.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = np.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return np.stack([h, s, maxc]).astype(np.float32)


def hsv2rgb(hsv):
    h, s, v = hsv
    i = np.floor(h * 6.0)
    f = (h * 6.0 - i).astype(np.float32)
    i = i.astype(np.int32) % 6
    p = np.clip(v * (1.0 - s), 0.0, 1.0)
    q = np.clip(v * (1.0 - s * f), 0.0, 1.0)
    t = np.clip(v * (1.0 - s * (1.0 - f)), 0.0, 1.0)
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([r, g, b]).astype(np.float32)


def adjust_hue(img, f):
    hsv = rgb2hsv(img)
    hsv[0] = np.fmod(hsv[0] + np.float32(f), 1.0)
    hsv[0] = np.where(hsv[0] < 0, hsv[0] + 1.0, hsv[0])         # python-style modulo of torch's `%`
    return hsv2rgb(hsv)


OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)     # ColorJitter's fn_id 0..3


def color_jitter(img, factors, order):
    """img [3, H, W] float32 in [0, 1]; factors = (brightness, contrast, saturation, hue); order = sequence of op ids
    (a permutation of 0..3; negative ids are skipped)."""
    for op in order:
        if op >= 0:
            img = OPS[op](img, factors[op])
    return img


def resize_bilinear(img, h, w):
    """Resize of a float tensor image [3, hs, ws] -> [3, h, w]: bilinear, half-pixel centres, no antialias
    (F.interpolate(align_corners=False)), source coordinate clamped at 0 like the framework."""
    _, hs, ws = img.shape
    fy = np.maximum((np.arange(h, dtype=np.float32) + 0.5) * np.float32(hs / h) - 0.5, 0.0).astype(np.float32)
    fx = np.maximum((np.arange(w, dtype=np.float32) + 0.5) * np.float32(ws / w) - 0.5, 0.0).astype(np.float32)
    y0, x0 = fy.astype(np.int32), fx.astype(np.int32)
    y1, x1 = np.minimum(y0 + 1, hs - 1), np.minimum(x0 + 1, ws - 1)
    wy, wx = (fy - y0)[None, :, None], (fx - x0)[None, None, :]
    a00, a01 = img[:, y0][:, :, x0], img[:, y0][:, :, x1]
    a10, a11 = img[:, y1][:, :, x0], img[:, y1][:, :, x1]
    top, bot = a00 + (a01 - a00) * wx, a10 + (a11 - a10) * wx
    return (top + (bot - top) * wy).astype(np.float32)


def input_pipeline(u8_hwc, h, w, factors=None, order=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """uint8 [hs, ws, 3] -> normalised float32 [3, h, w] (dataset.py:134-146; ColorJitter only when `order` is given)."""
    img = resize_bilinear(np.transpose(u8_hwc.astype(np.float32), (2, 0, 1)), h, w) * np.float32(1.0 / 255.0)
    if order is not None:
        img = color_jitter(img, factors, order)
    m, s = np.asarray(mean, np.float32)[:, None, None], np.asarray(std, np.float32)[:, None, None]
    return ((img - m) / s).astype(np.float32)
