"""Oracle (CPU, numpy): Gaussian target render, heatmap MSE loss, arg-max decode.

TEST INFRASTRUCTURE - see oracle/__init__.py.  Each function cites the reference
lines it restates (paths relative to the reference root).
"""
import math

import numpy as np

HEATMAP = 64          # src/modeling/simplebaseline/config.py:42 (HEATMAP_SIZE)
FEAT_STRIDE = 4       # src/tools/dataset.py:178
SIGMA = 2             # src/tools/dataset.py:192  (2 * 2**2 in the exponent)
RADIUS = 3 * SIGMA    # src/tools/dataset.py:174  (tmp_size)


def gaussian_patch(radius=RADIUS, sigma=SIGMA):
    """The (2r+1)^2 un-normalised Gaussian, fp32 like the reference (dataset.py:188-193).

    The reference builds x as float32 arange, so the exponent is evaluated in fp32.
    """
    size = 2 * radius + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, None]
    c = size // 2
    return np.exp(-((x - c) ** 2 + (y - c) ** 2) / (2 * sigma ** 2)).astype(np.float32)


def _trunc_center(v):
    # dataset.py:179-180: int(v / 4 + 0.5) -- Python int() truncates toward zero,
    # so negative coordinates round toward zero (not floor).
    return int(math.trunc(float(v) / FEAT_STRIDE + 0.5))


def generate_target(joints, num_joints=21, size=HEATMAP):
    """Restates CustomDataset.generate_target (src/tools/dataset.py:165-212).

    joints: [J, >=2] pixel coordinates in the 256x256 input frame.
    Returns float32 [J, size, size]; each joint's clipped 13x13 patch is ASSIGNED
    (not blended) into a zero map; joints whose patch lies wholly outside are skipped.
    """
    joints = np.asarray(joints)
    out = np.zeros((num_joints, size, size), np.float32)
    g = gaussian_patch()
    for j in range(num_joints):
        mx, my = _trunc_center(joints[j][0]), _trunc_center(joints[j][1])
        x0, y0 = mx - RADIUS, my - RADIUS            # ul
        x1, y1 = mx + RADIUS + 1, my + RADIUS + 1    # br (exclusive)
        if x0 >= size or y0 >= size or x1 < 0 or y1 < 0:
            continue
        gx0, gx1 = max(0, -x0), min(x1, size) - x0
        gy0, gy1 = max(0, -y0), min(y1, size) - y0
        ix0, ix1 = max(0, x0), min(x1, size)
        iy0, iy1 = max(0, y0), min(y1, size)
        if gx1 > gx0 and gy1 > gy0:
            out[j, iy0:iy1, ix0:ix1] = g[gy0:gy1, gx0:gx1]
    return out


def generate_heatmap_alt(points, res=HEATMAP, num_parts=21):
    """Restates GenerateHeatmap (src/utils/dataset_loader.py:22-53): sigma = res/64,
    9x9 patch, np.maximum blend, points already in heatmap coordinates; joints with
    x <= 0 or outside the map are skipped."""
    sigma = res / 64
    size = 6 * sigma + 3
    x = np.arange(0, size, 1, float)
    y = x[:, None]
    c = 3 * sigma + 1
    g = np.exp(-((x - c) ** 2 + (y - c) ** 2) / (2 * sigma ** 2))
    out = np.zeros((num_parts, res, res), np.float32)
    for idx, pt in enumerate(points):
        if not pt[0] > 0:
            continue
        px, py = int(pt[0]), int(pt[1])
        if px < 0 or py < 0 or px >= res or py >= res:
            continue
        ul = int(px - 3 * sigma - 1), int(py - 3 * sigma - 1)
        br = int(px + 3 * sigma + 2), int(py + 3 * sigma + 2)
        c0, d0 = max(0, -ul[0]), min(br[0], res) - ul[0]
        a0, b0 = max(0, -ul[1]), min(br[1], res) - ul[1]
        cc, dd = max(0, ul[0]), min(br[0], res)
        aa, bb = max(0, ul[1]), min(br[1], res)
        out[idx, aa:bb, cc:dd] = np.maximum(out[idx, aa:bb, cc:dd], g[a0:b0, c0:d0])
    return out


def joints_mse_loss(pred, target):
    """Restates JointsMSELoss(use_target_weight=False) (src/utils/loss.py:306-325).

    loss = (1/J) * sum_j 0.5 * mean_{b,hw} (p_j - g_j)^2; every joint has the same
    element count, so this equals 0.5 * mean over all elements.  Returns
    (loss float32, dloss/dpred float32) -- grad = (p - g) / (B*J*H*W).
    """
    p = np.asarray(pred, np.float32)
    g = np.asarray(target, np.float32)
    b, j = p.shape[:2]
    diff = (p - g).astype(np.float64)
    per_joint = 0.5 * (diff.reshape(b, j, -1) ** 2).mean(axis=(0, 2))
    loss = per_joint.sum() / j
    grad = (diff / diff.size).astype(np.float32)
    return np.float32(loss), grad


def get_max_preds(heatmaps):
    """Restates get_max_preds (src/utils/loss.py:327-355).

    heatmaps: ndarray [B, J, H, W].  First-occurrence arg-max over H*W (NaN counts
    as the maximum, numpy rule); x = idx % W, y = floor(idx / W) as float32, both
    zeroed when max <= 0.  Returns (preds [B,J,2] f32, maxvals [B,J,1]).
    """
    assert isinstance(heatmaps, np.ndarray) and heatmaps.ndim == 4
    b, j, h, w = heatmaps.shape
    flat = heatmaps.reshape(b, j, -1)
    preds = np.zeros((b, j, 2), np.float32)
    maxvals = np.zeros((b, j, 1), heatmaps.dtype)
    for bi in range(b):
        for ji in range(j):
            row = flat[bi, ji]
            nan = np.isnan(row)
            if nan.any():
                k = int(np.flatnonzero(nan)[0])
            else:
                k = int(np.flatnonzero(row == row.max())[0])
            v = row[k]
            maxvals[bi, ji, 0] = v
            keep = 1.0 if v > 0.0 else 0.0
            preds[bi, ji, 0] = np.float32(k % w) * keep
            preds[bi, ji, 1] = np.float32(math.floor(k / w)) * keep
    return preds, maxvals


def refine_quarter_pixel(heatmaps, preds, maxvals):
    """Opt-in extension, NOT in the reference (its TEST.POST_PROCESS switch, src/modeling/simplebaseline/config.py:109,
    is never read): restates the published SimpleBaseline `get_final_preds` post-processing on top of
    get_max_preds' UNSCALED output.  px = int(floor(x + 0.5)); if 1 < px < W-1 and 1 < py < H-1:
    coords += 0.25 * sign([hm[py][px+1] - hm[py][px-1], hm[py+1][px] - hm[py-1][px]])."""
    b, j, h, w = heatmaps.shape
    out = preds.astype(np.float32).copy()
    for bi in range(b):
        for ji in range(j):
            hm = heatmaps[bi, ji]
            px = int(math.floor(out[bi, ji, 0] + 0.5))
            py = int(math.floor(out[bi, ji, 1] + 0.5))
            if 1 < px < w - 1 and 1 < py < h - 1:
                diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]], np.float32)
                out[bi, ji] += np.sign(diff).astype(np.float32) * np.float32(0.25)
    return out


def soft_argmax(heatmaps, beta=100.0):
    """Opt-in extension named in the project brief, NOT in the reference: expected pixel coordinate under
    softmax(beta * heatmap) per (sample, joint), float64."""
    b, j, h, w = heatmaps.shape
    z = beta * heatmaps.reshape(b, j, -1).astype(np.float64)
    z -= z.max(-1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(-1, keepdims=True)
    idx = np.arange(h * w)
    return np.stack([(p * (idx % w)).sum(-1), (p * (idx // w)).sum(-1)], -1)

