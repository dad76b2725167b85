"""Heatmap target render, MSE heatmap loss and arg-max keypoint decode on the HIP device.

Mirrors the three free functions the reference's training loop uses
(``CustomDataset.generate_target`` src/tools/dataset.py:165-212, ``JointsMSELoss``
src/utils/loss.py:306-325, ``get_max_preds`` src/utils/loss.py:327-355) with the same
names, arguments and error behaviour; the arithmetic runs in liblighthand_hip.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check

HEATMAP_SIZE = 64
SIGMA = 2
RADIUS = 3 * SIGMA


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gaussian_patch_host(radius=RADIUS, sigma=SIGMA):
    """The (2r+1)^2 patch, evaluated on the host with numpy float32 exactly like the reference
    (dataset.py:188-193) so the rendered values are bit-identical to the reference's."""
    size = 2 * radius + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, np.newaxis]
    c = size // 2
    return np.exp(-((x - c) ** 2 + (y - c) ** 2) / (2 * sigma ** 2)).astype(np.float32)


_patch_cache = {}


def _patch_on(device):
    key = str(device)
    if key not in _patch_cache:
        _patch_cache[key] = torch.from_numpy(gaussian_patch_host()).to(device)
    return _patch_cache[key]


def render_targets(joints, size=HEATMAP_SIZE, out=None):
    """joints: device tensor [B, J, >=2] (pixel coordinates in the 256x256 frame) ->
    float32 [B, J, size, size] Gaussian targets (sigma 2, 13x13 patch, clipped assignment)."""
    if not joints.is_cuda:
        raise _lib.LightHandError("render_targets needs a HIP device tensor")
    j = joints.to(torch.float32).contiguous()
    b, nj, stride = j.shape
    if out is None:
        out = torch.empty(b, nj, size, size, dtype=torch.float32, device=j.device)
    patch = _patch_on(j.device)
    check(_lib.load().lh_gaussian_target(j.data_ptr(), stride, patch.data_ptr(), RADIUS, out.data_ptr(), b, nj, size, _stream()),
          "lh_gaussian_target")
    return out


def generate_target(joints, device="cuda"):
    """Per-sample form with the reference's signature: joints [21, >=2] (array-like) ->
    torch.float32 [21, 64, 64] (returned on the CPU like the reference's dataset method)."""
    j = torch.as_tensor(np.asarray(joints, dtype=np.float32)[:, :2].copy()).to(device)
    return render_targets(j[None])[0].cpu()


class GenerateHeatmap:
    """The reference's alternate renderer (src/utils/dataset_loader.py:22-53), same constructor and call: points
    ``[num_parts, >=2]`` already in heat-map coordinates -> float32 ``[num_parts, res, res]`` (CPU, like the reference);
    ``render(points)`` is the batched device form ``[B, J, >=2] -> [B, J, res, res]``.  sigma = output_res / 64 must be an
    integer (the reference instantiates it with output_res = 64)."""

    def __init__(self, output_res, num_parts, device="cuda"):
        if output_res % 64:
            raise ValueError("GenerateHeatmap on the device needs output_res = 64 * k (sigma = output_res / 64 integral)")
        self.output_res, self.num_parts, self.device = output_res, num_parts, device
        sigma = self.output_res / 64
        self.sigma = sigma
        size = 6 * sigma + 3
        x = np.arange(0, size, 1, float)
        y = x[:, np.newaxis]
        x0, y0 = 3 * sigma + 1, 3 * sigma + 1
        self.g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))       # float64, as in the reference
        self._patch = {}

    def render(self, points, out=None):
        if not points.is_cuda:
            raise _lib.LightHandError("GenerateHeatmap.render needs a HIP device tensor")
        p = points.to(torch.float32).contiguous()
        b, nj, stride = p.shape
        if out is None:
            out = torch.empty(b, nj, self.output_res, self.output_res, dtype=torch.float32, device=p.device)
        key = str(p.device)
        if key not in self._patch:
            self._patch[key] = torch.from_numpy(self.g.astype(np.float32)).to(p.device)
        check(_lib.load().lh_gaussian_target_alt(p.data_ptr(), stride, self._patch[key].data_ptr(), int(self.sigma), out.data_ptr(),
                                                 b, nj, self.output_res, _stream()), "lh_gaussian_target_alt")
        return out

    def __call__(self, p):
        pts = torch.as_tensor(np.asarray(p, dtype=np.float32)[:, :2].copy()).to(self.device)
        return self.render(pts[None])[0].cpu().numpy()


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target):
        lib = _lib.load()
        p = output.detach().to(torch.float32).contiguous()
        g = target.detach().to(torch.float32).contiguous()
        if p.shape != g.shape:
            raise ValueError(f"prediction {tuple(p.shape)} and target {tuple(g.shape)} differ")
        n = p.numel()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        ws = torch.empty(lib.lh_mse_workspace_bytes(n), dtype=torch.uint8, device=p.device)
        check(lib.lh_mse_heatmap(p.data_ptr(), g.data_ptr(), n, loss.data_ptr(), grad.data_ptr(), None, ws.data_ptr(), _stream()),
              "lh_mse_heatmap")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return grad * gout, None


class JointsMSELoss(nn.Module):
    """0.5 * MSE per joint averaged over joints (== 0.5 * mean over all elements).  The
    ``target_weight`` argument is accepted and ignored exactly like the reference does with
    ``use_target_weight=False`` (src/utils/method.py:49)."""

    def __init__(self, use_target_weight=False):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight=None):
        if not output.is_cuda:
            raise _lib.LightHandError("JointsMSELoss runs on the HIP device only")
        return _MseFn.apply(output, target)


def max_preds_device(heatmaps, scale=1.0, post_process=False):
    """Device overload: heatmaps float32 [B, J, H, W] on the device ->
    (preds [B, J, 2], maxvals [B, J, 1], flat indices [B, J]) device tensors.  ``post_process=True`` adds the opt-in
    quarter-pixel refinement (an extension: the reference's TEST.POST_PROCESS flag exists but is unused)."""
    if heatmaps.dim() != 4:
        raise AssertionError("batch_images should be 4-ndim")
    hm = heatmaps.to(torch.float32).contiguous()
    b, j, h, w = hm.shape
    preds = torch.empty(b, j, 2, dtype=torch.float32, device=hm.device)
    maxvals = torch.empty(b, j, 1, dtype=torch.float32, device=hm.device)
    idx = torch.empty(b, j, dtype=torch.int32, device=hm.device)
    check(_lib.load().lh_heatmap_argmax(hm.data_ptr(), b * j, h, w, float(scale), preds.data_ptr(), maxvals.data_ptr(),
                                         idx.data_ptr(), _stream()), "lh_heatmap_argmax")
    if post_process:
        check(_lib.load().lh_heatmap_refine(hm.data_ptr(), idx.data_ptr(), maxvals.data_ptr(), b * j, h, w, float(scale),
                                            preds.data_ptr(), _stream()), "lh_heatmap_refine")
    return preds, maxvals, idx


def soft_argmax_device(heatmaps, beta=100.0, scale=1.0):
    """Opt-in differentiable-style decode (an extension: the reference only has the hard arg-max): heatmaps float32
    [B, J, H, W] on the device -> expected (x, y) under softmax(beta * heatmap), [B, J, 2] device tensor."""
    if heatmaps.dim() != 4:
        raise AssertionError("batch_images should be 4-ndim")
    hm = heatmaps.to(torch.float32).contiguous()
    b, j, h, w = hm.shape
    preds = torch.empty(b, j, 2, dtype=torch.float32, device=hm.device)
    check(_lib.load().lh_heatmap_soft_argmax(hm.data_ptr(), b * j, h, w, float(beta), float(scale), preds.data_ptr(), _stream()),
          "lh_heatmap_soft_argmax")
    return preds


def get_max_preds(batch_heatmaps, post_process=False):
    """Reference signature (src/utils/loss.py:327-355): numpy [B, J, H, W] -> (preds float32
    [B, J, 2], maxvals [B, J, 1]) numpy arrays; device tensors are accepted too and then
    device tensors are returned (no host round trip).  ``post_process`` (default off = reference behaviour) enables
    the quarter-pixel refinement."""
    if isinstance(batch_heatmaps, torch.Tensor):
        p, m, _ = max_preds_device(batch_heatmaps, post_process=post_process)
        return p, m
    assert isinstance(batch_heatmaps, np.ndarray), "batch_heatmaps should be numpy.ndarray"
    assert batch_heatmaps.ndim == 4, "batch_images should be 4-ndim"
    p, m, _ = max_preds_device(torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).cuda(),
                               post_process=post_process)
    return p.cpu().numpy(), m.cpu().numpy().astype(batch_heatmaps.dtype)
