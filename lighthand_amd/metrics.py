"""Validation / evaluation metrics of the reference loop, host side (NumPy, like the reference):
``PCK_2d_loss`` and ``EPE_train`` (src/utils/loss.py:116-148, 50-67, used by Runner.run in validation,
src/utils/method.py:243-250) and ``pred_eval`` (src/utils/argparser.py:326-388).  The reference's quirks are
kept (they change the reported numbers): EPE_train sums joints 1..J-2 only; pred_eval's 'mean_auc' EPE is
diluted by 971 zero rows; visible joints only for PCK; px->mm constants 3.7795275591 / 2.83464567.
"""
import sys

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz      # np.trapz is deprecated (removed in newer NumPy)

PX_PER_MM_EVAL = 3.7795275591
PX_PER_MM_THRESH = 2.83464567


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def PCK_2d_loss(pred_2d, gt_2d, T=0.1, threshold="proportion"):
    pred, gt = _np(pred_2d).astype(np.float32), _np(gt_2d).astype(np.float32)[..., :2]
    diag = np.sqrt((gt[..., 0].max(1) - gt[..., 0].min(1)) ** 2 + (gt[..., 1].max(1) - gt[..., 1].min(1)) ** 2)
    dist = np.sqrt(((gt - pred) ** 2).sum(2))
    if threshold == "proportion":
        wrong = int((dist / diag[:, None] > T).sum())
    elif threshold == "mm":
        wrong = int((dist > T * 3.78).sum())
    else:
        assert False, "Please check variable threshold is right"
    return float((dist.size - wrong) / dist.size)


def EPE_train(pred_2d_joints, gt_2d_joint):
    """Returns ((sum, count), per_joint) like the reference; joints 0 and J-1 do not enter the sum."""
    pred, gt = _np(pred_2d_joints).astype(np.float32), _np(gt_2d_joint).astype(np.float32)[..., :2]
    b, j = pred.shape[:2]
    err = np.sqrt(((pred - gt) ** 2).sum(2)).astype(np.float32)
    distance = {f"{i}": [float(err[:, i].mean()), b] for i in range(1, j)}
    s = sum(distance[f"{i}"][0] * b for i in range(1, j - 1))
    return (s, float(b * (j - 2))), distance


def pred_eval(meta, T_list, method):
    """meta: the category dict of evaluation.json ({cat: {bb, pred, gt}}) -> {cat: [auc, epe_mm, pck_curve]}."""
    if method == "mm":
        thr = np.linspace(T_list[0], T_list[-1], 101)[1:] * PX_PER_MM_THRESH
    elif method == "pckb":
        thr = np.linspace(T_list[0], T_list[-1], 100)
    else:
        assert 0, "this method is the wrong"
    norm = _trapz(np.ones_like(thr), thr)
    out, vis_all, diff_all = {}, [], [np.zeros([971, 21])]
    for cat, d in meta.items():
        bb, pred, gt = np.array(d["bb"]), np.array(d["pred"]), np.array(d["gt"])
        diff = np.sqrt(np.sum(np.square(gt[:, :, :2] - pred[:, :, :2]), axis=-1))
        nd = diff / bb[:, None] if method == "pckb" else diff
        vis = nd[gt[:, :, -1] == 1]
        diff_all.append(diff)
        vis_all.insert(0, vis)
        curve = np.array([(vis < t).sum() / len(vis) * 100 for t in thr])
        out[cat] = [float(_trapz(curve, thr) / (norm + sys.float_info.epsilon)), float(diff.mean() / PX_PER_MM_EVAL), curve]
    vis = np.concatenate(vis_all)
    curve = np.array([(vis < t).sum() / len(vis) * 100 for t in thr])
    out["mean_auc"] = [float(_trapz(curve, thr) / (norm + sys.float_info.epsilon)),
                       float(np.concatenate(diff_all, 0).mean() / PX_PER_MM_EVAL), curve]
    return out


def device_pck_epe(pred_2d, gt_2d, T=0.2):
    """PCK@T ('proportion') and the EPE_train sum/count on the DEVICE (no host round trip): returns 0-d device
    tensors (pck, epe_sum, epe_count).  Same arithmetic as PCK_2d_loss / EPE_train above, quirks included."""
    import torch
    from . import _lib
    pred = pred_2d.to(torch.float32).contiguous()
    gt = gt_2d.to(torch.float32).contiguous()
    b, j = pred.shape[:2]
    wrong = torch.empty(b, dtype=torch.int32, device=pred.device)
    epe = torch.empty(b, dtype=torch.float32, device=pred.device)
    _lib.check(_lib.load().lh_keypoint_metrics(pred.data_ptr(), gt.data_ptr(), gt.shape[2], b, j, float(T), wrong.data_ptr(),
                                               epe.data_ptr(), torch.cuda.current_stream().cuda_stream), "lh_keypoint_metrics")
    pck = 1.0 - wrong.sum().to(torch.float32) / float(b * j)
    return pck, epe.sum(), torch.tensor(float(b * (j - 2)), device=pred.device)


def eval_thresholds(T_list, method):
    """The threshold grid of pred_eval (src/utils/argparser.py:334-341)."""
    if method == "mm":
        return np.linspace(T_list[0], T_list[-1], 101)[1:] * PX_PER_MM_THRESH
    if method == "pckb":
        return np.linspace(T_list[0], T_list[-1], 100)
    assert 0, "this method is the wrong"


def device_pck_curve(pred_2d, gt_3, bb, T_list, method, out=None):
    """pred_eval's counting on the DEVICE: pred [N, J, 2], gt [N, J, >=3] (x, y, visibility), bb [N] device tensors ->
    (counts int64 [T], nvis int64 [1], diff_sum float64 [1], n_joints int64 [1]) device tensors; pass ``out`` (a previous
    result) to accumulate over batches.  Integer counts: exact, so data-parallel ranks add them with one all-reduce and
    every rank then gets the same AUC (``auc_from_counts``)."""
    import torch
    from . import _lib
    thr = eval_thresholds(T_list, method)
    dev = pred_2d.device
    pred = pred_2d.to(torch.float32).contiguous()
    gt = gt_3.to(torch.float32).contiguous()
    n, j = pred.shape[:2]
    if out is None:
        out = (torch.zeros(len(thr), dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
               torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
    counts, nvis, diff_sum, n_all = out
    thr_dev = torch.from_numpy(np.ascontiguousarray(thr, dtype=np.float64)).to(dev)
    rows = torch.empty(n, dtype=torch.float64, device=dev)
    bbp = None
    if method == "pckb":
        bbt = bb.to(torch.float32).contiguous()
        bbp = bbt.data_ptr()
    _lib.check(_lib.load().lh_pck_curve(pred.data_ptr(), gt.data_ptr(), gt.shape[2], bbp, n, j, thr_dev.data_ptr(), len(thr),
                                        counts.data_ptr(), nvis.data_ptr(), rows.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "lh_pck_curve")
    diff_sum += rows.sum()
    n_all += n * j
    return counts, nvis, diff_sum, n_all


def auc_from_counts(counts, nvis, diff_sum, n_all, T_list, method):
    """[auc, epe_mm, pck_curve] exactly as pred_eval reports one category (src/utils/argparser.py:362-375)."""
    thr = eval_thresholds(T_list, method)
    norm = _trapz(np.ones_like(thr), thr)
    curve = np.asarray(counts, dtype=np.float64) / float(nvis) * 100
    return [float(_trapz(curve, thr) / (norm + sys.float_info.epsilon)), float(diff_sum) / float(n_all) / PX_PER_MM_EVAL, curve]


def pred_test(meta, T_list, method):
    """src/utils/argparser.py:391-438 -- the category-less evaluation (``pred_store_test`` output: lists of per-batch
    predictions, ground truth and bounding-box diagonals): AUC of the PCK curve over ALL joints (no visibility flag) and
    the mean error in PIXELS (the reference does not convert this one to mm).  Unlike the reference under NumPy >= 1.24
    a short last batch is accepted.  Returns (auc, mean_error_px)."""
    if method == "mm":          # this variant converts its mm thresholds with 3.7795 px/mm (argparser.py:400), pred_eval with 2.8346
        thr = np.linspace(T_list[0], T_list[-1], 101)[1:] * PX_PER_MM_EVAL
    else:
        thr = eval_thresholds(T_list, method)
    norm = _trapz(np.ones_like(thr), thr)
    bb = np.concatenate([np.asarray(b, dtype=np.float64).reshape(-1) for b in meta["bb"]])
    gt = np.concatenate([np.asarray(g, dtype=np.float64) for g in meta["gt"]])
    pred = np.concatenate([np.asarray(q, dtype=np.float64) for q in meta["pred"]])
    diff = np.sqrt(np.sum(np.square(gt[..., :2] - pred[..., :2]), axis=-1))
    nd = (diff / bb[:, None] if method == "pckb" else diff).flatten()
    curve = np.array([(nd < t).sum() / len(nd) * 100 for t in thr])
    return float(_trapz(curve, thr) / (norm + sys.float_info.epsilon)), float(diff.mean())
