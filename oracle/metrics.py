"""Oracle (CPU, numpy): validation / evaluation metrics with the reference's quirks.

TEST INFRASTRUCTURE - see oracle/__init__.py.
"""
import sys

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz      # np.trapz is deprecated (removed in newer NumPy)

PX_PER_MM_EVAL = 3.7795275591     # src/utils/argparser.py:374,385
PX_PER_MM_THRESH = 2.83464567     # src/utils/argparser.py:336
PX_TO_MM_LOG = 0.26               # src/utils/method.py:131


def pck_2d(pred, gt, T=0.1, threshold="proportion"):
    """Restates PCK_2d_loss (src/utils/loss.py:116-148): all joints count, error is
    normalised by the diagonal of the ground-truth joints' bounding box."""
    pred = np.asarray(pred, np.float32)
    gt = np.asarray(gt, np.float32)[..., :2]
    w = gt[..., 0].max(1) - gt[..., 0].min(1)
    h = gt[..., 1].max(1) - gt[..., 1].min(1)
    diag = np.sqrt(w ** 2 + h ** 2)
    dist = np.sqrt(((gt - pred) ** 2).sum(2))
    total = dist.size
    if threshold == "proportion":
        wrong = int((dist / diag[:, None] > T).sum())
    elif threshold == "mm":
        wrong = int((dist > T * 3.78).sum())
    else:
        raise AssertionError("Please check variable threshold is right")
    return float((total - wrong) / total)


def epe_train(pred, gt):
    """Restates EPE_train (src/utils/loss.py:50-67).

    Quirk kept: per-joint means are built for joints 1..J-1, but the final sum runs
    over ``range(1, len(distance))`` = joints 1..J-2 only, so for 21 joints the
    wrist (0) AND joint 20 are skipped; count = (J-2)*B.  Returns (sum, count).
    """
    pred = np.asarray(pred, np.float32)
    gt = np.asarray(gt, np.float32)[..., :2]
    b, j = pred.shape[:2]
    err = np.sqrt(((pred - gt) ** 2).sum(2)).astype(np.float32)       # [B, J]
    s, c = 0.0, 0.0
    for i in range(1, j - 1):
        m = float(np.mean(err[:, i]))
        s += m * b
        c += b
    return s, c


def pred_eval(meta, T_list, method):
    """Restates pred_eval (src/utils/argparser.py:326-388) on an in-memory
    ``evaluation.json`` category dict {cat: {bb, pred, gt}}.

    Quirks kept: 'mm' thresholds are linspace(T0,T1,101)[1:] * 2.83464567 px; PCK
    uses visible joints only (gt[...,2]==1) with a strict '<'; per-category EPE is
    diff.mean()/3.7795275591 over ALL joints; the 'mean_auc' EPE is diluted by the
    971x21 zero rows the reference starts its accumulator with (argparser.py:345).
    Returns {cat: [auc, epe_mm, pck_curve]}.
    """
    if method == "mm":
        thr = np.linspace(T_list[0], T_list[-1], 101)[1:] * PX_PER_MM_THRESH
    elif method == "pckb":
        thr = np.linspace(T_list[0], T_list[-1], 100)
    else:
        raise AssertionError("this method is the wrong")
    norm = _trapz(np.ones_like(thr), thr)
    out = {}
    all_vis = []
    all_diff = [np.zeros((971, 21))]
    for cat, d in meta.items():
        bb = np.array(d["bb"])
        pred = np.array(d["pred"])
        gt = np.array(d["gt"])
        diff = np.sqrt(((gt[:, :, :2] - pred[:, :, :2]) ** 2).sum(-1))
        nd = diff / bb[:, None] if method == "pckb" else diff
        vis = nd[gt[:, :, -1] == 1]
        all_diff.append(diff)
        all_vis.insert(0, vis)
        curve = np.array([(vis < t).sum() / len(vis) * 100 for t in thr])
        auc = _trapz(curve, thr) / (norm + sys.float_info.epsilon)
        out[cat] = [float(auc), float(diff.mean() / PX_PER_MM_EVAL), curve]
    vis = np.concatenate(all_vis)
    curve = np.array([(vis < t).sum() / len(vis) * 100 for t in thr])
    auc = _trapz(curve, thr) / (norm + sys.float_info.epsilon)
    out["mean_auc"] = [float(auc), float(np.concatenate(all_diff, 0).mean() / PX_PER_MM_EVAL), curve]
    return out


def pred_test(meta, T_list, method):
    """argparser.py:391-438 restated: flatten the per-batch lists, PCK curve over every joint (normalised by the
    bounding-box diagonal for 'pckb'), AUC by the trapezoid rule normalised to the threshold span, mean error in pixels."""
    if method == "mm":
        thr = np.linspace(T_list[0], T_list[-1], 101)[1:] * PX_PER_MM_EVAL      # argparser.py:400 (pred_eval uses 2.8346)
    elif method == "pckb":
        thr = np.linspace(T_list[0], T_list[-1], 100)
    else:
        raise AssertionError("this method is the wrong")
    norm = _trapz(np.ones_like(thr), thr)
    bbox = np.array([b for batch in meta["bb"] for b in batch])
    gt = np.array([g for batch in meta["gt"] for g in batch])
    pred = np.array([q for batch in meta["pred"] for q in batch])
    diff = np.sqrt(np.sum(np.square(gt - pred), axis=-1))
    nd = diff / bbox[:, None].repeat(gt.shape[1], axis=1) if method == "pckb" else diff
    nd = nd.flatten()
    pck = np.array([(len(nd[nd < t]) / len(nd)) * 100 for t in thr])
    return float(_trapz(pck, thr) / (norm + sys.float_info.epsilon)), float(diff.mean())
