#!/usr/bin/env python3
"""Evaluation entry point mirroring the reference's ``src/tools/wearable_eval_2d.py`` + ``pred_store`` /
``pred_eval`` (src/utils/argparser.py:246-388) on the HIP engine.

For every ``state_dict.bin`` under ``<root_path>/<model_path>``: load it (strict=False), run the model over the
evaluation set (keypoints decoded on the device, x4), write ``<root_path>/<name>/evaluation.json`` in the
reference's layout ``[{category: {bb, pred, gt}}]`` and evaluate AUC / EPE for the three threshold sets
(pckb [0.1,0.3], mm [0,30], mm [0,50]) into ``pck_eval_<...>_<type>_<T1>.txt`` with the reference's line format.

Reference quirk kept behind a flag: ``pred_store`` never calls ``model.eval()``, so BatchNorm normalises with
BATCH statistics at evaluation time.  ``--bn_train`` (default, = reference behaviour) reproduces that;
``--bn_eval`` uses the running statistics (BN folded into the conv epilogues).

The reference's evaluation set (Armo_hand_dataset) is not redistributable: ``--synthetic N`` builds a seeded
stand-in with the four occlusion categories and visibility flags.
"""
import argparse
import json
import os

import numpy as np
import torch

CATEGORIES = ["Standard", "Occlusion_by_Pinky", "Occlusion_by_Thumb", "Occlusion_by_Both"]    # argparser.py:247-252
THRESHOLDS = [["pckb", [0.1, 0.3]], ["mm", [0, 30]], ["mm", [0, 50]]]                         # wearable_eval_2d.py:40-44


class SyntheticEvalSet(torch.utils.data.Dataset):
    """(image[3,S,S], joint_2d_v[21,3] = x, y, visible, category) like eval_set.__getitem__ (dataset.py:255-300)."""

    def __init__(self, n, size, seed=9001):
        rng = np.random.RandomState(seed)
        self.images = torch.from_numpy(rng.randn(n, 3, size, size).astype(np.float32))
        xy = rng.uniform(20, size - 20, size=(n, 21, 2)).astype(np.float32)
        vis = (rng.rand(n, 21, 1) > 0.25).astype(np.float32)
        self.joints = torch.from_numpy(np.concatenate([xy, vis], -1))
        self.cats = [CATEGORIES[i] for i in rng.randint(0, 4, size=n)]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return self.images[i], self.joints[i], self.cats[i]


class _Steps:
    """One captured InferStep per batch size met: the last, short batch runs AS IS like in the reference loop
    (argparser.py:258-262) -- padding it would put foreign samples into the BatchNorm batch statistics that
    ``pred_store``'s train-mode forward uses, and into the running-statistics update."""

    def __init__(self, model, size, bn_train):
        self.model, self.size, self.bn_train, self.steps = model, size, bn_train, {}

    def __call__(self, images):
        from lighthand_amd.runtime import InferStep
        n = images.shape[0]
        step = self.steps.get(n)
        if step is None:
            step = self.steps[n] = InferStep(self.model, n, self.size, self.size, bn_train=self.bn_train)
        return step(images.cuda(non_blocking=True))


def pred_store(model, loader, out_json, batch, size, bn_train=True):
    """src/utils/argparser.py:246-281 with the forward + arg-max decode on the device."""
    meta = {c: {"bb": [], "pred": [], "gt": []} for c in CATEGORIES}
    step = _Steps(model, size, bn_train)
    for images, joints_v, cats in loader:
        preds = step(images).cpu()                       # already x4 (method.py:157)
        gt = joints_v[:, :, :2]
        w = gt[..., 0].max(1).values - gt[..., 0].min(1).values
        h = gt[..., 1].max(1).values - gt[..., 1].min(1).values
        bb = torch.sqrt(w ** 2 + h ** 2)
        for i, name in enumerate(cats):
            meta[name]["bb"].append(bb[i].item())
            meta[name]["pred"].append(preds[i].tolist())
            meta[name]["gt"].append(joints_v[i].tolist())
    os.makedirs(os.path.dirname(out_json), exist_ok=True)
    with open(out_json, "w") as f:
        json.dump([meta], f)
    return meta


def pred_store_test(model, loader, out_json, batch, size, bn_train=True):
    """src/utils/argparser.py:284-323 -- the category-less variant: one entry per BATCH of predictions (x4, 256-px frame),
    ground truth and bounding-box diagonals; read back by ``lighthand_amd.metrics.pred_test``.  The loader yields
    (images, gt_2d_joints[, ...])."""
    meta = {"pred": [], "gt": [], "bb": []}
    step = _Steps(model, size, bn_train)
    for item in loader:
        images, gt = item[0], item[1][:, :, :2]
        preds = step(images).cpu()
        w = gt[..., 0].max(1).values - gt[..., 0].min(1).values
        h = gt[..., 1].max(1).values - gt[..., 1].min(1).values
        meta["pred"].append(preds.tolist())
        meta["gt"].append(gt.tolist())
        meta["bb"].append(torch.sqrt(w ** 2 + h ** 2).tolist())
    os.makedirs(os.path.dirname(out_json), exist_ok=True)
    with open(out_json, "w") as f:
        json.dump([meta], f)
    return meta


def device_eval(model, loader, batch, size, bn_train=True):
    """Same evaluation reduced ON THE DEVICE: per threshold set the PCK-curve counts of all visible
    joints are accumulated by lh_pck_curve, summed across data-parallel ranks with one small all-reduce, and read by
    the host once.  Returns {(type, T1): [auc, epe_mm, curve]}; the AUC equals pred_eval's 'mean_auc' AUC (whose EPE is
    diluted by the reference's zeros quirk; the EPE here is the plain mean)."""
    from lighthand_amd.metrics import auc_from_counts, device_pck_curve
    step = _Steps(model, size, bn_train)
    acc = {(t, tuple(T)): None for t, T in THRESHOLDS}
    for images, joints_v, _ in loader:
        preds = step(images)
        gt = joints_v.cuda(non_blocking=True)
        w = gt[..., 0].max(1).values - gt[..., 0].min(1).values
        h = gt[..., 1].max(1).values - gt[..., 1].min(1).values
        bb = torch.sqrt(w ** 2 + h ** 2)
        for key in acc:
            acc[key] = device_pck_curve(preds, gt, bb, list(key[1]), key[0], out=acc[key])
    out = {}
    for key, tensors in acc.items():
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            for t in tensors:
                torch.distributed.all_reduce(t)
        counts, nvis, diff_sum, n_all = (t.cpu().numpy() for t in tensors)
        out[(key[0], key[1][1])] = auc_from_counts(counts, nvis[0], diff_sum[0], n_all[0], list(key[1]), key[0])
    return out


def write_report(path, results):
    """wearable_eval_2d.py:64-79: `category;model;auc;epe;pck...;` one line per category."""
    with open(path, "w") as f:
        for total_pck, name in results:
            for p_type in total_pck:
                f.write("{};{};{:.2f};{:.2f};".format(p_type, name, total_pck[p_type][0], total_pck[p_type][1]))
                curve = total_pck[p_type][2]
                for idx, pck in enumerate(curve):
                    f.write("{:.2f};".format(pck))
                    if idx == len(curve) - 1:
                        f.write("\n")


def main(argv=None):
    from lighthand_amd.metrics import pred_eval
    from lighthand_amd.tools.train import build_model
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_path", default="output")
    ap.add_argument("--model_path", default="simplebaseline/frei", help="sub-tree of root_path searched for *.bin (reference: output/simplebaseline/frei)")
    ap.add_argument("--batch_size", default=32, type=int)
    ap.add_argument("--depth", default=50, type=int)
    ap.add_argument("--hrnet_width", default=48, type=int)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--size", default=256, type=int)
    ap.add_argument("--synthetic", default=0, type=int)
    ap.add_argument("--bn_eval", action="store_true")
    ap.add_argument("--device_metrics", action="store_true", help="also reduce the PCK curves / AUC on the device and print them")
    args = ap.parse_args(argv)
    if not args.synthetic:
        raise SystemExit("the Armo_hand evaluation set is not shipped: pass --synthetic N or plug your own Dataset")
    data = SyntheticEvalSet(args.synthetic, args.size)
    loader = torch.utils.data.DataLoader(data, batch_size=args.batch_size, shuffle=False)
    root = os.path.join(args.root_path, args.model_path)
    ckpts = sorted(os.path.join(r, f) for r, _, fs in os.walk(root) for f in fs if f.endswith(".bin"))
    if not ckpts:
        raise SystemExit(f"no *.bin checkpoint under {root}")
    written = []
    for t_type, T_list in THRESHOLDS:
        results = []
        for path in ckpts:
            rel = os.path.relpath(path, args.root_path).split(os.sep)
            args.model, name = rel[0], os.sep.join(rel[:-2])
            model = build_model(args).cuda().set_precision(args.precision)
            model.load_state_dict(torch.load(path, map_location="cpu")["model_state_dict"], strict=False)
            model.train(not args.bn_eval)
            out_json = os.path.join(args.root_path, name, "evaluation.json")
            meta = pred_store(model, loader, out_json, args.batch_size, args.size, bn_train=not args.bn_eval)
            results.append([pred_eval({k: v for k, v in meta.items() if v["bb"]}, T_list, t_type), name])
            if args.device_metrics and t_type == THRESHOLDS[0][0] and T_list == THRESHOLDS[0][1]:
                for (ty, t1), (auc, epe, _) in device_eval(model, loader, args.batch_size, args.size, bn_train=not args.bn_eval).items():
                    host = pred_eval({k: v for k, v in meta.items() if v["bb"]}, [t for n_, t in THRESHOLDS if n_ == ty and t[1] == t1][0], ty)
                    print(f"device metrics {name} {ty} {t1}: auc {auc:.4f} (host mean_auc {host['mean_auc'][0]:.4f}) epe {epe:.3f} mm")
        fn = os.path.join(args.root_path, f"pck_eval_{'_'.join(args.model_path.split('/'))}_{t_type}_{T_list[1]}.txt")
        write_report(fn, results)
        written.append(fn)
        print("Writting ===> %s" % fn)
    return written


if __name__ == "__main__":
    main()
