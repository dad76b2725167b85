#!/usr/bin/env python3
"""Training entry point with the reference's flag surface (src/tools/train.py + src/utils/argparser.py:27-100)
running the hot loop on the HIP engine.

Same flags / defaults as the reference (``--root --name --root_path --model --dataset --view --batch_size
--milestone --count --num_our --ratio_of_other --ratio_of_aug --epoch --lr`` and the booleans ``--scale --plt
--transfer --eval --test --logger --reset --rot --optim --color --D3``), same seeds (9001), Adam(lr) +
CosineAnnealingLR(T_max=epoch) stepped per epoch, best-validation-loss checkpoint
``<root_path>/<root>/<name>/checkpoint-good/state_dict.bin`` with the reference's dict keys, early stop on
``--count``.  New flags (they do not change existing semantics): ``--depth`` (ResNet depth, the reference
hard-codes 50), ``--hrnet_width``, ``--precision {fp32,bf16,fp16}``, ``--size``, ``--synthetic N`` (seeded
synthetic samples; the reference's datasets are not redistributable), ``--no_graph``, ``--transfer_from PATH`` (where
``--transfer`` reads its checkpoint; default: the reference's ``output/<model>/frei/ori/checkpoint-good/state_dict.bin``),
``--drop_last`` (skip the short last batch: the reference trains and validates on it, src/tools/train.py:27-38 builds both
loaders with the default ``drop_last=False``), ``--lr_resume_fix`` (on resume, continue the cosine schedule where the saved run
left it; the reference builds a FRESH ``CosineAnnealingLR`` behind ``optimizer.load_state_dict``, src/tools/train.py:50-58, and
that is the default here too).

Datasets (src/tools/train.py:24-38 builds them from files this repository cannot ship): ``main(args, train_set=,
val_set=)`` takes any ``torch.utils.data.Dataset`` whose samples are tuples starting with ``(image, joint_2d)`` --
the reference's ``CustomDataset`` yields ``(image, joint_2d, heatmap)`` and fits as it is; the heatmap is re-rendered on
the device from ``joint_2d``.  ``image`` is either a normalised float tensor ``[3, S, S]`` or a RAW uint8 frame
``[H, W, 3]``: raw frames go through the fused device pipeline (ToTensor, Resize(S), ColorJitter(0.5 x 4) on the
samples with ``idx < len(dataset) * ratio_of_aug`` -- the reference's rule, src/tools/dataset.py:133 -- Normalize).

What changes vs the reference loop (src/utils/method.py:160-183): the per-iteration ``loss.item()`` and
full-heatmap D2H + NumPy arg-max are replaced by device-resident loss / keypoints that are read once per
logging interval; everything else (BN momentum, loss, decode rule, optimizer) is the same arithmetic.

Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N -m lighthand_amd.tools.train ...``
(one process per GPU, gradients averaged by bucketed RCCL all-reduce; see lighthand_amd/parallel.py).
"""
import argparse
import os
import random
import time

import numpy as np
import torch


def parse_args(argv=None, phase="train"):
    p = argparse.ArgumentParser()
    p.add_argument("--root", default="simplebaseline/ours", type=str, help="You write down to store the directory path")
    p.add_argument("--name", default="84k", type=str, help="You write down to store the directory path")
    p.add_argument("--root_path", default="output", type=str, help="The root directory to save location which you want")
    known, _ = p.parse_known_args(argv)
    p.add_argument("--model", default="ours", type=str)
    p.add_argument("--dataset", default=known.root.split("/")[-1], type=str)
    p.add_argument("--view", default="wrist", type=str)
    p.add_argument("--batch_size", default=32, type=int)
    p.add_argument("--milestone", default=10, type=int)
    p.add_argument("--count", default=30, type=int)
    p.add_argument("--num_our", default=300000, type=int)
    p.add_argument("--ratio_of_other", default=0, type=float)
    p.add_argument("--ratio_of_aug", default=0.6, type=float)
    p.add_argument("--epoch", default=100, type=int)
    p.add_argument("--lr", default=0.001, type=float)
    for flag in ("scale", "plt", "transfer", "eval", "test", "logger", "reset", "rot", "optim", "color", "D3"):
        p.add_argument("--" + flag, action="store_true")
    # additions of this engine
    p.add_argument("--depth", default=50, type=int, choices=[18, 34, 50, 101, 152])
    p.add_argument("--hrnet_width", default=48, type=int)
    p.add_argument("--precision", default="bf16", choices=["fp32", "bf16", "fp16"])
    p.add_argument("--size", default=256, type=int)
    p.add_argument("--synthetic", default=0, type=int, help="train on N seeded synthetic samples")
    p.add_argument("--val_synthetic", default=0, type=int)
    p.add_argument("--no_graph", action="store_true")
    p.add_argument("--transfer_from", default=None, type=str, help="checkpoint --transfer loads (default: the reference's path)")
    p.add_argument("--drop_last", action="store_true", help="skip the short last batch of the training epoch (reference: trained on)")
    p.add_argument("--lr_resume_fix", action="store_true", help="resume the cosine schedule at the saved epoch (reference: fresh schedule)")
    args = p.parse_args(argv)
    args.phase = phase
    args.model = args.root.split("/")[0]                  # src/tools/dataset.py:59 overwrites it from the name
    args.name = os.path.join(args.root, args.name)
    args.output_dir = os.path.join(args.root_path, args.name)      # src/utils/pre_argparser.py:9
    args.logging_steps, args.num_workers, args.device = 100, 8, "cuda"
    if args.D3:
        raise SystemExit("--D3 (3-D joint regression) is outside the heatmap-regression path this engine covers")
    return args


def build_model(args):
    from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
    from lighthand_amd.modeling.simplebaseline.config import default_config
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    if args.model == "hrnet":
        return get_hrnet(hrnet_cfg(args.hrnet_width), is_train=True)
    return get_pose_net(default_config(args.depth), is_train=True)


class SyntheticHands(torch.utils.data.Dataset):
    """Seeded stand-in for CustomDataset (src/tools/dataset.py:103-163): returns (image[3,S,S] ~ N(0,1) like
    a normalised crop, joint_2d[21,2] uniform in [20, S-20]).  Heatmaps are rendered on the device."""

    def __init__(self, n, size, seed):
        rng = np.random.RandomState(seed)
        self.images = torch.from_numpy(rng.randn(n, 3, size, size).astype(np.float32))
        self.joints = torch.from_numpy(rng.uniform(20, size - 20, size=(n, 21, 2)).astype(np.float32))

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return self.images[i], self.joints[i]


class _WithAugFlag(torch.utils.data.Dataset):
    """(image, joint_2d, ...) -> (image, joint_2d, jitter?) with the reference's rule for the colour augmentation: the
    FIXED subset idx < len(dataset) * ratio_of_aug is jittered (src/tools/dataset.py:133)."""

    def __init__(self, base, ratio_of_aug):
        # the reference compares the sample index with len(self.meta) * ratio_of_aug (dataset.py:133), and len(meta) can
        # differ from __len__ (= num_our): a dataset that carries `meta` supplies the limit the same way
        meta = getattr(base, "meta", None)
        self.base, self.limit = base, (len(meta) if meta is not None else len(base)) * ratio_of_aug

    def __len__(self):
        return len(self.base)

    def __getitem__(self, i):
        s = self.base[i]
        return s[0], torch.as_tensor(s[1], dtype=torch.float32)[:, :2], i < self.limit


def _sample_kind(ds):
    """'u8' for raw uint8 HWC frames, 'f32' for normalised [3, S, S] float tensors; (H, W) of a raw frame."""
    img = torch.as_tensor(ds[0][0])
    if img.dtype == torch.uint8:
        if img.dim() != 3 or img.shape[2] != 3:
            raise SystemExit(f"raw frames must be uint8 [H, W, 3], got {tuple(img.shape)}")
        return "u8", (int(img.shape[0]), int(img.shape[1]))
    if img.dim() != 3 or img.shape[0] != 3:
        raise SystemExit(f"normalised images must be float [3, S, S], got {tuple(img.shape)}")
    return "f32", None


def save_checkpoint(model, args, epoch, optimizer, best_loss, count, ment="good"):
    """Same file name and dict keys as src/tools/dataset.py:340-367; only rank 0 writes."""
    d = os.path.join(args.output_dir, "checkpoint-{}".format(ment))
    if int(os.environ.get("RANK", "0")) != 0:
        return d
    os.makedirs(d, exist_ok=True)
    torch.save({"epoch": epoch, "optimizer_state_dict": optimizer.state_dict(), "best_loss": best_loss, "count": count,
                "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}},
               os.path.join(d, "state_dict.bin"))
    return d


def resume_checkpoint(model, path):
    """src/utils/dir.py:38-47: strict=False load, epoch + 1."""
    sd = torch.load(path, map_location="cpu")
    model.load_state_dict(sd["model_state_dict"], strict=False)
    return sd["best_loss"], sd["epoch"] + 1, sd["count"], sd.get("optimizer_state_dict")


def transfer_path(args):
    """Where --transfer reads its weights: src/utils/argparser.py:167-175 hard-codes output/<model>/frei/ori (relative to the
    working directory, NOT --root_path)."""
    return args.transfer_from or os.path.join(f"output/{args.model}/frei/ori", "checkpoint-good/state_dict.bin")


def load_model_state(model, args):
    """The checkpoint logic of load_model (src/utils/argparser.py:100-189) on an already built model: resume from
    <output_dir>/checkpoint-good/state_dict.bin unless --reset, THEN (--transfer) overwrite the weights with those of the transfer
    checkpoint, keeping the resumed epoch / best loss / count / optimizer state (the reference discards the transfer checkpoint's:
    `_, _, _model, _, _ = resume_checkpoint(...)`, :169).  A missing transfer checkpoint is an error, as in the reference (torch.load).
    Returns (best_loss, first epoch, count, optimizer_state | None)."""
    best_loss, epo, count, opt_state = np.inf, 0, 0, None
    ckpt = os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin")
    if os.path.isfile(ckpt) and not args.reset:
        best_loss, epo, count, opt_state = resume_checkpoint(model, ckpt)
    if args.transfer:
        resume_checkpoint(model, transfer_path(args))
        if int(os.environ.get("RANK", "0")) == 0:
            print("Transfer_Loading ===> %s" % transfer_path(args))
    return best_loss, epo, count, opt_state


def make_scheduler(optimizer, args, epo, opt_state):
    """Optimizer state + learning-rate schedule at (re)start, src/tools/train.py:45-58: the saved optimizer state is loaded unless
    --optim, THEN a fresh CosineAnnealingLR(T_max=args.epoch) is built on the optimizer -- on a resume its param_groups carry the
    saved run's `lr` and `initial_lr`, so the schedule restarts its cosine from the saved learning rate (what the installed PyTorch
    does with such a group is the reference's behaviour, not restated here).  --lr_resume_fix instead fast-forwards a schedule that
    starts at args.lr by the epochs already run.  Call it AFTER the TrainStep exists (the fused Adam binds the arena there)."""
    if opt_state and not args.optim:
        optimizer.load_state_dict(opt_state)
    if getattr(args, "lr_resume_fix", False):
        for g in optimizer.param_groups:
            g["lr"] = args.lr
            g.pop("initial_lr", None)
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=args.epoch)
        for _ in range(epo):
            scheduler.step()
        return scheduler
    return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=args.epoch)


def run_epochs(args, epo, train_loader, train_batch, validate_fn, save_fn, optimizer, scheduler, stopper, log=print, rank=0, world=1):
    """The epoch loop of src/tools/train.py:60-117 with the device work behind three callables: train_batch(it, batch) runs one
    iteration (any batch size: the short last batch included) and returns a callable that reads the running loss (called once per
    logging interval: the only device->host read of the loop), validate_fn() -> (val_loss, pck, epe), save_fn(epoch, best, count).
    Same order as the reference: train, validate, best / count bookkeeping, checkpoint on improvement, break on count == --count,
    THEN scheduler.step()."""
    for epoch in range(epo, args.epoch):
        t0, seen = time.time(), 0
        for it, batch in enumerate(train_loader):
            read_loss = train_batch(it, batch)
            seen += len(batch[0])
            if it % args.logging_steps == 0 and rank == 0:
                log(f"epoch {epoch} iter {it}/{len(train_loader)} loss {float(read_loss()):.6f} "
                    f"{world * seen / (time.time() - t0 + 1e-9):.0f} img/s lr {optimizer.param_groups[0]['lr']:.2e}")
        val_loss, pck, epe = validate_fn()
        if rank == 0:
            log(f"epoch {epoch} valid loss {val_loss:.6f} pck {pck:.2f}% epe {epe * 0.26:.2f} mm")     # method.py:131
        improved, stop = stopper.update(val_loss)     # val_loss is rank-invariant (reduce_validation): collective decision
        if improved:
            save_fn(epoch, stopper.best_loss, stopper.count)
        if stop:
            break
        scheduler.step()
    return stopper.best_loss


def validate(model, loader, args, u8_step=None):
    """Runner.run validation branch (src/utils/method.py:218-287): loss, PCK@0.2 (bbox-normalised), EPE.
    ``u8_step``: an InferStep(input_u8=...) of the model for loaders that yield raw uint8 frames, or a callable
    batch size -> InferStep (the short last batch of the loader needs a step of its own shape)."""
    from lighthand_amd.heatmap import JointsMSELoss, max_preds_device, render_targets
    from lighthand_amd.metrics import device_pck_epe
    model.eval()
    crit = JointsMSELoss(False)
    acc = torch.zeros(5, device="cuda")          # loss*b, b, pck*b, epe sum, epe count -- reduced on the device
    with torch.no_grad():
        for batch in loader:
            images, joints = batch[0].cuda(non_blocking=True), batch[1][..., :2].float().cuda(non_blocking=True)
            if images.dtype == torch.uint8:
                st = u8_step(images.shape[0]) if callable(u8_step) and not hasattr(u8_step, "heatmaps") else u8_step
                st.refresh_weights()
                st(images)
                pred = st.heatmaps
            else:
                pred = model(images)
            hs = pred.shape[-1]
            target = render_targets(joints, size=hs)
            loss = crit(pred, target, None)
            kp, _, _ = max_preds_device(pred, scale=float(args.size // hs))
            b = images.shape[0]
            pck, esum, ecnt = device_pck_epe(kp, joints, T=0.2)
            acc += torch.stack([loss * b, torch.tensor(float(b), device="cuda"), pck * b, esum, ecnt])
    model.train()
    return reduce_validation(acc)


def reduce_validation(acc):
    """acc = [loss*b, b, pck*b, epe sum, epe count] of THIS rank's validation shard.  Summed over the data-parallel
    ranks first, so every rank derives the same (loss, pck, epe) and therefore the same best-checkpoint / early-stop
    decision: a rank leaving the epoch loop alone would leave the others inside the next bucketed all-reduce."""
    from lighthand_amd import parallel
    a = parallel.all_reduce_sum_(acc).tolist()     # the only host read of the validation pass
    return a[0] / max(a[1], 1), 100.0 * a[2] / max(a[1], 1), a[3] / max(a[4], 1)


class EarlyStop:
    """best-loss / patience bookkeeping of the reference loop (src/tools/train.py:84-112)."""

    def __init__(self, best_loss, count, patience):
        self.best_loss, self.count, self.patience = best_loss, count, patience

    def update(self, val_loss):
        """Returns (improved, stop)."""
        if self.best_loss > val_loss:
            self.best_loss, self.count = val_loss, 0
            return True, False
        self.count += 1
        return False, self.count == self.patience


def main(args, train_set=None, val_set=None):
    """``train_set`` / ``val_set``: any Dataset of (image, joint_2d, ...) samples (module docstring); without them
    ``--synthetic N`` builds seeded synthetic ones.  Under torch.distributed every rank must be given ITS shard."""
    from lighthand_amd import parallel
    from lighthand_amd.optim import Adam
    from lighthand_amd.runtime import InferStep, TrainStep

    seed = 9001                                           # src/tools/train.py:15-22
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    if train_set is None:
        if not args.synthetic:
            raise SystemExit("the reference's datasets (LightHand99K / FreiHAND / ...) are not shipped: pass --synthetic N, or call "
                             "lighthand_amd.tools.train.main(args, train_set=, val_set=) with Datasets of (image, joint_2d) samples")
        train_set = SyntheticHands(args.synthetic, args.size, seed + rank)
    if val_set is None:
        val_set = SyntheticHands(args.val_synthetic or max(args.batch_size, max(args.synthetic, args.batch_size * 8) // 8), args.size, seed + 1000)
    kind, raw_hw = _sample_kind(train_set)
    val_kind, val_hw = _sample_kind(val_set)
    train_set = _WithAugFlag(train_set, args.ratio_of_aug)
    workers = args.num_workers if not isinstance(train_set.base, SyntheticHands) else 0
    # persistent workers: a worker is forked from THIS process, which by the first epoch holds a HIP context and hundreds of
    # GB of mappings (~20 s per fork measured on the GPU box) -- fork them once per loader, not once per epoch
    # the reference builds both loaders with drop_last=False (src/tools/train.py:27-38): the short last batch is trained on and
    # validated on.  Data-parallel training keeps whole batches only (every rank must take part in every bucket's exchange, and the
    # reference has no multi-GPU loop to be faithful to); --drop_last asks for that on one GPU too.
    kw = dict(batch_size=args.batch_size, pin_memory=True, num_workers=workers, persistent_workers=workers > 0)
    train_loader = torch.utils.data.DataLoader(train_set, shuffle=True, drop_last=bool(args.drop_last or world > 1), **kw)
    val_loader = torch.utils.data.DataLoader(val_set, shuffle=False, drop_last=False, **kw)

    model = build_model(args).cuda().set_precision(args.precision)
    best_loss, epo, count, opt_state = load_model_state(model, args)
    optimizer = Adam(model.parameters(), lr=args.lr)
    sync = parallel.GradSync(world) if world > 1 else None
    # raw uint8 frames: ToTensor / Resize / ColorJitter(0.5, 0.5, 0.5, 0.5) / Normalize fused on the device (dataset.py:128-159)
    jitter = (0.5, 0.5, 0.5, 0.5) if kind == "u8" and args.ratio_of_aug > 0 else None
    step = TrainStep(model, args.batch_size, args.size, args.size, optimizer=optimizer, use_graph=not args.no_graph, grad_sync=sync,
                     input_u8=raw_hw, color_jitter=jitter)
    scheduler = make_scheduler(optimizer, args, epo, opt_state)       # src/tools/train.py:50-58, in the reference's order
    steps = {args.batch_size: step}

    def step_for(b):
        """The short last batch of an epoch runs on a step of its own shape (plans are per static shape): eager launches, the
        library's static kernel choice (one batch per epoch is not worth a measurement), the same optimizer -- moments and step
        count live in the optimizer, so full and short batches update one Adam state, as in the reference loop."""
        st = steps.get(b)
        if st is None:
            prev = os.environ.get("LH_AUTOTUNE")
            os.environ["LH_AUTOTUNE"] = "0"
            try:
                st = steps[b] = TrainStep(model, b, args.size, args.size, optimizer=optimizer, use_graph=False, grad_sync=sync,
                                          input_u8=raw_hw, color_jitter=jitter)
            finally:
                if prev is None:
                    os.environ.pop("LH_AUTOTUNE", None)
                else:
                    os.environ["LH_AUTOTUNE"] = prev
        return st

    val_steps = {}

    def val_step_for(b):
        if b not in val_steps:
            val_steps[b] = InferStep(model, b, args.size, args.size, input_u8=val_hw)
        return val_steps[b]

    def train_batch(it, batch):
        images, joints, aug = batch
        st = step_for(images.shape[0])
        st(images.cuda(non_blocking=True), joints.cuda(non_blocking=True), aug=aug if jitter else None)
        return lambda: st.loss

    stopper = EarlyStop(best_loss, count, args.count)
    return run_epochs(args, epo, train_loader, train_batch,
                      lambda: validate(model, val_loader, args, val_step_for if val_kind == "u8" else None),
                      lambda epoch, best, cnt: save_checkpoint(model, args, epoch, optimizer, best, cnt, "good"),
                      optimizer, scheduler, stopper, rank=rank, world=world)


if __name__ == "__main__":
    main(parse_args())
