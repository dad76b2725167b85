"""CPU restatement of the reference's HOST loop (epochs, loaders, checkpoint / resume / transfer, learning-rate schedule).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, statement by statement:
  * src/tools/train.py:13-120 `main` -- both DataLoaders with the default drop_last=False (:27-38), Adam(lr) (:45-48),
    `optimizer.load_state_dict(optimizer_state)` unless --optim (:50), a FRESH CosineAnnealingLR(T_max=args.epoch) behind it (:58),
    per epoch: train, valid, `is_best = best_loss > val_loss` (:93), checkpoint on improvement with count = 0 (:96-108), else
    count += 1 and break at count == args.count (:111-114), then lr_scheduler.step() (:117);
  * src/utils/argparser.py:100-189 `load_model` -- resume from <root_path>/<name>/checkpoint-good/state_dict.bin unless --reset,
    then --transfer overwrites the weights from output/<model>/frei/ori/checkpoint-good/state_dict.bin and keeps everything else (:167-175);
  * src/utils/dir.py:38-47 `resume_checkpoint` (strict=False, epoch + 1), src/tools/dataset.py:340-367 `save_checkpoint` (dict keys).
The device work (Runner_t.run, src/utils/method.py:158-287) is injected: `train_batch(model, batch)` and `valid_loss(epoch)`.
Pinned by construction: it is run against plain torch.optim / torch.utils.data of the installed PyTorch, the same classes the reference
instantiates; the reference ships no fixture for its loop."""
import os

import numpy as np
import torch
from torch.utils import data


def resume_checkpoint(model, path):
    sd = torch.load(path, map_location="cpu")
    model.load_state_dict(sd["model_state_dict"], strict=False)
    return sd["best_loss"], sd["epoch"] + 1, sd["count"], sd["optimizer_state_dict"]


def load_model(model, output_dir, reset, transfer, transfer_ckpt):
    epoch, best_loss, count, optimizer_state = 0, np.inf, 0, 0
    ckpt = os.path.join(output_dir, "checkpoint-good/state_dict.bin")
    if not reset and os.path.isfile(ckpt):
        best_loss, epoch, count, optimizer_state = resume_checkpoint(model, ckpt)
    if transfer:
        resume_checkpoint(model, transfer_ckpt)            # `_, _, _model, _, _ = resume_checkpoint(...)`: only the weights are kept
    return best_loss, epoch, count, optimizer_state


def save_checkpoint(model, output_dir, epoch, optimizer, best_loss, count):
    d = os.path.join(output_dir, "checkpoint-good")
    os.makedirs(d, exist_ok=True)
    torch.save({"epoch": epoch, "optimizer_state_dict": optimizer.state_dict(), "best_loss": best_loss, "count": count,
                "model_state_dict": model.state_dict()}, os.path.join(d, "state_dict.bin"))


def main(model, train_dataset, val_dataset, output_dir, batch_size, epochs, lr, patience, train_batch, valid_loss,
         reset=False, optim=False, transfer=False, transfer_ckpt=None, seed=9001):
    """Returns the trace: one entry per epoch = dict(epoch, lr, batch_sizes (train), val_batch_sizes, val_loss, saved, count)."""
    torch.manual_seed(seed)
    trainset_loader = data.DataLoader(dataset=train_dataset, batch_size=batch_size, num_workers=0, shuffle=True)
    valset_loader = data.DataLoader(dataset=val_dataset, batch_size=batch_size, num_workers=0, shuffle=False)
    best_loss, epo, count, optimizer_state = load_model(model, output_dir, reset, transfer, transfer_ckpt)
    optimizer = torch.optim.Adam(params=list(model.parameters()), lr=lr)
    if optimizer_state and not optim:
        optimizer.load_state_dict(optimizer_state)
    lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=epochs)
    trace = []
    for epoch in range(epo, epochs):
        ent = dict(epoch=epoch, lr=optimizer.param_groups[0]["lr"], batch_sizes=[], val_batch_sizes=[])
        for batch in trainset_loader:
            ent["batch_sizes"].append(len(batch[0]))
            train_batch(model, optimizer, batch)
        for batch in valset_loader:
            ent["val_batch_sizes"].append(len(batch[0]))
        val_loss = valid_loss(epoch)
        is_best = best_loss > val_loss
        best_loss = min(val_loss, best_loss)
        ent.update(val_loss=val_loss, saved=bool(is_best))
        if is_best:
            count = 0
            save_checkpoint(model, output_dir, epoch, optimizer, best_loss, count)
        else:
            count += 1
        ent["count"] = count
        trace.append(ent)
        if not is_best and count == patience:
            break
        lr_scheduler.step()
    return trace
