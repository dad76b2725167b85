import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from test_gpu_model import _build, rel
from oracle import models as omod
from lighthand_amd.heatmap import JointsMSELoss
for tag in sys.argv[1:] or ['mini_basic']:
    torch.manual_seed(11)
    model, fwd = _build(tag)
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(4, 3, 128, 128).astype(np.float32))
    tgt = torch.from_numpy(rng.rand(4, 21, 32, 32).astype(np.float32))
    sd = omod.clone_state(model.state_dict())
    loss_ref, pred_ref, g32 = omod.loss_and_grads(sd, lambda s, xx: fwd(s, xx, True), x, tgt)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    _, _, g64 = omod.loss_and_grads(sd64, lambda s, xx: fwd(s, xx, True), x.double(), tgt.double())
    model = model.cuda().train()
    pred = model(x.cuda())
    loss = JointsMSELoss(False)(pred, tgt.cuda(), None)
    loss.backward()
    rows=[]
    for k,p in model.named_parameters():
        gh, gc, gt = p.grad.cpu().double().numpy(), g32[k].double().numpy(), g64[k].numpy()
        rows.append((k, rel(gh,gt), rel(gc,gt), float(np.abs(gt).max())))
    for r in rows: print('  %-45s hip %.3e cpu %.3e  |g|max %.3e'%r)
