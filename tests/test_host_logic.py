"""CPU: host-side logic of the drop-in surface -- flag parsing, configs, metrics (vs the oracle and the
golden vectors), graph description / bucket planning.  No compute kernels are called."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import resnet_cfg


def test_cli_flags_and_defaults_match_reference():
    from lighthand_amd.tools.train import parse_args
    a = parse_args([])
    # src/utils/argparser.py:27-100 defaults
    assert (a.root, a.batch_size, a.milestone, a.count, a.num_our, a.epoch, a.lr) == ("simplebaseline/ours", 32, 10, 30, 300000, 100, 0.001)
    assert (a.ratio_of_other, a.ratio_of_aug, a.view, a.dataset) == (0, 0.6, "wrist", "ours")
    assert a.name == os.path.join("simplebaseline/ours", "84k") and a.output_dir == os.path.join("output", a.name)
    assert a.model == "simplebaseline" and a.logging_steps == 100 and a.num_workers == 8 and a.device == "cuda"
    for flag in ("scale", "plt", "transfer", "eval", "test", "logger", "reset", "rot", "optim", "color", "D3"):
        assert getattr(a, flag) is False
    b = parse_args(["--root", "hrnet/frei", "--name", "x", "--batch_size", "8", "--reset", "--epoch", "3"])
    assert b.model == "hrnet" and b.dataset == "frei" and b.reset and b.epoch == 3 and b.batch_size == 8
    with pytest.raises(SystemExit):
        parse_args(["--D3"])


def test_simplebaseline_config_surface(tmp_path):
    from lighthand_amd.modeling.simplebaseline.config import config, default_config, get_model_name, update_config
    assert config.MODEL.EXTRA.NUM_LAYERS == 50 and config.MODEL.STYLE == "pytorch" and config.MODEL.NUM_JOINTS == 21
    assert config.MODEL.EXTRA.NUM_DECONV_FILTERS == [256, 256, 256] and config.MODEL.EXTRA.FINAL_CONV_KERNEL == 1
    cfg = default_config()
    p = tmp_path / "c.yaml"
    p.write_text("MODEL:\n  EXTRA:\n    NUM_LAYERS: 18\n")
    update_config(str(p), cfg)
    assert cfg.MODEL.EXTRA.NUM_LAYERS == 18 and get_model_name(cfg)[0] == "pose_resnet_18"
    p.write_text("NOPE: 1\n")
    with pytest.raises(ValueError):
        update_config(str(p), cfg)


def test_metrics_match_oracle_and_golden(golden_dir):
    from lighthand_amd import metrics as M
    from oracle import metrics as om
    g = json.load(open(os.path.join(golden_dir, "g7_metrics.json")))
    cats = g["evaluation"][0]
    for key, T, method in (("pckb", [0.1, 0.3], "pckb"), ("mm30", [0, 30], "mm"), ("mm50", [0, 50], "mm")):
        got, ora = M.pred_eval(cats, T, method), om.pred_eval(cats, T, method)
        for cat, (auc, epe, curve) in g["pred_eval"][key].items():
            assert abs(got[cat][0] - auc) < 1e-9 * max(1, abs(auc)) and abs(got[cat][1] - epe) < 1e-9 * max(1, abs(epe))
            assert got[cat][0] == ora[cat][0] and np.allclose(got[cat][2], curve)
    pred, gt = torch.tensor(g["val_pred"]), torch.tensor(g["val_gt"])
    assert abs(M.PCK_2d_loss(pred, gt, T=0.2) - g["pck02"]) < 1e-12
    assert abs(M.PCK_2d_loss(pred, gt, T=5.0, threshold="mm") - g["pck_mm5"]) < 1e-12
    (s, c), dist = M.EPE_train(pred, gt)
    assert c == g["epe_cnt"] and abs(s - g["epe_sum"]) < 1e-4 * g["epe_sum"] and len(dist) == 20
    with pytest.raises(AssertionError):
        M.PCK_2d_loss(pred, gt, threshold="bogus")
    # the category-less variant (pred_store_test -> pred_test): golden, oracle, and a short last batch
    g8 = json.load(open(os.path.join(golden_dir, "g8_pred_test.json")))
    meta = g8["test"][0]
    for key, T, method in (("pckb", [0.1, 0.3], "pckb"), ("mm30", [0, 30], "mm"), ("mm50", [0, 50], "mm")):
        auc, epe = M.pred_test(meta, T, method)
        assert abs(auc - g8["pred_test"][key][0]) < 1e-9 * auc and abs(epe - g8["pred_test"][key][1]) < 1e-9 * epe
        assert (auc, epe) == om.pred_test(meta, T, method)
    ragged = {k: [v[0], v[1], v[2][:5]] for k, v in meta.items()}
    flat = {k: [[x for b in ragged[k] for x in b]] for k in ragged}
    assert M.pred_test(ragged, [0, 30], "mm") == M.pred_test(flat, [0, 30], "mm")


@pytest.mark.parametrize("tag,nodes", [("r18", None), ("r50", None), ("hrnet_w32", None)])
def test_graph_description_covers_every_parameter(tag, nodes):
    """describe() must reference each conv / BN parameter exactly once (no layer dropped or duplicated)."""
    from lighthand_amd.engine import GraphBuilder
    from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    model = get_hrnet(hrnet_cfg(32), True) if tag.startswith("hrnet") else get_pose_net(resnet_cfg(int(tag[1:])), True)
    params = dict(model.state_dict(keep_vars=True))
    gb = GraphBuilder(2, 64, 64, params)
    model.describe(gb)
    used = []
    for kind, nd in gb.nodes:
        if kind in ("conv", "deconv"):
            used.append(nd["w"] + ".weight")
            if nd["bias"]:
                used.append(nd["bias"])
        elif kind == "fuse":
            for _, bn, _ in nd["terms"]:
                if bn:
                    used += [bn + ".weight", bn + ".bias"]
    want = sorted(k for k, v in params.items() if isinstance(v, torch.nn.Parameter))
    assert sorted(used) == want
    assert gb.out is not None and (gb.out.h, gb.out.w, gb.out.c_valid) == (16, 16, 21)


def test_bucket_planner_covers_arena_in_backward_order():
    from lighthand_amd.parallel import plan_buckets
    offsets, off = {}, 0
    names = [f"p{i}" for i in range(10)]
    for i, k in enumerate(names):
        offsets[k] = (off, 1000 * (i + 1), (1000 * (i + 1),))
        off += 1000 * (i + 1)
    # backward finishes parameters in reverse order, two per mark, with one out-of-order pair
    order = [["p9", "p8"], ["p6", "p7"], ["p5"], ["p3", "p4"], ["p2"], ["p0", "p1"]]
    marks = [(10 * (i + 1), ps) for i, ps in enumerate(order)]
    segs = plan_buckets(marks, offsets, off, bucket_bytes=4 * 12000)
    covered = sorted(b for _, _, b in segs if b)
    assert covered[0][0] == 0 and covered[-1][1] == off
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))      # contiguous, no overlap
    assert [s[0] for s in segs] == [0] + [s[1] for s in segs[:-1]] and segs[-1][1] == marks[-1][0]
    assert len(segs) > 1
    with pytest.raises(RuntimeError):
        plan_buckets(marks[:-1], offsets, off, bucket_bytes=1 << 30)


def test_models_refuse_cpu_execution():
    from lighthand_amd import LightHandError
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    m = get_pose_net(resnet_cfg(18), True)
    with pytest.raises(LightHandError):
        m(torch.randn(1, 3, 64, 64))


def test_init_weights_rules_of_both_factories(tmp_path):
    """init_weights (pose_resnet.py:250-298, pose_hrnet.py:462-492; never called by the reference's factories, kept for
    API parity): head re-initialisation, 'module.' prefix stripping, strict=False, PRETRAINED_LAYERS filter, errors."""
    from collections import OrderedDict
    from lighthand_amd.modeling.hrnet.pose_hrnet import get_hrnet, hrnet_cfg
    from lighthand_amd.modeling.simplebaseline.pose_resnet import get_pose_net
    torch.manual_seed(0)
    m = get_pose_net(resnet_cfg(18), True)
    w1, w2 = torch.randn_like(m.conv1.weight), torch.randn_like(m.layer1[0].conv1.weight)
    path = str(tmp_path / "imagenet.pth")
    torch.save({"state_dict": OrderedDict([("module.conv1.weight", w1), ("layer1.0.conv1.weight", w2),
                                           ("module.fc.weight", torch.zeros(10, 512))])}, path)
    before = m.layer2[0].conv1.weight.detach().clone()
    m.init_weights(path)
    assert torch.equal(m.conv1.weight, w1) and torch.equal(m.layer1[0].conv1.weight, w2)
    assert torch.equal(m.layer2[0].conv1.weight, before)                    # keys absent from the file stay
    assert 5e-4 < float(m.deconv_layers[0].weight.std()) < 2e-3 and float(m.final_layer.bias.abs().max()) == 0.0
    assert float((m.deconv_layers[1].weight - 1).abs().max()) == 0.0
    with pytest.raises(ValueError):
        m.init_weights(str(tmp_path / "missing.pth"))

    h = get_hrnet(hrnet_cfg(32), True)
    c1, fl = torch.randn_like(h.conv1.weight), torch.randn_like(h.final_layer.weight)
    hpath = str(tmp_path / "hrnet.pth")
    torch.save(OrderedDict([("conv1.weight", c1), ("final_layer.weight", fl)]), hpath)
    h.init_weights(hpath)
    assert torch.equal(h.conv1.weight, c1)                                    # 'conv1' is in PRETRAINED_LAYERS
    assert not torch.equal(h.final_layer.weight, fl) and float(h.final_layer.weight.std()) < 2e-3   # 'final_layer' is not
    h.init_weights("")                                                        # no file requested: re-initialise only
    with pytest.raises(ValueError):
        h.init_weights(str(tmp_path / "missing.pth"))


def test_tuning_file_is_versioned_and_tolerates_bad_lines(tmp_path, monkeypatch):
    """The persisted kernel choices (engine.Plan._tune_cache_io): a file written by ANOTHER build of the kernel library
    (first line = size + mtime of the .so) is ignored as a whole; unparsable lines of a matching file are skipped instead
    of failing every Plan constructor; a save writes the stamp and only locally measured entries."""
    from lighthand_amd.engine import Plan
    path = tmp_path / "tune.txt"
    monkeypatch.setenv("LH_TUNE_CACHE", str(path))
    monkeypatch.setenv("LH_TUNE_DB", "0")
    saved = dict(Plan._TUNE_CACHE), set(Plan._tune_measured), Plan._tune_file_loaded
    try:
        good = repr((("k", 1), (128, 128, 2, 64)))
        path.write_text("# lib 1 2\n" + good + "\n")                               # another build's stamp
        Plan._TUNE_CACHE.clear(); Plan._tune_measured.clear(); Plan._tune_file_loaded = False
        Plan._tune_cache_io()
        assert ("k", 1) not in Plan._TUNE_CACHE
        path.write_text("# " + Plan._lib_stamp() + "\n" + good + "\nthis is not a tuple\n((\"k\", 2), (64,\n")
        Plan._TUNE_CACHE.clear(); Plan._tune_measured.clear(); Plan._tune_file_loaded = False
        Plan._tune_cache_io()
        assert Plan._TUNE_CACHE == {("k", 1): (128, 128, 2, 64)} and Plan._tune_measured == {("k", 1)}
        Plan._TUNE_CACHE[("k", 3)] = (64, 64, 2, 64)                                # not measured locally: a save must not write it
        Plan._tune_cache_io(save=True)
        lines = path.read_text().splitlines()
        assert lines[0] == "# " + Plan._lib_stamp() and lines[1:] == [good]
    finally:
        Plan._TUNE_CACHE.clear(); Plan._TUNE_CACHE.update(saved[0])
        Plan._tune_measured.clear(); Plan._tune_measured.update(saved[1])
        Plan._tune_file_loaded = saved[2]


def test_host_side_sanitizer_build_is_clean():
    """SURVEY section 5 (sanitizers), round-4 verdict item 7: `make -C lighthand_amd/csrc asan` builds the library with
    AddressSanitizer + UBSan on its HOST code, and tools/asan_host_check.py drives argument validation and every planner /
    size-query entry point over the R18 / R50 / HRNet convolution tables under it.  CPU only (no kernel is launched)."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not hipcc or not os.path.exists(clang):
        pytest.skip("no hipcc / clang on this machine")
    r = subprocess.run(["make", "-C", os.path.join(root, "lighthand_amd", "csrc"), "-j8", "asan"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1",
               LH_LIB_PATH=os.path.join(root, "lighthand_amd", "liblighthand_hip_asan.so"))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asan_host_check.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_pmc_traffic_tool_and_profile_guard(tmp_path):
    """tools/pmc_traffic.py on synthetic rocprofv3 counter tables: per-kernel bytes per launch (KiB unit, gfx950 x2 read correction) and,
    for a `bench.py --train-only` process (the once-per-step arg-max ran as often as Adam), the bytes all kernels moved per training
    step -- bench.py's roofline.step_traffic; a process that also replayed the inference graph gets no such entry.  And
    bench.profile_average_ns: the AverageNs bench.py compares with its live average before it attaches a committed PMC figure."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conv = "_Z17igemm_ring_kernelIDF16bLi128ELi128ELi2ELi4ELi3ELi128EEv9IgemmArgs"

    def table(path, counter, rows):
        with open(path, "w") as f:
            f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
            for i, (k, v) in enumerate(rows):
                f.write(f'{i + 1},"{k}","{counter}",{v}\n')
    steps = 4
    one_step = [(conv, 1000.0)] * 2 + [("heatmap_argmax_kernel(float const*)", 10.0), ("adam_kernel", 500.0)]
    # plan construction (fills) in front of the first step; the first (eager) step; then the steady state
    kernels = [("FillFunctor<float>", 7000.0)] * 3 + one_step * steps
    table(tmp_path / "f.csv", "FETCH_SIZE", kernels)
    table(tmp_path / "w.csv", "WRITE_SIZE", [(k, v / 2) for k, v in kernels])
    out_txt, out_json = tmp_path / "o.txt", tmp_path / "o.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), str(tmp_path / "f.csv"), str(tmp_path / "w.csv"), str(out_txt), str(out_json)], check=True)
    got = json.load(open(out_json))
    k = "igemm_ring_kernel<__bf16, 128, 128, 2, 4, 3, 128>"
    assert got[k] == {"read_bytes_per_launch": 2 * 1000 * 1024, "write_bytes_per_launch": 500 * 1024, "launches": 2 * steps}
    per_step_read = 2 * 1024 * (2 * 1000 + 500 + 10)
    per_step_write = 1024 * (2 * 500 + 250 + 5)
    # the steady state = what was dispatched behind the first Adam launch, over the Adam launches behind it: the construction fills
    # (and the first step) are not in it; round 5's whole-process figure is kept beside it
    assert got["__train_step__"]["steps"] == steps - 1
    assert got["__train_step__"]["read_bytes"] == per_step_read and got["__train_step__"]["write_bytes"] == per_step_write
    assert got["__train_step__"]["bytes"] == per_step_read + per_step_write
    assert got["__train_step__"]["whole_process_bytes_per_step"] == round(per_step_read + per_step_write + 3 * (2 * 7000 + 3500) * 1024 / steps)
    # an inference graph in the process: more arg-max launches than Adam launches -> no per-step figure
    table(tmp_path / "f2.csv", "FETCH_SIZE", kernels + [("heatmap_argmax_kernel(float const*)", 10.0)] * 4)
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), str(tmp_path / "f2.csv"), str(tmp_path / "w.csv"), str(out_txt), str(out_json)], check=True)
    assert "__train_step__" not in json.load(open(out_json))
    sys.path.insert(0, root)
    import bench
    stats = tmp_path / "stats.csv"
    stats.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                     f'"{conv}",10,300000,30000.5,9.0,1,2,3\n"adam_kernel(float*)",1,1,150000.0,1,1,1,0\n')
    assert bench.profile_average_ns(str(stats), k) == 30000.5
    assert bench.profile_average_ns(str(stats), "no_such_kernel<float>") is None
    assert bench.profile_average_ns(str(tmp_path / "missing.csv"), k) is None


def _loop_pieces(tmp_path, tag):
    """A tiny CPU model + index datasets for the host-loop tests: 2.5 training batches, 1.5 validation batches."""
    import torch

    class Idx(torch.utils.data.Dataset):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return torch.full((3,), float(i)), torch.zeros(21, 2)

    torch.manual_seed(0)
    model = torch.nn.Linear(3, 2)
    return model, Idx(20), Idx(12), str(tmp_path / tag)


def test_epoch_loop_matches_the_reference_loop(tmp_path):
    """tools/train.py's host loop (run_epochs + load_model_state + make_scheduler) against the oracle's restatement of
    src/tools/train.py:13-120 / src/utils/argparser.py:100-189 on the CPU, device work replaced by a real Adam step on a tiny model:
    a 2.5-batch epoch (the short last batch IS trained and validated on, train.py:27-38), best-loss / count / early-stop decisions,
    and a RESUME: the saved optimizer state is loaded and a fresh CosineAnnealingLR is built behind it (train.py:50-58) -- the
    learning rate of every epoch must equal the oracle's, as must the weights at the end.  --lr_resume_fix is the opt-in
    continuation of the first run's schedule; --transfer overwrites the weights and keeps the resumed counters."""
    import copy
    import types
    import torch
    from lighthand_amd.tools import train as T
    from oracle import loop as O

    val_seq = [1.0, 0.8, 0.9, 0.7, 0.75, 0.76, 0.77, 0.5, 0.6, 0.61, 0.62, 0.63]

    def train_batch_ref(model, optimizer, batch):
        optimizer.zero_grad()
        model(batch[0]).pow(2).mean().backward()
        optimizer.step()

    def run_product(model, train_set, val_set, out_dir, epochs, extra=()):
        args = T.parse_args(["--root_path", out_dir, "--batch_size", "8", "--epoch", str(epochs), "--count", "3", "--lr", "0.01", *extra])
        args.logging_steps = 2
        torch.manual_seed(9001)
        tl = torch.utils.data.DataLoader(train_set, batch_size=8, shuffle=True, drop_last=args.drop_last)
        vl = torch.utils.data.DataLoader(val_set, batch_size=8, shuffle=False, drop_last=False)
        best, epo, count, opt_state = T.load_model_state(model, args)
        optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)
        scheduler = T.make_scheduler(optimizer, args, epo, opt_state)
        trace, cur = [], {}

        def train_batch(it, batch):
            if it == 0:
                cur.update(lr=optimizer.param_groups[0]["lr"], batch_sizes=[])
            cur["batch_sizes"].append(len(batch[0]))
            train_batch_ref(model, optimizer, batch)
            return lambda: 0.0

        def validate_fn():
            cur["val_batch_sizes"] = [len(b[0]) for b in vl]
            cur["epoch"] = epo + len(trace)
            return val_seq[cur["epoch"]], 0.0, 0.0

        def save_fn(epoch, b, c):
            cur["saved"] = True
            T.save_checkpoint(model, args, epoch, optimizer, b, c)

        stopper = T.EarlyStop(best, count, args.count)
        orig_update = stopper.update

        def update(v):
            r = orig_update(v)
            trace.append(dict(cur, val_loss=v, saved=r[0], count=stopper.count))
            cur.clear()
            return r
        stopper.update = update
        T.run_epochs(args, epo, tl, train_batch, validate_fn, save_fn, optimizer, scheduler, stopper, log=lambda *_: None)
        return trace, args

    m0, train_set, val_set, out_p = _loop_pieces(tmp_path, "product")
    _, _, _, out_o = _loop_pieces(tmp_path, "oracle")
    mp, mo = copy.deepcopy(m0), copy.deepcopy(m0)
    # first run: best at epoch 3 (0.7), then 0.75, 0.76, 0.77 -> count reaches --count = 3 at epoch 6: stopped after 7 of 12 epochs
    tp, args = run_product(mp, train_set, val_set, out_p, 12)
    to = O.main(mo, train_set, val_set, os.path.join(out_o, "simplebaseline/ours/84k"), 8, 12, 0.01, 3, train_batch_ref, lambda e: val_seq[e])
    assert [e["batch_sizes"] for e in tp] == [e["batch_sizes"] for e in to] == [[8, 8, 4]] * len(to)
    assert [e["val_batch_sizes"] for e in tp] == [[8, 4]] * len(to)
    assert len(tp) == len(to) == 7 and to[-1]["count"] == 3                       # stopped early by --count
    for a, b in zip(tp, to):
        assert (a["epoch"], a["saved"], a["count"], a["val_loss"]) == (b["epoch"], b["saved"], b["count"], b["val_loss"])
        assert a["lr"] == b["lr"], (a["epoch"], a["lr"], b["lr"])
    for a, b in zip(mp.parameters(), mo.parameters()):
        assert torch.equal(a, b)
    ck_p = torch.load(os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin"))
    ck_o = torch.load(os.path.join(out_o, "simplebaseline/ours/84k", "checkpoint-good", "state_dict.bin"))
    assert sorted(ck_p) == sorted(ck_o) and (ck_p["epoch"], ck_p["count"], ck_p["best_loss"]) == (ck_o["epoch"], ck_o["count"], ck_o["best_loss"]) == (3, 0, 0.7)
    # resume (a new process in the reference: fresh model object, state from the checkpoint)
    mp2, mo2 = copy.deepcopy(m0), copy.deepcopy(m0)
    tp2, _ = run_product(mp2, train_set, val_set, out_p, 12)
    to2 = O.main(mo2, train_set, val_set, os.path.join(out_o, "simplebaseline/ours/84k"), 8, 12, 0.01, 3, train_batch_ref, lambda e: val_seq[e])
    assert [e["epoch"] for e in tp2] == [e["epoch"] for e in to2] and tp2[0]["epoch"] == 4
    assert [e["lr"] for e in tp2] == [e["lr"] for e in to2]
    assert [(e["saved"], e["count"]) for e in tp2] == [(e["saved"], e["count"]) for e in to2]
    for a, b in zip(mp2.parameters(), mo2.parameters()):
        assert torch.equal(a, b)
    # the reference's resumed schedule is NOT the continuation of the first run's: the opt-in flag gives that
    import math
    mp3 = copy.deepcopy(m0)
    saved_ck = torch.load(os.path.join(args.output_dir, "checkpoint-good", "state_dict.bin"))
    tp3, _ = run_product(mp3, train_set, val_set, out_p, 12, extra=("--lr_resume_fix",))
    e0 = saved_ck["epoch"] + 1
    assert tp3[0]["epoch"] == e0 and abs(tp3[0]["lr"] - 0.01 * (1 + math.cos(math.pi * e0 / 12)) / 2) < 1e-12
    # --optim: the saved optimizer state is ignored (train.py:50): a fresh Adam at args.lr
    tp4, _ = run_product(copy.deepcopy(m0), train_set, val_set, out_p, 12, extra=("--optim",))
    to4 = O.main(copy.deepcopy(m0), train_set, val_set, os.path.join(out_o, "simplebaseline/ours/84k"), 8, 12, 0.01, 3, train_batch_ref,
                 lambda e: val_seq[e], optim=True)
    assert [e["lr"] for e in tp4] == [e["lr"] for e in to4] and tp4[0]["lr"] == 0.01
    # --transfer: weights from the transfer checkpoint, counters from the resume point; a missing file is an error as in the reference
    donor = copy.deepcopy(m0)
    with torch.no_grad():
        for p in donor.parameters():
            p.fill_(0.25)
    tdir = tmp_path / "donor" / "checkpoint-good"
    tdir.mkdir(parents=True)
    torch.save({"epoch": 99, "optimizer_state_dict": {}, "best_loss": 0.0, "count": 7, "model_state_dict": donor.state_dict()}, tdir / "state_dict.bin")
    a5 = T.parse_args(["--root_path", out_p, "--transfer", "--transfer_from", str(tdir / "state_dict.bin")])
    m5 = copy.deepcopy(m0)
    best, epo, count, opt_state = T.load_model_state(m5, a5)
    m6 = copy.deepcopy(m0)
    assert (best, epo, count) == O.load_model(m6, os.path.join(out_o, "simplebaseline/ours/84k"), False, True, str(tdir / "state_dict.bin"))[:3]
    assert all(bool((p == 0.25).all()) for p in m5.parameters()) and all(torch.equal(a, b) for a, b in zip(m5.parameters(), m6.parameters()))
    assert opt_state and epo > 0                                  # counters and optimizer state are the RESUME point's, not the donor's (99 / 7)
    a7 = T.parse_args(["--root_path", out_p, "--transfer"])
    assert T.transfer_path(a7) == "output/simplebaseline/frei/ori/checkpoint-good/state_dict.bin"      # argparser.py:171-173
    with pytest.raises(FileNotFoundError):
        T.load_model_state(copy.deepcopy(m0), a7)
