"""Multi-rank paths on the GPU box (one MI355X: both ranks on cuda:0, gloo process group -- RCCL wants one GPU per rank, so
the 2-rank cases run over gloo and `test_single_rank_rccl_drives_every_exchange` sends the same exchanges through RCCL with one rank): BASELINE configs[3] (slides sharded over the ranks, one ragged
all-gather), the product CLI's --world_size launcher, and the classifier step's global weighted cross-entropy."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from ss25_hierarchical_multiscale_image_classification_amd import capi, extract, launch, synth

pytestmark = pytest.mark.gpu
HELPERS = os.path.join(os.path.dirname(__file__), "helpers")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(helper, args, n=2, timeout=900):
    cmds = launch.child_commands([os.path.join(HELPERS, helper)], args, n, launch.free_port())
    procs = [subprocess.Popen(cmd[:-1], env=env) for cmd, env in cmds]  # (the helpers take no --_child flag)
    assert [p.wait(timeout=timeout) for p in procs] == [0] * n


@pytest.mark.parametrize("precision", ["fp16q8", "fp16x3", "bf16"])
def test_two_rank_sharded_slides_equal_single_process(tmp_path, precision):
    """configs[3] in miniature: 3 synthetic slides, slide i -> rank i mod 2 (`dist.shard_units`), `extract.score_slide`
    per slide, `dist.gather_results`.  The rank-major result is the single-process result of slides (0, 2, 1) bit for
    bit -- features, logits and meta -- on both ranks; `dist.score_sharded` (what `main.py --world_size` calls) returns
    the rows in slide order 0, 1, 2 = exactly the single-process output."""
    specs = [[1500, 1300, 31], [1250, 1400, 32], [1600, 1100, 33]]
    torch.save(torch.tensor(specs), tmp_path / "specs.pt")
    _run_ranks("wsi_rank.py", [str(tmp_path), precision])
    net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision=precision)
    single = []
    for i, (w, h, seed) in enumerate(specs):
        slide = extract.DeviceSlide.synthetic(w, h, seed=seed, name=f"slide_{i}", with_polygons=True)
        f, l, _, meta = extract.score_slide(slide, net, levels=(1, 2, 3), fwd_batch=64)
        single.append((f.cpu(), l.cpu(), meta.cpu()))
    assert all(s[0].shape[0] > 0 for s in single)
    res = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(2)]
    for r in range(2):
        gf, gl, gm = res[r]["rank_major"]
        order = (0, 2, 1)  # rank 0 owns slides 0 and 2, rank 1 slide 1
        assert torch.equal(gf, torch.cat([single[i][0] for i in order]))
        assert torch.equal(gl, torch.cat([single[i][1] for i in order]))
        assert torch.equal(gm, torch.cat([single[i][2] for i in order])), (gm.tolist(), [single[i][2].tolist() for i in order])
        sf, sl, sm = res[r]["slide_order"]
        assert torch.equal(sf, torch.cat([s[0] for s in single])) and torch.equal(sl, torch.cat([s[1] for s in single]))
        assert torch.equal(sm[:, :4], torch.cat([s[2] for s in single])), (r, sm.tolist(), gm.tolist())
        assert sm[:, 4].tolist() == sum(([i] * single[i][0].shape[0] for i in range(3)), [])


def test_cli_world_size_writes_the_single_process_files(tmp_path):
    """`main.py --extract_features --world_size 2` (the launcher starts two fresh ranks; here both on cuda:0 over gloo):
    the three files rank 0 writes are byte-identical to those of the single-process run."""
    outs = []
    for world in (1, 2):
        d = tmp_path / f"w{world}"
        d.mkdir()
        cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "--extract_features", "--patch_level", "1", "--precision",
               "fp16", "--data_root", str(d / "none"), "--synthetic", "2600,2300,41,tumor_041", "--synthetic",
               "2300,2500,42,normal_042", "--synthetic", "2400,2400,43,tumor_043"]
        if world > 1:
            cmd += ["--world_size", "2", "--dist_backend", "gloo", "--one_device", "--rank_timeout", "600"]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(d)
    f1, f2 = np.load(outs[0] / "patch_features_1.npy"), np.load(outs[1] / "patch_features_1.npy")
    assert f1.shape[0] > 10 and np.array_equal(f1, f2)
    assert np.array_equal(np.load(outs[0] / "patch_labels_1.npy"), np.load(outs[1] / "patch_labels_1.npy"))
    assert (outs[0] / "patch_paths_1.txt").read_text() == (outs[1] / "patch_paths_1.txt").read_text()


def test_cli_world_size_reports_a_failing_rank(tmp_path):
    """A rank that dies takes the run down at once with its exit code (no hang in the rendezvous)."""
    cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "--extract_features", "--weights", str(tmp_path / "missing.pth"),
           "--synthetic", "1300,1300,5", "--world_size", "2", "--dist_backend", "gloo", "--one_device"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "exited with code" in r.stderr


def test_cli_train_world_size_two(tmp_path):
    """`main.py --train_strategy --strategy weighted_loss --world_size 2`: the launcher starts two ranks, each takes its
    share of every global batch (per-replica batch-norm, CE over the gathered logits, gradients all-reduced), rank 0
    prints the epoch line once and writes the checkpoint in the reference's key layout (src/main.py:605)."""
    from PIL import Image

    from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier

    rng = np.random.RandomState(0)
    root = tmp_path / "data" / "patches" / "level_3"
    for sl in range(4):
        d = root / f"tumor_{sl:03d}"
        d.mkdir(parents=True)
        for k in range(6):
            lab = "tumor" if (k + sl) % 2 == 0 else "normal"
            Image.fromarray(rng.randint(0, 256, (224, 224, 3), dtype=np.uint8), "RGB").save(d / f"tumor_{sl:03d}_x{224 * k}_y0_{lab}.png")
    cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "--train_strategy", "--strategy", "weighted_loss", "--patch_level", "3",
           "--data_root", str(tmp_path / "data"), "--epochs", "1", "--batch_size", "4", "--max_steps", "2", "--precision", "fp16",
           "--world_size", "2", "--dist_backend", "gloo", "--one_device", "--rank_timeout", "600"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("Epoch 1, Train Loss") == 1  # rank 0 only
    out = torch.load(tmp_path / "src" / "models" / "resnet18_patch_classifier_weighted_loss.pth", map_location="cpu", weights_only=True)
    assert set(out) == set(ResNet18Classifier().state_dict()) and all(bool(torch.isfinite(v.float()).all()) for v in out.values())


def test_cli_train_from_slides_world_size_two(tmp_path):
    """The same loop with the device input pipeline and no PNG tree, two ranks: `--train_strategy --from_slides --world_size 2`
    (every rank holds the slides and their pool in HBM, takes its share of every global batch made on the device)."""
    from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier

    cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "--train_strategy", "--strategy", "weighted_loss", "--patch_level", "3",
           "--data_root", str(tmp_path / "none"), "--from_slides", "--epochs", "1", "--batch_size", "8", "--max_steps", "2",
           "--precision", "fp16", "--world_size", "2", "--dist_backend", "gloo", "--one_device", "--rank_timeout", "600"]
    for k in range(3):
        cmd += ["--synthetic", f"{4000 + 300 * k},3600,{81 + k},tumor_08{k}"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("Epoch 1, Train Loss") == 1
    out = torch.load(tmp_path / "src" / "models" / "resnet18_patch_classifier_weighted_loss.pth", map_location="cpu", weights_only=True)
    assert set(out) == set(ResNet18Classifier().state_dict()) and all(bool(torch.isfinite(v.float()).all()) for v in out.values())


def test_two_rank_classifier_step(tmp_path):
    """Two ranks, 4 images each, class weights (1, 2.5) and DIFFERENT class mixes per rank: the loss both ranks report
    and the gradients after the SUM all-reduce are those of CrossEntropyLoss(weight) over the gathered logits -- sum_i
    w_i nll_i / sum_i w_i over ALL ranks, as the reference evaluates it under nn.DataParallel (src/main.py:499-506)."""
    import torch.nn.functional as F

    from oracle import train_ref as TR
    from oracle.resnet18_ref import canonical_state_dict
    from ss25_hierarchical_multiscale_image_classification_amd.resnet import ResNet18Classifier

    torch.manual_seed(29)
    model = ResNet18Classifier()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randn(2, 4, 3, 224, 224)
    y = torch.tensor([[0, 0, 0, 1], [1, 1, 0, 1]])
    w = torch.tensor([1.0, 2.5])
    torch.save(sd, tmp_path / "sd.pt")
    torch.save({"x": x, "y": y, "w": w}, tmp_path / "xy.pt")
    _run_ranks("classifier_rank.py", [str(tmp_path)])
    res = [torch.load(tmp_path / f"rank{r}.pt", weights_only=True) for r in range(2)]
    assert res[0]["loss"] == res[1]["loss"]
    for k, g in res[0]["grads"].items():
        assert torch.equal(g, res[1]["grads"][k]), k
    # oracle: each replica's images through the encoder separately (per-replica batch statistics), ONE loss
    bare = canonical_state_dict(sd)
    p, _ = TR._split(bare)
    logits = []
    for r in range(2):
        _, st = TR._split(bare)
        logits.append(F.linear(TR.encoder_train_forward(x[r], p, st), p["fc.weight"], p["fc.bias"]))
    loss = F.cross_entropy(torch.cat(logits), y.reshape(-1), weight=w)
    loss.backward()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert abs(res[0]["loss"] - float(loss)) <= 1e-5 * abs(float(loss)) + 1e-6
    for r in range(2):
        assert rel(res[r]["logits"], logits[r].detach()) <= 1e-4
    # the gradients nearest to the loss carry the normalisation (deep layers flip ReLU patterns, see test_gpu_train.py)
    for k in ("fc.weight", "fc.bias"):
        assert rel(res[0]["grads"][k], p[k].grad) <= 5e-3, (k, rel(res[0]["grads"][k], p[k].grad))
    # and it is NOT the per-rank-normalised sum the round-2 code computed (about world x larger)
    local = sum(F.cross_entropy(l.detach(), y[r], weight=w) for r, l in enumerate(logits))
    assert abs(float(local) - float(loss)) > 1e-3


def test_single_rank_rccl_drives_every_exchange(tmp_path):
    """RCCL on the hardware this box has: a one-rank "nccl" process group with HIPAC_DIST_FORCE=1, so that the slide-sharded
    scan (count + padded all-gathers), the SimCLR step (all-gather of z, all-reduce of the gradients, MAX all-reduce of the
    loss-scale flag), the rank-0 broadcast and the scalar all-reduce really call RCCL on device tensors.  With one rank every
    collective is the identity, so the results must equal a process that has no process group -- bit for bit."""
    specs = [[1500, 1300, 31], [1250, 1400, 32]]
    torch.save(torch.tensor(specs), tmp_path / "specs.pt")
    from ss25_hierarchical_multiscale_image_classification_amd import train_native as TN

    from ss25_hierarchical_multiscale_image_classification_amd.simclr import SimCLRModel

    torch.manual_seed(5)
    g = torch.Generator().manual_seed(5)
    sd = {k: v.clone() for k, v in SimCLRModel().state_dict().items()}
    x = torch.rand(2, 6, 3, 224, 224, generator=g)
    torch.save(sd, tmp_path / "sd.pt")
    torch.save(x, tmp_path / "x.pt")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(HELPERS, "rccl_one_rank.py"), str(tmp_path), str(launch.free_port())],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(tmp_path / "rccl.pt", weights_only=True)

    net = capi.PackedResNet18(synth.seeded_resnet18_state_dict(0, num_classes=2), precision="fp16x3")
    fs, ls, ms = [], [], []
    for i, (w, h, seed) in enumerate(specs):
        slide = extract.DeviceSlide.synthetic(w, h, seed=seed, name=f"slide_{i}", with_polygons=True)
        f, l, _, meta = extract.score_slide(slide, net, levels=(1, 2, 3), fwd_batch=64)
        fs.append(f.cpu()), ls.append(l.cpu()), ms.append(meta.cpu())
    sf, sl, sm = got["slides"]
    assert sf.shape[0] > 0 and torch.equal(sf, torch.cat(fs)) and torch.equal(sl, torch.cat(ls))
    assert torch.equal(sm[:, :4], torch.cat(ms)) and sm[:, 4].tolist() == sum(([i] * fs[i].shape[0] for i in range(2)), [])
    assert torch.equal(got["eq"], sf[:7]) and got["scal"] == [1.5, 2.0, -3.25]
    tr = TN.NativeSimCLRTrainer(sd, device="cuda:0", precision="fp16")
    loss = tr.step(x[0].cuda().contiguous(), x[1].cuda().contiguous())
    assert float(loss) == got["simclr"]["fp16"]["loss"]
    mine = tr.state_dict()  # the mixed-precision step is deterministic (two-stage reductions, no atomics): bit for bit
    assert set(mine) == set(got["simclr"]["fp16"]["params"])
    for k, v in mine.items():
        assert torch.equal(v.cpu(), got["simclr"]["fp16"]["params"][k]), k
    tr = TN.NativeSimCLRTrainer(sd, device="cuda:0", precision="fp32")
    loss = tr.forward_backward(x[0].cuda().contiguous(), x[1].cuda().contiguous())
    assert abs(float(loss) - got["simclr"]["fp32"]["loss"]) <= 1e-6 * abs(float(loss))
    for k, v in tr.grad_dict().items():  # the fp32 step is atomics-free and deterministic since round 3; the bound stays a rounding bound (the helper's process ran forward_backward once more before this one)
        ref = got["simclr"]["fp32"]["grads"][k]
        assert float((v.cpu() - ref).norm()) <= 1e-5 * float(ref.norm()) + 1e-12, k


def test_bench_two_ranks_default_line(tmp_path):
    """`bench.py --gpus 2` as the driver starts it (here: both ranks on cuda:0 over gloo): the parent launches two fresh
    ranks, each runs the default workload on its own patches / slide / views, rank 0 prints ONE JSON line whose value is the
    aggregate over both ranks and whose `wsi` and `simclr` objects come from the sharded runs."""
    import json

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one_device", "--steps", "2",
           "--warmup", "1", "--batch", "1024", "--no_cpu_baseline", "--wsi_sides", "3000", "--rank_timeout", "800"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["value"] > 0
    assert abs(rec["value"] - 2 * 1024 * 2 / (rec["ms_per_step"] * 2e-3)) <= 1e-6 * rec["value"]  # whole-job aggregate
    assert "cpu_baseline" not in rec or rec["cpu_baseline"] is None
    wsi = rec["wsi"]["3000x3000"]
    assert "error" not in wsi and wsi["slides"] == 2 and wsi["n_gpus"] == 2 and wsi["s_per_slide"] > 0
    # N > 1: the load balance of one-slide-per-rank is in the line (every rank's kept windows and own time)
    assert len(wsi["kept_per_rank"]) == 2 and min(wsi["kept_per_rank"]) > 0 and wsi["kept_all_ranks"] == sum(wsi["kept_per_rank"])
    assert 0 < wsi["rank_s_min"] <= wsi["rank_s_max"] <= wsi["s_per_slide"] * 1.5 and len(wsi["rank_s_per_slide"]) == 2
    assert all("(fused)" not in o["op"] for o in rec["per_op"])
    assert rec["simclr"]["n_gpus"] == 2 and rec["simclr"]["value"] > 0 and rec["simclr"]["final_loss"] > 0


def _write_npz_slide(path, w, h, seed):
    levels = synth.build_pyramid(synth.synth_level0(w, h, seed=seed, n_blobs=3), 4)
    np.savez(path, **{f"level{i}": l.numpy() for i, l in enumerate(levels)})


@pytest.mark.parametrize("world", [1, 2])
def test_cli_unreadable_slide_costs_that_slide_only(tmp_path, world):
    """The reference opens every slide under try / except ... continue and tolerates a bad annotation file
    (src/main.py:649-653, :670-675).  Three slide files, one of them a truncated .tif, plus a malformed XML: the run ends
    with exit code 0, names the file it could not open, and writes exactly the files of the run without the bad slide --
    with --world_size 2 the rank that owns the bad slide contributes zero rows for it and the exchange goes on."""
    outs = []
    for bad in (False, True):
        d = tmp_path / f"bad{int(bad)}"
        img = d / "data" / "train" / "img"
        ann = d / "data" / "train" / "mask" / "annotations"
        img.mkdir(parents=True), ann.mkdir(parents=True)
        _write_npz_slide(img / "tumor_001.npz", 2500, 2300, 51)
        _write_npz_slide(img / "tumor_003.npz", 2300, 2600, 53)
        if bad:
            (img / "tumor_002.tif").write_bytes(b"II*\x00" + os.urandom(2000))  # a TIFF header and nothing that follows it
            (ann / "tumor_003.xml").write_text("<ASAP_Annotations><Annotations><Annotation")  # cut off mid-tag
        cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "--extract_features", "--patch_level", "1", "--precision", "fp16",
               "--data_root", str(d / "data")]
        if world > 1:
            cmd += ["--world_size", "2", "--dist_backend", "gloo", "--one_device", "--rank_timeout", "600"]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        if bad:
            # (--world_size 2: slide 1 of 3 belongs to rank 1, whose stdout the launcher does not relay: it reports on stderr)
            assert "ould not open tumor_002" in r.stdout + r.stderr and "[WARNING] Failed to parse XML for tumor_003" in r.stdout
        outs.append(d)
    f0, f1 = np.load(outs[0] / "patch_features_1.npy"), np.load(outs[1] / "patch_features_1.npy")
    assert f0.shape[0] > 10 and np.array_equal(f0, f1)
    assert np.array_equal(np.load(outs[0] / "patch_labels_1.npy"), np.load(outs[1] / "patch_labels_1.npy"))
    assert (outs[0] / "patch_paths_1.txt").read_text() == (outs[1] / "patch_paths_1.txt").read_text()


def test_cli_patch_then_train_world_size_two_waits_for_every_rank(tmp_path):
    """`--patch --write_png --train_strategy --world_size 2` in ONE invocation, slides of unequal size: the training command
    lists the PNG tree the patch command wrote, so no rank may enter it before every rank has finished writing (a
    rendezvous between the commands); with it both ranks build the same dataset and the collectives match."""
    d = tmp_path / "data" / "train" / "img"
    d.mkdir(parents=True)
    _write_npz_slide(d / "tumor_011.npz", 7200, 5400, 61)   # rank 0's slide: ~4x the windows of rank 1's
    _write_npz_slide(d / "tumor_012.npz", 3600, 2700, 62)
    cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "--patch", "--write_png", "--train_strategy", "--strategy", "weighted_loss",
           "--patch_level", "3", "--data_root", str(tmp_path / "data"), "--epochs", "1", "--batch_size", "4", "--max_steps", "1",
           "--precision", "fp16", "--world_size", "2", "--dist_backend", "gloo", "--one_device", "--rank_timeout", "600"]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("Epoch 1, Train Loss") == 1
    assert (tmp_path / "src" / "models" / "resnet18_patch_classifier_weighted_loss.pth").exists()
