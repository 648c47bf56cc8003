"""Data-parallel step on the GPU box: 2 ranks (gloo over ONE GPU, since the test box has a single MI355X) against the
single-process result at equal global batch.  Exercises the real hook path: per-block gradient spans fired during the
HIP backward, side-stream all-reduce, parameter broadcast, centre all-reduce.  (RCCL itself is exercised by
`bench.py --gpus N` on the multi-GPU node.)"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["CHADAVIT_ROOT"])
import torch.distributed as dist
from chadavit_amd.parallel import GradSync, init_from_env
from chadavit_amd.methods.dino import DINO
from chadavit_amd.trainer import Trainer
from chadavit_amd.data.channels_strategies import one_channel_collate_fn
from oracle import procedural as P
from tests.golden_util import build_sd
from tests.test_model_gpu import _cfg
rank, world, local = init_from_env()
dev = torch.device("cuda", local)
nch_all = [3, 1, 2, 1]
per = len(nch_all) // world
mine = list(range(rank * per, (rank + 1) * per))
imgs_all = P.make_images(nch_all, [224, 224, 96], seed=11)
imgs = [imgs_all[i] for i in mine]
crops, labels, ncl = one_channel_collate_fn(imgs)
batch = ([c.to(dev) for c in crops], labels.to(dev), ncl)
cfg = _cfg(192, 4096, 2, 1)
if os.environ.get("CHADAVIT_TEST_STANDARD_MULTICROP"):   # the flagged standard-DINO option: two backward passes per network and step
    cfg.method_kwargs.standard_multicrop_loss = True
model = DINO(cfg)
sd = build_sd(192, 4096)
if rank != 0:   # non-zero ranks start from different weights: the broadcast must fix that
    sd = {k: v + 0.01 for k, v in sd.items()}
model.load_state_dict(sd)
model = model.to(dev)
tr = Trainer(max_epochs=10, steps_per_epoch=10, grad_sync=GradSync() if world > 1 else None).attach(model)
tr.current_epoch = 1
model.current_epoch = 1
model.on_train_epoch_start()
loss = model.training_step(batch, 1)
if tr.grad_sync is not None: tr.grad_sync.begin_backward()
loss.backward()
if tr.grad_sync is not None: tr.grad_sync.finish()
pending_after_step = model.dino_loss_func._pending is not None
model.dino_loss_func.sync_center()   # the centre's all-reduce was only started inside the step (losses/dino.py)
torch.cuda.synchronize()
named = dict(model.named_parameters())
out = {"loss": loss.item(),
       "gnorm": {n: named[n].grad.double().norm().item() for n in ("backbone.blocks.0.linear1.weight", "backbone.blocks.11.self_attn.in_proj_weight",
                                                                    "backbone.pos_embed", "backbone.norm.weight", "head.mlp.2.weight", "head.last_layer.weight_v")},
       "center": model.dino_loss_func.center.double().sum().item(),
       "center_pending_after_step": pending_after_step,
       "g0": named["backbone.norm.weight"].grad[:8].tolist()}
if rank == 0:
    print("RESULT " + json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _run(world, worker=None, extra_env=None):
    worker = WORKER if worker is None else worker
    env = dict(os.environ, CHADAVIT_ROOT=ROOT, PYTHONPATH=ROOT, **(extra_env or {}))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    if world == 1:
        cmd = [sys.executable, "-c", worker]
    else:
        env.update(CHADAVIT_DIST_BACKEND="gloo", CHADAVIT_SINGLE_DEVICE="1")
        import tempfile
        tmpdir = tempfile.mkdtemp(prefix="chadavit_ddp_")  # scratch, never the repo tree
        path = os.path.join(tmpdir, "_ddp_worker.py")
        with open(path, "w") as f:
            f.write(worker)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), path]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    import json
    return json.loads(lines[-1][7:])


@pytest.mark.timeout(1500)
def test_two_ranks_match_single_process():
    """World 2 (2 images per rank) vs world 1 (all 4 images): the averaged per-rank gradients equal the gradients of the
    global-batch mean loss (losses/dino.py: each rank's loss is its local mean; DDP averages gradients)."""
    one = _run(1)
    two = _run(2)
    # rank 0's local loss differs from the global one (different images); gradients and centre must agree
    for n, v in one["gnorm"].items():
        assert abs(two["gnorm"][n] - v) <= 3e-2 * v + 1e-7, (n, v, two["gnorm"][n])
    assert abs(one["center"] - two["center"]) <= 1e-3 * abs(one["center"]) + 1e-4
    # ... although with two ranks its all-reduce was still in flight when the step returned (finished by sync_center)
    assert two["center_pending_after_step"] and not one["center_pending_after_step"]
    for a, b in zip(one["g0"], two["g0"]):
        assert abs(a - b) <= 3e-2 * max(abs(a), abs(b)) + 5e-4


@pytest.mark.timeout(1500)
def test_two_ranks_match_single_process_with_the_standard_multicrop_loss():
    """The flagged standard-DINO option under data parallelism: every network runs TWO backward passes per step (global-crop pass,
    local-crop pass) that accumulate into one gradient slab; the per-block gradient spans may go to the reducer only during the LAST of
    them (`_pending_backwards`), or half-finished sums would be averaged.  World 2 vs world 1 on the same global batch, as above."""
    env = {"CHADAVIT_TEST_STANDARD_MULTICROP": "1"}
    one = _run(1, extra_env=env)
    two = _run(2, extra_env=env)
    for n, v in one["gnorm"].items():
        assert abs(two["gnorm"][n] - v) <= 3e-2 * v + 1e-7, (n, v, two["gnorm"][n])
    assert abs(one["center"] - two["center"]) <= 1e-3 * abs(one["center"]) + 1e-4
    for a, b in zip(one["g0"], two["g0"]):
        assert abs(a - b) <= 3e-2 * max(abs(a), abs(b)) + 5e-4


LINEAR_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["CHADAVIT_ROOT"])
import torch.distributed as dist
from chadavit_amd.parallel import GradSync, init_from_env
from chadavit_amd.backbones import vit_channels
from chadavit_amd.methods.linear import LinearModel
from chadavit_amd.trainer import Trainer
from chadavit_amd.data.channels_strategies import one_channel_collate_fn
from oracle import procedural as P
from tests.test_linear_gpu import _cfg
rank, world, local = init_from_env()
dev = torch.device("cuda", local)
nch_all = [3, 1, 2, 4]
per = len(nch_all) // world
imgs_all = P.make_images(nch_all, [224], seed=13)
x, labels, ncl = one_channel_collate_fn([imgs_all[i] for i in range(rank * per, (rank + 1) * per)])
bb = vit_channels("dino", patch_size=16, embed_dim=192, return_all_tokens=False, max_number_channels=10)
sdb = P.fill_state_dict(P.backbone_shapes(192), seed=1)
cl = P.fill_state_dict({"weight": (7, 192), "bias": (7,)}, seed=21)
if rank != 0:   # the broadcast at attach() must repair this
    sdb = {k: v + 0.01 for k, v in sdb.items()}
    cl = {k: v - 0.02 for k, v in cl.items()}
bb.load_state_dict(sdb)
m = LinearModel(bb, _cfg(192, False, 3, True, 7, True, 1e-3, 0.0, kwargs={"momentum": 0.9}))
m.classifier.load_state_dict(cl)
m = m.to(dev)
tr = Trainer(max_epochs=4, steps_per_epoch=4, grad_sync=GradSync() if world > 1 else None).attach(m)
m.train()
named = dict(m.named_parameters())
start = {"w": named["classifier.weight"].detach().double().sum().item(), "nw": named["backbone.norm.weight"].detach().double().sum().item()}
loss = m.training_step((x.to(dev), labels.to(dev), ncl), 0)
if tr.grad_sync is not None: tr.grad_sync.begin_backward()
loss.backward()
local_dw = named["classifier.weight"].grad.detach().clone()
if tr.grad_sync is not None: tr.grad_sync.finish()
torch.cuda.synchronize()
g = lambda n: named[n].grad.detach()
out = {"start": start, "dw": g("classifier.weight").flatten()[:16].cpu().tolist(), "dw_norm": g("classifier.weight").double().norm().item(),
       "db": g("classifier.bias").cpu().tolist(), "nw": g("backbone.norm.weight")[:8].cpu().tolist(),
       "gnorm": {n: g(n).double().norm().item() for n in ("backbone.blocks.0.linear1.weight", "backbone.blocks.11.self_attn.in_proj_weight",
                                                           "backbone.pos_embed", "backbone.norm.weight", "backbone.token_learner.proj.weight")},
       "moved_dw": (g("classifier.weight") - local_dw).double().norm().item() / local_dw.double().norm().item()}
tr.optimizer.step()    # (fused SGD over the averaged gradients: must run on every rank without error)
torch.cuda.synchronize()
if rank == 0:
    print("RESULT " + json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.timeout(1500)
def test_linear_model_two_ranks_match_single_process():
    """LinearModel fine-tuning under GradSync: world 2 (two images per rank, ranks starting from different weights) takes the same
    optimiser step as world 1 on all four images -- the backbone's spans and the classifier's gradient tensors are averaged, the
    parameters broadcast at attach()."""
    one = _run(1, LINEAR_WORKER)
    two = _run(2, LINEAR_WORKER)
    assert two["start"] == one["start"]                      # rank 0's weights (rank 0 reports; the other rank started elsewhere)
    assert one["moved_dw"] == 0.0
    # the exchange changed rank 0's local classifier gradient (other images; the backbone's spans are exchanged INSIDE backward) ...
    assert two["moved_dw"] > 0.05, two["moved_dw"]
    for n, v in one["gnorm"].items():                                  # ... into the gradients of the four-image mean loss
        assert abs(two["gnorm"][n] - v) <= 3e-2 * v + 1e-7, (n, v, two["gnorm"][n])
    assert abs(one["dw_norm"] - two["dw_norm"]) <= 2e-2 * one["dw_norm"]
    for a, b in zip(one["dw"] + one["db"] + one["nw"], two["dw"] + two["db"] + two["nw"]):
        assert abs(a - b) <= 3e-2 * max(abs(a), abs(b)) + 5e-4, (a, b)


def test_bench_contract_with_two_ranks():
    """`bench.py --gpus 2` launched exactly as the driver does (torch.distributed.run, one JSON line from rank 0), here with
    two gloo ranks sharing the box's single GPU.  Regression test for the roofline leg: it runs extra training steps, which
    contain gradient collectives, so every rank has to execute them (rank 0 alone used to, which hangs any N > 1 run)."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CHADAVIT_DIST_BACKEND="gloo", CHADAVIT_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"] is not None and out["roofline"]["avg_us_in_step"] > 0 and "cpu_baseline" not in out
    # the collectives of a step are reported (14 gradient spans: 12 blocks + final norm + tokenizer, + the head) ...
    assert out["rccl"]["world"] == 2 and out["rccl"]["spans"] >= 14 and out["rccl"]["bytes_per_step"] > 4 * 17e6
    # ... and the N-rank step equals the single-rank step on the same global batch (self-check of the first multi-GPU run)
    # (on by default when N > 1: no flag on the command line above)
    v = out["verify_equal_batch"]
    assert v["ok"] and v["rel_loss"] <= 3e-2 and v["rel_gradnorm"] <= 3e-2, v
    assert v["sharded_step_repeats_bit_for_bit_on_every_rank"], v   # determinism with the collectives' traffic beside the kernels
    # how much of the gradient exchange was NOT hidden behind the backward, per rank, and each rank's own step time
    r = out["rccl"]
    assert len(r["exposed_ms_per_step_per_rank"]) == 2 and r["exposed_ms_per_step"] >= 0 and 0 <= r["exposed_fraction_of_step"] < 1
    assert all(b > 0 for b in r["comm_busy_ms_per_step_per_rank"])
    assert len(out["step_ms_per_rank"]) == 2 and out["step_ms_rank_spread"] >= 0
    assert out["config"]["logged_loss_mean_over_ranks"] is not None and "other_workloads" not in out["config"]


def test_bench_with_a_dead_rank_fails_fast_instead_of_hanging():
    """one of two ranks dies (SIGKILL, injected in front of its first timed step) while the other is inside the
    step's collectives.  The launch, started exactly as the driver starts it, must come back NON-ZERO well inside the timeout and print
    no result line -- not sit in a collective until somebody kills it."""
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CHADAVIT_DIST_BACKEND="gloo", CHADAVIT_SINGLE_DEVICE="1", CHADAVIT_BENCH_FAULT="1:0", CHADAVIT_DIST_TIMEOUT_S="120")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "2", "--warmup", "1",
           "--no-verify-equal-batch", "--no-launch-profile", "--no-full-width-leg"]
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode != 0, r.stdout[-1000:]
    assert time.time() - t0 < 400
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-1000:]


DATA_WORKER = r'''
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.environ["CHADAVIT_ROOT"])
import torch.distributed as dist
from chadavit_amd.parallel import GradSync, init_from_env
from chadavit_amd.methods.dino import DINO
from chadavit_amd.trainer import Trainer
from chadavit_amd.data.idrcell import IDRCell100K
from chadavit_amd.data.sampler import TokenBalancedBatchSampler
from chadavit_amd.data.device_pipeline import CropSpec, DeviceMultiCropPipeline
from chadavit_amd.data.loader import DevicePrefetcher
from tests.golden_util import build_sd
from tests.test_model_gpu import _cfg
rank, world, local = init_from_env()
dev = torch.device("cuda", local)
ds = IDRCell100K(root_dir=os.environ["CHADAVIT_DATA"], train=True)
sampler = TokenBalancedBatchSampler(ds.num_channels(), global_batch=8, rank=rank, world=world, patches_per_channel=196, seed=5)
specs = [CropSpec(crop_size=224, num_crops=2, crop_min_scale=0.25, crop_max_scale=1.0, jitter_prob=0.8, blur_prob=0.5, flip_prob=0.5),
         CropSpec(crop_size=96, num_crops=2, crop_min_scale=0.05, crop_max_scale=0.25, jitter_prob=0.8, flip_prob=0.5)]
pipe = DeviceMultiCropPipeline(specs, dev, seed=100 + rank)
loader = DevicePrefetcher(ds, sampler, pipe, depth=2, workers=4)
model = DINO(_cfg(192, 4096, 2, 2))
model.load_state_dict(build_sd(192, 4096))
model = model.to(dev)
tr = Trainer(max_epochs=4, steps_per_epoch=2, grad_sync=GradSync() if world > 1 else None).attach(model)
losses, seen = [], []
it = iter(sampler)
for step, batch in enumerate(loader):
    seen.append(next(it))
    crops, labels, ncl = batch
    assert len(crops) == 4 and crops[0].shape[1:] == (1, 224, 224) and crops[2].shape[1:] == (1, 96, 96)
    assert ncl[0] == [ds.num_channels()[i] for i in seen[-1]] and labels.tolist() == [-1] * len(seen[-1])
    losses.append(tr.train_step(batch, step).item())
model.dino_loss_func.sync_center()
torch.cuda.synchronize()
sd = model.state_dict()
out = {"rank": rank, "losses": losses, "seen": seen, "steps": len(losses),
       "w": sd["backbone.blocks.3.linear1.weight"].double().sum().item(), "wt": sd["momentum_backbone.blocks.3.linear1.weight"].double().sum().item(),
       "center": sd["dino_loss_func.center"].double().sum().item(), "logged": model.logged_metrics().get("dino_loss_train")}
with open(os.path.join(os.environ["CHADAVIT_OUT"], f"rank{rank}.json"), "w") as f:   # (stdout of the two ranks can interleave)
    json.dump(out, f)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.timeout(900)
def test_disk_to_training_step_on_two_ranks(tmp_path):
    """The real-data shaped path end to end: an on-disk IDRCell100k-format set (csv of per-channel files) ->
    TokenBalancedBatchSampler -> reader threads -> DevicePrefetcher (H2D + crop / jitter / blur kernels on a side stream) ->
    two DINO training steps on 2 gloo ranks.  Every image of a global batch is used by exactly one rank, the replicas stay
    identical (same weights, EMA teacher and centre on both ranks) and the logged loss read back is the mean over ranks."""
    import json
    import tempfile
    import numpy as np
    from PIL import Image
    root = tmp_path
    os.makedirs(root / "images" / "plate")
    rs = np.random.RandomState(0)
    with open(root / "train.csv", "w") as f:
        for i in range(16):
            c = [1, 2, 3, 5][i % 4]
            paths = []
            for ch in range(c):
                rel = f"plate/img{i}_ch{ch}.png"
                Image.fromarray(rs.randint(0, 255, size=(120 + i, 130), dtype=np.uint8)).save(root / "images" / rel)
                paths.append(rel)
            f.write(f'id{i},"{paths}"\n')
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    tmpdir = tempfile.mkdtemp(prefix="chadavit_data_")
    env = dict(os.environ, CHADAVIT_ROOT=ROOT, PYTHONPATH=ROOT, CHADAVIT_DATA=str(root), CHADAVIT_DIST_BACKEND="gloo", CHADAVIT_SINGLE_DEVICE="1",
               CHADAVIT_OUT=tmpdir)
    path = os.path.join(tmpdir, "_data_worker.py")
    with open(path, "w") as f:
        f.write(DATA_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), path]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    res = [json.load(open(os.path.join(tmpdir, f"rank{k}.json"))) for k in range(2)]
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["steps"] == b["steps"] == 2 and all(np.isfinite(a["losses"] + b["losses"]))
    for sa, sb in zip(a["seen"], b["seen"]):   # a global batch of 8: four images per rank, disjoint
        assert len(sa) == len(sb) == 4 and not set(sa) & set(sb)
    assert len({i for s_ in a["seen"] + b["seen"] for i in s_}) == 16
    # replicas: identical after two optimiser steps (averaged gradients, replicated AdamW / EMA / centre)
    assert a["w"] == b["w"] and a["wt"] == b["wt"] and abs(a["center"] - b["center"]) <= 1e-6 * abs(a["center"]) + 1e-9
    # sync_dist: both ranks read the same mean of their last local losses
    assert abs(a["logged"] - b["logged"]) < 1e-9 and abs(a["logged"] - 0.5 * (a["losses"][-1] + b["losses"][-1])) < 1e-4


@pytest.mark.slow
@pytest.mark.timeout(1200)
def test_bench_contract_single_gpu_with_all_legs():
    """`python bench.py` as the driver runs it at N = 1 (smaller batch, no CPU baseline): ONE JSON line with the contract's keys, the
    roofline object, the full-width leg, the cfg2-mixed / cfg3 / cfg5 / cfg1 (hipGraph) / cfg2-at-512 / cfg2-standard legs without an error entry, and the data-path
    leg.  Guards the round-end bench run against a leg that raises."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "32", "--no-cpu-baseline",
                        "--data", "pipeline"], cwd=ROOT, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and out["dtype"] == "bf16" and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and 0 < rf["frac"] < 1 and rf["avg_us"] > 0 and rf["kernel"]
    # the power cap is the roof that binds on real operands: when the box exposes the GPU's hwmon the line carries board power and shader clock over
    # the timed steps, and the dominant kernel priced against the part's measured power model
    if out.get("power") is not None:
        assert 0 < out["power"]["board_w"] <= 1.05 * (out["power"]["cap_w"] or 1400.0) and 500 < out["power"]["sclk_mhz"] <= 2500 and out["power"]["samples"] > 0
    if rf.get("power_model") is not None:
        assert 0.3 < rf["power_model"]["essential_frac_of_cap"] < 1.05
    cfg = out["config"]
    assert cfg["images_per_s_with_full_width_last_block"] > 0 and "workload" in cfg
    legs = cfg["other_workloads"]
    assert set(legs) == {"cfg2-mixed", "cfg3", "cfg5", "cfg1-graph", "cfg2-512", "cfg2-standard"} and legs["cfg2-512"]["images_per_gpu"] == 512
    # the data-parallel MECHANISM in a group of one RCCL rank (child process): 15 hand-overs to the communication stream per step
    w1 = out["rccl"]["world1_mechanism"]
    assert "error" not in w1, w1
    assert w1["backend"] == "nccl" and w1["world"] == 1 and w1["spans"] >= 14 and w1["ms_per_step"] > 0 and w1["mechanism_ms_per_step"] is not None
    # the north star's literal target workload (Tiny/16, 1-10 channels per image) carries its dominant kernel and roofline fraction
    assert "1-10 channel" in legs["cfg2-mixed"]["workload"] and legs["cfg2-mixed"]["dominant_kernel"] and 0 < legs["cfg2-mixed"]["frac"] < 1
    for name, leg in legs.items():
        assert "error" not in leg, (name, leg)
        assert leg["images_per_s"] > 0 and leg["ms_per_step"] > 0
    assert legs["cfg1-graph"]["launch"].startswith("one hipGraph") and legs["cfg3"]["launch"] == "eager"   # (no timing assertions here)
    dp = cfg["data_path"]
    assert "error" not in dp and dp["step_fed_by_pipeline_images_per_s"] > 0 and dp["decode_images_per_s_per_reader_thread"] > 0


@pytest.mark.slow
@pytest.mark.timeout(900)
def test_example_scripts_run(tmp_path):
    """examples/pretrain.py (synthetic planes -> device augmentation -> 3 DINO steps -> Lightning-shaped checkpoint + optimiser state)
    and examples/linear_eval.py on that checkpoint (one epoch, frozen backbone): they run, the loss is finite, the files exist."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    ck = str(tmp_path / "pre.ckpt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "pretrain.py"), "--batch", "8", "--steps", "3", "--local-crops", "2", "--out", ck],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "step 0: loss" in r.stdout and "nan" not in r.stdout.lower() and os.path.isfile(ck) and os.path.isfile(ck + ".optimizer")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "linear_eval.py"), "--ckpt", ck, "--epochs", "1", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "epoch 0: train_loss" in r.stdout and "nan" not in r.stdout.lower()


RCCL_WORLD1_WORKER = r"""
import os, sys, json, torch
sys.path.insert(0, os.environ["CHADAVIT_ROOT"])
import torch.distributed as dist
from chadavit_amd.parallel import GradSync, init_from_env, force_collectives
from chadavit_amd.methods.dino import DINO
from chadavit_amd.trainer import Trainer
from chadavit_amd.data.channels_strategies import one_channel_collate_fn
from oracle import procedural as P
from tests.golden_util import build_sd
from tests.test_model_gpu import _cfg
dev = torch.device("cuda", 0)
def run(forced):
    if forced:
        os.environ["CHADAVIT_FORCE_COLLECTIVES"] = "1"
        rank, world, local = init_from_env("nccl")          # RCCL, a process group of ONE rank
        assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1 and force_collectives()
    model = DINO(_cfg(192, 4096, 2, 1))
    model.load_state_dict(build_sd(192, 4096))
    model = model.to(dev)
    gs = GradSync() if forced else None
    tr = Trainer(max_epochs=10, steps_per_epoch=10, grad_sync=gs).attach(model)
    if forced:
        assert gs.reducer.active and gs.reducer.native_avg
        gs.reducer.timing = True
    tr.current_epoch = 1
    losses, pend, nspans = [], [], []
    for step in range(2):
        crops, labels, ncl = one_channel_collate_fn(P.make_images([3, 1, 2, 1], [224, 224, 96], seed=11 + step))
        losses.append(tr.train_step(([c.to(dev) for c in crops], labels.to(dev), ncl), 1).item())
        pend.append(model.dino_loss_func._pending is not None)
        nspans.append(len(gs.reducer.spans) if forced else 0)
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}   # (state_dict finishes the pending centre update)
    timing = gs.reducer.timing_summary() if forced else None
    return losses, pend, nspans, sd, timing
def eval_modules():
    # GradSync on the evaluation modules (no head, no teacher), collectives forced: RegressionModel has no `classifier` attribute at all
    # (the reference deletes it, src/methods/regression.py) -- attach() must reach its `regressor` through `out_layer`; with the backbone
    # fine-tuned its spans go through the reducer, the linear layer's two gradients in finish(); averages over ONE rank are identities
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.methods.linear import LinearModel
    from chadavit_amd.methods.regression import RegressionModel
    from tests.test_linear_gpu import _cfg as lin_cfg
    ok = []
    for cls, ft in ((RegressionModel, True), (RegressionModel, False), (LinearModel, False)):
        bb = vit_channels("dino", patch_size=16, embed_dim=192, return_all_tokens=False, max_number_channels=10)
        m = cls(bb, lin_cfg(192, False, 3, False, 1 if cls is RegressionModel else 7, ft, 0.1, 0.0)).to(dev)
        gs = GradSync().attach(m)
        lin = m.out_layer if hasattr(m, "out_layer") else m.classifier
        good = (cls is LinearModel or not hasattr(m, "classifier")) and gs.reducer.active and [id(p) for p in gs._plain] == [id(p) for p in lin.parameters()]
        good = good and ((m.backbone.grad_ready_hook is not None) == ft)
        for p in lin.parameters():
            p.grad = torch.full_like(p, 3.0)
        gs.begin_backward(); gs.finish(); torch.cuda.synchronize()
        good = good and all(bool((p.grad == 3.0).all()) for p in lin.parameters()) and gs.reducer.bytes == 4 * sum(p.numel() for p in lin.parameters())
        ok.append(bool(good))
    return ok
la, pa, na, sda, _ = run(False)
lb, pb, nb, sdb, timing = run(True)
evals = eval_modules()
same = all(torch.equal(sda[k], sdb[k]) for k in sda)
diff = [k for k in sda if not torch.equal(sda[k], sdb[k])][:5]
print("RESULT " + json.dumps({"losses_plain": la, "losses_rccl": lb, "pending_plain": pa, "pending_rccl": pb, "spans": nb, "state_equal": same,
                              "diff": diff, "timing": timing, "comm_stream": True, "eval_modules": evals}), flush=True)
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
def test_rccl_code_path_on_one_gpu():
    """The `nccl` (= RCCL) branches of the data-parallel path -- `SpanAllReduce.submit / finish` on the communication stream with
    `ReduceOp.AVG`, `DINOLoss.update_center`'s comm-stream all-reduce finished by `sync_center`, `record_stream`, the exposed-time
    events -- executed on this box's single GPU in a process group of one rank (CHADAVIT_FORCE_COLLECTIVES: the `world == 1`
    early-outs are skipped; every collective is an identity), so the first multi-GPU run is not also the first RCCL run.  Two
    training steps with the collectives forced must leave EXACTLY the state of two steps without them (bit for bit: weights, EMA
    teacher, centre), the centre update must have been in flight when each step returned, and every gradient span must have gone
    through the reducer."""
    env_port = socket.socket(); env_port.bind(("127.0.0.1", 0)); port = env_port.getsockname()[1]; env_port.close()
    env = dict(os.environ, CHADAVIT_ROOT=ROOT, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CHADAVIT_FORCE_COLLECTIVES", None)
    r = subprocess.run([sys.executable, "-c", RCCL_WORLD1_WORKER], env=env, capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    import json
    out = json.loads(lines[-1][7:])
    assert out["losses_plain"] == out["losses_rccl"], out
    assert out["state_equal"], out["diff"]
    assert out["pending_rccl"] == [True, True] and out["pending_plain"] == [False, False]
    assert all(n >= 13 for n in out["spans"]), out["spans"]      # 12 blocks' spans (+ embeddings / final norm) + the head
    assert out["timing"] is not None and out["timing"]["steps"] == 2 and out["timing"]["comm_busy_ms_per_step"] > 0, out["timing"]
    assert out["eval_modules"] == [True, True, True], out["eval_modules"]   # GradSync.attach on RegressionModel (fine-tune / frozen) and LinearModel
