"""Multi-process (world_size 2, gloo, CPU) test of the span all-reduce that the N>1 GPU path uses over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from chadavit_amd.parallel import SpanAllReduce, init_from_env
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = SpanAllReduce()
    red.timing = True
    # spans fired in backward order (last block first), with a gap that must stay untouched
    for b, e in ((800, 1000), (400, 800), (64, 400)):
        red.submit(flat, b, e)
    red.finish()
    expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    expect[:64] = torch.arange(64, dtype=torch.float32) * (rank + 1)
    ok = torch.allclose(flat, expect)
    # centre statistics: SUM all-reduce then / world (losses/dino.py:112-114)
    cs = torch.full((8,), float(rank + 1))
    dist.all_reduce(cs)
    ok = ok and torch.allclose(cs / world, torch.full((8,), sum(range(1, world + 1)) / world))
    # bench.py's rccl.exposed_ms_per_step on the CPU path: host time blocked on the async works
    ts = red.timing_summary()
    ok = ok and ts is not None and ts["steps"] == 1 and ts["exposed_ms_per_step"] >= 0 and red.timing_summary() is None
    # self.log(..., sync_dist=True) (dino.py:319): the value READ from the log is the mean over ranks
    from chadavit_amd.methods.dino import _Base
    m = _Base()
    m.log("dino_loss_train", torch.tensor(float(rank + 1)), on_step=True, sync_dist=True)
    m.log("tau", 0.5 + rank)  # not synced: stays rank-local
    got = m.logged_metrics()
    ok = ok and abs(got["dino_loss_train"] - sum(range(1, world + 1)) / world) < 1e-12 and got["tau"] == 0.5 + rank
    q.put((rank, bool(ok), red.bytes))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_span_allreduce_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert all(ok for _, ok, _ in res), res
    assert all(b == (200 + 400 + 336) * 4 for _, _, b in res), res


def _worker_cfg4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import bench
    from chadavit_amd.parallel import SpanAllReduce, init_from_env
    from chadavit_amd.ragged import RaggedBatch
    r, w, _ = init_from_env("gloo")
    wl = bench.WORKLOADS["cfg3"]            # cfg4 = cfg3's model and channel mix on 8 ranks x 128 images (BASELINE configs[3])
    nch, tokens = bench.rank_channels(wl, wl["batch"], r, w)
    mine = {"rank": r, "nch": nch, "tokens": tokens}
    for p in (196, 36):                     # the 224-pixel and the 96-pixel crops' ragged descriptions, as the step builds them
        rb = RaggedBatch(nch, p, "cpu")
        work = rb.work.numpy()
        items = work[work[:, 0] >= 0]
        tile = 128
        per_img = {}
        for b, t in items.tolist():
            per_img.setdefault(b, []).append(t)
        ok = rb.n_work % 8 == 0 and len(per_img) == len(nch) and all(sorted(v) == list(range((1 + nch[b] * p + tile - 1) // tile)) for b, v in per_img.items())
        # all tiles of an image sit in ONE residue class mod 8 (its XCD's sub-list)
        pos = {}
        for j, (b, t) in enumerate(work.tolist()):
            if b >= 0:
                pos.setdefault(b, set()).add(j % 8)
        mine[f"work_ok_{p}"] = bool(ok and all(len(v) == 1 for v in pos.values()))
        mine[f"n_work_{p}"] = int(rb.n_work)
        mine[f"T_{p}"] = int(rb.T)
    gathered = [None] * w
    dist.all_gather_object(gathered, mine)
    # the gradient spans' collective on eight ranks: mean over ranks, untouched gap
    flat = torch.arange(512, dtype=torch.float32) * (r + 1)
    red = SpanAllReduce()
    for b, e in ((256, 512), (32, 256)):
        red.submit(flat, b, e)
    red.finish()
    expect = torch.arange(512, dtype=torch.float32) * (sum(range(1, w + 1)) / w)
    expect[:32] = torch.arange(32, dtype=torch.float32) * (r + 1)
    q.put((r, gathered, bool(torch.allclose(flat, expect))))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_cfg4_batch_construction_gloo_world8():
    """The first 8-GPU launch must not fail on partitioning (BASELINE configs[3]: Small, 1-10 channels, DDP 8 ranks, global batch 1024;
    /root/reference/main_pretrain.py:301-306 shards with Lightning's DistributedSampler): eight gloo ranks build bench.py's batch for it --
    every rank 128 images, together exactly the seeded global batch, global-crop tokens within 1 % across ranks, every rank's attention work
    lists (224- and 96-pixel crops) a multiple of 8 with each image's tiles in one XCD class -- and run the span all-reduce on eight ranks."""
    import bench
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_cfg4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    B = bench.WORKLOADS["cfg3"]["batch"]
    assert B * world == 1024
    glob = bench.channel_list("1-10", 1024, seed=1000)
    for r, gathered, ok in res:
        assert ok, r
        assert [g["rank"] for g in gathered] == list(range(world))
        assert gathered == res[0][1]                          # every rank saw the same eight descriptions
    g = res[0][1]
    assert all(len(x["nch"]) == B for x in g)
    assert sorted(c for x in g for c in x["nch"]) == sorted(glob)
    tokens = [sum(1 + c * 196 for c in x["nch"]) * 2 for x in g]
    assert all(x["tokens"] == tokens for x in g)
    assert max(tokens) / min(tokens) - 1.0 <= 0.01, tokens
    for x in g:
        assert x["work_ok_196"] and x["work_ok_36"] and x["n_work_196"] % 8 == 0 and x["n_work_36"] % 8 == 0
        assert x["T_196"] * 2 == x["tokens"][x["rank"]]


def test_library_exports_every_declared_symbol():
    """C-ABI contract: libchadavit_hip.so loads without a GPU and exports everything include/*.h declares."""
    from chadavit_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 25
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.chadavit_abi_version() == _lib.ABI_VERSION
    assert lib.chadavit_attn_tile_rows() == 128


def test_no_cpu_fallback():
    from chadavit_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm_fwd(torch.zeros(4, 64, dtype=torch.bfloat16), torch.ones(64), torch.zeros(64), 1e-5)


def test_ragged_batch_layout():
    from chadavit_amd.ragged import RaggedBatch
    rb = RaggedBatch([3, 1, 10], 196, "cpu")
    assert rb.T == 3 + 14 * 196 and rb.B == 3 and rb.n_chan == 14
    assert rb.cu_seqlens.tolist() == [0, 589, 786, 2747]
    assert rb.chan_img.tolist() == [0] * 3 + [1] + [2] * 10
    assert rb.chan_idx.tolist() == [0, 1, 2, 0] + list(range(10))
    w = rb.work.tolist()
    # entry j runs on XCD j % 8: an image's tiles all sit in one residue class; padding entries are (-1, 0)
    real = [tuple(e) for e in w if e[0] >= 0]
    assert sorted(real) == sorted([(0, t) for t in range(5)] + [(1, t) for t in range(2)] + [(2, t) for t in range(16)])
    assert len(w) % 8 == 0 and w[0] == [2, 0]  # longest image first
    for b in range(3):
        assert len({j % 8 for j, e in enumerate(w) if e[0] == b}) == 1
    many = RaggedBatch([3] * 64 + [1] * 7, 36, "cpu")
    per_xcd = [sum(1 for j, e in enumerate(many.work.tolist()) if e[0] >= 0 and j % 8 == x) for x in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 1
    assert rb.cls_rows.tolist() == [0, 589, 786]


def test_collate_matches_oracle():
    from chadavit_amd.data.channels_strategies import one_channel_collate_fn
    from oracle import chada_ref as R
    from oracle import procedural as P
    imgs = P.make_images([2, 5, 1], [32, 16], seed=1)
    a, la, na = one_channel_collate_fn(imgs)
    b, lb, nb = R.collate(imgs)
    assert na == nb and torch.equal(la, lb) and all(torch.equal(x, y) for x, y in zip(a, b))
    single = [([c[0]], l) for c, l in imgs]
    a1, _, n1 = one_channel_collate_fn([(c[0], l) for c, l in single])
    assert isinstance(a1, torch.Tensor) and a1.shape == (8, 1, 32, 32) and n1 == [[2, 5, 1]]


def test_collate_groups_equal_resolutions_into_one_buffer():
    """Crops of one resolution leave the collate back to back in one buffer (same tensors as the reference's collate), so the
    training step can take "all global crops" as a view: adjacent_view == torch.cat without the copy, None when it cannot."""
    from chadavit_amd.data.channels_strategies import adjacent_view, one_channel_collate_fn
    from oracle import chada_ref as R
    from oracle import procedural as P
    imgs = P.make_images([2, 5, 1], [32, 32, 16, 16, 16], seed=2)
    a, _, na = one_channel_collate_fn(imgs)
    b, _, nb = R.collate(imgs)
    assert na == nb and all(torch.equal(x, y) for x, y in zip(a, b))
    g, l = adjacent_view(a[:2]), adjacent_view(a[2:])
    assert g is not None and l is not None
    assert torch.equal(g, torch.cat(a[:2])) and torch.equal(l, torch.cat(a[2:]))
    assert g.data_ptr() == a[0].data_ptr() and l.data_ptr() == a[2].data_ptr()   # views, not copies
    assert adjacent_view([a[1], a[0]]) is None and adjacent_view([a[0], a[2]]) is None
    assert adjacent_view([t.clone() for t in a[:2]]) is None
    assert adjacent_view([a[0][:, :, ::2]]) is None


def test_schedules_and_optimizer_host_logic():
    import numpy as np
    from chadavit_amd.optim import WarmupCosineLR
    from chadavit_amd.utils.momentum import MomentumUpdater
    from oracle import chada_ref as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "schedules.npz"))
    mu = MomentumUpdater(0.9995, 1.0)
    for step, tau in zip(g["tau_steps"], g["taus"]):
        mu.update_tau(int(step), 100)
        assert abs(mu.cur_tau - float(tau)) < 1e-12
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([w], lr=float(g["base_lr"]))
    sch = WarmupCosineLR(opt, warmup_epochs=float(g["warmup"]), max_epochs=float(g["max_steps"]),
                         warmup_start_lr=float(g["warmup_start_lr"]), eta_min=float(g["eta_min"]))
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-9, atol=1e-12)


def test_weight_decay_split_matches_reference_golden():
    import numpy as np
    from chadavit_amd.optim import remove_bias_and_norm_from_weight_decay
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lars.npz"))
    shapes = [(64, 48), (64,), (16, 64, 3), (16,), (8, 8)]
    groups = [{"name": "backbone", "params": [torch.nn.Parameter(torch.zeros(s)) for s in shapes], "lr": 0.1},
              {"name": "head", "params": [torch.nn.Parameter(torch.zeros(3))], "weight_decay": 0.5}]
    split = remove_bias_and_norm_from_weight_decay(groups)
    assert [x["name"] for x in split] == [str(n) for n in g["split_names"]]
    assert [len(x["params"]) for x in split] == [int(c) for c in g["split_counts"]]
    assert [float(x.get("weight_decay", -1)) for x in split] == [float(w) for w in g["split_wd"]]


def test_state_dict_layout_matches_reference_and_checkpoint_roundtrip(tmp_path):
    """DINO.state_dict() has exactly the reference module's keys and shapes (golden from the unmodified reference), a
    Lightning-style checkpoint round-trips, and the evaluation scripts' backbone-key rewrite (main_linear.py:103-110) matches
    the oracle's restatement of it."""
    import numpy as np
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.utils.checkpoint import backbone_state_dict, load_backbone, load_checkpoint, save_checkpoint
    from chadavit_amd.utils.misc import AttrDict
    from oracle import chada_ref as R
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_knn_ckpt.npz"))
    cfg = AttrDict({
        "method": "dino",
        "backbone": {"name": "vit_channels", "kwargs": {"embed_dim": 192, "patch_size": 16, "return_all_tokens": False,
                                                        "max_number_channels": 10}},
        "data": {"dataset": "synthetic", "num_classes": 7, "max_img_channels": 10, "img_channels": 1, "num_large_crops": 2,
                 "num_small_crops": 8},
        "channels_strategy": "multi_channels", "mixed_channels": True, "weights_init": "random", "max_epochs": 10,
        "optimizer": {"name": "adamw", "batch_size": 4, "lr": 5e-4, "weight_decay": 1e-4, "classifier_lr": 0.1},
        "scheduler": {"name": "warmup_cosine"}, "momentum": {"base_tau": 0.9995, "final_tau": 1.0},
        "method_kwargs": {"proj_hidden_dim": 2048, "proj_output_dim": 256, "num_prototypes": 4096,
                          "warmup_teacher_temperature_epochs": 3},
    })
    model = DINO(cfg)
    sd = model.state_dict()
    want = {str(k): tuple(int(v) for v in str(s).split(",") if v != "") for k, s in zip(g["sd_keys"], g["sd_shapes"])}
    assert list(sd.keys()) == [str(k) for k in g["sd_keys"]]
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    path = str(tmp_path / "last.ckpt")
    save_checkpoint(model, path, epoch=3, global_step=42)
    other = DINO(cfg)
    ck = load_checkpoint(other, path)
    assert ck["epoch"] == 3 and ck["global_step"] == 42
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), other.state_dict().values()))
    bb = backbone_state_dict(ck["state_dict"])
    ref_bb = R.strip_backbone_prefix(ck["state_dict"])
    assert list(bb.keys()) == list(ref_bb.keys()) and "cls_token" in bb and "blocks.11.linear2.weight" in bb
    assert not any(k.startswith("head.") or k.startswith("backbone.") for k in bb)
    from chadavit_amd.backbones import vit_channels
    fresh = vit_channels("dino", patch_size=16, embed_dim=192, return_all_tokens=False, max_number_channels=10)
    res = load_backbone(fresh, path)
    assert not res.missing_keys
    assert torch.equal(fresh.state_dict()["blocks.3.linear1.weight"], sd["backbone.blocks.3.linear1.weight"])


def test_token_balanced_sampler():
    """Every image of the shuffled global batch is used exactly once per step, ranks get equal image counts, and the per-rank
    cost spread is far below that of a plain contiguous split (variable-channel data)."""
    import random
    from chadavit_amd.data.sampler import TokenBalancedBatchSampler, balanced_partition, image_cost
    rng = random.Random(0)
    nch = [rng.randint(1, 10) for _ in range(1024 * 3 + 17)]
    world, gb = 8, 1024
    samplers = [TokenBalancedBatchSampler(nch, gb, r, world, seed=3) for r in range(world)]
    for s in samplers:
        s.set_epoch(2)
    assert len(samplers[0]) == 3
    perm = torch.randperm(len(nch), generator=torch.Generator().manual_seed(3 + 2)).tolist()
    for step, batches in enumerate(zip(*[iter(s) for s in samplers])):
        flat = [i for b in batches for i in b]
        assert sorted(flat) == sorted(perm[step * gb:(step + 1) * gb])  # the same images a DistributedSampler step would use
        assert len(flat) == gb and len(set(flat)) == gb and all(len(b) == gb // world for b in batches)
        cost = [sum(image_cost(nch[i]) for i in b) for b in batches]
        shuffled = perm[step * gb:(step + 1) * gb]
        plain = [sum(image_cost(nch[i]) for i in shuffled[r::world]) for r in range(world)]  # DistributedSampler's strided split
        assert (max(cost) - min(cost)) / max(cost) < 0.01
        assert (max(cost) - min(cost)) < 0.2 * (max(plain) - min(plain))
    a = list(iter(samplers[0]))
    samplers[0].set_epoch(3)
    assert a != list(iter(samplers[0]))
    with pytest.raises(ValueError):
        balanced_partition([1.0] * 10, 4)
    assert image_cost(10) / image_cost(1) > 15


def test_bench_mixed_channel_batches_are_token_balanced_across_ranks():
    """bench.py shards ONE seeded global batch of a 1-10 channel workload with data/sampler.py's token-balanced partition:
    equal image counts per rank, per-rank token totals within 2 % (an independent draw per rank spreads them by tens of %)."""
    import bench
    from chadavit_amd.data.sampler import balanced_partition, image_cost
    for world in (2, 4, 8):
        for B in (32, 128):
            g = bench.channel_list("1-10", B * world, seed=1000)
            parts = balanced_partition([image_cost(c) for c in g], world)
            assert sorted(i for p in parts for i in p) == list(range(B * world)) and all(len(p) == B for p in parts)
            tokens = [sum(1 + g[i] * 196 for i in p) for p in parts]
            assert max(tokens) / min(tokens) - 1.0 <= 0.02, (world, B, tokens)
        naive = [sum(1 + c * 196 for c in bench.channel_list("1-10", 32, seed=1000 + r)) for r in range(world)]
        assert max(naive) / min(naive) - 1.0 > 0.02  # what the sampler is there to avoid


def test_bench_power_sampler_reads_the_gpus_own_hwmon_and_degrades_to_none(tmp_path, monkeypatch):
    """bench.py's `power` object: board power / shader clock of THIS rank's GPU from the hwmon directory of its PCI function, sampled from a thread over
    the timed steps -- and simply absent (no exception, no key) where sysfs does not expose it."""
    import glob as _glob
    import time
    import types
    import bench
    hw = tmp_path / "hwmon7"
    hw.mkdir()
    (hw / "power1_input").write_text("1370000000\n")
    (hw / "freq1_input").write_text("1938000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    seen = []
    monkeypatch.setattr(_glob, "glob", lambda pat: (seen.append(pat), [str(hw)])[1])
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x75, pci_device_id=0)
    fake = types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda dev: props))
    ps = bench.PowerSampler(fake, 0)
    assert seen == ["/sys/bus/pci/devices/0000:75:00.0/hwmon/hwmon*"] and ps.idle == (1370.0, 1938.0)
    ps.start()
    time.sleep(0.5)   # (the sampler reads at 5 Hz)
    out = ps.stop()
    assert out["board_w"] == 1370.0 and out["sclk_mhz"] == 1938.0 and out["cap_w"] == 1400.0 and out["samples"] >= 2
    monkeypatch.setattr(_glob, "glob", lambda pat: [])
    ps = bench.PowerSampler(fake, 0)
    ps.start()
    assert ps.dir is None and ps.idle is None and ps.stop() is None
    broken = types.SimpleNamespace(cuda=types.SimpleNamespace(get_device_properties=lambda dev: (_ for _ in ()).throw(RuntimeError("no device"))))
    assert bench.PowerSampler(broken, 0).stop() is None


def test_ragged_batch_cache_returns_the_same_description():
    from chadavit_amd.ragged import _CACHE, _CACHE_MAX, ragged_batch
    dev = torch.device("cpu")
    a = ragged_batch([3, 1, 2], 196, dev)
    assert ragged_batch((3, 1, 2), 196, dev) is a and ragged_batch([3, 1, 2], 36, dev) is not a
    for i in range(2 * _CACHE_MAX):
        ragged_batch([i + 1], 4, dev)
    assert len(_CACHE) == _CACHE_MAX
    b = ragged_batch([3, 1, 2], 196, dev)   # evicted meanwhile: rebuilt, same content
    assert b is not a and torch.equal(b.cu_seqlens, a.cu_seqlens) and torch.equal(b.work, a.work) and b.T == a.T


def test_step_scheduler_and_unknown_scheduler_as_the_reference():
    """scheduler.name = "step" -> MultiStepLR over scheduler.lr_decay_steps (base.py:472-473); an unknown name raises the reference's
    ValueError (base.py:474-475).  Host logic only: configure_optimizers builds the fused optimiser without touching the GPU."""
    import warnings
    import pytest
    from chadavit_amd.methods.dino import DINO
    from chadavit_amd.trainer import Trainer
    from chadavit_amd.utils.misc import AttrDict

    def cfg(name):
        return AttrDict({"method": "dino", "backbone": {"name": "vit_channels", "kwargs": {"embed_dim": 192, "patch_size": 16, "return_all_tokens": False,
                                                                                       "max_number_channels": 10}},
                         "data": {"dataset": "synthetic", "num_classes": 7, "max_img_channels": 10, "img_channels": 1, "num_large_crops": 2,
                                  "num_small_crops": 0},
                         "channels_strategy": "multi_channels", "mixed_channels": True, "weights_init": "random", "max_epochs": 10,
                         "optimizer": {"name": "sgd", "batch_size": 4, "lr": 0.5, "weight_decay": 0.0, "classifier_lr": 0.1},
                         "scheduler": {"name": name, "lr_decay_steps": [3, 6]}, "momentum": {"base_tau": 0.99, "final_tau": 1.0},
                         "method_kwargs": {"proj_hidden_dim": 64, "proj_output_dim": 32, "num_prototypes": 128}})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = DINO(cfg("step"))
        tr = Trainer(max_epochs=10, steps_per_epoch=1).attach(m)
        lrs = []
        for _ in range(8):
            lrs.append(tr.optimizer.param_groups[0]["lr"])
            tr.scheduler.step()
    assert lrs == pytest.approx([0.5, 0.5, 0.5, 0.05, 0.05, 0.05, 0.005, 0.005])
    assert tr.optimizer.param_groups[1]["lr"] == pytest.approx(0.1 * 0.01)    # the classifier group decays with the rest
    with pytest.raises(ValueError):
        Trainer(max_epochs=10, steps_per_epoch=1).attach(DINO(cfg("cosine_restarts")))


def test_linear_and_regression_host_logic():
    """Host-side pieces of the evaluation modules, no GPU: accuracy_at_k / weighted_mean as the reference's (src/utils/metrics.py), the
    regression metrics against numpy / scipy, the classifier's input width for CLS and all-token features (linear.py:118-138), the
    scheduler table of configure_optimizers (linear.py:326-369) and the parameter names of both modules' state_dict."""
    import warnings
    import numpy as np
    import pytest
    import torch
    from scipy import stats
    from chadavit_amd.backbones import vit_channels
    from chadavit_amd.methods.linear import LinearModel, accuracy_at_k, weighted_mean
    from chadavit_amd.methods.regression import RegressionModel, regression_metrics
    from chadavit_amd.trainer import Trainer
    from chadavit_amd.utils.misc import AttrDict
    from oracle import chada_ref as R
    g = torch.Generator().manual_seed(0)
    out, tgt = torch.randn(64, 9, generator=g), torch.randint(0, 9, (64,), generator=g)
    a1, a5 = accuracy_at_k(out, tgt)
    assert [float(a1), float(a5)] == pytest.approx(R.accuracy_at_k(out, tgt))
    assert float(accuracy_at_k(out[:, :3], tgt % 3)[1]) == 100.0          # fewer than five classes: top-5 is everything
    steps = [{"batch_size": 4, "v": torch.tensor([2.0])}, {"batch_size": 12, "v": torch.tensor([6.0])}]
    assert float(weighted_mean(steps, "v", "batch_size")) == pytest.approx(5.0)
    o, t = torch.randn(50, 1, generator=g), torch.randn(50, 1, generator=g)
    m = regression_metrics(o, t)
    on, tn = o.view(-1).numpy(), t.view(-1).numpy()
    assert float(m["mse"]) == pytest.approx(np.mean((on - tn) ** 2), rel=1e-5) and float(m["mae"]) == pytest.approx(np.mean(np.abs(on - tn)), rel=1e-5)
    assert float(m["pcc"]) == pytest.approx(stats.pearsonr(on, tn)[0], rel=1e-4)
    assert float(m["r2"]) == pytest.approx(1 - np.sum((on - tn) ** 2) / np.sum((tn - tn.mean()) ** 2), rel=1e-5)

    def cfg(rat, sched="none", opt="sgd", finetune=False):
        return AttrDict({"backbone": {"name": "vit_channels", "kwargs": {"embed_dim": 192, "patch_size": 16, "return_all_tokens": rat, "max_number_channels": 10}},
                         "data": {"dataset": "synthetic", "num_classes": 7, "img_channels": 3, "max_img_channels": 10},
                         "channels_strategy": "multi_channels", "mixed_channels": False, "max_epochs": 10, "finetune": finetune,
                         "optimizer": {"name": opt, "batch_size": 4, "lr": 0.1, "weight_decay": 0.5},
                         "scheduler": {"name": sched, "lr_decay_steps": [2, 4]}})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bb = vit_channels("dino", patch_size=16, embed_dim=192, return_all_tokens=False, max_number_channels=10)
        lm = LinearModel(bb, cfg(False))
        assert lm.classifier.in_features == 192 and not any(p.requires_grad for p in bb.parameters())
        assert LinearModel(bb, cfg(True)).classifier.in_features == 3 * 196 * 192
        assert sorted(k for k in lm.state_dict() if not k.startswith("backbone.")) == ["classifier.bias", "classifier.weight"]
        rm = RegressionModel(bb, cfg(False))
        assert rm.regressor.out_features == 1 and sorted(k for k in rm.state_dict() if not k.startswith("backbone.")) == ["regressor.bias", "regressor.weight"]
        assert isinstance(rm.loss_func, torch.nn.MSELoss) and rm.out_layer is rm.regressor
        for sched, kind in (("none", type(None)), ("step", torch.optim.lr_scheduler.MultiStepLR), ("exponential", torch.optim.lr_scheduler.ExponentialLR),
                            ("reduce", torch.optim.lr_scheduler.ReduceLROnPlateau), ("warmup_cosine", torch.optim.lr_scheduler.LRScheduler)):
            tr = Trainer(max_epochs=10, steps_per_epoch=5).attach(LinearModel(bb, cfg(False, sched)))
            assert isinstance(tr.scheduler, kind), sched
            assert [len(g_["params"]) for g_ in tr.optimizer.param_groups] == [2]           # the classifier alone
        tr = Trainer(max_epochs=10, steps_per_epoch=5).attach(LinearModel(bb, cfg(False, "none", "adamw", finetune=True)))
        assert [g_["name"] for g_ in tr.optimizer.param_groups] == ["backbone", "classifier"]
        with pytest.raises(ValueError):
            Trainer(max_epochs=10, steps_per_epoch=5).attach(LinearModel(bb, cfg(False, "cosine_restarts")))
        bad = cfg(False)
        bad.optimizer.layer_decay = 0.75
        with pytest.raises(RuntimeError):
            LinearModel(bb, bad)


def test_recorded_ragged_descriptions_outlive_the_cache():
    """A captured hipGraph bakes in the addresses of the ragged index arrays: `recording_uses` hands every description used during
    the capture to the graph's owner, so that the 8-entry cache forgetting them does not free them (advisor, round 3)."""
    from chadavit_amd.ragged import _CACHE, _CACHE_MAX, ragged_batch, recording_uses
    dev = torch.device("cpu")
    owned = []
    with recording_uses(owned):
        a = ragged_batch([2, 1], 196, dev)
        b = ragged_batch([2, 1], 36, dev)
        assert ragged_batch([2, 1], 196, dev) is a      # a second use is not recorded twice
    assert len(owned) == 2 and owned[0] is a and owned[1] is b
    ragged_batch([5], 4, dev)                           # outside the context: not recorded
    assert len(owned) == 2
    for i in range(2 * _CACHE_MAX):                     # the cache forgets both ...
        ragged_batch([i + 1], 9, dev)
    assert all(v is not a and v is not b for v in _CACHE.values())
    assert owned[0].cu_seqlens.tolist() == [0, 393, 590]   # ... the owner still holds them


def _forced_worker(port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", CHADAVIT_FORCE_COLLECTIVES="1")
    import torch.distributed as dist
    from chadavit_amd.parallel import SpanAllReduce, force_collectives, init_from_env
    rank, world, _ = init_from_env("gloo")
    ok = dist.is_initialized() and world == 1 and force_collectives()
    red = SpanAllReduce()
    ok = ok and red.active and red.world == 1
    flat = torch.arange(100, dtype=torch.float32)
    red.submit(flat, 10, 60)
    red.finish()
    ok = ok and len(red.spans) == 1 and torch.equal(flat, torch.arange(100, dtype=torch.float32))   # an identity at one rank
    q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_collectives_can_be_forced_in_a_group_of_one_rank():
    """CHADAVIT_FORCE_COLLECTIVES=1: the process group is created at world size 1 and the gradient reducer issues its collectives
    anyway (the GPU box runs the RCCL branches this way, tests/test_ddp_gpu.py::test_rccl_code_path_on_one_gpu); without the
    switch a single rank stays collective-free."""
    from chadavit_amd.parallel import SpanAllReduce
    assert not SpanAllReduce().active
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=100)
    p.join(timeout=30)


def test_dino_loss_module_accepts_the_standard_multicrop_view_count_and_checks_shapes():
    """DINOLoss(num_large_crops=V): V student views as the reference's chunking arithmetic allows (losses/dino.py:82); fewer than the
    teacher's two views is rejected, and so is a student tensor that is not V views of the teacher's images (checked before any kernel)."""
    from chadavit_amd.losses.dino import DINOLoss
    kw = dict(num_prototypes=64, warmup_teacher_temp=0.04, teacher_temp=0.07, warmup_teacher_temp_epochs=3, num_epochs=10)
    assert DINOLoss(**kw).num_large_crops == 2 and DINOLoss(num_large_crops=10, **kw).num_large_crops == 10
    with pytest.raises(RuntimeError):
        DINOLoss(num_large_crops=1, **kw)
    lf = DINOLoss(num_large_crops=5, **kw)
    with pytest.raises(RuntimeError, match="5 views"):
        lf(torch.zeros(8, 64), torch.zeros(4, 64))      # 8 rows are 4 views of 2 images, not 5 (and not the 2 global views)
    with pytest.raises(RuntimeError, match="5 views"):
        lf(torch.zeros(9, 64), torch.zeros(4, 64))      # not a whole number of views


LIGHTNING_BOUNDARY = r"""
import inspect, re, sys, types
import torch.nn as nn

# ---- a stand-in `pytorch_lightning` installed BEFORE the product is imported: faithful where the product could trip over the real one --
# `trainer` is a property that raises while unattached, `current_epoch` / `global_step` are READ-ONLY properties fed by the trainer
class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        self._trainer = None
    @property
    def trainer(self):
        if self._trainer is None:
            raise RuntimeError(f"{type(self).__qualname__} is not attached to a `Trainer`.")
        return self._trainer
    @trainer.setter
    def trainer(self, t):
        self._trainer = t
    @property
    def current_epoch(self):
        return self._trainer.current_epoch if self._trainer is not None else 0
    @property
    def global_step(self):
        return self._trainer.global_step if self._trainer is not None else 0
    def log(self, *a, **k): pass
    def log_dict(self, *a, **k): pass
pl = types.ModuleType("pytorch_lightning"); pl.LightningModule = LightningModule
sys.modules["pytorch_lightning"] = pl
sys.path.insert(0, ROOT)

import chadavit_amd.methods.dino as D
assert issubclass(D.DINO, LightningModule), D.DINO.__mro__          # the Lightning-installed branch of the boundary (methods/dino.py)

# ---- SURVEY 8(b): the hook set with Lightning's signatures
want = {"configure_optimizers": [], "on_train_start": [], "on_train_epoch_start": [], "training_step": ["batch", "batch_idx"],
        "on_after_backward": [], "optimizer_zero_grad": ["epoch", "batch_idx", "optimizer"], "on_train_batch_end": ["outputs", "batch", "batch_idx"],
        "validation_step": ["batch", "batch_idx"], "on_validation_epoch_end": [], "forward": ["X", "index"], "momentum_forward": ["X", "index"],
        "extract_features": ["batch"]}
for name, lead in want.items():
    params = [p for p in inspect.signature(getattr(D.DINO, name)).parameters.values()][1:]
    names = [p.name for p in params]
    assert names[:len(lead)] == lead, (name, names)
    assert all(p.default is not inspect.Parameter.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in params[len(lead):]), (name, names)
assert isinstance(inspect.getattr_static(D.DINO, "add_and_assert_specific_cfg"), staticmethod)
assert isinstance(D.DINO.learnable_params, property) and isinstance(D.DINO.momentum_pairs, property)

# ---- INTEGRATION.md section A, executed as written: the maintainer's src/mi355x.py against a stand-in `src.methods`
src = types.ModuleType("src"); methods = types.ModuleType("src.methods"); methods.METHODS = {}; src.methods = methods
sys.modules["src"] = src; sys.modules["src.methods"] = methods
code = re.search(r"```python\n(# src/mi355x.py.*?)```", open(ROOT + "/INTEGRATION.md").read(), re.S).group(1)
ns = {}
exec(code, ns)
ns["install"]()
ref_cv = sys.modules["src.backbones.vit.chada_vit"]
assert inspect.ismodule(ref_cv) and inspect.isclass(ref_cv.ChAdaViT)
from oracle import refshim
model = methods.METHODS["dino"](refshim.dino_cfg(embed_dim=192, num_prototypes=256))      # main_pretrain.py:87
assert isinstance(model, LightningModule)
assert isinstance(model.backbone, ref_cv.ChAdaViT) and isinstance(model.momentum_backbone, ref_cv.ChAdaViT)   # base.py:526-528
assert sys.modules["src.losses.dino"].DINOLoss is type(model.dino_loss_func)
assert callable(sys.modules["src.data.channels_strategies"].one_channel_collate_fn)
assert hasattr(sys.modules["src.utils.momentum"], "MomentumUpdater") and hasattr(sys.modules["src.utils.momentum"], "initialize_momentum_params")

# ---- the bundled loop on such a module: attaches, configures the optimiser, and never writes Lightning's read-only properties
from chadavit_amd.trainer import Trainer
tr = Trainer(max_epochs=3, steps_per_epoch=5).attach(model)
assert model.trainer is tr and tr.optimizer is not None
tr.current_epoch = 2
assert model.current_epoch == 2
print("LIGHTNING-BOUNDARY-OK")
"""


def test_lightning_boundary(tmp_path):
    """the boundary exercised the way the reference would.  A `pytorch_lightning` stand-in is installed BEFORE the
    product is imported (fresh interpreter): `DINO` must then subclass its LightningModule, expose SURVEY 8(b)'s hook set with
    Lightning's signatures, and INTEGRATION.md section A's `install()` -- executed from the document itself -- must make
    `isinstance(model.backbone, src.backbones.vit.chada_vit.ChAdaViT)` true (base.py:526-528).  Construction only: no GPU."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    script = tmp_path / "lightning_boundary.py"
    script.write_text("ROOT = %r\n" % root + LIGHTNING_BOUNDARY)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "LIGHTNING-BOUNDARY-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
