"""Determinism fuzz at the bench's sizes: build the workload, run ONE training step, keep loss + gradients; rebuild (same seeds) with the
allocator's free blocks refilled, repeat, compare bit for bit.   python scratch/r3/fuzz_bench_scale.py cfg2 [cfg3 cfg5 cfg2-mixed]"""
import gc, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench

def poison(dev, value, gb=24):
    bufs = []
    for mb in (4096, 2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
        for _ in range(max(1, min(8, int(gb * 1024 / 13 / mb)))):
            bufs.append(torch.full((mb * 1024 * 1024 // 2,), value, device=dev, dtype=torch.bfloat16))
    torch.cuda.synchronize()
    del bufs

def main():
    names = sys.argv[1:] or ["cfg2"]
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda:0")
    for name in names:
        wl = dict(bench.WORKLOADS[name])
        ref = None
        for it, val in enumerate([None, 3.0e4, float("nan"), -1.0e3]):
            if val is not None:
                poison(dev, val)
            model, tr, _, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
            model.current_epoch = tr.current_epoch
            model.on_train_epoch_start()
            loss = model.training_step(batch, 0)
            loss.backward()
            model.on_after_backward()
            torch.cuda.synchronize()
            got = (float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
            if ref is None:
                ref = got
                print(f"{name} run 0: loss {got[0]:.6f}, {len(got[1])} gradient tensors, overlap_streams={model.overlap_streams}", flush=True)
            else:
                bad = sorted(((float((t.float() - ref[1][n].float()).norm() / (ref[1][n].float().norm() + 1e-30)), n) for n, t in got[1].items()
                              if not torch.equal(t, ref[1][n])), reverse=True)
                print(f"{name} run {it} (poison {val}): loss {got[0]:.6f} vs {ref[0]:.6f}; {len(bad)} gradient tensors differ; worst {bad[:4]}", flush=True)
            del model, tr, batch, got
            gc.collect(); torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
