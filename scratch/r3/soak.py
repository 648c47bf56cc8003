"""Soak: N training steps on a constant synthetic batch at the bench's size; the loss must stay finite and fall."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
name, steps = sys.argv[1], int(sys.argv[2])
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS[name])
model, tr, _, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
losses = []
for i in range(steps):
    losses.append(tr.train_step(batch, i).detach())
    if (i + 1) % max(1, steps // 8) == 0:
        torch.cuda.synchronize()
        fin = all(torch.isfinite(p).all().item() for p in model.parameters())
        print(f"{name} step {i + 1}: loss {float(losses[-1]):.4f} params finite {fin} center |max| {float(model.dino_loss_func.center.abs().max()):.3f} "
              f"lr {tr.optimizer.param_groups[0]['lr']:.2e} tau {model.momentum_updater.cur_tau:.5f}", flush=True)
l = torch.stack(losses).float().cpu()
print(f"{name}: first {float(l[0]):.4f} min {float(l.min()):.4f} last {float(l[-1]):.4f} all finite {bool(torch.isfinite(l).all())}")
