"""cfg1 (Tiny/16, batch 4, 2 global crops) as one hipGraph: single-stream capture vs capture with DINO's side streams on (parallel branches)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from chadavit_amd.graphed import GraphedTrainStep

dev = torch.device("cuda:0")
args = argparse.Namespace(serial=False, overlap=False)
res = {}
for name in ("cfg1", "cfg2"):
    for par in (False, True):
        wl = dict(bench.WORKLOADS[name])
        if name == "cfg2":
            wl["batch"] = 16
        model, tr, _, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
        step = GraphedTrainStep(tr, parallel_streams=par)
        losses = [float(step(batch, i).item()) for i in range(4)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        for i in range(n):
            last = step(batch, 4 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        step.close()
        res[(name, par)] = (round(wl["batch"] / dt, 1), round(1e3 * dt, 3), losses, float(last.item()))
        print(name, "parallel" if par else "single-stream", res[(name, par)], flush=True)
        del model, tr, batch, step
        torch.cuda.empty_cache()
for name in ("cfg1", "cfg2"):
    a, b = res[(name, False)], res[(name, True)]
    print(name, "losses identical:", a[2] == b[2] and a[3] == b[3], " speedup", round(b[0] / a[0], 3))
