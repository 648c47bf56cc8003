"""Host-side cost of one training step of a bench workload: wall per step without syncs, cProfile over 3 steps (GPU async), and the time at which
the host returns from each step against the GPU's completion."""
import cProfile, pstats, sys, io, time, argparse, torch
sys.path.insert(0, '.')
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
sys.argv = ["bench.py", "--workload", name, "--data", "resident"]
args = bench.parse()
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS[name])
model, tr, gs, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
for i in range(3): tr.train_step(batch, i)
torch.cuda.synchronize()
# host time per step (no sync) vs GPU time per step
t0 = time.perf_counter(); hs = []
for i in range(4):
    a = time.perf_counter(); tr.train_step(batch, 3 + i); hs.append(time.perf_counter() - a)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{name}: host returns after {[round(1e3 * h, 1) for h in hs]} ms per step (sum {1e3 * t_host:.1f}); GPU done after {1e3 * t_all:.1f} ms = {1e3 * t_all / 4:.1f} per step")
pr = cProfile.Profile(); pr.enable()
for i in range(3): tr.train_step(batch, 7 + i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60); print(s.getvalue()[:12000])
