"""Host-side profile of the eager cfg1 step (launch-bound: ~700 launches, GPU time ~4 ms): where do the ~10 ms of interpreter go?"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
args = argparse.Namespace(serial=False, overlap=False)
model, tr, _, batch, nch, _ = bench.build_workload(dict(bench.WORKLOADS["cfg1"]), args, 0, 1, dev)
for i in range(5):
    tr.train_step(batch, i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(30):
    tr.train_step(batch, 5 + i)
torch.cuda.synchronize()
print("eager ms/step", round((time.perf_counter() - t0) / 30 * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for i in range(20):
    tr.train_step(batch, 40 + i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
