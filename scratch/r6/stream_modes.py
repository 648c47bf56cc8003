"""cfg2 step with the side streams switched on one group at a time (teacher / local-crop passes; weight-gradient GEMMs)."""
import sys, time, types, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
args = types.SimpleNamespace(serial=True, overlap=False)
model, tr, gs, batch, nch, _ = bench.build_workload(wl, args, 0, 1, dev)
def run(steps=8, warm=3):
    for i in range(warm): tr.train_step(batch, i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): tr.train_step(batch, warm + i)
    torch.cuda.synchronize(); return wl["batch"] * steps / (time.perf_counter() - t0)
for rep in range(2):
    for name, ov, dw in (("serial", False, False), ("teacher/local side streams", True, False), ("dW side stream", False, True), ("both", True, True)):
        model.overlap_streams = ov
        model.backbone.dw_side_stream = dw
        print(f"{name}: {run():.1f} images/s", flush=True)
