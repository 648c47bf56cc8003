"""Board power / shader clock while the D = 384 block kernels run back to back (set-up of scratch/blk384_time.py)."""
import sys, time, threading, glob, torch
src = open("scratch/blk384_time.py").read().split("fl = T *")[0]
ns = {}; exec(compile(src, "blk384", "exec"), ns)
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
def sample():
    best = (0, 0)
    for h in hw:
        try: v = (int(open(h + "/power1_input").read()) / 1e6, int(open(h + "/freq1_input").read()) / 1e6)
        except Exception: continue
        if v[0] > best[0]: best = v
    return best
for name in ("infer", "train"):
    fn = ns[name]
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop = False; rows = []
    def th():
        while not stop: rows.append(sample()); time.sleep(0.05)
    t = threading.Thread(target=th); t.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); n = 0; e0.record()
    while time.time() - t0 < 3.0:
        for _ in range(10): fn()
        n += 10; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop = True; t.join()
    half = rows[len(rows) // 2:]
    print(f"D=384 block kernel, {name:6s} {1e3 * e0.elapsed_time(e1) / n:9.1f} us  {sum(r[0] for r in half) / len(half):7.1f} W  {sum(r[1] for r in half) / len(half):7.1f} MHz", flush=True)
