"""Board power / shader clock while ONE entry point of scratch/one_op.py runs back to back: python scratch/r5/power_ops.py name[,name...] [seconds]"""
import os, sys, time, threading, glob, torch
names = sys.argv[1].split(","); secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
def setup(name):
    sys.argv = ["one_op.py", name, "."]
    ns = {}
    exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
    return ns
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
def sample():
    best = (0, 0)
    for h in hw:
        try: v = (int(open(h + "/power1_input").read()) / 1e6, int(open(h + "/freq1_input").read()) / 1e6)
        except Exception: continue
        if v[0] > best[0]: best = v
    return best
for name in names:
    g = setup(name); fn = g["fn"]
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop = False; rows = []
    def th():
        while not stop: rows.append(sample()); time.sleep(0.05)
    t = threading.Thread(target=th); t.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time(); n = 0; e0.record()
    while time.time() - t0 < secs:
        for _ in range(10): fn()
        n += 10; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize(); stop = True; t.join()
    half = rows[len(rows) // 2:]
    print(f"{name:18s} T={os.environ.get('ONE_OP_T', '150784'):>8s} {1e3 * e0.elapsed_time(e1) / n:9.1f} us  {sum(r[0] for r in half) / len(half):7.1f} W  {sum(r[1] for r in half) / len(half):7.1f} MHz", flush=True)
    del g, fn; torch.cuda.empty_cache()
