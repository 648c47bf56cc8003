"""What do the board power and the shader clock read while each hot kernel runs back to back for ~2 s?  (rocm-smi / sysfs sampled from a thread.)"""
import os, sys, time, threading, subprocess, glob, json, torch
os.environ.setdefault("ONE_OP_T", "603136")
def setup(name):
    sys.argv = ["one_op.py", name, "."]
    ns = {}
    exec(compile(open("scratch/one_op.py").read().split("for _ in range(4): fn()")[0], "one_op", "exec"), ns)
    return ns
hw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
print("hwmon:", hw, flush=True)
def rd(p):
    try: return open(p).read().strip()
    except Exception as e: return None
def sample():   # the card that draws the most is the one this process runs on (the box shows all eight in sysfs)
    best = {}
    for h in hw:
        out = {}
        for f in ("power1_input", "freq1_input"):
            v = rd(os.path.join(h, f))
            if v is not None: out[f] = int(v)
        if out.get("power1_input", 0) > best.get("power1_input", -1): best = out
    return best
def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20)
        return r.stdout.strip()[:1500]
    except Exception as e: return repr(e)
print("idle sysfs:", sample(), flush=True)
print("idle smi:", smi(), flush=True)
def run(label, fn, secs=4.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    stop = False; samples = []
    smis = []
    def th():
        while not stop:
            samples.append(sample()); time.sleep(0.05)
            if label.endswith("*") and len(samples) % 10 == 5: smis.append(smi())
    t = threading.Thread(target=th); t.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        n += 20; torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    s_txt = " | ".join(smis[-2:])
    stop = True; t.join()
    us = 1e3 * e0.elapsed_time(e1) / n
    keys = sorted({k for s in samples for k in s})
    half = samples[len(samples) // 2:]
    agg = {k: (sum(s.get(k, 0) for s in half) / max(1, len(half))) for k in keys}
    print(f"{label:34s} {us:8.1f} us  " + "  ".join(f"{k}={v / 1e6:.1f}" for k, v in agg.items()), flush=True)
    if s_txt: print("   smi:", s_txt, flush=True)
g = setup("proj_ffn"); ops = g["ops"]
a, xr, pkp, z0, ln, f0, bq, qkv = (g[k] for k in ("a", "xr", "pkp", "z0", "ln", "f0", "bq", "qkv"))
run("block training, random *", g["fn"])
run("block no-grad, random *", lambda: ops.proj_ffn_ln_fwd(a, xr, pkp, z0, ln, f0, z0, ln, ln_b=ln, qkv_bias=bq, qkv=qkv, want_x1=False, want_hn=False))
sa, sx, sp = a.clone(), xr.clone(), pkp.clone()
a.zero_(); xr.zero_(); pkp.zero_()
run("block training, zero operands *", g["fn"])
a.copy_(sa); xr.copy_(sx); pkp.copy_(sp); del sa, sx, sp
del g; torch.cuda.empty_cache()
g = setup("dW1"); run("dW1 weight gradient (HBM-bound) *", g["fn"]); del g; torch.cuda.empty_cache()
g = setup("attn_fwd"); run("attention forward", g["fn"]); del g; torch.cuda.empty_cache()
g = setup("attn_bwd"); run("attention backward *", g["fn"]); del g; torch.cuda.empty_cache()
x = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)
run("fill 1 GiB (HBM write)", lambda: x.zero_())
time.sleep(1.0); print("idle again:", sample(), flush=True)
