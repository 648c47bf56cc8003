"""Static audit: vmcnt retires in issue order and counts stores, so a `s_waitcnt vmcnt(n)` that has to cover a STORE older than the n newest
vector-memory operations exposes that store's round trip.  For every kernel of every csrc/*.hip (device asm, linear order -- loops
and branches are not followed, so this lists candidates, not proof): the waits that drain at least one store, with the line of the store.
usage: python scratch/r3/audit_store_drains.py [file.hip ...]"""
import re, subprocess, sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-S", "--cuda-device-only"]
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "chadavit_amd", "csrc", "*.hip")))
for f in files:
    out = "/tmp/asm/" + os.path.basename(f) + ".s"
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, f, "-o", out], check=True, capture_output=True)
    kern, ops, done, flagged = None, [], 0, []
    for ln, line in enumerate(open(out), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, ops, done, flagged = m.group(1), [], 0, []
            continue
        t = line.strip()
        if not kern or not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", op):
            ops.append(("S" if "store" in op else "L", ln))
        elif op == "s_waitcnt" and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            upto = max(done, len(ops) - n)
            st = [l for k, l in ops[done:upto] if k == "S"]
            if st:
                flagged.append((ln, n, len(st), st[0]))
            done = max(done, upto)
        elif op == "s_endpgm":
            if flagged:
                name = subprocess.run(["/usr/bin/c++filt", kern], capture_output=True, text=True).stdout.strip()[:110]
                print(f"{os.path.basename(f)}: {name}")
                print("   " + "  ".join(f"L{l}:vmcnt({n})<-{c}st" for l, n, c, s in flagged[:40]) + (" ..." if len(flagged) > 40 else ""))
            kern = None
