#!/bin/bash
# how much of a cfg2 step is the GPU idle between kernels?  rocprofv3 --kernel-trace of a 3-step run, gaps summed over the timed steps
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6_gaps; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads --no-full-width-leg --no-launch-profile --data resident --serial > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import csv, glob, os, json
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r6_gaps"
f=glob.glob(O+"/**/t_kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
d=json.loads(open(O+"/bench.json").read().strip().splitlines()[-1])
step_ns=d["ms_per_step"]*1e6
# the timed region = the last 3 steps' worth of time before the last kernel... take the window [end - 3*step, end] of the main stream of kernels
end=max(r[1] for r in rows if "ffn_fwd_kernel" in r[2]); start=end-3*step_ns
win=[r for r in rows if r[0]>=start and r[1]<=end+1e6]
busy=0; gap=0; last=None; biggest=[]
for s,e,n in win:
    if last is not None and s>last:
        gap+=s-last; biggest.append((s-last,n))
    busy+=max(0,e-(last if last and last>s else s)); last=max(last or 0,e)
biggest.sort(reverse=True)
print("window ms", (end-start)/1e6, "kernels", len(win), "busy ms/step", busy/3e6, "gap ms/step", gap/3e6)
print("largest gaps (us, before kernel):", [(round(g/1e3,1), n[-40:]) for g,n in biggest[:12]])
import collections
c=collections.Counter()
for g,n in biggest: c["<5us" if g<5e3 else "<20us" if g<2e4 else ">=20us"]+=g
print({k: round(v/3e6,3) for k,v in c.items()}, "ms/step by gap size")
PY
find $O -name "*kernel_trace*" -delete
