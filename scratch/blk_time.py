"""Time the forward whole-block kernel (training instance) and the FFN backward dX kernel."""
import os, sys, torch, subprocess
T = os.environ.get("ONE_OP_T", "301568")
for op in ("proj_ffn", "ffn_bwd_dx"):
    src = open("scratch/one_op.py").read()
    ns = {}
    sys.argv = ["one_op.py", op, "."]
    exec(compile(src, "one_op", "exec"), ns)
    fn = ns["fn"]
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(op, f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us", "lib", os.environ.get("CHADAVIT_HIP_LIB", "default"))
