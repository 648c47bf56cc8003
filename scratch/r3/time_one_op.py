"""HIP-event timing of one hot entry point (same set-ups as scratch/one_op.py): python scratch/r3/time_one_op.py <op> [<op> ...]"""
import sys, os, runpy, torch
T = os.environ.get("ONE_OP_T", "603136")
os.environ["ONE_OP_T"] = T
for name in sys.argv[1:]:
    sys.argv_saved = sys.argv
    g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "one_op.py"), init_globals={"__name__": "one_op"}, run_name="one_op") if False else None
    # one_op.py reads sys.argv[1:3]; run it in-process and take its `fn`
    import importlib.util
    argv = sys.argv
    sys.argv = ["one_op.py", name, "."]
    spec = importlib.util.spec_from_file_location("one_op_" + name, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "one_op.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    sys.argv = argv
    fn = m.fn
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{name} T={T}: {e0.elapsed_time(e1) * 100:.1f} us", flush=True)
    del m, fn
    torch.cuda.empty_cache()
