#!/bin/bash
# round-4 closing run after the augmentation-kernel rewrite: full GPU suite, smoke, the driver's command
O=gpurun_out/final_r04u; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python - <<'PY'
import subprocess, time, sys
t0 = time.time()
r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], capture_output=True, text=True)
open("gpurun_out/final_r04u/bench_cfg2.json", "w").write(r.stdout)
open("gpurun_out/final_r04u/bench_cfg2.err", "w").write(r.stderr)
print("driver command wall seconds:", round(time.time() - t0, 1), "rc", r.returncode)
open("gpurun_out/final_r04u/driver_run_s.txt", "w").write(f"{time.time() - t0:.1f}\n")
PY
tail -1 $O/bench_cfg2.json | cut -c1-300
