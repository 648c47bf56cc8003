"""Round 4, "prove the new assertions bite": a second build of the library with round 3's QKV-postlogue bug put back (the barrier
wait of the whole-block kernel's no-grad postlogue is `vmcnt(6)` on EVERY step again, so the last step reads a weight block that
is still landing: a wrong V third in 1-5 % of the teacher / local-crop rows at bench size).  Nothing of the product changes: the two
units are compiled from a patched COPY of ffn_fused.hip into scratch/sidebuild/postlogue_bug/ (outside the package since round 5) and
linked with the product's other objects.

    python scratch/r4/postlogue_bug_build.py       ->  scratch/sidebuild/postlogue_bug/libchadavit_hip_postlogue_bug.so
    CHADAVIT_HIP_LIB=<that> CHADAVIT_ALLOW_FOREIGN_LIB=1 python -m pytest tests/test_model_gpu.py -m gpu -k "bench_scale_replicated or vs_golden_and_oracle"
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from chadavit_amd import build as B  # noqa: E402

B.build(verbose=False)               # the product objects
out = os.path.join(B.SIDE, "postlogue_bug")
src = os.path.join(out, "csrc")
os.makedirs(src, exist_ok=True)
for f in ("common.h", "ffn_fused.hip", "ffn_fused_d384.hip"):
    shutil.copy(os.path.join(B.CSRC, f), os.path.join(src, f))
c = os.path.join(src, "common.h")
txt = open(c).read().replace('"../../include/chadavit_hip.h"', '"chadavit_hip.h"')
open(c, "w").write(txt)
p = os.path.join(src, "ffn_fused.hip")
s = open(p).read()
good = "if (NST == 3 && q + LA - 1 < 3 * NPB) asm volatile(\"s_waitcnt vmcnt(6) lgkmcnt(0)"
assert s.count(good) == 1
open(p, "w").write(s.replace(good, "if (NST == 3) asm volatile(\"s_waitcnt vmcnt(6) lgkmcnt(0)"))
objs = []
for name in B.SOURCES:
    o = os.path.join(B.HERE, "build", name.replace(".hip", ".o"))
    if name.startswith("ffn_fused"):
        o = os.path.join(out, name.replace(".hip", ".o"))
        subprocess.run([B._hipcc(), *B.FLAGS, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(src, name), "-o", o], check=True,
                       stderr=subprocess.DEVNULL)
    objs.append(o)
lib = os.path.join(out, "libchadavit_hip_postlogue_bug.so")
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
print(lib)
