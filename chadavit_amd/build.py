"""Build libchadavit_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain
C-ABI shared object (include/chadavit_hip.h) loaded through ctypes by chadavit_amd._lib."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libchadavit_hip.so")
SOURCES = ["gemm_nt.hip", "ffn_fused.hip", "ffn_fused_d384.hip", "gemm_tn.hip", "layernorm.hip", "attention.hip", "attention_m32.hip", "attention_cls.hip", "tokenizer.hip", "dino_ops.hip", "gemm_mx8.hip", "augment.hip", "host_draw.hip"]
# -packed-fp32-ops: no v_pk_{mul,add,fma}_f32.  On gfx950 a packed f32 op costs the VALU port 8 cycles -- the same as the two scalar ops it
# replaces -- and beside MFMAs it stalls the matrix pipe on top (scratch/r3/coissue*.hip: 2 v_pk_fma_f32 per 32x32x16 MFMA = 60 cycles per
# MFMA against 36 with 2 v_fma_f32).  hipcc forms them from every float4 / float2 expression.  Same-box A/B of the whole step: +1.2 %
# images/s (block kernels -1.5 ... -3 %); profiles/r03a_nopk_ab.txt.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


SIDE = os.path.join(ROOT, "scratch", "sidebuild")  # side builds (A/B variants, deliberately broken libraries): OUTSIDE the package


def side_lib(tag: str) -> str:
    return os.path.join(SIDE, tag, f"libchadavit_hip_{tag}.so")


def build(force: bool = False, verbose: bool = True, extra_flags=(), side: str | None = None, csrc: str = CSRC, units=None) -> str:
    """side = a tag: a second build of the same ABI (same-box A/B of compiler options / kernel variants) under scratch/sidebuild/<tag>/,
    never inside the package; load it with CHADAVIT_HIP_LIB=<path> CHADAVIT_ALLOW_FOREIGN_LIB=1.  csrc: another source directory (a
    checkout of an older commit) for such a build.  units: the sources `extra_flags` apply to (a side build of ONE kernel variant): the other
    objects are the product build's own (built first if stale)."""
    hipcc = _hipcc()
    lib_path = LIB if side is None else side_lib(side)
    objdir = os.path.join(HERE, "build") if side is None else os.path.join(SIDE, side)
    CSRC_ = csrc
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC_, "common.h"), os.path.join(ROOT, "include", "chadavit_hip.h")]
    jobs = []
    objs = []
    if units is not None and side is not None:
        build(verbose=False)   # the product objects the other units are taken from
    for src in SOURCES:
        sp = os.path.join(CSRC_, src)
        if units is not None and side is not None and src not in units:
            objs.append(os.path.join(HERE, "build", src.replace(".hip", ".o")))
            continue
        op = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(op)
        deps = [sp] + headers + ([os.path.join(CSRC_, "ffn_fused.hip")] if src == "ffn_fused_d384.hip" else [])
        if force or _stale(op, deps):
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        # (the HOST half of each compile does not know the device feature named in FLAGS and says so once per pass: not a finding)
        err = "\n".join(ln for ln in r.stderr.splitlines() if "is not a recognized feature for this target" not in ln).strip()
        if verbose and err:
            print(err, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib_path, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
    return lib_path


AB_TAG, AB_FLAGS = "ab", ("-DCHADA_AB_SWITCHES=1",)


def build_ab(verbose: bool = False) -> str:
    """The A/B side build: the same sources with -DCHADA_AB_SWITCHES=1 -- the superseded kernels (16x16x32 LDS-DMA forward at dh 96 / 192 / 384, its
    row-major-stage form, the fragment-major dQ kernel at dh 384) and the environment switches that select them or the weight-gradient GEMM's older
    tilings.  Not part of the product: lives under scratch/sidebuild/ab/, loaded only with CHADAVIT_HIP_LIB=<path> CHADAVIT_ALLOW_FOREIGN_LIB=1 (the
    bit-identity tests and same-box A/B runs do that in a child process)."""
    return build(verbose=verbose, side=AB_TAG, extra_flags=AB_FLAGS)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--ab" in sys.argv:
        print(build_ab(verbose=True))
