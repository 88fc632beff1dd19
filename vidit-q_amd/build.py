"""Build libviditq_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libviditq_hip.so")
SOURCES = ["api.hip", "rowquant.hip", "rowquant_fast.hip", "pack.hip", "gemm_i8.hip", "attention.hip", "sampler.hip", "fp_linear.hip", "clock_probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# sources whose MFMA accumulators stay in VGPRs: the AGPR form costs a v_accvgpr_read/write per softmax /
# epilogue operand (attention).  VQ_VGPR_FORM="a.hip,b.hip" overrides the set for experiments.
VGPR_FORM = {"attention.hip", "gemm_i8.hip", "clock_probe.hip"}


def _flags(src: str):
    env = os.environ.get("VQ_VGPR_FORM")
    vg = set(env.split(",")) if env is not None else VGPR_FORM
    extra = os.environ.get("VQ_EXTRA_HIPCC_FLAGS", "").split()       # A/B builds (e.g. -DVQ_GEMM_FP_DEQUANT=0 into out_dir)
    return FLAGS + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if src in vg else []) + extra


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(CSRC), "..", "include", "viditq.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


LAST_BUILD = {"compiled": 0, "seconds": 0.0, "lib": None}


def build(force: bool = False, verbose: bool = True, out_dir: str = None) -> str:
    """Compile every kernel source to objects (in parallel) and link the shared library.  ``out_dir``: build objects
    and library THERE instead of in-tree (always from source) - how smoke() proves a from-source build on the GPU box
    without touching the library the process has loaded."""
    import time
    lib = LIB if out_dir is None else os.path.join(out_dir, os.path.basename(LIB))
    if out_dir is not None:
        force = True
    if not force and not needs_build():
        return LIB
    t0 = time.perf_counter()
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build") if out_dir is None else os.path.join(out_dir, "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    n_compiled = 0
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_t):
            procs.append((src, obj, None))
            continue
        cmd = [hipcc, *_flags(src), "-c", sp, "-o", obj]
        if verbose:
            print("[viditq build]", " ".join(cmd), flush=True)
        n_compiled += 1
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for src, obj, p in procs:
        if p is not None:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode(errors="replace")))
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    if verbose:
        print("[viditq build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    LAST_BUILD.update(compiled=n_compiled, seconds=time.perf_counter() - t0, lib=lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
