"""Build tools/lab/libviditq_lab.so: the retired GEMM generations, profiling ablations and issue-rate probes.
Measurement equipment - NOT part of the product library (vidit-q_amd/csrc/libviditq_hip.so) and never loaded by it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libviditq_lab.so")
SOURCES = ["gemm_lab.hip", "probe.hip", "gemm_4w.hip", "gemm_loader.hip", "gemm_sp.hip", "gemm_persist6.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm",
         "-amdgpu-mfma-vgpr-form", "-I", HERE, "-I", os.path.join(HERE, "..", "..", "vidit-q_amd", "csrc")]


def build(force=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES]
    csrc = os.path.join(HERE, "..", "..", "vidit-q_amd", "csrc")
    deps = srcs + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) < os.path.getmtime(LIB) for d in deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for s in srcs:
        o = s.replace(".hip", ".o")
        objs.append(o)
        procs.append(subprocess.Popen([hipcc, *FLAGS, "-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
