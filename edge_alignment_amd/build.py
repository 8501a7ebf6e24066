"""Compiles the HIP sources for gfx950 into lib/libea_hip.so with an explicit hipcc command
(in-tree, so the built library travels with the repository snapshot)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# (source, extra flags).  ea_kernels.hip -- the plain functor's evaluation kernels, the fold and the LM step -- is scheduled
# for instruction-level parallelism: same-box A/B (scripts/archive/ab_sched.sh, bit-identical results) C2 2.99 -> 2.91 us, 32 x C2
# fp64 tile order 25.0 -> 23.9 us, fp32 12.95 -> 12.65 us, C5 fp64 17.9 -> 17.2 us, the 1e5-point fp64 solve 166 -> 160 us.
# The variant functors' instantiations (ea_kernels_var.hip = the same file under -DEA_TU_VARIANT) lose 4-7 % under that
# strategy (a wave of occupancy in fp64) and keep the default one.
SOURCES = [("csrc/ea_kernels.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]), ("csrc/ea_kernels_var.hip", []),
           ("csrc/ea_preprocess.hip", []), ("csrc/ea_capi.hip", []), ("csrc/ea_comm.hip", [])]
HEADERS = ["csrc/ea_types.h", "csrc/ea_lm.h", "csrc/ea_spin.h", "csrc/ea_hip_dev.h", "../include/ea_hip.h"]
LIB = os.path.join(_HERE, "lib", "libea_hip.so")
# -amdgpu-kernarg-preload-count: the command processor hands the first 16 dwords of the kernel-argument segment to
# every wave in SGPRs, so a kernel does not start with a scalar load of its own pointers and a wait (gfx950; kernels
# keep a loading prologue for firmware without the feature).  Latency-bound launches: C2's evaluation 3.27 -> 3.00 us,
# the 1e5-point fp32 evaluation 2.90 -> 2.62 us (same-box A/B, profiles/LOG.md section 5b); batches unchanged.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_library(force=False, verbose=False, out=None, defines=()):
    """hipcc -c per source (they differ in flags), in parallel, then one hipcc -shared; objects under lib/obj/"""
    lib = out or LIB
    srcs = [os.path.join(_HERE, s) for s, _ in SOURCES]
    deps = srcs + [os.path.join(_HERE, h) for h in HEADERS]
    if (not force and os.path.exists(lib)
            and all(os.path.getmtime(lib) >= os.path.getmtime(d) for d in deps)):
        return lib
    obj_dir = os.path.join(os.path.dirname(lib), "obj" + ("_" + os.path.basename(lib) if out else ""))
    os.makedirs(obj_dir, exist_ok=True)
    cc = _hipcc()
    procs, objs = [], []
    for (src, extra), path in zip(SOURCES, srcs):
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        cmd = [cc] + FLAGS + list(extra) + ["-D" + d for d in defines] + ["-c", "-o", obj, path]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=_HERE)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-ldl"]  # (librccl itself is opened lazily: ea_comm.hip)
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=_HERE)
    return lib


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
