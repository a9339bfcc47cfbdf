"""Builds avoid_mpc_amd/libavoid_mpc_amd.so (HIP, gfx950 only) in-tree with hipcc.

`python -m avoid_mpc_amd.build` or __graft_entry__.build().  hipcc cross-compiles without a GPU.
The built .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libavoid_mpc_amd.so")
ARCH = "gfx950"

FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "avoid_mpc_amd.h"))
    return hdrs


def build(force=False, verbose=False, extra_flags=()):
    if os.environ.get("AMK_SOLVE_TRACE"):
        extra_flags = tuple(extra_flags) + ("-DAMK_SOLVE_TRACE",)
    if os.environ.get("AMK_HIPCC_FLAGS"):  # experiments only (e.g. -DAMK_SOLVE_WAVES=4)
        extra_flags = tuple(extra_flags) + tuple(os.environ["AMK_HIPCC_FLAGS"].split())
    os.makedirs(OBJ, exist_ok=True)
    dep_mtime = max(os.path.getmtime(h) for h in _deps())
    objs, relink = [], force or not os.path.exists(LIB)
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_mtime)
        if stale:
            cmd = [hipcc(), *FLAGS, *extra_flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            relink = True
    if relink:
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
