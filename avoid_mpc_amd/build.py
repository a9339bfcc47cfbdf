"""Builds avoid_mpc_amd/libavoid_mpc_amd.so (HIP, gfx950 only) in-tree with hipcc.

`python -m avoid_mpc_amd.build` or __graft_entry__.build().  hipcc cross-compiles without a GPU.
The built .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libavoid_mpc_amd.so")
RES = os.path.join(HERE, "kernel_resources.json")   # per-kernel VGPRs / scratch / LDS as the compiler reports them
ARCH = "gfx950"

# -amdgpu-sched-strategy=max-ilp: every hot kernel here is bound by dependent-operation latency at a fixed occupancy (the
# solve's 2 waves per SIMD are set by LDS), so the machine scheduler should favour ILP over register economy; same-box A/B
# of the bench: +1.2 % (tools/experiments/ab_solve.sh).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


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
    flags = list(FLAGS)
    if os.environ.get("AMK_SCHED_STRATEGY"):  # experiments only: another machine-scheduler strategy than max-ilp
        flags[-1] = "-amdgpu-sched-strategy=" + os.environ["AMK_SCHED_STRATEGY"]
    os.makedirs(OBJ, exist_ok=True)
    dep_mtime = max(os.path.getmtime(h) for h in _deps())
    objs, relink = [], force or not os.path.exists(LIB)
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), dep_mtime)
        if stale:
            cmd = [hipcc(), *flags, *extra_flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            _record_resources(r.stderr, obj + ".res.json")
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            relink = True
    if relink:
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if relink or not os.path.exists(RES):
        table = {}
        for obj in objs:
            if os.path.exists(obj + ".res.json"):
                table.update(json.load(open(obj + ".res.json")))
        json.dump(table, open(RES, "w"), indent=1, sort_keys=True)
    return LIB


_REMARK = re.compile(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|"
                     r"Occupancy \[waves/SIMD\]): (\S+)")


def _record_resources(stderr, path):
    """Splits hipcc's stderr into the kernel-resource remarks (-> json: kernel -> {vgprs, scratch, ...}) and the rest
    (warnings / errors, echoed)."""
    table, cur = {}, None
    diag = re.compile(r"^\S+:\d+:\d+: (remark|warning|error|note|fatal error): ")
    pending, drop = [], False
    for line in stderr.splitlines():
        if line.startswith("In file included from"):
            pending.append(line)
            continue
        d = diag.match(line)
        if not d:
            if not drop:   # source snippet / summary lines of a diagnostic that is shown
                print(line, file=sys.stderr)
            continue
        drop = d.group(1) == "remark"
        if not drop:
            print("\n".join(pending + [line]), file=sys.stderr)
            pending = []
            continue
        pending = []
        m = _REMARK.search(line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = table.setdefault(val, {})
        elif cur is not None:
            cur[{"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
                 "LDS Size [bytes/block]": "lds_bytes", "Occupancy [waves/SIMD]": "occupancy"}[key]] = int(val)
    json.dump(table, open(path, "w"))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
