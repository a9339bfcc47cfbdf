"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares.
No compute calls (there is no GPU here and no CPU fallback to call)."""
import os
import re

import pytest

from avoid_mpc_amd import build as amk_build
from avoid_mpc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    amk_build.build()
    return capi.load()


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "avoid_mpc_amd.h")).read()
    declared = set(re.findall(r"\b(amk_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS)


def test_exports_every_declared_symbol(lib):
    assert capi.missing_symbols() == []
    assert lib.amk_version() >= 100
    assert lib.amk_status_string(0) == b"ok"


def test_casadi_plugin_header_and_exports(lib):
    """include/avoid_mpc_amd/casadi_plugin.h: every function it declares (7 x (entry + 15 helpers) + 2) is exported."""
    hdr = open(os.path.join(ROOT, "include", "avoid_mpc_amd", "casadi_plugin.h")).read()
    funcs = re.findall(r"^AMK_CASADI_DECLARE\((\w+)\)", hdr, re.M)
    assert funcs == capi.PLUGIN_FUNCTIONS
    macro = hdr[hdr.index("#define AMK_CASADI_DECLARE"):hdr.index("AMK_CASADI_DECLARE(nlp)")]
    helpers = re.findall(r"NAME##_(\w+)\(", macro)
    assert helpers == capi.PLUGIN_HELPERS
    extra = set(re.findall(r"\b(amk_plugin_\w+)\s*\(", hdr))
    assert extra == {"amk_plugin_configure", "amk_plugin_dims"}
    missing = [s for s in capi.PLUGIN_SYMBOLS if not hasattr(lib, s)]
    assert missing == []


def test_no_cpu_fallback(lib):
    """Without a GPU every constructor must fail loudly instead of computing on the host."""
    import ctypes as C
    if lib.amk_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.amk_kd_create(1, 100, C.byref(h)) == 3      # AMK_ERR_NO_DEVICE
    assert not h.value
    assert lib.amk_mpc_create(0.66, 0.033, 8, 1, C.byref(h)) == 3
    assert not h.value
    cfg = capi.PipelineConfig(2, 4, 100, 10, 0.33, 0.033, 3, 0, 0, capi.StepParams(10.0, 0.2, 3, 0))
    assert lib.amk_pipeline_create(C.byref(cfg), C.byref(h)) == 3 and not h.value
    assert lib.amk_shard_create(b"\0" * 128, 0, 1, C.byref(h)) == 3 and not h.value
    first, count = C.c_int(), C.c_int()   # the partition itself is host arithmetic
    assert lib.amk_shard_scene_range(1, 3, 8, C.byref(first), C.byref(count)) == 0 and (first.value, count.value) == (3, 3)
    assert lib.amk_shard_scene_range(2, 3, 8, C.byref(first), C.byref(count)) == 0 and (first.value, count.value) == (6, 2)


def test_kfmap_pool_bytes_is_host_arithmetic(lib):
    """amk_kfmap_pool_bytes: what a map's pools reserve -- ~30 B per obstacle point and slot + the edge pool, (max_frame_count + 2) slots per
    scene; no device needed."""
    import ctypes as C
    b = C.c_longlong()
    assert lib.amk_kfmap_pool_bytes(1, 3072, 3072, 100, C.byref(b)) == 0
    per_robot_yaml = b.value
    assert 10e6 < per_robot_yaml < 40e6                                   # the yaml's map: tens of MB per robot
    assert lib.amk_kfmap_pool_bytes(512, 50000, 5000, 100, C.byref(b)) == 0
    assert 70e9 < b.value < 110e9                                         # the advisor's example (86 GB): two pipeline slots of it already exceed half a 288 GB device
    assert lib.amk_kfmap_pool_bytes(512, 50000, 5000, 3, C.byref(b)) == 0 and 4e9 < b.value < 9e9
    small = C.c_longlong(); assert lib.amk_kfmap_pool_bytes(256, 50000, 5000, 3, C.byref(small)) == 0
    assert abs(b.value - 2 * small.value) < 1e-3 * b.value                # linear in the scenes
    assert lib.amk_kfmap_pool_bytes(0, 100, 100, 3, C.byref(b)) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_kfmap_pool_bytes(4, 100, 100, 0, C.byref(b)) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_kfmap_pool_bytes(4, 100, 100, 3, None) == capi.AMK_ERR_INVALID_ARG


def test_product_never_touches_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "avoid_mpc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"liboracle|_oracle|oracle/|mpc_oracle|kd_oracle", txt):
                    bad.append(f)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            txt = open(os.path.join(dirpath, f), errors="ignore").read()
            if re.search(r"liboracle|kdo_|mpco_", txt):
                bad.append(f)
    assert bad == []


def test_default_path_kernels_use_no_scratch_memory(lib):
    """A kernel that may touch scratch makes every hardware queue reserve it; with the bench's 20 streams / 32 queues a
    scratch-using kernel on the default path exhausted the runtime's resources (HSA_STATUS_ERROR_OUT_OF_RESOURCES).  No
    kernel of the library uses scratch (the traversal stacks of the nanoflann tie-order mode live in LDS); the solve
    must also stay within 256 VGPRs (2 waves per SIMD).  Read from the compiler's own resource report, written next to
    the library by the build."""
    import json
    table = json.load(open(amk_build.RES))
    assert len(table) >= 30
    for name, r in table.items():   # (round 2's thread-per-query traversal kernels kept their stack in scratch; they are wave-per-query now)
        assert r["scratch_bytes_per_lane"] == 0, (name, r)
    solve = [r for n, r in table.items() if "mpc_solve_kernel" in n]
    assert solve and all(r["vgprs"] <= 256 for r in solve)
    # a 512-thread index-build block is two waves per SIMD: it must fit beside ONE fp64 solve wave of the bench's horizon
    # in the 512-register file of a SIMD (with 24 points in flight per thread it did not, and the step rate fell by a third)
    build = [r for n, r in table.items() if n.startswith("_Z15kd_build_kernel")][0]
    solve20 = [r for n, r in table.items() if "mpc_solve_kernelILi20" in n][0]
    g = lambda v: (v + 7) // 8 * 8        # allocation granule
    assert 2 * g(build["vgprs"]) + g(solve20["vgprs"]) <= 512, (build, solve20)


def test_struct_layouts_of_the_header_equal_the_ctypes_mirrors(tmp_path):
    """The structs that cross the C ABI by value or by pointer (amk_pipeline_config / _frame, amk_task_params, amk_depth_params,
    amk_step_params, amk_frame_camera) are mirrored by hand in avoid_mpc_amd/capi.py: a C program compiled against the header
    prints sizeof and every offsetof, and they must equal the ctypes classes' -- a field added on one side only would
    silently shift every argument behind it."""
    import ctypes as C
    import subprocess
    pairs = {"amk_step_params": capi.StepParams, "amk_task_params": capi.TaskParams, "amk_depth_params": capi.DepthParams,
             "amk_frame_camera": capi.FrameCamera, "amk_pipeline_config": capi.PipelineConfig, "amk_pipeline_frame": capi.PipelineFrame}
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "avoid_mpc_amd.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        src.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _t in cls._fields_:
            src.append(f'printf(" {fname}=%zu", offsetof({cname}, {fname}));')
        src.append('printf("\\n");')
    src += ['return 0; }']
    cfile, exe = tmp_path / "layout.c", tmp_path / "layout"
    cfile.write_text("\n".join(src))
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)])   # plain C: the header is a C header
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.splitlines():
        tok = line.split()
        cls = pairs[tok[0]]
        assert int(tok[1]) == C.sizeof(cls), (tok[0], tok[1], C.sizeof(cls))
        for kv in tok[2:]:
            k, v = kv.split("=")
            assert int(v) == getattr(cls, k).offset, (tok[0], k, v, getattr(cls, k).offset)
