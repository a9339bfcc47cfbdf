#!/usr/bin/env python
"""Turns the raw outputs of tools/prof/r06_profile.sh (gpurun_out/<tag>/...) into the summaries committed under profiles/:
  profiles/r06_bench.json                 the bench line (default run) + the 20-step line
  profiles/r06_kernel_stats.md            rocprofv3 --kernel-trace --stats of `bench.py --steps 256` (10 slots x 4 steps per launch in flight)
  profiles/r06_kernel_stats_streams1.md   the same with --streams 1: clean per-kernel durations
  profiles/r06_pmc_traffic.{json,md}      HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate passes)
  profiles/r06_pmc_solve_issue.{json,md}  what bounds mpc_solve_kernel: VALU issue / LDS / wait fractions (SQ counters)
usage: tools/prof/r06_collect.py <tag> [--profiles-only]"""
import collections
import csv
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06a"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")


def run(cmd):
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(n[k]) for c, v in acc[k].items()} for k in acc}, {k: len(v) for k, v in n.items()}


PROFILES_ONLY = "--profiles-only" in sys.argv   # on the GPU box, before the bench lines are taken: they quote these summaries
if not PROFILES_ONLY:
    bench = last_json(os.path.join(src, "bench.json"))
    b20 = last_json(os.path.join(src, "bench_20steps.json"))
    json.dump({"default_run": bench, "driver_style_20_steps": b20,
               "streams_1": last_json(os.path.join(src, "bench_streams1.json")),
               "ipm_cap_40_same_box": last_json(os.path.join(src, "bench_cap40.json")),
               "streams_20_gang_1_same_box": last_json(os.path.join(src, "bench_20x1.json")),
               "streams_20_gang_1_20_steps_same_box": last_json(os.path.join(src, "bench_20x1_20steps.json")),
               "torchrun_1_rank": last_json(os.path.join(src, "bench_torchrun_1rank.json")),
               "flight_workload": last_json(os.path.join(src, "bench_flight.json")),
               "tie_order_nanoflann": last_json(os.path.join(src, "bench_tie_order1.json")),
               "self_launched_1_gpu": last_json(os.path.join(src, "bench_self_launch.json")),
               "under_rocprofv3_kernel_trace": last_json(os.path.join(src, "bench_under_rocprof.json")),
               "flight_workload_10x2_same_box": last_json(os.path.join(src, "bench_flight_10x2.json")),
               "flight_workload_keyframes_3_10x2": last_json(os.path.join(src, "bench_flight_keyframes3.json")),
               "flight_yaml_config_keyframes_100_16x4": last_json(os.path.join(src, "bench_flight_yaml_keyframes100.json")),
               "flight_yaml_config_keyframes_100_12x4_same_box": last_json(os.path.join(src, "bench_flight_yaml_keyframes100_12x4.json")),
               "flight_yaml_config_single_frame_16x4_same_box": last_json(os.path.join(src, "bench_flight_yaml_single_frame.json")),
               "solve_budget_16_same_box": last_json(os.path.join(src, "bench_budget16.json")),
               "solve_budget_16_20_steps_same_box": last_json(os.path.join(src, "bench_budget16_20steps.json"))},
              open(os.path.join(dst, "r06_bench.json"), "w"), indent=1)
hdr = ("rocprofv3 --kernel-trace --stats -- python bench.py %s (tools/prof/r06_profile.sh, raw .db under gpurun_out/%s; "
       "summary by tools/rocprof_summary.py).  Torch kernels in the list generate the synthetic frames (setup, untimed).  NOTE: rocprofv3's "
       "`vgpr` column is HALF the compiler's count on this chip (allocation granules): the compiler's figures -- %s -- are in "
       "avoid_mpc_amd/kernel_resources.json and are the ones DESIGN.md's occupancy arguments use.\n\n")


def _vgpr_note():
    """The compiler's VGPR counts of the three hot kernels, read from the table the build writes (the note quoted
    stale numbers)."""
    try:
        res = json.load(open(os.path.join(ROOT, "avoid_mpc_amd", "kernel_resources.json")))
    except OSError:
        return "see kernel_resources.json"
    pick = lambda frag: next((v["vgprs"] for k, v in sorted(res.items()) if frag in k), None)
    return "%s for the solve (N = 20), %s for the index build, %s for the searches" % (
        pick("mpc_solve_kernelILi20"), pick("kd_build_kernel"), pick("step_knn_grid_kernel"))
for sub, name, args in (("kt20", "r06_kernel_stats.md", "--steps 256 --no-cpu-baseline --no-parity --steady-steps 0"),
                        ("kt1", "r06_kernel_stats_streams1.md", "--steps 64 --warmup 4 --streams 1 --no-cpu-baseline --no-parity --steady-steps 0")):
    md = run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), os.path.join(src, sub, "kt_results.db")])
    rows = [l for l in md.splitlines() if l.startswith("|")]
    keep = rows[:2] + [l for l in rows[2:] if any(t in l for t in ("mpc_", "kd_", "step_", "rocclr"))]
    open(os.path.join(dst, name), "w").write(hdr % (args, tag, _vgpr_note()) + "\n".join(keep) + "\n")
    if True:   # as json too: bench.py quotes the rocprof durations of the same commands beside its own HIP-event figures
        st = {}
        for l in keep[2:]:
            c = [x.strip() for x in l.strip("|").split("|")]
            st[c[0].strip("`")] = {"calls": int(c[1]), "avg_us": float(c[3]), "min_us": float(c[4]), "max_us": float(c[5])}
        json.dump({"_source": "rocprofv3 --kernel-trace --stats, bench.py %s (%s)" % (args, tag), "kernels": st},
                  open(os.path.join(dst, name[:-3] + ".json"), "w"), indent=1)
out = io.StringIO()
sys.stdout = out
sys.argv = ["pmc_traffic.py", os.path.join(src, "pmc_fetch", "f_counter_collection.csv"),
            os.path.join(src, "pmc_write", "w_counter_collection.csv"), os.path.join(dst, "r06_pmc_traffic.json.tmp")]
exec(open(os.path.join(ROOT, "tools", "pmc_traffic.py")).read())
sys.stdout = sys.__stdout__
tj = json.load(open(os.path.join(dst, "r06_pmc_traffic.json.tmp"))); os.remove(os.path.join(dst, "r06_pmc_traffic.json.tmp"))
cfg = last_json(os.path.join(src, "bench_streams1.json"))["config"]   # the command of the PMC passes
n, ne, S = cfg["points"], cfg["points"] // 10, cfg["scenes_per_gpu"]
tj = {k: v for k, v in tj.items() if any(t in k for t in ("mpc_", "kd_", "step_"))}
G = cfg.get("steps_per_launch", 1)
json.dump({"_meta": {"scenes_per_gpu": S, "gang": G, "scenes_per_launch": S * G, "points": n, "horizon": cfg["horizon"], "K": cfg["K"], "streams": 1,
                     "units": "FETCH_SIZE / WRITE_SIZE in KiB (MI355X_MICROARCH.md, HBM section); x2 = gfx950 wide-read correction",
                     "algorithmic_bytes_per_launch": {"kd_build_kernel (one launch: obstacle + edge index of every frame of the gang)": 28 * S * G * (n + ne)}},
           "kernels": tj}, open(os.path.join(dst, "r06_pmc_traffic.json"), "w"), indent=1)
open(os.path.join(dst, "r06_pmc_traffic.md"), "w").write(
    "HBM traffic per launch (%d scenes = %d steps of %d), `bench.py --steps 8 --warmup 2 --streams 1` under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
    "(separate passes; raw CSVs under gpurun_out/%s).\n\n" % (S * G, G, S, tag) + "\n".join(l for l in out.getvalue().splitlines()
                                                                             if l.startswith("|") and ("launch" in l or "---" in l or any(t in l for t in ("mpc_", "kd_", "step_")))) + "\n")
ca, na = counters(os.path.join(src, "pmc_sq_a", "a_counter_collection.csv"))
cb, nb = counters(os.path.join(src, "pmc_sq_b", "b_counter_collection.csv"))
k = [x for x in ca if x.startswith("mpc_solve_kernel")][0]
a, b = ca[k], cb[k]
waves = b["SQ_WAVES"]
issue = {
    "kernel": k, "source": f"rocprofv3 --pmc (two passes of 8 SQ counters), bench.py --streams 1 --gang 1, gpurun_out/{tag}/pmc_sq_a|b; per-launch averages over {na[k]} launches",
    "waves_per_launch": waves,
    "valu_instructions_per_wave_solve": b["SQ_INSTS_VALU"] / waves, "lds_instructions_per_wave_solve": b["SQ_INSTS_LDS"] / waves,
    "salu_instructions_per_wave_solve": b["SQ_INSTS_SALU"] / waves,
    "wave_cycles_per_wave": 4 * a["SQ_WAVE_CYCLES"] / waves,
    "frac_of_wave_time_issuing_valu": a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"],
    "frac_of_wave_time_issuing_lds": a["SQ_ACTIVE_INST_LDS"] / a["SQ_WAVE_CYCLES"],
    "frac_of_wave_time_issuing_anything": a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"],
    "frac_of_wave_time_waiting (s_waitcnt / barrier)": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
    "lds_array_busy_cycles_per_wave": b["SQ_LDS_IDX_ACTIVE"] / waves,
    "lds_bank_conflict_frac_of_lds_busy": b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"],
    "lds_busy_frac_of_wave_time": b["SQ_LDS_IDX_ACTIVE"] / (4 * a["SQ_WAVE_CYCLES"]),
}
# saturated solve-only rate (tools/experiments/ms_parts.py: 16 streams x 256 scenes, 8 waves per CU): solves per us, chip
SAT = float(os.environ.get("AMK_SAT_SOLVES_PER_US", "0") or 0)
if not SAT:   # the same box's ms_parts.py line: "solve-only: ... -> X solves/us"
    import re
    SAT = float(re.search(r"-> ([0-9.]+) solves/us", open(os.path.join(src, "ms_parts.txt")).read()).group(1))
clk_per_solve_per_cu = 256 * 2400.0 / SAT          # 256 CUs, 2.4 GHz
issue["saturated_solves_per_us (ms_parts.py, 8 waves per CU)"] = SAT
# issue cycles a VALU instruction of this kernel holds its SIMD for, measured (4.27: the fp64 share; rounds 2-5 assumed 4)
issue["valu_issue_cycles_per_instruction"] = issue["frac_of_wave_time_issuing_valu"] * issue["wave_cycles_per_wave"] / issue["valu_instructions_per_wave_solve"]
issue["valu_issue_util_at_saturation"] = issue["valu_instructions_per_wave_solve"] * issue["valu_issue_cycles_per_instruction"] / 4 / clk_per_solve_per_cu
issue["lds_array_util_at_saturation (conflict level of the lone wave)"] = issue["lds_array_busy_cycles_per_wave"] / clk_per_solve_per_cu
issue["wave_time_stretch_at_saturation"] = 8 * clk_per_solve_per_cu / issue["wave_cycles_per_wave"]
issue["bound"] = ("dependent-operation latency at limited occupancy: a wave alone issues VALU %.0f %% of its time, LDS %.0f %%, "
                  "and waits %.0f %%; the LDS (17.8 KB per scene; 180 VGPRs per wave) allows 8 waves per CU = 2 per SIMD, and at that occupancy "
                  "the VALU pipes are %.0f %% busy and the LDS array %.0f %% -- neither is saturated, a wave just runs %.2fx "
                  "slower than alone because its dependent fp64 operations and LDS round trips interleave with one other "
                  "wave's; HBM and MFMA are not involved" % (
                      100 * issue["frac_of_wave_time_issuing_valu"], 100 * issue["frac_of_wave_time_issuing_lds"],
                      100 * issue["frac_of_wave_time_waiting (s_waitcnt / barrier)"],
                      100 * issue["valu_issue_util_at_saturation"],
                      100 * issue["lds_array_util_at_saturation (conflict level of the lone wave)"],
                      issue["wave_time_stretch_at_saturation"]))
json.dump(issue, open(os.path.join(dst, "r06_pmc_solve_issue.json"), "w"), indent=1)
with open(os.path.join(dst, "r06_pmc_solve_issue.md"), "w") as f:
    f.write("What bounds `mpc_solve_kernel` (SQ counters, `bench.py --streams 1 --gang 1`: 256-scene launches, one wave per CU).\n\n| quantity | value |\n|---|---|\n")
    for kk, v in issue.items():
        f.write(f"| {kk} | {v if isinstance(v, str) else round(v, 4)} |\n")
    f.write("\nRaw per-launch counter averages:\n\n| counter | value |\n|---|---|\n")
    for c, v in sorted({**a, **b}.items()):
        f.write(f"| {c} | {v:.0f} |\n")
# round 4 extras: the closed-loop workload's kernel mix, the flight parity report of tests/test_flight_gpu.py, the RCCL-presence and
# exact-tree cost measurements (plain text, as the tools print them)
import shutil
if os.path.exists(os.path.join(src, "kt_flight", "kt_results.db")):
    md = run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), os.path.join(src, "kt_flight", "kt_results.db")])
    rows = [l for l in md.splitlines() if l.startswith("|")]
    keep = rows[:2] + [l for l in rows[2:] if any(t in l for t in ("mpc_", "kd_", "step_", "pipeline_", "Cijk", "rocclr"))]
    open(os.path.join(dst, "r06_kernel_stats_flight.md"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python bench.py --workload flight --periods 24 --no-parity (10 slots x gang 4, TASK mode; raw .db "
        "under gpurun_out/%s).  Cijk_* is the vehicle's addmm (torch, part of the workload).  rocprofv3's vgpr column is half the compiler's.\n\n" % tag
        + "\n".join(keep) + "\n")
for sub, out, what in (("kt_flight_kf3", "r06_kernel_stats_flight_keyframes3_streams1.md",
                        "--workload flight --keyframes 3 --streams 1 --gang 2 (50 k-point frames, 512-scene launches, one stream: clean kernel durations)"),
                       ("kt_flight_yaml", "r06_kernel_stats_flight_yaml_keyframes100_streams1.md",
                        "--workload flight --config yaml --keyframes 100 --streams 1 (3072-point frames, N = 30, K = 3, max_frame_count 100, 1024-scene launches, one stream)")):
    dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(src, sub)) for f in fs if f.endswith(".db")]
    if dbs:
        md = run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), dbs[0]])
        rows = [l for l in md.splitlines() if l.startswith("|")]
        keep = rows[:2] + [l for l in rows[2:] if any(t in l for t in ("mpc_", "kd_", "kf_", "step_", "pipeline_", "Cijk", "rocclr"))]
        open(os.path.join(dst, out), "w").write(
            "rocprofv3 --kernel-trace --stats -- python bench.py %s (the closed loop with the keyframe map; raw .db under gpurun_out/%s).  "
            "rocprofv3's vgpr column is half the compiler's.\n\n" % (what, tag) + "\n".join(keep) + "\n")
for name, out in (("flight_c2_gpu_vs_oracle.json", "r06_flight_c2_gpu_vs_oracle.json"), ("rccl_presence.txt", "r06_rccl_presence.txt"),
                  ("exact_mode_cost.txt", "r06_exact_mode_cost.txt"), ("flight_tests.txt", "r06_flight_tests.txt"),
                  ("burst_timeline_10x4.txt", "r06_burst_timeline_10x4.txt")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out))
print("profiles written:", sorted(x for x in os.listdir(dst) if x.startswith("r06")))
