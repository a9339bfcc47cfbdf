"""CPU: the committed bench line (profiles/r06_bench.json) carries the contract's fields, and every roofline fraction in it
follows from the committed rocprofv3 summaries of the same commands (profiles/r06_kernel_stats_streams1.json, r06_pmc_*.json)
within 10 % -- VERDICT r2 item 2, kept current every round ("my recomputation from profiles/r06_* lands within 10 % of every frac").
Round 6: `roofline.frac` is SURVEY 8(d)'s ruler (frac_hbm_8d); the VALU-issue share of the dominant kernel rides beside it as
frac_valu_issue, with the measured issue cycles per VALU instruction."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: json.load(open(os.path.join(ROOT, "profiles", n)))


def test_bench_line_has_the_contract_fields_and_consistent_rooflines():
    line = P("r06_bench.json")["default_run"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["dtype"] == "f64" and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and line["config"]["scenes_per_gpu"] == 256 and line["config"]["points"] == 50000
    assert abs(line["value"] - 256 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) / line["value"] < 1e-3
    kt1 = P("r06_kernel_stats_streams1.json")["kernels"]
    issue = P("r06_pmc_solve_issue.json")
    traffic = P("r06_pmc_traffic.json")["kernels"]
    within = lambda a, b, tol=0.10: abs(a - b) <= tol * abs(b)
    # dominant kernel, HBM view: algorithmic bytes / rocprof's single-stream duration / 8 TB/s
    h = line["roofline_hbm"]
    assert within(h["avg_launch_us"], kt1["mpc_solve_kernel<20>"]["avg_us"])
    assert within(h["frac"], h["alg_bytes_per_launch"] / (kt1["mpc_solve_kernel<20>"]["avg_us"] * 1e-6) / 8e12)
    assert within(h["traffic"], traffic["mpc_solve_kernel<20>"]["hbm_bytes_per_launch_x2"], 0.02)
    # the contract's roofline block: SURVEY 8(d)'s ruler for the whole step, and beside it what the counters name as the dominant kernel's bound
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert within(r["frac"], line["value"] * 670544 / 8e12, 1e-3) and r["frac"] == r["frac_hbm_8d"]
    assert within(r["achieved"], line["value"] * 670544 / 1e9, 1e-3)
    valu = issue["valu_instructions_per_wave_solve"]
    cpi = issue["frac_of_wave_time_issuing_valu"] * issue["wave_cycles_per_wave"] / valu      # measured, not the nominal 4
    assert 4.0 <= cpi <= 4.6 and within(r["valu_issue_cycles_per_instruction_measured"], cpi, 1e-3)
    solves_per_s = line["value"] * line["config"]["solves_per_step"]
    assert within(r["frac_valu_issue"], solves_per_s * valu * cpi / (256 * 4 * 2.4e9), 0.02)
    dk = r["dominant_kernel"]
    assert within(dk["frac_valu_issue_at_saturation_solves_only"], issue["valu_issue_util_at_saturation"], 0.02)
    assert r["frac_valu_issue"] < dk["frac_valu_issue_at_saturation_solves_only"] < 1.0
    assert within(dk["hbm_frac"], h["frac"], 1e-6) and dk["hbm_traffic_per_launch"] == h["traffic"] == r["traffic"]
    # the HBM-bound kernel: one launch builds both trees of the 256 scenes of every step of a gang
    b = line["roofline_kd_build"]
    G = line["config"]["steps_per_launch"]
    assert b["alg_bytes_per_launch"] == 28 * 256 * G * 55000 and line["config"]["scenes_per_launch"] == 256 * G
    assert within(b["avg_launch_us"], kt1["kd_build_kernel"]["avg_us"], 0.12)   # the event bracket holds ~8 us of dispatch
    assert within(b["frac"], b["alg_bytes_per_launch"] / (kt1["kd_build_kernel"]["avg_us"] * 1e-6) / 8e12, 0.12)
    assert within(b["traffic"], traffic["kd_build_kernel"]["hbm_bytes_per_launch_x2"], 0.02)
    w = line["roofline_whole_step"]
    assert within(w["frac"], line["value"] * 670544 / 8e12, 1e-3)
    # the achievable HBM rate (a measured device copy) beside the vendor peak, and every HBM fraction against both
    m = line["hbm_peak_measured_gbs"]
    assert line["hbm_peak_vendor_gbs"] == 8000.0 and 5800.0 < m < 8000.0          # round 6: the flat copy shape reaches the guide's 6.29 TB/s
    assert within(b["frac_of_measured_copy"], b["achieved"] / m, 1e-3) and within(h["frac_of_measured_copy"], h["achieved"] / m, 1e-2)
    assert within(w["frac_of_measured_copy"], line["value"] * 670544 / (m * 1e9), 1e-3)
    assert b["frac"] < b["frac_of_measured_copy"] <= 1.05
    # parity and baseline blocks
    fx = line["parity"]["fixtures"]
    assert fx["ok"] and all(fx[c]["converged"] == 64 and fx[c]["du_max"] <= 1e-3 for c in ("C1", "C2", "C5"))
    assert line["parity"]["timed_workload_vs_cpu_oracle"]["ok"]
    cb = line["cpu_baseline"]
    assert cb["cores"] >= 1 and cb["value"] > 0 and cb["kind"] == "port" and "sample" in cb
