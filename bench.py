#!/usr/bin/env python
"""bench.py -- MPC steps/sec of the Avoid-MPC hot path on MI355X (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic scenes resident in HBM: for every
scene a fresh depth frame (build the obstacle + edge KD indices) and one control step (<= 3 outer
passes of {dual KD queries, pack P, interior-point solve <= 10 iterations}, zero warm start),
SURVEY.md §8(d).  value = scenes processed by all ranks / wall time (max over ranks).
Workload: BASELINE.json configs[1] (50k-point cloud, N = 20, K = 8) batched as configs[2]
(256 scenes per GPU); scenes are independent, so N GPUs run N x 256 scenes (weak scaling) and the
only collective is the gather of the controls (RCCL all_gather of 4 doubles per scene).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
KCLASS = ["kd_compact", "knn_obstacle", "knn_edge", "plan", "pack", "mpc_solve", "begin", "kd_grid_build"]


def alg_bytes_per_step(n, ne, N, K):
    """every input read once, every output written once, per scene-step."""
    nx = 10 + 14 * N
    plen = 54 + 10 * N + 3 * K * N
    return 12 * n + 12 * ne + 8 * plen + 8 * nx + 8 * nx + 32


def solve_alg_bytes(N, K):
    """Algorithmic HBM bytes of ONE launch of the dominant kernel (mpc_solve_kernel) per scene:
    reads vecRefStates (20+10N+3KN doubles), the 192-double parameter block and the warm start
    (nx); writes the solution (nx), u (4), x0Array (14N) and the refilled reference path (10N)."""
    nx = 10 + 14 * N
    return 8 * ((20 + 10 * N + 3 * K * N) + 192 + nx + nx + 4 + 14 * N + 10 * N)


# ------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle's restatement of the same step, one scene per core
# ------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    wid, n, T, K, budget_s, seed0 = args
    import numpy as np  # noqa: F401
    from avoid_mpc_amd import synth
    from tests import _oracle
    prm = synth.MpcParams(T=T, K=K)
    done, t_build, t_step = 0, 0.0, 0.0
    t_end = time.perf_counter() + budget_s
    i = 0
    while True:
        sc = synth.make_scene(n, seed0 + 1000 * wid + i, prm)          # generation not timed
        sq = _oracle.scene_state_quads(sc, prm)
        t0 = time.perf_counter()
        kd, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
        t1 = time.perf_counter()
        mpc = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mpc.configure(prm)
        _oracle.step_oracle(kd, ke, mpc, prm, sq, sc["pos"][0], sc["ref_path"].copy())
        t2 = time.perf_counter()
        t_build += t1 - t0; t_step += t2 - t1
        done += 1; i += 1
        kd.close(); ke.close(); mpc.close()
        if time.perf_counter() >= t_end and done >= 4:
            break
    return done, t_build, t_step


def cpu_baseline(n, T, K, budget_s=6.0):
    import multiprocessing as mp
    from tests import _oracle
    _oracle.build_oracle()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    workers = max(1, min(cores, 128))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(w, n, T, K, budget_s, 7000) for w in range(workers)])
    wall = time.perf_counter() - t0
    scenes = sum(r[0] for r in res)
    busy = sum(r[1] + r[2] for r in res)
    per_scene_ms = 1e3 * busy / scenes
    build_ms = 1e3 * sum(r[1] for r in res) / scenes
    # throughput of `workers` cores each running scenes back to back (generation/startup excluded)
    value = workers / (busy / scenes)
    return {"value": round(value, 2), "unit": "MPC steps/s", "cores": workers, "kind": "port",
            "sample": f"{scenes} scenes of the same workload (n={n}, N={int(T / 0.033)}, K={K}), one per core at a "
                      f"time, {workers} processes, {wall:.1f} s wall incl. startup",
            "single_thread_ms_per_step": round(per_scene_ms, 3), "kd_build_ms_per_step": round(build_ms, 3)}


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--scenes", type=int, default=256, help="scenes per GPU per step (BASELINE configs[2])")
    ap.add_argument("--points", type=int, default=50000)
    ap.add_argument("--T", type=float, default=0.66)
    ap.add_argument("--K", type=int, default=8)
    ap.add_argument("--streams", type=int, default=20,
                    help="independent steps in flight (each on its own HIP stream with its own handles)")
    ap.add_argument("--precision", type=int, default=64, choices=(32, 64),
                    help="arithmetic of the MPC solve (64 = the reference's; 32 = BASELINE configs[4] variant, not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="extra untimed pass with every kernel class timed")
    args = ap.parse_args()

    # one hardware queue per in-flight step (ROCm defaults to 4 and multiplexes streams onto them: two streams on one
    # queue serialise); measured on MI355X: 4 queues 280k, 8 -> 315k, 24 -> 390k steps/s at 16 streams
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    import numpy as np
    import torch
    from avoid_mpc_amd import capi, synth
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = capi.load()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.points, args.T, args.K)

    prm = synth.MpcParams(T=args.T, K=args.K)
    S, n, ne, N = args.scenes, args.points, args.points // 10, prm.N
    from avoid_mpc_amd import fsm
    clouds = torch.empty((S, n, 3), dtype=torch.float32, device=dev)
    edges = torch.empty((S, ne, 3), dtype=torch.float32, device=dev)
    sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
    for s in range(S):
        sc = synth.make_scene(n, 100000 + rank * S + s, prm)
        clouds[s] = torch.from_numpy(sc["cloud"]).to(dev)
        edges[s] = torch.from_numpy(sc["edge"]).to(dev)
        sq[s] = fsm.state_quads(sc["pos"], sc["vel"], sc["acc"], sc["yaw"], prm.decay, prm.max_iter); ref0[s] = sc["ref_path"]; posx[s] = sc["pos"][0]
    sq_d = torch.from_numpy(sq).to(dev); ref0_d = torch.from_numpy(ref0).to(dev); posx_d = torch.from_numpy(posx).to(dev)
    from avoid_mpc_amd import shard

    class Slot:
        """Everything one in-flight step owns: a HIP stream, the dual KD indices of its frame, its
        MPC batch (warm start, workspace), its reference path and outputs.  Consecutive steps are
        independent frames, so several are kept in flight: while one step sits in its latency-bound
        solve (256 wavefronts on 1024 SIMDs) another streams its clouds."""

        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev)
            self.kd_o, self.kd_e = KdBatch(S, n), KdBatch(S, ne)
            self.mpc = MpcBatch(prm.T, prm.dt, prm.K, S); self.mpc.configure(prm); self.mpc.set_precision(args.precision)
            self.ref = ref0_d.clone()
            self.u_all = torch.empty((S * world, 4), dtype=torch.float64, device=dev) if world > 1 else None
            self.out = dict(u=torch.empty((S, 4), dtype=torch.float64, device=dev),
                            x0array=torch.empty((S, N, 14), dtype=torch.float64, device=dev),
                            flags=torch.empty((S, 4), dtype=torch.int32, device=dev))

    slots = [Slot() for _ in range(max(1, args.streams))]
    out = slots[0].out
    step_no = [0]

    def one_step():
        sl = slots[step_no[0] % len(slots)]
        step_no[0] += 1
        with torch.cuda.stream(sl.stream):
            sl.ref.copy_(ref0_d, non_blocking=True)   # fresh frame: mRefPath after GetInitPath
            sl.mpc.reset_warm_start(sl.stream)        # zero warm start (HighLvlMpc.cpp:26-27,35)
            sl.kd_o.build(clouds, stream=sl.stream)   # FrameKDMap::AddVertex: obstacle index ...
            sl.kd_e.build(edges, stream=sl.stream)    # ... and edge index (FrameKDMap.cpp:44-47)
            step_batch(sl.kd_o, sl.kd_e, sl.mpc, prm, sq_d, posx_d, sl.ref, stream=sl.stream, out=sl.out)
            if world > 1:
                shard.gather_controls(sl.out["u"], out=sl.u_all)   # the one exchange step: controls to every rank

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(len(slots)):            # untimed priming: every slot allocates its workspace once
        one_step()
    barrier()
    for _ in range(args.warmup):
        one_step()
    barrier()
    lib.amk__timing_enable(1)              # HIP events around the dominant kernel, on its launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    t_enq = time.perf_counter() - t0       # host time to enqueue everything (diagnostic: launch-bound if ~= dt)
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_double * 8)(); cnt = (C.c_int * 8)()
    capi.check(lib.amk__timing_collect(ms, cnt), "timing")
    lib.amk__timing_enable(0)
    dt = shard.max_over_ranks(dt, dev)

    flags = out["flags"].cpu().numpy()
    solves = float(flags[:, 1].mean()); ipm_iters = float(flags[:, 3].mean())
    breakdown = None
    if args.breakdown and world == 1:   # (extra steps on one rank would unbalance the collectives)
        lib.amk__timing_enable(2)
        for _ in range(max(3, args.steps // 4)):
            one_step()
        torch.cuda.synchronize()
        ms2 = (C.c_double * 8)(); cnt2 = (C.c_int * 8)()
        capi.check(lib.amk__timing_collect(ms2, cnt2), "timing")
        lib.amk__timing_enable(0)
        reps = max(3, args.steps // 4)
        breakdown = {KCLASS[i]: {"ms_per_step": round(ms2[i] / reps, 4), "launches_per_step": cnt2[i] / reps}
                     for i in range(8)}

    if rank == 0:
        total_scenes = S * world * args.steps
        value = total_scenes / dt
        solve_launches = max(cnt[5], 1)
        solve_ms = ms[5] / solve_launches
        alg_launch = solve_alg_bytes(N, prm.K) * S
        achieved = alg_launch / (solve_ms * 1e-3) / 1e9
        step_bytes = alg_bytes_per_step(n, ne, N, prm.K)
        build_ms = ms[7] / max(cnt[7], 1)
        build_alg = 28 * S * (n + ne) // 2       # 12 B read + 16 B written per point; mean of the obstacle and the edge launch
        build_traffic = None
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):   # PMC passes cannot run inside this process: reuse the committed rocprofv3 result
            tj = json.load(open(tpath))
            m = tj.get("_meta", {})
            if (m.get("scenes_per_gpu"), m.get("points"), m.get("horizon"), m.get("K")) == (S, n, N, prm.K):
                kk = tj["kernels"].get(f"mpc_solve_kernel<{N}>")
                kb = tj["kernels"].get("kd_build_kernel")
                if kb:
                    build_traffic = round(kb["hbm_bytes_per_launch_x2"])
                if kk:
                    traffic = round(kk["hbm_bytes_per_launch_x2"])
                    traffic_src = "profiles/r01_pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, KiB, separate passes)"
        line = {
            "metric": f"MPC steps/sec ({n // 1000}k-pt cloud, N={N}, {prm.K} obstacle constraints)",
            "value": round(value, 1), "unit": "MPC steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] batched as configs[2]: {S} scenes/GPU x ({n}-pt obstacle "
                                   f"cloud + {ne}-pt edge cloud, N={N}, K={prm.K}), fresh frame + zero warm start "
                                   f"every step", "scenes_per_gpu": S, "points": n, "horizon": N, "K": prm.K,
                       "mpc_max_iter": prm.max_iter, "ipm_max_iter": 10,
                       "solves_per_step": round(solves, 3), "ipm_iters_per_step": round(ipm_iters, 2),
                       "streams_in_flight": len(slots), "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]),
                       "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 4),
                       "parallelism": f"scenes sharded over {world} GPU(s); all_gather of u" if world > 1
                       else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "mpc_solve_kernel", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(solve_ms, 4), "launches": cnt[5],
                         "alg_bytes_per_launch": alg_launch,
                         "note": "dominant kernel by time; it is bound by LDS bandwidth and dependent-op latency "
                                 "(DESIGN.md section 5), not by HBM; avg_launch_ms is submit-to-complete on the launch "
                                 "stream with 15 other steps in flight"},
            "roofline_kd_build": {"bound": "hbm", "kernel": "kd_build_kernel (obstacle + edge launch averaged)",
                                  "alg_bytes_per_launch": build_alg, "avg_launch_ms": round(build_ms, 4),
                                  "launches": cnt[7], "achieved": round(build_alg / (build_ms * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(build_alg / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "traffic": build_traffic,
                                  "note": "the HBM-heavy kernel: algorithmic 28 B per point (12 read, 16 written as a bucket "
                                          "record); the kernel reads the cloud three times (two of them L2/MALL-warm) and "
                                          "its scattered 16-byte stores leave L2 about twice"},
            "roofline_whole_step": {"alg_bytes_per_scene_step": step_bytes,
                                    "achieved": round(value * step_bytes / 1e9, 2), "unit": "GB/s",
                                    "frac": round(value * step_bytes / 1e9 / HBM_PEAK_GBS, 5)},
            "cpu_baseline": cpu,
        }
        if breakdown:
            line["kernel_breakdown"] = breakdown
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
