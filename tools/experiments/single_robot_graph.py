"""Single-robot latency (S = 1, the reference's deployment: AM/src/mpc_obstacle_avoidance_node.cpp:8): the whole control step -- fresh
frame, both index builds, amk_step_batch -- issued launch by launch on one stream against the same step replayed from a HIP graph
(capturable and bit-identical: tests/test_graph_gpu.py), at the reference's own configuration (3072-point frames, N = 30, K = 3) and at
BASELINE C2 (50 k points, N = 20, K = 8).  us per step, median of 200."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from avoid_mpc_amd import synth, fsm
from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch

def run(n, T, K, S=1, reps=200):
    prm = synth.MpcParams(T=T, K=K); dev = torch.device("cuda"); N = prm.N
    clouds, edges = synth.make_clouds_torch(n, S, 4242, dev)
    sq = np.zeros((S, prm.max_iter, 10)); ref0 = np.zeros((S, N, 10)); posx = np.zeros(S)
    for s in range(S):
        pos, vel, acc, yaw = synth.make_odom(4242 + s, prm)
        sq[s] = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0[s] = synth.make_ref_path(pos, prm); posx[s] = pos[0]
    sq = torch.from_numpy(sq).to(dev); ref0 = torch.from_numpy(ref0).to(dev); posx = torch.from_numpy(posx).to(dev)
    kd_o, kd_e = KdBatch(S, n), KdBatch(S, max(n // 10, 1))
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    ref = ref0.clone()
    out = dict(u=torch.empty((S, 4), dtype=torch.float64, device=dev), x0array=torch.empty((S, N, 14), dtype=torch.float64, device=dev),
               flags=torch.empty((S, 4), dtype=torch.int32, device=dev))
    st = torch.cuda.Stream()
    def step():
        ref.copy_(ref0, non_blocking=True); mpc.reset_warm_start(st)
        kd_o.build(clouds, stream=st); kd_e.build(edges, stream=st)
        step_batch(kd_o, kd_e, mpc, prm, sq, posx, ref, stream=st, out=out)
    with torch.cuda.stream(st):
        step(); step()
    st.synchronize()
    direct = {k: v.clone() for k, v in out.items()}
    def timed(fn):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.cuda.stream(st): fn()
            st.synchronize(); ts.append(time.perf_counter() - t0)
        return 1e6 * float(np.median(ts))
    t_direct = timed(step)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st): step()
    t_graph = timed(g.replay)
    same = all(torch.equal(out[k], direct[k]) for k in out)
    print(f"n {n} N {N} K {K} S {S}: launch by launch {t_direct:.0f} us, graph replay {t_graph:.0f} us per step (solves {int(out['flags'][0,1])}, iterations {int(out['flags'][0,3])}; same bits: {same})")

run(3072, 1.0, 3); run(50000, 0.66, 8); run(3072, 1.0, 3, S=8)
