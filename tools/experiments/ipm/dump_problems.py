"""HARNESS (CPU only).  The solves of the bench workload as standalone problems: S cold steps of the step oracle on the
bench's scene model (synth.make_clouds_torch on the CPU generator: same model, other random stream than the GPU's), every
Solve's vecRefStates logged together with the warm start it began from -> scratch/ipm/problems_<cfg>.npz.  Solver
prototypes (proto.py) are then timed in ITERATIONS on exactly the problems a step poses, pass by pass."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from avoid_mpc_amd import synth, fsm
from tests import _oracle

def main(S=128, n=50000, T=0.66, K=8, seed=100000, out="scratch/ipm/problems_c2.npz"):
    prm = synth.MpcParams(T=T, K=K); N = prm.N
    clouds, edges = synth.make_clouds_torch(n, S, seed, torch.device("cpu"))
    clouds, edges = clouds.numpy(), edges.numpy()
    nref = 20 + 10 * N + 3 * K * N
    refs, w0s, passes, scene_of, iters = [], [], [], [], []
    t0 = time.time()
    for s in range(S):
        pos, vel, acc, yaw = synth.make_odom(seed + s, prm)
        sq = fsm.state_quads(pos, vel, acc, yaw, prm.decay, prm.max_iter); ref0 = synth.make_ref_path(pos, prm)
        kd, ke = _oracle.kd_oracle(clouds[s]), _oracle.kd_oracle(edges[s])
        # run the step pass by pass to catch the warm start of each Solve: the log gives the vecRefStates; re-solve them in order
        mpc = _oracle.MpcOracle(prm.T, prm.dt, prm.K); mpc.configure(prm)
        r = _oracle.step_oracle(kd, ke, mpc, prm, sq, pos[0], ref0.copy(), want_log=True)
        ns = int(r["flags"][1])
        m2 = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m2.configure(prm)
        tot = 0
        for p in range(ns):
            w0s.append(m2.warm_start.copy()); refs.append(r["ref_log"][p].copy()); passes.append(p); scene_of.append(s)
            u, x0, info = m2.Solve(r["ref_log"][p], faster=(p == 0))
            iters.append(int(info[1])); tot += int(info[1])
        assert tot == int(r["flags"][3]), (tot, r["flags"])
        kd.close(); ke.close(); mpc.close(); m2.close()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, ref=np.array(refs), w0=np.array(w0s), pas=np.array(passes), scene=np.array(scene_of),
                        iters=np.array(iters), T=T, K=K)
    it = np.array(iters); pa = np.array(passes)
    print("scenes", S, "solves", len(it), "per step %.2f" % (len(it) / S), "iters/step %.1f" % (it.sum() / S),
          "by pass:", [(p, int((pa == p).sum()), round(float(it[pa == p].mean()), 1)) for p in range(prm.max_iter)], "%.0f s" % (time.time() - t0))

if __name__ == "__main__":
    main(S=int(sys.argv[1]) if len(sys.argv) > 1 else 128)
