"""GPU step vs CPU oracle on the 2048 scenes of BASELINE configs[3] (8 shards of 256, seeds as tests/test_step_gpu.py): how
many scenes have identical flags, how many took a different branch at a rounding-level tie but reached the same optimum,
how many ended in another local minimum.  CPU side on all usable cores."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from concurrent.futures import ProcessPoolExecutor
from avoid_mpc_amd import synth
from tests import _oracle

prm = synth.MpcParams(T=0.66, K=8)


def cpu_scene(g):
    sc = synth.make_scene(50000, 100000 + g, prm)
    ko, ke = _oracle.kd_oracle(sc["cloud"]), _oracle.kd_oracle(sc["edge"])
    m = _oracle.MpcOracle(prm.T, prm.dt, prm.K); m.configure(prm)
    r = _oracle.step_oracle(ko, ke, m, prm, _oracle.scene_state_quads(sc, prm), sc["pos"][0], sc["ref_path"].copy())
    return r["u"], r["flags"]


def main():
    from avoid_mpc_amd.host import KdBatch, MpcBatch, step_batch
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    S = 256
    gu, gf = [], []
    for lo in range(0, total, S):
        scenes = [synth.make_scene(50000, 100000 + g, prm) for g in range(lo, lo + S)]
        kd_o, kd_e = KdBatch(S, 50000), KdBatch(S, 5000)
        kd_o.build(torch.from_numpy(np.stack([sc["cloud"] for sc in scenes])).cuda())
        kd_e.build(torch.from_numpy(np.stack([sc["edge"] for sc in scenes])).cuda())
        mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
        sq = np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])
        ref = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
        posx = torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda()
        out = step_batch(kd_o, kd_e, mpc, prm, torch.from_numpy(sq).cuda(), posx, ref)
        torch.cuda.synchronize()
        gu.append(out["u"].cpu().numpy()); gf.append(out["flags"].cpu().numpy())
        kd_o.close(); kd_e.close()
    gu, gf = np.concatenate(gu), np.concatenate(gf)
    quota = 16
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = max(1, int(q) // int(p)) if q != "max" else os.cpu_count()
    except Exception:
        pass
    t0 = time.time()
    with ProcessPoolExecutor(min(quota, len(os.sched_getaffinity(0)))) as ex:
        res = list(ex.map(cpu_scene, range(total), chunksize=8))
    cu = np.stack([r[0] for r in res]); cf = np.stack([r[1] for r in res])
    same = np.all(cf == gf, axis=1)
    du = np.abs(cu - gu).max(axis=1)
    flipped = ~same
    print(f"{total} scenes (CPU oracle {time.time() - t0:.0f} s): identical flags {int(same.sum())}, max |du| there {du[same].max():.2e}; "
          f"different iteration counts {int(flipped.sum())} of which same optimum (|du| <= 1e-4) {int((flipped & (du <= 1e-4)).sum())}, "
          f"another local minimum {int((flipped & (du > 1e-4)).sum())}")
    ds = np.sort(du[same])
    print("  |du| over the scenes with identical flags: median %.1e  p99 %.1e  p99.9 %.1e  max %.1e; above 1e-9: %d scenes"
          % (np.median(ds), ds[int(0.99 * len(ds))], ds[int(0.999 * len(ds))], ds[-1], int((ds > 1e-9).sum())))
    for g in np.where(flipped)[0]:
        print("  scene", g, "gpu flags", gf[g].tolist(), "cpu flags", cf[g].tolist(), "|du| %.3e" % du[g])


if __name__ == "__main__":
    main()
