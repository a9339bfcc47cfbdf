"""Round 5: how often does the reference's tie order matter on the clouds the reference produces?
amk_kd_tie_flags over the obstacle K-NN and edge 1-NN queries of closed-loop flights whose frames come from rendered depth
images -- back-projected obstacle clouds and the edge clouds at their 8-bit-quantised depth (FrameKDMap.cpp:180-200) -- with the
sensor delivering 16UC1 millimetres and 32FC1 metres, at the test sensor (320 x 240 / 5) and at the yaml's (640 x 480 / 10).
A flag = an exact tie among the k + 1 nearest (two returned neighbours, or the k-th and the best rejected): only then can the
default lowest-index order differ from nanoflann's first-visited order.  python tools/experiments/tie_census.py"""
import json, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from tests import _flight
from avoid_mpc_amd import flight, fsm
from avoid_mpc_amd.host import Pipeline, depth_params, kd_tie_flags

dev = torch.device("cuda")


def census(cam, kind, cfg="C1", F=16, P=60, seed0=3000, world_kw=None):
    prm, _ = _flight.make_prm(cfg)
    kw = world_kw or dict(cyl_per_m=1.5, x_first=3.0, length=60.0)
    worlds = [flight.FlightWorld(seed0 + i, prm, 1000, **kw) for i in range(F)]
    st = [flight.initial_state(seed0 + i, prm) for i in range(F)]
    x = np.stack([a for a, _ in st]); ref0 = np.stack([b for _, b in st]); ref = ref0.copy()
    cap = int(cam["cols"] / cam["resize_scale"]) * int(cam["rows"] / cam["resize_scale"])
    p2m = cam["pixel2meter"] if kind == "u16" else 1.0
    dp = depth_params(p2m, cam["depth_min"], cam["depth_max"], cam["resize_scale"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["Tbc"])
    pl = Pipeline(1, F, cap, cap, prm, queue_depth=1, gang=1, depth=dp)
    N, K = prm.N, prm.K
    tot = dict(obstacle_queries=0, obstacle_ties=0, edge_queries=0, edge_ties=0, periods=0, periods_with_a_tie=0, edge_points=0, cloud_points=0,
               edge_duplicate_points=0)
    for t in range(P):
        imgs, Twbs = [], []
        for i in range(F):
            Twb = np.eye(4); Twb[:3, 3] = np.round(x[i, 0:3], 6)
            sel = (worlds[i].cx > x[i, 0] - 2.0) & (worlds[i].cx < x[i, 0] + 40.0)
            d = flight.render_depth((worlds[i].cx[sel], worlds[i].cy[sel], worlds[i].cr[sel]), Twb, cam["Tbc"], cam["rows"], cam["cols"],
                                    cam["fx"], cam["fy"], cam["cx"], cam["cy"])
            imgs.append(np.clip(np.round(d / cam["pixel2meter"]), 0, 65535).astype(np.uint16) if kind == "u16" else d.astype(np.float32))
            Twbs.append(Twb)
        depth = torch.from_numpy(np.stack(imgs).view(np.int16) if kind == "u16" else np.stack(imgs)).to(dev)
        Twb = torch.from_numpy(np.stack(Twbs)).to(dev)
        odom = torch.from_numpy(x).to(dev); cmd = torch.empty((F, 3), dtype=torch.float64, device=dev)
        # the path the step starts from (GetInitPath on the slot's own copy)
        start = ref.copy()
        for i in range(F):
            fsm.get_init_path(start[i], prm.speed, prm.T, x[i, 0], 500.0, prm.height)
        tk = pl.submit(None, None, ref_path_init=torch.from_numpy(ref0).to(dev) if t == 0 else None, odom=odom, cmd_out=cmd,
                       keep_warm_start=t > 0, depth=depth, Twb=Twb)
        pl.wait(tk)
        o = pl.outputs(tk)
        ref = o["ref_path"].copy()
        kd_o, kd_e = pl.kd(0, 0), pl.kd(0, 1)
        any_tie = np.zeros(F, bool)
        for path in (start, ref):   # the queries of the first pass and (where the step re-planned) of the last refill
            q = torch.from_numpy(np.ascontiguousarray(path[:, :, 0:3])).to(dev)
            fo = kd_tie_flags(kd_o, q, K).cpu().numpy()
            fe = kd_tie_flags(kd_e, q[:, :1].contiguous(), 1).cpu().numpy()
            tot["obstacle_queries"] += fo.size; tot["obstacle_ties"] += int(fo.sum())
            tot["edge_queries"] += fe.size; tot["edge_ties"] += int(fe.sum())
            any_tie |= (fo.sum(axis=1) + fe.sum(axis=1)) > 0
        tot["periods"] += F; tot["periods_with_a_tie"] += int(any_tie.sum())
        tot["cloud_points"] += int(kd_o.sizes().sum()); tot["edge_points"] += int(kd_e.sizes().sum())
        a = cmd.cpu().numpy()
        x = flight.apply_command(x, a, prm)
    pl.close()
    tot["obstacle_tie_rate"] = tot["obstacle_ties"] / max(tot["obstacle_queries"], 1)
    tot["edge_tie_rate"] = tot["edge_ties"] / max(tot["edge_queries"], 1)
    tot["period_tie_rate"] = tot["periods_with_a_tie"] / max(tot["periods"], 1)
    tot["points_per_frame"] = tot["cloud_points"] / max(tot["periods"], 1); tot["edge_points_per_frame"] = tot["edge_points"] / max(tot["periods"], 1)
    return tot


out = {}
test_cam = _flight.DEPTH_CAM
yaml_cam = dict(rows=480, cols=640, pixel2meter=1e-3, depth_min=0.1, depth_max=100.0, resize_scale=10.0, fx=320.0, fy=320.0, cx=320.0, cy=240.0,
                Tbc=flight.TBC_YAML)
for name, cam, P, F in (("test sensor 320x240/5", test_cam, 60, 16), ("yaml sensor 640x480/10", yaml_cam, 30, 8)):
    for kind in ("u16", "f32"):
        for cfg in ("C1", "C2"):
            key = f"{name}, {'16UC1 mm' if kind == 'u16' else '32FC1 m'}, {cfg}"
            out[key] = census(cam, kind, cfg=cfg, P=P, F=F)
            print(key, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in out[key].items()}, flush=True)
os.makedirs("gpurun_out/r05c", exist_ok=True)
json.dump(out, open("gpurun_out/r05c/tie_census.json", "w"), indent=1)
