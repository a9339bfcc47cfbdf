"""Round 5 (review item 2a): CU-masked streams.  (1) how the bits of hipExtStreamCreateWithCUMask map to (XCC, CU) on this part;
(2) how the index build's bandwidth scales with the CUs it may use, for masks inside one XCD and spread over all eight;
(3) builds on their own CUs beside a continuous supply of solves on the others (the pair that excludes each other through LDS and
registers when they share CUs).  python tools/experiments/cu_mask.py [probe] [scale] [overlap]"""
import os, sys, time, subprocess, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
sys.path.insert(0, '.')
import numpy as np, torch
from avoid_mpc_amd import synth, capi
from avoid_mpc_amd.host import MpcBatch, KdBatch, kd_build_pair

what = set(sys.argv[1:]) or {"probe", "scale", "overlap"}
lib = capi.load()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
N_CU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """bits: iterable of CU-mask bit numbers that are ON"""
    words = (N_CU + 31) // 32
    m = (C.c_uint32 * words)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), words, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


def probe():
    so = "/tmp/libcuprobe.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "tools/experiments/hip/cu_probe.hip", "-o", so])
    pl = C.CDLL(so)
    pl.cu_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    out = torch.zeros(2 * 4096, dtype=torch.int32, device="cuda")

    def where(bits):
        st = masked_stream(bits)
        out.zero_()
        pl.cu_probe(out.data_ptr(), 4096, 256, 200000, st.cuda_stream)   # 2 ms per block: every allowed CU gets blocks
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.uint32).reshape(-1, 2)
        hw, xcc = o[:, 0], o[:, 1] & 0xF
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
        return sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    print("device CUs", N_CU)
    for name, bits in (("bit 0", [0]), ("bit 1", [1]), ("bit 7", [7]), ("bit 8", [8]), ("bit 9", [9]), ("bit 31", [31]), ("bit 32", [32]),
                       ("bits 0-7", range(8)), ("bits 0-31", range(32)), ("bits 0,8,16,...,248", range(0, 256, 8)),
                       ("bits 224-255", range(224, 256)), ("all", range(N_CU))):
        w = where(list(bits))
        xs = sorted(set(x for x, _, _, _ in w))
        print(f"mask {name}: {len(w)} distinct (xcc, se, sh, cu); XCCs {xs}; first {w[:6]}")


S, n, ne = 256, 50000, 5000
G = 4


def scale_and_overlap():
    from tests.test_mpc_gpu import _scene_inputs
    prm = synth.MpcParams(T=0.66, K=8)
    base = torch.from_numpy(synth.make_cloud(n, 7)[0]).cuda()
    cl = torch.empty((G * S, n, 3), dtype=torch.float32, device="cuda")
    for s in range(G * S):
        cl[s] = base[torch.randperm(n, device="cuda")]
    ed = cl[:, :ne].contiguous()
    kdo = [KdBatch(G * S, n) for _ in range(4)]
    kde = [KdBatch(G * S, ne) for _ in range(4)]
    alg = G * S * (n + ne) * 28

    def builds(streams, reps):
        for i, st in enumerate(streams):
            kd_build_pair(kdo[i], cl, kde[i], ed, stream=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in range(reps):
            for i, st in enumerate(streams):
                kd_build_pair(kdo[i], cl, kde[i], ed, stream=st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * len(streams))

    layouts = {}
    # which bits make "one XCD" and which "k CUs of every XCD" depends on the mapping the probe prints; both readings are measured:
    # contiguous bits [0, c) and strided bits {i : i % 8 < c / 32}
    for c in (16, 32, 64, 96, 128, 192, 256):
        layouts[f"contiguous {c}"] = list(range(c))
        layouts[f"bits i with (i mod 8) < {c // 32}" if c >= 32 else f"bits 0,16,...({c})"] = \
            [i for i in range(N_CU) if (i % 8) < c // 32] if c >= 32 else list(range(0, N_CU, N_CU // c))
    if "scale" in what:
        for name, bits in layouts.items():
            sts = [masked_stream(bits) for _ in range(2)]
            t = builds(sts, 6)
            print(f"build of {G * S} scenes x ({n} + {ne}) points on {len(bits):3d} CUs [{name}], 2 streams: {t * 1e6:7.1f} us per launch = {alg / t / 1e12:.2f} TB/s")
    if "overlap" not in what:
        return
    logs = _scene_inputs(20000, [200, 201, 202, 203], prm)
    ref = torch.from_numpy(np.stack([logs[i % 4][0] for i in range(G * S)])).cuda()
    NS = 6
    mpcs = [MpcBatch(prm.T, prm.dt, prm.K, G * S) for _ in range(NS)]
    for m in mpcs:
        m.configure(prm)
    outs = [(torch.empty((G * S, 4), dtype=torch.float64, device="cuda"), torch.empty((G * S, 4), dtype=torch.int32, device="cuda")) for _ in range(NS)]

    def solve(i, st):
        mpcs[i].reset_warm_start(st)
        capi.check(lib.amk_mpc_solve(mpcs[i].h, capi.dptr(ref), capi.dptr(outs[i][0]), None, capi.dptr(outs[i][1]), 0, capi.stream_ptr(st)), 's')

    def run(ssts, bsts, rs, rb):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for r in range(max(rs, rb)):
            if r < rs:
                for i, st in enumerate(ssts):
                    solve(i, st)
            if r < rb:
                for i, st in enumerate(bsts):
                    kd_build_pair(kdo[i], cl, kde[i], ed, stream=st)
        e0 = time.perf_counter() - t0
        es = [torch.cuda.Event(enable_timing=False) for _ in ssts]; eb = [torch.cuda.Event() for _ in bsts]
        for e, st in zip(es, ssts): e.record(st)
        for e, st in zip(eb, bsts): e.record(st)
        for e in eb: e.synchronize()
        tb = time.perf_counter() - t0
        for e in es: e.synchronize()
        ts = time.perf_counter() - t0
        torch.cuda.synchronize()
        return ts * 1e3, tb * 1e3, e0 * 1e3
    RS, RB = 8, 24   # 8 x NS solve launches of 1024 scenes (~ 8 x 6 x 1024 / 2.4 per us = 20 ms), 24 x 2 builds (~ 16 ms alone)
    for name, bbits in (("no masks (shared CUs)", None), ("builds on bits 0-31", list(range(32))), ("builds on bits i mod 8 == 0", [i for i in range(N_CU) if i % 8 == 0]),
                        ("builds on bits 0-63", list(range(64))), ("builds on bits i mod 8 < 2", [i for i in range(N_CU) if i % 8 < 2])):
        if bbits is None:
            ssts = [torch.cuda.Stream() for _ in range(NS)]; bsts = [torch.cuda.Stream() for _ in range(2)]
        else:
            sb = set(bbits)
            ssts = [masked_stream([i for i in range(N_CU) if i not in sb]) for _ in range(NS)]
            bsts = [masked_stream(bbits) for _ in range(2)]
        run(ssts, bsts, 1, 1)
        ts, _, _ = run(ssts, [], RS, 0)
        _, tb, _ = run([], bsts, 0, RB)
        cs, cb, ce = run(ssts, bsts, RS, RB)
        print(f"{name}: solves alone {ts:.1f} ms ({RS * NS * G * S / ts / 1e3:.2f} per us), builds alone {tb:.1f} ms ({alg * RB * 2 / tb / 1e9:.2f} TB/s); "
              f"together: solves done at {cs:.1f} ms, builds at {cb:.1f} ms (enqueue {ce:.1f} ms); sum alone {ts + tb:.1f}")


if "probe" in what:
    probe()
if what & {"scale", "overlap"}:
    scale_and_overlap()
