"""GPU: amk_pipeline_* (several control steps in flight behind the C ABI): every submitted frame returns what the same frame
returns through the separate calls (amk_kd_build x 2 + amk_step_batch) -- at queue depth 1 (one step per slot) and 3 (steps
queued behind a running one on the same slot), with results read from the slot's buffers and from caller-supplied rows
(d_u_out), warm start kept or reset; wait / query / drain; argument errors."""
import ctypes as C

import numpy as np
import pytest

from avoid_mpc_amd import capi, synth

pytestmark = pytest.mark.gpu


def _frames(torch, prm, n_frames, S, n):
    from avoid_mpc_amd import fsm
    out = []
    for f in range(n_frames):
        scenes = [synth.make_scene(n, 5000 + 17 * f + s, prm) for s in range(S)]
        sq = np.stack([fsm.state_quads(sc["pos"], sc["vel"], sc["acc"], sc["yaw"], prm.decay, prm.max_iter) for sc in scenes])
        out.append(dict(cl=torch.from_numpy(np.stack([sc["cloud"] for sc in scenes])).cuda(),
                        ed=torch.from_numpy(np.stack([sc["edge"] for sc in scenes])).cuda(),
                        sq=torch.from_numpy(sq).cuda(), px=torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda(),
                        ref=torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()))
    return out


@pytest.mark.parametrize("depth", [1, 3])
def test_pipeline_equals_the_separate_calls(depth):
    import torch
    from avoid_mpc_amd.host import KdBatch, MpcBatch, Pipeline, step_batch
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne, n_slots, n_frames = 6, 5000, 500, 2, 7
    frames = _frames(torch, prm, n_frames, S, n)
    # reference: the separate calls, fresh warm start per frame
    want = []
    kd_o, kd_e = KdBatch(S, n), KdBatch(S, ne)
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    for fr in frames:
        kd_o.build(fr["cl"]); kd_e.build(fr["ed"]); mpc.reset_warm_start()
        ref = fr["ref"].clone()
        o = step_batch(kd_o, kd_e, mpc, prm, fr["sq"], fr["px"], ref)
        torch.cuda.synchronize()
        want.append(dict(u=o["u"].cpu().numpy().copy(), flags=o["flags"].cpu().numpy().copy(), x0=o["x0array"].cpu().numpy().copy(),
                         ref=ref.cpu().numpy().copy()))
    pl = Pipeline(n_slots, S, n, ne, prm, queue_depth=depth)
    assert pl.lib.amk_pipeline_slots(pl.h) == n_slots
    rows = torch.zeros((n_frames, S, 4), dtype=torch.float64, device="cuda")
    slots = [pl.submit(fr["cl"], fr["ed"], fr["sq"], fr["px"], fr["ref"], u_out=rows[i]) for i, fr in enumerate(frames)]
    assert slots == [i % n_slots for i in range(n_frames)]          # round robin
    pl.drain()
    assert all(pl.lib.amk_pipeline_query(pl.h, i) == 1 for i in range(n_slots))
    got = rows.cpu().numpy()
    for i in range(n_frames):
        assert np.array_equal(got[i], want[i]["u"]), i               # same kernels, same inputs: identical bits
    # the slots' own buffers hold the LAST frame each slot ran
    for sl in range(n_slots):
        last = max(i for i in range(n_frames) if i % n_slots == sl)
        o = pl.outputs(sl)
        assert np.array_equal(o["flags"], want[last]["flags"]) and np.array_equal(o["x0array"], want[last]["x0"])
        assert np.array_equal(o["ref_path"], want[last]["ref"])
    # without d_u_out the control lands in the slot's buffer; wait() on one slot
    sl = pl.submit(frames[0]["cl"], frames[0]["ed"], frames[0]["sq"], frames[0]["px"], frames[0]["ref"])
    pl.wait(sl)
    assert np.array_equal(pl.outputs(sl)["u"], want[0]["u"])
    # keep_warm_start: the second solve of the same frame starts from the first one's solution -> fewer iterations
    sl = pl.submit(frames[0]["cl"], frames[0]["ed"], frames[0]["sq"], frames[0]["px"], frames[0]["ref"])
    pl.wait(sl); cold = pl.outputs(sl)["flags"][:, 3].sum()
    for _ in range(n_slots):   # the same slot again
        sl2 = pl.submit(frames[0]["cl"], frames[0]["ed"], frames[0]["sq"], frames[0]["px"], frames[0]["ref"], keep_warm_start=True)
    pl.wait(sl2)
    assert sl2 == sl and pl.outputs(sl2)["flags"][:, 3].sum() < cold
    pl.close()


@pytest.mark.parametrize("gang,n_frames", [(2, 8), (3, 7), (8, 19)])
def test_pipeline_gang_equals_the_separate_calls(gang, n_frames):
    """gang frames per launch: every frame still returns what its own launches return, bit for bit -- also the frames of a
    gang that wait() / drain() had to launch partly filled; tickets address the frame's part of the slot's buffers."""
    import torch
    from avoid_mpc_amd.host import KdBatch, MpcBatch, Pipeline, step_batch
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne, n_slots = 5, 3000, 300, 2
    frames = _frames(torch, prm, n_frames, S, n)
    counts = [torch.full((S,), n - 7 * i, dtype=torch.int32, device="cuda") for i in range(n_frames)]   # ragged over frames
    want = []
    kd_o, kd_e = KdBatch(S, n), KdBatch(S, ne)
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    for fr, cnt in zip(frames, counts):
        kd_o.build(fr["cl"], counts=cnt); kd_e.build(fr["ed"]); mpc.reset_warm_start()
        ref = fr["ref"].clone()
        o = step_batch(kd_o, kd_e, mpc, prm, fr["sq"], fr["px"], ref)
        torch.cuda.synchronize()
        want.append(dict(u=o["u"].cpu().numpy().copy(), flags=o["flags"].cpu().numpy().copy(), x0=o["x0array"].cpu().numpy().copy(),
                         ref=ref.cpu().numpy().copy()))
    pl = Pipeline(n_slots, S, n, ne, prm, queue_depth=2, gang=gang)
    assert pl.gang == gang and pl.mpc(0).S == gang * S
    rows = torch.zeros((n_frames, S, 4), dtype=torch.float64, device="cuda")
    tickets = [pl.submit(fr["cl"], fr["ed"], fr["sq"], fr["px"], fr["ref"], cloud_counts=cnt, u_out=rows[i])
               for i, (fr, cnt) in enumerate(zip(frames, counts))]
    # G consecutive frames on one slot, then the next slot; ticket = position in the gang * n_slots + slot
    assert tickets == [(i % gang) * n_slots + (i // gang) % n_slots for i in range(n_frames)]
    if n_frames % gang:
        assert pl.lib.amk_pipeline_query(pl.h, tickets[-1]) == 0      # staged, not launched yet
    pl.drain()
    assert all(pl.lib.amk_pipeline_query(pl.h, t) == 1 for t in tickets)
    got = rows.cpu().numpy()
    for i in range(n_frames):
        assert np.array_equal(got[i], want[i]["u"]), i
    # the frame's part of the slot's buffers: the last frame that ran at that ticket
    for t in set(tickets):
        last = max(i for i in range(n_frames) if tickets[i] == t)
        if any(tickets[j] % n_slots == t % n_slots for j in range(last + 1, n_frames) if j // gang != last // gang):
            continue                                                   # a later gang of the slot overwrote it
        o = pl.outputs(t)
        assert np.array_equal(o["u"], want[last]["u"]) and np.array_equal(o["flags"], want[last]["flags"])
        assert np.array_equal(o["x0array"], want[last]["x0"]) and np.array_equal(o["ref_path"], want[last]["ref"])
    # wait() on a staged frame launches its gang (padded with copies of it)
    t = pl.submit(frames[1]["cl"], frames[1]["ed"], frames[1]["sq"], frames[1]["px"], frames[1]["ref"], cloud_counts=counts[1])
    pl.wait(t)
    assert np.array_equal(pl.outputs(t)["u"], want[1]["u"])
    # keep_warm_start is per frame: frame A keeps the slot's solution at its position, frame B at the same launch starts cold
    a, b = frames[2], frames[3]
    for _ in range(n_slots):   # run (a, b, ...) cold on every slot so that position 0 and 1 hold a's and b's solutions
        ta = pl.submit(a["cl"], a["ed"], a["sq"], a["px"], a["ref"], cloud_counts=counts[2])
        tb = pl.submit(b["cl"], b["ed"], b["sq"], b["px"], b["ref"], cloud_counts=counts[3])
        pl.wait(tb)
    cold_a, cold_b = pl.outputs(ta)["flags"][:, 3].sum(), pl.outputs(tb)["flags"][:, 3].sum()
    for _ in range(n_slots):
        ta2 = pl.submit(a["cl"], a["ed"], a["sq"], a["px"], a["ref"], cloud_counts=counts[2], keep_warm_start=True)
        tb2 = pl.submit(b["cl"], b["ed"], b["sq"], b["px"], b["ref"], cloud_counts=counts[3])
        pl.wait(tb2)
    assert (ta2, tb2) == (ta, tb)
    assert pl.outputs(ta2)["flags"][:, 3].sum() < cold_a and pl.outputs(tb2)["flags"][:, 3].sum() == cold_b
    assert np.array_equal(pl.outputs(tb2)["u"], want[3]["u"])
    pl.close()


@pytest.mark.parametrize("gang,n_slots,depth,seed", [(3, 3, 2, 0), (4, 2, 1, 1), (2, 4, 3, 2)])
def test_pipeline_random_interleavings(gang, n_slots, depth, seed):
    """Random sequences of submit / wait / query / drain on a ganged pipeline: every frame's controls (written to the caller's
    row) equal the separate calls' whatever gets launched partly filled, and a ticket's buffers hold its frame after wait()."""
    import torch
    from avoid_mpc_amd.host import KdBatch, MpcBatch, Pipeline, step_batch
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne, pool = 3, 1500, 150, 5
    frames = _frames(torch, prm, pool, S, n)
    want = []
    kd_o, kd_e = KdBatch(S, n), KdBatch(S, ne)
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    for fr in frames:
        kd_o.build(fr["cl"]); kd_e.build(fr["ed"]); mpc.reset_warm_start()
        ref = fr["ref"].clone()
        o = step_batch(kd_o, kd_e, mpc, prm, fr["sq"], fr["px"], ref)
        torch.cuda.synchronize()
        want.append(dict(u=o["u"].cpu().numpy().copy(), flags=o["flags"].cpu().numpy().copy()))
    rng = np.random.default_rng(seed)
    pl = Pipeline(n_slots, S, n, ne, prm, queue_depth=depth, gang=gang)
    n_ops = 70
    rows = torch.full((n_ops, S, 4), float("nan"), dtype=torch.float64, device="cuda")
    submitted = []          # (row, frame index, ticket, launch-independent order)
    live = {}               # ticket -> frame index of the LAST frame submitted with that ticket
    for op in range(n_ops):
        r = rng.random()
        if r < 0.7 or not submitted:
            f = int(rng.integers(pool))
            fr = frames[f]
            t = pl.submit(fr["cl"], fr["ed"], fr["sq"], fr["px"], fr["ref"], u_out=rows[op])
            assert 0 <= t < n_slots * gang
            submitted.append((op, f, t))
            live[t] = f
        elif r < 0.85:
            _, f, t = submitted[int(rng.integers(len(submitted)))]
            pl.wait(t)
            assert pl.lib.amk_pipeline_query(pl.h, t) == 1
            if live[t] == f:   # nothing newer was submitted at that ticket: its buffers hold this frame
                o = pl.outputs(t)
                # (a newer frame of the same slot at another position may have run since: only this position is checked)
                assert np.array_equal(o["u"], want[f]["u"]) and np.array_equal(o["flags"], want[f]["flags"]), (op, f, t)
        elif r < 0.95:
            _, _, t = submitted[int(rng.integers(len(submitted)))]
            assert pl.lib.amk_pipeline_query(pl.h, t) in (0, 1)
        else:
            pl.drain()
    pl.drain()
    got = rows.cpu().numpy()
    for row, f, _ in submitted:
        assert np.array_equal(got[row], want[f]["u"]), (row, f)
    pl.close()


def test_pipeline_argument_errors():
    import torch  # noqa: F401  (one HIP runtime per process)
    lib = capi.load()
    h = C.c_void_p()
    sp = capi.StepParams(10.0, 0.2, 3, 0)
    bad = capi.PipelineConfig(0, 4, 100, 10, 0.33, 0.033, 3, 0, 0, sp)
    assert lib.amk_pipeline_create(C.byref(bad), C.byref(h)) == capi.AMK_ERR_INVALID_ARG
    bad = capi.PipelineConfig(2, 4, 100, 10, 5.0, 0.033, 3, 0, 0, sp)                       # N = 151 > AMK_MAX_HORIZON
    assert lib.amk_pipeline_create(C.byref(bad), C.byref(h)) == capi.AMK_ERR_UNSUPPORTED and not h.value
    bad = capi.PipelineConfig(2, 4, 100, 10, 0.33, 0.033, 3, 0, 9, sp)                      # gang > AMK_PIPELINE_MAX_GANG
    assert lib.amk_pipeline_create(C.byref(bad), C.byref(h)) == capi.AMK_ERR_INVALID_ARG
    ok = capi.PipelineConfig(2, 4, 100, 10, 0.33, 0.033, 3, 0, 0, sp)
    assert lib.amk_pipeline_create(C.byref(ok), C.byref(h)) == 0
    fr = capi.PipelineFrame()                                                            # all NULL
    assert lib.amk_pipeline_submit(h, C.byref(fr), None) == capi.AMK_ERR_INVALID_ARG
    assert lib.amk_pipeline_wait(h, 5) == capi.AMK_ERR_INVALID_ARG and lib.amk_pipeline_query(h, -1) == -1
    assert lib.amk_pipeline_mpc(h, 2) is None and lib.amk_pipeline_kd(h, 0, 2) is None
    assert lib.amk_pipeline_query(h, 0) == 1 and lib.amk_pipeline_drain(h) == 0           # idle slots count as finished
    assert lib.amk_pipeline_destroy(h) == 0
