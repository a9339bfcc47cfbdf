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


@pytest.mark.parametrize("gang", [1, 2])
@pytest.mark.parametrize("stage", [1, 2])
def test_pipeline_survives_a_failed_launch(gang, stage):
    """A launch that fails half-way: the staged frames are dropped, the failure is reported by that
    submit() and by wait() / drain(), the round robin moves on, and every later frame returns what a fresh pipeline returns."""
    import torch
    from avoid_mpc_amd.host import Pipeline
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne, n_slots = 4, 5000, 500, 2
    frames = _frames(torch, prm, 3 * gang * n_slots, S, n)

    def run(pl, frs):
        rows = torch.zeros((len(frs), S, 4), dtype=torch.float64, device="cuda")
        for i, fr in enumerate(frs):
            pl.submit(fr["cl"], fr["ed"], fr["sq"], fr["px"], fr["ref"], u_out=rows[i])
        pl.drain()
        return rows.cpu().numpy()

    good = Pipeline(n_slots, S, n, ne, prm, gang=gang)
    want = run(good, frames)
    good.close()
    pl = Pipeline(n_slots, S, n, ne, prm, gang=gang)
    run(pl, frames[:gang * n_slots])                       # a healthy round first
    assert pl.lib.amk__pipeline_inject_failure(pl.h, stage) == capi.AMK_OK
    bad = torch.zeros((S, 4), dtype=torch.float64, device="cuda")
    st = [pl.lib.amk_pipeline_submit(pl.h, C.byref(capi.PipelineFrame(fr["cl"].data_ptr(), None, fr["ed"].data_ptr(), None, 3, 0,
                                                                       fr["sq"].data_ptr(), fr["px"].data_ptr(), fr["ref"].data_ptr(),
                                                                       bad.data_ptr(), None, 0.0, None, None)), None)
          for fr in frames[:gang]]
    assert st[:-1] == [capi.AMK_OK] * (gang - 1) and st[-1] == capi.AMK_ERR_HIP       # the submit that launched reports it
    assert pl.lib.amk_pipeline_wait(pl.h, 0) == capi.AMK_ERR_HIP                        # ... and so does wait() on that slot
    assert pl.lib.amk_pipeline_query(pl.h, 0) == -1                                     # nothing of it is left staged or running: query() says "failed" 
    assert pl.lib.amk_pipeline_drain(pl.h) == capi.AMK_ERR_HIP
    # the failed launch consumed slot 0's turn: the next frames start on slot 1, fill whole gangs, and are the fresh pipeline's
    got = run(pl, frames)
    assert np.array_equal(got, want)
    assert pl.lib.amk_pipeline_drain(pl.h) == capi.AMK_OK                                # the failure is forgotten once the slot ran again
    # argument errors leave nothing staged either
    fr = frames[0]
    f5 = capi.PipelineFrame(fr["cl"].data_ptr(), None, fr["ed"].data_ptr(), None, 5, 0, fr["sq"].data_ptr(), fr["px"].data_ptr(),
                            fr["ref"].data_ptr(), None, None, 0.0, None, None)
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(f5), None) == capi.AMK_ERR_INVALID_ARG
    assert np.array_equal(run(pl, frames), want)
    pl.close()


def test_task_frames_need_a_reference_path_and_lose_it_with_a_failed_launch():
    """a TASK frame shifts the mRefPath the slot keeps for its position -- the first one there must bring
    d_ref_path_init (InitCircleState's role, AvoidanceStateMachine.cpp:14-23); and a launch that fails after the prologue has
    already shifted the path: the slot's persistent state is declared lost, the next TASK frame must re-initialise it."""
    import torch
    from avoid_mpc_amd.host import Pipeline
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne = 4, 5000, 500
    fr = _frames(torch, prm, 1, S, n)[0]
    odom = torch.zeros((S, 10), dtype=torch.float64, device="cuda"); odom[:, 2] = prm.height; odom[:, 4] = prm.speed
    cmd = torch.zeros((S, 3), dtype=torch.float64, device="cuda")
    pl = Pipeline(1, S, n, ne, prm, gang=1)

    def task_frame(with_ref):
        return capi.PipelineFrame(fr["cl"].data_ptr(), None, fr["ed"].data_ptr(), None, 3, 1, None, None,
                                  fr["ref"].data_ptr() if with_ref else None, None, odom.data_ptr(), 0.0, cmd.data_ptr())
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(task_frame(False)), None) == capi.AMK_ERR_INVALID_ARG   # no path yet
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(task_frame(True)), None) == capi.AMK_OK
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(task_frame(False)), None) == capi.AMK_OK                # the slot's own path
    assert pl.lib.amk_pipeline_drain(pl.h) == capi.AMK_OK and pl.lib.amk_pipeline_query(pl.h, 0) == 1
    assert pl.lib.amk__pipeline_inject_failure(pl.h, 2) == capi.AMK_OK                                       # after GetInitPath ran
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(task_frame(False)), None) == capi.AMK_ERR_HIP
    assert pl.lib.amk_pipeline_query(pl.h, 0) == -1
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(task_frame(False)), None) == capi.AMK_ERR_INVALID_ARG   # the shifted path is not trusted
    assert pl.lib.amk_pipeline_submit(pl.h, C.byref(task_frame(True)), None) == capi.AMK_OK
    assert pl.lib.amk_pipeline_drain(pl.h) == capi.AMK_OK and pl.lib.amk_pipeline_query(pl.h, 0) == 1
    pl.close()


def test_pipeline_orders_inputs_after_the_producing_stream():
    """a slot runs on its own stream.  Pipeline.submit hands over an event recorded on torch's current stream
    (amk_pipeline_frame.input_ready), so inputs written by kernels still queued there are complete when the slot reads them."""
    import torch
    from avoid_mpc_amd.host import Pipeline
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne = 4, 5000, 500
    fr = _frames(torch, prm, 1, S, n)[0]
    pl = Pipeline(2, S, n, ne, prm, gang=2)
    pl.submit(fr["cl"], fr["ed"], fr["sq"], fr["px"], fr["ref"]); pl.drain()
    want = pl.outputs(0)["u"].copy()
    # the inputs are produced by a long chain of kernels on torch's stream, then submitted at once without a host synchronisation
    big = torch.zeros(64 << 20, device="cuda")
    for rep in range(3):
        cl = torch.zeros_like(fr["cl"]); ref = torch.zeros_like(fr["ref"]); sq = torch.zeros_like(fr["sq"])
        torch.cuda.synchronize()
        for _ in range(20):
            big.add_(1.0)                         # ~ms of queued work in front of the producers
        cl.copy_(fr["cl"]); ref.copy_(fr["ref"]); sq.copy_(fr["sq"])
        t = pl.submit(cl, fr["ed"], sq, fr["px"], ref)
        pl.wait(t)
        assert np.array_equal(pl.outputs(t)["u"], want), rep
    # and the way back: wait_stream orders torch's stream after the step without blocking the host
    t = pl.submit(fr["cl"], fr["ed"], fr["sq"], fr["px"], fr["ref"])
    pl.wait_stream(t)
    u = pl.output_tensors(t)["u"].clone()         # queued on torch's stream behind the slot's event
    torch.cuda.synchronize()
    assert np.array_equal(u.cpu().numpy(), want)
    pl.close()


def test_task_mode_slow_down_command():
    """TASK mode's epilogue: PubCmd when isSafety, PubSlowDownCmd (AvoidanceStateMachine.cpp:379-397) when PlanWapionts found no
    edge point next to a too-close obstacle (:270-274).  Scenes with an empty edge cloud and an obstacle point on the first
    reference point are unsafe; the command equals the host twin's.  One re-plan pass (mpc_max_iter = 1): isSafety is
    overwritten by every pass (:331), and a later pass plans from the PREDICTED path, which has left the obstacle."""
    import torch
    from avoid_mpc_amd import flight
    from avoid_mpc_amd.host import Pipeline
    prm = synth.MpcParams(T=0.33, K=3, max_iter=1)
    S, n, ne = 6, 5000, 500
    fr = _frames(torch, prm, 1, S, n)[0]
    rng = np.random.default_rng(3)
    x = np.zeros((S, 10)); x[:, 0:3] = [0.0, 0.2, prm.height]; x[:, 4:7] = rng.normal(size=(S, 3)) * [12.0, 40.0, 3.0]
    x[:, 7:10] = rng.normal(size=(S, 3)) * [3.0, 3.0, 40.0]
    ref = np.stack([synth.make_ref_path(x[s, 0:3], prm) for s in range(S)])
    shifted = ref.copy()
    flight.period_inputs(x, shifted, prm)                     # what GetInitPath leaves in mRefPath[0]
    cl = fr["cl"].clone()
    cl[:, 0, :] = torch.from_numpy(shifted[:, 0, 0:3].astype(np.float32)).cuda()   # an obstacle point ON the first reference point
    edge_counts = torch.tensor([0, ne, 0, ne, 0, ne], dtype=torch.int32, device="cuda")
    pl = Pipeline(1, S, n, ne, prm)
    cmd = torch.zeros((S, 3), dtype=torch.float64, device="cuda")
    t = pl.submit(cl, fr["ed"], ref_path_init=torch.from_numpy(ref).cuda(), odom=torch.from_numpy(x).cuda(), cmd_out=cmd,
                  edge_counts=edge_counts)
    pl.wait(t)
    o = pl.outputs(t)
    assert list(o["flags"][:, 0]) == [0, 1, 0, 1, 0, 1]
    want = flight.command(o["u"], o["flags"], x, prm)
    assert np.array_equal(cmd.cpu().numpy(), want)
    assert np.abs(want[0] - o["u"][0, :3]).max() > 1e-3      # the slow-down command is not the solver's
    pl.close()


def test_pipeline_with_keyframes_equals_step_batch_frames():
    """amk_pipeline_frame.kf_*: the slot's own indices of the frame + the caller's keyframe handles = amk_step_batch_frames over
    [cur, keyframes ...] (FrameKDMap.cpp:64-74,215-231,254-427), bit for bit; a gang refuses keyframes."""
    import torch
    from avoid_mpc_amd.host import KdBatch, MpcBatch, Pipeline, step_batch_frames
    from tests import _oracle
    prm = synth.MpcParams(T=0.66, K=8)
    S, n = 6, 20000
    scenes = [synth.make_scene(n, 900 + i, prm) for i in range(S)]
    spans = [(6.0, 30.0), (0.0, 8.0), (3.0, 12.0)]
    Twc = np.array([[0, 0, 1, -2.0], [-1, 0, 0, 0.0], [0, -1, 0, 1.5], [0, 0, 0, 1.0]])
    cam = capi.FrameCamera(32.0, 32.0, 32.0, 24.0, 6.0, 64, 48)

    def packed(lst, cap):
        buf = np.zeros((S, cap, 3), np.float32); cnt = np.zeros(S, np.int32)
        for s, x in enumerate(lst):
            buf[s, :len(x)] = x; cnt[s] = len(x)
        return torch.from_numpy(buf).cuda(), torch.from_numpy(cnt).cuda()

    ne = n // 10
    clouds = [packed([sc["cloud"][(sc["cloud"][:, 0] >= a) & (sc["cloud"][:, 0] < b)] for sc in scenes], n) for a, b in spans]
    edges = [packed([sc["edge"][(sc["edge"][:, 0] >= a) & (sc["edge"][:, 0] < b)] for sc in scenes], ne) for a, b in spans]
    kd_o = [KdBatch(S, n) for _ in spans]; kd_e = [KdBatch(S, ne) for _ in spans]
    for f in range(3):
        kd_o[f].build(*clouds[f]); kd_e[f].build(*edges[f])
    sq = torch.from_numpy(np.stack([_oracle.scene_state_quads(sc, prm) for sc in scenes])).cuda()
    pos_x = torch.from_numpy(np.array([sc["pos"][0] for sc in scenes])).cuda()
    ref0 = torch.from_numpy(np.stack([sc["ref_path"] for sc in scenes])).cuda()
    Tw = torch.from_numpy(np.repeat(Twc[None], S, 0).copy()).cuda()
    mpc = MpcBatch(prm.T, prm.dt, prm.K, S); mpc.configure(prm)
    ref = ref0.clone()
    want = step_batch_frames(kd_o, kd_e, mpc, prm, sq, pos_x, ref, Twc=Tw, cam=cam)
    torch.cuda.synchronize()
    pl = Pipeline(1, S, n, ne, prm)
    t = pl.submit(clouds[0][0], edges[0][0], sq, pos_x, ref0, cloud_counts=clouds[0][1], edge_counts=edges[0][1],
                  keyframes=[(kd_o[1], kd_e[1]), (kd_o[2], kd_e[2])], Twc_cur=Tw, cam=cam)
    pl.wait(t)
    o = pl.outputs(t)
    assert np.array_equal(o["u"], want["u"].cpu().numpy()) and np.array_equal(o["flags"], want["flags"].cpu().numpy())
    assert np.array_equal(o["ref_path"], ref.cpu().numpy())
    mpc1 = MpcBatch(prm.T, prm.dt, prm.K, S); mpc1.configure(prm)       # and the keyframes matter on these scenes
    ref1 = ref0.clone()
    single = step_batch_frames(kd_o[:1], kd_e[:1], mpc1, prm, sq, pos_x, ref1, Twc=Tw, cam=cam)
    torch.cuda.synchronize()
    assert not torch.equal(single["u"], want["u"])
    pl.close()
    pg = Pipeline(1, S, n, ne, prm, gang=2)
    with pytest.raises(capi.AmkError):
        pg.submit(clouds[0][0], edges[0][0], sq, pos_x, ref0, keyframes=[(kd_o[1], kd_e[1])])
    pg.close()


@pytest.mark.parametrize("use_odom_est,age,iter_time", [(True, 0.004, 0.007), (False, 0.0, 0.0)])
def test_task_mode_clock_model_and_odometry_age(use_odom_est, age, iter_time):
    """TASK mode's prologue against the host twins for the non-default settings: odometry that is `age` seconds old
    (GetCurStateQuad extrapolates over now + decay - mTimePos, AvoidanceStateMachine.cpp:183-184,330), a pass duration other
    than `decay`, and use_odom_est = false (:186-191)."""
    import torch
    from avoid_mpc_amd import flight, fsm
    from avoid_mpc_amd.host import Pipeline
    prm = synth.MpcParams(T=0.33, K=3)
    S, n, ne = 5, 5000, 500
    fr = _frames(torch, prm, 1, S, n)[0]
    rng = np.random.default_rng(8)
    x = np.zeros((S, 10)); x[:, 0:3] = [0.0, 0.1, prm.height]; x[:, 3] = 0.02; x[:, 4:7] = [prm.speed, 0.3, -0.1] + rng.normal(size=(S, 3)) * 0.2
    x[:, 7:10] = rng.normal(size=(S, 3))
    ref0 = np.stack([synth.make_ref_path(x[s, 0:3], prm) for s in range(S)])
    # host twin: GetInitPath, then the per-pass states
    ref_h = ref0.copy()
    it = iter_time if iter_time > 0 else prm.decay
    sq = np.zeros((S, prm.max_iter, 10))
    for s in range(S):
        fsm.get_init_path(ref_h[s], prm.speed, prm.T, x[s, 0], 500.0, prm.height)
        sq[s] = np.stack([fsm.cur_state_quad(x[s, 0:3], x[s, 4:7], x[s, 7:10], x[s, 3], age + prm.decay + i * it, use_odom_est)
                          for i in range(prm.max_iter)])
    ph = Pipeline(1, S, n, ne, prm)
    t = ph.submit(fr["cl"], fr["ed"], torch.from_numpy(sq).cuda(), torch.from_numpy(x[:, 0].copy()).cuda(), torch.from_numpy(ref_h).cuda())
    ph.wait(t); want = ph.outputs(t); ph.close()
    pt = Pipeline(1, S, n, ne, prm, iter_time=iter_time, use_odom_est=use_odom_est)
    cmd = torch.zeros((S, 3), dtype=torch.float64, device="cuda")
    t = pt.submit(fr["cl"], fr["ed"], ref_path_init=torch.from_numpy(ref0).cuda(), odom=torch.from_numpy(x).cuda(), odom_age=age, cmd_out=cmd)
    pt.wait(t); got = pt.outputs(t); pt.close()
    assert np.array_equal(got["u"], want["u"]) and np.array_equal(got["flags"], want["flags"]) and np.array_equal(got["ref_path"], want["ref_path"])
    assert np.array_equal(cmd.cpu().numpy(), flight.command(got["u"], got["flags"], x, prm))
