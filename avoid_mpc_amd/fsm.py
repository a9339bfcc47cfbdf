"""Host-side pieces of the TASK branch that stay on the host in this design: the clock model and
the reference-path bookkeeping around amk_step_batch.  (The C++ twin for a ROS node is
include/avoid_mpc_amd/avoidance_step.hpp.)

  cur_state_quad  <- AvoidanceStateMachine::GetCurStateQuad   AM/src/AvoidanceStateMachine.cpp:183-203
  get_init_path   <- AvoidanceStateMachine::GetInitPath       AM/src/AvoidanceStateMachine.cpp:24-54 ("forward")
  state_quads     the per-outer-iteration initial states handed to amk_step_batch
"""
import numpy as np


def cur_state_quad(pos, vel, acc, yaw, dt, use_odom_est=True):
    """mVecStateQuad at time stamp + dt: constant-acceleration extrapolation (:186-191)."""
    pos, vel, acc = (np.asarray(v, np.float64) for v in (pos, vel, acc))
    sq = np.zeros(10)
    if use_odom_est:
        sq[0:3] = pos + vel * dt + 0.5 * acc * dt * dt
        sq[4:7] = vel + acc * dt
    else:
        sq[0:3] = pos
        sq[4:7] = vel
    sq[3] = yaw
    sq[7:10] = acc
    return sq


def state_quads(pos, vel, acc, yaw, decay, max_iter, iter_time=None):
    """[max_iter][10].  The reference reads the wall clock inside the re-plan loop
    (:329-330,343): iteration 0 extrapolates by `decay`, iteration i by the measured duration of
    iteration i-1 on top of the time already spent.  The device loop has no host round trip, so the
    caller supplies a clock model: every outer iteration is assumed to take `iter_time` seconds
    (default: decay, the reference's own compute-latency assumption, mpc_parameters.yaml:77)."""
    it = decay if iter_time is None else iter_time
    return np.stack([cur_state_quad(pos, vel, acc, yaw, decay + i * it) for i in range(max_iter)])


def get_init_path(ref_path, speed, T, pos_x, farest_point, height):
    """GetInitPath for task "forward": shift by one, append the goal (:29-33,46-53).  In place."""
    N = ref_path.shape[0]
    goalx = min(speed * T + pos_x, farest_point)
    for i in range(N - 1):
        nxt = ref_path[i + 1].copy()
        ref_path[i] = nxt
        ref_path[i, 2] = height
    ref_path[N - 1] = [goalx, 0.0, height, 0.0, speed, 0.0, 0.0, 0.0, 0.0, 0.0]
    return ref_path
