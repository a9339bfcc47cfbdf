"""Host-side pieces of the TASK branch that stay on the host in this design: the clock model and
the reference-path bookkeeping around amk_step_batch.  (The C++ twin for a ROS node is
include/avoid_mpc_amd/avoidance_step.hpp.)

  cur_state_quad  <- AvoidanceStateMachine::GetCurStateQuad   AM/src/AvoidanceStateMachine.cpp:183-203
  get_init_path   <- AvoidanceStateMachine::GetInitPath       AM/src/AvoidanceStateMachine.cpp:24-54 ("forward", "global_goal")
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


def state_quads(pos, vel, acc, yaw, decay, max_iter, iter_time=None, age=0.0):
    """[max_iter][10].  The reference reads the wall clock inside the re-plan loop
    (:329-330,343): iteration 0 extrapolates by `decay`, iteration i by the measured duration of
    iteration i-1 on top of the time already spent.  The device loop has no host round trip, so the
    caller supplies a clock model: every outer iteration is assumed to take `iter_time` seconds
    (default: decay, the reference's own compute-latency assumption, mpc_parameters.yaml:77): pass 0 extrapolates by decay,
    pass i >= 1 by (i + 1) * iter_time = i passes spent + the duration of pass i - 1 (written iter_time + i * iter_time: the
    bits of the default iter_time = decay are those of decay + i * decay).  age: now - mTimePos at the start of the step
    (:183-184), on top of every pass."""
    it = decay if iter_time is None else iter_time
    return np.stack([cur_state_quad(pos, vel, acc, yaw, (age + (decay if i == 0 else it)) + i * it) for i in range(max_iter)])


def get_init_path(ref_path, speed, T, pos_x, farest_point, height, task="forward", global_goal=None, dt=None):
    """GetInitPath: shift by one, append the goal (:24-54).  In place.  task "forward" (:29-33): the goal is speed * T ahead of
    the odometry position, capped at farest_point; task "global_goal" (:34-45): the path's last point walks towards
    global_goal (mStateGlobalGoal; default the constructor's {0, 0, height}, :22) by at most speed * dt, and its z is written
    into every shifted point (goalz, :46-52)."""
    N = ref_path.shape[0]
    goalx, goaly, goalz = min(speed * T + pos_x, farest_point), 0.0, height
    if task == "global_goal":
        g = np.array([0.0, 0.0, height]) if global_goal is None else np.asarray(global_goal, np.float64)
        last = ref_path[N - 1, 0:3]
        d = g - last
        z = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
        nrm = np.sqrt(z)
        e = d / nrm if z > 0.0 else d                       # Eigen's normalized()
        step = min(nrm, speed * dt)
        goalx, goaly, goalz = (float(last[i] + e[i] * step) for i in range(3))
    else:
        assert task == "forward", task
    for i in range(N - 1):
        nxt = ref_path[i + 1].copy()
        ref_path[i] = nxt
        ref_path[i, 2] = goalz
    ref_path[N - 1] = [goalx, goaly, goalz, 0.0, speed, 0.0, 0.0, 0.0, 0.0, 0.0]
    return ref_path
