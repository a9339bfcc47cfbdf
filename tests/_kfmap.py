"""The reference's multi-frame map over a flight (TEST INFRASTRUCTURE): FrameKDMap's keyframe list on the CPU oracle's trees.

Restates, statement by statement, what happens to the map between two control steps of the reference when the keyframe thread
is on (its default: max_frame_count = 100, only_trust_vel = false, AM/src/FrameKDMap.cpp:29-32):

  add_vertex   FrameKDMap::AddVertex after ProcessDepth          AM/src/FrameKDMap.cpp:39-51   (two fresh trees, Twc, flag)
  update       one pass of KeyframeThreadWorker's body           :443-486  (30 ms cadence: once per depth frame)
               - empty list: InsertKeyFrame                       :446-449
               - pop the oldest keyframes while the list is longer than max_frame_count or the drone has passed them
                 (DroneBehindPts: the <= 10 points nearest to the drone, body-frame x <= depth_min)   :450-459, 233-252
               - n x 1-NN sweep of the NEWEST keyframe's points against the current frame; with >= th_count outliers the
                 keyframe's tree is rebuilt from them, in place, and the current frame is appended   :463-486
  frames       mVecQueryVector = [current, every keyframe but the newest]                            :64-74
The step over these frames is oracle/step_oracle.c: stepo_run_frames (PtIsInFrame fast path, per-frame merge).

Twb = Twc * Tbc^-1 is formed with the rigid inverse [R' | -R' t] of Tbc (Eigen's general 4 x 4 inverse is not restated:
parity with it is unpinned, like the other Eigen products of the depth path)."""
import numpy as np

from tests import _oracle


def rigid_inverse(T):
    T = np.asarray(T, np.float64).reshape(4, 4)
    R, t = T[:3, :3], T[:3, 3]
    out = np.eye(4)
    out[:3, :3] = R.T
    out[:3, 3] = -(R.T @ t)
    return out


def drone_pose(Twc, Tbc_inv):
    """(twb [3], body x axis in the world [3]) of Twb = Twc * Tbc^-1, summed in index order like the device kernel."""
    Twc = np.asarray(Twc, np.float64).reshape(4, 4)
    Twb = np.zeros((4, 4))
    for i in range(4):
        for j in range(4):
            acc = 0.0
            for k in range(4):
                acc += Twc[i, k] * Tbc_inv[k, j]
            Twb[i, j] = acc
    return Twb[:3, 3].copy(), Twb[:3, 0].copy()


class Frame:
    def __init__(self, kd, ke, Twc, stamp):
        self.kd, self.ke, self.Twc, self.stamp = kd, ke, Twc, stamp


class MapOracle:
    def __init__(self, max_frame_count, th_dist, th_count, depth_min, Tbc):
        self.max_frame_count, self.th_dist, self.th_count, self.depth_min = int(max_frame_count), float(th_dist), int(th_count), float(depth_min)
        self.Tbc_inv = rigid_inverse(Tbc)
        self.cur = None
        self.kfs = []          # mKeyFrameMap, oldest first
        self.need = False      # mbNeedProcessPtCloud
        self.Twc = np.eye(4)   # mCurFrame.Twc (identity before the first frame)
        self.last_outliers = -1

    # FrameKDMap::AddVertex (:39-51); the caller ran ProcessDepth / BuildEdgeCloud (with self.Twc, the STALE pose, :209)
    def add_vertex(self, cloud, edge, Twc, stamp=0):
        if len(cloud) == 0:
            return                                                     # :39-41
        self.cur = Frame(_oracle.kd_oracle(cloud), _oracle.kd_oracle(edge), None, stamp)
        self.Twc = np.asarray(Twc, np.float64).reshape(4, 4).copy()    # :50
        self.need = True

    def drone_behind_pts(self, frame):                                 # :233-252
        twb, bx = drone_pose(self.Twc, self.Tbc_inv)
        cnt = min(frame.kd.size(), 10)
        _, _, pts = frame.kd.search(twb, cnt)                          # SearchForNearest: nothing when size == cnt
        for p in pts.astype(np.float64):
            ptbx = (bx[0] * (p[0] - twb[0]) + bx[1] * (p[1] - twb[1])) + bx[2] * (p[2] - twb[2])
            if ptbx <= self.depth_min:
                return False
        return True

    def update(self):                                                  # KeyframeThreadWorker's body (:443-486)
        self.last_outliers = -1                                       # (bookkeeping of this restatement: outliers of THIS pass's sweep)
        if self.max_frame_count <= 0 or not self.need:
            return
        self.need = False
        if not self.kfs:                                               # :446-449
            self.kfs.append(self.cur)
            return
        while self.kfs:                                                # :450-459
            if len(self.kfs) > self.max_frame_count or not self.drone_behind_pts(self.kfs[0]):
                self.kfs.pop(0)
            else:
                break
        if not self.kfs:
            return
        last = self.kfs[-1]
        if last.kd is self.cur.kd:                                     # a tree swept against itself has no outlier
            return
        rebuilt, n_out = last.kd.keyframe_sweep(self.cur.kd, self.th_dist, self.th_count)   # :463-485, rebuilds in place
        self.last_outliers = n_out
        if not rebuilt:
            return
        self.kfs.append(self.cur)                                      # InsertKeyFrame :486

    def frames(self):                                                  # UpdateQueryVector (:64-74)
        if self.cur is None:
            return []
        return [self.cur] + self.kfs[:-1]

    def step(self, mpc, prm, state_quad, pos_x, ref_path, cam):
        fr = self.frames()
        return _oracle.step_oracle_frames([f.kd for f in fr], [f.ke for f in fr], mpc, prm, state_quad, pos_x, ref_path,
                                          Twc=self.Twc if cam is not None else None, cam=cam)

    def summary(self):
        """(number of keyframes, [size of every query frame's obstacle cloud])"""
        return len(self.kfs), [f.kd.size() for f in self.frames()]
