/* TEST INFRASTRUCTURE ONLY.  CPU restatement of one control step: the TASK branch of
 * AvoidanceStateMachine::Step with the single-frame FrameKDMap queries it makes.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Restated (AM = /root/reference/roswrapper/ros/src/avoid_mpc):
 *   map_query            FrameKDMap::QueryNearest (+WithCurFrame, ThreadWorker)  AM/src/FrameKDMap.cpp:254-376
 *   map_nearest_distance FrameKDMap::GetNearestDistance (+worker)               AM/src/FrameKDMap.cpp:378-427
 *   plan_waypoints       AvoidanceStateMachine::PlanWapionts                    AM/src/AvoidanceStateMachine.cpp:259-281
 *   process_waypoints    AvoidanceStateMachine::ProcessWaypoints                :204-235
 *   get_ref_states       AvoidanceStateMachine::GetRefStates                    :236-257
 *   stepo_run            TASK branch of Step                                    :322-355
 *   stepo_get_init_path  GetInitPath ("forward" task)                           :24-54
 *   stepo_cur_state_quad GetCurStateQuad                                        :183-203
 * The map holds the current frame only (mVecQueryVector = [cur], FrameKDMap.cpp:64-74 with no
 * keyframes), which is what BASELINE.json's synthetic configs exercise.  The wall-clock reads of
 * the reference (ros::Time::now, :329,343) are replaced by caller-supplied per-iteration states.
 * KD queries go through kd_oracle.c (pinned to the reference's nanoflann), the solve through
 * mpc_oracle.c (PARITY UNPINNED, see there).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* kd_oracle.c */
int kdo_size(void *h);
int kdo_search(void *h, double x, double y, double z, int n, int *indices, double *sqdist, float *pts_xyz);
/* mpc_oracle.c */
int mpco_horizon(void *h);
int mpco_Solve(void *h, const double *ref_states, double *u, double *x0array, int faster);
const int *mpco_last_info(void *h);

#define MAXK 64

/* QueryNearest for a single-frame map.  Fast path (cur frame has >= k points and the query is
 * inside the camera frustum, FrameKDMap.cpp:339-345) and slow path (k' = min(k, size), :298) end in
 * the same SearchForNearest call except for k' -- both are restated; `in_frame` selects. */
static int map_query(void *kd, const double *p, int k, int in_frame, double pts[][3], double *d2) {
    if (!kd) return 0;
    int size = kdo_size(kd);
    int kq = (size >= k && in_frame) ? k : (k < size ? k : size);
    if (kq <= 0) return 0;
    int idx[MAXK];
    double dd[MAXK];
    float pf[MAXK * 3];
    int cnt = kdo_search(kd, p[0], p[1], p[2], kq, idx, dd, pf);
    /* slow path sorts by distance and truncates to k (:371-375): already sorted, cnt <= k */
    for (int i = 0; i < cnt; ++i) {
        pts[i][0] = pf[3 * i];
        pts[i][1] = pf[3 * i + 1];
        pts[i][2] = pf[3 * i + 2];
        d2[i] = dd[i];
    }
    return cnt;
}

static double map_nearest_distance(void *kd, const double *p) {
    double nearest = DBL_MAX;
    if (kd && kdo_size(kd) > 0) { /* worker skips empty clouds, FrameKDMap.cpp:385-387 */
        int idx[1];
        double dd[1];
        float pf[3];
        int cnt = kdo_search(kd, p[0], p[1], p[2], 1, idx, dd, pf);
        if (cnt > 0 && dd[0] < nearest) nearest = dd[0];
    }
    return sqrt(nearest);
}

/* GetCurStateQuad (:183-203): constant-acceleration extrapolation by dt seconds */
void stepo_cur_state_quad(const double *pos, const double *vel, const double *acc, double yaw, double dt,
                          int use_odom_est, double *sq) {
    for (int i = 0; i < 3; ++i) {
        double p = pos[i], v = vel[i];
        if (use_odom_est) {
            p = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt;
            v = vel[i] + acc[i] * dt;
        }
        sq[i] = p;
        sq[4 + i] = v;
        sq[7 + i] = acc[i];
    }
    sq[3] = yaw;
}

/* GetInitPath, "forward" task (:24-54): shift the path by one and append the goal */
void stepo_get_init_path(double *ref_path, int N, double speed, double T, double pos_x, double farest, double height) {
    double goalx = speed * T + pos_x;
    goalx = fmin(goalx, farest);
    double goaly = 0, goalz = height;
    for (int i = 0; i < N - 1; ++i) {
        double *a = ref_path + 10 * i, *b = ref_path + 10 * (i + 1);
        a[0] = b[0]; a[1] = b[1]; a[2] = goalz;
        for (int j = 3; j < 10; ++j) a[j] = b[j];
    }
    double *l = ref_path + 10 * (N - 1);
    memset(l, 0, sizeof(double) * 10);
    l[0] = goalx; l[1] = goaly; l[2] = goalz; l[4] = speed;
}

/* flags[4] = {isSafety, solves done, worst solver status over the solves (-1: none ran), total interior-point iterations}
 * ref_log (optional): [mpc_max_iter][20+10N+3KN] the vecRefStates handed to each Solve. */
int stepo_run(void *kd_obs, void *kd_edge, void *mpc, int K, double speed, double T, double safety_distance,
              int mpc_max_iter, const double *state_quad, double pos_x, double *ref_path, double *u, double *x0array,
              int *flags, double *ref_log) {
    const int N = mpco_horizon(mpc);
    const int nref = 20 + 10 * N + 3 * K * N;
    double *ref_states = (double *)malloc(sizeof(double) * nref);
    double *x0 = (double *)calloc((size_t)14 * N, sizeof(double));
    double *obst = (double *)malloc(sizeof(double) * 3 * K * N);
    int is_safety = 1, solves = 0, status = -1, iters = 0;
    memset(u, 0, sizeof(double) * 4);
    for (int iter = 0; iter < mpc_max_iter; ++iter) {
        const double *sq = state_quad + 10 * iter; /* GetCurStateQuad(start + decay), :330 */
        /* PlanWapionts :259-281 (only ref point 0) */
        is_safety = 1;
        {
            double *p1 = ref_path;
            double nd = map_nearest_distance(kd_obs, p1);
            if (!(nd > safety_distance)) {
                double ep[1][3], ed[1];
                int cnt = map_query(kd_edge, p1, 1, 1, ep, ed);
                if (cnt == 0) is_safety = 0;
                else {
                    p1[0] = ep[0][0]; p1[1] = ep[0][1]; p1[2] = ep[0][2];
                    is_safety = 1;
                }
            }
        }
        /* ProcessWaypoints :204-235 */
        int need_replan = 0;
        for (int i = 0; i < N; ++i) {
            double pts[MAXK][3], d2[MAXK];
            int cnt = map_query(kd_obs, ref_path + 10 * i, K, 1, pts, d2);
            for (int j = 0; j < K; ++j) {
                double *o = obst + 3 * (K * i + j);
                if (j < cnt) { o[0] = pts[j][0]; o[1] = pts[j][1]; o[2] = pts[j][2]; }
                else { o[0] = o[1] = o[2] = 10000; } /* :223-226 */
            }
            if (cnt == 0 || sqrt(d2[0]) <= safety_distance) need_replan = 1; /* :228-231 */
        }
        if (!need_replan && iter > 0 && is_safety) break; /* :333-335 */
        /* GetRefStates :236-257 */
        memcpy(ref_states, sq, sizeof(double) * 10);
        memcpy(ref_states + 10, ref_path, sizeof(double) * 10 * N);
        memcpy(ref_states + 10 + 10 * N, obst, sizeof(double) * 3 * K * N);
        {
            double *tg = ref_states + 10 + 10 * N + 3 * K * N;
            memcpy(tg, ref_path + 10 * (N - 1), sizeof(double) * 10);
            double dX = speed * T - fmax(0., tg[0] - pos_x);
            dX = fmax(0., dX);
            tg[0] += dX;
            tg[1] = 0.;
        }
        if (ref_log) memcpy(ref_log + (size_t)nref * iter, ref_states, sizeof(double) * nref);
        { const int st = mpco_Solve(mpc, ref_states, u, x0, iter == 0); /* :337 */
          if (st > status) status = st; }
        iters += mpco_last_info(mpc)[1];
        ++solves;
        for (int i = 0; i < N; ++i) memcpy(ref_path + 10 * i, x0 + 14 * i, sizeof(double) * 10); /* :338-342 */
    }
    if (x0array) memcpy(x0array, x0, sizeof(double) * 14 * N);
    if (flags) {
        flags[0] = is_safety;
        flags[1] = solves;
        flags[2] = status;
        flags[3] = iters;
    }
    free(ref_states);
    free(x0);
    free(obst);
    return 0;
}

/* ---------------------------------------------------------------- multi-frame map ------------------------------------
 * FrameKDMap with keyframes: mVecQueryVector = [cur, keyframes...] (AM/src/FrameKDMap.cpp:64-74).  Restated:
 *   pt_is_in_frame        PtIsInFrame                                     FrameKDMap.cpp:215-231
 *   mapf_query            QueryNearest (+WithCurFrame, ThreadWorker)      :254-376
 *   mapf_nearest_distance GetNearestDistance (+worker)                    :378-427
 *   stepo_run_frames      TASK branch of Step on that map                 AvoidanceStateMachine.cpp:322-355
 * cam[7] = {fx, fy, cx, cy (already divided by the resize scale, :21-24), depth_max, width, height}; Twc row-major 4x4
 * (NULL: every point counts as inside the current frame).  The reference sorts the merged candidates with std::sort on
 * the squared distance alone (:371, FrameKDMap.h:52-54: equal distances in unspecified order); here ties keep the
 * earlier frame / earlier neighbour -- the rule of the product. */
static int pt_is_in_frame(const double *p, const double *T, const double *cam) {
    if (!T) return 1;
    /* Twc.inverse() * p for a rigid Twc: R'(p - t) */
    const double dx = p[0] - T[3], dy = p[1] - T[7], dz = p[2] - T[11];
    const double x = T[0] * dx + T[4] * dy + T[8] * dz;
    const double y = T[1] * dx + T[5] * dy + T[9] * dz;
    const double z = T[2] * dx + T[6] * dy + T[10] * dz;
    if (z > cam[4] || z < 0) return 0;
    const double u = cam[0] * x / z + cam[2];
    const double v = cam[1] * y / z + cam[3];
    if (u < 0 || u >= cam[5] || v < 0 || v >= cam[6]) return 0;
    return 1;
}

typedef struct { double pt[3], d; } ptd;

static int mapf_query(void **kds, int F, const double *p, int k, const double *Twc, const double *cam, double pts[][3],
                      double *d2) {
    int idx[MAXK];
    double dd[MAXK];
    float pf[MAXK * 3];
    if (kds[0]) {
        const int first = kdo_size(kds[0]);
        if (first >= k && pt_is_in_frame(p, Twc, cam)) { /* fast path :339-345 */
            int cnt = kdo_search(kds[0], p[0], p[1], p[2], k, idx, dd, pf);
            for (int i = 0; i < cnt; ++i) {
                pts[i][0] = pf[3 * i]; pts[i][1] = pf[3 * i + 1]; pts[i][2] = pf[3 * i + 2];
                d2[i] = dd[i];
            }
            return cnt;
        }
    }
    ptd all[64 * MAXK];
    int n = 0;
    for (int f = 0; f < F; ++f) { /* one worker per frame in the reference, :347-364 */
        if (!kds[f]) continue;
        const int size = kdo_size(kds[f]);
        const int kq = k < size ? k : size; /* :298 */
        int cnt = kdo_search(kds[f], p[0], p[1], p[2], kq, idx, dd, pf);
        for (int j = 0; j < cnt; ++j) {
            all[n].pt[0] = pf[3 * j]; all[n].pt[1] = pf[3 * j + 1]; all[n].pt[2] = pf[3 * j + 2];
            all[n].d = dd[j];
            ++n;
        }
    }
    /* stable insertion sort on the distance (:371) */
    for (int i = 1; i < n; ++i) {
        ptd t = all[i];
        int j = i - 1;
        while (j >= 0 && all[j].d > t.d) { all[j + 1] = all[j]; --j; }
        all[j + 1] = t;
    }
    int cnt = n < k ? n : k; /* :372-375 */
    for (int i = 0; i < cnt; ++i) {
        pts[i][0] = all[i].pt[0]; pts[i][1] = all[i].pt[1]; pts[i][2] = all[i].pt[2];
        d2[i] = all[i].d;
    }
    return cnt;
}

static double mapf_nearest_distance(void **kds, int F, const double *p) {
    double nearest = DBL_MAX;
    for (int f = 0; f < F; ++f)
        if (kds[f] && kdo_size(kds[f]) > 0) { /* :385-387 */
            int idx[1];
            double dd[1];
            float pf[3];
            int cnt = kdo_search(kds[f], p[0], p[1], p[2], 1, idx, dd, pf);
            if (cnt > 0 && dd[0] < nearest) nearest = dd[0];
        }
    return sqrt(nearest);
}

int stepo_run_frames(void **kd_obs, void **kd_edge, int F, const double *Twc, const double *cam, void *mpc, int K,
                     double speed, double T, double safety_distance, int mpc_max_iter, const double *state_quad,
                     double pos_x, double *ref_path, double *u, double *x0array, int *flags) {
    const int N = mpco_horizon(mpc);
    const int nref = 20 + 10 * N + 3 * K * N;
    double *ref_states = (double *)malloc(sizeof(double) * nref);
    double *x0 = (double *)calloc((size_t)14 * N, sizeof(double));
    double *obst = (double *)malloc(sizeof(double) * 3 * K * N);
    int is_safety = 1, solves = 0, status = -1, iters = 0;
    memset(u, 0, sizeof(double) * 4);
    for (int iter = 0; iter < mpc_max_iter; ++iter) {
        const double *sq = state_quad + 10 * iter;
        is_safety = 1;
        {
            double *p1 = ref_path;
            double nd = mapf_nearest_distance(kd_obs, F, p1);
            if (!(nd > safety_distance)) {
                double ep[1][3], ed[1];
                int cnt = mapf_query(kd_edge, F, p1, 1, Twc, cam, ep, ed);
                if (cnt == 0) is_safety = 0;
                else { p1[0] = ep[0][0]; p1[1] = ep[0][1]; p1[2] = ep[0][2]; }
            }
        }
        int need_replan = 0;
        for (int i = 0; i < N; ++i) {
            double pts[MAXK][3], d2[MAXK];
            int cnt = mapf_query(kd_obs, F, ref_path + 10 * i, K, Twc, cam, pts, d2);
            for (int j = 0; j < K; ++j) {
                double *o = obst + 3 * (K * i + j);
                if (j < cnt) { o[0] = pts[j][0]; o[1] = pts[j][1]; o[2] = pts[j][2]; }
                else { o[0] = o[1] = o[2] = 10000; }
            }
            if (cnt == 0 || sqrt(d2[0]) <= safety_distance) need_replan = 1;
        }
        if (!need_replan && iter > 0 && is_safety) break;
        memcpy(ref_states, sq, sizeof(double) * 10);
        memcpy(ref_states + 10, ref_path, sizeof(double) * 10 * N);
        memcpy(ref_states + 10 + 10 * N, obst, sizeof(double) * 3 * K * N);
        {
            double *tg = ref_states + 10 + 10 * N + 3 * K * N;
            memcpy(tg, ref_path + 10 * (N - 1), sizeof(double) * 10);
            double dX = speed * T - fmax(0., tg[0] - pos_x);
            dX = fmax(0., dX);
            tg[0] += dX;
            tg[1] = 0.;
        }
        { const int st = mpco_Solve(mpc, ref_states, u, x0, iter == 0);
          if (st > status) status = st; }
        iters += mpco_last_info(mpc)[1];
        ++solves;
        for (int i = 0; i < N; ++i) memcpy(ref_path + 10 * i, x0 + 14 * i, sizeof(double) * 10);
    }
    if (x0array) memcpy(x0array, x0, sizeof(double) * 14 * N);
    if (flags) { flags[0] = is_safety; flags[1] = solves; flags[2] = status; flags[3] = iters; }
    free(ref_states); free(x0); free(obst);
    return 0;
}
