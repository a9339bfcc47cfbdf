/* avoid_mpc_amd.h -- C ABI of the MI355X-native Avoid-MPC hot path.
 *
 * One shared library (avoid_mpc_amd/libavoid_mpc_amd.so, HIP for gfx950).  Plain pointers and
 * sizes only; every `d_` pointer is DEVICE memory on the current HIP device, every `h_` pointer is
 * host memory; `stream` is a hipStream_t passed as void* (NULL = the default stream).  All entry
 * points return an amk_status and are re-entrant per handle (no global state; the reference's
 * KDTreeTwo keeps results in members and is not re-entrant, AM/include/kd_tree_two.h:109-111).
 *
 * A handle holds a BATCH of S independent objects ("scenes"): scene s of an amk_kd is one
 * KDTreeTwo<double>, scene s of an amk_mpc is one ObstacleAvoidanceMPC.  S = 1 reproduces the
 * reference's single-robot use; S >> 1 is how the GPU is filled.
 *
 * Reference interfaces replaced (AM = roswrapper/ros/src/avoid_mpc in the reference tree):
 *   amk_kd_*     AM/include/kd_tree_two.h:53-144   (KDTreeTwo<double>)  and, through it,
 *                AM/include/nanoflann_two.hpp:1518-1541,1563-1586 (buildIndex / findNeighbors)
 *   amk_mpc_*    AM/include/HighLvlMpc.h:4-33, AM/src/HighLvlMpc.cpp:5-137 (ObstacleAvoidanceMPC)
 *                and the plugin it loads (AM/tools/mpc_obstacle_casadi.py:51-242,338-357)
 *   amk_step_*   AM/src/AvoidanceStateMachine.cpp:204-281,322-355 (TASK branch of Step) with
 *                AM/src/FrameKDMap.cpp:254-275,322-427 (QueryNearest / GetNearestDistance)
 * C++ adapters with the reference's class names/signatures: the .hpp files under include/avoid_mpc_amd/.
 */
#ifndef AVOID_MPC_AMD_H
#define AVOID_MPC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum amk_status {
    AMK_OK = 0,
    AMK_ERR_INVALID_ARG = 1, /* NULL handle, k < 0, size mismatch ...                           */
    AMK_ERR_HIP = 2,         /* a HIP runtime call failed; see amk_last_hip_error()              */
    AMK_ERR_NO_DEVICE = 3,   /* no gfx950 device visible: there is NO CPU fallback               */
    AMK_ERR_UNSUPPORTED = 4, /* e.g. k > AMK_MAX_K, N > AMK_MAX_HORIZON                          */
    AMK_ERR_TIMEOUT = 5      /* amk_shard_wait: a collective did not finish in time               */
} amk_status;

#define AMK_MAX_K 64        /* neighbours per query (reference uses 1, 3, 8, <=10)               */
#define AMK_MAX_QUERIES 64  /* queries per scene per call (reference: N <= 30 per outer pass)    */
#define AMK_MAX_HORIZON 32  /* N = int(T/dt); reference default 30                               */
#define AMK_S_DIM 10        /* [px,py,pz,yaw,vx,vy,vz,ax,ay,az]  mpc_obstacle_casadi.py:41-46    */
#define AMK_U_DIM 4         /* [ax_cmd,ay_cmd,az_cmd,yaw_dot]    mpc_obstacle_casadi.py:75       */
#define AMK_MPC_DEFAULT_MAX_ITER 100 /* iteration cap of this library's interior-point method (see amk_mpc_create) */
#define AMK_MAX_BUDGET_ROUNDS 16    /* amk_mpc_set_solve_budget: budgeted rounds of a control step */

int amk_version(void);
const char *amk_status_string(int status);
int amk_last_hip_error(void); /* hipError_t of the last failing HIP call on this thread          */
int amk_device_count(void);

/* ------------------------------------------------------------------------------------------ */
/* KD index: batch of KDTreeTwo<double>                                    kd_tree_two.h:53-144 */
/* ------------------------------------------------------------------------------------------ */
typedef struct amk_kd amk_kd;

/* KDTreeTwo() x n_scenes.  max_points = capacity per scene.                                      */
int amk_kd_create(int n_scenes, int max_points, amk_kd **out);
int amk_kd_destroy(amk_kd *kd);

/* InitializeNew(cloud) for every scene (kd_tree_two.h:76-78,88-106): the handle COPIES the points
 * whose x is not NaN (order preserved, :96-101) and builds its index.  Scene s reads
 * d_xyz[s*scene_stride + i*point_stride + {0,1,2}], i < counts[s]; point_stride = 3 (packed) or
 * 4 (pcl::PointXYZ's 16-byte layout).  d_counts == NULL means every scene has max_points.       */
int amk_kd_build(amk_kd *kd, const float *d_xyz, int point_stride, long long scene_stride,
                 const int *d_counts, void *stream);

/* FrameKDMap::AddVertex's two InitializeNew calls (obstacle cloud + edge cloud of one depth frame, FrameKDMap.cpp:44-47) as
 * ONE launch: both clouds packed [S][max_points of the handle][point_stride]; results identical to two amk_kd_build calls. */
int amk_kd_build_pair(amk_kd *obstacle, const float *d_xyz, const int *d_counts, amk_kd *edge, const float *d_edge_xyz,
                      const int *d_edge_counts, int point_stride, void *stream);

/* cloud.pts.size() per scene after the NaN-x filter (synchronises the stream).                   */
int amk_kd_sizes(amk_kd *kd, int *h_sizes, void *stream);

/* SearchForNearest(x,y,z,n) (kd_tree_two.h:108-133) for n_queries query points per scene.
 *   d_queries  [S][n_queries][3] double
 *   d_indices  [S][n_queries][k] int     -> KDTreeTwo::indices            (may be NULL)
 *   d_sqdist   [S][n_queries][k] double  -> KDTreeTwo::squared_distances  (may be NULL)
 *   d_pts      [S][n_queries][k][3] float-> KDTreeTwo::closest_pts        (may be NULL)
 *   d_counts   [S][n_queries]    int     -> number of results, reproducing :119-124 exactly:
 *                                           size < k -> size ; size > k -> k ; size == k -> 0
 * Results ascend in squared distance; equal distances order by index (nanoflann orders ties by
 * tree-traversal order, nanoflann_two.hpp:224-229 -- see DESIGN.md "tie policy").  Slots beyond
 * the count hold index -1 / distance DBL_MAX / point (0,0,0).  Distances are the IEEE
 * left-to-right fp64 sums of kd_tree_two.h:24-27 with no FMA contraction.                       */
int amk_kd_search(amk_kd *kd, const double *d_queries, int n_queries, int k, int *d_indices,
                  double *d_sqdist, float *d_pts, int *d_counts, void *stream);

/* Tie visibility.  For every query: d_tie_flags[s][q] = 1 when, among the k + 1 nearest points, two are at exactly the same
 * squared distance (two returned neighbours, or the k-th returned one and the best rejected one), else 0.  nanoflann keeps
 * the first VISITED of equal distances (KNNResultSet::addPoint, nanoflann_two.hpp:219-246; traversal order), this library
 * the lowest index: with flag 0 indices and neighbour set are identical to the reference's, with flag 1 only the distance
 * list is guaranteed identical (DESIGN.md "tie policy").  Queries are read at d_queries[(s*n_queries + q)*query_stride
 * + {0,1,2}] (query_stride 3 = packed; 10 = the position part of mRefPath).  k + 1 <= AMK_MAX_K.                     */
int amk_kd_tie_flags(amk_kd *kd, const double *d_queries, int query_stride, int n_queries, int k, int *d_tie_flags,
                     void *stream);

/* Tie ORDER.  AMK_TIES_LOWEST_INDEX (default): equal squared distances order by cloud index.  AMK_TIES_NANOFLANN: the
 * handle also builds the reference's own tree (divideTree / middleSplit_ / planeSplit, leaf size 10: nanoflann_two.hpp:
 * 1055-1106,1197-1294, kd_tree_two.h:68) at every amk_kd_build / keyframe rebuild and amk_kd_search answers by nanoflann's
 * own traversal (searchLevel + KNNResultSet, :1729-1793,219-246): index lists identical to the reference's on ANY cloud,
 * ties included.  Costs milliseconds per build (level-synchronous, one wavefront per tree node) where the bucketed index
 * takes a fraction of one: meant for the reference's real frame sizes (<= 3072 points, quantised edge clouds), set it
 * before amk_kd_build.  amk_step_batch and amk_step_batch_frames honour it (queries and the edge snap's
 * re-query then go through nanoflann's traversal too).  A scene whose tree would exceed the node capacity
 * (cap / 2 + 64) or a traversal depth of 48 -- pathological data -- keeps the bucketed index's answer.  The mode takes effect at the NEXT
 * build: a search between amk_kd_set_tie_order(NANOFLANN) and that build still answers from the bucketed index (the
 * handle tracks whether its exact tree belongs to the cloud it currently holds).                                        */
#define AMK_TIES_LOWEST_INDEX 0
#define AMK_TIES_NANOFLANN 1
int amk_kd_set_tie_order(amk_kd *kd, int mode);
/* Whether AMK_TIES_NANOFLANN holds for a scene: d_status[s] (device, [n_scenes], stream-ordered after the build it describes)
 *   AMK_EXACT_OFF      (-1) the mode is off, or no build has run since it was switched on: the bucketed index answers (by design)
 *   AMK_EXACT_IN_USE   ( 0) the reference-shaped tree answers every search of the scene: nanoflann's index lists, ties included
 *   AMK_EXACT_GAVE_UP  ( 1) the build gave the tree up (node capacity cap / 2 + 64, its ring of open nodes, or its watchdog):
 *                           the bucketed index answers -- same distances, equal distances in cloud-index order
 *   AMK_EXACT_TOO_DEEP ( 2) the tree exists but is deeper than the traversal stack (48 levels): queries that would descend past
 *                           it are answered by the bucketed index, the others by the tree
 * A caller that NEEDS the reference's index lists checks this after the build; 1 and 2 are pathological data (never seen on
 * depth-derived or synthetic clouds; tests force them with a shrunken ring and a geometric point sequence).              */
#define AMK_EXACT_OFF (-1)
#define AMK_EXACT_IN_USE 0
#define AMK_EXACT_GAVE_UP 1
#define AMK_EXACT_TOO_DEEP 2
int amk_kd_exact_status(amk_kd *kd, int *d_status, void *stream);
int amk_kd_exact_status_host(amk_kd *kd, int *h_status);   /* the same into host memory; synchronises the device */

/* Keyframe sweep of FrameKDMap::KeyframeThreadWorker (AM/src/FrameKDMap.cpp:462-485), for every scene:
 * for each point of `keyframe` the nearest-neighbour distance in `current` (SearchForNearest(pt, 1));
 * points with sqrt(d2) > th_dist are outliers (a point gets no result, hence is no outlier, when `current`
 * holds <= 1 point, kd_tree_two.h:119-124).  A scene with >= th_count outliers has its `keyframe` rebuilt
 * from them (InitializeNew(newCloud), order preserved) and d_rebuilt[s] = 1; otherwise `keyframe` is left
 * untouched and d_rebuilt[s] = 0.  d_outliers[s] = number of outliers (either may be NULL).
 * Both handles must hold the same number of scenes.                                               */
int amk_kd_keyframe_sweep(amk_kd *keyframe, amk_kd *current, double th_dist, int th_count,
                          int *d_outliers, int *d_rebuilt, void *stream);

/* Same, synchronising, with the per-scene results in host memory.                                 */
int amk_kd_keyframe_sweep_host(amk_kd *keyframe, amk_kd *current, double th_dist, int th_count,
                               int *h_outliers, int *h_rebuilt);

/* GetPointCloud().pts (kd_tree_two.h:134-136): the handle's copy of the cloud (points whose x is not NaN,
 * in order) as packed xyz, h_xyz[s*max_points*3 + i*3 + c], and the per-scene sizes.  Synchronises.  */
int amk_kd_points_host(amk_kd *kd, float *h_xyz, int *h_sizes);

/* Host-buffer conveniences (stage through internal device buffers, synchronise).                 */
int amk_kd_build_host(amk_kd *kd, const float *h_xyz, int point_stride, long long scene_stride,
                      const int *h_counts);
int amk_kd_search_host(amk_kd *kd, const double *h_queries, int n_queries, int k, int *h_indices,
                       double *h_sqdist, float *h_pts, int *h_counts);

/* ------------------------------------------------------------------------------------------ */
/* MPC: batch of ObstacleAvoidanceMPC                          HighLvlMpc.h:4-33, .cpp:5-137    */
/* ------------------------------------------------------------------------------------------ */
typedef struct amk_mpc amk_mpc;

/* ObstacleAvoidanceMPC(T, dt, soPath) x n_scenes (HighLvlMpc.cpp:5-57).  The reference bakes
 * N = int(T/dt) and K = nearest_point_num into the generated plugin behind soPath
 * (mpc_obstacle_casadi.py:36-37,76-85); here they are constructor arguments.  Defaults after
 * create are the constructor's: weights/tau/gains of HighLvlMpc.cpp:53-56, control bounds
 * [-10,-10,1,-10]..[10,10,20,10] (:13-16,29-30), zero warm start (:26-27,35), tol 1e-4 (:19).
 * The iteration cap defaults to AMK_MPC_DEFAULT_MAX_ITER, not to the reference's ipopt.max_iter = 10 (:20): that
 * number counts IPOPT's iterations, which this library does not reproduce.  The default is a
 * safety net, not a budget: a solve runs until its last barrier problem is solved to tol (mean 10 / 21 / 25 iterations
 * at N = 10 / 20 / 30, the slowest of 1024 + 64 + 256 bench scenes 54; a cap of 40 cut 1.5 % of the cold starts short).
 * A solve that does hit the cap reports status 1 (amk_mpc_solve: info[0]; amk_step_batch: flags[2]).  Real-time hosts:
 * in a batched launch the slowest scene sets the launch time, and a non-converging scene runs to the cap (~25 us per
 * iteration and solve when the chip is full); set a tighter cap per handle (amk_mpc_set_solver_options; for a pipeline:
 * on amk_pipeline_mpc(p, slot) of every slot) and fall back to PubSlowDownCmd when flags[2] > 0.      */
int amk_mpc_create(double T, double dt, int nearest_point_num, int n_scenes, amk_mpc **out);
int amk_mpc_destroy(amk_mpc *mpc);
int amk_mpc_horizon(const amk_mpc *mpc);   /* N                                                  */
int amk_mpc_nx(const amk_mpc *mpc);        /* 10 + 14 N                                          */
int amk_mpc_ref_len(const amk_mpc *mpc);   /* 20 + 10 N + 3 K N  (what GetRefStates produces)    */

int amk_mpc_setup_weights(amk_mpc *mpc, const double *h_weights25); /* SetupWeights  .cpp:58-60  */
int amk_mpc_setup_tau(amk_mpc *mpc, const double *h_tau4);          /* SetupTau      .cpp:61-63  */
int amk_mpc_setup_gains(amk_mpc *mpc, const double *h_gains4);      /* SetupGains    .cpp:67-69  */
/* The generator's use_drag_coefficient switch (mpc_obstacle_casadi.py:95-105, mpc_parameters.yaml:4; off by default, hard-coded 0.033).
 * Its expression `rotmat * diag(k, k, k) * rotmat.T * (vx, vy, vz)`, read as the matrix products it describes, is k v whatever the attitude
 * (R (k I) R' = k I): v' = a - k .* v, linear in the state, so the RK4 map stays affine with the same sparsity and the solver is unchanged
 * -- only A differs.  (As written, with CasADi's element-wise `*`, a 3 x 3 times a 3 x 1 is a dimension error; the reference cannot have
 * exercised it, and it cannot be checked here: CasADi is absent.)  One coefficient per world axis, >= 0; 0, 0, 0 = the reference's default. */
int amk_mpc_set_drag_coefficient(amk_mpc *mpc, double kx, double ky, double kz);
int amk_mpc_set_drone_radius(amk_mpc *mpc, double radius);          /* SetDroneRadius .cpp:64-66 */
int amk_mpc_set_drone_accel_limits(amk_mpc *mpc, double aMinZ, double aMaxZ, double aMaxXy,
                                   double aMaxYawDot);              /* .cpp:70-92                */
/* ipopt.tol of HighLvlMpc.cpp:19 and the iteration cap of this library's method (see amk_mpc_create) */
int amk_mpc_set_solver_options(amk_mpc *mpc, double tol, int max_iter);
/* Scheduling of the solves inside amk_step_batch -- results are unchanged, bit for bit.  budget > 0: a solve launch of the
 * step's first `budget_rounds` rounds (0: mpc_max_iter - 1) ends after `budget` interior-point iterations per scene; a scene
 * whose solve has not converged by then is paused (iterate + multipliers kept on the device) and resumed inside the NEXT
 * round's launch, while the scenes that did converge make their next re-plan pass there -- a launch no longer lasts as long
 * as its slowest scene.  Every scene still runs PlanWapionts -> ProcessWaypoints -> GetRefStates -> Solve to convergence ->
 * refill per pass, in order (AvoidanceStateMachine.cpp:322-344; one Solve per pass, warm start in / out, HighLvlMpc.cpp:93-137);
 * mpc_max_iter catch-up rounds without a budget follow the budgeted ones.  budget = 0 (default): one round per pass, every
 * launch runs its scenes to convergence.  No effect on amk_mpc_solve / amk_step_batch_frames.                              */
int amk_mpc_set_solve_budget(amk_mpc *mpc, int budget, int budget_rounds);
/* Arithmetic of the NLP evaluation and the interior-point method: 64 (default; the reference's CasADi/IPOPT
 * path is fp64 throughout) or 32 (BASELINE.json configs[4], "fp32 tolerance check vs CPU trajectory").  The
 * interface stays double and the KD queries stay fp64 (neighbour indices remain bit-exact); only the solve
 * narrows.  Anything else: AMK_ERR_UNSUPPORTED.                                                              */
int amk_mpc_set_precision(amk_mpc *mpc, int bits);

/* Solve(vecRefStates, u, x0Array, faster) (HighLvlMpc.cpp:93-137) for every scene.
 *   d_ref_states [S][20+10N+3KN]  = [x_init | ref_k | obstacles | target]  (GetRefStates layout,
 *                                    AvoidanceStateMachine.cpp:236-257); gains, taus, weights and
 *                                    radius are appended internally as in .cpp:97-107
 *   d_u          [S][4]           = sol[10..13]                         (.cpp:124-128)
 *   d_x0array    [S][N][14]       = rows [X_k, U_k], k < N              (.cpp:130-136) (may be NULL)
 *   d_info       [S][4] int       = {status (0 converged, 1 iteration cap, 2 regularisation overflow), iterations,
 *                                    regularisations, line-search failures}   (may be NULL)
 * The full primal solution is kept as the next call's warm start (.cpp:110,129).  The reference
 * never inspects the solver status (.cpp:116-122); neither does this function -- it reports it.  */
int amk_mpc_solve(amk_mpc *mpc, const double *d_ref_states, double *d_u, double *d_x0array,
                  int *d_info, int faster, void *stream);

/* mNlpW0 access: [S][nx] in the reference's decision-vector order [X_0,U_0,...,U_{N-1},X_N].    */
int amk_mpc_get_warm_start(amk_mpc *mpc, double *d_w, void *stream);
int amk_mpc_set_warm_start(amk_mpc *mpc, const double *d_w, void *stream);
int amk_mpc_reset_warm_start(amk_mpc *mpc, void *stream); /* back to the constructor's zeros      */

int amk_mpc_solve_host(amk_mpc *mpc, const double *h_ref_states, double *h_u, double *h_x0array,
                       int *h_info, int faster);

/* The NLP functions of the generated solver plugin (mpc_obstacle_casadi.py:290-300; loaded at HighLvlMpc.cpp:50,52),
 * SURVEY.md section 8 rows a14-a18, evaluated for every scene at caller-supplied points:
 *   nlp_f      f(x, p)                  objective                     mpc_obstacle_casadi.py:153-214
 *   nlp_g      g(x, p)                  [X_0 - x_init ; F(X_k,U_k) - X_{k+1}]           :156-160,219,338-357
 *   nlp_grad_f grad_x f                 (d|s|/ds = sign(s), as CasADi differentiates fabs)
 *   nlp_jac_g  dg/dx, CCS values        constant: the dynamics are affine (drag off, mpc_parameters.yaml:4)
 *   nlp_hess_l lam_f * triu(hess_x f)   CCS values; the constraints are linear, lam_g contributes nothing
 *   d_w          [S][nx]    x = [X_0, U_0, X_1, ..., U_{N-1}, X_N]                        :158,164,217,224
 *   d_ref_states [S][nref]  P[0 : 20+10N+3KN]; gains, taus, weights, radius come from the handle (HighLvlMpc.cpp:97-107)
 *   d_lam_f      [S] or NULL (= 1)
 *   d_f [S], d_grad_f [S][nx], d_g [S][ng], d_jac_g [S][jac_nnz], d_hess_l [S][hess_nnz]: each may be NULL.
 * ng = 10 + 10 N; jac_nnz = 10 + 39 N; hess_nnz = 25 (N - 1) + 10 + 4 N.  The CCS patterns
 * (column pointers colind[nx + 1], row indices, rows ascending inside a column) come from the *_sparsity calls.      */
int amk_mpc_ng(const amk_mpc *mpc);
int amk_mpc_jac_nnz(const amk_mpc *mpc);
int amk_mpc_hess_nnz(const amk_mpc *mpc);
int amk_mpc_jac_sparsity(const amk_mpc *mpc, int *h_colind, int *h_row);
int amk_mpc_hess_sparsity(const amk_mpc *mpc, int *h_colind, int *h_row);
int amk_mpc_eval(amk_mpc *mpc, const double *d_w, const double *d_ref_states, const double *d_lam_f, double *d_f,
                 double *d_grad_f, double *d_g, double *d_jac_g, double *d_hess_l, void *stream);
int amk_mpc_eval_host(amk_mpc *mpc, const double *h_w, const double *h_ref_states, const double *h_lam_f, double *h_f,
                      double *h_grad_f, double *h_g, double *h_jac_g, double *h_hess_l);

/* nlp_grad, the sixth function CasADi's generate_dependencies emits for an nlpsol (mpc_obstacle_casadi.py:294-303): with
 * gamma(x, p) = lam_f f + lam_g' g,
 *   d_grad_x [S][nx]  grad_x gamma = lam_f grad f + (dg/dx)' lam_g
 *   d_grad_p [S][np]  grad_p gamma over the WHOLE parameter vector p = [P prefix (nref) | gain(4) | tau(4) | weights(25) |
 *                     radius], np = nref + 34 (:76-85): what nlpsol reports as lam_p.  The tail's values are taken from the
 *                     handle (setters), its derivatives are returned here; d/d gain = d/d tau_yaw = 0 (unused by the model,
 *                     :114-121), d/d tau through the RK4 map (dual-number probe of the same integrator).
 *   d_lam_f [S] or NULL (= 1);  d_lam_g [S][ng] or NULL (= 0).  Either output may be NULL.                                  */
int amk_mpc_np(const amk_mpc *mpc);
int amk_mpc_eval_gamma(amk_mpc *mpc, const double *d_w, const double *d_ref_states, const double *d_lam_f,
                       const double *d_lam_g, double *d_grad_x, double *d_grad_p, void *stream);
int amk_mpc_eval_gamma_host(amk_mpc *mpc, const double *h_w, const double *h_ref_states, const double *h_lam_f,
                            const double *h_lam_g, double *h_grad_x, double *h_grad_p);

/* ------------------------------------------------------------------------------------------ */
/* One control step: TASK branch of AvoidanceStateMachine::Step       AvoidanceStateMachine.cpp */
/* ------------------------------------------------------------------------------------------ */
typedef struct amk_step_params {
    double speed;           /* mSpeed                      mpc_parameters.yaml:46                 */
    double safety_distance; /* mParamSafteyDistance        mpc_parameters.yaml:56                 */
    int mpc_max_iter;       /* mParamMPCMaxIter (<= 8)     mpc_parameters.yaml:3                  */
    int reserved;
} amk_step_params;

#define AMK_MAX_OUTER_ITER 8

/* For every scene: for iter < mpc_max_iter { PlanWapionts (:259-281) ; ProcessWaypoints
 * (:204-235) ; early exit (:333-335) ; GetRefStates (:236-257) ; Solve ; refill the reference path
 * with the predicted states (:338-342) }, all on the device, no host round trip.
 *   obstacle, edge  the dual KD indices of the current frame (FrameKDMap's mCurFrame.pointCloud /
 *                   .edgeCloud, FrameKDMap.cpp:44-50); single-frame map (mVecQueryVector = [cur])
 *   d_state_quad    [S][mpc_max_iter][10]  mVecStateQuad as GetCurStateQuad (:183-203) would give
 *                   it at the start of each outer iteration (the host owns the clock model)
 *   d_pos_x         [S]                    mPos.x() used by GetRefStates (:251)
 *   d_ref_path      [S][N][10] in/out      mRefPath (after GetInitPath on entry)
 *   d_u             [S][4]                 control of the last solve
 *   d_x0array       [S][N][14]             predicted trajectory of the last solve (may be NULL)
 *   d_flags         [S][4] int             {isSafety, solves done, WORST solver status over the step's
 *                                           solves (0 converged, 1 iteration cap, 2 regularisation
 *                                           overflow; -1 no solve ran), total interior-point
 *                                           iterations}.  The reference ignores IPOPT's status
 *                                           (HighLvlMpc.cpp:116-122); a host that wants to fall back to
 *                                           PubSlowDownCmd on an unconverged solve tests flags[2] > 0.  */
int amk_step_batch(amk_kd *obstacle, amk_kd *edge, amk_mpc *mpc, const amk_step_params *params,
                   const double *d_state_quad, const double *d_pos_x, double *d_ref_path,
                   double *d_u, double *d_x0array, int *d_flags, void *stream);

/* The same control step over a MULTI-FRAME map: mVecQueryVector = [current frame, keyframes ...] (FrameKDMap.cpp:64-74).
 * Every query follows FrameKDMap::QueryNearest (:322-376): the current frame alone when it holds >= k points and the query
 * projects into the current image (PtIsInFrame, :215-231), otherwise the k' = min(k, size_f) nearest of every frame merged by
 * squared distance; GetNearestDistance (:400-427) is the minimum over the frames.  obstacle[f] / edge[f]: the dual KD indices of
 * frame f, f = 0 the current frame; n_frames <= AMK_MAX_FRAMES; every handle holds the same number of scenes as `mpc`.
 *   d_Twc [S][16]  mCurFrame.Twc (world <- camera, row-major) per scene; NULL: every query counts as inside the current frame
 *   cam            camera model of PtIsInFrame: intrinsics ALREADY divided by the resize scale (:21-24), depth_max, and the
 *                  size of the down-scaled image (mParamWidth / mParamHeight, :106-107)
 * Remaining arguments as amk_step_batch.  With n_frames = 1 and d_Twc = NULL the results equal amk_step_batch's.            */
#define AMK_MAX_FRAMES 16
#define AMK_MAX_MAP_FRAMES 101   /* frames of a keyframe map (amk_kfmap): current + max_frame_count <= 100 (mpc_parameters.yaml:73) */
typedef struct amk_frame_camera {
    double fx, fy, cx, cy;
    double depth_max;
    int width, height;
} amk_frame_camera;
int amk_step_batch_frames(amk_kd *const *obstacle, amk_kd *const *edge, int n_frames, const double *d_Twc,
                          const amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *params,
                          const double *d_state_quad, const double *d_pos_x, double *d_ref_path, double *d_u,
                          double *d_x0array, int *d_flags, void *stream);

/* Same with host buffers (stages through device memory, synchronises); what a single-robot host
 * (S = 1) calls once per control period.                                                         */
int amk_step_batch_host(amk_kd *obstacle, amk_kd *edge, amk_mpc *mpc, const amk_step_params *params,
                        const double *h_state_quad, const double *h_pos_x, double *h_ref_path,
                        double *h_u, double *h_x0array, int *h_flags);

/* ------------------------------------------------------------------------------------------ */
/* Several control steps in flight on one GPU (the throughput configuration)                    */
/* ------------------------------------------------------------------------------------------ */
/* One step (both index builds of a fresh frame + amk_step_batch) is a chain of dependent launches around a latency-bound
 * solve: a single stream reaches ~15 % of what the chip does with 20 independent steps in flight.
 * A pipeline owns n_slots slots = {HIP stream, obstacle + edge amk_kd, amk_mpc, reference-path buffer, outputs}; submit()
 * enqueues on the next slot (round robin) -- per frame the reference's sequence FrameKDMap::AddVertex (FrameKDMap.cpp:34-52: the two
 * InitializeNew) -> AvoidanceStateMachine::Step TASK branch (AvoidanceStateMachine.cpp:322-355) -- and returns at once;
 * it blocks only when queue_depth steps of that slot are still unfinished.  The host should export GPU_MAX_HW_QUEUES >= n_slots
 * before the first HIP call (ROCm multiplexes streams onto 4 hardware queues by default; two streams on one queue
 * serialise).  Input buffers belong to the caller and must stay valid until the slot has finished.  They must also be
 * COMPLETE: a slot runs on its own non-blocking stream, which is not ordered against the stream that produced the inputs, and
 * with a gang the inputs are only read when the gang is launched (the G-th submit, wait or drain).  Either synchronise the
 * producing stream before submit(), or record an event on it and pass it as amk_pipeline_frame.input_ready -- the slot's stream
 * then waits for it on the device (hipStreamWaitEvent), the host does not.                                                */
typedef struct amk_pipeline amk_pipeline;
/* FrameKDMap's perception parameters (used by amk_depth_to_cloud / _edge_cloud below and by pipeline frames that start at the
 * raw depth image).                                                                                                       */
typedef struct amk_depth_params {
    double pixel2meter;   /* mParamPixel2Meter   mpc_parameters.yaml:64                            */
    double depth_min;     /* mParamDepthMin      :66                                               */
    double depth_max;     /* mParamDepthMax      :65                                               */
    double resize_scale;  /* mParamDepthScale    :63  (output is cols/scale x rows/scale, truncated) */
    double fx, fy, cx, cy; /* FULL-resolution intrinsics (:59-62); divided by resize_scale inside,
                             as the FrameKDMap constructor does (FrameKDMap.cpp:21-24)             */
    double Tbc[16];       /* body <- camera, row-major 4x4 (mParamTbc)                             */
} amk_depth_params;
/* What the TASK case does around the re-plan loop (AvoidanceStateMachine.cpp:322-355), for frames submitted in TASK mode.  */
typedef struct amk_task_params {
    double decay;           /* mParamDecay: assumed compute latency, mpc_parameters.yaml:77                                 */
    double iter_time;       /* assumed duration of one re-plan pass (the reference measures it, :329,343); <= 0: decay      */
    double farest_point;    /* goal_x, mpc_parameters.yaml:55 (GetInitPath :31)                                             */
    double height;          /* mHeight, :54                                                                                 */
    double slow_down_kp, slow_down_kd;   /* :79-80  (PubSlowDownCmd :379-397)                                               */
    double a_max_xy, a_max_z;            /* clamp of the slow-down command (:383-388; z clamped to +-aMaxZ as there)        */
    int use_odom_est;       /* mParamIsUseOdomEstimate, :78                                                                 */
    int task;               /* mStrTask of GetInitPath (:29-45): AMK_TASK_FORWARD (0, the launch file's default,             */
                            /* mpc_obstacle_avoidance_sim.launch:9) or AMK_TASK_GLOBAL_GOAL (1): the path's last point walks  */
                            /* towards amk_pipeline_frame.d_global_goal by at most speed * dt per period (:34-45)            */
} amk_task_params;
#define AMK_TASK_FORWARD 0
#define AMK_TASK_GLOBAL_GOAL 1
#define AMK_PIPELINE_MAX_SLOTS 64
#define AMK_PIPELINE_DEFAULT_DEPTH 3
#define AMK_PIPELINE_MAX_DEPTH 64
#define AMK_PIPELINE_MAX_GANG 8
/* FrameKDMap's keyframe parameters (the batched map itself: amk_kfmap_*, below) */
typedef struct amk_kfmap amk_kfmap;
typedef struct amk_kfmap_params {
    int max_frame_count;       /* mParamMaxFrameCount, mpc_parameters.yaml:73; 1 ... AMK_MAX_MAP_FRAMES - 1; 0 (pipeline      */
                               /* config only): no keyframe map, mVecQueryVector = [current]                                  */
    int keyframe_th_count;     /* mParamKeyframeCountTh, :72 (>= 1)                                                           */
    double keyframe_th_dist;   /* mParamKeyframeDistanceTh, :71                                                               */
    double depth_min;          /* mParamDepthMin (DroneBehindPts :247)                                                        */
    double Tbc[16];            /* mParamTbc, row-major (amk_pipeline_config: all zero = take amk_depth_params.Tbc, and if that  */
                               /* is all zero too the identity: camera frame = body frame)                                    */
} amk_kfmap_params;
typedef struct amk_pipeline_config {
    int n_slots;            /* independent steps in flight                                                              */
    int n_scenes;           /* scenes per step (S of every handle)                                                      */
    int max_points;         /* capacity per scene of the obstacle cloud ...                                             */
    int max_edge_points;    /* ... and of the edge cloud                                                                */
    double T, dt;           /* ObstacleAvoidanceMPC(T, dt, .)                                     HighLvlMpc.cpp:5-12  */
    int nearest_point_num;  /* K                                                                  mpc_parameters.yaml:5 */
    int queue_depth;        /* steps that may be queued per slot before submit() blocks (0 = AMK_PIPELINE_DEFAULT_DEPTH).   */
                            /* A slot's steps run in order on its stream; with depth >= 2 the next step is already queued   */
                            /* when one ends (no host round trip between them).  Results of a step must be read -- wait(),  */
                            /* outputs() -- before a later submit() on the same slot overwrites them, or go to d_u_out.      */
    int gang;               /* frames per launch (0 / 1 = every frame has its own launches; <= AMK_PIPELINE_MAX_GANG).       */
                            /* G > 1: a slot's handles hold G x n_scenes scenes and G consecutive submit()s share one set   */
                            /* of launches (both index builds of all G frames in one launch, one amk_step_batch); a frame   */
                            /* is staged until its gang is full -- wait() / drain() launch a partly filled gang.  Results   */
                            /* are those of separate launches, bit for bit (scenes are independent).  On the bench workload */
                            /* 10 slots x 4 frames of 256 scenes beat 20 slots x 1 by 11 %.           */
    amk_step_params step;
    amk_task_params task;   /* only read for frames submitted with d_odom (TASK mode, below)                                */
    amk_depth_params depth; /* only read for frames submitted with d_depth (below)                                          */
    amk_kfmap_params keyframes;  /* max_frame_count > 0: every slot keeps a keyframe map (amk_kfmap) over its gang x n_scenes    */
                            /* scenes and every frame runs AddVertex -> KeyframeThreadWorker's body -> the step over            */
                            /* mVecQueryVector = [current, keyframes ...] (the reference's default regime: FrameKDMap.cpp:29-32, */
                            /* 64-74).  The scene at position g of slot s must be the same robot every period (as in TASK mode).*/
                            /* PtIsInFrame: for depth frames the camera is amk_pipeline_config.depth's, down-scaled (:21-24,     */
                            /* 106-107); for cloud frames amk_pipeline_frame.camera + d_Twc_cur (BOTH required then, else         */
                            /* AMK_ERR_INVALID_ARG: mCurFrame.Twc also feeds DroneBehindPts, and without a camera no keyframe    */
                            /* would ever be merged).  The frames of one gang must agree on depth-vs-cloud, the depth image      */
                            /* size and the camera (AMK_ERR_INVALID_ARG): the map step takes one camera model per launch.        */
                            /* keyframes.Tbc is ignored: depth.Tbc is used.  0: single-frame map.                                */
                            /* A TASK frame that brings d_ref_path_init (a robot that starts over) also resets its scenes' map.   */
} amk_pipeline_config;
typedef struct amk_pipeline_frame {
    const float *d_cloud;          /* [S][max_points][point_stride]      obstacle cloud of the frame (NULL with d_depth)*/
    const int *d_cloud_counts;     /* [S] or NULL (= max_points)                                                        */
    const float *d_edge;           /* [S][max_edge_points][point_stride] edge cloud                                     */
    const int *d_edge_counts;      /* [S] or NULL                                                                       */
    int point_stride;              /* 3, 4 (pcl::PointXYZ) or 0 (= 3)                                                   */
    int keep_warm_start;           /* 0: zero warm start (a fresh ObstacleAvoidanceMPC, HighLvlMpc.cpp:26-27,35);        */
                                   /* 1: the slot's last solution (mNlpW0 = sol, :129)                                  */
    const double *d_state_quad;    /* [S][mpc_max_iter][10]  as amk_step_batch                                          */
    const double *d_pos_x;         /* [S]                                                                               */
    const double *d_ref_path_init; /* [S][N][10]  mRefPath after GetInitPath; copied, the slot's copy is refilled       */
    double *d_u_out;               /* [S][4] or NULL: where the control goes instead of the slot's own buffer (e.g. a   */
                                   /* row of the sweep's result array that amk_shard_gather exchanges at the end)       */
    /* TASK mode: the slot keeps the state machine's persistent state -- mRefPath next to mNlpW0 -- and the caller supplies  */
    /* per control period what the reference's callbacks supply: the frame and the odometry.  With d_odom != NULL the slot   */
    /* runs, on the device, GetInitPath (:24-54, task "forward") on ITS OWN mRefPath (position g of slot s keeps the path of */
    /* the frame submitted there last: submit the same robots in the same order every period), GetCurStateQuad (:183-203)    */
    /* for every re-plan pass (odom_age + decay for pass 0, odom_age + (i + 1) * iter_time for pass i >= 1), the step, PubCmd /  */
    /* PubSlowDownCmd (:345-350,369-397).                                                                                   */
    /* d_state_quad / d_pos_x are then ignored (may be NULL); d_ref_path_init != NULL first (re)sets mRefPath               */
    /* (InitCircleState :14-23 or any re-initialisation) BEFORE GetInitPath, NULL keeps the slot's -- which must exist: the   */
    /* first TASK frame at a position, and the first one after a launch of the slot failed half-way (the slot's persistent   */
    /* state may have been shifted / reset by what was enqueued before the failure), must bring it (AMK_ERR_INVALID_ARG).     */
    const double *d_odom;          /* [S][10] or NULL: [mPos(3), yaw, mVel(3), mAcc(3)] as the callbacks left them       */
    double odom_age;               /* now - mTimePos at the start of the step, seconds (:183-184); 0 = fresh odometry    */
    double *d_cmd_out;             /* [S][3] or NULL: Command.acceleration -- u[0..2] when isSafety, else the slow-down  */
                                   /* command (TASK mode only)                                                           */
    /* Frames that start where FrameKDMap::AddVertex starts (FrameKDMap.cpp:34-52): the raw depth image.  With d_depth != NULL  */
    /* d_cloud / d_edge are ignored (may be NULL) and the slot runs ProcessDepth (:90-130) and BuildEdgeCloud (:176-214) on the */
    /* device with amk_pipeline_config.depth -- the edge cloud through the slot's own mCurFrame.Twc (the PREVIOUS frame's        */
    /* Twb * Tbc of the same position, identity at first, as the reference: :50,209) -- then both index builds; a scene whose  */
    /* frame yields no obstacle point keeps its previous frame and its Twc (:39-41).  The down-scaled image must fit the       */
    /* pipeline's capacities: (cols / resize_scale) * (rows / resize_scale) <= max_points and <= max_edge_points.              */
    const void *d_depth;           /* [S][rows][cols] uint16 / float32 (depth_type: AMK_DEPTH_U16 / AMK_DEPTH_F32) or NULL */
    int depth_type, depth_rows, depth_cols, reserved;
    const double *d_Twb;           /* [S][16] world <- body, row-major (DepthCallback, AvoidanceStateMachine.cpp:153-164) */
    /* A multi-frame map: mVecQueryVector = [this frame, keyframes ...] (FrameKDMap.cpp:64-74).  With n_keyframes > 0 the slot  */
    /* runs amk_step_batch_frames over [its own two indices of this frame, kf_obstacle[i] / kf_edge[i] ...] instead of          */
    /* amk_step_batch (PtIsInFrame fast path, per-frame merge, minimum distance over the frames: :215-231,254-427).  The         */
    /* keyframe handles are the caller's (built with amk_kd_build, maintained with amk_kd_keyframe_sweep), hold n_scenes        */
    /* scenes and must not be rebuilt while the frame is in flight; gang == 1 only (AMK_ERR_UNSUPPORTED otherwise);             */
    /* n_keyframes <= AMK_MAX_FRAMES - 1.  d_Twc_cur: mCurFrame.Twc per scene for PtIsInFrame (NULL: the slot's own Twc for a   */
    /* depth frame, else every query counts as inside the current frame); camera: host pointer, copied at submit (NULL: no     */
    /* frustum test).                                                                                                        */
    amk_kd *const *kf_obstacle;
    amk_kd *const *kf_edge;
    int n_keyframes, reserved2;
    const double *d_Twc_cur;
    const struct amk_frame_camera *camera;
    void *input_ready;             /* hipEvent_t or NULL: recorded by the caller on the stream that produces this       */
                                   /* frame's inputs; the slot's stream waits for it before it reads them.  The event   */
                                   /* must stay alive (and must not be re-recorded) until the frame has been launched   */
    const double *d_global_goal;   /* [S][3] or NULL: mStateGlobalGoal of every scene (GlobalGoalCallback, :166-172), read by   */
                                   /* GetInitPath when amk_task_params.task == AMK_TASK_GLOBAL_GOAL; NULL = the constructor's   */
                                   /* {0, 0, height} (:22)                                                                   */
} amk_pipeline_frame;
int amk_pipeline_create(const amk_pipeline_config *cfg, amk_pipeline **out);
int amk_pipeline_destroy(amk_pipeline *p);
int amk_pipeline_slots(const amk_pipeline *p);
int amk_pipeline_gang(const amk_pipeline *p);       /* frames per launch (>= 1)                                            */
/* The slot's handles, to configure them (weights, limits, tie order, precision ...) and its stream.                       */
amk_mpc *amk_pipeline_mpc(amk_pipeline *p, int slot);
amk_kd *amk_pipeline_kd(amk_pipeline *p, int slot, int which /* 0 obstacle, 1 edge */);
amk_kfmap *amk_pipeline_kfmap(amk_pipeline *p, int slot);   /* the slot's keyframe map (NULL without one), e.g. for amk_kfmap_state_host */
void *amk_pipeline_stream(amk_pipeline *p, int slot);
/* submit() hands back a ticket = position_in_gang * n_slots + slot (without a gang: the slot index, as before);
 * ticket % n_slots is the slot, for amk_pipeline_mpc / _kd / _stream.                                                    */
int amk_pipeline_submit(amk_pipeline *p, const amk_pipeline_frame *frame, int *ticket_out);
int amk_pipeline_wait(amk_pipeline *p, int ticket);   /* until that frame's step has finished (launches an open gang)     */
/* Device-side wait: work queued on `stream` after this call runs after the frame's step has finished; the host does not
 * block (launches an open gang).  A closed loop -- outputs -> the caller's kernels -> next submit(input_ready) -- then never
 * synchronises with the host.  The slot's NEWEST launch is the one waited for: call it before the next submit on that slot. */
int amk_pipeline_wait_stream(amk_pipeline *p, int ticket, void *stream);
int amk_pipeline_query(amk_pipeline *p, int ticket);  /* 1 finished / idle, 0 running or staged, -1 error (also: the slot's     */
                                                      /* newest launch failed half-way and dropped its frames; wait() says why) */
int amk_pipeline_drain(amk_pipeline *p);              /* wait for every slot                                              */
/* Device pointers of the frame's results (valid after wait): u [S][4], x0array [S][N][14], flags [S][4], ref_path [S][N][10] */
int amk_pipeline_outputs(amk_pipeline *p, int ticket, double **d_u, double **d_x0array, int **d_flags, double **d_ref_path);

/* ------------------------------------------------------------------------------------------ */
/* The multi-frame map with keyframes, batched: FrameKDMap's keyframe list on the device       */
/* ------------------------------------------------------------------------------------------ */
/* The reference's keyframe thread is on by default (max_frame_count = 100, only_trust_vel = false: FrameKDMap.cpp:29-32),
 * and every QueryNearest / GetNearestDistance of a control step runs over mVecQueryVector = [current frame, every keyframe
 * but the newest] (:64-74,322-427).  amk_kfmap is that map for S scenes at once -- every scene with its own deque of
 * keyframes -- entirely on the device: per control period
 *   amk_kfmap_add_vertex   FrameKDMap::AddVertex after ProcessDepth (:39-51) for the scenes whose frame is not empty
 *   amk_kfmap_update       one pass of KeyframeThreadWorker's body (:443-486): first keyframe / pop while the list is longer
 *                          than max_frame_count or DroneBehindPts fails for the oldest (:233-252) / n x 1-NN sweep of the
 *                          newest keyframe against the current frame, rebuild from >= keyframe_th_count outliers farther than
 *                          keyframe_th_dist, InsertKeyFrame
 *   amk_kfmap_step         the TASK branch over the map (amk_step_batch_frames' semantics: PtIsInFrame fast path, per-frame
 *                          k' = min(k, size), merge, minimum distance over the frames)
 * all stream-ordered, no host round trip.  A keyframe is not a copy: every scene owns max_frame_count + 2 physical slots in
 * two pool indices, its current frame is built into a slot no keyframe holds, the deque is a list of slot numbers.
 * amk_pipeline runs the three calls inside a TASK-mode slot when amk_pipeline_config.keyframes.max_frame_count > 0
 * (any gang).  Twb = Twc * Tbc^-1 of DroneBehindPts uses the rigid inverse of Tbc.                                          */
/* Memory.  The pools are allocated at creation, at full capacity: with cap(p) = round_up(p, 256) + 1024 points per pool scene
 *   bytes ~ (max_frame_count + 2) x n_scenes x [ 16 cap(max_points) + 16 cap(max_edge_points) + 13 cap(max_points) + directories ]
 *           + n_scenes x 2 x 16 cap(max_points)   (the sweep's grids: two generations per scene)    (amk_kfmap_pool_bytes)
 * i.e. ~ 30 B per obstacle point per slot + the edge pool: the reference's max_frame_count = 100 at its own 3072-point frames is 20 MB per
 * robot, at 50 k-point frames 170 MB per robot.  A flight holds ~ 6 frames; slots a scene never uses are still reserved.
 * amk_kfmap_create compares the figure with the device's free memory first and returns AMK_ERR_UNSUPPORTED (message with both
 * numbers on stderr) instead of failing half-way through the allocations with AMK_ERR_HIP.
 * Call order.  add_vertex -> update -> step per control period is what amk_pipeline issues.  add_vertex alone already points the
 * query vector's first frame at the new trees (SetCurPtCloud -> UpdateQueryVector, FrameKDMap.cpp:53-58); the keyframe rows of the
 * query vector are the worker's (update).                                                                                    */
int amk_kfmap_pool_bytes(int n_scenes, int max_points, int max_edge_points, int max_frame_count, long long *bytes_out);
int amk_kfmap_create(int n_scenes, int max_points, int max_edge_points, const amk_kfmap_params *prm, amk_kfmap **out);
int amk_kfmap_destroy(amk_kfmap *map);
/* scenes [first_scene, first_scene + n_scenes) start over: no current frame, no keyframe, Twc = identity (stream-ordered) */
int amk_kfmap_reset(amk_kfmap *map, int first_scene, int n_scenes, void *stream);
int amk_kfmap_scenes(const amk_kfmap *map);
int amk_kfmap_frames(const amk_kfmap *map);          /* frames of the query vector: 1 + max_frame_count                      */
const double *amk_kfmap_twc(const amk_kfmap *map);   /* device [S][16]: mCurFrame.Twc (what BuildEdgeCloud multiplies with, :209) */
/* Scenes [first_scene, first_scene + n_scenes): d_xyz [n][max_points][stride] / d_counts [n] (NULL = max_points) and the edge
 * cloud likewise, d_Twc [n][16] = mat4Twb * mParamTbc of the frame.  A scene whose count is 0 keeps its map untouched (:39-41). */
int amk_kfmap_add_vertex(amk_kfmap *map, int first_scene, int n_scenes, const float *d_xyz, const int *d_counts,
                         const float *d_edge_xyz, const int *d_edge_counts, int point_stride, const double *d_Twc, void *stream);
int amk_kfmap_update(amk_kfmap *map, void *stream);
/* cam == NULL: no frustum test (every query counts as inside the current frame, as amk_step_batch_frames with d_Twc = NULL) */
int amk_kfmap_step(amk_kfmap *map, const struct amk_frame_camera *cam, amk_mpc *mpc, const amk_step_params *prm,
                   const double *d_state_quad, const double *d_pos_x, double *d_ref_path, double *d_u, double *d_x0array,
                   int *d_flags, void *stream);
/* Introspection (synchronises): per scene mKeyFrameMap.size(), mVecQueryVector.size(), the outliers of the last sweep and --
 * h_frame_sizes [S][amk_kfmap_frames()] or NULL -- the obstacle-cloud size of every query frame (-1 behind the scene's last). */
int amk_kfmap_state_host(amk_kfmap *map, int *h_n_keyframes, int *h_n_query_frames, int *h_last_outliers, int *h_frame_sizes);

/* ------------------------------------------------------------------------------------------ */
/* Scenes sharded over the GPUs of a node (one process per GPU), RCCL over xGMI                 */
/* ------------------------------------------------------------------------------------------ */
/* Scenes are independent (the reference runs a single instance, mpc_obstacle_avoidance_node.cpp:8): block partition, no
 * data-path collective; the one exchange step is an ncclAllGather of the controls.  Rank 0 makes the id, the host
 * distributes its AMK_SHARD_ID_BYTES bytes by its own means (MPI, TCP, a file), every rank calls amk_shard_create with
 * its GPU current.  RCCL is bound at run time; AMK_ERR_UNSUPPORTED when librccl cannot be loaded.                        */
typedef struct amk_shard amk_shard;
#define AMK_SHARD_ID_BYTES 128
int amk_shard_scene_range(int rank, int world, int total, int *first, int *count);
int amk_shard_unique_id(char *id_out /* [AMK_SHARD_ID_BYTES] */);
int amk_shard_create(const char *id, int rank, int world, amk_shard **out);
int amk_shard_destroy(amk_shard *s);
int amk_shard_rank(const amk_shard *s);
int amk_shard_world(const amk_shard *s);
int amk_shard_last_rccl_error(void);
/* d_all[r * n + i] = rank r's d_local[i]; stream-ordered.  ncclAllGather needs the SAME count on every rank: when
 * total % world != 0 the counts of amk_shard_scene_range differ by one, so every rank passes the PADDED shard size
 * amk_shard_padded_count(world, total) = ceil(total / world) (its buffers sized for it; rows beyond its own count are
 * padding) -- passing the natural `count` would hang or corrupt d_all.  amk_shard_gather_u: n_local_scenes is that padded
 * size too (4 doubles per scene).                                                                                      */
int amk_shard_padded_count(int world, int total);
int amk_shard_gather(amk_shard *s, const double *d_local, long long n_doubles_per_rank, double *d_all, void *stream);
int amk_shard_gather_u(amk_shard *s, const double *d_u_local, int n_local_scenes, double *d_u_all, void *stream);
int amk_shard_max(amk_shard *s, double *d_values, int n, void *stream);   /* in-place max over ranks (timing)              */
/* Which RCCL was bound (the copy already in the process -- e.g. PyTorch's -- if there is one, else librccl.so.1 from the loader
 * path): file of ncclAllGather, ncclGetVersion's code, 1 if it had been loaded before this library asked.  A host prints
 * this next to its results so that the first multi-GPU run can be diagnosed (bench.py: config.rccl).                       */
int amk_shard_rccl_info(char *path, int path_len, int *version, int *was_loaded);
/* Watchdog: host-side wait (polling, <= timeout_s) for everything queued on `stream`, the gathers included.
 * AMK_ERR_TIMEOUT: a collective hangs (a rank that never entered it, a link, two RCCLs in one process): exit and re-run with
 * NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL.                                                                             */
int amk_shard_wait(amk_shard *s, void *stream, double timeout_s);

/* ------------------------------------------------------------------------------------------ */
/* Depth image -> obstacle cloud: FrameKDMap::ProcessDepth            FrameKDMap.cpp:75-138   */
/* (the step immediately before the tree build)                   */
/* ------------------------------------------------------------------------------------------ */
/* amk_depth_params: defined with the pipeline, above */

#define AMK_DEPTH_U16 0   /* CV_16UC1 */
#define AMK_DEPTH_F32 1   /* CV_32FC1 */

/* Width / height of the down-scaled image = capacity of the output cloud per scene (FrameKDMap.cpp:106-107). */
int amk_depth_out_size(int rows, int cols, double resize_scale, int *out_w, int *out_h);

/* For every scene: inverse depth of the raw pixels with the range gate (GetInvDepthImg, :76-89), bilinear
 * down-scale (cv::resize with dsize given: its 4th positional argument cv::INTER_MAX lands in `fx`, so the
 * interpolation is the default INTER_LINEAR, :109), back-projection of every pixel with inverse depth >= 1e-2
 * and min < depth < max through the scaled intrinsics (:110-118,131-138), transform by Twb * Tbc (:119-120),
 * points appended in row-major pixel order as float32 (:122).
 *   d_depth   [S][rows][cols] uint16 or float32, scene_stride in ELEMENTS
 *   d_Twb     [S][16] world <- body, row-major
 *   d_cloud   [S][W*H][point_stride] float32 out (point_stride 3 or 4: feeds amk_kd_build directly),
 *             cloud_scene_stride in floats;  d_counts [S] out: points written per scene.
 * Arithmetic contract: DESIGN.md section 10 (float bilinear taps in OpenCV's order without FMA, double
 * back-projection and transform without FMA); bit-exact against oracle/depth_oracle.c.  Parity with OpenCV's
 * own resize and Eigen's -march=native products is unpinned (neither is in the image).                       */
int amk_depth_to_cloud(const void *d_depth, int depth_type, int rows, int cols, long long scene_stride,
                       int n_scenes, const amk_depth_params *params, const double *d_Twb, float *d_cloud,
                       int point_stride, long long cloud_scene_stride, int *d_counts, void *stream);
int amk_depth_to_cloud_host(const void *h_depth, int depth_type, int rows, int cols, long long scene_stride,
                            int n_scenes, const amk_depth_params *params, const double *h_Twb, float *h_cloud,
                            int point_stride, long long cloud_scene_stride, int *h_counts);

/* Depth image -> edge cloud: FrameKDMap::BuildEdgeCloud (FrameKDMap.cpp:176-214), the
 * input of the Edge-KD-tree.  On the down-scaled inverse-depth image of amk_depth_to_cloud: 8-bit quantisation
 * uchar(depth / (max - min) * 200), 255 where invalid (:180-193); cv::erode with a 3x3 kernel (:194);
 * cv::Canny(img, 0.1, 0.3) (:196: aperture 3, L1 norm, thresholds floor to 0, so every pixel that survives the
 * non-maximum suppression with a non-zero Sobel magnitude is an edge); the edge pixels are back-projected at their
 * QUANTISED, ERODED depth (:199-200) and transformed by d_Twc * Tbc, appended in row-major order.
 *   d_Twc [S][16]  the matrix the reference multiplies with Tbc at :209: mCurFrame.Twc, i.e. the PREVIOUS frame's
 *                  Twb * Tbc (it is updated only after ProcessDepth, :50).  Pass Twb of the current frame instead to
 *                  get what the authors presumably meant.
 * The edge cloud of a scene is empty when its obstacle cloud is (:126-128).  Down-scaled images of more than
 * AMK_EDGE_MAX_PIXELS pixels: AMK_ERR_UNSUPPORTED.  Bit-exact against oracle/depth_oracle.c; parity with OpenCV's
 * own erode / Canny is unpinned.                                                          */
#define AMK_EDGE_MAX_PIXELS 16000
int amk_depth_to_edge_cloud(const void *d_depth, int depth_type, int rows, int cols, long long scene_stride,
                            int n_scenes, const amk_depth_params *params, const double *d_Twc, float *d_cloud,
                            int point_stride, long long cloud_scene_stride, int *d_counts, void *stream);
int amk_depth_to_edge_cloud_host(const void *h_depth, int depth_type, int rows, int cols, long long scene_stride,
                                 int n_scenes, const amk_depth_params *params, const double *h_Twc, float *h_cloud,
                                 int point_stride, long long cloud_scene_stride, int *h_counts);

#ifdef __cplusplus
}
#endif
#endif /* AVOID_MPC_AMD_H */
