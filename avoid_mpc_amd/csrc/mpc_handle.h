// Host-side handle of a batch of ObstacleAvoidanceMPC objects (AM/include/HighLvlMpc.h:4-33) and the
// device workspace of the fused control step.  Shared by mpc_solve.hip and step.hip.
#pragma once
#include <vector>
#include "mpc_device.h"

struct amk_mpc {
    double T = 0, dt = 0;
    int N = 0, K = 0, S = 0, nx = 0, nref = 0;
    int run_scenes = 0;       // internal (amk_pipeline, a gang that is not full): launches cover scenes [0, run_scenes); 0 = all S
    int launch_scenes() const { return run_scenes > 0 && run_scenes < S ? run_scenes : S; }
    double h_prm[amk::PRM_LEN];
    double drag[3] = {0.0, 0.0, 0.0};   // amk_mpc_set_drag_coefficient: v' = a - drag .* v (host-side: only A, B, c see it)
    amk::SolveOpts opt;
    size_t lds_bytes = 0;     // fp64 scratchpad; the fp32 kernels take half
    int precision = 64;       // arithmetic of the solve: 64 (default) or 32 (amk_mpc_set_precision)
    amk::DevBuf<double> prm;  // [PRM_LEN]
    amk::DevBuf<double> w0;   // [S][nx]  mNlpW0
    amk::DevBuf<double> ybuf; // [S][N][K][4]  per collision term: its two multipliers + the geometry cache of the derivative pass (scratch of one solve)
    amk::DevBuf<double> gains; // [S][N][GAIN_STAGE = 32] Riccati feedback gains, structural zeros left out (scratch of one interior-point iteration)
    amk::DevBuf<double> plan_coef;  // item coefficients + lane-role constants of the Riccati plan (mpc_device.h)
    amk::DevBuf<int> plan_meta;     // [PLAN_ITEMS] PlanItemMeta + [64] LaneRole
    // staging for amk_mpc_eval_host
    amk::DevBuf<double> ev_w, ev_ref, ev_out;
    // amk_mpc_eval_gamma: d(A, B, c)/d tau ([3][150], recomputed when tau changes) and its host staging
    amk::DevBuf<double> ev_dtau, ev_gam;
    double ev_dtau_tau[4] = {0, 0, 0, 0};
    bool ev_dtau_valid = false;
    // staging for amk_mpc_solve_host
    amk::DevBuf<double> st_ref, st_u, st_x0;
    amk::DevBuf<int> st_info;
    // control-step workspace (allocated by the first amk_step_batch)
    amk::DevBuf<float> knn_pts;     // [S][N][K][3] neighbours of every reference point
    amk::DevBuf<double> knn_d2;     // [S][N][K]
    amk::DevBuf<float> edge_pt;     // [S][3]       nearest edge point of reference point 0
    amk::DevBuf<double> edge_d2;    // [S]
    amk::DevBuf<double> ref_states; // [S][nref]    vecRefStates handed to Solve
    amk::DevBuf<int> done;          // [S]          1: scene left the re-plan loop (:333-335); 2: its solve is paused (iteration budget)
    amk::DevBuf<double> resume_rec; // [S][8 N + 8] what a paused solve needs beside w0 and ybuf (mpc_device.h: SolveSched)
    int solve_budget = 0;           // amk_mpc_set_solve_budget: iterations per launch in the first budget_rounds rounds of a step (0: off)
    int budget_rounds = 0;
    // per-frame raw query results of amk_step_batch_frames (allocated by its first call)
    int mf_frames = 0;
    amk::DevBuf<float> mf_knn_pts, mf_edge_pt;    // [F][S][N][K][3], [F][S][3]
    amk::DevBuf<double> mf_knn_d2, mf_edge_d2;    // [F][S][N][K],    [F][S]
    amk::DevBuf<char> mf_exact;                   // step_frames.hip: FrameExact, when a frame is in AMK_TIES_NANOFLANN mode
    std::vector<char> mf_exact_host;
    // staging for amk_step_batch_host
    amk::DevBuf<double> sh_sq, sh_posx, sh_ref, sh_u, sh_x0;
    amk::DevBuf<int> sh_flags;
};

namespace amk {
// Launches the solve for every scene of `m` (skipping scenes with d_done[s] != 0 when given).
// d_ref_path / d_step_flags: control-step mode (refill mRefPath, count solves / iterations).
// budget / max_passes (control step only, amk_step_batch): see mpc_device.h SolveSched; both 0 = the plain solve.
int launch_solve(amk_mpc *m, const double *d_ref_states, double *d_u, double *d_x0array, int *d_info, const int *d_done,
                 double *d_ref_path, int *d_step_flags, hipStream_t stream, int budget = 0, int max_passes = 0);
}  // namespace amk
