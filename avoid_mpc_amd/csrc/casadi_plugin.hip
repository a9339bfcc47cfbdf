// Drop-in for the reference's generated solver plugin.
//
// The reference generates `so/mpc_obstacle_v2.so` with CasADi's `solver.generate_dependencies("nmpc_v0.c")`
// (AM/tools/mpc_obstacle_casadi.py:290-300) and hands its path to `casadi::nlpsol("solve", "ipopt", soPath, opts)`
// (AM/src/HighLvlMpc.cpp:50,52; ROS parameter `mpc_so`, AM/launch/mpc_obstacle_avoidance_sim.launch:7,75).  CasADi
// dlopen()s the file and looks up, for each of nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l, the entry points of
// its code-generation C API.  This file exports exactly those symbols from libavoid_mpc_amd.so, so the stock node can
// be started with `mpc_so:=<path>/libavoid_mpc_amd.so`; every call is evaluated on the GPU by amk_mpc_eval (one
// scene).  The generated code bakes N = int(T/dt) and K = nearest_point_num in (:36-37,76-85); here they come from
// amk_plugin_configure(T, dt, K) or, without it, from the environment at the first call:  AMK_MPC_T (default 1.0),
// AMK_MPC_DT (0.033), AMK_MPC_K (3) -- the yaml defaults (AM/config/mpc_parameters.yaml:1-2,5).
//
// NOT VERIFIED AGAINST CASADI: CasADi 3.6.4 is neither in the reference tree nor in this image (SURVEY.md section
// 8(c), appendix B).  The API below follows the generator's convention as recalled there -- signature
// `int f(const double** arg, double** res, long long* iw, double* w, int mem)`, helper names, compressed-column
// sparsity arrays [nrow, ncol, colind..., row...] with the dense shorthand [nrow, ncol, 1] -- and the function
// signatures nlp_f(x,p)->f, nlp_g(x,p)->g, nlp_grad_f(x,p)->(f,grad_f_x), nlp_jac_g(x,p)->(g,jac_g_x),
// nlp_hess_l(x,p,lam_f,lam_g)->triu(hess_gamma_x_x), nlp(x,p)->(f,g), nlp_grad(x,p,lam_f,lam_g)->(f,g,grad_gamma_x,grad_gamma_p);
// arg[i] == NULL is an all-zero input, res[i] == NULL an output that is not wanted.  tests/test_casadi_plugin_gpu.py drives the symbols through
// ctypes the way CasADi's importer would and checks values and patterns against the oracle.
#include "mpc_handle.h"
#include "../../include/avoid_mpc_amd/casadi_plugin.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

typedef amk_casadi_int casadi_int;

namespace {

struct Plugin {
    amk_mpc *mpc = nullptr;
    int N = 0, K = 0, nx = 0, np = 0, ng = 0, nref = 0, nnz_j = 0, nnz_h = 0;
    double tail[34];  // gains, taus, weights, radius last pushed into the handle
    bool have_tail = false;
    std::vector<casadi_int> sp_x, sp_p, sp_g, sp_scalar, sp_jac, sp_hess;
    std::vector<double> f, grad, g, jac, hess, gx, gp, zx, zp, zg;
    int status = AMK_ERR_INVALID_ARG;
    bool configured = false;
};

std::mutex g_mu;   // IPOPT calls from one thread; the lock only orders a configure against a first use
Plugin g_plugin;

int configure_locked(Plugin &P, double T, double dt, int K) {
    if (P.mpc) { amk_mpc_destroy(P.mpc); P.mpc = nullptr; }
    P = Plugin();
    P.configured = true;
    P.status = amk_mpc_create(T, dt, K, 1, &P.mpc);
    if (P.status != AMK_OK) return P.status;
    P.N = amk_mpc_horizon(P.mpc);
    P.K = K;
    P.nx = amk_mpc_nx(P.mpc);
    P.nref = amk_mpc_ref_len(P.mpc);
    P.np = amk_mpc_np(P.mpc);  // nref + gain(4) tau(4) weights(25) radius(1), mpc_obstacle_casadi.py:76-85
    P.ng = amk_mpc_ng(P.mpc);
    P.nnz_j = amk_mpc_jac_nnz(P.mpc);
    P.nnz_h = amk_mpc_hess_nnz(P.mpc);
    auto dense = [](casadi_int n) { return std::vector<casadi_int>{n, 1, 1}; };  // dense column vector, shorthand
    P.sp_x = dense(P.nx); P.sp_p = dense(P.np); P.sp_g = dense(P.ng); P.sp_scalar = dense(1);
    auto ccs = [&](int nrow, int nnz, int (*fn)(const amk_mpc *, int *, int *)) {
        std::vector<int> colind(P.nx + 1), row(nnz);
        P.status = fn(P.mpc, colind.data(), row.data());
        std::vector<casadi_int> sp;
        sp.push_back(nrow); sp.push_back(P.nx);
        for (int v : colind) sp.push_back(v);
        for (int v : row) sp.push_back(v);
        return sp;
    };
    P.sp_jac = ccs(P.ng, P.nnz_j, amk_mpc_jac_sparsity);
    if (P.status == AMK_OK) P.sp_hess = ccs(P.nx, P.nnz_h, amk_mpc_hess_sparsity);
    P.f.resize(1); P.grad.resize(P.nx); P.g.resize(P.ng); P.jac.resize(P.nnz_j); P.hess.resize(P.nnz_h);
    P.gx.resize(P.nx); P.gp.resize(P.np);
    P.zx.assign(P.nx, 0.0); P.zp.assign(P.np, 0.0); P.zg.assign(P.ng, 0.0);
    return P.status;
}

Plugin &plugin() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_plugin.configured) {
        const char *eT = std::getenv("AMK_MPC_T"), *eD = std::getenv("AMK_MPC_DT"), *eK = std::getenv("AMK_MPC_K");
        configure_locked(g_plugin, eT ? std::atof(eT) : 1.0, eD ? std::atof(eD) : 0.033, eK ? std::atoi(eK) : 3);
    }
    return g_plugin;
}

// p = [P prefix | gain(4) | tau(4) | weights(25) | radius]: the tail goes through the handle's setters
// (HighLvlMpc.cpp:97-107 appends exactly these), re-uploaded only when it changed
int push_tail(Plugin &P, const double *p) {
    const double *t = p + P.nref;
    if (P.have_tail && std::memcmp(t, P.tail, sizeof P.tail) == 0) return AMK_OK;
    int st = amk_mpc_setup_gains(P.mpc, t);
    if (st == AMK_OK) st = amk_mpc_setup_tau(P.mpc, t + 4);
    if (st == AMK_OK) st = amk_mpc_setup_weights(P.mpc, t + 8);
    if (st == AMK_OK) st = amk_mpc_set_drone_radius(P.mpc, t[33]);
    if (st == AMK_OK) {
        std::memcpy(P.tail, t, sizeof P.tail);
        P.have_tail = true;
    }
    return st;
}

enum { WANT_F = 1, WANT_GRAD = 2, WANT_G = 4, WANT_JAC = 8, WANT_HESS = 16, WANT_GX = 32, WANT_GP = 64 };

// one evaluation; returns 0 on success (the generated code's convention), 1 otherwise.  NULL inputs are all-zero vectors
// (CasADi's convention for arg[i] == 0): x = 0, p = 0, lam_f = 0, lam_g = 0.
int run(const double *x, const double *p, const double *lam_f, const double *lam_g, int want) {
    Plugin &P = plugin();
    if (P.status != AMK_OK) return 1;
    if (!x) x = P.zx.data();
    if (!p) p = P.zp.data();
    if (push_tail(P, p) != AMK_OK) return 1;
    const double zero = 0.0;
    if (!lam_f) lam_f = &zero;
    if (want & (WANT_F | WANT_GRAD | WANT_G | WANT_JAC | WANT_HESS)) {
        const int st = amk_mpc_eval_host(P.mpc, x, p, lam_f, (want & WANT_F) ? P.f.data() : nullptr,
                                         (want & WANT_GRAD) ? P.grad.data() : nullptr, (want & WANT_G) ? P.g.data() : nullptr,
                                         (want & WANT_JAC) ? P.jac.data() : nullptr, (want & WANT_HESS) ? P.hess.data() : nullptr);
        if (st != AMK_OK) return 1;
    }
    if (want & (WANT_GX | WANT_GP)) {
        const int st = amk_mpc_eval_gamma_host(P.mpc, x, p, lam_f, lam_g ? lam_g : P.zg.data(),
                                               (want & WANT_GX) ? P.gx.data() : nullptr, (want & WANT_GP) ? P.gp.data() : nullptr);
        if (st != AMK_OK) return 1;
    }
    return 0;
}

void copy_out(double *dst, const std::vector<double> &src) {
    if (dst) std::memcpy(dst, src.data(), sizeof(double) * src.size());
}

const char *pick(casadi_int i, const char *const *names, int n) { return (i >= 0 && i < n) ? names[i] : nullptr; }

// sparsity of a named input / output
const casadi_int *sp_of(const char *name) {
    Plugin &P = plugin();
    if (!name || P.status != AMK_OK) return nullptr;
    if (!std::strcmp(name, "x") || !std::strcmp(name, "grad_f_x") || !std::strcmp(name, "grad_gamma_x")) return P.sp_x.data();
    if (!std::strcmp(name, "p") || !std::strcmp(name, "grad_gamma_p")) return P.sp_p.data();
    if (!std::strcmp(name, "g") || !std::strcmp(name, "lam_g")) return P.sp_g.data();
    if (!std::strcmp(name, "f") || !std::strcmp(name, "lam_f")) return P.sp_scalar.data();
    if (!std::strcmp(name, "jac_g_x")) return P.sp_jac.data();
    if (!std::strcmp(name, "triu_hess_gamma_x_x")) return P.sp_hess.data();
    return nullptr;
}

}  // namespace

// ---- the helper entry points every generated function carries ------------------------------------------------------
#define AMK_PLUGIN_COMMON(NAME, N_IN, N_OUT, ...)                                                                        \
    static const char *const NAME##_names[] = {__VA_ARGS__};                                                             \
    int NAME##_alloc_mem(void) { return 0; }                                                                             \
    int NAME##_init_mem(int) { return 0; }                                                                               \
    void NAME##_free_mem(int) {}                                                                                         \
    int NAME##_checkout(void) { return 0; }                                                                              \
    void NAME##_release(int) {}                                                                                          \
    void NAME##_incref(void) {}                                                                                          \
    void NAME##_decref(void) {}                                                                                          \
    casadi_int NAME##_n_in(void) { return N_IN; }                                                                        \
    casadi_int NAME##_n_out(void) { return N_OUT; }                                                                      \
    double NAME##_default_in(casadi_int) { return 0.0; }                                                                 \
    const char *NAME##_name_in(casadi_int i) { return pick(i, NAME##_names, N_IN); }                                     \
    const char *NAME##_name_out(casadi_int i) { return i < 0 ? nullptr : pick(i + N_IN, NAME##_names, N_IN + N_OUT); }   \
    const casadi_int *NAME##_sparsity_in(casadi_int i) { return sp_of(NAME##_name_in(i)); }                              \
    const casadi_int *NAME##_sparsity_out(casadi_int i) { return sp_of(NAME##_name_out(i)); }                            \
    int NAME##_work(casadi_int *sz_arg, casadi_int *sz_res, casadi_int *sz_iw, casadi_int *sz_w) {                       \
        if (sz_arg) *sz_arg = N_IN;                                                                                      \
        if (sz_res) *sz_res = N_OUT;                                                                                     \
        if (sz_iw) *sz_iw = 0;                                                                                           \
        if (sz_w) *sz_w = 0;                                                                                             \
        return 0;                                                                                                        \
    }

extern "C" {

// ---- nlp(x, p) -> (f, g): the oracle nlpsol(name, "ipopt", "<file>.so", opts) loads first (nx, np, ng and the dense
// sparsities come from it)
AMK_PLUGIN_COMMON(nlp, 2, 2, "x", "p", "f", "g")
int nlp(const double **arg, double **res, casadi_int *, double *, int) {
    const int want = (res[0] ? WANT_F : 0) | (res[1] ? WANT_G : 0);
    if (want && run(arg[0], arg[1], nullptr, nullptr, want)) return 1;
    copy_out(res[0], plugin().f);
    copy_out(res[1], plugin().g);
    return 0;
}

// ---- nlp_f(x, p) -> f --------------------------------------------------------------------------------------------
AMK_PLUGIN_COMMON(nlp_f, 2, 1, "x", "p", "f")
int nlp_f(const double **arg, double **res, casadi_int *, double *, int) {
    if (!res[0]) return 0;
    if (run(arg[0], arg[1], nullptr, nullptr, WANT_F)) return 1;
    copy_out(res[0], plugin().f);
    return 0;
}

// ---- nlp_g(x, p) -> g --------------------------------------------------------------------------------------------
AMK_PLUGIN_COMMON(nlp_g, 2, 1, "x", "p", "g")
int nlp_g(const double **arg, double **res, casadi_int *, double *, int) {
    if (!res[0]) return 0;
    if (run(arg[0], arg[1], nullptr, nullptr, WANT_G)) return 1;
    copy_out(res[0], plugin().g);
    return 0;
}

// ---- nlp_grad_f(x, p) -> (f, grad_f_x) ------------------------------------------------------------------------------
AMK_PLUGIN_COMMON(nlp_grad_f, 2, 2, "x", "p", "f", "grad_f_x")
int nlp_grad_f(const double **arg, double **res, casadi_int *, double *, int) {
    const int want = (res[0] ? WANT_F : 0) | (res[1] ? WANT_GRAD : 0);
    if (want && run(arg[0], arg[1], nullptr, nullptr, want)) return 1;
    copy_out(res[0], plugin().f);
    copy_out(res[1], plugin().grad);
    return 0;
}

// ---- nlp_jac_g(x, p) -> (g, jac_g_x) ----------------------------------------------------------------------------------
AMK_PLUGIN_COMMON(nlp_jac_g, 2, 2, "x", "p", "g", "jac_g_x")
int nlp_jac_g(const double **arg, double **res, casadi_int *, double *, int) {
    const int want = (res[0] ? WANT_G : 0) | (res[1] ? WANT_JAC : 0);
    if (want && run(arg[0], arg[1], nullptr, nullptr, want)) return 1;
    copy_out(res[0], plugin().g);
    copy_out(res[1], plugin().jac);
    return 0;
}

// ---- nlp_hess_l(x, p, lam_f, lam_g) -> triu(hess_gamma_x_x) ---------------------------------------------------------
// gamma = lam_f f + lam_g' g; g is linear in x, so lam_g drops out
AMK_PLUGIN_COMMON(nlp_hess_l, 4, 1, "x", "p", "lam_f", "lam_g", "triu_hess_gamma_x_x")
int nlp_hess_l(const double **arg, double **res, casadi_int *, double *, int) {
    if (!res[0]) return 0;
    if (run(arg[0], arg[1], arg[2], nullptr, WANT_HESS)) return 1;
    copy_out(res[0], plugin().hess);
    return 0;
}

// ---- nlp_grad(x, p, lam_f, lam_g) -> (f, g, grad_gamma_x, grad_gamma_p): what nlpsol evaluates after the solve for lam_p
AMK_PLUGIN_COMMON(nlp_grad, 4, 4, "x", "p", "lam_f", "lam_g", "f", "g", "grad_gamma_x", "grad_gamma_p")
int nlp_grad(const double **arg, double **res, casadi_int *, double *, int) {
    const int want = (res[0] ? WANT_F : 0) | (res[1] ? WANT_G : 0) | (res[2] ? WANT_GX : 0) | (res[3] ? WANT_GP : 0);
    if (want && run(arg[0], arg[1], arg[2], arg[3], want)) return 1;
    copy_out(res[0], plugin().f);
    copy_out(res[1], plugin().g);
    copy_out(res[2], plugin().gx);
    copy_out(res[3], plugin().gp);
    return 0;
}

// ---- not part of CasADi's API -----------------------------------------------------------------------------------
int amk_plugin_configure(double T, double dt, int nearest_point_num) {
    std::lock_guard<std::mutex> lk(g_mu);
    return configure_locked(g_plugin, T, dt, nearest_point_num);
}

int amk_plugin_dims(int *N, int *K, int *nx, int *np, int *ng) {
    Plugin &P = plugin();
    if (P.status != AMK_OK) return P.status;
    if (N) *N = P.N;
    if (K) *K = P.K;
    if (nx) *nx = P.nx;
    if (np) *np = P.np;
    if (ng) *ng = P.ng;
    return AMK_OK;
}

}  // extern "C"
