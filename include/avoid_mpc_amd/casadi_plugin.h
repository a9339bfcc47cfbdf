/* casadi_plugin.h -- the symbols libavoid_mpc_amd.so exports so that it can stand in for the reference's generated solver
 * plugin.
 *
 * The reference produces `so/mpc_obstacle_v2.so` with CasADi's `solver.generate_dependencies("nmpc_v0.c")`
 * (AM/tools/mpc_obstacle_casadi.py:294-303; AM = roswrapper/ros/src/avoid_mpc) and loads it with
 * `casadi::nlpsol("solve", "ipopt", soPath, opts)` (AM/src/HighLvlMpc.cpp:50,52; ROS parameter `mpc_so`,
 * AM/src/ParameterManager.cpp:90).  nlpsol resolves, by name, the NLP oracle `nlp` and the functions IPOPT's interface
 * asks for; every one of them carries CasADi's code-generation C API:
 *
 *     int  NAME(const double** arg, double** res, casadi_int* iw, double* w, int mem);     0 = success
 *     int  NAME_alloc_mem(void);  int NAME_init_mem(int mem);  void NAME_free_mem(int mem);
 *     int  NAME_checkout(void);   void NAME_release(int mem);
 *     void NAME_incref(void);     void NAME_decref(void);
 *     casadi_int NAME_n_in(void); casadi_int NAME_n_out(void);
 *     casadi_real NAME_default_in(casadi_int i);
 *     const char* NAME_name_in(casadi_int i);  const char* NAME_name_out(casadi_int i);
 *     const casadi_int* NAME_sparsity_in(casadi_int i);  const casadi_int* NAME_sparsity_out(casadi_int i);
 *     int  NAME_work(casadi_int* sz_arg, casadi_int* sz_res, casadi_int* sz_iw, casadi_int* sz_w);
 *
 * (casadi_int = long long, casadi_real = double; a sparsity is [nrow, ncol, colind[ncol+1], row[nnz]] in compressed-column
 * form, or [nrow, ncol, 1] for a dense block; arg[i] == NULL means an all-zero input, res[i] == NULL an output that is not
 * requested.)  THIS CONVENTION IS RECALLED FROM CasADi 3.6, WHICH IS NEITHER UNDER /root/reference NOR IN THE BUILD IMAGE:
 * the plugin has never been loaded by a real libcasadi -- treat the B4 route as experimental and the
 * class-level route (include/avoid_mpc_amd/high_lvl_mpc.hpp) as the supported one.
 *
 *   NAME         inputs                         outputs
 *   nlp          x, p                           f, g                                   the oracle nlpsol loads first
 *   nlp_f        x, p                           f
 *   nlp_g        x, p                           g
 *   nlp_grad_f   x, p                           f, grad_f_x
 *   nlp_jac_g    x, p                           g, jac_g_x           (CCS, 10 + 39 N non-zeros)
 *   nlp_hess_l   x, p, lam_f, lam_g             triu_hess_gamma_x_x  (CCS, 25 (N - 1) + 10 + 4 N non-zeros)
 *   nlp_grad     x, p, lam_f, lam_g             f, g, grad_gamma_x, grad_gamma_p
 *
 * x = [X_0, U_0, ..., U_{N-1}, X_N] (10 + 14 N), p = the full parameter vector (54 + 10 N + 3 K N), g = 10 + 10 N rows.
 * N = int(T / dt) and K are baked into the generated file (mpc_obstacle_casadi.py:36-37,76-85); here they are set by
 * amk_plugin_configure() before the first call, or read once from the environment (AMK_MPC_T, AMK_MPC_DT, AMK_MPC_K;
 * defaults = AM/config/mpc_parameters.yaml:1-2,5).  Every evaluation runs on the GPU (amk_mpc_eval / amk_mpc_eval_gamma with
 * one scene); a host that needs two horizons at once loads two copies of the library file.
 */
#ifndef AVOID_MPC_AMD_CASADI_PLUGIN_H
#define AVOID_MPC_AMD_CASADI_PLUGIN_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long amk_casadi_int;

#define AMK_CASADI_DECLARE(NAME)                                                                      \
    int NAME(const double **arg, double **res, amk_casadi_int *iw, double *w, int mem);              \
    int NAME##_alloc_mem(void);                                                                       \
    int NAME##_init_mem(int mem);                                                                     \
    void NAME##_free_mem(int mem);                                                                    \
    int NAME##_checkout(void);                                                                        \
    void NAME##_release(int mem);                                                                     \
    void NAME##_incref(void);                                                                         \
    void NAME##_decref(void);                                                                         \
    amk_casadi_int NAME##_n_in(void);                                                                 \
    amk_casadi_int NAME##_n_out(void);                                                                \
    double NAME##_default_in(amk_casadi_int i);                                                       \
    const char *NAME##_name_in(amk_casadi_int i);                                                     \
    const char *NAME##_name_out(amk_casadi_int i);                                                    \
    const amk_casadi_int *NAME##_sparsity_in(amk_casadi_int i);                                       \
    const amk_casadi_int *NAME##_sparsity_out(amk_casadi_int i);                                      \
    int NAME##_work(amk_casadi_int *sz_arg, amk_casadi_int *sz_res, amk_casadi_int *sz_iw, amk_casadi_int *sz_w);

AMK_CASADI_DECLARE(nlp)
AMK_CASADI_DECLARE(nlp_f)
AMK_CASADI_DECLARE(nlp_g)
AMK_CASADI_DECLARE(nlp_grad_f)
AMK_CASADI_DECLARE(nlp_jac_g)
AMK_CASADI_DECLARE(nlp_hess_l)
AMK_CASADI_DECLARE(nlp_grad)

/* Not part of CasADi's API.  amk_plugin_configure: (re)creates the plugin's one-scene MPC handle for horizon int(T / dt) and
 * K neighbours; returns an AMK_* status.  Without it the first call of any symbol above configures from the environment.
 * amk_plugin_dims: the sizes in force.                                                                               */
int amk_plugin_configure(double T, double dt, int nearest_point_num);
int amk_plugin_dims(int *N, int *K, int *nx, int *np, int *ng);

#ifdef __cplusplus
}
#endif
#endif
