// Type-generic body of the MPC device algorithm; included twice by mpc_device.h (AMK_REAL = double in namespace
// amk, float in namespace amk32).  `real` is the arithmetic type of the NLP evaluation and the interior-point
// method and the element type of the LDS scratchpad; everything that crosses global memory (parameters, P,
// warm start, outputs, plan coefficients) stays double.  RL(x) is a literal of type real.
namespace AMK_RNS {
using namespace amk;
typedef AMK_REAL real;
#define RL(x) ((real)(x))
constexpr real kGz = RL(9.81);   // mpc_obstacle_casadi.py:39
constexpr int YB = amk::kTermRecord;  // doubles per collision term in the global scratch: y1, y2, 1/rho, g'/g

// ---- cross-lane reductions on DPP (no LDS crossbar): quad_perm for lane^1 / lane^2, row_half_mirror and
// row_mirror for the 8- and 16-lane levels (valid because every lane of the lower level already holds
// that level's result), v_readlane for the four rows.  __shfl_xor would be two ds_bpermute_b32 + a wait
// per double per level; the collision Hessian reduction alone is 28 values x 3 levels x 3 rounds.
template <int CTRL>
__device__ __forceinline__ real dpp_r(real v) {
#if AMK_REAL_F32
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
#else
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
#endif
}
__device__ __forceinline__ real readlane_r(real v, int l) {
#if AMK_REAL_F32
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
#else
    return amk::readlane_f64(v, l);
#endif
}

// 1/x to working precision: v_rcp + Newton steps (the IEEE division sequence is ~4x longer)
__device__ __forceinline__ real fast_rcp(real x) {
#if AMK_REAL_F32
    real r = __builtin_amdgcn_rcpf(x);
    r = fma(fma(-x, r, 1.0f), r, r);
    return r;
#else
    return amk::rcp_f64(x);  // one cubic correction: a dependent operation less than two Newton steps
#endif
}

// exp / log of the collision terms and barrier sums: the short-chain versions of fast_math.h in fp64, the library's
// (hardware-assisted) ones in fp32
__device__ __forceinline__ real real_exp(real x) {
#if AMK_REAL_F32
    return exp(x);
#else
    return amk::fast_exp(x);
#endif
}
__device__ __forceinline__ real real_log(real x) {
#if AMK_REAL_F32
    return log(x);
#else
    return amk::fast_log(x);
#endif
}

struct OpSum { __device__ __forceinline__ static real f(real a, real b) { return a + b; } };
struct OpMin { __device__ __forceinline__ static real f(real a, real b) { return fmin(a, b); } };
struct OpMax { __device__ __forceinline__ static real f(real a, real b) { return fmax(a, b); } };

template <class Op>
__device__ __forceinline__ real row_reduce(real v) {  // every lane of a 16-lane row gets the row result
    v = Op::f(v, dpp_r<DPP_XOR1>(v));
    v = Op::f(v, dpp_r<DPP_XOR2>(v));
    v = Op::f(v, dpp_r<DPP_HALF_MIRROR>(v));
    v = Op::f(v, dpp_r<DPP_MIRROR>(v));
    return v;
}
template <class Op>
__device__ __forceinline__ real wave_reduce(real v) {
    v = row_reduce<Op>(v);
    const real r0 = readlane_r(v, 0), r1 = readlane_r(v, 16), r2 = readlane_r(v, 32), r3 = readlane_r(v, 48);
    return Op::f(Op::f(r0, r1), Op::f(r2, r3));
}
__device__ __forceinline__ real wave_sum(real v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ real wave_min(real v) { return wave_reduce<OpMin>(v); }
__device__ __forceinline__ real wave_max(real v) { return wave_reduce<OpMax>(v); }

// sum over aligned segments of `seg` (power of two) consecutive lanes; every lane gets its segment's sum
__device__ __forceinline__ real seg_sum(real v, int seg) {
    if (seg >= 2) v += dpp_r<DPP_XOR1>(v);
    if (seg >= 4) v += dpp_r<DPP_XOR2>(v);
    if (seg >= 8) v += dpp_r<DPP_HALF_MIRROR>(v);
    if (seg >= 16) v += dpp_r<DPP_MIRROR>(v);
    if (seg >= 32) v += __shfl_xor(v, 16);
    if (seg >= 64) v += __shfl_xor(v, 32);
    return v;
}

// ---- collision terms.  c |s| is handled as an l1 term of an interior-point method:
// epigraph variable t >= |s| kept at the optimum of its own barrier, t = mu + sqrt(mu^2 + s^2), slacks w1 = t - s,
// w2 = t + s, multipliers yh1, yh2 (global memory, two doubles per term; yh1 < 0 = none yet), barrier form
//   F = c (t - mu log(w1 w2)).
// Mirrors oracle term_geo / slacks / term_mult / term_merit / term_derivs / term_step statement by statement.
__device__ __forceinline__ void slacks(real s, real mu, real &w1, real &w2) {
    const real as = fabs(s), r = sqrt(mu * mu + s * s);
    const real ws = mu + mu * mu * fast_rcp(r + as), wb = mu + r + as;  // no cancellation in the small slack
    w1 = s > RL(0.0) ? ws : wb;
    w2 = s > RL(0.0) ? wb : ws;
}
__device__ __forceinline__ void term_mult(real y1, real y2, real mu, real kappa_sigma, real iw1, real iw2, real &a1,
                                          real &a2) {
    if (y1 < RL(0.0)) {  // dormant so far: central path
        a1 = mu * iw1;
        a2 = mu * iw2;
    } else {             // within a factor kappa_sigma of the central path
        const real ik = fast_rcp(kappa_sigma);
        a1 = fmax(fmin(y1, kappa_sigma * mu * iw1), mu * iw1 * ik);
        a2 = fmax(fmin(y2, kappa_sigma * mu * iw2), mu * iw2 * ik);
    }
}

// MODE 0: merit value only.  MODE 1: also the condensed gradient (6) and Hessian (21 unique, row-major lower triangle
// of the (p,v) 6x6 block) ADDED to the stage's LDS cells with ds_add_f64 (no per-lane accumulators, no cross-lane
// reduction; lanes of one wave-instruction that hit the same cell are applied in lane order, so the sums are
// reproducible), and the complementarity maxima cmax = max yh_i w_i, cdev = max |yh_i w_i - mu|.
// MODE 2: Newton step of the term's multipliers for the state step (dp, dv), own fraction-to-the-boundary length;
// y1/y2 are updated in place (a term that is dormant here forgets its multipliers).
// c_ir, c_gpg: 1 / |o - p| and g'/g of the term at the iterate, written by MODE 1 (c_ir = 0 for a dormant term) and read
// back by MODE 2, which runs at the same point: the multiplier pass skips the sqrt -> exp -> log chain (~100 fp64
// instructions per term) and finds the same bits.
template <int MODE>
__device__ __forceinline__ real collide_term(const real p[3], const real v[3], const real o[3], real lam, real radius,
                                               real mu, real kappa_sigma, real maj, real tau, real &y1, real &y2,
                                               const real *dp, const real *dv, real *gq, real *h21, real &cmax,
                                               real &cdev, real &c_ir, real &c_gpg) {
    const real d0 = o[0] - p[0], d1 = o[1] - p[1], d2 = o[2] - p[2];
    real ir, g = RL(0.0), c = RL(0.0), ex = RL(0.0);
    if (MODE == 2) {
        ir = c_ir;
        if (!(ir > RL(0.0))) { y1 = -RL(1.0); y2 = -RL(1.0); return RL(0.0); }  // dormant at the iterate
    } else {
        // Exact-zero pre-test (round 6): the reference's naive softplus log(1 + exp(x)) is exactly 0 once exp(x) < 2^-53, i.e. x < -36.74, i.e.
        // beyond 1.148 m of clearance.  A squared-distance test with a margin (x < -37: clearance > 1.15625 m) decides that without the
        // sqrt -> exp -> log chain; a term inside the margin takes the chain and its own `c > 0` test as before, so the result is the same
        // bits either way.  A round whose lanes are ALL beyond it (sparse worlds, the 1e4 padding of short neighbour lists) skips the
        // chain altogether; on the bench's dense scenes 150 of 152 terms are live and this is two instructions per term.
        const real rr2 = d0 * d0 + d1 * d1 + d2 * d2;
        const real far = radius + (AMK_REAL_F32 ? RL(0.6) : RL(1.15625));   // (fp32: exp(x) < 2^-24 at x < -16.7: clearance 0.52 m)
        if (rr2 > far * far * (RL(1.0) + RL(1e-6))) {
            if (MODE == 1) c_ir = RL(0.0);
            return RL(0.0);
        }
        const real rho = sqrt(rr2);
        const real x = -RL(32.0) * (rho - radius);
        ex = real_exp(x);
        g = real_log(RL(1.0) + ex);  // naive softplus, mpc_obstacle_casadi.py:250-251
        c = lam * g;
        if (!(c > RL(0.0))) {          // dormant: c == 0 exactly, the term and all its derivatives vanish
            if (MODE == 1) c_ir = RL(0.0);
            return RL(0.0);
        }
        ir = fast_rcp(rho);
    }
    const real n[3] = {d0 * ir, d1 * ir, d2 * ir};
    const real s = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
    real w1, w2;
    slacks(s, mu, w1, w2);
    const real t = RL(0.5) * (w1 + w2);
    if (MODE == 0) return c * (t - mu * real_log(w1 * w2));
    real sg = RL(0.0), gp = RL(0.0);
    if (MODE == 1) {
        sg = ex * fast_rcp(RL(1.0) + ex);  // = 1/(1+exp(-x)); x <= 32 r, no overflow
        gp = -RL(32.0) * sg;
        c_ir = ir;
        c_gpg = gp * fast_rcp(g);
    }
    const real tv[3] = {v[0] - s * n[0], v[1] - s * n[1], v[2] - s * n[2]};
    const real iw1 = fast_rcp(w1), iw2 = fast_rcp(w2);
    real a1, a2;
    term_mult(y1, y2, mu, kappa_sigma, iw1, iw2, a1, a2);
    const real D1 = a1 * iw1, D2 = a2 * iw2, Dh = D1 + D2, dD = D2 - D1, iDh = fast_rcp(Dh);
    const real e = RL(1.0) - a1 - a2;
    if (MODE == 2) {
        const real ndp = n[0] * dp[0] + n[1] * dp[1] + n[2] * dp[2];
        const real ds = -(tv[0] * dp[0] + tv[1] * dp[1] + tv[2] * dp[2]) * ir + (n[0] * dv[0] + n[1] * dv[1] + n[2] * dv[2]);
        const real dlc = -c_gpg * ndp;                   // (grad c / c)' dz
        const real dt = (-e * dlc - dD * ds) * iDh;      // the t row of the Newton system (its residual is 0)
        const real dy1 = mu * iw1 - a1 - D1 * (dt - ds), dy2 = mu * iw2 - a2 - D2 * (dt + ds);
        real al = RL(1.0);
        if (dy1 < RL(0.0)) al = fmin(al, -tau * a1 * fast_rcp(dy1));
        if (dy2 < RL(0.0)) al = fmin(al, -tau * a2 * fast_rcp(dy2));
        y1 = a1 + al * dy1;
        y2 = a2 + al * dy2;
        return RL(0.0);
    }
    const real gpp = RL(1024.0) * sg * (RL(1.0) - sg);
    const real sig = a1 - a2;
    const real bs = mu * (iw1 - iw2);  // d/ds of the barrier form; its d/dt vanishes at the optimal t
    const real Psi = t - mu * real_log(w1 * w2);
    const real lgp = lam * gp;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        unsafeAtomicAdd(gq + i, -lgp * Psi * n[i] - c * bs * tv[i] * ir);  // state slots 0,1,2 = p
        unsafeAtomicAdd(gq + 4 + i, c * bs * n[i]);                        // state slots 4,5,6 = v
    }
    // Hessian with t eliminated, in the three rank-structured blocks (n = unit vector to the obstacle, tv = v - s n):
    //   H_pp = al tv tv' + A n n' + C (n tv' + tv n') + B I     H_vp = -be n tv' + D n n' + E I     H_vv = k2 n n'
    const real k1 = sig - e * dD * iDh, k2 = RL(4.0) * c * D1 * D2 * iDh + maj * c * fast_rcp(t), k3 = e * e * iDh * fast_rcp(c);
    const real ir2 = ir * ir, cs = c * sig;
    const real al = k2 * ir2, be = k2 * ir;
    const real B = Psi * lgp * ir - cs * s * ir2;
    const real A = Psi * lam * gpp - B - k3 * lgp * lgp;
    const real Cc = k1 * lgp * ir - cs * ir2;
    const real D = -k1 * lgp + cs * ir, E = -cs * ir;
    // lower triangle, row-major: (i,j), j <= i, index i(i+1)/2 + j
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            real h = al * tv[i] * tv[j] + A * n[i] * n[j] + Cc * (n[i] * tv[j] + tv[i] * n[j]);
            if (i == j) h += B;
            unsafeAtomicAdd(h21 + i * (i + 1) / 2 + j, h);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int i = 3 + a;
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // H[v_a][p_j]
            real h = n[a] * (D * n[j] - be * tv[j]);
            if (a == j) h += E;
            unsafeAtomicAdd(h21 + i * (i + 1) / 2 + j, h);
        }
#pragma unroll
        for (int bb = 0; bb <= a; ++bb) unsafeAtomicAdd(h21 + i * (i + 1) / 2 + 3 + bb, k2 * n[a] * n[bb]);
    }
    cmax = fmax(cmax, fmax(a1 * w1, a2 * w2));
    cdev = fmax(cdev, fmax(fabs(a1 * w1 - mu), fabs(a2 * w2 - mu)));
    return c * Psi;
}

// ---- the plugin's own form of one obstacle term (amk_mpc_eval: nlp_f / nlp_grad_f / nlp_hess_l): value lam g |s|,
// gradient and Hessian as CasADi differentiates it -- d|s|/ds = sign(s) (0 at s = 0), no curvature from the abs()
// itself.  Same LDS accumulation as collide_term.  Mirrors oracle collide_point (majorise = 0).
template <bool DERIV>
__device__ __forceinline__ real collide_exact(const real p[3], const real v[3], const real o[3], real lam,
                                                real radius, real *gq, real *h21) {
    const real d0 = o[0] - p[0], d1 = o[1] - p[1], d2 = o[2] - p[2];
    const real rho = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    const real x = -RL(32.0) * (rho - radius);
    const real ex = exp(x);
    const real g = log(RL(1.0) + ex);  // naive softplus, mpc_obstacle_casadi.py:250-251
    const real ir = fast_rcp(rho);
    const real n[3] = {d0 * ir, d1 * ir, d2 * ir};
    const real s = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
    const real as = fabs(s);
    const real cost = lam * g * as;
    if (!DERIV) return cost;
    const real sg = ex * fast_rcp(RL(1.0) + ex);  // = 1/(1+exp(-x)); x <= 32 r, no overflow
    const real gp = -RL(32.0) * sg;
    const real gpp = RL(1024.0) * sg * (RL(1.0) - sg);
    const real sgn = (s > RL(0.0)) ? RL(1.0) : ((s < RL(0.0)) ? -RL(1.0) : RL(0.0));
    const real t[3] = {v[0] - s * n[0], v[1] - s * n[1], v[2] - s * n[2]};
    const real ls = lam * sgn;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        unsafeAtomicAdd(gq + i, ls * (gp * (-n[i]) * s + g * (-t[i] * ir)));       // state slots 0,1,2 = p
        unsafeAtomicAdd(gq + 4 + i, ls * g * n[i]);                                // state slots 4,5,6 = v
    }
    // Model Hessian of the term w.r.t. (p, v), written out in the three rank-structured blocks (n = unit vector to the
    // obstacle, t = v - s n, s = v.n, g/g'/g'' the softplus and its derivatives in rho):
    //   H_pp = al t t' + A n n' + C (n t' + t n') + B I     H_vp = -be n t' + D n n' + E I     H_vv = wk n n'
    // (the same sums as oracle collide_point's entry-by-entry expressions, common factors pulled out)
    const real wk = RL(0.0);
    const real ir2 = ir * ir, gir = g * ir, gpir = gp * ir;
    const real al = wk * ir2, be = wk * ir;
    const real A = ls * s * (gpp - gpir + g * ir2), B = ls * s * (gpir - g * ir2), Cc = ls * (gpir - g * ir2);
    const real D = ls * (gir - gp), E = -ls * gir;
    // lower triangle, row-major: (i,j), j <= i, index i(i+1)/2 + j
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            real h = al * t[i] * t[j] + A * n[i] * n[j] + Cc * (n[i] * t[j] + t[i] * n[j]);
            if (i == j) h += B;
            unsafeAtomicAdd(h21 + i * (i + 1) / 2 + j, h);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int i = 3 + a;
#pragma unroll
        for (int j = 0; j < 3; ++j) {  // H[v_a][p_j]
            real h = n[a] * (D * n[j] - be * t[j]);
            if (a == j) h += E;
            unsafeAtomicAdd(h21 + i * (i + 1) / 2 + j, h);
        }
#pragma unroll
        for (int bb = 0; bb <= a; ++bb) unsafeAtomicAdd(h21 + i * (i + 1) / 2 + 3 + bb, wk * n[a] * n[bb]);
    }
    return cost;
}


// Objective on (Xs, Us) held in LDS, in the solver's barrier form (EXACT = false) or as the plugin defines it (EXACT).  DERIV: also q, r, H6 (21 per stage), and acc[0] = max yh_i w_i,
// acc[1] = max |yh_i w_i - mu| over the collision terms (wave-uniform).  Returns the value (wave-uniform).  Reads the
// multipliers (ybuf, [N-1][K][YB] doubles in global memory: y1, y2, then the cache of collide_term) and, with DERIV, writes the cache.  One wave; caller syncs before/after.
template <bool DERIV, bool EXACT = false>
__device__ __forceinline__ real evaluate(real *sm, const LdsMap &L, const SceneIO &io, int N, int K, const real *Xs,
                           const real *Us, real mu, real kappa_sigma, real maj, double *ybuf, real *acc,
                           long long *tclk = nullptr) {
    int lane = threadIdx.x; asm volatile("" : "+v"(lane));   // (opaque per phase: nothing derived from the lane is carried between the phases)
    const real *prm = sm + L.prm;
    const real lamw = prm[PRM_W + 24], radius = prm[PRM_RADIUS];
    real Jloc = RL(0.0), cmax = RL(0.0), cdev = RL(0.0);
    // ---- collision terms: lane = (stage, obstacle point), 64 terms per round
    if (DERIV) {  // the cells the terms add into
        for (int e = lane; e < (N - 1) * 21; e += 64) sm[L.H6 + e] = RL(0.0);
        for (int e = lane; e < (N - 1) * 6; e += 64) sm[L.q + (e / 6 + 1) * SD + pv_of(e % 6)] = RL(0.0);
        __syncthreads();
    }
    const long long tc0 = AMK_CLK();
    // lane = (obstacle slot, stage), stage fastest: a round covers `per` obstacle points of EVERY stage, so a stage's
    // LDS cells see `per` (3 at N = 20) colliding ds_add_f64 per instruction instead of K (8) with the obstacle
    // index fastest -- the atomics are a fifth of the kernel's LDS time
    const int ns = N - 1, per = 64 / ns;
    const int jl = lane / ns, k = lane - jl * ns;
    for (int j0 = 0; j0 < K; j0 += per) {
        const int j = j0 + jl;
        if (jl < per && j < K) {
            const real *xk = Xs + (k + 1) * SD;
            const real p[3] = {xk[0], xk[1], xk[2]}, v[3] = {xk[4], xk[5], xk[6]};
            const double *op = io.obs + ((size_t)k * K + j) * 3;  // [k][j][3]
            const real o[3] = {(real)op[0], (real)op[1], (real)op[2]};
            if (EXACT) {  // the plugin's form c |s| (amk_mpc_eval)
                Jloc += collide_exact<DERIV>(p, v, o, lamw, radius, sm + L.q + (k + 1) * SD, sm + L.H6 + k * 21);
            } else {
                real y1 = RL(0.0), y2 = RL(0.0), c_ir = RL(0.0), c_gpg = RL(0.0);
                double *yrec = ybuf + ((size_t)k * K + j) * YB;
                if (DERIV) {
                    const double2 yy = *reinterpret_cast<const double2 *>(yrec);
                    y1 = (real)yy.x; y2 = (real)yy.y;
                }
                Jloc += collide_term<DERIV ? 1 : 0>(p, v, o, lamw, radius, mu, kappa_sigma, maj, RL(0.0), y1, y2, nullptr,
                                                    nullptr, sm + L.q + (k + 1) * SD, sm + L.H6 + k * 21, cmax, cdev, c_ir,
                                                    c_gpg);
                if (DERIV) *reinterpret_cast<double2 *>(yrec + 2) = make_double2((double)c_ir, (double)c_gpg);
            }
        }
    }
    if (kTrace && tclk) tclk[0] += AMK_CLK() - tc0;
    if (DERIV) __syncthreads();
    const long long tc3 = AMK_CLK();
    // ---- per-stage quadratic terms: lane = stage
    if (lane < N) {
        const int k = lane;
        const real *uk = Us + k * UD;
        const real *xk = Xs + (k + 1) * SD;
        const real uref[4] = {RL(0.0), RL(0.0), kGz, RL(0.0)};
#pragma unroll
        for (int i = 0; i < UD; ++i) {
            const real du = uk[i] - uref[i];
            const real w = prm[PRM_W + 20 + i];
            Jloc += du * w * du;  // :209-210
            if (DERIV) sm[L.r + k * UD + i] = RL(2.0) * w * du;
        }
        if (k >= N - 1) {  // goal stage :168-170
            const real *tg = sm + L.target;
#pragma unroll
            for (int i = 0; i < SD; ++i) {
                const real d = xk[i] - tg[i];
                const real w = prm[PRM_W + i];
                Jloc += d * w * d;
                if (DERIV) sm[L.q + (k + 1) * SD + i] = RL(2.0) * w * d;
            }
        } else {  // path stage :171-208
            const double *rf = io.ref + k * SD;
            const real cy = sm[L.cy + k], sy = sm[L.sy + k];
            real d[SD], y[SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) { d[i] = xk[i] - (real)rf[i]; y[i] = d[i]; }
            y[0] = cy * d[0] - sy * d[1];
            y[1] = sy * d[0] + cy * d[1];
            y[4] = cy * d[4] - sy * d[5];
            y[5] = sy * d[4] + cy * d[5];
            real wy[SD];
#pragma unroll
            for (int i = 0; i < SD; ++i) {
                const real w = prm[PRM_W + 10 + i];
                Jloc += y[i] * w * y[i];
                wy[i] = RL(2.0) * w * y[i];
            }
            if (DERIV) {
                real qq[SD];
#pragma unroll
                for (int i = 0; i < SD; ++i) qq[i] = wy[i];
                qq[0] = cy * wy[0] + sy * wy[1];
                qq[1] = -sy * wy[0] + cy * wy[1];
                qq[4] = cy * wy[4] + sy * wy[5];
                qq[5] = -sy * wy[4] + cy * wy[5];
                real *qk = sm + L.q + (k + 1) * SD;
#pragma unroll
                for (int i = 0; i < SD; ++i) {
                    const bool inpv = (i < 3) || (i >= 4 && i <= 6);
                    qk[i] = qq[i] + (inpv ? qk[i] : RL(0.0));
                }
            }
        }
    }
    if (kTrace && tclk) tclk[2] += AMK_CLK() - tc3;
    if (DERIV && !EXACT) {
        acc[0] = wave_max(cmax);
        acc[1] = wave_max(cdev);
    }
    return wave_sum(Jloc);
}

// Newton step of the collision terms' multipliers for the state step dX (oracle term_step), same lane = (slot, stage) map.
__device__ __forceinline__ void update_term_multipliers(real *sm, const LdsMap &L, const SceneIO &io, int N, int K, real mu,
                                                        real tau, real kappa_sigma, double *ybuf) {
    int lane = threadIdx.x; asm volatile("" : "+v"(lane));   // (opaque per phase: nothing derived from the lane is carried between the phases)
    const real *prm = sm + L.prm;
    const real lamw = prm[PRM_W + 24], radius = prm[PRM_RADIUS];
    const int ns = N - 1, per = 64 / ns;
    const int jl = lane / ns, k = lane - jl * ns;
    real dummy0 = RL(0.0), dummy1 = RL(0.0);
    for (int j0 = 0; j0 < K; j0 += per) {
        const int j = j0 + jl;
        if (jl < per && j < K) {
            const real *xk = sm + L.X + (k + 1) * SD, *dk = sm + L.dX + (k + 1) * SD;
            const real p[3] = {xk[0], xk[1], xk[2]}, v[3] = {xk[4], xk[5], xk[6]};
            const real dp[3] = {dk[0], dk[1], dk[2]}, dv[3] = {dk[4], dk[5], dk[6]};
            const double *op = io.obs + ((size_t)k * K + j) * 3;
            const real o[3] = {(real)op[0], (real)op[1], (real)op[2]};
            double2 *yp = reinterpret_cast<double2 *>(ybuf + ((size_t)k * K + j) * YB);
            const double2 yy = yp[0], cc = yp[1];
            real y1 = (real)yy.x, y2 = (real)yy.y, c_ir = (real)cc.x, c_gpg = (real)cc.y;
            collide_term<2>(p, v, o, lamw, radius, mu, kappa_sigma, RL(0.0), tau, y1, y2, dp, dv, nullptr, nullptr, dummy0,
                            dummy1, c_ir, c_gpg);
            yp[0] = make_double2((double)y1, (double)y2);
        }
    }
}

#ifdef AMK_ADJOINT_SEQ  // the sequential sweep the scans below replaced (kept for A/B: tools/experiments/ab_solve.sh)
// Reduced gradient gU_k = r_k + B' lam_{k+1} by the adjoint sweep lam_k = q_k + A' lam_{k+1}, lam_N = q_N (oracle
// eval_iterate).  Lane i < 10 owns lam[i], lanes 10..13 own gU[a]; column i of A / a of B has <= 3 non-zero rows
// (rows_of_A / rows_of_B), read from a double-buffered copy of lam_{k+1} in LDS: one barrier per stage.
__device__ __forceinline__ void adjoint_sweep_seq(real *sm, const LdsMap &L, int N, const double *prm_g) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));  // the lane's coefficients are re-derived per sweep: hoisted out of the iteration loop
                                    // they would be live across the objective evaluation, which sets the register peak
    const double *A = prm_g + PRM_A, *B = prm_g + PRM_B;  // three coefficients per lane, L2-resident
    int rows[3] = {0, 0, 0};
    real cf[3] = {RL(0.0), RL(0.0), RL(0.0)};
    if (lane < SD) {
        const int nr = rows_of_A(lane, rows);
        for (int t = 0; t < 3; ++t) cf[t] = t < nr ? (real)A[rows[t] * SD + lane] : RL(0.0);
    } else if (lane < SD + UD) {
        const int nr = rows_of_B(lane - SD, rows);
        for (int t = 0; t < 3; ++t) cf[t] = t < nr ? (real)B[rows[t] * UD + (lane - SD)] : RL(0.0);
    }
    real *buf0 = sm + L.lam, *buf1 = sm + L.Atl;
    if (lane < SD) buf0[lane] = sm[L.q + N * SD + lane];
    __syncthreads();
#pragma unroll 1
    for (int k = N - 1; k >= 0; --k) {
        const real *src = ((N - 1 - k) & 1) ? buf1 : buf0;
        real *dst = ((N - 1 - k) & 1) ? buf0 : buf1;
        const real l0 = src[rows[0]], l1 = src[rows[1]], l2 = src[rows[2]];
        const real acc3 = (cf[0] * l0 + cf[1] * l1) + cf[2] * l2;
        if (lane < SD) {
            if (k > 0) dst[lane] = sm[L.q + k * SD + lane] + acc3;
        } else if (lane < SD + UD) {
            sm[L.gU + k * UD + (lane - SD)] = sm[L.r + k * UD + (lane - SD)] + acc3;
        }
        __syncthreads();
    }
}
#endif

// value of lane k + d of the same 32-lane half (0 beyond it): two ds_bpermute per double, no LDS memory
__device__ __forceinline__ real half_shift_down(real v, int d, int k) {
    const real y = __shfl_down(v, d, 32);
    return (k + d < 32) ? y : RL(0.0);
}
// x_e <- sum_{j >= 0} alpha^j x_{e+j} over the 32 lanes of a half (suffix scan of the recurrence x_e = b_e + alpha x_{e+1})
__device__ __forceinline__ void suffix_scan2(real &x0, real &x1, real al0, real al1, int k) {
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        const real y0 = half_shift_down(x0, d, k), y1 = half_shift_down(x1, d, k);
        x0 = fma(al0, y0, x0);
        x1 = fma(al1, y1, x1);
        al0 *= al0;
        al1 *= al1;
    }
}

// The same reduced gradient without the N sequential stages.  The dynamics are chains p_a <- v_a <- c_a <- u_a per axis
// (+ yaw <- u_3; build_plan checks it), so the adjoint recursion splits into scalar first-order recurrences with CONSTANT
// coefficients, level by level:  lam_p[m] = q_p[m] + A_pp lam_p[m+1];  lam_v[m] = q_v[m] + A_pv lam_p[m+1] + A_vv lam_v[m+1];
// lam_c[m] = q_c[m] + A_pc lam_p[m+1] + A_vc lam_v[m+1] + A_cc lam_c[m+1]  (m = N .. 1, lam[N+1] = 0), each a suffix scan:
// 5 shuffle + FMA steps for up to 32 stages.  Lane = (half h, element e = m - 1); a lane carries two channels (axes 2h and
// 2h + 1: x, y | z, yaw).  gU_e[a] = r_e[a] + B_pa lam_p[e+1] + B_va lam_v[e+1] + B_ca lam_c[e+1] needs no shift.
// 10 LDS accesses and 72 ds_bpermute per sweep instead of 120 LDS accesses and 20 dependent stages; sums are taken in
// a different order than the sequential recursion (relative differences ~1e-16).
__device__ __forceinline__ void adjoint_sweep(real *sm, const LdsMap &L, int N, const double *prm_g) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));  // the lane's coefficients are re-derived per sweep: hoisted out of the iteration loop
                                    // they would be live across the objective evaluation, which sets the register peak
    const double *A = prm_g + PRM_A, *B = prm_g + PRM_B;
    const int h = lane >> 5, e = lane & 31;
    const int a0 = 2 * h, a1 = 2 * h + 1;       // control / axis of the two channels; a1 == 3 is the yaw channel
    const bool yaw1 = a1 == 3;
    const int p0 = a0, v0 = 4 + a0, c0 = 7 + a0;
    const int p1 = a1, v1 = yaw1 ? 3 : 4 + a1, c1 = yaw1 ? 3 : 7 + a1;
    const bool on = e < N;
    const real *qe = sm + L.q + (on ? e + 1 : 1) * SD;  // q_m, m = e + 1
    // level p
    real lp0 = on ? qe[p0] : RL(0.0), lp1 = on ? qe[p1] : RL(0.0);
    suffix_scan2(lp0, lp1, (real)A[p0 * SD + p0], (real)A[p1 * SD + p1], e);
    // level v
    const real sp0 = half_shift_down(lp0, 1, e), sp1 = half_shift_down(lp1, 1, e);  // lam_p[m+1]
    real lv0 = on ? fma((real)A[p0 * SD + v0], sp0, qe[v0]) : RL(0.0);
    real lv1 = (on && !yaw1) ? fma((real)A[p1 * SD + v1], sp1, qe[v1]) : RL(0.0);
    suffix_scan2(lv0, lv1, (real)A[v0 * SD + v0], yaw1 ? RL(0.0) : (real)A[v1 * SD + v1], e);
    // level c
    const real sv0 = half_shift_down(lv0, 1, e), sv1 = half_shift_down(lv1, 1, e);
    real lc0 = on ? fma((real)A[v0 * SD + c0], sv0, fma((real)A[p0 * SD + c0], sp0, qe[c0])) : RL(0.0);
    real lc1 = (on && !yaw1) ? fma((real)A[v1 * SD + c1], sv1, fma((real)A[p1 * SD + c1], sp1, qe[c1])) : RL(0.0);
    suffix_scan2(lc0, lc1, (real)A[c0 * SD + c0], yaw1 ? RL(0.0) : (real)A[c1 * SD + c1], e);
    if (on) {
        const real *re = sm + L.r + e * UD;
        real g0 = fma((real)B[p0 * UD + a0], lp0, re[a0]);
        g0 = fma((real)B[v0 * UD + a0], lv0, g0);
        g0 = fma((real)B[c0 * UD + a0], lc0, g0);
        real g1 = fma((real)B[p1 * UD + a1], lp1, re[a1]);
        if (!yaw1) {
            g1 = fma((real)B[v1 * UD + a1], lv1, g1);
            g1 = fma((real)B[c1 * UD + a1], lc1, g1);
        }
        sm[L.gU + e * UD + a0] = g0;
        sm[L.gU + e * UD + a1] = g1;
    }
    __syncthreads();
}

struct LanePlan {                // the two items + the role of this lane, in registers for one backward sweep
    real coef[2][PLAN_TERMS + 1];
    int idx[2][PLAN_TERMS];
    int out[2], out_kstride[2], aux[2], aux_kstride[2];
    LaneRole role;
    real bconst, bdelta;
};

__device__ __forceinline__ void load_lane_plan(LanePlan &lp, const double *plan_coef, const int *plan_meta) {
    int lane = threadIdx.x; asm volatile("" : "+v"(lane));   // (opaque per phase: nothing derived from the lane is carried between the phases)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int e = lane + 64 * h;
        const PlanItemMeta *m = reinterpret_cast<const PlanItemMeta *>(plan_meta) + e;
#pragma unroll
        for (int t = 0; t < PLAN_TERMS; ++t)
            if (h == 0 || t < PLAN_TERMS_LIGHT) lp.idx[h][t] = m->idx[t];
#pragma unroll
        for (int t = 0; t <= PLAN_TERMS; ++t)
            if (h == 0 || t < PLAN_TERMS_LIGHT || t == PLAN_TERMS) lp.coef[h][t] = plan_coef[e * (PLAN_TERMS + 1) + t];
        lp.out[h] = m->out; lp.out_kstride[h] = m->out_kstride; lp.aux[h] = m->aux; lp.aux_kstride[h] = m->aux_kstride;
    }
    lp.role = reinterpret_cast<const LaneRole *>(plan_meta + PLAN_ITEMS * (PLAN_TERMS + 4))[lane];
    lp.bconst = plan_coef[PLAN_ITEMS * (PLAN_TERMS + 1) + 2 * lane];
    lp.bdelta = plan_coef[PLAN_ITEMS * (PLAN_TERMS + 1) + 2 * lane + 1];
}

// Backward Riccati sweep.  Returns false when a
// control block is not positive definite.  Gains go to the scene's global scratch ([k][GAIN_STAGE], gain_offset; column 10 =
// feed-forward): stores that nothing in the sweep waits for.
//
// A stage is two LDS rounds.  Both are straight-line code (every lane runs the same instructions on its own
// host-built indices) so that the loads of a round are all in flight before the first dependent fp64 op: this
// sweep is a chain of dependent fp64 operations (32 cycles each here) and LDS round trips, nothing else.
__device__ __forceinline__ bool riccati_backward(real *sm, const LdsMap &L, const double *plan_coef,
                                                 const int *plan_meta, int N, real delta, double *gains) {
    int lane = threadIdx.x; asm volatile("" : "+v"(lane));   // (opaque per phase: nothing derived from the lane is carried between the phases)
    // the lane's plan is (re)loaded per sweep -- L2-resident words -- instead of being held for the whole
    // solve: it is dead weight (~90 VGPRs) during the objective evaluation, which sets the register peak
    LanePlan lp;
    load_lane_plan(lp, plan_coef, plan_meta);
    real *P = sm + L.P, *pv = sm + L.p;
    // terminal: P = Q_N + delta I = diag(2 Qgoal) + delta I, p = q_N
    for (int e = lane; e < 100; e += 64) {
        const int i = e / 10, j = e % 10;
        P[e] = (i == j) ? RL(2.0) * sm[L.prm + PRM_W + i] + delta : RL(0.0);
    }
    if (lane < SD) pv[lane] = sm[L.q + N * SD + lane];
    if (lane == 0) sm[L.red + 10] = RL(0.0);  // the zero cell of the plan
    __syncthreads();
    const LaneRole &R = lp.role;
#pragma unroll 1
    for (int k = N - 1; k >= 0; --k) {
        // ---- round A: the two plan items of this lane (entries of A'PA, B'PB + R_bar, B'PA, q + A'p, r_bar + B'p)
        {
            // item 0: up to 9 terms; item 1: up to 3 (the items are sorted by term count on the host: 431 terms in
            // 95 items, so the light half of the slots needs a third of the loads and FMAs)
            real v[PLAN_TERMS], u[PLAN_TERMS_LIGHT], ax[2];
#pragma unroll
            for (int t = 0; t < PLAN_TERMS; ++t) v[t] = sm[lp.idx[0][t]];
#pragma unroll
            for (int t = 0; t < PLAN_TERMS_LIGHT; ++t) u[t] = sm[lp.idx[1][t]];
#pragma unroll
            for (int h = 0; h < 2; ++h) ax[h] = sm[lp.aux[h] + k * lp.aux_kstride[h]];
            {   // four partial sums: a dependent fp64 FMA is 32 cycles on this chip, an independent one 4
                const real *c = lp.coef[0];
                real a0 = c[0] * v[0], a1 = c[1] * v[1], a2 = c[2] * v[2], a3 = fma(c[PLAN_TERMS], delta, ax[0]);
                a0 = fma(c[3], v[3], a0); a1 = fma(c[4], v[4], a1); a2 = fma(c[5], v[5], a2);
                a0 = fma(c[6], v[6], a0); a1 = fma(c[7], v[7], a1); a2 = fma(c[8], v[8], a2);
                sm[lp.out[0] + k * lp.out_kstride[0]] = (a0 + a1) + (a2 + a3);
            }
            {
                const real *c = lp.coef[1];
                const real a0 = c[0] * u[0], a1 = c[1] * u[1], a2 = c[2] * u[2], a3 = fma(c[PLAN_TERMS], delta, ax[1]);
                sm[lp.out[1] + k * lp.out_kstride[1]] = (a0 + a1) + (a2 + a3);
            }
        }
        __syncthreads();
        // ---- rounds B+C: Hm = blkdiag(3x3, h33) because the yaw chain is decoupled (build_plan checks).  With
        // adj = adjugate of the 3x3 block:  G(:,i)' Hm^-1 g = (G(0:3,i)' adj g(0:3)) / det + G(3,i) g(3) / h33.
        // The products with adj run beside det -> 1/det (ONE reciprocal on the critical path; an LDL' has four
        // sequential pivots).  Positive definite <=> leading principal minors > 0, the test an LDL' makes.
        {
            const real *h = sm + L.Hm;  // lower: h00 h10 h11 h20 h21 h22 h30 h31 h32 h33
            const real h00 = h[0], h10 = h[1], h11 = h[2], h20 = h[3], h21 = h[4], h22 = h[5], h33 = h[9];
            real gi[4], gj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                gi[a] = sm[R.gi + a * R.gi_stride];
                gj[a] = sm[R.gj + a * R.gj_stride];
            }
            const real b0 = sm[R.b_idx[0] + k * R.b_ks[0]], b1 = sm[R.b_idx[1] + k * R.b_ks[1]];
            const real b2 = sm[R.b_idx[2] + k * R.b_ks[2]];
            const real c00 = h11 * h22 - h21 * h21, c10 = h21 * h20 - h10 * h22, c20 = h10 * h21 - h11 * h20;
            const real c11 = h00 * h22 - h20 * h20, c21 = h10 * h20 - h00 * h21, c22 = h00 * h11 - h10 * h10;
            const real det = (h00 * c00 + h10 * c10) + h20 * c20;
            if (!(h00 > RL(0.0) && c22 > RL(0.0) && det > RL(0.0) && h33 > RL(0.0))) return false;
            const real rd = fast_rcp(det), r33 = fast_rcp(h33);
            const real u0 = (c00 * gj[0] + c10 * gj[1]) + c20 * gj[2];
            const real u1 = (c10 * gj[0] + c11 * gj[1]) + c21 * gj[2];
            const real u2 = (c20 * gj[0] + c21 * gj[1]) + c22 * gj[2];
            const real u3 = gj[3] * r33;
            const real t = (gi[0] * u0 + gi[1] * u1) + gi[2] * u2;
            const real base = (fma(lp.bdelta, delta, lp.bconst) + b0) + (b1 + b2);
            const real val = fma(-t, rd, fma(-gi[3], u3, base));
            if (R.gain_col >= 0) {  // K(:,c) = -Hm^-1 g_c (column 10 = feed-forward); structural zeros are not stored
                double *kk = gains + k * GAIN_STAGE;
                const int c = R.gain_col, o0 = gain_offset(0, c), o3 = gain_offset(3, c);
                if (o0 >= 0) { kk[o0] = (double)(-u0 * rd); kk[o0 + 10] = (double)(-u1 * rd); kk[o0 + 20] = (double)(-u2 * rd); }
                if (o3 >= 0) kk[o3] = (double)(-u3);
            }
            if (k > 0) {  // P_k = Q_k + delta I + A'PA - G'Hm^-1 G ; p_k = q_k + A'p - G'Hm^-1 qu ; lam_k
                sm[R.out1] = val;
            }
        }
        __syncthreads();
    }
    return true;
}

// forward roll of the Newton step: dX_0 = 0, dU_k = K_k dX_k + d_k, dX_{k+1} = A dX_k + B dU_k.
// Lane 16 a + j (j < 11) holds K_k[a][j] (j = 10: the feed-forward term; structural zeros are not loaded) -- one coalesced global load per stage,
// prefetched kFwdPrefetch stages ahead -- multiplies it with dX_k[j] from LDS, and a DPP row reduction leaves dU_k[a] in
// every lane of row a.  State i is owned by a spare lane of the row of ITS control (p_a, v_a, a_a in row a; yaw in row 3:
// B couples a state to that control only, cols_of_A_row / rows_of_B), so dX_{k+1}[i] needs no second exchange: a stage
// is ONE LDS round trip (the previous version: two, plus 11 LDS reads of the gains per lane).
constexpr int kFwdPrefetch = 5;  // divides the baked horizons 10, 20, 30
__device__ __forceinline__ int cols_of_A_row(int i, int c[3]) {  // transpose view of rows_of_A
    if (i < 3) { c[0] = i; c[1] = 4 + i; c[2] = 7 + i; return 3; }   // p_a <- p_a, v_a, a_a
    if (i == 3) { c[0] = 3; return 1; }                              // yaw
    if (i < 7) { c[0] = i; c[1] = i + 3; return 2; }                 // v_a <- v_a, a_a
    c[0] = i; return 1;                                              // a_a
}
__device__ __forceinline__ void riccati_forward(real *sm, const LdsMap &L, int N, const double *prm_g,
                                                const double *gains) {
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));  // as in adjoint_sweep: nothing of the lane's role may be hoisted out of the iteration loop
    const int a = lane >> 4, j = lane & 15;
    const double *A = prm_g + PRM_A, *B = prm_g + PRM_B;
    int own = -1;  // the state this lane produces
    if (a < 3 && j >= 11 && j <= 13) own = j == 11 ? a : (j == 12 ? 4 + a : 7 + a);
    if (a == 3 && j == 11) own = 3;
    int cols[3] = {0, 0, 0};
    real ac[3] = {RL(0.0), RL(0.0), RL(0.0)}, bc = RL(0.0);
    if (own >= 0) {
        const int nc = cols_of_A_row(own, cols);
        for (int t = 0; t < 3; ++t) ac[t] = t < nc ? (real)A[own * SD + cols[t]] : RL(0.0);
        bc = (real)B[own * UD + a];
    }
    const int goff = j <= SD ? gain_offset(a, j) : -1;
    const bool gain_lane = goff >= 0;
    real kv[kFwdPrefetch];
#pragma unroll
    for (int u = 0; u < kFwdPrefetch; ++u) kv[u] = (u < N && gain_lane) ? (real)gains[u * GAIN_STAGE + goff] : RL(0.0);
    if (lane < SD) sm[L.dX + lane] = RL(0.0);
    __syncthreads();
    const int xsrc = j < SD ? j : 0;
#pragma unroll 1
    for (int k0 = 0; k0 < N; k0 += kFwdPrefetch) {
#pragma unroll
        for (int u = 0; u < kFwdPrefetch; ++u) {
            const int k = k0 + u;
            if (k >= N) break;
            const real g = kv[u];
            kv[u] = (k + kFwdPrefetch < N && gain_lane) ? (real)gains[(k + kFwdPrefetch) * GAIN_STAGE + goff] : RL(0.0);
            const real *xk = sm + L.dX + k * SD;
            const real xj = xk[xsrc], x0 = xk[cols[0]], x1 = xk[cols[1]], x2 = xk[cols[2]];
            const real du = row_reduce<OpSum>(g * (j < SD ? xj : RL(1.0)));  // lanes j > 10 hold g = 0
            const real ax = fma(ac[2], x2, ac[0] * x0) + ac[1] * x1;
            if (j == 0) sm[L.dU + k * UD + a] = du;
            if (own >= 0) sm[L.dX + (k + 1) * SD + own] = fma(bc, du, ax);
            __syncthreads();
        }
    }
}

// Box part of the barrier function and the optimality errors at the iterate (oracle eval_iterate, second half):
// returns -mu sum log(sl su); err[0] = E_mu, err[1] = E_0 (IPOPT eq. (5) with c == 0; the collision terms enter through
// acc = their complementarity maxima from evaluate<true>).
__device__ __forceinline__ real box_errors(const real *sm, const LdsMap &L, int nvar, real mu, real s_max, const real *acc,
                                           real *err) {
    int lane = threadIdx.x; asm volatile("" : "+v"(lane));   // (opaque per phase: nothing derived from the lane is carried between the phases)
    const real *prm = sm + L.prm;
    real zs = RL(0.0), ed = RL(0.0), ec = RL(0.0), ecm = RL(0.0), lg = RL(0.0);
    for (int e = lane; e < nvar; e += 64) {
        const int i = e % UD;
        const real u = sm[L.U + e], zl = sm[L.zl + e], zu = sm[L.zu + e];
        const real sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
        zs += zl + zu;
        ed = fmax(ed, fabs(sm[L.gU + e] - zl + zu));
        ec = fmax(ec, fmax(sl * zl, su * zu));
        ecm = fmax(ecm, fmax(fabs(sl * zl - mu), fabs(su * zu - mu)));
        lg -= mu * real_log(sl * su);  // both slacks are positive and bounded by the box: no overflow
    }
    zs = wave_sum(zs); ed = wave_max(ed); ec = fmax(wave_max(ec), acc[0]); ecm = fmax(wave_max(ecm), acc[1]);
    const real is_d = s_max * fast_rcp(fmax(s_max, zs / (RL(2.0) * nvar)));
    err[0] = fmax(ed, ecm) * is_d;
    err[1] = fmax(ed, ec) * is_d;
    return wave_sum(lg);
}

__device__ __forceinline__ real next_mu(real mu, real mu_min, real kappa_mu) {
    const real m = fmax(mu_min, fmin(kappa_mu * mu, mu * sqrt(mu)));  // IPOPT eq. (7), theta_mu = 1.5
    return m < RL(10.0) * mu_min ? mu_min : m;                       // no short last level
}

// The whole solve for one scene.  w0/w_out: decision vector [X_0,U_0,...,U_{N-1},X_N] in global
// memory (warm start in, solution out; may alias).  info[4] as in the C ABI.  ybuf: [N][K][YB] doubles of scratch.
// Statement by statement the algorithm of DESIGN.md section 5, as the CPU restatement of the tests (one difference in bookkeeping only: an accepted first
// line-search trial is evaluated WITH derivatives, so the next iteration finds q, r, H6 of its iterate in LDS).
//
// Resumable (round 5): with budget > 0 the call makes at most `budget` interior-point iterations and, if the solve has not
// ended by then, PAUSES: the iterate goes to w_out as usual (X and U: X is carried by the updates X += a dX, not re-rolled, so it is
// state), the bound multipliers and five scalars (mu, delta_last, it, n_reg, ls_fail) to `rec` ([8 N + 8] doubles), the term
// multipliers are in ybuf already; sm[L.red + 2] = 1 tells the caller.  A later call with resume = true picks the solve up at
// the top of the iteration it stopped before: the derivatives, the reduced gradient and the optimality errors of the iterate
// are recomputed by the same eval_iterate(false, ...) the uninterrupted solve runs whenever its line search did not hand them
// over, and find the same bits (the speculative evaluation of an accepted first trial ran the same code on the same values:
// Xt = X + a dX is the expression of the update).  A paused-and-resumed solve therefore returns the bits of the uninterrupted
// one (tests/test_mpc_resume_gpu.py); what it costs is one derivative evaluation per pause.
#define SC_MU sm[L.red + 3]
#define SC_PHI0 sm[L.red + 4]
#define SC_DELTA_LAST sm[L.red + 7]
#define SC_A sm[L.red + 9]
#define SC_A_DU sm[L.red + 13]
#define SC_DPHI sm[L.red + 14]
__device__ __forceinline__ void solve_scene(real *sm, const LdsMap &L, int N, int K, const double *prm_g, const SolveOpts &opt,
                            const double *x_init, const double *target, const SceneIO &io, const double *w0,
                            double *w_out, int *info, const double *plan_coef, const int *plan_meta, double *ybuf,
                            double *gains, double *trace = nullptr, bool resume = false, int budget = 0,
                            double *rec = nullptr) {
    const int lane = threadIdx.x;
    const real o_tol = (real)opt.tol, o_mu_init = (real)opt.mu_init, o_bound_push = (real)opt.bound_push, o_bound_frac = (real)opt.bound_frac, o_kappa_mu = (real)opt.kappa_mu, o_tau_min = (real)opt.tau_min, o_eta_phi = (real)opt.eta_phi, o_s_max = (real)opt.s_max, o_kappa_sigma = (real)opt.kappa_sigma;
    const real o_kappa_eps = (real)opt.kappa_eps, o_maj = (real)opt.maj;
    for (int e = lane; e < PRM_LDS_LEN; e += 64) sm[L.prm + e] = (real)prm_g[e];
    if (lane < SD) {
        sm[L.xinit + lane] = (real)x_init[lane];
        sm[L.target + lane] = (real)target[lane];
    }
    if (lane < N - 1) {  // rot of ref yaw, :174-185
        const real yaw = (real)io.ref[lane * SD + 3];
        sm[L.cy + lane] = cos(yaw);
        sm[L.sy + lane] = sin(-yaw);
    }
    __syncthreads();
    const real *prm = sm + L.prm;
    for (int e = lane; e < 56 + 10 + 40; e += 64) sm[L.M + e] = RL(0.0);  // structurally-zero outputs stay zero
    if (lane < N - 1) {  // constant part of Q on the rotated (px,py) and (vx,vy) blocks
        const real cy = sm[L.cy + lane], sy = sm[L.sy + lane];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const real w0q = RL(2.0) * prm[PRM_W + 10 + 4 * b], w1q = RL(2.0) * prm[PRM_W + 11 + 4 * b];
            sm[L.rotQ + lane * 6 + b * 3 + 0] = cy * cy * w0q + sy * sy * w1q;
            sm[L.rotQ + lane * 6 + b * 3 + 1] = -cy * sy * w0q + sy * cy * w1q;
            sm[L.rotQ + lane * 6 + b * 3 + 2] = sy * sy * w0q + cy * cy * w1q;
        }
    }
    const int nvar = UD * N;
    if (resume) {  // the paused iterate, its bound multipliers (the term multipliers never left ybuf)
        for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.X + e] = (real)w0[14 * (e / SD) + (e % SD)];
        for (int e = lane; e < nvar; e += 64) {
            sm[L.U + e] = (real)w0[14 * (e / UD) + 10 + (e % UD)];
            sm[L.zl + e] = (real)rec[e];
            sm[L.zu + e] = (real)rec[nvar + e];
        }
        __syncthreads();
    } else {
    // warm start pushed into the interior
    for (int e = lane; e < nvar; e += 64) {
        const int k = e / UD, i = e % UD;
        const real lb = prm[PRM_LB + i], ub = prm[PRM_UB + i];
        const real pl = fmin(o_bound_push * fmax(RL(1.0), fabs(lb)), o_bound_frac * (ub - lb));
        const real pu = fmin(o_bound_push * fmax(RL(1.0), fabs(ub)), o_bound_frac * (ub - lb));
        sm[L.U + e] = fmin(fmax((real)w0[14 * k + 10 + i], lb + pl), ub - pu);
    }
    if (lane < SD) sm[L.X + lane] = sm[L.xinit + lane];
    for (int e = lane; e < (N - 1) * K; e += 64)  // no collision term has multipliers yet
        *reinterpret_cast<double2 *>(ybuf + (size_t)e * YB) = make_double2(-1.0, -1.0);
    __syncthreads();
    }
    if (!resume) {  // rollout X_{k+1} = A X_k + B U_k + c
        const real *c = prm + PRM_C;
        const int row = lane < SD ? lane : 0;
        real arow[SD], brow[UD];  // row `lane` of A and B
#pragma unroll
        for (int j = 0; j < SD; ++j) arow[j] = (real)prm_g[PRM_A + row * SD + j];
#pragma unroll
        for (int j = 0; j < UD; ++j) brow[j] = (real)prm_g[PRM_B + row * UD + j];
        for (int k = 0; k < N; ++k) {
            if (lane < SD) {
                real a = RL(0.0);
#pragma unroll
                for (int j = 0; j < SD; ++j) a += arow[j] * sm[L.X + k * SD + j];
#pragma unroll
                for (int j = 0; j < UD; ++j) a += brow[j] * sm[L.U + k * UD + j];
                sm[L.X + (k + 1) * SD + lane] = a + c[lane];
            }
            __syncthreads();
        }
    }
    const real mu_min = o_tol * (real)opt.mu_min_fac;
    // The solver's wave-uniform scalars live in LDS BETWEEN the phases (every lane stores the same value and reads its own store
    // back: no barrier needed), and a phase takes what it needs by value: as ordinary variables they were ~25 registers pinned
    // across the objective evaluation and the Riccati sweep, the two phases that set the kernel's register count.
    SC_MU = resume ? (real)rec[2 * nvar + 0] : o_mu_init;
    SC_PHI0 = RL(0.0);
    real *err = sm + L.red + 5, *acc = sm + L.red + 11;
    err[0] = RL(0.0); err[1] = RL(0.0); acc[0] = RL(0.0); acc[1] = RL(0.0);
    // derivatives, reduced gradient and optimality errors of the iterate under SC_MU (oracle eval_iterate)
    auto eval_iterate = [&](bool derivs_in_lds, real J_known, long long *tclk) -> real {
        const real J = derivs_in_lds ? J_known
                                     : evaluate<true>(sm, L, io, N, K, sm + L.X, sm + L.U, SC_MU, o_kappa_sigma,
                                                      o_maj * SC_MU / o_mu_init, ybuf, acc, tclk);
        __syncthreads();
        #ifdef AMK_ADJOINT_SEQ
        adjoint_sweep_seq(sm, L, N, prm_g);
#else
        adjoint_sweep(sm, L, N, prm_g);
#endif
        return J + box_errors(sm, L, nvar, SC_MU, o_s_max, acc, err);
    };
    // starting barrier parameter: the duals start on the central path of whatever SC_MU is chosen, so mu_init is lowered
    // level by level while the start already solves that level's barrier problem to the accuracy at which the barrier
    // update below would leave it (a warm start from a previous solution begins several levels down)
#pragma unroll 1
    for (;;) {
        if (!resume) {
            for (int e = lane; e < nvar; e += 64) {
                const int i = e % UD;
                const real u = sm[L.U + e];
                sm[L.zl + e] = SC_MU / (u - prm[PRM_LB + i]);
                sm[L.zu + e] = SC_MU / (prm[PRM_UB + i] - u);
            }
            __syncthreads();
        }
        SC_PHI0 = eval_iterate(false, RL(0.0), nullptr);
        if (resume || !(err[0] <= o_kappa_eps * SC_MU) || SC_MU <= mu_min) break;   // (a resumed solve only needs the evaluation)
        SC_MU = next_mu(SC_MU, mu_min, o_kappa_mu);
    }
    SC_DELTA_LAST = resume ? (real)rec[2 * nvar + 1] : RL(0.0);
    const int it_begin = resume ? (int)rec[2 * nvar + 2] : 0;
    int status = 1, n_reg = resume ? (int)rec[2 * nvar + 3] : 0, ls_fail = resume ? (int)rec[2 * nvar + 4] : 0, it = 0;
    bool paused = false;
#pragma unroll 1
    for (it = it_begin; it < opt.max_iter; ++it) {
        const long long t0 = AMK_CLK();
        long long tclk[3] = {0, 0, 0};
        if (kTrace && trace && lane == 0) {
            trace[16 * it + 0] = SC_PHI0; trace[16 * it + 1] = err[1]; trace[16 * it + 2] = SC_MU; trace[16 * it + 3] = err[0];
        }
        if (SC_MU <= mu_min && err[0] <= o_tol) { status = 0; break; }  // the last barrier problem is solved to tol
        if (budget > 0 && it - it_begin >= budget) { paused = true; break; }  // this launch's share is used up: continue later
        if (it > 0 && err[0] <= o_kappa_eps * SC_MU && SC_MU > mu_min) {   // barrier update (one level per iteration)
            SC_MU = next_mu(SC_MU, mu_min, o_kappa_mu);
            __syncthreads();
            SC_PHI0 = eval_iterate(false, RL(0.0), kTrace ? tclk : nullptr);
        }
        __syncthreads();
        const long long t1 = AMK_CLK();
        const real tau = fmax(o_tau_min, RL(1.0) - SC_MU);
        for (int e = lane; e < nvar; e += 64) {
            const int i = e % UD;
            const real u = sm[L.U + e];
            const real sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
            const real isl = fast_rcp(sl), isu = fast_rcp(su);
            sm[L.rb + e] = sm[L.r + e] - SC_MU * isl + SC_MU * isu;
            sm[L.Rb + e] = RL(2.0) * prm[PRM_W + 20 + i] + sm[L.zl + e] * isl + sm[L.zu + e] * isu;
        }
        for (int e = lane; e < 56 + 10 + 40; e += 64) sm[L.M + e] = RL(0.0);  // (the line search's trial states lay here)
        __syncthreads();
        real delta = RL(0.0);
        int reg_now = 0;
        bool ok = riccati_backward(sm, L, plan_coef, plan_meta, N, delta, gains);
        while (!ok) {
            __syncthreads();
            if (delta == RL(0.0)) delta = (SC_DELTA_LAST == RL(0.0)) ? RL(1.0) : fmax(RL(1e-20), SC_DELTA_LAST / RL(3.0));
            else delta *= (SC_DELTA_LAST == RL(0.0)) ? RL(100.0) : RL(8.0);
            ++reg_now;
            if (delta > (AMK_REAL_F32 ? RL(1e30) : RL(1e40))) break;
            ok = riccati_backward(sm, L, plan_coef, plan_meta, N, delta, gains);
        }
        if (!ok) { status = 2; break; }
        n_reg += reg_now;
        if (delta > RL(0.0)) SC_DELTA_LAST = delta;
        const long long t2 = AMK_CLK();
        riccati_forward(sm, L, N, prm_g, gains);
        const long long t3 = AMK_CLK();
        // dual steps, fraction to the boundary, directional derivative: control box ...
        real a_pr = RL(1.0), a_du = RL(1.0), dphi = RL(0.0);
        for (int e = lane; e < nvar; e += 64) {
            const int i = e % UD;
            const real u = sm[L.U + e], zl = sm[L.zl + e], zu = sm[L.zu + e], du = sm[L.dU + e];
            const real sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
            const real isl = fast_rcp(sl), isu = fast_rcp(su);
            const real dzl = SC_MU * isl - zl - (zl * isl) * du;
            const real dzu = SC_MU * isu - zu + (zu * isu) * du;
            sm[L.dzl + e] = dzl;
            sm[L.dzu + e] = dzu;
            // fraction to the boundary: one reciprocal per ratio, only the binding side of each bound
            const real idu = fast_rcp(du);
            if (du < RL(0.0)) a_pr = fmin(a_pr, -tau * sl * idu);
            if (du > RL(0.0)) a_pr = fmin(a_pr, tau * su * idu);
            if (dzl < RL(0.0)) a_du = fmin(a_du, -tau * zl * fast_rcp(dzl));
            if (dzu < RL(0.0)) a_du = fmin(a_du, -tau * zu * fast_rcp(dzu));
            dphi += (sm[L.gU + e] - SC_MU * isl + SC_MU * isu) * du;
        }
        a_pr = wave_min(a_pr); SC_A_DU = wave_min(a_du); SC_DPHI = wave_sum(dphi);   // (parked: see SC_MU)
        // ... and the multipliers of the collision terms (they do not enter the merit function)
        update_term_multipliers(sm, L, io, N, K, SC_MU, tau, o_kappa_sigma, ybuf);
        // backtracking Armijo line search on the barrier function
        const long long t4 = AMK_CLK();
        SC_A = a_pr;
        real phi_t = RL(0.0), J_t = RL(0.0);
        bool accepted = false, have_derivs = false;
        const bool speculate = it + 1 < opt.max_iter;  // the last iteration has no successor to hand derivatives to
        // a Newton step whose predicted decrease is below the rounding level of phi is taken as it is
        const bool tiny = -SC_DPHI <= RL(100.0) * (AMK_REAL_F32 ? RL(1.1920929e-7) : RL(2.220446049250313e-16)) * (RL(1.0) + fabs(SC_PHI0));
        for (int ls = 0; ls < opt.max_ls; ++ls) {
            __syncthreads();
            for (int e = lane; e < nvar; e += 64) sm[L.Ut + e] = sm[L.U + e] + SC_A * sm[L.dU + e];
            for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.Xt + e] = sm[L.X + e] + SC_A * sm[L.dX + e];
            __syncthreads();
            if (ls == 0 && speculate) {
                J_t = evaluate<true>(sm, L, io, N, K, sm + L.Xt, sm + L.Ut, SC_MU, o_kappa_sigma, o_maj * SC_MU / o_mu_init, ybuf, acc);
                have_derivs = true;
            } else {
                J_t = evaluate<false>(sm, L, io, N, K, sm + L.Xt, sm + L.Ut, SC_MU, o_kappa_sigma, RL(0.0), ybuf, nullptr);
                have_derivs = false;  // the iterate moves on to a point that has no derivatives yet
            }
            real lg = RL(0.0);
            for (int e = lane; e < nvar; e += 64) {
                const int i = e % UD;
                const real u = sm[L.Ut + e];
                lg -= SC_MU * real_log((u - prm[PRM_LB + i]) * (prm[PRM_UB + i] - u));
            }
            phi_t = J_t + wave_sum(lg);
            if (tiny || phi_t <= SC_PHI0 + o_eta_phi * SC_A * SC_DPHI) { accepted = true; break; }
            if (ls + 1 < opt.max_ls) SC_A *= RL(0.5);
        }
        if (!accepted) {  // no decrease found (rounding level of phi): stay; the duals still move
            ++ls_fail;
            SC_A = RL(0.0);
            have_derivs = false;
        }
        if (kTrace && trace && lane == 0) {
            trace[16 * it + 4] = SC_A; trace[16 * it + 5] = a_pr; trace[16 * it + 6] = SC_A_DU; trace[16 * it + 7] = SC_DPHI;
            trace[16 * it + 8] = (double)(t1 - t0); trace[16 * it + 9] = (double)(t2 - t1);
            trace[16 * it + 10] = (double)(t3 - t2); trace[16 * it + 11] = (double)(t4 - t3);
            trace[16 * it + 12] = (double)(AMK_CLK() - t4); trace[16 * it + 13] = (double)tclk[0]; trace[16 * it + 14] = delta; trace[16 * it + 15] = (double)tclk[2];
        }
        __syncthreads();
        for (int e = lane; e < (N + 1) * SD; e += 64) sm[L.X + e] += SC_A * sm[L.dX + e];
        for (int e = lane; e < nvar; e += 64) {
            const int i = e % UD;
            const real u = sm[L.U + e] + SC_A * sm[L.dU + e];
            sm[L.U + e] = u;
            const real sl = u - prm[PRM_LB + i], su = prm[PRM_UB + i] - u;
            real zl = sm[L.zl + e] + SC_A_DU * sm[L.dzl + e], zu = sm[L.zu + e] + SC_A_DU * sm[L.dzu + e];
            zl = fmax(fmin(zl, o_kappa_sigma * SC_MU / sl), SC_MU / (o_kappa_sigma * sl));
            zu = fmax(fmin(zu, o_kappa_sigma * SC_MU / su), SC_MU / (o_kappa_sigma * su));
            sm[L.zl + e] = zl;
            sm[L.zu + e] = zu;
        }
        __syncthreads();
        if (speculate) SC_PHI0 = eval_iterate(have_derivs, J_t, nullptr);
    }
    __syncthreads();
    for (int e = lane; e < (N + 1) * SD; e += 64) w_out[14 * (e / SD) + (e % SD)] = sm[L.X + e];
    for (int e = lane; e < nvar; e += 64) w_out[14 * (e / UD) + 10 + (e % UD)] = sm[L.U + e];
    if (paused) {
        for (int e = lane; e < nvar; e += 64) {
            rec[e] = (double)sm[L.zl + e];
            rec[nvar + e] = (double)sm[L.zu + e];
        }
        if (lane == 0) {
            rec[2 * nvar + 0] = (double)SC_MU; rec[2 * nvar + 1] = (double)SC_DELTA_LAST; rec[2 * nvar + 2] = (double)it;
            rec[2 * nvar + 3] = (double)n_reg; rec[2 * nvar + 4] = (double)ls_fail;
        }
    }
    if (lane == 0) {  // kept for the control-step bookkeeping of the calling kernel
        sm[L.red + 0] = (real)status;
        sm[L.red + 1] = (real)it;
        sm[L.red + 2] = paused ? RL(1.0) : RL(0.0);
    }
    if (info && lane == 0) {
        info[0] = status;
        info[1] = it;
        info[2] = n_reg;
        info[3] = ls_fail;
    }
}

#undef SC_MU
#undef SC_PHI0
#undef SC_DELTA_LAST
#undef SC_A
#undef SC_A_DU
#undef SC_DPHI
__device__ __forceinline__ int sm_status(const real *sm, const LdsMap &L) { return (int)sm[L.red + 0]; }
__device__ __forceinline__ int sm_iters(const real *sm, const LdsMap &L) { return (int)sm[L.red + 1]; }
__device__ __forceinline__ bool sm_paused(const real *sm, const LdsMap &L) { return sm[L.red + 2] != RL(0.0); }


#undef RL
}  // namespace AMK_RNS
