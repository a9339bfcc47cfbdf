// exp / log in double precision for the solve's collision terms and barrier sums (gfx950).
//
// Why not the device library's: its fp64 log evaluates in double-double arithmetic (~60 DEPENDENT fp64 operations),
// its exp is a degree-11 Horner chain behind a two-step reduction.  The solve is one wavefront per scene walking a
// chain of dependent operations, so the length of these chains -- not the instruction count -- is
// what an interior-point iteration pays: ten log and six exp latencies per iteration, ~7 of its ~30 us.  The versions
// here have the SHORTEST dependency chain that still gives ~1 ulp (measured on the device against the host's libm,
// tests/test_fast_math_gpu.py: max 1.5 ulp / 1.0 ulp):
//   fast_log  m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), the classical odd series in s with the 7-coefficient
//             minimax fit of the FreeBSD/fdlibm e_log.c lineage (Lg1..Lg7), evaluated as two interleaved Horner
//             chains in s^4; k ln2 added in hi/lo parts.  ~18 dependent operations.
//   fast_exp  k = rint(x / ln2), r = x - k ln2 (hi/lo), e^r = 1 + r + r^2 Q(r) with Q the degree-11 Taylor tail
//             evaluated by Estrin's scheme (depth 4), scaled by 2^k with v_ldexp.  ~11 dependent operations.
// Domain: fast_log wants a positive, finite, normal argument (every call site passes products of positive slacks or
// 1 + e^x >= 1); denormals are rescaled, 0 / inf / negative / NaN arguments get the IEEE answers by selects.  fast_exp clamps x to [-745.2, 709.7] (results 0 / inf
// outside are not needed: the callers' arguments are bounded above by 32 x the drone radius).
#pragma once
#include <hip/hip_runtime.h>

namespace amk {

__device__ __forceinline__ double rcp_f64(double x) {  // 1/x, <= 1 ulp: v_rcp_f64 (24 bits) + one cubic correction
    double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(fma(e, e, e), r, r);  // r (1 + e + e^2): relative error e^3 ~ 1e-22
}

__device__ __forceinline__ double fast_log(double x) {
#pragma clang fp contract(off)  // every step below is the IEEE operation it is written as (the tests' CPU restatement repeats them)
#ifdef AMK_FAST_LOG_LIBRARY_FALLBACK  // (the library's log inlined at every call site: ~150 instructions each, 12 sites)
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);  // <= 0, denormal, inf, NaN
#else
    // outside the domain (never reached by the solver): NaN for negative / NaN, -inf for 0, +inf for +inf; a denormal is
    // scaled into the normal range first
    const bool special = !(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308);
    const double x_in = x;
    double bias = 0.0;
    if (special) {
        if (x > 0.0 && x < 2.2250738585072014e-308) { x = x * 18014398509481984.0; bias = -54.0; }  // 2^54
        else x = 1.0;
    }
#endif
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? 2.0 * m : m;                       // [sqrt(1/2), sqrt(2))
    k = lo ? k - 1 : k;
    const double f = m - 1.0;                   // exact
    const double s = f * rcp_f64(2.0 + f);
#ifdef AMK_FAST_LOG_LIBRARY_FALLBACK
    const double dk = (double)k;
#else
    const double dk = (double)k + bias;
#endif
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    // log(1 + f) = f - hfsq + s (hfsq + R);  + k ln2 in two parts (ln2_hi has 21 trailing zero bits: k ln2_hi is exact)
    const double res = dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
#ifdef AMK_FAST_LOG_LIBRARY_FALLBACK
    return res;
#else
    if (special && !(bias < 0.0))
        return x_in == 0.0 ? -__builtin_huge_val() : (x_in > 1.0 ? __builtin_huge_val() : __builtin_nan(""));
    return res;
#endif
}

__device__ __forceinline__ double fast_exp(double x) {
#pragma clang fp contract(off)
    x = fmin(fmax(x, -745.2), 709.7);
    const double kd = __builtin_rint(x * 1.44269504088896338700e+00);
    double r = fma(-kd, 6.93147180369123816490e-01, x);
    r = fma(-kd, 1.90821492927058770002e-10, r);  // |r| <= 0.3466 (+ rounding)
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    // Q(r) = sum_{i=0}^{11} r^i / (i + 2)!  (truncation < 2^-57 relative to e^r on |r| <= 0.3466)
    const double q01 = fma(r, 1.66666666666666666667e-01, 5.00000000000000000000e-01);
    const double q23 = fma(r, 8.33333333333333333333e-03, 4.16666666666666666667e-02);
    const double q45 = fma(r, 1.98412698412698412698e-04, 1.38888888888888888889e-03);
    const double q67 = fma(r, 2.75573192239858906526e-06, 2.48015873015873015873e-05);
    const double q89 = fma(r, 2.50521083854417187751e-08, 2.75573192239858906526e-07);
    const double qab = fma(r, 1.60590438368216145994e-10, 2.08767569878680989792e-09);
    const double q03 = fma(r2, q23, q01), q47 = fma(r2, q67, q45), q8b = fma(r2, qab, q89);
    const double Q = fma(r8, q8b, fma(r4, q47, q03));
    const double p = fma(r2, Q, r);
    return __builtin_amdgcn_ldexp(1.0 + p, (int)kd);
}

}  // namespace amk
