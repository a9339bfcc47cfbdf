/* HARNESS (CPU only, not product, not oracle).  Prototypes of changes to the interior-point method of oracle/mpc_oracle.c,
 * built on the oracle's own static pieces (the file is #included, nothing is copied): proto_solve() is mpco_solve() with
 * switches.  With every switch off it returns mpco_solve()'s bits (checked by run.py). */
#include <stdio.h>
#include "../../../oracle/mpc_oracle.c"

typedef struct {
    int diag;          /* print the parts of E_mu per iteration */
    int variant;       /* bit field of experiments, see proto_solve */
    double p[8];       /* experiment parameters */
} proto_opts;

static double eval_iterate_parts(work *w, const pview *pp, const double *lbu, const double *ubu, double mu,
                                 const mpco_opts *opt, double *err, double *parts) {
    const int N = w->N;
    double acc[2];
    const double maj = opt->maj * mu / opt->mu_init;
    double phi = total_merit(w, pp, w->X, w->U, mu, opt->kappa_sigma, maj, 1, acc);
    double lam[SD], ln[SD];
    memcpy(lam, w->q[N], sizeof lam);
    for (int k = N - 1; k >= 0; --k) {
        for (int i = 0; i < UD; ++i) {
            double a = 0.0;
            for (int l = 0; l < SD; ++l) a += w->B[l * UD + i] * lam[l];
            w->gU[k][i] = w->r[k][i] + a;
        }
        if (k > 0) {
            for (int i = 0; i < SD; ++i) {
                double a = 0.0;
                for (int l = 0; l < SD; ++l) a += w->A[l * SD + i] * lam[l];
                ln[i] = w->q[k][i] + a;
            }
            memcpy(lam, ln, sizeof lam);
        }
    }
    double zsum = 0.0, e_d = 0.0, e_c = acc[0], e_cm = acc[1], e_cmb = 0.0;
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < UD; ++i) {
            double sl = w->U[k][i] - lbu[i], su = ubu[i] - w->U[k][i];
            zsum += w->zl[k][i] + w->zu[k][i];
            e_d = dmax(e_d, fabs(w->gU[k][i] - w->zl[k][i] + w->zu[k][i]));
            e_c = dmax(e_c, dmax(sl * w->zl[k][i], su * w->zu[k][i]));
            e_cmb = dmax(e_cmb, dmax(fabs(sl * w->zl[k][i] - mu), fabs(su * w->zu[k][i] - mu)));
            phi -= mu * slog(sl * su);
        }
    const double s_d = dmax(opt->s_max, zsum / (2.0 * UD * N)) / opt->s_max;
    if (parts) { parts[0] = e_d; parts[1] = e_cmb; parts[2] = acc[1]; parts[3] = s_d; }
    e_cm = dmax(e_cm, e_cmb);
    err[0] = dmax(e_d, e_cm) / s_d;
    err[1] = dmax(e_d, e_c) / s_d;
    return phi;
}


/* ---- experiment 1: the barrier update as a primal-dual method makes it.  In the iteration that lowers mu the terms' epigraph
 * variable t is NOT re-optimised for the new mu before the step: slacks from mu_s (= the old mu), barrier targets from mu; the
 * t row of the Newton system then has the residual c (1 - mu / mu_s), eliminated with the row. */
static double term_derivs_f(const tgeo *G, double lam, double mu, double mu_s, double kappa_sigma, double maj, double y1, double y2,
                            double *g6, double *H6) {
    const double c = lam * G->g;
    double w1, w2;
    slacks(G->s, mu_s, &w1, &w2);
    const double t = 0.5 * (w1 + w2), iw1 = 1.0 / w1, iw2 = 1.0 / w2;
    double a1, a2;
    term_mult(y1, y2, mu_s, kappa_sigma, iw1, iw2, &a1, &a2);
    const double D1 = a1 * iw1, D2 = a2 * iw2, Dh = D1 + D2, dD = D2 - D1, iDh = 1.0 / Dh;
    const double sig = a1 - a2, e = 1.0 - a1 - a2;
    const double bs = mu * (iw1 - iw2);
    const double Psi = t - mu * slog(w1 * w2);
    const double gt = 1.0 - mu / mu_s; /* residual of the t row / c */
    const double ir = G->ir, lgp = lam * G->gp;
    const double f = -gt * iDh;
    for (int i = 0; i < 3; ++i) {
        g6[i] += -lgp * Psi * G->n[i] - c * bs * G->tv[i] * ir + f * (e * (-lgp * G->n[i]) + c * dD * (-G->tv[i] * ir));
        g6[3 + i] += c * bs * G->n[i] + f * (c * dD * G->n[i]);
    }
    const double k1 = sig - e * dD * iDh, k2 = 4.0 * c * D1 * D2 * iDh + maj * c / t, k3 = e * e * iDh / c;
    const double ir2 = ir * ir, cs = c * sig;
    const double al = k2 * ir2, be = k2 * ir;
    const double Bc = Psi * lgp * ir - cs * G->s * ir2;
    const double Ac = Psi * lam * G->gpp - Bc - k3 * lgp * lgp;
    const double Cc = k1 * lgp * ir - cs * ir2;
    const double Dc = -k1 * lgp + cs * ir, Ec = -cs * ir;
    const double *n = G->n, *tv = G->tv;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double hpp = al * tv[i] * tv[j] + Ac * n[i] * n[j] + Cc * (n[i] * tv[j] + tv[i] * n[j]) + (i == j ? Bc : 0.0);
            double hvp = n[i] * (Dc * n[j] - be * tv[j]) + (i == j ? Ec : 0.0);
            H6[i * 6 + j] += hpp;
            H6[(3 + i) * 6 + j] += hvp;
            H6[j * 6 + 3 + i] += hvp;
            H6[(3 + i) * 6 + 3 + j] += k2 * n[i] * n[j];
        }
    return c * Psi;
}
static void term_step_f(const tgeo *G, double mu, double mu_s, double tau, double kappa_sigma, double *y1, double *y2, const double *dp,
                        const double *dv) {
    double w1, w2, a1, a2;
    slacks(G->s, mu_s, &w1, &w2);
    const double iw1 = 1.0 / w1, iw2 = 1.0 / w2;
    term_mult(*y1, *y2, mu_s, kappa_sigma, iw1, iw2, &a1, &a2);
    const double D1 = a1 * iw1, D2 = a2 * iw2, Dh = D1 + D2, dD = D2 - D1;
    const double e = 1.0 - a1 - a2, gt = 1.0 - mu / mu_s;
    const double ndp = G->n[0] * dp[0] + G->n[1] * dp[1] + G->n[2] * dp[2];
    const double ds = -(G->tv[0] * dp[0] + G->tv[1] * dp[1] + G->tv[2] * dp[2]) * G->ir +
                      (G->n[0] * dv[0] + G->n[1] * dv[1] + G->n[2] * dv[2]);
    const double dlc = -(G->gp / G->g) * ndp;
    const double dt = (-gt - e * dlc - dD * ds) / Dh;
    const double dy1 = mu * iw1 - a1 - D1 * (dt - ds), dy2 = mu * iw2 - a2 - D2 * (dt + ds);
    double al = 1.0;
    if (dy1 < 0.0) al = dmin(al, -tau * a1 / dy1);
    if (dy2 < 0.0) al = dmin(al, -tau * a2 / dy2);
    *y1 = a1 + al * dy1;
    *y2 = a2 + al * dy2;
}
/* derivatives q, Q (and r, Rd) at the iterate with the terms in the frozen-t form, then the reduced gradient */
static void eval_frozen(work *w, const pview *pp, double mu, double mu_s, const mpco_opts *opt) {
    const int N = w->N, K = w->K;
    const double maj = opt->maj * mu / opt->mu_init;
    for (int k = 0; k < N; ++k) {
        stage_smooth(pp, k, w->X[k + 1], w->U[k], w->q[k + 1], w->Q[k + 1], w->r[k], w->Rd[k]);
        if (k >= N - 1) continue;
        const double p[3] = {w->X[k + 1][0], w->X[k + 1][1], w->X[k + 1][2]}, v[3] = {w->X[k + 1][4], w->X[k + 1][5], w->X[k + 1][6]};
        double g6[6] = {0}, H6[36] = {0};
        for (int j = 0; j < K; ++j) {
            tgeo G;
            term_geo(p, v, pp->obs + 3 * (K * k + j), pp->radius, 1, &G);
            if (!(pp->lam * G.g > 0.0)) continue;
            term_derivs_f(&G, pp->lam, mu, mu_s, opt->kappa_sigma, maj, w->y1[k][j], w->y2[k][j], g6, H6);
        }
        for (int i = 0; i < 6; ++i) {
            w->q[k + 1][PV[i]] += g6[i];
            for (int j = 0; j < 6; ++j) w->Q[k + 1][PV[i] * SD + PV[j]] += H6[i * 6 + j];
        }
    }
}

/* diagnostics: the dual infeasibility max |gU - zl + zu| at the iterate when the terms' gradient takes mu_psi in Psi (the factor of grad c)
 * and mu_bs in the smoothed sign (the factor of grad s) */
static double ed_split(work *w, const pview *pp, double mu_psi, double mu_bs) {
    const int N = w->N, K = w->K;
    double q[MAXN + 1][SD], r[MAXN][UD];
    for (int k = 0; k < N; ++k) {
        double Q[SD * SD], Rd[UD];
        stage_smooth(pp, k, w->X[k + 1], w->U[k], q[k + 1], Q, r[k], Rd);
        if (k >= N - 1) continue;
        const double p[3] = {w->X[k + 1][0], w->X[k + 1][1], w->X[k + 1][2]}, v[3] = {w->X[k + 1][4], w->X[k + 1][5], w->X[k + 1][6]};
        for (int j = 0; j < K; ++j) {
            tgeo G;
            term_geo(p, v, pp->obs + 3 * (K * k + j), pp->radius, 1, &G);
            const double c = pp->lam * G.g;
            if (!(c > 0.0)) continue;
            double w1, w2;
            slacks(G.s, mu_psi, &w1, &w2);
            const double Psi = 0.5 * (w1 + w2) - mu_psi * slog(w1 * w2);
            slacks(G.s, mu_bs, &w1, &w2);
            const double bs = mu_bs * (1.0 / w1 - 1.0 / w2);
            const double lgp = pp->lam * G.gp;
            for (int i = 0; i < 3; ++i) {
                q[k + 1][PV[i]] += -lgp * Psi * G.n[i] - c * bs * G.tv[i] * G.ir;
                q[k + 1][PV[3 + i]] += c * bs * G.n[i];
            }
        }
    }
    double lam[SD], ln[SD], ed = 0.0;
    memcpy(lam, q[N], sizeof lam);
    for (int k = N - 1; k >= 0; --k) {
        for (int i = 0; i < UD; ++i) {
            double a = 0.0;
            for (int l = 0; l < SD; ++l) a += w->B[l * UD + i] * lam[l];
            ed = dmax(ed, fabs(r[k][i] + a - w->zl[k][i] + w->zu[k][i]));
        }
        if (k > 0) {
            for (int i = 0; i < SD; ++i) {
                double a = 0.0;
                for (int l = 0; l < SD; ++l) a += w->A[l * SD + i] * lam[l];
                ln[i] = q[k][i] + a;
            }
            memcpy(lam, ln, sizeof lam);
        }
    }
    return ed;
}

/* mpco_solve with switches.  variant bits:
 *   1  : (reserved)
 */
int proto_solve(const double *P, const double *w0, const double *lbu, const double *ubu, int N, int K, double dt,
                const mpco_opts *opt_in, const proto_opts *po, double *w_out, int *info, double *stats) {
    mpco_opts opt;
    if (opt_in) opt = *opt_in;
    else mpco_default_opts(&opt);
    if (N > MAXN || K > MAXKO) return -1;
    work *w = (work *)calloc(1, sizeof(work));
    w->N = N;
    w->K = K;
    pview pp = split_p(P, N, K);
    mpco_affine(pp.tau, dt, w->A, w->B, w->c);
    double pl[UD], pu[UD], parts[4];
    for (int i = 0; i < UD; ++i) {
        pl[i] = dmin(opt.bound_push * dmax(1.0, fabs(lbu[i])), opt.bound_frac * (ubu[i] - lbu[i]));
        pu[i] = dmin(opt.bound_push * dmax(1.0, fabs(ubu[i])), opt.bound_frac * (ubu[i] - lbu[i]));
    }
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < UD; ++i) w->U[k][i] = dmin(dmax(UK(w0, k)[i], lbu[i] + pl[i]), ubu[i] - pu[i]);
    rollout(w, pp.x_init, w->U, w->X);
    const double mu_min = opt.tol * opt.mu_min_fac;
    double mu = opt.mu_init, err[2] = {DBL_MAX, DBL_MAX}, phi0;
    for (;;) {
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < UD; ++i) {
                w->zl[k][i] = mu / (w->U[k][i] - lbu[i]);
                w->zu[k][i] = mu / (ubu[i] - w->U[k][i]);
            }
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < K; ++j) w->y1[k][j] = w->y2[k][j] = -1.0;
        phi0 = eval_iterate_parts(w, &pp, lbu, ubu, mu, &opt, err, parts);
        if (!(err[0] <= opt.kappa_eps * mu) || mu <= mu_min) break;
        mu = next_mu(mu, mu_min, &opt);
    }
    double delta_last = 0.0;
    int status = 1, n_reg = 0, ls_fail = 0, it, n_ls = 0;
    double(*rbp)[UD] = w->rb, (*Rbp)[UD] = w->Rb;
    double mu_s = 0.0; double gU_true[MAXN][UD];
    double prev_delta = 0.0;
    int spec_n = 0, spec_hit = 0, spec_waste = 0, spec_more = 0, nospec_n = 0, nospec_fail = 0;
    for (it = 0; it < opt.max_iter; ++it) {
        mu_s = 0.0;
        if (po->diag)
            printf("  it %2d mu %8.2e Emu %9.2e | e_d %9.2e cm_box %9.2e cm_term %9.2e s_d %.2f", it, mu, err[0], parts[0], parts[1], parts[2], parts[3]);
        if (mu <= mu_min && err[0] <= opt.tol) { status = 0; if (po->diag) printf("\n"); break; }
        const int tail = (po->variant & 2) && mu <= po->p[1];
        if (it > 0 && err[0] <= (tail ? po->p[2] : opt.kappa_eps) * mu && mu > mu_min) {
            const double mu_old = mu;
            if (tail) { mu = dmax(mu_min, po->p[3] * mu); if (mu < po->p[4] * mu_min) mu = mu_min; }
            else mu = next_mu(mu, mu_min, &opt);
            phi0 = eval_iterate_parts(w, &pp, lbu, ubu, mu, &opt, err, parts);
            if (po->diag) printf(" -> mu %8.2e Emu %9.2e (e_d %9.2e cm_box %9.2e cm_term %9.2e)", mu, err[0], parts[0], parts[1], parts[2]);
            if (po->diag) printf("\n        e_d(old,old) %.2e (new,old) %.2e (old,new) %.2e (new,new) %.2e\n       ", ed_split(w, &pp, mu_old, mu_old), ed_split(w, &pp, mu, mu_old), ed_split(w, &pp, mu_old, mu), ed_split(w, &pp, mu, mu));
            if ((po->variant & 1) && mu_old <= po->p[0]) {
                mu_s = mu_old;
                memcpy(gU_true, w->gU, sizeof(double) * UD * N);
                eval_frozen(w, &pp, mu, mu_s, &opt);
                /* (the adjoint sweep for the frozen q is not needed: riccati() takes q, r; gU only enters dphi) */
            }
        }
        const double tau = dmax(opt.tau_min, 1.0 - mu);
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < UD; ++i) {
                double sl = w->U[k][i] - lbu[i], su = ubu[i] - w->U[k][i];
                rbp[k][i] = w->r[k][i] - mu / sl + mu / su;
                Rbp[k][i] = w->Rd[k][i] + w->zl[k][i] / sl + w->zu[k][i] / su;
            }
        double delta = 0.0;
        int tries = 0;
        const int spec = prev_delta > 0.0;   /* census for a speculative double sweep (delta = 0 and delta_last / 3 in one pass) */
        while (!riccati(w, (const double(*)[UD])rbp, (const double(*)[UD])Rbp, delta)) {
            if (delta == 0.0) delta = (delta_last == 0.0) ? 1.0 : dmax(1e-20, delta_last / 3.0);
            else delta *= (delta_last == 0.0) ? 100.0 : 8.0;
            ++n_reg;
            ++tries;
            if (delta > 1e40) break;
        }
        prev_delta = delta;
        if (spec) { ++spec_n; if (tries == 0) ++spec_waste; else if (tries == 1) ++spec_hit; else ++spec_more; }
        else { ++nospec_n; nospec_fail += tries; }
        if (delta > 1e40) { status = 2; break; }
        if (delta > 0.0) delta_last = delta;
        double a_pr = 1.0, a_du = 1.0, dphi = 0.0;
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < UD; ++i) {
                double sl = w->U[k][i] - lbu[i], su = ubu[i] - w->U[k][i], du = w->dU[k][i];
                double dzl = mu / sl - w->zl[k][i] - (w->zl[k][i] / sl) * du;
                double dzu = mu / su - w->zu[k][i] + (w->zu[k][i] / su) * du;
                w->dzl[k][i] = dzl;
                w->dzu[k][i] = dzu;
                if (du < 0.0) a_pr = dmin(a_pr, -tau * sl / du);
                if (du > 0.0) a_pr = dmin(a_pr, tau * su / du);
                if (dzl < 0.0) a_du = dmin(a_du, -tau * w->zl[k][i] / dzl);
                if (dzu < 0.0) a_du = dmin(a_du, -tau * w->zu[k][i] / dzu);
                dphi += (w->gU[k][i] - mu / sl + mu / su) * du;
            }
        for (int k = 0; k + 1 < N; ++k) {
            const double *x = w->X[k + 1], *dx = w->dX[k + 1];
            const double p[3] = {x[0], x[1], x[2]}, v[3] = {x[4], x[5], x[6]};
            const double dp[3] = {dx[0], dx[1], dx[2]}, dv[3] = {dx[4], dx[5], dx[6]};
            for (int j = 0; j < K; ++j) {
                tgeo G;
                term_geo(p, v, pp.obs + 3 * (K * k + j), pp.radius, 1, &G);
                if (pp.lam * G.g > 0.0) {
                    if (mu_s > 0.0) term_step_f(&G, mu, mu_s, tau, opt.kappa_sigma, &w->y1[k][j], &w->y2[k][j], dp, dv);
                    else term_step(&G, mu, tau, opt.kappa_sigma, &w->y1[k][j], &w->y2[k][j], dp, dv);
                }
                else w->y1[k][j] = w->y2[k][j] = -1.0;
            }
        }
        const int tiny = -dphi <= 100.0 * DBL_EPSILON * (1.0 + fabs(phi0));
        double a = a_pr;
        int accepted = 0;
        for (int ls = 0; ls < opt.max_ls; ++ls) {
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < UD; ++i) w->Ut[k][i] = w->U[k][i] + a * w->dU[k][i];
            for (int k = 0; k <= N; ++k)
                for (int i = 0; i < SD; ++i) w->Xt[k][i] = w->X[k][i] + a * w->dX[k][i];
            double phi = total_merit(w, &pp, w->Xt, w->Ut, mu, opt.kappa_sigma, 0.0, 0, NULL);
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < UD; ++i) phi -= mu * slog((w->Ut[k][i] - lbu[i]) * (ubu[i] - w->Ut[k][i]));
            ++n_ls;
            if (tiny || phi <= phi0 + opt.eta_phi * a * dphi) { accepted = 1; break; }
            if (ls + 1 < opt.max_ls) a *= 0.5;
        }
        if (!accepted) { ++ls_fail; a = 0.0; }
        if (po->diag) printf(" | a %.4f (a_pr %.4f) a_du %.4f delta %.2e\n", a, a_pr, a_du, delta);
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < UD; ++i) w->U[k][i] += a * w->dU[k][i];
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < SD; ++i) w->X[k][i] += a * w->dX[k][i];
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < UD; ++i) {
                double zl = w->zl[k][i] + a_du * w->dzl[k][i], zu = w->zu[k][i] + a_du * w->dzu[k][i];
                double sl = w->U[k][i] - lbu[i], su = ubu[i] - w->U[k][i];
                zl = dmax(dmin(zl, opt.kappa_sigma * mu / sl), mu / (opt.kappa_sigma * sl));
                zu = dmax(dmin(zu, opt.kappa_sigma * mu / su), mu / (opt.kappa_sigma * su));
                w->zl[k][i] = zl;
                w->zu[k][i] = zu;
            }
        phi0 = eval_iterate_parts(w, &pp, lbu, ubu, mu, &opt, err, parts);
    }
    double J = 0.0;
    for (int k = 0; k < N; ++k) J += stage_cost(&pp, k, w->X[k + 1], w->U[k], NULL, NULL, NULL, NULL);
    for (int k = 0; k < N; ++k) {
        memcpy(XK(w_out, k), w->X[k], sizeof(double) * SD);
        memcpy(UK(w_out, k), w->U[k], sizeof(double) * UD);
    }
    memcpy(XK(w_out, N), w->X[N], sizeof(double) * SD);
    if (info) { info[0] = status; info[1] = it; info[2] = n_reg; info[3] = ls_fail; info[4] = n_ls; info[5] = spec_n; info[6] = spec_hit; info[7] = spec_waste; }
    if (stats) { (void)spec_more; (void)nospec_n; (void)nospec_fail; }
    if (stats) { stats[0] = J; stats[1] = err[0]; stats[2] = mu; stats[3] = delta_last; }
    free(w);
    return status;
}
