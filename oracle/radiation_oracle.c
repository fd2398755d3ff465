/*
 * radiation_oracle.c -- CPU restatement of the reference's radiation columns.   TEST INFRASTRUCTURE ONLY
 * (same rules as dycore_oracle.h: only tests/ may load it; the product path never does).
 *
 * Follows rad_main.py:121-205 (Radiation.simple_radiation), rad_longwave.py:36-137 + 140-210
 * (org_longwave with the pure-numpy helpers, run_cython = 0) and rad_shortwave.py:27-170
 * (org_shortwave / rad_calc_SW_fluxes_toon).  The reference solves its column systems with
 * scipy.linalg.solve_banded, i.e. LAPACK dgbsv for the (2,2)-banded longwave matrix and dgtsv for
 * the tridiagonal shortwave one (scipy and LAPACK are third-party, not part of /root/reference; scipy
 * 1.x calls ?gtsv when both bandwidths are 1, ?gbsv otherwise): both are restated here after the
 * published LAPACK algorithms (unblocked dgbtf2 + dgbtrs, dgtsv), partial pivoting included.
 *
 * Parity status: PINNED by tests/golden/g10x4_rad.npz and g4x8_rad.npz (oracle/gen_golden.py runs the
 * reference's Radiation.simple_radiation on a model state) to 1e-12 relative (BLAS-internal rounding
 * of the two solvers is the only difference).
 *
 * Layout: reference arrays [i][j][k] with one halo cell, shape (nx+2, ny+2, nk).
 */
#define _GNU_SOURCE      /* M_PI under -std=c11 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* io_constants.py:15-32, namelist.py:131-139 */
static const double CON_G = 9.81, CON_CP = 1005.0;
static const double CON_H = 6.6256E-34, CON_C = 2.9979E8, CON_KB = 1.38E-23;
static const double SIGMA_ABS_SW = 1.7E-5, SIGMA_SCA_SW = 1.72E-5;
static const double SIGMA_ABS_LW = 3.7E-4, SIGMA_SCA_LW = 1.72E-7;
static const double EMISSIVITY_SURFACE = 1.0;

/* rad_longwave.py:140-166: sum over the spectral bins of the Planck radiance */
static double planck_intensity(double temp, int nb, const double *lam, const double *dlam)
{
    double B = 0.0;
    for (int c = 0; c < nb; c++) {
        const double spectral = 2 * CON_H * (CON_C * CON_C) / pow(lam[c], 5) *
                                1 / (exp(CON_H * CON_C / (lam[c] * CON_KB * temp)) - 1);
        const double radiance = spectral * -dlam[c];
        B = B + radiance;
    }
    return B;
}

/* LAPACK dgbsv for one right-hand side: dgbtf2 (unblocked, chosen by dgbtrf whenever the block
 * size exceeds kl) followed by dgbtrs.  ab: (2*kl+ku+1) x n, column-major with leading dimension
 * ld = 2*kl+ku+1, matrix entry (i,j) at ab[(kl+ku+i-j) + j*ld]. */
static int gbsv(int n, int kl, int ku, double *ab, int *ipiv, double *b)
{
    const int ld = 2 * kl + ku + 1, kv = ku + kl;
#define AB(r, c) ab[(r) + (size_t)(c) * ld]
    for (int j = ku + 1; j < (kv < n ? kv : n); j++)          /* fill-in rows start at zero */
        for (int i = kv - j; i < kl; i++) AB(i, j) = 0.0;
    int ju = 0;
    for (int j = 0; j < n; j++) {
        if (j + kv < n)
            for (int i = 0; i < kl; i++) AB(i, j + kv) = 0.0;
        const int km = kl < n - 1 - j ? kl : n - 1 - j;
        int jp = 0;
        double best = fabs(AB(kv, j));
        for (int i = 1; i <= km; i++)
            if (fabs(AB(kv + i, j)) > best) { best = fabs(AB(kv + i, j)); jp = i; }
        ipiv[j] = jp + j;
        if (AB(kv + jp, j) == 0.0) return j + 1;
        const int cand = j + ku + jp < n - 1 ? j + ku + jp : n - 1;
        if (cand > ju) ju = cand;
        if (jp != 0)
            for (int c = j; c <= ju; c++) {
                const double t = AB(kv + jp - (c - j), c);
                AB(kv + jp - (c - j), c) = AB(kv - (c - j), c);
                AB(kv - (c - j), c) = t;
            }
        if (km > 0) {
            const double r = 1.0 / AB(kv, j);
            for (int i = 1; i <= km; i++) AB(kv + i, j) *= r;
            for (int c = j + 1; c <= ju; c++) {
                const double y = AB(kv - (c - j), c);
                if (y != 0.0)
                    for (int i = 1; i <= km; i++) AB(kv + i - (c - j), c) -= AB(kv + i, j) * y;
            }
        }
    }
    for (int j = 0; j < n - 1; j++) {                         /* L: interchanges + multipliers */
        const int lm = kl < n - 1 - j ? kl : n - 1 - j;
        const int l = ipiv[j];
        if (l != j) { const double t = b[l]; b[l] = b[j]; b[j] = t; }
        for (int i = 1; i <= lm; i++) b[j + i] -= b[j] * AB(kv + i, j);
    }
    for (int j = n - 1; j >= 0; j--) {                        /* U: dtbsv, upper, non-unit */
        if (b[j] != 0.0) {
            b[j] /= AB(kv, j);
            const double t = b[j];
            const int lo = j - kv > 0 ? j - kv : 0;
            for (int i = j - 1; i >= lo; i--) b[i] -= t * AB(kv - (j - i), j);
        }
    }
#undef AB
    return 0;
}

/* LAPACK dgtsv for one right-hand side */
static int gtsv(int n, double *dl, double *d, double *du, double *b)
{
    for (int i = 0; i < n - 1; i++) {
        if (fabs(d[i]) >= fabs(dl[i])) {
            if (d[i] == 0.0) return i + 1;
            const double fact = dl[i] / d[i];
            d[i + 1] -= fact * du[i];
            b[i + 1] -= fact * b[i];
            if (i < n - 2) dl[i] = 0.0;
        } else {
            const double fact = d[i] / dl[i];
            d[i] = dl[i];
            double temp = d[i + 1];
            d[i + 1] = du[i] - fact * temp;
            if (i < n - 2) {
                dl[i] = du[i + 1];
                du[i + 1] = -fact * dl[i];
            }
            du[i] = temp;
            temp = b[i];
            b[i] = b[i + 1];
            b[i + 1] = temp - fact * b[i + 1];
        }
    }
    if (d[n - 1] == 0.0) return n;
    b[n - 1] /= d[n - 1];
    if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / d[n - 2];
    for (int i = n - 3; i >= 0; i--) b[i] = (b[i] - du[i] * b[i + 1] - dl[i] * b[i + 2]) / d[i];
    return 0;
}

/* rad_longwave.py:36-137.  Outputs LWFLXDO = -down_diffuse, LWFLXUP = up_diffuse (rad_main.py:163-164), nz+1 values */
static void longwave_column(int nz, const double *dz, const double *tair, const double *rho, double tsurf,
                            double albedo_lw, const double *qc, int nb, const double *lam, const double *dlam,
                            double *lwdo, double *lwup, double *work, int *iwork)
{
    const int n = 2 * nz + 2;
    const double cloud_fact = 1E-2, g_a = 0.0, my1 = 1 / sqrt(3.0);
    double *ab = work, *g = work + 7 * n;
    memset(ab, 0, sizeof(double) * 7 * n);
#define A_(i, j) ab[(4 + (i) - (j)) + (size_t)(j) * 7]         /* kl + ku = 4 */
    g[0] = 0.0;
    A_(0, 0) = 1.0;
    A_(n - 1, n - 1) = 1.0;
    for (int k = 0; k < nz; k++) {
        const double q = qc[k] < 0.002 ? qc[k] : 0.002;
        const double s_abs = SIGMA_ABS_LW + q * cloud_fact, s_sca = SIGMA_SCA_LW;
        const double s_tot = s_abs + s_sca;
        const double dtau = s_tot * dz[k] * rho[k];
        double omega = s_sca / s_tot;
        if (isnan(omega)) omega = 0;
        const double gamma1 = (1 - omega * (1 + g_a) / 2) / my1;
        const double gamma2 = omega * (1 - g_a) / (2 * my1);
        const double B_air = 2 * M_PI * (1 - omega) * planck_intensity(tair[k], nb, lam, dlam);
        const double C1 = 0.5 * dtau * gamma1, C2 = 0.5 * dtau * gamma2;
        /* rad_longwave.py:169-210: the five diagonals, entry (row, column) */
        g[1 + 2 * k] = dtau * B_air;
        g[2 + 2 * k] = -(dtau * B_air);
        A_(1 + 2 * k, 1 + 2 * k) = 1 + C1;                      /* d0 */
        A_(2 + 2 * k, 2 + 2 * k) = -(1 + C1);
        A_(1 + 2 * k, 2 + 2 * k) = -C2;                         /* dp1[2+2k]: entry (j-1, j) */
        A_(2 + 2 * k, 3 + 2 * k) = -(-C2);
        A_(1 + 2 * k, 3 + 2 * k) = -(1 - C1);                   /* dp2[3+2k]: entry (j-2, j) */
        A_(1 + 2 * k, 2 * k) = -C2;                             /* dm1[2k]: entry (j+1, j) */
        A_(2 + 2 * k, 1 + 2 * k) = -(-C2);
        A_(2 + 2 * k, 2 * k) = 1 - C1;                          /* dm2[2k]: entry (j+2, j) */
    }
    A_(n - 1, n - 2) = -albedo_lw;                              /* dm1[-2] */
    g[n - 1] = EMISSIVITY_SURFACE * M_PI * planck_intensity(tsurf, nb, lam, dlam);
#undef A_
    gbsv(n, 2, 2, ab, iwork, g);
    for (int k = 0; k <= nz; k++) {
        lwdo[k] = -(-g[2 * k]);
        lwup[k] = g[2 * k + 1];
    }
}

/* rad_shortwave.py:27-170.  Outputs as stored by rad_main.py:187-190 */
static void shortwave_column(int nz, const double *dz, double solar_constant, const double *rho, double swintoa,
                             double mysun, double albedo_sw, double *swdifdo, double *swdirdo, double *swup,
                             double *swdo, double *work)
{
    const int n = 2 * nz;
    const double g_a = 0.0, my1 = 1 / sqrt(3.0);
    double *dtau = work, *taus = dtau + nz, *e1 = taus + nz + 1, *e2 = e1 + nz, *e3 = e2 + nz, *e4 = e3 + nz;
    double *cpt = e4 + nz, *cp0 = cpt + nz, *cmt = cp0 + nz, *cm0 = cmt + nz;
    double *dm1 = cm0 + nz, *d0 = dm1 + n, *dp1 = d0 + n, *src = dp1 + n;
    const double s_tot = SIGMA_ABS_SW + SIGMA_SCA_SW;
    double omega = SIGMA_SCA_SW / s_tot;
    if (isnan(omega)) omega = 0;
    const double gamma1 = (1 - omega * (1 + g_a) / 2) / my1;
    const double gamma2 = omega * (1 - g_a) / (2 * my1);
    const double gamma3 = (1 - 3 * g_a * my1 * mysun) / 2;
    const double gamma4 = 1 - gamma3;
    const double lamb = sqrt(gamma1 * gamma1 - gamma2 * gamma2);
    const double tau2 = gamma2 / (gamma1 + lamb);
    taus[0] = 0.0;
    for (int k = 0; k < nz; k++) {
        dtau[k] = s_tot * dz[k] * rho[k];
        taus[k + 1] = taus[k] + dtau[k];
    }
    const double surf_reflected = albedo_sw * mysun * solar_constant * exp(-taus[nz] / mysun);
    const double den = lamb * lamb - 1 / (mysun * mysun);
    for (int k = 0; k < nz; k++) {
        const double ex = exp(-lamb * dtau[k]);
        e1[k] = 1 + tau2 * ex;
        e2[k] = 1 - tau2 * ex;
        e3[k] = tau2 + ex;
        e4[k] = tau2 - ex;
        const double fp = ((gamma1 - 1 / mysun) * gamma3 + gamma4 * gamma2), fm = ((gamma1 + 1 / mysun) * gamma4 + gamma2 * gamma3);
        cpt[k] = omega * solar_constant * exp(-(taus[k + 1]) / mysun) * fp / den;
        cp0[k] = omega * solar_constant * exp(-(taus[k]) / mysun) * fp / den;
        cmt[k] = omega * solar_constant * exp(-(taus[k + 1]) / mysun) * fm / den;
        cm0[k] = omega * solar_constant * exp(-(taus[k]) / mysun) * fm / den;
    }
    for (int m = 0; m < nz - 1; m++) {                          /* lodd = 2m+2, leven = 2m+1 */
        const int lo = 2 * m + 2, le = 2 * m + 1;
        dm1[lo] = e2[m] * e3[m] - e4[m] * e1[m];
        dm1[le] = e2[m + 1] * e1[m] - e3[m] * e4[m + 1];
        d0[lo] = e1[m] * e1[m + 1] - e3[m] * e3[m + 1];
        d0[le] = e2[m] * e2[m + 1] - e4[m] * e4[m + 1];
        dp1[lo] = e3[m] * e4[m + 1] - e1[m] * e2[m + 1];
        dp1[le] = e1[m + 1] * e4[m + 1] - e2[m + 1] * e3[m + 1];
        src[lo] = e3[m] * (cp0[m + 1] - cpt[m]) + e1[m] * (cmt[m] - cm0[m + 1]);
        src[le] = e2[m + 1] * (cp0[m + 1] - cpt[m]) + e4[m + 1] * (cm0[m + 1] - cmt[m]);
    }
    dm1[0] = 0;
    dm1[n - 1] = e1[nz - 1] - albedo_sw * e3[nz - 1];
    d0[0] = e1[0];
    d0[n - 1] = e2[nz - 1] - albedo_sw * e4[nz - 1];
    dp1[0] = -e2[0];
    dp1[n - 1] = 0;
    src[0] = 0 - cm0[0];
    src[n - 1] = surf_reflected - cpt[nz - 1] + albedo_sw * cmt[nz - 1];
    /* A_mat rows: superdiagonal dp1[:-1], diagonal d0, subdiagonal dm1[1:] */
    gtsv(n, dm1 + 1, d0, dp1, src);
    double *down_diffuse = swdifdo, *up_diffuse = swup, *down_direct = swdirdo;   /* reused as temporaries */
    down_diffuse[0] = 0.0;
    for (int k = 0; k < nz; k++) {
        const double Y1 = src[2 * k], Y2 = src[2 * k + 1];
        down_diffuse[k + 1] = +(-Y1 * e3[k] - Y2 * e4[k] - cmt[k]);
        down_direct[k + 1] = -mysun * solar_constant * exp(-taus[k + 1] / mysun);
        up_diffuse[k + 1] = +(Y1 * e1[k] + Y2 * e2[k] + cpt[k]);
    }
    const double ext = up_diffuse[1] - (-up_diffuse[3] + 4 * up_diffuse[2] - 3 * up_diffuse[1]) * dz[0] / (dz[1] + dz[2]);
    up_diffuse[0] = 0 > ext ? 0 : ext;          /* Python max(0, x): 0 unless x is greater */
    if (!(ext > 0)) up_diffuse[0] = 0;
    down_direct[0] = -swintoa;
    for (int k = 0; k <= nz; k++) {             /* rad_main.py:187-190 */
        const double dd = down_diffuse[k], dr = down_direct[k];
        swdifdo[k] = -dd;
        swdirdo[k] = -dr;
        swdo[k] = -dd - dr;
    }
}

/* rad_main.py:121-205 for the interior columns; halo columns of the outputs are left untouched.
 * All 3-D arrays (nx+2, ny+2, nk); SOILTEMP, albedos, MYSUN, SWINTOA (nx+2, ny+2, 1). */
void oracle_radiation(int nx, int ny, int nz, const double *PHIVB, const double *TAIR, const double *RHO,
                      const double *SOILTEMP, const double *ALB_LW, const double *ALB_SW, const double *QC,
                      const double *MYSUN, const double *SWINTOA, double solar_constant,
                      int nb, const double *lam, const double *dlam,
                      double *LWFLXDO, double *LWFLXUP, double *SWDIFFLXDO, double *SWDIRFLXDO, double *SWFLXUP,
                      double *SWFLXDO, double *LWFLXNET, double *SWFLXNET, double *SWFLXDIV, double *LWFLXDIV,
                      double *TOTFLXDIV, double *dPOTTdt_RAD)
{
    const int nzs = nz + 1, n = 2 * nz + 2;
#pragma omp parallel
    {
        double *work = malloc(sizeof(double) * (8 * (size_t)n + 16 * (size_t)nzs + 64));
        int *iwork = malloc(sizeof(int) * n);
        double *dz = malloc(sizeof(double) * nz);
#pragma omp for collapse(2)
        for (int i = 1; i <= nx; i++)
            for (int j = 1; j <= ny; j++) {
                const size_t c2 = (size_t)i * (ny + 2) + j;
                const size_t c = c2 * nz, cs = c2 * nzs;
                for (int k = 0; k < nz; k++) dz[k] = PHIVB[cs + k] / CON_G - PHIVB[cs + k + 1] / CON_G;
                longwave_column(nz, dz, TAIR + c, RHO + c, SOILTEMP[c2], ALB_LW[c2], QC + c, nb, lam, dlam,
                                LWFLXDO + cs, LWFLXUP + cs, work, iwork);
                if (MYSUN[c2] > 0)
                    shortwave_column(nz, dz, solar_constant, RHO + c, SWINTOA[c2], MYSUN[c2], ALB_SW[c2],
                                     SWDIFFLXDO + cs, SWDIRFLXDO + cs, SWFLXUP + cs, SWFLXDO + cs, work);
                else
                    for (int k = 0; k < nzs; k++) SWDIFFLXDO[cs + k] = SWDIRFLXDO[cs + k] = SWFLXUP[cs + k] = SWFLXDO[cs + k] = 0;
                for (int k = 0; k < nzs; k++) {
                    LWFLXNET[cs + k] = LWFLXDO[cs + k] - LWFLXUP[cs + k];
                    SWFLXNET[cs + k] = SWFLXDO[cs + k] - SWFLXUP[cs + k];
                }
                for (int k = 0; k < nz; k++) {
                    SWFLXDIV[c + k] = (SWFLXNET[cs + k] - SWFLXNET[cs + k + 1]) / dz[k];
                    LWFLXDIV[c + k] = (LWFLXNET[cs + k] - LWFLXNET[cs + k + 1]) / dz[k];
                    TOTFLXDIV[c + k] = SWFLXDIV[c + k] + LWFLXDIV[c + k];
                    dPOTTdt_RAD[c + k] = 1 / (CON_CP * RHO[c + k]) * TOTFLXDIV[c + k];
                }
            }
        free(work); free(iwork); free(dz);
    }
}
