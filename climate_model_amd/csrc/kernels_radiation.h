// kernels_radiation.h -- radiation columns of the reference (rad_main.py:121-205) on the device.
//
// One thread = one (i, j) column: longwave two-stream system (rad_longwave.py:36-137, 169-210:
// 2nz+2 unknowns, (2,2)-banded, solved like LAPACK dgbsv with partial pivoting, which is what the
// reference's scipy.linalg.solve_banded calls) and the Toon et al. shortwave system
// (rad_shortwave.py:27-170: 2nz unknowns, tridiagonal, dgtsv) for sunlit columns, then net fluxes,
// flux divergences and the heating rate dPOTTdt_RAD that the temperature tendency consumes.
//
// The band matrix and right-hand side of a column (18 nz + 18 doubles) live in a global workspace
// laid out [element][column]: lane l of a wavefront owns column base + l, so every access to
// element e is one coalesced 512-B row.  The scheme runs every few simulated hours
// (rad_nth_hour); it is written for clarity and exactness, not tuned.
#pragma once

#include "dycore_common.h"

namespace dyc {

constexpr int RAD_MAX_BINS = 16;
struct RadParams {
    double solar_constant;
    int nbins;
    double lam[RAD_MAX_BINS], dlam[RAD_MAX_BINS];      // planck_lambdas_center, planck_dlambdas
};
// io_constants.py:30-32, namelist.py:131-139
constexpr double RAD_H = 6.6256E-34, RAD_C = 2.9979E8, RAD_KB = 1.38E-23;
constexpr double RAD_ABS_SW = 1.7E-5, RAD_SCA_SW = 1.72E-5, RAD_ABS_LW = 3.7E-4, RAD_SCA_LW = 1.72E-7;
constexpr double RAD_EMISSIVITY_SURFACE = 1.0;
constexpr double RAD_PI = 3.141592653589793;

__host__ __device__ inline size_t rad_ws_per_column(int nz) { return 18 * (size_t)nz + 18; }

// rad_longwave.py:140-166
__device__ __forceinline__ double rad_planck(double temp, const RadParams &rp)
{
    double B = 0.0;
    for (int c = 0; c < rp.nbins; c++) {
        const double spectral = 2 * RAD_H * (RAD_C * RAD_C) / pow(rp.lam[c], 5.0) *
                                1 / (exp(RAD_H * RAD_C / (rp.lam[c] * RAD_KB * temp)) - 1);
        B = B + spectral * -rp.dlam[c];
    }
    return B;
}

// W(e): element e of this thread's workspace column
#define W(e) ws[(size_t)(e) * stride]

// LAPACK dgbsv (dgbtf2 + dgbtrs, one right-hand side) for kl = ku = 2 on the band matrix stored at
// W(0 .. 7n-1) (column-major, leading dimension 7, entry (i,j) at 4+i-j + 7j) and b at W(7n ..).
// The row interchanges and multipliers are applied to b as they are produced (same operations in
// the same order as dgbtrs' forward sweep), so neither pivots nor L are kept.
__device__ inline void rad_gbsv22(int n, double *ws, size_t stride)
{
    constexpr int kl = 2, ku = 2, ld = 7, kv = 4;
#define AB(r, c) W((r) + (c) * ld)
#define BV(i) W(7 * n + (i))
    int ju = 0;
    for (int j = 0; j < n; j++) {
        const int km = kl < n - 1 - j ? kl : n - 1 - j;
        int jp = 0;
        double piv = AB(kv, j), best = fabs(piv);
        for (int i = 1; i <= km; i++) {
            const double v = AB(kv + i, j);
            if (fabs(v) > best) { best = fabs(v); jp = i; piv = v; }
        }
        if (piv == 0.0) return;                              // singular: leaves NaN-free garbage like an unchecked gbsv
        const int cand = j + ku + jp < n - 1 ? j + ku + jp : n - 1;
        if (cand > ju) ju = cand;
        if (jp != 0) {
            for (int c = j; c <= ju; c++) {
                const double t = AB(kv + jp - (c - j), c);
                AB(kv + jp - (c - j), c) = AB(kv - (c - j), c);
                AB(kv - (c - j), c) = t;
            }
            const double t = BV(j + jp); BV(j + jp) = BV(j); BV(j) = t;
        }
        if (km > 0) {
            const double r = 1.0 / piv;
            double m[kl];
            for (int i = 1; i <= km; i++) { m[i - 1] = AB(kv + i, j) * r; }
            for (int c = j + 1; c <= ju; c++) {
                const double y = AB(kv - (c - j), c);
                if (y != 0.0)
                    for (int i = 1; i <= km; i++) AB(kv + i - (c - j), c) -= m[i - 1] * y;
            }
            const double bj = BV(j);
            for (int i = 1; i <= km; i++) BV(j + i) -= bj * m[i - 1];
        }
    }
    for (int j = n - 1; j >= 0; j--) {                        // dtbsv: upper, non-unit
        double bj = BV(j);
        if (bj != 0.0) {
            bj /= AB(kv, j);
            BV(j) = bj;
            const int lo = j - kv > 0 ? j - kv : 0;
            for (int i = j - 1; i >= lo; i--) BV(i) -= bj * AB(kv - (j - i), j);
        }
    }
#undef AB
#undef BV
}

// LAPACK dgtsv (one right-hand side): dl, d, du, b are workspace offsets
__device__ inline void rad_gtsv(int n, double *ws, size_t stride, int o_dl, int o_d, int o_du, int o_b)
{
#define DL(i) W(o_dl + (i))
#define D(i) W(o_d + (i))
#define DU(i) W(o_du + (i))
#define B(i) W(o_b + (i))
    for (int i = 0; i < n - 1; i++) {
        const double di = D(i), dli = DL(i);
        if (fabs(di) >= fabs(dli)) {
            if (di == 0.0) return;
            const double fact = dli / di;
            D(i + 1) -= fact * DU(i);
            B(i + 1) -= fact * B(i);
            if (i < n - 2) DL(i) = 0.0;
        } else {
            const double fact = di / dli;
            D(i) = dli;
            double temp = D(i + 1);
            D(i + 1) = DU(i) - fact * temp;
            if (i < n - 2) {
                const double dun = DU(i + 1);
                DL(i) = dun;
                DU(i + 1) = -fact * dun;
            }
            DU(i) = temp;
            temp = B(i);
            const double bn = B(i + 1);
            B(i) = bn;
            B(i + 1) = temp - fact * bn;
        }
    }
    if (D(n - 1) == 0.0) return;
    B(n - 1) /= D(n - 1);
    if (n > 1) B(n - 2) = (B(n - 2) - DU(n - 2) * B(n - 1)) / D(n - 2);
    for (int i = n - 3; i >= 0; i--) B(i) = (B(i) - DU(i) * B(i + 1) - DL(i) * B(i + 2)) / D(i);
#undef DL
#undef D
#undef DU
#undef B
}

// columns [col0, col0 + ncols) of the row-major owned-cell order; ws holds ncols_ws >= ncols columns
__global__ __launch_bounds__(256) void k_radiation(Geo g, Ptrs p, RadParams rp, double *wsbase, long long stride,
                                                   long long col0, long long ncols)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncols) return;
    const long long col = col0 + t;
    const int j = (int)(col / g.nx) + 1, i = (int)(col - (long long)(j - 1) * g.nx) + 1;
    const int nz = g.nz, nzs = nz + 1;
    const size_t c = at2(g, i, j);
    const size_t pl = (size_t)g.plane;
    double *ws = wsbase + t;
#define DZ(k) (p.PHIVB[c + (size_t)(k) * pl] / CON_G - p.PHIVB[c + (size_t)((k) + 1) * pl] / CON_G)

    // ---------------- longwave: rad_longwave.py:36-137 ----------------
    {
        const int n = 2 * nz + 2;
        const double cloud_fact = 1E-2, g_a = 0.0, my1 = 1 / sqrt(3.0);
        for (int e = 0; e < 7 * n; e++) W(e) = 0.0;
#define A_(r, cc) W((4 + (r) - (cc)) + (cc) * 7)
#define G_(r) W(7 * n + (r))
        G_(0) = 0.0;
        A_(0, 0) = 1.0;
        A_(n - 1, n - 1) = 1.0;
        for (int k = 0; k < nz; k++) {
            const size_t q = c + (size_t)k * pl;
            const double qc = p.QC[q];
            const double qcl = qc < 0.002 ? qc : 0.002;
            const double s_abs = RAD_ABS_LW + qcl * cloud_fact, s_sca = RAD_SCA_LW;
            const double s_tot = s_abs + s_sca;
            const double dtau = s_tot * DZ(k) * p.RHO[q];
            double omega = s_sca / s_tot;
            if (isnan(omega)) omega = 0;
            const double gamma1 = (1 - omega * (1 + g_a) / 2) / my1;
            const double gamma2 = omega * (1 - g_a) / (2 * my1);
            const double B_air = 2 * RAD_PI * (1 - omega) * rad_planck(p.TAIR[q], rp);
            const double C1 = 0.5 * dtau * gamma1, C2 = 0.5 * dtau * gamma2;
            G_(1 + 2 * k) = dtau * B_air;
            G_(2 + 2 * k) = -(dtau * B_air);
            A_(1 + 2 * k, 1 + 2 * k) = 1 + C1;
            A_(2 + 2 * k, 2 + 2 * k) = -(1 + C1);
            A_(1 + 2 * k, 2 + 2 * k) = -C2;
            A_(2 + 2 * k, 3 + 2 * k) = C2;
            A_(1 + 2 * k, 3 + 2 * k) = -(1 - C1);
            A_(1 + 2 * k, 2 * k) = -C2;
            A_(2 + 2 * k, 1 + 2 * k) = C2;
            A_(2 + 2 * k, 2 * k) = 1 - C1;
        }
        A_(n - 1, n - 2) = -p.SURFALBEDLW[c];
        G_(n - 1) = RAD_EMISSIVITY_SURFACE * RAD_PI * rad_planck(p.SOILTEMP[c], rp);
        rad_gbsv22(n, ws, (size_t)stride);
        for (int k = 0; k < nzs; k++) {                         // rad_main.py:163-166
            const size_t q = c + (size_t)k * pl;
            const double dn = G_(2 * k), up = G_(2 * k + 1);
            p.LWFLXDO[q] = dn;
            p.LWFLXUP[q] = up;
            p.LWFLXNET[q] = dn - up;
        }
#undef A_
#undef G_
    }

    // ---------------- shortwave: rad_shortwave.py:27-170 ----------------
    const double mysun = p.MYSUN[c];
    if (mysun > 0) {
        const int n = 2 * nz;
        const int o_dtau = 0, o_taus = nz, o_e1 = 2 * nz + 1, o_e2 = o_e1 + nz, o_e3 = o_e2 + nz, o_e4 = o_e3 + nz;
        const int o_cpt = o_e4 + nz, o_cp0 = o_cpt + nz, o_cmt = o_cp0 + nz, o_cm0 = o_cmt + nz;
        const int o_dm1 = o_cm0 + nz, o_d0 = o_dm1 + n, o_dp1 = o_d0 + n, o_src = o_dp1 + n;
        const double S = rp.solar_constant, albedo = p.SURFALBEDSW[c];
        const double g_a = 0.0, my1 = 1 / sqrt(3.0);
        const double s_tot = RAD_ABS_SW + RAD_SCA_SW;
        double omega = RAD_SCA_SW / s_tot;
        if (isnan(omega)) omega = 0;
        const double gamma1 = (1 - omega * (1 + g_a) / 2) / my1;
        const double gamma2 = omega * (1 - g_a) / (2 * my1);
        const double gamma3 = (1 - 3 * g_a * my1 * mysun) / 2;
        const double gamma4 = 1 - gamma3;
        const double lamb = sqrt(gamma1 * gamma1 - gamma2 * gamma2);
        const double tau2 = gamma2 / (gamma1 + lamb);
        double taus = 0.0;
        W(o_taus) = 0.0;
        for (int k = 0; k < nz; k++) {
            const double dtau = s_tot * DZ(k) * p.RHO[c + (size_t)k * pl];
            W(o_dtau + k) = dtau;
            taus = taus + dtau;
            W(o_taus + k + 1) = taus;
        }
        const double surf_reflected = albedo * mysun * S * exp(-taus / mysun);
        const double den = lamb * lamb - 1 / (mysun * mysun);
        const double fp = ((gamma1 - 1 / mysun) * gamma3 + gamma4 * gamma2), fm = ((gamma1 + 1 / mysun) * gamma4 + gamma2 * gamma3);
        for (int k = 0; k < nz; k++) {
            const double ex = exp(-lamb * W(o_dtau + k));
            W(o_e1 + k) = 1 + tau2 * ex;
            W(o_e2 + k) = 1 - tau2 * ex;
            W(o_e3 + k) = tau2 + ex;
            W(o_e4 + k) = tau2 - ex;
            const double x1 = exp(-(W(o_taus + k + 1)) / mysun), x0 = exp(-(W(o_taus + k)) / mysun);
            W(o_cpt + k) = omega * S * x1 * fp / den;
            W(o_cp0 + k) = omega * S * x0 * fp / den;
            W(o_cmt + k) = omega * S * x1 * fm / den;
            W(o_cm0 + k) = omega * S * x0 * fm / den;
        }
#define E1(k) W(o_e1 + (k))
#define E2(k) W(o_e2 + (k))
#define E3(k) W(o_e3 + (k))
#define E4(k) W(o_e4 + (k))
        for (int m = 0; m < nz - 1; m++) {
            const int lo = 2 * m + 2, le = 2 * m + 1;
            const double e1 = E1(m), e2 = E2(m), e3 = E3(m), e4 = E4(m);
            const double f1 = E1(m + 1), f2 = E2(m + 1), f3 = E3(m + 1), f4 = E4(m + 1);
            const double cp0n = W(o_cp0 + m + 1), cptm = W(o_cpt + m), cmtm = W(o_cmt + m), cm0n = W(o_cm0 + m + 1);
            W(o_dm1 + lo) = e2 * e3 - e4 * e1;
            W(o_dm1 + le) = f2 * e1 - e3 * f4;
            W(o_d0 + lo) = e1 * f1 - e3 * f3;
            W(o_d0 + le) = e2 * f2 - e4 * f4;
            W(o_dp1 + lo) = e3 * f4 - e1 * f2;
            W(o_dp1 + le) = f1 * f4 - f2 * f3;
            W(o_src + lo) = e3 * (cp0n - cptm) + e1 * (cmtm - cm0n);
            W(o_src + le) = f2 * (cp0n - cptm) + f4 * (cm0n - cmtm);
        }
        W(o_dm1) = 0;
        W(o_dm1 + n - 1) = E1(nz - 1) - albedo * E3(nz - 1);
        W(o_d0) = E1(0);
        W(o_d0 + n - 1) = E2(nz - 1) - albedo * E4(nz - 1);
        W(o_dp1) = -E2(0);
        W(o_dp1 + n - 1) = 0;
        W(o_src) = 0 - W(o_cm0);
        W(o_src + n - 1) = surf_reflected - W(o_cpt + nz - 1) + albedo * W(o_cmt + nz - 1);
        rad_gtsv(n, ws, (size_t)stride, o_dm1 + 1, o_d0, o_dp1, o_src);
        double up1 = 0.0, up2 = 0.0, up3 = 0.0;
        for (int k = 0; k < nz; k++) {
            const double Y1 = W(o_src + 2 * k), Y2 = W(o_src + 2 * k + 1);
            const double down_diffuse = +(-Y1 * E3(k) - Y2 * E4(k) - W(o_cmt + k));
            const double down_direct = -mysun * S * exp(-W(o_taus + k + 1) / mysun);
            const double up_diffuse = +(Y1 * E1(k) + Y2 * E2(k) + W(o_cpt + k));
            if (k == 0) up1 = up_diffuse;
            if (k == 1) up2 = up_diffuse;
            if (k == 2) up3 = up_diffuse;
            const size_t q = c + (size_t)(k + 1) * pl;
            const double dif = -down_diffuse, dir = -down_direct, dn = -down_diffuse - down_direct;
            p.SWDIFFLXDO[q] = dif;
            p.SWDIRFLXDO[q] = dir;
            p.SWFLXUP[q] = up_diffuse;
            p.SWFLXDO[q] = dn;
            p.SWFLXNET[q] = dn - up_diffuse;
        }
#undef E1
#undef E2
#undef E3
#undef E4
        // top interface: rad_shortwave.py:96-101
        const double ext = up1 - (-up3 + 4 * up2 - 3 * up1) * DZ(0) / (DZ(1) + DZ(2));
        const double up0 = ext > 0 ? ext : 0.0;
        const double dir0 = -(-p.SWINTOA[c]);
        p.SWDIFFLXDO[c] = -0.0;
        p.SWDIRFLXDO[c] = dir0;
        p.SWFLXUP[c] = up0;
        p.SWFLXDO[c] = -0.0 - (-p.SWINTOA[c]);
        p.SWFLXNET[c] = p.SWFLXDO[c] - up0;
    } else {
        for (int k = 0; k < nzs; k++) {
            const size_t q = c + (size_t)k * pl;
            p.SWDIFFLXDO[q] = 0; p.SWDIRFLXDO[q] = 0; p.SWFLXUP[q] = 0; p.SWFLXDO[q] = 0;
            p.SWFLXNET[q] = 0;
        }
    }

    // ---------------- flux divergences and heating rate: rad_main.py:196-204 ----------------
    double sw_k = p.SWFLXNET[c], lw_k = p.LWFLXNET[c];
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * pl;
        const double sw_kp1 = p.SWFLXNET[q + pl], lw_kp1 = p.LWFLXNET[q + pl];
        const double dz = DZ(k);
        const double swdiv = (sw_k - sw_kp1) / dz, lwdiv = (lw_k - lw_kp1) / dz;
        const double tot = swdiv + lwdiv;
        p.SWFLXDIV[q] = swdiv;
        p.LWFLXDIV[q] = lwdiv;
        p.TOTFLXDIV[q] = tot;
        p.dPOTTdt_RAD[q] = 1 / (CON_CP * p.RHO[q]) * tot;
        sw_k = sw_kp1; lw_k = lw_kp1;
    }
#undef DZ
}
#undef W

}  // namespace dyc
