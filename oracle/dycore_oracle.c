/*
 * dycore_oracle.c -- CPU restatement of the reference's Matsuno dycore step.
 * TEST INFRASTRUCTURE ONLY (see dycore_oracle.h).  Plain C + OpenMP, fp64.
 *
 * Structured like the reference's CPU kernels on purpose (it doubles as the
 * "restated CPU baseline" timed by bench.py): [i][j][k] k-contiguous arrays,
 * one function per reference kernel, `omp parallel for` over i where the
 * reference uses njit(parallel=True)+prange(i), serial where it does not
 * (dyn_timestep.py:297).
 *
 * Deliberate differences from the reference (none changes a value that
 * survives the following exchange_BC):
 *   - loads the reference issues out of bounds / from never-written halos and
 *     then discards are not issued; the cells they would have
 *     produced (dUFLXdt column nx+1, dVFLXdt rows 1 and ny+1, UWIND/VWIND/POTT
 *     halo cells in make_timestep) are left untouched here;
 *   - dCOLPdt is summed sequentially in k (numba order), not numpy-pairwise.
 */
#include "dycore_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* io_constants.py:15-20 */
#define CON_G 9.81
#define CON_RE 6371000.0
#define CON_RD 287.058
#define CON_CP 1005.0
#define CON_KAPPA (CON_RD / CON_CP)
#define CON_LH 2264E3   /* io_constants.py:24 */

#define AT(F, i, j, k, stgy, nk) \
    (F)[((size_t)(i) * (size_t)(ny + 2 + (stgy)) + (size_t)(j)) * (size_t)(nk) + (size_t)(k)]
/* 2-D (.., .., 1) fields */
#define A2(F, i, j) AT(F, i, j, 0, 0, 1)
#define A2S(F, i, j) AT(F, i, j, 0, 1, 1)

#ifdef _OPENMP
#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(nthr)")
static int nthreads_of(const oracle_cfg *c)
{
    return c->nthreads > 0 ? c->nthreads : omp_get_max_threads();
}
#else
#define PAR_FOR
static int nthreads_of(const oracle_cfg *c) { (void)c; return 1; }
#endif

/* ------------------------------------------------------------------------- */
/* misc_boundaries.py:22-42 */
void oracle_exchange_bc(const oracle_cfg *c, real *F, int stgx, int stgy, int nk)
{
    const int nx = c->nx, ny = c->ny;
    const int nxs = nx + 1, nys = ny + 1;
    const int fny = ny + 2 + stgy;
    const size_t col = (size_t)fny * nk;
    if (stgx) {
        memcpy(F, F + (size_t)(nxs - 1) * col, col * sizeof(real));
        memcpy(F + (size_t)nxs * col, F + 1 * col, col * sizeof(real));
        memcpy(F + (size_t)(nxs + 1) * col, F + 2 * col, col * sizeof(real));
    } else {
        memcpy(F, F + (size_t)nx * col, col * sizeof(real));
        memcpy(F + (size_t)(nx + 1) * col, F + 1 * col, col * sizeof(real));
    }
    const int fnx = nx + 2 + stgx;
    for (int i = 0; i < fnx; i++) {
        real *Fi = F + (size_t)i * col;
        if (stgy) {
            const int rows[4] = {0, 1, nys, nys + 1};
            for (int r = 0; r < 4; r++)
                for (int k = 0; k < nk; k++) Fi[(size_t)rows[r] * nk + k] = 0.0;
        } else {
            for (int k = 0; k < nk; k++) {
                Fi[k] = Fi[(size_t)1 * nk + k];
                Fi[(size_t)(ny + 1) * nk + k] = Fi[(size_t)ny * nk + k];
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* dyn_continuity.py:40-47 (formulae), :170-232 (loops) */
#define TERM_ON(c, bit) (((c)->term_off & (bit)) == 0)
void oracle_continuity(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    PAR_FOR
    for (int i = 1; i < nx + 1; i++) {
        for (int j = 1; j < ny + 1; j++) {
            const real colp = A2(f->COLP, i, j);
            const real c_im1 = A2(f->COLP, i - 1, j), c_ip1 = A2(f->COLP, i + 1, j);
            const real c_jm1 = A2(f->COLP, i, j - 1), c_jp1 = A2(f->COLP, i, j + 1);
            const real dy_i = AT(g->dyis, i, j, 0, 0, 1), dy_ip1 = AT(g->dyis, i + 1, j, 0, 0, 1);
            const real dx_j = A2S(g->dxjs, i, j), dx_jp1 = A2S(g->dxjs, i, j + 1);
            const real area = A2(g->A, i, j);
            real sum = 0.0;
            for (int k = 0; k < nz; k++) {
                const real uflx_i = (c_im1 + colp) / 2.0 * AT(f->UWIND, i, j, k, 0, nz) * dy_i;
                const real uflx_ip1 = (colp + c_ip1) / 2.0 * AT(f->UWIND, i + 1, j, k, 0, nz) * dy_ip1;
                const real vflx_j = (c_jm1 + colp) / 2.0 * AT(f->VWIND, i, j, k, 1, nz) * dx_j;
                const real vflx_jp1 = (colp + c_jp1) / 2.0 * AT(f->VWIND, i, j + 1, k, 1, nz) * dx_jp1;
                AT(f->UFLX, i, j, k, 0, nz) = uflx_i;
                AT(f->VFLX, i, j, k, 1, nz) = vflx_j;
                const real div = (uflx_ip1 - uflx_i + vflx_jp1 - vflx_j) * g->dsigma[k] / area;
                AT(f->FLXDIV, i, j, k, 0, nz) = div;
                sum += div;
            }
            const real dcolpdt = TERM_ON(c, ORACLE_TERM_COLP_MAIN) ? -sum : (real)0.0;   /* :211-214 */
            const real colp_new = A2(f->COLP_OLD, i, j) + dt * dcolpdt; /* :218 */
            A2(f->dCOLPdt, i, j) = dcolpdt;
            A2(f->COLP_NEW, i, j) = colp_new;
            real run = AT(f->FLXDIV, i, j, 0, 0, nz);                /* :222-229 */
            for (int k = 1; k < nz; k++) {
                AT(f->WWIND, i, j, k, 0, nzs) =
                    -run / colp_new - g->sigma_vb[k] * dcolpdt / colp_new;
                run += AT(f->FLXDIV, i, j, k, 0, nz);
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* dyn_UVFLX_prepare.py:249-439 */
void oracle_uvflx_prepare(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    /* ---- WWIND_UWIND (:262-280), rigid wall at j = 1 and j = ny ---- */
    const int vadv = TERM_ON(c, ORACLE_TERM_UVFLX_VERT_ADV), hadv = TERM_ON(c, ORACLE_TERM_UVFLX_HOR_ADV);   /* :260, :356 */
    PAR_FOR
    for (int i = 1; i < (vadv ? nx + 2 : 0); i++)
        for (int j = 1; j < ny + 1; j++) {
            AT(f->WWIND_UWIND, i, j, 0, 0, nzs) = 0.0;
            AT(f->WWIND_UWIND, i, j, nz, 0, nzs) = 0.0;
            const int lo = (j == 1), hi = (j == ny);
            for (int k = 1; k < nz; k++) {
                /* products are formed as COLP_NEW*A*WWIND, doubled terms as 2*COLP_NEW*A*WWIND */
                const real p_pp1_dm1 = A2(f->COLP_NEW, i - 1, j + 1) * A2(g->A, i - 1, j + 1) * AT(f->WWIND, i - 1, j + 1, k, 0, nzs);
                const real p_pp1 = A2(f->COLP_NEW, i, j + 1) * A2(g->A, i, j + 1) * AT(f->WWIND, i, j + 1, k, 0, nzs);
                const real p_pm1_dm1 = A2(f->COLP_NEW, i - 1, j - 1) * A2(g->A, i - 1, j - 1) * AT(f->WWIND, i - 1, j - 1, k, 0, nzs);
                const real p_pm1 = A2(f->COLP_NEW, i, j - 1) * A2(g->A, i, j - 1) * AT(f->WWIND, i, j - 1, k, 0, nzs);
                real s;
                if (lo || hi) {
                    const real p_dm1 = A2(f->COLP_NEW, i - 1, j) * A2(g->A, i - 1, j) * AT(f->WWIND, i - 1, j, k, 0, nzs);
                    const real p_0 = A2(f->COLP_NEW, i, j) * A2(g->A, i, j) * AT(f->WWIND, i, j, k, 0, nzs);
                    s = lo ? 0.25 * (p_pp1_dm1 + p_pp1 + p_dm1 + p_0)
                           : 0.25 * (p_dm1 + p_0 + p_pm1_dm1 + p_pm1);
                } else {
                    const real d_dm1 = 2.0 * A2(f->COLP_NEW, i - 1, j) * A2(g->A, i - 1, j) * AT(f->WWIND, i - 1, j, k, 0, nzs);
                    const real d_0 = 2.0 * A2(f->COLP_NEW, i, j) * A2(g->A, i, j) * AT(f->WWIND, i, j, k, 0, nzs);
                    s = 0.125 * (p_pp1_dm1 + p_pp1 + d_dm1 + d_0 + p_pm1_dm1 + p_pm1);
                }
                const real u_ks = (g->dsigma[k] * AT(f->UWIND, i, j, k - 1, 0, nz) +
                                     g->dsigma[k - 1] * AT(f->UWIND, i, j, k, 0, nz)) /
                                    (g->dsigma[k] + g->dsigma[k - 1]);
                AT(f->WWIND_UWIND, i, j, k, 0, nzs) = s * u_ks;
            }
        }
    /* ---- WWIND_VWIND (:282-300), no wall variant ---- */
    PAR_FOR
    for (int i = 1; i < (vadv ? nx + 1 : 0); i++)
        for (int j = 1; j < ny + 2; j++) {
            AT(f->WWIND_VWIND, i, j, 0, 1, nzs) = 0.0;
            AT(f->WWIND_VWIND, i, j, nz, 1, nzs) = 0.0;
            for (int k = 1; k < nz; k++) {
                const real p_pp1_dm1 = A2(f->COLP_NEW, i + 1, j - 1) * A2(g->A, i + 1, j - 1) * AT(f->WWIND, i + 1, j - 1, k, 0, nzs);
                const real p_pp1 = A2(f->COLP_NEW, i + 1, j) * A2(g->A, i + 1, j) * AT(f->WWIND, i + 1, j, k, 0, nzs);
                const real d_dm1 = 2.0 * A2(f->COLP_NEW, i, j - 1) * A2(g->A, i, j - 1) * AT(f->WWIND, i, j - 1, k, 0, nzs);
                const real d_0 = 2.0 * A2(f->COLP_NEW, i, j) * A2(g->A, i, j) * AT(f->WWIND, i, j, k, 0, nzs);
                const real p_pm1_dm1 = A2(f->COLP_NEW, i - 1, j - 1) * A2(g->A, i - 1, j - 1) * AT(f->WWIND, i - 1, j - 1, k, 0, nzs);
                const real p_pm1 = A2(f->COLP_NEW, i - 1, j) * A2(g->A, i - 1, j) * AT(f->WWIND, i - 1, j, k, 0, nzs);
                const real s = 0.125 * (p_pp1_dm1 + p_pp1 + d_dm1 + d_0 + p_pm1_dm1 + p_pm1);
                const real v_ks = (g->dsigma[k] * AT(f->VWIND, i, j, k - 1, 1, nz) +
                                     g->dsigma[k - 1] * AT(f->VWIND, i, j, k, 1, nz)) /
                                    (g->dsigma[k] + g->dsigma[k - 1]);
                AT(f->WWIND_VWIND, i, j, k, 1, nzs) = s * v_ks;
            }
        }
    /* ---- KMOM_dUWINDdz / KMOM_dVWINDdz (:303-352; dyn_functions.py:276-377) ---- */
    if (c->i_turb) {
#define RK(ii, jj) (AT(f->RHOVB, ii, jj, k, 0, nzs) * A2(f->COLP, ii, jj) * A2(g->A, ii, jj) * AT(f->KMOM, ii, jj, k, 0, nzs))
#define RK2(ii, jj) (2.0 * AT(f->RHOVB, ii, jj, k, 0, nzs) * A2(f->COLP, ii, jj) * A2(g->A, ii, jj) * AT(f->KMOM, ii, jj, k, 0, nzs))
#define PH(ii, jj, kk) AT(f->PHI, ii, jj, kk, 0, nz)
        PAR_FOR
        for (int i = 1; i < nx + 2; i++)
            for (int j = 1; j < ny + 1; j++) {
                /* k = 0 and k = nz: the reference returns an unassigned local
                 * (dyn_functions.py:310 vs :377); never consumed -> 0 */
                AT(f->KMOM_dUWINDdz, i, j, 0, 0, nzs) = 0.0;
                AT(f->KMOM_dUWINDdz, i, j, nz, 0, nzs) = 0.0;
                const int lo = (j == 1), hi = (j == ny);
                for (int k = 1; k < nz; k++) {
                    real cak, alt_km1, alt;
                    if (lo) {
                        cak = 0.25 * (RK(i - 1, j + 1) + RK(i, j + 1) + RK(i - 1, j) + RK(i, j));
                        alt_km1 = 0.25 * (PH(i - 1, j + 1, k - 1) + PH(i, j + 1, k - 1) + PH(i - 1, j, k - 1) + PH(i, j, k - 1)) / CON_G;
                        alt = 0.25 * (PH(i - 1, j + 1, k) + PH(i, j + 1, k) + PH(i - 1, j, k) + PH(i, j, k)) / CON_G;
                    } else if (hi) {
                        cak = 0.25 * (RK(i - 1, j) + RK(i, j) + RK(i - 1, j - 1) + RK(i, j - 1));
                        alt_km1 = 0.25 * (PH(i - 1, j, k - 1) + PH(i, j, k - 1) + PH(i - 1, j - 1, k - 1) + PH(i, j - 1, k - 1)) / CON_G;
                        alt = 0.25 * (PH(i - 1, j, k) + PH(i, j, k) + PH(i - 1, j - 1, k) + PH(i, j - 1, k)) / CON_G;
                    } else {
                        cak = 0.125 * (RK(i - 1, j + 1) + RK(i, j + 1) + RK2(i - 1, j) + RK2(i, j) + RK(i - 1, j - 1) + RK(i, j - 1));
                        alt_km1 = 0.125 * (PH(i - 1, j + 1, k - 1) + PH(i, j + 1, k - 1) + 2.0 * PH(i - 1, j, k - 1) + 2.0 * PH(i, j, k - 1) + PH(i - 1, j - 1, k - 1) + PH(i, j - 1, k - 1)) / CON_G;
                        alt = 0.125 * (PH(i - 1, j + 1, k) + PH(i, j + 1, k) + 2.0 * PH(i - 1, j, k) + 2.0 * PH(i, j, k) + PH(i - 1, j - 1, k) + PH(i, j - 1, k)) / CON_G;
                    }
                    const real dudz = (AT(f->UWIND, i, j, k - 1, 0, nz) - AT(f->UWIND, i, j, k, 0, nz)) / (alt_km1 - alt);
                    AT(f->KMOM_dUWINDdz, i, j, k, 0, nzs) = cak * dudz;
                }
            }
        PAR_FOR
        for (int i = 1; i < nx + 1; i++)
            for (int j = 1; j < ny + 2; j++) {
                AT(f->KMOM_dVWINDdz, i, j, 0, 1, nzs) = 0.0;
                AT(f->KMOM_dVWINDdz, i, j, nz, 1, nzs) = 0.0;
                for (int k = 1; k < nz; k++) {
                    const real cak = 0.125 * (RK(i + 1, j - 1) + RK(i + 1, j) + RK2(i, j - 1) + RK2(i, j) + RK(i - 1, j - 1) + RK(i - 1, j));
                    const real alt_km1 = 0.125 * (PH(i + 1, j - 1, k - 1) + PH(i + 1, j, k - 1) + 2.0 * PH(i, j - 1, k - 1) + 2.0 * PH(i, j, k - 1) + PH(i - 1, j - 1, k - 1) + PH(i - 1, j, k - 1)) / CON_G;
                    const real alt = 0.125 * (PH(i + 1, j - 1, k) + PH(i + 1, j, k) + 2.0 * PH(i, j - 1, k) + 2.0 * PH(i, j, k) + PH(i - 1, j - 1, k) + PH(i - 1, j, k)) / CON_G;
                    const real dvdz = (AT(f->VWIND, i, j, k - 1, 1, nz) - AT(f->VWIND, i, j, k, 1, nz)) / (alt_km1 - alt);
                    AT(f->KMOM_dVWINDdz, i, j, k, 1, nzs) = cak * dvdz;
                }
            }
#undef RK
#undef RK2
#undef PH
    }
    /* ---- the eight Arakawa-C momentum fluxes (:357-436; dyn_functions.py:429-536) ---- */
#define UF(ii, jj) AT(f->UFLX, ii, jj, k, 0, nz)
#define VF(ii, jj) AT(f->VFLX, ii, jj, k, 1, nz)
    PAR_FOR
    for (int i = 1; i < (hadv ? nx + 2 : 0); i++)
        for (int j = 1; j < ny + 2; j++)
            for (int k = 0; k < nz; k++) {
                /* (is,js): CFLX, QFLX */
                AT(f->CFLX, i, j, k, 1, nz) = 1.0 / 12.0 *
                    (VF(i - 1, j - 1) + VF(i, j - 1) + 2.0 * (VF(i - 1, j) + VF(i, j)) + VF(i - 1, j + 1) + VF(i, j + 1));
                AT(f->QFLX, i, j, k, 1, nz) = 1.0 / 12.0 *
                    (UF(i - 1, j - 1) + UF(i - 1, j) + 2.0 * (UF(i, j - 1) + UF(i, j)) + UF(i + 1, j - 1) + UF(i + 1, j));
                if (i < nx + 1) { /* (i,js): DFLX, EFLX */
                    AT(f->DFLX, i, j, k, 1, nz) = 1.0 / 24.0 *
                        (VF(i, j - 1) + 2.0 * VF(i, j) + VF(i, j + 1) + UF(i, j - 1) + UF(i, j) + UF(i + 1, j - 1) + UF(i + 1, j));
                    AT(f->EFLX, i, j, k, 1, nz) = 1.0 / 24.0 *
                        (VF(i, j - 1) + 2.0 * VF(i, j) + VF(i, j + 1) - UF(i, j - 1) - UF(i, j) - UF(i + 1, j - 1) - UF(i + 1, j));
                }
                if (j < ny + 1) { /* (is,j): SFLX, TFLX */
                    AT(f->SFLX, i, j, k, 0, nz) = 1.0 / 24.0 *
                        (VF(i - 1, j) + VF(i - 1, j + 1) + VF(i, j) + VF(i, j + 1) + UF(i - 1, j) + 2.0 * UF(i, j) + UF(i + 1, j));
                    AT(f->TFLX, i, j, k, 0, nz) = 1.0 / 24.0 *
                        (VF(i - 1, j) + VF(i - 1, j + 1) + VF(i, j) + VF(i, j + 1) - UF(i - 1, j) - 2.0 * UF(i, j) - UF(i + 1, j));
                }
                if (i < nx + 1 && j < ny + 1) { /* (i,j): BFLX, RFLX */
                    AT(f->BFLX, i, j, k, 0, nz) = 1.0 / 12.0 *
                        (UF(i, j - 1) + UF(i + 1, j - 1) + 2.0 * (UF(i, j) + UF(i + 1, j)) + UF(i, j + 1) + UF(i + 1, j + 1));
                    AT(f->RFLX, i, j, k, 0, nz) = 1.0 / 12.0 *
                        (VF(i - 1, j) + VF(i - 1, j + 1) + 2.0 * (VF(i, j) + VF(i, j + 1)) + VF(i + 1, j) + VF(i + 1, j + 1));
                }
            }
#undef UF
#undef VF
}

/* ------------------------------------------------------------------------- */
/* dyn_functions.py:383-424 (note: divides by g itself) */
static inline real var_ds(real v0, real v_dm1, real v_pm1, real v_pp1,
                            real v_pm1_dm1, real v_pp1_dm1, int wall_lo, int wall_hi)
{
    if (wall_lo) return 0.25 * (v_pp1_dm1 + v_pp1 + v_dm1 + v0) / CON_G;
    if (wall_hi) return 0.25 * (v_dm1 + v0 + v_pm1_dm1 + v_pm1) / CON_G;
    return 0.125 * (v_pp1_dm1 + v_pp1 + 2.0 * v_dm1 + 2.0 * v0 + v_pm1_dm1 + v_pm1) / CON_G;
}

/* dyn_functions.py:177-207 */
static inline real pre_grad(real phi, real phi_dm1, real colp, real colp_dm1,
                              real pott, real pott_dm1, real pvtf, real pvtf_dm1,
                              real pvtfvb, real pvtfvb_dm1, real pvtfvb_dm1_kp1,
                              real pvtfvb_kp1, real dsig, real svb, real svb_kp1,
                              real dgrid)
{
    return -dgrid * ((phi - phi_dm1) * (colp + colp_dm1) / 2.0 +
                     (colp - colp_dm1) * CON_CP / 2.0 *
                         (+pott_dm1 / dsig * (svb_kp1 * (pvtfvb_dm1_kp1 - pvtf_dm1) + svb * (pvtf_dm1 - pvtfvb_dm1)) +
                          pott / dsig * (svb_kp1 * (pvtfvb_kp1 - pvtf) + svb * (pvtf - pvtfvb))));
}

/* dyn_UFLX.py:339-434, add_up_tendencies_py :69-199 */
void oracle_uflx_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
#define U(ii, jj) AT(f->UWIND, ii, jj, k, 0, nz)
#define V(ii, jj) AT(f->VWIND, ii, jj, k, 1, nz)
    PAR_FOR
    for (int i = 1; i < nx + 1; i++) {
        const int im1 = (i == 1) ? nx : i - 1;          /* :366-374 */
        for (int j = 1; j < ny + 1; j++) {
            const real colp = A2(f->COLP, i, j), colp_im1 = A2(f->COLP, i - 1, j);
            const real corf = AT(g->corf_is, i, j, 0, 0, 1), lat = AT(g->lat_is_rad, i, j, 0, 0, 1);
            const real dlon = A2S(g->dlon_rad, i, j), dlat = AT(g->dlat_rad, i, j, 0, 0, 1);
            const real dyis = AT(g->dyis, i, j, 0, 0, 1);
            const int lo = (j == 1), hi = (j == ny);
            real smom_is = 0.0;
            if (c->i_turb)
                smom_is = var_ds(A2(f->SMOMXFLX, i, j), A2(f->SMOMXFLX, i - 1, j), A2(f->SMOMXFLX, i, j - 1),
                                 A2(f->SMOMXFLX, i, j + 1), A2(f->SMOMXFLX, i - 1, j - 1), A2(f->SMOMXFLX, i - 1, j + 1), lo, hi);
            for (int k = 0; k < nz; k++) {
                real bflx = AT(f->BFLX, i, j, k, 0, nz), bflx_im1 = AT(f->BFLX, im1, j, k, 0, nz);
                real cflx = AT(f->CFLX, i, j, k, 1, nz), cflx_jp1 = AT(f->CFLX, i, j + 1, k, 1, nz);
                real dflx_im1 = AT(f->DFLX, im1, j, k, 1, nz), dflx_jp1 = AT(f->DFLX, i, j + 1, k, 1, nz);
                real eflx = AT(f->EFLX, i, j, k, 1, nz), eflx_im1_jp1 = AT(f->EFLX, im1, j + 1, k, 1, nz);
                if (lo) { dflx_im1 = 0.0; cflx = 0.0; eflx = 0.0; }               /* :377-380 */
                if (hi) { dflx_jp1 = 0.0; cflx_jp1 = 0.0; eflx_im1_jp1 = 0.0; }   /* :381-384 */
                const real u = U(i, j);
                real t = 0.0;
                /* horizontal advection, dyn_functions.py:541-572, sign +1 */
                if (TERM_ON(c, ORACLE_TERM_UVFLX_HOR_ADV))
                t = t + (+bflx_im1 * (U(i - 1, j) + u) / 2.0 - bflx * (u + U(i + 1, j)) / 2.0
                         + cflx * (U(i, j - 1) + u) / 2.0 - cflx_jp1 * (u + U(i, j + 1)) / 2.0
                         + dflx_im1 * (U(i - 1, j - 1) + u) / 2.0 - dflx_jp1 * (u + U(i + 1, j + 1)) / 2.0
                         + 1.0 * (+eflx * (U(i + 1, j - 1) + u) / 2.0 - eflx_im1_jp1 * (u + U(i - 1, j + 1)) / 2.0));
                /* vertical advection */
                if (TERM_ON(c, ORACLE_TERM_UVFLX_VERT_ADV))
                t = t + (AT(f->WWIND_UWIND, i, j, k, 0, nzs) - AT(f->WWIND_UWIND, i, j, k + 1, 0, nzs)) / g->dsigma[k];
                /* vertical turbulent transport (:137-170); interp_VAR_ds already divides by g,
                 * ALTVB is divided by g a second time, RHO and SMOMXFLX once (reference quirk) */
                if (c->i_turb) {
#define PV(ii, jj, kk) AT(f->PHIVB, ii, jj, kk, 0, nzs)
#define RH(ii, jj) AT(f->RHO, ii, jj, k, 0, nz)
                    const real altvb = var_ds(PV(i, j, k), PV(i - 1, j, k), PV(i, j - 1, k), PV(i, j + 1, k), PV(i - 1, j - 1, k), PV(i - 1, j + 1, k), lo, hi) / CON_G;
                    const real altvb_kp1 = var_ds(PV(i, j, k + 1), PV(i - 1, j, k + 1), PV(i, j - 1, k + 1), PV(i, j + 1, k + 1), PV(i - 1, j - 1, k + 1), PV(i - 1, j + 1, k + 1), lo, hi) / CON_G;
                    const real rho_is = var_ds(RH(i, j), RH(i - 1, j), RH(i, j - 1), RH(i, j + 1), RH(i - 1, j - 1), RH(i - 1, j + 1), lo, hi);
                    const real kd = AT(f->KMOM_dUWINDdz, i, j, k, 0, nzs), kd_kp1 = AT(f->KMOM_dUWINDdz, i, j, k + 1, 0, nzs);
                    real turb;
                    if (k == 0) turb = (0.0 - kd_kp1) / ((altvb - altvb_kp1) * rho_is);
                    else if (k == nz - 1) turb = (kd + smom_is) / ((altvb - altvb_kp1) * rho_is);
                    else turb = (kd - kd_kp1) / ((altvb - altvb_kp1) * rho_is);
                    t = t + turb;
                    AT(f->dUFLXdt_TURB, i, j, k, 0, nz) = turb;
#undef PV
#undef RH
                }
                /* Coriolis + spherical metric, dyn_UFLX.py:39-66 */
                if (TERM_ON(c, ORACLE_TERM_UVFLX_CORIOLIS))
                t = t + CON_RE * dlon * dlat / 2.0 *
                            (colp_im1 * (V(i - 1, j) + V(i - 1, j + 1)) / 2.0 *
                                 (corf * CON_RE * cos(lat) + (U(i - 1, j) + u) / 2.0 * sin(lat)) +
                             colp * (V(i, j) + V(i, j + 1)) / 2.0 *
                                 (corf * CON_RE * cos(lat) + (u + U(i + 1, j)) / 2.0 * sin(lat)));
                /* pressure gradient */
                if (TERM_ON(c, ORACLE_TERM_UVFLX_PRE_GRAD))
                t = t + pre_grad(AT(f->PHI, i, j, k, 0, nz), AT(f->PHI, i - 1, j, k, 0, nz), colp, colp_im1,
                                 AT(f->POTT, i, j, k, 0, nz), AT(f->POTT, i - 1, j, k, 0, nz),
                                 AT(f->PVTF, i, j, k, 0, nz), AT(f->PVTF, i - 1, j, k, 0, nz),
                                 AT(f->PVTFVB, i, j, k, 0, nzs), AT(f->PVTFVB, i - 1, j, k, 0, nzs),
                                 AT(f->PVTFVB, i - 1, j, k + 1, 0, nzs), AT(f->PVTFVB, i, j, k + 1, 0, nzs),
                                 g->dsigma[k], g->sigma_vb[k], g->sigma_vb[k + 1], dyis);
                /* horizontal diffusion, dyn_functions.py:158-170 */
                if (g->UVFLX_dif_coef[k] > 0.0 && TERM_ON(c, ORACLE_TERM_UVFLX_NUM_DIF))
                    t = t + g->UVFLX_dif_coef[k] *
                                (AT(f->UFLX, i - 1, j, k, 0, nz) + AT(f->UFLX, i + 1, j, k, 0, nz) +
                                 AT(f->UFLX, i, j - 1, k, 0, nz) + AT(f->UFLX, i, j + 1, k, 0, nz) -
                                 4.0 * AT(f->UFLX, i, j, k, 0, nz));
                AT(f->dUFLXdt, i, j, k, 0, nz) = TERM_ON(c, ORACLE_TERM_UVFLX_MAIN) ? t : (real)0.0;   /* dyn_UFLX.py:115-117 */
            }
        }
    }
#undef U
#undef V
}

/* dyn_VFLX.py:322-408, add_up_tendencies_py :67-198.  Rows js = 1 and ny+1 (the
 * poles) are skipped: the reference computes them from NaN halos and
 * exchange_BC zeroes VWIND there (misc_boundaries.py:36-38). */
void oracle_vflx_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
#define U(ii, jj) AT(f->UWIND, ii, jj, k, 0, nz)
#define V(ii, jj) AT(f->VWIND, ii, jj, k, 1, nz)
    PAR_FOR
    for (int i = 1; i < nx + 1; i++) {
        const int ip1 = (i == nx) ? 1 : i + 1;          /* :349-356 */
        for (int j = 2; j < ny + 1; j++) {
            const real colp = A2(f->COLP, i, j), colp_jm1 = A2(f->COLP, i, j - 1);
            const real corf = A2(g->corf, i, j), corf_jm1 = A2(g->corf, i, j - 1);
            const real lat = A2(g->lat_rad, i, j), lat_jm1 = A2(g->lat_rad, i, j - 1);
            const real dlon = A2S(g->dlon_rad, i, j), dlat = AT(g->dlat_rad, i, j, 0, 0, 1);
            const real dxjs = A2S(g->dxjs, i, j);
            real smom_js = 0.0;
            if (c->i_turb)
                smom_js = var_ds(A2(f->SMOMYFLX, i, j), A2(f->SMOMYFLX, i, j - 1), A2(f->SMOMYFLX, i - 1, j),
                                 A2(f->SMOMYFLX, i + 1, j), A2(f->SMOMYFLX, i - 1, j - 1), A2(f->SMOMYFLX, i + 1, j - 1), 0, 0);
            for (int k = 0; k < nz; k++) {
                const real rflx = AT(f->RFLX, i, j, k, 0, nz), rflx_jm1 = AT(f->RFLX, i, j - 1, k, 0, nz);
                const real qflx = AT(f->QFLX, i, j, k, 1, nz), qflx_ip1 = AT(f->QFLX, ip1, j, k, 1, nz);
                const real sflx_jm1 = AT(f->SFLX, i, j - 1, k, 0, nz), sflx_ip1 = AT(f->SFLX, ip1, j, k, 0, nz);
                const real tflx = AT(f->TFLX, i, j, k, 0, nz), tflx_ip1_jm1 = AT(f->TFLX, ip1, j - 1, k, 0, nz);
                const real v = V(i, j);
                real t = 0.0;
                /* horizontal advection, sign -1 on the T term */
                if (TERM_ON(c, ORACLE_TERM_UVFLX_HOR_ADV))
                t = t + (+rflx_jm1 * (V(i, j - 1) + v) / 2.0 - rflx * (v + V(i, j + 1)) / 2.0
                         + qflx * (V(i - 1, j) + v) / 2.0 - qflx_ip1 * (v + V(i + 1, j)) / 2.0
                         + sflx_jm1 * (V(i - 1, j - 1) + v) / 2.0 - sflx_ip1 * (v + V(i + 1, j + 1)) / 2.0
                         + -1.0 * (+tflx * (V(i - 1, j + 1) + v) / 2.0 - tflx_ip1_jm1 * (v + V(i + 1, j - 1)) / 2.0));
                if (TERM_ON(c, ORACLE_TERM_UVFLX_VERT_ADV))
                t = t + (AT(f->WWIND_VWIND, i, j, k, 1, nzs) - AT(f->WWIND_VWIND, i, j, k + 1, 1, nzs)) / g->dsigma[k];
                if (c->i_turb) {
#define PV(ii, jj, kk) AT(f->PHIVB, ii, jj, kk, 0, nzs)
#define RH(ii, jj) AT(f->RHO, ii, jj, k, 0, nz)
                    const real altvb = var_ds(PV(i, j, k), PV(i, j - 1, k), PV(i - 1, j, k), PV(i + 1, j, k), PV(i - 1, j - 1, k), PV(i + 1, j - 1, k), 0, 0) / CON_G;
                    const real altvb_kp1 = var_ds(PV(i, j, k + 1), PV(i, j - 1, k + 1), PV(i - 1, j, k + 1), PV(i + 1, j, k + 1), PV(i - 1, j - 1, k + 1), PV(i + 1, j - 1, k + 1), 0, 0) / CON_G;
                    const real rho_js = var_ds(RH(i, j), RH(i, j - 1), RH(i - 1, j), RH(i + 1, j), RH(i - 1, j - 1), RH(i + 1, j - 1), 0, 0);
                    const real kd = AT(f->KMOM_dVWINDdz, i, j, k, 1, nzs), kd_kp1 = AT(f->KMOM_dVWINDdz, i, j, k + 1, 1, nzs);
                    real turb;
                    if (k == 0) turb = (0.0 - kd_kp1) / ((altvb - altvb_kp1) * rho_js);
                    else if (k == nz - 1) turb = (kd + smom_js) / ((altvb - altvb_kp1) * rho_js);
                    else turb = (kd - kd_kp1) / ((altvb - altvb_kp1) * rho_js);
                    t = t + turb;
                    AT(f->dVFLXdt_TURB, i, j, k, 1, nz) = turb;
#undef PV
#undef RH
                }
                /* Coriolis + metric, dyn_VFLX.py:39-64 */
                if (TERM_ON(c, ORACLE_TERM_UVFLX_CORIOLIS))
                t = t + -CON_RE * dlon * dlat / 2.0 *
                            (colp_jm1 * (U(i, j - 1) + U(i + 1, j - 1)) / 2.0 *
                                 (corf_jm1 * CON_RE * cos(lat_jm1) + (U(i, j - 1) + U(i + 1, j - 1)) / 2.0 * sin(lat_jm1)) +
                             colp * (U(i, j) + U(i + 1, j)) / 2.0 *
                                 (corf * CON_RE * cos(lat) + (U(i, j) + U(i + 1, j)) / 2.0 * sin(lat)));
                if (TERM_ON(c, ORACLE_TERM_UVFLX_PRE_GRAD))
                t = t + pre_grad(AT(f->PHI, i, j, k, 0, nz), AT(f->PHI, i, j - 1, k, 0, nz), colp, colp_jm1,
                                 AT(f->POTT, i, j, k, 0, nz), AT(f->POTT, i, j - 1, k, 0, nz),
                                 AT(f->PVTF, i, j, k, 0, nz), AT(f->PVTF, i, j - 1, k, 0, nz),
                                 AT(f->PVTFVB, i, j, k, 0, nzs), AT(f->PVTFVB, i, j - 1, k, 0, nzs),
                                 AT(f->PVTFVB, i, j - 1, k + 1, 0, nzs), AT(f->PVTFVB, i, j, k + 1, 0, nzs),
                                 g->dsigma[k], g->sigma_vb[k], g->sigma_vb[k + 1], dxjs);
                if (g->UVFLX_dif_coef[k] > 0.0 && TERM_ON(c, ORACLE_TERM_UVFLX_NUM_DIF))
                    t = t + g->UVFLX_dif_coef[k] *
                                (AT(f->VFLX, i - 1, j, k, 1, nz) + AT(f->VFLX, i + 1, j, k, 1, nz) +
                                 AT(f->VFLX, i, j - 1, k, 1, nz) + AT(f->VFLX, i, j + 1, k, 1, nz) -
                                 4.0 * AT(f->VFLX, i, j, k, 1, nz));
                AT(f->dVFLXdt, i, j, k, 1, nz) = TERM_ON(c, ORACLE_TERM_UVFLX_MAIN) ? t : (real)0.0;
            }
        }
    }
#undef U
#undef V
}

/* ------------------------------------------------------------------------- */
/* dyn_POTT.py:180-216, add_up_tendencies_py :55-110 */
void oracle_pott_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
#define P(ii, jj, kk) AT(f->POTT, ii, jj, kk, 0, nz)
    PAR_FOR
    for (int i = 1; i < nx + 1; i++)
        for (int j = 1; j < ny + 1; j++) {
            const real colp = A2(f->COLP, i, j), colp_new = A2(f->COLP_NEW, i, j);
            const real c_im1 = A2(f->COLP, i - 1, j), c_ip1 = A2(f->COLP, i + 1, j);
            const real c_jm1 = A2(f->COLP, i, j - 1), c_jp1 = A2(f->COLP, i, j + 1);
            const real area = A2(g->A, i, j);
            for (int k = 0; k < nz; k++) {
                const real p = P(i, j, k);
                real t = 0.0;
                /* hor_adv_py, dyn_functions.py:105-114 */
                if (TERM_ON(c, ORACLE_TERM_POTT_HOR_ADV))
                t = t + (+AT(f->UFLX, i, j, k, 0, nz) * (P(i - 1, j, k) + p) / 2.0
                         - AT(f->UFLX, i + 1, j, k, 0, nz) * (p + P(i + 1, j, k)) / 2.0
                         + AT(f->VFLX, i, j, k, 1, nz) * (P(i, j - 1, k) + p) / 2.0
                         - AT(f->VFLX, i, j + 1, k, 1, nz) * (p + P(i, j + 1, k)) / 2.0) / area;
                /* vert_adv_py, dyn_functions.py:118-138 */
                if (TERM_ON(c, ORACLE_TERM_POTT_VERT_ADV)) {
                    const real w_kp1 = AT(f->WWIND, i, j, k + 1, 0, nzs), pvb_kp1 = AT(f->POTTVB, i, j, k + 1, 0, nzs);
                    if (k == 0)
                        t = t + colp_new * (-w_kp1 * pvb_kp1) / g->dsigma[k];
                    else
                        t = t + colp_new * (+AT(f->WWIND, i, j, k, 0, nzs) * AT(f->POTTVB, i, j, k, 0, nzs) - w_kp1 * pvb_kp1) / g->dsigma[k];
                }
                /* turb_flux_tendency_py, dyn_functions.py:26-67 */
                if (c->i_turb) {
                    const real alt = AT(f->PHI, i, j, k, 0, nz) / CON_G;
                    const real altvb = AT(f->PHIVB, i, j, k, 0, nzs) / CON_G, altvb_kp1 = AT(f->PHIVB, i, j, k + 1, 0, nzs) / CON_G;
                    const real rho = AT(f->RHO, i, j, k, 0, nz);
                    const real surf = A2(f->SSHFLX, i, j) / CON_CP;
                    real turb;
                    if (k == 0) {
                        const real alt_kp1 = AT(f->PHI, i, j, k + 1, 0, nz) / CON_G;
                        turb = colp * ((+0.0 - ((p - P(i, j, k + 1)) / (alt - alt_kp1) * AT(f->RHOVB, i, j, k + 1, 0, nzs) * AT(f->KHEAT, i, j, k + 1, 0, nzs))) /
                                       ((altvb - altvb_kp1) * rho));
                    } else if (k == nz - 1) {
                        const real alt_km1 = AT(f->PHI, i, j, k - 1, 0, nz) / CON_G;
                        turb = colp * ((+((P(i, j, k - 1) - p) / (alt_km1 - alt) * AT(f->RHOVB, i, j, k, 0, nzs) * AT(f->KHEAT, i, j, k, 0, nzs)) + surf) /
                                       ((altvb - altvb_kp1) * rho));
                    } else {
                        const real alt_kp1 = AT(f->PHI, i, j, k + 1, 0, nz) / CON_G;
                        const real alt_km1 = AT(f->PHI, i, j, k - 1, 0, nz) / CON_G;
                        turb = colp * ((+((P(i, j, k - 1) - p) / (alt_km1 - alt) * AT(f->RHOVB, i, j, k, 0, nzs) * AT(f->KHEAT, i, j, k, 0, nzs))
                                        - ((p - P(i, j, k + 1)) / (alt - alt_kp1) * AT(f->RHOVB, i, j, k + 1, 0, nzs) * AT(f->KHEAT, i, j, k + 1, 0, nzs))) /
                                       ((altvb - altvb_kp1) * rho));
                    }
                    t = t + turb;
                    AT(f->dPOTTdt_TURB, i, j, k, 0, nz) = turb / colp * 3600.0;   /* dyn_POTT.py:93-94 */
                }
                /* num_dif_pw_py, dyn_functions.py:142-155 */
                if (g->POTT_dif_coef[k] > 0.0 && TERM_ON(c, ORACLE_TERM_POTT_NUM_DIF))
                    t = t + g->POTT_dif_coef[k] * (+c_im1 * P(i - 1, j, k) + c_ip1 * P(i + 1, j, k) + c_jm1 * P(i, j - 1, k) + c_jp1 * P(i, j + 1, k) - 4.0 * colp * p);
                if (c->i_radiation)
                    t = t + AT(f->dPOTTdt_RAD, i, j, k, 0, nz) * colp;
                AT(f->dPOTTdt, i, j, k, 0, nz) = TERM_ON(c, ORACLE_TERM_POTT_MAIN) ? t : (real)0.0;   /* dyn_POTT.py:70-72 */
            }
        }
#undef P
}


/* ------------------------------------------------------------------------- */
/* dyn_functions.py:70-95: interface value by logarithmic interpolation.  max(VAR, min_val)
 * is Python's: the first argument unless the second is greater. */
static inline real varvb_log(real var, real var_km1)
{
    const real min_val = 0.0000001;
    if (min_val > var) var = min_val;
    if (min_val > var_km1) var_km1 = min_val;
    if (var_km1 == var) return var;
    return (log(var_km1) - log(var)) / (1.0 / var - 1.0 / var_km1);
}

/* dyn_functions.py:26-67 */
static inline real turb_flux_tendency(real phi, real phi_kp1, real phi_km1, real phivb,
                                        real phivb_kp1, real var, real var_kp1, real var_km1,
                                        real kvar, real kvar_kp1, real rho, real rhovb,
                                        real rhovb_kp1, real colp, real surf_flux, int k, int nz)
{
    const real alt = phi / CON_G, alt_kp1 = phi_kp1 / CON_G, alt_km1 = phi_km1 / CON_G;
    const real altvb = phivb / CON_G, altvb_kp1 = phivb_kp1 / CON_G;
    if (k == 0)
        return colp * ((+0.0 - ((var - var_kp1) / (alt - alt_kp1) * rhovb_kp1 * kvar_kp1)) / ((altvb - altvb_kp1) * rho));
    if (k == nz - 1)
        return colp * ((+((var_km1 - var) / (alt_km1 - alt) * rhovb * kvar) + surf_flux) / ((altvb - altvb_kp1) * rho));
    return colp * ((+((var_km1 - var) / (alt_km1 - alt) * rhovb * kvar)
                    - ((var - var_kp1) / (alt - alt_kp1) * rhovb_kp1 * kvar_kp1)) / ((altvb - altvb_kp1) * rho));
}

/* dyn_moist.py:201-243 (loop) and :49-129 (terms).  QV[i,j,k+1] at k = nz-1 is an unchecked
 * out-of-range read in the reference whose interface value is multiplied by WWIND[nz] = 0;
 * the product is taken as 0 here.  The k-1 neighbour at k = 0 wraps to nz-1 in the reference
 * and is not used by any term. */
void oracle_moist_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
#define QVA(ii, jj, kk) AT(f->QV, ii, jj, kk, 0, nz)
#define QCA(ii, jj, kk) AT(f->QC, ii, jj, kk, 0, nz)
    PAR_FOR
    for (int i = 1; i < nx + 1; i++)
        for (int j = 1; j < ny + 1; j++) {
            const real colp = A2(f->COLP, i, j), colp_new = A2(f->COLP_NEW, i, j);
            const real c_im1 = A2(f->COLP, i - 1, j), c_ip1 = A2(f->COLP, i + 1, j);
            const real c_jm1 = A2(f->COLP, i, j - 1), c_jp1 = A2(f->COLP, i, j + 1);
            const real area = A2(g->A, i, j);
            for (int k = 0; k < nz; k++) {
                const real uflx = AT(f->UFLX, i, j, k, 0, nz), uflx_ip1 = AT(f->UFLX, i + 1, j, k, 0, nz);
                const real vflx = AT(f->VFLX, i, j, k, 1, nz), vflx_jp1 = AT(f->VFLX, i, j + 1, k, 1, nz);
                const real w = AT(f->WWIND, i, j, k, 0, nzs), w_kp1 = AT(f->WWIND, i, j, k + 1, 0, nzs);
                const real ds = g->dsigma[k], dif = g->moist_dif_coef[k];
                real t[2];
                real turb_qv = 0.0;
                for (int s = 0; s < 2; s++) {
                    real *Q = s == 0 ? f->QV : f->QC;
#define QA(ii, jj, kk) AT(Q, ii, jj, kk, 0, nz)
                    const real q = QA(i, j, k);
                    const real q_im1 = QA(i - 1, j, k), q_ip1 = QA(i + 1, j, k), q_jm1 = QA(i, j - 1, k), q_jp1 = QA(i, j + 1, k);
                    real d = 0.0;
                    /* hor_adv_py */
                    if (TERM_ON(c, ORACLE_TERM_MOIST_HOR_ADV))
                    d = d + (+uflx * (q_im1 + q) / 2.0 - uflx_ip1 * (q + q_ip1) / 2.0
                             + vflx * (q_jm1 + q) / 2.0 - vflx_jp1 * (q + q_jp1) / 2.0) / area;
                    /* vertical advection with log-interpolated interface values */
                    if (TERM_ON(c, ORACLE_TERM_MOIST_VERT_ADV)) {
                        real flux_kp1 = 0.0;                           /* W[nz] * (unchecked read) = 0 */
                        if (k < nz - 1) flux_kp1 = w_kp1 * varvb_log(QA(i, j, k + 1), q);
                        if (k == 0)
                            d = d + colp_new * (-flux_kp1) / ds;
                        else
                            d = d + colp_new * (+w * varvb_log(q, QA(i, j, k - 1)) - flux_kp1) / ds;
                    }
                    if (c->i_turb) {
                        const real surf = s == 0 ? A2(f->SLHFLX, i, j) / CON_LH : 0.0;
                        const real tb = turb_flux_tendency(
                            AT(f->PHI, i, j, k, 0, nz), k < nz - 1 ? AT(f->PHI, i, j, k + 1, 0, nz) : 0.0,
                            k > 0 ? AT(f->PHI, i, j, k - 1, 0, nz) : 0.0,
                            AT(f->PHIVB, i, j, k, 0, nzs), AT(f->PHIVB, i, j, k + 1, 0, nzs),
                            q, k < nz - 1 ? QA(i, j, k + 1) : 0.0, k > 0 ? QA(i, j, k - 1) : 0.0,
                            AT(f->KHEAT, i, j, k, 0, nzs), AT(f->KHEAT, i, j, k + 1, 0, nzs),
                            AT(f->RHO, i, j, k, 0, nz), AT(f->RHOVB, i, j, k, 0, nzs), AT(f->RHOVB, i, j, k + 1, 0, nzs),
                            colp, surf, k, nz);
                        d = d + tb;
                        if (s == 0) turb_qv = tb;
                    }
                    if (dif > 0.0 && TERM_ON(c, ORACLE_TERM_MOIST_NUM_DIF))
                        d = d + dif * (+c_im1 * q_im1 + c_ip1 * q_ip1 + c_jm1 * q_jm1 + c_jp1 * q_jp1 - 4.0 * colp * q);
                    t[s] = d;
#undef QA
                }
                AT(f->dQVdt, i, j, k, 0, nz) = t[0];
                AT(f->dQCdt, i, j, k, 0, nz) = t[1];
                if (c->i_turb) AT(f->dQVdt_TURB, i, j, k, 0, nz) = turb_qv;
            }
        }
#undef QVA
#undef QCA
}

/* ------------------------------------------------------------------------- */
/* dyn_timestep.py:212-297 (serial in the reference: plain njit) */
void oracle_make_timestep(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz;
#define CN(ii, jj) (A2(f->COLP, ii, jj))
#define CO(ii, jj) (A2(f->COLP_OLD, ii, jj))
#define AR(ii, jj) (A2(g->A, ii, jj))
    for (int i = 1; i < nx + 2; i++)
        for (int j = 1; j < ny + 2; j++) {
            if (i < nx + 1 && j < ny + 1) {   /* UWIND: is in [1,nx], j in [1,ny]; column nx+1 is the
                                                 periodic duplicate refreshed by exchange_BC */
                real ca, ca_old;
                if (j == 1) {
                    ca = 1.0 / 4.0 * (CN(i - 1, j + 1) * AR(i - 1, j + 1) + CN(i, j + 1) * AR(i, j + 1) + CN(i - 1, j) * AR(i - 1, j) + CN(i, j) * AR(i, j));
                    ca_old = 1.0 / 4.0 * (CO(i - 1, j + 1) * AR(i - 1, j + 1) + CO(i, j + 1) * AR(i, j + 1) + CO(i - 1, j) * AR(i - 1, j) + CO(i, j) * AR(i, j));
                } else if (j == ny) {
                    ca = 1.0 / 4.0 * (CN(i - 1, j - 1) * AR(i - 1, j - 1) + CN(i, j - 1) * AR(i, j - 1) + CN(i - 1, j) * AR(i - 1, j) + CN(i, j) * AR(i, j));
                    ca_old = 1.0 / 4.0 * (CO(i - 1, j - 1) * AR(i - 1, j - 1) + CO(i, j - 1) * AR(i, j - 1) + CO(i - 1, j) * AR(i - 1, j) + CO(i, j) * AR(i, j));
                } else {
                    ca = 1.0 / 8.0 * (CN(i - 1, j + 1) * AR(i - 1, j + 1) + CN(i, j + 1) * AR(i, j + 1) + 2.0 * CN(i - 1, j) * AR(i - 1, j) + 2.0 * CN(i, j) * AR(i, j) + CN(i - 1, j - 1) * AR(i - 1, j - 1) + CN(i, j - 1) * AR(i, j - 1));
                    ca_old = 1.0 / 8.0 * (CO(i - 1, j + 1) * AR(i - 1, j + 1) + CO(i, j + 1) * AR(i, j + 1) + 2.0 * CO(i - 1, j) * AR(i - 1, j) + 2.0 * CO(i, j) * AR(i, j) + CO(i - 1, j - 1) * AR(i - 1, j - 1) + CO(i, j - 1) * AR(i, j - 1));
                }
                for (int k = 0; k < nz; k++)
                    AT(f->UWIND, i, j, k, 0, nz) = AT(f->UWIND_OLD, i, j, k, 0, nz) * ca_old / ca + dt * AT(f->dUFLXdt, i, j, k, 0, nz) / ca;
            }
            if (i < nx + 1 && j > 1 && j < ny + 1) {   /* VWIND: js in [2,ny] (poles are zeroed by exchange_BC) */
                const real ca = 1.0 / 8.0 * (CN(i + 1, j - 1) * AR(i + 1, j - 1) + CN(i + 1, j) * AR(i + 1, j) + 2.0 * CN(i, j - 1) * AR(i, j - 1) + 2.0 * CN(i, j) * AR(i, j) + CN(i - 1, j - 1) * AR(i - 1, j - 1) + CN(i - 1, j) * AR(i - 1, j));
                const real ca_old = 1.0 / 8.0 * (CO(i + 1, j - 1) * AR(i + 1, j - 1) + CO(i + 1, j) * AR(i + 1, j) + 2.0 * CO(i, j - 1) * AR(i, j - 1) + 2.0 * CO(i, j) * AR(i, j) + CO(i - 1, j - 1) * AR(i - 1, j - 1) + CO(i - 1, j) * AR(i - 1, j));
                for (int k = 0; k < nz; k++)
                    AT(f->VWIND, i, j, k, 1, nz) = AT(f->VWIND_OLD, i, j, k, 1, nz) * ca_old / ca + dt * AT(f->dVFLXdt, i, j, k, 1, nz) / ca;
            }
            if (i < nx + 1 && j < ny + 1) {
                const real cn = CN(i, j), co = CO(i, j);
                for (int k = 0; k < nz; k++)
                    AT(f->POTT, i, j, k, 0, nz) = AT(f->POTT_OLD, i, j, k, 0, nz) * co / cn + dt * AT(f->dPOTTdt, i, j, k, 0, nz) / cn;
                if (c->i_moist)   /* no clipping of QV/QC on the CPU path (dyn_timestep.py:291-296) */
                    for (int k = 0; k < nz; k++) {
                        AT(f->QV, i, j, k, 0, nz) = AT(f->QV_OLD, i, j, k, 0, nz) * co / cn + dt * AT(f->dQVdt, i, j, k, 0, nz) / cn;
                        AT(f->QC, i, j, k, 0, nz) = AT(f->QC_OLD, i, j, k, 0, nz) * co / cn + dt * AT(f->dQCdt, i, j, k, 0, nz) / cn;
                        if (c->i_moist_clip) {      /* dyn_timestep.py:145-154 (accelerator path) */
                            if (AT(f->QV, i, j, k, 0, nz) < 0.0) AT(f->QV, i, j, k, 0, nz) = 0.0;
                            if (AT(f->QC, i, j, k, 0, nz) < 0.0) AT(f->QC, i, j, k, 0, nz) = 0.0;
                        }
                    }
            }
        }
#undef CN
#undef CO
#undef AR
}

/* ------------------------------------------------------------------------- */
/* dyn_diagnostics.py:139-156 */
void oracle_diag_pvtf(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    PAR_FOR
    for (int i = 0; i < nx + 2; i++)
        for (int j = 0; j < ny + 2; j++) {
            const real colp = A2(f->COLP, i, j);
            for (int k = 0; k < nz; k++) {
                const real p_km12 = c->pair_top + g->sigma_vb[k] * colp;
                const real p_kp12 = c->pair_top + g->sigma_vb[k + 1] * colp;
                AT(f->PVTF, i, j, k, 0, nz) = 1.0 / (1.0 + CON_KAPPA) *
                    (pow(p_kp12 / 100000.0, CON_KAPPA) * p_kp12 - pow(p_km12 / 100000.0, CON_KAPPA) * p_km12) / (p_kp12 - p_km12);
                AT(f->PVTFVB, i, j, k, 0, nzs) = pow(p_km12 / 100000.0, CON_KAPPA);
                if (k == nz - 1) AT(f->PVTFVB, i, j, k + 1, 0, nzs) = pow(p_kp12 / 100000.0, CON_KAPPA);
            }
        }
}

/* dyn_diagnostics.py:160-174 */
void oracle_diag_phi(const oracle_cfg *c, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    PAR_FOR
    for (int i = 0; i < nx + 2; i++)
        for (int j = 0; j < ny + 2; j++) {
            AT(f->PHIVB, i, j, nz, 0, nzs) = A2(f->HSURF, i, j) * CON_G;
            for (int k = nz - 1; k >= 0; k--) {
                const real pott = AT(f->POTT, i, j, k, 0, nz), pvtf = AT(f->PVTF, i, j, k, 0, nz);
                const real phi = AT(f->PHIVB, i, j, k + 1, 0, nzs) - CON_CP * (pott * (pvtf - AT(f->PVTFVB, i, j, k + 1, 0, nzs)));
                AT(f->PHI, i, j, k, 0, nz) = phi;
                AT(f->PHIVB, i, j, k, 0, nzs) = phi - CON_CP * (pott * (AT(f->PVTFVB, i, j, k, 0, nzs) - pvtf));
            }
        }
}

/* dyn_diagnostics.py:179-195 */
void oracle_diag_pottvb(const oracle_cfg *c, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    PAR_FOR
    for (int i = 0; i < nx + 2; i++)
        for (int j = 0; j < ny + 2; j++)
            for (int k = 1; k < nzs - 1; k++) {
                const real v = (+(AT(f->PVTFVB, i, j, k, 0, nzs) - AT(f->PVTF, i, j, k - 1, 0, nz)) * AT(f->POTT, i, j, k - 1, 0, nz)
                                  + (AT(f->PVTF, i, j, k, 0, nz) - AT(f->PVTFVB, i, j, k, 0, nzs)) * AT(f->POTT, i, j, k, 0, nz)) /
                                 (AT(f->PVTF, i, j, k, 0, nz) - AT(f->PVTF, i, j, k - 1, 0, nz));
                AT(f->POTTVB, i, j, k, 0, nzs) = v;
                if (k == 1)          /* extrapolate to the model top */
                    AT(f->POTTVB, i, j, k - 1, 0, nzs) = AT(f->POTT, i, j, k - 1, 0, nz) - (v - AT(f->POTT, i, j, k - 1, 0, nz));
                else if (k == nzs - 2) /* ... and to the surface (elif in the reference) */
                    AT(f->POTTVB, i, j, k + 1, 0, nzs) = AT(f->POTT, i, j, k, 0, nz) - (v - AT(f->POTT, i, j, k, 0, nz));
            }
}

/* dyn_diagnostics.py:199-222, the RHO / RHOVB part */
void oracle_diag_rho(const oracle_cfg *c, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    PAR_FOR
    for (int i = 0; i < nx + 2; i++)
        for (int j = 0; j < ny + 2; j++) {
            for (int k = 0; k < nzs; k++) {
                const real pairvb = 100000.0 * pow(AT(f->PVTFVB, i, j, k, 0, nzs), 1.0 / CON_KAPPA);
                const real tairvb = AT(f->POTTVB, i, j, k, 0, nzs) * AT(f->PVTFVB, i, j, k, 0, nzs);
                AT(f->RHOVB, i, j, k, 0, nzs) = pairvb / (CON_RD * tairvb);
            }
            for (int k = 0; k < nz; k++) {
                const real tair = AT(f->POTT, i, j, k, 0, nz) * AT(f->PVTF, i, j, k, 0, nz);
                const real pair = 100000.0 * pow(AT(f->PVTF, i, j, k, 0, nz), 1.0 / CON_KAPPA);
                AT(f->RHO, i, j, k, 0, nz) = pair / (CON_RD * tair);
            }
        }
}

/* dyn_diagnostics.py:199-222, all outputs.  x ** 2 and x ** (1/2) are written as products
 * and sqrt. */
void oracle_diag_secondary(const oracle_cfg *c, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    PAR_FOR
    for (int i = 0; i < nx + 2; i++)
        for (int j = 0; j < ny + 2; j++) {
            for (int k = 0; k < nzs; k++) {
                const real pairvb = 100000.0 * pow(AT(f->PVTFVB, i, j, k, 0, nzs), 1.0 / CON_KAPPA);
                const real tairvb = AT(f->POTTVB, i, j, k, 0, nzs) * AT(f->PVTFVB, i, j, k, 0, nzs);
                AT(f->PAIRVB, i, j, k, 0, nzs) = pairvb;
                AT(f->TAIRVB, i, j, k, 0, nzs) = tairvb;
                AT(f->RHOVB, i, j, k, 0, nzs) = pairvb / (CON_RD * tairvb);
            }
            for (int k = 0; k < nz; k++) {
                const real tair = AT(f->POTT, i, j, k, 0, nz) * AT(f->PVTF, i, j, k, 0, nz);
                const real pair = 100000.0 * pow(AT(f->PVTF, i, j, k, 0, nz), 1.0 / CON_KAPPA);
                AT(f->TAIR, i, j, k, 0, nz) = tair;
                AT(f->PAIR, i, j, k, 0, nz) = pair;
                AT(f->RHO, i, j, k, 0, nz) = pair / (CON_RD * tair);
                const real wx = (AT(f->UWIND, i, j, k, 0, nz) + AT(f->UWIND, i + 1, j, k, 0, nz)) / 2.0;
                const real wy = (AT(f->VWIND, i, j, k, 1, nz) + AT(f->VWIND, i, j + 1, k, 1, nz)) / 2.0;
                AT(f->WINDX, i, j, k, 0, nz) = wx;
                AT(f->WINDY, i, j, k, 0, nz) = wy;
                AT(f->WIND, i, j, k, 0, nz) = sqrt(wx * wx + wy * wy);
            }
        }
}

/* turb_compute.py:28-48 (constants), :54-142 (formulae), :191-208 (loop) */
void oracle_compute_turbulence(const oracle_cfg *c, oracle_fields *f)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz, nzs = nz + 1;
    const int nthr = nthreads_of(c);
    (void)nthr;
    const real Ri_c = 1.0, free_mix_len = 200.0, con_k = 0.35, con_Pr = 0.72;
    const real min_wind_diff = 0.0001, min_KMOM = 0.000001, max_KMOM = 0.01;
    PAR_FOR
    for (int i = 0; i < nx + 2; i++)
        for (int j = 0; j < ny + 2; j++)
            for (int k = 1; k < nzs - 1; k++) {
                real wx_km = AT(f->WINDX, i, j, k - 1, 0, nz), wx_kp = AT(f->WINDX, i, j, k, 0, nz);
                real wy_km = AT(f->WINDY, i, j, k - 1, 0, nz), wy_kp = AT(f->WINDY, i, j, k, 0, nz);
                if (wx_km == wx_kp) wx_km += min_wind_diff;
                if (wy_km == wy_kp) wy_km += min_wind_diff;
                const real alt_k = AT(f->PHIVB, i, j, k, 0, nzs) / CON_G;
                const real alt_km = AT(f->PHI, i, j, k - 1, 0, nz) / CON_G, alt_kp = AT(f->PHI, i, j, k, 0, nz) / CON_G;
                const real hgt = alt_k - A2(f->HSURF, i, j);
                const real mix_len = con_k * hgt / (1.0 + con_k * hgt / free_mix_len);
                const real qv_k = varvb_log(AT(f->QV, i, j, k, 0, nz), AT(f->QV, i, j, k - 1, 0, nz));
                const real pott_k = AT(f->POTTVB, i, j, k, 0, nzs);
                const real pott_v = pott_k * (1.0 + qv_k / 0.622) / (1.0 + qv_k);   /* misc_meteo_utilities.py:38-52 */
                const real dwx = wx_km - wx_kp, dwy = wy_km - wy_kp, dz = alt_km - alt_kp;
                const real ri_b = (CON_G / pott_v * (AT(f->POTT, i, j, k - 1, 0, nz) - AT(f->POTT, i, j, k, 0, nz)) * dz) /
                                    (dwx * dwx + dwy * dwy);
                const real shear = sqrt((dwx / dz) * (dwx / dz) + (dwy / dz) * (dwy / dz));
                real kmom = mix_len * mix_len * shear * (Ri_c - ri_b) / Ri_c;
                if (kmom < min_KMOM) kmom = min_KMOM;
                if (kmom > max_KMOM) kmom = max_KMOM;
                AT(f->KMOM, i, j, k, 0, nzs) = kmom;
                AT(f->KHEAT, i, j, k, 0, nzs) = kmom / con_Pr;
            }
}

/* ------------------------------------------------------------------------- */
/* dyn_org_discretizations.py:89-117 */
void oracle_stage_continuity(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt)
{
    const int nzs = c->nz + 1;
    oracle_continuity(c, g, f, dt);
    oracle_exchange_bc(c, f->UFLX, 1, 0, c->nz);
    oracle_exchange_bc(c, f->VFLX, 0, 1, c->nz);
    oracle_exchange_bc(c, f->WWIND, 0, 0, nzs);
    oracle_exchange_bc(c, f->COLP_NEW, 0, 0, 1);
}

/* dyn_org_discretizations.py:121-249 */
void oracle_stage_momentum(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    const int nzs = c->nz + 1;
    if (c->i_turb) oracle_exchange_bc(c, f->KMOM, 0, 0, nzs);
    oracle_uvflx_prepare(c, g, f);
    if (c->i_turb) {
        oracle_exchange_bc(c, f->KMOM_dUWINDdz, 1, 0, nzs);
        oracle_exchange_bc(c, f->KMOM_dVWINDdz, 0, 1, nzs);
        oracle_exchange_bc(c, f->SMOMXFLX, 0, 0, 1);
        oracle_exchange_bc(c, f->SMOMYFLX, 0, 0, 1);
    }
    oracle_uflx_tendency(c, g, f);
    oracle_vflx_tendency(c, g, f);
}

/* dyn_org_discretizations.py:253-273 */
void oracle_stage_temperature(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    oracle_pott_tendency(c, g, f);
}

/* dyn_org_discretizations.py:277-294 */
void oracle_stage_moisture(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    if (c->i_moist) oracle_moist_tendency(c, g, f);
}

/* dyn_org_discretizations.py:359-393 */
void oracle_stage_euler_forward(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt)
{
    oracle_make_timestep(c, g, f, dt);
    oracle_exchange_bc(c, f->POTT, 0, 0, c->nz);
    oracle_exchange_bc(c, f->VWIND, 0, 1, c->nz);
    oracle_exchange_bc(c, f->UWIND, 1, 0, c->nz);
    if (c->i_moist) {
        oracle_exchange_bc(c, f->QV, 0, 0, c->nz);
        oracle_exchange_bc(c, f->QC, 0, 0, c->nz);
    }
}

/* dyn_org_discretizations.py:310-326 */
void oracle_stage_primary_diag(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f)
{
    oracle_diag_pvtf(c, g, f);
    oracle_diag_phi(c, f);
    oracle_diag_pottvb(c, f);
}

/* dyn_matsuno.py:28-131 */
void oracle_step_matsuno(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f,
                         double dt, int nsteps, int with_rho)
{
    const int nx = c->nx, ny = c->ny, nz = c->nz;
    const size_t n2 = (size_t)(nx + 2) * (ny + 2);
    for (int s = 0; s < nsteps; s++) {
        if (with_rho) oracle_diag_rho(c, f);                        /* solver.py:99-101 */
        memcpy(f->COLP_OLD, f->COLP, n2 * sizeof(real));          /* :34-41 */
        memcpy(f->UWIND_OLD, f->UWIND, (size_t)(nx + 3) * (ny + 2) * nz * sizeof(real));
        memcpy(f->VWIND_OLD, f->VWIND, (size_t)(nx + 2) * (ny + 3) * nz * sizeof(real));
        memcpy(f->POTT_OLD, f->POTT, n2 * nz * sizeof(real));
        if (c->i_moist) {
            memcpy(f->QV_OLD, f->QV, n2 * nz * sizeof(real));
            memcpy(f->QC_OLD, f->QC, n2 * nz * sizeof(real));
        }
        for (int sub = 0; sub < 2; sub++) {
            oracle_stage_continuity(c, g, f, dt);                   /* dyn_tendencies.py:25-70 */
            oracle_stage_momentum(c, g, f);
            oracle_stage_temperature(c, g, f);
            oracle_stage_moisture(c, g, f);
            memcpy(f->COLP, f->COLP_NEW, n2 * sizeof(real));      /* :64-67 */
            oracle_stage_euler_forward(c, g, f, dt);                /* :75-77 */
            oracle_stage_primary_diag(c, g, f);                     /* :86-88 */
        }
    }
}
