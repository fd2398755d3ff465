// kernels_stage.h -- one HIP kernel per reference kernel of the Matsuno step.
//
// Thread mapping (all kernels): lane <-> longitude i (64 consecutive i per wavefront, so
// every global access is a coalesced 512-B row segment), threadIdx.y <-> latitude row j,
// the k loop runs inside the thread with the k+-1 neighbours carried in registers.
// These "stage" kernels read their horizontal neighbours straight from global memory
// (L1/L2 absorb the 3x3 reuse); they back the per-stage C ABI and the parity tests.
// The fused fast path used by dycore_step_matsuno lives in kernels_fused.h.
#pragma once
#include "dycore_common.h"

namespace dyc {

// Workgroup shape of the column kernels: TI consecutive longitudes x TJ rows.  128 x 2 (or 256 x 1) makes
// every workgroup touch one contiguous 2-KB segment per field and level, which the HBM likes
// better than 64 x 4 (four 512-B segments 11.8 KB apart): +14 % in tests/microbench/bw.hip.
#ifndef DYC_TI
#define DYC_TI 128
#endif
#ifndef DYC_TJ
#define DYC_TJ 2
#endif
constexpr int TI = DYC_TI;   // lanes along i
constexpr int TJ = DYC_TJ;   // rows per workgroup

#define DYC_IJ(i0, j0)                                   \
    const int i = blockIdx.x * TI + threadIdx.x + (i0);  \
    const int j = blockIdx.y * TJ + threadIdx.y + (j0);

// Thread -> owned cell (i in [1,nx], j in [1,ny]) of the column kernels of the fused step
// (1-D launch of 8*ceil(n/8) workgroups of TI x TJ threads, n from cells_blocks()).
//  * XCD-aware order: hardware deals workgroup b to XCD b % 8; every XCD gets a contiguous run of
//    the row-major block order, so the rows j-1 / j+1 a workgroup reads beyond its own are the
//    ones its neighbours on the SAME XCD (same L2) fetch at about the same time.  Dealt round-robin
//    those neighbours always sit on other XCDs and POTT / VFLX come from HBM two to three times.
//  * lin = 0: blocks are TI x TJ tiles.  lin != 0: blocks are runs of TI*TJ cells of the row-major
//    cell order -- no idle lanes at the row ends, so a latitude band of 1440 x 90 columns is 507
//    workgroups instead of 540 and fits two per CU on 256 CUs.  Used when nx is a multiple of 32,
//    so that a wavefront's row segment still starts on a 256-B boundary.
// lin: bit 0 = cell runs instead of tiles, bit 1 = XCD-aware block order
__device__ __forceinline__ bool owned_cell(const Geo &g, int lin, int &i, int &j)
{
    const int chunk = gridDim.x / 8;
    const int b = (lin & 2) ? (blockIdx.x % 8) * chunk + blockIdx.x / 8 : blockIdx.x;
    if (lin & 1) {
        const long long t = (long long)b * (TI * TJ) + threadIdx.y * TI + threadIdx.x;
        if (t >= (long long)g.nx * g.ny) return false;
        j = (int)(t / g.nx);
        i = (int)(t - (long long)j * g.nx) + 1;
        j += 1;
        return true;
    }
    const int gbx = (g.nx + TI - 1) / TI;
    const int by = b / gbx, bx = b - by * gbx;
    i = bx * TI + threadIdx.x + 1;
    j = by * TJ + threadIdx.y + 1;
    return i <= g.nx && j <= g.ny;
}

// a tendency term of namelist.py:62-88 is on unless its DYCORE_TERM_* bit is set (wave-uniform: g is a kernel argument)
__device__ __forceinline__ bool term_on(const Geo &g, int bit) { return (g.term_off & bit) == 0; }

// ------------------------------------------------------------------------------------
// continuity: dyn_continuity.py:170-232.  Owned cells i in [1,nx], j in [1,ny]; the lateral
// halo of UFLX, VFLX, WWIND, COLP_NEW (dyn_org_discretizations.py:107-117) is refreshed by
// one k_halo_multi launch afterwards.
__global__ __launch_bounds__(TI *TJ) void k_continuity(Geo g, Ptrs p, real dt)
{
    DYC_IJ(1, 1)
    if (i > g.nx || j > g.ny) return;
    const size_t c = at2(g, i, j);
    const real colp = p.COLP[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1];
    const real c_jm1 = p.COLP[c - g.P], c_jp1 = p.COLP[c + g.P];
    const real dy_i = p.dyis[c], dy_ip1 = p.dyis[c + 1];
    const real dx_j = p.dxjs[c], dx_jp1 = p.dxjs[c + g.P];
    const real area = p.A[c];
    real sum = 0.0;
#ifdef DYC_CONT_UNROLL
#pragma unroll DYC_CONT_UNROLL
#endif
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real uflx_i = (c_im1 + colp) / 2.0 * p.UWIND[q] * dy_i;
        const real uflx_ip1 = (colp + c_ip1) / 2.0 * p.UWIND[q + 1] * dy_ip1;
        const real vflx_j = (c_jm1 + colp) / 2.0 * p.VWIND[q] * dx_j;
        const real vflx_jp1 = (colp + c_jp1) / 2.0 * p.VWIND[q + g.P] * dx_jp1;
        p.UFLX[at(g, i, j, k)] = uflx_i;
        p.VFLX[at(g, i, j, k)] = vflx_j;
        const real div = (uflx_ip1 - uflx_i + vflx_jp1 - vflx_j) * p.dsigma[k] / area;
        p.FLXDIV[q] = div;
        sum += div;
    }
    const real dcolpdt = term_on(g, DYCORE_TERM_COLP_MAIN) ? -sum : (real)0.0;       // dyn_continuity.py:211-214
    const real colp_new = p.COLP_OLD[c] + dt * dcolpdt;
    p.dCOLPdt[c] = dcolpdt;
    p.COLP_NEW[at(g, i, j, 0)] = colp_new;
    // sigma-dot vanishes at the model top and at the surface (io_initial_conditions.py:45-46)
    p.WWIND[at(g, i, j, 0)] = 0.0;
    p.WWIND[at(g, i, j, g.nz)] = 0.0;
    real run = p.FLXDIV[c];
    for (int k = 1; k < g.nz; k++) {
        const real w = -run / colp_new - p.sigma_vb[k] * dcolpdt / colp_new;
        p.WWIND[at(g, i, j, k)] = w;
        run += p.FLXDIV[c + (size_t)k * g.plane];
    }
}

// ------------------------------------------------------------------------------------
// momentum-flux preparation: dyn_UVFLX_prepare.py:249-439.
// Threads cover is in [1,nx+1], js in [1,ny+1]; each output keeps its own staggered range.
// A band that does not touch the south wall also produces R, S, T on its halo row j = 0
// (dVFLXdt at its first v-row reads them at j-1); nothing else is computed there.
template <bool TURB>
__global__ __launch_bounds__(TI *TJ) void k_prepare(Geo g, Ptrs p)
{
    DYC_IJ(1, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const int P = g.P;
    const size_t c = at2(g, i, j);
    // dyn_UVFLX_prepare.py:260 / :299 / :356: each group of outputs only with its term (a field that is not computed keeps
    // what it held, as in the reference)
    const bool vadv = term_on(g, DYCORE_TERM_UVFLX_VERT_ADV), hadv = term_on(g, DYCORE_TERM_UVFLX_HOR_ADV);
    if (j == 0) {
        if (g.pole_s || i > g.nx + 1 || !hadv) return;
        for (int k = 0; k < g.nz; k++) {
            const size_t q = c + (size_t)k * g.plane;
            const real vs = p.VFLX[q - 1] + p.VFLX[q + P - 1] + p.VFLX[q] + p.VFLX[q + P];
            p.SFLX[q] = 1.0 / 24.0 * (vs + p.UFLX[q - 1] + 2.0 * p.UFLX[q] + p.UFLX[q + 1]);
            p.TFLX[q] = 1.0 / 24.0 * (vs - p.UFLX[q - 1] - 2.0 * p.UFLX[q] - p.UFLX[q + 1]);
            if (i <= g.nx)
                p.RFLX[q] = 1.0 / 12.0 * (p.VFLX[q - 1] + p.VFLX[q + P - 1] + 2.0 * (p.VFLX[q] + p.VFLX[q + P]) +
                                          p.VFLX[q + 1] + p.VFLX[q + P + 1]);
        }
        return;
    }
    const int jg = j + g.joff;
    const bool u_ok = (j <= g.ny);               // WWIND_UWIND, KMOM_dUWINDdz, S, T  rows
    const bool v_ok = (i <= g.nx);               // WWIND_VWIND, KMOM_dVWINDdz, D, E  columns
    const bool lo = (jg == 1), hi = (jg == g.ny_global);

    // COLP_NEW*A at the nine points (2-D, reused for every level)
    real ca[3][3];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int ii = i + a - 1, jj = j + b - 1;
            const bool in = (ii <= g.nx + 1) && (jj <= g.ny + 1);
            ca[b][a] = in ? p.COLP_NEW[at2(g, ii, jj)] * p.A[at2(g, ii, jj)] : 0.0;
        }
    if (u_ok && vadv) { p.WWIND_UWIND[c] = 0.0; p.WWIND_UWIND[c + (size_t)g.nz * g.plane] = 0.0; }
    if (v_ok && vadv) { p.WWIND_VWIND[c] = 0.0; p.WWIND_VWIND[c + (size_t)g.nz * g.plane] = 0.0; }
    if (TURB) {
        // k = 0 and k = nz: never consumed (dyn_UFLX.py:158-167); the reference returns an
        // unassigned local there (dyn_functions.py:310/377) -> 0
        if (u_ok) { p.KMOM_dUWINDdz[at(g, i, j, 0)] = 0.0; p.KMOM_dUWINDdz[at(g, i, j, g.nz)] = 0.0; }
        if (v_ok) { p.KMOM_dVWINDdz[at(g, i, j, 0)] = 0.0; p.KMOM_dVWINDdz[at(g, i, j, g.nz)] = 0.0; }
    }

    real u_km1 = 0.0, v_km1 = 0.0;
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        // ---- interface quantities (k = 1..nz-1) ----
        const real u_k = u_ok ? p.UWIND[q] : 0.0;
        const real v_k = v_ok ? p.VWIND[q] : 0.0;
        if (k >= 1) {
            const real ds = p.dsigma[k], ds_km1 = p.dsigma[k - 1];
            // W(a,b) = COLP_NEW*A*WWIND at (i+a-1, j+b-1)
#define WT(a, b) (ca[b][a] * p.WWIND[q + ((b) - 1) * P + ((a) - 1)])
#define WT2(a, b) (2.0 * ca[b][a] * p.WWIND[q + ((b) - 1) * P + ((a) - 1)])
            if (u_ok && vadv) {   // dyn_functions.py:211-270, d = i, p = j, rigid wall at the poles
                real s;
                if (lo) s = 0.25 * (WT(0, 2) + WT(1, 2) + WT(0, 1) + WT(1, 1));
                else if (hi) s = 0.25 * (WT(0, 1) + WT(1, 1) + WT(0, 0) + WT(1, 0));
                else s = 0.125 * (WT(0, 2) + WT(1, 2) + WT2(0, 1) + WT2(1, 1) + WT(0, 0) + WT(1, 0));
                const real u_ks = (ds * u_km1 + ds_km1 * u_k) / (ds + ds_km1);
                p.WWIND_UWIND[q] = s * u_ks;
            }
            if (v_ok && vadv) {   // d = j, p = i, no wall
                const real s = 0.125 * (WT(2, 0) + WT(2, 1) + WT2(1, 0) + WT2(1, 1) + WT(0, 0) + WT(0, 1));
                const real v_ks = (ds * v_km1 + ds_km1 * v_k) / (ds + ds_km1);
                p.WWIND_VWIND[q] = s * v_ks;
            }
#undef WT
#undef WT2
            if (TURB) {   // dyn_functions.py:276-377 (uses COLP, not COLP_NEW: dyn_UVFLX_prepare.py:318-320)
#define KT(a, b) (p.RHOVB[q + ((b) - 1) * P + ((a) - 1)] * p.COLP[c + ((b) - 1) * P + ((a) - 1)] * p.A[c + ((b) - 1) * P + ((a) - 1)] * p.KMOM[q + ((b) - 1) * P + ((a) - 1)])
#define KT2(a, b) (2.0 * p.RHOVB[q + ((b) - 1) * P + ((a) - 1)] * p.COLP[c + ((b) - 1) * P + ((a) - 1)] * p.A[c + ((b) - 1) * P + ((a) - 1)] * p.KMOM[q + ((b) - 1) * P + ((a) - 1)])
#define PH(a, b) p.PHI[q + ((b) - 1) * P + ((a) - 1)]
#define PHM(a, b) p.PHI[q - g.plane + ((b) - 1) * P + ((a) - 1)]
                if (u_ok) {
                    real k6, alt_km1, alt;
                    if (lo) {
                        k6 = 0.25 * (KT(0, 2) + KT(1, 2) + KT(0, 1) + KT(1, 1));
                        alt_km1 = 0.25 * (PHM(0, 2) + PHM(1, 2) + PHM(0, 1) + PHM(1, 1)) / CON_G;
                        alt = 0.25 * (PH(0, 2) + PH(1, 2) + PH(0, 1) + PH(1, 1)) / CON_G;
                    } else if (hi) {
                        k6 = 0.25 * (KT(0, 1) + KT(1, 1) + KT(0, 0) + KT(1, 0));
                        alt_km1 = 0.25 * (PHM(0, 1) + PHM(1, 1) + PHM(0, 0) + PHM(1, 0)) / CON_G;
                        alt = 0.25 * (PH(0, 1) + PH(1, 1) + PH(0, 0) + PH(1, 0)) / CON_G;
                    } else {
                        k6 = 0.125 * (KT(0, 2) + KT(1, 2) + KT2(0, 1) + KT2(1, 1) + KT(0, 0) + KT(1, 0));
                        alt_km1 = 0.125 * (PHM(0, 2) + PHM(1, 2) + 2.0 * PHM(0, 1) + 2.0 * PHM(1, 1) + PHM(0, 0) + PHM(1, 0)) / CON_G;
                        alt = 0.125 * (PH(0, 2) + PH(1, 2) + 2.0 * PH(0, 1) + 2.0 * PH(1, 1) + PH(0, 0) + PH(1, 0)) / CON_G;
                    }
                    p.KMOM_dUWINDdz[at(g, i, j, k)] = k6 * ((u_km1 - u_k) / (alt_km1 - alt));
                }
                if (v_ok) {
                    const real k6 = 0.125 * (KT(2, 0) + KT(2, 1) + KT2(1, 0) + KT2(1, 1) + KT(0, 0) + KT(0, 1));
                    const real alt_km1 = 0.125 * (PHM(2, 0) + PHM(2, 1) + 2.0 * PHM(1, 0) + 2.0 * PHM(1, 1) + PHM(0, 0) + PHM(0, 1)) / CON_G;
                    const real alt = 0.125 * (PH(2, 0) + PH(2, 1) + 2.0 * PH(1, 0) + 2.0 * PH(1, 1) + PH(0, 0) + PH(0, 1)) / CON_G;
                    p.KMOM_dVWINDdz[at(g, i, j, k)] = k6 * ((v_km1 - v_k) / (alt_km1 - alt));
                }
#undef KT
#undef KT2
#undef PH
#undef PHM
            }
        }
        u_km1 = u_k;
        v_km1 = v_k;
        if (!hadv) continue;
        // ---- the eight Arakawa-C fluxes (dyn_functions.py:429-536) ----
        // UF/VF(a,b) at (i+a-1, j+b-1); loads outside a field's extent are not issued
#define UF(a, b) p.UFLX[q + ((b) - 1) * P + ((a) - 1)]
#define VF(a, b) p.VFLX[q + ((b) - 1) * P + ((a) - 1)]
        {   // (is,js): C, Q
            p.CFLX[q] = 1.0 / 12.0 * (VF(0, 0) + VF(1, 0) + 2.0 * (VF(0, 1) + VF(1, 1)) + VF(0, 2) + VF(1, 2));
            p.QFLX[q] = 1.0 / 12.0 * (UF(0, 0) + UF(0, 1) + 2.0 * (UF(1, 0) + UF(1, 1)) + UF(2, 0) + UF(2, 1));
        }
        if (v_ok) {   // (i,js): D, E
            const real vs = VF(1, 0) + 2.0 * VF(1, 1) + VF(1, 2);
            p.DFLX[q] = 1.0 / 24.0 * (vs + UF(1, 0) + UF(1, 1) + UF(2, 0) + UF(2, 1));
            p.EFLX[q] = 1.0 / 24.0 * (vs - UF(1, 0) - UF(1, 1) - UF(2, 0) - UF(2, 1));
        }
        if (u_ok) {   // (is,j): S, T
            const real vs = VF(0, 1) + VF(0, 2) + VF(1, 1) + VF(1, 2);
            p.SFLX[q] = 1.0 / 24.0 * (vs + UF(0, 1) + 2.0 * UF(1, 1) + UF(2, 1));
            p.TFLX[q] = 1.0 / 24.0 * (vs - UF(0, 1) - 2.0 * UF(1, 1) - UF(2, 1));
        }
        if (u_ok && v_ok) {   // (i,j): B, R
            p.BFLX[q] = 1.0 / 12.0 * (UF(1, 0) + UF(2, 0) + 2.0 * (UF(1, 1) + UF(2, 1)) + UF(1, 2) + UF(2, 2));
            p.RFLX[q] = 1.0 / 12.0 * (VF(0, 1) + VF(0, 2) + 2.0 * (VF(1, 1) + VF(1, 2)) + VF(2, 1) + VF(2, 2));
        }
#undef UF
#undef VF
    }
}

// ------------------------------------------------------------------------------------
// dyn_functions.py:383-424 (divides by g itself)
__device__ __forceinline__ real var_ds(real v0, real v_dm1, real v_pm1, real v_pp1,
                                         real v_pm1_dm1, real v_pp1_dm1, bool lo, bool hi)
{
    if (lo) return 0.25 * (v_pp1_dm1 + v_pp1 + v_dm1 + v0) / CON_G;
    if (hi) return 0.25 * (v_dm1 + v0 + v_pm1_dm1 + v_pm1) / CON_G;
    return 0.125 * (v_pp1_dm1 + v_pp1 + 2.0 * v_dm1 + 2.0 * v0 + v_pm1_dm1 + v_pm1) / CON_G;
}

// dyn_functions.py:177-207
__device__ __forceinline__ real pre_grad(real phi, real phi_dm1, real colp, real colp_dm1,
                                           real pott, real pott_dm1, real pvtf, real pvtf_dm1,
                                           real pvtfvb, real pvtfvb_dm1, real pvtfvb_dm1_kp1,
                                           real pvtfvb_kp1, real dsig, real svb, real svb_kp1,
                                           real dgrid)
{
    return -dgrid * ((phi - phi_dm1) * (colp + colp_dm1) / 2.0 +
                     (colp - colp_dm1) * CON_CP / 2.0 *
                         (pott_dm1 / dsig * (svb_kp1 * (pvtfvb_dm1_kp1 - pvtf_dm1) + svb * (pvtf_dm1 - pvtfvb_dm1)) +
                          pott / dsig * (svb_kp1 * (pvtfvb_kp1 - pvtf) + svb * (pvtf - pvtfvb))));
}

// ------------------------------------------------------------------------------------
// dUFLXdt: dyn_UFLX.py:339-434 (loop) and :69-199 (terms).  Owned is in [1,nx], j in [1,ny];
// column nx+1 is the periodic duplicate the reference computes from NaN halos and discards.
template <bool TURB>
__global__ __launch_bounds__(TI *TJ) void k_uflx_tendency(Geo g, Ptrs p)
{
    DYC_IJ(1, 1)
    if (i > g.nx || j > g.ny) return;
    const int P = g.P;
    const size_t c = at2(g, i, j);
    const int jg = j + g.joff;
    const bool lo = (jg == 1), hi = (jg == g.ny_global);
    const int dm = (i == 1) ? (g.nx - 1) : -1;        // offset to the i-1 flux column (dyn_UFLX.py:366-374)
    const real colp = p.COLP[c], colp_im1 = p.COLP[c - 1];
    const real corf = p.corf_is[c], lat = p.lat_is_rad[c];
    const real coslat = cos(lat), sinlat = sin(lat);
    const real dlon = p.dlon_rad[c], dlat = p.dlat_rad[c], dyis = p.dyis[c];
    real smom_is = 0.0;
    if (TURB)
        smom_is = var_ds(p.SMOMXFLX[c], p.SMOMXFLX[c - 1], p.SMOMXFLX[c - P], p.SMOMXFLX[c + P],
                         p.SMOMXFLX[c - P - 1], p.SMOMXFLX[c + P - 1], lo, hi);
    real wu_k = p.WWIND_UWIND[c];
    real pvb_k = p.PVTFVB[c], pvb_im1_k = p.PVTFVB[c - 1];
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        real bflx = p.BFLX[q], bflx_im1 = p.BFLX[q + dm];
        real cflx = p.CFLX[q], cflx_jp1 = p.CFLX[q + P];
        real dflx_im1 = p.DFLX[q + dm], dflx_jp1 = p.DFLX[q + P];
        real eflx = p.EFLX[q], eflx_im1_jp1 = p.EFLX[q + P + dm];
        if (lo) { dflx_im1 = 0.0; cflx = 0.0; eflx = 0.0; }
        if (hi) { dflx_jp1 = 0.0; cflx_jp1 = 0.0; eflx_im1_jp1 = 0.0; }
#define U(a, b) p.UWIND[q + (b) * P + (a)]
#define V(a, b) p.VWIND[q + (b) * P + (a)]
        const real u = U(0, 0), u_im1 = U(-1, 0), u_ip1 = U(1, 0);
        real t = 0.0;
        if (term_on(g, DYCORE_TERM_UVFLX_HOR_ADV))
        t = t + (+bflx_im1 * (u_im1 + u) / 2.0 - bflx * (u + u_ip1) / 2.0
                 + cflx * (U(0, -1) + u) / 2.0 - cflx_jp1 * (u + U(0, 1)) / 2.0
                 + dflx_im1 * (U(-1, -1) + u) / 2.0 - dflx_jp1 * (u + U(1, 1)) / 2.0
                 + 1.0 * (+eflx * (U(1, -1) + u) / 2.0 - eflx_im1_jp1 * (u + U(-1, 1)) / 2.0));
        const real wu_kp1 = p.WWIND_UWIND[q + g.plane];
        if (term_on(g, DYCORE_TERM_UVFLX_VERT_ADV)) t = t + (wu_k - wu_kp1) / p.dsigma[k];
        wu_k = wu_kp1;
        if (TURB) {
#define PV(a, b, kk) p.PHIVB[q + (size_t)(kk) * g.plane + (b) * P + (a)]
#define RH(a, b) p.RHO[q + (b) * P + (a)]
            const real altvb = var_ds(PV(0, 0, 0), PV(-1, 0, 0), PV(0, -1, 0), PV(0, 1, 0), PV(-1, -1, 0), PV(-1, 1, 0), lo, hi) / CON_G;
            const real altvb_kp1 = var_ds(PV(0, 0, 1), PV(-1, 0, 1), PV(0, -1, 1), PV(0, 1, 1), PV(-1, -1, 1), PV(-1, 1, 1), lo, hi) / CON_G;
            const real rho_is = var_ds(RH(0, 0), RH(-1, 0), RH(0, -1), RH(0, 1), RH(-1, -1), RH(-1, 1), lo, hi);
            const real kd = p.KMOM_dUWINDdz[q], kd_kp1 = p.KMOM_dUWINDdz[q + g.plane];
            real turb;
            if (k == 0) turb = (0.0 - kd_kp1) / ((altvb - altvb_kp1) * rho_is);
            else if (k == g.nz - 1) turb = (kd + smom_is) / ((altvb - altvb_kp1) * rho_is);
            else turb = (kd - kd_kp1) / ((altvb - altvb_kp1) * rho_is);
            t = t + turb;
            p.dUFLXdt_TURB[q] = turb;
#undef PV
#undef RH
        }
        // Coriolis + metric: dyn_UFLX.py:39-66
        if (term_on(g, DYCORE_TERM_UVFLX_CORIOLIS))
        t = t + CON_RE * dlon * dlat / 2.0 *
                    (colp_im1 * (V(-1, 0) + V(-1, 1)) / 2.0 * (corf * CON_RE * coslat + (u_im1 + u) / 2.0 * sinlat) +
                     colp * (V(0, 0) + V(0, 1)) / 2.0 * (corf * CON_RE * coslat + (u + u_ip1) / 2.0 * sinlat));
        const real pvb_kp1 = p.PVTFVB[q + g.plane], pvb_im1_kp1 = p.PVTFVB[q + g.plane - 1];
        if (term_on(g, DYCORE_TERM_UVFLX_PRE_GRAD))
        t = t + pre_grad(p.PHI[q], p.PHI[q - 1], colp, colp_im1, p.POTT[q], p.POTT[q - 1],
                         p.PVTF[q], p.PVTF[q - 1], pvb_k, pvb_im1_k, pvb_im1_kp1, pvb_kp1,
                         p.dsigma[k], p.sigma_vb[k], p.sigma_vb[k + 1], dyis);
        pvb_k = pvb_kp1;
        pvb_im1_k = pvb_im1_kp1;
        const real dif = p.UVFLX_dif_coef[k];
        if (dif > 0.0 && term_on(g, DYCORE_TERM_UVFLX_NUM_DIF))
            t = t + dif * (p.UFLX[q - 1] + p.UFLX[q + 1] + p.UFLX[q - P] + p.UFLX[q + P] - 4.0 * p.UFLX[q]);
        p.dUFLXdt[q] = term_on(g, DYCORE_TERM_UVFLX_MAIN) ? t : (real)0.0;       // dyn_UFLX.py:115-117
#undef U
#undef V
    }
}

// dVFLXdt: dyn_VFLX.py:322-408 and :67-198.  Owned i in [1,nx], js in [2,ny] globally (the
// pole rows js = 1, nys read NaN halos in the reference and VWIND is zeroed there).
template <bool TURB>
__global__ __launch_bounds__(TI *TJ) void k_vflx_tendency(Geo g, Ptrs p, int jlo, int jhi)
{
    DYC_IJ(1, 0)
    const int jj = j + jlo;
    if (i > g.nx || jj > jhi) return;
    const int P = g.P;
    const size_t c = at2(g, i, jj);
    const int dp = (i == g.nx) ? -(g.nx - 1) : 1;       // offset to the i+1 flux column (dyn_VFLX.py:349-356)
    const real colp = p.COLP[c], colp_jm1 = p.COLP[c - P];
    const real corf = p.corf[c], corf_jm1 = p.corf[c - P];
    const real lat = p.lat_rad[c], lat_jm1 = p.lat_rad[c - P];
    const real coslat = cos(lat), sinlat = sin(lat), coslat_jm1 = cos(lat_jm1), sinlat_jm1 = sin(lat_jm1);
    const real dlon = p.dlon_rad[c], dlat = p.dlat_rad[c], dxjs = p.dxjs[c];
    real smom_js = 0.0;
    if (TURB)
        smom_js = var_ds(p.SMOMYFLX[c], p.SMOMYFLX[c - P], p.SMOMYFLX[c - 1], p.SMOMYFLX[c + 1],
                         p.SMOMYFLX[c - P - 1], p.SMOMYFLX[c - P + 1], false, false);
    real wv_k = p.WWIND_VWIND[c];
    real pvb_k = p.PVTFVB[c], pvb_jm1_k = p.PVTFVB[c - P];
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real rflx = p.RFLX[q], rflx_jm1 = p.RFLX[q - P];
        const real qflx = p.QFLX[q], qflx_ip1 = p.QFLX[q + dp];
        const real sflx_jm1 = p.SFLX[q - P], sflx_ip1 = p.SFLX[q + dp];
        const real tflx = p.TFLX[q], tflx_ip1_jm1 = p.TFLX[q - P + dp];
#define U(a, b) p.UWIND[q + (b) * P + (a)]
#define V(a, b) p.VWIND[q + (b) * P + (a)]
        const real v = V(0, 0);
        real t = 0.0;
        if (term_on(g, DYCORE_TERM_UVFLX_HOR_ADV))
        t = t + (+rflx_jm1 * (V(0, -1) + v) / 2.0 - rflx * (v + V(0, 1)) / 2.0
                 + qflx * (V(-1, 0) + v) / 2.0 - qflx_ip1 * (v + V(1, 0)) / 2.0
                 + sflx_jm1 * (V(-1, -1) + v) / 2.0 - sflx_ip1 * (v + V(1, 1)) / 2.0
                 + -1.0 * (+tflx * (V(-1, 1) + v) / 2.0 - tflx_ip1_jm1 * (v + V(1, -1)) / 2.0));
        const real wv_kp1 = p.WWIND_VWIND[q + g.plane];
        if (term_on(g, DYCORE_TERM_UVFLX_VERT_ADV)) t = t + (wv_k - wv_kp1) / p.dsigma[k];
        wv_k = wv_kp1;
        if (TURB) {
#define PV(a, b, kk) p.PHIVB[q + (size_t)(kk) * g.plane + (b) * P + (a)]
#define RH(a, b) p.RHO[q + (b) * P + (a)]
            const real altvb = var_ds(PV(0, 0, 0), PV(0, -1, 0), PV(-1, 0, 0), PV(1, 0, 0), PV(-1, -1, 0), PV(1, -1, 0), false, false) / CON_G;
            const real altvb_kp1 = var_ds(PV(0, 0, 1), PV(0, -1, 1), PV(-1, 0, 1), PV(1, 0, 1), PV(-1, -1, 1), PV(1, -1, 1), false, false) / CON_G;
            const real rho_js = var_ds(RH(0, 0), RH(0, -1), RH(-1, 0), RH(1, 0), RH(-1, -1), RH(1, -1), false, false);
            const real kd = p.KMOM_dVWINDdz[q], kd_kp1 = p.KMOM_dVWINDdz[q + g.plane];
            real turb;
            if (k == 0) turb = (0.0 - kd_kp1) / ((altvb - altvb_kp1) * rho_js);
            else if (k == g.nz - 1) turb = (kd + smom_js) / ((altvb - altvb_kp1) * rho_js);
            else turb = (kd - kd_kp1) / ((altvb - altvb_kp1) * rho_js);
            t = t + turb;
            p.dVFLXdt_TURB[q] = turb;
#undef PV
#undef RH
        }
        // Coriolis + metric: dyn_VFLX.py:39-64
        const real ub_jm1 = (U(0, -1) + U(1, -1)) / 2.0, ub = (U(0, 0) + U(1, 0)) / 2.0;
        if (term_on(g, DYCORE_TERM_UVFLX_CORIOLIS))
        t = t + -CON_RE * dlon * dlat / 2.0 *
                    (colp_jm1 * ub_jm1 * (corf_jm1 * CON_RE * coslat_jm1 + ub_jm1 * sinlat_jm1) +
                     colp * ub * (corf * CON_RE * coslat + ub * sinlat));
        const real pvb_kp1 = p.PVTFVB[q + g.plane], pvb_jm1_kp1 = p.PVTFVB[q + g.plane - P];
        if (term_on(g, DYCORE_TERM_UVFLX_PRE_GRAD))
        t = t + pre_grad(p.PHI[q], p.PHI[q - P], colp, colp_jm1, p.POTT[q], p.POTT[q - P],
                         p.PVTF[q], p.PVTF[q - P], pvb_k, pvb_jm1_k, pvb_jm1_kp1, pvb_kp1,
                         p.dsigma[k], p.sigma_vb[k], p.sigma_vb[k + 1], dxjs);
        pvb_k = pvb_kp1;
        pvb_jm1_k = pvb_jm1_kp1;
        const real dif = p.UVFLX_dif_coef[k];
        if (dif > 0.0 && term_on(g, DYCORE_TERM_UVFLX_NUM_DIF))
            t = t + dif * (p.VFLX[q - 1] + p.VFLX[q + 1] + p.VFLX[q - P] + p.VFLX[q + P] - 4.0 * p.VFLX[q]);
        p.dVFLXdt[q] = term_on(g, DYCORE_TERM_UVFLX_MAIN) ? t : (real)0.0;
#undef U
#undef V
    }
}

// ------------------------------------------------------------------------------------
// dPOTTdt: dyn_POTT.py:180-216 and :55-110.  Owned i in [1,nx], j in [1,ny].
template <bool TURB>
__global__ __launch_bounds__(TI *TJ) void k_pott_tendency(Geo g, Ptrs p)
{
    DYC_IJ(1, 1)
    if (i > g.nx || j > g.ny) return;
    const int P = g.P;
    const size_t c = at2(g, i, j);
    const real colp = p.COLP[c], colp_new = p.COLP_NEW[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1], c_jm1 = p.COLP[c - P], c_jp1 = p.COLP[c + P];
    const real area = p.A[c];
    real surf = 0.0;
    if (TURB) surf = p.SSHFLX[c] / CON_CP;
    real w_k = 0.0, pvb_k = 0.0;     // k = 0: first term omitted (dyn_functions.py:124-127)
    real pott_km1 = 0.0, phi_km1 = 0.0;
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real pt = p.POTT[q];
        const real p_im1 = p.POTT[q - 1], p_ip1 = p.POTT[q + 1], p_jm1 = p.POTT[q - P], p_jp1 = p.POTT[q + P];
        real t = 0.0;
        if (term_on(g, DYCORE_TERM_POTT_HOR_ADV))
        t = t + (+p.UFLX[q] * (p_im1 + pt) / 2.0 - p.UFLX[q + 1] * (pt + p_ip1) / 2.0
                 + p.VFLX[q] * (p_jm1 + pt) / 2.0 - p.VFLX[q + P] * (pt + p_jp1) / 2.0) / area;
        const real w_kp1 = p.WWIND[q + g.plane], pvb_kp1 = p.POTTVB[q + g.plane];
        if (!term_on(g, DYCORE_TERM_POTT_VERT_ADV)) { }
        else if (k == 0) t = t + colp_new * (-w_kp1 * pvb_kp1) / p.dsigma[k];
        else t = t + colp_new * (+w_k * pvb_k - w_kp1 * pvb_kp1) / p.dsigma[k];
        w_k = w_kp1;
        pvb_k = pvb_kp1;
        if (TURB) {   // dyn_functions.py:26-67
            const real phi = p.PHI[q];
            const real alt = phi / CON_G;
            const real altvb = p.PHIVB[q] / CON_G, altvb_kp1 = p.PHIVB[q + g.plane] / CON_G;
            const real rho = p.RHO[q];
            real turb;
            if (k == 0) {
                const real alt_kp1 = p.PHI[q + g.plane] / CON_G;
                turb = colp * ((0.0 - ((pt - p.POTT[q + g.plane]) / (alt - alt_kp1) * p.RHOVB[q + g.plane] * p.KHEAT[q + g.plane])) /
                               ((altvb - altvb_kp1) * rho));
            } else if (k == g.nz - 1) {
                const real alt_km1 = phi_km1 / CON_G;
                turb = colp * ((((pott_km1 - pt) / (alt_km1 - alt) * p.RHOVB[q] * p.KHEAT[q]) + surf) /
                               ((altvb - altvb_kp1) * rho));
            } else {
                const real alt_kp1 = p.PHI[q + g.plane] / CON_G, alt_km1 = phi_km1 / CON_G;
                turb = colp * ((((pott_km1 - pt) / (alt_km1 - alt) * p.RHOVB[q] * p.KHEAT[q])
                                - ((pt - p.POTT[q + g.plane]) / (alt - alt_kp1) * p.RHOVB[q + g.plane] * p.KHEAT[q + g.plane])) /
                               ((altvb - altvb_kp1) * rho));
            }
            t = t + turb;
            p.dPOTTdt_TURB[q] = turb / colp * 3600.0;
            phi_km1 = phi;
        }
        pott_km1 = pt;
        const real dif = p.POTT_dif_coef[k];
        if (dif > 0.0 && term_on(g, DYCORE_TERM_POTT_NUM_DIF))
            t = t + dif * (c_im1 * p_im1 + c_ip1 * p_ip1 + c_jm1 * p_jm1 + c_jp1 * p_jp1 - 4.0 * colp * pt);
        if (g.i_rad) t = t + p.dPOTTdt_RAD[q] * colp;
        p.dPOTTdt[q] = term_on(g, DYCORE_TERM_POTT_MAIN) ? t : (real)0.0;        // dyn_POTT.py:70-72
    }
}

// ------------------------------------------------------------------------------------
// dyn_functions.py:70-95: interface value by logarithmic interpolation (Python's max(VAR,
// min_val): the first argument unless the second is greater)
__device__ __forceinline__ real varvb_log(real var, real var_km1)
{
    const real min_val = 0.0000001;
    if (min_val > var) var = min_val;
    if (min_val > var_km1) var_km1 = min_val;
    if (var_km1 == var) return var;
    return (log(var_km1) - log(var)) / (1.0 / var - 1.0 / var_km1);
}

// The same value with the logarithm and reciprocal of each clamped level value formed at most once: a level is the
// lower argument of one interface and the upper argument of the next (k_moist carries the cache down the column),
// and neither is formed where the two clamped values are equal (cloud water is 1e-7 = the clamp in most of the
// atmosphere).  Same operations on the same operands as varvb_log: bit-identical.
struct LogCache { real c, l, r; bool have; };
__device__ __forceinline__ LogCache log_cache(real var)
{
    const real min_val = 0.0000001;
    LogCache x;
    x.c = (min_val > var) ? min_val : var;
    x.l = 0.0; x.r = 0.0; x.have = false;
    return x;
}
__device__ __forceinline__ void log_need(LogCache &x)
{
    if (!x.have) { x.l = log(x.c); x.r = 1.0 / x.c; x.have = true; }
}
__device__ __forceinline__ real varvb_log_cached(LogCache &var, LogCache &var_km1)
{
    if (var_km1.c == var.c) return var.c;
    log_need(var);
    log_need(var_km1);
    return (var_km1.l - var.l) / (var.r - var_km1.r);
}

// dyn_functions.py:26-67
__device__ __forceinline__ real turb_flux(real phi, real phi_kp1, real phi_km1, real phivb,
                                            real phivb_kp1, real var, real var_kp1, real var_km1,
                                            real kvar, real kvar_kp1, real rho, real rhovb,
                                            real rhovb_kp1, real colp, real surf, int k, int nz)
{
    const real alt = phi / CON_G, alt_kp1 = phi_kp1 / CON_G, alt_km1 = phi_km1 / CON_G;
    const real altvb = phivb / CON_G, altvb_kp1 = phivb_kp1 / CON_G;
    if (k == 0)
        return colp * ((0.0 - ((var - var_kp1) / (alt - alt_kp1) * rhovb_kp1 * kvar_kp1)) / ((altvb - altvb_kp1) * rho));
    if (k == nz - 1)
        return colp * ((((var_km1 - var) / (alt_km1 - alt) * rhovb * kvar) + surf) / ((altvb - altvb_kp1) * rho));
    return colp * ((((var_km1 - var) / (alt_km1 - alt) * rhovb * kvar)
                    - ((var - var_kp1) / (alt - alt_kp1) * rhovb_kp1 * kvar_kp1)) / ((altvb - altvb_kp1) * rho));
}

// dQVdt, dQCdt: dyn_moist.py:201-243 and :49-129.  Owned i in [1,nx], j in [1,ny].
// EULER: apply the pressure-weighted Euler step (dyn_timestep.py:291-296) in the same thread and
// write the new QV/QC to q?_out instead of storing the tendencies (fused production path).
// The interface value below the bottom level is an unchecked out-of-range read in the
// reference, multiplied by WWIND[nz] = 0: taken as 0 here.
// FROM_WINDS: UFLX, VFLX are re-formed from the winds (fused step without stored fluxes, see
// k_continuity_reg).
// TURB_PRE: the turbulent tendencies come from k_turb_tendencies (fused turbulence-coupled step: by the
// time this kernel runs, PHI already belongs to the new state).
template <bool TURB, bool EULER, bool FROM_WINDS = false, bool TURB_PRE = false>
__global__ __launch_bounds__(TI *TJ) void k_moist(Geo g, Ptrs p, const real *qv_old, const real *qc_old,
                                                  real *qv_out, real *qc_out, real dt, int lin)
{
    int i, j;
    if (!owned_cell(g, lin, i, j)) return;
    const int P = g.P, nz = g.nz;
    const size_t c = at2(g, i, j);
    const real colp = p.COLP[c], colp_new = p.COLP_NEW[c], colp_old = p.COLP_OLD[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1], c_jm1 = p.COLP[c - P], c_jp1 = p.COLP[c + P];
    const real area = p.A[c];
    real surf = 0.0;
    if (TURB) surf = p.SLHFLX[c] / CON_LH;
    const real r_area = 1.0 / area, r_cn = 1.0 / colp_new;
    real qv_km1 = 0.0, qc_km1 = 0.0, phi_km1 = 0.0;
    real fv_k = 0.0, fc_k = 0.0;                 // W * interface value at interface k (0 at the top)
    real qv_k = p.QV[c], qc_k = p.QC[c];
    LogCache lv_k = log_cache(qv_k), lc_k = log_cache(qc_k);
    // software pipeline as in k_pott_fused: the inputs of level k+1 are requested before level k is
    // computed, so the HBM latency overlaps the two logarithms per interface
    __shared__ real s_met[FROM_WINDS ? 4 : 1][TI * TJ];
    const int st = threadIdx.y * TI + threadIdx.x;
    if (FROM_WINDS) {
        s_met[0][st] = p.dyis[c]; s_met[1][st] = p.dyis[c + 1]; s_met[2][st] = p.dxjs[c]; s_met[3][st] = p.dxjs[c + P];
    }
    const real *fu = FROM_WINDS ? p.UWIND : p.UFLX, *fv = FROM_WINDS ? p.VWIND : p.VFLX;
    struct In { real uf, uf1, vf, vf1, w1, qv1, qc1, v_im1, v_ip1, v_jm1, v_jp1, c_im1, c_ip1, c_jm1, c_jp1, vo, co; };
    auto load = [&](int k) {
        In r;
        const size_t q = c + (size_t)k * g.plane;
        r.uf = fu[q]; r.uf1 = fu[q + 1]; r.vf = fv[q]; r.vf1 = fv[q + P];
        const bool last = (k == nz - 1);
        r.w1 = last ? 0.0 : p.WWIND[q + g.plane];
        r.qv1 = last ? 0.0 : p.QV[q + g.plane];
        r.qc1 = last ? 0.0 : p.QC[q + g.plane];
        r.v_im1 = p.QV[q - 1]; r.v_ip1 = p.QV[q + 1]; r.v_jm1 = p.QV[q - P]; r.v_jp1 = p.QV[q + P];
        r.c_im1 = p.QC[q - 1]; r.c_ip1 = p.QC[q + 1]; r.c_jm1 = p.QC[q - P]; r.c_jp1 = p.QC[q + P];
        r.vo = 0.0; r.co = 0.0;
        if (EULER) { r.vo = qv_old[q]; r.co = qc_old[q]; }
        return r;
    };
    In nxt = load(0);
#pragma unroll 1
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const In cur = nxt;
        if (k + 1 < nz) nxt = load(k + 1);
        const real ds = p.dsigma[k], dif = p.moist_dif_coef[k];
        const real rds = EULER ? p.rdsigma[k] : 0.0;
        real uflx = cur.uf, uflx_ip1 = cur.uf1, vflx = cur.vf, vflx_jp1 = cur.vf1;
        if (FROM_WINDS) {           // dyn_continuity.py:40-45, operand order of k_continuity_reg
            uflx = (c_im1 + colp) / 2.0 * uflx * s_met[0][st];
            uflx_ip1 = (colp + c_ip1) / 2.0 * uflx_ip1 * s_met[1][st];
            vflx = (c_jm1 + colp) / 2.0 * vflx * s_met[2][st];
            vflx_jp1 = (colp + c_jp1) / 2.0 * vflx_jp1 * s_met[3][st];
        }
        const bool last = (k == nz - 1);
        const real qv_kp1 = cur.qv1, qc_kp1 = cur.qc1;
        real fv_kp1 = 0.0, fc_kp1 = 0.0;
        if (!last) {
            const real w_kp1 = cur.w1;
            LogCache lv_kp1 = log_cache(qv_kp1), lc_kp1 = log_cache(qc_kp1);
            fv_kp1 = w_kp1 * varvb_log_cached(lv_kp1, lv_k);
            fc_kp1 = w_kp1 * varvb_log_cached(lc_kp1, lc_k);
            lv_k = lv_kp1; lc_k = lc_kp1;
        }
        real tq[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const real x = s == 0 ? qv_k : qc_k;
            const real x_im1 = s == 0 ? cur.v_im1 : cur.c_im1, x_ip1 = s == 0 ? cur.v_ip1 : cur.c_ip1;
            const real x_jm1 = s == 0 ? cur.v_jm1 : cur.c_jm1, x_jp1 = s == 0 ? cur.v_jp1 : cur.c_jp1;
            const real f_k = s == 0 ? fv_k : fc_k, f_kp1 = s == 0 ? fv_kp1 : fc_kp1;
            real d = 0.0;
            const real adv = (+uflx * (x_im1 + x) / 2.0 - uflx_ip1 * (x + x_ip1) / 2.0
                                + vflx * (x_jm1 + x) / 2.0 - vflx_jp1 * (x + x_jp1) / 2.0);
            // fused step (EULER): per-column / per-level divisors as reciprocals, as in k_pott_fused
            // (the per-term switches of dyn_moist.py:70-115 are consulted by the stage instances only: a handle with a
            // term switched off never takes the fused step)
            if (EULER || term_on(g, DYCORE_TERM_MOIST_HOR_ADV)) d = d + (EULER ? adv * r_area : adv / area);
            if (!EULER && !term_on(g, DYCORE_TERM_MOIST_VERT_ADV)) { }
            else if (k == 0) d = d + (EULER ? colp_new * (-f_kp1) * rds : colp_new * (-f_kp1) / ds);
            else d = d + (EULER ? colp_new * (+f_k - f_kp1) * rds : colp_new * (+f_k - f_kp1) / ds);
            if (TURB) {
                const real tb = turb_flux(p.PHI[q], last ? 0.0 : p.PHI[q + g.plane], phi_km1, p.PHIVB[q], p.PHIVB[q + g.plane],
                                            x, s == 0 ? qv_kp1 : qc_kp1, s == 0 ? qv_km1 : qc_km1,
                                            p.KHEAT[q], p.KHEAT[q + g.plane], p.RHO[q], p.RHOVB[q], p.RHOVB[q + g.plane],
                                            colp, s == 0 ? surf : 0.0, k, nz);
                d = d + tb;
                if (s == 0 && !EULER) p.dQVdt_TURB[q] = tb;
            }
            if (TURB_PRE) d = d + (s == 0 ? p.TURB_QV[q] : p.TURB_QC[q]);
            if (dif > 0.0 && (EULER || term_on(g, DYCORE_TERM_MOIST_NUM_DIF)))
                d = d + dif * (c_im1 * x_im1 + c_ip1 * x_ip1 + c_jm1 * x_jm1 + c_jp1 * x_jp1 - 4.0 * colp * x);
            tq[s] = d;
        }
        if (EULER) {
            real nv = cur.vo * colp_old * r_cn + dt * tq[0] * r_cn;
            real nc = cur.co * colp_old * r_cn + dt * tq[1] * r_cn;
            if (g.i_moist_clip) {       // dyn_timestep.py:145-154: `if QV < 0: QV = 0` (a NaN stays a NaN)
                if (nv < 0.0) nv = 0.0;
                if (nc < 0.0) nc = 0.0;
            }
            qv_out[q] = nv;
            qc_out[q] = nc;
        } else {
            p.dQVdt[q] = tq[0];
            p.dQCdt[q] = tq[1];
        }
        if (TURB) phi_km1 = p.PHI[q];
        qv_km1 = qv_k; qc_km1 = qc_k;
        qv_k = qv_kp1; qc_k = qc_kp1;
        fv_k = fv_kp1; fc_k = fc_kp1;
    }
}

// ------------------------------------------------------------------------------------
// Euler forward with pressure weighting: dyn_timestep.py:212-297 (the exchange_BC of POTT,
// VWIND, UWIND, dyn_org_discretizations.py:388-390, follows as one k_halo_multi launch).
// Threads cover i in [1,nx],
// j in [1,ny+1]; U and POTT own rows 1..ny, V owns js rows not on a pole.
// `cur` pointers are the fields being written (UWIND, VWIND, POTT); *_OLD and the
// tendencies are read at the same cell only, so writing in place is race-free.
__global__ __launch_bounds__(TI *TJ) void k_euler_forward(Geo g, Ptrs p, real dt)
{
    DYC_IJ(1, 1)
    if (i > g.nx || j > g.ny + 1) return;
    const int P = g.P;
    const size_t c = at2(g, i, j);
    const int jg = j + g.joff;
#define CN(a, b) (p.COLP[c + (b) * P + (a)] * p.A[c + (b) * P + (a)])
#define CO(a, b) (p.COLP_OLD[c + (b) * P + (a)] * p.A[c + (b) * P + (a)])
#define CN2(a, b) (2.0 * p.COLP[c + (b) * P + (a)] * p.A[c + (b) * P + (a)])
#define CO2(a, b) (2.0 * p.COLP_OLD[c + (b) * P + (a)] * p.A[c + (b) * P + (a)])
    const bool row_c = (j <= g.ny);
    const bool v_pole = (jg == 1) || (jg == g.ny_global + 1);
    const bool v_own = row_c || g.pole_n;   // js = ny+1 belongs to the northern neighbour band
    real ca_is = 1.0, cao_is = 1.0, ca_js = 1.0, cao_js = 1.0, cn = 1.0, co = 1.0;
    if (row_c) {
        if (jg == 1) {
            ca_is = 1.0 / 4.0 * (CN(-1, 1) + CN(0, 1) + CN(-1, 0) + CN(0, 0));
            cao_is = 1.0 / 4.0 * (CO(-1, 1) + CO(0, 1) + CO(-1, 0) + CO(0, 0));
        } else if (jg == g.ny_global) {
            ca_is = 1.0 / 4.0 * (CN(-1, -1) + CN(0, -1) + CN(-1, 0) + CN(0, 0));
            cao_is = 1.0 / 4.0 * (CO(-1, -1) + CO(0, -1) + CO(-1, 0) + CO(0, 0));
        } else {
            ca_is = 1.0 / 8.0 * (CN(-1, 1) + CN(0, 1) + CN2(-1, 0) + CN2(0, 0) + CN(-1, -1) + CN(0, -1));
            cao_is = 1.0 / 8.0 * (CO(-1, 1) + CO(0, 1) + CO2(-1, 0) + CO2(0, 0) + CO(-1, -1) + CO(0, -1));
        }
        cn = p.COLP[c];
        co = p.COLP_OLD[c];
    }
    if (!v_pole) {
        ca_js = 1.0 / 8.0 * (CN(1, -1) + CN(1, 0) + CN2(0, -1) + CN2(0, 0) + CN(-1, -1) + CN(-1, 0));
        cao_js = 1.0 / 8.0 * (CO(1, -1) + CO(1, 0) + CO2(0, -1) + CO2(0, 0) + CO(-1, -1) + CO(-1, 0));
    }
#undef CN
#undef CO
#undef CN2
#undef CO2
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        if (row_c) {
            p.UWIND[at(g, i, j, k)] = p.UWIND_OLD[q] * cao_is / ca_is + dt * p.dUFLXdt[q] / ca_is;
            p.POTT[at(g, i, j, k)] = p.POTT_OLD[q] * co / cn + dt * p.dPOTTdt[q] / cn;
            if (g.i_moist) {   // the reference's CPU kernel does not clip (dyn_timestep.py:291-296), its accelerator kernel does (:145-154)
                real nv = p.QV_OLD[q] * co / cn + dt * p.dQVdt[q] / cn;
                real nc = p.QC_OLD[q] * co / cn + dt * p.dQCdt[q] / cn;
                if (g.i_moist_clip) {
                    if (nv < 0.0) nv = 0.0;
                    if (nc < 0.0) nc = 0.0;
                }
                p.QV[q] = nv;
                p.QC[q] = nc;
            }
        }
        if (v_own) {
            if (!v_pole) p.VWIND[at(g, i, j, k)] = p.VWIND_OLD[q] * cao_js / ca_js + dt * p.dVFLXdt[q] / ca_js;
            else p.VWIND[at(g, i, j, k)] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------
// (p / 1e5) ** kappa of dyn_diagnostics.py:146-153: the Exner function at an interface.
// The generic pow() of the device library costs 226 vector instructions per call -- 70 % of the temperature sweep's
// arithmetic per level, the largest single item of the whole substep (profiles/r03_notes.md).  Its argument here lives in
// a narrow domain (0.05 < p / 1e5 < 1.3, kappa a constant), so it is evaluated directly, ~85 instructions, every rounding
// written out (fma / contract off: all kernels that call it produce the same bits, which the band decomposition relies on):
//   x = p * 1e-5 as a double-double (no division);  x = 2^e m, m in [sqrt(1/2), sqrt(2));
//   log m = 2 atanh(s), s = (m - 1) / (m + 1) as a double-double quotient, odd series up to s^21 (|s| <= 0.172);
//   y = kappa (e ln 2 + log m) as a double-double (relative error of the logarithm < 2^-58);
//   exp(y) = 2^n exp(r), r = y - n ln 2 exact in its high part, Taylor polynomial of degree 13, the last sum compensated.
// Error <= 0.8 ulp against the exact power (mean 0.26; glibc's pow: 0.51, ocml's: ~1; exp(kappa log p): 3.3) on 4e6
// random arguments of the domain (prototype against quad precision; the device instruction sequence is the same).
// -DDYC_EXNER_OCML (experiment builds): the library's pow, as in rounds 1-2.
__device__ __forceinline__ double exner_pow(double p)
{
#pragma clang fp contract(off)
    constexpr double C_HI = 0x1.4f8b588e368f1p-17, C_LO = -0x1.ee78183f91e64p-71;        // 1e-5
    constexpr double LN2_HI = 0x1.62e42fefa38p-1, LN2_LO = 0x1.ef35793c7673p-45;           // ln 2, the high part with 11 trailing zero bits
    constexpr double LOG2E = 0x1.71547652b82fep+0;
    constexpr double KAP = (double)CON_KAPPA;               // (the float build's kappa is the float one, as in the reference's float32 mode)
    const double xh = p * C_HI;
    const double xl = fma(p, C_LO, fma(p, C_HI, -xh));
    double m = __builtin_amdgcn_frexp_mant(xh);             // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(xh);
    const bool low = m < 0.70710678118654752;
    m = m * (low ? 2.0 : 1.0);
    e = low ? e - 1 : e;
    const double corr = xl * __builtin_amdgcn_rcp(xh);      // log(xh + xl) - log(xh)
    const double f = m - 1.0, g = m + 1.0, gl = m - (g - 1.0);        // m - 1 and g - 1 are exact; gl = rounding error of m + 1
    double r = __builtin_amdgcn_rcp(g);
    r = fma(r, fma(-g, r, 1.0), r);
    r = fma(r, fma(-g, r, 1.0), r);
    const double s = f * r;
    const double sl = fma(-s, gl, fma(-s, g, f)) * r;       // s + sl = (m - 1) / (m + 1)
    const double s2 = s * s;
    double P = 1.0 / 21.0;
    P = fma(P, s2, 1.0 / 19.0); P = fma(P, s2, 1.0 / 17.0); P = fma(P, s2, 1.0 / 15.0); P = fma(P, s2, 1.0 / 13.0);
    P = fma(P, s2, 1.0 / 11.0); P = fma(P, s2, 1.0 / 9.0); P = fma(P, s2, 1.0 / 7.0); P = fma(P, s2, 1.0 / 5.0);
    P = fma(P, s2, 1.0 / 3.0);
    P = P * s2;
    const double lh = 2.0 * s;
    const double ll = fma(lh, P, 2.0 * sl) + corr;          // log m = lh + ll
    const double E = (double)e;
    const double A = E * LN2_HI;                            // exact
    double th = A + lh;
    double tl = ((A - th) + lh) + fma(E, LN2_LO, ll);       // |A| >= |lh| or A = 0
    { const double t2 = th + tl; tl = (th - t2) + tl; th = t2; }
    const double yh = KAP * th;
    const double yl = fma(KAP, tl, fma(KAP, th, -yh));
    const double n = __builtin_rint(yh * LOG2E);
    const double rh = fma(-n, LN2_HI, yh);                  // exact
    const double rl = fma(-n, LN2_LO, yl);
    double q = 1.0 / 6227020800.0;
    q = fma(q, rh, 1.0 / 479001600.0); q = fma(q, rh, 1.0 / 39916800.0); q = fma(q, rh, 1.0 / 3628800.0); q = fma(q, rh, 1.0 / 362880.0);
    q = fma(q, rh, 1.0 / 40320.0); q = fma(q, rh, 1.0 / 5040.0); q = fma(q, rh, 1.0 / 720.0); q = fma(q, rh, 1.0 / 120.0);
    q = fma(q, rh, 1.0 / 24.0); q = fma(q, rh, 1.0 / 6.0); q = fma(q, rh, 0.5);
    const double t = (rh * rh) * q;
    const double u = 1.0 + rh, ul = (1.0 - u) + rh;
    const double res = u + fma(rl, u + t, ul + t);
    return __builtin_amdgcn_ldexp(res, (int)n);
}
__device__ __forceinline__ real exner_vb(real p)
{
#ifdef DYC_EXNER_OCML
    return pow(p / 100000.0, CON_KAPPA);
#else
    return (real)exner_pow((double)p);
#endif
}

// One level of the primary diagnostics (dyn_diagnostics.py:139-195), shared by the stage kernel, the
// halo-row kernel of a latitude band and the fused temperature kernel.  Contraction to FMA is
// switched off inside: the compiler fuses a*b+c differently in different surroundings, and a band
// recomputes the diagnostics of rows it received from its neighbour -- they must come out bit
// for bit as the owner computed them.
__device__ __forceinline__ real interface_pressure(real pair_top, real svb, real colp)
{
#pragma clang fp contract(off)
    return pair_top + svb * colp;
}
struct DiagLevel { real p_km12, pvb_k, pvtf, phi, phivb; };
// ... with the interface pressure and its Exner value already at hand
__device__ __forceinline__ DiagLevel diag_level_pre(real p_km12, real pvb_k, real pott,
                                                    real p_kp12, real pvb_kp1, real phivb_kp1)
{
#pragma clang fp contract(off)
    DiagLevel d;
    d.p_km12 = p_km12;
    d.pvb_k = pvb_k;
    d.pvtf = 1.0 / (1.0 + CON_KAPPA) * (pvb_kp1 * p_kp12 - d.pvb_k * d.p_km12) / (p_kp12 - d.p_km12);
    d.phi = phivb_kp1 - CON_CP * (pott * (d.pvtf - pvb_kp1));
    d.phivb = d.phi - CON_CP * (pott * (d.pvb_k - d.pvtf));
    return d;
}
__device__ __forceinline__ DiagLevel diag_level(real pair_top, real svb_k, real colp, real pott,
                                                real p_kp12, real pvb_kp1, real phivb_kp1)
{
    const real p_km12 = interface_pressure(pair_top, svb_k, colp);
    return diag_level_pre(p_km12, exner_vb(p_km12), pott, p_kp12, pvb_kp1, phivb_kp1);
}
// ---- the diagnostics as the FUSED step evaluates them ("canonical" forms) --------------------------------------------
// The fused step stores neither PVTF nor PHI: the momentum kernel forms both in the thread, for its own column and its
// western and southern neighbour, from PVTFVB, POTT, the column pressure and the surface geopotential; the temperature
// kernel forms them for the new column where a level chunk of the momentum kernel needs a stored PHIVB to start from.
// Different kernels, tiles, level chunks and latitude bands must get the SAME bits for the same cell (a chunk continues
// the hydrostatic integration of the chunk below it from a stored plane, a band that of the single-domain run), so
// every rounding is written out here and every kernel calls these functions.
//   PVTF (dyn_diagnostics.py:146-153): the reference divides by (1 + kappa)(p[k+1] - p[k]); p[k+1] - p[k] is dsigma[k] COLP, so
//   the quotient is a product with rr = (1 / ((1 + kappa) COLP)) (1 / dsigma[k]) -- one division per column instead of one
//   per level and cell; it differs from the reference's quotient by the rounding of p[k+1] - p[k] (~4e-15 relative).
//   (double intermediates: the difference of the two products cancels five digits; a no-op in the double build)
__device__ __forceinline__ double canon_pk(real pair_top, real svb, real colp)
{
#pragma clang fp contract(off)
    return (double)pair_top + (double)svb * (double)colp;       // dyn_diagnostics.py:146, two roundings as written there
}
__device__ __forceinline__ real canon_rcol(real colp) { return (real)(1.0 / ((1.0 + (double)CON_KAPPA) * (double)colp)); }
__device__ __forceinline__ real canon_pvtf(real pvb_kp1, double p_kp1, real pvb_k, double p_k, real rcol, real rds)
{
#pragma clang fp contract(off)
    const double lo = (double)pvb_k * p_k;
    return (real)(fma((double)pvb_kp1, p_kp1, -lo) * ((double)rcol * (double)rds));
}
// Exner function at an interface from its (double) pressure
__device__ __forceinline__ real canon_exner(double pk) { return (real)exner_pow(pk); }
// hydrostatic integration, dyn_diagnostics.py:160-174: PHI[k] from PHIVB[k+1], PHIVB[k] from PHI[k]
__device__ __forceinline__ real canon_phi(real phivb_kp1, real pott, real pvtf, real pvb_kp1)
{
#pragma clang fp contract(off)
    return phivb_kp1 - CON_CP * (pott * (pvtf - pvb_kp1));
}
__device__ __forceinline__ real canon_phivb(real phi, real pott, real pvb_k, real pvtf)
{
#pragma clang fp contract(off)
    return phi - CON_CP * (pott * (pvb_k - pvtf));
}
// interface value between levels k and k+1, dyn_diagnostics.py:183-186
__device__ __forceinline__ real pottvb_between(real pvb_kp1, real pvtf, real pott, real pvtf_kp1, real pott_kp1)
{
#pragma clang fp contract(off)
    return ((pvb_kp1 - pvtf) * pott + (pvtf_kp1 - pvb_kp1) * pott_kp1) / (pvtf_kp1 - pvtf);
}

// primary diagnostics: dyn_diagnostics.py:139-195 (diag_PVTF, diag_PHI, diag_POTTVB) in one
// bottom-up sweep over the whole array incl. halos (i in [0,nx+1], j in [0,ny+1]).
// One pow() per interface instead of the reference's two per cell: PVTFVB[k+1] is carried.
// `pvb` (optional): the Exner values of the column's interfaces, pvb[k * stride], computed beforehand.
__device__ __forceinline__ void diag_column(const Geo &g, const Ptrs &p, int i, int j, const real *pvb = nullptr, int stride = 0)
{
    const size_t c = at2(g, i, j);
    const int nz = g.nz;
    const real colp = p.COLP[c];
    real p_kp12 = interface_pressure(g.pair_top, p.sigma_vb[nz], colp);
    real pvb_kp1 = pvb ? pvb[nz * stride] : exner_vb(p_kp12);
    real phivb_kp1 = p.HSURF[c] * CON_G;
    p.PVTFVB[c + (size_t)nz * g.plane] = pvb_kp1;
    p.PHIVB[c + (size_t)nz * g.plane] = phivb_kp1;
    real pott_kp1 = 0.0, pvtf_kp1 = 0.0, pottvb_first = 0.0;
    for (int k = nz - 1; k >= 0; k--) {
        const size_t q = c + (size_t)k * g.plane;
        const real pott = p.POTT[q];
        const DiagLevel dl = pvb ? diag_level_pre(interface_pressure(g.pair_top, p.sigma_vb[k], colp), pvb[k * stride], pott, p_kp12, pvb_kp1, phivb_kp1)
                                 : diag_level(g.pair_top, p.sigma_vb[k], colp, pott, p_kp12, pvb_kp1, phivb_kp1);
        const real p_km12 = dl.p_km12, pvb_k = dl.pvb_k, pvtf = dl.pvtf, phi = dl.phi, phivb = dl.phivb;
        p.PVTF[q] = pvtf;
        p.PVTFVB[q] = pvb_k;
        p.PHI[q] = phi;
        p.PHIVB[q] = phivb;
        if (k < nz - 1) {   // interface k+1 between levels k and k+1
            const real vb = pottvb_between(pvb_kp1, pvtf, pott, pvtf_kp1, pott_kp1);
            p.POTTVB[q + g.plane] = vb;
            if (k == nz - 2 && nz > 2)   // surface extrapolation (elif branch, dyn_diagnostics.py:191-194)
                p.POTTVB[c + (size_t)nz * g.plane] = pott_kp1 - (vb - pott_kp1);
            if (k == 0) pottvb_first = vb;
        }
        p_kp12 = p_km12;
        pvb_kp1 = pvb_k;
        phivb_kp1 = phivb;
        pott_kp1 = pott;
        pvtf_kp1 = pvtf;
    }
    if (nz >= 2) p.POTTVB[c] = pott_kp1 - (pottvb_first - pott_kp1);   // model top, :187-190
}

__global__ __launch_bounds__(TI *TJ) void k_primary_diag(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    diag_column(g, p, i, j);
}

// secondary diagnostics, RHO / RHOVB only: dyn_diagnostics.py:199-222
__global__ __launch_bounds__(TI *TJ) void k_diag_rho(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    for (int k = 0; k <= g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real pvb = p.PVTFVB[q];
        p.RHOVB[q] = 100000.0 * pow(pvb, 1.0 / CON_KAPPA) / (CON_RD * (p.POTTVB[q] * pvb));
        if (k < g.nz) {
            const real pv = p.PVTF[q];
            p.RHO[q] = 100000.0 * pow(pv, 1.0 / CON_KAPPA) / (CON_RD * (p.POTT[q] * pv));
        }
    }
}

// secondary diagnostics, every output: dyn_diagnostics.py:199-222 (whole array incl. halos)
__global__ __launch_bounds__(TI *TJ) void k_diag_secondary(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    for (int k = 0; k <= g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real pvb = p.PVTFVB[q];
        const real pairvb = 100000.0 * pow(pvb, 1.0 / CON_KAPPA), tairvb = p.POTTVB[q] * pvb;
        p.PAIRVB[q] = pairvb;
        p.TAIRVB[q] = tairvb;
        p.RHOVB[q] = pairvb / (CON_RD * tairvb);
        if (k < g.nz) {
            const real pv = p.PVTF[q];
            const real tair = p.POTT[q] * pv, pair = 100000.0 * pow(pv, 1.0 / CON_KAPPA);
            p.TAIR[q] = tair;
            p.PAIR[q] = pair;
            p.RHO[q] = pair / (CON_RD * tair);
            const real wx = (p.UWIND[q] + p.UWIND[q + 1]) / 2.0, wy = (p.VWIND[q] + p.VWIND[q + g.P]) / 2.0;
            p.WINDX[q] = wx;
            p.WINDY[q] = wy;
            p.WIND[q] = sqrt(wx * wx + wy * wy);
        }
    }
}

// turbulence scheme: turb_compute.py:191-208 (loop), :54-142 (formulae), :28-48 (constants).
// KMOM, KHEAT at interfaces 1..nz-1 of the whole array incl. halos.
__global__ __launch_bounds__(TI *TJ) void k_turbulence(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const real Ri_c = 1.0, free_mix_len = 200.0, con_k = 0.35, con_Pr = 0.72;
    const real min_wind_diff = 0.0001, min_KMOM = 0.000001, max_KMOM = 0.01;
    const size_t c = at2(g, i, j);
    const real hsurf = p.HSURF[c];
    real wx_km = p.WINDX[c], wy_km = p.WINDY[c], phi_km = p.PHI[c], qv_km = p.QV[c], pott_km = p.POTT[c];
    for (int k = 1; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real wx_kp = p.WINDX[q], wy_kp = p.WINDY[q], phi_kp = p.PHI[q], qv_kp = p.QV[q], pott_kp = p.POTT[q];
        real ax = wx_km, ay = wy_km;
        if (ax == wx_kp) ax += min_wind_diff;
        if (ay == wy_kp) ay += min_wind_diff;
        const real alt_k = p.PHIVB[q] / CON_G, alt_km = phi_km / CON_G, alt_kp = phi_kp / CON_G;
        const real hgt = alt_k - hsurf;
        const real mix_len = con_k * hgt / (1.0 + con_k * hgt / free_mix_len);
        const real qv_k = varvb_log(qv_kp, qv_km);
        const real pott_v = p.POTTVB[q] * (1.0 + qv_k / 0.622) / (1.0 + qv_k);
        const real dwx = ax - wx_kp, dwy = ay - wy_kp, dz = alt_km - alt_kp;
        const real ri_b = (CON_G / pott_v * (pott_km - pott_kp) * dz) / (dwx * dwx + dwy * dwy);
        const real shear = sqrt((dwx / dz) * (dwx / dz) + (dwy / dz) * (dwy / dz));
        real kmom = mix_len * mix_len * shear * (Ri_c - ri_b) / Ri_c;
        if (kmom < min_KMOM) kmom = min_KMOM;
        if (kmom > max_KMOM) kmom = max_KMOM;
        p.KMOM[q] = kmom;
        p.KHEAT[q] = kmom / con_Pr;
        wx_km = wx_kp; wy_km = wy_kp; phi_km = phi_kp; qv_km = qv_kp; pott_km = pott_kp;
    }
}

// ------------------------------------------------------------------------------------
// lateral boundary conditions of one field: misc_boundaries.py:47-82.  Every halo cell
// pulls from the interior cell it mirrors, so one pass suffices (corners included).
__global__ void k_exchange_bc(Geo g, real *F, int stgx, int stgy, int nk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int k = blockIdx.z;
    const int fnx = g.nx + 2 + stgx, fny = g.ny + 2 + stgy;
    if (i >= fnx || j >= fny || k >= nk) return;
    int si = i;
    if (i == 0) si = g.nx;
    else if (i == g.nx + 1) si = 1;
    else if (stgx && i == g.nx + 2) si = 2;
    real *pl = F + (size_t)k * g.plane;
    if (stgy) {
        const bool zero = (g.pole_s && j <= 1) || (g.pole_n && j >= g.ny + 1);
        if (zero) pl[(size_t)j * g.P + i] = 0.0;
        else if (si != i) pl[(size_t)j * g.P + i] = pl[(size_t)j * g.P + si];
    } else {
        int sj = j;
        if (j == 0 && g.pole_s) sj = 1;
        else if (j == g.ny + 1 && g.pole_n) sj = g.ny;
        if (si != i || sj != j) pl[(size_t)j * g.P + i] = pl[(size_t)sj * g.P + si];
    }
}

// The same for up to 12 fields in ONE launch, touching halo cells only (the production
// path: every kernel writes interior cells with plain coalesced stores, then one such launch
// refreshes the lateral halo of everything it produced).  Part A = wall rows (all columns),
// part B = the periodic columns of the interior rows.  Rows that belong to a neighbouring
// latitude band are left alone: they arrive by message, periodic columns included.
// The PHIVB planes of rows j_lo .. j_hi (columns 0 .. nx+1): PHIVB at every PHIVB_STRIDE-th interface, integrated from the
// surface with the canonical forms -- what the momentum wavefronts of a level chunk of the whole-substep kernel start from
// (that kernel keeps no planes: its temperature wavefronts do not integrate PHI; its work list cuts only the tile rows beyond
// the last full round of workgroups into chunks, a few rows of 720).  Evaluated by k_phivb_rows, or as a rider of the
// k_halo_multi launch that precedes the whole-substep launch anyway (one launch less per substep: on a 1 deg grid a
// launch gap is 3 % of the step).  j_hi < j_lo: no rows.
struct PlaneRows {
    const real *colp, *hsurf, *pvtfvb, *pott, *sigma_vb, *rdsigma;
    real *phivb;
    int j_lo, j_hi;
};
__device__ __forceinline__ void phivb_planes_column(const Geo &g, const PlaneRows &r, int i, int j)
{
    const size_t c = at2(g, i, j);
    const int nz = g.nz;
    const real colp = r.colp[c], rcol = canon_rcol(colp);
    real phivb_kp1 = r.hsurf[c] * CON_G;
    double pk1 = canon_pk(g.pair_top, r.sigma_vb[nz], colp);
    real pvb_kp1 = r.pvtfvb[c + (size_t)nz * g.plane];
    // Batches of NB levels with all their loads issued first: the integration is a chain of four multiply-adds per level,
    // the launch a few hundred wavefronts -- left as one load pair per iteration the kernel waits nz times for HBM
    constexpr int NB = 16;
    for (int kb = nz - 1; kb >= 1; kb -= NB) {
        real pvb_[NB], pott_[NB];
#pragma unroll
        for (int d = 0; d < NB; d++) {
            const int k = max(kb - d, 1);
            const size_t q = c + (size_t)k * g.plane;
            pvb_[d] = r.pvtfvb[q]; pott_[d] = r.pott[q];
        }
#pragma unroll
        for (int d = 0; d < NB; d++) {
            const int k = kb - d;
            if (k >= 1) {
                const double pk = canon_pk(g.pair_top, r.sigma_vb[k], colp);
                const real pvb_k = pvb_[d], pott = pott_[d];
                const real pvtf = canon_pvtf(pvb_kp1, pk1, pvb_k, pk, rcol, r.rdsigma[k]);
                phivb_kp1 = canon_phivb(canon_phi(phivb_kp1, pott, pvtf, pvb_kp1), pott, pvb_k, pvtf);
                if (k % PHIVB_STRIDE == 0) r.phivb[c + (size_t)k * g.plane] = phivb_kp1;
                pk1 = pk; pvb_kp1 = pvb_k;
            }
        }
    }
}
__global__ __launch_bounds__(TI *TJ) void k_phivb_rows(Geo g, PlaneRows r)
{
    const int i = blockIdx.x * TI + threadIdx.x, j = r.j_lo + blockIdx.y * TJ + threadIdx.y;
    if (i > g.nx + 1 || j > r.j_hi) return;
    phivb_planes_column(g, r, i, j);
}

struct HaloList {
    int n;
    real *base[12];
    int stgx[12], stgy[12], nk[12];
    // optional rider: a plain copy of one 2-D plane (COLP <- COLP_NEW of dyn_matsuno.py:64-67) on the
    // z-slice after the last field, so that the step does not pay a launch for it
    const real *copy_src;
    real *copy_dst, *copy_dst2;     // (copy_dst2: a second destination of the same plane -- COLP_OLD of the NEXT step, which
    long long copy_n;               //  begins with COLP_OLD <- COLP, dyn_matsuno.py:34-41 -- or null)
    // optional rider: the PHIVB planes of a few rows (PlaneRows) on the z-slice after that
    PlaneRows rows;
};
__global__ void k_halo_multi(Geo g, HaloList l)
{
    const int e = blockIdx.z, k = blockIdx.y;
    if (e >= l.n) {
        const int e_copy = l.copy_dst ? l.n : -1, e_rows = l.n + (l.copy_dst ? 1 : 0);
        if (e == e_copy) {
            const long long stride = (long long)gridDim.x * gridDim.y * blockDim.x;
            for (long long t = ((long long)k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; t < l.copy_n; t += stride)
            {
                const real v = l.copy_src[t];
                l.copy_dst[t] = v;
                if (l.copy_dst2) l.copy_dst2[t] = v;
            }
        } else if (e == e_rows && l.rows.j_hi >= l.rows.j_lo) {      // (the launch has at least (nx + 2) * rows threads per slice)
            const long long t = ((long long)k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
            const int nxx = g.nx + 2;
            const int j = l.rows.j_lo + (int)(t / nxx), i = (int)(t - (long long)(t / nxx) * nxx);
            if (j <= l.rows.j_hi) phivb_planes_column(g, l.rows, i, j);
        }
        return;
    }
    if (e >= l.n || k >= l.nk[e]) return;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int stgx = l.stgx[e], stgy = l.stgy[e];
    const int fnx = g.nx + 2 + stgx;
    real *pl = l.base[e] + (size_t)k * g.plane;
    // ---- part A: wall rows ----
    const int rows_s = g.pole_s ? (stgy ? 2 : 1) : 0, rows_n = g.pole_n ? (stgy ? 2 : 1) : 0;
    const int nA = (rows_s + rows_n) * fnx;
    if (t < nA) {
        const int r = t / fnx, i = t - r * fnx;
        const int j = r < rows_s ? r : g.ny + 1 + (r - rows_s);
        if (stgy) { pl[(size_t)j * g.P + i] = 0.0; return; }
        int si = i;
        if (i == 0) si = g.nx;
        else if (i == g.nx + 1) si = 1;
        else if (stgx && i == g.nx + 2) si = 2;
        const int sj = (j == 0) ? 1 : g.ny;
        pl[(size_t)j * g.P + i] = pl[(size_t)sj * g.P + si];
        return;
    }
    // ---- part B: periodic columns of the rows this band owns ----
    t -= nA;
    const int jlo = (stgy && g.pole_s) ? 2 : 1, jhi = g.ny;
    const int nj = jhi - jlo + 1, ncol = 2 + stgx;
    if (nj <= 0 || t >= ncol * nj) return;
    const int cidx = t / nj, j = jlo + (t - cidx * nj);
    const int i = cidx == 0 ? 0 : (cidx == 1 ? g.nx + 1 : g.nx + 2);
    const int si = cidx == 0 ? g.nx : (cidx == 1 ? 1 : 2);
    pl[(size_t)j * g.P + i] = pl[(size_t)j * g.P + si];
}

// ---- layout adaptor: host [i][j][k] (k contiguous) <-> device [k][j][i] ---------------
// One workgroup transposes a 32(i) x 32(k) tile of row j through LDS, so both the read
// and the write side are contiguous.
__global__ void k_to_device(Geo g, const real *__restrict__ src, real *__restrict__ dst,
                            int n0, int n1, int n2)
{
    __shared__ real tile[32][33];
    const int j = blockIdx.z;
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int i = i0 + r, k = k0 + threadIdx.x;
        if (i < n0 && k < n2) tile[r][threadIdx.x] = src[((size_t)i * n1 + j) * n2 + k];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int k = k0 + r, i = i0 + threadIdx.x;
        if (i < n0 && k < n2) dst[(size_t)k * g.plane + (size_t)j * g.P + i] = tile[threadIdx.x][r];
    }
}
__global__ void k_to_host(Geo g, const real *__restrict__ src, real *__restrict__ dst,
                          int n0, int n1, int n2)
{
    __shared__ real tile[32][33];
    const int j = blockIdx.z;
    const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int k = k0 + r, i = i0 + threadIdx.x;
        if (i < n0 && k < n2) tile[r][threadIdx.x] = src[(size_t)k * g.plane + (size_t)j * g.P + i];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int i = i0 + r, k = k0 + threadIdx.x;
        if (i < n0 && k < n2) dst[((size_t)i * n1 + j) * n2 + k] = tile[threadIdx.x][r];
    }
}
// ---- halo rows of a latitude band <-> contiguous message buffer (multi-GPU) -------------
// One launch moves up to 24 (field,row) pairs: entry e copies the nk[e] levels of row
// row[e], columns 0..nx+2, to/from buf + off[e].
// (k0, kstep: the levels k0, k0 + kstep, ... -- the PHIVB planes at every PHIVB_STRIDE-th interface travel this way)
constexpr int ROWLIST_MAX = 24;
struct RowList {
    int n;
    real *base[ROWLIST_MAX];
    int row[ROWLIST_MAX], nk[ROWLIST_MAX], k0[ROWLIST_MAX], kstep[ROWLIST_MAX];
    long long off[ROWLIST_MAX];
};
__global__ void k_rows(Geo g, RowList rl, real *buf, int unpack)
{
    const int e = blockIdx.z, k = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rl.n || k >= rl.nk[e] || i > g.nx + 2) return;
    real *f = rl.base[e] + (size_t)(rl.k0[e] + k * rl.kstep[e]) * g.plane + (size_t)rl.row[e] * g.P;
    real *b = buf + rl.off[e] + (size_t)k * (g.nx + 3) + i;
    if (unpack) { f[i] = *b; return; }
    // a packed row carries the periodic images of its interior columns (misc_boundaries.py:22-31), so a
    // sender need not refresh the lateral halo of a field before its rows leave
    const int si = i == 0 ? g.nx : (i == g.nx + 1 ? 1 : (i == g.nx + 2 ? 2 : i));
    *b = f[si];
}

__global__ void k_fill(real *dst, size_t n, real v)
{
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
        dst[t] = v;
}

}  // namespace dyc
