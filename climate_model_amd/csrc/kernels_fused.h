// kernels_fused.h -- the production kernels of dycore_step_matsuno (dry / moist dycore).
//
// k_momentum_dma   : dyn_UVFLX_prepare.py:249-439 + dyn_UFLX.py:339-434 + dyn_VFLX.py:322-408 + the
//                    wind part of dyn_timestep.py:212-297 in one pass (k_momentum_fused is its
//                    register-staged predecessor, kept as the fallback for nz > DMAXZ).  The eight
//                    Arakawa-C fluxes B,C,D,E,R,Q,S,T and the interface products WWIND*U, WWIND*V are
//                    never written to HBM: they are re-derived per thread from UFLX/VFLX/WWIND tiles
//                    in LDS (3x3 halo, and one more column / row for the radius-2 composite
//                    stencil); the UFLX/VFLX tiles themselves are formed in LDS from the wind tiles.
//                    HBM traffic per launch: 7-9 reads + 2 writes instead of the 41 passes of the
//                    three stage kernels.
// k_continuity_reg : dyn_continuity.py:170-232, vertical sums in registers, WWIND the only 3-D output.
// k_pott_fused     : dyn_POTT.py + POTT part of dyn_timestep.py + dyn_diagnostics.py:139-195.
//
// Mapping: lane <-> longitude i (64 per wavefront), threadIdx.y <-> latitude row, k loop in
// the thread; one workgroup = 64 x FTJ columns.  Tiles are (FTJ+3) rows x 68 columns of
// doubles, rows 512-B+ apart, read with conflict-free ds_read_b64 (consecutive lanes ->
// consecutive doubles).  Workgroups are dealt to the XCDs in contiguous latitude bands so
// that the halo rows two workgroups share are served by one L2.
#pragma once
#include "dycore_common.h"
#include "kernels_stage.h"

namespace dyc {

constexpr int FTI = 64;                // lanes along i per workgroup (one wavefront per row)
constexpr int FTJ = 8;                 // rows per workgroup
constexpr int VW = 16 / sizeof(real);   // values per 16-B DMA chunk: 2 (double), 4 (float)
constexpr int FXO = VW;                // k_momentum_dma: tile column 0 is longitude i0 - FXO, a 16-B boundary (i = 1 sits on a 512-B one)
constexpr int FTW = ((64 + FXO + 2 + VW - 1) / VW) * VW;    // tile width: i-1 .. i+2 for 64 lanes behind FXO-1 spare columns: 68 / 72
constexpr int FTH = FTJ + 3;           // tile height: j-1 .. j+2
constexpr int FTN = FTW * FTH;         // doubles per tile
constexpr int FNT = FTI * FTJ;          // threads per workgroup
constexpr int FNF = 9;                 // staged fields

// XCD-aware workgroup order: hardware deals workgroup b to XCD b % 8; give every XCD a
// contiguous chunk of the (bx fastest, by slowest) tile order.  Pure speed; any placement
// is correct.
__device__ __forceinline__ bool xcd_tile(int gbx, int gby, int &bx, int &by)
{
    const int nblk = gbx * gby;
    const int b = blockIdx.x;
    const int chunk = (nblk + 7) / 8;
    const int lin = (b % 8) * chunk + b / 8;
    if (lin >= nblk) return false;
    bx = lin % gbx;
    by = lin / gbx;
    return true;
}

// Tile-row subset of a launch: launch row r (0 .. rs.n-1) is tile row rs.first + r, plus rs.gap
// once r has reached rs.split -- one contiguous range (gap 0) or the two edge ranges of a
// latitude band (the tile rows whose stencils touch rows received from the neighbours).
struct RowSet { int first, n, split, gap; };
__host__ __device__ __forceinline__ int rowset_row(const RowSet &rs, int r) { return rs.first + r + (r >= rs.split ? rs.gap : 0); }

// Work list of one k_momentum_dma launch: up to two segments of (tile rows, level chunks per tile).
// With one workgroup per CU, T equal tiles take ceil(T / CUs) rounds; the tiles beyond the last full
// round form the second segment and are cut into short level chunks, so that the last round is a
// fraction of a tile's time (do_momentum_fused picks the plan).  Workgroups [0, nblk_a) belong to
// segment a (nblk_a a multiple of 8, so both segments keep the XCD-aware order), the rest to b.
struct MomPlan { RowSet a; int ks_a, nblk_a; RowSet b; int ks_b; int patch; };
// `patch` = H > 1: inside bands of H tile rows the order runs down the columns first, so that the ~32 tiles an XCD
// works on at one time form an H x 32/H block: H-1 of H horizontal tile boundaries then have both neighbours in
// flight together and their shared halo rows are served by that XCD's L2 instead of being fetched twice
__device__ __forceinline__ bool xcd_tile_at(int b, int gbx, int gby, int &bx, int &by, int patch = 1)
{
    const int nblk = gbx * gby;
    const int chunk = (nblk + 7) / 8;
    const int lin = (b % 8) * chunk + b / 8;
    if (lin >= nblk) return false;
    if (patch > 1) {
        const int per = patch * gbx, q = lin / per, r = lin - q * per;
        const int hq = min(patch, gby - q * patch);         // rows of this band (the last one may be short)
        bx = r / hq;
        by = q * patch + (r - bx * hq);
    } else {
        bx = lin % gbx;
        by = lin / gbx;
    }
    return true;
}

// EULER: the pressure-weighted Euler step (dyn_timestep.py:212-297, UWIND/VWIND part) is
// applied in the same thread -- dUFLXdt/dVFLXdt never reach HBM; u_old/v_old hold the state at
// step start (read at the own cell only), u_out/v_out receive the new winds (interior cells;
// k_halo_multi refreshes the lateral halo afterwards).  `p.UWIND/p.VWIND` are the winds the tendencies are evaluated on.
// In the corrector u_out aliases u_old (in-place update of the own cell), hence no __restrict__.
template <bool EULER>
__global__ __launch_bounds__(FNT) void k_momentum_fused(Geo g, Ptrs p, int gbx, int gby, int jv_lo,
                                                        const real *u_old, const real *v_old,
                                                        real *u_out, real *v_out, real dt)
{
    __shared__ real sm[FNF][FTN];
    int bx, by;
    if (!xcd_tile(gbx, gby, bx, by)) return;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * FTI + tx;
    const int i0 = bx * FTI + 1, j0 = by * FTJ + 1;
    const int i = i0 + tx, j = j0 + ty;
    const int P = g.P;
    const bool active = (i <= g.nx) && (j <= g.ny);

    // tile element(s) this thread stages: e0 = tid, e1 = tid + FNT
    const int e0 = tid, e1 = tid + FNT;
    const bool has1 = e1 < FTN;
    int goff0, goff1 = 0;
    {
        int r = e0 / FTW, c = e0 - r * FTW;
        int gi = min(i0 - 1 + c, P - 1), gj = min(j0 - 1 + r, g.NJ - 1);
        goff0 = gj * P + gi;
        if (has1) {
            r = e1 / FTW; c = e1 - r * FTW;
            gi = min(i0 - 1 + c, P - 1); gj = min(j0 - 1 + r, g.NJ - 1);
            goff1 = gj * P + gi;
        }
    }
    const real *src[FNF] = {p.UFLX, p.VFLX, p.UWIND, p.VWIND, p.WWIND, p.PHI, p.POTT, p.PVTF, p.PVTFVB};
    // WWIND and PVTFVB are staged at interface k+1
    const int kofs[FNF] = {0, 0, 0, 0, 1, 0, 0, 0, 1};
    enum { S_UF = 0, S_VF, S_UW, S_VW, S_W, S_PHI, S_POTT, S_PVTF, S_PVB };

    // ---- per-column (2-D) quantities ----
    const size_t c = at2(g, active ? i : 1, active ? j : 1);
    const int jg = j + g.joff;
    const bool lo = (jg == 1), hi = (jg == g.ny_global);
    const bool v_on = active && (j >= jv_lo);            // dVFLXdt row owned and not a pole
    real ca[3][3];                                     // COLP_NEW * A at (i+a-1, j+b-1)
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const size_t q2 = c + (b - 1) * P + (a - 1);
            ca[b][a] = p.COLP_NEW[q2] * p.A[q2];
        }
    const real colp = p.COLP[c], colp_im1 = p.COLP[c - 1], colp_jm1 = p.COLP[c - P];
    const real corf_is = p.corf_is[c], lat_is = p.lat_is_rad[c];
    const real cos_is = cos(lat_is), sin_is = sin(lat_is);
    const real corf = p.corf[c], corf_jm1 = p.corf[c - P];
    const real lat = p.lat_rad[c], lat_jm1 = p.lat_rad[c - P];
    const real cos_c = cos(lat), sin_c = sin(lat), cos_jm1 = cos(lat_jm1), sin_jm1 = sin(lat_jm1);
    const real dlon = p.dlon_rad[c], dlat = p.dlat_rad[c], dyis = p.dyis[c], dxjs = p.dxjs[c];
    // pressure weights of the Euler step at the u- and v-point: dyn_timestep.py:42-79
    real ca_is = 1.0, cao_is = 1.0, ca_js = 1.0, cao_js = 1.0;
    if (EULER) {
        real co[3][3];                                 // COLP_OLD * A
#pragma unroll
        for (int b = 0; b < 3; b++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const size_t q2 = c + (b - 1) * P + (a - 1);
                co[b][a] = p.COLP_OLD[q2] * p.A[q2];
            }
#define CA(a, b) ca[(b) + 1][(a) + 1]
#define CO(a, b) co[(b) + 1][(a) + 1]
        if (lo) {
            ca_is = 1.0 / 4.0 * (CA(-1, 1) + CA(0, 1) + CA(-1, 0) + CA(0, 0));
            cao_is = 1.0 / 4.0 * (CO(-1, 1) + CO(0, 1) + CO(-1, 0) + CO(0, 0));
        } else if (hi) {
            ca_is = 1.0 / 4.0 * (CA(-1, -1) + CA(0, -1) + CA(-1, 0) + CA(0, 0));
            cao_is = 1.0 / 4.0 * (CO(-1, -1) + CO(0, -1) + CO(-1, 0) + CO(0, 0));
        } else {
            ca_is = 1.0 / 8.0 * (CA(-1, 1) + CA(0, 1) + 2.0 * CA(-1, 0) + 2.0 * CA(0, 0) + CA(-1, -1) + CA(0, -1));
            cao_is = 1.0 / 8.0 * (CO(-1, 1) + CO(0, 1) + 2.0 * CO(-1, 0) + 2.0 * CO(0, 0) + CO(-1, -1) + CO(0, -1));
        }
        ca_js = 1.0 / 8.0 * (CA(1, -1) + CA(1, 0) + 2.0 * CA(0, -1) + 2.0 * CA(0, 0) + CA(-1, -1) + CA(-1, 0));
        cao_js = 1.0 / 8.0 * (CO(1, -1) + CO(1, 0) + 2.0 * CO(0, -1) + 2.0 * CO(0, 0) + CO(-1, -1) + CO(-1, 0));
#undef CA
#undef CO
    }

    const int lx = tx + 1, ly = ty + 1;                  // own cell inside the tile
#define SM(f, a, b) sm[f][(ly + (b)) * FTW + (lx + (a))]

    real wu_k = 0.0, wv_k = 0.0;                       // interface products at k (0 at the top)
    real u_k = active ? p.UWIND[c] : 0.0, v_k = active ? p.VWIND[c] : 0.0;
    real pvb_k = p.PVTFVB[c], pvb_im1_k = p.PVTFVB[c - 1], pvb_jm1_k = p.PVTFVB[c - P];

#pragma unroll 1
    for (int k = 0; k < g.nz; k++) {
        // ---- stage the nine tiles of this level ----
        real r0[FNF], r1[FNF];
#pragma unroll
        for (int f = 0; f < FNF; f++) {
            const real *s = src[f] + (size_t)(k + kofs[f]) * g.plane;
            r0[f] = s[goff0];
            r1[f] = has1 ? s[goff1] : 0.0;
        }
        // next level's own-cell winds (vertical interpolation to interface k+1)
        const bool last = (k == g.nz - 1);
        const size_t qn = c + (size_t)(last ? k : k + 1) * g.plane;
        const real u_kp1 = p.UWIND[qn], v_kp1 = p.VWIND[qn];
        __syncthreads();                                  // previous level fully consumed
#pragma unroll
        for (int f = 0; f < FNF; f++) {
            sm[f][e0] = r0[f];
            if (has1) sm[f][e1] = r1[f];
        }
        __syncthreads();

        if (active) {
            const size_t q = c + (size_t)k * g.plane;
            const real ds = p.dsigma[k];
            // ---- UFLX (i-1..i+2, j-1..j+1) and VFLX (i-1..i+1, j-1..j+2) neighbourhoods ----
            real uf[3][4], vf[4][3];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int a = 0; a < 4; a++) uf[b][a] = SM(S_UF, a - 1, b - 1);
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int a = 0; a < 3; a++) vf[b][a] = SM(S_VF, a - 1, b - 1);
#define UF(a, b) uf[(b) + 1][(a) + 1]
#define VF(a, b) vf[(b) + 1][(a) + 1]
            // wind neighbourhoods
            real uw[3][3], vw[3][3];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int a = 0; a < 3; a++) { uw[b][a] = SM(S_UW, a - 1, b - 1); vw[b][a] = SM(S_VW, a - 1, b - 1); }
#define U(a, b) uw[(b) + 1][(a) + 1]
#define V(a, b) vw[(b) + 1][(a) + 1]
            // ---- interface k+1: WWIND*U, WWIND*V (dyn_functions.py:211-270) ----
            real wu_kp1 = 0.0, wv_kp1 = 0.0;
            if (!last) {
                const real ds1 = p.dsigma[k + 1];
#define WT(a, b) (ca[(b) + 1][(a) + 1] * SM(S_W, a, b))
#define WT2(a, b) (2.0 * ca[(b) + 1][(a) + 1] * SM(S_W, a, b))
                real s;
                if (lo) s = 0.25 * (WT(-1, 1) + WT(0, 1) + WT(-1, 0) + WT(0, 0));
                else if (hi) s = 0.25 * (WT(-1, 0) + WT(0, 0) + WT(-1, -1) + WT(0, -1));
                else s = 0.125 * (WT(-1, 1) + WT(0, 1) + WT2(-1, 0) + WT2(0, 0) + WT(-1, -1) + WT(0, -1));
                wu_kp1 = s * ((ds1 * u_k + ds * u_kp1) / (ds1 + ds));
                const real sv = 0.125 * (WT(1, -1) + WT(1, 0) + WT2(0, -1) + WT2(0, 0) + WT(-1, -1) + WT(-1, 0));
                wv_kp1 = sv * ((ds1 * v_k + ds * v_kp1) / (ds1 + ds));
#undef WT
#undef WT2
            }
            const real svb = p.sigma_vb[k], svb1 = p.sigma_vb[k + 1];
            const real dif = p.UVFLX_dif_coef[k];
            const real pvb_kp1 = SM(S_PVB, 0, 0), pvb_im1_kp1 = SM(S_PVB, -1, 0), pvb_jm1_kp1 = SM(S_PVB, 0, -1);
            const real phi = SM(S_PHI, 0, 0), pott = SM(S_POTT, 0, 0), pvtf = SM(S_PVTF, 0, 0);

            // =============================== dUFLXdt ===============================
            {
                real bflx = 1.0 / 12.0 * (UF(0, -1) + UF(1, -1) + 2.0 * (UF(0, 0) + UF(1, 0)) + UF(0, 1) + UF(1, 1));
                real bflx_im1 = 1.0 / 12.0 * (UF(-1, -1) + UF(0, -1) + 2.0 * (UF(-1, 0) + UF(0, 0)) + UF(-1, 1) + UF(0, 1));
                real cflx = 1.0 / 12.0 * (VF(-1, -1) + VF(0, -1) + 2.0 * (VF(-1, 0) + VF(0, 0)) + VF(-1, 1) + VF(0, 1));
                real cflx_jp1 = 1.0 / 12.0 * (VF(-1, 0) + VF(0, 0) + 2.0 * (VF(-1, 1) + VF(0, 1)) + VF(-1, 2) + VF(0, 2));
                real dflx_im1 = 1.0 / 24.0 * (VF(-1, -1) + 2.0 * VF(-1, 0) + VF(-1, 1) + UF(-1, -1) + UF(-1, 0) + UF(0, -1) + UF(0, 0));
                real dflx_jp1 = 1.0 / 24.0 * (VF(0, 0) + 2.0 * VF(0, 1) + VF(0, 2) + UF(0, 0) + UF(0, 1) + UF(1, 0) + UF(1, 1));
                real eflx = 1.0 / 24.0 * (VF(0, -1) + 2.0 * VF(0, 0) + VF(0, 1) - UF(0, -1) - UF(0, 0) - UF(1, -1) - UF(1, 0));
                real eflx_im1_jp1 = 1.0 / 24.0 * (VF(-1, 0) + 2.0 * VF(-1, 1) + VF(-1, 2) - UF(-1, 0) - UF(-1, 1) - UF(0, 0) - UF(0, 1));
                if (lo) { dflx_im1 = 0.0; cflx = 0.0; eflx = 0.0; }
                if (hi) { dflx_jp1 = 0.0; cflx_jp1 = 0.0; eflx_im1_jp1 = 0.0; }
                const real u = U(0, 0), u_im1 = U(-1, 0), u_ip1 = U(1, 0);
                real t = 0.0;
                t = t + (+bflx_im1 * (u_im1 + u) / 2.0 - bflx * (u + u_ip1) / 2.0
                         + cflx * (U(0, -1) + u) / 2.0 - cflx_jp1 * (u + U(0, 1)) / 2.0
                         + dflx_im1 * (U(-1, -1) + u) / 2.0 - dflx_jp1 * (u + U(1, 1)) / 2.0
                         + 1.0 * (+eflx * (U(1, -1) + u) / 2.0 - eflx_im1_jp1 * (u + U(-1, 1)) / 2.0));
                t = t + (wu_k - wu_kp1) / ds;
                t = t + CON_RE * dlon * dlat / 2.0 *
                            (colp_im1 * (V(-1, 0) + V(-1, 1)) / 2.0 * (corf_is * CON_RE * cos_is + (u_im1 + u) / 2.0 * sin_is) +
                             colp * (V(0, 0) + V(0, 1)) / 2.0 * (corf_is * CON_RE * cos_is + (u + u_ip1) / 2.0 * sin_is));
                t = t + pre_grad(phi, SM(S_PHI, -1, 0), colp, colp_im1, pott, SM(S_POTT, -1, 0),
                                 pvtf, SM(S_PVTF, -1, 0), pvb_k, pvb_im1_k, pvb_im1_kp1, pvb_kp1,
                                 ds, svb, svb1, dyis);
                if (dif > 0.0)
                    t = t + dif * (UF(-1, 0) + UF(1, 0) + UF(0, -1) + UF(0, 1) - 4.0 * UF(0, 0));
                if (EULER) u_out[at(g, i, j, k)] = u_old[q] * cao_is / ca_is + dt * t / ca_is;
                else p.dUFLXdt[q] = t;
            }
            // =============================== dVFLXdt ===============================
            if (v_on) {
                const real rflx = 1.0 / 12.0 * (VF(-1, 0) + VF(-1, 1) + 2.0 * (VF(0, 0) + VF(0, 1)) + VF(1, 0) + VF(1, 1));
                const real rflx_jm1 = 1.0 / 12.0 * (VF(-1, -1) + VF(-1, 0) + 2.0 * (VF(0, -1) + VF(0, 0)) + VF(1, -1) + VF(1, 0));
                const real qflx = 1.0 / 12.0 * (UF(-1, -1) + UF(-1, 0) + 2.0 * (UF(0, -1) + UF(0, 0)) + UF(1, -1) + UF(1, 0));
                const real qflx_ip1 = 1.0 / 12.0 * (UF(0, -1) + UF(0, 0) + 2.0 * (UF(1, -1) + UF(1, 0)) + UF(2, -1) + UF(2, 0));
                const real sflx_jm1 = 1.0 / 24.0 * (VF(-1, -1) + VF(-1, 0) + VF(0, -1) + VF(0, 0) + UF(-1, -1) + 2.0 * UF(0, -1) + UF(1, -1));
                const real sflx_ip1 = 1.0 / 24.0 * (VF(0, 0) + VF(0, 1) + VF(1, 0) + VF(1, 1) + UF(0, 0) + 2.0 * UF(1, 0) + UF(2, 0));
                const real tflx = 1.0 / 24.0 * (VF(-1, 0) + VF(-1, 1) + VF(0, 0) + VF(0, 1) - UF(-1, 0) - 2.0 * UF(0, 0) - UF(1, 0));
                const real tflx_ip1_jm1 = 1.0 / 24.0 * (VF(0, -1) + VF(0, 0) + VF(1, -1) + VF(1, 0) - UF(0, -1) - 2.0 * UF(1, -1) - UF(2, -1));
                const real v = V(0, 0);
                real t = 0.0;
                t = t + (+rflx_jm1 * (V(0, -1) + v) / 2.0 - rflx * (v + V(0, 1)) / 2.0
                         + qflx * (V(-1, 0) + v) / 2.0 - qflx_ip1 * (v + V(1, 0)) / 2.0
                         + sflx_jm1 * (V(-1, -1) + v) / 2.0 - sflx_ip1 * (v + V(1, 1)) / 2.0
                         + -1.0 * (+tflx * (V(-1, 1) + v) / 2.0 - tflx_ip1_jm1 * (v + V(1, -1)) / 2.0));
                t = t + (wv_k - wv_kp1) / ds;
                const real ub_jm1 = (U(0, -1) + U(1, -1)) / 2.0, ub = (U(0, 0) + U(1, 0)) / 2.0;
                t = t + -CON_RE * dlon * dlat / 2.0 *
                            (colp_jm1 * ub_jm1 * (corf_jm1 * CON_RE * cos_jm1 + ub_jm1 * sin_jm1) +
                             colp * ub * (corf * CON_RE * cos_c + ub * sin_c));
                t = t + pre_grad(phi, SM(S_PHI, 0, -1), colp, colp_jm1, pott, SM(S_POTT, 0, -1),
                                 pvtf, SM(S_PVTF, 0, -1), pvb_k, pvb_jm1_k, pvb_jm1_kp1, pvb_kp1,
                                 ds, svb, svb1, dxjs);
                if (dif > 0.0)
                    t = t + dif * (VF(-1, 0) + VF(1, 0) + VF(0, -1) + VF(0, 1) - 4.0 * VF(0, 0));
                if (EULER) v_out[at(g, i, j, k)] = v_old[q] * cao_js / ca_js + dt * t / ca_js;
                else p.dVFLXdt[q] = t;
            }
#undef UF
#undef VF
#undef U
#undef V
            wu_k = wu_kp1; wv_k = wv_kp1;
            pvb_k = pvb_kp1; pvb_im1_k = pvb_im1_kp1; pvb_jm1_k = pvb_jm1_kp1;
        }
        u_k = u_kp1; v_k = v_kp1;
    }
#undef SM
}


// ------------------------------------------------------------------------------------
// k_momentum_dma: the same computation as k_momentum_fused<true>, restructured around
// LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS without passing through VGPRs):
//   * the inputs of the next DEPTH levels are in flight, 16 B per lane / 1 KB per
//     wave-instruction, into a ring of DEPTH+1 LDS slots while level k is computed: one
//     barrier per level, no vector-memory load inside the loop whose result the loop waits
//     for;
//   * arrays are allocated with longitude i = 1 on a 512-B boundary; tiles start at i0-2, an
//     even element, so every 16-B chunk is aligned in HBM and the tile is a linear LDS image
//     (the DMA writes wave-base + lane*16);
//   * the vertical advection is evaluated one level late ("skewed"): iteration k forms
//     WWIND*U at interface k from the six-point sum carried from iteration k-1 and the
//     winds of levels k-1 and k, which closes the tendency of level k-1.
// OLD_IS_CUR: predictor (u_old is the field the tendencies are evaluated on); otherwise the
// step-start winds are staged as two more (interior-height) tiles.
//
// DEPTH = 2 (production, dry / moist): __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`, i.e.
// it drains every DMA the wave has in flight -- with it a level's data movement is one exposed memory
// latency per level whatever the number of buffers (the kernel without its arithmetic took 1.32 ms of the
// full kernel's 1.36 ms, the arithmetic alone 0.91 ms: profiles/r02_notes.md).  The level loop therefore
// synchronises with a bare s_barrier and counts its own DMA instructions: vmcnt is in order, so
// `s_waitcnt vmcnt(n)` with n = the instructions this wave issued for level k+2 means "my part of level k+1
// has landed" while level k+2 stays in flight.  The two stores of an iteration are issued BEFORE that
// iteration's DMA, so they are older than it and the count is exact.
constexpr int DMAXZ = 160;                        // levels up to which the LDS-DMA kernel is used (beyond: k_momentum_fused)

// Tile rows a staged field needs from row j0-1 on: UFLX, UWIND, WWIND j-1..j+1 (TJ+2 rows), VFLX (and VWIND
// when VFLX is re-formed from it) j-1..j+2 (TJ+3), PHI / POTT / PVTF / PVTFVB only j-1 and j (TJ+1).
// With `fw` (the production instances) PVTF is not staged either: the mid-level Exner value is
// (PVTFVB[k+1] p[k+1] - PVTFVB[k] p[k]) / ((1 + kappa) (p[k+1] - p[k])) (dyn_diagnostics.py:146-153) and p[k+1] - p[k] =
// dsigma[k] COLP, so the three values a thread needs cost it a dozen multiply-adds from the PVTFVB values it
// holds anyway, with a per-column reciprocal 1 / ((1 + kappa) COLP) and 1 / dsigma[k]: 58 instead of 67 tile rows
// through the DMA per level.  The kernel's time is its data movement (profiles/r02_notes.md), the arithmetic
// rides along.  The quotient differs from the stored PVTF by the rounding of p[k+1] - p[k] (~4e-15 relative).
enum { S_UF = 0, S_VF, S_UW, S_VW, S_W, S_PHI, S_POTT, S_PVTF, S_PVB, S_QV, S_QC };
constexpr int DNF = 11;                 // staged field slots of k_momentum_dma: the nine of k_momentum_fused + QV, QC (`mo`, below)
constexpr int mom_rows(int f, int tj, bool fw, bool pl = false, bool wt = false, bool mo = false)
{
    // `mo` (MOIST, with WITH_TEMP): the temperature wavefronts advect QV and QC as well (k_moist_canon's arithmetic): their
    // tiles with one halo row and column on each side, like POTT's
    if (f == S_QV || f == S_QC) return mo ? tj + 2 : 0;
#ifdef DYC_MOM_NOHALO       // experiment builds only (wrong results): no halo rows staged
    return tj;
#endif
    if (f == S_PVTF && fw) return 0;        // formed in the thread from PVTFVB and the column pressure (below)
    if (f == S_PHI && pl) return 0;         // `pl` (PHI_LOCAL): carried up the column by the thread (below)
    if (f == S_POTT && wt) return tj + 2;   // `wt` (WITH_TEMP): the temperature wavefronts read POTT at j+1 as well
    // (WITH_TEMP, 64 x 4: one row more of PVTFVB than anybody reads, so that every staged field has more than 3 x 64 chunks and
    // ALL FOUR momentum wavefronts issue one DMA instruction per field and level: their count per level is then a compile-time
    // constant and the loop's `s_waitcnt vmcnt(n)` an immediate instead of a ten-way ladder of scalar branches -- ~25 of the
    // ~560 instructions a lone momentum wavefront issues per level, one every four clocks)
    if (f == S_PVB && wt && tj == 4) return tj + 2;
    return f == S_UF ? tj + 2 : f == S_VF ? tj + 3 : f == S_UW ? tj + 2 : f == S_VW ? (fw ? tj + 3 : tj + 2) : f == S_W ? tj + 2 : tj + 1;
}
// LDS layout (doubles).  A DMA slot holds the staged fields of one level back to back (each rows x FTW), then the
// two step-start tiles of the corrector (tj rows each); with `fw` (fluxes re-formed from the winds) UFLX / VFLX
// are not staged: they live in their own two-slot ring, written by the DMA wavefronts (flux_own).
constexpr int mom_off(int f, int tj, bool fw, bool pl = false, bool wt = false, bool mo = false)     // offset of field f inside its slot; f = DNF: the step-start tiles
{
    if (fw && f < 2) return f == S_UF ? 0 : (tj + 2) * FTW;
    int o = 0;
    for (int g = (fw ? 2 : 0); g < f; g++) o += mom_rows(g, tj, fw, pl, wt, mo) * FTW;
    return o;
}
// own-cell tiles behind the staged ones (FTI wide): the step-start winds of the corrector; WITH_TEMP: + the step-start POTT of the corrector, + dPOTTdt_RAD
// number of staged fields the momentum wavefronts move when each of them reaches into the last wavefront's 64-chunk slice (then
// every momentum wavefront issues exactly that many DMA instructions per level), else 0; WITH_TEMP only -- elsewhere some
// momentum wavefronts move the step-start tiles as well
constexpr int mom_ni_const(int tj, bool fw, bool pl, bool wt, unsigned mv_t, bool mo = false)
{
    if (!wt) return 0;
    int n = 0;
    for (int f = (fw ? 2 : 0); f < DNF; f++) {
        if ((mv_t >> f) & 1u) continue;
        const int chunks = mom_rows(f, tj, fw, pl, wt, mo) * (FTW / VW);
        if (chunks == 0) continue;
        if (chunks <= (tj - 1) * 64) return 0;
        n++;
    }
    return n;
}
// own-cell tiles: [u_old, v_old (corrector)] [pott_old (corrector, WITH_TEMP)] [qv_old, qc_old (corrector, MOIST)] [dPOTTdt_RAD (WITH_TEMP; staged
// only on a handle with i_radiation)] [TURB_U, TURB_V (`tb`: turbulence-coupled instances)]
constexpr int mom_own_tiles(bool cur, bool wt, bool tb = false, bool mo = false) { return (cur ? 0 : 2) + (wt ? (cur ? 0 : 1) + 1 : 0) + ((mo && !cur) ? 2 : 0) + (tb ? 2 : 0); }
constexpr int mom_dma_slot(int tj, bool fw, bool cur, bool pl = false, bool wt = false, bool tb = false, bool mo = false) { return mom_off(DNF, tj, fw, pl, wt, mo) + mom_own_tiles(cur, wt, tb, mo) * FTI * tj; }
constexpr int mom_flux_slot(int tj, bool fw) { return fw ? (2 * tj + 5) * FTW : 0; }
// ... then (fw) two factor tiles, dyis (tj+2 rows) and dxjs (tj+3 rows), shaped like the UFLX / VFLX tiles
constexpr size_t dma_lds_bytes(int tj, bool fw, bool cur, int depth, bool pl = false, bool wt = false, bool tb = false, bool mo = false)
{
    return ((size_t)(depth + 1) * mom_dma_slot(tj, fw, cur, pl, wt, tb, mo) + 3 * (size_t)mom_flux_slot(tj, fw)) * sizeof(real);
}

#ifndef DYC_DMA_AUX
#define DYC_DMA_AUX 0       // cache policy of the LDS-DMA loads (2 = nt)
#endif
#ifndef DYC_MV_T
#define DYC_MV_T 0          // whole-substep kernel: staged fields whose DMA the TEMPERATURE wavefronts issue (bit f = field f; k_momentum_dma)
#endif
#ifndef DYC_EXP_TEMP
#define DYC_EXP_TEMP 0     // experiment builds only (wrong results): k_temp_canon without the PHIVB chain (1), the interface division (2), the Exner function (3)
#endif
#ifndef DYC_WT_SKIP
#define DYC_WT_SKIP 0       // experiment builds only (wrong results): WITH_TEMP without the temperature (1) / momentum (2) arithmetic
#endif
#ifndef DYC_EARLY_DMA
#define DYC_EARLY_DMA 1     // the first DMA levels of a workgroup ahead of its prologue (k_momentum_dma)
#endif
#ifndef DYC_MOM_MODE
#define DYC_MOM_MODE 0      // experiment builds only: 1 = data movement without the arithmetic, 2 = arithmetic without the data movement
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
// One LDS-DMA instruction: 16 B per lane from `gp` to LDS byte address `lds` + 16 * lane (`lds` wave-uniform, in M0).
// Written as assembly, not as __builtin_amdgcn_global_load_lds: the compiler tracks the builtin's LDS writes and puts
// `s_waitcnt vmcnt(0)` in front of any LDS read it cannot prove disjoint from them -- which drains the levels in flight.
// Whether it finds such a read depends on details of the loop body (a build that staged one field less ran 9 % slower
// for one such wait, profiles/r02_notes.md); the level loop does its own counting (wait_vmcnt), so the DMA is kept
// out of the compiler's bookkeeping altogether.
__device__ __forceinline__ void lds_dma16(const void *gp, const void *lds)
{
    // (the low half of a generic pointer into LDS is the LDS byte address)
    const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)lds);
#if DYC_DMA_AUX == 2
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gp), "s"(m) : "memory", "m0");
#else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(m) : "memory", "m0");
#endif
}

// The same instruction with the global address as SCALAR base + 32-bit unsigned lane offset (`saddr` form) and the LDS address as
// scalar base + immediate: a staged field's address is (field base + k plane) -- wave-uniform -- plus the lane's chunk offset,
// which never changes; formed as a 64-bit lane address it cost the momentum wavefront a dozen instructions per field and
// level (scalar adds, a 64-bit vector add, shifts for the LDS address), five or six here.  A lone momentum wavefront issues one
// instruction every four clocks, whatever its kind (profiles/r03_notes.md).
template <unsigned LDS_IMM>
__device__ __forceinline__ void lds_dma16s(const void *sbase_p, unsigned voff, unsigned lds_base)
{
    // (the base is wave-uniform by construction; spelt out so that the compiler keeps it in a scalar register pair -- a no-op
    // where it already is one)
    const unsigned long long sb_ = (unsigned long long)(size_t)sbase_p;
    const unsigned lo_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb_), hi_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb_ >> 32));
    const unsigned long long sbase = ((unsigned long long)hi_ << 32) | lo_;
#if DYC_DMA_AUX == 2
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(lds_base), "n"(LDS_IMM) : "memory", "m0", "scc");
#else
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_base), "n"(LDS_IMM) : "memory", "m0", "scc");
#endif
}

// a real that is the same in every lane of the wavefront, moved to scalar registers
__device__ __forceinline__ double wave_uniform(double x)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_uniform(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// `s_waitcnt vmcnt(n)` for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
// Expressions that k_momentum_dma evaluates at two places -- inside the level loop, and in the prologue / epilogue of a
// level chunk for the interface the chunk shares with its neighbour -- with the contraction to FMA written out: left
// to the compiler, a0*b0 + a1*b1 may become fma(a0, b0, a1*b1) at one place and fma(a1, b1, a0*b0) at the other, and the
// result of a step would depend on how the work list cuts the columns (one element in 6e7 did, by one ulp).
__device__ __forceinline__ real mul_rn(real a, real b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ real mix2(real a, real x, real b, real y) { return fma(a, x, mul_rn(b, y)); }
__device__ __forceinline__ real dot6(const real (&w)[6], real x0, real x1, real x2, real x3, real x4, real x5)
{
    real s = mul_rn(w[0], x0);
    s = fma(w[1], x1, s); s = fma(w[2], x2, s); s = fma(w[3], x3, s); s = fma(w[4], x4, s);
    return fma(w[5], x5, s);
}
// One level of the temperature kernel's arithmetic, shared by k_pott_fused and the temperature wavefronts of the
// whole-substep kernel (k_momentum_dma<..., WITH_TEMP>): dPOTTdt (dyn_POTT.py:180-216; horizontal advection, vertical
// advection, diffusion, radiative heating) and the pressure-weighted Euler step (dyn_timestep.py:70-79).  Every
// multiply-add is written out, so both kernels produce the same bits (a latitude band steps on the separate kernels
// and must reproduce the single-GPU run).  The "/ 2" of the four face means is applied once (a power of two passes
// through every rounding unchanged).  A macro, not a function: as a function (arguments in a struct or not) it cost
// k_pott_fused 24 VGPRs and its third wavefront per SIMD (1.06 -> 1.29 ms per launch).
// COLP_ ... RCN_: colp, cnew, cold, c_im1, c_ip1, c_jm1, c_jp1, hr_area = 0.5 / A, r_cnew = 1 / cnew as separate scalars (in a
// struct they cost the same registers as the function)
#define TEMP_NEW_POTT(OUT_, COLP_, CNEW_, COLD_, CIM1_, CIP1_, CJM1_, CJP1_, HRA_, RCN_, UF_, UF1_, VF_, VF1_, PT_, PIM1_, PIP1_, PJM1_, PJP1_, WK_, PVBOK_, WKP1_, PVBOKP1_, RDS_, DIF_, HAS_TURB_, TURB_, HAS_RAD_, RAD_, PO_, DT_) \
    real OUT_;                                                                                                          \
    {                                                                                                                   \
        const real sx_ = fma(-(UF1_), (PT_) + (PIP1_), mul_rn((UF_), (PIM1_) + (PT_)));                                 \
        const real sy_ = fma(-(VF1_), (PT_) + (PJP1_), mul_rn((VF_), (PJM1_) + (PT_)));                                 \
        real t_ = mul_rn(sx_ + sy_, (HRA_));                                                                            \
        t_ = fma(mul_rn((CNEW_), fma((WK_), (PVBOK_), -mul_rn((WKP1_), (PVBOKP1_)))), (RDS_), t_);   /* w = 0 at the model top */ \
        if (HAS_TURB_) t_ = t_ + (TURB_);                                                                               \
        const real dif_ = (DIF_);                                                                                       \
        if (dif_ > 0.0) {                                                                                               \
            real d_ = mul_rn((CIM1_), (PIM1_));                                                                         \
            d_ = fma((CIP1_), (PIP1_), d_); d_ = fma((CJM1_), (PJM1_), d_); d_ = fma((CJP1_), (PJP1_), d_);             \
            d_ = fma((real)-4.0 * (COLP_), (PT_), d_);                                                                  \
            t_ = fma(dif_, d_, t_);                                                                                     \
        }                                                                                                               \
        if (HAS_RAD_) t_ = fma((RAD_), (COLP_), t_);                                                                    \
        OUT_ = fma(mul_rn((DT_), t_), (RCN_), mul_rn(mul_rn((PO_), (COLD_)), (RCN_)));                                  \
    }
// The same arithmetic in two pieces, for the kernels that do not read a stored POTTVB (k_temp_canon, the temperature
// wavefronts of the whole-substep kernel): the interface value between levels k and k+1 (dyn_diagnostics.py:183-186) of the
// state the tendency is evaluated on is formed in the thread from POTT and PVTFVB of the two levels, so the vertical
// advection of a level can only be added once the level above it has been read -- a level is "closed" one iteration late,
// as in the momentum wavefronts.  TEMP_PARTIAL: horizontal advection, diffusion, radiative heating of level k;
// TEMP_CLOSE: + vertical advection between the level's two interfaces (LP_ = mul_rn(w, POTTVB) at the lower one) and the
// pressure-weighted Euler step.  Every rounding written out: both kernels produce the same bits.
#define TEMP_PARTIAL(OUT_, COLP_, CIM1_, CIP1_, CJM1_, CJP1_, HRA_, UF_, UF1_, VF_, VF1_, PT_, PIM1_, PIP1_, PJM1_, PJP1_, DIF_, HAS_RAD_, RAD_) \
    real OUT_;                                                                                                          \
    {                                                                                                                   \
        const real sx_ = fma(-(UF1_), (PT_) + (PIP1_), mul_rn((UF_), (PIM1_) + (PT_)));                                 \
        const real sy_ = fma(-(VF1_), (PT_) + (PJP1_), mul_rn((VF_), (PJM1_) + (PT_)));                                 \
        real t_ = mul_rn(sx_ + sy_, (HRA_));                                                                            \
        const real dif_ = (DIF_);                                                                                       \
        if (dif_ > 0.0) {                                                                                               \
            real d_ = mul_rn((CIM1_), (PIM1_));                                                                         \
            d_ = fma((CIP1_), (PIP1_), d_); d_ = fma((CJM1_), (PJM1_), d_); d_ = fma((CJP1_), (PJP1_), d_);             \
            d_ = fma((real)-4.0 * (COLP_), (PT_), d_);                                                                  \
            t_ = fma(dif_, d_, t_);                                                                                     \
        }                                                                                                               \
        if (HAS_RAD_) t_ = fma((RAD_), (COLP_), t_);                                                                    \
        OUT_ = t_;                                                                                                      \
    }
#define TEMP_CLOSE(OUT_, TPART_, CNEW_, COLD_, RCN_, WK_, VBK_, LP_, RDS_, PO_, DT_)                                    \
    real OUT_;                                                                                                          \
    {                                                                                                                   \
        const real t_ = fma(mul_rn((CNEW_), fma((WK_), (VBK_), -(LP_))), (RDS_), (TPART_));                             \
        OUT_ = fma(mul_rn((DT_), t_), (RCN_), mul_rn(mul_rn((PO_), (COLD_)), (RCN_)));                                  \
    }
// workgroup barrier that publishes this wave's LDS writes but leaves its vector-memory operations in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FROM_WINDS: UFLX and VFLX are not staged (k_continuity_reg<., false> does not store them); their
// tiles are re-formed in LDS every level from the UWIND / VWIND tiles and the 2-D factors,
// UFLX = UCOL * UWIND * dyis and VFLX = VCOL * VWIND * dxjs (operand order of the continuity
// kernel, so the values are the ones it would have stored): 7 instead of 9 fields through the DMA.
// Each DMA lane converts the chunks it moved itself once its own loads have landed (flux_own; its factors
// never change, so they sit in registers), and the one barrier per level publishes the DMA data and the
// flux tiles together.
// TURB: the turbulent tendencies of k_turb_tendencies (own cell, fetched one level ahead into
// registers) join the sums; DEPTH = 1 (the compiler's own waits for those loads drain the DMA anyway).
// The per-level constants (weights of the interface means, 1 / dsigma, sigma_vb, diffusion coefficient: p.mom_c,
// formed by the library when the vertical grid is set) are wave-uniform scalar loads: they use neither LDS nor vmcnt.
// WITH_TEMP (the whole-substep kernel of a single-GPU dry step): TJ more wavefronts per workgroup run k_pott_fused's
// column sweep -- dPOTTdt, Euler step, primary diagnostics of the new column -- on the SAME tiles: the temperature
// kernel reads POTT, UWIND, VWIND, WWIND like the momentum kernel, so one staging serves both (4 of its 5 read passes
// disappear: 14.5 instead of 18.5 field passes per substep).  Each SIMD then holds one momentum and one temperature
// wavefront -- two different instruction streams between the same per-level barriers.  The new PVTFVB and PHI go to
// alternate buffer (the momentum wavefronts of neighbouring tiles still read the old one), which the library swaps in
// after the launch.  Level chunks (MomPlan) as in the momentum-only instances: the temperature sweep carries nothing further
// than one level since it stopped integrating PHI (round 3), the momentum wavefronts of a chunk that ends inside the column
// start their hydrostatic integration from a stored PHIVB plane (k_phivb_rows, launched for the tile rows that are cut).
struct TempArgs { const real *pott_old; real *pott_out, *pvb_out; long long *trace; const real *qv_old, *qc_old; real *qv_out, *qc_out; };
// -DDYC_TRACE (experiment builds only): the wavefronts of ONE workgroup of the whole-substep kernel write their cycle counter at
// four points of every level: [wave][level][point], point 0 = top of the iteration, 1 = stores and DMA issued, 2 = arithmetic
// done, 3 = own DMA landed / flux entries formed (the barrier follows)
#ifdef DYC_TRACE
#define TRACE_PT(k_, pt_) do { if (ta.trace && blockIdx.x == DYC_TRACE && lane == 0) ta.trace[((size_t)wave * 256 + (k_)) * 8 + (pt_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TRACE_PT(k_, pt_) do { } while (0)
#endif
// MOIST (with WITH_TEMP, a handle with i_moist): the temperature wavefronts advance QV and QC too -- dyn_moist.py:201-243 + the QV /
// QC part of dyn_timestep.py, the arithmetic of k_moist_canon level for level (TEMP_PARTIAL / TEMP_CLOSE with the logarithmic
// interface value of dyn_functions.py:70-95; the same bits) -- on two more staged tiles (one halo row / column each side) and, in
// the corrector, two more own-cell tiles: k_moist's launch and its passes over UWIND, VWIND, WWIND disappear (8 -> 5 passes for
// the moisture of a substep), at the price of a temperature wavefront about as long as the momentum wavefront of its SIMD.
template <bool OLD_IS_CUR, int TJ, bool FROM_WINDS, bool TURB = false, int DEPTH = 1, bool PHI_LOCAL = false, bool WITH_TEMP = false, bool MOIST = false>
__global__ __launch_bounds__(FTI *TJ *(WITH_TEMP ? 2 : 1), ((TJ <= 4 && !WITH_TEMP) ? 2 : 1)) void k_momentum_dma(Geo g, Ptrs p, int gbx, MomPlan plan, int jv_lo,
                                                      const real *u_old, const real *v_old,
                                                      real *u_out, real *v_out, real dt, TempArgs ta)
{
    static_assert(!WITH_TEMP || (FROM_WINDS && !TURB && PHI_LOCAL && DEPTH == 2), "the temperature wavefronts read the re-formed flux tiles; nobody stores PHI for the momentum wavefronts");
    static_assert(DEPTH == 1 || DEPTH == 2, "ring of 2 or 3 slots");
    static_assert(!TURB || (TJ == 8 && !WITH_TEMP), "the turbulent tendencies travel as own-cell tiles of the two step-start wavefronts (64 x 8 tiles)");
    static_assert(!PHI_LOCAL || FROM_WINDS, "PHI in the thread needs PVTF in the thread");
    static_assert(!MOIST || WITH_TEMP, "the moisture sweep is the temperature wavefronts'");
#define MROWS(f) mom_rows(f, TJ, FROM_WINDS, PHI_LOCAL, WITH_TEMP, MOIST)
#define MOFF(f) mom_off(f, TJ, FROM_WINDS, PHI_LOCAL, WITH_TEMP, MOIST)
    // segment of the work list this workgroup belongs to (MomPlan)
    const bool seg_b = (int)blockIdx.x >= plan.nblk_a;
    const RowSet trows = seg_b ? plan.b : plan.a;
    const int ksplit = seg_b ? plan.ks_b : plan.ks_a;
    const int gby = trows.n;            // tile rows of this segment (a subset of the band's, see RowSet)
    constexpr int CPR = FTW / VW;                                       // 16-B chunks per tile row
    constexpr int NCH = (TJ + 3) * CPR, NSL = (NCH + 63) / 64;          // chunks of the tallest tile, 64-chunk slices
    constexpr int ON = FTI * TJ, OCPR = FTI / VW, ONSL = ON / VW / 64;  // step-start ("old") tiles: own cells, FTI wide; 64-chunk slices
    static_assert(ON % (VW * 64) == 0, "whole slices");
    // levels of DMA in flight / slots of the ring.  (WITH_TEMP with a third level in flight, and with the flux tiles formed by
    // the temperature wavefronts instead -- which that makes possible --, was tried: +3 % / +10 % per launch; the
    // temperature wavefront is the slower of the two on a SIMD, profiles/r02_notes.md)
    constexpr int AHEAD = DEPTH;
    constexpr int NS = AHEAD + 1;
    constexpr int DSLOT = mom_dma_slot(TJ, FROM_WINDS, OLD_IS_CUR, PHI_LOCAL, WITH_TEMP, TURB, MOIST), FSLOT = mom_flux_slot(TJ, FROM_WINDS);
    constexpr int OFF_OLD = MOFF(DNF);
    // the step-start tiles: two wavefronts of their own (8 rows), or one slice more for the first 2 ONSL wavefronts (4 rows)
    constexpr bool OLD_SHARED = NSL + 2 > TJ;
    static_assert(!(TURB && OLD_SHARED), "TURB_U / TURB_V ride with the two wavefronts that move the step-start tiles");
    // own-cell tiles behind the staged ones: [u_old, v_old (corrector)] [TURB_U, TURB_V (turbulence-coupled)]; OFF_TB = the latter
    constexpr int OFF_TB = OFF_OLD + (OLD_IS_CUR ? 0 : 2) * ON;
    static_assert(NSL <= TJ && 2 * ONSL <= TJ, "not enough wavefronts for the DMA slices");
    extern __shared__ __attribute__((aligned(16))) real dsm[];
    // workgroup -> (tile, level chunk): `ksplit` chunks of the k range per tile give a small
    // latitude band enough workgroups to balance over the CUs (one tile per CU otherwise)
    int bx, by;
    if (!xcd_tile_at((int)blockIdx.x - (seg_b ? plan.nblk_a : 0), gbx, gby * ksplit, bx, by, ksplit == 1 ? plan.patch : 1)) return;
    const int kc = by / gby;
    by = rowset_row(trows, by - kc * gby);
    // level chunks are whole blocks of PHIVB_STRIDE levels: a chunk starts its hydrostatic integration from a stored PHIVB plane
    const int nzb = (g.nz + PHIVB_STRIDE - 1) / PHIVB_STRIDE;
    const int k0 = min(g.nz, PHIVB_STRIDE * (int)((long long)kc * nzb / ksplit)), k1 = min(g.nz, PHIVB_STRIDE * (int)((long long)(kc + 1) * nzb / ksplit));
    if (k1 <= k0) return;
    constexpr int NTW = TJ * (WITH_TEMP ? 2 : 1);                       // wavefronts per workgroup
    const int tx = threadIdx.x;
    const int tid = (int)threadIdx.y * FTI + tx;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool role_t = WITH_TEMP && wave >= TJ;                        // a temperature wavefront (wave-uniform)
    const int ty = role_t ? (int)threadIdx.y - TJ : (int)threadIdx.y;   // its row of the tile
    const int i0 = bx * FTI + 1, j0 = by * TJ + 1;
    const int i = i0 + tx, j = j0 + ty;
    const int P = g.P, nz = g.nz;
    const bool active = (i <= g.nx) && (j <= g.ny);

    // ---- DMA assignment: a staged field is moved in slices of 64 chunks, slice s by the s-th wavefront of the role that
    // moves the field -- the momentum wavefronts, or (WITH_TEMP, bit f of MV_T) the temperature wavefronts: the momentum
    // wavefront is the longer of the two instruction streams of a SIMD (timeline in profiles/r03_notes.md), and issuing a
    // vector-memory instruction costs the issuing wavefront ~100 cycles; the role that moves the winds also forms the flux
    // tiles from them (flux_own).  Without WITH_TEMP the last two waves move the UWIND / VWIND step-start tiles ----
    constexpr unsigned MV_T = WITH_TEMP ? (unsigned)(DYC_MV_T) : 0u;
    static_assert(((MV_T >> S_UW) & 1u) == ((MV_T >> S_VW) & 1u), "one role moves both wind fields (it forms the flux tiles)");
    constexpr bool FLUX_T = ((MV_T >> S_UW) & 1u) != 0;       // the temperature wavefronts form the flux tiles
    const int sl = role_t ? wave - TJ : wave;        // this wavefront's slice
    const int ch = sl * 64 + lane;                  // this lane's chunk (full tiles)
    int goff = 0;                                   // its plane offset
    bool dma_on = false;
    int n_issue = 0;                                // DMA instructions this wave issues per level (wave-uniform)
    if (sl < NSL) {
        dma_on = ch < NCH;
        const int e = VW * (dma_on ? ch : 0), r = e / FTW, c2 = e - r * FTW;
        goff = min(j0 - 1 + r, g.NJ - 1) * P + min(i0 - FXO + c2, P - VW);
#pragma unroll
        for (int f = (FROM_WINDS ? 2 : 0); f < DNF; f++)
            if ((((MV_T >> f) & 1u) != 0) == role_t && sl * 64 < MROWS(f) * CPR) n_issue++;
    }
    // a slice of a step-start tile = VW rows of OCPR chunks (16 B each, i0 is 16-B aligned): lane -> (row, column)
    const int orow = lane / OCPR, ocol = i0 + VW * (lane % OCPR);
    constexpr int NOWN = mom_own_tiles(OLD_IS_CUR, WITH_TEMP, false, MOIST);         // own-cell tiles: [u_old, v_old, pott_old, [qv_old, qc_old,]] [dPOTTdt_RAD]
    const int nown = (WITH_TEMP && !g.i_rad) ? NOWN - 1 : NOWN;        // ... the last one only with radiative heating
    if (WITH_TEMP) {            // the temperature wavefronts move them: slice s (of nown * ONSL) by wavefront TJ + s % TJ
        if (role_t)
            for (int s5 = wave - TJ; s5 < nown * ONSL; s5 += TJ) n_issue++;
    } else {
        if (!OLD_SHARED && wave >= TJ - 2) n_issue = ONSL * ((OLD_IS_CUR ? 0 : 1) + (TURB ? 1 : 0));
        if (!OLD_IS_CUR && OLD_SHARED && wave < 2 * ONSL) n_issue++;
    }
    const real *src[DNF] = {p.UFLX, p.VFLX, p.UWIND, p.VWIND, p.WWIND, p.PHI, p.POTT, p.PVTF, p.PVTFVB, p.QV, p.QC};
    // the momentum wavefronts' DMA instructions per level where that is the same number for all of them (WITH_TEMP: they move
    // staged fields only, and every staged field reaches into the last wavefront's slice); 0: counted per wavefront (n_issue)
    constexpr int NI_CONST = mom_ni_const(TJ, FROM_WINDS, PHI_LOCAL, WITH_TEMP, MV_T, MOIST);
    const real *osrc = OLD_SHARED ? (wave < ONSL ? u_old : v_old) : ((wave == TJ - 2) ? u_old : v_old);

    // the lane's chunk as a byte offset from (field base + k plane - one row): tile column 0 of the first tile is longitude -1 of
    // its row, so `goff` can be -1; biased by a row the offset is unsigned
    const unsigned voff = (unsigned)(goff + P) * (unsigned)sizeof(real);
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dsm);       // LDS byte address of the ring
    auto issue = [&](int k, int slot) {
        real *buf = dsm + slot * DSLOT;
        if (sl < NSL) {
            if (dma_on) {
                const size_t kp = (size_t)k * g.plane - (size_t)P;
                const unsigned lds_sl = lds0 + (unsigned)((slot * DSLOT + sl * 64 * VW) * (int)sizeof(real));
                // (spelt out per field: the field's offset inside the slot is an immediate of the instruction that sets M0)
#define DMA_FIELD(F_)                                                                                                        \
                if ((F_) >= (FROM_WINDS ? 2 : 0) && ((((MV_T >> (F_)) & 1u) != 0) == role_t) && ch < MROWS(F_) * CPR)             \
                    lds_dma16s<(unsigned)(MOFF(F_) * sizeof(real))>(src[F_] + kp, voff, lds_sl);
                DMA_FIELD(0) DMA_FIELD(1) DMA_FIELD(2) DMA_FIELD(3) DMA_FIELD(4) DMA_FIELD(5) DMA_FIELD(6) DMA_FIELD(7) DMA_FIELD(8) DMA_FIELD(9) DMA_FIELD(10)
#undef DMA_FIELD
                static_assert(DNF == 11, "DMA_FIELD list");
            }
        }
        if (WITH_TEMP) {
            if (role_t) {
                // (in the order of mom_own_tiles: [u_old, v_old, pott_old, [qv_old, qc_old,]] dPOTTdt_RAD)
                const real *own[6] = {OLD_IS_CUR ? p.dPOTTdt_RAD : u_old, v_old, ta.pott_old, MOIST ? ta.qv_old : p.dPOTTdt_RAD, ta.qc_old, p.dPOTTdt_RAD};
#pragma unroll
                for (int s5 = 0; s5 < NOWN * ONSL; s5++) {
                    if (s5 % TJ == wave - TJ && s5 < nown * ONSL) {
                        const real *gp = own[s5 / ONSL] + (size_t)k * g.plane + min(j0 + VW * (s5 % ONSL) + orow, g.NJ - 1) * P + ocol;
                        lds_dma16(gp, buf + OFF_OLD + s5 * 64 * VW);
                    }
                }
            }
        } else if (!OLD_IS_CUR && OLD_SHARED) {
            if (wave < 2 * ONSL) {          // slice wave % ONSL of tile wave / ONSL: consecutive wavefronts, consecutive LDS
                const real *gp = osrc + (size_t)k * g.plane + min(j0 + VW * (wave % ONSL) + orow, g.NJ - 1) * P + ocol;
                lds_dma16(gp, buf + OFF_OLD + wave * 64 * VW);
            }
        } else if ((!OLD_IS_CUR || TURB) && wave >= TJ - 2) {
            if (!OLD_IS_CUR) {
                real *ob = buf + OFF_OLD + (wave - (TJ - 2)) * ON;
#pragma unroll
                for (int s5 = 0; s5 < ONSL; s5++) {
                    const real *gp = osrc + (size_t)k * g.plane + min(j0 + VW * s5 + orow, g.NJ - 1) * P + ocol;
                    lds_dma16(gp, ob + s5 * 64 * VW);
                }
            }
            if (TURB) {     // the turbulent tendencies of k_turb_tendencies, own cell: no vector load left in the level loop
                real *ob = buf + OFF_TB + (wave - (TJ - 2)) * ON;
                const real *tsrc = (wave == TJ - 2) ? p.TURB_U : p.TURB_V;
#pragma unroll
                for (int s5 = 0; s5 < ONSL; s5++) {
                    const real *gp = tsrc + (size_t)k * g.plane + min(j0 + VW * s5 + orow, g.NJ - 1) * P + ocol;
                    lds_dma16(gp, ob + s5 * 64 * VW);
                }
            }
        }
    };

    // The first levels go out HERE, ahead of the workgroup's 2-D loads, factor tiles and per-column arithmetic (sines, cosines,
    // reciprocals: ~4 levels' worth of time), so that their latency lies under that prologue instead of behind it: what a
    // workgroup pays once matters where a column is 32 levels or a chunk 4 (-DDYC_EARLY_DMA=0: issued behind the prologue)
#if DYC_EARLY_DMA
    issue(k1 - 1, 0);
    if (AHEAD >= 2 && k1 - 2 >= k0) issue(k1 - 2, 1);
#endif
    // per-level constants (p.mom_c, stride nz + 2): weights c_wa[k], c_wb[k] of the winds of levels k-1 and k in the
    // dsigma-weighted mean at interface k (dyn_functions.py:236-241), dsigma[k] / (dsigma[k] + dsigma[k-1]) and
    // dsigma[k-1] / (...); c_rds = 1 / dsigma; sigma_vb; UVFLX_dif_coef
    // Read through the constant address space: the kernel stores through u_out / v_out, so the compiler cannot prove
    // a plain global load unclobbered and would issue it as a VECTOR load -- whose wait (vmcnt is in order) drains the
    // DMA prefetch.  Constant-address-space loads at a wave-uniform index are scalar loads (s_load, lgkmcnt).
    const int cs = nz + 2;
    typedef const __attribute__((address_space(4))) real cdouble_t;
    cdouble_t *c_wa = (cdouble_t *)p.mom_c, *c_wb = c_wa + cs, *c_rds = c_wa + 2 * cs, *c_svb = c_wa + 3 * cs, *c_dif = c_wa + 4 * cs;
    // factors of the flux-tile entries this lane forms (FROM_WINDS), same element -> grid cell map as its DMA chunks
    // (the column pressures in registers, the metric factors dyis / dxjs in two LDS tiles: the register file is full)
    const int chf = ch;
    const bool flux_lane = sl < NSL && dma_on && (role_t == FLUX_T);
    real f_uc[VW], f_vc[VW];
#pragma unroll
    for (int d = 0; d < VW; d++) { f_uc[d] = 0.0; f_vc[d] = 0.0; }
    real *fac = dsm + NS * DSLOT + 2 * FSLOT;
    // The factor tiles' values are requested here and written to LDS at the END of the role's prologue (fac_publish), so that
    // their round trip to memory lies under the role's own 2-D loads instead of in front of them
    constexpr int NFAC = ((TJ + 3) * FTW + FTI * NTW - 1) / (FTI * NTW);
    real fd_y[NFAC], fd_x[NFAC];
    if (FROM_WINDS) {
#pragma unroll
        for (int n = 0; n < NFAC; n++) {
            const int e = min(tid + n * FTI * NTW, (TJ + 3) * FTW - 1);
            const int r = e / FTW, cc = e - r * FTW, c2 = cc - cc % VW;
            const int gi = min(i0 - FXO + c2, P - VW) + (cc - c2);
            const int o = min(j0 - 1 + r, g.NJ - 1) * P + gi;
            // the reference never fills column nx+2 of its x-staggered grid arrays (main_grid.py:317-357):
            // take the periodic image the UFLX halo copy would have delivered
            fd_y[n] = p.dyis[gi == g.nx + 2 ? o - g.nx : o];
            fd_x[n] = p.dxjs[o];
        }
        if (flux_lane) {
#pragma unroll
            for (int d = 0; d < VW; d++) {
                const int e = VW * chf + d;
                const int r = e / FTW, cc = e - r * FTW, c2 = cc - cc % VW;
                const int o = min(j0 - 1 + r, g.NJ - 1) * P + min(i0 - FXO + c2, P - VW) + (cc - c2);
                f_uc[d] = p.UCOL[o];
                f_vc[d] = p.VCOL[o];
            }
        }
    }
    auto fac_publish = [&]() {
        if (FROM_WINDS) {
#pragma unroll
            for (int n = 0; n < NFAC; n++) {
                const int e = tid + n * FTI * NTW;
                if (e < (TJ + 2) * FTW) fac[mom_off(S_UF, TJ, true) + e] = fd_y[n];
                if (e < (TJ + 3) * FTW) fac[mom_off(S_VF, TJ, true) + e] = fd_x[n];
            }
            lds_barrier();          // the factor tiles are complete before the first flux entries are formed
        }
    };
    // flux tiles of a level (FROM_WINDS): every DMA lane turns the chunks IT moved into UFLX / VFLX
    // entries as soon as its own loads have landed -- no barrier between the DMA and this step, the one at
    // the end of the level publishes both
    auto flux_own = [&](int slot, int fslot) {
        if (FROM_WINDS && flux_lane) {
            const real *nb = dsm + slot * DSLOT;
            real *nf = dsm + NS * DSLOT + fslot * FSLOT;
            const int e = VW * chf;
            // every LDS read first, then the products, then the writes: written as one loop the compiler orders each write
            // before the next element's reads (it cannot tell the tiles apart) -- three LDS round trips one after the other
            // on the critical path between "my DMA has landed" and the barrier.  (A chunk lies inside the UFLX tile or
            // beyond it as a whole; beyond it the reads hit the neighbouring tiles and are discarded.)
            real wu_[VW], wv_[VW], fu_[VW], fv_[VW];
#pragma unroll
            for (int d = 0; d < VW; d++) {
                wu_[d] = nb[mom_off(S_UW, TJ, true) + e + d]; wv_[d] = nb[mom_off(S_VW, TJ, true) + e + d];
                fu_[d] = fac[mom_off(S_UF, TJ, true) + e + d]; fv_[d] = fac[mom_off(S_VF, TJ, true) + e + d];
            }
#pragma unroll
            for (int d = 0; d < VW; d++) { wu_[d] = f_uc[d] * wu_[d] * fu_[d]; wv_[d] = f_vc[d] * wv_[d] * fv_[d]; }
            if (e < (TJ + 2) * FTW) {
#pragma unroll
                for (int d = 0; d < VW; d++) nf[mom_off(S_UF, TJ, true) + e + d] = wu_[d];
            }
#pragma unroll
            for (int d = 0; d < VW; d++) nf[mom_off(S_VF, TJ, true) + e + d] = wv_[d];
        }
    };
    const int lx = tx + FXO, ly = ty + 1; // tile column 0 is longitude i0 - FXO (16-B aligned in HBM)
    // field f of the level being computed: `db` = its DMA slot, `fb` = its flux slot
#define SM(f, a, b) ((FROM_WINDS && (f) < 2) ? fb : db)[MOFF(f) + (ly + (b)) * FTW + (lx + (a))]

    // =====================================================================================================
    // temperature wavefronts (WITH_TEMP): k_pott_fused's bottom-up sweep of one column on the tiles the momentum
    // wavefronts stage -- same helper functions, same bits.  They move the own-cell tiles and keep step with the
    // momentum wavefronts' barriers; the outputs of a level and the DMA of their own tiles go out at the END of the level
    // (below), away from the momentum wavefronts' burst at the top of it.
    if (WITH_TEMP && role_t) {
        const size_t c = at2(g, active ? i : 1, active ? j : 1);
        const real cnew = p.COLP_NEW[c];
        real colp = p.COLP[c], cold = p.COLP_OLD[c], c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1], c_jm1 = p.COLP[c - P], c_jp1 = p.COLP[c + P];
        real hr_area = (real)(0.5 * (1.0 / p.A[c])), r_cnew = (real)(1.0 / cnew);
        real rcolO = canon_rcol(colp);
        cdouble_t *c_sig = (cdouble_t *)p.sigma_vb, *c_rdst = (cdouble_t *)p.rdsigma, *c_tdif = (cdouble_t *)p.POTT_dif_coef;
        const size_t pl = (size_t)g.plane;
        // the surface interface: Exner value of the new column (stored), of the old one (read: the lower interface of the
        // bottom level).  No hydrostatic integration here: the momentum wavefronts of the NEXT substep carry it themselves
        if (active && k1 == nz) ta.pvb_out[c + (size_t)nz * pl] = canon_exner(canon_pk(g.pair_top, c_sig[nz], cnew));
        // Level chunk [k0, k1): nothing in this sweep is carried further than one level -- a level is closed one iteration
        // late, with the interface value between it and the level above.  A chunk that ends inside the column (k1 < nz)
        // therefore starts from level k1's POTT, Exner value and sigma-dot (read here; level k1 itself belongs to the chunk
        // below), and one that begins inside it (k0 > 0) closes its top level with level k0-1's (epilogue).  Both chunks
        // evaluate the interface they share with the same call on the same operands: the same bits as the whole column.
        real pvbO_kp1 = p.PVTFVB[c + (size_t)k1 * pl];
        double pkO_kp1 = canon_pk(g.pair_top, c_sig[k1], colp);
        real pvtfO_kp1 = 0.0, pt_kp1 = 0.0, w_kp1 = 0.0, lp_kp2 = 0.0, tpart_kp1 = 0.0, po_kp1 = 0.0;
        if (k1 < nz) {
            const real pvbO_kp2 = p.PVTFVB[c + (size_t)(k1 + 1) * pl];
            pvtfO_kp1 = canon_pvtf(pvbO_kp2, canon_pk(g.pair_top, c_sig[k1 + 1], colp), pvbO_kp1, pkO_kp1, rcolO, c_rdst[k1]);
            pt_kp1 = p.POTT[c + (size_t)k1 * pl];
            w_kp1 = p.WWIND[c + (size_t)k1 * pl];
        }
        // MOIST: QV, QC carried like POTT -- the level below (its value with the logarithm / reciprocal cache of the interface
        // interpolation), sigma-dot times the interface value at interface k+2, the partial tendency and the step-start value
        // of level k+1
        cdouble_t *c_mdif = (cdouble_t *)p.moist_dif_coef;
        LogCache lqv_kp1 = log_cache((real)0.0), lqc_kp1 = log_cache((real)0.0);
        real lpv_kp2 = 0.0, lpc_kp2 = 0.0, tqv_kp1 = 0.0, tqc_kp1 = 0.0, qvo_kp1 = 0.0, qco_kp1 = 0.0;
        real o_qv = 0.0, o_qc = 0.0;
        if (MOIST && k1 < nz) {
            real qv1 = p.QV[c + (size_t)k1 * pl], qc1 = p.QC[c + (size_t)k1 * pl];
            asm volatile("" : "+v"(qv1), "+v"(qc1));
            lqv_kp1 = log_cache(qv1); lqc_kp1 = log_cache(qc1);
        }
        real o_pott = 0.0, o_pvb = 0.0;       // outputs stored at the end of a level: POTT of level k+1, PVTFVB at interface k
        {   // every value loaded above is complete before the level loop starts (see the momentum wavefronts)
            asm volatile("" : "+v"(colp), "+v"(cold), "+v"(c_im1), "+v"(c_ip1), "+v"(c_jm1), "+v"(c_jp1), "+v"(hr_area), "+v"(r_cnew), "+v"(rcolO), "+v"(pvbO_kp1), "+v"(pvtfO_kp1), "+v"(pt_kp1), "+v"(w_kp1));
        }
        constexpr int OWN_PO = OFF_OLD + 2 * ON, OWN_QV = OFF_OLD + 3 * ON, OWN_QC = OFF_OLD + 4 * ON, OWN_RAD = OFF_OLD + (NOWN - 1) * ON;
        (void)OWN_QV; (void)OWN_QC;
        const bool has_rad = g.i_rad;
#if !DYC_EARLY_DMA
        issue(k1 - 1, 0);
        if (k1 - 2 >= k0) issue(k1 - 2, 1);
#endif
        fac_publish();
        wait_vmcnt(k1 - 2 >= k0 ? n_issue : 0);
        if (FLUX_T) flux_own(0, (k1 - 1) & 1);
        lds_barrier();          // (the flux tiles of level k1-1: whichever role moves the winds)
        int slot = 0;
#pragma unroll 1
        for (int k = k1 - 1; k >= k0; k--) {
            int slot1 = slot + 1;
            if (slot1 == NS) slot1 = 0;
            TRACE_PT(k, 0);
            TRACE_PT(k, 1);
            const real *db = dsm + slot * DSLOT;
            const real *fb = dsm + NS * DSLOT + (k & 1) * FSLOT;
            if (active && DYC_MOM_MODE != 1 && DYC_WT_SKIP != 1) {
                const real pt = SM(S_POTT, 0, 0);
                const real w_k = k > 0 ? SM(S_W, 0, 0) : (real)0.0;
                const real pvbO_k = SM(S_PVB, 0, 0);
                const real po = OLD_IS_CUR ? pt : db[OWN_PO + ty * FTI + tx];
                // old column: Exner value of this level, POTTVB at its lower interface k+1 -- which closes level k+1
                // (k_temp_canon: the same calls, the same bits)
                const double pkO_k = canon_pk(g.pair_top, c_sig[k], colp);
                const real pvtfO_k = canon_pvtf(pvbO_kp1, pkO_kp1, pvbO_k, pkO_k, rcolO, c_rdst[k]);
                real lp_kp1 = 0.0;
                if (k < nz - 1) {
                    const real vb = pottvb_between(pvbO_kp1, pvtfO_k, pt, pvtfO_kp1, pt_kp1);
                    if (k < k1 - 1) {       // (level k1 is closed by the chunk below)
                        TEMP_CLOSE(pottn, tpart_kp1, cnew, cold, r_cnew, w_kp1, vb, lp_kp2, c_rdst[k + 1], po_kp1, dt)
                        o_pott = pottn;
                    }
                    lp_kp1 = mul_rn(w_kp1, vb);
                }
                // tendency of level k on the re-formed mass-flux tiles, without the vertical advection
                TEMP_PARTIAL(tpart, colp, c_im1, c_ip1, c_jm1, c_jp1, hr_area, SM(S_UF, 0, 0), SM(S_UF, 1, 0), SM(S_VF, 0, 0), SM(S_VF, 0, 1),
                             pt, SM(S_POTT, -1, 0), SM(S_POTT, 1, 0), SM(S_POTT, 0, -1), SM(S_POTT, 0, 1), c_tdif[k], has_rad, db[OWN_RAD + ty * FTI + tx])
                o_pvb = canon_exner(canon_pk(g.pair_top, c_sig[k], cnew));
                if (MOIST) {        // k_moist_canon's level, bottom-up: interface k+1 closes level k+1, then level k's partial tendency
                    const real qv = SM(S_QV, 0, 0), qc = SM(S_QC, 0, 0);
                    const real qvo = OLD_IS_CUR ? qv : db[OWN_QV + ty * FTI + tx], qco = OLD_IS_CUR ? qc : db[OWN_QC + ty * FTI + tx];
                    LogCache lqv = log_cache(qv), lqc = log_cache(qc);
                    real lpv_kp1 = 0.0, lpc_kp1 = 0.0;
                    if (k < nz - 1) {
                        const real vbv = varvb_log_cached(lqv_kp1, lqv), vbc = varvb_log_cached(lqc_kp1, lqc);
                        if (k < k1 - 1) {
                            TEMP_CLOSE(qvn, tqv_kp1, cnew, cold, r_cnew, w_kp1, vbv, lpv_kp2, c_rdst[k + 1], qvo_kp1, dt)
                            TEMP_CLOSE(qcn, tqc_kp1, cnew, cold, r_cnew, w_kp1, vbc, lpc_kp2, c_rdst[k + 1], qco_kp1, dt)
                            o_qv = qvn; o_qc = qcn;
                            if (g.i_moist_clip) {       // dyn_timestep.py:145-154
                                if (o_qv < 0.0) o_qv = 0.0;
                                if (o_qc < 0.0) o_qc = 0.0;
                            }
                        }
                        lpv_kp1 = mul_rn(w_kp1, vbv); lpc_kp1 = mul_rn(w_kp1, vbc);
                    }
                    const real mdif = c_mdif[k];
                    TEMP_PARTIAL(tqv, colp, c_im1, c_ip1, c_jm1, c_jp1, hr_area, SM(S_UF, 0, 0), SM(S_UF, 1, 0), SM(S_VF, 0, 0), SM(S_VF, 0, 1),
                                 qv, SM(S_QV, -1, 0), SM(S_QV, 1, 0), SM(S_QV, 0, -1), SM(S_QV, 0, 1), mdif, false, 0.0)
                    TEMP_PARTIAL(tqc, colp, c_im1, c_ip1, c_jm1, c_jp1, hr_area, SM(S_UF, 0, 0), SM(S_UF, 1, 0), SM(S_VF, 0, 0), SM(S_VF, 0, 1),
                                 qc, SM(S_QC, -1, 0), SM(S_QC, 1, 0), SM(S_QC, 0, -1), SM(S_QC, 0, 1), mdif, false, 0.0)
                    lqv_kp1 = lqv; lqc_kp1 = lqc;
                    lpv_kp2 = lpv_kp1; lpc_kp2 = lpc_kp1; tqv_kp1 = tqv; tqc_kp1 = tqc; qvo_kp1 = qvo; qco_kp1 = qco;
                }
                pvbO_kp1 = pvbO_k; pkO_kp1 = pkO_k; pvtfO_kp1 = pvtfO_k; pt_kp1 = pt; w_kp1 = w_k;
                lp_kp2 = lp_kp1; tpart_kp1 = tpart; po_kp1 = po;
            }
            TRACE_PT(k, 2);
            // Everything this wavefront sends to memory goes out HERE, at the end of the level: the momentum wavefronts' burst
            // of DMA and stores at the top of a level fills the queues, and a vector-memory instruction that finds them full
            // blocks its wavefront (timeline in profiles/r02_notes.md: 750 ticks per level with the stores and the DMA at the
            // top, 80 now).  Its own-cell tiles of level k-1 went out one level ago, behind that level's stores: vmcnt(0).
            if (k - 1 >= k0 && DYC_MOM_MODE != 2) {
                wait_vmcnt(0);
                if (FLUX_T) flux_own(slot1, (k - 1) & 1);      // the winds of level k-1 are among what it moved: their flux entries
            }
            TRACE_PT(k, 5);
            // ---- the outputs of this iteration: a whole level to retire before the next wait ----
            asm volatile("" ::: "memory");
            if (active) {
                const size_t q0 = c + (size_t)k * pl;
                if (k < k1 - 1) {
                    __builtin_nontemporal_store(o_pott, ta.pott_out + q0 + pl);
                    if (MOIST) { __builtin_nontemporal_store(o_qv, ta.qv_out + q0 + pl); __builtin_nontemporal_store(o_qc, ta.qc_out + q0 + pl); }
                }
                __builtin_nontemporal_store(o_pvb, ta.pvb_out + q0);
            }
            asm volatile("" ::: "memory");
            if (k - 2 >= k0 && DYC_MOM_MODE != 2) issue(k - 2, (slot + 2) % NS);        // its own-cell tiles of level k-2
            TRACE_PT(k, 3);
            lds_barrier();
            slot = slot1;
        }
        if (active) {           // the top level of the chunk
            real w_top = 0.0, vb_top = 0.0;     // its upper interface is the model top (sigma-dot = 0) ...
            if (k0 > 0) {                       // ... or interface k0, shared with the chunk above: level k0-1's values
                const real pvbO_m = p.PVTFVB[c + (size_t)(k0 - 1) * pl], pt_m = p.POTT[c + (size_t)(k0 - 1) * pl];
                const real pvtfO_m = canon_pvtf(pvbO_kp1, pkO_kp1, pvbO_m, canon_pk(g.pair_top, c_sig[k0 - 1], colp), rcolO, c_rdst[k0 - 1]);
                vb_top = pottvb_between(pvbO_kp1, pvtfO_m, pt_m, pvtfO_kp1, pt_kp1);
                w_top = w_kp1;
            }
            TEMP_CLOSE(pott0, tpart_kp1, cnew, cold, r_cnew, w_top, vb_top, lp_kp2, c_rdst[k0], po_kp1, dt)
            __builtin_nontemporal_store(pott0, ta.pott_out + c + (size_t)k0 * pl);
            if (MOIST) {
                real vbv_top = 0.0, vbc_top = 0.0;
                if (k0 > 0) {
                    LogCache lqv_m = log_cache(p.QV[c + (size_t)(k0 - 1) * pl]), lqc_m = log_cache(p.QC[c + (size_t)(k0 - 1) * pl]);
                    vbv_top = varvb_log_cached(lqv_kp1, lqv_m); vbc_top = varvb_log_cached(lqc_kp1, lqc_m);
                }
                TEMP_CLOSE(qv0, tqv_kp1, cnew, cold, r_cnew, w_top, vbv_top, lpv_kp2, c_rdst[k0], qvo_kp1, dt)
                TEMP_CLOSE(qc0, tqc_kp1, cnew, cold, r_cnew, w_top, vbc_top, lpc_kp2, c_rdst[k0], qco_kp1, dt)
                real qv0c = qv0, qc0c = qc0;
                if (g.i_moist_clip) {
                    if (qv0c < 0.0) qv0c = 0.0;
                    if (qc0c < 0.0) qc0c = 0.0;
                }
                __builtin_nontemporal_store(qv0c, ta.qv_out + c + (size_t)k0 * pl);
                __builtin_nontemporal_store(qc0c, ta.qc_out + c + (size_t)k0 * pl);
            }
        }
        return;
    }
    // =====================================================================================================
    // momentum wavefronts
    // (their instruction stream has the LDS reads the level's arithmetic waits for: issue priority over the temperature
    // wavefront of the same SIMD, -2 % per launch)
    if (WITH_TEMP) __builtin_amdgcn_s_setprio(3);
    // ---- per-column (2-D) quantities ----
    const size_t c = at2(g, active ? i : 1, active ? j : 1);
    const int jg = j + g.joff;
    const bool lo = (jg == 1), hi = (jg == g.ny_global);
    const bool v_on = active && (j >= jv_lo);
    // a wavefront is one latitude row: the polar-row switches of the C, D, E fluxes are wave-uniform weights
    const bool lo_u = __builtin_amdgcn_readfirstlane((int)lo) != 0, hi_u = __builtin_amdgcn_readfirstlane((int)hi) != 0;
    const real w24_lo = lo_u ? 0.0 : 1.0 / 24.0, w48_lo = lo_u ? 0.0 : 1.0 / 48.0;
    const real w24_hi = hi_u ? 0.0 : 1.0 / 24.0, w48_hi = hi_u ? 0.0 : 1.0 / 48.0;
    // Every 2-D value of the prologue is REQUESTED here, in one batch, before anything is computed from them: written where
    // they are used, the ~45 loads went out in five groups with a wait behind each -- five round trips to memory, ~3 levels'
    // worth of time, in each of a launch's 4278 workgroups
    const size_t c1 = c + (size_t)k1 * g.plane;         // interface k1, the lower end of the chunk
    real ca[3][3], a_[3][3], co[3][3];
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const size_t q2 = c + (b - 1) * P + (a - 1);
            ca[b][a] = p.COLP_NEW[q2]; a_[b][a] = p.A[q2]; co[b][a] = p.COLP_OLD[q2];
        }
    const real colp = p.COLP[c], colp_im1 = p.COLP[c - 1], colp_jm1 = p.COLP[c - P];
    const real dlon_c = p.dlon_rad[c], dlat_c = p.dlat_rad[c], dyis_c = p.dyis[c], dxjs_c = p.dxjs[c];
    real pvb_kp1 = p.PVTFVB[c1], pvb_im1_kp1 = p.PVTFVB[c1 - 1], pvb_jm1_kp1 = p.PVTFVB[c1 - P];
    // PHIVB at interface k1 of the own column and its western / southern neighbour: the surface geopotential, or
    // (chunks) what k_phivb_rows / k_temp_canon stored for every PHIVB_STRIDE-th interface.  The western neighbour of i = 1 is
    // taken from its periodic image (misc_boundaries.py:22-31): nobody refreshes the lateral halo of those planes.
    real phivb_c = 0.0, phivb_w = 0.0, phivb_s = 0.0;
    if (PHI_LOCAL) {
        const size_t cw = (active && i == 1) ? c - 1 + g.nx : c - 1;
        if (k1 == nz) { phivb_c = p.HSURF[c]; phivb_w = p.HSURF[cw]; phivb_s = p.HSURF[c - P]; }
        else { const size_t o = (size_t)k1 * g.plane; phivb_c = p.PHIVB[c + o]; phivb_w = p.PHIVB[cw + o]; phivb_s = p.PHIVB[c - P + o]; }
    }
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
        for (int a = 0; a < 3; a++) { ca[b][a] = ca[b][a] * a_[b][a]; co[b][a] = co[b][a] * a_[b][a]; }
    if (PHI_LOCAL && k1 == nz) { phivb_c = phivb_c * CON_G; phivb_w = phivb_w * CON_G; phivb_s = phivb_s * CON_G; }
    // Latitude and Coriolis parameter depend on the row only (a wavefront is one row of a latitude-longitude grid;
    // dycore_set_grid checks it and sends a handle with zonally varying arrays to the stage kernels): the six
    // Coriolis / metric factors live in scalar registers
    // -- formed once per grid by k_row_constants (the same expressions, the same device functions: the same bits as when
    // every workgroup evaluated three sines and cosines per wavefront in its prologue, ~1000 instructions of a launch's
    // 4278 prologues) and read as scalar loads at the wavefront's row
    real kc_is, sin_is, kc_c, sin_c, kc_jm1, sin_jm1;
    {
        const int jr = __builtin_amdgcn_readfirstlane((j <= g.ny) ? j : 1);       // (the row lane 0 -- always an owned column -- computes for)
        cdouble_t *rc = (cdouble_t *)p.row_c;
        kc_is = rc[jr]; sin_is = rc[g.NJ + jr];
        kc_c = rc[2 * g.NJ + jr]; sin_c = rc[3 * g.NJ + jr];
        kc_jm1 = rc[2 * g.NJ + jr - 1]; sin_jm1 = rc[3 * g.NJ + jr - 1];
    }
    // per-column factors of the Coriolis / metric and pressure-gradient terms (formed once; the "/ 2" of the
    // reference's means are folded in exactly, the products are associated differently: <= 1 ulp each)
    const real hsin_is = 0.5 * sin_is, hsin_c = 0.5 * sin_c, hsin_jm1 = 0.5 * sin_jm1;
    real cor_im1, cor_c, cov_jm1, cov_c, pgu_a, pgu_b, pgv_a, pgv_b;
    {
        const real metric = CON_RE * dlon_c * dlat_c / 2.0;
        const real dyis = dyis_c, dxjs = dxjs_c;
        cor_im1 = 0.5 * metric * colp_im1; cor_c = 0.5 * metric * colp;
        cov_jm1 = -0.5 * metric * colp_jm1; cov_c = -0.5 * metric * colp;
        pgu_a = -dyis * ((colp + colp_im1) / 2.0); pgu_b = -dyis * ((colp - colp_im1) * CON_CP / 2.0);
        pgv_a = -dxjs * ((colp + colp_jm1) / 2.0); pgv_b = -dxjs * ((colp - colp_jm1) * CON_CP / 2.0);
    }
    real ca_is, cao_is, ca_js, cao_js;
    real wu[6], wv[6];
    {
#define CA(a, b) ca[(b) + 1][(a) + 1]
#define CO(a, b) co[(b) + 1][(a) + 1]
        if (lo) {
            ca_is = 1.0 / 4.0 * (CA(-1, 1) + CA(0, 1) + CA(-1, 0) + CA(0, 0));
            cao_is = 1.0 / 4.0 * (CO(-1, 1) + CO(0, 1) + CO(-1, 0) + CO(0, 0));
        } else if (hi) {
            ca_is = 1.0 / 4.0 * (CA(-1, -1) + CA(0, -1) + CA(-1, 0) + CA(0, 0));
            cao_is = 1.0 / 4.0 * (CO(-1, -1) + CO(0, -1) + CO(-1, 0) + CO(0, 0));
        } else {
            ca_is = 1.0 / 8.0 * (CA(-1, 1) + CA(0, 1) + 2.0 * CA(-1, 0) + 2.0 * CA(0, 0) + CA(-1, -1) + CA(0, -1));
            cao_is = 1.0 / 8.0 * (CO(-1, 1) + CO(0, 1) + 2.0 * CO(-1, 0) + 2.0 * CO(0, 0) + CO(-1, -1) + CO(0, -1));
        }
        ca_js = 1.0 / 8.0 * (CA(1, -1) + CA(1, 0) + 2.0 * CA(0, -1) + 2.0 * CA(0, 0) + CA(-1, -1) + CA(-1, 0));
        cao_js = 1.0 / 8.0 * (CO(1, -1) + CO(1, 0) + 2.0 * CO(0, -1) + 2.0 * CO(0, 0) + CO(-1, -1) + CO(-1, 0));
        // weights of the six-point COLP_NEW * A * WWIND means at the u- and v-point (dyn_functions.py:211-270)
        // with 1/8 (1/4 at a polar row, "rigid wall") and the doubled centre pair folded in: powers of two,
        // exact.  A polar row gives its two outside points weight 0 (their WWIND halo values are finite).
        const real ws = lo ? 0.0 : (hi ? 0.25 : 0.125), wn = hi ? 0.0 : (lo ? 0.25 : 0.125);
        wu[0] = wn * CA(-1, 1); wu[1] = wn * CA(0, 1); wu[2] = 0.25 * CA(-1, 0); wu[3] = 0.25 * CA(0, 0);
        wu[4] = ws * CA(-1, -1); wu[5] = ws * CA(0, -1);
        wv[0] = 0.125 * CA(1, -1); wv[1] = 0.125 * CA(1, 0); wv[2] = 0.25 * CA(0, -1); wv[3] = 0.25 * CA(0, 0);
        wv[4] = 0.125 * CA(-1, -1); wv[5] = 0.125 * CA(-1, 0);
#undef CA
#undef CO
    }
    // Divisions by per-column (ca_is, ca_js) and per-level (dsigma) quantities are multiplications by
    // reciprocals formed once: <= 1 ulp from the reference's x / y per operation, 8 fp64 divisions
    // (~80 instructions) less per level -- -9 % kernel time.  The stage kernels keep the divisions.
    // ... and the pressure-weighted Euler step X_new = X_old * P_old / P + dt * dX / P (dyn_timestep.py:34-79) as
    // X_old * eu1 + dX * eu2 with eu1 = P_old * (1 / P), eu2 = dt * (1 / P) per column
    const real eu1 = cao_is * (1.0 / ca_is), eu2 = dt * (1.0 / ca_is), ev1 = cao_js * (1.0 / ca_js), ev2 = dt * (1.0 / ca_js);
    // PVTF in the thread (FROM_WINDS): 1 / ((1 + kappa) COLP) at the own cell and its western / southern neighbour
    const real rcp_c = canon_rcol(colp), rcp_im1 = canon_rcol(colp_im1), rcp_jm1 = canon_rcol(colp_jm1);

    // carried across levels.  The column is walked BOTTOM-UP (level k1-1 first), the direction of the hydrostatic
    // integration, so that PHI can be carried along (PHI_LOCAL); the vertical advection of a level is closed one
    // iteration late, when the winds of the level above it are at hand for its upper interface
    real su_kp1 = 0.0, sv_kp1 = 0.0;      // six-point COLP_NEW*A*WWIND sums at interface k+1
    real wu_b = 0.0, wv_b = 0.0;          // WWIND*U, WWIND*V at interface k+2 (0 at the surface)
    real u_kp1 = 0.0, v_kp1 = 0.0, uo_kp1 = 0.0, vo_kp1 = 0.0;
    real tu_kp1 = 0.0, tv_kp1 = 0.0;      // tendencies of level k+1 without the vertical advection
    if (k1 < nz) {      // a chunk that ends inside the column: interface k1 and the winds below it
#define WG(a, b) p.WWIND[c1 + (b) * P + (a)]
        su_kp1 = dot6(wu, WG(-1, 1), WG(0, 1), WG(-1, 0), WG(0, 0), WG(-1, -1), WG(0, -1));
        sv_kp1 = dot6(wv, WG(1, -1), WG(1, 0), WG(0, -1), WG(0, 0), WG(-1, -1), WG(-1, 0));
#undef WG
        u_kp1 = p.UWIND[c1];
        v_kp1 = p.VWIND[c1];
    }

    // the values loaded above are complete before the level loop starts: left to itself the compiler waits for them at
    // their first use, inside the loop, where a vector-memory wait drains the DMA levels in flight
    asm volatile("" : "+v"(su_kp1), "+v"(sv_kp1), "+v"(u_kp1), "+v"(v_kp1), "+v"(pvb_kp1), "+v"(pvb_im1_kp1), "+v"(pvb_jm1_kp1), "+v"(phivb_c), "+v"(phivb_w), "+v"(phivb_s));
    // ---- prologue: the first DEPTH levels go out; level k1-1 lands and its flux tiles are formed ----
#if !DYC_EARLY_DMA
    issue(k1 - 1, 0);
    if (AHEAD >= 2 && k1 - 2 >= k0) issue(k1 - 2, 1);
#endif
    if (AHEAD >= 3 && k1 - 3 >= k0) issue(k1 - 3, 2);
    fac_publish();
    if (n_issue > 0) wait_vmcnt((AHEAD >= 2 && k1 - AHEAD >= k0) ? n_issue : 0);        // everything but the last level issued has landed
    flux_own(0, (k1 - 1) & 1);
    lds_barrier();
    int slot = 0;                       // ring slot of level k
#pragma unroll 1
    for (int k = k1 - 1; k >= k0; k--) {
        int slot1 = slot + 1;           // ... of level k-1
        if (slot1 == NS) slot1 = 0;
        // DEPTH 1: level k-1 goes out first thing.  DEPTH 2: level k-2 goes out after this iteration's two stores
        // (below), so that every older vector-memory operation of the wave precedes it and `vmcnt(n_issue)` at the
        // end of the iteration means exactly "level k-1 has landed"
        if (DEPTH == 1 && k - 1 >= k0 && DYC_MOM_MODE != 2) issue(k - 1, slot1);
        if (WITH_TEMP) TRACE_PT(k, 0);
        const real *db = dsm + slot * DSLOT;
        const real *fb = dsm + NS * DSLOT + (k & 1) * FSLOT;
        (void)fb;
        real u = 0.0, v = 0.0, uo = 0.0, vo = 0.0;
        real su_k = 0.0, sv_k = 0.0;
        if (DYC_MOM_MODE == 1) {
            if (active) {
                const size_t qm = c + (size_t)k * g.plane;
                u_out[qm] = SM(S_UW, 0, 0) + SM(S_POTT, 0, 0) + SM(S_UF, 1, 0);
                if (v_on) v_out[qm] = SM(S_VW, 0, 0) + SM(S_PVB, 0, 0) + SM(S_VF, 0, 1) + (OLD_IS_CUR ? 0.0 : db[OFF_OLD + ty * FTI + tx] + db[OFF_OLD + ON + ty * FTI + tx]);
            }
        } else if (active) {
            u = SM(S_UW, 0, 0); v = SM(S_VW, 0, 0);
            uo = u; vo = v;
            if (!OLD_IS_CUR) {
                uo = db[OFF_OLD + ty * FTI + tx];
                vo = db[OFF_OLD + ON + ty * FTI + tx];
            }
            // ---- interface k+1 closes level k+1 ----
            if (k + 1 < nz) {
                const real wa = c_wa[k + 1], wb = c_wb[k + 1];
                const real wu_t = mul_rn(su_kp1, mix2(wa, u, wb, u_kp1));
                const real wv_t = mul_rn(sv_kp1, mix2(wa, v, wb, v_kp1));
                if (k + 1 < k1) {
                    const size_t qm = c + (size_t)(k + 1) * g.plane;
                    const real rds1 = c_rds[k + 1];
                    const real tu = fma(wu_t - wu_b, rds1, tu_kp1);
                    u_out[qm] = mix2(uo_kp1, eu1, tu, eu2);
                    if (v_on) {
                        const real tv = fma(wv_t - wv_b, rds1, tv_kp1);
                        v_out[qm] = mix2(vo_kp1, ev1, tv, ev2);
                    }
                }
                wu_b = wu_t; wv_b = wv_t;
            }
        }
        if (WITH_TEMP) TRACE_PT(k, 4);
        if (DEPTH == 2) {
            asm volatile("" ::: "memory");      // the stores above stay above the DMA below
            if (k - AHEAD >= k0 && DYC_MOM_MODE != 2) issue(k - AHEAD, (slot + AHEAD) % NS);
        }
        if (WITH_TEMP) TRACE_PT(k, 1);
        if (DYC_MOM_MODE != 1 && active && !(WITH_TEMP && DYC_WT_SKIP == 2)) {
            real uf[3][4], vf[4][3], uw[3][3], vw[3][3];
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int a = 0; a < 4; a++) uf[b][a] = SM(S_UF, a - 1, b - 1);
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int a = 0; a < 3; a++) vf[b][a] = SM(S_VF, a - 1, b - 1);
#define UF(a, b) uf[(b) + 1][(a) + 1]
#define VF(a, b) vf[(b) + 1][(a) + 1]
#define U(a, b) uw[(b) + 1][(a) + 1]
#define V(a, b) vw[(b) + 1][(a) + 1]
            // ---- the sixteen Arakawa-C flux means B,C,D,E / R,Q,S,T (dyn_UVFLX_prepare.py:249-439) from shared pair
            // sums of the UFLX (3 x 4) and VFLX (4 x 3) neighbourhoods.  They carry HALF their weight (1/24, 1/48):
            // the "/ 2" of the wind mean in every advective term is applied here, where it is free -- a power of
            // two passes through every rounding unchanged.  The sums are associated pairwise instead of left to
            // right (<= 1 ulp per sum from the reference's order); 80 instead of ~125 fp64 instructions.
            // Everything the flux neighbourhoods feed is formed first, so that their 24 registers are free again
            // before the wind neighbourhoods are read.
            const real hx_m_m = UF(-1, -1) + UF(0, -1), hx_0_m = UF(0, -1) + UF(1, -1), hx_p_m = UF(1, -1) + UF(2, -1);
            const real hx_m_0 = UF(-1, 0) + UF(0, 0), hx_0_0 = UF(0, 0) + UF(1, 0), hx_p_0 = UF(1, 0) + UF(2, 0);
            const real hx_m_p = UF(-1, 1) + UF(0, 1), hx_0_p = UF(0, 1) + UF(1, 1);
            const real gy_m_m = VF(-1, -1) + VF(-1, 0), gy_0_m = VF(0, -1) + VF(0, 0), gy_p_m = VF(1, -1) + VF(1, 0);
            const real gy_m_0 = VF(-1, 0) + VF(-1, 1), gy_0_0 = VF(0, 0) + VF(0, 1), gy_p_0 = VF(1, 0) + VF(1, 1);
            const real gy_m_p = VF(-1, 1) + VF(-1, 2), gy_0_p = VF(0, 1) + VF(0, 2);
            const real gx_m = VF(-1, -1) + VF(0, -1), gx_0 = VF(-1, 0) + VF(0, 0), gx_p = VF(-1, 1) + VF(0, 1), gx_pp = VF(-1, 2) + VF(0, 2);
            const real hy_m = UF(-1, -1) + UF(-1, 0), hy_0 = UF(0, -1) + UF(0, 0), hy_p = UF(1, -1) + UF(1, 0), hy_pp = UF(2, -1) + UF(2, 0);
            const real lap_u = UF(-1, 0) + UF(1, 0) + UF(0, -1) + UF(0, 1) - 4.0 * UF(0, 0);     // five-point diffusion stencils
            const real lap_v = VF(-1, 0) + VF(1, 0) + VF(0, -1) + VF(0, 1) - 4.0 * VF(0, 0);
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
                for (int a = 0; a < 3; a++) { uw[b][a] = SM(S_UW, a - 1, b - 1); vw[b][a] = SM(S_VW, a - 1, b - 1); }
            // ---- six-point sums at interface k, the upper one of this level (dyn_functions.py:211-270); sigma-dot
            // is zero at the model top ----
            if (k > 0) {
                const real w_c = SM(S_W, 0, 0), w_w = SM(S_W, -1, 0), w_s = SM(S_W, 0, -1), w_sw = SM(S_W, -1, -1);
                su_k = dot6(wu, SM(S_W, -1, 1), SM(S_W, 0, 1), w_w, w_c, w_sw, w_s);
                sv_k = dot6(wv, SM(S_W, 1, -1), SM(S_W, 1, 0), w_s, w_c, w_sw, w_w);
            }
            const real svb = c_svb[k], svb1 = c_svb[k + 1];
            const real dif = c_dif[k];
            const real rds = c_rds[k];
            const real pvb_k = SM(S_PVB, 0, 0), pvb_im1_k = SM(S_PVB, -1, 0), pvb_jm1_k = SM(S_PVB, 0, -1);
            const real pott = SM(S_POTT, 0, 0), pott_im1 = SM(S_POTT, -1, 0), pott_jm1 = SM(S_POTT, 0, -1);
            real pvtf, pvtf_im1, pvtf_jm1;
            if (FROM_WINDS) {       // the canonical forms (kernels_stage.h): the same bits in every kernel, chunk and band
                pvtf = canon_pvtf(pvb_kp1, canon_pk(g.pair_top, svb1, colp), pvb_k, canon_pk(g.pair_top, svb, colp), rcp_c, rds);
                pvtf_im1 = canon_pvtf(pvb_im1_kp1, canon_pk(g.pair_top, svb1, colp_im1), pvb_im1_k, canon_pk(g.pair_top, svb, colp_im1), rcp_im1, rds);
                pvtf_jm1 = canon_pvtf(pvb_jm1_kp1, canon_pk(g.pair_top, svb1, colp_jm1), pvb_jm1_k, canon_pk(g.pair_top, svb, colp_jm1), rcp_jm1, rds);
            } else {
                pvtf = SM(S_PVTF, 0, 0); pvtf_im1 = SM(S_PVTF, -1, 0); pvtf_jm1 = SM(S_PVTF, 0, -1);
            }
            // PHI at the own cell and its western / southern neighbour (dyn_diagnostics.py:160-174): staged, or
            // (PHI_LOCAL) carried up the column from the surface geopotential -- PHIVB[k+1] is at hand from the level
            // below, PHI[k] = PHIVB[k+1] - cp POTT (PVTF - PVTFVB[k+1]), PHIVB[k] = PHI[k] - cp POTT (PVTFVB[k] - PVTF),
            // the expressions of diag_level_pre on the thread's own PVTF
            real phi, phi_im1, phi_jm1;
            if (PHI_LOCAL) {
                phi = canon_phi(phivb_c, pott, pvtf, pvb_kp1);
                phi_im1 = canon_phi(phivb_w, pott_im1, pvtf_im1, pvb_im1_kp1);
                phi_jm1 = canon_phi(phivb_s, pott_jm1, pvtf_jm1, pvb_jm1_kp1);
                phivb_c = canon_phivb(phi, pott, pvb_k, pvtf);
                phivb_w = canon_phivb(phi_im1, pott_im1, pvb_im1_k, pvtf_im1);
                phivb_s = canon_phivb(phi_jm1, pott_jm1, pvb_jm1_k, pvtf_jm1);
            } else {
                phi = SM(S_PHI, 0, 0); phi_im1 = SM(S_PHI, -1, 0); phi_jm1 = SM(S_PHI, 0, -1);
            }
            constexpr real W24 = 1.0 / 24.0, W48 = 1.0 / 48.0;
            const real prs = pott * rds;                 // POTT / dsigma of the own cell (pressure-gradient terms)
            const real own = svb1 * (pvb_kp1 - pvtf) + svb * (pvtf - pvb_k);
            // =============================== dUFLXdt (level k, without vertical advection) ====
            {
                const real bflx = W24 * (hx_0_m + 2.0 * hx_0_0 + hx_0_p);
                const real bflx_im1 = W24 * (hx_m_m + 2.0 * hx_m_0 + hx_m_p);
                const real cflx = w24_lo * (gx_m + 2.0 * gx_0 + gx_p);          // 0 at the southern polar row (dyn_UFLX.py:377-384)
                const real cflx_jp1 = w24_hi * (gx_0 + 2.0 * gx_p + gx_pp);     // 0 at the northern polar row
                const real dflx_im1 = w48_lo * ((gy_m_m + gy_m_0) + (hx_m_m + hx_m_0));
                const real dflx_jp1 = w48_hi * ((gy_0_0 + gy_0_p) + (hx_0_0 + hx_0_p));
                const real eflx = w48_lo * ((gy_0_m + gy_0_0) - (hx_0_m + hx_0_0));
                const real eflx_im1_jp1 = w48_hi * ((gy_m_0 + gy_m_p) - (hx_m_0 + hx_m_p));
                const real u_im1 = U(-1, 0), u_ip1 = U(1, 0);
                const real um = u_im1 + u, up = u + u_ip1;
                real t = (+bflx_im1 * um - bflx * up
                            + cflx * (U(0, -1) + u) - cflx_jp1 * (u + U(0, 1))
                            + dflx_im1 * (U(-1, -1) + u) - dflx_jp1 * (u + U(1, 1))
                            + (+eflx * (U(1, -1) + u) - eflx_im1_jp1 * (u + U(-1, 1))));
                // Coriolis and metric term (dyn_UFLX.py:39-66); cor_im1 = metric / 2 * COLP[i-1], hsin_is = sin / 2
                t = t + (cor_im1 * (V(-1, 0) + V(-1, 1)) * (kc_is + um * hsin_is) +
                         cor_c * (V(0, 0) + V(0, 1)) * (kc_is + up * hsin_is));
                // pressure gradient (dyn_functions.py:177-207) with the per-column factors pg_a = -dy (COLP + COLP[i-1]) / 2,
                // pg_b = -dy (COLP - COLP[i-1]) cp / 2 formed once and POTT / dsigma as POTT * (1 / dsigma)
                t = t + ((phi - phi_im1) * pgu_a +
                         pgu_b * ((pott_im1 * rds) * (svb1 * (pvb_im1_kp1 - pvtf_im1) + svb * (pvtf_im1 - pvb_im1_k)) +
                                  prs * own));
                t = t + dif * lap_u;      // dif >= 0; 0 adds nothing
                if (TURB) t = t + db[OFF_TB + ty * FTI + tx];
                tu_kp1 = t;
            }
            // =============================== dVFLXdt ===============================
            if (v_on) {
                const real rflx = W24 * (gy_m_0 + 2.0 * gy_0_0 + gy_p_0);
                const real rflx_jm1 = W24 * (gy_m_m + 2.0 * gy_0_m + gy_p_m);
                const real qflx = W24 * (hy_m + 2.0 * hy_0 + hy_p);
                const real qflx_ip1 = W24 * (hy_0 + 2.0 * hy_p + hy_pp);
                const real sflx_jm1 = W48 * ((gy_m_m + gy_0_m) + (hx_m_m + hx_0_m));
                const real sflx_ip1 = W48 * ((gy_0_0 + gy_p_0) + (hx_0_0 + hx_p_0));
                const real tflx = W48 * ((gy_m_0 + gy_0_0) - (hx_m_0 + hx_0_0));
                const real tflx_ip1_jm1 = W48 * ((gy_0_m + gy_p_m) - (hx_0_m + hx_p_m));
                real t = (+rflx_jm1 * (V(0, -1) + v) - rflx * (v + V(0, 1))
                            + qflx * (V(-1, 0) + v) - qflx_ip1 * (v + V(1, 0))
                            + sflx_jm1 * (V(-1, -1) + v) - sflx_ip1 * (v + V(1, 1))
                            - (+tflx * (V(-1, 1) + v) - tflx_ip1_jm1 * (v + V(1, -1))));
                // dyn_VFLX.py:39-64; us = 2 x the mean u at the v-point, cov_* = -metric / 2 * COLP, hsin = sin / 2
                const real us_jm1 = U(0, -1) + U(1, -1), us = U(0, 0) + U(1, 0);
                t = t + (cov_jm1 * us_jm1 * (kc_jm1 + us_jm1 * hsin_jm1) + cov_c * us * (kc_c + us * hsin_c));
                t = t + ((phi - phi_jm1) * pgv_a +
                         pgv_b * ((pott_jm1 * rds) * (svb1 * (pvb_jm1_kp1 - pvtf_jm1) + svb * (pvtf_jm1 - pvb_jm1_k)) +
                                  prs * own));
                t = t + dif * lap_v;
                if (TURB) t = t + db[OFF_TB + ON + ty * FTI + tx];
                tv_kp1 = t;
            }
#undef UF
#undef VF
#undef U
#undef V
            u_kp1 = u; v_kp1 = v; uo_kp1 = uo; vo_kp1 = vo;
            su_kp1 = su_k; sv_kp1 = sv_k;
            pvb_kp1 = pvb_k; pvb_im1_kp1 = pvb_im1_k; pvb_jm1_kp1 = pvb_jm1_k;
        }
        if (WITH_TEMP) TRACE_PT(k, 2);
        // this wave's part of level k-1 has landed (the deeper prefetch stays in flight); form its flux entries
        if (k - 1 >= k0 && DYC_MOM_MODE != 2) {
            // level k - AHEAD + 1 has landed (a wavefront that moves nothing does not wait for its stores either)
            if (NI_CONST > 0) {             // (every momentum wavefront issues NI_CONST instructions per level: an immediate)
                if (DEPTH == 2 && k - AHEAD >= k0) wait_vmcnt(NI_CONST);
                else wait_vmcnt(0);
            } else if (n_issue > 0) wait_vmcnt((DEPTH == 2 && k - AHEAD >= k0) ? n_issue : 0);
            if (WITH_TEMP) TRACE_PT(k, 5);
            flux_own(slot1, (k - 1) & 1);
        }
        if (WITH_TEMP) TRACE_PT(k, 3);
        lds_barrier();         // level k-1 is in LDS for every wave; slot `slot` is free again
        slot = slot1;
    }
    // ---- the top level of the chunk: its upper interface is k0 (WWIND*U = 0 at the model top) ----
    if (active) {
        const size_t qm = c + (size_t)k0 * g.plane;
        real wu_t = 0.0, wv_t = 0.0;
        if (k0 > 0) {
            wu_t = mul_rn(su_kp1, mix2(c_wa[k0], p.UWIND[qm - g.plane], c_wb[k0], u_kp1));
            wv_t = mul_rn(sv_kp1, mix2(c_wa[k0], p.VWIND[qm - g.plane], c_wb[k0], v_kp1));
        }
        u_out[qm] = mix2(uo_kp1, eu1, fma(wu_t - wu_b, c_rds[k0], tu_kp1), eu2);
        if (v_on) v_out[qm] = mix2(vo_kp1, ev1, fma(wv_t - wv_b, c_rds[k0], tv_kp1), ev2);
    }
#undef SM
#undef MROWS
#undef MOFF
}
// ------------------------------------------------------------------------------------
// k_turb_tendencies: the vertical-turbulence terms of the momentum, temperature and moisture
// tendencies for the fused step of a turbulence-coupled handle (i_turb), evaluated once per substep
// on the state the tendencies are taken at and kept as five raw 3-D fields that the fused kernels
// add at their own cell:
//   TURB_U, TURB_V : dyn_UVFLX_prepare.py:318-439 (KMOM_dUWINDdz, KMOM_dVWINDdz; dyn_functions.py:276-377)
//                    + dyn_UFLX.py:158-167 / dyn_VFLX.py:157-166 -- the expressions of k_prepare<true> and
//                    k_uflx/vflx_tendency<true>, the interface values carried instead of stored
//   TURB_T, TURB_QV, TURB_QC : dyn_functions.py:26-67 (turb_flux) as in k_pott_tendency<true>, k_moist<true>
// One thread per owned column, levels top-down.  `rs`: the rows of this launch (a latitude band evaluates the rows
// next to a neighbour once that neighbour's rows have arrived: every term reads rows j-1 .. j+1 at most).
// the per-row Coriolis / metric constants of k_momentum_dma (Ptrs::row_c): one thread per row, column 1 (dycore_set_grid has
// checked that the four arrays do not vary along a row)
__global__ void k_row_constants(Geo g, Ptrs p, real *out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.NJ) return;
    const size_t c = at2(g, 1, j);
    const real lat_is = p.lat_is_rad[c], lat = p.lat_rad[c];
    out[j] = p.corf_is[c] * CON_RE * cos(lat_is);
    out[g.NJ + j] = sin(lat_is);
    out[2 * g.NJ + j] = p.corf[c] * CON_RE * cos(lat);
    out[3 * g.NJ + j] = sin(lat);
}
__global__ __launch_bounds__(TI *TJ) void k_turb_tendencies(Geo g, Ptrs p, int jv_lo, int lin, RowSet rs)
{
    int i, j;
    {
        Geo gs = g;
        gs.ny = rs.n;                       // owned_cell over the nx x rs.n cells of the launch
        if (!owned_cell(gs, lin, i, j)) return;
        j = 1 + rowset_row(rs, j - 1);
    }
    const int P = g.P, nz = g.nz;
    const size_t c = at2(g, i, j), pl = (size_t)g.plane;
    const int jg = j + g.joff;
    const bool lo = (jg == 1), hi = (jg == g.ny_global);
    const bool v_on = j >= jv_lo;
    const real colp = p.COLP[c];
    constexpr real RG = 1.0 / CON_G;
    // Everything that depends on the column only is formed once: COLP * A of the 3 x 3 neighbourhood folded into the
    // weights of the six-point means at the u-point (columns i-1, i; rows j-1 .. j+1; four points with weight 1/4
    // at a polar row: zero weights) and at the v-point (rows j-1, j; columns i-1 .. i+1), and the plain six-point
    // weights of interp_VAR_ds (dyn_functions.py:383-424) with its division by g as a factor.  Divisions by g are
    // multiplications by 1/g; the products are associated differently from the stage kernels (a few ulp).
    // point order u: (-1,1) (0,1) (-1,0) (0,0) (-1,-1) (0,-1);  v: (1,-1) (1,0) (0,-1) (0,0) (-1,-1) (-1,0)
    const int ou[6] = {P - 1, P, -1, 0, -P - 1, -P}, ov[6] = {-P + 1, 1, -P, 0, -P - 1, -1};
    real wu[6], wv[6], ku[6], kv[6];
    {
        const real wn = hi ? 0.0 : (lo ? 0.25 : 0.125), ws = lo ? 0.0 : (hi ? 0.25 : 0.125);
        const real w6u[6] = {wn, wn, 0.25, 0.25, ws, ws}, w6v[6] = {0.125, 0.125, 0.25, 0.25, 0.125, 0.125};
#pragma unroll
        for (int n = 0; n < 6; n++) {
            wu[n] = w6u[n] * RG; wv[n] = w6v[n] * RG;
            ku[n] = w6u[n] * (p.COLP[c + ou[n]] * p.A[c + ou[n]]);
            kv[n] = w6v[n] * (p.COLP[c + ov[n]] * p.A[c + ov[n]]);
        }
    }
    // six-point means of a field at the u- / v-point with the given weights (a polar row gives its outside points
    // weight 0; their halo values are finite)
    auto m6u = [&](const real *F, size_t q, const real *w) {
        return w[0] * F[q + ou[0]] + w[1] * F[q + ou[1]] + w[2] * F[q + ou[2]] + w[3] * F[q + ou[3]] + w[4] * F[q + ou[4]] + w[5] * F[q + ou[5]];
    };
    auto m6v = [&](const real *F, size_t q, const real *w) {
        return w[0] * F[q + ov[0]] + w[1] * F[q + ov[1]] + w[2] * F[q + ov[2]] + w[3] * F[q + ov[3]] + w[4] * F[q + ov[4]] + w[5] * F[q + ov[5]];
    };
    // RHOVB * COLP * A * KMOM means (dyn_functions.py:276-377)
    auto k6u = [&](size_t q) {
        real s = 0.0;
#pragma unroll
        for (int n = 0; n < 6; n++) s += ku[n] * (p.RHOVB[q + ou[n]] * p.KMOM[q + ou[n]]);
        return s;
    };
    auto k6v = [&](size_t q) {
        real s = 0.0;
#pragma unroll
        for (int n = 0; n < 6; n++) s += kv[n] * (p.RHOVB[q + ov[n]] * p.KMOM[q + ov[n]]);
        return s;
    };
    const real smom_is = m6u(p.SMOMXFLX, c, wu), smom_js = m6v(p.SMOMYFLX, c, wv);
    const real surf_t = p.SSHFLX[c] / CON_CP, surf_q = g.i_moist ? p.SLHFLX[c] / CON_LH : 0.0;

    real kdu_k = 0.0, kdv_k = 0.0;                         // K dU/dz, K dV/dz at interface k (0 at the top)
    real u_k = p.UWIND[c], v_k = p.VWIND[c];
    real altu_k = m6u(p.PHI, c, wu), altv_k = m6v(p.PHI, c, wv);
    real avu_k = m6u(p.PHIVB, c, wu) * RG, avv_k = m6v(p.PHIVB, c, wv) * RG;   // ALTVB at the u- / v-point (g twice: dyn_UFLX.py:141)
    real pott_km1 = 0.0, qv_km1 = 0.0, qc_km1 = 0.0, alt_km1 = 0.0;
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * pl, q1 = q + pl;
        const bool last = (k == nz - 1);
        // ---- momentum ----
        real kdu_kp1 = 0.0, kdv_kp1 = 0.0, u_kp1 = 0.0, v_kp1 = 0.0, altu_kp1 = 0.0, altv_kp1 = 0.0;
        if (!last) {
            u_kp1 = p.UWIND[q1]; v_kp1 = p.VWIND[q1];
            altu_kp1 = m6u(p.PHI, q1, wu); altv_kp1 = m6v(p.PHI, q1, wv);
            kdu_kp1 = k6u(q1) * ((u_k - u_kp1) / (altu_k - altu_kp1));
            kdv_kp1 = k6v(q1) * ((v_k - v_kp1) / (altv_k - altv_kp1));
        }
        const real avu_kp1 = m6u(p.PHIVB, q1, wu) * RG, avv_kp1 = m6v(p.PHIVB, q1, wv) * RG;
        const real rho_is = m6u(p.RHO, q, wu), rho_js = m6v(p.RHO, q, wv);
        const real du = (k == 0 ? 0.0 : kdu_k) - (last ? -smom_is : kdu_kp1);
        const real dv = (k == 0 ? 0.0 : kdv_k) - (last ? -smom_js : kdv_kp1);
        p.TURB_U[q] = du / ((avu_k - avu_kp1) * rho_is);
        if (v_on) p.TURB_V[q] = dv / ((avv_k - avv_kp1) * rho_js);
        kdu_k = kdu_kp1; kdv_k = kdv_kp1; u_k = u_kp1; v_k = v_kp1;
        altu_k = altu_kp1; altv_k = altv_kp1; avu_k = avu_kp1; avv_k = avv_kp1;
        // ---- scalars (dyn_functions.py:26-67): the three quotients that do not depend on the species once ----
        const real alt = p.PHI[q] * RG, alt_kp1 = last ? 0.0 : p.PHI[q1] * RG;
        const real thick = (p.PHIVB[q] * RG - p.PHIVB[q1] * RG) * p.RHO[q];
        const real f_up = k == 0 ? 0.0 : p.RHOVB[q] * p.KHEAT[q] / (alt_km1 - alt);          // flux through interface k per unit difference
        const real f_dn = last ? 0.0 : p.RHOVB[q1] * p.KHEAT[q1] / (alt - alt_kp1);          // ... through interface k+1
        const real r_thick = colp / thick;
        const real pt = p.POTT[q], pt_kp1 = last ? 0.0 : p.POTT[q1];
        p.TURB_T[q] = r_thick * ((pott_km1 - pt) * f_up - (pt - pt_kp1) * f_dn + (last ? surf_t : 0.0));
        pott_km1 = pt;
        if (g.i_moist) {
            const real qv = p.QV[q], qv_kp1 = last ? 0.0 : p.QV[q1], qc = p.QC[q], qc_kp1 = last ? 0.0 : p.QC[q1];
            p.TURB_QV[q] = r_thick * ((qv_km1 - qv) * f_up - (qv - qv_kp1) * f_dn + (last ? surf_q : 0.0));
            p.TURB_QC[q] = r_thick * ((qc_km1 - qc) * f_up - (qc - qc_kp1) * f_dn);
            qv_km1 = qv; qc_km1 = qc;
        }
        alt_km1 = alt;
    }
}

// ------------------------------------------------------------------------------------
// k_continuity_reg: dyn_continuity.py:170-232 with the running sums of the flux divergence kept
// in registers (NZ is a compile-time constant, both level loops fully unrolled), so FLXDIV
// makes no round trip through HBM: 2 reads + 3 writes per cell, the compulsory minimum of
// this stage.  FLXDIV itself is not stored (k_derive_flxdiv re-derives it on demand).
// STORE_FLX = false: UFLX and VFLX are not written either -- every consumer of the fused step
// re-forms them from the winds it reads anyway and the 2-D factors UCOL, VCOL written here
// (UFLX = UCOL * UWIND * dyis, VFLX = VCOL * VWIND * dxjs, the same operations in the same order),
// which leaves WWIND as the only 3-D output: 2 reads + 1 write per cell.
template <int NZ, bool STORE_FLX>
__global__ __launch_bounds__(TI *TJ) void k_continuity_reg(Geo g, Ptrs p, real dt, int lin)
{
    int i, j;
    if (!owned_cell(g, lin, i, j)) return;
    const size_t c = at2(g, i, j);
    const real colp = p.COLP[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1];
    const real c_jm1 = p.COLP[c - g.P], c_jp1 = p.COLP[c + g.P];
    const real dy_i = p.dyis[c], dy_ip1 = p.dyis[c + 1];
    const real dx_j = p.dxjs[c], dx_jp1 = p.dxjs[c + g.P];
    const real area = p.A[c];
    // inputs and outputs never alias: lets the compiler issue the loads of later levels ahead
    // of the stores of earlier ones in the unrolled loop
    const real *__restrict__ uw = p.UWIND;
    const real *__restrict__ vw = p.VWIND;
    real *__restrict__ uflx = p.UFLX;
    real *__restrict__ vflx = p.VFLX;
    real pre[NZ];                     // pre[k] = sum of FLXDIV over the levels above k
    real run = 0.0;
#pragma unroll
    for (int k = 0; k < NZ; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real uflx_i = (c_im1 + colp) / 2.0 * uw[q] * dy_i;
        const real uflx_ip1 = (colp + c_ip1) / 2.0 * uw[q + 1] * dy_ip1;
        const real vflx_j = (c_jm1 + colp) / 2.0 * vw[q] * dx_j;
        const real vflx_jp1 = (colp + c_jp1) / 2.0 * vw[q + g.P] * dx_jp1;
        if (STORE_FLX) {
            uflx[q] = uflx_i;
            vflx[q] = vflx_j;
        }
        const real div = (uflx_ip1 - uflx_i + vflx_jp1 - vflx_j) * p.dsigma[k] / area;
        pre[k] = run;
        run += div;
    }
    const real dcolpdt = -run;
    const real colp_new = p.COLP_OLD[c] + dt * dcolpdt;
    p.dCOLPdt[c] = dcolpdt;
    p.COLP_NEW[c] = colp_new;
    p.UCOL[c] = (c_im1 + colp) / 2.0;
    p.VCOL[c] = (c_jm1 + colp) / 2.0;
    p.WWIND[c] = 0.0;
    p.WWIND[c + (size_t)NZ * g.plane] = 0.0;
#pragma unroll
    for (int k = 1; k < NZ; k++)       // streaming stores (nt): 0.405 -> 0.39 ms per launch (profiles/r02_notes.md)
        __builtin_nontemporal_store((real)(-pre[k] / colp_new - p.sigma_vb[k] * dcolpdt / colp_new), &p.WWIND[c + (size_t)k * g.plane]);
}

// Fields the fused step does not store, re-derived when somebody asks for them:
// UFLX, VFLX (dyn_continuity.py:40-45) from the winds the last continuity call read and UCOL, VCOL;
// FLXDIV (dyn_continuity.py:46-47) from UFLX, VFLX;  PHIVB (dyn_diagnostics.py:160-174) from
// PHI, POTT, PVTF, PVTFVB, HSURF -- whole array including halos, like diag_PHI.
__global__ __launch_bounds__(TI *TJ) void k_derive_uvflx(Geo g, Ptrs p, const real *uw, const real *vw)
{
    DYC_IJ(1, 1)
    if (i > g.nx || j > g.ny) return;
    const size_t c = at2(g, i, j);
    const real ucol = p.UCOL[c], vcol = p.VCOL[c], dy = p.dyis[c], dx = p.dxjs[c];
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        p.UFLX[q] = ucol * uw[q] * dy;
        p.VFLX[q] = vcol * vw[q] * dx;
    }
}
__global__ __launch_bounds__(TI *TJ) void k_derive_flxdiv(Geo g, Ptrs p)
{
    DYC_IJ(1, 1)
    if (i > g.nx || j > g.ny) return;
    const size_t c = at2(g, i, j);
    const real area = p.A[c];
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        p.FLXDIV[q] = (p.UFLX[q + 1] - p.UFLX[q] + p.VFLX[q + g.P] - p.VFLX[q]) * p.dsigma[k] / area;
    }
}
// PVTF (dyn_diagnostics.py:146-153) from PVTFVB and COLP: whole array including halos, the expression of diag_level_pre
// (`canon`: the canonical form the production step evaluates in its threads, kernels_stage.h)
__global__ __launch_bounds__(TI *TJ) void k_derive_pvtf(Geo g, Ptrs p, int canon)
{
#pragma clang fp contract(off)
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    const real colp = p.COLP[c];
    if (canon) {
        const real rcol = canon_rcol(colp);
        double pk = canon_pk(g.pair_top, p.sigma_vb[0], colp);
        real pvb_k = p.PVTFVB[c];
        for (int k = 0; k < g.nz; k++) {
            const size_t q = c + (size_t)k * g.plane;
            const double pk1 = canon_pk(g.pair_top, p.sigma_vb[k + 1], colp);
            const real pvb_kp1 = p.PVTFVB[q + g.plane];
            p.PVTF[q] = canon_pvtf(pvb_kp1, pk1, pvb_k, pk, rcol, p.rdsigma[k]);
            pk = pk1; pvb_k = pvb_kp1;
        }
        return;
    }
    real p_km12 = interface_pressure(g.pair_top, p.sigma_vb[0], colp), pvb_k = p.PVTFVB[c];
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const real p_kp12 = interface_pressure(g.pair_top, p.sigma_vb[k + 1], colp), pvb_kp1 = p.PVTFVB[q + g.plane];
        p.PVTF[q] = 1.0 / (1.0 + CON_KAPPA) * (pvb_kp1 * p_kp12 - pvb_k * p_km12) / (p_kp12 - p_km12);
        p_km12 = p_kp12; pvb_k = pvb_kp1;
    }
}
// POTTVB (dyn_diagnostics.py:179-195) from POTT, PVTFVB and COLP: whole array including halos, for a handle whose step does
// not store it (k_temp_canon / the whole-substep kernel form the interface values they need in the thread, with these calls)
__global__ __launch_bounds__(TI *TJ) void k_derive_pottvb(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    const int nz = g.nz;
    const real colp = p.COLP[c], rcol = canon_rcol(colp);
    double pk1 = canon_pk(g.pair_top, p.sigma_vb[nz], colp);
    real pvb_kp1 = p.PVTFVB[c + (size_t)nz * g.plane];
    real pott_kp1 = 0.0, pvtf_kp1 = 0.0, first = 0.0;
    for (int k = nz - 1; k >= 0; k--) {
        const size_t q = c + (size_t)k * g.plane;
        const double pk = canon_pk(g.pair_top, p.sigma_vb[k], colp);
        const real pvb_k = p.PVTFVB[q], pott = p.POTT[q];
        const real pvtf = canon_pvtf(pvb_kp1, pk1, pvb_k, pk, rcol, p.rdsigma[k]);
        if (k < nz - 1) {
            const real vb = pottvb_between(pvb_kp1, pvtf, pott, pvtf_kp1, pott_kp1);
            p.POTTVB[q + g.plane] = vb;
            if (k == nz - 2 && nz > 2) p.POTTVB[c + (size_t)nz * g.plane] = pott_kp1 - (vb - pott_kp1);     // surface, dyn_diagnostics.py:191-194
            if (k == 0) first = vb;
        }
        pk1 = pk; pvb_kp1 = pvb_k; pott_kp1 = pott; pvtf_kp1 = pvtf;
    }
    if (nz >= 2) p.POTTVB[c] = pott_kp1 - (first - pott_kp1);       // model top, :187-190
}
// PHI and PHIVB (dyn_diagnostics.py:160-174) from POTT, PVTFVB, COLP and HSURF: whole array including halos, the
// expressions of diag_level_pre on the stored PVTFVB -- the bits k_pott_fused would have stored; `canon`: the canonical
// forms, i.e. the values the momentum kernel of the production step carries in its threads
__global__ __launch_bounds__(TI *TJ) void k_derive_phi(Geo g, Ptrs p, int canon)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    const int nz = g.nz;
    const real colp = p.COLP[c];
    real phivb_kp1 = p.HSURF[c] * CON_G;
    p.PHIVB[c + (size_t)nz * g.plane] = phivb_kp1;
    if (canon) {
        const real rcol = canon_rcol(colp);
        double pk1 = canon_pk(g.pair_top, p.sigma_vb[nz], colp);
        real pvb_kp1 = p.PVTFVB[c + (size_t)nz * g.plane];
        for (int k = nz - 1; k >= 0; k--) {
            const size_t q = c + (size_t)k * g.plane;
            const double pk = canon_pk(g.pair_top, p.sigma_vb[k], colp);
            const real pvb_k = p.PVTFVB[q], pott = p.POTT[q];
            const real pvtf = canon_pvtf(pvb_kp1, pk1, pvb_k, pk, rcol, p.rdsigma[k]);
            const real phi = canon_phi(phivb_kp1, pott, pvtf, pvb_kp1);
            phivb_kp1 = canon_phivb(phi, pott, pvb_k, pvtf);
            p.PHI[q] = phi;
            p.PHIVB[q] = phivb_kp1;
            pk1 = pk; pvb_kp1 = pvb_k;
        }
        return;
    }
    real p_kp12 = interface_pressure(g.pair_top, p.sigma_vb[nz], colp), pvb_kp1 = p.PVTFVB[c + (size_t)nz * g.plane];
    for (int k = nz - 1; k >= 0; k--) {
        const size_t q = c + (size_t)k * g.plane;
        const DiagLevel dl = diag_level_pre(interface_pressure(g.pair_top, p.sigma_vb[k], colp), p.PVTFVB[q], p.POTT[q], p_kp12, pvb_kp1, phivb_kp1);
        p.PHI[q] = dl.phi;
        p.PHIVB[q] = dl.phivb;
        p_kp12 = dl.p_km12; pvb_kp1 = dl.pvb_k; phivb_kp1 = dl.phivb;
    }
}
__global__ __launch_bounds__(TI *TJ) void k_derive_phivb(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    p.PHIVB[c + (size_t)g.nz * g.plane] = p.HSURF[c] * CON_G;
    for (int k = 0; k < g.nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        p.PHIVB[q] = p.PHI[q] - CON_CP * (p.POTT[q] * (p.PVTFVB[q] - p.PVTF[q]));
    }
}

// ------------------------------------------------------------------------------------
// k_pott_fused: dPOTTdt (dyn_POTT.py:180-216) + Euler step of POTT (dyn_timestep.py:70-79)
// + primary diagnostics PVTF, PVTFVB, PHI, PHIVB, POTTVB of the NEW state
// (dyn_diagnostics.py:139-195) in one bottom-up sweep per column.  dPOTTdt never reaches
// HBM.  `p.POTT` is the field the tendency is evaluated on (5-point neighbourhood);
// pott_old is the state at step start (own cell only); pott_out gets the new POTT (interior
// cells; halos of every output by k_halo_multi afterwards).  POTTVB is updated in place: interface k+1 is rewritten only after its old
// value has been consumed by the vertical advection of levels k+1 and k.
// Dry dycore only (the turbulence-coupled path uses the stage kernels).
// FROM_WINDS: the mass fluxes are re-formed from UWIND, VWIND and the column pressures
// (k_continuity_reg<., false> does not store UFLX, VFLX): 16 more fp64 instructions per level, which
// cost this issue-bound kernel ~10 % (profiles/r01_notes.md) against 2 passes saved elsewhere.
// Addressing: every field has the same geometry, so ONE 32-bit byte offset per level (plus two for
// the rows j-1, j+1) serves all of them -- `base + zero-extended VGPR offset` is the native form
// of global_load/global_store, where 64-bit pointer arithmetic per access was ~50 VALU
// instructions per level.  (dycore_create refuses fields of 4 GiB and more.)
__device__ __forceinline__ real ldb(const real *base, unsigned boff) { return *(const real *)((const char *)base + boff); }
// (streaming load, `nt`: for the fields a column reads at its own cell only -- WWIND, POTTVB, the step-start POTT; the
// neighbour-shared ones keep the default policy.  k_pott_fused 1.051 -> 1.034 ms, step -1.5 %, profiles/r02_notes.md)
__device__ __forceinline__ real ldn(const real *base, unsigned boff) { return __builtin_nontemporal_load((const real *)((const char *)base + boff)); }
// (streaming store, `nt`: the temperature kernel's five outputs are read again only by later kernels, after ~20 GB of
// other traffic; k_pott_fused 1.17 -> 1.10 ms per launch, profiles/r02_notes.md)
__device__ __forceinline__ void stb(real *base, unsigned boff, real v) { __builtin_nontemporal_store(v, (real *)((char *)base + boff)); }

// WRITE_PVTF = false: PVTF is not stored either -- the production momentum kernel forms it from PVTFVB and COLP, every
// other reader gets it on demand (k_derive_pvtf, the formula of diag_level_pre on the stored PVTFVB: the same bits).
// WRITE_PHI = false: PHI is not stored either -- the production momentum kernel carries it up the column (PHI_LOCAL) from
// the surface geopotential; for its level chunks PHIVB is stored at every PHIVB_STRIDE-th interface (1/4 of a pass
// instead of one).  Every other reader gets PHI (and PHIVB) on demand from k_derive_phi.
// (three workgroups of four wavefronts per CU = three wavefronts per SIMD, <= 168 VGPRs: the kernel waits for its own
// dependent fp64 chain, a third wavefront is worth 7 %; without the bound the register allocator takes 191)
template <bool WRITE_PHIVB, bool FROM_WINDS, bool TURB = false, bool WRITE_PVTF = true, bool WRITE_PHI = true>
__global__ __launch_bounds__(TI *TJ) void k_pott_fused(Geo g, Ptrs p, const real *pott_old, real *pott_out, real dt, int lin)
{
    int i, j;
    if (!owned_cell(g, lin, i, j)) return;
    const int P = g.P, nz = g.nz;
    const size_t c = at2(g, i, j);
    const unsigned bc = (unsigned)(c * sizeof(real)), bpl = (unsigned)(g.plane * sizeof(real)), bP = (unsigned)(P * sizeof(real));
    const real colp = p.COLP[c], cnew = p.COLP_NEW[c], cold = p.COLP_OLD[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1], c_jm1 = p.COLP[c - P], c_jp1 = p.COLP[c + P];
    const real area = p.A[c];
    // as in k_momentum_dma: divisions by per-column / per-level quantities as multiplications by
    // reciprocals formed once (<= 1 ulp per operation); the diagnostics below keep their divisions
    const real hr_area = (real)(0.5 * (1.0 / area)), r_cnew = (real)(1.0 / cnew);
    // surface interface of the new column
    real p_kp12 = interface_pressure(g.pair_top, p.sigma_vb[nz], cnew);
    real pvb_kp1 = exner_vb(p_kp12);
    real phivb_kp1 = p.HSURF[c] * CON_G;
    stb(p.PVTFVB, bc + (unsigned)nz * bpl, pvb_kp1);
    if (WRITE_PHIVB) stb(p.PHIVB, bc + (unsigned)nz * bpl, phivb_kp1);
    real w_kp1 = ldb(p.WWIND, bc + (unsigned)nz * bpl);
    real pvbo_kp1 = ldb(p.POTTVB, bc + (unsigned)nz * bpl);        // old interface values
    real pottn_kp1 = 0.0, pvtf_kp1 = 0.0, pottvb_first = 0.0;
    real p_up = interface_pressure(g.pair_top, p.sigma_vb[nz - 1], cnew), pvb_up = exner_vb(p_up);   // upper interface of the bottom level
    // software pipeline: the inputs of level k-1 are requested before level k is computed, so the
    // HBM latency overlaps the pow/division chain instead of preceding it.  One level ahead is
    // enough (rings of 2-4 levels cost the third wave per SIMD and ran 5-15 % slower,
    // profiles/r01_notes.md); the two input sets alternate roles in a loop unrolled by two, so
    // no register is copied from "next" to "current".
    const real m_dy0 = FROM_WINDS ? p.dyis[c] : 0.0, m_dy1 = FROM_WINDS ? p.dyis[c + 1] : 0.0;
    const real m_dx0 = FROM_WINDS ? p.dxjs[c] : 0.0, m_dx1 = FROM_WINDS ? p.dxjs[c + P] : 0.0;
    const real *fu = FROM_WINDS ? p.UWIND : p.UFLX, *fv = FROM_WINDS ? p.VWIND : p.VFLX;
    struct In { real pt, im1, ip1, jm1, jp1, uf, uf1, vf, vf1, w, pvb, po, tb; };
    auto load = [&](int k, In &r) {
        const unsigned q = bc + (unsigned)k * bpl;
        r.tb = TURB ? ldb(p.TURB_T, q) : 0.0;
        r.pt = ldb(p.POTT, q); r.im1 = ldb(p.POTT, q - (unsigned)sizeof(real)); r.ip1 = ldb(p.POTT, q + (unsigned)sizeof(real));
        r.jm1 = ldb(p.POTT, q - bP); r.jp1 = ldb(p.POTT, q + bP);
        r.uf = ldb(fu, q); r.uf1 = ldb(fu, q + (unsigned)sizeof(real)); r.vf = ldb(fv, q); r.vf1 = ldb(fv, q + bP);
        r.w = ldn(p.WWIND, q); r.pvb = ldn(p.POTTVB, q); r.po = ldn(pott_old, q);
    };
    auto level = [&](int k, In &cur) {
        const unsigned q = bc + (unsigned)k * bpl;
        if (FROM_WINDS) {           // dyn_continuity.py:40-45, operand order of k_continuity_reg
            cur.uf = (c_im1 + colp) / 2.0 * cur.uf * m_dy0;
            cur.uf1 = (colp + c_ip1) / 2.0 * cur.uf1 * m_dy1;
            cur.vf = (c_jm1 + colp) / 2.0 * cur.vf * m_dx0;
            cur.vf1 = (colp + c_jp1) / 2.0 * cur.vf1 * m_dx1;
        }
        real w_k = 0.0, pvbo_k = 0.0;
        if (k > 0) { w_k = cur.w; pvbo_k = cur.pvb; }
        // tendency (turbulent part from k_turb_tendencies, dyn_functions.py:26-67) and Euler forward, pressure weighted
        TEMP_NEW_POTT(pott, colp, cnew, cold, c_im1, c_ip1, c_jm1, c_jp1, hr_area, r_cnew, cur.uf, cur.uf1, cur.vf, cur.vf1, cur.pt, cur.im1, cur.ip1, cur.jm1, cur.jp1,
                      w_k, pvbo_k, w_kp1, pvbo_kp1, p.rdsigma[k], p.POTT_dif_coef[k], TURB, cur.tb, g.i_rad, ldb(p.dPOTTdt_RAD, q), cur.po, dt)
        stb(pott_out, q, pott);
        // diagnostics of the new column, level k
        // the interface pressure of this level's upper interface and its Exner value (one pow) were formed during the
        // level below: the pow depends on the column pressure only, so it need not sit in front of this level's
        // dependent chain (PVTF -> PHI -> PHIVB, POTTVB); the one for the next level up is issued here
        const DiagLevel dl = diag_level_pre(p_up, pvb_up, pott, p_kp12, pvb_kp1, phivb_kp1);
        if (k > 0) {
            p_up = interface_pressure(g.pair_top, p.sigma_vb[k - 1], cnew);
            pvb_up = exner_vb(p_up);
        }
        const real p_km12 = dl.p_km12, pvb_k = dl.pvb_k, pvtf = dl.pvtf, phi = dl.phi, phivb = dl.phivb;
        if (WRITE_PVTF) stb(p.PVTF, q, pvtf);
        stb(p.PVTFVB, q, pvb_k);
        if (WRITE_PHI) stb(p.PHI, q, phi);
        if (WRITE_PHIVB || (!WRITE_PHI && k % PHIVB_STRIDE == 0)) stb(p.PHIVB, q, phivb);
        if (k < nz - 1) {
            const real vb = pottvb_between(pvb_kp1, pvtf, pott, pvtf_kp1, pottn_kp1);
            stb(p.POTTVB, q + bpl, vb);
            if (k == nz - 2 && nz > 2) stb(p.POTTVB, bc + (unsigned)nz * bpl, pottn_kp1 - (vb - pottn_kp1));
            if (k == 0) pottvb_first = vb;
        }
        p_kp12 = p_km12; pvb_kp1 = pvb_k; phivb_kp1 = phivb;
        pottn_kp1 = pott; pvtf_kp1 = pvtf;
        w_kp1 = w_k; pvbo_kp1 = pvbo_k;
    };
    In A, B;
    load(nz - 1, A);
    int k = nz - 1;
#pragma unroll 1
    for (; k >= 1; k -= 2) {
        load(k - 1, B);
        level(k, A);
        if (k >= 2) load(k - 2, A);
        level(k - 1, B);
    }
    if (k == 0) level(0, A);
    if (nz >= 2) stb(p.POTTVB, bc, pottn_kp1 - (pottvb_first - pottn_kp1));
}


// ------------------------------------------------------------------------------------
// k_diag_rho_canon: RHO / RHOVB (dyn_diagnostics.py:199-222) of a turbulence-coupled handle on the fused step, whose temperature
// kernel (k_temp_canon<true>) stores neither PVTF nor POTTVB: both are formed in the thread from POTT, PVTFVB and COLP with the
// canonical forms (the expressions of k_derive_pvtf / k_derive_pottvb: the values a caller would get on demand), one bottom-up sweep,
// whole array including halos like k_diag_rho.  2 reads instead of 4, and no derive launches in front of it.
__global__ __launch_bounds__(TI *TJ) void k_diag_rho_canon(Geo g, Ptrs p)
{
    DYC_IJ(0, 0)
    if (i > g.nx + 1 || j > g.ny + 1) return;
    const size_t c = at2(g, i, j);
    const int nz = g.nz;
    const real colp = p.COLP[c], rcol = canon_rcol(colp);
    double pk1 = canon_pk(g.pair_top, p.sigma_vb[nz], colp);
    real pvb_kp1 = p.PVTFVB[c + (size_t)nz * g.plane];
    const real pvb_surf = pvb_kp1;
    real pott_kp1 = 0.0, pvtf_kp1 = 0.0, first = 0.0;
    auto rho_of = [](real pv, real pt) { return (real)(100000.0 * pow(pv, 1.0 / CON_KAPPA) / (CON_RD * (pt * pv))); };
    for (int k = nz - 1; k >= 0; k--) {
        const size_t q = c + (size_t)k * g.plane;
        const double pk = canon_pk(g.pair_top, p.sigma_vb[k], colp);
        const real pvb_k = p.PVTFVB[q], pott = p.POTT[q];
        const real pvtf = canon_pvtf(pvb_kp1, pk1, pvb_k, pk, rcol, p.rdsigma[k]);
        p.RHO[q] = rho_of(pvtf, pott);
        if (k < nz - 1) {
            const real vb = pottvb_between(pvb_kp1, pvtf, pott, pvtf_kp1, pott_kp1);       // interface k+1
            p.RHOVB[q + g.plane] = rho_of(pvb_kp1, vb);
            if (k == nz - 2 && nz > 2) p.RHOVB[c + (size_t)nz * g.plane] = rho_of(pvb_surf, pott_kp1 - (vb - pott_kp1));     // surface, dyn_diagnostics.py:191-194
            if (k == 0) first = vb;
        }
        pk1 = pk; pvb_kp1 = pvb_k; pott_kp1 = pott; pvtf_kp1 = pvtf;
    }
    if (nz >= 2) p.RHOVB[c] = rho_of(pvb_kp1, pott_kp1 - (first - pott_kp1));       // model top, :187-190
}

// ------------------------------------------------------------------------------------
// k_moist_canon: QV / QC of the fused step (dyn_moist.py:201-243 + the QV / QC part of dyn_timestep.py) -- k_moist<., EULER>'s
// data flow and software pipeline with the arithmetic of a level written as the temperature sweep's is (TEMP_PARTIAL /
// TEMP_CLOSE: horizontal advection and diffusion of level k, then the vertical advection between its two interfaces and the
// pressure-weighted Euler step, every rounding spelt out).  Moisture IS that arithmetic with another interface value -- the
// logarithmic interpolation of dyn_functions.py:70-95 instead of POTTVB -- and written with the same macros the kernel
// produces the same bits wherever the level is evaluated: here (latitude bands, turbulence-coupled handles), or by the
// temperature wavefronts of the whole-substep kernel (k_momentum_dma<..., WITH_TEMP, MOIST>), which walk the column the other
// way round.  Against k_moist<., EULER> the sums are associated differently (the "/ 2" of the face means applied once, the
// diffusion joined to the advection before the vertical term): a few ulp per level.
// FROM_WINDS / TURB_PRE as in k_moist.
template <bool FROM_WINDS, bool TURB_PRE = false>
__global__ __launch_bounds__(TI *TJ) void k_moist_canon(Geo g, Ptrs p, const real *qv_old, const real *qc_old,
                                                        real *qv_out, real *qc_out, real dt, int lin)
{
    int i, j;
    if (!owned_cell(g, lin, i, j)) return;
    const int P = g.P, nz = g.nz;
    const size_t c = at2(g, i, j);
    const real colp = p.COLP[c], cnew = p.COLP_NEW[c], cold = p.COLP_OLD[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1], c_jm1 = p.COLP[c - P], c_jp1 = p.COLP[c + P];
    const real hr_area = (real)(0.5 * (1.0 / p.A[c])), r_cnew = (real)(1.0 / cnew);
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
        r.vo = qv_old[q]; r.co = qc_old[q];
        return r;
    };
    real qv_k = p.QV[c], qc_k = p.QC[c];
    LogCache lv_k = log_cache(qv_k), lc_k = log_cache(qc_k);
    real w_k = 0.0, vbv_k = 0.0, vbc_k = 0.0;       // sigma-dot and the interface values at interface k (0 at the model top)
    In nxt = load(0);
#pragma unroll 1
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * g.plane;
        const In cur = nxt;
        if (k + 1 < nz) nxt = load(k + 1);
        const real dif = p.moist_dif_coef[k], rds = p.rdsigma[k];
        real uflx = cur.uf, uflx_ip1 = cur.uf1, vflx = cur.vf, vflx_jp1 = cur.vf1;
        if (FROM_WINDS) {           // dyn_continuity.py:40-45, operand order of k_continuity_reg (= the flux tiles of k_momentum_dma)
            uflx = (c_im1 + colp) / 2.0 * uflx * s_met[0][st];
            uflx_ip1 = (colp + c_ip1) / 2.0 * uflx_ip1 * s_met[1][st];
            vflx = (c_jm1 + colp) / 2.0 * vflx * s_met[2][st];
            vflx_jp1 = (colp + c_jp1) / 2.0 * vflx_jp1 * s_met[3][st];
        }
        const bool last = (k == nz - 1);
        // the lower interface k+1: sigma-dot times the interface values (0 at the surface: sigma-dot vanishes there)
        real w_kp1 = 0.0, vbv_kp1 = 0.0, vbc_kp1 = 0.0;
        LogCache lv_kp1 = log_cache(cur.qv1), lc_kp1 = log_cache(cur.qc1);
        if (!last) {
            w_kp1 = cur.w1;
            vbv_kp1 = varvb_log_cached(lv_kp1, lv_k);
            vbc_kp1 = varvb_log_cached(lc_kp1, lc_k);
        }
        const real lpv = mul_rn(w_kp1, vbv_kp1), lpc = mul_rn(w_kp1, vbc_kp1);
        TEMP_PARTIAL(tpv, colp, c_im1, c_ip1, c_jm1, c_jp1, hr_area, uflx, uflx_ip1, vflx, vflx_jp1, qv_k, cur.v_im1, cur.v_ip1, cur.v_jm1, cur.v_jp1, dif, false, 0.0)
        TEMP_PARTIAL(tpc, colp, c_im1, c_ip1, c_jm1, c_jp1, hr_area, uflx, uflx_ip1, vflx, vflx_jp1, qc_k, cur.c_im1, cur.c_ip1, cur.c_jm1, cur.c_jp1, dif, false, 0.0)
        if (TURB_PRE) { tpv = tpv + p.TURB_QV[q]; tpc = tpc + p.TURB_QC[q]; }
        TEMP_CLOSE(nv_, tpv, cnew, cold, r_cnew, w_k, vbv_k, lpv, rds, cur.vo, dt)
        TEMP_CLOSE(nc_, tpc, cnew, cold, r_cnew, w_k, vbc_k, lpc, rds, cur.co, dt)
        real nv = nv_, nc = nc_;
        if (g.i_moist_clip) {       // dyn_timestep.py:145-154: `if QV < 0: QV = 0` (a NaN stays a NaN)
            if (nv < 0.0) nv = 0.0;
            if (nc < 0.0) nc = 0.0;
        }
        qv_out[q] = nv;
        qc_out[q] = nc;
        qv_k = cur.qv1; qc_k = cur.qc1;
        lv_k = lv_kp1; lc_k = lc_kp1;
        w_k = w_kp1; vbv_k = vbv_kp1; vbc_k = vbc_kp1;
    }
}

// ------------------------------------------------------------------------------------
// k_temp_canon: the temperature kernel of the production dry / moist step (latitude bands; one GPU with
// DYCORE_STEP_FUSED=0): dPOTTdt (dyn_POTT.py:180-216) + Euler step of POTT (dyn_timestep.py:70-79) + the Exner function
// of the new column (dyn_diagnostics.py:139-156) in one bottom-up sweep.  Against k_pott_fused it moves two fields less:
//   * POTTVB is neither read nor written: the interface values of the state the tendency is taken at are formed in the
//     thread from POTT and PVTFVB of the two adjacent levels (canon_pvtf, pottvb_between) -- PVTFVB is read at the own
//     cell where POTTVB was (and updated in place behind the read), so -1 write;
//   * PHI is not written: the momentum kernel carries the hydrostatic integration in its threads (PHI_LOCAL).  Its level
//     chunks start from the PHIVB planes stored here at every PHIVB_STRIDE-th interface (1/4 of a pass), computed with
//     the same canonical forms, so that a chunk continues the chunk below it bit for bit.
// 5 reads (+1 in the corrector) + 2 1/4 writes.  POTTVB, PHI, PHIVB, PVTF of the new state are derived on demand
// (k_derive_*).  A level is closed one iteration late (TEMP_PARTIAL / TEMP_CLOSE above).
// TURB (turbulence-coupled handles, late in round 3; k_pott_fused<true, ...> before): + the turbulent tendency of
// k_turb_tendencies at the own cell, and PHI / PHIVB of the new column stored at every level / interface -- the inputs of the next
// substep's k_turb_tendencies and of the momentum kernel's staged PHI -- from the hydrostatic integration this sweep carries
// anyway.  PVTF and POTTVB are not stored (derived once per step for the RHO / RHOVB diagnosis): 10 / 11 passes instead of 12 / 13.
template <bool TURB>
__global__ __launch_bounds__(TI *TJ) void k_temp_canon(Geo g, Ptrs p, const real *pott_old, real *pott_out, real dt, int lin)
{
    int i, j;
    if (!owned_cell(g, lin, i, j)) return;
    const int P = g.P, nz = g.nz;
    const size_t c = at2(g, i, j);
    const unsigned bc = (unsigned)(c * sizeof(real)), bpl = (unsigned)(g.plane * sizeof(real)), bP = (unsigned)(P * sizeof(real));
    const real colp = p.COLP[c], cnew = p.COLP_NEW[c], cold = p.COLP_OLD[c];
    const real c_im1 = p.COLP[c - 1], c_ip1 = p.COLP[c + 1], c_jm1 = p.COLP[c - P], c_jp1 = p.COLP[c + P];
    const real hr_area = (real)(0.5 * (1.0 / p.A[c])), r_cnew = (real)(1.0 / cnew);
    const real rcolO = canon_rcol(colp), rcolN = canon_rcol(cnew);
    const real m_dy0 = p.dyis[c], m_dy1 = p.dyis[c + 1], m_dx0 = p.dxjs[c], m_dx1 = p.dxjs[c + P];
    // per-level constants through the constant address space: scalar loads.  As plain loads next to the kernel's stores
    // they compile to VECTOR loads, each followed by s_waitcnt vmcnt(0) -- which drains the level prefetched ahead
    typedef const __attribute__((address_space(4))) real cdouble_t;
    cdouble_t *c_sig = (cdouble_t *)p.sigma_vb, *c_rds = (cdouble_t *)p.rdsigma, *c_dif = (cdouble_t *)p.POTT_dif_coef;
    // the surface interface: new column (written), old column (read first: PVTFVB is updated in place, own cell only)
    double pkN_kp1 = canon_pk(g.pair_top, c_sig[nz], cnew), pkN_kp2 = 0.0;
    real pvbN_kp1 = canon_exner(pkN_kp1), pvbN_kp2 = 0.0;
    real phivbN_kp1 = p.HSURF[c] * CON_G, phivbN_kp2 = 0.0;
    if (TURB) stb(p.PHIVB, bc + (unsigned)nz * bpl, phivbN_kp1);
    real pvbO_kp1 = ldn(p.PVTFVB, bc + (unsigned)nz * bpl);
    double pkO_kp1 = canon_pk(g.pair_top, c_sig[nz], colp);
    stb(p.PVTFVB, bc + (unsigned)nz * bpl, pvbN_kp1);
    real pvtfO_kp1 = 0.0, pt_kp1 = 0.0, w_kp1 = 0.0, lp_kp2 = 0.0, tpart_kp1 = 0.0, po_kp1 = 0.0;
    struct In { real pt, im1, ip1, jm1, jp1, uf, uf1, vf, vf1, w, pvb, po, tb; };
    auto load = [&](int k, In &r) {
        const unsigned q = bc + (unsigned)k * bpl;
        r.tb = TURB ? ldn(p.TURB_T, q) : (real)0.0;
        r.pt = ldb(p.POTT, q); r.im1 = ldb(p.POTT, q - (unsigned)sizeof(real)); r.ip1 = ldb(p.POTT, q + (unsigned)sizeof(real));
        r.jm1 = ldb(p.POTT, q - bP); r.jp1 = ldb(p.POTT, q + bP);
        r.uf = ldb(p.UWIND, q); r.uf1 = ldb(p.UWIND, q + (unsigned)sizeof(real)); r.vf = ldb(p.VWIND, q); r.vf1 = ldb(p.VWIND, q + bP);
        r.w = ldn(p.WWIND, q); r.pvb = ldn(p.PVTFVB, q); r.po = ldn(pott_old, q);
    };
    auto level = [&](int k, In &cur) {
        const unsigned q = bc + (unsigned)k * bpl;
        // the mass fluxes, dyn_continuity.py:40-45, operand order of k_continuity_reg
        cur.uf = (c_im1 + colp) / 2.0 * cur.uf * m_dy0;
        cur.uf1 = (colp + c_ip1) / 2.0 * cur.uf1 * m_dy1;
        cur.vf = (c_jm1 + colp) / 2.0 * cur.vf * m_dx0;
        cur.vf1 = (colp + c_jp1) / 2.0 * cur.vf1 * m_dx1;
        // old column: Exner value of this level, POTTVB at its lower interface k+1 -- which closes level k+1
        const double pkO_k = canon_pk(g.pair_top, c_sig[k], colp);
        const real pvtfO_k = canon_pvtf(pvbO_kp1, pkO_kp1, cur.pvb, pkO_k, rcolO, c_rds[k]);
        real lp_kp1 = 0.0;          // sigma-dot = 0 at the surface
        if (k < nz - 1) {
#if DYC_EXP_TEMP == 2
            const real vb = cur.pt + pvtfO_k;
#else
            const real vb = pottvb_between(pvbO_kp1, pvtfO_k, cur.pt, pvtfO_kp1, pt_kp1);
#endif
            TEMP_CLOSE(pottn, tpart_kp1, cnew, cold, r_cnew, w_kp1, vb, lp_kp2, c_rds[k + 1], po_kp1, dt)
            stb(pott_out, q + bpl, pottn);
            lp_kp1 = mul_rn(w_kp1, vb);
            // hydrostatic integration of the new column through level k+1; PHIVB at interface k+1
#if DYC_EXP_TEMP != 1
            const real pvtfN = canon_pvtf(pvbN_kp2, pkN_kp2, pvbN_kp1, pkN_kp1, rcolN, c_rds[k + 1]);
            const real phiN = canon_phi(phivbN_kp2, pottn, pvtfN, pvbN_kp2);
            phivbN_kp1 = canon_phivb(phiN, pottn, pvbN_kp1, pvtfN);
            if (TURB) stb(p.PHI, q + bpl, phiN);
            if (TURB || (k + 1) % PHIVB_STRIDE == 0) stb(p.PHIVB, q + bpl, phivbN_kp1);
#endif
        }
        TEMP_PARTIAL(tpart, colp, c_im1, c_ip1, c_jm1, c_jp1, hr_area, cur.uf, cur.uf1, cur.vf, cur.vf1, cur.pt, cur.im1, cur.ip1, cur.jm1, cur.jp1,
                     c_dif[k], g.i_rad, ldb(p.dPOTTdt_RAD, q))
        if (TURB) tpart = tpart + cur.tb;
        // Exner function of the new column at interface k
        const double pkN_k = canon_pk(g.pair_top, c_sig[k], cnew);
#if DYC_EXP_TEMP == 3
        const real pvbN_k = (real)pkN_k;
#else
        const real pvbN_k = canon_exner(pkN_k);
#endif
        stb(p.PVTFVB, q, pvbN_k);
        pvbO_kp1 = cur.pvb; pkO_kp1 = pkO_k; pvtfO_kp1 = pvtfO_k; pt_kp1 = cur.pt;
        w_kp1 = k > 0 ? cur.w : (real)0.0;
        lp_kp2 = lp_kp1; tpart_kp1 = tpart; po_kp1 = cur.po;
        pvbN_kp2 = pvbN_kp1; pkN_kp2 = pkN_kp1; pvbN_kp1 = pvbN_k; pkN_kp1 = pkN_k; phivbN_kp2 = phivbN_kp1;
    };
    In A, B;
    load(nz - 1, A);
    int k = nz - 1;
#pragma unroll 1
    for (; k >= 1; k -= 2) {
        load(k - 1, B);
        level(k, A);
        if (k >= 2) load(k - 2, A);
        level(k - 1, B);
    }
    if (k == 0) level(0, A);
    // the top level: its upper interface is the model top (sigma-dot = 0)
    TEMP_CLOSE(pott0, tpart_kp1, cnew, cold, r_cnew, (real)0.0, (real)0.0, lp_kp2, c_rds[0], po_kp1, dt)
    stb(pott_out, bc, pott0);
    if (TURB) {     // PHI of the top level and PHIVB at the model top (after the last level: _kp1 = interface 0, _kp2 = interface 1)
        const real pvtfN = canon_pvtf(pvbN_kp2, pkN_kp2, pvbN_kp1, pkN_kp1, rcolN, c_rds[0]);
        const real phiN = canon_phi(phivbN_kp2, pott0, pvtfN, pvbN_kp2);
        stb(p.PHI, bc, phiN);
        stb(p.PHIVB, bc, canon_phivb(phiN, pott0, pvbN_kp1, pvtfN));
    }
}

}  // namespace dyc
