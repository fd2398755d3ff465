// dycore_common.h -- geometry, field registry and device-side addressing shared by the
// HIP kernels of the Matsuno dycore step (gfx950 only).
//
// Device layout (DESIGN.md "Data layout in HBM"): every 2-D/3-D array, staggered or not,
// lives in planes of NJ = ny+3 rows x P columns, P = (nx+3) rounded up to 64 elements, so
// that element (i,j,k) of ANY field is at  k*plane + j*P + i .  i is the fastest index:
// a wavefront's 64 lanes read 64 consecutive longitudes (512 B, fully coalesced); the
// k loop runs inside a thread.  Host arrays arrive in the reference's [i][j][k] layout
// (main_fields.py:477-485) and are transposed on the device at upload/download.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dyc {

// io_constants.py:15-20
constexpr double CON_G = 9.81;
constexpr double CON_RE = 6371000.0;
constexpr double CON_RD = 287.058;
constexpr double CON_CP = 1005.0;
constexpr double CON_KAPPA = CON_RD / CON_CP;
constexpr double CON_LH = 2264E3;       // io_constants.py:24

struct Geo {
    int nx, ny, nz;        // local extents (ny = band height)
    int P;                 // i-pitch in elements (multiple of 64)
    int NJ;                // rows per plane = ny + 3
    long long plane;       // P * NJ
    int ny_global, joff;   // global row = local row + joff
    int pole_s, pole_n;    // this band touches the south / north wall
    int i_turb, i_rad, i_moist;
    int i_moist_clip;      // QV, QC < 0 -> 0 after the Euler step (dyn_timestep.py:145-154, the reference's accelerator path)
    double pair_top;
};

// Every array the kernels touch.  Names follow the reference (main_fields.py:234-475,
// main_grid.py:299-314).
struct Ptrs {
    // 2-D prognostic / diagnostic
    double *COLP, *COLP_OLD, *COLP_NEW, *dCOLPdt, *HSURF;
    // 3-D
    double *UWIND, *UWIND_OLD, *VWIND, *VWIND_OLD, *WWIND;
    double *UWIND_EST, *VWIND_EST, *POTT_EST;   // predictor state of the fused step (library scratch)
    // raw turbulent tendencies of the fused turbulence-coupled step (library scratch; only with i_turb)
    double *TURB_U, *TURB_V, *TURB_T, *TURB_QV, *TURB_QC;
    double *UCOL, *VCOL;                        // 2-D: column pressure at the u- / v-point, (COLP[i-1]+COLP)/2, (COLP[j-1]+COLP)/2
    double *POTT, *POTT_OLD, *POTTVB;
    double *UFLX, *VFLX, *FLXDIV;
    double *BFLX, *CFLX, *DFLX, *EFLX, *RFLX, *QFLX, *SFLX, *TFLX;
    double *WWIND_UWIND, *WWIND_VWIND, *KMOM_dUWINDdz, *KMOM_dVWINDdz;
    double *dUFLXdt, *dVFLXdt, *dPOTTdt;
    double *dUFLXdt_TURB, *dVFLXdt_TURB, *dPOTTdt_TURB;
    double *PVTF, *PVTFVB, *PHI, *PHIVB;
    double *RHO, *RHOVB, *KMOM, *KHEAT;
    double *SSHFLX, *SMOMXFLX, *SMOMYFLX, *dPOTTdt_RAD;
    // moisture (dyn_moist.py)
    double *QV, *QV_OLD, *QC, *QC_OLD, *dQVdt, *dQCdt, *dQVdt_TURB, *SLHFLX, *QV_EST, *QC_EST;
    // full secondary diagnostics (dyn_diagnostics.py:199-222)
    double *TAIR, *TAIRVB, *PAIR, *PAIRVB, *WINDX, *WINDY, *WIND;
    // radiation (rad_main.py:33-38); allocated only on handles created with i_radiation
    double *SOILTEMP, *SURFALBEDLW, *SURFALBEDSW, *SOLZEN, *MYSUN, *SWINTOA;
    double *LWFLXDO, *LWFLXUP, *SWDIFFLXDO, *SWDIRFLXDO, *SWFLXUP, *SWFLXDO, *LWFLXNET, *SWFLXNET;
    double *SWFLXDIV, *LWFLXDIV, *TOTFLXDIV;
    // grid constants, 2-D
    const double *A, *dxjs, *dyis, *corf, *corf_is, *lat_rad, *lat_is_rad, *dlon_rad, *dlat_rad;
    // grid constants, vertical 1-D
    const double *sigma_vb, *dsigma, *UVFLX_dif_coef, *POTT_dif_coef, *moist_dif_coef;
    const double *rdsigma;      // 1 / dsigma, formed by the library when dsigma is set
    // per-level constants of the momentum kernel, five arrays of stride nz + 2 (formed by the library from dsigma,
    // sigma_vb, UVFLX_dif_coef): wa[k] = dsigma[k] / (dsigma[k] + dsigma[k-1]), wb[k] = dsigma[k-1] / (...) (0 at k = 0),
    // 1 / dsigma[k], sigma_vb[k] (nz + 1 entries), UVFLX_dif_coef[k]
    const double *mom_c;
};

__device__ __forceinline__ size_t at(const Geo &g, int i, int j, int k)
{
    return (size_t)k * (size_t)g.plane + (size_t)j * (size_t)g.P + (size_t)i;
}
__device__ __forceinline__ size_t at2(const Geo &g, int i, int j)
{
    return (size_t)j * (size_t)g.P + (size_t)i;
}

}  // namespace dyc
