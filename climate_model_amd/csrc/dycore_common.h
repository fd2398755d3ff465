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

// Working precision (namelist.py:225-226, io_read_namelist.py:47-53: working_precision = 'float64' | 'float32').
// One build of the library per precision: libdycore_hip.so (real = double, the reference's default and the parity mode)
// and libdycore_hip_f32.so (-DDYCORE_FLOAT32: fields, grid constants and the host arrays of the C ABI are float; scalar
// arguments of the ABI stay double).  Every kernel is written on `real`; expressions that contain a double literal are
// evaluated in double and rounded when stored, so the float build is closer to the float64 solution than a pure float32
// evaluation (tests/test_f32.py).  The radiation columns compute in double in either build.
#ifdef DYCORE_FLOAT32
typedef float real;
#else
typedef double real;
#endif
constexpr bool REAL_IS_DOUBLE = sizeof(real) == 8;

// io_constants.py:15-20
constexpr real CON_G = 9.81;
constexpr real CON_RE = 6371000.0;
constexpr real CON_RD = 287.058;
constexpr real CON_CP = 1005.0;
constexpr real CON_KAPPA = CON_RD / CON_CP;
constexpr real CON_LH = 2264E3;       // io_constants.py:24

constexpr int PHIVB_STRIDE = 4;       // PHIVB is kept at every 4th interface when PHI is not stored (the level chunks of the momentum kernel start there)

struct Geo {
    int nx, ny, nz;        // local extents (ny = band height)
    int P;                 // i-pitch in elements (multiple of 64)
    int NJ;                // rows per plane = ny + 3
    long long plane;       // P * NJ
    int ny_global, joff;   // global row = local row + joff
    int pole_s, pole_n;    // this band touches the south / north wall
    int i_turb, i_rad, i_moist;
    int i_moist_clip;      // QV, QC < 0 -> 0 after the Euler step (dyn_timestep.py:145-154, the reference's accelerator path)
    int term_off;          // DYCORE_TERM_* bits: tendency terms switched off (namelist.py:62-88); 0 on every fused path
    real pair_top;
};

// Every array the kernels touch.  Names follow the reference (main_fields.py:234-475,
// main_grid.py:299-314).
struct Ptrs {
    // 2-D prognostic / diagnostic
    real *COLP, *COLP_OLD, *COLP_NEW, *dCOLPdt, *HSURF;
    // 3-D
    real *UWIND, *UWIND_OLD, *VWIND, *VWIND_OLD, *WWIND;
    real *UWIND_EST, *VWIND_EST, *POTT_EST;   // predictor state of the fused step (library scratch)
    // raw turbulent tendencies of the fused turbulence-coupled step (library scratch; only with i_turb)
    real *TURB_U, *TURB_V, *TURB_T, *TURB_QV, *TURB_QC;
    real *UCOL, *VCOL;                        // 2-D: column pressure at the u- / v-point, (COLP[i-1]+COLP)/2, (COLP[j-1]+COLP)/2
    real *POTT, *POTT_OLD, *POTTVB;
    real *UFLX, *VFLX, *FLXDIV;
    real *BFLX, *CFLX, *DFLX, *EFLX, *RFLX, *QFLX, *SFLX, *TFLX;
    real *WWIND_UWIND, *WWIND_VWIND, *KMOM_dUWINDdz, *KMOM_dVWINDdz;
    real *dUFLXdt, *dVFLXdt, *dPOTTdt;
    real *dUFLXdt_TURB, *dVFLXdt_TURB, *dPOTTdt_TURB;
    real *PVTF, *PVTFVB, *PHI, *PHIVB;
    real *RHO, *RHOVB, *KMOM, *KHEAT;
    real *SSHFLX, *SMOMXFLX, *SMOMYFLX, *dPOTTdt_RAD;
    // moisture (dyn_moist.py)
    real *QV, *QV_OLD, *QC, *QC_OLD, *dQVdt, *dQCdt, *dQVdt_TURB, *SLHFLX, *QV_EST, *QC_EST;
    // full secondary diagnostics (dyn_diagnostics.py:199-222)
    real *TAIR, *TAIRVB, *PAIR, *PAIRVB, *WINDX, *WINDY, *WIND;
    // radiation (rad_main.py:33-38); allocated only on handles created with i_radiation
    real *SOILTEMP, *SURFALBEDLW, *SURFALBEDSW, *SOLZEN, *MYSUN, *SWINTOA;
    real *LWFLXDO, *LWFLXUP, *SWDIFFLXDO, *SWDIRFLXDO, *SWFLXUP, *SWFLXDO, *LWFLXNET, *SWFLXNET;
    real *SWFLXDIV, *LWFLXDIV, *TOTFLXDIV;
    // grid constants, 2-D
    const real *A, *dxjs, *dyis, *corf, *corf_is, *lat_rad, *lat_is_rad, *dlon_rad, *dlat_rad;
    // grid constants, vertical 1-D
    const real *sigma_vb, *dsigma, *UVFLX_dif_coef, *POTT_dif_coef, *moist_dif_coef;
    const real *rdsigma;      // 1 / dsigma, formed by the library when dsigma is set
    // per-level constants of the momentum kernel, five arrays of stride nz + 2 (formed by the library from dsigma,
    // sigma_vb, UVFLX_dif_coef): wa[k] = dsigma[k] / (dsigma[k] + dsigma[k-1]), wb[k] = dsigma[k-1] / (...) (0 at k = 0),
    // 1 / dsigma[k], sigma_vb[k] (nz + 1 entries), UVFLX_dif_coef[k]
    const real *mom_c;
    // per-row constants of the momentum kernel, four arrays of stride NJ (formed on the device by k_row_constants whenever
    // lat_rad / lat_is_rad / corf / corf_is are set: a latitude-longitude grid, they depend on the row only):
    // corf_is RE cos(lat_is), sin(lat_is), corf RE cos(lat), sin(lat)
    const real *row_c;
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
