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

struct Geo {
    int nx, ny, nz;        // local extents (ny = band height)
    int P;                 // i-pitch in elements (multiple of 64)
    int NJ;                // rows per plane = ny + 3
    long long plane;       // P * NJ
    int ny_global, joff;   // global row = local row + joff
    int pole_s, pole_n;    // this band touches the south / north wall
    int i_turb, i_rad;
    double pair_top;
};

// Every array the kernels touch.  Names follow the reference (main_fields.py:234-475,
// main_grid.py:299-314).
struct Ptrs {
    // 2-D prognostic / diagnostic
    double *COLP, *COLP_OLD, *COLP_NEW, *dCOLPdt, *HSURF;
    // 3-D
    double *UWIND, *UWIND_OLD, *VWIND, *VWIND_OLD, *WWIND;
    double *POTT, *POTT_OLD, *POTTVB;
    double *UFLX, *VFLX, *FLXDIV;
    double *BFLX, *CFLX, *DFLX, *EFLX, *RFLX, *QFLX, *SFLX, *TFLX;
    double *WWIND_UWIND, *WWIND_VWIND, *KMOM_dUWINDdz, *KMOM_dVWINDdz;
    double *dUFLXdt, *dVFLXdt, *dPOTTdt;
    double *dUFLXdt_TURB, *dVFLXdt_TURB, *dPOTTdt_TURB;
    double *PVTF, *PVTFVB, *PHI, *PHIVB;
    double *RHO, *RHOVB, *KMOM, *KHEAT;
    double *SSHFLX, *SMOMXFLX, *SMOMYFLX, *dPOTTdt_RAD;
    // grid constants, 2-D
    const double *A, *dxjs, *dyis, *corf, *corf_is, *lat_rad, *lat_is_rad, *dlon_rad, *dlat_rad;
    // grid constants, vertical 1-D
    const double *sigma_vb, *dsigma, *UVFLX_dif_coef, *POTT_dif_coef;
};

__device__ __forceinline__ size_t at(const Geo &g, int i, int j, int k)
{
    return (size_t)k * (size_t)g.plane + (size_t)j * (size_t)g.P + (size_t)i;
}
__device__ __forceinline__ size_t at2(const Geo &g, int i, int j)
{
    return (size_t)j * (size_t)g.P + (size_t)i;
}

// ---- stores that also refresh the lateral halo (misc_boundaries.py:22-42 fused into the
// producing kernel).  The thread that owns interior cell (i,j) writes every halo image of
// that cell; halo cells have exactly one owner, so there is no race and no second pass.
// Zonal: periodic (i = nx -> 0, 1 -> nx+1, and 2 -> nx+2 for x-staggered fields).
__device__ __forceinline__ void put_row(double *row, const Geo &g, int i, double v, int stgx)
{
    row[i] = v;
    if (i == g.nx) row[0] = v;
    if (i == 1) row[g.nx + 1] = v;
    if (stgx && i == 2) row[g.nx + 2] = v;
}
// y-unstaggered field: wall rows copy their neighbour (F[:,0]=F[:,1], F[:,ny+1]=F[:,ny]),
// only where the band touches the wall.
__device__ __forceinline__ void put_c(double *F, const Geo &g, int i, int j, int k, double v, int stgx)
{
    double *pl = F + (size_t)k * (size_t)g.plane;
    put_row(pl + (size_t)j * g.P, g, i, v, stgx);
    if (j == 1 && g.pole_s) put_row(pl, g, i, v, stgx);
    if (j == g.ny && g.pole_n) put_row(pl + (size_t)(g.ny + 1) * g.P, g, i, v, stgx);
}
// y-staggered field (x-unstaggered), j in [1, ny+1]: rows 0,1,nys,nys+1 are zero at the walls.
__device__ __forceinline__ void put_s(double *F, const Geo &g, int i, int j, int k, double v)
{
    double *pl = F + (size_t)k * (size_t)g.plane;
    const bool s = (j == 1 && g.pole_s), n = (j == g.ny + 1 && g.pole_n);
    if (s || n) v = 0.0;
    put_row(pl + (size_t)j * g.P, g, i, v, 0);
    if (s) put_row(pl, g, i, 0.0, 0);
    if (n) put_row(pl + (size_t)(g.ny + 2) * g.P, g, i, 0.0, 0);
}

}  // namespace dyc
