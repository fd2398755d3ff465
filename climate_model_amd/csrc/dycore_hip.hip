// dycore_hip.hip -- host side of libdycore_hip.so: handle, field registry, layout adaptor,
// stage launches, the fused Matsuno step and measurement.  C ABI in include/dycore.h.
// Target: MI355X / gfx950 only.
#include "../../include/dycore.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "dycore_common.h"
#include "kernels_stage.h"
#include "kernels_fused.h"
#include "kernels_radiation.h"

using namespace dyc;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(DYCORE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

struct FieldDesc {
    const char *name;
    int stgx, stgy;
    int nk;                         // 0 -> nz, -1 -> nz+1, 1 -> 1
    real *Ptrs::*member;
    int rad;                        // 1: exists only on handles created with i_radiation; 2: only with i_turb
};

// main_fields.py:234-475
#define F_(n, sx, sy, nk) {#n, sx, sy, nk, &Ptrs::n, 0}
#define R_(n, nk) {#n, 0, 0, nk, &Ptrs::n, 1}
#define T_(n, sx, sy) {#n, sx, sy, 0, &Ptrs::n, 2}
const FieldDesc kFields[] = {
    F_(COLP, 0, 0, 1), F_(COLP_OLD, 0, 0, 1), F_(COLP_NEW, 0, 0, 1), F_(dCOLPdt, 0, 0, 1), F_(HSURF, 0, 0, 1),
    F_(UWIND, 1, 0, 0), F_(UWIND_OLD, 1, 0, 0), F_(VWIND, 0, 1, 0), F_(VWIND_OLD, 0, 1, 0), F_(WWIND, 0, 0, -1),
    F_(POTT, 0, 0, 0), F_(POTT_OLD, 0, 0, 0), F_(POTTVB, 0, 0, -1),
    F_(UWIND_EST, 1, 0, 0), F_(VWIND_EST, 0, 1, 0), F_(POTT_EST, 0, 0, 0), F_(UCOL, 1, 0, 1), F_(VCOL, 0, 1, 1),
    F_(UFLX, 1, 0, 0), F_(VFLX, 0, 1, 0), F_(FLXDIV, 0, 0, 0),
    F_(BFLX, 0, 0, 0), F_(CFLX, 1, 1, 0), F_(DFLX, 0, 1, 0), F_(EFLX, 0, 1, 0),
    F_(RFLX, 0, 0, 0), F_(QFLX, 1, 1, 0), F_(SFLX, 1, 0, 0), F_(TFLX, 1, 0, 0),
    F_(WWIND_UWIND, 1, 0, -1), F_(WWIND_VWIND, 0, 1, -1), F_(KMOM_dUWINDdz, 1, 0, -1), F_(KMOM_dVWINDdz, 0, 1, -1),
    F_(dUFLXdt, 1, 0, 0), F_(dVFLXdt, 0, 1, 0), F_(dPOTTdt, 0, 0, 0),
    F_(dUFLXdt_TURB, 1, 0, 0), F_(dVFLXdt_TURB, 0, 1, 0), F_(dPOTTdt_TURB, 0, 0, 0),
    F_(PVTF, 0, 0, 0), F_(PVTFVB, 0, 0, -1), F_(PHI, 0, 0, 0), F_(PHIVB, 0, 0, -1),
    F_(RHO, 0, 0, 0), F_(RHOVB, 0, 0, -1), F_(KMOM, 0, 0, -1), F_(KHEAT, 0, 0, -1),
    F_(SSHFLX, 0, 0, 1), F_(SMOMXFLX, 0, 0, 1), F_(SMOMYFLX, 0, 0, 1), F_(dPOTTdt_RAD, 0, 0, 0),
    F_(QV, 0, 0, 0), F_(QV_OLD, 0, 0, 0), F_(QC, 0, 0, 0), F_(QC_OLD, 0, 0, 0), F_(dQVdt, 0, 0, 0), F_(dQCdt, 0, 0, 0),
    F_(dQVdt_TURB, 0, 0, 0), F_(SLHFLX, 0, 0, 1), F_(QV_EST, 0, 0, 0), F_(QC_EST, 0, 0, 0),
    F_(TAIR, 0, 0, 0), F_(TAIRVB, 0, 0, -1), F_(PAIR, 0, 0, 0), F_(PAIRVB, 0, 0, -1),
    F_(WINDX, 0, 0, 0), F_(WINDY, 0, 0, 0), F_(WIND, 0, 0, 0),
    // radiation: rad_main.py:33-38 (SOILTEMP: the top soil level, the only one the scheme reads)
    R_(SOILTEMP, 1), R_(SURFALBEDLW, 1), R_(SURFALBEDSW, 1), R_(SOLZEN, 1), R_(MYSUN, 1), R_(SWINTOA, 1),
    R_(LWFLXDO, -1), R_(LWFLXUP, -1), R_(SWDIFFLXDO, -1), R_(SWDIRFLXDO, -1), R_(SWFLXUP, -1), R_(SWFLXDO, -1),
    R_(LWFLXNET, -1), R_(SWFLXNET, -1), R_(SWFLXDIV, 0), R_(LWFLXDIV, 0), R_(TOTFLXDIV, 0),
    // raw turbulent tendencies of the fused turbulence-coupled step
    T_(TURB_U, 1, 0), T_(TURB_V, 0, 1), T_(TURB_T, 0, 0), T_(TURB_QV, 0, 0), T_(TURB_QC, 0, 0),
};
#undef F_
#undef R_
#undef T_
constexpr int kNFields = sizeof(kFields) / sizeof(kFields[0]);

struct GridDesc {
    const char *name;
    int stgx, stgy;                 // 2-D: staggering; 1-D: stgx = -1
    int nk;                         // 1-D: 0 -> nz, -1 -> nz+1
    const real *Ptrs::*member;
};
#define G_(n, sx, sy, nk) {#n, sx, sy, nk, &Ptrs::n}
const GridDesc kGrid[] = {
    G_(A, 0, 0, 1), G_(dxjs, 0, 1, 1), G_(dyis, 1, 0, 1), G_(corf, 0, 0, 1), G_(corf_is, 1, 0, 1),
    G_(lat_rad, 0, 0, 1), G_(lat_is_rad, 1, 0, 1), G_(dlon_rad, 0, 1, 1), G_(dlat_rad, 1, 0, 1),
    G_(sigma_vb, -1, 0, -1), G_(dsigma, -1, 0, 0), G_(UVFLX_dif_coef, -1, 0, 0), G_(POTT_dif_coef, -1, 0, 0),
    G_(moist_dif_coef, -1, 0, 0),       // must stay last: optional unless i_moist
};
#undef G_
constexpr int kNGrid = sizeof(kGrid) / sizeof(kGrid[0]);

// kernels tracked by the profiler
enum KernelId {
    K_CONTINUITY = 0, K_PREPARE, K_UFLX, K_VFLX, K_POTT, K_EULER, K_DIAG, K_RHO, K_EXCHANGE, K_COPY, K_HALO, K_MOM_FUSED, K_POTT_FUSED, K_TEMP_CANON, K_HALO_MULTI, K_CONT_REG, K_DERIVE, K_MOIST, K_SECONDARY, K_TURB, K_RAD, K_TURB_TEND, K_MOM_DMA, K_SUBSTEP,
    K_COUNT
};
const char *kKernelNames[K_COUNT] = {
    "k_continuity", "k_prepare", "k_uflx_tendency", "k_vflx_tendency", "k_pott_tendency",
    "k_euler_forward", "k_primary_diag", "k_diag_rho", "k_exchange_bc", "copy_field", "k_rows", "k_momentum_fused", "k_pott_fused", "k_temp_canon", "k_halo_multi", "k_continuity_reg", "k_derive", "k_moist", "k_diag_secondary", "k_turbulence", "k_radiation", "k_turb_tendencies", "k_momentum_dma", "k_substep_fused",
};

struct ProfRec { int kid; hipEvent_t a, b; };

}  // namespace

struct dycore {
    DycoreConfig cfg;
    Geo g;
    Ptrs p;
    hipStream_t stream = nullptr;   // where launches go (dycore_use_stream can redirect them)
    hipStream_t home = nullptr;     // the handle's own stream (DycoreConfig.stream or created here)
    bool own_stream = false;
    std::unordered_map<std::string, int> field_idx, grid_idx;
    std::vector<real *> allocs;
    double *rad_ws = nullptr;     // workspace of the radiation columns, [element][column] (the column solves run in double in either precision)
    long long rad_ws_cols = 0;
    real *stage = nullptr;        // staging buffer for the layout adaptor
    std::vector<real> h_dsigma, h_sigma_vb, h_uvdif;      // host copies of the vertical grid (-> Ptrs::mom_c)
    size_t stage_elems = 0;
    int grid_set = 0;
    bool zonal = true;              // lat_rad, lat_is_rad, corf, corf_is do not vary along a row (the fused momentum kernel keeps them per wavefront)
    int nonzonal = 0;               // ... which of the four do (bits 1, 2, 4, 8), as of their last dycore_set_grid
    bool fused = true;              // production kernels in dycore_step_matsuno (DYCORE_FUSED=0: stage kernels)
    bool lazy = true;               // fused step leaves FLXDIV / PHIVB to be derived on demand (DYCORE_LAZY=0: store them)
    bool fused_turb = true;         // DYCORE_FUSED_TURB=0: a turbulence-coupled handle steps on the stage kernels
    int fw_mask = 7;                // DYCORE_FW_MASK (debug): 1 continuity stores no fluxes, 2 momentum, 4 temperature/moisture re-form them
    bool lazy_flx = true;           // ... nor UFLX / VFLX: consumers re-form them from the winds (DYCORE_LAZY_FLX=0: store them)
    bool stale_flxdiv = false, stale_phivb = false, stale_uvflx = false, stale_pvtf = false, stale_phi = false, stale_pottvb = false;
    // ... nor PHI and POTTVB (DYCORE_LAZY_PHI=0: store them): the LDS-DMA momentum kernel carries the hydrostatic integration up
    // the column in its threads and the temperature kernel forms the interface values of POTT it needs from POTT and PVTFVB;
    // the temperature kernel stores PHIVB at every 4th interface for the level chunks of the momentum kernel, evaluated with
    // the same canonical forms (kernels_stage.h), so that results do not depend on the work list or the band decomposition
    bool lazy_phi = true;
    int step_fused_bands = -1;      // a latitude band on the whole-substep kernel: -1 when it is large enough (substep_fused), DYCORE_BANDS_STEP_FUSED=1 / 0 always / never
    bool moist_fused = true;        // QV / QC advanced by the whole-substep kernel's temperature wavefronts (DYCORE_MOIST_FUSED=0: k_moist_canon behind it)
    bool row_c_valid = false;       // Ptrs::row_c holds the per-row constants of the current lat_rad / lat_is_rad / corf / corf_is
    bool colp_old_ready = false;    // COLP_OLD already holds COLP (second destination of the copy rider that closed the previous step)
    bool rows_ready = false;        // the PHIVB planes of the rows the next whole-substep launch cuts into chunks are current (rider of the halo launch behind the continuity kernel)
    bool planes_valid = false;      // the PHIVB planes at every PHIVB_STRIDE-th interface hold the canonical integration of the current state (k_temp_canon keeps them)
    bool lazy_pvtf = true;          // ... nor PVTF (DYCORE_LAZY_PVTF=0: store it); only where the momentum kernel forms its own (flux_from_winds, no i_turb)
    const real *flx_u = nullptr, *flx_v = nullptr;   // winds the last fused continuity evaluated on
    bool winds_bc_dirty = true;     // UWIND / VWIND came from the host: their lateral BC images (column nx+2!) may be unset
    int n_cu = 256;                 // compute units of the device
    // momentum + temperature of a substep in ONE launch (k_momentum_dma<..., WITH_TEMP>; DYCORE_STEP_FUSED=0: two kernels):
    // the whole grid on one GPU, dry or moist, with or without radiative heating, no turbulence coupling
#ifdef DYCORE_FLOAT32
    // (float build: off -- there the compiler contracts the momentum arithmetic of the two instantiations differently, a few
    // cells per 1e4 differ by rounding between a band on the separate kernels and the single-GPU run, tests/test_f32.py)
    bool step_fused = false;
#else
    bool step_fused = true;
#endif
    real *pvb_alt = nullptr;
    long long *trace_buf = nullptr;                    // (-DDYC_TRACE experiment builds)       // the buffers that launch writes the new PVTFVB / PHI into (swapped in afterwards)
    int mom_tj = 4;                 // tile rows of the LDS-DMA momentum kernel (8: one workgroup per CU; 4: two, each with its own barrier)
    int cell_lin = -1;              // DYCORE_CELL_LINEAR: column-kernel launch, bit 0 = cell runs, bit 1 = XCD-aware order (default 3 or 2)
    int mom_ks = 0;                 // DYCORE_MOM_KS: force the level chunks per momentum tile (1 .. 16)
    int mom_split = 1;              // DYCORE_MOM_SPLIT=0: no second segment (short chunks for the tiles beyond the last full round)
    double mom_pro = 2.0;           // DYCORE_MOM_PRO: prologue of a level chunk in units of one level (plan cost model)
    int mom_dma = 1;                // momentum kernel: 1 = LDS-DMA (default); 0 = register-staged tiles (DYCORE_MOM_DMA)
    int mom_patch = 4;              // tile order: bands of this many tile rows traversed column by column (DYCORE_MOM_PATCH; 1 = row by row)
    int mom_depth = 2;              // levels the LDS-DMA kernel keeps in flight (DYCORE_MOM_DEPTH=1: one, the round-1 schedule)
    // profiler
    int prof = 0;                   // 0 off, 1 every launch, 2 dominant kernel only
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> ev_pool;
    int64_t launches[K_COUNT] = {0};
    double total_ms[K_COUNT] = {0};

    int nk_of(int code) const { return code == 0 ? g.nz : (code == -1 ? g.nz + 1 : 1); }
    size_t field_elems(int code) const { return (size_t)g.plane * nk_of(code); }
};

namespace {

int get_event(dycore *h, hipEvent_t *e)
{
    if (!h->ev_pool.empty()) { *e = h->ev_pool.back(); h->ev_pool.pop_back(); return 0; }
    HIP_TRY(hipEventCreate(e));
    return 0;
}

int prof_flush(dycore *h)
{
    if (h->recs.empty()) return 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto &r : h->recs) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        h->total_ms[r.kid] += ms;
        h->launches[r.kid] += 1;
        h->ev_pool.push_back(r.a);
        h->ev_pool.push_back(r.b);
    }
    h->recs.clear();
    return 0;
}

struct Scope {          // brackets one launch with events when profiling is on
    dycore *h; int kid; ProfRec r; bool on;
    Scope(dycore *h_, int kid_) : h(h_), kid(kid_), on(false)
    {
        if (!h->prof || (h->prof == 2 && kid != K_MOM_FUSED && kid != K_MOM_DMA && kid != K_SUBSTEP)) return;
        if (get_event(h, &r.a) || get_event(h, &r.b)) return;
        r.kid = kid;
        on = hipEventRecord(r.a, h->stream) == hipSuccess;
    }
    ~Scope()
    {
        if (!on) return;
        (void)hipEventRecord(r.b, h->stream);
        h->recs.push_back(r);
        if (h->recs.size() >= 4096) prof_flush(h);
    }
};

dim3 grid_for(const Geo &g, int ni, int nj) { return dim3((ni + TI - 1) / TI, (nj + TJ - 1) / TJ, 1); }
// launch geometry of the column kernels over the owned cells: see owned_cell (kernels_stage.h)
int cells_linear(const dycore *h) { return h->cell_lin >= 0 ? h->cell_lin : ((h->g.nx % 32 == 0) ? 3 : 2); }
dim3 grid_cells(const dycore *h)
{
    long long n;
    if (cells_linear(h) & 1) n = ((long long)h->g.nx * h->g.ny + TI * TJ - 1) / (TI * TJ);
    else n = (long long)((h->g.nx + TI - 1) / TI) * ((h->g.ny + TJ - 1) / TJ);
    return dim3((unsigned)(8 * ((n + 7) / 8)));
}

int check_ready(dycore *h)
{
    if (!h) return fail(DYCORE_EINVAL, "null handle");
    int need = (1 << kNGrid) - 1;
    if (!h->g.i_moist) need &= ~(1 << (kNGrid - 1));      // moist_dif_coef only with moisture
    if ((h->grid_set & need) != need)
        return fail(DYCORE_ESTATE, "grid constants missing: call dycore_set_grid for all %d arrays first", kNGrid);
    return 0;
}

enum { LAZY_FLX = 1, LAZY_FLXDIV = 2, LAZY_PHIVB = 4, LAZY_PVTF = 8, LAZY_PHI = 16, LAZY_POTTVB = 32, LAZY_ALL = 63 };
int refresh_lazy(dycore *h, int need = LAZY_ALL);
// stage entry points read what the side stream may still be writing and what the fused step
// left to be derived (`need`: which of the derived fields the stage reads)
int ready_join(dycore *h, int need = LAZY_ALL)
{
    int rc = check_ready(h);
    if (rc) return rc;
    h->planes_valid = false; h->rows_ready = false; h->colp_old_ready = false;        // a stage call may rewrite what the stored PHIVB planes were integrated from
    return refresh_lazy(h, need);
}

int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DYCORE_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// lateral halo (misc_boundaries.py:22-42) of up to 12 fields in one launch
struct HaloSpec { real *ptr; int stgx, stgy, nk; };
int do_halo_noscope(dycore *h, const std::vector<HaloSpec> &fields, real *copy_dst = nullptr, const real *copy_src = nullptr, size_t copy_n = 0, const PlaneRows *rows = nullptr, real *copy_dst2 = nullptr);
int do_halo(dycore *h, const std::vector<HaloSpec> &fields, real *copy_dst = nullptr, const real *copy_src = nullptr, size_t copy_n = 0, const PlaneRows *rows = nullptr, real *copy_dst2 = nullptr)
{
    Scope s(h, K_HALO_MULTI);
    return do_halo_noscope(h, fields, copy_dst, copy_src, copy_n, rows, copy_dst2);
}
int do_halo_noscope(dycore *h, const std::vector<HaloSpec> &fields, real *copy_dst, const real *copy_src, size_t copy_n, const PlaneRows *rows, real *copy_dst2)
{
    HaloList l;
    l.copy_dst = copy_dst; l.copy_src = copy_src; l.copy_n = (long long)copy_n; l.copy_dst2 = copy_dst ? copy_dst2 : nullptr;
    if (rows) l.rows = *rows;
    else { memset(&l.rows, 0, sizeof l.rows); l.rows.j_lo = 1; l.rows.j_hi = 0; }
    const bool with_rows = l.rows.j_hi >= l.rows.j_lo;
    l.n = 0;
    int maxk = 1, maxcells = 0;
    for (const HaloSpec &f : fields) {
        const int e = l.n++;
        l.base[e] = f.ptr; l.stgx[e] = f.stgx; l.stgy[e] = f.stgy; l.nk[e] = f.nk;
        if (f.nk > maxk) maxk = f.nk;
        const int cells = 4 * (h->g.nx + 3) + 3 * (h->g.ny + 1);
        if (cells > maxcells) maxcells = cells;
    }
    if (!l.n) return 0;
    int gx = (maxcells + 255) / 256;
    if (with_rows) {        // the rider's slice covers (nx + 2) x rows columns
        const long long need = (long long)(h->g.nx + 2) * (l.rows.j_hi - l.rows.j_lo + 1);
        gx = std::max(gx, (int)((need + 256LL * maxk - 1) / (256LL * maxk)));
    }
    hipLaunchKernelGGL(k_halo_multi, dim3(gx, maxk, l.n + (copy_dst ? 1 : 0) + (with_rows ? 1 : 0)), dim3(256), 0, h->stream, h->g, l);
    return launch_check("k_halo_multi");
}

// ---- stage launches -------------------------------------------------------------------
int do_continuity(dycore *h, const Ptrs &p, real dt)
{
    {
        Scope s(h, K_CONTINUITY);
        hipLaunchKernelGGL(k_continuity, grid_for(h->g, h->g.nx, h->g.ny), dim3(TI, TJ), 0, h->stream, h->g, p, dt);
        int rc = launch_check("k_continuity");
        if (rc) return rc;
    }
    const int nz = h->g.nz;
    return do_halo(h, {{p.UFLX, 1, 0, nz}, {p.VFLX, 0, 1, nz}, {p.WWIND, 0, 0, nz + 1}, {p.COLP_NEW, 0, 0, 1}});
}
int do_prepare(dycore *h, const Ptrs &p)
{
    {
        Scope s(h, K_PREPARE);
        dim3 gr = grid_for(h->g, h->g.nx + 1, h->g.ny + 2);      // rows 0..ny+1
        if (h->g.i_turb) hipLaunchKernelGGL(k_prepare<true>, gr, dim3(TI, TJ), 0, h->stream, h->g, p);
        else hipLaunchKernelGGL(k_prepare<false>, gr, dim3(TI, TJ), 0, h->stream, h->g, p);
        int rc = launch_check("k_prepare");
        if (rc) return rc;
    }
    if (!h->g.i_turb) return 0;     // dyn_org_discretizations.py:212-213
    return do_halo(h, {{p.KMOM_dUWINDdz, 1, 0, h->g.nz + 1}, {p.KMOM_dVWINDdz, 0, 1, h->g.nz + 1}});
}
int do_uflx(dycore *h, const Ptrs &p)
{
    Scope s(h, K_UFLX);
    dim3 gr = grid_for(h->g, h->g.nx, h->g.ny);
    if (h->g.i_turb) hipLaunchKernelGGL(k_uflx_tendency<true>, gr, dim3(TI, TJ), 0, h->stream, h->g, p);
    else hipLaunchKernelGGL(k_uflx_tendency<false>, gr, dim3(TI, TJ), 0, h->stream, h->g, p);
    return launch_check("k_uflx_tendency");
}
int do_vflx(dycore *h, const Ptrs &p)
{
    Scope s(h, K_VFLX);
    const int jlo = h->g.pole_s ? 2 : 1;
    const int jhi = h->g.ny;                      // js = ny+1 is a pole or the neighbour's row
    if (jhi < jlo) return 0;
    dim3 gr = grid_for(h->g, h->g.nx, jhi - jlo + 1);
    if (h->g.i_turb) hipLaunchKernelGGL(k_vflx_tendency<true>, gr, dim3(TI, TJ), 0, h->stream, h->g, p, jlo, jhi);
    else hipLaunchKernelGGL(k_vflx_tendency<false>, gr, dim3(TI, TJ), 0, h->stream, h->g, p, jlo, jhi);
    return launch_check("k_vflx_tendency");
}
int do_pott(dycore *h, const Ptrs &p)
{
    Scope s(h, K_POTT);
    dim3 gr = grid_for(h->g, h->g.nx, h->g.ny);
    if (h->g.i_turb) hipLaunchKernelGGL(k_pott_tendency<true>, gr, dim3(TI, TJ), 0, h->stream, h->g, p);
    else hipLaunchKernelGGL(k_pott_tendency<false>, gr, dim3(TI, TJ), 0, h->stream, h->g, p);
    return launch_check("k_pott_tendency");
}
int do_euler(dycore *h, const Ptrs &p, real dt)
{
    {
        Scope s(h, K_EULER);
        hipLaunchKernelGGL(k_euler_forward, grid_for(h->g, h->g.nx, h->g.ny + 1), dim3(TI, TJ), 0, h->stream, h->g, p, dt);
        int rc = launch_check("k_euler_forward");
        if (rc) return rc;
    }
    const int nz = h->g.nz;
    if (h->g.i_moist)
        return do_halo(h, {{p.POTT, 0, 0, nz}, {p.VWIND, 0, 1, nz}, {p.UWIND, 1, 0, nz}, {p.QV, 0, 0, nz}, {p.QC, 0, 0, nz}});
    return do_halo(h, {{p.POTT, 0, 0, nz}, {p.VWIND, 0, 1, nz}, {p.UWIND, 1, 0, nz}});
}
int do_diag(dycore *h, const Ptrs &p)
{
    Scope s(h, K_DIAG);
    hipLaunchKernelGGL(k_primary_diag, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, p);
    if (&p == &h->p || p.PVTF == h->p.PVTF) { h->stale_pvtf = false; h->stale_phivb = false; h->stale_phi = false; h->stale_pottvb = false; }      // it writes all of them, whole array
    return launch_check("k_primary_diag");
}
int do_rho(dycore *h, const Ptrs &p)
{
    Scope s(h, K_RHO);
    hipLaunchKernelGGL(k_diag_rho, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, p);
    return launch_check("k_diag_rho");
}
int do_secondary(dycore *h, const Ptrs &p)
{
    Scope s(h, K_SECONDARY);
    hipLaunchKernelGGL(k_diag_secondary, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, p);
    return launch_check("k_diag_secondary");
}
int do_turbulence(dycore *h, const Ptrs &p)
{
    Scope s(h, K_TURB);
    hipLaunchKernelGGL(k_turbulence, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, p);
    return launch_check("k_turbulence");
}
int do_exchange(dycore *h, real *F, int stgx, int stgy, int nk)
{
    Scope s(h, K_EXCHANGE);
    const int fnx = h->g.nx + 2 + stgx, fny = h->g.ny + 2 + stgy;
    hipLaunchKernelGGL(k_exchange_bc, dim3((fnx + 255) / 256, fny, nk), dim3(256), 0, h->stream, h->g, F, stgx, stgy, nk);
    return launch_check("k_exchange_bc");
}
int do_copy(dycore *h, real *dst, const real *src, size_t elems)
{
    Scope s(h, K_COPY);
    HIP_TRY(hipMemcpyAsync(dst, src, elems * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

bool fused_path(const dycore *h);
// continuity of the fused step: running sums in registers, FLXDIV (and UFLX, VFLX) not stored
bool flux_from_winds(const dycore *h)
{
    const int nz = h->g.nz;
    return h->lazy && h->lazy_flx && h->mom_dma == 1 && nz <= DMAXZ && (nz == 8 || nz == 16 || nz == 32 || nz == 64);
}
// the fused step leaves PVTF to be derived on demand: only where every kernel of that step that needs it forms its own
// (LDS-DMA momentum kernel on fluxes re-formed from the winds, no turbulence coupling, every re-forming switch on)
bool pvtf_on_demand(const dycore *h)
{
    return h->lazy && h->lazy_pvtf && !h->g.i_turb && flux_from_winds(h) && h->fw_mask == 7;
}
// ... and PHI: the same momentum kernel carries it up the column from the surface geopotential (its level chunks start
// from the PHIVB planes k_pott_fused stores at every PHIVB_STRIDE-th interface)
bool phi_on_demand(const dycore *h) { return pvtf_on_demand(h) && h->lazy_phi && h->mom_tj == 4; }      // (the 64 x 8 tiles have no registers left for it)
int do_continuity_fused(dycore *h, const Ptrs &p, real dt, const PlaneRows *rows = nullptr)
{
    const int nz = h->g.nz;
    if (!h->lazy || !(nz == 8 || nz == 16 || nz == 32 || nz == 64)) return do_continuity(h, p, dt);
    const bool fw = flux_from_winds(h) && (h->fw_mask & 1);
    if (flux_from_winds(h) && h->winds_bc_dirty) {
        // Re-forming UFLX(nx+2), VFLX(ny+2) reads the winds' own boundary images.  The reference's host-side
        // initialisation leaves column nx+2 of x-staggered arrays unset (main_grid.py:317-357) and reaches it
        // only through the exchange of UFLX; apply the winds' lateral BC (misc_boundaries.py:22-42) once.
        int rc = do_halo(h, {{const_cast<real *>(p.UWIND), 1, 0, nz}, {const_cast<real *>(p.VWIND), 0, 1, nz}});
        if (rc) return rc;
        h->winds_bc_dirty = false;
    }
    {
        Scope s(h, K_CONT_REG);
        const dim3 gr = grid_cells(h), bl(TI, TJ);
        const int lin = cells_linear(h);
#define CONT_LAUNCH(NZ_)                                                                                  \
        if (fw) hipLaunchKernelGGL((k_continuity_reg<NZ_, false>), gr, bl, 0, h->stream, h->g, p, dt, lin); \
        else hipLaunchKernelGGL((k_continuity_reg<NZ_, true>), gr, bl, 0, h->stream, h->g, p, dt, lin);
        switch (nz) {
        case 8: CONT_LAUNCH(8) break;
        case 16: CONT_LAUNCH(16) break;
        case 32: CONT_LAUNCH(32) break;
        default: CONT_LAUNCH(64) break;
        }
#undef CONT_LAUNCH
        int rc = launch_check("k_continuity_reg");
        if (rc) return rc;
    }
    h->stale_flxdiv = true;
    if (fw) {
        h->stale_uvflx = true;
        h->flx_u = p.UWIND; h->flx_v = p.VWIND;
        int rc = do_halo(h, {{p.UCOL, 1, 0, 1}, {p.VCOL, 0, 1, 1}, {p.WWIND, 0, 0, nz + 1}, {p.COLP_NEW, 0, 0, 1}}, nullptr, nullptr, 0, rows);
        if (!rc && rows) h->rows_ready = true;
        return rc;
    }
    return do_halo(h, {{p.UFLX, 1, 0, nz}, {p.VFLX, 0, 1, nz}, {p.WWIND, 0, 0, nz + 1}, {p.COLP_NEW, 0, 0, 1}});
}
// bring the on-demand fields up to date (called wherever they can be observed)
int refresh_lazy(dycore *h, int need)
{
    if (need & LAZY_FLXDIV) need |= LAZY_FLX;          // FLXDIV is derived from UFLX, VFLX
    if ((need & LAZY_PHIVB) && h->stale_phivb && h->stale_phi) need |= LAZY_PHI;      // k_derive_phi writes PHI and PHIVB
    if ((need & LAZY_PHIVB) && h->stale_phivb && !(need & LAZY_PHI)) need |= LAZY_PVTF;      // PHIVB from PHI, POTT, PVTFVB and PVTF
    const bool d_phi = h->stale_phi && (need & LAZY_PHI);
    const bool d_pottvb = h->stale_pottvb && (need & LAZY_POTTVB);
    const bool d_flx = h->stale_uvflx && (need & LAZY_FLX), d_div = h->stale_flxdiv && (need & LAZY_FLXDIV),
               d_phivb = h->stale_phivb && (need & LAZY_PHIVB) && !d_phi, d_pvtf = h->stale_pvtf && (need & LAZY_PVTF);
    if (!d_flx && !d_div && !d_phivb && !d_pvtf && !d_phi && !d_pottvb) return 0;
    int rc;
    Scope s(h, K_DERIVE);
    if (d_flx) {
        hipLaunchKernelGGL(k_derive_uvflx, grid_for(h->g, h->g.nx, h->g.ny), dim3(TI, TJ), 0, h->stream, h->g, h->p, h->flx_u, h->flx_v);
        if ((rc = launch_check("k_derive_uvflx"))) return rc;
        h->stale_uvflx = false;
        if ((rc = do_halo_noscope(h, {{h->p.UFLX, 1, 0, h->g.nz}, {h->p.VFLX, 0, 1, h->g.nz}}))) return rc;
    }
    if (d_div) {
        hipLaunchKernelGGL(k_derive_flxdiv, grid_for(h->g, h->g.nx, h->g.ny), dim3(TI, TJ), 0, h->stream, h->g, h->p);
        if ((rc = launch_check("k_derive_flxdiv"))) return rc;
        h->stale_flxdiv = false;
    }
    // (a handle whose step keeps PHI in the momentum kernel's threads derives with the canonical forms that step uses)
    const int canon = (phi_on_demand(h) || (h->g.i_turb && fused_path(h))) ? 1 : 0;      // (... and the turbulence-coupled fused step: k_temp_canon<true>)
    if (d_pvtf) {
        hipLaunchKernelGGL(k_derive_pvtf, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, h->p, canon);
        if ((rc = launch_check("k_derive_pvtf"))) return rc;
        h->stale_pvtf = false;
    }
    if (d_pottvb) {
        hipLaunchKernelGGL(k_derive_pottvb, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, h->p);
        if ((rc = launch_check("k_derive_pottvb"))) return rc;
        h->stale_pottvb = false;
    }
    if (d_phi) {
        hipLaunchKernelGGL(k_derive_phi, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, h->p, canon);
        if ((rc = launch_check("k_derive_phi"))) return rc;
        h->stale_phi = false;
        h->stale_phivb = false;
    }
    if (d_phivb) {
        hipLaunchKernelGGL(k_derive_phivb, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, h->p);
        if ((rc = launch_check("k_derive_phivb"))) return rc;
        h->stale_phivb = false;
    }
    return 0;
}

// Which tile rows of the momentum kernel a launch covers.  A latitude band overlaps its
// neighbour exchange with the tiles whose stencils stay clear of the received halo rows
// (INNER) and finishes the edge tiles (EDGE) once the rows have arrived; a band that touches a
// pole has no neighbour on that side.  ALL = INNER + EDGE in one launch.
enum { ROWS_ALL = 0, ROWS_INNER = 1, ROWS_EDGE = 2, ROWS_INNER_A = 3, ROWS_INNER_B = 4 };      // INNER = INNER_A + INNER_B (its first / second half)
// inner range out of [0, n-1] given how many rows at each end depend on a neighbour
bool row_subset(int n, int dep_s, int dep_n, int mode, RowSet &rs)
{
    int lo = dep_s, hi = n - 1 - dep_n;
    if (lo > hi) { lo = n; hi = n - 1; }              // no inner part: everything is edge
    const int mid = lo + (hi - lo + 2) / 2;       // first row of the second half
    if (mode == ROWS_ALL) rs = {0, n, n, 0};
    else if (mode == ROWS_INNER) rs = {lo, hi - lo + 1, n, 0};
    else if (mode == ROWS_INNER_A) rs = {lo, mid - lo, n, 0};
    else if (mode == ROWS_INNER_B) rs = {mid, hi - mid + 1, n, 0};
    else rs = {0, n - (hi - lo + 1), lo, hi - lo + 1};
    return rs.n > 0;
}

// The level chunks of the PHI_LOCAL momentum kernel start from stored PHIVB planes.  k_temp_canon keeps them current from
// substep to substep; after an upload, a stage call or a whole-substep launch nobody has integrated them for the current
// state: once, with the forms the threads use (whole array, halo rows included: a band's halo rows of POTT, PVTFVB, COLP,
// HSURF are in place whenever its step starts).  Called when a step begins, i.e. before the band's first exchange of
// the step packs rows of these planes for its neighbours.
bool phi_on_demand(const dycore *h);
int ensure_planes(dycore *h)
{
    if (h->planes_valid || !phi_on_demand(h)) return 0;
    Scope s(h, K_DERIVE);
    hipLaunchKernelGGL(k_derive_phi, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, h->p, 1);
    int rc = launch_check("k_derive_phi");
    if (rc) return rc;
    h->stale_phi = false; h->stale_phivb = false;
    h->planes_valid = true;
    return 0;
}
// the per-row constants of k_momentum_dma, once per set of grid arrays
int ensure_row_constants(dycore *h)
{
    if (h->row_c_valid) return 0;
    hipLaunchKernelGGL(k_row_constants, dim3((h->g.NJ + 63) / 64), dim3(64), 0, h->stream, h->g, h->p, const_cast<real *>(h->p.row_c));
    int rc = launch_check("k_row_constants");
    if (rc) return rc;
    h->row_c_valid = true;
    return 0;
}
// Work list of a k_momentum_dma launch (MomPlan).  With `slots` workgroups resident at a time, T tiles of nz levels take
// ceil(T / slots) rounds; ks level chunks per tile take ceil(T * ks / slots) rounds of nz / ks levels + the chunk's
// prologue (2-D loads, factor tiles, first DMA: ~mom_pro levels).  Candidates: the whole row set in 1 .. 16 chunks per
// tile, or -- for a contiguous row set of more than one round -- the tile rows that fill whole rounds as full tiles
// (segment a) and the remaining rows cut into up to 16 chunks (segment b), so that the last round costs a fraction of a tile.
MomPlan make_mom_plan(const dycore *h, int gbx, const RowSet &trows, int slots)
{
    MomPlan plan = {trows, 1, 0, {0, 0, 0, 0}, 1, h->mom_patch};
    const long tiles = (long)gbx * trows.n;
    const double pro = h->mom_pro;
    auto rounds = [&](long wg) { return (double)((wg + slots - 1) / slots); };
    double best = -1.0;
    // any number of chunks up to 16 (chunk kc covers levels [kc nz / ks, (kc + 1) nz / ks)); a chunk of fewer
    // than 4 levels is all prologue
    const int nzb = (h->g.nz + PHIVB_STRIDE - 1) / PHIVB_STRIDE;     // level chunks are whole blocks of PHIVB_STRIDE levels (k_momentum_dma)
    auto chunk_levels = [&](int cand) { return (double)std::min(h->g.nz, PHIVB_STRIDE * ((nzb + cand - 1) / cand)); };
    for (int cand = 1; cand <= 16; cand++) {
        if (cand > nzb) break;
        const double cost = rounds(tiles * cand) * (chunk_levels(cand) + (cand > 1 ? pro : 0.0));
        if (best < 0 || cost < best - 1e-9) { best = cost; plan.ks_a = cand; }
    }
    if (h->mom_ks) plan.ks_a = std::min(h->mom_ks, nzb);
    const long full = tiles / slots;              // whole rounds of full tiles
    if (h->mom_split && !h->mom_ks && trows.gap == 0 && full >= 1 && tiles % slots) {
        const int r1 = (int)(full * slots / gbx);       // tile rows inside those rounds
        const int r2 = trows.n - r1;
        if (r1 >= 1 && r2 >= 1) {
            for (int cand = 1; cand <= 16; cand++) {
                if (cand > nzb) break;
                const double cost = rounds((long)gbx * r1) * h->g.nz + rounds((long)gbx * r2 * cand) * (chunk_levels(cand) + (cand > 1 ? pro : 0.0));
                if (cost < best - 1e-9) {
                    best = cost;
                    plan.a = {trows.first, r1, r1, 0}; plan.ks_a = 1;
                    plan.b = {trows.first + r1, r2, r2, 0}; plan.ks_b = cand;
                }
            }
        }
    }
    plan.nblk_a = 8 * ((gbx * plan.a.n * plan.ks_a + 7) / 8);
    return plan;
}
// prepare + dUFLXdt + dVFLXdt (+ Euler step of the winds when u_out != nullptr)
int do_momentum_fused(dycore *h, const Ptrs &p, const real *u_old, const real *v_old,
                      real *u_out, real *v_out, real dt, int mode = ROWS_ALL)
{
    // only the LDS-DMA kernel takes a tile-row subset; the others run whole with the edge part
    const bool dma = u_out && h->mom_dma == 1 && h->g.nz <= DMAXZ;
    if (!dma) {
        if (mode == ROWS_INNER) return 0;
        mode = ROWS_ALL;
    }
    if (!(dma && flux_from_winds(h) && (h->fw_mask & 2))) {       // every other momentum kernel reads the stored PVTF (and PHI)
        int rc = refresh_lazy(h, LAZY_PVTF | LAZY_PHI);
        if (rc) return rc;
    }
    if (dma) { int rc = ensure_row_constants(h); if (rc) return rc; }
    Scope s(h, dma ? K_MOM_DMA : K_MOM_FUSED);      // stats under the name of the kernel that runs
    const int gbx = (h->g.nx + FTI - 1) / FTI, gby = (h->g.ny + FTJ - 1) / FTJ;
    const int nblk = gbx * gby;
    const dim3 gr(8 * ((nblk + 7) / 8)), bl(FTI, FTJ);
    const int jv_lo = h->g.pole_s ? 2 : 1;
    if (dma) {
        // rows per workgroup: 8 (8 wavefronts = 2 per SIMD, up to 256 VGPRs each).  Taller tiles
        // (9, 10 rows: fewer halo rows, better fit of 90-row bands on 256 CUs) need a third
        // wavefront on one SIMD, which caps the kernel at 168 VGPRs: 2x slower on MI355X
        // (profiles/r01_notes.md).
        // 64 x 4 tiles (production; DYCORE_MOM_TJ=8 for 64 x 8): two workgroups of four wavefronts per CU instead of one of
        // eight.  Each has its own per-level barrier, so one workgroup's LDS reads meet the other's arithmetic (the kernel
        // without its DMA: 0.94 -> 0.80 ms per launch) at the price of 17 % more rows staged (7 halo rows per 4 instead of
        // 10 per 8; they hit in L2); the corrector keeps one DMA level in flight (two do not fit 80 KB of LDS).
        // 1.28-1.30 -> 1.23-1.25 ms per launch, profiles/r02_notes.md.
        const bool half = h->mom_tj == 4 && !h->g.i_turb && flux_from_winds(h) && (h->fw_mask & 2);
        const int tj = half ? 4 : FTJ;
        const int slots = h->n_cu * (half ? 2 : 1);           // workgroups resident at a time
        const int gby2 = (h->g.ny + tj - 1) / tj;
        // level chunks per tile: with one workgroup per CU, `tiles` workgroups of 64 levels take
        // ceil(tiles / CUs) rounds; ks chunks take ceil(tiles*ks / CUs) rounds of (64/ks + 1)
        // levels (one extra level of prologue per chunk).  Pick the cheapest of 1, 2, 4.
        // a tile reads rows j0-1 .. j0+tj+1 (VFLX): the first tile row of a band with a southern
        // neighbour and every tile row that reaches beyond row ny of one with a northern neighbour
        int dep_n = 0;
        if (!h->g.pole_n)
            for (int by = gby2 - 1; by >= 0 && by * tj + 1 + tj + 1 > h->g.ny; by--) dep_n++;
        RowSet trows;
        if (!row_subset(gby2, h->g.pole_s ? 0 : 1, dep_n, mode, trows)) return 0;
        const MomPlan plan = make_mom_plan(h, gbx, trows, slots);
        const dim3 gr2(plan.nblk_a + 8 * ((gbx * plan.b.n * plan.ks_b + 7) / 8)), blk(FTI, tj);
        const bool cur = (u_old == p.UWIND);
        const bool fw = flux_from_winds(h) && (h->fw_mask & 2);
        // ring depth: two levels in flight for the production (dry / moist, fluxes from the winds) instances
#define MOM_LAUNCH_P(CUR_, TJ_, FW_, TB_, DP_, PL_) hipLaunchKernelGGL((k_momentum_dma<CUR_, TJ_, FW_, TB_, DP_, PL_>), gr2, blk, dma_lds_bytes(TJ_, FW_, CUR_, DP_, PL_, false, TB_), h->stream, h->g, p, gbx, plan, jv_lo, u_old, v_old, u_out, v_out, dt, TempArgs{nullptr, nullptr, nullptr, nullptr})
#define MOM_LAUNCH_T(CUR_, TJ_, FW_, TB_, DP_) MOM_LAUNCH_P(CUR_, TJ_, FW_, TB_, DP_, false)
#define MOM_LAUNCH(CUR_, FW_, TB_, DP_) MOM_LAUNCH_T(CUR_, FTJ, FW_, TB_, DP_)
        const bool pl = phi_on_demand(h);
        if (half && pl) { int rc = ensure_planes(h); if (rc) return rc; }       // (step_begin has done it; a caller of the parts may not have)
#ifdef DYCORE_AB_SWITCHES
        if (half && pl) { if (!cur) MOM_LAUNCH_P(false, 4, true, false, 1, true); else if (h->mom_depth == 2) MOM_LAUNCH_P(true, 4, true, false, 2, true); else MOM_LAUNCH_P(true, 4, true, false, 1, true); }
        else if (half) { if (!cur) MOM_LAUNCH_T(false, 4, true, false, 1); else if (h->mom_depth == 2) MOM_LAUNCH_T(true, 4, true, false, 2); else MOM_LAUNCH_T(true, 4, true, false, 1); }
        else if (h->g.i_turb) { if (cur) MOM_LAUNCH(true, true, true, 2); else MOM_LAUNCH(false, true, true, 1); }
        else if (cur) { if (!fw) MOM_LAUNCH(true, false, false, 1); else if (h->mom_depth == 2) MOM_LAUNCH(true, true, false, 2); else MOM_LAUNCH(true, true, false, 1); }
        else { if (!fw) MOM_LAUNCH(false, false, false, 1); else if (h->mom_depth == 2) MOM_LAUNCH(false, true, false, 2); else MOM_LAUNCH(false, true, false, 1); }
#else
        // the shipped instances: 64 x 4 tiles with PHI in the thread (production), with PHI staged (DYCORE_LAZY_PHI=0), 64 x 8
        // turbulence-coupled, 64 x 8 on stored mass fluxes (level counts the register continuity kernel has no instance for)
        (void)fw;
        if (half && pl) { if (!cur) MOM_LAUNCH_P(false, 4, true, false, 1, true); else MOM_LAUNCH_P(true, 4, true, false, 2, true); }
        else if (half) { if (!cur) MOM_LAUNCH_T(false, 4, true, false, 1); else MOM_LAUNCH_T(true, 4, true, false, 2); }
        else if (h->g.i_turb) { if (cur) MOM_LAUNCH(true, true, true, 2); else MOM_LAUNCH(false, true, true, 1); }
        else if (cur) MOM_LAUNCH(true, false, false, 1);
        else MOM_LAUNCH(false, false, false, 1);
#endif
#undef MOM_LAUNCH
#undef MOM_LAUNCH_T
#undef MOM_LAUNCH_P
    } else if (u_out)
        hipLaunchKernelGGL(k_momentum_fused<true>, gr, bl, 0, h->stream, h->g, p, gbx, gby, jv_lo, u_old, v_old, u_out, v_out, dt);
    else
        hipLaunchKernelGGL(k_momentum_fused<false>, gr, bl, 0, h->stream, h->g, p, gbx, gby, jv_lo, u_old, v_old, u_out, v_out, dt);
    return launch_check(dma ? "k_momentum_dma" : "k_momentum_fused");
}
// momentum + temperature in one launch: see k_momentum_dma<..., WITH_TEMP>
bool substep_fused(const dycore *h)
{
    // A latitude band takes it when it is large (DYCORE_BANDS_STEP_FUSED=1 / 0: always / never): its pieces -- inner tile
    // rows in two halves around the edge rows, so that both exchanges stay hidden -- are launches of their own, and on a small
    // band, where each is about one round of workgroups, their ramps cost more than the single launch gains on a whole domain
    // (5 %): 0.25 deg x 64 in 2 / 4 / 8 emulated bands 2.49 / 1.35 / 0.815 ms per band-step against 2.585 / 1.35 / 0.727 on
    // the two kernels (profiles/r03_notes.md).  Bitwise the same step either way.
    bool band_ok = h->g.pole_s && h->g.pole_n;
    if (!band_ok) {
        const long tiles = (long)((h->g.nx + FTI - 1) / FTI) * ((h->g.ny + 3) / 4);
        band_ok = h->step_fused_bands > 0 || (h->step_fused_bands < 0 && tiles >= 6L * h->n_cu);
    }
    return h->step_fused && h->pvb_alt && band_ok && phi_on_demand(h) && h->g.nz >= 2;
}
// work list of the whole-substep launch: one workgroup of eight wavefronts per CU; the tile rows beyond the last full round
// are cut into level chunks
// `mode`: the tile rows of the launch (a latitude band: those clear of the neighbours' rows, in one piece or two, and the
// edge rows -- the dependence footprint is the momentum wavefronts', which covers the temperature wavefronts')
bool substep_plan(const dycore *h, int mode, MomPlan &plan)
{
    constexpr int tj = 4;
    const int gbx = (h->g.nx + FTI - 1) / FTI, gby = (h->g.ny + tj - 1) / tj;
    int dep_n = 0;
    if (!h->g.pole_n)
        for (int by = gby - 1; by >= 0 && by * tj + 1 + tj + 1 > h->g.ny; by--) dep_n++;
    RowSet trows;
    if (!row_subset(gby, h->g.pole_s ? 0 : 1, dep_n, mode, trows)) return false;
    plan = make_mom_plan(h, gbx, trows, h->n_cu);
    return true;
}
// the rows whose PHIVB planes that launch needs (the chunks' momentum wavefronts start from them: own row, southern
// neighbour), of the state `p` it evaluates on -- nobody else keeps them on this path.  At most one segment of a plan is cut.
// (whole domain in one launch: one contiguous run of rows, the rider of the halo launch)
bool substep_plane_rows(const dycore *h, const Ptrs &p, PlaneRows &r)
{
    constexpr int tj = 4;
    MomPlan plan;
    r = PlaneRows{p.COLP, p.HSURF, p.PVTFVB, p.POTT, p.sigma_vb, p.rdsigma, p.PHIVB, 1, 0};
    if (!substep_plan(h, ROWS_ALL, plan)) return false;
    for (int seg = 0; seg < 2; seg++) {
        const RowSet &rs = seg ? plan.b : plan.a;
        const int ks = seg ? plan.ks_b : plan.ks_a;
        if (rs.n <= 0 || ks <= 1) continue;
        const int j_lo = std::max(0, rs.first * tj), j_hi = std::min(h->g.ny + 1, (rs.first + rs.n) * tj);
        if (r.j_hi < r.j_lo) { r.j_lo = j_lo; r.j_hi = j_hi; }
        else { r.j_lo = std::min(r.j_lo, j_lo); r.j_hi = std::max(r.j_hi, j_hi); }
    }
    return r.j_hi >= r.j_lo;
}
// the whole-substep launch has written the new PVTFVB into the alternate buffer (momentum wavefronts of neighbouring tiles,
// and of a band's other pieces, were still reading the old one): swap it in -- once per substep, behind its last piece
void finish_substep_fused(dycore *h)
{
    std::swap(h->p.PVTFVB, h->pvb_alt);
    h->stale_pvtf = true; h->stale_phi = true; h->stale_phivb = true; h->stale_pottvb = true;
    h->planes_valid = false;        // this launch neither reads nor writes the planes as a whole
}
// the whole-substep launch advances QV / QC as well
bool moist_in_substep(const dycore *h) { return h->g.i_moist && h->moist_fused && substep_fused(h); }
int do_substep_fused(dycore *h, const Ptrs &p, const real *u_old, const real *v_old, const real *pott_old,
                     real *u_out, real *v_out, real *pott_out, real dt, int mode = ROWS_ALL,
                     const real *qv_old = nullptr, const real *qc_old = nullptr, real *qv_out = nullptr, real *qc_out = nullptr)
{
    constexpr int tj = 4;
    const int gbx = (h->g.nx + FTI - 1) / FTI;
    MomPlan plan;
    if (!substep_plan(h, mode, plan)) return 0;
    { int rc = ensure_row_constants(h); if (rc) return rc; }
    Scope s(h, K_SUBSTEP);
    const dim3 gr(plan.nblk_a + 8 * ((gbx * plan.b.n * plan.ks_b + 7) / 8)), blk(FTI, 2 * tj);
    const bool cur = (u_old == p.UWIND);
    // the PHIVB planes of the rows that are cut: as a rider of the halo launch behind the continuity kernel (substep_part),
    // or -- a caller that ran something in between -- by a launch of their own
    if (!(mode == ROWS_ALL && h->rows_ready)) {
        PlaneRows pr{p.COLP, p.HSURF, p.PVTFVB, p.POTT, p.sigma_vb, p.rdsigma, p.PHIVB, 1, 0};
        for (int seg = 0; seg < 2; seg++) {
            const RowSet &rs = seg ? plan.b : plan.a;
            const int ks = seg ? plan.ks_b : plan.ks_a;
            if (rs.n <= 0 || ks <= 1) continue;
            // contiguous runs of tile rows (a band's edge piece is its first and its last tile rows); a run needs the rows of
            // its tiles and the one south of them (a band's halo row 0: its POTT, PVTFVB rows have arrived by the time the
            // edge piece is issued)
            for (int r0 = 0; r0 < rs.n;) {
                int r1 = r0;
                while (r1 + 1 < rs.n && rowset_row(rs, r1 + 1) == rowset_row(rs, r1) + 1) r1++;
                pr.j_lo = std::max(0, rowset_row(rs, r0) * tj);
                pr.j_hi = std::min(h->g.ny + 1, (rowset_row(rs, r1) + 1) * tj);
                hipLaunchKernelGGL(k_phivb_rows, dim3((h->g.nx + 2 + TI - 1) / TI, (pr.j_hi - pr.j_lo + 1 + TJ - 1) / TJ), dim3(TI, TJ), 0, h->stream, h->g, pr);
                int rc = launch_check("k_phivb_rows");
                if (rc) return rc;
                r0 = r1 + 1;
            }
        }
    }
    if (mode == ROWS_ALL) h->rows_ready = false;
#ifdef DYC_TRACE
    static long long *trace_buf = nullptr;
    if (!trace_buf && hipMalloc((void **)&trace_buf, 8 * 256 * 8 * sizeof(long long)) == hipSuccess) (void)hipMemset(trace_buf, 0, 8 * 256 * 8 * sizeof(long long));
    const TempArgs ta = {pott_old, pott_out, h->pvb_alt, trace_buf, qv_old, qc_old, qv_out, qc_out};
    h->trace_buf = trace_buf;
#else
    const TempArgs ta = {pott_old, pott_out, h->pvb_alt, nullptr, qv_old, qc_old, qv_out, qc_out};
#endif
    const int jv_lo = h->g.pole_s ? 2 : 1;
    // (whole columns: PHI starts from the surface geopotential in every thread, no stored PHIVB plane is read or written)
    if (moist_in_substep(h)) {
        if (!qv_out || !qc_out || !qv_old || !qc_old) return fail(DYCORE_ESTATE, "whole-substep launch of a moist handle without its QV / QC buffers");
        if (cur) hipLaunchKernelGGL((k_momentum_dma<true, tj, true, false, 2, true, true, true>), gr, blk, dma_lds_bytes(tj, true, true, 2, true, true, false, true), h->stream,
                                    h->g, p, gbx, plan, jv_lo, u_old, v_old, u_out, v_out, dt, ta);
        else hipLaunchKernelGGL((k_momentum_dma<false, tj, true, false, 2, true, true, true>), gr, blk, dma_lds_bytes(tj, true, false, 2, true, true, false, true), h->stream,
                                h->g, p, gbx, plan, jv_lo, u_old, v_old, u_out, v_out, dt, ta);
    } else if (cur) hipLaunchKernelGGL((k_momentum_dma<true, tj, true, false, 2, true, true>), gr, blk, dma_lds_bytes(tj, true, true, 2, true, true), h->stream,
                                h->g, p, gbx, plan, jv_lo, u_old, v_old, u_out, v_out, dt, ta);
    else hipLaunchKernelGGL((k_momentum_dma<false, tj, true, false, 2, true, true>), gr, blk, dma_lds_bytes(tj, true, false, 2, true, true), h->stream,
                            h->g, p, gbx, plan, jv_lo, u_old, v_old, u_out, v_out, dt, ta);
    return launch_check("k_substep_fused");
}
int do_pott_fused(dycore *h, const Ptrs &p, const real *pott_old, real *pott_out, real dt)
{
    Scope s(h, (h->g.i_turb || (h->lazy && phi_on_demand(h))) ? K_TEMP_CANON : K_POTT_FUSED);
    const dim3 gr = grid_cells(h), bl(TI, TJ);
    const int lin = cells_linear(h);
#define POTT_LAUNCH_P(W_, F_, T_, V_, P_) hipLaunchKernelGGL((k_pott_fused<W_, F_, T_, V_, P_>), gr, bl, 0, h->stream, h->g, p, pott_old, pott_out, dt, lin)
#define POTT_LAUNCH(W_, F_, T_, V_) POTT_LAUNCH_P(W_, F_, T_, V_, true)
    if (h->g.i_turb) {                  // fused turbulence-coupled step: PHI and PHIVB are inputs of k_turb_tendencies, so they are stored;
        hipLaunchKernelGGL(k_temp_canon<true>, gr, bl, 0, h->stream, h->g, p, pott_old, pott_out, dt, lin);      // PVTF and POTTVB are derived when asked for
        h->stale_pvtf = true; h->stale_pottvb = true;
    } else if (h->lazy) {
        if (phi_on_demand(h)) {         // production: neither PHI nor POTTVB is stored (k_temp_canon)
            hipLaunchKernelGGL(k_temp_canon<false>, gr, bl, 0, h->stream, h->g, p, pott_old, pott_out, dt, lin);
            h->stale_pvtf = true; h->stale_phi = true; h->stale_pottvb = true;
            h->planes_valid = true;     // ... and of the halo rows once their owner's rows have arrived (a band's exchange)
        }
        else if (pvtf_on_demand(h)) { POTT_LAUNCH(false, true, false, false); h->stale_pvtf = true; }
#ifdef DYCORE_AB_SWITCHES
        else if (flux_from_winds(h) && (h->fw_mask & 4)) POTT_LAUNCH(false, true, false, true);
#endif
        else POTT_LAUNCH(false, false, false, true);
        h->stale_phivb = true;
    } else {
#ifdef DYCORE_AB_SWITCHES
        POTT_LAUNCH(true, false, false, true);
#else
        return fail(DYCORE_ESTATE, "the fused step without on-demand fields is an experiment build's (DYCORE_AB_SWITCHES)");
#endif
    }
#undef POTT_LAUNCH
#undef POTT_LAUNCH_P
    return launch_check("k_pott_fused");
}
// the turbulent tendencies of the fused turbulence-coupled step (one launch per substep)
// `mode`: all rows, or -- for a band that hides its exchange -- the rows clear of the neighbours' rows (INNER) and the
// one row next to each neighbour (EDGE)
int do_turb_tendencies(dycore *h, const Ptrs &p, int mode = ROWS_ALL)
{
    RowSet rs;
    if (!row_subset(h->g.ny, h->g.pole_s ? 0 : 1, h->g.pole_n ? 0 : 1, mode, rs)) return 0;
    Scope s(h, K_TURB_TEND);
    const int lin = cells_linear(h);
    long long n;
    if (lin & 1) n = ((long long)h->g.nx * rs.n + TI * TJ - 1) / (TI * TJ);
    else n = (long long)((h->g.nx + TI - 1) / TI) * ((rs.n + TJ - 1) / TJ);
    hipLaunchKernelGGL(k_turb_tendencies, dim3((unsigned)(8 * ((n + 7) / 8))), dim3(TI, TJ), 0, h->stream, h->g, p, h->g.pole_s ? 2 : 1, lin, rs);
    return launch_check("k_turb_tendencies");
}
// dQVdt, dQCdt (stage) or the new QV, QC directly (fused): dyn_moist.py
int do_moist(dycore *h, const Ptrs &p, const real *qv_old, const real *qc_old, real *qv_out, real *qc_out, real dt)
{
    Scope s(h, K_MOIST);
    const dim3 gr = grid_cells(h), bl(TI, TJ);
    const int lin = cells_linear(h);
    if (qv_out) {
        // the fused step: k_moist_canon (the level's arithmetic in the forms the whole-substep kernel's temperature wavefronts use)
        if (h->g.i_turb && fused_path(h)) hipLaunchKernelGGL((k_moist_canon<true, true>), gr, bl, 0, h->stream, h->g, p, qv_old, qc_old, qv_out, qc_out, dt, lin);
        else if (h->g.i_turb) hipLaunchKernelGGL((k_moist<true, true>), gr, bl, 0, h->stream, h->g, p, qv_old, qc_old, qv_out, qc_out, dt, lin);
        else if (fused_path(h) && flux_from_winds(h) && (h->fw_mask & 4)) hipLaunchKernelGGL((k_moist_canon<true>), gr, bl, 0, h->stream, h->g, p, qv_old, qc_old, qv_out, qc_out, dt, lin);
        else hipLaunchKernelGGL((k_moist_canon<false>), gr, bl, 0, h->stream, h->g, p, qv_old, qc_old, qv_out, qc_out, dt, lin);
    } else {
        if (h->g.i_turb) hipLaunchKernelGGL((k_moist<true, false>), gr, bl, 0, h->stream, h->g, p, qv_old, qc_old, qv_out, qc_out, dt, lin);
        else hipLaunchKernelGGL((k_moist<false, false>), gr, bl, 0, h->stream, h->g, p, qv_old, qc_old, qv_out, qc_out, dt, lin);
    }
    return launch_check("k_moist");
}

int do_momentum(dycore *h, const Ptrs &p)
{
    int rc;
    if (h->g.i_turb) {   // dyn_org_discretizations.py:195-196, 212-216
        if ((rc = do_exchange(h, p.KMOM, 0, 0, h->g.nz + 1))) return rc;
    }
    if ((rc = do_prepare(h, p))) return rc;
    if (h->g.i_turb) {   // KMOM_d*dz halos are written by k_prepare itself
        if ((rc = do_exchange(h, p.SMOMXFLX, 0, 0, 1))) return rc;
        if ((rc = do_exchange(h, p.SMOMYFLX, 0, 0, 1))) return rc;
    }
    if ((rc = do_uflx(h, p))) return rc;
    return do_vflx(h, p);
}

// dyn_matsuno.py:28-131 with the *_OLD copies replaced by pointer rotation: after the swap
// X_OLD holds the state at step start; the predictor evaluates tendencies on X_OLD and
// writes the estimate into X, the corrector evaluates on X and overwrites X in place
// (k_euler_forward reads X_OLD and the tendency at the written cell only).
// The step is cut into parts at the two points where a latitude band needs rows from its
// neighbours (after continuity, after the Euler step); a single band runs them back to back.
// the turbulence-coupled fused step needs the kernels that re-form the fluxes from the winds (LDS-DMA
// momentum, register continuity); otherwise an i_turb handle steps on the stage kernels
bool fused_path(const dycore *h) { return h->fused && h->zonal && (!h->g.i_turb || (h->fused_turb && flux_from_winds(h) && h->fw_mask == 7)); }

// Fused (production, dry) scheme -- no time-level swaps: X holds the state S0 at step start
// through the predictor, X_EST receives the estimate S1, the corrector evaluates on X_EST and
// overwrites X in place with S2 (it reads X at the own cell only).  X_OLD is not maintained.
// Staged scheme (turbulence-coupled, or DYCORE_FUSED=0): time levels rotate by pointer swap,
// after which X_OLD holds S0; the predictor evaluates on X_OLD and writes X, the corrector
// evaluates on X and overwrites X in place.
int step_begin(dycore *h, int with_rho)
{
    int rc;
    Ptrs &p = h->p;
    if (with_rho) {
        if (h->g.i_turb && fused_path(h)) {     // PVTF / POTTVB are not stored on this path (k_temp_canon<true>): formed in the thread
            Scope s(h, K_RHO);
            hipLaunchKernelGGL(k_diag_rho_canon, grid_for(h->g, h->g.nx + 2, h->g.ny + 2), dim3(TI, TJ), 0, h->stream, h->g, p);
            if ((rc = launch_check("k_diag_rho_canon"))) return rc;
        } else if ((rc = refresh_lazy(h, LAZY_PVTF | LAZY_POTTVB)) || (rc = do_rho(h, p))) return rc;
    }
    if (fused_path(h)) {
        if (!h->g.i_turb && !substep_fused(h) && (rc = ensure_planes(h))) return rc;
        if (h->g.i_turb) {
            // dyn_org_discretizations.py:195-196, 212-216: lateral BC of the K field and the surface momentum
            // fluxes, once per step (they do not change within one).  Whole-field exchange as in the staged
            // step: these inputs never travel between bands, so the periodic columns of a band's halo rows
            // are refreshed here too.
            if ((rc = do_exchange(h, p.KMOM, 0, 0, h->g.nz + 1))) return rc;
            if ((rc = do_exchange(h, p.SMOMXFLX, 0, 0, 1))) return rc;
            if ((rc = do_exchange(h, p.SMOMYFLX, 0, 0, 1))) return rc;
        }
        if (h->colp_old_ready) return 0;        // (filled by the launch that closed the previous step; nothing has touched the handle since)
        return do_copy(h, p.COLP_OLD, p.COLP, (size_t)h->g.plane);
    }
    std::swap(p.UWIND, p.UWIND_OLD);
    std::swap(p.VWIND, p.VWIND_OLD);
    std::swap(p.POTT, p.POTT_OLD);
    std::swap(p.COLP, p.COLP_OLD);
    if (h->g.i_moist) { std::swap(p.QV, p.QV_OLD); std::swap(p.QC, p.QC_OLD); }
    return 0;
}
// the fields the tendencies of a substep are evaluated on
Ptrs substep_view(dycore *h, int sub)
{
    Ptrs q = h->p;
    if (fused_path(h)) {
        if (sub == 1) { q.UWIND = h->p.UWIND_EST; q.VWIND = h->p.VWIND_EST; q.POTT = h->p.POTT_EST; q.QV = h->p.QV_EST; q.QC = h->p.QC_EST; }
    } else if (sub == 0) {         // predictor: the current state IS the *_OLD buffers
        q.UWIND = h->p.UWIND_OLD; q.VWIND = h->p.VWIND_OLD; q.POTT = h->p.POTT_OLD; q.COLP = h->p.COLP_OLD;
        q.QV = h->p.QV_OLD; q.QC = h->p.QC_OLD;
    }
    return q;
}
int substep_part(dycore *h, real dt, int sub, int part)
{
    int rc;
    Ptrs &p = h->p;
    const Ptrs q = substep_view(h, sub);
    if (fused_path(h)) {
        real *u_out = sub == 0 ? p.UWIND_EST : p.UWIND;
        real *v_out = sub == 0 ? p.VWIND_EST : p.VWIND;
        real *t_out = sub == 0 ? p.POTT_EST : p.POTT;
        real *qv_out = sub == 0 ? p.QV_EST : p.QV, *qc_out = sub == 0 ? p.QC_EST : p.QC;
        switch (part) {
        case 0: {                   // -> UFLX, VFLX, WWIND, COLP_NEW halo rows
            PlaneRows pr;           // (+ the PHIVB planes the whole-substep launch of part 1 starts its level chunks from)
            // (a whole domain only: a band issues the launch in pieces, each with the planes of its own cut rows)
            const bool rider = substep_fused(h) && h->g.pole_s && h->g.pole_n && substep_plane_rows(h, q, pr);
            return do_continuity_fused(h, q, dt, rider ? &pr : nullptr);
        }
        case DYCORE_PART_MOM_INNER:     // turbulence-coupled: the turbulent tendencies of the same rows first
        case DYCORE_PART_MOM_INNER_A:
        case DYCORE_PART_MOM_INNER_B:
            // momentum + temperature in one launch (round 3: latitude bands too): a piece writes the new winds, POTT and -- into
            // the alternate buffer, swapped in by part TEMP -- PVTFVB of its rows
            if (substep_fused(h))
                return do_substep_fused(h, q, p.UWIND, p.VWIND, p.POTT, u_out, v_out, t_out, dt,
                                        part == DYCORE_PART_MOM_INNER ? ROWS_INNER : part == DYCORE_PART_MOM_INNER_A ? ROWS_INNER_A : ROWS_INNER_B,
                                        p.QV, p.QC, qv_out, qc_out);
            if (part == DYCORE_PART_MOM_INNER_B) return 0;      // (two kernels: INNER_A is the whole inner piece, part TEMP hides the wind rows)
            if (h->g.i_turb && (rc = do_turb_tendencies(h, q, ROWS_INNER))) return rc;
            return do_momentum_fused(h, q, p.UWIND, p.VWIND, u_out, v_out, dt, ROWS_INNER);
        case DYCORE_PART_MOM_EDGE:  // -> new winds complete: their rows can travel while part TEMP runs
            if (substep_fused(h)) return do_substep_fused(h, q, p.UWIND, p.VWIND, p.POTT, u_out, v_out, t_out, dt, ROWS_EDGE, p.QV, p.QC, qv_out, qc_out);
            if (h->g.i_turb && (rc = do_turb_tendencies(h, q, ROWS_EDGE))) return rc;
            // the rows that leave now are packed with their periodic images (k_rows); the lateral halo of
            // the new winds is refreshed with everything else at the end of part TEMP
            return do_momentum_fused(h, q, p.UWIND, p.VWIND, u_out, v_out, dt, ROWS_EDGE);
        case 1:
        case DYCORE_PART_TEMP: {    // momentum reads the old diagnostics, so it runs first
            if (substep_fused(h)) {
                if (part == 1 && (rc = do_substep_fused(h, q, p.UWIND, p.VWIND, p.POTT, u_out, v_out, t_out, dt, ROWS_ALL, p.QV, p.QC, qv_out, qc_out))) return rc;
                finish_substep_fused(h);        // (the pieces of a split part 1 have all been issued by now)
            } else {
                if (part == 1) {
                    if (h->g.i_turb && (rc = do_turb_tendencies(h, q))) return rc;
                    if ((rc = do_momentum_fused(h, q, p.UWIND, p.VWIND, u_out, v_out, dt))) return rc;
                }
                if ((rc = do_pott_fused(h, q, p.POTT, t_out, dt))) return rc;
            }
            // (QV, QC: by the whole-substep launch itself, or by the moisture kernel behind it)
            if (h->g.i_moist && !moist_in_substep(h) && (rc = do_moist(h, q, p.QV, p.QC, qv_out, qc_out, dt))) return rc;
            const int nz = h->g.nz;
            std::vector<HaloSpec> hl = {{t_out, 0, 0, nz}, {p.PVTFVB, 0, 0, nz + 1}};
            if (h->g.i_moist) { hl.push_back({qv_out, 0, 0, nz}); hl.push_back({qc_out, 0, 0, nz}); }     // (one launch for everything: at most 10 of its 12 slots)
            // (a turbulence-coupled handle stores PHI and PHIVB, not POTTVB and PVTF: k_temp_canon<true>)
            if (!phi_on_demand(h)) { if (!h->g.i_turb) hl.push_back({p.POTTVB, 0, 0, nz + 1}); hl.push_back({p.PHI, 0, 0, nz}); }
            if (!pvtf_on_demand(h) && !h->g.i_turb) hl.push_back({p.PVTF, 0, 0, nz});
            if (!h->lazy || h->g.i_turb) hl.push_back({p.PHIVB, 0, 0, nz + 1});
            hl.push_back({u_out, 1, 0, nz});   // owned rows and pole rows only: rows in flight from a neighbour are not touched
            hl.push_back({v_out, 0, 1, nz});
            // + COLP <- COLP_NEW (dyn_matsuno.py:64-67) as a rider of the same launch  -> winds, POTT halo rows
            // (the corrector's also fills COLP_OLD: the next step begins with COLP_OLD <- COLP, a launch of its own otherwise --
            // one of nine per step, which is what a coarse grid's step is made of)
            rc = do_halo(h, hl, p.COLP, p.COLP_NEW, (size_t)h->g.plane, nullptr, sub == 1 ? p.COLP_OLD : nullptr);
            if (!rc && sub == 1) h->colp_old_ready = true;
            return rc;
        }
        case 2:             // nothing: the diagnostics of a band's halo rows arrive from their owner with the POTT rows
            return 0;       // (dycore_band_row_fields), like the rows themselves -- same bits as the single-domain run
        }
        return fail(DYCORE_EINVAL, "substep part %d out of range", part);
    }
    switch (part) {
    case DYCORE_PART_MOM_INNER:
    case DYCORE_PART_MOM_INNER_A:
    case DYCORE_PART_MOM_INNER_B:
    case DYCORE_PART_MOM_EDGE:
    case DYCORE_PART_TEMP:
        return fail(DYCORE_ESTATE, "split parts need the fused step (dycore_overlap_parts)");
    case 0:                         // -> needs UFLX, VFLX, WWIND, COLP_NEW halo rows afterwards
        return do_continuity(h, q, dt);
    case 1:
        if ((rc = do_momentum(h, q))) return rc;
        if ((rc = do_pott(h, q))) return rc;
        if (h->g.i_moist && (rc = do_moist(h, q, nullptr, nullptr, nullptr, nullptr, dt))) return rc;
        // COLP <- COLP_NEW (dyn_matsuno.py:64-67), then the Euler step writes p.UWIND/VWIND/POTT
        if ((rc = do_copy(h, p.COLP, p.COLP_NEW, (size_t)h->g.plane))) return rc;
        return do_euler(h, p, dt);  // -> needs UWIND, VWIND, POTT halo rows afterwards
    case 2:
        return do_diag(h, p);
    }
    return fail(DYCORE_EINVAL, "substep part %d out of range", part);
}
int do_step(dycore *h, real dt, int with_rho)
{
    int rc;
    if ((rc = step_begin(h, with_rho))) return rc;
    for (int sub = 0; sub < 2; sub++)
        for (int part = 0; part < 3; part++)
            if ((rc = substep_part(h, dt, sub, part))) return rc;
    return 0;
}

}  // namespace

// =========================================================================================
extern "C" {

const char *dycore_last_error(void) { return g_err.c_str(); }
int dycore_abi_version(void) { return DYCORE_ABI_VERSION; }
int dycore_real_bytes(void) { return (int)sizeof(real); }

int dycore_create(const DycoreConfig *cfg, dycore_t **out)
{
    if (!cfg || !out) return fail(DYCORE_EINVAL, "null argument");
    if (cfg->abi_version != DYCORE_ABI_VERSION)
        return fail(DYCORE_EINVAL, "ABI version mismatch: caller %d, library %d", cfg->abi_version, DYCORE_ABI_VERSION);
    if (cfg->nx < 3 || cfg->ny < 2 || cfg->nz < 2)
        return fail(DYCORE_EINVAL, "extents too small: nx=%d ny=%d nz=%d (need nx>=3, ny>=2, nz>=2)", cfg->nx, cfg->ny, cfg->nz);
    if (cfg->ny_global < cfg->ny || cfg->j_offset < 0 || cfg->j_offset + cfg->ny > cfg->ny_global)
        return fail(DYCORE_EINVAL, "band [%d,%d] outside the global grid of %d rows", cfg->j_offset + 1, cfg->j_offset + cfg->ny, cfg->ny_global);
    if ((double)(((cfg->nx + 3 + 63) / 64) * 64) * (cfg->ny + 3) * (cfg->nz + 1) * sizeof(real) >= 4294967296.0)
        return fail(DYCORE_EINVAL, "a field of %d x %d x %d cells is 4 GiB or more: cut the grid into latitude bands", cfg->nx, cfg->ny, cfg->nz);
    if (cfg->term_off & ~DYCORE_TERM_ALL) return fail(DYCORE_EINVAL, "term_off = 0x%x has bits outside DYCORE_TERM_ALL", cfg->term_off);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(DYCORE_EINVAL, "device %d not present (%d visible)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    dycore *h = new dycore();
    h->cfg = *cfg;
    Geo &g = h->g;
    g.nx = cfg->nx; g.ny = cfg->ny; g.nz = cfg->nz;
    g.P = ((cfg->nx + 3 + 63) / 64) * 64;
    g.NJ = cfg->ny + 3;
    g.plane = (long long)g.P * g.NJ;
    g.ny_global = cfg->ny_global; g.joff = cfg->j_offset;
    g.pole_s = (cfg->j_offset == 0);
    g.pole_n = (cfg->j_offset + cfg->ny == cfg->ny_global);
    g.i_turb = cfg->i_turb != 0; g.i_rad = cfg->i_radiation != 0; g.i_moist = cfg->i_moist != 0;
    g.i_moist_clip = cfg->i_moist_clip != 0;
    g.term_off = cfg->term_off;
    g.pair_top = cfg->pair_top;
    memset(&h->p, 0, sizeof h->p);
    // Switches of the shipped library ("Switches"): the documented fallback and what the parity tests
    // compare the production step with.  None of them changes a result beyond rounding.
    if (const char *e = getenv("DYCORE_FUSED")) h->fused = atoi(e) != 0;                    // 0: stage kernels for everything (the fallback)
    if (g.term_off) h->fused = false;       // a tendency term switched off (namelist.py:62-88): the stage kernels carry the switches
    if (const char *e = getenv("DYCORE_FUSED_TURB")) h->fused_turb = atoi(e) != 0;          // 0: stage kernels for turbulence-coupled handles
    if (const char *e = getenv("DYCORE_BANDS_STEP_FUSED")) h->step_fused_bands = atoi(e) != 0;  // 1 / 0: latitude bands always / never on the whole-substep kernel (default: large bands, see substep_fused)
    if (const char *e = getenv("DYCORE_MOIST_FUSED")) h->moist_fused = atoi(e) != 0;        // 0: the moisture kernel as a launch of its own behind the whole-substep kernel
    if (const char *e = getenv("DYCORE_STEP_FUSED")) h->step_fused = atoi(e) != 0;          // 0: momentum and temperature as two kernels on one GPU too (what a band runs)
    if (const char *e = getenv("DYCORE_LAZY_PHI")) h->lazy_phi = atoi(e) != 0;              // 0: PHI and POTTVB stored and staged (round 2's data flow)
    if (const char *e = getenv("DYCORE_MOM_KS")) h->mom_ks = atoi(e);                       // level chunks per tile forced (work-list tests)
    if (const char *e = getenv("DYCORE_MOM_SPLIT")) h->mom_split = atoi(e);                 // 0: no second work-list segment
    if (const char *e = getenv("DYCORE_CELL_LINEAR")) h->cell_lin = atoi(e);                // launch shape of the column kernels (bitwise-neutral, tested)
#ifdef DYCORE_AB_SWITCHES
    // A/B switches of experiment builds (-DDYCORE_AB_SWITCHES: tools/experiments/build_variant.sh): older data flows and tile
    // shapes whose kernel instances the shipped library does not contain
    if (const char *e = getenv("DYCORE_MOM_DMA")) h->mom_dma = atoi(e);
    if (const char *e = getenv("DYCORE_LAZY")) h->lazy = atoi(e) != 0;
    if (const char *e = getenv("DYCORE_LAZY_FLX")) h->lazy_flx = atoi(e) != 0;
    if (const char *e = getenv("DYCORE_LAZY_PVTF")) h->lazy_pvtf = atoi(e) != 0;
    if (const char *e = getenv("DYCORE_FW_MASK")) h->fw_mask = atoi(e);
    if (const char *e = getenv("DYCORE_MOM_PATCH")) h->mom_patch = atoi(e) > 1 ? atoi(e) : 1;
    if (const char *e = getenv("DYCORE_MOM_TJ")) h->mom_tj = atoi(e) == 4 ? 4 : 8;
    if (const char *e = getenv("DYCORE_MOM_DEPTH")) h->mom_depth = atoi(e) == 1 ? 1 : 2;
    if (const char *e = getenv("DYCORE_MOM_PRO")) h->mom_pro = atof(e);
#endif
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
    }
    {   // the LDS-DMA kernel keeps up to 155 KB of tiles in LDS: raise the dynamic-LDS limit of every instantiation once
        bool ok_plain = true, ok_fw = true, ok_deep = true, ok_half = true;
#define MOM_ATTR_P(CUR_, TJ_, FW_, TB_, DP_, PL_, FLAG_)                                                              \
        if (hipFuncSetAttribute((const void *)k_momentum_dma<CUR_, TJ_, FW_, TB_, DP_, PL_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)dma_lds_bytes(TJ_, FW_, CUR_, DP_, PL_, false, TB_)) != hipSuccess) { (void)hipGetLastError(); FLAG_ = false; }
#define MOM_ATTR_T(CUR_, TJ_, FW_, TB_, DP_, FLAG_) MOM_ATTR_P(CUR_, TJ_, FW_, TB_, DP_, false, FLAG_)
        bool ok_pl = true;
#ifdef DYCORE_AB_SWITCHES
        MOM_ATTR_P(true, 4, true, false, 1, true, ok_pl)
#endif
        MOM_ATTR_P(true, 4, true, false, 2, true, ok_pl) MOM_ATTR_P(false, 4, true, false, 1, true, ok_pl)
        if (!ok_pl) h->lazy_phi = false;
        bool ok_wt = true;
        if (hipFuncSetAttribute((const void *)k_momentum_dma<true, 4, true, false, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dma_lds_bytes(4, true, true, 2, true, true)) != hipSuccess) { (void)hipGetLastError(); ok_wt = false; }
        if (hipFuncSetAttribute((const void *)k_momentum_dma<false, 4, true, false, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dma_lds_bytes(4, true, false, 2, true, true)) != hipSuccess) { (void)hipGetLastError(); ok_wt = false; }
        if (!ok_wt) h->step_fused = false;
        bool ok_mo = true;      // ... and with the moisture sweep in its temperature wavefronts
        if (hipFuncSetAttribute((const void *)k_momentum_dma<true, 4, true, false, 2, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dma_lds_bytes(4, true, true, 2, true, true, false, true)) != hipSuccess) { (void)hipGetLastError(); ok_mo = false; }
        if (hipFuncSetAttribute((const void *)k_momentum_dma<false, 4, true, false, 2, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)dma_lds_bytes(4, true, false, 2, true, true, false, true)) != hipSuccess) { (void)hipGetLastError(); ok_mo = false; }
        if (!ok_mo) h->moist_fused = false;
#define MOM_ATTR(CUR_, FW_, TB_, DP_, FLAG_) MOM_ATTR_T(CUR_, FTJ, FW_, TB_, DP_, FLAG_)
#ifdef DYCORE_AB_SWITCHES
        MOM_ATTR_T(true, 4, true, false, 1, ok_half)
        MOM_ATTR(true, true, false, 1, ok_fw) MOM_ATTR(false, true, false, 1, ok_fw)
        MOM_ATTR(true, true, false, 2, ok_deep) MOM_ATTR(false, true, false, 2, ok_deep)
#endif
        MOM_ATTR_T(true, 4, true, false, 2, ok_half) MOM_ATTR_T(false, 4, true, false, 1, ok_half)
        MOM_ATTR(true, false, false, 1, ok_plain) MOM_ATTR(false, false, false, 1, ok_plain)
        MOM_ATTR(true, true, true, 2, ok_fw) MOM_ATTR(false, true, true, 1, ok_fw)
#undef MOM_ATTR
#undef MOM_ATTR_T
#undef MOM_ATTR_P
        if (!ok_plain) h->mom_dma = 0;
        if (!ok_fw) h->lazy_flx = false;
#ifdef DYCORE_AB_SWITCHES
        if (!ok_deep) h->mom_depth = 1;
        if (!ok_half) h->mom_tj = 8;
#else
        (void)ok_deep;
        if (!ok_half) h->fused = false;     // (never seen: the 64 x 4 instances need the least LDS) -> stage kernels
#endif
    }
    if (cfg->stream) h->stream = (hipStream_t)cfg->stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(DYCORE_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    h->home = h->stream;
    const real nan = std::numeric_limits<real>::quiet_NaN();
    // Every array is placed so that column i = 1 (the first owned longitude) of every row sits
    // on a 512-B boundary: a wavefront's 64 x 8 B row segment then covers exactly four 128-B
    // lines instead of straddling five (FETCH_SIZE -20 % on the column kernels).  The 63
    // doubles in front also absorb the i0-2 start of the first momentum tile.
    auto alloc = [&](size_t elems, real **out_p) -> int {
        char *raw = nullptr;
        constexpr size_t LEAD = 512 / sizeof(real) - 1;       // elements in front of i = 0, so that i = 1 sits on the 512-B boundary
        const size_t bytes = (elems + LEAD + 1 + 128) * sizeof(real) + 512;
        hipError_t e = hipMalloc((void **)&raw, bytes);
        if (e != hipSuccess) return fail(DYCORE_ENOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        h->allocs.push_back((real *)raw);
        char *al = (char *)(((uintptr_t)raw + 511) & ~(uintptr_t)511);
        real *d = (real *)al + LEAD;
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, h->stream, (real *)al, elems + LEAD + 1 + 128, nan);   // main_fields.py:484
        *out_p = d;
        return 0;
    };
    int rc = 0;
    for (int f = 0; f < kNFields && !rc; f++) {
        real *d = nullptr;
        if (!kFields[f].rad || (kFields[f].rad == 1 && g.i_rad) || (kFields[f].rad == 2 && g.i_turb)) rc = alloc(h->field_elems(kFields[f].nk), &d);
        h->p.*(kFields[f].member) = d;
        h->field_idx[kFields[f].name] = f;
    }
    if (!rc && g.i_rad) {           // columns per radiation launch: bounded workspace (<= 1 GiB)
        const long long cols = (long long)g.nx * g.ny, per = (long long)rad_ws_per_column(g.nz);
        long long chunk = ((1LL << 27) / per) / 256 * 256;
        if (chunk < 256) chunk = 256;
        if (chunk > cols) chunk = (cols + 255) / 256 * 256;
        void *d = nullptr;          // doubles in either precision, outside the field registry
        if (hipMalloc(&d, (size_t)(chunk * per) * sizeof(double)) != hipSuccess) rc = fail(DYCORE_ENOMEM, "hipMalloc of the radiation workspace failed");
        else h->allocs.push_back((real *)d);
        h->rad_ws = (double *)d;
        h->rad_ws_cols = chunk;
    }
    for (int f = 0; f < kNGrid && !rc; f++) {
        real *d = nullptr;
        const size_t elems = kGrid[f].stgx < 0 ? (size_t)(g.nz + 1) : (size_t)g.plane;
        rc = alloc(elems, &d);
        h->p.*(kGrid[f].member) = d;
        h->grid_idx[kGrid[f].name] = f;
    }
    if (!rc) {
        real *d = nullptr;
        rc = alloc((size_t)(g.nz + 1), &d);
        h->p.rdsigma = d;
    }
    if (!rc) {
        real *d = nullptr;
        rc = alloc((size_t)(5 * (g.nz + 2)), &d);
        h->p.mom_c = d;
    }
    if (!rc) {
        real *d = nullptr;
        rc = alloc((size_t)(4 * g.NJ), &d);
        h->p.row_c = d;
    }
    if (!rc && h->step_fused && !g.i_turb) {       // (substep_fused: the other conditions can change with the grid)
        rc = alloc(h->field_elems(-1), &h->pvb_alt);         // nz + 1 planes
    }
    if (!rc) {
        h->stage_elems = (size_t)(g.nx + 3) * (g.ny + 3) * (g.nz + 1);
        real *d = nullptr;
        rc = alloc(h->stage_elems, &d);
        h->stage = d;
    }
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(DYCORE_EHIP, "initial fill failed");
    if (rc) { dycore_destroy(h); return rc; }
    *out = h;
    return 0;
}

int dycore_destroy(dycore_t *h)
{
    if (!h) return 0;
    (void)hipStreamSynchronize(h->stream);
    for (auto &r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    for (real *d : h->allocs) (void)hipFree(d);
    if (h->own_stream) (void)hipStreamDestroy(h->home);
    delete h;
    return 0;
}

int dycore_use_stream(dycore_t *h, void *stream)
{
    if (!h) return fail(DYCORE_EINVAL, "null handle");
    h->stream = stream ? (hipStream_t)stream : h->home;
    return 0;
}
void *dycore_get_stream(dycore_t *h) { return h ? (void *)h->home : nullptr; }
int dycore_overlap_parts(dycore_t *h) { return h && fused_path(h) && flux_from_winds(h) ? 1 : 0; }
// Diagnostic fields whose halo rows a latitude band receives from their owner together with the POTT rows (the band does
// not re-diagnose rows it does not own: the owner's bits travel).  Which ones depends on what the handle's step stores.
static void band_row_fields(const dycore *h, std::vector<const char *> &out)
{
    if (!fused_path(h)) return;                 // the staged step diagnoses the whole band, halo rows included (part 2)
    out.push_back("PVTFVB");
    if (substep_fused(h)) return;               // the whole-substep kernel keeps nothing else: PHI is carried in its threads
    if (!h->g.i_turb && phi_on_demand(h)) { out.push_back("PHIVB"); return; }       // (the planes the level chunks start from)
    out.push_back("PHI");
    if (h->g.i_turb) { out.push_back("PHIVB"); return; }        // (k_temp_canon<true>: PVTF and POTTVB are derived, halo rows included, when asked for)
    out.push_back("POTTVB");
    if (!pvtf_on_demand(h)) out.push_back("PVTF");
    if (!h->lazy) out.push_back("PHIVB");
}
int dycore_band_row_fields(dycore_t *h, char *buf, int32_t nbytes)
{
    if (!h || !buf || nbytes < 1) return fail(DYCORE_EINVAL, "bad argument");
    std::vector<const char *> v;
    band_row_fields(h, v);
    std::string s;
    for (const char *n : v) { if (!s.empty()) s += ','; s += n; }
    if ((int)s.size() + 1 > nbytes) return fail(DYCORE_EINVAL, "buffer of %d bytes too small", nbytes);
    memcpy(buf, s.c_str(), s.size() + 1);
    return 0;
}
int dycore_flux_from_winds(dycore_t *h) { return h && fused_path(h) && flux_from_winds(h) ? 1 : 0; }
#ifdef DYC_TRACE
extern "C" int dycore_debug_trace(dycore_t *h, long long *host)      // 8 waves x 256 levels x 8 points
{
    if (!h || !h->trace_buf) return fail(DYCORE_ESTATE, "no trace");
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(host, h->trace_buf, 8 * 256 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif
int dycore_sync(dycore_t *h)
{
    if (!h) return fail(DYCORE_EINVAL, "null handle");
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int dycore_set_grid(dycore_t *h, const char *name, const real *host, int32_t n0, int32_t n1, int32_t n2)
{
    if (!h || !name || !host) return fail(DYCORE_EINVAL, "null argument");
    auto it = h->grid_idx.find(name);
    if (it == h->grid_idx.end()) return fail(DYCORE_EINVAL, "unknown grid array '%s'", name);
    const GridDesc &d = kGrid[it->second];
    const Geo &g = h->g;
    real *dev = const_cast<real *>(h->p.*(d.member));
    if (d.stgx < 0) {
        const int n = d.nk == 0 ? g.nz : g.nz + 1;
        if ((long long)n0 * n1 * n2 != n) return fail(DYCORE_EINVAL, "grid array '%s': expected %d elements, got %dx%dx%d", name, n, n0, n1, n2);
        HIP_TRY(hipMemcpyAsync(dev, host, n * sizeof(real), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (d.member == &Ptrs::dsigma) {
            std::vector<real> r(n);
            for (int k = 0; k < n; k++) r[k] = 1.0 / host[k];
            HIP_TRY(hipMemcpyAsync(const_cast<real *>(h->p.rdsigma), r.data(), n * sizeof(real), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            h->h_dsigma.assign(host, host + n);
        }
        if (d.member == &Ptrs::sigma_vb) h->h_sigma_vb.assign(host, host + n);
        if (d.member == &Ptrs::UVFLX_dif_coef) h->h_uvdif.assign(host, host + n);
        if ((d.member == &Ptrs::dsigma || d.member == &Ptrs::sigma_vb || d.member == &Ptrs::UVFLX_dif_coef) &&
            !h->h_dsigma.empty() && !h->h_sigma_vb.empty() && !h->h_uvdif.empty()) {
            // the momentum kernel's per-level constants (Ptrs::mom_c); plain IEEE operations, no contraction possible
            const int nz = g.nz, cs = nz + 2;
            std::vector<real> m(5 * (size_t)cs, 0.0);
            for (int t = 0; t <= nz; t++) {
                m[3 * cs + t] = h->h_sigma_vb[t];
                if (t < nz) {
                    const real ds = h->h_dsigma[t];
                    const real rsum = t > 0 ? 1.0 / (ds + h->h_dsigma[t - 1]) : 0.0;
                    m[t] = ds * rsum;
                    m[cs + t] = t > 0 ? h->h_dsigma[t - 1] * rsum : 0.0;
                    m[2 * cs + t] = 1.0 / ds;
                    m[4 * cs + t] = h->h_uvdif[t];
                }
            }
            HIP_TRY(hipMemcpyAsync(const_cast<real *>(h->p.mom_c), m.data(), m.size() * sizeof(real), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
    } else {
        const int e0 = g.nx + 2 + d.stgx, e1 = g.ny + 2 + d.stgy;
        if (n0 != e0 || n1 != e1 || n2 != 1)
            return fail(DYCORE_EINVAL, "grid array '%s': expected shape (%d,%d,1), got (%d,%d,%d)", name, e0, e1, n0, n1, n2);
        const size_t n = (size_t)n0 * n1;
        if (d.member == &Ptrs::lat_rad || d.member == &Ptrs::lat_is_rad || d.member == &Ptrs::corf || d.member == &Ptrs::corf_is) {
            // a latitude-longitude grid: these depend on the row only (main_grid.py:132-189, 229-240).  The fused
            // momentum kernel relies on it (a wavefront keeps them in scalar registers); anything else steps on the
            // stage kernels.  Every row the fused kernels read is checked -- they take corf and lat_rad at j-1 = 0 too
            // (a pole's halo row, which the reference leaves NaN: NaN along the whole row counts as constant) --, and
            // the verdict is taken anew from all four arrays whenever one of them is set again.
            bool ok = true;
            for (int j = 0; j < n1 && ok; j++) {
                const real r0 = host[(size_t)1 * n1 + j];
                for (int i = 2; i <= g.nx + d.stgx; i++) {
                    const real v = host[(size_t)i * n1 + j];
                    if (v != r0 && !(v != v && r0 != r0)) { ok = false; break; }
                }
            }
            const int bit = d.member == &Ptrs::lat_rad ? 1 : d.member == &Ptrs::lat_is_rad ? 2 : d.member == &Ptrs::corf ? 4 : 8;
            h->nonzonal = ok ? (h->nonzonal & ~bit) : (h->nonzonal | bit);
            h->zonal = h->nonzonal == 0;
            h->row_c_valid = false;         // Ptrs::row_c is formed from these four (ensure_row_constants)
        }
        HIP_TRY(hipMemcpyAsync(h->stage, host, n * sizeof(real), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_to_device, dim3((n0 + 31) / 32, 1, n1), dim3(32, 8), 0, h->stream, g, h->stage, dev, n0, n1, 1);
        int rc = launch_check("k_to_device");
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    h->grid_set |= 1 << it->second;
    h->planes_valid = false; h->rows_ready = false; h->colp_old_ready = false;
    return 0;
}

static int find_field(dycore_t *h, const char *name, const FieldDesc **d)
{
    if (!h || !name) return fail(DYCORE_EINVAL, "null argument");
    auto it = h->field_idx.find(name);
    if (it == h->field_idx.end()) return fail(DYCORE_EINVAL, "unknown field '%s'", name);
    *d = &kFields[it->second];
    if (!(h->p.*((*d)->member)))
        return fail(DYCORE_ESTATE, "field '%s' exists only on handles created with %s", name, (*d)->rad == 2 ? "i_turb" : "i_radiation");
    return 0;
}

int dycore_has_field(dycore_t *h, const char *name)
{
    if (!h || !name) return 0;
    auto it = h->field_idx.find(name);
    return it != h->field_idx.end() && (h->p.*(kFields[it->second].member)) != nullptr;
}

int dycore_field_shape(dycore_t *h, const char *name, int32_t shape[3])
{
    const FieldDesc *d;
    int rc = find_field(h, name, &d);
    if (rc) return rc;
    shape[0] = h->g.nx + 2 + d->stgx;
    shape[1] = h->g.ny + 2 + d->stgy;
    shape[2] = h->nk_of(d->nk);
    return 0;
}

int dycore_upload(dycore_t *h, const char *name, const real *host)
{
    const FieldDesc *d;
    int rc = find_field(h, name, &d);
    if (rc) return rc;
    if (!host) return fail(DYCORE_EINVAL, "null host pointer");
    if (d->member == &Ptrs::UWIND || d->member == &Ptrs::VWIND) h->winds_bc_dirty = true;
    h->planes_valid = false; h->rows_ready = false; h->colp_old_ready = false;
    if (d->member == &Ptrs::UFLX || d->member == &Ptrs::VFLX || d->member == &Ptrs::FLXDIV || d->member == &Ptrs::PHIVB || d->member == &Ptrs::PVTF || d->member == &Ptrs::PHI || d->member == &Ptrs::POTTVB)
        if ((rc = refresh_lazy(h))) return rc;      // a caller's value must not be overwritten by a later derivation
    const Geo &g = h->g;
    const int n0 = g.nx + 2 + d->stgx, n1 = g.ny + 2 + d->stgy, n2 = h->nk_of(d->nk);
    const size_t n = (size_t)n0 * n1 * n2;
    HIP_TRY(hipMemcpyAsync(h->stage, host, n * sizeof(real), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_to_device, dim3((n0 + 31) / 32, (n2 + 31) / 32, n1), dim3(32, 8), 0, h->stream,
                       g, h->stage, h->p.*(d->member), n0, n1, n2);
    if ((rc = launch_check("k_to_device"))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));   // the staging buffer is reused by the next call
    return 0;
}

// which derivation a read of this field needs (0: the field is always current)
static int lazy_need(const FieldDesc *d)
{
    if (d->member == &Ptrs::UFLX || d->member == &Ptrs::VFLX) return LAZY_FLX;
    if (d->member == &Ptrs::FLXDIV) return LAZY_FLXDIV;
    if (d->member == &Ptrs::PHIVB) return LAZY_PHIVB;
    if (d->member == &Ptrs::PVTF) return LAZY_PVTF;
    if (d->member == &Ptrs::PHI) return LAZY_PHI;
    if (d->member == &Ptrs::POTTVB) return LAZY_POTTVB;
    return 0;
}

int dycore_download(dycore_t *h, const char *name, real *host)
{
    const FieldDesc *d;
    int rc = find_field(h, name, &d);
    if (rc) return rc;
    if (!host) return fail(DYCORE_EINVAL, "null host pointer");
    if ((rc = refresh_lazy(h, lazy_need(d)))) return rc;
    const Geo &g = h->g;
    const int n0 = g.nx + 2 + d->stgx, n1 = g.ny + 2 + d->stgy, n2 = h->nk_of(d->nk);
    const size_t n = (size_t)n0 * n1 * n2;
    hipLaunchKernelGGL(k_to_host, dim3((n0 + 31) / 32, (n2 + 31) / 32, n1), dim3(32, 8), 0, h->stream,
                       g, h->p.*(d->member), h->stage, n0, n1, n2);
    if ((rc = launch_check("k_to_host"))) return rc;
    HIP_TRY(hipMemcpyAsync(host, h->stage, n * sizeof(real), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int dycore_field_ptr(dycore_t *h, const char *name, void **dev, int64_t *pitch, int64_t *rows, int64_t *levels)
{
    const FieldDesc *d;
    int rc = find_field(h, name, &d);
    if (rc) return rc;
    if ((rc = refresh_lazy(h, lazy_need(d)))) return rc;
    h->planes_valid = false; h->rows_ready = false; h->colp_old_ready = false;        // the caller may write through the pointer
    if (dev) *dev = h->p.*(d->member);
    if (pitch) *pitch = h->g.P;
    if (rows) *rows = h->g.NJ;
    if (levels) *levels = h->nk_of(d->nk);
    return 0;
}

int dycore_copy_field(dycore_t *h, const char *dst, const char *src)
{
    const FieldDesc *a, *b;
    int rc;
    if ((rc = find_field(h, dst, &a)) || (rc = find_field(h, src, &b))) return rc;
    if (a->stgx != b->stgx || a->stgy != b->stgy || a->nk != b->nk)
        return fail(DYCORE_EINVAL, "copy_field: '%s' and '%s' differ in shape", dst, src);
    h->planes_valid = false; h->rows_ready = false; h->colp_old_ready = false;
    if ((rc = refresh_lazy(h, lazy_need(a) | lazy_need(b)))) return rc;
    return do_copy(h, h->p.*(a->member), h->p.*(b->member), h->field_elems(a->nk));
}

int dycore_exchange_bc(dycore_t *h, const char *name)
{
    const FieldDesc *d;
    int rc = find_field(h, name, &d);
    if (rc) return rc;
    return do_exchange(h, h->p.*(d->member), d->stgx, d->stgy, h->nk_of(d->nk));
}

int dycore_continuity(dycore_t *h, double dt) { int rc = ready_join(h); return rc ? rc : do_continuity(h, h->p, dt); }
int dycore_momentum(dycore_t *h) { int rc = ready_join(h); return rc ? rc : do_momentum(h, h->p); }
int dycore_momentum_prepare(dycore_t *h) { int rc = ready_join(h); return rc ? rc : do_prepare(h, h->p); }
int dycore_uflx_tendency(dycore_t *h) { int rc = ready_join(h); return rc ? rc : do_uflx(h, h->p); }
int dycore_vflx_tendency(dycore_t *h) { int rc = ready_join(h); return rc ? rc : do_vflx(h, h->p); }
int dycore_pott_tendency(dycore_t *h) { int rc = ready_join(h); return rc ? rc : do_pott(h, h->p); }
int dycore_moist_tendency(dycore_t *h)
{
    int rc = ready_join(h);
    if (rc) return rc;
    if (!h->g.i_moist) return fail(DYCORE_ESTATE, "handle created without i_moist");
    return do_moist(h, h->p, nullptr, nullptr, nullptr, nullptr, 0.0);
}
int dycore_euler_forward(dycore_t *h, double dt) { int rc = ready_join(h); return rc ? rc : do_euler(h, h->p, dt); }
int dycore_primary_diag(dycore_t *h) { int rc = ready_join(h); return rc ? rc : do_diag(h, h->p); }
int dycore_secondary_diag_rho(dycore_t *h) { int rc = ready_join(h, LAZY_PVTF | LAZY_POTTVB); return rc ? rc : do_rho(h, h->p); }
int dycore_secondary_diag(dycore_t *h) { int rc = ready_join(h, LAZY_PVTF | LAZY_POTTVB); return rc ? rc : do_secondary(h, h->p); }
int dycore_compute_turbulence(dycore_t *h) { int rc = ready_join(h, LAZY_PHIVB | LAZY_PHI | LAZY_POTTVB); return rc ? rc : do_turbulence(h, h->p); }

int dycore_radiation(dycore_t *h, double solar_constant, int32_t nbins, const double *lambdas_center, const double *dlambdas)
{
    if (!h) return fail(DYCORE_EINVAL, "null handle");
    if (!h->g.i_rad) return fail(DYCORE_ESTATE, "handle created without i_radiation");
    int rc = ready_join(h, LAZY_PHIVB | LAZY_PHI);
    if (rc) return rc;
    if (!lambdas_center || !dlambdas || nbins < 1 || nbins > RAD_MAX_BINS)
        return fail(DYCORE_EINVAL, "need 1..%d spectral bins", RAD_MAX_BINS);
    if (h->g.nz < 3) return fail(DYCORE_EINVAL, "the shortwave scheme needs nz >= 3 (rad_shortwave.py:99-100)");
    RadParams rp;
    rp.solar_constant = solar_constant;
    rp.nbins = nbins;
    for (int b = 0; b < RAD_MAX_BINS; b++) { rp.lam[b] = b < nbins ? lambdas_center[b] : 1.0; rp.dlam[b] = b < nbins ? dlambdas[b] : 0.0; }
    Scope s(h, K_RAD);
    const long long cols = (long long)h->g.nx * h->g.ny;
    for (long long c0 = 0; c0 < cols; c0 += h->rad_ws_cols) {
        const long long n = cols - c0 < h->rad_ws_cols ? cols - c0 : h->rad_ws_cols;
        hipLaunchKernelGGL(k_radiation, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->g, h->p, rp,
                           h->rad_ws, h->rad_ws_cols, c0, n);
        if ((rc = launch_check("k_radiation"))) return rc;
    }
    return 0;
}

int dycore_step_matsuno(dycore_t *h, double dt, int32_t nsteps, int32_t with_rho)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (nsteps < 0) return fail(DYCORE_EINVAL, "nsteps < 0");
    for (int s = 0; s < nsteps; s++)
        if ((rc = do_step(h, dt, with_rho))) return rc;
    return 0;
}


int dycore_step_begin(dycore_t *h, int32_t with_rho)
{
    int rc = check_ready(h);
    return rc ? rc : step_begin(h, with_rho);
}

int dycore_substep_part(dycore_t *h, double dt, int32_t substep, int32_t part)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (substep < 0 || substep > 1) return fail(DYCORE_EINVAL, "substep must be 0 (estimate) or 1 (final)");
    return substep_part(h, dt, substep, part);
}

const char *dycore_substep_field(dycore_t *h, int32_t substep, const char *name)
{
    if (!h || !name) return nullptr;
    if (fused_path(h) && substep == 0) {
        if (!strcmp(name, "UWIND")) return "UWIND_EST";
        if (!strcmp(name, "VWIND")) return "VWIND_EST";
        if (!strcmp(name, "POTT")) return "POTT_EST";
        if (!strcmp(name, "QV")) return "QV_EST";
        if (!strcmp(name, "QC")) return "QC_EST";
    }
    return name;
}

static int rows_op(dycore_t *h, int32_t n, const char *const *names, const int32_t *rows,
                   void *dev_buf, int64_t *n_doubles, int unpack)
{
    if (!h || n < 0 || (n && (!names || !rows))) return fail(DYCORE_EINVAL, "bad argument");
    const Ptrs &q = h->p;
    long long off = 0;
    int done = 0;
    while (done < n) {
        RowList rl;
        rl.n = 0;
        int maxk = 1;
        for (; done < n && rl.n < ROWLIST_MAX; done++) {
            const FieldDesc *d;
            int rc = find_field(h, names[done], &d);
            if (rc) return rc;
            if (rows[done] < 0 || rows[done] > h->g.ny + 1 + d->stgy)
                return fail(DYCORE_EINVAL, "row %d outside field '%s'", rows[done], names[done]);
            const int e = rl.n++;
            rl.base[e] = q.*(d->member);
            if (unpack && dev_buf && (d->member == &Ptrs::COLP || d->member == &Ptrs::COLP_OLD)) h->colp_old_ready = false;
            rl.row[e] = rows[done];
            rl.nk[e] = h->nk_of(d->nk);
            rl.k0[e] = 0; rl.kstep[e] = 1;
            rl.off[e] = off;
            off += (long long)rl.nk[e] * (h->g.nx + 3);
            if (rl.nk[e] > maxk) maxk = rl.nk[e];
        }
        if (dev_buf && rl.n) {
            Scope s(h, K_HALO);
            hipLaunchKernelGGL(k_rows, dim3((h->g.nx + 3 + 255) / 256, maxk, rl.n), dim3(256), 0, h->stream,
                               h->g, rl, (real *)dev_buf, unpack);
            int rc = launch_check("k_rows");
            if (rc) return rc;
        }
    }
    if (n_doubles) *n_doubles = off;
    return 0;
}

int dycore_rows_size(dycore_t *h, int32_t n, const char *const *names, int64_t *n_doubles)
{
    std::vector<int32_t> zero(n > 0 ? n : 0, 0);
    return rows_op(h, n, names, zero.data(), nullptr, n_doubles, 0);
}
int dycore_pack_rows(dycore_t *h, int32_t n, const char *const *names, const int32_t *rows, void *dev_buf)
{
    if (!dev_buf) return fail(DYCORE_EINVAL, "null buffer");
    return rows_op(h, n, names, rows, dev_buf, nullptr, 0);
}
int dycore_unpack_rows(dycore_t *h, int32_t n, const char *const *names, const int32_t *rows, const void *dev_buf)
{
    if (!dev_buf) return fail(DYCORE_EINVAL, "null buffer");
    return rows_op(h, n, names, rows, const_cast<void *>(dev_buf), nullptr, 1);
}

int dycore_time_steps(dycore_t *h, double dt, int32_t nsteps, int32_t with_rho, double *ms)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!ms) return fail(DYCORE_EINVAL, "null ms");
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, h->stream));
    for (int s = 0; s < nsteps && !rc; s++) rc = do_step(h, dt, with_rho);
    HIP_TRY(hipEventRecord(b, h->stream));
    HIP_TRY(hipEventSynchronize(b));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms = f;
    return rc;
}

int dycore_profile(dycore_t *h, int32_t enable)
{
    if (!h) return fail(DYCORE_EINVAL, "null handle");
    int rc = prof_flush(h);
    h->prof = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    if (enable) { memset(h->launches, 0, sizeof h->launches); memset(h->total_ms, 0, sizeof h->total_ms); }
    return rc;
}

int dycore_n_kernels(void) { return K_COUNT; }

int dycore_kernel_stats(dycore_t *h, int32_t idx, const char **name, int64_t *launches, double *total_ms)
{
    if (!h || idx < 0 || idx >= K_COUNT) return fail(DYCORE_EINVAL, "bad kernel index");
    int rc = prof_flush(h);
    if (rc) return rc;
    if (name) *name = kKernelNames[idx];
    if (launches) *launches = h->launches[idx];
    if (total_ms) *total_ms = h->total_ms[idx];
    return 0;
}

}  // extern "C"

#include "dycore_bands.h"
