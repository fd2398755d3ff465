/*
 * dycore.h -- C ABI of libdycore_hip.so, the MI355X (gfx950) implementation of the
 * hydrostatic dynamical-core Matsuno time step.
 *
 * This is the drop-in boundary for the hot path of Potopoles/Climate_Model.  The
 * reference has no FFI of its own: the path sits behind the three factory classes
 * of dyn_org_discretizations.py, which dispatch to numba CPU/CUDA kernels.  Each
 * entry point below names the reference interface it replaces (file:line relative
 * to the reference tree).  The Python classes in climate_model_amd/ bind these
 * symbols with ctypes and reproduce the factories' method names and `fields_*`
 * lists; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C: opaque handle, POD config struct, raw pointers and sizes; every
 *    function returns 0 on success or a DYCORE_E* code and never throws;
 *    dycore_last_error() gives the message for the calling thread's last failure.
 *  - host arrays cross the boundary in the reference's layout: C-order [i][j][k],
 *    k contiguous, halo nb = 1, shapes as in main_fields.py:477-485
 *    (nx+2+stgx, ny+2+stgy, nz | nz+1 | 1).  Device storage is [k][j][i] with a
 *    padded i-pitch; dycore_upload/_download convert on the device.
 *  - fields are addressed by the reference's names ("UWIND", "COLP_NEW", ...).
 *  - all work of one handle is issued on one HIP stream; calls return without
 *    synchronising unless documented (download, sync).
 *  - one handle per GPU / per latitude band; a handle is not thread-safe, distinct
 *    handles are independent.
 */
#ifndef DYCORE_H
#define DYCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYCORE_ABI_VERSION 1

enum {
    DYCORE_OK = 0,
    DYCORE_EINVAL = 1,     /* bad argument / unknown field name / shape mismatch */
    DYCORE_EHIP = 2,       /* a HIP runtime call failed                           */
    DYCORE_ENOMEM = 3,
    DYCORE_ESTATE = 4      /* call not valid in the handle's current state       */
};

typedef struct dycore dycore_t;

/* Working precision of the fields (namelist.py:225-226 working_precision): one build of the library per precision --
 * libdycore_hip.so (double, the reference's default and the parity mode) and libdycore_hip_f32.so (built with
 * -DDYCORE_FLOAT32).  Host arrays cross the boundary in that type; scalar arguments are double in both.
 * dycore_real_bytes() tells a binding which one it has loaded. */
#ifdef DYCORE_FLOAT32
typedef float dycore_real_t;
#else
typedef double dycore_real_t;
#endif
int dycore_real_bytes(void);

/* Per-term dynamics switches (namelist.py:62-88), consumed where the reference consumes them: dyn_continuity.py:211,
 * dyn_UVFLX_prepare.py:260-356 and dyn_org_discretizations.py:144, dyn_UFLX.py:117-193, dyn_VFLX.py:115-192,
 * dyn_POTT.py:72-108, dyn_moist.py:68-115.  A set bit means "that namelist entry is 0".  The vertical-turbulence entries
 * travel as DycoreConfig.i_turb, i_POTT_radiation as i_radiation, i_moist_main_switch as i_moist. */
#define DYCORE_TERM_COLP_MAIN       (1 << 0)
#define DYCORE_TERM_UVFLX_MAIN      (1 << 1)
#define DYCORE_TERM_UVFLX_HOR_ADV   (1 << 2)
#define DYCORE_TERM_UVFLX_VERT_ADV  (1 << 3)
#define DYCORE_TERM_UVFLX_CORIOLIS  (1 << 4)
#define DYCORE_TERM_UVFLX_PRE_GRAD  (1 << 5)
#define DYCORE_TERM_UVFLX_NUM_DIF   (1 << 6)
#define DYCORE_TERM_POTT_MAIN       (1 << 7)
#define DYCORE_TERM_POTT_HOR_ADV    (1 << 8)
#define DYCORE_TERM_POTT_VERT_ADV   (1 << 9)
#define DYCORE_TERM_POTT_NUM_DIF    (1 << 10)
#define DYCORE_TERM_MOIST_HOR_ADV   (1 << 11)
#define DYCORE_TERM_MOIST_VERT_ADV  (1 << 12)
#define DYCORE_TERM_MOIST_NUM_DIF   (1 << 13)
#define DYCORE_TERM_ALL             ((1 << 14) - 1)

/* Replaces the module-level globals the reference freezes into its kernels:
 * main_grid.py:42-51 (extents), namelist.py:62-88 (switches), pair_top. */
typedef struct DycoreConfig {
    int32_t abi_version;   /* DYCORE_ABI_VERSION                                        */
    int32_t nx;            /* number of longitudes (periodic)                           */
    int32_t ny;            /* number of latitude rows held by THIS handle (band height) */
    int32_t nz;            /* number of sigma levels                                    */
    int32_t ny_global;     /* rows of the whole grid; == ny for a single-GPU run        */
    int32_t j_offset;      /* global row of local row 1, minus 1 (0 for the first band) */
    int32_t device;        /* HIP device ordinal                                        */
    int32_t i_turb;        /* i_UVFLX_vert_turb & i_POTT_vert_turb (namelist.py:66,77)  */
    int32_t i_radiation;   /* i_POTT_radiation (io_read_namelist.py:66-67)              */
    int32_t i_moist;       /* i_moist_main_switch (namelist.py:80): QV/QC advection, dyn_moist */
    int32_t i_moist_clip;  /* != 0: QV, QC < 0 -> 0 after every Euler step, the rule of the reference's accelerator
                            * kernel (dyn_timestep.py:145-154); 0 = its CPU kernel (:291-296), which does not clip
                            * and which the parity fixtures follow */
    int32_t term_off;      /* bit mask of DYCORE_TERM_*: tendency terms switched OFF (namelist.py:62-88: the i_*_main_switch /
                            * _hor_adv / _vert_adv / _coriolis / _pre_grad / _num_dif entries set to 0).  0 = the reference's
                            * default, every term on.  A handle with any bit set steps on the stage kernels. */
    int32_t reserved[4];
    double  pair_top;      /* model-top pressure [Pa] (namelist.py:37)                  */
    void   *stream;        /* hipStream_t to issue work on; NULL = handle creates one   */
} DycoreConfig;

/* ---- lifetime ------------------------------------------------------------------- */
int dycore_create(const DycoreConfig *cfg, dycore_t **out);
int dycore_destroy(dycore_t *h);
const char *dycore_last_error(void);
int dycore_abi_version(void);
/* blocks until all work issued on the handle's stream has finished */
int dycore_sync(dycore_t *h);

/* ---- grid constants: main_grid.py:299-314 (GRF dict) ------------------------------ */
/* name in {A,dxjs,dyis,corf,corf_is,lat_rad,lat_is_rad,dlon_rad,dlat_rad} with host
 * shape (n0,n1,1), or {sigma_vb,dsigma,UVFLX_dif_coef,POTT_dif_coef,moist_dif_coef} with
 * n0*n1*n2 = nz+1 / nz elements (moist_dif_coef is required only with i_moist). */
int dycore_set_grid(dycore_t *h, const char *name, const dycore_real_t *host,
                    int32_t n0, int32_t n1, int32_t n2);

/* ---- fields: main_fields.py:184-215 (copy_host_to_device / copy_device_to_host) ---- */
/* 1 if the handle holds a buffer for `name` (the radiation fields exist only with i_radiation, the
 * TURB_* scratch fields only with i_turb), else 0: lets main_fields.ModelFields build its ALL_FIELDS
 * group (main_fields.py:150-182) from what can actually be transferred */
int dycore_has_field(dycore_t *h, const char *name);
int dycore_field_shape(dycore_t *h, const char *name, int32_t shape[3]);
int dycore_upload(dycore_t *h, const char *name, const dycore_real_t *host);
int dycore_download(dycore_t *h, const char *name, dycore_real_t *host);   /* synchronises */
/* raw device view for callers that keep data resident (torch, other kernels):
 * element (i,j,k) is dev[(k*rows + j)*pitch + i].
 * PVTFVB and PHI of a single-GPU handle are double-buffered by the whole-substep kernel (one swap per substep): a pointer
 * handed out here is that field's again after every WHOLE step (dycore_step_matsuno), not between the two substeps of
 * one (dycore_substep_part); ask again there.  The derived fields (UFLX, VFLX, FLXDIV, PVTF, PHIVB) are brought up to date
 * by this call. */
int dycore_field_ptr(dycore_t *h, const char *name, void **dev,
                     int64_t *pitch, int64_t *rows, int64_t *levels);
/* dst <- src, whole field: misc_gpu_functions.py:35-40 (set_equal) and the slice
 * copies of dyn_matsuno.py:34-49,64-67 */
int dycore_copy_field(dycore_t *h, const char *dst, const char *src);
/* lateral boundary conditions of one field: misc_boundaries.py:22-82 */
int dycore_exchange_bc(dycore_t *h, const char *name);

/* ---- stages: one entry per factory method of dyn_org_discretizations.py ----------- */
/* TendencyFactory.continuity  (:89-117)  incl. exchange of UFLX,VFLX,WWIND,COLP_NEW  */
int dycore_continuity(dycore_t *h, double dt);
/* TendencyFactory.momentum    (:121-249) = prepare + UFLX + VFLX tendencies           */
int dycore_momentum(dycore_t *h);
/*   the three kernels of .momentum individually (dyn_UVFLX_prepare.py:249,
 *   dyn_UFLX.py:339, dyn_VFLX.py:322) */
int dycore_momentum_prepare(dycore_t *h);
int dycore_uflx_tendency(dycore_t *h);
int dycore_vflx_tendency(dycore_t *h);
/* TendencyFactory.temperature (:253-273) */
int dycore_pott_tendency(dycore_t *h);
/* TendencyFactory.moisture (:277-294): dQVdt, dQCdt, dQVdt_TURB (handles created with i_moist) */
int dycore_moist_tendency(dycore_t *h);
/* PrognosticsFactory.euler_forward (:359-393) incl. exchange of POTT,VWIND,UWIND      */
int dycore_euler_forward(dycore_t *h, double dt);
/* DiagnosticsFactory.primary_diag (:310-326): PVTF, PVTFVB, PHI, PHIVB, POTTVB         */
int dycore_primary_diag(dycore_t *h);
/* DiagnosticsFactory.secondary_diag (:329-347), the part the dycore consumes: RHO, RHOVB */
int dycore_secondary_diag_rho(dycore_t *h);
/* DiagnosticsFactory.secondary_diag (:329-347), every output: TAIR, TAIRVB, PAIR, PAIRVB, RHO,
 * RHOVB, WINDX, WINDY, WIND (dyn_diagnostics.py:199-222) */
int dycore_secondary_diag(dycore_t *h);
/* Turbulence.compute_turbulence (turb_main.py:36-48, turb_compute.py:191-208): KMOM, KHEAT from
 * PHIVB, HSURF, PHI, QV, WINDX, WINDY, POTTVB, POTT */
int dycore_compute_turbulence(dycore_t *h);
/* rad_main.py:121-205 (Radiation.simple_radiation) for every owned column: longwave and, where MYSUN > 0,
 * shortwave two-stream fluxes, net fluxes, flux divergences and the heating rate dPOTTdt_RAD.  Reads
 * PHIVB, TAIR, RHO, QC, SOILTEMP (top soil level), SURFALBEDLW, SURFALBEDSW, MYSUN, SWINTOA; the caller
 * supplies the spectral bins of rad_main.py:50-65 and the current solar constant
 * (rad_shortwave.calc_current_solar_constant).  Needs i_radiation at dycore_create and nz >= 3. */
int dycore_radiation(dycore_t *h, double solar_constant, int32_t nbins, const double *lambdas_center,
                     const double *dlambdas);

/* ---- the whole hot loop body: dyn_matsuno.py:28-131, nsteps times ------------------- */
/* with_rho != 0 re-diagnoses RHO/RHOVB before each step (solver.py:99-101); needed only
 * when i_turb is set. */
int dycore_step_matsuno(dycore_t *h, double dt, int32_t nsteps, int32_t with_rho);

/* ---- latitude-band decomposition (multi-GPU; new functionality, the reference is
 * single-device: SURVEY.md 5 and 8e) -------------------------------------------------
 * A band handle (ny < ny_global) runs the step in parts and refreshes its north/south halo
 * rows from the neighbouring bands in between; the transport (RCCL send/recv through
 * torch.distributed) lives in climate_model_amd/bands.py, which owns the message buffers.
 *   dycore_step_begin          rotate the time levels (the *_OLD copies of dyn_matsuno.py:34-49)
 *   dycore_substep_part(s, 0)  continuity            -> exchange UFLX, VFLX, WWIND, COLP_NEW rows
 *   dycore_substep_part(s, 1)  momentum, temperature, COLP<-COLP_NEW, Euler forward
 *                                                     -> exchange UWIND, VWIND, POTT rows
 *   dycore_substep_part(s, 2)  primary diagnostics (halo rows included)
 * for s = 0 (estimate) and 1 (final).  To overlap the exchanges with compute, part 1 of the fused
 * step (dycore_overlap_parts(h) != 0) can be issued in three pieces:
 *   DYCORE_PART_MOM_INNER  momentum tiles whose stencils stay clear of the halo rows: may run while
 *                          the exchange after part 0 is in flight
 *   DYCORE_PART_MOM_EDGE   the remaining momentum tiles (need those rows and the diagnostics of the
 *                          previous part 2)      -> the new UWIND, VWIND rows can be exchanged
 *   DYCORE_PART_TEMP       temperature (+ moisture), COLP<-COLP_NEW: does not touch the winds, so the
 *                          wind rows travel meanwhile -> exchange POTT (QV, QC) rows, then part 2
 * On a turbulence-coupled handle the two momentum pieces evaluate the turbulent tendencies of the same rows first, and
 * dycore_step_begin(with_rho) needs the POTT halo rows of the state the step starts from (exchange them at the end of
 * every step, not only behind the last one).
 * Part 0 of the next substep needs the wind rows only.
 * dycore_use_stream redirects every subsequent launch of the handle to another HIP stream (NULL:
 * back to the handle's own stream) so that pack/unpack and part 2 can be queued on the stream the
 * transport runs on; ordering between the streams is the caller's (events).
 * pack/unpack move whole rows (columns 0..nx+2, all
 * levels) of the named fields between the field buffers and one contiguous device buffer,
 * entry e at offset sum_{e'<e} levels(e')*(nx+3) doubles; packed columns 0, nx+1, nx+2 are the periodic
 * images of columns nx, 1, 2 of the same row. */
int dycore_step_begin(dycore_t *h, int32_t with_rho);
/* name of the registered field that holds `name` (UWIND, VWIND or POTT) right after part 1 of
 * the given substep: the fused step writes the predictor state into *_EST scratch fields */
const char *dycore_substep_field(dycore_t *h, int32_t substep, const char *name);
enum { DYCORE_PART_MOM_INNER = 10, DYCORE_PART_MOM_EDGE = 11, DYCORE_PART_TEMP = 12,
       /* MOM_INNER in two halves (first / second half of its tile rows), for a schedule that puts the edge piece between
        * them: INNER_A - wait for the rows - EDGE - send the new wind rows - INNER_B.  On a handle whose momentum and
        * temperature kernels are one launch (the whole-substep kernel; round 3: latitude bands too) part TEMP no longer
        * holds a kernel that could hide the wind rows' journey -- INNER_B does.  Other handles: INNER_A = MOM_INNER,
        * INNER_B = nothing. */
       DYCORE_PART_MOM_INNER_A = 13, DYCORE_PART_MOM_INNER_B = 14 };
int dycore_overlap_parts(dycore_t *h);
/* != 0: the fused step does not store UFLX / VFLX (its kernels re-form them from the winds and the
 * 2-D column pressures UCOL, VCOL at the u- / v-points; dycore_download still returns them, derived on
 * demand).  A band then exchanges rows of UCOL, VCOL (2-D) after part 0 instead of UFLX, VFLX, and row 2
 * of VWIND with the winds. */
int dycore_flux_from_winds(dycore_t *h);
int dycore_substep_part(dycore_t *h, double dt, int32_t substep, int32_t part);
/* Comma-separated names of the diagnostic fields whose halo rows travel with the POTT rows of a band exchange (phase B):
 * a band never re-diagnoses rows it does not own, it receives their owner's values (PVTFVB and, depending on what the
 * handle's step stores, PHIVB / PHI / POTTVB / PVTF).  Empty for a handle on the staged step, whose part 2 diagnoses the
 * whole band including its halo rows (dyn_org_discretizations.py:321-326). */
int dycore_band_row_fields(dycore_t *h, char *buf, int32_t nbytes);
int dycore_use_stream(dycore_t *h, void *stream);
/* the handle's own stream (DycoreConfig.stream, or the one the library created) */
void *dycore_get_stream(dycore_t *h);
int dycore_rows_size(dycore_t *h, int32_t n, const char *const *names, int64_t *n_doubles);
int dycore_pack_rows(dycore_t *h, int32_t n, const char *const *names, const int32_t *rows,
                     void *dev_buf);
int dycore_unpack_rows(dycore_t *h, int32_t n, const char *const *names, const int32_t *rows,
                       const void *dev_buf);

/* ---- the band step inside the library ------------------------------------------------
 * The schedule above (continuity, exchange AB2 behind the inner momentum tiles, edge tiles, exchange B1 behind the
 * temperature kernel, ...) issued by ONE call per step sequence, with the transport inside the library:
 *   dycore_bands_create_rccl   one band per process / GPU; neighbour exchange = ncclSend / ncclRecv (RCCL over xGMI)
 *                              on the library's comm stream.  `id` = the 128 bytes rank 0 obtained from
 *                              dycore_bands_unique_id and passed to every rank by whatever bootstrap the host has
 *                              (torch.distributed, MPI, a file); south_rank / north_rank = the ranks holding the
 *                              adjacent bands, -1 on the side where this band touches a pole.  librccl.so is
 *                              resolved at run time (dlopen), so a single-GPU host does not need it.
 *   dycore_bands_create_local  all bands are handles of THIS process on one stream (south to north); exchange =
 *                              device-to-device copies.  Rehearsal of the same schedule on one GPU.
 *   dycore_bands_connect       one transport of the send buffers, nothing unpacked (peer connections up before a
 *                              timed region); every rank calls it.
 *   dycore_bands_step_matsuno  nsteps Matsuno steps; returns with every halo row of the final state in place and
 *                              the comm stream joined into the handles' stream.
 * Needs handles whose step is the fused one with split parts (dycore_overlap_parts, dycore_flux_from_winds != 0): dry,
 * moist or turbulence-coupled.
 * An N-band run equals the single-domain run bit for bit. */
typedef struct dycore_bands dycore_bands_t;
int dycore_bands_unique_id(void *id, int32_t nbytes);                 /* nbytes = 128 */
int dycore_bands_create_rccl(dycore_t *h, const void *id, int32_t nbytes, int32_t rank, int32_t nranks,
                             int32_t south_rank, int32_t north_rank, dycore_bands_t **out);
int dycore_bands_create_local(dycore_t *const *handles, int32_t n, dycore_bands_t **out);
int dycore_bands_connect(dycore_bands_t *b);
int dycore_bands_step_matsuno(dycore_bands_t *b, double dt, int32_t nsteps, int32_t with_rho);
int dycore_bands_destroy(dycore_bands_t *b);

/* ---- measurement ------------------------------------------------------------------- */
/* Runs nsteps Matsuno steps bracketed by HIP events on the handle's stream and returns
 * the elapsed device time in milliseconds (synchronises). */
int dycore_time_steps(dycore_t *h, double dt, int32_t nsteps, int32_t with_rho, double *ms);
/* Per-kernel device time accumulated by HIP events since the last reset, for the kernel
 * with the given index (0 <= idx < dycore_n_kernels()); enable with dycore_profile(h,1).
 * dycore_profile(h,2) brackets only the dominant kernel (momentum): two events per launch are
 * a measurable share of a small latitude band's step, so timed multi-GPU runs use this level. */
int dycore_profile(dycore_t *h, int32_t enable);
int dycore_n_kernels(void);
int dycore_kernel_stats(dycore_t *h, int32_t idx, const char **name,
                        int64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* DYCORE_H */
