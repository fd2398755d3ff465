/*
 * dycore_oracle.h -- CPU restatement of the reference's dry/turbulence-coupled
 * hydrostatic dynamical-core Matsuno step.   TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (climate_model_amd + libdycore_hip.so) never
 * calls into it and has no CPU fallback.
 *
 * Parity status: PINNED by the .npz files under tests/golden/, which oracle/gen_golden.py
 * produced in the build container by executing the reference's own Python
 * functions (see that script's header for the import stand-ins it needs because
 * numba is absent).  The reference repository itself holds no known-answer
 * tests for this path.
 *
 * Layout mirrors the reference: C-order [i][j][k], k contiguous
 * (main_fields.py:477-485), halo nb = 1.
 *   unstaggered (nx+2, ny+2, nk)   x-staggered (nx+3, ny+2, nk)
 *   y-staggered (nx+2, ny+3, nk)   xy-staggered (nx+3, ny+3, nk)
 *   nk = nz, nz+1 or 1.  Vertical 1-D fields are plain arrays of nz / nz+1.
 */
#ifndef DYCORE_ORACLE_H
#define DYCORE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Working precision (namelist.py:225-226): libdycore_oracle.so is built with real = double (the parity mode),
 * libdycore_oracle_f32.so with -DORACLE_FLOAT32 (real = float: fields and grid constants; scalars stay double). */
#ifdef ORACLE_FLOAT32
typedef float real;
#else
typedef double real;
#endif

typedef struct {
    int nx, ny, nz;
    int i_turb;        /* i_UVFLX_vert_turb / i_POTT_vert_turb (namelist.py:66,77) */
    int i_radiation;   /* i_POTT_radiation after io_read_namelist.py:66-67         */
    int nthreads;      /* OpenMP threads, 0 = runtime default                      */
    double pair_top;   /* namelist.py pair_top                                     */
    int i_moist;       /* i_moist_main_switch (namelist.py:80): QV/QC advection     */
    int i_moist_clip;  /* QV, QC < 0 -> 0 after the Euler step: the rule of the reference's GPU kernel
                          (dyn_timestep.py:145-154); its CPU kernel (:291-296) does not clip       */
    int term_off;      /* ORACLE_TERM_* bits: namelist.py:62-88 entries set to 0 (same bits as DYCORE_TERM_* of
                          include/dycore.h, restated here: the oracle shares no header with the product)  */
} oracle_cfg;

enum {
    ORACLE_TERM_COLP_MAIN = 1 << 0,
    ORACLE_TERM_UVFLX_MAIN = 1 << 1, ORACLE_TERM_UVFLX_HOR_ADV = 1 << 2, ORACLE_TERM_UVFLX_VERT_ADV = 1 << 3,
    ORACLE_TERM_UVFLX_CORIOLIS = 1 << 4, ORACLE_TERM_UVFLX_PRE_GRAD = 1 << 5, ORACLE_TERM_UVFLX_NUM_DIF = 1 << 6,
    ORACLE_TERM_POTT_MAIN = 1 << 7, ORACLE_TERM_POTT_HOR_ADV = 1 << 8, ORACLE_TERM_POTT_VERT_ADV = 1 << 9,
    ORACLE_TERM_POTT_NUM_DIF = 1 << 10,
    ORACLE_TERM_MOIST_HOR_ADV = 1 << 11, ORACLE_TERM_MOIST_VERT_ADV = 1 << 12, ORACLE_TERM_MOIST_NUM_DIF = 1 << 13
};

/* grid constants: main_grid.py:299-314 (GRF dict) */
typedef struct {
    const real *A, *dxjs, *dyis, *corf, *corf_is, *lat_rad, *lat_is_rad,
                 *dlon_rad, *dlat_rad;                 /* 2-D (.., .., 1) */
    const real *sigma_vb, *dsigma, *UVFLX_dif_coef, *POTT_dif_coef; /* 1-D */
    const real *moist_dif_coef;                                     /* 1-D */
} oracle_grid;

/* model fields touched by the hot path: main_fields.py:234-475 */
typedef struct {
    real *COLP, *COLP_OLD, *COLP_NEW, *dCOLPdt, *HSURF;
    real *UWIND, *UWIND_OLD, *VWIND, *VWIND_OLD, *WWIND;
    real *POTT, *POTT_OLD, *POTTVB;
    real *UFLX, *VFLX, *FLXDIV;
    real *BFLX, *CFLX, *DFLX, *EFLX, *RFLX, *QFLX, *SFLX, *TFLX;
    real *WWIND_UWIND, *WWIND_VWIND, *KMOM_dUWINDdz, *KMOM_dVWINDdz;
    real *dUFLXdt, *dVFLXdt, *dPOTTdt;
    real *dUFLXdt_TURB, *dVFLXdt_TURB, *dPOTTdt_TURB;
    real *PVTF, *PVTFVB, *PHI, *PHIVB;
    real *RHO, *RHOVB, *KMOM, *KHEAT;
    real *SSHFLX, *SMOMXFLX, *SMOMYFLX, *dPOTTdt_RAD;
    /* moisture (dyn_moist.py) */
    real *QV, *QV_OLD, *QC, *QC_OLD, *dQVdt, *dQCdt, *dQVdt_TURB, *SLHFLX;
    /* full secondary diagnostics (dyn_diagnostics.py:199-222) */
    real *TAIR, *TAIRVB, *PAIR, *PAIRVB, *WINDX, *WINDY, *WIND;
} oracle_fields;

/* misc_boundaries.py:22-42 ; stgx/stgy = 1 if staggered in that direction */
void oracle_exchange_bc(const oracle_cfg *c, real *F, int stgx, int stgy, int nk);

/* dyn_continuity.py:170-232 (kernel only, no exchange) */
void oracle_continuity(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt);
/* dyn_UVFLX_prepare.py:249-439 */
void oracle_uvflx_prepare(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
/* dyn_UFLX.py:339-434 */
void oracle_uflx_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
/* dyn_VFLX.py:322-408 */
void oracle_vflx_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
/* dyn_POTT.py:180-216 */
void oracle_pott_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
/* dyn_moist.py:201-243 (loop), :49-129 (terms) */
void oracle_moist_tendency(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
/* dyn_timestep.py:212-297 (UWIND, VWIND, POTT; QV, QC when i_moist) */
void oracle_make_timestep(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt);
/* dyn_diagnostics.py:139-195 */
void oracle_diag_pvtf(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
void oracle_diag_phi(const oracle_cfg *c, oracle_fields *f);
void oracle_diag_pottvb(const oracle_cfg *c, oracle_fields *f);
/* dyn_diagnostics.py:199-222, RHO and RHOVB only (the two the dycore consumes) */
void oracle_diag_rho(const oracle_cfg *c, oracle_fields *f);

/* dyn_diagnostics.py:199-222, every output */
void oracle_diag_secondary(const oracle_cfg *c, oracle_fields *f);
/* turb_compute.py:191-208 (loop), :54-142 (formulae): KMOM, KHEAT at interfaces 1..nz-1 */
void oracle_compute_turbulence(const oracle_cfg *c, oracle_fields *f);

/* factory-level stages: dyn_org_discretizations.py:89-117, 121-249, 253-273, 359-393, 310-326 */
void oracle_stage_continuity(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt);
void oracle_stage_momentum(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
void oracle_stage_temperature(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
void oracle_stage_moisture(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);
void oracle_stage_euler_forward(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f, double dt);
void oracle_stage_primary_diag(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f);

/* dyn_matsuno.py:28-131, nsteps times; with_rho != 0 re-diagnoses RHO/RHOVB
 * before every step as solver.py:99-101 does */
void oracle_step_matsuno(const oracle_cfg *c, const oracle_grid *g, oracle_fields *f,
                         double dt, int nsteps, int with_rho);

/* radiation columns (radiation_oracle.c): rad_main.py:121-205.  3-D arrays (nx+2, ny+2, nz | nz+1),
 * 2-D ones (nx+2, ny+2, 1); interior columns only. */
void oracle_radiation(int nx, int ny, int nz, const double *PHIVB, const double *TAIR, const double *RHO,
                      const double *SOILTEMP, const double *ALB_LW, const double *ALB_SW, const double *QC,
                      const double *MYSUN, const double *SWINTOA, double solar_constant,
                      int nb, const double *lam, const double *dlam,
                      double *LWFLXDO, double *LWFLXUP, double *SWDIFFLXDO, double *SWDIRFLXDO, double *SWFLXUP,
                      double *SWFLXDO, double *LWFLXNET, double *SWFLXNET, double *SWFLXDIV, double *LWFLXDIV,
                      double *TOTFLXDIV, double *dPOTTdt_RAD);

#ifdef __cplusplus
}
#endif
#endif
