/* A host written in plain C against include/dycore.h (no Python, no torch): builds the metric terms of
 * a 5-degree grid (main_grid.py:166-242), starts an atmosphere at rest over flat terrain (uniform
 * column pressure, potential temperature a function of the level only) and checks that ten Matsuno
 * steps keep it at rest: zero wind, unchanged column pressure.  Links at run time:
 *     gcc -O1 -Iinclude examples/c_client.c -o c_client -ldl -lm
 *     ./c_client climate_model_amd/csrc/libdycore_hip.so
 * Exit code 0 and "C_CLIENT_OK" on success; 3 when the library reports an error (e.g. no GPU). */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "dycore.h"

#define NX 72
#define NY 36
#define NZ 8
#define RE 6371000.0
#define OMEGA 7.292115e-5

static void *lib;
#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define TRY(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, p_dycore_last_error()); return 3; } } while (0)

/* reference layout: [i][j][k], C order, one halo cell each side, +1 on a staggered axis */
static double *field(int stgx, int stgy, int nk, double val)
{
    size_t n = (size_t)(NX + 2 + stgx) * (NY + 2 + stgy) * nk;
    double *a = malloc(n * sizeof *a);
    for (size_t e = 0; e < n; e++) a[e] = val;
    return a;
}
#define AT(a, stgy, nk, i, j, k) (a)[((size_t)(i) * (NY + 2 + (stgy)) + (j)) * (nk) + (k)]

int main(int argc, char **argv)
{
    lib = dlopen(argc > 1 ? argv[1] : "climate_model_amd/csrc/libdycore_hip.so", RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    SYM(dycore_create) SYM(dycore_destroy) SYM(dycore_last_error) SYM(dycore_abi_version)
    SYM(dycore_set_grid) SYM(dycore_upload) SYM(dycore_download) SYM(dycore_primary_diag) SYM(dycore_step_matsuno)
    if (p_dycore_abi_version() != DYCORE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }

    DycoreConfig cfg = {0};
    cfg.abi_version = DYCORE_ABI_VERSION;
    cfg.nx = NX; cfg.ny = NY; cfg.nz = NZ; cfg.ny_global = NY; cfg.pair_top = 10000.0;
    dycore_t *h = NULL;
    TRY(p_dycore_create(&cfg, &h));

    const double dlon = 360.0 / NX / 180.0 * M_PI, dlat = 180.0 / NY / 180.0 * M_PI;
    double *A = field(0, 0, 1, NAN), *dxjs = field(0, 1, 1, NAN), *dyis = field(1, 0, 1, NAN);
    double *corf = field(0, 0, 1, NAN), *corf_is = field(1, 0, 1, NAN);
    double *lat = field(0, 0, 1, NAN), *lat_is = field(1, 0, 1, NAN);
    double *dlon_a = field(0, 1, 1, dlon), *dlat_a = field(1, 0, 1, dlat);
    for (int i = 0; i < NX + 3; i++)
        for (int j = 0; j < NY + 3; j++) {
            /* halo rows mirror the wall rule of misc_boundaries.py:36-41 */
            int jm = j < 1 ? 1 : (j > NY ? NY : j);
            double phi = (-90.0 + (jm - 0.5) * 180.0 / NY) / 180.0 * M_PI;
            int js = j < 1 ? 1 : (j > NY + 1 ? NY + 1 : j);
            double phi_js = (-90.0 + (js - 1) * 180.0 / NY) / 180.0 * M_PI;
            if (i < NX + 2 && j < NY + 2) {
                AT(A, 0, 1, i, j, 0) = cos(phi) * dlon * dlat * RE * RE;
                AT(corf, 0, 1, i, j, 0) = 2 * OMEGA * sin(phi);
                AT(lat, 0, 1, i, j, 0) = phi;
            }
            if (j < NY + 2) {
                AT(dyis, 0, 1, i, j, 0) = dlat * RE;
                AT(corf_is, 0, 1, i, j, 0) = 2 * OMEGA * sin(phi);
                AT(lat_is, 0, 1, i, j, 0) = phi;
            }
            if (i < NX + 2) AT(dxjs, 1, 1, i, j, 0) = cos(phi_js) * dlon * RE;
        }
    double sigma_vb[NZ + 1], dsigma[NZ], uvdif[NZ], ptdif[NZ];
    for (int k = 0; k <= NZ; k++) sigma_vb[k] = pow((double)k / NZ, 1.5);
    for (int k = 0; k < NZ; k++) {
        dsigma[k] = sigma_vb[k + 1] - sigma_vb[k];
        uvdif[k] = 4.0;                                         /* namelist.py:286-316 at 5 deg */
        ptdif[k] = 1e-5 * exp(-1.5 * (NZ - k - 1) / (double)NZ);
    }
    TRY(p_dycore_set_grid(h, "A", A, NX + 2, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "corf", corf, NX + 2, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "lat_rad", lat, NX + 2, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "dyis", dyis, NX + 3, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "corf_is", corf_is, NX + 3, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "lat_is_rad", lat_is, NX + 3, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "dlat_rad", dlat_a, NX + 3, NY + 2, 1));
    TRY(p_dycore_set_grid(h, "dxjs", dxjs, NX + 2, NY + 3, 1));
    TRY(p_dycore_set_grid(h, "dlon_rad", dlon_a, NX + 2, NY + 3, 1));
    TRY(p_dycore_set_grid(h, "sigma_vb", sigma_vb, NZ + 1, 1, 1));
    TRY(p_dycore_set_grid(h, "dsigma", dsigma, NZ, 1, 1));
    TRY(p_dycore_set_grid(h, "UVFLX_dif_coef", uvdif, NZ, 1, 1));
    TRY(p_dycore_set_grid(h, "POTT_dif_coef", ptdif, NZ, 1, 1));
    TRY(p_dycore_set_grid(h, "moist_dif_coef", ptdif, NZ, 1, 1));

    double *HSURF = field(0, 0, 1, 0.0), *COLP = field(0, 0, 1, 90000.0);
    double *U = field(1, 0, NZ, 0.0), *V = field(0, 1, NZ, 0.0), *T = field(0, 0, NZ, 0.0);
    double *W = field(0, 0, NZ + 1, 0.0), *TVB = field(0, 0, NZ + 1, 0.0);
    for (int i = 0; i < NX + 2; i++)
        for (int j = 0; j < NY + 2; j++)
            for (int k = 0; k < NZ; k++) AT(T, 0, NZ, i, j, k) = 300.0 + 5.0 * (NZ - k);
    TRY(p_dycore_upload(h, "HSURF", HSURF)); TRY(p_dycore_upload(h, "COLP", COLP));
    TRY(p_dycore_upload(h, "UWIND", U)); TRY(p_dycore_upload(h, "VWIND", V)); TRY(p_dycore_upload(h, "POTT", T));
    TRY(p_dycore_upload(h, "WWIND", W)); TRY(p_dycore_upload(h, "POTTVB", TVB));
    TRY(p_dycore_primary_diag(h));                              /* solver.py:70-73 */
    TRY(p_dycore_step_matsuno(h, 60.0, 10, 0));                 /* 10 x dyn_matsuno.step_matsuno */
    TRY(p_dycore_download(h, "UWIND", U)); TRY(p_dycore_download(h, "VWIND", V));
    TRY(p_dycore_download(h, "COLP", COLP)); TRY(p_dycore_download(h, "POTT", T));

    double umax = 0, vmax = 0, dcolp = 0, dpott = 0;
    for (int i = 1; i <= NX; i++)
        for (int j = 1; j <= NY; j++) {
            dcolp = fmax(dcolp, fabs(AT(COLP, 0, 1, i, j, 0) - 90000.0));
            for (int k = 0; k < NZ; k++) {
                umax = fmax(umax, fabs(AT(U, 0, NZ, i, j, k)));
                vmax = fmax(vmax, fabs(AT(V, 1, NZ, i, j, k)));
                dpott = fmax(dpott, fabs(AT(T, 0, NZ, i, j, k) - (300.0 + 5.0 * (NZ - k))));
            }
        }
    printf("after 10 steps: max|U| %.3e  max|V| %.3e  max|dCOLP| %.3e  max|dPOTT| %.3e\n", umax, vmax, dcolp, dpott);
    TRY(p_dycore_destroy(h));
    if (!(umax < 1e-9 && vmax < 1e-9 && dcolp < 1e-6 && dpott < 1e-9)) { fprintf(stderr, "atmosphere at rest did not stay at rest\n"); return 1; }
    printf("C_CLIENT_OK\n");
    return 0;
}
