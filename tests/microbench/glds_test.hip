// Semantics check of __builtin_amdgcn_global_load_lds (16 B/lane) for the dycore tile staging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int FTW = 68, FTH = 11, FTN = FTW * FTH, NCH = FTN / 2;   // 374 16-byte chunks

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__global__ __launch_bounds__(512) void tile_copy(const double *src, double *dst, int P, int NJ, int i0, int j0)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int ch = wave * 64 + lane;
    if (ch < NCH) {
        const int e = 2 * ch, r = e / FTW, c = e - r * FTW;
        const int gi = min(i0 + c, P - 2), gj = min(j0 + r, NJ - 1);
        const double *g = src + (size_t)gj * P + gi;
        double *l = sm + wave * 128;                 // wave-uniform LDS base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)l, 16, 0, 0);
    }
    __syncthreads();
    for (int e = tid; e < FTN; e += 512) dst[e] = sm[e];
}

int main()
{
    const int P = 1472, NJ = 723;
    std::vector<double> h((size_t)P * NJ);
    for (size_t t = 0; t < h.size(); t++) h[t] = (double)t;
    double *d, *o;
    CK(hipMalloc((void **)&d, h.size() * 8));
    CK(hipMalloc((void **)&o, FTN * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    int bad_total = 0;
    for (int i0 : {0, 64, 1408}) for (int j0 : {0, 8, 712}) {
        tile_copy<<<1, dim3(64, 8), FTN * 8>>>(d, o, P, NJ, i0, j0);
        std::vector<double> r(FTN);
        CK(hipMemcpy(r.data(), o, FTN * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int e = 0; e < FTN; e++) {
            int rr = e / FTW, c = e % FTW;
            int cc = (c / 2) * 2;                        // chunk base column
            int gi = std::min(i0 + cc, P - 2) + (c - cc), gj = std::min(j0 + rr, NJ - 1);
            double want = (double)((size_t)gj * P + gi);
            if (r[e] != want) { if (bad < 3) printf("  i0=%d j0=%d e=%d got %.0f want %.0f\n", i0, j0, e, r[e], want); bad++; }
        }
        printf("tile i0=%d j0=%d: %d mismatches\n", i0, j0, bad);
        bad_total += bad;
    }
    printf(bad_total ? "FAIL\n" : "PASS\n");
    return bad_total != 0;
}
