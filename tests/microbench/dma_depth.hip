// Does a deeper LDS-DMA prefetch ring raise the streaming rate of the momentum kernel's access
// pattern?  Workgroup = 64 x 8 columns, NF fields staged per level as (8 + HALO) x 68 tiles with
// global_load_lds (16 B/lane), DEPTH buffers in flight, one barrier per level, token compute
// (one LDS read per field and thread), two outputs per cell.  Same tile order as k_momentum_dma.
//   hipcc -O3 --offload-arch=gfx950 dma_depth.hip -o dma_depth && ./dma_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
constexpr int NF = 9, TW = 68;

template <int DEPTH, int ROWS>
__global__ __launch_bounds__(512) void stream(const double *base, double *out0, double *out1, int P, int NJ, long plane,
                                              long fstride, int gbx, int gby, int nz)
{
    constexpr int TN = TW * ROWS, NCH = TN / 2, NSL = (NCH + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int nblk = gbx * gby, chunk = (nblk + 7) / 8;
    const int lin = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
    if (lin >= nblk) return;
    const int bx = lin % gbx, by = lin / gbx;
    const int tid = threadIdx.y * 64 + threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i0 = bx * 64 + 1, j0 = by * 8 + 1;
    const int ch = wave * 64 + lane;
    const bool on = wave < NSL && ch < NCH;
    const int e = 2 * (on ? ch : 0), r = e / TW, c2 = e - r * TW;
    const long goff = (long)min(j0 - 1 + r, NJ - 1) * P + min(i0 - 2 + c2, P - 2);
    auto issue = [&](int k, int b) {
        if (on) {
            double *buf = sm + (size_t)b * NF * TN;
#pragma unroll
            for (int f = 0; f < NF; f++)
                __builtin_amdgcn_global_load_lds((gbl_void *)(base + f * fstride + (long)k * plane + goff),
                                                 (lds_void *)(buf + f * TN + wave * 128), 16, 0, 0);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; d++) issue(d < nz ? d : nz - 1, d);
    const int i = i0 + threadIdx.x, j = j0 + threadIdx.y;
    const bool act = i <= P - 3 && j <= NJ - 3;
    const long cell = (long)j * P + i;
    for (int k = 0; k < nz; k++) {
        // everything issued so far except the newest DEPTH-2 levels must have landed
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");       // NF loads of the level after this one may be outstanding
        __syncthreads();
        if (k + DEPTH - 1 < nz) issue(k + DEPTH - 1, (k + DEPTH - 1) % DEPTH);
        else if (DEPTH > 2) issue(nz - 1, (k + DEPTH - 1) % DEPTH);   // keep the vmcnt bookkeeping uniform
        const double *buf = sm + (size_t)(k % DEPTH) * NF * TN;
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            s0 += buf[f * TN + (threadIdx.y + 1) * TW + threadIdx.x + 2];
            s1 += buf[f * TN + (threadIdx.y + 2) * TW + threadIdx.x + 1];
        }
        if (act) { out0[(long)k * plane + cell] = s0; out1[(long)k * plane + cell] = s1; }
    }
}

template <int DEPTH, int ROWS>
int run(const double *d, double *o0, double *o1, int P, int NJ, long plane, long fstride, int nx, int ny, int nz)
{
    const int gbx = (nx + 63) / 64, gby = (ny + 7) / 8;
    const size_t lds = (size_t)DEPTH * NF * TW * ROWS * 8;
    CK(hipFuncSetAttribute((const void *)stream<DEPTH, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const dim3 gr(8 * ((gbx * gby + 7) / 8)), bl(64, 8);
    for (int w = 0; w < 2; w++) stream<DEPTH, ROWS><<<gr, bl, lds>>>(d, o0, o1, P, NJ, plane, fstride, gbx, gby, nz);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int w = 0; w < reps; w++) stream<DEPTH, ROWS><<<gr, bl, lds>>>(d, o0, o1, P, NJ, plane, fstride, gbx, gby, nz);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double compulsory = (double)(NF + 2) * nx * ny * nz * 8;
    printf("depth %d rows %2d  LDS %6.1f KB  %.3f ms  compulsory-bytes rate %.2f TB/s\n", DEPTH, ROWS, lds / 1024.0, ms, compulsory / ms / 1e9);
    return 0;
}

int main()
{
    const int nx = 1440, ny = 720, nz = 64, P = 1472, NJ = ny + 3;
    const long plane = (long)P * NJ, fstride = plane * (nz + 1) + 64;
    double *d, *o0, *o1;
    CK(hipMalloc((void **)&d, (size_t)NF * fstride * 8 + 4096));
    CK(hipMalloc((void **)&o0, (size_t)plane * nz * 8)); CK(hipMalloc((void **)&o1, (size_t)plane * nz * 8));
    CK(hipMemset(d, 0, (size_t)NF * fstride * 8 + 4096));
    const double *base = d + 63;      // i = 1 on a 512-B boundary as in the library
    if (run<2, 11>(base, o0, o1, P, NJ, plane, fstride, nx, ny, nz)) return 1;
    if (run<2, 10>(base, o0, o1, P, NJ, plane, fstride, nx, ny, nz)) return 1;
    if (run<3, 10>(base, o0, o1, P, NJ, plane, fstride, nx, ny, nz)) return 1;
    if (run<3, 9>(base, o0, o1, P, NJ, plane, fstride, nx, ny, nz)) return 1;
    if (run<2, 9>(base, o0, o1, P, NJ, plane, fstride, nx, ny, nz)) return 1;
    return 0;
}
