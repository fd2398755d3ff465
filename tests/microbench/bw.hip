// Bandwidth calibration for the dycore's access pattern (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// lane <-> i, ty <-> j, k loop in thread: NR reads + 1 write per cell
template <int NR, int TJ, int UNR>
__global__ void kloop(const double *const *in, double *out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * TJ + threadIdx.y + 1;
    if (i > nx || j > ny) return;
    const size_t c = (size_t)j * P + i;
#pragma unroll UNR
    for (int k = 0; k < nz; k++) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) s += in[r][c + (size_t)k * plane];
        out[c + (size_t)k * plane] = s;
    }
}
// k loop in thread with __restrict__ pointers passed directly (compiler may hoist the loads of
// level k+1.. above the stores of level k) and explicit unroll
template <int TJ, int UNR>
__global__ void kloop_r4(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c4,
                         const double *__restrict__ d, double *__restrict__ out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * TJ + threadIdx.y + 1;
    if (i > nx || j > ny) return;
    const size_t c = (size_t)j * P + i;
#pragma unroll UNR
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * plane;
        out[q] = a[q] + b[q] + c4[q] + d[q];
    }
}
// explicit software prefetch: loads of level k+1 issued before the store of level k
template <int TJ>
__global__ void kloop_pf4(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c4,
                          const double *__restrict__ d, double *__restrict__ out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * TJ + threadIdx.y + 1;
    if (i > nx || j > ny) return;
    const size_t c = (size_t)j * P + i;
    double a0 = a[c], b0 = b[c], c0 = c4[c], d0 = d[c];
    double a1 = a[c + plane], b1 = b[c + plane], c1 = c4[c + plane], d1 = d[c + plane];
#pragma unroll 1
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * plane;
        const size_t qn = c + (size_t)(k + 2 < nz ? k + 2 : k) * plane;
        const double a2 = a[qn], b2 = b[qn], c2 = c4[qn], d2 = d[qn];
        out[q] = a0 + b0 + c0 + d0;
        a0 = a1; b0 = b1; c0 = c1; d0 = d1;
        a1 = a2; b1 = b2; c1 = c2; d1 = d2;
    }
}
// wide blocks: WX lanes along i, one row per block
template <int WX, int SYNC>
__global__ void kloop_wide(const double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c4,
                           const double *__restrict__ d, double *__restrict__ out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = blockIdx.x * WX + threadIdx.x + 1, j = blockIdx.y + 1;
    const bool ok = (i <= nx);
    const size_t c = (size_t)j * P + (ok ? i : 1);
    for (int k = 0; k < nz; k++) {
        const size_t q = c + (size_t)k * plane;
        const double v = a[q] + b[q] + c4[q] + d[q];
        if (ok) out[q] = v;
        if (SYNC) __syncthreads();
    }
}
// same mapping, [j][k][i] layout: k stride = P, j stride = nk*P
template <int NR, int TJ>
__global__ void kloop_jki(const double *const *in, double *out, int nx, int ny, int nz, int P, int nk)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * TJ + threadIdx.y + 1;
    if (i > nx || j > ny) return;
    const size_t c = (size_t)j * nk * P + i;
    for (int k = 0; k < nz; k++) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) s += in[r][c + (size_t)k * P];
        out[c + (size_t)k * P] = s;
    }
}
// [j][k][i] layout with a 3-point j stencil (j-1, j, j+1) on every input: NR reads x3
template <int NR, int TJ>
__global__ void kloop_jki_st(const double *const *in, double *out, int nx, int ny, int nz, int P, int nk)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * TJ + threadIdx.y + 1;
    if (i > nx || j > ny) return;
    const size_t sj = (size_t)nk * P;
    const size_t c = (size_t)j * sj + i;
    for (int k = 0; k < nz; k++) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) { const double *f = in[r] + c + (size_t)k * P; s += f[0] + f[-1] + f[1] + f[-(long long)sj] + f[sj]; }
        out[c + (size_t)k * P] = s;
    }
}
template <int NR, int TJ>
__global__ void kloop_st(const double *const *in, double *out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 1, j = blockIdx.y * TJ + threadIdx.y + 1;
    if (i > nx || j > ny) return;
    const size_t c = (size_t)j * P + i;
    for (int k = 0; k < nz; k++) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) { const double *f = in[r] + c + (size_t)k * plane; s += f[0] + f[-1] + f[1] + f[-P] + f[P]; }
        out[c + (size_t)k * plane] = s;
    }
}
// one thread per cell, k in blockIdx.z
template <int NR>
__global__ void flat(const double *const *in, double *out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = blockIdx.x * 256 + threadIdx.x + 1, j = blockIdx.y + 1, k = blockIdx.z;
    if (i > nx) return;
    const size_t q = (size_t)k * plane + (size_t)j * P + i;
    double s = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) s += in[r][q];
    out[q] = s;
}
// float4-style: each thread 2 doubles (16 B), aligned, one thread per pair, k in blockIdx.z
template <int NR>
__global__ void flat2(const double *const *in, double *out, int nx, int ny, int nz, int P, long long plane)
{
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2, j = blockIdx.y + 1, k = blockIdx.z;
    if (i > nx + 1) return;
    const size_t q = (size_t)k * plane + (size_t)j * P + i;
    double2 s = {0, 0};
#pragma unroll
    for (int r = 0; r < NR; r++) { double2 v = *(const double2 *)(in[r] + q); s.x += v.x; s.y += v.y; }
    *(double2 *)(out + q) = s;
}

int main()
{
    const int nx = 1440, ny = 720, nz = 64, P = 1472, NJ = 723;
    const long long plane = (long long)P * NJ;
    const size_t n = (size_t)plane * (nz + 1);
    const int NF = 9;
    std::vector<double *> f(NF);
    for (auto &p : f) { CK(hipMalloc((void **)&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
    double **din;
    CK(hipMalloc((void **)&din, NF * sizeof(double *)));
    CK(hipMemcpy(din, f.data(), NF * sizeof(double *), hipMemcpyHostToDevice));
    double *out = f[NF - 1];
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double cellbytes = (double)nx * ny * nz * 8;
    auto report = [&](const char *name, int passes, float ms) {
        printf("%-34s %8.3f ms  %7.1f GB/s (%d passes)\n", name, ms, passes * cellbytes / ms / 1e6, passes);
    };
#define TIME(name, passes, ...)                                                \
    do {                                                                        \
        for (int w = 0; w < 2; w++) { __VA_ARGS__; }                                 \
        CK(hipEventRecord(a));                                                  \
        for (int it = 0; it < 5; it++) { __VA_ARGS__; }                              \
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));                      \
        float ms; CK(hipEventElapsedTime(&ms, a, b)); report(name, passes, ms / 5); \
    } while (0)
    TIME("hipMemcpyDtoD (1R+1W)", 2, CK(hipMemcpyAsync(f[1], f[0], (size_t)plane * nz * 8, hipMemcpyDeviceToDevice, 0)));
    dim3 g4((nx + 63) / 64, (ny + 3) / 4), b4(64, 4), g8((nx + 63) / 64, (ny + 7) / 8), b8(64, 8), g16((nx + 63) / 64, (ny + 15) / 16), b16(64, 16);
    TIME("kloop 1R+1W tj4", 2, kloop<1, 4, 1><<<g4, b4>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop 2R+1W tj4", 3, kloop<2, 4, 1><<<g4, b4>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop 4R+1W tj4", 5, kloop<4, 4, 1><<<g4, b4>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop 4R+1W tj4 unroll4", 5, kloop<4, 4, 4><<<g4, b4>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop 8R+1W tj4", 9, kloop<8, 4, 1><<<g4, b4>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop 4R+1W tj8", 5, kloop<4, 8, 1><<<g8, b8>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop 4R+1W tj16", 5, kloop<4, 16, 1><<<g16, b16>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop restrict 4R+1W unroll1", 5, kloop_r4<4, 1><<<g4, b4>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop restrict 4R+1W unroll2", 5, kloop_r4<4, 2><<<g4, b4>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop restrict 4R+1W unroll4", 5, kloop_r4<4, 4><<<g4, b4>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop restrict 4R+1W unroll8", 5, kloop_r4<4, 8><<<g4, b4>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop prefetch2 4R+1W", 5, kloop_pf4<4><<<g4, b4>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop restrict 4R+1W unroll4 tj1", 5, kloop_r4<1, 4><<<dim3((nx + 63) / 64, ny), dim3(64, 1)>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop restrict 4R+1W unroll4 tj16", 5, kloop_r4<16, 4><<<g16, b16>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop wide256x1 4R+1W", 5, kloop_wide<256, 0><<<dim3((nx + 255) / 256, ny), dim3(256)>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop wide512x1 4R+1W", 5, kloop_wide<512, 0><<<dim3((nx + 511) / 512, ny), dim3(512)>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop wide1024x1 4R+1W", 5, kloop_wide<1024, 0><<<dim3((nx + 1023) / 1024, ny), dim3(1024)>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop wide256x1 sync 4R+1W", 5, kloop_wide<256, 1><<<dim3((nx + 255) / 256, ny), dim3(256)>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    TIME("kloop wide1024x1 sync 4R+1W", 5, kloop_wide<1024, 1><<<dim3((nx + 1023) / 1024, ny), dim3(1024)>>>(f[0], f[1], f[2], f[3], out, nx, ny, nz, P, plane));
    for (int nyy : {640, 700, 710, 720, 730, 760, 800}) {   // tail-effect probe (NJ=723 rows allocated: cap)
        int nyc = nyy > 720 ? 720 : nyy;
        (void)nyc;
    }
    for (int nyy : {512, 600, 680, 704, 712, 720}) {
        dim3 gg((nx + 63) / 64, (nyy + 3) / 4);
        for (int w = 0; w < 2; w++) kloop_r4<4, 1><<<gg, b4>>>(f[0], f[1], f[2], f[3], out, nx, nyy, nz, P, plane);
        CK(hipEventRecord(a));
        for (int it = 0; it < 5; it++) kloop_r4<4, 1><<<gg, b4>>>(f[0], f[1], f[2], f[3], out, nx, nyy, nz, P, plane);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        const double waves = (double)((nx + 63) / 64) * nyy;
        printf("kloop ny=%d  waves=%.0f (%.3f x 8192)  %.3f ms  %.1f GB/s\n", nyy, waves, waves / 8192, ms, 5.0 * nx * nyy * nz * 8 / ms / 1e6);
    }
    TIME("kloop[j][k][i] 1R+1W tj4", 2, kloop_jki<1, 4><<<g4, b4>>>(din, out, nx, ny, nz, P, nz + 1));
    TIME("kloop[j][k][i] 4R+1W tj4", 5, kloop_jki<4, 4><<<g4, b4>>>(din, out, nx, ny, nz, P, nz + 1));
    TIME("kloop[j][k][i] 8R+1W tj4", 9, kloop_jki<8, 4><<<g4, b4>>>(din, out, nx, ny, nz, P, nz + 1));
    TIME("kloop[j][k][i] 4R+1W tj8", 5, kloop_jki<4, 8><<<g8, b8>>>(din, out, nx, ny, nz, P, nz + 1));
    TIME("kloop[j][k][i] 4R+1W tj1", 5, kloop_jki<4, 1><<<dim3((nx + 63) / 64, ny), dim3(64, 1)>>>(din, out, nx, ny, nz, P, nz + 1));
    TIME("kloop[k][j][i] 5pt 4R+1W tj4", 5, kloop_st<4, 4><<<g4, b4>>>(din, out, nx, ny, nz, P, plane));
    TIME("kloop[j][k][i] 5pt 4R+1W tj4", 5, kloop_jki_st<4, 4><<<g4, b4>>>(din, out, nx, ny, nz, P, nz + 1));
    TIME("kloop[j][k][i] 5pt 4R+1W tj8", 5, kloop_jki_st<4, 8><<<g8, b8>>>(din, out, nx, ny, nz, P, nz + 1));
    dim3 gf((nx + 255) / 256, ny, nz), gf2((nx / 2 + 1 + 255) / 256, ny, nz);
    TIME("flat 1R+1W", 2, flat<1><<<gf, dim3(256)>>>(din, out, nx, ny, nz, P, plane));
    TIME("flat 4R+1W", 5, flat<4><<<gf, dim3(256)>>>(din, out, nx, ny, nz, P, plane));
    TIME("flat 8R+1W", 9, flat<8><<<gf, dim3(256)>>>(din, out, nx, ny, nz, P, plane));
    TIME("flat2 (16B/lane) 1R+1W", 2, flat2<1><<<gf2, dim3(256)>>>(din, out, nx, ny, nz, P, plane));
    TIME("flat2 (16B/lane) 4R+1W", 5, flat2<4><<<gf2, dim3(256)>>>(din, out, nx, ny, nz, P, plane));
    TIME("flat2 (16B/lane) 8R+1W", 9, flat2<8><<<gf2, dim3(256)>>>(din, out, nx, ny, nz, P, plane));
    return 0;
}
