#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NSP = 53, NE = NSP * NSP;
__global__ void __launch_bounds__(256) k_a(const double* __restrict__ src, long m, double* __restrict__ dst)
{
    __shared__ double tile[64][65];
    const long s0 = (long)blockIdx.x * 64;
    const int e0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int r = ly; r < 64; r += 4) { const int e = e0 + r; const long sidx = s0 + lx; if (e < NE && sidx < m) tile[r][lx] = __builtin_nontemporal_load(&src[(long)e * m + sidx]); }
    __syncthreads();
    for (int r = ly; r < 64; r += 4) { const long sidx = s0 + r; const int e = e0 + lx; if (sidx < m && e < NE) __builtin_nontemporal_store(tile[lx][r], &dst[sidx * NE + e]); }
}
// a workgroup walks a contiguous range of entry tiles of its 64 states: the boundary lines of neighbouring tiles are
// written by the same CU back to back; two tiles in LDS so that the loads of tile t + 1 travel while tile t is stored
template <int YS>
__global__ void __launch_bounds__(256) k_b(const double* __restrict__ src, long m, double* __restrict__ dst)
{
    __shared__ double tile[2][64][65];
    constexpr int NT = (NE + 63) / 64, PER = (NT + YS - 1) / YS;
    const long s0 = (long)blockIdx.x * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int t0 = blockIdx.y * PER, t1 = t0 + PER < NT ? t0 + PER : NT;
    double v[16];
    auto ld = [&](int t) { for (int q = 0; q < 16; ++q) { const int e = t * 64 + ly + 4 * q; const long sidx = s0 + lx; v[q] = (e < NE && sidx < m) ? __builtin_nontemporal_load(&src[(long)e * m + sidx]) : 0.0; } };
    if (t0 < t1) ld(t0);
    for (int t = t0; t < t1; ++t) {
        const int b = (t - t0) & 1;
        for (int q = 0; q < 16; ++q) tile[b][ly + 4 * q][lx] = v[q];
        if (t + 1 < t1) ld(t + 1);
        __syncthreads();
        for (int q = 0; q < 16; ++q) { const int r = ly + 4 * q; const long sidx = s0 + r; const int e = t * 64 + lx; if (sidx < m && e < NE) __builtin_nontemporal_store(tile[b][lx][r], &dst[sidx * NE + e]); }
    }
}
int main()
{
    const long m = 65536;
    double *src, *dst;
    hipMalloc(&src, sizeof(double) * NE * m); hipMalloc(&dst, sizeof(double) * NE * m);
    hipMemset(src, 1, sizeof(double) * NE * m);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* nm, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %.3f ms  %.2f TB/s\n", nm, ms, 2.0 * 8 * NE * m / ms / 1e9);
    };
    time("tile per workgroup (shipped)", [&] { hipLaunchKernelGGL(k_a, dim3(m / 64, (NE + 63) / 64), dim3(256), 0, 0, src, m, dst); });
    time("tile range, 1 per state tile", [&] { hipLaunchKernelGGL(k_b<1>, dim3(m / 64, 1), dim3(256), 0, 0, src, m, dst); });
    time("tile range, 2", [&] { hipLaunchKernelGGL(k_b<2>, dim3(m / 64, 2), dim3(256), 0, 0, src, m, dst); });
    time("tile range, 4", [&] { hipLaunchKernelGGL(k_b<4>, dim3(m / 64, 4), dim3(256), 0, 0, src, m, dst); });
    time("tile range, 11", [&] { hipLaunchKernelGGL(k_b<11>, dim3(m / 64, 11), dim3(256), 0, 0, src, m, dst); });
    time("tile per workgroup (shipped)", [&] { hipLaunchKernelGGL(k_a, dim3(m / 64, (NE + 63) / 64), dim3(256), 0, 0, src, m, dst); });
    return 0;
}
