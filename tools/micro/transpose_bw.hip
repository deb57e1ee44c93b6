// SoA -> AoS transposition of a chunk of Jacobians (m states x NE entries): the kernel of pj_rblk.hip (64 x 64 tiles,
// 8 bytes per lane both ways) against variants that move 16 bytes per lane on the store / on both sides.
// hipcc --offload-arch=gfx950 -O3 -o transpose_bw transpose_bw.hip && ./transpose_bw 65536 2809
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NEc = 2809;
__global__ void __launch_bounds__(256) k_base(const double* __restrict__ src, long m, double* __restrict__ dst, int NE)
{
    __shared__ double tile[64][65];
    const long s0 = (long)blockIdx.x * 64;
    const int e0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int r = ly; r < 64; r += 4) {
        const int e = e0 + r;
        const long sidx = s0 + lx;
        if (e < NE && sidx < m) tile[r][lx] = __builtin_nontemporal_load(&src[(long)e * m + sidx]);
    }
    __syncthreads();
    for (int r = ly; r < 64; r += 4) {
        const long sidx = s0 + r;
        const int e = e0 + lx;
        if (sidx < m && e < NE) __builtin_nontemporal_store(tile[lx][r], &dst[sidx * NE + e]);
    }
}
// 128 entries x 64 states per workgroup; stores: a wavefront writes 1 KB of one state (16 bytes per lane)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_st16(const double* __restrict__ src, long m, double* __restrict__ dst, int NE)
{
    __shared__ double tile[128][65];
    const long s0 = (long)blockIdx.x * 64;
    const int e0 = blockIdx.y * 128;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll 8
    for (int r = ly; r < 128; r += 4) {
        const int e = e0 + r;
        const long sidx = s0 + lx;
        if (e < NE && sidx < m) tile[r][lx] = __builtin_nontemporal_load(&src[(long)e * m + sidx]);
    }
    __syncthreads();
#pragma unroll 8
    for (int r = ly; r < 64; r += 4) {
        const long sidx = s0 + r;
        const int e = e0 + 2 * lx;
        if (sidx < m) {
            double* p = &dst[sidx * NE + e];
            // (state blocks start at odd multiples of 8 bytes when NE is odd: 8-byte stores, the pair is still one run)
            if (e < NE) __builtin_nontemporal_store(tile[2 * lx][r], p);
            if (e + 1 < NE) __builtin_nontemporal_store(tile[2 * lx + 1][r], p + 1);
        }
    }
}
// 64 entries x 128 states: loads 16 bytes per lane (two states), stores 8 bytes
__global__ void __launch_bounds__(256) k_ld16(const double* __restrict__ src, long m, double* __restrict__ dst, int NE)
{
    __shared__ double tile[64][129];
    const long s0 = (long)blockIdx.x * 128;
    const int e0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll 8
    for (int r = ly; r < 64; r += 4) {
        const int e = e0 + r;
        const long sidx = s0 + 2 * lx;
        if (e < NE && sidx + 1 < m) {
            const d2 v = __builtin_nontemporal_load((const d2*)&src[(long)e * m + sidx]);
            tile[r][2 * lx] = v.x; tile[r][2 * lx + 1] = v.y;
        }
    }
    __syncthreads();
#pragma unroll 8
    for (int r = ly; r < 128; r += 4) {
        const long sidx = s0 + r;
        const int e = e0 + lx;
        if (sidx < m && e < NE) __builtin_nontemporal_store(tile[lx][r], &dst[sidx * NE + e]);
    }
}
int main(int argc, char** argv)
{
    const long m = argc > 1 ? atol(argv[1]) : 65536;
    const int NE = argc > 2 ? atoi(argv[2]) : NEc;
    double *a, *b;
    hipMalloc(&a, sizeof(double) * m * NE); hipMalloc(&b, sizeof(double) * m * NE);
    hipMemset(a, 0, sizeof(double) * m * NE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-8s %.3f ms  %.0f GB/s (read + write)\n", name, ms, 2.0 * 8.0 * m * NE / ms / 1e6);
    };
    run("base", [&] { hipLaunchKernelGGL(k_base, dim3((m + 63) / 64, (NE + 63) / 64), dim3(256), 0, 0, a, m, b, NE); });
    run("st16", [&] { hipLaunchKernelGGL(k_st16, dim3((m + 63) / 64, (NE + 127) / 128), dim3(256), 0, 0, a, m, b, NE); });
    run("ld16", [&] { hipLaunchKernelGGL(k_ld16, dim3((m + 127) / 128, (NE + 63) / 64), dim3(256), 0, 0, a, m, b, NE); });
    hipMemcpyAsync(b, a, sizeof(double) * m * NE, hipMemcpyDeviceToDevice, 0);
    run("memcpy", [&] { hipMemcpyAsync(b, a, sizeof(double) * m * NE, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
