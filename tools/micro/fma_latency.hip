// fp64 FMA issue / dependent-issue cost on one wavefront per SIMD (gfx950): K independent chains of
// dependent v_fma_f64, s_memtime around 4096 x K instructions.  Prints shader cycles per instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o fma_latency fma_latency.hip ; run: ./fma_latency
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void __launch_bounds__(256) chains(double* out, long long* cyc, double a, double b)
{
    double x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = (double)threadIdx.x + k;
    long long t0 = clock64();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < K; ++k) x[k] = __builtin_fma(x[k], a, b);
        }
    }
    long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int K>
void run(double* d_out, long long* d_cyc, int blocks)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(chains<K>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 0.999999, 1e-9);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(chains<K>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 0.999999, 1e-9);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    const double n = 4096.0 * K;
    printf("chains %d, %d workgroups of 256: %.2f clock64 ticks per fma, %.3f ns per fma per wavefront\n", K, blocks,
           (double)c / n, ms * 1e6 / n);
}
int main()
{
    double* d_out; long long* d_cyc;
    (void)hipMalloc(&d_out, 8 * 256 * 1024); (void)hipMalloc(&d_cyc, 8);
    for (int blocks : {1, 256}) {
        run<1>(d_out, d_cyc, blocks); run<2>(d_out, d_cyc, blocks); run<3>(d_out, d_cyc, blocks);
        run<4>(d_out, d_cyc, blocks); run<8>(d_out, d_cyc, blocks);
    }
    return 0;
}
