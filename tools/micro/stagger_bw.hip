// Microbenchmark: do the four wavefronts of a workgroup (one per SIMD, started together, same code) hide their
// store bursts better when their compute / store phases are shifted against each other?
// One state per lane, NB blocks of [C dependent FMAs, then W 16-byte SoA stores]; stagger modes:
//   0: none   1: wavefront w of a workgroup starts with w * C / 4 extra FMAs   2: workgroup g with (g & 3) * C / 4
//   3: both (wavefront and workgroup shifts)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int W>
__global__ void __launch_bounds__(256) k_burst(double* out, long n, int nb, int C, int do_store, int stag)
{
    __shared__ double pad[100 * 128];
    if (n < 0) pad[threadIdx.x] = 1.0;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    double x = (double)s * 1e-9 + 0.5;
    const int lane = threadIdx.x & 63;
    int extra = 0;
    if (stag & 1) extra += (int)(threadIdx.x >> 6) * C / 4;
    if (stag & 2) extra += (int)(blockIdx.x & 3) * C / 4;
    for (int c = 0; c < extra; ++c) x = x * 0.999999 + 1e-7;
    for (int b = 0; b < nb; ++b) {
        for (int c = 0; c < C; ++c) x = x * 0.999999 + 1e-7;      // dependent chain
        if (!do_store) continue;
        const long sp = s - (lane & 1);
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 v = {x, x + 1.0};
#pragma unroll
        for (int u = 0; u < W; u += 2) {
            d2* p = (d2*)(out + (long)(b * W + u + (lane & 1)) * n + sp);
            __builtin_nontemporal_store(v, p);
        }
    }
    if (x == 123.456) out[s] = x;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    const int nb = 46, W = 60;
    double* buf;
    CHK(hipMalloc(&buf, sizeof(double) * n * nb * W));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto run = [&](int C, int st, int stag) {
        const unsigned full = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_burst<W>, dim3(full), dim3(256), 0, 0, buf, n, nb, C, st, stag); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_burst<W>, dim3(full), dim3(256), 0, 0, buf, n, nb, C, st, stag);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / 3;
    };
    printf("n=%ld, %d blocks x %d entries (16-byte stores), 1 workgroup of 256 per CU; store-only time %.3f ms\n", n, nb, W, run(0, 1, 0));
    for (int C : {125, 250, 375, 500, 750, 1000}) {
        printf("C=%5d  compute-only %7.3f ms |", C, run(C, 0, 0));
        for (int stag = 0; stag < 4; ++stag) printf(" stagger %d: %7.3f ms (compute-only with the shift %7.3f) |", stag, run(C, 1, stag), run(C, 0, stag));
        printf("\n");
    }
    return 0;
}
