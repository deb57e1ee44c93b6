// Microbenchmark: alternating compute / store bursts of the row-block kernels at one wavefront per SIMD.
// One state per lane, NB blocks of [C dependent FMAs, then W SoA stores 8*n bytes apart].
//   mode 0: 8 B per lane (one 512 B run per wavefront store)
//   mode 1: 16 B per lane, even lanes entry e / odd lanes entry e+1 (two 512 B runs per store, half the
//           store instructions)
//   mode 2: as 0 with plain (not nontemporal) stores
// Prints achieved store GB/s and the time a pure-compute run of the same C takes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int W, bool PERM = false>
__global__ void __launch_bounds__(256) k_burst(double* out, long n, int nb, int C, int do_store)
{
    __shared__ double pad[100 * 128];
    if (n < 0) pad[threadIdx.x] = 1.0;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    double x = (double)s * 1e-9 + 0.5;
    const int lane = threadIdx.x & 63;
    for (int b = 0; b < nb; ++b) {
        for (int c = 0; c < C; ++c) x = x * 0.999999 + 1e-7;      // dependent chain: ~8 cycles each
        if (!do_store) continue;
        if constexpr (MODE == 1) {
            // lane pair (2m, 2m+1): even lane writes states 2m, 2m+1 of entry e, odd lane those of entry e + 1
            const long sp = s - (lane & 1);
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 v = {x, x + 1.0};
#pragma unroll
            for (int u = 0; u < W; u += 2) {
                d2* p = (d2*)(out + (long)(b * W + u + (lane & 1)) * n + sp);
                __builtin_nontemporal_store(v, p);
            }
        } else {
#pragma unroll
            for (int u = 0; u < W; ++u) {
                // PERM: entries of one block 53 rows apart (a row block writes one row of every column)
                double* p = out + (long)(PERM ? u * 53 + b : b * W + u) * n + s;
                if (MODE == 0) __builtin_nontemporal_store(x + u, p); else *p = x + u;
            }
        }
    }
    if (x == 123.456) out[s] = x;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    const int nb = 46, W = 60;
    double* buf;
    CHK(hipMalloc(&buf, sizeof(double) * n * 53 * W));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto run = [&](auto kern, int C, int st) {
        const unsigned full = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(kern, dim3(full), dim3(256), 0, 0, buf, n, nb, C, st); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(full), dim3(256), 0, 0, buf, n, nb, C, st);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); return ms / 2;
    };
    printf("n=%ld, %d blocks x %d stores, 1 workgroup of 256 per CU\n", n, nb, W);
    for (int C : {0, 250, 500, 1000, 2000, 4000}) {
        const float tc = C ? run(k_burst<0, W>, C, 0) : 0.f;
        const float t0 = run(k_burst<0, W>, C, 1), t1 = run(k_burst<1, W>, C, 1), t2 = run(k_burst<2, W>, C, 1);
        const float t3 = run(k_burst<0, 52, true>, C, 1);
        printf("C=%5d  8B nt, 52 stores 53 rows apart %7.3f ms %6.0f GB/s\n", C, t3, 8.0 * n * nb * 52 / 1e6 / t3);
        const double gb = 8.0 * n * nb * W / 1e6;
        printf("C=%5d  compute-only %7.3f ms | 8B nt %7.3f ms %6.0f GB/s | 16B nt %7.3f ms %6.0f GB/s | 8B plain %7.3f ms %6.0f GB/s\n",
               C, tc, t0, gb / t0, t1, gb / t1, t2, gb / t2);
    }
    return 0;
}
