// Microbenchmark: what does ONE CU sustain on the Jacobian store pattern?  G workgroups of 256 threads (one per CU up to 256),
// every wavefront issues back-to-back nontemporal 16-byte-per-lane stores, two 512-byte runs per instruction a "column" apart
// (the pair stores of pj_rblk.hip), consecutive instructions 8 n bytes apart.  Prints bytes per clock and CU for G = 8 .. 1024:
// if the per-CU rate is the same at 32 busy CUs as at 256, the limit is the CU's store path, not HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_store(double* out, long n, int nst, int reps)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wavefront w of workgroup b owns states [64 * (b * WAVES + w), +64): lanes 0..31 column c, lanes 32..63 column c + 1
    const long s0 = 64L * ((long)blockIdx.x * WAVES + wave);
    const long sp = s0 + 2 * (lane & 31);
    const d2 v = {(double)lane, (double)wave};
    for (int r = 0; r < reps; ++r)
#pragma unroll 8
        for (int e = 0; e < nst; ++e) {
            d2* p = (d2*)(out + (long)(2 * e + (lane >> 5)) * n + sp);
            __builtin_nontemporal_store(v, p);
        }
}

int main(int argc, char** argv)
{
    const long n = 1000000;          // states (row length of the SoA Jacobian)
    const int nst = 1024;            // store instructions per wavefront and repetition (2 columns each): 2048 "entries"
    double* buf;
    CHK(hipMalloc(&buf, sizeof(double) * n * 2 * nst));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    int clk_khz = 0; CHK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("pair stores (16 B per lane, 1 KB per instruction), %d instructions per wavefront, clock attribute %.2f GHz\n", nst, clk_khz / 1e6);
    auto run = [&](auto kern, int G, int waves, int reps) {
        hipLaunchKernelGGL(kern, dim3(G), dim3(64 * waves), 0, 0, buf, n, nst, 1); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(G), dim3(64 * waves), 0, 0, buf, n, nst, reps);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        const double bytes = 1024.0 * nst * reps * waves * G;
        const int cus = G < 256 ? G : 256;
        printf("  %4d workgroups x %d wavefront(s): %8.3f ms  %7.0f GB/s  %6.2f B/clk/CU (at 2.1 GHz), %5.0f clk per store instruction and wavefront\n",
               G, waves, ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.1e9 / cus, ms * 1e-3 * 2.1e9 / (nst * reps * (G > 256 ? G / 256.0 : 1.0)));
    };
    for (int G : {8, 32, 64, 128, 256, 512, 1024}) run(k_store<4>, G, 4, 8);
    for (int G : {8, 32, 128, 256}) run(k_store<1>, G, 1, 8);
    for (int G : {32, 256}) run(k_store<2>, G, 2, 8);
    for (int G : {32, 256}) run(k_store<8>, G, 8, 8);
    return 0;
}
