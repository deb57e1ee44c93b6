// Microbenchmark: does a per-workgroup scratch region that is rewritten and re-read every tile
// stay in the 256 MB Infinity Cache while a Jacobian-sized output stream passes through?
// Workgroup = 4 waves x the same 64 states; resident workgroups loop over state tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int NT>   // MODE 0: output only; 1: + scratch write/read;  NT: nontemporal output stores
__global__ void __launch_bounds__(256) k_fused(double* scr, double* out, long n, int nslot, int nout, int reread)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* my = scr + (long)blockIdx.x * nslot * 64 + lane;
    for (long t = blockIdx.x; t * 64 < n; t += gridDim.x) {
        const long s = t * 64 + lane;
        double acc = (double)s;
        if (MODE) {
            for (int q = w; q < nslot; q += 4) my[(long)q * 64] = acc + q;
            __threadfence();
            __syncthreads();
        }
        const int per = nout / 4;
        int q = w * 37;
        for (int e = 0; e < per; e += 8) {
            if (MODE) {
                double v[8];
                // reread * nslot loads per workgroup-tile in total, spread over the stores
#pragma unroll
                for (int u = 0; u < 8; ++u) { v[u] = my[(long)(q % nslot) * 64]; q += 53; }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double* p = out + (long)(w * per + e + u) * n + s;
                if (NT) __builtin_nontemporal_store(acc + u, p); else *p = acc + u;
            }
        }
        if (MODE) __syncthreads();
    }
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 262144;
    const int nslot = 771, nout = 2752;
    double *out, *scr;
    CHK(hipMalloc(&out, sizeof(double) * n * nout));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int wgs : {256, 512}) {
        CHK(hipMalloc(&scr, sizeof(double) * (size_t)wgs * nslot * 64));
        auto time = [&](const char* name, auto launch, double bytes) {
            launch(); CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) launch(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
            float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
            printf("wgs %d %-44s %8.3f ms  out %7.1f GB/s  all-traffic-if-HBM %7.1f GB/s\n", wgs, name, ms,
                   8.0 * n * nout / ms / 1e6, bytes / ms / 1e6);
        };
        const double ob = 8.0 * n * nout, sb = 8.0 * n * (nslot + nout);   // writes + reads (nout loads, ~3.6x nslot)
        time("output only", [&] { hipLaunchKernelGGL((k_fused<0, 0>), dim3(wgs), dim3(256), 0, 0, scr, out, n, nslot, nout, 0); }, ob);
        time("output + scratch (write 771, read 2752)", [&] { hipLaunchKernelGGL((k_fused<1, 0>), dim3(wgs), dim3(256), 0, 0, scr, out, n, nslot, nout, 0); }, ob + sb);
        time("same, nontemporal output stores", [&] { hipLaunchKernelGGL((k_fused<1, 1>), dim3(wgs), dim3(256), 0, 0, scr, out, n, nslot, nout, 0); }, ob + sb);
        CHK(hipFree(scr));
    }
    return 0;
}
