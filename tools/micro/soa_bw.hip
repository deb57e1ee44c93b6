// Microbenchmark: HBM bandwidth of the state-per-lane SoA access pattern (each wave stores /
// loads 512 B to thousands of streams that are n*8 bytes apart) against a plain stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NT, bool PERSIST>
__global__ void __launch_bounds__(256) k_store(double* out, long n, int E, long tile)
{
    for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < n; s += PERSIST ? (long)gridDim.x * 256 : n) {
        const double v = (double)s;
        // tile == 0: out[e*n + s];  tile > 0: [s/tile][e][tile]
        double* p = tile ? out + (s / tile) * tile * E + (s % tile) : out + s;
        const long st = tile ? tile : n;
        for (int e = 0; e < E; ++e) {
            if (NT) __builtin_nontemporal_store(v + e, p + (long)e * st);
            else p[(long)e * st] = v + e;
        }
    }
}
template <bool PERSIST>
__global__ void __launch_bounds__(256) k_load(const double* in, double* out, long n, int E, long tile)
{
    for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < n; s += PERSIST ? (long)gridDim.x * 256 : n) {
        const double* p = tile ? in + (s / tile) * tile * E + (s % tile) : in + s;
        const long st = tile ? tile : n;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int e = 0;
        for (; e + 16 <= E; e += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[(long)(e + u) * st];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
        }
        out[s] = a0 + a1 + a2 + a3;
    }
}
// the row-kernel pattern: per block, RB loads from a tiled scratch array, then WB SoA stores
template <int RB, int WB, int LDSKB = 0>
__global__ void __launch_bounds__(256) k_mixed(const double* scr, double* out, long n, int nblk, int nslot)
{
    __shared__ double pad[LDSKB * 128 + 1];
    if (n < 0) pad[threadIdx.x] = 1.0;   // keep the allocation
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const double* p = scr + (s / 256) * 256 * nslot + (s % 256);
    double* o = out + s;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) {
        double v[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) v[u] = p[(long)((b * 37 + u * 11) % nslot) * 256];
#pragma unroll
        for (int u = 0; u < RB; ++u) acc += v[u];
#pragma unroll
        for (int u = 0; u < WB; ++u) o[(long)(b * WB + u) * n] = acc + u;
    }
}
__global__ void k_stream(double* out, long total)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) out[i] = (double)i;
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 262144;
    const int E = argc > 2 ? atoi(argv[2]) : 2809;
    double *buf, *o2;
    CHK(hipMalloc(&buf, sizeof(double) * n * E));
    CHK(hipMalloc(&o2, sizeof(double) * n));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto time = [&](const char* name, auto launch) {
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) launch(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, 8.0 * n * E / ms / 1e6);
    };
    const unsigned full = (unsigned)((n + 255) / 256);
    time("stream store (grid-stride, 1024 WG x 256)", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, buf, n * E); });
    time("soa store, 1 state/lane, all WGs", [&] { hipLaunchKernelGGL((k_store<0, false>), dim3(full), dim3(256), 0, 0, buf, n, E, 0L); });
    time("soa store nontemporal", [&] { hipLaunchKernelGGL((k_store<1, false>), dim3(full), dim3(256), 0, 0, buf, n, E, 0L); });
    time("soa store persistent 256 WGs (1 wave/SIMD)", [&] { hipLaunchKernelGGL((k_store<0, true>), dim3(256), dim3(256), 0, 0, buf, n, E, 0L); });
    time("soa store persistent 512 WGs", [&] { hipLaunchKernelGGL((k_store<0, true>), dim3(512), dim3(256), 0, 0, buf, n, E, 0L); });
    time("tiled(256) store, all WGs", [&] { hipLaunchKernelGGL((k_store<0, false>), dim3(full), dim3(256), 0, 0, buf, n, E, 256L); });
    time("tiled(256) store persistent 256 WGs", [&] { hipLaunchKernelGGL((k_store<0, true>), dim3(256), dim3(256), 0, 0, buf, n, E, 256L); });
    time("tiled(4096) store, all WGs", [&] { hipLaunchKernelGGL((k_store<0, false>), dim3(full), dim3(256), 0, 0, buf, n, E, 4096L); });
    time("soa load, all WGs", [&] { hipLaunchKernelGGL((k_load<false>), dim3(full), dim3(256), 0, 0, buf, o2, n, E, 0L); });
    time("soa load persistent 256 WGs", [&] { hipLaunchKernelGGL((k_load<true>), dim3(256), dim3(256), 0, 0, buf, o2, n, E, 0L); });
    time("tiled(256) load, all WGs", [&] { hipLaunchKernelGGL((k_load<false>), dim3(full), dim3(256), 0, 0, buf, o2, n, E, 256L); });
    time("tiled(256) load persistent 256 WGs", [&] { hipLaunchKernelGGL((k_load<true>), dim3(256), dim3(256), 0, 0, buf, o2, n, E, 256L); });
    {
        const int nslot = 771, nblk = 46;
        double* scr; CHK(hipMalloc(&scr, sizeof(double) * n * nslot));
        CHK(hipMemset(scr, 0, sizeof(double) * n * nslot));
        hipLaunchKernelGGL((k_mixed<60, 60>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mixed<60, 60>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        printf("%-44s %8.3f ms  %7.1f GB/s (read+write)\n", "mixed 60 tiled loads + 60 soa stores x 46", ms, 8.0 * n * nblk * 120 / ms / 1e6);
        hipLaunchKernelGGL((k_mixed<24, 60>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mixed<24, 60>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        printf("%-44s %8.3f ms  %7.1f GB/s (read+write)\n", "mixed 24 tiled loads + 60 soa stores x 46", ms, 8.0 * n * nblk * 84 / ms / 1e6);
    }
    {
        const int nslot = 771, nblk = 46;
        double* scr; CHK(hipMalloc(&scr, sizeof(double) * n * nslot));
        float ms;
        hipLaunchKernelGGL((k_mixed<24, 60, 100>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mixed<24, 60, 100>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        printf("%-44s %8.3f ms  %7.1f GB/s (read+write)\n", "mixed 24+60, 1 WG/CU (100 KB LDS)", ms, 8.0 * n * nblk * 84 / ms / 1e6);
        hipLaunchKernelGGL((k_mixed<24, 60, 60>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a)); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_mixed<24, 60, 60>), dim3(full), dim3(256), 0, 0, scr, buf, n, nblk, nslot);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        CHK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        printf("%-44s %8.3f ms  %7.1f GB/s (read+write)\n", "mixed 24+60, 2 WG/CU (60 KB LDS)", ms, 8.0 * n * nblk * 84 / ms / 1e6);
    }
    return 0;
}
