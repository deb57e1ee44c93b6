// hip_shim.h -- just enough of the HIP host/device surface to compile the state-per-lane
// kernels (pj_lane.hip, pj_rblk.hip) with g++ and run them one "thread" per workgroup on the
// CPU -- or, for the kernels whose workgroup is several lane groups on the same states (pj_rblk.hip, PJQ_HALVES > 1:
// -DPJQ_BLOCK=1 makes a group one lane), one OS thread per lane with a real barrier behind __syncthreads().  Test infrastructure only: lets the CPU suite check the kernels' arithmetic and
// indexing against the oracle without a GPU.  Build with -DPJR_BLOCK=1 / -DPJL_BLOCK=1.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <pthread.h>
#include <thread>
#include <vector>

#define __HIPCC__ 1
#define __global__
#define __device__
#define __host__
#define __constant__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
static thread_local dim3 threadIdx(0), blockIdx(0), blockDim(1), gridDim(1);
static pthread_barrier_t* hip_shim_barrier = nullptr;      // set while a multi-thread workgroup runs
static inline void __syncthreads() { if (hip_shim_barrier) pthread_barrier_wait(hip_shim_barrier); }
inline void __threadfence() {}
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipDeviceAttributeMultiprocessorCount = 0 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 4; return hipSuccess; }
template <class K>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* v, K, int, int) { *v = 1; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { static int dummy; *s = &dummy; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, int) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return hipSuccess; }
using std::exp; using std::log; using std::fmax; using std::pow; using std::floor;

// clang / AMDGPU builtins used by the kernels
#define __builtin_nontemporal_store(val, ptr) (*(ptr) = (val))
#define __builtin_nontemporal_load(ptr) (*(ptr))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)

// blocks run one after the other; a block of one thread runs inline, a block of several threads as that many OS
// threads (function-local `static` = __shared__ arrays are shared by them, threadIdx is thread-local)
// (static, not inline: threadIdx / blockIdx are per translation unit, and an inline function template would be merged
// across the translation units of a library -- the survivor would set ITS unit's blockIdx only)
template <class K, class... Args>
static void hip_shim_launch(K kernel, dim3 grid, dim3 block, Args... args)
{
    if (block.x > 64) abort();
    if (block.x == 1) {
        gridDim = grid; blockDim = block; threadIdx = dim3(0);
        for (unsigned b = 0; b < grid.x; ++b) { blockIdx = dim3(b); kernel(args...); }
        return;
    }
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, block.x);
    hip_shim_barrier = &bar;
    for (unsigned b = 0; b < grid.x; ++b) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < block.x; ++t)
            th.emplace_back([=]() {
                gridDim = grid; blockDim = block; blockIdx = dim3(b); threadIdx = dim3(t);
                kernel(args...);
            });
        for (auto& x : th) x.join();
    }
    hip_shim_barrier = nullptr;
    pthread_barrier_destroy(&bar);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hip_shim_launch(kernel, grid, block, __VA_ARGS__)
