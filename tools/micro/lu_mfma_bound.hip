// Microbenchmark: what would a blocked LU of 56 x 56 (padded 64 x 64) blocks with v_mfma_f64_16x16x4 trailing updates
// cost at best?  ("build the MFMA variant for one size and put its time next to k_lu<56>".)
//
// This is a LOWER BOUND of such a kernel, not a factorisation: one wavefront per block, the block in the MFMA accumulator
// layout (16 tiles of 16 x 16, 4 doubles per lane and tile: 64 doubles per lane), and per panel of 4 columns
//   * the rank-4 updates of the trailing tiles as real v_mfma_f64_16x16x4 instructions on real operands
//     (4 row tiles x the column tiles right of the panel: 112 MFMAs per block),
//   * the operand preparation a real kernel cannot avoid, as real cross-lane instructions: the multipliers of a row tile
//     re-laid as an A operand (row l % 16, k = l / 16: one ds_bpermute pair per k from the accumulator layout), the four
//     pivot rows' entries of a column tile re-laid as a B operand (a bpermute pair + a register select per k),
//   * the panel factorisation as a PROXY: PANEL dependent-ish VALU instructions per column (pivot search over 4 lanes x
//     16 registers, implicit row exchange, scaling, the rank-1 updates inside the panel) -- PANEL = 80 is DESIGN.md's
//     count; 40 is given as an optimistic variant.
// Loads and stores of the blocks are the real ones (per-state layout).  Prints ms per 1e6 blocks; k_lu<56> of the
// product takes 15.8 ms (profiles/r04_lu_probe.txt: 53 x 53, factor).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bperm(const double v, const int byte_addr)
{
    const long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int PANEL, int WAVES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) k_bound(const double* A, double* LU, long n)
{
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n) return;
    constexpr int N = 56;
    d4 t[4][4];          // tile (ti, tj): rows 16 ti + (lane >> 4) + 4 r, column 16 tj + (lane & 15)
    const double* a = A + s * N * N;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                t[ti][tj][r] = (row < N && col < N) ? a[row + N * col] : (row == col ? 1.0 : 0.0);
            }
    double px = 1.0 + 1e-9 * lane;
#pragma unroll
    for (int p = 0; p < N / 4; ++p) {
        const int tp = p / 4;                       // column tile of the panel
        // panel factorisation proxy: PANEL instructions per column on the panel's registers (dependent chain + cross-lane)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int q = 0; q < PANEL / 8; ++q) {
                px = __builtin_fma(px, 0.999999, t[q & 3][tp][c & 3]);
                px = fmax(px, bperm(px, ((lane + 16) & 63) * 4));          // (3 instructions: 2 bpermute + max)
                t[q & 3][tp][(c + 1) & 3] = __builtin_fma(-px, 1e-9, t[q & 3][tp][(c + 1) & 3]);
                px = __builtin_fma(px, 1.000001, -1e-7);
                px = px * 0.5 + 0.25;
            }
        }
        // operands: L of every row tile (A operand: row l % 16, k = l / 16), U of every column tile right of the panel
        double La[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            // multiplier of row (l % 16) for panel column k = l / 16: sits in lane ((l % 16) % 4) * 16 + 4 p % 16 + k, register (l % 16) / 4
            const int src = (((lane & 15) & 3) * 16 + ((4 * p) & 15) + (lane >> 4)) * 4;
            const int rsel = (lane & 15) >> 2;
            const double v0 = bperm(t[ti][tp][0], src), v1 = bperm(t[ti][tp][1], src), v2 = bperm(t[ti][tp][2], src), v3 = bperm(t[ti][tp][3], src);
            La[ti] = rsel == 0 ? v0 : rsel == 1 ? v1 : rsel == 2 ? v2 : v3;
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            if (16 * tj + 15 <= 4 * p + 3) continue;       // (compile-time after unrolling: column tiles left of the panel)
            // pivot row k of the panel (a run-time row in a real kernel: here row 4 p + k): entry of column 16 tj + l % 16
            const int prow = 4 * p + (lane >> 4);
            const int src = ((prow & 3) * 16 + (lane & 15)) * 4;
            const int rsel = (prow >> 2) & 3;
            const d4& pt = t[(prow >> 4) & 3][tj];
            const double u0 = bperm(pt[0], src), u1 = bperm(pt[1], src), u2 = bperm(pt[2], src), u3 = bperm(pt[3], src);
            const double Ub = rsel == 0 ? u0 : rsel == 1 ? u1 : rsel == 2 ? u2 : u3;
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
                t[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-La[ti], Ub, t[ti][tj], 0, 0, 0);
        }
    }
    double* o = LU + s * N * N;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                if (row < N && col < N) o[row + N * col] = t[ti][tj][r] + (r == 0 && ti == 0 && tj == 0 ? px * 1e-300 : 0.0);
            }
}

int main(int argc, char** argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 1000000;
    double *a, *lu;
    CHK(hipMalloc(&a, sizeof(double) * n * 56 * 56));
    CHK(hipMalloc(&lu, sizeof(double) * n * 56 * 56));
    CHK(hipMemset(a, 0, sizeof(double) * n * 56 * 56));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* label) {
        const unsigned blocks = (unsigned)((n + 3) / 4);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, lu, n); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, lu, n);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.3f ms per %ld blocks\n", label, ms / 3, n);
    };
    printf("lower bound of a blocked 56 x 56 LU with v_mfma_f64_16x16x4 trailing updates (112 MFMAs + operand re-layout + panel proxy per block)\n");
    run(k_bound<80, 2>, "panel proxy 80 instr / column, 2 wavefronts per SIMD");
    run(k_bound<80, 3>, "panel proxy 80 instr / column, 3 wavefronts per SIMD");
    run(k_bound<40, 2>, "panel proxy 40 instr / column (optimistic), 2 wavefronts per SIMD");
    run(k_bound<40, 3>, "panel proxy 40 instr / column (optimistic), 3 wavefronts per SIMD");
    run(k_bound<0, 3>, "no panel work at all (loads, operand re-layout, MFMAs, stores), 3 per SIMD");
    printf("k_lu<54> of the product (53 x 53, real factorisation with partial pivoting): 15.8 ms per 1e6 blocks (profiles/r04_lu_probe.txt)\n");
    return 0;
}
