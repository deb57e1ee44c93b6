// Microbenchmark of csrc/pj_lu.h: factor / fused solve of n random diagonally dominant NSP x NSP blocks.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -freciprocal-math [-DPJ_LU_BPERM=0|1|2] -o lu_bw lu_bw.hip; ./lu_bw [nsp] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pyjac_amd/csrc/pj_lu.h"

int main(int argc, char** argv)
{
    const int nsp = argc > 1 ? atoi(argv[1]) : 53;
    const long n = argc > 2 ? atol(argv[2]) : 1000000;
    const double gamma = argc > 3 ? atof(argv[3]) : 0.0;
    const long ne = (long)nsp * nsp;
    std::vector<double> h(ne * 4096);
    srand(1);
    for (long s = 0; s < 4096; ++s)
        for (int r = 0; r < nsp; ++r)
            for (int c = 0; c < nsp; ++c) h[s * ne + r + nsp * c] = (rand() / (double)RAND_MAX - 0.5) + (r == c ? 10.0 * nsp : 0.0);
    double *a, *lu, *b, *x; int* perm;
    hipMalloc((void**)&a, 8 * ne * n); hipMalloc((void**)&lu, 8 * ne * n); hipMalloc((void**)&b, 8 * nsp * n);
    hipMalloc((void**)&x, 8 * nsp * n); hipMalloc((void**)&perm, 4 * nsp * n);
    for (long s0 = 0; s0 < n; s0 += 4096) hipMemcpy(a + s0 * ne, h.data(), 8 * ne * (n - s0 < 4096 ? n - s0 : 4096), hipMemcpyHostToDevice);
    hipMemset(b, 0, 8 * nsp * n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode : {(int)pj::LU_FACTOR, (int)(pj::LU_FACTOR | pj::LU_SOLVE)}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            for (int it = 0; it < 5; ++it)
                pj::lu_launch(nsp, n, a, pj::LuLay{1, (long)nsp * nsp, 1, (long)nsp}, gamma, mode == pj::LU_FACTOR ? lu : nullptr, mode == pj::LU_FACTOR ? perm : nullptr, b, x, mode, 256, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("BPERM=%d nsp %d n %ld mode %d: %.3f ms\n", PJ_LU_BPERM, nsp, n, mode, ms / 5);
        }
    }
    return hipGetLastError() != hipSuccess;
}
