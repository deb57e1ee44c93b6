// pj_lu.hip -- the batched LU / Newton-solve kernels of pj_lu.h as translation units of their own, so that the C-ABI library
// builds in parallel (k_lu4 is four instantiations of about two minutes each; inside pj_api.hip they made that file an
// eight-minute compile -- VERDICT round 5, item 6).
//   -DPJ_LU_PART=0     k_lu16 / k_lu / k_lu_lds and the dispatcher (pj::lu_launch_x, what pj_api.hip calls)
//   -DPJ_LU_PART=80 | 96 | 112 | 128   one k_lu4 instantiation and its launcher
#ifndef PJ_LU_PART
#define PJ_LU_PART 0
#endif
#if PJ_LU_PART == 0
#define PJ_LU4_SPLIT 1
#include "pj_lu.h"
namespace pj {
int lu_launch_x(int nsp, long n, const double* A, LuLay Y, double gamma, double* lu, int* perm, const double* b, double* x,
                int mode, int cus, hipStream_t st)
{
    return lu_launch(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st);
}
}
#else
#define PJ_LU4_ONLY PJ_LU_PART
#include "pj_lu.h"
#endif
