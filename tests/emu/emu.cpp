// emu.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the device phases of pyjac_amd/csrc/pj_kernel.h thread-by-thread on the
// host (each phase for every thread of a workgroup, phases in kernel order =
// what __syncthreads() enforces on the GPU), so indexing / scheduling logic can
// be checked against the oracle in the GPU-less container.  Never shipped,
// never imported by the product package; the product path has no CPU fallback.
#include <cmath>
#include <cstring>
#include <vector>
struct uint2 { unsigned x, y; };
using std::exp; using std::log; using std::fmax;
#define PJ_DEV static inline
#define PJ_WAVE_SYNC() ((void)0)
#include "../../pyjac_amd/csrc/pj_kernel.h"
#include "../../pyjac_amd/csrc/pj_tables.cpp"

using namespace pj;

template <int TS>
static void run_tiles(const DevMech& M, const Batch& B, int NT, int want_jac)
{
    std::vector<double> V((size_t)M.nv * TS + NT + (M.prog_words + 1) / 2 + 1);
    for (int tid = 0; tid < NT; ++tid) stage_prog<TS>(M, V.data(), tid, NT);
    std::vector<Lane> L(NT);
    const long ntiles = (B.n + TS - 1) / TS;
    for (long t = 0; t < ntiles; ++t) {
        for (int tid = 0; tid < NT; ++tid) phase0a<TS>(M, B, V.data(), tid, NT, t, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase0b<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase0c_scale<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase2<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase3<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase3c<TS>(M, B, V.data(), tid, NT, L[tid]);
        if (want_jac) {
            const int nr = phase4_rounds<TS>(M, NT);
            for (int r = 0; r < nr; ++r) {
                for (int tid = 0; tid < NT; ++tid) phase4a<TS>(M, B, V.data(), tid, NT, L[tid], r);
                for (int tid = 0; tid < NT; ++tid) phase4b<TS>(M, B, V.data(), tid, NT, L[tid], r);
            }
        }
    }
}

extern "C" int emu_run(const int32_t* I, long nI, const double* D, long nD, long n,
                       const double* pres, const double* y_soa, double* jac, int jac_aos,
                       double* conc, double* fwd, double* rev, double* pres_mod,
                       double* spec_rates, double* dy, int TS, int NT, int sum_last)
{
    Programs P;
    if (!build_programs(I, nI, D, nD, P)) return -1;
    DevMech M;
    M.nsp = P.nsp; M.nrxn = P.nrxn; M.ng = P.ng; M.ne = P.ne; M.nv = P.vm.NV;
    M.lastq_rxn = P.lastq_rxn; M.sum_last = sum_last; M.v = P.vm;
    M.sp = P.sp.data(); M.ri = P.ri.data(); M.rd = P.rd.data();
    M.eff_sp = P.eff_sp.data(); M.eff_am1 = P.eff_am1.data(); M.kcg = P.kcg.data();
    M.plog = P.plog.data(); M.net_sp = P.net_sp.data(); M.net_nu = P.net_nu.data();
    M.sp_ptr = P.sp_ptr.data(); M.sp_rxn = P.sp_rxn.data(); M.sp_nu = P.sp_nu.data();
    M.prog = P.prog.data(); M.prog_words = (int)P.prog.size();
    M.p4en = P.p4en; M.p4c = P.p4c; M.p3en = P.p3en; M.p3c = P.p3c; M.prog_in_lds = (TS % 2 == 0);
    Batch B;
    B.n = n; B.pres = pres; B.y = y_soa; B.y_si = n; B.y_ss = 1;
    B.jac = jac;
    if (jac_aos) { B.j_si = 1; B.j_ss = (long)P.nsp * P.nsp; } else { B.j_si = n; B.j_ss = 1; }
    B.conc = conc; B.fwd = fwd; B.rev = rev; B.pres_mod = pres_mod; B.spec_rates = spec_rates; B.dy = dy;
    B.o_ld = n;
    switch (TS) {
        case 64: run_tiles<64>(M, B, NT, jac != nullptr); break;
        case 16: run_tiles<16>(M, B, NT, jac != nullptr); break;
        case 4: run_tiles<4>(M, B, NT, jac != nullptr); break;
        case 1: run_tiles<1>(M, B, NT, jac != nullptr); break;
        default: return -2;
    }
    return 0;
}
