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
using std::exp; using std::log; using std::fmax; using std::pow; using std::floor;
#define PJ_DEV static inline
#define PJ_LDS_ADD(ptr, v) (*(ptr) += (v))
#define PJT_CONST
#define PJ_UNIFORM(x) (x)
#define PJT_FETCH_ALL 1
#include "../../pyjac_amd/csrc/pj_kernel.h"
#include "../../pyjac_amd/csrc/pj_tab.h"
#include "../../pyjac_amd/csrc/pj_tables.cpp"
#include "../../pyjac_amd/csrc/pj_tabprog.cpp"

using namespace pj;

template <int TS>
static void run_tiles(const DevMech& M, const Batch& B, int NT, int want_jac)
{
    std::vector<double> V((size_t)M.v.NSLOT * TS + 3 * M.nsp);
    for (int tid = 0; tid < NT; ++tid) stage_consts<TS>(M, V.data(), tid, NT);
    std::vector<Lane> L(NT);
    const long ntiles = (B.n + TS - 1) / TS;
    for (long t = 0; t < ntiles; ++t) {
        for (int tid = 0; tid < NT; ++tid) phase0a<TS>(M, B, V.data(), tid, NT, t, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase_zero_tile<TS>(M, V.data(), tid, NT);
        for (int tid = 0; tid < NT; ++tid) phase0b<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase0c_scale<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase2<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase_scatter<TS>(M, V.data(), tid, NT, !want_jac);
        for (int tid = 0; tid < NT; ++tid) phase_fin1<TS>(M, V.data(), tid, NT);
        for (int tid = 0; tid < NT; ++tid) phase_fin2a<TS>(M, B, V.data(), tid, NT, L[tid]);
        for (int tid = 0; tid < NT; ++tid) phase_fin2b<TS>(M, B, V.data(), tid, NT, L[tid]);
        if (want_jac) {
            for (int tid = 0; tid < NT; ++tid) phase_out_energy<TS>(M, B, V.data(), tid, NT, L[tid]);
            for (int tid = 0; tid < NT; ++tid) phase_out_block<TS>(M, B, V.data(), tid, NT, L[tid]);
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
    M.rti = P.rti.data(); M.rtd = P.rtd.data(); M.nrp = P.nrp;
    M.smap = P.smap.data(); M.ecol_ptr = P.ecol_ptr.data(); M.ecol = P.ecol.data();
    M.eff_sp = P.eff_sp.data(); M.eff_am1 = P.eff_am1.data(); M.kcg = P.kcg.data();
    M.plog = P.plog.data(); M.sri = P.sri.data(); M.cheb = P.cheb.data(); M.net_sp = P.net_sp.data(); M.net_nu = P.net_nu.data();
    M.sp_ptr = P.sp_ptr.data(); M.sp_rxn = P.sp_rxn.data(); M.sp_nu = P.sp_nu.data();
    M.gen_sp = P.gen_sp.data(); M.gen_nu = P.gen_nu.data();
    for (int t = 0; t < NUTAB_N; ++t) M.nutab[t] = P.nutab[t];
    Schedule S;
    if (!build_schedule(P, NT / 64, 64 / TS, S)) return -3;
    M.v = P.vm; M.nv = P.vm.NV;
    M.sched = S.codes.data();
    for (int w = 0; w < 16; ++w) { M.sched_off[w] = S.off[w]; M.sched_rounds[w] = S.rounds[w]; M.sched_rounds_dense[w] = S.rounds_dense[w]; }
    M.fin_tgt = S.fin_tgt.data(); M.fin_part = S.fin_part.data(); M.fin_cnt = S.fin_cnt.data(); M.nfin = (int)S.fin_tgt.size();
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


// k_tab / k_tab_fin (pj_tab.h) thread by thread: the stage of every thread of a workgroup, then its row blocks
// (what __syncthreads() separates on the GPU), then the energy-row kernel the same way.  info[0..5]: L, G, B,
// blocks, visits, LDS bytes of the program that build_tab_program chose for `lds_avail`.
extern "C" int emu_tab_run(const int32_t* I, long nI, const double* D, long nD, long n, const double* pres,
                           const double* y_soa, double* jac, int jac_aos, int sum_last, long lds_avail, int* info)
{
    Programs P;
    if (!build_programs(I, nI, D, nD, P)) return -1;
    TabProg T;
    if (!build_tab_program(P, (size_t)lds_avail, T)) return -2;
    if (info) { info[0] = T.L; info[1] = T.G; info[2] = T.B; info[3] = T.nblk; info[4] = T.nvisit; info[5] = (int)T.lds_bytes; }
    DevMech M;
    memset(&M, 0, sizeof(M));
    M.nsp = P.nsp; M.nrxn = P.nrxn; M.lastq_rxn = P.lastq_rxn; M.sum_last = sum_last;
    M.sp = P.sp.data(); M.ri = P.ri.data(); M.rd = P.rd.data();
    M.eff_sp = P.eff_sp.data(); M.eff_am1 = P.eff_am1.data(); M.kcg = P.kcg.data();
    M.plog = P.plog.data(); M.sri = P.sri.data(); M.cheb = P.cheb.data();
    M.gen_sp = P.gen_sp.data(); M.gen_nu = P.gen_nu.data();
    std::vector<double> scr((size_t)(P.nsp + 1) * n);
    TabDev X;
    X.L = T.L; X.G = T.G; X.B = T.B; X.ZERO = T.ZERO; X.TRASH = T.TRASH;
    X.I = T.I.data(); X.D = T.D.data();
    X.scr = scr.data(); X.scr_ld = n; X.dbg = 0;
    Batch B;
    memset(&B, 0, sizeof(B));
    B.n = n; B.pres = pres; B.y = y_soa; B.y_si = n; B.y_ss = 1; B.jac = jac;
    if (jac_aos) { B.j_si = 1; B.j_ss = (long)P.nsp * P.nsp; } else { B.j_si = n; B.j_ss = 1; }
    std::vector<double> lds(T.lds_bytes / 8 + 256 * TAB_RING_WORDS, std::nan(""));      // (a ring per emulated thread)
    std::vector<TabLane> Ln(256);
    for (long wg = 0; wg * T.L < n; ++wg) {
        for (int tid = 0; tid < 256; ++tid) tab_stage(M, X, B, lds.data(), tid, wg, Ln[tid]);
        for (int tid = 0; tid < 256; ++tid) tab_blocks(M, X, B, lds.data(), tid, Ln[tid]);
    }
    std::vector<double> lds2((size_t)(P.nsp + 28) * 64 + 16, std::nan(""));
    for (long wg = 0; wg * 64 < n; ++wg) {
        for (int tid = 0; tid < 256; ++tid) tab_fin_stage(M, X, B, lds2.data(), tid, wg);
        for (int tid = 0; tid < 256; ++tid) tab_fin_cols(M, X, B, lds2.data(), tid, wg);
    }
    return 0;
}
