// pj_rows.hip -- state-per-lane Jacobian kernels for MEDIUM / LARGE mechanisms.
//
// pj_lane.hip keeps a whole Jacobian in the registers of one lane; that stops at ~15 species.
// Here the same formulation is cut into kernels whose live set fits the 512 registers a lane
// owns at one wavefront per SIMD, all with one thermochemical state per lane and SoA-coalesced
// memory traffic:
//
//   k_rates<R0,R1>   reactions [R0,R1): Arrhenius / PLOG, K_c, third-body / falloff / Troe factor
//                    and the per-reaction derivative scalars.  The d/dT column (sum nu theta) is
//                    finished here; what the row kernels need goes to an HBM scratch array
//                    scr[slot][state]: c*k_f, c*k_r and, for pressure-dependent reactions,
//                    rp, b_M, b_col (~2.4 doubles per reaction)
//   k_rows<B0,B1>    row blocks [B0,B1) of the Jacobian: a block is a group of species rows whose
//                    accumulators (omega_k, P_k, Q_k and the structurally non-zero
//                    S_kj of those rows) fit the register budget; it re-reads the scratch values
//                    of every reaction that touches one of its rows, rebuilds the cheap
//                    concentration products, accumulates with compile-time register indices and
//                    stores its rows of the Jacobian.  Concentrations sit in LDS (one column per
//                    lane); energy-row partial sums are added to row 0 in memory at the end.
//   k_fin            turns the raw energy-row sums into the d(dT/dt)/d. row.
//
// The mechanism is injected as constexpr tables (pj::emit_spec_header + pj::emit_rows_tables ->
// PJS_HEADER); every loop is a compile-time loop.  The kernels of one library are compiled as
// separate translation units (PJR_PART) so that a 53-species mechanism builds in parallel.
//
// Same formulation as pj_lane.hip / pj_kernel.h; reference emitters:
// pyjac/core/rate_subs.py:254-2335, pyjac/core/create_jacobian.py:2189-3298.
//
// PJR_PART = 0: host entry points + k_fin;  PJR_PART = 1: k_rates<PJR_R0,PJR_R1>;
// PJR_PART = 2: k_rows<PJR_B0,PJR_B1>.  PJR_ID is the launch-order index of the part.
// PJR_PART = 3: the fused single-kernel variant (k_fused, one translation unit, own host entry).
#ifdef PJR_HOST_EMU
#include "hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "pj_tables.h"
#include PJS_HEADER

using namespace pj;

#ifndef PJR_BLOCK
#define PJR_BLOCK 256
#endif
#ifndef PJR_DEPTH
#define PJR_DEPTH 16       // visits whose scratch values are in flight
#endif
#ifndef PJR_C_LDS
#define PJR_C_LDS 0         // rate kernels: concentrations in LDS (set for large mechanisms)
#endif
#ifndef PJR_RECOMPUTE_KR
#define PJR_RECOMPUTE_KR 0   // 1: row kernels rebuild c*k_r = c*k_f * exp(-ln K_c(T)) instead of reading it back
#endif
#ifndef PJR_RECOMPUTE_KF
#define PJR_RECOMPUTE_KF 0   // 1: ... and k_f of plain Arrhenius reactions (no third body / falloff / PLOG) as well
#endif
#define PJR_TICK_B(ph)      // phase timing hook of debug builds (fused kernel, -DPJR_TIMING)
#define PJR_TILE 256        // states per scratch tile
// Jacobian entries are written once and never read back by these kernels
#if defined(PJR_NT_STORE) && PJR_NT_STORE && !defined(PJR_HOST_EMU)
#define PJR_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define PJR_STORE(ptr, val) (*(ptr) = (val))
#endif
#ifdef PJR_HOST_EMU
#define PJR_SCHED_BARRIER()
#else
#define PJR_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

struct PjrArgs {
    long n;                            // states of this chunk
    const double* pres;                // chunk base
    const double* y; long y_si, y_ss;  // chunk base
    double* jac; long j_si, j_ss;      // chunk base
    double* scr; long ld;              // scratch [ld / PJR_TILE][NSCR + 3][PJR_TILE]
    int sum_last;
    // rate-output launches (k_rates<true>, k_dy): SoA, leading dimension o_ld; sr is never null
    double *conc, *fwd, *rev, *pres_mod, *sr, *dy; long o_ld, sr_ld;   // sr has its own leading dimension
};
typedef void (*pjr_launch_fn)(const PjrArgs&, void* stream);
extern "C" void pjr_register(int id, int kind, pjr_launch_fn fn);

namespace {

constexpr double RU_ = 8314.4621;
constexpr double INV_LN10 = 0.434294481903251828;
constexpr int NSP = pjs::NSP, NRXN = pjs::NRXN, LAST = pjs::NSP - 1, ONE = pjs::NSP;
constexpr int S_TH = 0, S_KF = 1, S_KR = 2, S_RP = 3, S_BM = 4, S_BC = 5;
constexpr int SUM_H = pjs::NSCR, SUM_SCP = pjs::NSCR + 1, SUM_SJT = pjs::NSCR + 2;

// plain Arrhenius reaction: c = 1 and k_f = sgn * exp(ln A + b ln T - Ta / T) is a function of T alone
constexpr bool kf_plain(int i) { return (pjs::RI[i][RI_FLAGS] & (F_THD | F_PDEP | F_PLOG | F_CHEB)) == 0; }

#define PJR_INL __attribute__((always_inline))
#define PJR_SLOT(i_, c_) pjs::SCR[i_][c_]      // hand-over slots of pj_rows_rate.inc
// compile-time loop: f(std::integral_constant<int, I0 + i>) for i = 0..N-1 (flat fold, no recursion)
template <int I0, class F, int... Is>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, I0 + Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (N > 0) static_for_seq<0>(f, std::make_integer_sequence<int, N>{});
}
template <int I0, int I1, class F>
__device__ __forceinline__ void static_range(F&& f)
{
    if constexpr (I1 > I0) static_for_seq<I0>(f, std::make_integer_sequence<int, I1 - I0>{});
}

// scratch layout [state tile][slot][PJR_TILE]: everything one workgroup reads and writes is one
// contiguous (NSCR + 3) * 2 KB region (DRAM-page and TLB locality), each wave access is 512 B
#ifndef PJR_SCR_TILED
#define PJR_SCR_TILED 1
#endif
#if PJR_SCR_TILED
#define PJR_SSTRIDE(A) PJR_TILE
__device__ __forceinline__ double* scr_of(const PjrArgs& A, long s)
{
    return A.scr + (s / PJR_TILE) * ((long)(pjs::NSCR + 3) * PJR_TILE) + (s % PJR_TILE);
}
#else
#define PJR_SSTRIDE(A) (A).ld
__device__ __forceinline__ double* scr_of(const PjrArgs& A, long s) { return A.scr + s; }
#endif

// per-state scalars and concentrations shared by the kernels
struct State {
    double T, p, Wbar, rho, invrho, mconc;
    double C[NSP + 1];
};

__device__ __forceinline__ void load_state(const PjrArgs& A, long s, State& L)
{
    const double* y = A.y + s * A.y_ss;
    L.T = y[0];
    L.p = A.pres[s];
    double sumY = 0.0, sumYW = 0.0;
    static_for<LAST>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        L.C[k] = y[(k + 1) * A.y_si];
        sumY += L.C[k];
        sumYW += L.C[k] * pjs::SP[k][0];
    });
    const double yN = 1.0 - sumY;
    L.C[LAST] = yN;
    sumYW += yN * pjs::SP[LAST][0];
    L.Wbar = 1.0 / sumYW;
    L.rho = L.p * L.Wbar / (RU_ * L.T);
    L.invrho = 1.0 / L.rho;
    L.mconc = L.p / (RU_ * L.T);
}
// mass fractions -> concentrations (after the mass-fraction weighted sums were taken)
__device__ __forceinline__ void to_conc(State& L)
{
    static_for<NSP>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        L.C[k] = L.rho * L.C[k] * pjs::SP[k][0];
    });
    L.C[ONE] = 1.0;
}

#if PJR_PART == 1
// ------------------------------------------------------------------------------------------
// k_rates<R0,R1>
// ------------------------------------------------------------------------------------------
constexpr int R0_ = PJR_R0, R1_ = PJR_R1;
constexpr int kc_lo()
{
    int lo = 1 << 30;
    for (int i = R0_; i < R1_; ++i)
        if ((pjs::RI[i][RI_FLAGS] & F_REV) && pjs::RI[i][RI_KC_CNT] > 0 && pjs::RI[i][RI_KC_PTR] < lo)
            lo = pjs::RI[i][RI_KC_PTR];
    return lo == (1 << 30) ? 0 : lo;
}
constexpr int kc_hi()
{
    int hi = 0;
    for (int i = R0_; i < R1_; ++i)
        if ((pjs::RI[i][RI_FLAGS] & F_REV) && pjs::RI[i][RI_KC_PTR] + pjs::RI[i][RI_KC_CNT] > hi)
            hi = pjs::RI[i][RI_KC_PTR] + pjs::RI[i][RI_KC_CNT];
    return hi < kc_lo() ? kc_lo() : hi;
}
constexpr int KC_LO = kc_lo(), KC_N = kc_hi() - kc_lo();
// first reaction of [R0,R1) that uses K_c class c evaluates it
constexpr bool kc_first_in_range(int i)
{
    for (int h = R0_; h < i; ++h)
        if ((pjs::RI[h][RI_FLAGS] & F_REV) && pjs::KC_CLASS[h][0] == pjs::KC_CLASS[i][0]) return false;
    return true;
}
constexpr int NEFF = (int)(sizeof(pjs::EFF_AM1) / sizeof(pjs::EFF_AM1[0]));

// RATES_OUT = false: hand-over to the row kernels (scratch array, d/dT column);
// RATES_OUT = true: conc, fwd, rev, pres_mod and this range's share of spec_rates (pj_eval_rates_dev)
template <bool RATES_OUT>
__global__ void __launch_bounds__(PJR_BLOCK) k_rates(PjrArgs A0)
{
    // real-valued coefficient tables of this reaction range, staged once per workgroup and
    // read with uniform ds_reads (as 64-bit literals they would be hoisted and spilled)
    __shared__ __attribute__((aligned(16))) double LT[(KC_N > 0 ? KC_N : 1) * 16];
    __shared__ __attribute__((aligned(16))) double RDL[R1_ - R0_][RDW];
    __shared__ __attribute__((aligned(16))) double EFL[NEFF];
#if PJR_C_LDS
    __shared__ double CLr[NSP][PJR_BLOCK];
#endif
    for (int w = threadIdx.x; w < KC_N * 16; w += PJR_BLOCK) LT[w] = pjs::LTAB[pjs::LT_KC + KC_LO * 16 + w];
    for (int w = threadIdx.x; w < (R1_ - R0_) * RDW; w += PJR_BLOCK) (&RDL[0][0])[w] = (&pjs::RDT[R0_][0])[w];
    for (int w = threadIdx.x; w < NEFF; w += PJR_BLOCK) EFL[w] = pjs::EFFT[w][0];
    __syncthreads();
    for (long s = (long)blockIdx.x * PJR_BLOCK + threadIdx.x; s < A0.n; s += (long)gridDim.x * PJR_BLOCK) {
        // The strides are the same every iteration; hide that from the optimiser, which would otherwise
        // hoist every entry offset e * j_si (and slot * ld ...) out of this loop as SGPR pairs and spill
        // them (SGPR -> VGPR lanes -> scratch).
        PjrArgs A = A0;
#ifndef PJR_HOST_EMU
        asm volatile("" : "+s"(A.j_si), "+s"(A.y_si), "+s"(A.o_ld), "+s"(A.sr_ld), "+s"(A.ld));
#endif
#if PJR_C_LDS
        // large mechanisms: concentrations in LDS (one column per lane) instead of 2*NSP VGPRs
        double T, p, invrho, Wbar, mconc;
        {
            State L;
            load_state(A, s, L);
            to_conc(L);
            T = L.T; p = L.p; invrho = L.invrho; Wbar = L.Wbar; mconc = L.mconc;
            static_for<NSP>([&](auto kc) PJR_INL { CLr[decltype(kc)::value][threadIdx.x] = L.C[decltype(kc)::value]; });
        }
#define CC(idx) ((idx) == ONE ? 1.0 : CLr[(idx) == ONE ? 0 : (idx)][threadIdx.x])
#else
        State L;
        load_state(A, s, L);
        to_conc(L);
        const double T = L.T, p = L.p;
        const double invrho = L.invrho, Wbar = L.Wbar, mconc = L.mconc;
#define CC(idx) L.C[idx]
#endif
        const double logT = log(T), invT = 1.0 / T, logp = log(p);
        double* const scr = scr_of(A, s);
#define SCR_(slot) scr[(long)(slot) * PJR_SSTRIDE(A)]
// written once per state, read back by later kernels from HBM: nontemporal unless PJR_SCR_NT=0
#if defined(PJR_HOST_EMU) || (defined(PJR_SCR_NT) && !PJR_SCR_NT)
#define SCR_ST(slot, val) do { if constexpr (!RATES_OUT) SCR_(slot) = (val); } while (0)
#else
#define SCR_ST(slot, val) do { if constexpr (!RATES_OUT) __builtin_nontemporal_store((val), &SCR_(slot)); } while (0)
#endif
        double* const Jl = RATES_OUT ? nullptr : A.jac + s * A.j_ss;
#define J_(e) Jl[(long)(e) * A.j_si]
        auto rate_out = [&](auto ic, const double Rf, const double Rr, const double c) PJR_INL {
            constexpr int i = decltype(ic)::value;
            // rate_subs.py:634-658, 811-840, 1076-1283: indices are positions in the mechanism file
            if (A.fwd) PJR_STORE(&A.fwd[pjs::RI[i][RI_ORIG] * A.o_ld + s], Rf);
            if constexpr (pjs::RI[i][RI_REV_IDX] >= 0) { if (A.rev) PJR_STORE(&A.rev[pjs::RI[i][RI_REV_IDX] * A.o_ld + s], Rr); }
            if constexpr (pjs::RI[i][RI_PRES_IDX] >= 0) {
                if (A.pres_mod) PJR_STORE(&A.pres_mod[pjs::RI[i][RI_PRES_IDX] * A.o_ld + s], c);
            }
        };
        if constexpr (!RATES_OUT && R0_ == 0) {
            // this part also clears what the row kernels accumulate into
            SCR_(SUM_H) = 0.0; SCR_(SUM_SCP) = 0.0;
            static_for<LAST>([&](auto jc) PJR_INL { J_(NSP * (decltype(jc)::value + 1)) = 0.0; });
        }
        // With few unconditional global stores in the loop body the optimiser treats the LDS tables as
        // loop invariant and hoists hundreds of coefficient reads out of the persistent loop (spills).
        // An opaque zero offset per iteration stops that and, unlike laundering the pointers
        // themselves, keeps them recognisable as LDS addresses (ds_read, not flat_load).
        unsigned zoff = 0;
#ifndef PJR_HOST_EMU
        if constexpr (RATES_OUT) asm volatile("" : "+s"(zoff));
#endif
        const double (*rdl)[RDW] = (const double (*)[RDW])((const char*)RDL + zoff);
        const double* lt = (const double*)((const char*)LT + zoff);
        const double* efl = (const double*)((const char*)EFL + zoff);
        double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
        // d/dT column: sum_q nu_kq theta_q needs nothing but theta, so it is finished here and
        // theta never goes through the scratch array
        double jt[NSP], jtq = 0.0;
        static_for<NSP>([&](auto kc) PJR_INL { jt[decltype(kc)::value] = 0.0; });
        static_range<R0_, R1_>([&](auto ic) PJR_INL {
            constexpr int i = decltype(ic)::value;
#define PJR_RD(i_) rdl[(i_) - R0_]
#define PJR_KCROW(g_) (lt + ((g_) - KC_LO) * 16)
#define PJR_EFL(e_) efl[e_]
#define PJR_KC_FIRST(i_) kc_first_in_range(i_)
#include "pj_rows_rate.inc"
#undef PJR_RD
#undef PJR_KCROW
#undef PJR_EFL
#undef PJR_KC_FIRST
        });
        if constexpr (RATES_OUT) {
            // eval_conc output and this range's share of omega_k (rate_subs.py:1297-1542)
            static_for<NSP>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                if constexpr (R0_ == 0) { if (A.conc) PJR_STORE(&A.conc[k * A.o_ld + s], CC(k)); }
                if constexpr (R0_ == 0) A.sr[k * A.sr_ld + s] = jt[k];
                else A.sr[k * A.sr_ld + s] += jt[k];
            });
            continue;
        }
        // reference quirk (create_jacobian.py:2786-2818): the last species keeps only the d/dT
        // term of one reaction unless sum_last is set (see pj_kernel.h)
        if (!A.sum_last) jt[LAST] = (pjs::LASTQ >= R0_ && pjs::LASTQ < R1_) ? jtq : 0.0;
        double sjt = 0.0;
        static_for<NSP>([&](auto kc) PJR_INL {
            constexpr int k = decltype(kc)::value;
            const bool lo = T <= pjs::SP[k][2];
            double a[6];
            static_for<6>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value;
                a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
            });
            const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                     T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
            sjt += hW * jt[k];
            if constexpr (k < LAST) {
                if constexpr (R0_ == 0) J_(k + 1) = pjs::SP[k][1] * jt[k];
                else J_(k + 1) += pjs::SP[k][1] * jt[k];
            }
        });
        if constexpr (R0_ == 0) SCR_(SUM_SJT) = sjt; else SCR_(SUM_SJT) += sjt;
#undef J_
#undef SCR_
#undef SCR_ST
#undef CC
    }
}

#ifndef PJR_RATES_LIB
void launch_part(const PjrArgs& A, void* stream)
{
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rates<false>, PJR_BLOCK, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    long blocks = (A.n + PJR_BLOCK - 1) / PJR_BLOCK;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(k_rates<false>, dim3((unsigned)blocks), dim3(PJR_BLOCK), 0, (hipStream_t)stream, A);
}
#endif
void launch_part_out(const PjrArgs& A, void* stream)
{
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rates<true>, PJR_BLOCK, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    long blocks = (A.n + PJR_BLOCK - 1) / PJR_BLOCK;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(k_rates<true>, dim3((unsigned)blocks), dim3(PJR_BLOCK), 0, (hipStream_t)stream, A);
}
#ifdef PJR_RATES_LIB   // rate outputs only (the Jacobian kernels of the library are pj_rblk.hip's)
struct Reg { Reg() { pjr_register(PJR_ID, 3, launch_part_out); } } reg_;
#else
struct Reg { Reg() { pjr_register(PJR_ID, 1, launch_part); pjr_register(PJR_ID, 3, launch_part_out); } } reg_;
#endif
#endif  // PJR_PART == 1

#if PJR_PART == 2
// ------------------------------------------------------------------------------------------
// k_rows<B0,B1>
// ------------------------------------------------------------------------------------------
constexpr int B0_ = PJR_B0, B1_ = PJR_B1;

// does reaction i carry an enhanced efficiency of the last species?
template <int i>
constexpr bool has_anm1() { return pjs::RD[i][RD_ANM1] != 0.0; }

// ln K_c of reaction i from the pre-summed NASA polynomials of its groups (rate_subs.py:660-809)
template <int i>
__device__ __forceinline__ double kc_ln(const double* lt, double T, double logT, double invT)
{
    double lnKc = pjs::RD[i][RD_LNPREF];
    static_for<pjs::RI[i][RI_KC_CNT]>([&](auto cc) PJR_INL {
        constexpr int g = pjs::RI[i][RI_KC_PTR] + decltype(cc)::value;
        const double* a = lt + g * 16 + ((T <= pjs::KCG[g][0]) ? 0 : 8);
        lnKc += a[0] + a[1] * logT + T * (a[2] + T * (a[3] + T * (a[4] + a[5] * T))) - a[6] * invT;
    });
    return lnKc;
}

__global__ void __launch_bounds__(PJR_BLOCK) k_rows(PjrArgs A)
{
    // Concentrations live in LDS, one column per lane (bank-conflict free): the registers they
    // would occupy are worth more as landing space for scratch loads in flight -- at one
    // wavefront per SIMD the bytes in flight per lane bound the achieved HBM bandwidth.
    __shared__ double CL[NSP][PJR_BLOCK];
#if PJR_RECOMPUTE_KR
    // NASA row pairs of every K_c group: c*k_r is rebuilt from c*k_f here (one exp and a 7-term
    // polynomial per visit, in issue slots the memory-bound kernel leaves idle) instead of being
    // written by the rate kernel and read back ~3.6 times
    __shared__ __attribute__((aligned(16))) double LTK[(pjs::LT_SP / 16 > 0 ? pjs::LT_SP / 16 : 1) * 16];
    for (int x = threadIdx.x; x < pjs::LT_SP; x += PJR_BLOCK) LTK[x] = pjs::LTAB[pjs::LT_KC + x];
    __syncthreads();
#endif
    const long s = (long)blockIdx.x * PJR_BLOCK + threadIdx.x;
    if (s >= A.n) return;
    const int tid = threadIdx.x;
    double T, invrho, Wbar;
    {
        State L;
        load_state(A, s, L);
        to_conc(L);
        T = L.T; invrho = L.invrho; Wbar = L.Wbar;
        static_for<NSP>([&](auto kc) PJR_INL { CL[decltype(kc)::value][tid] = L.C[decltype(kc)::value]; });
    }
#if PJR_RECOMPUTE_KR
    const double logT = log(T), invT = 1.0 / T;
#define PJR_KR_FROM_KF(i_, ckf_) ((ckf_) * exp(-kc_ln<i_>(LTK, T, logT, invT)))
#if PJR_RECOMPUTE_KF
#define PJR_KF_PLAIN(i_) ((pjs::RD[i_][RD_SGN] < 0.0 ? -1.0 : 1.0) * \
                          exp(pjs::RD[i_][RD_LNA] + pjs::RD[i_][RD_B] * logT - pjs::RD[i_][RD_TA] * invT))
#endif
#endif
    auto conc = [&](auto spc) PJR_INL {
        constexpr int sp = decltype(spc)::value;
        if constexpr (sp == ONE) return 1.0; else return CL[sp][tid];
    };
    // energy-row partial sums: rarely touched, so the register allocator parks them in AGPRs
    double E[LAST > 0 ? LAST : 1];
    const double* const scr = scr_of(A, s);
#ifndef PJR_LD_NT
#define PJR_LD_NT 1        // scratch values are streamed through once per kernel: nontemporal loads (+1.5 %)
#endif
#if !defined(PJR_HOST_EMU) && PJR_LD_NT
#define LD_(slot) __builtin_nontemporal_load(&scr[(long)(slot) * PJR_SSTRIDE(A)])
#else
#define LD_(slot) scr[(long)(slot) * PJR_SSTRIDE(A)]
#endif
    double* const Jl = A.jac + s * A.j_ss;
#define J_(e) Jl[(long)(e) * A.j_si]
    static_for<LAST>([&](auto jc) PJR_INL { E[decltype(jc)::value] = 0.0; });
    double H = 0.0, SCP = 0.0;

    // scratch values are fetched PJR_DEPTH visits ahead into a register ring; the scheduling
    // barrier after every visit keeps the loads where they are issued (left alone, the
    // scheduler hoists all of a block's loads to its top and spills the accumulators).
    double ring[PJR_DEPTH][6];
    auto issue_bv = [&](auto bc, auto vc) PJR_INL {
        constexpr int v = decltype(vc)::value;
        constexpr int i = pjs::BLK_RX[pjs::BLK_RX_PTR[decltype(bc)::value][0] + v][0];
        static_for<6>([&](auto cc) PJR_INL {
            constexpr int c = decltype(cc)::value;
            if constexpr (pjs::SCR[i][c] >= 0 && !(PJR_RECOMPUTE_KR && c == S_KR) &&
                          !(PJR_RECOMPUTE_KF && c == S_KF && kf_plain(i)))
                ring[v % PJR_DEPTH][c] = LD_(pjs::SCR[i][c]);
        });
    };
    auto prologue = [&](auto bc) PJR_INL {
        constexpr int b = decltype(bc)::value;
        constexpr int nv = pjs::BLK_RX_PTR[b + 1][0] - pjs::BLK_RX_PTR[b][0];
        static_for<(nv < PJR_DEPTH ? nv : PJR_DEPTH)>([&](auto vc) PJR_INL { issue_bv(bc, vc); });
    };
#define PJR_SP(k_, c_) pjs::SP[k_][c_]
#define PJR_EFF(e_) pjs::EFF_AM1[e_][0]
#define PJR_NASA(k_, lo_, c_) ((lo_) ? pjs::SP[k_][4 + (c_)] : pjs::SP[k_][11 + (c_)])
#define PJR_ANM1(i_) pjs::RD[i_][RD_ANM1]
    static_range<B0_, B1_>([&](auto bc) PJR_INL {
        constexpr int b = decltype(bc)::value;
#include "pj_rows_block.inc"
    });

    // hand the partial sums to k_fin through memory (kernels of one batch run in stream order)
    double* const sw = scr_of(A, s);
    sw[(long)SUM_H * PJR_SSTRIDE(A)] += H;
    sw[(long)SUM_SCP * PJR_SSTRIDE(A)] += SCP;
    static_for<LAST>([&](auto jc) PJR_INL {
        constexpr int j = decltype(jc)::value;
        J_(NSP * (j + 1)) += E[j];
    });
#undef J_
#undef LD_
}

void launch_part(const PjrArgs& A, void* stream)
{
    const long blocks = (A.n + PJR_BLOCK - 1) / PJR_BLOCK;
    hipLaunchKernelGGL(k_rows, dim3((unsigned)blocks), dim3(PJR_BLOCK), 0, (hipStream_t)stream, A);
}
struct Reg { Reg() { pjr_register(PJR_ID, 2, launch_part); } } reg_;
#endif  // PJR_PART == 2

#if PJR_PART == 3
// ------------------------------------------------------------------------------------------
// k_fused: the whole Jacobian of a 64-state tile by one workgroup of 4 wavefronts.
//
// The multi-kernel path above is HBM-bound on its scratch traffic (every c*k_f / c*k_r is written
// once and re-read ~3.6 times, 2.8x the algorithmic bytes in total).  Here the four wavefronts of
// a workgroup own the SAME 64 states and split the work instead of the states (reaction i ->
// wavefront i % 4, row blocks dealt by estimated cost: pjs::ARM_BLKS), and resident workgroups walk the state tiles,
// so the scratch array is one fixed (NSCR x 64 doubles) region per resident workgroup, ~100 MB for
// the whole device: rewritten and re-read every tile, it stays in the 256 MB Infinity Cache while
// the Jacobian streams out through nontemporal stores.  Cross-wavefront sums (d/dT column, energy
// row, H / SCP / SJT) go through LDS atomics; nothing but T, p, Y is read from and nothing but the
// Jacobian is written to memory outside that region.
// ------------------------------------------------------------------------------------------
#ifndef PJR_WLANES
#define PJR_WLANES 64      // lanes per wavefront = states per tile
#endif
#ifndef PJR_NW
#define PJR_NW 4           // wavefronts per workgroup; arm a runs on wavefront a % PJR_NW
#endif
static_assert(!PJR_RECOMPUTE_KR, "the fused kernel reads c*k_r back from its scratch region");
constexpr int NARM = 4;
constexpr int NKC = pjs::LT_SP / 16;
constexpr int NEFF = (int)(sizeof(pjs::EFF_AM1) / sizeof(pjs::EFF_AM1[0]));
template <int i>
constexpr bool has_anm1() { return pjs::RD[i][RD_ANM1] != 0.0; }
// first reaction of its arm that uses a K_c class evaluates it
constexpr bool kc_first_arm(int i)
{
    for (int h = i % NARM; h < i; h += NARM)
        if ((pjs::RI[h][RI_FLAGS] & F_REV) && pjs::KC_CLASS[h][0] == pjs::KC_CLASS[i][0]) return false;
    return true;
}
#ifdef PJR_HOST_EMU
#define PJR_LDS_ADD(ptr, val) (*(ptr) += (val))
#define PJR_STORE_NT(ptr, val) (*(ptr) = (val))
#else
#define PJR_LDS_ADD(ptr, val) ((void)__hip_atomic_fetch_add((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define PJR_STORE_NT(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif

#ifdef PJR_TIMING
// debug build: cycles per phase and wavefront, summed over the tiles of a workgroup
__device__ long long g_tim[8][4][512];
#define PJR_TICK(ph) { const long long tn_ = clock64(); tacc[ph] += tn_ - tprev; tprev = tn_; }
#undef PJR_TICK_B
#define PJR_TICK_B(ph) PJR_TICK(ph)
#else
#define PJR_TICK(ph)
#endif

__global__ void __launch_bounds__(PJR_WLANES * PJR_NW) k_fused(PjrArgs A)
{
    __shared__ __attribute__((aligned(16))) double LT[(NKC > 0 ? NKC : 1) * 16];
    __shared__ __attribute__((aligned(16))) double RDL[NRXN][RDW];
    __shared__ __attribute__((aligned(16))) double EFL[NEFF];
    // species constants (1/W, W, T_mid, W/W_N, NASA rows) read with uniform ds_reads: as literals
    // the persistent tile loop would hoist ~2000 64-bit constants out of the loop and spill them
    __shared__ __attribute__((aligned(16))) double SPL[NSP][SPW];
    __shared__ double CL[NSP][PJR_WLANES];     // concentrations of the tile
    __shared__ double RED[NSP][PJR_WLANES];    // sum nu theta per species, later the energy-row sums
    __shared__ double SUMS[5][PJR_WLANES];     // H, SCP, SJT, cp_avg, d(cp_avg)/dT
    constexpr int NT = PJR_WLANES * PJR_NW;
    for (int x = threadIdx.x; x < NKC * 16; x += NT) LT[x] = pjs::LTAB[pjs::LT_KC + x];
    for (int x = threadIdx.x; x < NRXN * RDW; x += NT) (&RDL[0][0])[x] = (&pjs::RDT[0][0])[x];
    for (int x = threadIdx.x; x < NEFF; x += NT) EFL[x] = pjs::EFFT[x][0];
    for (int x = threadIdx.x; x < NSP * SPW; x += NT) (&SPL[0][0])[x] = (&pjs::SPF[0][0])[x];
#define PJR_SP(k_, c_) SPL[k_][c_]
#define PJR_EFF(e_) EFL[e_]
#define PJR_NASA(k_, lo_, c_) SPL[k_][((lo_) ? 4 : 11) + (c_)]
#define PJR_ANM1(i_) RDL[i_][RD_ANM1]
    const int lane = threadIdx.x % PJR_WLANES;
#ifdef PJR_HOST_EMU
    const int w = threadIdx.x / PJR_WLANES;
#else
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / PJR_WLANES);   // wavefront-uniform: scalar branches
#endif
    double* const scr0 = A.scr + (long)blockIdx.x * ((long)pjs::NSCR * PJR_WLANES) + lane;
#define SCR_(slot) scr[(slot) * PJR_WLANES]
#define SCR_ST(slot, val) (SCR_(slot) = (val))
#define LD_(slot) scr[(slot) * PJR_WLANES]
#define CC(idx) ((idx) == ONE ? 1.0 : CL[(idx) == ONE ? 0 : (idx)][lane])
    auto conc = [&](auto spc) PJR_INL {
        constexpr int sp = decltype(spc)::value;
        if constexpr (sp == ONE) return 1.0; else return CL[sp][lane];
    };
    const long ntiles = (A.n + PJR_WLANES - 1) / PJR_WLANES;
#ifdef PJR_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // lanes past the end repeat the last state (same values to the same addresses)
        long s = t * PJR_WLANES + lane;
        if (s >= A.n) s = A.n - 1;
        // The scratch region is the same every tile; hide that from the optimiser, which would
        // otherwise hoist the 64-bit address of every scratch slot out of this loop (~NSCR live
        // pointers per lane, all spilled).
        double* scr = scr0;
        long j_si = A.j_si;                 // same for the per-entry offsets e * j_si (SGPR pairs)
#ifndef PJR_HOST_EMU
        asm volatile("" : "+v"(scr));
        asm volatile("" : "+s"(j_si));
#endif
        double T, p, rho, invrho, Wbar, mconc;
        __syncthreads();                    // previous tile done with CL / RED / SUMS / scratch
        {
            // eval_conc (every wavefront needs the sums; each stores its share of C_k)
            const double* y = A.y + s * A.y_ss;
            T = y[0];
            p = A.pres[s];
            double Y[NSP], sumY = 0.0, sumYW = 0.0;
            static_for<LAST>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                Y[k] = y[(k + 1) * A.y_si];
                sumY += Y[k];
                sumYW += Y[k] * PJR_SP(k, 0);
            });
            Y[LAST] = 1.0 - sumY;
            sumYW += Y[LAST] * PJR_SP(LAST, 0);
            Wbar = 1.0 / sumYW;
            rho = p * Wbar / (RU_ * T);
            invrho = 1.0 / rho;
            mconc = p / (RU_ * T);
            static_for<NSP>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                if (k % PJR_NW == w) { CL[k][lane] = rho * Y[k] * PJR_SP(k, 0); RED[k][lane] = 0.0; }
            });
            if (w == 0) static_for<5>([&](auto cc) PJR_INL { SUMS[decltype(cc)::value][lane] = 0.0; });
        }
        double* const Jl = A.jac + s * A.j_ss;
#define J_(e) Jl[(long)(e) * j_si]
        PJR_TICK(0)
        __syncthreads();
        PJR_TICK(2)

        // ---- phase 1: rates of this wavefront's reactions -> scratch; d/dT partial sums ----
        {
            const double logT = log(T), invT = 1.0 / T, logp = log(p);
            static_for<NARM>([&](auto ac) PJR_INL {
                constexpr int arm = decltype(ac)::value;
                if (arm % PJR_NW == w) {
                    double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
                    double jt[NSP], jtq = 0.0;
                    constexpr bool RATES_OUT = false;
                    auto rate_out = [](auto, double, double, double) {};
                    static_for<NSP>([&](auto kc) PJR_INL { jt[decltype(kc)::value] = 0.0; });
                    static_for<(NRXN - arm + NARM - 1) / NARM>([&](auto nc) PJR_INL {
                        constexpr int i = arm + NARM * decltype(nc)::value;
#define PJR_RD(i_) RDL[i_]
#define PJR_KCROW(g_) (LT + (g_) * 16)
#define PJR_EFL(e_) EFL[e_]
#define PJR_KC_FIRST(i_) kc_first_arm(i_)
#include "pj_rows_rate.inc"
#undef PJR_RD
#undef PJR_KCROW
#undef PJR_EFL
#undef PJR_KC_FIRST
                    });
                    // reference quirk (create_jacobian.py:2786-2818), see pj_kernel.h
                    if (!A.sum_last) jt[LAST] = (pjs::LASTQ >= 0 && pjs::LASTQ % NARM == arm) ? jtq : 0.0;
                    static_for<NSP>([&](auto kc) PJR_INL {
                        constexpr int k = decltype(kc)::value;
                        PJR_LDS_ADD(&RED[k][lane], jt[k]);
                    });
                }
            });
        }
        PJR_TICK(1)
        __syncthreads();
        PJR_TICK(2)

        // ---- d/dT column out, SJT; RED is handed over (zeroed) to the energy row ----
        {
            double sjt = 0.0, cpa = 0.0, dcpa = 0.0;
            static_for<NSP>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                if (k % PJR_NW == w) {
                    const double jtk = RED[k][lane];
                    RED[k][lane] = 0.0;
                    const bool lo = T <= PJR_SP(k, 2);
                    double a[6];
                    static_for<6>([&](auto cc) PJR_INL {
                        constexpr int c = decltype(cc)::value;
                        a[c] = PJR_NASA(k, lo, c);
                    });
                    const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                             T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
                    sjt += hW * jtk;
                    if constexpr (k < LAST) PJR_STORE_NT(&J_(k + 1), PJR_SP(k, 1) * jtk);
                    // mass-fraction weighted c_p sums from the concentrations: Y_k c_p,k = C_k R (a0 + ...) / rho
                    const double Ck = CL[k][lane];
                    cpa += Ck * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
                    dcpa += Ck * (a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T)));
                }
            });
            PJR_LDS_ADD(&SUMS[2][lane], sjt);
            PJR_LDS_ADD(&SUMS[3][lane], cpa * (RU_ * invrho));
            PJR_LDS_ADD(&SUMS[4][lane], dcpa * (RU_ * invrho));
        }
        PJR_TICK(3)
        __syncthreads();
        PJR_TICK(2)

        // ---- phase 2: this wavefront's row blocks ----
        {
            double E[LAST > 0 ? LAST : 1];
            static_for<LAST>([&](auto jc) PJR_INL { E[decltype(jc)::value] = 0.0; });
            double H = 0.0, SCP = 0.0;
            double ring[PJR_DEPTH][6];
            auto issue_bv = [&](auto bc, auto vc) PJR_INL {
                constexpr int v = decltype(vc)::value;
                constexpr int i = pjs::BLK_RX[pjs::BLK_RX_PTR[decltype(bc)::value][0] + v][0];
                static_for<6>([&](auto cc) PJR_INL {
                    constexpr int c = decltype(cc)::value;
                    if constexpr (pjs::SCR[i][c] >= 0) ring[v % PJR_DEPTH][c] = LD_(pjs::SCR[i][c]);
                });
            };
            auto prologue = [&](auto bc) PJR_INL {
                constexpr int b = decltype(bc)::value;
                constexpr int nv = pjs::BLK_RX_PTR[b + 1][0] - pjs::BLK_RX_PTR[b][0];
                static_for<(nv < PJR_DEPTH ? nv : PJR_DEPTH)>([&](auto vc) PJR_INL { issue_bv(bc, vc); });
            };
#undef PJR_STORE
#define PJR_STORE(ptr, val) PJR_STORE_NT(ptr, val)
            static_for<NARM>([&](auto ac) PJR_INL {
                constexpr int arm = decltype(ac)::value;
                if (arm % PJR_NW == w) {
                    static_for<pjs::ARM_BLK_PTR[arm + 1][0] - pjs::ARM_BLK_PTR[arm][0]>([&](auto nc) PJR_INL {
                        constexpr int b = pjs::ARM_BLKS[pjs::ARM_BLK_PTR[arm][0] + decltype(nc)::value][0];
                        const std::integral_constant<int, b> bc{};
#include "pj_rows_block.inc"
                    });
                }
            });
            static_for<LAST>([&](auto jc) PJR_INL {
                constexpr int j = decltype(jc)::value;
                PJR_LDS_ADD(&RED[j][lane], E[j]);
            });
            PJR_LDS_ADD(&SUMS[0][lane], H);
            PJR_LDS_ADD(&SUMS[1][lane], SCP);
        }
        PJR_TICK(4)
        __syncthreads();
        PJR_TICK(2)

        // ---- energy row ----
        {
            auto cp_of = [&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                const bool lo = T <= PJR_SP(k, 2);
                double a[5];
                static_for<5>([&](auto cc) PJR_INL {
                    constexpr int c = decltype(cc)::value;
                    a[c] = PJR_NASA(k, lo, c);
                });
                return (RU_ * PJR_SP(k, 0)) * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
            };
            const double cpavg = SUMS[3][lane], dcpavg = SUMS[4][lane];
            const double cpN = cp_of(std::integral_constant<int, LAST>{});
            const double H = SUMS[0][lane], SCP = SUMS[1][lane], SJT = SUMS[2][lane];
            const double icp = 1.0 / cpavg;
            if (w == 0) PJR_STORE_NT(&J_(0), -(SCP - (dcpavg * icp) * H + rho * SJT) / (rho * cpavg));
            static_for<LAST>([&](auto jc) PJR_INL {
                constexpr int j = decltype(jc)::value;
                if (j % PJR_NW == w) {
                    const double cpj = cp_of(jc);
                    PJR_STORE_NT(&J_(NSP * (j + 1)),
                                 -RED[j][lane] * PJR_SP(j, 0) * icp + (cpj - cpN) * H * invrho * icp * icp);
                }
            });
        }
        PJR_TICK(5)
#undef J_
    }
#ifdef PJR_TIMING
    if (lane == 0 && blockIdx.x < 512)
        for (int ph = 0; ph < 8; ++ph) g_tim[ph][w][blockIdx.x] = tacc[ph];
#endif
#undef SCR_
#undef SCR_ST
#undef LD_
#undef CC
#undef PJR_SP
#undef PJR_EFF
#undef PJR_NASA
#undef PJR_ANM1
}

double* g_scr = nullptr;
long g_scr_wgs = 0;
#endif  // PJR_PART == 3

#if PJR_PART == 0
// ------------------------------------------------------------------------------------------
// k_fin + host side
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PJR_BLOCK) k_fin(PjrArgs A)
{
    const long s = (long)blockIdx.x * PJR_BLOCK + threadIdx.x;
    if (s >= A.n) return;
    State L;
    load_state(A, s, L);
    const double T = L.T;
    double cpavg = 0.0, dcpavg = 0.0, cpN = 0.0;
    auto cp_of = [&](auto kc, double& dcp) PJR_INL {
        constexpr int k = decltype(kc)::value;
        const bool lo = T <= pjs::SP[k][2];
        double a[5];
        static_for<5>([&](auto cc) PJR_INL {
            constexpr int c = decltype(cc)::value;
            a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
        });
        const double RW = RU_ * pjs::SP[k][0];
        dcp = RW * (a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T)));
        return RW * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
    };
    static_for<NSP>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        double dcp;
        const double cp = cp_of(kc, dcp);
        cpavg += L.C[k] * cp;
        dcpavg += L.C[k] * dcp;
        if constexpr (k == LAST) cpN = cp;
    });
    const double* const scr = scr_of(A, s);
    const double H = scr[(long)SUM_H * PJR_SSTRIDE(A)], SCP = scr[(long)SUM_SCP * PJR_SSTRIDE(A)],
                 SJT = scr[(long)SUM_SJT * PJR_SSTRIDE(A)];
    const double rho = L.rho, invrho = L.invrho, icp = 1.0 / cpavg;
    double* const Jl = A.jac + s * A.j_ss;
#define J_(e) Jl[(long)(e) * A.j_si]
    J_(0) = -(SCP - (dcpavg * icp) * H + rho * SJT) / (rho * cpavg);
    static_for<LAST>([&](auto jc) PJR_INL {
        constexpr int j = decltype(jc)::value;
        double dcp;
        const double cpj = cp_of(jc, dcp);
        const double tot = J_(NSP * (j + 1));
        J_(NSP * (j + 1)) = -tot * pjs::SP[j][0] * icp + (cpj - cpN) * H * invrho * icp * icp;
    });
#undef J_
}

// dydt from the complete omega_k (rate_subs.py:2171-2335): dT/dt = -sum h_k W_k omega_k / (rho cp),
// dY_k/dt = omega_k W_k / rho
__global__ void __launch_bounds__(PJR_BLOCK) k_dy(PjrArgs A)
{
    const long s = (long)blockIdx.x * PJR_BLOCK + threadIdx.x;
    if (s >= A.n) return;
    State L;
    load_state(A, s, L);
    const double T = L.T;
    double cpavg = 0.0, Hs = 0.0;
    static_for<NSP>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        const bool lo = T <= pjs::SP[k][2];
        double a[6];
        static_for<6>([&](auto cc) PJR_INL {
            constexpr int c = decltype(cc)::value;
            a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
        });
        cpavg += L.C[k] * (RU_ * pjs::SP[k][0]) * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
        const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                 T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
        const double om = A.sr[k * A.sr_ld + s];
        Hs += hW * om;
        if constexpr (k < LAST) A.dy[(k + 1) * A.o_ld + s] = om * pjs::SP[k][1] * L.invrho;
    });
    A.dy[s] = -Hs / (L.rho * cpavg);
}

constexpr int MAXPARTS = 512;
// rows of a scratch array: the hand-over slots + 3 sums, or NSP rows when pj_spec_rates keeps omega_k there
constexpr int SCR_ROWS = (pjs::NSCR + 3 > NSP) ? pjs::NSCR + 3 : NSP;
pjr_launch_fn g_rates[MAXPARTS], g_rows[MAXPARTS], g_rates_out[MAXPARTS];
double* g_scr[2] = {nullptr, nullptr};
long g_scr_ld[2] = {0, 0};
hipStream_t g_streams[2] = {nullptr, nullptr};
enum { EV_START, EV_RATES0, EV_RATES1, EV_ROWS0, EV_ROWS1, EV_COUNT };
hipEvent_t g_events[EV_COUNT];
#endif

}  // namespace

#if PJR_PART == 0
extern "C" {

void pjr_register(int id, int kind, pjr_launch_fn fn)
{
    if (id < 0 || id >= MAXPARTS) return;
    (kind == 1 ? g_rates : kind == 3 ? g_rates_out : g_rows)[id] = fn;
}

#ifndef PJR_RATES_LIB
unsigned long long pj_spec_hash(void) { return PJS_HASH; }
int pj_spec_nsp(void) { return NSP; }
int pj_spec_kind(void) { return 2; }   // 1: pj_lane.hip, 2: pj_rows.hip
long pj_spec_scratch_doubles_per_state(void) { return pjs::NSCR + 3; }

// layouts as in include/pyjac_amd.h: element (i, s) at base[i*si + s*ss].  One batch at a time
// per library (the scratch arrays are shared): calls on different streams must not overlap.
//
// The batch runs in chunks through two scratch arrays on two internal streams: the rate kernels
// of chunk c+1 (compute-bound) overlap the row kernels of chunk c (HBM-bound); both are forked
// from and joined to the caller's stream with events, so the call is asynchronous and ordered
// like a single kernel launch on `stream`.  PJ_ROWS_OVERLAP=0: everything on the caller's stream.
int pj_spec_jacobian(long n, const double* pres, const double* y, long y_si, long y_ss, double* jac,
                     long j_si, long j_ss, int sum_last, void* stream)
{
    if (n <= 0) return 0;
    static int overlap = -1;
    if (overlap < 0) { const char* e = getenv("PJ_ROWS_OVERLAP"); overlap = (e && atoi(e) == 0) ? 0 : 1; }
    long chunk = 262144;
    if (const char* e = getenv("PJ_ROWS_CHUNK")) { const long v = atol(e); if (v >= 256) chunk = v; }
    // at least two chunks when the batch fills the device several times over (measured: splitting a
    // 2e5-state batch of the 111-species mechanism costs 10 %, splitting 1e6 states gains 3 %)
    if (overlap && n > 2 * 131072 && n < 2 * chunk) chunk = (n + 1) / 2;
    chunk = (chunk + PJR_TILE - 1) / PJR_TILE * PJR_TILE;
    if (chunk > n) chunk = (n + PJR_TILE - 1) / PJR_TILE * PJR_TILE;
    const int nbuf = (overlap && n > chunk) ? 2 : 1;
    for (int b = 0; b < nbuf; ++b) {
        if (g_scr_ld[b] >= chunk) continue;
        if (g_scr[b]) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr[b]); g_scr[b] = nullptr; g_scr_ld[b] = 0; }
        if (hipMalloc((void**)&g_scr[b], sizeof(double) * (size_t)SCR_ROWS * (size_t)chunk) != hipSuccess) return -4;
        g_scr_ld[b] = chunk;
    }
    hipStream_t user = (hipStream_t)stream, s_rates = user, s_rows = user;
    if (nbuf == 2) {
        if (!g_streams[0]) {
            if (hipStreamCreateWithFlags(&g_streams[0], hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&g_streams[1], hipStreamNonBlocking) != hipSuccess) return -3;
            for (auto& e : g_events)
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -3;
        }
        s_rates = g_streams[0]; s_rows = g_streams[1];
        // fork
        (void)hipEventRecord(g_events[EV_START], user);
        (void)hipStreamWaitEvent(s_rates, g_events[EV_START], 0);
        (void)hipStreamWaitEvent(s_rows, g_events[EV_START], 0);
    }
    long c = 0;
    for (long s0 = 0; s0 < n; s0 += chunk, ++c) {
        const long m = s0 + chunk < n ? chunk : n - s0;
        const int b = (int)(c % nbuf);
        PjrArgs A{m, pres + s0, y + s0 * y_ss, y_si, y_ss, jac + s0 * j_ss, j_si, j_ss, g_scr[b], g_scr_ld[b], sum_last,
                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
        // scratch buffer b is free again once the row kernels of chunk c-2 are done
        if (nbuf == 2 && c >= 2) (void)hipStreamWaitEvent(s_rates, g_events[EV_ROWS0 + b], 0);
        for (int i = 0; i < MAXPARTS; ++i) if (g_rates[i]) g_rates[i](A, s_rates);
        if (nbuf == 2) {
            (void)hipEventRecord(g_events[EV_RATES0 + b], s_rates);
            (void)hipStreamWaitEvent(s_rows, g_events[EV_RATES0 + b], 0);
        }
        for (int i = 0; i < MAXPARTS; ++i) if (g_rows[i]) g_rows[i](A, s_rows);
        hipLaunchKernelGGL(k_fin, dim3((unsigned)((m + PJR_BLOCK - 1) / PJR_BLOCK)), dim3(PJR_BLOCK), 0, s_rows, A);
        if (nbuf == 2) (void)hipEventRecord(g_events[EV_ROWS0 + b], s_rows);
    }
    if (nbuf == 2) {
        // join: the last event recorded on s_rows covers every chunk (one stream, in order); the rate
        // stream finished before it by construction
        (void)hipStreamWaitEvent(user, g_events[EV_ROWS0 + (int)((c - 1) % 2)], 0);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

#endif  // !PJR_RATES_LIB

// Rate outputs of one pass (pyjacob.cu:18-35 k_dydt): any pointer may be null; SoA, leading dimension n.
int pj_spec_rates(long n, const double* pres, const double* y, long y_si, long y_ss, double* conc, double* fwd,
                  double* rev, double* pres_mod, double* spec_rates, double* dy, void* stream)
{
    if (n <= 0) return 0;
    // omega_k is accumulated across the rate kernels in memory: the caller's spec_rates array, or a
    // chunk of the scratch array when only dydt is wanted
    long chunk = n;
    if (!spec_rates) {
        chunk = 262144;
        if (chunk > n) chunk = (n + PJR_TILE - 1) / PJR_TILE * PJR_TILE;
        if (g_scr_ld[0] < chunk) {
            if (g_scr[0]) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr[0]); g_scr[0] = nullptr; g_scr_ld[0] = 0; }
            if (hipMalloc((void**)&g_scr[0], sizeof(double) * (size_t)SCR_ROWS * (size_t)chunk) != hipSuccess) return -4;
            g_scr_ld[0] = chunk;
        }
    }
    for (long s0 = 0; s0 < n; s0 += chunk) {
        const long m = s0 + chunk < n ? chunk : n - s0;
        PjrArgs A{m, pres + s0, y + s0 * y_ss, y_si, y_ss, nullptr, 0, 0, nullptr, 0, 0,
                  conc ? conc + s0 : nullptr, fwd ? fwd + s0 : nullptr, rev ? rev + s0 : nullptr,
                  pres_mod ? pres_mod + s0 : nullptr, spec_rates ? spec_rates + s0 : g_scr[0],
                  dy ? dy + s0 : nullptr, n, spec_rates ? n : g_scr_ld[0]};
        for (int i = 0; i < MAXPARTS; ++i) if (g_rates_out[i]) g_rates_out[i](A, stream);
        if (dy)
            hipLaunchKernelGGL(k_dy, dim3((unsigned)((m + PJR_BLOCK - 1) / PJR_BLOCK)), dim3(PJR_BLOCK), 0,
                               (hipStream_t)stream, A);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
#endif

#if PJR_PART == 3
extern "C" {

unsigned long long pj_spec_hash(void) { return PJS_HASH; }
int pj_spec_nsp(void) { return NSP; }
int pj_spec_kind(void) { return 3; }   // 1: pj_lane.hip, 2: pj_rows.hip kernels, 3: pj_rows.hip fused
#ifdef PJR_TIMING
int pj_spec_debug_timing(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tim), sizeof(g_tim)); }
#endif

// layouts as in include/pyjac_amd.h: element (i, s) at base[i*si + s*ss].  One batch at a time
// per library (the per-workgroup scratch regions are shared): calls on different streams must
// not overlap.
int pj_spec_jacobian(long n, const double* pres, const double* y, long y_si, long y_ss, double* jac,
                     long j_si, long j_ss, int sum_last, void* stream)
{
    if (n <= 0) return 0;
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fused, PJR_WLANES * PJR_NW, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    long wgs = (n + PJR_WLANES - 1) / PJR_WLANES;
    if (wgs > resident) wgs = resident;
    if (g_scr_wgs < wgs) {
        if (g_scr) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr); g_scr = nullptr; g_scr_wgs = 0; }
        if (hipMalloc((void**)&g_scr, sizeof(double) * (size_t)pjs::NSCR * PJR_WLANES * (size_t)wgs) != hipSuccess) return -4;
        g_scr_wgs = wgs;
    }
    PjrArgs A{n, pres, y, y_si, y_ss, jac, j_si, j_ss, g_scr, 0, sum_last,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(k_fused, dim3((unsigned)wgs), dim3(PJR_WLANES * PJR_NW), 0, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
#endif
