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
};
typedef void (*pjr_launch_fn)(const PjrArgs&, void* stream);
extern "C" void pjr_register(int id, int kind, pjr_launch_fn fn);

namespace {

constexpr double RU_ = 8314.4621;
constexpr double INV_LN10 = 0.434294481903251828;
constexpr int NSP = pjs::NSP, NRXN = pjs::NRXN, LAST = pjs::NSP - 1, ONE = pjs::NSP;
constexpr int S_TH = 0, S_KF = 1, S_KR = 2, S_RP = 3, S_BM = 4, S_BC = 5;
constexpr int SUM_H = pjs::NSCR, SUM_SCP = pjs::NSCR + 1, SUM_SJT = pjs::NSCR + 2;

#define PJR_INL __attribute__((always_inline))
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

__global__ void __launch_bounds__(PJR_BLOCK) k_rates(PjrArgs A)
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
    for (long s = (long)blockIdx.x * PJR_BLOCK + threadIdx.x; s < A.n; s += (long)gridDim.x * PJR_BLOCK) {
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
        double* const Jl = A.jac + s * A.j_ss;
#define J_(e) Jl[(long)(e) * A.j_si]
        if constexpr (R0_ == 0) {
            // this part also clears what the row kernels accumulate into
            SCR_(SUM_H) = 0.0; SCR_(SUM_SCP) = 0.0;
            static_for<LAST>([&](auto jc) PJR_INL { J_(NSP * (decltype(jc)::value + 1)) = 0.0; });
        }
        double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
        // d/dT column: sum_q nu_kq theta_q needs nothing but theta, so it is finished here and
        // theta never goes through the scratch array
        double jt[NSP], jtq = 0.0;
        static_for<NSP>([&](auto kc) PJR_INL { jt[decltype(kc)::value] = 0.0; });
        static_range<R0_, R1_>([&](auto ic) PJR_INL {
            constexpr int i = decltype(ic)::value;
#define PJR_RD(i_) RDL[(i_) - R0_]
#define PJR_KCROW(g_) (LT + ((g_) - KC_LO) * 16)
#define PJR_EFL(e_) EFL[e_]
#define PJR_KC_FIRST(i_) kc_first_in_range(i_)
#include "pj_rows_rate.inc"
#undef PJR_RD
#undef PJR_KCROW
#undef PJR_EFL
#undef PJR_KC_FIRST
        });
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
#undef CC
    }
}

void launch_part(const PjrArgs& A, void* stream)
{
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rates, PJR_BLOCK, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    long blocks = (A.n + PJR_BLOCK - 1) / PJR_BLOCK;
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(k_rates, dim3((unsigned)blocks), dim3(PJR_BLOCK), 0, (hipStream_t)stream, A);
}
struct Reg { Reg() { pjr_register(PJR_ID, 1, launch_part); } } reg_;
#endif  // PJR_PART == 1

#if PJR_PART == 2
// ------------------------------------------------------------------------------------------
// k_rows<B0,B1>
// ------------------------------------------------------------------------------------------
constexpr int B0_ = PJR_B0, B1_ = PJR_B1;

// does reaction i carry an enhanced efficiency of the last species?
template <int i>
constexpr bool has_anm1() { return pjs::RD[i][RD_ANM1] != 0.0; }

__global__ void __launch_bounds__(PJR_BLOCK) k_rows(PjrArgs A)
{
    // Concentrations live in LDS, one column per lane (bank-conflict free): the registers they
    // would occupy are worth more as landing space for scratch loads in flight -- at one
    // wavefront per SIMD the bytes in flight per lane bound the achieved HBM bandwidth.
    __shared__ double CL[NSP][PJR_BLOCK];
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
    auto conc = [&](auto spc) PJR_INL {
        constexpr int sp = decltype(spc)::value;
        if constexpr (sp == ONE) return 1.0; else return CL[sp][tid];
    };
    // energy-row partial sums: rarely touched, so the register allocator parks them in AGPRs
    double E[LAST > 0 ? LAST : 1];
    const double* const scr = scr_of(A, s);
#define LD_(slot) scr[(long)(slot) * PJR_SSTRIDE(A)]
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
            if constexpr (pjs::SCR[i][c] >= 0) ring[v % PJR_DEPTH][c] = LD_(pjs::SCR[i][c]);
        });
    };
    auto prologue = [&](auto bc) PJR_INL {
        constexpr int b = decltype(bc)::value;
        constexpr int nv = pjs::BLK_RX_PTR[b + 1][0] - pjs::BLK_RX_PTR[b][0];
        static_for<(nv < PJR_DEPTH ? nv : PJR_DEPTH)>([&](auto vc) PJR_INL { issue_bv(bc, vc); });
    };
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

constexpr int MAXPARTS = 512;
pjr_launch_fn g_rates[MAXPARTS], g_rows[MAXPARTS];
double* g_scr = nullptr;
long g_scr_ld = 0;
#endif

}  // namespace

#if PJR_PART == 0
extern "C" {

void pjr_register(int id, int kind, pjr_launch_fn fn)
{
    if (id < 0 || id >= MAXPARTS) return;
    (kind == 1 ? g_rates : g_rows)[id] = fn;
}

unsigned long long pj_spec_hash(void) { return PJS_HASH; }
int pj_spec_nsp(void) { return NSP; }
int pj_spec_kind(void) { return 2; }   // 1: pj_lane.hip, 2: pj_rows.hip
long pj_spec_scratch_doubles_per_state(void) { return pjs::NSCR + 3; }

// layouts as in include/pyjac_amd.h: element (i, s) at base[i*si + s*ss].  One batch at a time
// per library (the scratch array is shared): calls on different streams must not overlap.
int pj_spec_jacobian(long n, const double* pres, const double* y, long y_si, long y_ss, double* jac,
                     long j_si, long j_ss, int sum_last, void* stream)
{
    if (n <= 0) return 0;
    long chunk = 262144;
    if (const char* e = getenv("PJ_ROWS_CHUNK")) { const long v = atol(e); if (v >= 256) chunk = v; }
    chunk = (chunk + PJR_TILE - 1) / PJR_TILE * PJR_TILE;
    if (chunk > n) chunk = (n + PJR_TILE - 1) / PJR_TILE * PJR_TILE;
    if (g_scr_ld < chunk) {
        if (g_scr) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr); g_scr = nullptr; g_scr_ld = 0; }
        if (hipMalloc((void**)&g_scr, sizeof(double) * (size_t)(pjs::NSCR + 3) * (size_t)chunk) != hipSuccess) return -4;
        g_scr_ld = chunk;
    }
    for (long s0 = 0; s0 < n; s0 += chunk) {
        const long m = s0 + chunk < n ? chunk : n - s0;
        PjrArgs A{m, pres + s0, y + s0 * y_ss, y_si, y_ss, jac + s0 * j_ss, j_si, j_ss, g_scr, g_scr_ld, sum_last};
        for (int i = 0; i < MAXPARTS; ++i) if (g_rates[i]) g_rates[i](A, stream);
        for (int i = 0; i < MAXPARTS; ++i) if (g_rows[i]) g_rows[i](A, stream);
        hipLaunchKernelGGL(k_fin, dim3((unsigned)((m + PJR_BLOCK - 1) / PJR_BLOCK)), dim3(PJR_BLOCK), 0,
                           (hipStream_t)stream, A);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
#endif
