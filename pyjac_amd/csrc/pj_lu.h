// pj_lu.h -- batched dense LU (partial pivoting) and triangular solves on per-state NSP x NSP blocks: the consumer
// of the Jacobians an implicit integrator's Newton iteration needs (SURVEY 8f N2 "fused consumer: batched LU";
// the reference hands its per-state Jacobian to such a solver, docs/examples.rst:106-170, and has no batched
// form of it).  The blocks are pyJac's per-state C layout (column-major, state-major: A[s*NSP*NSP + r + NSP*c]).
//
// One wavefront per matrix, one LANE PER ROW, the row's entries in registers with compile-time indices:
//  * column k's pivot is a wavefront reduction (6 DPP steps on |a_k|, then a ballot picks the first lane that holds
//    the maximum -- LAPACK's choice);
//  * rows are never swapped: a lane remembers the position its row was chosen for (implicit pivoting) and rows are
//    written out at their positions, so the result is the usual P A = L U block with the row permutation next to it;
//  * the pivot row reaches the other lanes through v_readlane (lane index in a scalar register), i.e. as scalar
//    operands of the update FMAs: no LDS, no shuffles.
// Elimination step k costs 3 instructions per remaining column (two v_readlane, one fused multiply-add over all
// rows at once) + ~40 for the pivot search: 6.4 k instructions for a 53 x 53 block.  Bound: HBM (reads and writes
// 8 NSP^2 bytes per state) for large NSP, instruction issue below that.  Blocks of 65 .. 140 rows: k_lu_lds below.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

// Translation units (csrc/pj_lu.hip; __graft_entry__.build compiles them in parallel -- k_lu4 alone is four instantiations of
// two minutes each):
//   PJ_LU_DECL_ONLY   declarations for callers (pj_api.hip): enum, LuLay, LU_MAX_LDS, lu_launch_x()
//   PJ_LU4_SPLIT      lu_launch() reaches the k_lu4 instantiations through lu4_go_<NP>() defined elsewhere
//   PJ_LU4_ONLY=NP    only the helpers, k_lu4 and lu4_go_<NP>()
// None of them defined: everything inline in the including file (tools/micro/lu_bw.hip).
namespace pj {
enum { LU_FACTOR = 1, LU_SOLVE = 2, LU_PREFACTORED = 4 };
// where the input blocks and the vectors live: entry (r, c) of block s at a[(r + NSP c) * a_si + s * a_ss], entry i of
// vector s at v[i * v_si + s * v_ss] -- pyJac's per-state layout (a_si = 1, a_ss = NSP^2; v_si = 1, v_ss = NSP) or the
// state-fastest batch layout the row kernels write (a_si = n, a_ss = 1; v_si = n, v_ss = 1).  Factors are always
// written / read per state (P A = L U blocks are consumed by these kernels only).
struct LuLay { long a_si, a_ss, v_si, v_ss; };
constexpr int LU_MAX_LDS = 140;        // (140 | 1) * 140 + 3 * 140 doubles = 159 KB of the 160 KB
int lu_launch_x(int nsp, long n, const double* A, LuLay Y, double gamma, double* lu, int* perm, const double* b, double* x,
                int mode, int cus, hipStream_t st);
#define PJ_LU4_GO_DECL(NP_) void lu4_go_##NP_(unsigned blocks, hipStream_t st, int nsp, long n, const double* A, LuLay Y, double gamma, \
                                              double* lu, int* perm, const double* b, double* x, int mode);
PJ_LU4_GO_DECL(80) PJ_LU4_GO_DECL(96) PJ_LU4_GO_DECL(112) PJ_LU4_GO_DECL(128)
}
#ifndef PJ_LU_DECL_ONLY

namespace pj {

template <int I, int N, class F>
__device__ __forceinline__ void lu_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lu_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ double lu_readlane(const double v, const int lane)
{
    const long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the same broadcast through the LDS crossbar (ds_bpermute_b32: no LDS memory, no VALU issue slot; the value arrives
// in a vector register)
__device__ __forceinline__ double lu_bpermute(const double v, const int byte_addr)
{
    const long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#ifndef PJ_LU_ABL
#define PJ_LU_ABL 0          // timing experiments on k_lu_lds (results wrong): 1 no trailing update, 2 no row exchange, 4 no pivot search
#endif
#ifndef PJ_LU_LOOKAHEAD
#define PJ_LU_LOOKAHEAD 1    // k_lu: next column's pivot search in slices between this step's column updates (0: after them)
#endif
#ifndef PJ_LU_BPERM
#define PJ_LU_BPERM 4       // broadcasts of the pivot row: 0 all v_readlane, 1 all ds_bpermute, 2 every other column, 4 every third, 5 two of five
#endif
#ifndef PJ_LU_WAVES
#define PJ_LU_WAVES (NP <= 56 ? 3 : 2)   // k_lu: wavefronts per SIMD the register allocation aims at (168 / 256 registers)
#endif
// column k + d of elimination step k: pivot-row entry through the LDS crossbar (true) or v_readlane (false)
constexpr bool lu_via_lds(const int d)
{
    return PJ_LU_BPERM == 1 || (PJ_LU_BPERM == 2 && (d & 1) == 0) || (PJ_LU_BPERM == 4 && d % 3 == 0) ||
           (PJ_LU_BPERM == 5 && (d % 5 == 0 || d % 5 == 2)) || (PJ_LU_BPERM == 6 && d % 4 == 0);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double lu_dpp_max(const double v)
{
    // Every row takes part (ROWMASK 0xf): lanes without a source lane read 0.0, which never beats a candidate
    // (|a| >= 0) -- no `old` operand, i.e. no register copies in front of the DPP moves.  Otherwise lanes in rows outside
    // ROWMASK keep their own value: max(v, v) = v.
    const long long u = __double_as_longlong(v);
    int lo, hi;
    if constexpr (ROWMASK == 0xf) {
        lo = __builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp((int)(unsigned)u, (int)(unsigned)u, CTRL, ROWMASK, 0xf, false);
        hi = __builtin_amdgcn_update_dpp((int)(unsigned)(u >> 32), (int)(unsigned)(u >> 32), CTRL, ROWMASK, 0xf, false);
    }
    const double o = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));     // (fmax() would canonicalise both operands first)
    return r;
}

// the same step on unsigned keys: v_max_u32 takes the DPP operand itself (lanes without a source lane read 0, the
// identity; rows outside ROWMASK keep their value)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned lu_dpp_umax(const unsigned v)
{
    unsigned o;
    if constexpr (ROWMASK == 0xf) o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
    else o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xf, false);
    return v > o ? v : o;
}

// 1 / u correctly rounded in all but pathological cases (v_rcp_f64 + three Newton steps: 8 instructions instead of
// the 25 of an IEEE division sequence); LAPACK's dgetf2 scales the column by the reciprocal of the pivot as well
__device__ __forceinline__ double lu_rcp(const double u)
{
    double r = __builtin_amdgcn_rcp(u);
    r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
    return r;
}
// b / u from r = 1 / u with one residual correction
__device__ __forceinline__ double lu_div(const double b, const double u, const double r)
{
    const double q = b * r;
    return __builtin_fma(__builtin_fma(-q, u, b), r, q);
}

// maximum over the 64 lanes (values >= -1, no NaN handling beyond fmax's): row_shr 1, 2, 4, 8 inside the rows of 16,
// then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3: lane 63 holds the maximum
__device__ __forceinline__ double lu_wave_max(double v)
{
    v = lu_dpp_max<0x111, 0xf>(v);
    v = lu_dpp_max<0x112, 0xf>(v);
    v = lu_dpp_max<0x114, 0xf>(v);
    v = lu_dpp_max<0x118, 0xf>(v);
    v = lu_dpp_max<0x142, 0xa>(v);
    v = lu_dpp_max<0x143, 0xc>(v);
    return lu_readlane(v, 63);
}

#ifndef PJ_LU4_ONLY
// NP: NSP rounded up to a multiple of 8, or 54 for 53 / 54 rows (rows / columns beyond NSP are the identity's: they are never pivots of a
// real column and contribute zeros -- for FINITE input: the padding is formed as 0 * (a clamped copy of the last row /
// column) + delta_ij, so an Inf / NaN there puts NaN into the padding as well; a block with non-finite entries gives
// non-finite factors either way, only which of them are NaN differs from a select-based padding).  mode: LU_FACTOR (A -> lu, perm), LU_FACTOR | LU_SOLVE (A, b -> x, and lu / perm
// if given), LU_PREFACTORED | LU_SOLVE (lu, perm, b -> x).  gamma != 0: the matrix is I - gamma A (the Newton
// matrix of an implicit step).
template <int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PJ_LU_WAVES))) k_lu(const int nsp, const long n, const double* A, const LuLay Y, const double gamma,
                                            double* lu, int* __restrict__ perm, const double* __restrict__ b,
                                            double* __restrict__ x, const int mode)
{
    const int lane = (int)(threadIdx.x & 63);
    const long ne = (long)nsp * nsp;
    const int lane0 = lane, nsp0 = nsp;
    // Blocks s .. s + 3 go to the four wavefronts of a workgroup, and the four workgroups that take the 16 blocks of a
    // run are b, b + 8, b + 16, b + 24: dispatched together and -- round-robin over the 8 XCDs -- onto the same XCD, so
    // that in the batch layout the 16 states of a 128-byte line are fetched into one L2 once and hit there three more
    // times (one workgroup taking the four quarters of a run one after the other: 2.5x the algorithmic HBM bytes,
    // the lines are evicted in between).  A wavefront past the end repeats the last block -- the batch-layout path
    // below has workgroup barriers -- and stores nothing.
    __shared__ double stage[4][8][64];
    const long nvirt = ((n + 15) / 16 + 7) / 8 * 32;          // runs padded to a multiple of 8, four workgroups each
    for (long v = blockIdx.x; v < nvirt; v += gridDim.x) {
        const long slot = v >> 3;
        const long s4_ = 16 * ((slot >> 2) * 8 + (v & 7)) + 4 * (slot & 3);
        if (s4_ >= n) continue;                               // (the whole workgroup)
        const long s_ = s4_ + (threadIdx.x >> 6);
        const bool valid = s_ < n;
        const long s = valid ? s_ : n - 1;
        // Every predicate of the body (j < nsp, lane == j, lane > k ...) is invariant across matrices, and the
        // optimiser knows: it computes hundreds of lane masks once, in front of the loop, and spills them.  Opaque
        // copies of `nsp` and `lane` per phase keep each predicate next to its use.
        int nsp = nsp0, lane = lane0;
        asm volatile("" : "+s"(nsp), "+v"(lane));
        const bool act = lane < nsp;
        double a[NP];
        const bool pre_ = (mode & LU_PREFACTORED) != 0;
        const long a_si = pre_ ? 1 : Y.a_si;
        const double* As = pre_ ? lu + s * ne : A + s * Y.a_ss;
        // every load is issued before anything depends on one (clamped, always valid addresses instead of
        // predicates: a branch per column would serialise 53 memory round trips); the identity padding and the Newton
        // matrix I - gamma A are applied afterwards, branch-free
        const int lane_c = lane < nsp ? lane : nsp - 1;
        if (!pre_ && Y.a_ss == 1) {
            // Batch layout: entry (r, c) of the workgroup's four blocks is 32 contiguous bytes, a lane-per-row load
            // would touch a different line in every lane.  The four wavefronts fetch eight columns of the four
            // blocks together (a thread: one entry of one block, lanes 4 e .. 4 e + 3 = the four states) into LDS
            // and each wavefront picks its block's values up from there.
            const long s4 = s_ & ~3L;                         // first block of this round's four
            const int tid = (int)threadIdx.x, w4 = tid & 3, e0 = tid >> 2;
            const long sl = s4 + w4 < n ? s4 + w4 : n - 1;
            // (the loads of chunk c + 1 are issued before chunk c goes through its two barriers and the LDS: one memory
            // round trip per chunk off the critical path)
            constexpr int NCH = (NP + 7) / 8;
            double v0[8], v1[8];
            // a thread's eight entries of a chunk: row e0 of columns j0 .. j0 + 7 -- a per-thread base and uniform
            // column offsets (scalar arithmetic; the general index expression cost ~10 vector instructions per load)
            const double* const Ab = A + (long)(e0 < nsp ? e0 : nsp - 1) * Y.a_si + sl;
            const long cs = (long)nsp * Y.a_si;
            auto fetch = [&](auto cc, double (&v)[8]) {
                constexpr int j0 = 8 * decltype(cc)::value;
#pragma unroll
                for (int q = 0; q < 8; ++q) {                  // 8 columns x 64 rows = 512 entries, 64 per pass
                    const int c = j0 + q;
                    v[q] = Ab[(long)(c < nsp ? c : nsp - 1) * cs];
                }
            };
            fetch(std::integral_constant<int, 0>{}, v0);
            lu_for<0, NCH>([&](auto cc) {
                constexpr int ci = decltype(cc)::value, j0 = 8 * ci;
                double (&v)[8] = (ci & 1) ? v1 : v0;
                if constexpr (ci + 1 < NCH) fetch(std::integral_constant<int, ci + 1>{}, (ci & 1) ? v0 : v1);
                __syncthreads();                               // the previous chunk has been picked up
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = e0 + 64 * q;
                    stage[w4][e >> 6][e & 63] = v[q];
                }
                __syncthreads();
                lu_for<0, 8>([&](auto jj) {
                    constexpr int j = j0 + decltype(jj)::value;
                    if constexpr (j < NP) a[j] = stage[threadIdx.x >> 6][decltype(jj)::value][lane_c];
                });
            });
        } else {
            lu_for<0, NP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                a[j] = As[(lane_c + (long)nsp * (j < nsp ? j : nsp - 1)) * a_si];
            });
        }
        {
            // a -> sh * delta_ij + sc * a in the block, delta_ij in the padding: one FMA per entry with a per-lane scale
            // (0 in the padding rows, whose loads are clamped copies) and a diagonal term picked by its upper word
            // (0.0 and 1.0 share the lower one)
            const bool newton = !(mode & LU_PREFACTORED) && gamma != 0.0;
            const double sc = newton ? -gamma : 1.0, sh = newton ? 1.0 : 0.0;
            const double sc_l = act ? sc : 0.0;
            const int dg_l = act ? (int)((unsigned long long)__double_as_longlong(sh) >> 32) : 0x3ff00000;
            lu_for<0, NP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const bool col = j < NP - 7 || j < nsp;          // (uniform; compile-time for all but seven columns)
                const double dj = __hiloint2double(j == lane ? (col ? dg_l : 0x3ff00000) : 0, 0);
                a[j] = __builtin_fma(col ? sc_l : 0.0, a[j], dj);
            });
        }
        // pos: the position this lane's row was chosen for (-1: not yet); a prefactored block is read row by position
        int pos = (mode & LU_PREFACTORED) ? lane : -1;
        double bb = 0.0, myinv = 1.0;             // myinv: 1 / u_kk in the lane whose row became row k
        if (mode & LU_SOLVE) {
            if (mode & LU_PREFACTORED) bb = act ? b[perm[s * nsp + lane] * Y.v_si + s * Y.v_ss] : 0.0;
            else bb = act ? b[lane * Y.v_si + s * Y.v_ss] : 0.0;
        }
        asm volatile("" : "+s"(nsp), "+v"(lane));
#if PJ_LU_LOOKAHEAD
        if (!(mode & LU_PREFACTORED)) {
            // The pivot of column k + 1 depends on step k through that one column only: step k updates it first and
            // the search for its pivot -- a chain of ~25 dependent instructions (magnitude key, six DPP maxima,
            // ballot, 1 / u from v_rcp_f64 + three Newton steps) -- is issued in slices between the updates of the
            // other columns, which hide its latency (two wavefronts per SIMD cannot).  The search runs on the upper
            // 32 bits of |a| as an unsigned key (v_max_u32 with a DPP operand: one instruction per reduction step
            // instead of three); keys that tie (one in 2^20 per pair of rows, or exact ties) are settled by the
            // same reduction on the lower 32 bits of the tying rows, so the choice is dgetf2's as before.
            // the rows not chosen yet as a wavefront mask: lane predicates come out of it for free (inverse ballot)
            unsigned long long openmask = nsp >= 64 ? ~0ull : (1ull << nsp) - 1ull;
            unsigned key = 0, kmax = 0;
            int pn = 0;
            double ukkn = 1.0, invn = 1.0;
            auto stage = [&](auto knc, auto sc_) {
                constexpr int kn = decltype(knc)::value, S = decltype(sc_)::value;
                if constexpr (kn < NP) {
                    if constexpr (S == 0) {
                        const bool open = __builtin_amdgcn_inverse_ballot_w64(openmask);
                        const unsigned hi = (unsigned)((unsigned long long)__double_as_longlong(a[kn]) >> 32) & 0x7fffffffu;
                        key = (open & (a[kn] == a[kn])) ? hi : 0u;         // a NaN never beats a number
                        kmax = key;
                    } else if constexpr (S == 1) {
                        kmax = lu_dpp_umax<0x111, 0xf>(kmax);
                        kmax = lu_dpp_umax<0x112, 0xf>(kmax);
                    } else if constexpr (S == 2) {
                        kmax = lu_dpp_umax<0x114, 0xf>(kmax);
                        kmax = lu_dpp_umax<0x118, 0xf>(kmax);
                    } else if constexpr (S == 3) {
                        kmax = lu_dpp_umax<0x142, 0xa>(kmax);
                        kmax = lu_dpp_umax<0x143, 0xc>(kmax);
                    } else if constexpr (S == 4) {
                        const unsigned mx = (unsigned)__builtin_amdgcn_readlane((int)kmax, 63);
                        unsigned long long hit = __builtin_amdgcn_ballot_w64(key == mx) & openmask;
                        if (__builtin_popcountll(hit) > 1) {        // (wavefront-uniform, rare)
                            const bool in = __builtin_amdgcn_inverse_ballot_w64(hit);
                            // the same reduction on the lower 32 bits of the tying rows
                            const unsigned k2 = (in && a[kn] == a[kn]) ? (unsigned)(unsigned long long)__double_as_longlong(a[kn]) : 0u;
                            unsigned m2 = k2;
                            m2 = lu_dpp_umax<0x111, 0xf>(m2);
                            m2 = lu_dpp_umax<0x112, 0xf>(m2);
                            m2 = lu_dpp_umax<0x114, 0xf>(m2);
                            m2 = lu_dpp_umax<0x118, 0xf>(m2);
                            m2 = lu_dpp_umax<0x142, 0xa>(m2);
                            m2 = lu_dpp_umax<0x143, 0xc>(m2);
                            const unsigned mx2 = (unsigned)__builtin_amdgcn_readlane((int)m2, 63);
                            hit = __builtin_amdgcn_ballot_w64(in && k2 == mx2);      // (never empty)
                        }
                        pn = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit | (1ull << 63)));
                        ukkn = lu_readlane(a[kn], pn);
                    } else if constexpr (S == 5) {
                        invn = __builtin_amdgcn_rcp(ukkn);
                        invn = __builtin_fma(__builtin_fma(-ukkn, invn, 1.0), invn, invn);
                    } else if constexpr (S == 6) {
                        invn = __builtin_fma(__builtin_fma(-ukkn, invn, 1.0), invn, invn);
                        invn = __builtin_fma(__builtin_fma(-ukkn, invn, 1.0), invn, invn);
                    }
                }
            };
            constexpr int NSTAGE = 7;
            lu_for<0, NSTAGE>([&](auto sc_) { stage(std::integral_constant<int, 0>{}, sc_); });
            lu_for<0, NP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nsp) {                                   // wavefront-uniform
                    const int p = pn;
                    const double inv = invn;
                    const bool me = __builtin_amdgcn_inverse_ballot_w64(1ull << (p & 63));
                    pos = me ? k : pos;
                    myinv = me ? inv : myinv;
                    openmask &= ~(1ull << (p & 63));
                    const bool below = __builtin_amdgcn_inverse_ballot_w64(openmask);   // rows not chosen yet: eliminated by this pivot
                    const double l = below ? a[k] * inv : 0.0;
                    if (below) a[k] = l;
                    // slices of the next search after columns k + 1, k + 1 + SP, k + 1 + 2 SP ...
                    constexpr int NC = NP - 1 - k;
                    constexpr int SP = NC >= 4 * NSTAGE ? 4 : NC >= 2 * NSTAGE ? 2 : 1;
                    lu_for<k + 1, NP>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const double ukj = lu_via_lds(j - k) ? lu_bpermute(a[j], p * 4) : lu_readlane(a[j], p);
                        a[j] = __builtin_fma(-l, ukj, a[j]);     // l = 0 in the rows already chosen
                        constexpr int d = j - k - 1;
                        if constexpr (d % SP == 0 && d / SP < NSTAGE) {
                            __builtin_amdgcn_sched_barrier(0);
                            stage(std::integral_constant<int, k + 1>{}, std::integral_constant<int, d / SP>{});
                            __builtin_amdgcn_sched_barrier(0);
                        } else if constexpr (((j - k) & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                    });
                    constexpr int DONE = NC <= 0 ? 0 : ((NC - 1) / SP + 1 < NSTAGE ? (NC - 1) / SP + 1 : NSTAGE);
                    lu_for<DONE, NSTAGE>([&](auto sc_) { stage(std::integral_constant<int, k + 1>{}, sc_); });
                    // forward substitution rides along: y_k is the pivot row's right-hand side
                    if (mode & LU_SOLVE) {
                        const double yk = lu_readlane(bb, p);
                        bb = __builtin_fma(-l, yk, bb);
                    }
                }
            });
#else
        if (!(mode & LU_PREFACTORED)) {
            lu_for<0, NP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nsp) {                                   // wavefront-uniform
                    const bool open = pos < 0 && act;           // rows not chosen yet (nsp - k of them)
                    const double cand = open ? fabs(a[k]) : -1.0;
                    const double mx = lu_wave_max(cand);
                    const unsigned long long avail = __builtin_amdgcn_ballot_w64(open);
                    unsigned long long hit = __builtin_amdgcn_ballot_w64(cand == mx) & avail;
                    if (hit == 0) hit = avail;                   // a column of NaNs: any open row, never a chosen one
                    const int p = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit));
                    if (lane == p) pos = k;
                    const double ukk = lu_readlane(a[k], p);
                    const double inv = lu_rcp(ukk);
                    if (lane == p) myinv = inv;
                    const bool below = pos < 0;                  // rows not chosen yet: eliminated by this pivot
                    const double l = below ? a[k] * inv : 0.0;
                    if (below) a[k] = l;
                    lu_for<k + 1, NP>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        // (the VALU issues the v_readlane pairs and the updates, the LDS pipeline the other broadcasts)
                        const double ukj = lu_via_lds(j - k) ? lu_bpermute(a[j], p * 4) : lu_readlane(a[j], p);
                        a[j] = __builtin_fma(-l, ukj, a[j]);     // l = 0 in the rows already chosen
                        // the broadcasts of a step are independent of its updates: left alone the scheduler issues
                        // them all first and spills a thousand scalar registers; four columns at a time
                        if constexpr (((j - k) & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                    });
                    // forward substitution rides along: y_k is the pivot row's right-hand side
                    if (mode & LU_SOLVE) {
                        const double yk = lu_readlane(bb, p);
                        bb = __builtin_fma(-l, yk, bb);
                    }
                }
            });
#endif
        } else {
            // L y = P b, column by column: position k's lane is lane k
            lu_for<0, NP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nsp) {
                    const double yk = lu_readlane(bb, k);
                    if (lane > k) bb = __builtin_fma(-a[k], yk, bb);
                }
            });
        }
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (!(mode & LU_PREFACTORED) && lu != nullptr && act && valid) {
            double* Ls = lu + s * ne;
            lu_for<0, NP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < nsp) Ls[pos + (long)nsp * j] = a[j];
            });
            if (perm != nullptr) perm[s * nsp + pos] = lane;
        }
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (mode & LU_SOLVE) {
            // U x = y, last column first: the lane at position k owns x_k
            lu_for<0, NP>([&](auto kr) {
                constexpr int k = NP - 1 - decltype(kr)::value;
                if (k < nsp) {
                    const unsigned long long own = __builtin_amdgcn_ballot_w64(pos == k);
                    const int p = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(own));
                    // every lane divides its own right-hand side by its own a[k] (u_kk in the owner lane, whose
                    // quotient is the one that is broadcast)
                    if (mode & LU_PREFACTORED) myinv = lu_rcp(a[k]);
                    const double xk = lu_readlane(lu_div(bb, a[k], myinv), p);
                    if (pos == k) bb = xk;
                    else if (pos < k && pos >= 0) bb = __builtin_fma(-a[k], xk, bb);
                }
            });
            if (act && valid) x[pos * Y.v_si + s * Y.v_ss] = bb;
        }
    }
}


// ---- blocks of 65 .. LU_MAX_LDS rows: the matrix in LDS, one workgroup per block -------------------------------
// A lane per row stops at 64 rows.  Larger blocks (the 111-species mechanisms) take the textbook right-looking
// factorisation on an LDS-resident copy: column-major with an odd leading dimension (bank-conflict free down a
// column), pivot search by a wavefront argmax per 64 rows + a four-entry exchange, physical row exchange, 32 x 8
// thread tiles over the trailing block with the multipliers of the thread's rows in registers.  Same results and
// same (lu, perm) convention as k_lu; latency-bound by its ~4 barriers per column.

__global__ void __launch_bounds__(256) k_lu_lds(const int nsp, const long n, const double* A, const LuLay Y, const double gamma,
                                                double* lu, int* __restrict__ perm, const double* __restrict__ b,
                                                double* __restrict__ x, const int mode)
{
    extern __shared__ double lds_lu[];
    const int ld = nsp | 1;
    double* const M = lds_lu;                       // M[c * ld + r]
    double* const bv = M + (long)ld * nsp;          // right-hand side / solution
    double* const red_v = bv + nsp;                 // per-wavefront pivot candidates
    int* const red_i = (int*)(red_v + 4);
    int* const pm = red_i + 4;                      // row permutation
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long ne = (long)nsp * nsp;
    const bool pre = (mode & LU_PREFACTORED) != 0, solve = (mode & LU_SOLVE) != 0;
    // workgroups b, b + 8, ... b + 120 sit on one XCD (round-robin dispatch) and take 16 consecutive blocks: in the
    // batch layout a 128-byte line of 16 states is fetched into one L2 instead of eight
    for (long t = blockIdx.x; t < ((n + 127) / 128) * 128; t += gridDim.x) {
        const long s = (t / 128) * 128 + 16 * (t % 8) + (t % 128) / 8;
        if (s >= n) continue;                            // (uniform in the workgroup)
        const long a_si = pre ? 1 : Y.a_si;
        const double* As = pre ? lu + s * ne : A + s * Y.a_ss;
        const bool newton = !pre && gamma != 0.0;
        for (int c = wave; c < nsp; c += 4) {              // a wavefront per column: no index divisions
            const double* Ac = As + (long)c * nsp * a_si;
            double* Mc = M + c * ld;
            for (int r = lane; r < nsp; r += 64) {
                double v = Ac[r * a_si];
                if (newton) v = (r == c ? 1.0 : 0.0) - gamma * v;
                Mc[r] = v;
            }
        }
        if (tid < nsp) {
            pm[tid] = pre ? perm[s * nsp + tid] : tid;
        }
        __syncthreads();
        if (solve && tid < nsp) bv[tid] = b[(pre ? pm[tid] : tid) * Y.v_si + s * Y.v_ss];
        __syncthreads();
        if (!pre) {
            // pivot of column c0 among rows c0 .. nsp-1, by the lower half of wavefront 0 (lane tr: rows c0 + tr + 32 q --
            // for c0 = k + 1 exactly the entries those lanes have just updated): maximum magnitude, ties to the lowest
            // row.  Left in red_i[0] for the next step: no barrier, no pivot-column read for the other wavefronts.
            auto search = [&](const int c0) {
                double best = -1.0; int brow = c0;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int r = c0 + (tid & 31) + 32 * q;
                    if (tid < 32 && r < nsp) {
                        const double v = fabs(M[c0 * ld + r]);
                        if (v > best) { best = v; brow = r; }
                    }
                }
                const double mx = lu_wave_max(best);
                // among the lanes that hold the maximum the LOWEST ROW wins (a lane scans rows c0 + tr + 32 q, so the
                // lowest lane need not hold the lowest row): the first row of maximum magnitude, as k_lu and dgetf2
                const bool mine = best == mx && best >= 0.0;
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(mine);
                if (hit == 0) { if (tid == 0) red_i[0] = c0; }             // a column of NaNs
                else {
                    int lowest;
                    if (__builtin_popcountll(hit) == 1) {        // (uniform; the usual case: no tie, no reduction)
                        lowest = __builtin_amdgcn_readlane(brow, __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit)));
                    } else {
                        lowest = mine ? brow : 0x7fffffff;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) {
                            const int o = __shfl_xor(lowest, off, 64);
                            lowest = o < lowest ? o : lowest;
                        }
                    }
                    if (lane == 0) red_i[0] = lowest;
                }
            };
            if (wave == 0) search(0);
            __syncthreads();
            for (int k = 0; k < nsp; ++k) {
                const int p = red_i[0];
                if (p != k && !(PJ_LU_ABL & 2)) {
                    if (tid < nsp) { const double t = M[tid * ld + k]; M[tid * ld + k] = M[tid * ld + p]; M[tid * ld + p] = t; }
                    if (tid == 255) { const int t = pm[k]; pm[k] = pm[p]; pm[p] = t; }
                    if (solve && tid == 254) { const double t = bv[k]; bv[k] = bv[p]; bv[p] = t; }
                }
                __syncthreads();
                const double inv = lu_rcp(M[k * ld + k]);
                const double bk = solve ? bv[k] : 0.0;
                // multipliers of this thread's rows (kept for the trailing update), forward substitution rides along
                const int tr = tid & 31, tc = tid >> 5;
                double lr[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const int r = k + 1 + tr + 32 * q;
                    lr[q] = r < nsp ? M[k * ld + r] * inv : 0.0;
                }
                __syncthreads();            // pivot, b_k and every multiplier read before column k and b change
                if (tc == 0) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        const int r = k + 1 + tr + 32 * q;
                        if (r < nsp) { M[k * ld + r] = lr[q]; if (solve) bv[r] = __builtin_fma(-lr[q], bk, bv[r]); }
                    }
                }
                {
                    // Trailing update.  Rows k + 1 + tr + 32 q, q < NQ with NQ the same for every thread (a switch on
                    // the rows still in play: compile-time trip shapes); reads are unconditional -- a row past the end
                    // reads the last row instead -- so that all LDS reads of a trip (three columns x NQ rows + three
                    // pivot-row entries) are in flight before the first multiply; only the writes are predicated.
                    // (Per-thread row counts as predicates around the reads: 178 ms per 2e5 111 x 111 blocks; one
                    // column per trip: 206 ms.)
                    const int step8 = 8 * ld;
                    const int r0 = k + 1 + tr;
                    auto upd = [&](auto nqc) {
                        constexpr int NQ = decltype(nqc)::value;
                        int off[NQ]; bool ok[NQ];
#pragma unroll
                        for (int q = 0; q < NQ; ++q) { ok[q] = r0 + 32 * q < nsp; off[q] = ok[q] ? 32 * q : nsp - 1 - r0; }
                        double* col = M + (k + 1 + tc) * ld + r0;
                        int c = (PJ_LU_ABL & 1) ? nsp : k + 1 + tc;
                        // (written out on purpose: the same trip as nested generic lambdas over NC columns compiles to
                        // code that is 10 % (111 rows) to 80 % (65 rows) slower, and 4 / 6 / 8 columns per trip gain nothing)
                        for (; c + 16 < nsp; c += 24, col += 3 * step8) {
                            double* const c1 = col + step8;
                            double* const c2 = col + 2 * step8;
                            const double u0 = col[-(1 + tr)], u1 = c1[-(1 + tr)], u2 = c2[-(1 + tr)];
                            double v0[NQ], v1[NQ], v2[NQ];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) { v0[q] = col[off[q]]; v1[q] = c1[off[q]]; v2[q] = c2[off[q]]; }
#pragma unroll
                            for (int q = 0; q < NQ; ++q)
                                if (ok[q]) {
                                    col[32 * q] = __builtin_fma(-lr[q], u0, v0[q]);
                                    c1[32 * q] = __builtin_fma(-lr[q], u1, v1[q]);
                                    c2[32 * q] = __builtin_fma(-lr[q], u2, v2[q]);
                                }
                        }
                        for (; c < nsp; c += 8, col += step8) {
                            const double ukc = col[-(1 + tr)];
                            double v[NQ];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) v[q] = col[off[q]];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) if (ok[q]) col[32 * q] = __builtin_fma(-lr[q], ukc, v[q]);
                        }
                    };
                    switch ((nsp - (k + 1) + 31) >> 5) {
                    case 5: upd(std::integral_constant<int, 5>{}); break;
                    case 4: upd(std::integral_constant<int, 4>{}); break;
                    case 3: upd(std::integral_constant<int, 3>{}); break;
                    case 2: upd(std::integral_constant<int, 2>{}); break;
                    case 1: upd(std::integral_constant<int, 1>{}); break;
                    default: break;
                    }
                }
                if (wave == 0 && k + 1 < nsp && !(PJ_LU_ABL & 4)) search(k + 1);       // (its lanes read back what they wrote themselves)
                __syncthreads();
            }
            if (lu != nullptr) {
                double* Ls = lu + s * ne;
                for (int c = wave; c < nsp; c += 4)
                    for (int r = lane; r < nsp; r += 64) Ls[c * nsp + r] = M[c * ld + r];
                if (perm != nullptr && tid < nsp) perm[s * nsp + tid] = pm[tid];
            }
        } else if (solve) {
            for (int k = 0; k < nsp; ++k) {          // L y = P b
                const double yk = bv[k];
                __syncthreads();
                const int r = k + 1 + tid;
                if (r < nsp) bv[r] = __builtin_fma(-M[k * ld + r], yk, bv[r]);
                __syncthreads();
            }
        }
        if (solve) {
            for (int k = nsp - 1; k >= 0; --k) {     // U x = y
                const double ukk = M[k * ld + k];
                const double xk = lu_div(bv[k], ukk, lu_rcp(ukk));
                __syncthreads();
                if (tid == k) bv[k] = xk;
                else if (tid < k) bv[tid] = __builtin_fma(-M[k * ld + tid], xk, bv[tid]);
                __syncthreads();
            }
            if (tid < nsp) x[tid * Y.v_si + s * Y.v_ss] = bv[tid];
        }
        __syncthreads();
    }
}

#endif  // !PJ_LU4_ONLY

// ---- blocks of 65 .. 128 rows: FOUR wavefronts per block, the matrix in registers ------------------------------------
// k_lu_lds keeps the block in LDS and pays ~4 barriers and an LDS round trip per entry and column: 168 ms per 2e5
// 111 x 111 blocks, 0.03 of the HBM roofline, 27x the time of the Jacobians it consumes.  Here the
// workgroup's four wavefronts each hold EVERY row of a quarter of the columns: lane l of wavefront w holds rows l and
// l + RH (RH = NP / 2) of the columns 4 c + w, c < NC = NP / 4 -- 2 NC doubles per lane (112 registers for 111 x 111),
// compile-time indices, columns dealt cyclically so that all four wavefronts have work until the last step.  Since a
// wavefront sees whole columns, the pivot search of column k is local to its owner (k mod 4), and since it sees every
// row of its columns, the pivot row's entries reach its lanes without leaving the wavefront (one lane parks them in
// the wavefront's LDS strip, 16 bytes at a time; the others read them back at a uniform address).  What has to cross
// wavefronts is the multiplier column and the pivot's position: ONE barrier per elimination step, and the owner of
// column k + 1 updates that column first, searches it and publishes step k + 1's multipliers while the others are
// still updating (double-buffered), so the search is off the critical path.
// (First version, round 5: wavefronts = row halves x column parity, two barriers per step, candidates exchanged through
// LDS: 50.8 ms per 2e5 111 x 111 factorisations.)
// Rows are never exchanged (implicit pivoting: a row remembers its position; ties go to the lowest original row, as
// k_lu); the right-hand side and the positions are replicated in the four wavefronts, so forward substitution needs
// no exchange; the back substitution exchanges one partial sum per wavefront and step (one barrier).  Same (lu, perm)
// convention as k_lu / k_lu_lds; stored factors are solved by k_lu_lds (LU_PREFACTORED).
template <int NP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_lu4(const int nsp, const long n, const double* A, const LuLay Y, const double gamma, double* lu, int* __restrict__ perm,
      const double* __restrict__ b, double* __restrict__ x, const int mode)
{
    static_assert(NP % 8 == 0 && NP <= 128, "k_lu4: up to 128 rows");
    constexpr int RH = NP / 2, NC = NP / 4;
    __shared__ __attribute__((aligned(16))) double prow[4][NC + (NC & 1)];     // a wavefront's share of the pivot row
    __shared__ double lcol[2][2][64];                                          // multipliers of a step: [buffer][row set][lane]
    __shared__ double hdr_inv[2];
    __shared__ int hdr_lp[2], hdr_rs[2];
    __shared__ double part[2][4];
    const int tid = (int)threadIdx.x, lane0 = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long ne = (long)nsp * nsp;
    const bool solve = (mode & LU_SOLVE) != 0;
    const int nsp0 = nsp;
    // workgroups b, b + 8, ... b + 120 sit on one XCD (round-robin dispatch) and take 16 consecutive blocks: in the
    // batch layout a 128-byte line of 16 states is fetched into one L2 instead of eight (as k_lu_lds)
    for (long t = blockIdx.x; t < ((n + 127) / 128) * 128; t += gridDim.x) {
        const long s = (t / 128) * 128 + 16 * (t % 8) + (t % 128) / 8;
        if (s >= n) continue;                            // (uniform in the workgroup)
        int nsp = nsp0, lane = lane0;
        asm volatile("" : "+s"(nsp), "+v"(lane));
        const int row0 = lane, row1 = lane + RH;
        const bool act0 = lane < RH && row0 < nsp, act1 = lane < RH && row1 < nsp;
        double a0[NC], a1[NC];
        {
            const double* As = A + s * Y.a_ss;
            const int r0c = act0 ? row0 : nsp - 1, r1c = act1 ? row1 : nsp - 1;
            // all loads first (clamped addresses instead of predicates), then identity padding / I - gamma A branch-free
            lu_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int g = 4 * c + w;
                const long col = (long)nsp * (g < nsp ? g : nsp - 1);
                a0[c] = As[(r0c + col) * Y.a_si];
                a1[c] = As[(r1c + col) * Y.a_si];
            });
            const bool newton = gamma != 0.0;
            const double sc = newton ? -gamma : 1.0, sh = newton ? 1.0 : 0.0;
            lu_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int g = 4 * c + w;
                const double id0 = (g == row0 && lane < RH) ? 1.0 : 0.0, id1 = (g == row1 && lane < RH) ? 1.0 : 0.0;
                a0[c] = (act0 && g < nsp) ? __builtin_fma(sc, a0[c], sh * id0) : id0;
                a1[c] = (act1 && g < nsp) ? __builtin_fma(sc, a1[c], sh * id1) : id1;
            });
        }
        int pos0 = -1, pos1 = -1;
        double bb0 = 0.0, bb1 = 0.0, inv0 = 1.0, inv1 = 1.0;      // inv: 1 / u_kk in the lane whose row became row k
        if (solve) {
            bb0 = act0 ? b[row0 * Y.v_si + s * Y.v_ss] : 0.0;
            bb1 = act1 ? b[row1 * Y.v_si + s * Y.v_ss] : 0.0;
        }
        // rows not chosen yet, per row set, as wavefront masks
        auto rows_mask = [&](const int first) {
            const int cnt = nsp - first;
            const int m = cnt < 0 ? 0 : cnt > RH ? RH : cnt;
            return m >= 64 ? ~0ull : (1ull << m) - 1ull;
        };
        unsigned long long open0 = rows_mask(0), open1 = rows_mask(RH);

        // pivot of the column whose entries are (c0, c1) among the open rows; publishes the step's multipliers
        auto search_publish = [&](double& c0, double& c1, const unsigned long long o0, const unsigned long long o1, const int buf) {
            const bool op0 = __builtin_amdgcn_inverse_ballot_w64(o0), op1 = __builtin_amdgcn_inverse_ballot_w64(o1);
            const unsigned long long u0 = (unsigned long long)__double_as_longlong(c0), u1 = (unsigned long long)__double_as_longlong(c1);
            const bool n0 = c0 == c0, n1 = c1 == c1;
            const unsigned k0 = (op0 & n0) ? (unsigned)(u0 >> 32) & 0x7fffffffu : 0u;      // a NaN never beats a number
            const unsigned k1 = (op1 & n1) ? (unsigned)(u1 >> 32) & 0x7fffffffu : 0u;
            unsigned kmax = k0 > k1 ? k0 : k1;
            kmax = lu_dpp_umax<0x111, 0xf>(kmax);
            kmax = lu_dpp_umax<0x112, 0xf>(kmax);
            kmax = lu_dpp_umax<0x114, 0xf>(kmax);
            kmax = lu_dpp_umax<0x118, 0xf>(kmax);
            kmax = lu_dpp_umax<0x142, 0xa>(kmax);
            kmax = lu_dpp_umax<0x143, 0xc>(kmax);
            const unsigned mx = (unsigned)__builtin_amdgcn_readlane((int)kmax, 63);
            unsigned long long h0 = __builtin_amdgcn_ballot_w64(k0 == mx) & o0, h1 = __builtin_amdgcn_ballot_w64(k1 == mx) & o1;
            if (__builtin_popcountll(h0) + __builtin_popcountll(h1) > 1) {      // (uniform, rare): the lower 32 bits decide
                const bool i0 = __builtin_amdgcn_inverse_ballot_w64(h0), i1 = __builtin_amdgcn_inverse_ballot_w64(h1);
                const unsigned q0 = (i0 && n0) ? (unsigned)u0 : 0u, q1 = (i1 && n1) ? (unsigned)u1 : 0u;
                unsigned m2 = q0 > q1 ? q0 : q1;
                m2 = lu_dpp_umax<0x111, 0xf>(m2);
                m2 = lu_dpp_umax<0x112, 0xf>(m2);
                m2 = lu_dpp_umax<0x114, 0xf>(m2);
                m2 = lu_dpp_umax<0x118, 0xf>(m2);
                m2 = lu_dpp_umax<0x142, 0xa>(m2);
                m2 = lu_dpp_umax<0x143, 0xc>(m2);
                const unsigned mx2 = (unsigned)__builtin_amdgcn_readlane((int)m2, 63);
                h0 = __builtin_amdgcn_ballot_w64(i0 && q0 == mx2);
                h1 = __builtin_amdgcn_ballot_w64(i1 && q1 == mx2);
            }
            // the first row of maximum magnitude in the original order: row set 0 (rows 0 .. RH-1) before row set 1
            const int rs = h0 ? 0 : 1;
            const unsigned long long hh = h0 ? h0 : (h1 ? h1 : (o0 ? o0 : o1));   // (a column of NaNs: any open row)
            const int rs_ = h0 ? 0 : (h1 ? 1 : (o0 ? 0 : 1));
            (void)rs;
            const int lp = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hh | (1ull << 63)));
            const double ukk = lu_readlane(rs_ ? c1 : c0, lp);
            const double inv = lu_rcp(ukk);
            // multipliers of the rows that stay open (kept as L in the column itself)
            const unsigned long long o0n = rs_ == 0 ? o0 & ~(1ull << lp) : o0, o1n = rs_ == 1 ? o1 & ~(1ull << lp) : o1;
            const bool b0 = __builtin_amdgcn_inverse_ballot_w64(o0n), b1 = __builtin_amdgcn_inverse_ballot_w64(o1n);
            const double l0 = b0 ? c0 * inv : 0.0, l1 = b1 ? c1 * inv : 0.0;
            if (b0) c0 = l0;
            if (b1) c1 = l1;
            lcol[buf][0][lane] = l0;
            lcol[buf][1][lane] = l1;
            if (lane == 0) { hdr_lp[buf] = lp; hdr_rs[buf] = rs_; hdr_inv[buf] = inv; }
        };
        if (w == 0) search_publish(a0[0], a1[0], open0, open1, 0);
        lu_for<0, NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            lu_for<0, 4>([&](auto qc) {
                constexpr int q = decltype(qc)::value, k = 4 * c + q, buf = k & 1;
                if (k < nsp) {                                   // (uniform)
                    __syncthreads();
                    const int lp = __builtin_amdgcn_readfirstlane(hdr_lp[buf]), rs = __builtin_amdgcn_readfirstlane(hdr_rs[buf]);
                    const double inv = hdr_inv[buf];
                    const double l0 = lcol[buf][0][lane], l1 = lcol[buf][1][lane];
                    const bool me = lane == lp;
                    if (rs == 0) { pos0 = me ? k : pos0; inv0 = me ? inv : inv0; open0 &= ~(1ull << lp); }
                    else { pos1 = me ? k : pos1; inv1 = me ? inv : inv1; open1 &= ~(1ull << lp); }
                    // this wavefront's share of the pivot row, right of column k: parked by the pivot's lane, 16 bytes at a time
                    constexpr int CF = c + 1;                    // columns c' >= CF are right of k in every wavefront; c' == c if w > q
                    constexpr int EV = CF + (CF & 1);
                    if (me) {
                        auto park = [&](const double (&ar)[NC]) {
                            if (w > q) prow[w][c] = ar[c];
                            if constexpr ((CF & 1) != 0 && CF < NC) prow[w][CF] = ar[CF];
                            lu_for<0, (NC - EV) / 2>([&](auto jc) {
                                constexpr int j = EV + 2 * decltype(jc)::value;
                                double2 v;
                                v.x = ar[j];
                                v.y = ar[j + 1];
                                *(double2*)&prow[w][j] = v;
                            });
                            if constexpr (((NC - EV) & 1) != 0) prow[w][NC - 1] = ar[NC - 1];
                        };
                        if (rs == 0) park(a0); else park(a1);
                    }
                    // (the pivot lane's stores sit in a divergent region; every lane of the SAME wavefront reads the strip next:
                    // a wavefront-scope release keeps the LDS accesses in program order whatever the compiler does with the branch)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (solve) {                                 // forward substitution rides along (replicated)
                        const double yk = lu_readlane(rs ? bb1 : bb0, lp);
                        bb0 = __builtin_fma(-l0, yk, bb0);
                        bb1 = __builtin_fma(-l1, yk, bb1);
                    }
                    // the owner of column k + 1 brings that column up to date first, finds its pivot and publishes step k + 1
                    constexpr int qn = (q + 1) & 3, cn = q == 3 ? c + 1 : c;
                    if constexpr (cn < NC) {
                        if (w == qn && k + 1 < nsp) {
                            const double u = prow[w][cn];
                            a0[cn] = __builtin_fma(-l0, u, a0[cn]);
                            a1[cn] = __builtin_fma(-l1, u, a1[cn]);
                            search_publish(a0[cn], a1[cn], open0, open1, buf ^ 1);
                        }
                    }
                    // the other columns right of k
                    if (w > q && !(cn == c && w == qn)) {
                        const double u = prow[w][c];
                        a0[c] = __builtin_fma(-l0, u, a0[c]);
                        a1[c] = __builtin_fma(-l1, u, a1[c]);
                    }
                    if constexpr ((CF & 1) != 0 && CF < NC) {
                        if (!(cn == CF && w == qn)) {
                            const double u = prow[w][CF];
                            a0[CF] = __builtin_fma(-l0, u, a0[CF]);
                            a1[CF] = __builtin_fma(-l1, u, a1[CF]);
                        }
                    }
                    lu_for<0, (NC - EV) / 2>([&](auto jc) {
                        constexpr int j = EV + 2 * decltype(jc)::value;
                        const double2 u = *(const double2*)&prow[w][j];
                        if (!(cn == j && w == qn)) {
                            a0[j] = __builtin_fma(-l0, u.x, a0[j]);
                            a1[j] = __builtin_fma(-l1, u.x, a1[j]);
                        }
                        if (!(cn == j + 1 && w == qn)) {
                            a0[j + 1] = __builtin_fma(-l0, u.y, a0[j + 1]);
                            a1[j + 1] = __builtin_fma(-l1, u.y, a1[j + 1]);
                        }
                    });
                    if constexpr (((NC - EV) & 1) != 0) {
                        if (!(cn == NC - 1 && w == qn)) {
                            const double u = prow[w][NC - 1];
                            a0[NC - 1] = __builtin_fma(-l0, u, a0[NC - 1]);
                            a1[NC - 1] = __builtin_fma(-l1, u, a1[NC - 1]);
                        }
                    }
                }
            });
        });
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (lu != nullptr) {
            double* Ls = lu + s * ne;
            lu_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int g = 4 * c + w;
                if (g < nsp) {
                    if (act0) Ls[pos0 + (long)nsp * g] = a0[c];
                    if (act1) Ls[pos1 + (long)nsp * g] = a1[c];
                }
            });
            if (perm != nullptr && w == 0) {
                if (act0) perm[s * nsp + pos0] = row0;
                if (act1) perm[s * nsp + pos1] = row1;
            }
        }
        if (solve) {
            // U x = y, last column first.  Every wavefront holds y and the positions; the products a[r][k'] x_k' of the columns it
            // owns accumulate in acc (per row).  Step k: the four partial sums of the row at position k meet in LDS (one
            // barrier), every wavefront forms x_k, the owner of column k adds its products.
            double acc0 = 0.0, acc1 = 0.0, xs0 = 0.0, xs1 = 0.0;
            lu_for<0, NC>([&](auto cr) {
                constexpr int c = NC - 1 - decltype(cr)::value;
                lu_for<0, 4>([&](auto qr) {
                    constexpr int q = 3 - decltype(qr)::value, k = 4 * c + q, buf = k & 1;
                    if (k < nsp) {
                        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(pos0 == k), m1 = __builtin_amdgcn_ballot_w64(pos1 == k);
                        const int rs = m0 ? 0 : 1;
                        const int lp = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll((m0 ? m0 : m1) | (1ull << 63)));
                        const double accr = lu_readlane(rs ? acc1 : acc0, lp);
                        if (lane == 0) part[buf][w] = accr;
                        __syncthreads();
                        const double sum = (part[buf][0] + part[buf][1]) + (part[buf][2] + part[buf][3]);
                        const double yk = lu_readlane(rs ? bb1 : bb0, lp), ik = lu_readlane(rs ? inv1 : inv0, lp);
                        const double xk = (yk - sum) * ik;
                        if (w == q) {
                            if (pos0 >= 0 && pos0 < k) acc0 = __builtin_fma(a0[c], xk, acc0);
                            if (pos1 >= 0 && pos1 < k) acc1 = __builtin_fma(a1[c], xk, acc1);
                        }
                        xs0 = pos0 == k ? xk : xs0;
                        xs1 = pos1 == k ? xk : xs1;
                    }
                });
            });
            if (w == 0) {
                if (act0) x[pos0 * Y.v_si + s * Y.v_ss] = xs0;
                if (act1) x[pos1 * Y.v_si + s * Y.v_ss] = xs1;
            }
        }
        __syncthreads();
    }
}

#ifdef PJ_LU4_ONLY
#define PJ_LU4_GO_DEF(NP_) void lu4_go_##NP_(unsigned blocks, hipStream_t st, int nsp, long n, const double* A, LuLay Y, double gamma, \
                                             double* lu, int* perm, const double* b, double* x, int mode) \
    { hipLaunchKernelGGL((k_lu4<NP_>), dim3(blocks), dim3(256), 0, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode); }
#define PJ_LU4_GO_DEF_(NP_) PJ_LU4_GO_DEF(NP_)
PJ_LU4_GO_DEF_(PJ_LU4_ONLY)
#else
// ---- blocks of up to 16 rows: four blocks per wavefront ---------------------------------------------------------
// A 10 x 10 block (the H2-size mechanisms) leaves 54 of k_lu's 64 lanes idle.  k_lu16 gives every block one DPP row
// of 16 lanes: lane = 16 g + i holds row i of block g, the pivot is a maximum over the row of lanes (four DPP
// rotations of unsigned magnitude keys: every lane ends up with it), the first lane that holds it comes out of the
// ballot's 16-bit field of the group, and the pivot row travels through the LDS crossbar (ds_bpermute_b32 with a per-lane source: each group
// reads its own pivot lane).  Same (lu, perm) results as k_lu.
// first lane of this lane's group of GW for which `pred` holds (the lane after the group if none does)
template <int GW>
__device__ __forceinline__ int lu_group_first(const bool pred, const int lane)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
    const int base = lane & ~(GW - 1);
    const unsigned long long f = (m >> base) & ((1ull << GW) - 1ull);
    return base + (int)__builtin_ctzll(f | (1ull << GW));
}
// maximum of unsigned keys over the group of GW lanes, in every lane of it: four rotations inside the DPP rows of 16
// (v_max_u32 takes the rotated operand itself); for groups of 32 the two rows of a group then exchange through
// v_permlane16_swap (odd rows of one copy <-> even rows of the other)
template <int GW>
__device__ __forceinline__ unsigned lu_group_umax(unsigned m)
{
    auto rot = [](const unsigned v, auto ctrl) {
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xf, 0xf, true);
        return v > o ? v : o;
    };
    m = rot(m, std::integral_constant<int, 0x121>{});      // row_ror:1, 2, 4, 8
    m = rot(m, std::integral_constant<int, 0x122>{});
    m = rot(m, std::integral_constant<int, 0x124>{});
    m = rot(m, std::integral_constant<int, 0x128>{});
    if constexpr (GW == 32) {
        const auto sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);
        m = sw[0] > sw[1] ? sw[0] : sw[1];
    }
    return m;
}

template <int NP, int GW>      // NP <= GW: 16 (four blocks per wavefront) or 32 (two)
__global__ void __launch_bounds__(256) k_lu16(const int nsp, const long n, const double* A, const LuLay Y, const double gamma,
                                              double* lu, int* __restrict__ perm, const double* __restrict__ b,
                                              double* __restrict__ x, const int mode)
{
    const int lane0 = (int)(threadIdx.x & 63);
    const long nw = (long)gridDim.x * 4;
    const long ne = (long)nsp * nsp;
    const int nsp0 = nsp;
    const bool pre = (mode & LU_PREFACTORED) != 0, solve = (mode & LU_SOLVE) != 0;
    constexpr int BPW = 64 / GW;
    // a workgroup takes 4 BPW consecutive blocks; the 16 / (4 BPW) workgroups of a 16-state run are b, b + 8, ... on one
    // XCD (as in k_lu): a 128-byte line of the batch layout is fetched into one L2 once (2.3x the bytes otherwise)
    constexpr int HPW = 16 / (4 * BPW);
    const long nvirt = ((n + 15) / 16 + 7) / 8 * 8 * HPW;
    (void)nw;
    for (long v = blockIdx.x; v < nvirt; v += gridDim.x) {
        const long slot = v >> 3;
        const long s_base = 16 * ((slot / HPW) * 8 + (v & 7)) + (slot % HPW) * (4 * BPW);
        if (s_base >= n) continue;
        int nsp = nsp0, lane = lane0;
        asm volatile("" : "+s"(nsp), "+v"(lane));
        const int i = lane & (GW - 1);
        const long s = s_base + (long)(threadIdx.x >> 6) * BPW + lane / GW;
        const bool act = i < nsp && s < n;
        const long sc = s < n ? s : n - 1;               // clamped: loads are unconditional
        const int ic = i < nsp ? i : nsp - 1;
        const long a_si = pre ? 1 : Y.a_si;
        const double* As = pre ? lu + sc * ne : A + sc * Y.a_ss;
        double a[NP];
        lu_for<0, NP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            a[j] = As[(ic + (long)nsp * (j < nsp ? j : nsp - 1)) * a_si];
        });
        {
            const bool newton = !pre && gamma != 0.0;
            const double scl = newton ? -gamma : 1.0, sh = newton ? 1.0 : 0.0;
            lu_for<0, NP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double id = (j == i) ? 1.0 : 0.0;
                a[j] = (act && j < nsp) ? __builtin_fma(scl, a[j], sh * id) : id;
            });
        }
        int pos = pre ? i : -1;
        double bb = 0.0, myinv = 1.0;
        if (solve) bb = act ? b[(pre ? perm[sc * nsp + ic] : ic) * Y.v_si + sc * Y.v_ss] : 0.0;
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (!pre) {
            lu_for<0, NP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nsp) {
                    // pivot search on unsigned keys, as in k_lu: the upper 32 bits of |a| (a NaN counts as 0, so the
                    // rows that tie with the group's maximum are never none while a row is open), ties settled on the
                    // lower 32 bits -- for every group at once if any group has one (wavefront-uniform, rare)
                    const bool open = (pos < 0) & act;
                    const unsigned long long ub = (unsigned long long)__double_as_longlong(a[k]);
                    const bool num = a[k] == a[k];
                    const unsigned key = (open & num) ? (unsigned)(ub >> 32) & 0x7fffffffu : 0u;
                    const unsigned km = lu_group_umax<GW>(key);          // the maximum of the group, in all of its lanes
                    const bool in = open & (key == km);
                    const int base = lane & ~(GW - 1);
                    constexpr unsigned long long GMASK = (1ull << GW) - 1ull;
                    unsigned long long f = (__builtin_amdgcn_ballot_w64(in) >> base) & GMASK;
                    if (__builtin_amdgcn_ballot_w64((f & (f - 1ull)) != 0ull) != 0ull) {
                        const unsigned k2 = (in & num) ? (unsigned)ub : 0u;
                        const unsigned k2m = lu_group_umax<GW>(k2);
                        f = (__builtin_amdgcn_ballot_w64(in & (k2 == k2m)) >> base) & GMASK;
                    }
                    int p = base + (int)__builtin_ctzll(f | (1ull << GW));    // first row of maximum magnitude (dgetf2)
                    p = p > 63 ? lane : p;                                     // (a group without blocks: harmless values)
                    if (lane == p) pos = k;
                    const int src = p * 4;
                    const double ukk = lu_bpermute(a[k], src);
                    const double inv = lu_rcp(ukk);
                    if (lane == p) myinv = inv;
                    const bool below = pos < 0;
                    const double l = below ? a[k] * inv : 0.0;
                    if (below) a[k] = l;
                    lu_for<k + 1, NP>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const double ukj = lu_bpermute(a[j], src);
                        a[j] = __builtin_fma(-l, ukj, a[j]);
                    });
                    if (solve) {
                        const double yk = lu_bpermute(bb, src);
                        bb = __builtin_fma(-l, yk, bb);
                    }
                }
            });
        } else if (solve) {
            lu_for<0, NP>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nsp) {
                    const double yk = lu_bpermute(bb, ((lane & ~(GW - 1)) + k) * 4);
                    if (i > k) bb = __builtin_fma(-a[k], yk, bb);
                }
            });
        }
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (!pre && lu != nullptr && act) {
            double* Ls = lu + s * ne;
            lu_for<0, NP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < nsp) Ls[pos + (long)nsp * j] = a[j];
            });
            if (perm != nullptr) perm[s * nsp + pos] = i;
        }
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (solve) {
            lu_for<0, NP>([&](auto kr) {
                constexpr int k = NP - 1 - decltype(kr)::value;
                if (k < nsp) {
                    int p = lu_group_first<GW>(pos == k, lane);
                    p = p > 63 ? lane : p;
                    if (pre) myinv = lu_rcp(a[k]);
                    const double xk = lu_bpermute(lu_div(bb, a[k], myinv), p * 4);
                    if (pos == k) bb = xk;
                    else if (pos < k && pos >= 0) bb = __builtin_fma(-a[k], xk, bb);
                }
            });
            if (act) x[pos * Y.v_si + s * Y.v_ss] = bb;
        }
    }
}

template <int NP, int GW>
inline void lu_launch16(int nsp, long n, const double* A, LuLay Y, double gamma, double* lu, int* perm, const double* b, double* x,
                        int mode, int cus, hipStream_t st)
{
    constexpr int PER_WG = 4 * (64 / GW);   // four wavefronts of four (two) blocks per workgroup
    long blocks = ((n + 15) / 16 + 7) / 8 * 8 * (16 / PER_WG);
    const long cap = (long)cus * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_lu16<NP, GW>), dim3((unsigned)blocks), dim3(256), 0, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
}

template <int NP>
inline void lu_launch_np(int nsp, long n, const double* A, LuLay Y, double gamma, double* lu, int* perm, const double* b, double* x,
                         int mode, int cus, hipStream_t st)
{
    long blocks = ((n + 15) / 16 + 7) / 8 * 32;   // four blocks per workgroup, runs of 16 per XCD (k_lu)
    const long cap = (long)cus * 8;         // grid-stride beyond a few workgroups per CU (a multiple of 32)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_lu<NP>, dim3((unsigned)blocks), dim3(256), 0, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
}

inline int lu_launch(int nsp, long n, const double* A, LuLay Y, double gamma, double* lu, int* perm, const double* b, double* x,
                     int mode, int cus, hipStream_t st)
{
    if (nsp < 1 || nsp > LU_MAX_LDS) return -1;
#ifndef PJ_LU4
#define PJ_LU4 1            // 65 .. 128 rows: the four-wavefront register-resident kernel (0: k_lu_lds for everything above 64)
#endif
    if (PJ_LU4 && nsp > 64 && nsp <= 128 && !(mode & LU_PREFACTORED)) {
        const long slots = (n + 127) / 128 * 128;
        const long blocks = slots < (long)cus * 8 ? slots : (long)cus * 8;
#ifdef PJ_LU4_SPLIT
        if (nsp <= 80) lu4_go_80((unsigned)blocks, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
        else if (nsp <= 96) lu4_go_96((unsigned)blocks, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
        else if (nsp <= 112) lu4_go_112((unsigned)blocks, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
        else lu4_go_128((unsigned)blocks, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
#else
        auto go = [&](auto ncc) {
            hipLaunchKernelGGL((k_lu4<decltype(ncc)::value>), dim3((unsigned)blocks), dim3(256), 0, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
        };
        if (nsp <= 80) go(std::integral_constant<int, 80>{});
        else if (nsp <= 96) go(std::integral_constant<int, 96>{});
        else if (nsp <= 112) go(std::integral_constant<int, 112>{});
        else go(std::integral_constant<int, 128>{});
#endif
        return 0;
    }
    if (nsp > 64) {
        const int ld = nsp | 1;
        const size_t lds = sizeof(double) * ((size_t)ld * nsp + nsp + 4) + sizeof(int) * (4 + (size_t)nsp);
        // (per device: a function attribute belongs to the device that was current when it was set)
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -2;
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            if (hipFuncSetAttribute((const void*)k_lu_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        const long slots = (n + 127) / 128 * 128;
        long blocks = slots < (long)cus * 4 ? slots : (long)cus * 4;
        hipLaunchKernelGGL(k_lu_lds, dim3((unsigned)blocks), dim3(256), lds, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
        return 0;
    }
#ifdef PJ_LU_ONLY_NP    // (microbenchmark builds: one instantiation)
    lu_launch_np<PJ_LU_ONLY_NP>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st);
    return 0;
#endif
    // (the 53-species GRI-Mech-3.0 family, the headline mechanism: its own instantiation instead of three padding
    // columns in every elimination step -- 7 % of the column work)
    if (nsp == 53 || nsp == 54) { lu_launch_np<54>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); return 0; }
    switch ((nsp + 7) / 8) {
    case 1: lu_launch16<8, 16>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    case 2: lu_launch16<16, 16>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    case 3: lu_launch16<24, 32>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    case 4: lu_launch16<32, 32>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    case 5: lu_launch_np<40>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    case 6: lu_launch_np<48>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    case 7: lu_launch_np<56>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    default: lu_launch_np<64>(nsp, n, A, Y, gamma, lu, perm, b, x, mode, cus, st); break;
    }
    return 0;
}
#endif  // !PJ_LU4_ONLY

}  // namespace pj
#endif  // !PJ_LU_DECL_ONLY
