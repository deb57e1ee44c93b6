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

enum { LU_FACTOR = 1, LU_SOLVE = 2, LU_PREFACTORED = 4 };
// where the input blocks and the vectors live: entry (r, c) of block s at a[(r + NSP c) * a_si + s * a_ss], entry i of
// vector s at v[i * v_si + s * v_ss] -- pyJac's per-state layout (a_si = 1, a_ss = NSP^2; v_si = 1, v_ss = NSP) or the
// state-fastest batch layout the row kernels write (a_si = n, a_ss = 1; v_si = n, v_ss = 1).  Factors are always
// written / read per state (P A = L U blocks are consumed by these kernels only).
struct LuLay { long a_si, a_ss, v_si, v_ss; };

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
constexpr int LU_MAX_LDS = 140;        // (140 | 1) * 140 + 3 * 140 doubles = 159 KB of the 160 KB

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

// ---- blocks of 65 .. 128 rows: FOUR wavefronts per block, the matrix in registers ------------------------------------
// k_lu_lds keeps the block in LDS and pays ~4 barriers and an LDS round trip per entry and column: 168 ms per 2e5
// 111 x 111 blocks, 0.03 of the HBM roofline, 27x the time of the Jacobians it consumes.  Here the
// workgroup's four wavefronts are (row half rh) x (column parity ch): lane l of wavefront (rh, ch) holds row
// rh * NC + l, and of it the columns 2 c + ch, c < NC -- NC doubles per lane (112 registers for 111 x 111), compile-time
// indices, columns dealt cyclically so that all four wavefronts have work until the last step.  Elimination step
// k = 2 c + h (TWO barriers):
//   1. the two wavefronts that hold column k (ch == h) reduce their open rows' |a_k| to a candidate each (unsigned keys,
//      DPP maxima, ties to the lowest row, as k_lu) and leave value + lane in LDS;                          -- barrier --
//   2. every wavefront picks the winner (the larger magnitude, the lower row half on a tie: dgetf2's choice unless an
//      earlier step displaced a tied row, as k_lu); the winner row's lane in either column half parks its remaining
//      entries in LDS (one lane, 16-byte writes), the column owners scale their column (multipliers: kept as L, and
//      left in LDS for the other column half);                                                              -- barrier --
//   3. a[c'] -= l * u[c'] for the columns right of k: the pivot row arrives as uniform 16-byte LDS reads.
// Rows are never exchanged (implicit pivoting: a row remembers its position), forward substitution rides along, the
// back substitution runs on an LDS copy of the right-hand side (two barriers per column as well).  Same (lu, perm)
// convention as k_lu / k_lu_lds; stored factors are solved by k_lu_lds (LU_PREFACTORED).
template <int NC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_lu4(const int nsp, const long n, const double* A, const LuLay Y, const double gamma, double* lu, int* __restrict__ perm,
      const double* __restrict__ b, double* __restrict__ x, const int mode)
{
    static_assert(NC % 2 == 0 && NC <= 64, "k_lu4: up to 128 rows, an even number of columns per wavefront");
    __shared__ __attribute__((aligned(16))) double prow[2][NC];     // the pivot row, per column parity
    __shared__ double lcol[128];                                     // the multipliers of the step, per row
    __shared__ double candv[2];
    __shared__ int candl[2];
    __shared__ double ys[2 * NC + 2];                                // right-hand side by position (back substitution)
    __shared__ double yk_s;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane0 = tid & 63;
    const int rh = __builtin_amdgcn_readfirstlane(wave >> 1), ch = __builtin_amdgcn_readfirstlane(wave & 1);
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
        const int row = rh * NC + lane;
        const bool act = lane < NC && row < nsp;
        const int row_c = act ? row : nsp - 1;
        double a[NC];
        {
            const double* As = A + s * Y.a_ss;
            // all loads first (clamped addresses instead of predicates), then identity padding / I - gamma A branch-free
            lu_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int g = 2 * c + ch;
                a[c] = As[(row_c + (long)nsp * (g < nsp ? g : nsp - 1)) * Y.a_si];
            });
            const bool newton = gamma != 0.0;
            const double sc = newton ? -gamma : 1.0, sh = newton ? 1.0 : 0.0;
            lu_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int g = 2 * c + ch;
                const double id = (g == row && lane < NC) ? 1.0 : 0.0;
                a[c] = (act && g < nsp) ? __builtin_fma(sc, a[c], sh * id) : id;
            });
        }
        int pos = -1;
        double bb = 0.0;
        if (solve && ch == 0) bb = act ? b[row * Y.v_si + s * Y.v_ss] : 0.0;
        unsigned long long openmask = 0;
        {
            const int nrow = nsp - rh * NC;                  // open rows of this row half: lanes 0 .. nrow - 1
            openmask = nrow >= 64 ? ~0ull : nrow > 0 ? (1ull << (nrow < NC ? nrow : NC)) - 1ull : 0ull;
        }
        lu_for<0, NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            lu_for<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value, k = 2 * c + h;
                if (k < nsp) {                                   // (uniform)
                    // ---- 1. candidates of the two wavefronts that hold column k
                    if (ch == h) {
                        const bool open = __builtin_amdgcn_inverse_ballot_w64(openmask);
                        const unsigned long long ub = (unsigned long long)__double_as_longlong(a[c]);
                        const bool num = a[c] == a[c];
                        const unsigned key = (open & num) ? (unsigned)(ub >> 32) & 0x7fffffffu : 0u;   // a NaN never beats a number
                        unsigned kmax = key;
                        kmax = lu_dpp_umax<0x111, 0xf>(kmax);
                        kmax = lu_dpp_umax<0x112, 0xf>(kmax);
                        kmax = lu_dpp_umax<0x114, 0xf>(kmax);
                        kmax = lu_dpp_umax<0x118, 0xf>(kmax);
                        kmax = lu_dpp_umax<0x142, 0xa>(kmax);
                        kmax = lu_dpp_umax<0x143, 0xc>(kmax);
                        const unsigned mx = (unsigned)__builtin_amdgcn_readlane((int)kmax, 63);
                        unsigned long long hit = __builtin_amdgcn_ballot_w64(key == mx) & openmask;
                        if (__builtin_popcountll(hit) > 1) {        // (uniform, rare): the lower 32 bits of the tying rows
                            const bool in = __builtin_amdgcn_inverse_ballot_w64(hit);
                            const unsigned k2 = (in && num) ? (unsigned)ub : 0u;
                            unsigned m2 = k2;
                            m2 = lu_dpp_umax<0x111, 0xf>(m2);
                            m2 = lu_dpp_umax<0x112, 0xf>(m2);
                            m2 = lu_dpp_umax<0x114, 0xf>(m2);
                            m2 = lu_dpp_umax<0x118, 0xf>(m2);
                            m2 = lu_dpp_umax<0x142, 0xa>(m2);
                            m2 = lu_dpp_umax<0x143, 0xc>(m2);
                            const unsigned mx2 = (unsigned)__builtin_amdgcn_readlane((int)m2, 63);
                            hit = __builtin_amdgcn_ballot_w64(in && k2 == mx2);
                        }
                        const int lp = hit ? __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(hit)) : -1;
                        const double val = lu_readlane(a[c], lp < 0 ? 0 : lp);
                        if (lane == 0) { candv[rh] = val; candl[rh] = lp; }
                    }
                    __syncthreads();
                    // ---- 2. the winner; its row to LDS, the multipliers of column k
                    const double v0 = candv[0], v1 = candv[1];
                    const int l0 = candl[0], l1 = candl[1];
                    // magnitudes as integers (a NaN counts as 0, as in the keys); the lower row half wins a tie
                    const unsigned long long m0 = (v0 == v0) ? ((unsigned long long)__double_as_longlong(v0) & 0x7fffffffffffffffull) : 0ull;
                    const unsigned long long m1 = (v1 == v1) ? ((unsigned long long)__double_as_longlong(v1) & 0x7fffffffffffffffull) : 0ull;
                    const bool second = l0 < 0 || (l1 >= 0 && m1 > m0);
                    const int rhp = __builtin_amdgcn_readfirstlane(second ? 1 : 0);
                    const int lp = __builtin_amdgcn_readfirstlane(second ? l1 : l0);
                    const double ukk = second ? v1 : v0;
                    const double inv = lu_rcp(ukk);
                    const bool mine = rh == rhp;                 // (uniform)
                    if (mine) {
                        const bool me = __builtin_amdgcn_inverse_ballot_w64(1ull << (lp & 63));
                        pos = me ? k : pos;
                        openmask &= ~(1ull << (lp & 63));
                        if (me) {
                            // the pivot row's entries right of column k (this wavefront's share), 16 bytes at a time
                            constexpr int ST = c + 1, EV = ST + (ST & 1);            // first pair-aligned column
                            if (ch > h) prow[ch][c] = a[c];                          // (column 2 c + 1 at step 2 c)
                            if constexpr ((ST & 1) != 0 && ST < NC) prow[ch][ST] = a[ST];
                            lu_for<0, (NC - EV) / 2>([&](auto qc) {
                                constexpr int j = EV + 2 * decltype(qc)::value;
                                double2 v;
                                v.x = a[j];
                                v.y = a[j + 1];
                                *(double2*)&prow[ch][j] = v;
                            });
                            if (solve && ch == 0) yk_s = bb;
                        }
                    }
                    const bool below = __builtin_amdgcn_inverse_ballot_w64(openmask);   // rows not chosen yet
                    double l = 0.0;
                    if (ch == h) {
                        l = below ? a[c] * inv : 0.0;
                        if (below) a[c] = l;
                        lcol[rh * 64 + lane] = l;
                    }
                    __syncthreads();
                    // ---- 3. update of the columns right of k
                    if (ch != h) l = lcol[rh * 64 + lane];
                    if (ch > h) a[c] = __builtin_fma(-l, prow[ch][c], a[c]);
                    {
                        constexpr int ST = c + 1, EV = ST + (ST & 1);
                        if constexpr ((ST & 1) != 0 && ST < NC) a[ST] = __builtin_fma(-l, prow[ch][ST], a[ST]);
                        lu_for<0, (NC - EV) / 2>([&](auto qc) {
                            constexpr int j = EV + 2 * decltype(qc)::value;
                            const double2 u = *(const double2*)&prow[ch][j];
                            a[j] = __builtin_fma(-l, u.x, a[j]);
                            a[j + 1] = __builtin_fma(-l, u.y, a[j + 1]);
                        });
                    }
                    if (solve && ch == 0) bb = __builtin_fma(-l, yk_s, bb);      // forward substitution rides along
                }
            });
        });
        asm volatile("" : "+s"(nsp), "+v"(lane));
        if (lu != nullptr && act) {
            double* Ls = lu + s * ne;
            lu_for<0, NC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                const int g = 2 * c + ch;
                if (g < nsp) Ls[pos + (long)nsp * g] = a[c];
            });
            if (perm != nullptr && ch == 0) perm[s * nsp + pos] = row;
        }
        if (solve) {
            // U x = y, last column first, on an LDS copy of y by position: the lane at position k divides, the column's
            // owners (parity ch == k & 1) update the positions above
            if (ch == 0 && act) ys[pos] = bb;
            __syncthreads();
            lu_for<0, NC>([&](auto cr) {
                constexpr int c = NC - 1 - decltype(cr)::value;
                lu_for<0, 2>([&](auto hr) {
                    constexpr int h = 1 - decltype(hr)::value, k = 2 * c + h;
                    if (k < nsp) {
                        if (ch == h && pos == k) ys[k] = lu_div(ys[k], a[c], lu_rcp(a[c]));
                        __syncthreads();
                        if (ch == h && pos >= 0 && pos < k) ys[pos] = __builtin_fma(-a[c], ys[k], ys[pos]);
                        __syncthreads();
                    }
                });
            });
            if (ch == 0 && act) x[pos * Y.v_si + s * Y.v_ss] = ys[pos];
        }
        __syncthreads();
    }
}

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
        auto go = [&](auto ncc) {
            hipLaunchKernelGGL((k_lu4<decltype(ncc)::value>), dim3((unsigned)blocks), dim3(256), 0, st, nsp, n, A, Y, gamma, lu, perm, b, x, mode);
        };
        if (nsp <= 80) go(std::integral_constant<int, 40>{});
        else if (nsp <= 96) go(std::integral_constant<int, 48>{});
        else if (nsp <= 112) go(std::integral_constant<int, 56>{});
        else go(std::integral_constant<int, 64>{});
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

}  // namespace pj
