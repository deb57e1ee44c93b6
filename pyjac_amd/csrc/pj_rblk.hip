// pj_rblk.hip -- state-per-lane Jacobian kernels for MEDIUM / LARGE mechanisms, second generation:
// row blocks that REBUILD the reaction rates they need instead of reading them back.
//
// Round 1's family (pj_rows.hip, retired) evaluated every reaction once, handed c*k_f (and more) to row-block
// kernels through an HBM scratch array and read it back ~3.6 times: 2.2x the algorithmic bytes, a
// load queue that sits behind the Jacobian stores of the previous block (vmcnt is one in-order
// counter on gfx9), and a compute-bound rate kernel that cannot share a SIMD with the row kernels.
// Measured (profiles/r02_*): the row kernels wait 55 % and issue 24 % of their wave-cycles, the
// instruction cache hits 99.5 % -- arithmetic is the resource that is left over.  So here:
//
//   k_rblk<B0,B1>  row blocks [B0,B1) of the Jacobian, one thermochemical state per lane,
//                  concentrations in LDS (one column per lane).  A visit of reaction i rebuilds
//                  k_f = exp(ln A + b ln T - Ta/T), K_c (pre-summed NASA polynomials + one exp), the
//                  third-body concentration, theta_i = dq_i/dT and the cheap concentration products
//                  from T, p and the LDS columns, and accumulates omega_k, P_k, Q_k, sum nu theta
//                  (the d/dT column, finished here too) and the structurally non-zero S_kj of the
//                  block's rows in registers with compile-time indices.  No loads in the steady
//                  state: the Jacobian stores of block b drain while block b+1 is being computed.
//   k_pre          the few reactions whose rate factor is expensive (falloff: Lindemann / Troe / SRI,
//                  PLOG, Chebyshev) are evaluated once per state and handed over: theta, c*k_f, rp, b_M, b_col --
//                  4-5 doubles for ~10 % of the reactions.  Their loads for block b+1 are issued
//                  BEFORE the stores of block b.
//   energy row     column j is finished by the block that holds ROW j (E_j = sum_i Hr_i G_ij over the reactions it
//                  visits anyway; see BCOL below), long-lived sums only for what that block cannot see; scalar sums
//                  and those few columns travel between the kernels of one library through hand-over slots
//                  that the next kernel loads with its state; the last kernel finishes the row and jac[0].
//   lane groups    PJQ_HALVES = 2 / 4: the wavefronts of a workgroup work on the SAME states and split the row blocks;
//   PJQ_KCF        up to 53 species: equilibrium constants as products of per-species factors kept in LDS columns,
//                  64 states per workgroup, four lane groups, ONE row kernel (see k_rblk).
//
//   k_jvd          w = J v per state -- the consumer of pyJac's sparse_multiplier (create_jacobian.py:3301-3404) -- as a
//                  directional derivative: every reaction visited ONCE, its derivative row times the vector scattered to
//                  its net species like its rate (PJQ_PART == 5 below); NSP instead of NSP^2 doubles written per state.
//   PJQ_JV         (rounds 2 - 4; built on request, PJ_RBLK_ROW_JV) the same product from the row kernels with their
//                  Jacobian stores replaced by w_k += J(k, c) v_c: 3.6 visits per reaction, 3x the time of k_jvd.
//
// The mechanism is injected as constexpr tables (pj::emit_spec_header + pj::emit_rows_tables ->
// PJS_HEADER); every loop is a compile-time loop.  One translation unit per kernel (PJQ_PART).
//
// Same formulation as pj_lane.hip / pj_kernel.h; reference emitters:
// pyjac/core/rate_subs.py:254-2335, pyjac/core/create_jacobian.py:2189-3298.
//
// PJQ_PART = 0: host entry points;  1: k_pre;  2: k_rblk, row blocks [PJQ_B0, PJQ_B1) (PJQ_FIRST / PJQ_LAST: first /
// last row kernel of the library);  3: k_rate, reactions [PJQ_R0, PJQ_R1);  4: k_fin (option PJQ_FIN, off);  5: k_jvd,
// reactions [PJQ_R0, PJQ_R1).  PJQ_ID is the launch-order index of a kernel; the ranges come from the kernel plan in the
// header (pj::emit_rows_tables: KER_B, KER_BM, RATE_R) unless given explicitly (tests; k_jvd without LDS copies of the K_c
// rows: the whole mechanism).
#ifdef PJR_HOST_EMU
#include "hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <type_traits>
#include <utility>

#include "pj_tables.h"
#include PJS_HEADER

using namespace pj;

#ifndef PJQ_BLOCK
#define PJQ_BLOCK 256
#endif
#ifndef PJQ_JV
#define PJQ_JV 0            // 1: the Jacobian is consumed in registers, w = J v per state (no Jacobian stores)
#endif
#ifndef PJQ_PAIR
#define PJQ_PAIR 0          // 1: two Jacobian columns per 16-byte store (SoA output, whole workgroups)
#endif
#ifndef PJQ_SB_EVERY
#define PJQ_SB_EVERY 0      // scheduling barrier after every n-th visit (0: none)
#endif
#ifndef PJQ_DEPTH
#define PJQ_DEPTH 4         // falloff / PLOG visits whose hand-over values are in flight
#endif
#ifndef PJQ_CONC_OPAQUE
#define PJQ_CONC_OPAQUE 0
#endif
#ifndef PJQ_KC_AHEAD
#define PJQ_KC_AHEAD 1      // K_c rows of the next visit are read while the current one is computed
#endif
#ifndef PJQ_SPLIT
#define PJQ_SPLIT 0         // scheduling barrier between the two phases of a visit
#endif
#ifndef PJQ_STREAMS
#define PJQ_STREAMS 1       // internal streams the chunks of a batch are dealt to
#endif
#ifndef PJQ_HALVES
#define PJQ_HALVES 1        // 2 / 4: a workgroup is that many groups of PJQ_BLOCK lanes that share the PJQ_BLOCK states'
                            // concentration columns and run different row blocks (k_rblk)
#endif
#ifndef PJQ_KCF
#define PJQ_KCF 0           // 1: equilibrium constants as products of per-species factors X_k that the prologue puts
                            // into LDS columns next to the concentrations (one exp per species and state) instead of a
                            // 7-term polynomial and an exp per reversible visit (k_rblk; pyjac_amd/kcfactors.py)
#endif
#ifndef PJQ_SINGLE
#define PJQ_SINGLE 0        // 1: the library has ONE row kernel (nothing is handed from row kernel to row kernel)
#endif
#ifndef PJQ_SUMSETS
#define PJQ_SUMSETS (PJQ_SINGLE ? 0 : 2 * PJQ_HALVES)   // slot sets of the energy-row sums between row kernels: a pair
                            // per lane group of the ROW kernels (every translation unit of a library gets the same value)
#endif
#ifndef PJQ_ECL
#define PJQ_ECL 0           // 1: what the block of row j cannot see of column j of the energy row (enhanced colliders, a falloff
                            // collider, a species on both sides) is summed ONCE per state by k_pre and handed over: the row
                            // kernels carry no long-lived energy-row sums at all (every translation unit gets the same value)
#endif
#ifndef PJQ_COOP
#define PJQ_COOP (PJQ_KCF && PJQ_HALVES > 1)   // k_rblk's prologue: lane group g loads the mass fractions of ITS species, the
                            // groups exchange partial sums and each writes its species' concentration columns (the
                            // factor-column builds always; the polynomial K_c builds on request: 4 KB of LDS per 64 states)
#endif
#ifndef PJQ_FIN
#define PJQ_FIN 0           // 1: the energy row is finished by a kernel of its own (k_fin) behind the row kernels instead of in the
                            // last row kernel's epilogue, whose loads sit behind that kernel's last Jacobian stores (needs the
                            // column sums in the hand-over array: PJQ_ECOLS, and PJQ_ECL; every translation unit gets the same value)
#endif
#ifndef PJQ_DEFER
#define PJQ_DEFER 0         // 1: the Jacobian rows of block b are stored DURING the visits of block b + 1, a slice behind every
                            // visit, instead of in one burst behind the block's own visits (k_rblk; not the w = J v builds)
#endif
#ifndef PJQ_CONC_AHEAD
#define PJQ_CONC_AHEAD 1    // 1: a visit's concentration reads are issued during the previous visit (-1 %)
#endif
#ifndef PJQ_SPLIT_TAIL
#define PJQ_SPLIT_TAIL 1     // batches with a partially filled last round: two unequal parts on two streams
#endif
#ifndef PJQ_CHUNK
#define PJQ_CHUNK (1L << 20) // states per chunk
#endif
#ifndef PJQ_LAUNDER_EVERY
#define PJQ_LAUNDER_EVERY 1  // blocks between opaque copies of T's functions (0: never)
#endif
#ifndef PJQ_C_LDS
#define PJQ_C_LDS 0         // k_pre: concentrations in LDS (set for large mechanisms)
#endif
#ifndef PJQ_XCD
#define PJQ_XCD 0           // k_rblk: XCD-aware workgroup -> states mapping (a contiguous eighth of the batch per XCD)
#endif
#ifndef PJQ_STAGGER0
#define PJQ_STAGGER0 0      // default spread of the first round of k_rblk's workgroups (PjqArgs::stagger)
#endif
#define PJQ_TILE 256        // states per scratch tile
#if defined(PJR_HOST_EMU)
#define PJQ_STORE(ptr, val) (*(ptr) = (val))
#define PJQ_LOAD_NT(ptr) (*(ptr))
#define PJQ_SCHED_BARRIER()
#else
// Jacobian entries are written once and never read back by these kernels
#ifndef PJQ_NT_STORE
#define PJQ_NT_STORE 1
#endif
#if defined(PJQ_NO_STORE) && PJQ_PART == 2
// experiment: the arithmetic without the Jacobian stores (values folded into one sink per lane)
#define PJQ_STORE(ptr, val) (pjq_sink += (val))
#define PJQ_STORE2(ptr, val) (pjq_sink += (val).x + (val).y)
#elif PJQ_NT_STORE
#define PJQ_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define PJQ_STORE2(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define PJQ_STORE(ptr, val) (*(ptr) = (val))
#define PJQ_STORE2(ptr, val) (*(ptr) = (val))
#endif
#define PJQ_LOAD_NT(ptr) __builtin_nontemporal_load(ptr)
#define PJQ_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// debug builds (-DPJQ_TIMING): shader cycles per phase and wavefront, summed over a kernel
// (0 prologue, 1 Arrhenius visits, 2 hand-over visits, 3 output phase, 4 energy-row epilogue: its last part -- 5: c_p sums,
// 6: fence + barriers + exchange of the scalar sums, 7: loads of the column sums)
#ifdef PJQ_TIMING
#define PJQ_TICK(ph) { const long long tn_ = clock64(); tacc[ph] += tn_ - tprev; tprev = tn_; }
#else
#define PJQ_TICK(ph) {}
#endif

struct PjqArgs {
    long n;                            // states of this chunk
    const double* pres;                // chunk base
    const double* y; long y_si, y_ss;  // chunk base
    double* jac; long j_si, j_ss;      // chunk base
    double* scr;                       // hand-over array [tile][NSLOTS][PJQ_TILE]
    int sum_last;
    // PJQ_JV kernels (w = J v per state): chunk bases, element (i, s) at base[i*si + s*ss]
    const double* v; long v_si, v_ss;
    double* w; long w_si, w_ss;
    // k_rate (rate outputs of pyJac's k_dydt pass): SoA arrays with leading dimension o_ld, any may be null;
    // sr: omega_k between the rate kernels of a library that has several (leading dimension sr_ld)
    // (k_rate with per-reaction outputs stores unconditionally: an array the caller does not want is pointed at
    // a dummy row with row stride 0 -- fwd_ld / rev_ld / pm_ld)
    double *conc, *fwd, *rev, *pres_mod, *spec_rates, *dy; long o_ld;
    long fwd_ld, rev_ld, pm_ld;
    double* sr; long sr_ld;
    // PJQ_TILE again, as a run-time value: slot offsets of the hand-over array formed with it are scalar arithmetic the
    // optimiser cannot fold into the per-lane address (k_rblk: ScrRef)
    long tile_rt;
    // k_rblk: spread of the first round of workgroups in units of 64 x 128 cycles (0: all start together)
    int stagger;
};
typedef void (*pjq_launch_fn)(const PjqArgs&, void* stream);
extern "C" void pjq_register(int id, int kind, pjq_launch_fn fn);

namespace {

constexpr double RU_ = 8314.4621;
constexpr double INV_LN10 = 0.434294481903251828;
constexpr int NSP = pjs::NSP, NRXN = pjs::NRXN, LAST = pjs::NSP - 1, ONE = pjs::NSP;
constexpr int S_TH = 0, S_KF = 1, S_KR = 2, S_RP = 3, S_BM = 4, S_BC = 5;
// Energy-row sums (H, SCP, SJT and LAST partial sums E_j) travel from row kernel to row kernel through
// two sets of slots: kernel q reads set q % 2 and writes set (q + 1) % 2.  Never in place: a state that
// two lanes evaluate (the shifted last workgroup of the pair-store kernels, the lanes past the end of
// the general ones) is then read identically and written identically by both, whatever the order the
// two workgroups run in (in place, the later one could pick up the earlier one's write-back and add this
// kernel's share twice: seen once two parts of a batch ran on two streams).
constexpr int NSUM = 5 + (pjs::NSP - 1);   // H, SCP, SJT, HP, HQN, E_j
#ifdef PJQ_ID
constexpr int SUM_IN = pjs::NSCQ + (PJQ_ID % 2) * NSUM, SUM_OUT = pjs::NSCQ + ((PJQ_ID + 1) % 2) * NSUM;
#endif
// (Tried and removed: the LAST partial sums E_j as no-return fp64 atomic adds into a per-state array instead of
// registers -- 8 % fewer instructions, half the AGPRs, and 7.0 -> 11.1 ms (GRI-shaped), 6.5 -> 8.8 ms (USC-shaped):
// 1.8 k / 5.7 k atomic adds per state are more than the L2 takes next to the Jacobian stores;
// profiles/r03_rblk_energy_row_atomics.txt.)
// Energy row by columns (round 4).  E_j = sum_k h_kW_k S_kj with S_kj = sum_i nu_ki G_ij (G_ij: what reaction i's
// molecule slots, efficiencies and collider put into column j), i.e. E_j = sum_i Hr_i G_ij with the reaction enthalpy
// Hr_i = sum_k nu_ki h_kW_k = R T (T dlnK_c/dT + sum nu) -- which a visit has anyway.  The block that owns ROW j visits
// every reaction that has j as a net reactant / product, so it can finish COLUMN j of the energy row itself: one
// running sum per row of the block instead of LAST sums carried through the whole kernel by every lane group (104 /
// 220 registers -- what made the kernels spill, and a spill reload sits behind every Jacobian store).  What a block
// cannot see -- slots of species that are not net species of the reaction: enhanced third-body colliders, a falloff
// collider, a species on both sides -- goes into long-lived sums as before, but only the few columns that have such
// contributions exist (constant zero otherwise), added at ONE visit of the reaction (its first block).
constexpr bool in_net(int i, int sp)
{
    for (int q = 0; q < pjs::RI[i][RI_NET_CNT]; ++q)
        if (pjs::NET_SP[pjs::RI[i][RI_NET_PTR] + q][0] == sp) return true;
    return false;
}
constexpr double net_sum(int i)
{
    double s = 0.0;
    for (int q = 0; q < pjs::RI[i][RI_NET_CNT]; ++q) s += pjs::NET_NU[pjs::RI[i][RI_NET_PTR] + q][0];
    return s;
}
struct BCols { bool b[pjs::NSP + 1]; int n; };
constexpr BCols make_bcols()
{
    BCols m{};
    auto mark = [&](int i, int sp) { if (sp >= 0 && sp < pjs::NSP - 1 && !in_net(i, sp)) m.b[sp] = true; };
    for (int i = 0; i < pjs::NRXN; ++i) {
        const int fl = pjs::RI[i][RI_FLAGS];
        for (int c = RI_R0; c <= RI_R2; ++c) mark(i, pjs::RI[i][c]);
        if (fl & F_REV) for (int c = RI_P0; c <= RI_P2; ++c) mark(i, pjs::RI[i][c]);
        if (fl & F_GEN) {
            const int nf = pjs::RI[i][RI_GEN_NR] + ((fl & F_REV) ? pjs::RI[i][RI_GEN_NP] : 0);
            for (int f = 0; f < nf; ++f) mark(i, pjs::GEN_SP[pjs::RI[i][RI_GEN_PTR] + f][0]);
        }
        if (fl & F_EFFTYPE) for (int e = 0; e < pjs::RI[i][RI_EFF_CNT]; ++e) mark(i, pjs::EFF_SP[pjs::RI[i][RI_EFF_PTR] + e][0]);
        if ((fl & F_COLLIDER) && pjs::RI[i][RI_COLLIDER] >= 0) mark(i, pjs::RI[i][RI_COLLIDER]);
    }
    for (int j = 0; j < pjs::NSP - 1; ++j) m.n += m.b[j] ? 1 : 0;
    return m;
}
constexpr BCols BCOL = make_bcols();
// With several lane groups AND several row kernels a finished column sum travels to the last kernel through one
// slot per column behind the slot sets (written once, by the kernel that holds the row)
#ifndef PJQ_ECOLS
#define PJQ_ECOLS (PJQ_SUMSETS > 2)      // (also set for ONE row kernel whose LDS has no room for the column sums: EJ below)
#endif
constexpr int E_COL0 = pjs::NSCQ + PJQ_SUMSETS * NSUM;
// PJQ_ECL: one slot per marked column (BCOL) behind them, written by k_pre, read by the last row kernel's epilogue
constexpr int ECL0 = E_COL0 + (PJQ_ECOLS ? pjs::NSP - 1 : 0);
constexpr int ecl_index(int j)
{
    int c = 0;
    for (int q = 0; q < j; ++q) c += BCOL.b[q] ? 1 : 0;
    return c;
}
constexpr int NSLOTS = ECL0 + (PJQ_ECL ? BCOL.n : 0);
// does reaction i put anything into a column whose species is not one of its net species?
constexpr bool has_ecl(int i)
{
    bool any = false;
    auto mark = [&](int sp) { if (sp >= 0 && sp < pjs::NSP - 1 && !in_net(i, sp)) any = true; };
    const int fl = pjs::RI[i][RI_FLAGS];
    for (int c = RI_R0; c <= RI_R2; ++c) mark(pjs::RI[i][c]);
    if (fl & F_REV) for (int c = RI_P0; c <= RI_P2; ++c) mark(pjs::RI[i][c]);
    if (fl & F_GEN) {
        const int nf = pjs::RI[i][RI_GEN_NR] + ((fl & F_REV) ? pjs::RI[i][RI_GEN_NP] : 0);
        for (int f = 0; f < nf; ++f) mark(pjs::GEN_SP[pjs::RI[i][RI_GEN_PTR] + f][0]);
    }
    if (fl & F_EFFTYPE) for (int e = 0; e < pjs::RI[i][RI_EFF_CNT]; ++e) mark(pjs::EFF_SP[pjs::RI[i][RI_EFF_PTR] + e][0]);
    if ((fl & F_COLLIDER) && pjs::RI[i][RI_COLLIDER] >= 0) mark(pjs::RI[i][RI_COLLIDER]);
    return any;
}

// Columns whose species weighs (nearly) what the last species weighs.  Entry (k, j) is (W_k / W_j)(P_k - w_j Q_k + S_kj) with
// w_j = W_j / W_N and Q_k = P_k + QN_k: for w_j = 1 -- an isomer of the last species; pyJac leaves whatever species the file
// lists last in that place when it has no N2 / AR / HE (create_jacobian.py:3521-3542) -- the dense parts cancel EXACTLY,
// P_k - Q_k = -QN_k, and the reference, which forms a_i (1 - W_j / W_N) per reaction (create_jacobian.py:341-489), gets the
// small remainder right, while (1 / W_j) W_k P_k - (W_k / W_N) Q_k carries the rounding error of P_k: entries 1e-16 of their
// row scale off by percents (found by the random-mechanism sweep of round 6: sweep_r2, HCNO next to HOCN).  So the blocks
// accumulate QN_k = sum nu gN instead of Q_k (one addition per visit less, and nothing at all for the reactions that do not
// see the last species), a row's W_k Q_k / W_N is formed once from P_k + QN_k, and the columns with w_j = 1 take
// (1 / W_j - 1 / W_N) W_k P_k - W_k QN_k / W_N [+ (1 / W_j) W_k S_kj]: no cancellation between sums.  The energy row
// likewise carries HQN = sum hW_k QN_k and forms (1 - w_j) HP - w_j HQN.
constexpr bool near_last(int j)
{
    // (only w_j = 1 -- up to the last bits of two differently ordered element sums -- is special: for w_j = 1 - 1e-4, CO next to
    // N2, the coefficient 1 - w_j itself is known to eps / (1 - w_j) only, in the reference's arithmetic as in this one, and both
    // forms lose the same digits)
    const double d = 1.0 - pjs::SP[j][3];
    return d < 1.0 / (1 << 30) && d > -1.0 / (1 << 30);
}
constexpr bool any_near_last()
{
    for (int j = 0; j < pjs::NSP - 1; ++j) if (near_last(j)) return true;
    return false;
}
constexpr bool ANY_NEAR = any_near_last();
// does a visit of reaction i put anything into QN (a molecule slot, general factor or collider that IS the last species, or
// an enhanced efficiency of the last species)?
constexpr bool has_gn(int i)
{
    const int fl = pjs::RI[i][RI_FLAGS];
    if (pjs::RD[i][RD_ANM1] != 0.0) return true;
    for (int c = RI_R0; c <= RI_R2; ++c) if (pjs::RI[i][c] == pjs::NSP - 1) return true;
    if (fl & F_REV) for (int c = RI_P0; c <= RI_P2; ++c) if (pjs::RI[i][c] == pjs::NSP - 1) return true;
    if (fl & F_GEN) {
        const int nf = pjs::RI[i][RI_GEN_NR] + ((fl & F_REV) ? pjs::RI[i][RI_GEN_NP] : 0);
        for (int f = 0; f < nf; ++f) if (pjs::GEN_SP[pjs::RI[i][RI_GEN_PTR] + f][0] == pjs::NSP - 1) return true;
    }
    if ((fl & F_COLLIDER) && pjs::RI[i][RI_COLLIDER] == pjs::NSP - 1) return true;
    return false;
}

// reactions evaluated once per state by k_pre and handed over
constexpr bool is_pre(int i) { return (pjs::RI[i][RI_FLAGS] & (F_PDEP | F_PLOG | F_CHEB)) != 0; }

#define PJR_INL __attribute__((always_inline))
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
// f(integral_constant<g>) for the one g == grp of [G0, G1): an if / else-if chain on a wavefront-uniform value
template <int G0, int G1, class F>
__device__ __forceinline__ void group_dispatch(const int grp, F&& f)
{
    if constexpr (G0 + 1 >= G1) {
        f(std::integral_constant<int, G0>{});
    } else {
        if (grp == G0) f(std::integral_constant<int, G0>{});
        else group_dispatch<G0 + 1, G1>(grp, f);
    }
}

#include "pj_math.h"

// exponentials and logarithms of the falloff / PLOG / Chebyshev body (pj_rate_pre.inc: k_pre, k_rate, k_jvd): the lean forms of
// pj_math.h (18 / 40 instructions) instead of the device library's (36 / 92) -- a Troe reaction calls them seven times / twice,
// which is where the 427 instructions per hand-over reaction of round 5's k_pre went
#ifndef PJR_LEAN_MATH
#define PJR_LEAN_MATH 1
#endif
#if PJR_LEAN_MATH && !defined(PJR_HOST_EMU)
#define PJR_EXP(x_) exp_one(x_)
#define PJR_LOG(x_) log_lean(x_)
#else
#define PJR_EXP(x_) exp(x_)
#define PJR_LOG(x_) log(x_)
#endif

// General stoichiometry (F_GEN: a fractional coefficient or more than three molecules on a side; pj_tables.h):
// C^nu of factor F of the header's GEN_SP / GEN_NU lists -- whole-number coefficients by repeated multiplication,
// as the reference emits them, fractional ones through pow() (rate_subs.py:634-658) -- and nu C^(nu-1), where the
// power of C is only there "if (nu - 1) > 0" (create_jacobian.py:417-427: reference quirk kept for parity)
template <int F>
__device__ __forceinline__ double gen_pow(const double C)
{
    constexpr double nu = pjs::GEN_NU[F][0];
    if constexpr (nu == (double)(int)nu) {
        double r = 1.0;
        static_for<(int)nu>([&](auto) PJR_INL { r *= C; });
        return r;
    } else {
        return pow(C, nu);
    }
}
template <int F>
__device__ __forceinline__ double gen_dpow(const double C)
{
    constexpr double nu = pjs::GEN_NU[F][0];
    double r = nu;
    if constexpr (nu - 1.0 > 0.0) {
        if constexpr (nu == (double)(int)nu) static_for<(int)nu - 1>([&](auto) PJR_INL { r *= C; });
        else r *= pow(C, nu - 1.0);
    }
    return r;
}

// Real-valued reaction constants in the visit bodies: as 64-bit literals (two s_mov_b32 each, an issue slot
// apiece at one wavefront per SIMD) or, PJQ_RD_CONST = 1, read through the scalar cache from the __constant__
// tables of the header (RDT, EFFT: s_load_dwordx2..x16 at immediate offsets from one opaque base; the
// scheduler batches neighbouring fields).  Constants that encode for free (0, +-0.5, +-1, +-2, +-4, or 32 zero low
// bits: a single literal dword) stay literals either way.
#ifndef PJQ_RD_CONST
#define PJQ_RD_CONST 0
#endif
constexpr bool cheap_literal(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    return (u & 0xffffffffull) == 0;
}
#if PJQ_RD_CONST && !defined(PJR_HOST_EMU)
typedef const __attribute__((address_space(4))) double* pjq_cptr;
#define PJQ_CONST_BASES() \
    pjq_cptr rdt_ = (pjq_cptr)&pjs::RDT[0][0], efft_ = (pjq_cptr)&pjs::EFFT[0][0], invw_ = (pjq_cptr)&pjs::INVWT[0][0]; \
    asm volatile("" : "+s"(rdt_), "+s"(efft_), "+s"(invw_));
#define RDC(i_, f_) (cheap_literal(pjs::RD[i_][f_]) ? pjs::RD[i_][f_] : rdt_[(i_) * RDW + (f_)])
#define EFC(e_) (cheap_literal(pjs::EFF_AM1[e_][0]) ? pjs::EFF_AM1[e_][0] : efft_[e_])
#define INVW(j_) invw_[j_]
#else
#define PJQ_CONST_BASES()
#define RDC(i_, f_) pjs::RD[i_][f_]
#define EFC(e_) pjs::EFF_AM1[e_][0]
#define INVW(j_) pjs::SP[j_][0]
#endif

#if PJQ_PAIR
typedef double d2s __attribute__((ext_vector_type(2)));
// Pair stores: a lane writes 16 bytes, two neighbouring states of one Jacobian entry.  The lanes of a
// wavefront are mapped to states so that one v_permlane32_swap per 32-bit half does the exchange: lane
// l < 32 holds state 2 l, lane l + 32 holds state 2 l + 1.  Swapping the upper half of x (entry c) with
// the lower half of y (entry c + 1) leaves the lower lanes with entry c of states (2 l, 2 l + 1) and the
// upper lanes with entry c + 1 of the same two states: (x, y) is the 16-byte piece, no selects.
// (An (even, odd) lane pairing needs 2 DPP moves and 6 selects for the same exchange.)
__device__ __forceinline__ void swap_halves(double& x, double& y)
{
    const unsigned long long ux = __builtin_bit_cast(unsigned long long, x), uy = __builtin_bit_cast(unsigned long long, y);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ux, (unsigned)uy, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ux >> 32), (unsigned)(uy >> 32), false, false);
    x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
    y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
}
#endif

// hand-over layout [state tile][slot][PJQ_TILE]: what one workgroup reads and writes is one
// contiguous NSLOTS * 2 KB region, each wave access is 512 B
__device__ __forceinline__ double* scr_of(const PjqArgs& A, long s)
{
    return A.scr + (s / PJQ_TILE) * ((long)NSLOTS * PJQ_TILE) + (s % PJQ_TILE);
}

struct State {
    double T, p, Wbar, rho, invrho, mconc;
    double C[NSP + 1];
};
__device__ __forceinline__ void load_state(const PjqArgs& A, long s, State& L)
{
    const double* y = A.y + s * A.y_ss;
    L.T = y[0];
    L.p = A.pres[s];
    // all loads first: left to itself the scheduler keeps two of them in flight and pays the memory
    // latency NSP / 2 times per kernel
    static_for<LAST>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        L.C[k] = y[(k + 1) * A.y_si];
    });
    PJQ_SCHED_BARRIER();
    double sumY = 0.0, sumYW = 0.0;
    static_for<LAST>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
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
__device__ __forceinline__ void to_conc(State& L)
{
    static_for<NSP>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        L.C[k] = L.rho * L.C[k] * pjs::SP[k][0];
    });
    L.C[ONE] = 1.0;
}

// K_c groups a kernel needs, in order of first use: only their NASA row pairs are staged in LDS
// (the whole table of a 111-species mechanism is 100 KB)
constexpr int NKC_ALL = pjs::LT_SP / 16;
struct KcMap { int loc[NKC_ALL > 0 ? NKC_ALL : 1]; int list[NKC_ALL > 0 ? NKC_ALL : 1]; int n; };
constexpr void kcmap_add(KcMap& m, int i)
{
    if (!(pjs::RI[i][RI_FLAGS] & F_REV)) return;
    for (int c = 0; c < pjs::RI[i][RI_KC_CNT]; ++c) {
        const int g = pjs::RI[i][RI_KC_PTR] + c;
        if (m.loc[g] < 0) { m.loc[g] = m.n; m.list[m.n++] = g; }
    }
}
// NQ 16-byte pieces per thread: rows of the listed groups -> registers (issue), registers -> LDS (land)
template <int NQ, int NT = PJQ_BLOCK, class M, class D2>
__device__ __forceinline__ void kc_issue(const M& list, int nrows, D2* lt)
{
    static_for<NQ>([&](auto qc) PJR_INL {
        constexpr int q = decltype(qc)::value;
        const int x = (int)threadIdx.x + q * NT;
        const int row = x < nrows * 8 ? x >> 3 : 0;
        lt[q] = ((const D2*)(pjs::LTAB + pjs::LT_KC + (long)list[row] * 16))[x & 7];
    });
}
template <int NQ, int NT = PJQ_BLOCK, class D2>
__device__ __forceinline__ void kc_land(double* table, int nrows, const D2* lt)
{
    static_for<NQ>([&](auto qc) PJR_INL {
        constexpr int q = decltype(qc)::value;
        const int x = (int)threadIdx.x + q * NT;
        if (x < nrows * 8) ((D2*)table)[x] = lt[q];
    });
}
struct __attribute__((aligned(16))) d2 { double x, y; };

#if PJQ_PART == 1
// ------------------------------------------------------------------------------------------
// k_pre: falloff / PLOG reactions once per state -> hand-over array
// ------------------------------------------------------------------------------------------
#define PJR_RECOMPUTE_KF 0
#define PJR_RECOMPUTE_KR 1          // c*k_r is rebuilt by the row kernels
#define PJR_SLOT(i_, c_) pjs::SCQ[i_][c_]
constexpr bool kf_plain(int) { return false; }
constexpr int NEFF = (int)(sizeof(pjs::EFF_AM1) / sizeof(pjs::EFF_AM1[0]));
// k_pre's reactions: the hand-over reactions and -- PJQ_ECL -- every reaction that puts something into a column of the
// energy row whose species is not one of its net species (enhanced colliders, a falloff collider, a species on both
// sides: the block that finishes the column never sees it).  Those are summed here, once per state:
// E^B_j = sum_i Hr_i G_ij over exactly the (i, j) that make_bcols marks.
constexpr bool in_pre(int i) { return is_pre(i) || (PJQ_ECL && has_ecl(i)); }
constexpr int NECL = PJQ_ECL ? BCOL.n : 0;
constexpr KcMap make_kcmap()
{
    KcMap m{};
    for (int g = 0; g < NKC_ALL; ++g) m.loc[g] = -1;
    for (int i = 0; i < NRXN; ++i) if (in_pre(i)) kcmap_add(m, i);
    return m;
}
constexpr KcMap KCM = make_kcmap();
constexpr int NKC = KCM.n;
struct KcList { int v[NKC > 0 ? NKC : 1]; };
constexpr KcList make_list() { KcList l{}; for (int q = 0; q < NKC; ++q) l.v[q] = KCM.list[q]; return l; }
__device__ const KcList KCL = make_list();

// ordinal of reaction i among k_pre's reactions (PJQ_HALVES == 2: even ones to half 0, odd ones to half 1)
constexpr int pre_ordinal(int i)
{
    int c = 0;
    for (int q = 0; q < i; ++q) c += in_pre(q) ? 1 : 0;
    return c;
}
constexpr int NTHR = PJQ_BLOCK * PJQ_HALVES;
__global__ void __launch_bounds__(NTHR) k_pre(PjqArgs A)
{
    // NASA row pairs of the K_c groups: the range select is per lane, so the rows are read from LDS
    __shared__ __attribute__((aligned(16))) double LT[(NKC > 0 ? NKC : 1) * 16];
#if PJQ_C_LDS
    __shared__ double CLr[NSP][PJQ_BLOCK];
#endif
    {
        constexpr int NQ = (NKC * 8 + NTHR - 1) / NTHR;
        d2 lt[NQ > 0 ? NQ : 1];
        kc_issue<NQ, NTHR>(KCL.v, NKC, lt);
        kc_land<NQ, NTHR>(LT, NKC, lt);
    }
    // two halves (see k_rblk): both on the same PJQ_BLOCK states, each with every other hand-over reaction
    const int half = PJQ_HALVES == 2 ? (int)(threadIdx.x >= PJQ_BLOCK) : 0;
    const int tid = (int)threadIdx.x - half * PJQ_BLOCK;
    const long s_nom = (long)blockIdx.x * PJQ_BLOCK + tid;
    const long s = s_nom < A.n ? s_nom : A.n - 1;
#if PJQ_C_LDS
    double T, p, invrho, Wbar, mconc;
    {
        State L;
        load_state(A, s, L);
        to_conc(L);
        T = L.T; p = L.p; invrho = L.invrho; Wbar = L.Wbar; mconc = L.mconc;
        if (PJQ_HALVES == 1 || half == 0)
            static_for<NSP>([&](auto kc) PJR_INL { CLr[decltype(kc)::value][tid] = L.C[decltype(kc)::value]; });
    }
    __syncthreads();
    // (lanes past the end repeat the last state: same values to the same addresses)
#define CC(idx) ((idx) == ONE ? 1.0 : CLr[(idx) == ONE ? 0 : (idx)][tid])
#else
    __syncthreads();
    State L;
    load_state(A, s, L);
    to_conc(L);
    const double T = L.T, p = L.p;
    const double invrho = L.invrho, Wbar = L.Wbar, mconc = L.mconc;
#define CC(idx) L.C[idx]
#endif
    const double logT = log(T), invT = 1.0 / T, logp = log(p);
    double* const scr = scr_of(A, s);
    constexpr bool RATES_OUT = false;
    auto rate_out = [](auto, double, double, double) {};
#ifdef PJR_HOST_EMU
#define SCR_ST(slot, val) (scr[(long)(slot) * PJQ_TILE] = (val))
#else
#define SCR_ST(slot, val) __builtin_nontemporal_store((val), &scr[(long)(slot) * PJQ_TILE])
#endif
    double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
    double jt[NSP], jtq = 0.0;          // d/dT sums are taken by the row kernels: dead here
    double ecl[NECL > 0 ? NECL : 1];
    static_for<NECL>([&](auto cc) PJR_INL { ecl[decltype(cc)::value] = 0.0; });
    auto run_pre = [&](auto hc) PJR_INL {
    constexpr int HALF_ = decltype(hc)::value;
    static_for<NRXN>([&](auto ic) PJR_INL {
        constexpr int i = decltype(ic)::value;
        if constexpr (in_pre(i) && (PJQ_HALVES == 1 || pre_ordinal(i) % 2 == HALF_)) {
#define PJR_RD(i_) pjs::RD[i_]
#define PJR_KCROW(g_) (LT + KCM.loc[g_] * 16)
#define PJR_EFL(e_) pjs::EFF_AM1[e_][0]
#define PJR_KC_FIRST(i_) true
#define PJR_SCHED_BARRIER() PJQ_SCHED_BARRIER()
#include "pj_rate_pre.inc"
#undef PJR_RD
#undef PJR_KCROW
#undef PJR_KC_FIRST
#if PJQ_ECL
            if constexpr (has_ecl(i)) {
                // the slots of k_rblk's visit (create_jacobian.py:341-489, 2850-2938) whose species is not a net species
                // of the reaction, times the reaction enthalpy Hr_i = sum_k nu_ki h_kW_k = R T (T dlnK_c/dT + sum nu)
                constexpr int np0 = pjs::RI[i][RI_NET_PTR], ncnt = pjs::RI[i][RI_NET_CNT];
                double Hr;
                if constexpr ((fl & F_REV) != 0) {
                    constexpr double nsum = net_sum(i);
                    Hr = (RU_ * T) * (TdlnKc + nsum);
                } else {
                    Hr = 0.0;
                    static_for<ncnt>([&](auto qc) PJR_INL {
                        constexpr int k = pjs::NET_SP[np0 + decltype(qc)::value][0];
                        const bool lo = T <= pjs::SP[k][2];
                        double a[6];
                        static_for<6>([&](auto cc) PJR_INL {
                            constexpr int c_ = decltype(cc)::value;
                            a[c_] = lo ? pjs::SP[k][4 + c_] : pjs::SP[k][11 + c_];
                        });
                        Hr += pjs::NET_NU[np0 + decltype(qc)::value][0] *
                              (RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                      T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T))))));
                    });
                }
                double gkf = c * kf, gkr = c * kr;
                if constexpr (pjs::SCQ[i][S_KR] >= 0) {       // Chebyshev: eval_jacob's own k_f (pj_rate_pre.inc)
                    gkf = kfj_;
                    if constexpr ((fl & F_REV) != 0) gkr = kfj_ * ekc[pjs::KC_CLASS[i][0]];
                }
                auto eslot = [&](auto spc, const double gv) PJR_INL {
                    constexpr int sp = decltype(spc)::value;
                    if constexpr (sp < LAST) {
                        if constexpr (!in_net(i, sp)) {
                            static_assert(BCOL.b[sp], "energy row: column not marked");
                            // (a constexpr variable: called in place, ecl_index() is a RUN-TIME loop over BCOL, the index
                            // a variable, and the whole array lives in scratch memory)
                            constexpr int ci = ecl_index(sp);
                            ecl[ci] += Hr * gv;
                        }
                    }
                };
                eslot(std::integral_constant<int, pjs::RI[i][RI_R0]>{}, gkf * (cr1 * cr2));
                eslot(std::integral_constant<int, pjs::RI[i][RI_R1]>{}, gkf * (cr0 * cr2));
                eslot(std::integral_constant<int, pjs::RI[i][RI_R2]>{}, gkf * (cr0 * cr1));
                if constexpr ((fl & F_REV) != 0) {
                    eslot(std::integral_constant<int, pjs::RI[i][RI_P0]>{}, -gkr * (cp1 * cp2));
                    eslot(std::integral_constant<int, pjs::RI[i][RI_P1]>{}, -gkr * (cp0 * cp2));
                    eslot(std::integral_constant<int, pjs::RI[i][RI_P2]>{}, -gkr * (cp0 * cp1));
                }
                if constexpr ((fl & F_GEN) != 0) {
                    constexpr int GP = pjs::RI[i][RI_GEN_PTR], GNR = pjs::RI[i][RI_GEN_NR];
                    constexpr int GNP = (fl & F_REV) ? pjs::RI[i][RI_GEN_NP] : 0;
                    double gcf[GNR + GNP > 0 ? GNR + GNP : 1], gpw[GNR + GNP > 0 ? GNR + GNP : 1];
                    static_for<GNR + GNP>([&](auto fc) PJR_INL {
                        constexpr int f = decltype(fc)::value;
                        gcf[f] = CC(pjs::GEN_SP[GP + f][0]);
                        gpw[f] = gen_pow<GP + f>(gcf[f]);
                    });
                    static_for<GNR + GNP>([&](auto fc) PJR_INL {
                        constexpr int f = decltype(fc)::value;
                        constexpr int f0 = f < GNR ? 0 : GNR, f1 = f < GNR ? GNR : GNR + GNP;
                        double gv = (f < GNR ? gkf : -gkr) * gen_dpow<GP + f>(gcf[f]);
                        static_range<f0, f1>([&](auto hc2) PJR_INL { if constexpr (decltype(hc2)::value != f) gv *= gpw[decltype(hc2)::value]; });
                        eslot(std::integral_constant<int, pjs::GEN_SP[GP + f][0]>{}, gv);
                    });
                }
                if constexpr ((fl & F_COLLIDER) != 0)
                    eslot(std::integral_constant<int, (pjs::RI[i][RI_COLLIDER] >= 0 ? pjs::RI[i][RI_COLLIDER] : ONE)>{}, bcol);
                if constexpr ((fl & F_EFFTYPE) != 0) {
                    static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJR_INL {
                        constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                        eslot(std::integral_constant<int, pjs::EFF_SP[e][0]>{}, PJR_EFL(e) * bM);
                    });
                }
            }
#endif
#undef PJR_EFL
            PJQ_SCHED_BARRIER();
        }
    });
    };
    if constexpr (PJQ_HALVES == 2) {
        if (half == 0) run_pre(std::integral_constant<int, 0>{});
        else run_pre(std::integral_constant<int, 1>{});
    } else {
        run_pre(std::integral_constant<int, 0>{});
    }
    (void)jt; (void)jtq;
    if constexpr (NECL > 0) {
        if constexpr (PJQ_HALVES == 2) {
#if PJQ_C_LDS
            // the halves' shares meet in the concentration columns (nobody reads them any more)
            __syncthreads();
            if (half == 1) static_for<NECL>([&](auto cc) PJR_INL { CLr[decltype(cc)::value][tid] = ecl[decltype(cc)::value]; });
            __syncthreads();
            if (half == 0) static_for<NECL>([&](auto cc) PJR_INL {
                SCR_ST(ECL0 + decltype(cc)::value, ecl[decltype(cc)::value] + CLr[decltype(cc)::value][tid]);
            });
#else
            static_assert(PJQ_HALVES != 2, "k_pre: two lane groups need the concentration columns (PJQ_C_LDS)");
#endif
        } else {
            static_for<NECL>([&](auto cc) PJR_INL { SCR_ST(ECL0 + decltype(cc)::value, ecl[decltype(cc)::value]); });
        }
    }
#undef SCR_ST
#undef CC
}

void launch_pre(const PjqArgs& A, void* stream)
{
    const long blocks = (A.n + PJQ_BLOCK - 1) / PJQ_BLOCK;
    hipLaunchKernelGGL(k_pre, dim3((unsigned)blocks), dim3(NTHR), 0, (hipStream_t)stream, A);
}
struct Reg { Reg() { pjq_register(0, 1, launch_pre); } } reg_;
#endif  // PJQ_PART == 1

#if PJQ_PART == 2
// ------------------------------------------------------------------------------------------
// k_rblk<B0,B1>
// ------------------------------------------------------------------------------------------
#ifndef PJQ_B0      // from the kernel plan
#define PJQ_B0 pjs::KER_B[PJQ_ID][0]
#define PJQ_B1 pjs::KER_B[PJQ_ID + 1][0]
#define PJQ_PLAN 1
#define PJQ_FIRST (PJQ_ID == 0)
#define PJQ_LAST (PJQ_ID == pjs::NKER - 1)
#else
#define PJQ_PLAN 0
#endif
constexpr int B0_ = PJQ_B0, B1_ = PJQ_B1;
constexpr bool FIRST_ = PJQ_FIRST != 0, LASTK_ = PJQ_LAST != 0 && !PJQ_FIN;
static_assert(!PJQ_FIN || (PJQ_ECOLS && PJQ_ECL && PJQ_HALVES > 1), "PJQ_FIN: column sums and the pre-pass's sums in the hand-over array");
static_assert(!PJQ_KCF || pjs::KCF_OK, "PJQ_KCF needs the per-species factor rows (pj_mech_set_kc_factors)");
static_assert(!PJQ_SINGLE || (FIRST_ && (LASTK_ || PJQ_FIN)), "PJQ_SINGLE: one row kernel");
constexpr KcMap make_kcmap()
{
    KcMap m{};
    for (int g = 0; g < NKC_ALL; ++g) m.loc[g] = -1;
    if (PJQ_KCF) return m;          // no per-reaction polynomial rows at all
    for (int b = B0_; b < B1_; ++b)
        for (int v = pjs::BLK_RX_PTR[b][0]; v < pjs::BLK_RX_PTR[b + 1][0]; ++v) kcmap_add(m, pjs::BLK_RX[v][0]);
    return m;
}
constexpr KcMap KCM = make_kcmap();
constexpr int NKC = KCM.n;
struct KcList { int v[NKC > 0 ? NKC : 1]; };
constexpr KcList make_list() { KcList l{}; for (int q = 0; q < NKC; ++q) l.v[q] = KCM.list[q]; return l; }
__device__ const KcList KCL = make_list();

constexpr int max_kc_cnt()
{
    int m = 1;
    for (int i = 0; i < NRXN; ++i)
        if ((pjs::RI[i][RI_FLAGS] & F_REV) && pjs::RI[i][RI_KC_CNT] > m) m = pjs::RI[i][RI_KC_CNT];
    return m;
}
constexpr int MAXKC = max_kc_cnt();
constexpr int max_net_cnt()
{
    int m = 1;
    for (int i = 0; i < NRXN; ++i)
        if ((pjs::RI[i][RI_FLAGS] & F_REV) && pjs::RI[i][RI_NET_CNT] > m) m = pjs::RI[i][RI_NET_CNT];
    return m;
}
constexpr int MAXNET = max_net_cnt();

template <int i>
constexpr bool has_anm1() { return pjs::RD[i][RD_ANM1] != 0.0; }

// PJQ_VCT: the real-valued constants of a visit -- ln A, b, T_a, the K_c prefactor, the enhanced third-body efficiencies --
// through the SCALAR cache instead of the instruction stream.  A 64-bit literal is two s_mov_b32 (or two v_mov_b32 where an
// instruction would need a second constant: the constant bus of gfx9 carries one), each an issue slot of its own at one
// wavefront per SIMD: 11.3 k s_mov_b32 + 5.8 k v_mov_b32 of the 112 k issue slots per state of the 53-species kernel
// (round 6).  Here a reaction's constants are one 32-byte record {ln A, b, T_a, prefactor} of a table in constant memory
// (10 KB for 325 reactions: resident in the scalar cache from workgroup to workgroup), read by ONE s_load_dwordx8 -- a visit
// ahead, next to the LDS reads of the look-ahead, so that the single lgkmcnt(0) at the top of a visit covers both (scalar
// loads return out of order: any wait on them is a wait for everything, which is what killed round 3's PJQ_RD_CONST, whose
// loads sat next to their uses) -- and enter the arithmetic as SGPR operands.  Efficiencies: a second table, 64-byte
// records {alpha_N - 1, alpha_0 - 1 .. alpha_6 - 1} for the reactions that have any.
#ifndef PJQ_VCT
#define PJQ_VCT 0
#endif
#if PJQ_VCT && !defined(PJR_HOST_EMU)
#define PJQ_VCT_ON 1
static_assert(PJQ_KC_AHEAD, "PJQ_VCT: the constants travel with the look-ahead reads of the next visit (PJQ_KC_AHEAD)");
constexpr int VEW = 8;
constexpr int max_eff_cnt()
{
    int m = 0;
    for (int i = 0; i < NRXN; ++i) if ((pjs::RI[i][RI_FLAGS] & (F_THD | F_EFFTYPE)) && pjs::RI[i][RI_EFF_CNT] > m) m = pjs::RI[i][RI_EFF_CNT];
    return m;
}
static_assert(max_eff_cnt() <= VEW - 1, "PJQ_VCT: more enhanced colliders per reaction than an efficiency record holds");
constexpr bool vct_has_eff(int i) { return (pjs::RI[i][RI_FLAGS] & (F_THD | F_EFFTYPE)) && (pjs::RI[i][RI_EFF_CNT] > 0 || pjs::RD[i][RD_ANM1] != 0.0); }
constexpr int vct_eff_index(int i)
{
    int c = 0;
    for (int q = 0; q < i; ++q) c += vct_has_eff(q) ? 1 : 0;
    return c;
}
constexpr int NVE = vct_eff_index(NRXN);
struct __attribute__((aligned(64))) VctB { double v[NRXN > 0 ? NRXN : 1][4]; };
struct __attribute__((aligned(64))) VctE { double v[NVE > 0 ? NVE : 1][VEW]; };
constexpr VctB make_vctb()
{
    VctB t{};
    for (int i = 0; i < NRXN; ++i) {
        t.v[i][0] = pjs::RD[i][RD_LNA]; t.v[i][1] = pjs::RD[i][RD_B]; t.v[i][2] = pjs::RD[i][RD_TA];
        t.v[i][3] = PJQ_KCF ? pjs::KCF_PREFINV[i][0] : pjs::RD[i][RD_LNPREF];
    }
    return t;
}
constexpr VctE make_vcte()
{
    VctE t{};
    int r = 0;
    for (int i = 0; i < NRXN; ++i) {
        if (!vct_has_eff(i)) continue;
        t.v[r][0] = pjs::RD[i][RD_ANM1];
        for (int e = 0; e < pjs::RI[i][RI_EFF_CNT]; ++e) t.v[r][1 + e] = pjs::EFF_AM1[pjs::RI[i][RI_EFF_PTR] + e][0];
        ++r;
    }
    return t;
}
__device__ const VctB VCTB = make_vctb();
__device__ const VctE VCTE = make_vcte();
typedef const __attribute__((address_space(4))) double* pjq_vptr;
#else
#define PJQ_VCT_ON 0
#endif

// visits of block b that read hand-over values
template <int b>
constexpr int n_pre_visits()
{
    int c = 0;
    for (int v = pjs::BLK_RX_PTR[b][0]; v < pjs::BLK_RX_PTR[b + 1][0]; ++v) c += is_pre(pjs::BLK_RX[v][0]) ? 1 : 0;
    return c;
}
// the first row block that visits a reaction: where its contributions to the long-lived energy-row sums are added
struct OwnerMap { int b[NRXN > 0 ? NRXN : 1]; };
constexpr OwnerMap make_owner()
{
    OwnerMap m{};
    for (int i = 0; i < NRXN; ++i) m.b[i] = -1;
    for (int b = pjs::NBLK - 1; b >= 0; --b)
        for (int v = pjs::BLK_RX_PTR[b][0]; v < pjs::BLK_RX_PTR[b + 1][0]; ++v) m.b[pjs::BLK_RX[v][0]] = b;
    return m;
}
constexpr OwnerMap OWNER = make_owner();
#ifdef PJQ_TIMING
__device__ long long g_tim[8][1024][8];
#endif

// Lane groups (PJQ_HALVES = G: 1, 2 or 4).  The workgroup is G groups of PJQ_BLOCK lanes ON THE SAME PJQ_BLOCK
// STATES -- lane l of every group is state l -- that share one set of concentration columns (and factor columns /
// K_c rows) in LDS and run different row blocks: group g takes the blocks [GB(g), GB(g + 1)) of the kernel.  All four
// SIMDs of a CU work for the LDS of PJQ_BLOCK states: 128 states and two groups where 111 species leave room for
// no more; 64 states and four groups where the per-species factor columns of PJQ_KCF take the room (then ONE kernel
// covers every row block: the state is read once, nothing is handed from kernel to kernel).  Each group carries its
// own energy-row sums (own slot sets between kernels); in the last kernel the groups exchange them through the
// then free columns and share the columns of the energy row.
constexpr int G_ = PJQ_HALVES;
constexpr int NTHR = PJQ_BLOCK * G_;
static_assert(G_ == 1 || G_ == 2 || G_ == 4 || G_ == 8, "PJQ_HALVES: 1, 2, 4 or 8 lane groups (8: two wavefronts per SIMD, 256 registers each)");
// a column of the energy row has a long-lived sum only if something outside its own row's block contributes (BCOL) --
// or, with one lane group, always: the block's finished column sum is then kept there until the epilogue
// (PJQ_ECL: those contributions come from k_pre -- no long-lived sums with several lane groups)
constexpr bool e_live_col(int j) { return G_ == 1 || (BCOL.b[j] && !PJQ_ECL); }
constexpr int group_first_block(int g)
{
#if PJQ_PLAN
    if (pjs::KER_NG == G_) return pjs::KER_GB[PJQ_ID][g];
#endif
#ifdef PJQ_BM
    if (G_ == 2) return g == 0 ? B0_ : g == 1 ? (int)(PJQ_BM) : B1_;
#endif
    return B0_ + (int)((long)(B1_ - B0_) * g / G_);        // (tests: explicit block ranges, even split)
}
// species (prologue of the PJQ_KCF kernels) and energy-row columns (epilogue) of group g: contiguous, even
constexpr int group_first_species(int g) { return (int)((long)NSP * g / G_); }
constexpr int group_first_col(int g) { return (int)((long)LAST * g / G_); }
constexpr int col_owner(int j)
{
    for (int g = 0; g < G_; ++g) if (j < group_first_col(g + 1)) return g;
    return G_ - 1;
}
// is species k a net reactant / product of some reversible reaction (its factor column is read)?
constexpr bool kcf_used(int k)
{
    for (int i = 0; i < NRXN; ++i)
        if (pjs::RI[i][RI_FLAGS] & F_REV)
            for (int q = 0; q < pjs::RI[i][RI_NET_CNT]; ++q)
                if (pjs::NET_SP[pjs::RI[i][RI_NET_PTR] + q][0] == k) return true;
    return false;
}

// LDS of a workgroup, one raw array (doubles):
//   main phase   CL[NSP][BLOCK] concentrations | PJQ_KCF: XT[NSP][BLOCK] {X_k, t_k}, IXT[NSP][BLOCK] {1/X_k, t_k}
//                (t_k = h_k/RT - 1 rides along: one 16-byte read per net species and visit; tried: three 8-byte
//                columns X, 1/X, t -- the 27 KB of the second copy of t_k hold 14 more energy-row sums per lane
//                group, see below -- and the kernels spill MORE: 8.3 .. 11 ms against 5.9) | else LTK[NKC][16] the
//                kernel's K_c polynomial rows | PRED23[2][G][BLOCK] partial c_p sums of the cooperative prologue (read
//                by the epilogue) | EL[G][NEL][BLOCK] energy-row sums (below); the prologue's first partial sums
//                PRED01[2][G][BLOCK] use the same room before EL is cleared
//   epilogue     (last kernel, G > 1, after a barrier) EX[LAST - NEL][G-1][BLOCK] register-resident energy-row sums on
//                their way to the group that owns the column | RED[6][G][BLOCK] the scalar sums
// Long-lived energy-row sums in LDS (PJQ_NEL > 0; measured, no longer the default).  While every lane group carried
// LAST partial sums E_j through the kernel (104 registers of the 512 a lane has) the one-kernel builds spilled, and a
// spill reload sits behind every Jacobian store issued before it (vmcnt is in order): 9.2 ms per 1e6 GRI-shaped states
// against 5.7 ms for the same kernel without the sums.  Keeping the sums of the columns with the most structural
// non-zeros in LDS -- one slot per lane group, column and lane: a single writer each, so the order of the additions is
// fixed; an update is ONE ds_add_f64 instead of five instructions on an AGPR pair -- brought that to 5.9 ms; since the
// energy row is finished column by column (BCOL) far fewer long-lived sums exist and they fit the registers.
// PJQ_EXPT: the exponentials of k_rblk through the 64-entry table of pj_math.h (exp_tab), which sits in front of everything
// else so that its reads carry their offset in the instruction
#ifndef PJQ_EXPT
#define PJQ_EXPT 0
#endif
#if PJQ_EXPT && !defined(PJR_HOST_EMU)
#define PJQ_EXPT_ON 1
#define PJQ_EXP1(x_) exp_tab((x_), SM)
#define PJQ_EXP2(x0_, x1_, y0_, y1_) exp_tab_pair((x0_), (x1_), (y0_), (y1_), SM)
#else
#define PJQ_EXPT_ON 0
#define PJQ_EXP1(x_) exp_one(x_)
#define PJQ_EXP2(x0_, x1_, y0_, y1_) exp_pair((x0_), (x1_), (y0_), (y1_))
#endif
__device__ const double EXPT_G[64] = {PJM_EXPT[0], PJM_EXPT[1], PJM_EXPT[2], PJM_EXPT[3], PJM_EXPT[4], PJM_EXPT[5], PJM_EXPT[6], PJM_EXPT[7],
    PJM_EXPT[8], PJM_EXPT[9], PJM_EXPT[10], PJM_EXPT[11], PJM_EXPT[12], PJM_EXPT[13], PJM_EXPT[14], PJM_EXPT[15],
    PJM_EXPT[16], PJM_EXPT[17], PJM_EXPT[18], PJM_EXPT[19], PJM_EXPT[20], PJM_EXPT[21], PJM_EXPT[22], PJM_EXPT[23],
    PJM_EXPT[24], PJM_EXPT[25], PJM_EXPT[26], PJM_EXPT[27], PJM_EXPT[28], PJM_EXPT[29], PJM_EXPT[30], PJM_EXPT[31],
    PJM_EXPT[32], PJM_EXPT[33], PJM_EXPT[34], PJM_EXPT[35], PJM_EXPT[36], PJM_EXPT[37], PJM_EXPT[38], PJM_EXPT[39],
    PJM_EXPT[40], PJM_EXPT[41], PJM_EXPT[42], PJM_EXPT[43], PJM_EXPT[44], PJM_EXPT[45], PJM_EXPT[46], PJM_EXPT[47],
    PJM_EXPT[48], PJM_EXPT[49], PJM_EXPT[50], PJM_EXPT[51], PJM_EXPT[52], PJM_EXPT[53], PJM_EXPT[54], PJM_EXPT[55],
    PJM_EXPT[56], PJM_EXPT[57], PJM_EXPT[58], PJM_EXPT[59], PJM_EXPT[60], PJM_EXPT[61], PJM_EXPT[62], PJM_EXPT[63]};
constexpr int SM_CL = PJQ_EXPT_ON ? 64 : 0;
constexpr int SM_XT = SM_CL + NSP * PJQ_BLOCK;
constexpr int SM_IXT = SM_XT + (PJQ_KCF ? 2 * NSP * PJQ_BLOCK : 0);
constexpr int SM_LTK = SM_IXT + (PJQ_KCF ? 2 * NSP * PJQ_BLOCK : 0);
// EJ[LAST][BLOCK]: the finished column sums of the energy row on their way from the lane group that holds the row to
// the group that writes the column (one kernel, several groups); the prologue's partial sums PRED[4][G][BLOCK] use the
// same room before the first block ends
constexpr bool EJ_LDS = PJQ_SINGLE && G_ > 1 && !PJQ_ECOLS;
constexpr int SM_EJ = SM_LTK + (NKC > 0 ? NKC * 16 : 0);
// (w = J v builds: the room holds the vector v instead -- NSP columns; a finished column sum is folded into w_0 at once)
constexpr bool JV_LDS = PJQ_JV && EJ_LDS;
static_assert(!PJQ_KCF || G_ == 1 || PJQ_COOP, "the factor-column builds with several lane groups have a cooperative prologue");
constexpr int PRED_DOUBLES = (PJQ_COOP && G_ > 1) ? (PJQ_KCF ? 4 : 2) * G_ * PJQ_BLOCK : 0;
constexpr int SM_EJ_DOUBLES = (EJ_LDS ? (JV_LDS ? NSP : LAST) * PJQ_BLOCK : 0) > PRED_DOUBLES
                                  ? (EJ_LDS ? (JV_LDS ? NSP : LAST) * PJQ_BLOCK : 0) : PRED_DOUBLES;
constexpr int SM_EL = SM_EJ + SM_EJ_DOUBLES;
#ifndef PJQ_NEL
#define PJQ_NEL 0           // long-lived energy-row sums per lane group that live in LDS (ds_add_f64); -1: as many as fit
                            // (since the energy row is finished column by column only a few such sums exist: registers)
#endif
constexpr int nel_fit()
{
    const long room = (160L * 1024 / 8 - SM_EL) / ((long)G_ * PJQ_BLOCK);
    const long need01 = (PJQ_KCF && G_ > 1) ? 2 : 0;        // PRED01 borrows the room of two columns' worth per group
    long n = room < LAST ? room : LAST;
    if (n < need01 && room >= need01) n = n;                // (PRED01 only needs the ROOM, see SM_MAIN)
    return (int)(n < 0 ? 0 : n);
}
constexpr int NEL_ = PJQ_NEL >= 0 ? (PJQ_NEL < LAST ? PJQ_NEL : LAST) : nel_fit();
constexpr int SM_EL_DOUBLES = G_ * PJQ_BLOCK * NEL_;
constexpr int SM_MAIN = SM_EL + SM_EL_DOUBLES;
// (the epilogue's exchange area: over the then free columns if it fits below the sums that are still needed, else behind)
// (PJQ_ECL: no lane group carries long-lived sums, nothing is exchanged but the scalar sums)
constexpr int NEX = PJQ_ECL ? 0 : LAST - NEL_;
constexpr int SM_EPI_SIZE = (G_ > 1 && LASTK_) ? NEX * (G_ - 1) * PJQ_BLOCK + 6 * G_ * PJQ_BLOCK : 0;
constexpr int SM_EX = SM_EPI_SIZE <= SM_EJ ? 0 : SM_MAIN;
constexpr int SM_RED = SM_EX + NEX * (G_ - 1) * PJQ_BLOCK;
constexpr int SM_EPI = SM_EPI_SIZE ? SM_EX + SM_EPI_SIZE : 0;
constexpr int SM_DOUBLES = SM_MAIN > SM_EPI ? SM_MAIN : SM_EPI;
static_assert(SM_DOUBLES * 8 <= 160 * 1024, "LDS: columns of PJQ_BLOCK states do not fit");
// Per-state scalars that every visit of every block reads -- 1 / T, ln T, T, 1 / rho, Wbar / rho, p / (R T) -- are kernel-long
// values, and in a kernel that is short of registers exactly those are what the allocator parks in scratch memory
// (reloaded at every visit, each reload behind every Jacobian store issued before it).  Whatever LDS is left over holds a
// column of each (in that order, as many as fit): they are re-read at every row block through an opaque offset, so no block
// inherits them in registers.
#ifndef PJQ_PARK
#define PJQ_PARK 0           // (measured, round 5: no gain -- USC-shaped 4.99 -> 5.24 ms, GRI-shaped 5.93 -> 6.00 ms with it; kept as a build option)
#endif
constexpr int NPARK = (160 * 1024 / 8 - SM_DOUBLES) / PJQ_BLOCK < PJQ_PARK ? (160 * 1024 / 8 - SM_DOUBLES) / PJQ_BLOCK : PJQ_PARK;
constexpr int SM_PS = SM_DOUBLES;
constexpr int SM_TOTAL = SM_DOUBLES + NPARK * PJQ_BLOCK;
// which columns' sums live in LDS: those with the most structural non-zeros (the most updates)
constexpr int col_nnz(int j)
{
    int c = 0;
    for (int k = 0; k < NSP; ++k) c += pjs::SLOC[k][j] >= 0 ? 1 : 0;
    return c;
}
struct ElMap { int slot[LAST > 0 ? LAST : 1]; };
constexpr ElMap make_elmap()
{
    ElMap m{};
    int cnt[LAST > 0 ? LAST : 1] = {};
    for (int j = 0; j < LAST; ++j) cnt[j] = col_nnz(j);
    for (int j = 0; j < LAST; ++j) {
        int rank = 0;
        for (int q = 0; q < LAST; ++q) rank += (cnt[q] > cnt[j] || (cnt[q] == cnt[j] && q < j)) ? 1 : 0;
        m.slot[j] = rank < NEL_ ? rank : -1;
    }
    return m;
}
constexpr ElMap ELM = make_elmap();
// index of a register-resident column among the register-resident ones (the exchange area of the epilogue)
constexpr int ex_index(int j)
{
    int c = 0;
    for (int q = 0; q < j; ++q) c += ELM.slot[q] < 0 ? 1 : 0;
    return c;
}
#ifdef PJR_HOST_EMU
#define PJQ_LDS_ADD(ptr, val) (*(ptr) += (val))
#else
#define PJQ_LDS_ADD(ptr, val) ((void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)(ptr), (val)))
#endif

__global__ void __launch_bounds__(NTHR) k_rblk(PjqArgs A)
{
    __shared__ __attribute__((aligned(16))) double SM[SM_TOTAL];
    // Concentrations live in LDS, one column per lane (bank-conflict free)
    double (*const CL)[PJQ_BLOCK] = (double (*)[PJQ_BLOCK])(SM + SM_CL);
    // NASA row pairs of every K_c group (the low / high range select is per lane)
    double* const LTK = SM + SM_LTK;
#if PJQ_KCF
    d2 (*const XT)[PJQ_BLOCK] = (d2 (*)[PJQ_BLOCK])(SM + SM_XT);
    d2 (*const IXT)[PJQ_BLOCK] = (d2 (*)[PJQ_BLOCK])(SM + SM_IXT);
#endif
#ifdef PJQ_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
    // grp: wavefront-uniform -- and the compiler has to know (readfirstlane: a scalar), or the group branches below
    // become divergent regions that every wavefront walks through under an execution mask, with the registers of
    // all of them allocated together; tid: the lane's index within its group = its state's index in the workgroup
#ifdef PJR_HOST_EMU
    const int grp = G_ > 1 ? (int)threadIdx.x / PJQ_BLOCK : 0;
#else
    const int grp = G_ > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / PJQ_BLOCK) : 0;
#endif
    const int tid = (int)threadIdx.x - grp * PJQ_BLOCK;
    // lanes past the end repeat the last state (same values to the same addresses): no divergence
    // Workgroup -> states.  Workgroups are dealt to the eight XCDs round-robin (workgroup b runs on XCD b % 8), so with states
    // b * BLOCK .. neighbouring 512-byte pieces of every Jacobian entry are written by eight different L2s.  PJQ_XCD: XCD x
    // takes a CONTIGUOUS eighth of the batch instead -- the pieces that one L2 collects at a time are neighbours.
#if PJQ_XCD && !defined(PJR_HOST_EMU)
    long wgid;
    {
        const unsigned b = blockIdx.x, g = gridDim.x, q = g / 8u, r = g % 8u, x = b % 8u;
        wgid = (long)(x * q + (x < r ? x : r) + b / 8u);
    }
#else
    const long wgid = (long)blockIdx.x;
#endif
#if PJQ_PAIR
    // pair stores need whole lane pairs: the last workgroup is shifted back over states its neighbour
    // also evaluates (n >= PJQ_BLOCK, host-checked)
    long s0_wg = wgid * PJQ_BLOCK;
    if (s0_wg + PJQ_BLOCK > A.n) s0_wg = A.n - PJQ_BLOCK;
    // lanes 0..31 of a wavefront: its even states, lanes 32..63: the odd ones (swap_halves)
    const long s = s0_wg + (tid & ~63) + 2 * (tid & 31) + ((tid >> 5) & 1);
#else
    long s = wgid * PJQ_BLOCK + tid;
    if (s >= A.n) s = A.n - 1;
#endif
#ifdef PJQ_NO_STORE
    double pjq_sink = 0.0;
#endif
#if PJQ_EXPT_ON
    if (threadIdx.x < 64) SM[threadIdx.x] = EXPT_G[threadIdx.x];
    if constexpr (G_ == 1 || !PJQ_KCF) __syncthreads();      // (the factor-column prologue has a barrier in front of its exponentials)
#endif
#ifndef PJR_HOST_EMU
    // Every workgroup of a launch runs the same instruction stream on the same schedule, and the first round of workgroups
    // (one per CU) starts at the same moment: lane group g of ALL 256 CUs reaches the output phase of its block b together,
    // the chip asks for 256 x 27 KB at once, and between those bursts the memory system idles -- the later rounds inherit the
    // lockstep, a workgroup starts when one ends.  Phase map of the 53-species kernel (round 6): 231 k cycles per wavefront
    // with stores, 161 k without, while the stores alone need 130 k at the achievable 6.1 TB/s: neither overlapped (161 k)
    // nor serialised (291 k).  So the FIRST round is spread out: workgroup i < stagger_wgs sleeps (i * 37 mod 128) x 64 x
    // A.stagger cycles (at A.stagger = 1 up to 8 k cycles, about one row block's period; once per launch, 0.03 % of it).
    if (A.stagger > 0 && blockIdx.x < 256u) {
        const int ph = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 37u) & 127u)) * A.stagger;
        for (int q = 0; q < ph; ++q) __builtin_amdgcn_s_sleep(1);
    }
#endif
    double T, rho, invrho, Wbar, mconc;
    double cpa = 0.0, dcpa = 0.0;       // sum_k C_k cp_k / R and its d/dT: PJQ_KCF prologue, else the last kernel's epilogue
    PJQ_CONST_BASES()
    // PJQ_ECL: k_pre's sums of the marked columns this lane group owns.  One row kernel with several lane groups: requested
    // right behind the prologue (no store of this workgroup is in flight yet; requested next to the state they stay in
    // registers through the prologue's exponentials and the kernel keeps 128 bytes of scratch memory) and put where the
    // finished column sums go (EJ, or w_0's share of a w = J v build); otherwise the epilogue fetches them.
    constexpr bool ECL_PRO = PJQ_ECL && EJ_LDS;
    constexpr int NCOLG = (LAST + G_ - 1) / G_ + 1;
    double ECLV[PJQ_ECL ? NCOLG : 1];
    // (by POSITION p within the group's column range, the slot picked by scalar selects on grp: with a branch per lane
    // group around the loads the optimiser sinks the branches' common tail, the array index becomes a variable and ECLV
    // lives in scratch memory -- 128 bytes that cost the one-kernel builds a third of their speed)
    auto ecl_slot = [&](auto pc) PJR_INL {      // hand-over slot of this group's p-th column, -1: none / not marked
        constexpr int p = decltype(pc)::value;
        int sl = -1;
        static_for<G_>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value, j = group_first_col(g) + p;
            constexpr int v = (j < group_first_col(g + 1) && BCOL.b[j < LAST ? j : 0]) ? ECL0 + ecl_index(j < LAST ? j : 0) : -1;
            sl = grp == g ? v : sl;
        });
        return sl;
    };
    auto ecl_fetch = [&]() PJR_INL {
        const double* const scr_ = scr_of(A, s);
        static_for<(PJQ_ECL ? NCOLG : 0)>([&](auto pc) PJR_INL {
            const int sl = ecl_slot(pc);
            const double v = PJQ_LOAD_NT(&scr_[(long)(sl < 0 ? ECL0 : sl) * PJQ_TILE]);
            ECLV[decltype(pc)::value] = sl < 0 ? 0.0 : v;
        });
    };
#if PJQ_KCF
    {
        // Cooperative prologue: group g loads the mass fractions of ITS species, the groups exchange partial sums,
        // and each turns its species into concentration, factor and enthalpy columns for all of them:
        // X_k = exp(ln X_k), 1 / X_k and t_k = h_k/RT - 1 -- NSP / G exponential pairs per lane instead of one per
        // reversible visit.
        double (*const PRED)[G_ > 1 ? G_ : 1][PJQ_BLOCK] = (double (*)[G_ > 1 ? G_ : 1][PJQ_BLOCK])(SM + SM_EJ);
        const double* const y = A.y + s * A.y_ss;
        T = y[0];
        const double p = A.pres[s];
        constexpr int KMAXG = (NSP + G_ - 1) / G_ + 1;
        double Y[KMAXG];
        double sumY = 0.0, sumYW = 0.0;
        static_for<G_>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            constexpr int k0 = group_first_species(g), k1 = group_first_species(g + 1);
            if (G_ == 1 || grp == g) {
                static_range<k0, k1>([&](auto kc) PJR_INL {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < LAST) Y[k - k0] = y[(k + 1) * A.y_si];
                });
                PJQ_SCHED_BARRIER();
                static_range<k0, k1>([&](auto kc) PJR_INL {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < LAST) { sumY += Y[k - k0]; sumYW += Y[k - k0] * pjs::SP[k][0]; }
                });
            }
        });
        if constexpr (G_ > 1) {
            PRED[0][grp][tid] = sumY; PRED[1][grp][tid] = sumYW;
            __syncthreads();
            sumY = 0.0; sumYW = 0.0;
            static_for<G_>([&](auto gc) PJR_INL { sumY += PRED[0][decltype(gc)::value][tid]; sumYW += PRED[1][decltype(gc)::value][tid]; });
        }
        const double yN = 1.0 - sumY;
        sumYW += yN * pjs::SP[LAST][0];
        Wbar = 1.0 / sumYW;
        rho = p * Wbar / (RU_ * T);
        invrho = 1.0 / rho;
        mconc = p / (RU_ * T);
        const double logT_ = log(T), invT_ = 1.0 / T;
        static_for<G_>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            constexpr int k0 = group_first_species(g), k1 = group_first_species(g + 1);
            if (G_ == 1 || grp == g) {
                static_range<k0, k1>([&](auto kc) PJR_INL {
                    constexpr int k = decltype(kc)::value;
                    const double Ck = rho * (k < LAST ? Y[k < LAST ? k - k0 : 0] : yN) * pjs::SP[k][0];
                    CL[k][tid] = Ck;
                    const bool lo = T <= pjs::SP[k][2];
                    double a[6];
                    static_for<6>([&](auto cc) PJR_INL {
                        constexpr int c = decltype(cc)::value;
                        a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
                    });
                    cpa += Ck * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
                    dcpa += Ck * (a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T)));
                    // t_k = h_k/RT - 1 = T dlnX_k/dT (rate_subs.py:660-809: the d/dT of the K_c polynomial)
                    const double tq = (a[0] - 1.0) + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) + T * (a[3] * (1.0 / 4.0) +
                                      a[4] * (1.0 / 5.0) * T))) + a[5] * invT_;
                    double X = 1.0, IX = 1.0;
                    if constexpr (kcf_used(k)) {
                        const bool lx = T <= pjs::KCF_ROW[k][0];
                        double b[7];
                        static_for<7>([&](auto cc) PJR_INL {
                            constexpr int c = decltype(cc)::value;
                            b[c] = lx ? pjs::KCF_ROW[k][1 + c] : pjs::KCF_ROW[k][8 + c];
                        });
                        const double lnX = b[0] + b[1] * logT_ + T * (b[2] + T * (b[3] + T * (b[4] + b[5] * T))) - b[6] * invT_;
                        PJQ_EXP2(lnX, -lnX, X, IX);
                    }
                    d2 v; v.x = X; v.y = tq;
                    XT[k][tid] = v;
                    v.x = IX;
                    IXT[k][tid] = v;
                });
            }
        });
        if constexpr (G_ > 1) {
            PRED[2][grp][tid] = cpa; PRED[3][grp][tid] = dcpa;
            __syncthreads();
            cpa = 0.0; dcpa = 0.0;
            static_for<G_>([&](auto gc) PJR_INL { cpa += PRED[2][decltype(gc)::value][tid]; dcpa += PRED[3][decltype(gc)::value][tid]; });
            if constexpr (EJ_LDS) __syncthreads();      // (the room is EJ's from here on)
        } else {
            __syncthreads();
        }
    }
#else
    {
        // one round trip for everything the prologue reads: the state and this thread's share of the
        // K_c table are requested before anything waits (a load behind other workgroups' Jacobian
        // stores takes microseconds)
        constexpr int NQ = (NKC * 8 + NTHR - 1) / NTHR;     // 16-byte pieces per thread
        d2 lt[NQ > 0 ? NQ : 1];
        kc_issue<NQ, NTHR>(KCL.v, NKC, lt);
#if PJQ_COOP
        // cooperative (several lane groups): group g loads the mass fractions of ITS species only; the groups exchange the
        // partial sums of Y_N and W and each writes its species' concentration columns
        static_assert(G_ > 1, "PJQ_COOP: several lane groups");
        double (*const PRED)[G_][PJQ_BLOCK] = (double (*)[G_][PJQ_BLOCK])(SM + SM_EJ);
        const double* const y = A.y + s * A.y_ss;
        T = y[0];
        const double p = A.pres[s];
        constexpr int KMAXG = (NSP + G_ - 1) / G_ + 1;
        double Y[KMAXG];
        double sumY = 0.0, sumYW = 0.0;
        group_dispatch<0, G_>(grp, [&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            constexpr int k0 = group_first_species(g), k1 = group_first_species(g + 1);
            static_range<k0, k1>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < LAST) Y[k - k0] = y[(k + 1) * A.y_si];
            });
            PJQ_SCHED_BARRIER();
            static_range<k0, k1>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                if constexpr (k < LAST) { sumY += Y[k - k0]; sumYW += Y[k - k0] * pjs::SP[k][0]; }
            });
        });
        kc_land<NQ, NTHR>(LTK, NKC, lt);
        PRED[0][grp][tid] = sumY; PRED[1][grp][tid] = sumYW;
        __syncthreads();
        sumY = 0.0; sumYW = 0.0;
        static_for<G_>([&](auto gc) PJR_INL { sumY += PRED[0][decltype(gc)::value][tid]; sumYW += PRED[1][decltype(gc)::value][tid]; });
        const double yN = 1.0 - sumY;
        sumYW += yN * pjs::SP[LAST][0];
        Wbar = 1.0 / sumYW;
        rho = p * Wbar / (RU_ * T);
        invrho = 1.0 / rho;
        mconc = p / (RU_ * T);
        group_dispatch<0, G_>(grp, [&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            constexpr int k0 = group_first_species(g), k1 = group_first_species(g + 1);
            static_range<k0, k1>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                CL[k][tid] = rho * (k < LAST ? Y[k < LAST ? k - k0 : 0] : yN) * pjs::SP[k][0];
            });
        });
#else
        State L;
        load_state(A, s, L);        // all loads, scheduling barrier, sums (every group: each needs T, rho, ...)
        kc_land<NQ, NTHR>(LTK, NKC, lt);
        to_conc(L);
        T = L.T; rho = L.rho; invrho = L.invrho; Wbar = L.Wbar; mconc = L.mconc;
        if (G_ == 1 || grp == 0)
            static_for<NSP>([&](auto kc) PJR_INL { CL[decltype(kc)::value][tid] = L.C[decltype(kc)::value]; });
#endif
    }
    __syncthreads();
#endif
    if constexpr (ECL_PRO) ecl_fetch();
    if constexpr (ECL_PRO && EJ_LDS && !(PJQ_JV && EJ_LDS)) {
        int jfirst = 0;
        static_for<G_>([&](auto gc) PJR_INL { jfirst = grp == decltype(gc)::value ? group_first_col(decltype(gc)::value) : jfirst; });
        static_for<NCOLG>([&](auto pc) PJR_INL {
            constexpr int p = decltype(pc)::value;
            if (ecl_slot(pc) >= 0) SM[SM_EJ + (jfirst + p) * PJQ_BLOCK + tid] = ECLV[p];        // (wavefront-uniform)
        });
        __syncthreads();        // (the block that holds row j ADDS its finished sum: any lane group)
    }
#if defined(PJQ_STAGGER) && !defined(PJR_HOST_EMU)
    // experiment: shift the compute / store phases of neighbouring workgroups against each other
    {
        const int ph = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 3));
        for (int q = 0; q < ph; ++q) __builtin_amdgcn_s_sleep(PJQ_STAGGER);
    }
#endif
    // (not const: see PJQ_LAUNDER_EVERY)
    double logT = log(T), invT = 1.0 / T;
    // powers of T for the K_c polynomials (sum-of-products form: no dependent Horner chain)
    double T2 = T * T, T3 = T2 * T, T4 = T2 * T2;
    double T2d = 2.0 * T2, T3d = 3.0 * T3, T4d = 4.0 * T4;
    double WR = Wbar * invrho;
    // (every lane group writes the same values: no barrier, a wavefront reads what it -- or a twin -- wrote)
    if constexpr (NPARK > 0) SM[SM_PS + tid] = invT;
    if constexpr (NPARK > 1) SM[SM_PS + PJQ_BLOCK + tid] = logT;
    if constexpr (NPARK > 2) SM[SM_PS + 2 * PJQ_BLOCK + tid] = T;
    if constexpr (NPARK > 3) SM[SM_PS + 3 * PJQ_BLOCK + tid] = invrho;
    if constexpr (NPARK > 4) SM[SM_PS + 4 * PJQ_BLOCK + tid] = WR;
    if constexpr (NPARK > 5) SM[SM_PS + 5 * PJQ_BLOCK + tid] = mconc;
    // A DS instruction reaches 64 KB from its address register; the columns of a 53-species mechanism
    // span 106 KB.  Left alone the compiler keeps one address VGPR per column beyond the first 64 KB;
    // an opaque zero offset per 32-column group gives it one base per group instead.
    constexpr int CGRP = 65536 / (8 * PJQ_BLOCK) > 0 ? 65536 / (8 * PJQ_BLOCK) : 1;
    constexpr int NCG = (NSP + CGRP - 1) / CGRP;
    // (the bases are derived anew at every row block -- rebase() below: as kernel-long values they are what the register
    // allocator parks in scratch memory first, reloaded at every visit: 2 900 reloads in the w = J v kernel of the 53-species
    // mechanism, round 5)
    const double* clb[NCG];
    auto rebase_c = [&]() PJR_INL {
        static_for<NCG>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            unsigned zo = 0;
#ifndef PJR_HOST_EMU
            if constexpr (g > 0) asm volatile("" : "+v"(zo));
#endif
            clb[g] = (const double*)((const char*)&CL[g * CGRP][tid] + zo);
        });
    };
    rebase_c();
#if PJQ_VCT_ON
    // (opaque: the tables are compile-time constants and the optimiser would fold the loads back into literals; renewed per
    // row block like the column bases, or a reaction's constants are kept in SGPRs from one of its visits to the next)
    pjq_vptr vtb_ = (pjq_vptr)&VCTB.v[0][0], vte_ = (pjq_vptr)&VCTE.v[0][0];
    auto rebase_v = [&]() PJR_INL { asm volatile("" : "+s"(vtb_), "+s"(vte_)); };
    rebase_v();
#endif
    unsigned vzo = 0;      // opaque zero, renewed per visit (PJQ_CONC_OPAQUE): see the visit loop
    auto conc = [&](auto spc) PJR_INL {
        constexpr int sp = decltype(spc)::value;
        if constexpr (sp == ONE) return 1.0;
        else return ((const double*)((const char*)clb[sp / CGRP] + vzo))[(sp % CGRP) * PJQ_BLOCK];
    };
#if PJQ_KCF
    // factor columns: one base address each (16 bytes per species and lane: 64 KB hold 64 species of 64 states)
    constexpr int XGRP = 65536 / (16 * PJQ_BLOCK) > 0 ? 65536 / (16 * PJQ_BLOCK) : 1;
    constexpr int NXG = (NSP + XGRP - 1) / XGRP;
    const d2* xtb[NXG];
    const d2* ixtb[NXG];
    auto rebase_x = [&]() PJR_INL {
        static_for<NXG>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            unsigned z0 = 0, z1 = 0;
#ifndef PJR_HOST_EMU
            asm volatile("" : "+v"(z0), "+v"(z1));
#endif
            xtb[g] = (const d2*)((const char*)&XT[g * XGRP][tid] + z0);
            ixtb[g] = (const d2*)((const char*)&IXT[g * XGRP][tid] + z1);
        });
    };
    rebase_x();
    // {X_k or 1 / X_k, t_k} of net species q of reaction i: a product (nu > 0) divides K_c's inverse
    auto factor = [&](auto ic, auto qc) PJR_INL {
        constexpr int i = decltype(ic)::value, q = pjs::RI[i][RI_NET_PTR] + decltype(qc)::value;
        constexpr int k = pjs::NET_SP[q][0];
        if constexpr (pjs::NET_NU[q][0] > 0.0) return ixtb[k / XGRP][(k % XGRP) * PJQ_BLOCK];
        else return xtb[k / XGRP][(k % XGRP) * PJQ_BLOCK];
    };
#endif
#if PJQ_JV
    // the vector this state's Jacobian is applied to: read once per kernel, in registers (AGPRs) -- or, in the
    // one-kernel builds with several lane groups, in an LDS column set that the groups fill together (53 doubles next
    // to everything else spill, and every row block would reload them from scratch memory: 85 KB per state through
    // the vector memory path, 9.8 ms per 1e6 products)
    // What is kept is the SCALED vector vs_0 = v_0, vs_c = v_c / W_{c-1}: row k of the species block is
    //   J(k, c) = (1 / W_{c-1}) (W_k (P_k + S_k,c-1)) - W_k Q_k / W_N   (the output phase's form), so
    //   w_k = W_k JT_k v_0 + W_k P_k SV1 - (W_k Q_k / W_N) SV0 + W_k sum_{c-1 in nz(k)} S_k,c-1 vs_c
    // with the per-state sums SV0 = sum_{c>=1} v_c, SV1 = sum_{c>=1} vs_c: a row costs one LDS read and one FMA per
    // STRUCTURAL NON-ZERO instead of a read and three FMAs per column (round 4: 7.4 ms per 1e6 GRI-shaped products,
    // slower than writing the Jacobians).
    double V[JV_LDS ? 1 : NSP];
    double WE = 0.0;        // JV_LDS: sum_j E^A_j v_{j+1} / W_j, the finished column sums' share of w_0
    double SV0 = 0.0, SV1 = 0.0;
    {
        const double* vp = A.v + s * A.v_ss;
        if constexpr (JV_LDS) {
            static_for<G_>([&](auto gc) PJR_INL {
                constexpr int g = decltype(gc)::value;
                if (grp == g)
                    static_range<group_first_species(g), group_first_species(g + 1)>([&](auto cc) PJR_INL {
                        constexpr int c = decltype(cc)::value;
                        const double vc = vp[c * A.v_si];
                        double vsc = vc;
                        if constexpr (c > 0) vsc = vc * pjs::SP[c - 1][0];
                        SM[SM_EJ + c * PJQ_BLOCK + tid] = vsc;
                    });
            });
            __syncthreads();
            // every lane group needs the whole sums: each takes them from the columns (NSP - 1 reads: once per state)
            static_range<1, NSP>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value;
                const double vsc = SM[SM_EJ + c * PJQ_BLOCK + tid];
                SV1 += vsc;
                SV0 += vsc * pjs::SP[c - 1][1];
            });
        } else {
            static_for<NSP>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value;
                const double vc = vp[c * A.v_si];
                if constexpr (c > 0) { V[c] = vc * pjs::SP[c - 1][0]; SV0 += vc; SV1 += V[c]; } else V[c] = vc;
            });
        }
    }
    // vs_c
    auto vv = [&](auto cc) PJR_INL {
        if constexpr (JV_LDS) return SM[SM_EJ + decltype(cc)::value * PJQ_BLOCK + tid];
        else return V[decltype(cc)::value];
    };
    if constexpr (ECL_PRO && JV_LDS) {
        int jfirst = 0;
        static_for<G_>([&](auto gc) PJR_INL { jfirst = grp == decltype(gc)::value ? group_first_col(decltype(gc)::value) : jfirst; });
        static_for<NCOLG>([&](auto pc) PJR_INL {
            constexpr int p = decltype(pc)::value;
            // 1 / W_j of this group's p-th column (scalar selects) and v_{j+1} (LDS, run-time column); unmarked: ECLV = 0
            double iw = 0.0;
            static_for<G_>([&](auto gc) PJR_INL {
                constexpr int g = decltype(gc)::value, j = group_first_col(g) + p;
                iw = grp == g ? pjs::SP[j < LAST ? j : 0][0] : iw;
            });
            const int jv_ = jfirst + p + 1 < NSP ? jfirst + p + 1 : 0;
            (void)iw;
            WE += ECLV[p] * SM[SM_EJ + jv_ * PJQ_BLOCK + tid];          // (the column holds v_{j+1} / W_j)
        });
    }
    double* const wp = A.w + s * A.w_ss;
#endif
    // energy-row partial sums: touched once per block, the register allocator parks them in AGPRs
    double E[LAST > 0 ? LAST : 1];
    double H = 0.0, SCP = 0.0, SJT = 0.0, HP = 0.0, HQN = 0.0;
    // this lane's column of the hand-over array: a wavefront-uniform 64-bit base (the first lane's address: scalar
    // registers, slot offsets by scalar arithmetic) + a 32-bit per-lane byte offset -- as a per-lane 64-bit pointer it is a
    // kernel-long register pair that the allocator parks in scratch memory and reloads in front of every hand-over load
    // (and two vector additions per access)
    // (idx = slot * PJQ_TILE at every call site; the slot is multiplied by the RUN-TIME tile size -- with the constant the
    // optimiser folds base + off into one per-lane 64-bit address again and adds the slot offsets with vector arithmetic)
    struct ScrRef {
        double* base;
        unsigned off;
        long tile;
        __device__ __forceinline__ double& operator[](const long idx) const
        {
            return *(double*)((char*)(base + (idx / PJQ_TILE) * tile + (idx % PJQ_TILE)) + off);
        }
    };
    ScrRef scr;
    scr.tile = A.tile_rt;
    {
        double* const mine = scr_of(A, s);
#ifdef PJR_HOST_EMU
        scr.base = mine;
        scr.off = 0u;
#else
        // (pointer arithmetic on the kernel argument with a SCALAR state index: the address space survives -- rebuilt from
        // readfirstlane'd integer halves the pointer is a flat one, and flat loads count on lgkmcnt as well)
        const long sw_ = ((long)__builtin_amdgcn_readfirstlane((int)((unsigned long)s >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s);
        scr.base = scr_of(A, sw_);
        scr.off = (unsigned)((mine - scr.base) * 8);
#endif
    }
    const long hset = (long)grp * (2 * NSUM) * PJQ_TILE;      // this group's pair of slot sets
    // (sums that live in LDS: this lane's slot of column j is el[ELM.slot[j] * PJQ_BLOCK])
    double* const el = SM + SM_EL + (long)grp * NEL_ * PJQ_BLOCK + tid;
    auto e_add = [&](auto jc, const double v) PJR_INL {
        constexpr int j = decltype(jc)::value;
        if constexpr (ELM.slot[j] >= 0) PJQ_LDS_ADD(el + ELM.slot[j] * PJQ_BLOCK, v);
        else E[j] += v;
    };
    if constexpr (FIRST_) {
        static_for<LAST>([&](auto jc) PJR_INL {
            constexpr int j = decltype(jc)::value;
            if constexpr (ELM.slot[j] >= 0) el[ELM.slot[j] * PJQ_BLOCK] = 0.0; else E[j] = 0.0;
        });
    } else {
        // partial sums of the previous row kernel: fetched here, next to the state loads, so that no
        // kernel ever waits for a load behind its own Jacobian stores
        static_for<LAST>([&](auto jc) PJR_INL {
            constexpr int j = decltype(jc)::value;
            double e_in = 0.0;
#ifndef PJQ_NO_E
            if constexpr (e_live_col(j)) e_in = scr[hset + (long)(SUM_IN + 5 + j) * PJQ_TILE];
#endif
            if constexpr (ELM.slot[j] >= 0) el[ELM.slot[j] * PJQ_BLOCK] = e_in; else E[j] = e_in;
        });
        H = scr[hset + (long)SUM_IN * PJQ_TILE];
        SCP = scr[hset + (long)(SUM_IN + 1) * PJQ_TILE];
        SJT = scr[hset + (long)(SUM_IN + 2) * PJQ_TILE];
        HP = scr[hset + (long)(SUM_IN + 3) * PJQ_TILE];
        HQN = scr[hset + (long)(SUM_IN + 4) * PJQ_TILE];
    }
    // Jacobian entry e of this lane's state: wavefront-uniform 64-bit base (entry offset e * j_si and
    // the wavefront's first state: scalar arithmetic) + a 32-bit per-lane byte offset, so that a store
    // is one instruction with an SGPR base and no 64-bit vector address arithmetic
#ifdef PJR_HOST_EMU
    const long s_wave = s;
#else
    const long s_wave = ((long)__builtin_amdgcn_readfirstlane((int)((unsigned long)s >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s);
#endif
    double* const Jw = A.jac + s_wave * A.j_ss;
    const unsigned jvo = (unsigned)((s - s_wave) * A.j_ss) * 8u;
    // The kernel argument block arrives as ONE 16-dword scalar load, and the register allocator treats its 16 SGPRs as one
    // value: whoever needs a field in the steady state (the entry stride of every store address, sum_last) keeps all 16
    // alive, they are spilled to VGPR lanes as a tuple and come back as a tuple -- 137 x 16 v_readlane_b32 per state in the
    // 53-species kernel (round 6: 2 192 of its 3 861 v_readlane, each an issue slot at one wavefront per SIMD).  Opaque copies
    // of the two fields the blocks read cut them loose.
    long jsi = A.j_si;
    int sum_last_ = A.sum_last;
#ifndef PJR_HOST_EMU
    // (real moves: an empty asm with "+s" operands is coalesced with the tuple's sub-registers again, and the 16 SGPRs stay one
    // live value -- measured, round 6: 3 884 v_readlane_b32 with it, as many as without)
    asm volatile("s_mov_b64 %0, %2\n\ts_mov_b32 %1, %3" : "=&s"(jsi), "=&s"(sum_last_) : "s"(A.j_si), "s"(A.sum_last));
#endif
#define J_(e) (*(double*)((char*)(Jw + (long)(e) * jsi) + jvo))
#if PJQ_PAIR
    // pair stores (SoA only, host-checked: j_ss == 1, 8 * NSP * j_si < 2^32): a lane of the lower half
    // addresses its own (even) state in column c, its partner in the upper half that state in column c + 1
    const bool upper = (tid & 32) != 0;
    const unsigned jvo2 = upper ? jvo - 8u + (unsigned)(NSP * jsi) * 8u : jvo;
#endif

    // hand-over values of the falloff / PLOG visits: those visits come last in a block
    // (pj::emit_rows_tables), their values are fetched PJQ_DEPTH visits ahead into a register ring.
    // vmcnt counts loads and stores in order, so a load waits for every store issued before it: the
    // first loads of a block go out right after the previous block's stores and are consumed after
    // the block's Arrhenius visits, by which time those stores have drained.
    double ring[PJQ_DEPTH][6];
    auto issue_pre = [&](auto bc, auto pc) PJR_INL {
        constexpr int b = decltype(bc)::value, pp = decltype(pc)::value;
        constexpr int v1 = pjs::BLK_RX_PTR[b + 1][0];
        constexpr int i = pjs::BLK_RX[v1 - n_pre_visits<b>() + pp][0];
        static_assert(is_pre(i), "falloff / PLOG visits must come last in a block");
        static_for<6>([&](auto cc) PJR_INL {
            constexpr int c = decltype(cc)::value;
            if constexpr (pjs::SCQ[i][c] >= 0)
                ring[pp % PJQ_DEPTH][c] = PJQ_LOAD_NT(&scr[(long)pjs::SCQ[i][c] * PJQ_TILE]);
        });
    };

    // One row of a block as store instructions: `piece` h of row r is the column pair (2 h, 2 h + 1) of the pair-store builds
    // (the odd last column alone), column h otherwise; WPr / WQNr / JTr: the row's constants, Sb: the block's sparse sums
    constexpr int PIECES = PJQ_PAIR ? (NSP + 1) / 2 : NSP;
    auto row_piece = [&](auto bc, auto rc, auto hc, const double WPr, const double WQNr, const double WCNr, const double JTr, const double* Sb) PJR_INL {
        constexpr int b = decltype(bc)::value, r = decltype(rc)::value, h = decltype(hc)::value;
        constexpr int k = pjs::BLK_ROWS[pjs::BLK_ROW_PTR[b][0] + r][0];
        // Jacobian column c of the row: c = 0 is the d/dT column, c = j + 1 belongs to species j
        auto cv = [&](auto cc) PJR_INL {
            constexpr int c = decltype(cc)::value;
            if constexpr (c == 0) {
                return pjs::SP[k][1] * JTr;                  // create_jacobian.py:2786-2818
            } else {
                constexpr int j = c - 1;
                constexpr int si = pjs::SLOC[k][j];
                if constexpr (near_last(j)) {
                    // (1 / W_j - 1 / W_N) W_k P_k - W_k QN_k / W_N [+ (1 / W_j) W_k S_kj]: see near_last()
                    constexpr double a1 = pjs::SP[j][0] * (1.0 - pjs::SP[j][3]);
                    if constexpr (si >= 0) return INVW(j) * (pjs::SP[k][1] * Sb[si]) + (a1 * WPr - WCNr);
                    else return a1 * WPr - WCNr;
                } else {
                    if constexpr (si >= 0) return INVW(j) * (WPr + pjs::SP[k][1] * Sb[si]) - WQNr;
                    else return INVW(j) * WPr - WQNr;
                }
            }
        };
        if constexpr (k < LAST) {
#if PJQ_PAIR
            constexpr int c = 2 * h;
            if constexpr (c + 1 < NSP) {
                double v0 = cv(std::integral_constant<int, c>{});
                double v1 = cv(std::integral_constant<int, c + 1>{});
                swap_halves(v0, v1);
                d2s out;
                out.x = v0;
                out.y = v1;
#if defined(PJQ_ST_BITS) && !defined(PJQ_NO_STORE)
                // experiment: the pair store with explicit cache-policy bits (sc0 / sc1 / nt in any combination) instead of the
                // compiler's "nt": scalar base + 32-bit lane offset, as the compiler forms it
                {
                    double* const sb_ = Jw + (long)(k + 1 + NSP * c) * jsi;
                    asm volatile("global_store_dwordx4 %0, %1, %2 " PJQ_ST_BITS :: "v"(jvo2), "v"(out), "s"(sb_) : "memory");
                }
#else
                PJQ_STORE2((d2s*)((char*)(Jw + (long)(k + 1 + NSP * c) * jsi) + jvo2), out);
#endif
            } else {
                PJQ_STORE(&J_(k + 1 + NSP * c), cv(std::integral_constant<int, c>{}));
            }
#elif !PJQ_JV
            PJQ_STORE(&J_(k + 1 + NSP * h), cv(hc));
#endif
        }
    };
    auto run_blocks = [&](auto lo_c, auto hi_c) PJR_INL {
    constexpr int LO_ = decltype(lo_c)::value, HI_ = decltype(hi_c)::value;
    // PJQ_DEFER: what the stores of a block need, kept until the next block has issued them
    constexpr bool DEFER = PJQ_DEFER && !PJQ_JV;
    double pWP[DEFER ? pjs::BLK_MAXROWS : 1], pWQN[DEFER ? pjs::BLK_MAXROWS : 1], pWCN[DEFER ? pjs::BLK_MAXROWS : 1], pJT[DEFER ? pjs::BLK_MAXROWS : 1];
    double pS[DEFER ? pjs::BLK_MAXNNZ : 1];
    static_range<LO_, HI_>([&](auto bc) PJR_INL {
        constexpr int b = decltype(bc)::value;
        constexpr int r0 = pjs::BLK_ROW_PTR[b][0], nrows = pjs::BLK_ROW_PTR[b + 1][0] - r0;
        constexpr int v0 = pjs::BLK_RX_PTR[b][0], nv = pjs::BLK_RX_PTR[b + 1][0] - v0;
#if PJQ_LAUNDER_EVERY && !defined(PJR_HOST_EMU)
        // k_f, K_c and T dlnK_c/dT of a reaction are functions of T alone, and the optimiser knows: it
        // evaluates them at the first visit and keeps them for every later block of the kernel -- in
        // AGPRs while they last, then in scratch memory, whose loads queue behind the Jacobian
        // stores.  Opaque copies of T's functions every few blocks bound that cache.
        // (with several lane groups also at a group's first block: k_f of a reaction that two groups visit is
        // otherwise computed once, in front of the group branches, and kept)
        if constexpr ((b - LO_) % PJQ_LAUNDER_EVERY == 0 && (b != LO_ || G_ > 1))
        {
#ifndef PJQ_NO_REBASE
            rebase_c();
#if PJQ_KCF
            rebase_x();
#endif
#if PJQ_VCT_ON
            rebase_v();
#endif
#endif
            // (the powers of T are REBUILT from the opaque copy, six multiplications per block: as opaque copies of their own
            // they are six more per-state values that live through the whole kernel, and in the kernels that are short of
            // registers exactly those are kept in scratch memory and reloaded at every visit -- each reload behind every
            // Jacobian store issued before it: 2 900 reloads in the first kernel of a 64-state / four-group USC-shaped build)
            if constexpr (NPARK > 0) {
                unsigned pzo = 0;
                asm volatile("" : "+v"(pzo));
                const double* const ps = (const double*)((const char*)(SM + SM_PS + tid) + pzo);
                invT = ps[0];
                if constexpr (NPARK > 1) logT = ps[PJQ_BLOCK];
                if constexpr (NPARK > 2) T = ps[2 * PJQ_BLOCK];
                if constexpr (NPARK > 3) invrho = ps[3 * PJQ_BLOCK];
                if constexpr (NPARK > 4) WR = ps[4 * PJQ_BLOCK];
                if constexpr (NPARK > 5) mconc = ps[5 * PJQ_BLOCK];
            }
            if constexpr (NPARK == 0) asm volatile("" : "+v"(T), "+v"(logT), "+v"(invT));
            else if constexpr (NPARK == 1) asm volatile("" : "+v"(T), "+v"(logT));
            else if constexpr (NPARK == 2) asm volatile("" : "+v"(T));
            T2 = T * T; T3 = T2 * T; T4 = T2 * T2;
            T2d = 2.0 * T2; T3d = 3.0 * T3; T4d = 4.0 * T4;
        }
#endif
        double om[nrows], P[nrows], QN[nrows], JT[nrows], S[pjs::BLK_NNZ[b][0] > 0 ? pjs::BLK_NNZ[b][0] : 1];
        double EA[nrows];           // energy row, column of each row of the block: sum_i Hr_i G_ij over the block's visits
        double JTQ = 0.0;
        static_for<nrows>([&](auto rc) PJR_INL {
            constexpr int r = decltype(rc)::value;
            om[r] = 0.0; P[r] = 0.0; QN[r] = 0.0; JT[r] = 0.0; EA[r] = 0.0;
        });
        static_for<pjs::BLK_NNZ[b][0]>([&](auto ec) PJR_INL { S[decltype(ec)::value] = 0.0; });
        constexpr int npre = n_pre_visits<b>();
        static_for<(npre < PJQ_DEPTH ? npre : PJQ_DEPTH)>([&](auto pc) PJR_INL { issue_pre(bc, pc); });
        PJQ_SCHED_BARRIER();
        if constexpr (b == LO_) PJQ_TICK(0)

#if PJQ_KC_AHEAD
        // The K_c data of visit v + 1 -- polynomial rows, or the net species' factor pairs (PJQ_KCF) -- are read
        // from LDS while visit v is computed (software pipelining by hand: a visit's first consumer of LDS data
        // are those values, and at one wavefront per SIMD nothing else covers the ~120-cycle LDS round trip at
        // the top of every visit)
#if PJQ_VCT_ON
        double vcb[2][4], veb[2][VEW];
        auto fetch_vc = [&](auto vc) PJR_INL {
            constexpr int v = decltype(vc)::value;
            constexpr int i = pjs::BLK_RX[v0 + v][0];
            constexpr int fl_ = pjs::RI[i][RI_FLAGS];
            if constexpr (!is_pre(i) || (fl_ & F_REV))
                static_for<4>([&](auto cc) PJR_INL { vcb[v & 1][decltype(cc)::value] = vtb_[i * 4 + decltype(cc)::value]; });
            if constexpr (vct_has_eff(i))
                static_for<VEW>([&](auto cc) PJR_INL { veb[v & 1][decltype(cc)::value] = vte_[vct_eff_index(i) * VEW + decltype(cc)::value]; });
        };
#endif
#if PJQ_KCF
        d2 xab[2][MAXNET];
        auto fetch_ka = [&](auto vc) PJR_INL {
            constexpr int v = decltype(vc)::value;
            constexpr int i = pjs::BLK_RX[v0 + v][0];
            constexpr int NN = (pjs::RI[i][RI_FLAGS] & F_REV) ? pjs::RI[i][RI_NET_CNT] : 0;
            static_for<NN>([&](auto qc) PJR_INL { xab[v & 1][decltype(qc)::value] = factor(std::integral_constant<int, i>{}, qc); });
#if PJQ_VCT_ON
            fetch_vc(vc);
#endif
        };
#else
        double kab[2][MAXKC][7];
        auto fetch_ka = [&](auto vc) PJR_INL {
            constexpr int v = decltype(vc)::value;
            constexpr int i = pjs::BLK_RX[v0 + v][0];
            constexpr int KC = (pjs::RI[i][RI_FLAGS] & F_REV) ? pjs::RI[i][RI_KC_CNT] : 0;
            static_for<KC>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value, g = pjs::RI[i][RI_KC_PTR] + c;
                const double* a = LTK + KCM.loc[g] * 16 + ((T <= pjs::KCG[g][0]) ? 0 : 8);
                static_for<7>([&](auto ec) PJR_INL { kab[v & 1][c][decltype(ec)::value] = a[decltype(ec)::value]; });
            });
#if PJQ_VCT_ON
            fetch_vc(vc);
#endif
        };
#endif
        if constexpr (nv > 0) fetch_ka(std::integral_constant<int, 0>{});
#endif
#if PJQ_CONC_AHEAD
        // the six concentration reads of visit v + 1 likewise travel while visit v is computed
        double cab[2][6];
        auto fetch_c = [&](auto vc) PJR_INL {
            constexpr int v = decltype(vc)::value;
            constexpr int i = pjs::BLK_RX[v0 + v][0];
            cab[v & 1][0] = conc(std::integral_constant<int, pjs::RI[i][RI_R0]>{});
            cab[v & 1][1] = conc(std::integral_constant<int, pjs::RI[i][RI_R1]>{});
            cab[v & 1][2] = conc(std::integral_constant<int, pjs::RI[i][RI_R2]>{});
            cab[v & 1][3] = conc(std::integral_constant<int, pjs::RI[i][RI_P0]>{});
            cab[v & 1][4] = conc(std::integral_constant<int, pjs::RI[i][RI_P1]>{});
            cab[v & 1][5] = conc(std::integral_constant<int, pjs::RI[i][RI_P2]>{});
        };
        if constexpr (nv > 0) fetch_c(std::integral_constant<int, 0>{});
#endif
        if constexpr (DEFER && b > LO_ && nv - npre == 0) {
            // (a block without Arrhenius visits: the previous block's rows go out here)
            constexpr int pb = b > LO_ ? b - 1 : b;      // (b - 1; the first block has none)
            constexpr int pnrows = pjs::BLK_ROW_PTR[pb + 1][0] - pjs::BLK_ROW_PTR[pb][0];
            static_for<pnrows>([&](auto rc) PJR_INL {
                static_for<PIECES>([&](auto hc) PJR_INL {
                    row_piece(std::integral_constant<int, pb>{}, rc, hc, pWP[decltype(rc)::value], pWQN[decltype(rc)::value],
                              pWCN[decltype(rc)::value], pJT[decltype(rc)::value], pS);
                });
            });
        }
        static_for<nv>([&](auto vc) PJR_INL {
            constexpr int v = decltype(vc)::value;
            constexpr int i = pjs::BLK_RX[v0 + v][0];
            constexpr int fl = pjs::RI[i][RI_FLAGS];
            constexpr double nr = pjs::RD[i][RD_NR], np_ = pjs::RD[i][RD_NP];
#if PJQ_VCT_ON
            // this visit's constants: SGPRs filled a visit ago (fetch_vc); a constant that is zero or free to encode stays a
            // literal, so that b = 0 / T_a = 0 still fold their terms away
#define VC_(f_, lit_) (cheap_literal(lit_) ? (lit_) : vcb[v & 1][f_])
#define VE_(e_) (cheap_literal(pjs::EFF_AM1[e_][0]) ? pjs::EFF_AM1[e_][0] : veb[v & 1][1 + (e_) - pjs::RI[i][RI_EFF_PTR]])
            const double c_lna = VC_(0, pjs::RD[i][RD_LNA]), c_b = VC_(1, pjs::RD[i][RD_B]), c_ta = VC_(2, pjs::RD[i][RD_TA]);
            const double c_pref = VC_(3, (PJQ_KCF ? pjs::KCF_PREFINV[i][0] : pjs::RD[i][RD_LNPREF]));
            const double c_anm1 = cheap_literal(pjs::RD[i][RD_ANM1]) ? pjs::RD[i][RD_ANM1] : veb[v & 1][0];
#else
#define VE_(e_) EFC(e_)
            const double c_lna = RDC(i, RD_LNA), c_b = RDC(i, RD_B), c_ta = RDC(i, RD_TA);
            const double c_pref = PJQ_KCF ? pjs::KCF_PREFINV[i][0] : RDC(i, RD_LNPREF);
            const double c_anm1 = RDC(i, RD_ANM1);
#endif
            (void)c_lna; (void)c_b; (void)c_ta; (void)c_pref; (void)c_anm1;
#if PJQ_CONC_OPAQUE && !defined(PJR_HOST_EMU)
            // the concentration columns never change, so the optimiser would merge all reads of a
            // column into one load and keep the value live across the kernel (register pressure)
            asm volatile("" : "+v"(vzo));
#endif
#if PJQ_CONC_AHEAD
            const double cr0 = cab[v & 1][0], cr1 = cab[v & 1][1], cr2 = cab[v & 1][2];
            const double cp0 = cab[v & 1][3], cp1 = cab[v & 1][4], cp2 = cab[v & 1][5];
            if constexpr (v + 1 < nv) fetch_c(std::integral_constant<int, v + 1>{});
#else
            const double cr0 = conc(std::integral_constant<int, pjs::RI[i][RI_R0]>{}),
                         cr1 = conc(std::integral_constant<int, pjs::RI[i][RI_R1]>{}),
                         cr2 = conc(std::integral_constant<int, pjs::RI[i][RI_R2]>{});
            const double cp0 = conc(std::integral_constant<int, pjs::RI[i][RI_P0]>{}),
                         cp1 = conc(std::integral_constant<int, pjs::RI[i][RI_P1]>{}),
                         cp2 = conc(std::integral_constant<int, pjs::RI[i][RI_P2]>{});
#endif
            // ---- phase A: everything that has to travel (LDS reads of the concentration columns and
            //      of the K_c polynomial rows / factor pairs) next to arithmetic that needs none of it (k_f) ----
            constexpr int KCNT = (fl & F_REV) ? pjs::RI[i][RI_KC_CNT] : 0;
#if PJQ_KCF
            constexpr int NNET = (fl & F_REV) ? pjs::RI[i][RI_NET_CNT] : 0;
            d2 xf[NNET > 0 ? NNET : 1];
#if PJQ_KC_AHEAD
            static_for<NNET>([&](auto qc) PJR_INL { xf[decltype(qc)::value] = xab[v & 1][decltype(qc)::value]; });
            if constexpr (v + 1 < nv) fetch_ka(std::integral_constant<int, v + 1>{});
#else
            static_for<NNET>([&](auto qc) PJR_INL { xf[decltype(qc)::value] = factor(std::integral_constant<int, i>{}, qc); });
#endif
#else
            double ka[KCNT > 0 ? KCNT : 1][7];
#if PJQ_KC_AHEAD
            static_for<KCNT>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value;
                static_for<7>([&](auto ec) PJR_INL { ka[c][decltype(ec)::value] = kab[v & 1][c][decltype(ec)::value]; });
            });
            if constexpr (v + 1 < nv) fetch_ka(std::integral_constant<int, v + 1>{});
#else
            static_for<KCNT>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value, g = pjs::RI[i][RI_KC_PTR] + c;
                const double* a = LTK + KCM.loc[g] * 16 + ((T <= pjs::KCG[g][0]) ? 0 : 8);
                static_for<7>([&](auto ec) PJR_INL { ka[c][decltype(ec)::value] = a[decltype(ec)::value]; });
            });
#endif
#endif
#if PJQ_SPLIT
            PJQ_SCHED_BARRIER();
#endif
            // ---- phase B ----
            double pr_ = cr0 * cr1 * cr2, pp_ = cp0 * cp1 * cp2;
            // general stoichiometry: the molecule slots are empty, the factors multiply in here
            constexpr int GP = pjs::RI[i][RI_GEN_PTR];
            constexpr int GNR = (fl & F_GEN) ? pjs::RI[i][RI_GEN_NR] : 0;
            constexpr int GNP = ((fl & F_GEN) && (fl & F_REV)) ? pjs::RI[i][RI_GEN_NP] : 0;
            double gcf[GNR + GNP > 0 ? GNR + GNP : 1], gpw[GNR + GNP > 0 ? GNR + GNP : 1];
            static_for<GNR + GNP>([&](auto fc) PJR_INL {
                constexpr int f = decltype(fc)::value;
                gcf[f] = conc(std::integral_constant<int, pjs::GEN_SP[GP + f][0]>{});
                gpw[f] = gen_pow<GP + f>(gcf[f]);
                if constexpr (f < GNR) pr_ *= gpw[f]; else pp_ *= gpw[f];
            });
            // Arrhenius (rate_subs.py:113-147); exp(-ln K_c) and T dlnK_c/dT: from the pre-summed NASA
            // polynomials of the reaction's groups (rate_subs.py:660-809; the two exponentials of a
            // reversible reaction are evaluated side by side), or -- PJQ_KCF -- as the product of the net
            // species' factors and the sum of their t_k: no polynomial, no second exponential
            double kf = 0.0, ekc = 0.0, td = 0.0, lnk = 0.0, lnKc = 0.0;
            // (T_a / T first: it is wanted again for d ln k / dT, and every instruction then carries ONE constant -- the
            // constant bus of gfx9 takes one scalar operand, a second constant is copied into a vector register pair first)
            double taT = 0.0;
            if constexpr (!is_pre(i)) {
#if PJQ_VCT_ON
                if constexpr (pjs::RD[i][RD_TA] != 0.0) taT = c_ta * invT;
                if constexpr (pjs::RD[i][RD_B] != 0.0) lnk = (c_b * logT - taT) + c_lna;
                else lnk = c_lna - taT;
#else
                lnk = c_lna + c_b * logT - c_ta * invT;
#endif
            }
#if PJQ_KCF
            if constexpr ((fl & F_REV) != 0) {
                ekc = c_pref;
                static_for<NNET>([&](auto qc) PJR_INL {
                    constexpr int q = decltype(qc)::value;
                    constexpr double nu = pjs::NET_NU[pjs::RI[i][RI_NET_PTR] + q][0];
                    constexpr int m = (int)(nu < 0.0 ? -nu : nu);
                    static_assert((double)m == (nu < 0.0 ? -nu : nu), "PJQ_KCF: whole net coefficients");
                    static_for<m>([&](auto) PJR_INL { ekc *= xf[q].x; });
                    if constexpr (!is_pre(i)) td += nu * xf[q].y;
                });
            }
            if constexpr (!is_pre(i)) kf = PJQ_EXP1(lnk);
            (void)lnKc;
#else
            if constexpr ((fl & F_REV) != 0) {
                lnKc = c_pref;
                static_for<KCNT>([&](auto cc) PJR_INL {
                    constexpr int c = decltype(cc)::value;
                    const double* a = ka[c];
                    lnKc += a[0] + a[1] * logT + a[2] * T + a[3] * T2 + a[4] * T3 + a[5] * T4 - a[6] * invT;
                    if constexpr (!is_pre(i))
                        td += a[1] + a[2] * T + a[3] * T2d + a[4] * T3d + a[5] * T4d + a[6] * invT;
                });
            }
            if constexpr (!is_pre(i) && (fl & F_REV) != 0) PJQ_EXP2(lnk, -lnKc, kf, ekc);
            else if constexpr (!is_pre(i)) kf = PJQ_EXP1(lnk);
            else if constexpr ((fl & F_REV) != 0) ekc = PJQ_EXP1(-lnKc);
#endif
            if constexpr (pjs::RD[i][RD_SGN] < 0.0) kf = -kf;

            double ckf, ckr = 0.0, theta = 0.0, rp, bM = 0.0, bcol = 0.0;
            double kf_slot = 0.0;       // Chebyshev: the k_f eval_jacob uses in its dR/dY_j terms (pj_rate_pre.inc)
            if constexpr (is_pre(i)) {
                // falloff / PLOG: handed over by k_pre
                constexpr int pp = v - (nv - npre);
                static_assert(pp >= 0, "falloff / PLOG visits must come last in a block");
                ckf = ring[pp % PJQ_DEPTH][S_KF];
                theta = ring[pp % PJQ_DEPTH][S_TH];
                double rp_ld = 0.0;
                if constexpr (pjs::SCQ[i][S_BM] >= 0) bM = ring[pp % PJQ_DEPTH][S_BM];
                if constexpr (pjs::SCQ[i][S_BC] >= 0) bcol = ring[pp % PJQ_DEPTH][S_BC];
                if constexpr (pjs::SCQ[i][S_RP] >= 0) rp_ld = ring[pp % PJQ_DEPTH][S_RP];
                double kfj_ld = 0.0;
                if constexpr (pjs::SCQ[i][S_KR] >= 0) kfj_ld = ring[pp % PJQ_DEPTH][S_KR];
                if constexpr (pp + PJQ_DEPTH < npre) issue_pre(bc, std::integral_constant<int, pp + PJQ_DEPTH>{});
                if constexpr (pjs::SCQ[i][S_KR] >= 0) kf_slot = kfj_ld;
                if constexpr ((fl & F_REV) != 0) ckr = ckf * ekc;
                if constexpr (pjs::SCQ[i][S_RP] >= 0) rp = rp_ld;
                else rp = WR * ((1.0 - nr) * (ckf * pr_) - ((fl & F_REV) ? (1.0 - np_) * (ckr * pp_) : 0.0));
            } else {
                // optional third body (rate_subs.py:1076-1130)
                const double Rf = kf * pr_;
                double Rr = 0.0;
                if constexpr ((fl & F_REV) != 0) Rr = (kf * ekc) * pp_;
                const double R = Rf - Rr;
                double c = 1.0, lead = 0.0;
                if constexpr ((fl & F_THD) != 0) {
                    double Mc = mconc;
                    static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJR_INL {
                        constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                        Mc += VE_(e) * conc(std::integral_constant<int, pjs::EFF_SP[e][0]>{});
                    });
                    c = Mc;
                    lead = -c * R * invT;
                    if constexpr ((fl & F_EFFTYPE) != 0) bM = R;
                }
                if constexpr ((fl & F_NO_DT) == 0) {
#if PJQ_VCT_ON
                    const double dlnk = c_b + taT;
#else
                    const double dlnk = c_b + c_ta * invT;
#endif
                    double el = R * dlnk + Rf * (1.0 - nr);
                    if constexpr ((fl & F_REV) != 0) el -= Rr * ((1.0 - np_) - td);
                    theta = (lead + c * invT * el) * invrho;
                }
                ckf = c * kf;
                if constexpr ((fl & F_REV) != 0) ckr = ckf * ekc;
                if constexpr ((fl & F_THD) != 0) {
                    // create_jacobian.py:341-489: a_i and the b_i * [M] term of a third-body reaction
                    double a = c * (nr * Rf - ((fl & F_REV) ? np_ * Rr : 0.0));
                    if constexpr ((fl & F_EFFTYPE) != 0) a += c * R;
                    rp = WR * (c * R - a) + bM;
                } else {
                    rp = WR * ((1.0 - nr) * Rf - ((fl & F_REV) ? (1.0 - np_) * Rr : 0.0));
                }
            }
            const double q_ = ckf * pr_ - ckr * pp_;

            double gN = 0.0;
            if constexpr (has_anm1<i>()) gN = bM * c_anm1;
            constexpr int np0 = pjs::RI[i][RI_NET_PTR], ncnt = pjs::RI[i][RI_NET_CNT];
            // reaction enthalpy Hr_i = sum_k nu_ki h_kW_k = R T (sum_k nu_ki t_k + sum nu), t_k = h_k/RT - 1
            // (dead code in the visits that add nothing to the energy row)
            double hrt = 0.0;
#if PJQ_KCF
            if constexpr ((fl & F_REV) != 0) {
                static_for<NNET>([&](auto qc) PJR_INL {
                    hrt += pjs::NET_NU[np0 + decltype(qc)::value][0] * xf[decltype(qc)::value].y;
                });
            } else {
                static_for<ncnt>([&](auto qc) PJR_INL {
                    constexpr int k = pjs::NET_SP[np0 + decltype(qc)::value][0];
                    hrt += pjs::NET_NU[np0 + decltype(qc)::value][0] * xtb[k / XGRP][(k % XGRP) * PJQ_BLOCK].y;
                });
            }
            constexpr double nsum = net_sum(i);
            const double Hr = (RU_ * T) * (hrt + nsum);
#else
            double Hr;
            if constexpr ((fl & F_REV) != 0) {
                static_for<KCNT>([&](auto cc) PJR_INL {
                    const double* a = ka[decltype(cc)::value];
                    hrt += a[1] + a[2] * T + a[3] * T2d + a[4] * T3d + a[5] * T4d + a[6] * invT;
                });
                constexpr double nsum = net_sum(i);
                Hr = (RU_ * T) * (hrt + nsum);
            } else {
                // an irreversible reaction has no K_c polynomial: the species' enthalpies from their NASA coefficients
                Hr = 0.0;
                static_for<ncnt>([&](auto qc) PJR_INL {
                    constexpr int k = pjs::NET_SP[np0 + decltype(qc)::value][0];
                    const bool lo = T <= pjs::SP[k][2];
                    double a[6];
                    static_for<6>([&](auto cc) PJR_INL {
                        constexpr int c = decltype(cc)::value;
                        a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
                    });
                    Hr += pjs::NET_NU[np0 + decltype(qc)::value][0] *
                          (RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                  T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T))))));
                });
            }
            (void)hrt;
#endif
            auto slot = [&](auto spc, const double gv) PJR_INL {
                constexpr int sp = decltype(spc)::value;
                if constexpr (sp == LAST) gN += gv;
                else if constexpr (sp != ONE) {
                    static_for<ncnt>([&](auto qc) PJR_INL {
                        constexpr int q = np0 + decltype(qc)::value;
                        constexpr int k = pjs::NET_SP[q][0];
                        if constexpr (pjs::ROW_BLK[k][0] == b) {
                            constexpr int si = pjs::SLOC[k][sp];
                            static_assert(si >= 0, "sparse pattern and program disagree");
                            S[si] += pjs::NET_NU[q][0] * gv;
                        }
                    });
#ifndef PJQ_NO_E
                    // energy row: column sp is finished by the block that holds row sp (it visits every reaction
                    // with sp as a net species); what that block cannot see is added once, at the reaction's first visit
                    if constexpr (in_net(i, sp)) {
                        if constexpr (pjs::ROW_BLK[sp][0] == b) EA[pjs::ROWLOC[sp][0]] += Hr * gv;
                    } else if constexpr (OWNER.b[i] == b && !PJQ_ECL) {
                        static_assert(BCOL.b[sp], "energy row: column not marked");
                        e_add(std::integral_constant<int, sp>{}, Hr * gv);
                    }
#endif
                }
            };
            double gkf = ckf, gkr = ckr;
            if constexpr (is_pre(i) && pjs::SCQ[i][S_KR] >= 0) { gkf = kf_slot; gkr = kf_slot * ekc; }
            slot(std::integral_constant<int, pjs::RI[i][RI_R0]>{}, gkf * (cr1 * cr2));
            slot(std::integral_constant<int, pjs::RI[i][RI_R1]>{}, gkf * (cr0 * cr2));
            slot(std::integral_constant<int, pjs::RI[i][RI_R2]>{}, gkf * (cr0 * cr1));
            if constexpr ((fl & F_REV) != 0) {
                slot(std::integral_constant<int, pjs::RI[i][RI_P0]>{}, -gkr * (cp1 * cp2));
                slot(std::integral_constant<int, pjs::RI[i][RI_P1]>{}, -gkr * (cp0 * cp2));
                slot(std::integral_constant<int, pjs::RI[i][RI_P2]>{}, -gkr * (cp0 * cp1));
            }
            static_for<GNR + GNP>([&](auto fc) PJR_INL {
                // one value per factor: c k nu C^(nu-1) prod_others (create_jacobian.py:400-448)
                constexpr int f = decltype(fc)::value;
                constexpr int f0 = f < GNR ? 0 : GNR, f1 = f < GNR ? GNR : GNR + GNP;
                double gv = (f < GNR ? gkf : -gkr) * gen_dpow<GP + f>(gcf[f]);
                static_range<f0, f1>([&](auto hc) PJR_INL { if constexpr (decltype(hc)::value != f) gv *= gpw[decltype(hc)::value]; });
                slot(std::integral_constant<int, pjs::GEN_SP[GP + f][0]>{}, gv);
            });
            if constexpr ((fl & F_COLLIDER) != 0)
                slot(std::integral_constant<int, (pjs::RI[i][RI_COLLIDER] >= 0 ? pjs::RI[i][RI_COLLIDER] : ONE)>{}, bcol);
            if constexpr ((fl & F_EFFTYPE) != 0) {
                static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJR_INL {
                    constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                    constexpr int es = pjs::EFF_SP[e][0];
                    // the last species' enhanced efficiency is already in gN (RD_ANM1)
                    if constexpr (es != LAST) slot(std::integral_constant<int, es>{}, VE_(e) * bM);
                });
            }
#undef VE_
#if PJQ_VCT_ON
#undef VC_
#endif
            static_for<ncnt>([&](auto qc) PJR_INL {
                constexpr int q = np0 + decltype(qc)::value;
                constexpr int k = pjs::NET_SP[q][0];
                if constexpr (pjs::ROW_BLK[k][0] == b) {
                    constexpr int r = pjs::ROWLOC[k][0];
                    constexpr double nu = pjs::NET_NU[q][0];
                    om[r] += nu * q_;
                    P[r] += nu * rp;
                    if constexpr (has_gn(i)) QN[r] += nu * gN;        // (Q_k = P_k + QN_k: see near_last())
                    JT[r] += nu * theta;
                    // reference quirk (create_jacobian.py:2786-2818): J_nplusone is assigned, not
                    // accumulated -- the last species keeps the d/dT term of one reaction only
                    if constexpr (k == LAST && i == pjs::LASTQ) JTQ = nu * theta;
                }
            });
            if constexpr (DEFER && b > LO_ && v < nv - npre) {
                // the previous block's rows: an even share of their store instructions behind every Arrhenius visit
                // (the hand-over visits at the block's end see no new store in front of their ring loads)
                constexpr int pb = b > LO_ ? b - 1 : b;      // (b - 1; the first block has none)
                constexpr int pnrows = pjs::BLK_ROW_PTR[pb + 1][0] - pjs::BLK_ROW_PTR[pb][0];
                constexpr int NP_ = pnrows * PIECES, na = nv - npre;
                constexpr int e0 = (int)((long)NP_ * v / na), e1 = (int)((long)NP_ * (v + 1) / na);
                static_range<e0, e1>([&](auto ec) PJR_INL {
                    constexpr int e = decltype(ec)::value, r = e / PIECES, h = e % PIECES;
                    row_piece(std::integral_constant<int, pb>{}, std::integral_constant<int, r>{}, std::integral_constant<int, h>{},
                              pWP[r], pWQN[r], pWCN[r], pJT[r], pS);
                });
                PJQ_SCHED_BARRIER();
            }
#if PJQ_SB_EVERY
            if constexpr ((v + 1) % PJQ_SB_EVERY == 0) PJQ_SCHED_BARRIER();
#endif
            if constexpr (v + 1 == nv - npre) PJQ_TICK(1)
            if constexpr (v + 1 == nv && npre > 0) PJQ_TICK(2)
        });
        PJQ_SCHED_BARRIER();

        // rows of this block: NASA properties of its species, outputs, energy-row partials.
        // Row k of the species block is (W_k / W_j)(P_k - w_j Q_k + S_kj) with w_j = W_j / W_N, i.e.
        //   J(k, j) = (1 / W_j) (W_k (P_k + S_kj)) - W_k Q_k / W_N:
        // one literal per column (1 / W_j) instead of two, and where S_kj is structurally zero the entry
        // is one fused multiply-add on row constants.  The energy row Sum_k hW_k (P_k - w_j Q_k + S_kj)
        // travels as the scalars HP = Sum hW_k P_k, HQN = Sum hW_k Q_k and E_j = Sum_k hW_k S_kj
        // (structural non-zeros only); the last kernel puts them together.
        double hW[nrows], WP[nrows], WQN[nrows], WCN[nrows];
        static_for<nrows>([&](auto rc) PJR_INL {
            constexpr int r = decltype(rc)::value;
            constexpr int k = pjs::BLK_ROWS[r0 + r][0];
            const bool lo = T <= pjs::SP[k][2];
            double a[6];
            static_for<6>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value;
                a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
            });
#if PJQ_KCF
            // h_k W_k = R T (t_k + 1): t_k sits in the factor columns
            hW[r] = (RU_ * T) * (xtb[k / XGRP][(k % XGRP) * PJQ_BLOCK].y + 1.0);
#else
            hW[r] = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                           T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
#endif
            const double cpk = (RU_ * pjs::SP[k][0]) * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
            H += hW[r] * om[r];
            SCP += om[r] * pjs::SP[k][1] * cpk;
            if constexpr (k == LAST) SJT += hW[r] * (sum_last_ ? JT[r] : JTQ);
            else SJT += hW[r] * JT[r];
            HP += hW[r] * P[r];
            HQN += hW[r] * QN[r];
            WP[r] = pjs::SP[k][1] * P[r];
            WQN[r] = (pjs::SP[k][1] * pjs::SP[LAST][0]) * (P[r] + QN[r]);
            WCN[r] = ANY_NEAR ? (pjs::SP[k][1] * pjs::SP[LAST][0]) * QN[r] : 0.0;
        });
        // Jacobian column c of a row: c = 0 is the d/dT column, c = j + 1 belongs to species j
        auto col_val = [&](auto rc, auto cc) PJR_INL {
            constexpr int r = decltype(rc)::value, c = decltype(cc)::value;
            constexpr int k = pjs::BLK_ROWS[r0 + r][0];
            if constexpr (c == 0) {
                return pjs::SP[k][1] * JT[r];                  // create_jacobian.py:2786-2818
            } else {
                constexpr int j = c - 1;
                constexpr int si = pjs::SLOC[k][j];
                if constexpr (si >= 0) {
                    return INVW(j) * (WP[r] + pjs::SP[k][1] * S[si]) - WQN[r];
                } else {
                    return INVW(j) * WP[r] - WQN[r];
                }
                // (only reached for the last species' pseudo-row, whose values are discarded: row_piece holds the accurate form)
            }
        };
        static_for<nrows>([&](auto rc) PJR_INL {
            constexpr int r = decltype(rc)::value;
            constexpr int k = pjs::BLK_ROWS[r0 + r][0];
            if constexpr (k < LAST) {
#if PJQ_JV
                double sk = 0.0;
                static_for<LAST>([&](auto jc) PJR_INL {
                    constexpr int j = decltype(jc)::value;
                    constexpr int si = pjs::SLOC[k][j];
                    if constexpr (si >= 0) sk += S[si] * vv(std::integral_constant<int, j + 1>{});
                });
                wp[(k + 1) * A.w_si] = pjs::SP[k][1] * (JT[r] * vv(std::integral_constant<int, 0>{}) + sk) + WP[r] * SV1 - WQN[r] * SV0;
#else
                if constexpr (DEFER && b + 1 < HI_) {
                    // (stored during the next block's visits)
                    pWP[r] = WP[r]; pWQN[r] = WQN[r]; pWCN[r] = WCN[r]; pJT[r] = JT[r];
                } else {
                    // pair stores: two columns per store instruction -- the halves of the wavefront exchange one value each,
                    // the lower half then writes two states of column c, the upper half the same two states of column c + 1
                    // (16 bytes per lane: half the store instructions in flight for the same bytes)
                    static_for<PIECES>([&](auto hc) PJR_INL { row_piece(bc, rc, hc, WP[r], WQN[r], WCN[r], JT[r], S); });
                }
#endif
            } else {
                // the last species has no row of its own; its terms still enter the energy row
                static_for<LAST>([&](auto jc) PJR_INL { (void)col_val(rc, std::integral_constant<int, decltype(jc)::value + 1>{}); });
            }
        });
        if constexpr (DEFER && b + 1 < HI_)
            static_for<pjs::BLK_NNZ[b][0]>([&](auto ec) PJR_INL { pS[decltype(ec)::value] = S[decltype(ec)::value]; });
        // the block's finished columns of the energy row: to the long-lived sums (one lane group), to the LDS
        // array the epilogue reads (one kernel), or to the column's slot of the hand-over array (several kernels)
#ifndef PJQ_NO_E
        static_for<nrows>([&](auto rc) PJR_INL {
            constexpr int r = decltype(rc)::value;
            constexpr int k = pjs::BLK_ROWS[r0 + r][0];
            if constexpr (k < LAST) {
                if constexpr (G_ == 1) e_add(std::integral_constant<int, k>{}, EA[r]);
#if PJQ_JV
                else if constexpr (JV_LDS) WE += EA[r] * vv(std::integral_constant<int, k + 1>{});       // (v_{k+1} / W_k)
#endif
                else if constexpr (EJ_LDS && PJQ_ECL && BCOL.b[k < LAST ? k : 0]) SM[SM_EJ + k * PJQ_BLOCK + tid] += EA[r];
                else if constexpr (EJ_LDS) SM[SM_EJ + k * PJQ_BLOCK + tid] = EA[r];
                else PJQ_STORE(&scr[(long)(E_COL0 + k) * PJQ_TILE], EA[r]);
            }
        });
#endif
        PJQ_SCHED_BARRIER();
        PJQ_TICK(3)
    });
    };      // run_blocks
    // one of G_ mutually exclusive branches (an if / else-if chain: ONE join behind them -- as G_ independent ifs every
    // sum that a group carries would be merged G_ times)
    group_dispatch<0, G_>(grp, [&](auto gc) PJR_INL {
        constexpr int g = decltype(gc)::value;
        run_blocks(std::integral_constant<int, group_first_block(g)>{}, std::integral_constant<int, group_first_block(g + 1)>{});
    });

    // ---- energy row: partial sums travel from kernel to kernel through hand-over slots (stored here,
    //      loaded in the next kernel's prologue); the last kernel turns them into d(dT/dt)/d. ----
    if constexpr (!LASTK_) {
        scr[hset + (long)SUM_OUT * PJQ_TILE] = H;
        scr[hset + (long)(SUM_OUT + 1) * PJQ_TILE] = SCP;
        scr[hset + (long)(SUM_OUT + 2) * PJQ_TILE] = SJT;
        scr[hset + (long)(SUM_OUT + 3) * PJQ_TILE] = HP;
        scr[hset + (long)(SUM_OUT + 4) * PJQ_TILE] = HQN;
        static_for<LAST>([&](auto jc) PJR_INL {
            constexpr int j = decltype(jc)::value;
#ifndef PJQ_NO_E
            if constexpr (e_live_col(j)) {
                if constexpr (ELM.slot[j] >= 0) scr[hset + (long)(SUM_OUT + 5 + j) * PJQ_TILE] = el[ELM.slot[j] * PJQ_BLOCK];
                else scr[hset + (long)(SUM_OUT + 5 + j) * PJQ_TILE] = E[j];
            }
#endif
        });
    } else {
        // rate_subs.py:2171-2335 / create_jacobian.py:2940-3120: mass-fraction weighted c_p sums
        // from the concentrations, Y_k c_p,k = C_k R (a0 + ...) / rho
        double cpN = 0.0;
        // (Tc: the temperature, or an opaque copy of it -- the range-selected coefficients of a species are wanted twice,
        // in the c_p sums and in the species' column of the energy row; as common subexpressions they are kept from the
        // first use to the second, 5 NSP doubles: 4.3 KB of scratch memory per lane in the 111-species kernel)
        auto cp_of = [&](auto kc, const double Tc, double& cpm, double& dcpm) PJR_INL {
            constexpr int k = decltype(kc)::value;
            const bool lo = Tc <= pjs::SP[k][2];
            double a[5];
            static_for<5>([&](auto cc) PJR_INL {
                constexpr int c = decltype(cc)::value;
                a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
            });
            cpm = a[0] + Tc * (a[1] + Tc * (a[2] + Tc * (a[3] + a[4] * Tc)));
            dcpm = a[1] + Tc * (2.0 * a[2] + Tc * (3.0 * a[3] + 4.0 * a[4] * Tc));
        };
        double Te = T;
#if PJQ_KCF
        {
            double cpm, dcpm;
            cp_of(std::integral_constant<int, LAST>{}, T, cpm, dcpm);
            cpN = (RU_ * pjs::SP[LAST][0]) * cpm;
        }
#else
        static_for<NSP>([&](auto kc) PJR_INL {
            constexpr int k = decltype(kc)::value;
            double cpm, dcpm;
            cp_of(kc, T, cpm, dcpm);
            const double Ck = conc(kc);
            cpa += Ck * cpm;
            dcpa += Ck * dcpm;
            if constexpr (k == LAST) cpN = (RU_ * pjs::SP[k][0]) * cpm;
        });
#ifndef PJR_HOST_EMU
        // (both sums finished HERE: left alone, the d/dT sum is evaluated behind the barriers below, from range-selected
        // coefficients that are kept in scratch memory until then)
        asm volatile("" : "+v"(cpa), "+v"(dcpa));
#endif
#endif
        const double cpavg = cpa * (RU_ * invrho), dcpavg = dcpa * (RU_ * invrho);
        const double icp = 1.0 / cpavg;
        PJQ_TICK(5)
        if constexpr (G_ > 1) {
            // Every group hands the sums of the columns it does not own to their owners through the columns (nobody
            // reads them any more) and the scalar sums to everybody; then each finishes its share of the energy row.
            double (*const EX)[G_ > 1 ? G_ - 1 : 1][PJQ_BLOCK] = (double (*)[G_ > 1 ? G_ - 1 : 1][PJQ_BLOCK])(SM + SM_EX);
            double (*const RED)[G_][PJQ_BLOCK] = (double (*)[G_][PJQ_BLOCK])(SM + SM_RED);
            // (column sums this kernel's other lane groups left in the hand-over array: a WORKGROUP-scope release -- the
            // groups sit on one CU and share its L1.  __threadfence() is a device-scope fence: on this chip an L2 write-back,
            // with the L2 full of Jacobian lines -- the last 111-species kernel spent 116 k of its 400 k cycles per
            // wavefront in this epilogue, profiles/r05_phase_ecl.txt)
            if constexpr (!EJ_LDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();
            static_for<G_>([&](auto gc) PJR_INL {
                constexpr int g = decltype(gc)::value;
                if (grp == g)
                    static_for<LAST>([&](auto jc) PJR_INL {
                        constexpr int j = decltype(jc)::value, o = col_owner(j);
                        constexpr int xi = ex_index(j);
                        if constexpr (o != g && ELM.slot[j] < 0 && e_live_col(j)) EX[xi][g < o ? g : g - 1][tid] = E[j];
                    });
            });
            RED[0][grp][tid] = H; RED[1][grp][tid] = SCP; RED[2][grp][tid] = SJT; RED[3][grp][tid] = HP; RED[4][grp][tid] = HQN;
            __syncthreads();
            H = 0.0; SCP = 0.0; SJT = 0.0; HP = 0.0; HQN = 0.0;
            static_for<G_>([&](auto gc) PJR_INL {
                constexpr int g = decltype(gc)::value;
                H += RED[0][g][tid]; SCP += RED[1][g][tid]; SJT += RED[2][g][tid]; HP += RED[3][g][tid]; HQN += RED[4][g][tid];
            });
        }
        PJQ_TICK(6)
        // Several kernels: the finished column sums of this group's columns come out of the hand-over array -- ALL requested
        // here, in one batch (a load per column next to its use is a memory round trip per column, each behind the store
        // of the column before: 170 k of the 500 k cycles of the 111-species mechanism's last kernel)
        constexpr bool ECOL_MEM = G_ > 1 && !EJ_LDS;
        double ECOL[ECOL_MEM ? (LAST + G_ - 1) / G_ + 1 : 1];       // (indexed within the group's column range)
        if constexpr (ECOL_MEM) {
            group_dispatch<0, G_>(grp, [&](auto gc) PJR_INL {
                constexpr int g = decltype(gc)::value;
                static_range<group_first_col(g), group_first_col(g + 1)>([&](auto jc) PJR_INL {
                    constexpr int jl = decltype(jc)::value - group_first_col(g);
                    ECOL[jl] = PJQ_LOAD_NT(&scr[(long)(E_COL0 + decltype(jc)::value) * PJQ_TILE]);
                });
            });
            PJQ_SCHED_BARRIER();
        }
        if constexpr (PJQ_ECL && !ECL_PRO) { ecl_fetch(); PJQ_SCHED_BARRIER(); }
#ifdef PJQ_TIMING
        if constexpr (ECOL_MEM) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        PJQ_TICK(7)
#endif
        // the total of column j, for the lane group that owns it (called once per column, right where the value is
        // used: gathered in front of the energy row, the sums of 55 columns spill)
        auto ecol = [&](auto jc) PJR_INL {
            constexpr int j = decltype(jc)::value;
            constexpr int jloc = j - group_first_col(col_owner(j));      // (constexpr variables: see ecl_index in k_pre)
            double e = 0.0;
            if constexpr (G_ == 1) {
                if constexpr (ELM.slot[j] >= 0) e = el[ELM.slot[j] * PJQ_BLOCK]; else e = E[j];
            } else {
                double (*const EX)[G_ > 1 ? G_ - 1 : 1][PJQ_BLOCK] = (double (*)[G_ > 1 ? G_ - 1 : 1][PJQ_BLOCK])(SM + SM_EX);
                if constexpr (ELM.slot[j] >= 0) {
                    // the groups' slots of this column, in group order
                    static_for<G_>([&](auto qc) PJR_INL {
                        e += SM[SM_EL + ((long)decltype(qc)::value * NEL_ + ELM.slot[j]) * PJQ_BLOCK + tid];
                    });
                } else if constexpr (e_live_col(j)) {
                    e = E[j];
                    constexpr int xi = ex_index(j);
                    static_for<G_ - 1>([&](auto qc) PJR_INL { e += EX[xi][decltype(qc)::value][tid]; });
                }
                // + the column sum that the block of row j finished (w = J v with v in LDS: already in WE)
                if constexpr (JV_LDS) {}
                else if constexpr (EJ_LDS) e += SM[SM_EJ + j * PJQ_BLOCK + tid];
                else e += ECOL[jloc];
            }
            if constexpr (PJQ_ECL && !ECL_PRO && BCOL.b[j]) e += ECLV[jloc];
            return e;
        };
        // column j + 1 of the energy row (create_jacobian.py:2940-3120)
        auto erow = [&](auto jc) PJR_INL {
            constexpr int j = decltype(jc)::value;
            double cpm, dcpm;
            cp_of(jc, Te, cpm, dcpm);
            const double cpj = (RU_ * pjs::SP[j][0]) * cpm;
            // ((1 - w_j) HP - w_j HQN: HQ = HP + HQN, see near_last())
            return -(((1.0 - pjs::SP[j][3]) * HP - pjs::SP[j][3] * HQN) + ecol(jc)) * pjs::SP[j][0] * icp + (cpj - cpN) * H * invrho * icp * icp;
        };
#ifndef PJR_HOST_EMU
        asm volatile("" : "+v"(Te));        // (here: behind the barriers, or the columns' polynomials are evaluated in front of them and kept)
#endif
        // (rho is not carried through the kernel: one division here; rounding of 1 / (1 / rho) is below the sums')
        const double rho_e = 1.0 / invrho;
        const double e0 = -(SCP - (dcpavg * icp) * H + rho_e * SJT) / (rho_e * cpavg);
#if PJQ_JV
        double w0 = 0.0;
        static_for<G_>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            if (G_ == 1 || grp == g)
                static_range<group_first_col(g), group_first_col(g + 1)>([&](auto jc) PJR_INL {
                    // (erow carries the factor 1 / W_j itself: the plain v_{j+1} = W_j vs_{j+1})
                    w0 += erow(jc) * (pjs::SP[decltype(jc)::value][1] * vv(std::integral_constant<int, decltype(jc)::value + 1>{}));
                });
        });
        w0 -= icp * WE;         // (0 unless JV_LDS: erow's term -E_j / (W_j c_p) v_{j+1} of the finished column sums)
        if constexpr (G_ > 1) {
            double (*const RED)[G_][PJQ_BLOCK] = (double (*)[G_][PJQ_BLOCK])(SM + SM_RED);
            RED[5][grp][tid] = w0;
            __syncthreads();
            w0 = 0.0;
            static_for<G_>([&](auto gc) PJR_INL { w0 += RED[5][decltype(gc)::value][tid]; });
        }
        if (G_ == 1 || grp == 0) wp[0] = w0 + e0 * vv(std::integral_constant<int, 0>{});
#else
        if (G_ == 1 || grp == 0) PJQ_STORE(&J_(0), e0);
        static_for<G_>([&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value;
            if (G_ == 1 || grp == g)
                static_range<group_first_col(g), group_first_col(g + 1)>([&](auto jc) PJR_INL {
                    PJQ_STORE(&J_(NSP * (decltype(jc)::value + 1)), erow(jc));
                    // (the NASA selects of all the columns gathered in front of the stores would spill)
                    if constexpr (decltype(jc)::value % 4 == 3) PJQ_SCHED_BARRIER();
                });
        });
#endif
    }
#ifdef PJQ_NO_STORE
    if (pjq_sink == 1.2345e-300) J_(0) = pjq_sink;
#endif
#ifdef PJQ_TIMING
    PJQ_TICK(4)
    if ((tid & 63) == 0 && blockIdx.x < 1024)
        for (int ph = 0; ph < 8; ++ph) g_tim[ph][blockIdx.x][(tid >> 6) + grp * (PJQ_BLOCK / 64)] = tacc[ph];
#endif
#undef J_
}

void launch_part(const PjqArgs& A, void* stream)
{
    const long blocks = (A.n + PJQ_BLOCK - 1) / PJQ_BLOCK;
    hipLaunchKernelGGL(k_rblk, dim3((unsigned)blocks), dim3(NTHR), 0, (hipStream_t)stream, A);
}
#ifdef PJQ_TIMING
void read_timing(const PjqArgs& A, void*)   // A.scr: host buffer of 5 * 1024 * 4 long long
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol((void*)A.scr, HIP_SYMBOL(g_tim), sizeof(g_tim));
}
struct Reg { Reg() { pjq_register(PJQ_ID, PJQ_JV ? 6 : PJQ_PAIR ? 2 : 4, launch_part); pjq_register(PJQ_ID, 5, read_timing); } } reg_;
#else
struct Reg { Reg() { pjq_register(PJQ_ID, PJQ_JV ? 6 : PJQ_PAIR ? 2 : 4, launch_part); } } reg_;   // 2: pair stores, 4: general, 6: w = J v
#endif
#endif  // PJQ_PART == 2

#if PJQ_PART == 4
// ------------------------------------------------------------------------------------------
// k_fin: the energy row (row 0 of the Jacobian) from what the row kernels and k_pre left in the hand-over array
// ------------------------------------------------------------------------------------------
// In the last row kernel's epilogue the loads of the column sums sit behind that kernel's last Jacobian stores (one
// in-order vmcnt queue), its workgroups keep their CU until the store queue has drained, and the lane groups wait for
// the slowest: 95 .. 110 k of the 280 k cycles of the last 111-species kernel.  As a kernel of its own -- one state per
// lane, 2.4 KB read and 888 bytes written per 111-species state -- it costs a fraction of that, and every row kernel ends
// right behind its last store.  create_jacobian.py:2940-3120 (energy row), 1853-1905 (jac[0]).
#ifndef PJQ_NKER        // row kernels of the library: from the kernel plan unless given (tests)
#define PJQ_NKER pjs::NKER
#endif
constexpr int G_ = PJQ_HALVES;
constexpr int NKER_ = PJQ_NKER;
constexpr int SUM_FIN = pjs::NSCQ + (NKER_ % 2) * NSUM;      // the slot set the last row kernel wrote
#ifdef PJR_HOST_EMU
constexpr int FINB = 1;         // (the emulation runs a one-thread workgroup inline)
#else
constexpr int FINB = 256;
#endif
__global__ void __launch_bounds__(FINB) k_fin(PjqArgs A)
{
    long s = (long)blockIdx.x * FINB + (long)threadIdx.x;
    const bool valid = s < A.n;
    if (!valid) s = A.n - 1;
    const double* const scr = scr_of(A, s);
    double H = 0.0, SCP = 0.0, SJT = 0.0, HP = 0.0, HQN = 0.0;
    static_for<G_>([&](auto gc) PJR_INL {
        const long hset = (long)decltype(gc)::value * (2 * NSUM) * PJQ_TILE;
        H += scr[hset + (long)SUM_FIN * PJQ_TILE];
        SCP += scr[hset + (long)(SUM_FIN + 1) * PJQ_TILE];
        SJT += scr[hset + (long)(SUM_FIN + 2) * PJQ_TILE];
        HP += scr[hset + (long)(SUM_FIN + 3) * PJQ_TILE];
        HQN += scr[hset + (long)(SUM_FIN + 4) * PJQ_TILE];
    });
    const double* const y = A.y + s * A.y_ss;
    const double T = y[0], p = A.pres[s];
    // (Tc: the temperature, or an opaque copy of it -- the range-selected coefficients of a species are wanted twice, in the
    // c_p sums and in the species' column; as common subexpressions they are kept from the first use to the second,
    // 5 NSP doubles: 4.3 KB of scratch memory per lane)
    auto cp_of = [&](auto kc, const double Tc, double& cpm, double& dcpm) PJR_INL {
        constexpr int k = decltype(kc)::value;
        const bool lo = Tc <= pjs::SP[k][2];
        double a[5];
        static_for<5>([&](auto cc) PJR_INL {
            constexpr int c = decltype(cc)::value;
            a[c] = lo ? pjs::SP[k][4 + c] : pjs::SP[k][11 + c];
        });
        cpm = a[0] + Tc * (a[1] + Tc * (a[2] + Tc * (a[3] + a[4] * Tc)));
        dcpm = a[1] + Tc * (2.0 * a[2] + Tc * (3.0 * a[3] + 4.0 * a[4] * Tc));
    };
    // eval_conc (rate_subs.py:1625-1710) and the mass-fraction weighted c_p sums (rate_subs.py:2171-2335,
    // create_jacobian.py:2940-3120) in one pass over the mass fractions -- sum_k C_k c_p,k = rho sum_k (Y_k / W_k) c_p,k:
    // nothing per species is kept
    double sumY = 0.0, sumYW = 0.0, cps = 0.0, dcps = 0.0;
    static_for<LAST>([&](auto kc) PJR_INL {
        constexpr int k = decltype(kc)::value;
        const double yk = y[(k + 1) * A.y_si];
        double cpm, dcpm;
        cp_of(kc, T, cpm, dcpm);
        const double yw = yk * pjs::SP[k][0];
        sumY += yk;
        sumYW += yw;
        cps += yw * cpm;
        dcps += yw * dcpm;
        if constexpr (k % 8 == 7) PJQ_SCHED_BARRIER();
    });
    double cpN;
    {
        double cpm, dcpm;
        cp_of(std::integral_constant<int, LAST>{}, T, cpm, dcpm);
        const double yw = (1.0 - sumY) * pjs::SP[LAST][0];
        sumYW += yw;
        cps += yw * cpm;
        dcps += yw * dcpm;
        cpN = (RU_ * pjs::SP[LAST][0]) * cpm;
    }
    const double Wbar = 1.0 / sumYW;
    const double rho = p * Wbar / (RU_ * T);
    const double invrho = 1.0 / rho;
    const double cpa = rho * cps, dcpa = rho * dcps;
    const double cpavg = cpa * (RU_ * invrho), dcpavg = dcpa * (RU_ * invrho);
    const double icp = 1.0 / cpavg;
    const double rho_e = 1.0 / invrho;
    const double e0 = -(SCP - (dcpavg * icp) * H + rho_e * SJT) / (rho_e * cpavg);
    PJQ_SCHED_BARRIER();        // (the concentrations are dead from here on)
    double Te = T;
#ifndef PJR_HOST_EMU
    asm volatile("" : "+v"(Te));
#endif
    auto erow = [&](auto jc) PJR_INL {
        constexpr int j = decltype(jc)::value;
        double e = scr[(long)(E_COL0 + j) * PJQ_TILE];
        if constexpr (BCOL.b[j]) {
            constexpr int ci = ecl_index(j);
            e += scr[(long)(ECL0 + ci) * PJQ_TILE];
        }
        double cpm, dcpm;
        cp_of(jc, Te, cpm, dcpm);
        const double cpj = (RU_ * pjs::SP[j][0]) * cpm;
        return -(((1.0 - pjs::SP[j][3]) * HP - pjs::SP[j][3] * HQN) + e) * pjs::SP[j][0] * icp + (cpj - cpN) * H * invrho * icp * icp;
    };
#if PJQ_JV
    const double* vp = A.v + s * A.v_ss;
    double w0 = e0 * vp[0];
    static_for<LAST>([&](auto jc) PJR_INL {
        w0 += erow(jc) * vp[(decltype(jc)::value + 1) * A.v_si];
        if constexpr (decltype(jc)::value % 8 == 7) PJQ_SCHED_BARRIER();
    });
    if (valid) (A.w + s * A.w_ss)[0] = w0;
#else
    double* const J = A.jac + s * A.j_ss;
    if (valid) PJQ_STORE(&J[0], e0);
    static_for<LAST>([&](auto jc) PJR_INL {
        const double v = erow(jc);
        if (valid) PJQ_STORE(&J[(long)(NSP * (decltype(jc)::value + 1)) * A.j_si], v);
        // (eight columns' loads in flight at a time: left alone the scheduler requests all 2 (NSP - 1) sums up front and
        // keeps them, next to the NSP concentrations, in 4 KB of scratch memory per lane)
        if constexpr (decltype(jc)::value % 8 == 7) PJQ_SCHED_BARRIER();
    });
#endif
}
void launch_fin(const PjqArgs& A, void* stream)
{
    hipLaunchKernelGGL(k_fin, dim3((unsigned)((A.n + FINB - 1) / FINB)), dim3(FINB), 0, (hipStream_t)stream, A);
}
// behind the row kernels of its kind (2: pair stores, 4: general -- the same kernel --, 6: w = J v)
#if PJQ_JV
struct Reg { Reg() { pjq_register(NKER_, 6, launch_fin); } } reg_;
#else
struct Reg { Reg() { pjq_register(NKER_, 2, launch_fin); pjq_register(NKER_, 4, launch_fin); } } reg_;
#endif
#endif  // PJQ_PART == 4

#if PJQ_PART == 3
// ------------------------------------------------------------------------------------------
// k_rate: the rate outputs of pyJac's k_dydt pass (pyjacob.cu:18-25, 153: eval_conc, eval_rxn_rates,
// get_rxn_pres_mod, eval_spec_rates, dydt) in one pass over the reactions [R0, R1)
// ------------------------------------------------------------------------------------------
// One state per lane, every reaction visited ONCE: k_f and exp(-ln K_c) side by side (exp_pair), the
// third-body concentration, q_f, q_r; omega_k accumulates in registers with compile-time indices and dydt is
// finished in the same kernel -- no hand-over through memory when one kernel covers the mechanism (a kernel
// per reaction range otherwise: omega_k then travels through `sr`).  Concentrations sit in registers
// (PJQ_C_LDS = 0: up to ~64 species) or in LDS columns; the K_c polynomial rows of the range in LDS (the low /
// high range select is per lane).  Falloff / PLOG / Chebyshev reactions take the shared body pj_rate_pre.inc.
// PJQ_FULL = 1: also fwd / rev / pres_mod (per reaction stores); 0: conc, spec_rates, dydt only.
#ifndef PJQ_FULL
#define PJQ_FULL 0
#endif
#define PJR_RECOMPUTE_KF 0
#define PJR_RECOMPUTE_KR 1
#define PJR_SLOT(i_, c_) (-1)
constexpr bool kf_plain(int) { return false; }
#ifndef PJQ_R0      // from the kernel plan
#define PJQ_R0 pjs::RATE_R[PJQ_ID][0]
#define PJQ_R1 pjs::RATE_R[PJQ_ID + 1][0]
#define PJQ_FIRST (PJQ_ID == 0)
#define PJQ_LAST (PJQ_ID == pjs::NRATE - 1)
#endif
constexpr int R0_ = PJQ_R0, R1_ = PJQ_R1;
constexpr bool FIRST_ = PJQ_FIRST != 0, LASTK_ = PJQ_LAST != 0;
constexpr KcMap make_kcmap()
{
    KcMap m{};
    for (int g = 0; g < NKC_ALL; ++g) m.loc[g] = -1;
    for (int i = R0_; i < R1_; ++i) kcmap_add(m, i);
    return m;
}
constexpr KcMap KCM = make_kcmap();
constexpr int NKC = KCM.n;
struct KcList { int v[NKC > 0 ? NKC : 1]; };
constexpr KcList make_list() { KcList l{}; for (int q = 0; q < NKC; ++q) l.v[q] = KCM.list[q]; return l; }
__device__ const KcList KCL = make_list();
constexpr int NTHR = PJQ_BLOCK * PJQ_HALVES;

__global__ void __launch_bounds__(NTHR) k_rate(PjqArgs A)
{
    __shared__ __attribute__((aligned(16))) double LTK[(NKC > 0 ? NKC : 1) * 16];
#if PJQ_C_LDS
    __shared__ double CL[NSP][PJQ_BLOCK];
#endif
    // two halves (PJQ_HALVES == 2, see k_rblk): both on the same PJQ_BLOCK states, every other reaction each
    const int half = PJQ_HALVES == 2 ? (int)(threadIdx.x >= PJQ_BLOCK) : 0;
    const int tid = (int)threadIdx.x - half * PJQ_BLOCK;
    // lanes past the end repeat the last state (same values to the same addresses): no divergence
    long s = (long)blockIdx.x * PJQ_BLOCK + tid;
    if (s >= A.n) s = A.n - 1;
    double om[NSP];
    double T, p, rho, invrho, Wbar, mconc;
    PJQ_CONST_BASES()
#if !PJQ_C_LDS
    State L;
#endif
    {
        constexpr int NQ = (NKC * 8 + NTHR - 1) / NTHR;
        d2 lt[NQ > 0 ? NQ : 1];
        kc_issue<NQ, NTHR>(KCL.v, NKC, lt);
#if PJQ_C_LDS
        State L;
#endif
        // omega_k of the reactions before R0: requested with the state, before anything waits
        if constexpr (!FIRST_) {
            if (PJQ_HALVES == 1 || half == 0)
                static_for<NSP>([&](auto kc) PJR_INL { om[decltype(kc)::value] = A.sr[decltype(kc)::value * A.sr_ld + s]; });
        }
        load_state(A, s, L);
        kc_land<NQ, NTHR>(LTK, NKC, lt);
        to_conc(L);
        T = L.T; p = L.p; rho = L.rho; invrho = L.invrho; Wbar = L.Wbar; mconc = L.mconc;
#if PJQ_C_LDS
        if (PJQ_HALVES == 1 || half == 0)
            static_for<NSP>([&](auto kc) PJR_INL { CL[decltype(kc)::value][tid] = L.C[decltype(kc)::value]; });
#endif
    }
    if (FIRST_ || (PJQ_HALVES == 2 && half == 1))
        static_for<NSP>([&](auto kc) PJR_INL { om[decltype(kc)::value] = 0.0; });
    __syncthreads();
#if PJQ_C_LDS
#define CC(idx) ((idx) == ONE ? 1.0 : CL[(idx) == ONE ? 0 : (idx)][tid])
#else
#define CC(idx) L.C[idx]
#endif
    const double logT = log(T), invT = 1.0 / T, logp = log(p);
    const double T2 = T * T, T3 = T2 * T, T4 = T2 * T2;
    // output addresses: wavefront-uniform 64-bit base (row offset + the wavefront's first state: scalar
    // arithmetic) + a 32-bit per-lane byte offset
#ifdef PJR_HOST_EMU
    const long s_wave = s;
#else
    const long s_wave = ((long)__builtin_amdgcn_readfirstlane((int)((unsigned long)s >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s);
#endif
    const unsigned lvo = (unsigned)(s - s_wave) * 8u;
#define OUT_(base, row) (*(double*)((char*)((base) + (long)(row) * A.o_ld + s_wave) + lvo))
#define OUTL_(base, row, ld) (*(double*)((char*)((base) + (long)(row) * (ld) + s_wave) + lvo))
    if constexpr (FIRST_) {
        // eval_conc (rate_subs.py:1625-1710)
        if (A.conc && (PJQ_HALVES == 1 || half == 0))
            static_for<NSP>([&](auto kc) PJR_INL { PJQ_STORE(&OUT_(A.conc, decltype(kc)::value), CC(decltype(kc)::value)); });
    }
    constexpr bool RATES_OUT = true;
    auto rate_out = [&](auto ic, const double Rf, const double Rr, const double c) PJR_INL {
        constexpr int i = decltype(ic)::value;
        (void)Rf; (void)Rr; (void)c;
#if PJQ_FULL
        // rate_subs.py:634-658, 811-840, 1076-1283: indices are positions in the mechanism file
        // (no branches around the stores: ~700 of them would cut the kernel into as many scheduling regions)
        PJQ_STORE(&OUTL_(A.fwd, pjs::RI[i][RI_ORIG], A.fwd_ld), Rf);
        if constexpr (pjs::RI[i][RI_REV_IDX] >= 0) PJQ_STORE(&OUTL_(A.rev, pjs::RI[i][RI_REV_IDX], A.rev_ld), Rr);
        if constexpr (pjs::RI[i][RI_PRES_IDX] >= 0) PJQ_STORE(&OUTL_(A.pres_mod, pjs::RI[i][RI_PRES_IDX], A.pm_ld), c);
#endif
    };
    double (&jt)[NSP] = om;         // pj_rate_pre.inc accumulates omega_k into jt[] when RATES_OUT
    double jtq = 0.0;
    double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
#define SCR_ST(slot, val) ((void)0)
    auto run_rx = [&](auto hc) PJR_INL {
    constexpr int HALF_ = decltype(hc)::value;
    static_range<R0_, R1_>([&](auto ic) PJR_INL {
        constexpr int i = decltype(ic)::value;
        if constexpr (PJQ_HALVES == 1 || (i - R0_) % 2 == HALF_) {
        if constexpr (is_pre(i)) {
#define PJR_RD(i_) pjs::RD[i_]
#define PJR_KCROW(g_) (LTK + KCM.loc[g_] * 16)
#define PJR_EFL(e_) pjs::EFF_AM1[e_][0]
#define PJR_KC_FIRST(i_) true
#define PJR_SCHED_BARRIER() PJQ_SCHED_BARRIER()
#include "pj_rate_pre.inc"
#undef PJR_RD
#undef PJR_KCROW
#undef PJR_EFL
#undef PJR_KC_FIRST
#undef PJR_SCHED_BARRIER
        } else {
            // Arrhenius (+ optional third body): rate_subs.py:113-147, 634-658, 660-840, 1076-1134
            constexpr int fl = pjs::RI[i][RI_FLAGS];
            const double cr0 = CC(pjs::RI[i][RI_R0]), cr1 = CC(pjs::RI[i][RI_R1]), cr2 = CC(pjs::RI[i][RI_R2]);
            const double lnk = RDC(i, RD_LNA) + RDC(i, RD_B) * logT - RDC(i, RD_TA) * invT;
            double kf, Rr = 0.0;
            if constexpr ((fl & F_REV) != 0) {
                const double cp0 = CC(pjs::RI[i][RI_P0]), cp1 = CC(pjs::RI[i][RI_P1]), cp2 = CC(pjs::RI[i][RI_P2]);
                double lnKc = RDC(i, RD_LNPREF);
                static_for<pjs::RI[i][RI_KC_CNT]>([&](auto cc) PJR_INL {
                    constexpr int g = pjs::RI[i][RI_KC_PTR] + decltype(cc)::value;
                    const double* a = LTK + KCM.loc[g] * 16 + ((T <= pjs::KCG[g][0]) ? 0 : 8);
                    lnKc += a[0] + a[1] * logT + a[2] * T + a[3] * T2 + a[4] * T3 + a[5] * T4 - a[6] * invT;
                });
                double ekc_;
                exp_pair(lnk, -lnKc, kf, ekc_);
                if constexpr (pjs::RD[i][RD_SGN] < 0.0) kf = -kf;
                Rr = (kf * ekc_) * (cp0 * cp1 * cp2);
            } else {
                kf = exp_one(lnk);
                if constexpr (pjs::RD[i][RD_SGN] < 0.0) kf = -kf;
            }
            double Rf = kf * (cr0 * cr1 * cr2);
            if constexpr ((fl & F_GEN) != 0) {
                // general stoichiometry: the molecule slots are empty, the factors multiply in here
                constexpr int GP = pjs::RI[i][RI_GEN_PTR], GNR = pjs::RI[i][RI_GEN_NR], GNP = pjs::RI[i][RI_GEN_NP];
                static_for<GNR>([&](auto fc) PJR_INL { Rf *= gen_pow<GP + decltype(fc)::value>(CC(pjs::GEN_SP[GP + decltype(fc)::value][0])); });
                if constexpr ((fl & F_REV) != 0)
                    static_for<GNP>([&](auto fc) PJR_INL { Rr *= gen_pow<GP + GNR + decltype(fc)::value>(CC(pjs::GEN_SP[GP + GNR + decltype(fc)::value][0])); });
            }
            double c = 1.0;
            if constexpr ((fl & F_THD) != 0) {
                c = mconc;
                static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJR_INL {
                    constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                    c += EFC(e) * CC(pjs::EFF_SP[e][0]);
                });
            }
            rate_out(ic, Rf, Rr, c);
            const double q_ = c * (Rf - Rr);
            static_for<pjs::RI[i][RI_NET_CNT]>([&](auto qc) PJR_INL {
                constexpr int q = pjs::RI[i][RI_NET_PTR] + decltype(qc)::value;
                om[pjs::NET_SP[q][0]] += pjs::NET_NU[q][0] * q_;
            });
#if PJQ_SB_EVERY
            if constexpr ((i - R0_ + 1) % PJQ_SB_EVERY == 0) PJQ_SCHED_BARRIER();
#endif
        }
        }
    });
    };
    if constexpr (PJQ_HALVES == 2) {
        if (half == 0) run_rx(std::integral_constant<int, 0>{});
        else run_rx(std::integral_constant<int, 1>{});
    } else {
        run_rx(std::integral_constant<int, 0>{});
    }
    (void)jtq; (void)ekc; (void)tdk; (void)logp; (void)p; (void)Wbar;

    // mass-fraction weighted c_p sum from the concentrations (Y_k c_p,k = C_k R (a0 + ...) / rho): taken before
    // the halves reuse the concentration columns
    double cpa = 0.0;
    if constexpr (LASTK_) {
        static_for<NSP>([&](auto kc) PJR_INL {
            constexpr int k = decltype(kc)::value;
            const bool lo = T <= pjs::SP[k][2];
            double a[5];
            static_for<5>([&](auto cc) PJR_INL { a[decltype(cc)::value] = lo ? pjs::SP[k][4 + decltype(cc)::value] : pjs::SP[k][11 + decltype(cc)::value]; });
            cpa += CC(k) * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
        });
    }
#if PJQ_HALVES == 2
    // half 1 hands its share of omega_k to half 0 through the concentration columns (nobody reads them any more)
    static_assert(PJQ_C_LDS, "two lane groups share the concentration columns");
    __syncthreads();
    if (half == 1) static_for<NSP>([&](auto kc) PJR_INL { CL[decltype(kc)::value][tid] = om[decltype(kc)::value]; });
    __syncthreads();
    if (half == 1) return;
    static_for<NSP>([&](auto kc) PJR_INL { om[decltype(kc)::value] += CL[decltype(kc)::value][tid]; });
#endif
    if constexpr (!LASTK_) {
        static_for<NSP>([&](auto kc) PJR_INL { A.sr[decltype(kc)::value * A.sr_ld + s] = om[decltype(kc)::value]; });
    } else {
        // eval_spec_rates output (rate_subs.py:1297-1542: the last species through dy_N = sp_rates[NSP - 1]) and
        // dydt (rate_subs.py:2171-2335): dT/dt = -sum_k h_k W_k omega_k / (rho c_p), dY_k/dt = omega_k W_k / rho
        if (A.spec_rates)
            static_for<NSP>([&](auto kc) PJR_INL { PJQ_STORE(&OUT_(A.spec_rates, decltype(kc)::value), om[decltype(kc)::value]); });
        if (A.dy) {
            const double cpavg = cpa * (RU_ * invrho);
            double Hs = 0.0;
            static_for<NSP>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                const bool lo = T <= pjs::SP[k][2];
                double a[6];
                static_for<6>([&](auto cc) PJR_INL { a[decltype(cc)::value] = lo ? pjs::SP[k][4 + decltype(cc)::value] : pjs::SP[k][11 + decltype(cc)::value]; });
                const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                         T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
                Hs += hW * om[k];
                if constexpr (k < LAST) PJQ_STORE(&OUT_(A.dy, k + 1), om[k] * pjs::SP[k][1] * invrho);
            });
            PJQ_STORE(&OUT_(A.dy, 0), -Hs / (rho * cpavg));
        }
    }
#undef OUT_
#undef OUTL_
#undef CC
#undef SCR_ST
}

void launch_rate(const PjqArgs& A, void* stream)
{
    const long blocks = (A.n + PJQ_BLOCK - 1) / PJQ_BLOCK;
    hipLaunchKernelGGL(k_rate, dim3((unsigned)blocks), dim3(NTHR), 0, (hipStream_t)stream, A);
}
struct Reg { Reg() { pjq_register(PJQ_ID, PJQ_FULL ? 8 : 7, launch_rate); } } reg_;   // 7: conc / spec_rates / dydt, 8: every rate output
#endif  // PJQ_PART == 3

#if PJQ_PART == 5
// ------------------------------------------------------------------------------------------
// k_jvd: w = J v per state as a DIRECTIONAL DERIVATIVE -- every reaction visited ONCE (pj_spec_jacvec)
// ------------------------------------------------------------------------------------------
// Every sum of the formulation is a sum over reactions of nu_ki times a per-reaction scalar:
//   omega_k = sum nu q,  JT_k = sum nu theta,  P_k = sum nu rp,  Q_k = sum nu rq,  S_kj = sum nu G_ij,
// and row k of the species block applied to v is  w_k = W_k D_k  with
//   D_k = JT_k v_0 + P_k SV1 - Q_k SV0 / W_N + sum_j S_kj vs_j = sum_i nu_ki d_i,
//   d_i = theta_i v_0 + rp_i SV1 - rq_i SV0 / W_N + sum_{slots of i} G_ij vs_j      (vs_j = v_{j+1} / W_j,
//   SV0 = sum_j v_{j+1}, SV1 = sum_j vs_j: k_rblk's scaled vector)
// -- ONE scalar d_i per reaction: the reaction's sparse derivative row times the vector, scattered to its net species
// exactly like q_i into omega_k.  The row kernels visit a reaction once per row block that holds one of its net species
// (3.6 times on average) because a Jacobian ROW needs all of its reactions in one place; the product does not.
// The energy row needs no sums of its own:  sum_k h_kW_k (P_k SV1 - Q_k SV0/W_N + sum_j S_kj vs_j) = sum_k h_kW_k (D_k - JT_k v_0),
// and with jac[0]'s term -(sum_k h_kW_k JT*_k / c_p) v_0  (JT*: the J_nplusone quirk's value for the last species)
//   w_0 = -[SCP - (dc_p/c_p) H] / (rho c_p) v_0 - [sum_k h_kW_k D_k + h_NW_N (JTQ - JT_N) v_0] / c_p
//         + (H / (rho c_p^2)) (sum_j c_p,j v_{j+1} - c_p,N SV0)
// (k_rblk's epilogue, create_jacobian.py:2940-3120).  The rates enter only through H = sum_k h_kW_k omega_k and
// SCP = sum_k omega_k W_k c_p,k, and those are sums over reactions too:  H = sum_i Hr_i q_i,  SCP = sum_i dCp_i q_i  with the
// reaction enthalpy Hr_i = sum_k nu_ki h_kW_k = R T (T dlnK_c/dT + sum nu) -- the visit has it -- and its temperature
// derivative dCp_i = sum_k nu_ki W_k c_p,k, four more multiply-adds on the K_c row the visit has read anyway (an irreversible
// reaction has no row: its net species' NASA polynomials).  So a lane carries ONE array, D_k, and four scalars.
// Geometry (pyjac_amd/specbuild.py: jvd_geometry): D_k in registers; concentrations and the scaled vector in LDS columns
// shared by four lane groups that take every fourth reaction -- 128 states per workgroup up to 64 species (512 threads, two
// wavefronts per SIMD), 64 states beyond (there with the K_c rows read from the mechanism table in global memory:
// PJQ_JVD_KC_GLOBAL) -- or, small mechanisms / tests, everything in registers and one group.  Reaction ranges as k_rate
// (RATE_R) where the K_c rows are staged in LDS; between the kernels of a library D_k and the scalars travel through `sr`.
#ifndef PJQ_V_LDS
#define PJQ_V_LDS 0         // the scaled vector in LDS columns (large mechanisms) instead of registers
#endif
#ifndef PJQ_JVD_SB
#define PJQ_JVD_SB 2        // visits between scheduling barriers
#endif
#ifndef PJQ_JVD_DYDT
#define PJQ_JVD_DYDT 0      // 1: the same kernel WITHOUT the vector -- q_i instead of d_i is scattered to the net species, the
                            // derivative parts of a visit are dead code, and the outputs are pyJac's dydt / eval_spec_rates /
                            // eval_conc arrays (k_rate's lean kernel in k_jvd's geometry: several lane groups on shared
                            // concentration columns, two wavefronts per SIMD); registered as the fast lean rate kernel when the
                            // library has ONE such kernel (nothing travels through `sr`)
#endif
constexpr bool DYDT_ = PJQ_JVD_DYDT != 0;
#ifndef PJQ_JVD_KC_GLOBAL
#define PJQ_JVD_KC_GLOBAL 0 // 1: the K_c rows are read from the mechanism table in global memory (L1 / L2 resident, a 16-byte load per
                            // lane at one of two addresses) instead of from a copy in LDS: the vector memory path idles in this
                            // kernel while the LDS is what bounds the large-mechanism geometry; and without the copy a kernel's
                            // reaction range is not limited by the LDS (one kernel for the mechanism)
#endif
#ifndef PJQ_JVD_AHEAD
#define PJQ_JVD_AHEAD (PJQ_C_LDS != 0)      // 1 / 2: the K_c rows and concentrations of a lane group's next / next but one reaction
                            // are requested while the current one is computed (k_rblk's PJQ_KC_AHEAD / PJQ_CONC_AHEAD: at one
                            // wavefront per SIMD nothing else covers the round trips at the top of every visit)
#endif
#define PJR_RECOMPUTE_KF 0
#define PJR_RECOMPUTE_KR 1
// which scalars of a falloff / PLOG / Chebyshev reaction pj_rate_pre.inc "hands over" (here: to local variables): the rule
// of pj_tables.cpp's SCQ
constexpr int jvd_slot(int fl, int c)
{
    return c == S_KR ? ((fl & F_CHEB) ? c : -1) : c == S_RP ? ((fl & (F_THD | F_PDEP)) ? c : -1) :
           c == S_BM ? ((fl & F_EFFTYPE) ? c : -1) : c == S_BC ? ((fl & F_COLLIDER) ? c : -1) : c;
}
#define PJR_SLOT(i_, c_) jvd_slot(pjs::RI[i_][RI_FLAGS], c_)
constexpr bool kf_plain(int) { return false; }
template <int i>
constexpr bool has_anm1() { return pjs::RD[i][RD_ANM1] != 0.0; }
constexpr int max_kc_cnt()
{
    int m = 1;
    for (int i = 0; i < NRXN; ++i)
        if ((pjs::RI[i][RI_FLAGS] & F_REV) && pjs::RI[i][RI_KC_CNT] > m) m = pjs::RI[i][RI_KC_CNT];
    return m;
}
constexpr int MAXKC = max_kc_cnt();
#ifndef PJQ_R0      // from the kernel plan
#define PJQ_R0 pjs::RATE_R[PJQ_ID][0]
#define PJQ_R1 pjs::RATE_R[PJQ_ID + 1][0]
#define PJQ_FIRST (PJQ_ID == 0)
#define PJQ_LAST (PJQ_ID == pjs::NRATE - 1)
#endif
constexpr int R0_ = PJQ_R0, R1_ = PJQ_R1;
constexpr bool FIRST_ = PJQ_FIRST != 0, LASTK_ = PJQ_LAST != 0;
constexpr KcMap make_kcmap()
{
    KcMap m{};
    for (int g = 0; g < NKC_ALL; ++g) m.loc[g] = -1;
    for (int i = R0_; i < R1_; ++i) kcmap_add(m, i);
    return m;
}
constexpr KcMap KCM = make_kcmap();
constexpr int NKC = PJQ_JVD_KC_GLOBAL ? 0 : KCM.n;
struct KcList { int v[NKC > 0 ? NKC : 1]; };
constexpr KcList make_list() { KcList l{}; for (int q = 0; q < NKC; ++q) l.v[q] = KCM.list[q]; return l; }
__device__ const KcList KCL = make_list();
constexpr int G_ = PJQ_HALVES;
constexpr int NTHR = PJQ_BLOCK * G_;
static_assert(G_ == 1 || PJQ_C_LDS, "k_jvd: several lane groups share the concentration columns (PJQ_C_LDS)");
static_assert(!PJQ_V_LDS || PJQ_C_LDS, "k_jvd: the vector columns sit next to the concentration columns");
constexpr long JVD_LDS = 8L * ((NKC > 0 ? NKC : 1) * 16 + (PJQ_C_LDS ? (long)NSP * PJQ_BLOCK : 0) + (PJQ_V_LDS ? (long)NSP * PJQ_BLOCK : 0) +
                               (G_ > 1 ? 4L * PJQ_BLOCK : 0));
#ifndef PJR_HOST_EMU
static_assert(JVD_LDS <= 160L * 1024, "k_jvd: columns + K_c rows exceed the LDS (specbuild.py picks the geometry)");
#endif
// rows of `sr` between the kernels of a library: D_k, then four scalars
constexpr int SR_D = 0, SR_SC = NSP;       // JT_N, JTQ, H, SCP

__global__ void __launch_bounds__(NTHR) k_jvd(PjqArgs A)
{
    __shared__ __attribute__((aligned(16))) double LTK[(NKC > 0 ? NKC : 1) * 16];
#if PJQ_C_LDS
    __shared__ double CL[NSP][PJQ_BLOCK];
#endif
#if PJQ_V_LDS && !PJQ_JVD_DYDT
    __shared__ double VL[NSP][PJQ_BLOCK];
#endif
    __shared__ double RED[G_ > 1 ? 4 : 1][PJQ_BLOCK];
    // lane groups (see k_rblk): all on the same PJQ_BLOCK states, reaction R0 + q to group q % G_
    const int grp = G_ > 1 ? (int)threadIdx.x / PJQ_BLOCK : 0;
    const int tid = (int)threadIdx.x - grp * PJQ_BLOCK;
    // lanes past the end repeat the last state (same values to the same addresses): no divergence
    long s = (long)blockIdx.x * PJQ_BLOCK + tid;
    if (s >= A.n) s = A.n - 1;
    double acc[NSP];                    // D_k
    double JTN = 0.0, JTQ = 0.0, Hs = 0.0, SCP = 0.0;
    double T, p, rho, invrho, Wbar, mconc;
    PJQ_CONST_BASES()
    // a lane's byte offset inside an LDS column (an opaque copy per pass and epilogue part: nothing read from the columns
    // is kept from one part of the kernel to the next)
    unsigned lo_ = (unsigned)tid * 8u;
#if !PJQ_C_LDS
    State L;
#endif
#if !PJQ_V_LDS && !PJQ_JVD_DYDT
    double VR[NSP];                     // vs_c: v_0, v_c / W_{c-1}
#endif
    double SV0 = 0.0, SV1 = 0.0;
    {
        constexpr int NQ = (NKC * 8 + NTHR - 1) / NTHR;
        d2 lt[NQ > 0 ? NQ : 1];
        kc_issue<NQ, NTHR>(KCL.v, NKC, lt);
#if PJQ_C_LDS
        State L;
#endif
        // the sums of the reactions before R0: requested with the state, before anything waits
        if constexpr (!FIRST_) {
            if (grp == 0) {
                static_for<NSP>([&](auto kc) PJR_INL { acc[decltype(kc)::value] = A.sr[(SR_D + decltype(kc)::value) * A.sr_ld + s]; });
                JTN = A.sr[(SR_SC + 0) * A.sr_ld + s]; JTQ = A.sr[(SR_SC + 1) * A.sr_ld + s];
                Hs = A.sr[(SR_SC + 2) * A.sr_ld + s]; SCP = A.sr[(SR_SC + 3) * A.sr_ld + s];
            }
        }
#if !PJQ_JVD_DYDT
        const double* vp = A.v + s * A.v_ss;
#endif
#if PJQ_JVD_DYDT
#elif PJQ_V_LDS
        // the groups fill the vector columns together: group g the components g, g + G_, ... (ONE branch per group with all
        // of its loads in flight -- a test per component is a load, a wait and a store per component, NSP / G_ memory round
        // trips in a row)
        group_dispatch<0, G_>(grp, [&](auto gc) PJR_INL {
            constexpr int g = decltype(gc)::value, cnt = (NSP - g + G_ - 1) / G_;
            double vt[cnt > 0 ? cnt : 1];
            static_for<cnt>([&](auto jc) PJR_INL { vt[decltype(jc)::value] = vp[(g + decltype(jc)::value * G_) * A.v_si]; });
            static_for<cnt>([&](auto jc) PJR_INL {
                constexpr int c = g + decltype(jc)::value * G_;
                if constexpr (c > 0) VL[c][tid] = vt[decltype(jc)::value] * pjs::SP[c > 0 ? c - 1 : 0][0];
                else VL[c][tid] = vt[decltype(jc)::value];
            });
        });
#else
        static_for<NSP>([&](auto cc) PJR_INL { VR[decltype(cc)::value] = vp[decltype(cc)::value * A.v_si]; });
#endif
        load_state(A, s, L);
        kc_land<NQ, NTHR>(LTK, NKC, lt);
        to_conc(L);
        T = L.T; p = L.p; rho = L.rho; invrho = L.invrho; Wbar = L.Wbar; mconc = L.mconc;
#if PJQ_C_LDS
        if (grp == 0)
            static_for<NSP>([&](auto kc) PJR_INL { CL[decltype(kc)::value][tid] = L.C[decltype(kc)::value]; });
#endif
    }
    if (FIRST_ || grp != 0) static_for<NSP>([&](auto kc) PJR_INL { acc[decltype(kc)::value] = 0.0; });
    __syncthreads();
#if PJQ_C_LDS
#define CC(idx) ((idx) == ONE ? 1.0 : *(const double*)((const char*)&CL[(idx) == ONE ? 0 : (idx)][0] + lo_))
#else
#define CC(idx) L.C[idx]
#endif
#if PJQ_JVD_DYDT
#define VS(c_) 0.0
    static_assert(FIRST_ && LASTK_, "PJQ_JVD_DYDT: one kernel for the mechanism (nothing travels through sr)");
    // output addresses as in k_rate: wavefront-uniform 64-bit base + a 32-bit per-lane byte offset
#ifdef PJR_HOST_EMU
    const long s_wave = s;
#else
    const long s_wave = ((long)__builtin_amdgcn_readfirstlane((int)((unsigned long)s >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s);
#endif
    const unsigned lvo = (unsigned)(s - s_wave) * 8u;
#define OUT_(base, row) (*(double*)((char*)((base) + (long)(row) * A.o_ld + s_wave) + lvo))
    // eval_conc (rate_subs.py:1625-1710)
    if (A.conc && grp == 0)
        static_for<NSP>([&](auto kc) PJR_INL { PJQ_STORE(&OUT_(A.conc, decltype(kc)::value), CC(decltype(kc)::value)); });
#elif PJQ_V_LDS
#define VS(c_) (*(const double*)((const char*)&VL[c_][0] + lo_))
    static_range<1, NSP>([&](auto cc) PJR_INL {
        constexpr int c = decltype(cc)::value;
        const double vsc = VL[c][tid];
        SV1 += vsc;
        SV0 += vsc * pjs::SP[c - 1][1];
    });
#else
#define VS(c_) VR[c_]
    static_range<1, NSP>([&](auto cc) PJR_INL {
        constexpr int c = decltype(cc)::value;
        SV0 += VR[c];
        VR[c] *= pjs::SP[c - 1][0];
        SV1 += VR[c];
    });
#endif
    const double v0 = VS(0);
    const double SV0N = SV0 * pjs::SP[LAST][0];
    double logT = log(T), invT = 1.0 / T;
    const double logp = log(p);
    double T2, T3, T4, T2d, T3d, T4d, Tc1, Tc2, Tc3, Tc4;
    // opaque copies of T and its functions: k_f, K_c, the NASA selects ... of one part of the kernel are not kept for the
    // next one (two passes over the reactions: 2 NRXN values; three sweeps over the species: 6 NSP range-selected
    // coefficients -- in scratch memory)
    auto fresh = [&]() PJR_INL {
#ifndef PJR_HOST_EMU
        asm volatile("" : "+v"(T), "+v"(logT), "+v"(invT), "+v"(lo_));
#endif
        T2 = T * T; T3 = T2 * T; T4 = T2 * T2;
        T2d = 2.0 * T2; T3d = 3.0 * T3; T4d = 4.0 * T4;
        Tc1 = 2.0 * T; Tc2 = 6.0 * T2; Tc3 = 12.0 * T3; Tc4 = 20.0 * T4;
    };
    fresh();
    const double WR = Wbar * invrho;
    // NASA row pair of K_c group g
    auto kcrow = [&](auto gc) PJR_INL -> const double* {
        if constexpr (PJQ_JVD_KC_GLOBAL) return pjs::LTAB + pjs::LT_KC + (long)decltype(gc)::value * 16;
        else return LTK + KCM.loc[decltype(gc)::value] * 16;
    };
    double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
    double jtd[NSP], jtq = 0.0;         // pj_rate_pre.inc's d/dT sums: dead here
    auto rate_out = [](auto, double, double, double) {};
    double hv[6];                       // pj_rate_pre.inc's hand-over values of the reaction at hand
#define SCR_ST(slot, val) (hv[slot] = (val))

    // NASA properties of species k: h_kW_k and c_p,k W_k / R (rate_subs.py:2171-2335)
    auto nasa = [&](auto kc, double& hW, double& cpm, double& dcpm) PJR_INL {
        constexpr int k = decltype(kc)::value;
        const bool lo = T <= pjs::SP[k][2];
        double a[6];
        static_for<6>([&](auto cc) PJR_INL { a[decltype(cc)::value] = lo ? pjs::SP[k][4 + decltype(cc)::value] : pjs::SP[k][11 + decltype(cc)::value]; });
        hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) + T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
        cpm = a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T)));
        dcpm = a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T));
    };
    // what an Arrhenius visit reads from LDS first: requested one visit ahead (PJQ_JVD_AHEAD), into the buffer of its parity
    constexpr int NBUF = PJQ_JVD_AHEAD + 1;
    double cab[NBUF][6], kab[NBUF][MAXKC][7];
    auto fetch = [&](auto ic) PJR_INL {
        constexpr int i = decltype(ic)::value, par = ((i - R0_) / G_) % NBUF;
        if constexpr (i < R1_) {
            if constexpr (!is_pre(i < R1_ ? i : R0_)) {
                constexpr int FL = pjs::RI[i][RI_FLAGS];
                cab[par][0] = CC(pjs::RI[i][RI_R0]); cab[par][1] = CC(pjs::RI[i][RI_R1]); cab[par][2] = CC(pjs::RI[i][RI_R2]);
                if constexpr ((FL & F_REV) != 0) {
                    cab[par][3] = CC(pjs::RI[i][RI_P0]); cab[par][4] = CC(pjs::RI[i][RI_P1]); cab[par][5] = CC(pjs::RI[i][RI_P2]);
                    static_for<pjs::RI[i][RI_KC_CNT]>([&](auto cc) PJR_INL {
                        constexpr int c = decltype(cc)::value, g = pjs::RI[i][RI_KC_PTR] + c;
                        const double* a = kcrow(std::integral_constant<int, g>{}) + ((T <= pjs::KCG[g][0]) ? 0 : 8);
                        static_for<7>([&](auto ec) PJR_INL { kab[par][c][decltype(ec)::value] = a[decltype(ec)::value]; });
                    });
                }
            }
        }
    };
    // One reaction: q_i and d_i, the reaction's derivative row times the scaled vector
    auto visit = [&](auto ic) PJR_INL {
        constexpr int i = decltype(ic)::value;
        constexpr int par = ((i - R0_) / G_) % (PJQ_JVD_AHEAD + 1);
        constexpr int FL = pjs::RI[i][RI_FLAGS];
        constexpr int np0 = pjs::RI[i][RI_NET_PTR], ncnt = pjs::RI[i][RI_NET_CNT];
        constexpr int GP = pjs::RI[i][RI_GEN_PTR];
        constexpr int GNR = (FL & F_GEN) ? pjs::RI[i][RI_GEN_NR] : 0;
        constexpr int GNP = ((FL & F_GEN) && (FL & F_REV)) ? pjs::RI[i][RI_GEN_NP] : 0;
        constexpr int KCNT = (FL & F_REV) ? pjs::RI[i][RI_KC_CNT] : 0;
        constexpr double nsum = net_sum(i);
        double q_ = 0.0, theta = 0.0, rp = 0.0, bM_ = 0.0, bcol_ = 0.0, gkf = 0.0, gkr = 0.0;
        double a0 = 1.0, a1 = 1.0, a2 = 1.0, b0 = 1.0, b1 = 1.0, b2 = 1.0;     // the molecule slots' concentrations
        double hrt = 0.0, dcr = 0.0;        // Hr_i / (R T) - sum nu and dCp_i / R - sum nu, from the K_c rows
        if constexpr (is_pre(i)) {
            // falloff / PLOG / Chebyshev: the shared body (rate_subs.py:254-2335, create_jacobian.py:2189-3298)
            constexpr bool RATES_OUT = false;
            double (&jt)[NSP] = jtd;
#define PJR_RD(i_) pjs::RD[i_]
#define PJR_KCROW(g_) kcrow(std::integral_constant<int, g_>{})
#define PJR_EFL(e_) pjs::EFF_AM1[e_][0]
#define PJR_KC_FIRST(i_) true
#define PJR_SCHED_BARRIER() ((void)0)
#include "pj_rate_pre.inc"
#undef PJR_RD
#undef PJR_KCROW
#undef PJR_EFL
#undef PJR_KC_FIRST
#undef PJR_SCHED_BARRIER
            q_ = c * R;
            theta = hv[S_TH];
            bM_ = bM; bcol_ = bcol;
            gkf = c * kf; gkr = c * kr;
            if constexpr (PJR_SLOT(i, S_RP) >= 0) rp = hv[S_RP];
            else rp = WR * ((1.0 - pjs::RD[i][RD_NR]) * (c * Rf) - ((FL & F_REV) ? (1.0 - pjs::RD[i][RD_NP]) * (c * Rr) : 0.0));
            if constexpr ((FL & F_CHEB) != 0) {         // eval_jacob's own k_f in the dR/dY_j terms (pj_rate_pre.inc)
                gkf = kfj_;
                if constexpr ((FL & F_REV) != 0) gkr = kfj_ * ekc[pjs::KC_CLASS[i][0]];
            }
            a0 = cr0; a1 = cr1; a2 = cr2; b0 = cp0; b1 = cp1; b2 = cp2;
            hrt = TdlnKc;
            if constexpr (PJQ_JVD_AHEAD) fetch(std::integral_constant<int, i + PJQ_JVD_AHEAD * G_>{});
            static_for<KCNT>([&](auto cc) PJR_INL {
                constexpr int g = pjs::RI[i][RI_KC_PTR] + decltype(cc)::value;
                const double* a = kcrow(std::integral_constant<int, g>{}) + ((T <= pjs::KCG[g][0]) ? 0 : 8);
                dcr += a[1] + a[2] * Tc1 + a[3] * Tc2 + a[4] * Tc3 + a[5] * Tc4;
            });
        } else {
            // Arrhenius (+ optional third body): k_rblk's visit (rate_subs.py:113-147, 660-840, 1076-1134;
            // create_jacobian.py:341-489)
            constexpr double nr = pjs::RD[i][RD_NR], np_ = pjs::RD[i][RD_NP];
            if constexpr (!PJQ_JVD_AHEAD) fetch(ic);
            a0 = cab[par][0]; a1 = cab[par][1]; a2 = cab[par][2];
            if constexpr ((FL & F_REV) != 0) { b0 = cab[par][3]; b1 = cab[par][4]; b2 = cab[par][5]; }
            double ka[KCNT > 0 ? KCNT : 1][7];
            static_for<KCNT>([&](auto cc) PJR_INL {
                static_for<7>([&](auto ec) PJR_INL { ka[decltype(cc)::value][decltype(ec)::value] = kab[par][decltype(cc)::value][decltype(ec)::value]; });
            });
            if constexpr (PJQ_JVD_AHEAD) fetch(std::integral_constant<int, i + PJQ_JVD_AHEAD * G_>{});
            double pr_ = a0 * a1 * a2, pp_ = b0 * b1 * b2;
            static_for<GNR + GNP>([&](auto fc) PJR_INL {
                constexpr int f = decltype(fc)::value;
                const double gp_ = gen_pow<GP + f>(CC(pjs::GEN_SP[GP + f][0]));
                if constexpr (f < GNR) pr_ *= gp_; else pp_ *= gp_;
            });
            const double lnk = RDC(i, RD_LNA) + RDC(i, RD_B) * logT - RDC(i, RD_TA) * invT;
            double kf, ekc_ = 0.0;
            if constexpr ((FL & F_REV) != 0) {
                double lnKc = RDC(i, RD_LNPREF);
                static_for<KCNT>([&](auto cc) PJR_INL {
                    const double* a = ka[decltype(cc)::value];
                    lnKc += a[0] + a[1] * logT + a[2] * T + a[3] * T2 + a[4] * T3 + a[5] * T4 - a[6] * invT;
                    hrt += a[1] + a[2] * T + a[3] * T2d + a[4] * T3d + a[5] * T4d + a[6] * invT;
                    dcr += a[1] + a[2] * Tc1 + a[3] * Tc2 + a[4] * Tc3 + a[5] * Tc4;
                });
                exp_pair(lnk, -lnKc, kf, ekc_);
            } else {
                kf = exp_one(lnk);
            }
            if constexpr (pjs::RD[i][RD_SGN] < 0.0) kf = -kf;
            const double Rf = kf * pr_;
            double Rr = 0.0;
            if constexpr ((FL & F_REV) != 0) Rr = (kf * ekc_) * pp_;
            const double R = Rf - Rr;
            double c = 1.0, lead = 0.0;
            if constexpr ((FL & F_THD) != 0) {
                double Mc = mconc;
                static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJR_INL {
                    constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                    Mc += EFC(e) * CC(pjs::EFF_SP[e][0]);
                });
                c = Mc;
                lead = -c * R * invT;
                if constexpr ((FL & F_EFFTYPE) != 0) bM_ = R;
            }
            if constexpr ((FL & F_NO_DT) == 0) {
                const double dlnk = RDC(i, RD_B) + RDC(i, RD_TA) * invT;
                double el = R * dlnk + Rf * (1.0 - nr);
                if constexpr ((FL & F_REV) != 0) el -= Rr * ((1.0 - np_) - hrt);
                theta = (lead + c * invT * el) * invrho;
            }
            gkf = c * kf;
            if constexpr ((FL & F_REV) != 0) gkr = gkf * ekc_;
            if constexpr ((FL & F_THD) != 0) {
                double a = c * (nr * Rf - ((FL & F_REV) ? np_ * Rr : 0.0));
                if constexpr ((FL & F_EFFTYPE) != 0) a += c * R;
                rp = WR * (c * R - a) + bM_;
            } else {
                rp = WR * ((1.0 - nr) * Rf - ((FL & F_REV) ? (1.0 - np_) * Rr : 0.0));
            }
            q_ = gkf * pr_ - gkr * pp_;
        }
        // H and SCP: the reaction's enthalpy and its temperature derivative times q_i
        if constexpr (DYDT_) {
            // (the dydt build takes H from the species, as k_rate does)
        } else if constexpr ((FL & F_REV) != 0) {
            Hs += ((RU_ * T) * (hrt + nsum)) * q_;
            SCP += (RU_ * (dcr + nsum)) * q_;
        } else {
            // an irreversible reaction has no K_c row: its net species' NASA polynomials
            double Hr = 0.0, dC = 0.0;
            static_for<ncnt>([&](auto qc) PJR_INL {
                constexpr int k = pjs::NET_SP[np0 + decltype(qc)::value][0];
                double hW, cpm, dcpm;
                nasa(std::integral_constant<int, k>{}, hW, cpm, dcpm);
                Hr += pjs::NET_NU[np0 + decltype(qc)::value][0] * hW;
                dC += pjs::NET_NU[np0 + decltype(qc)::value][0] * (RU_ * cpm);
            });
            Hs += Hr * q_;
            SCP += dC * q_;
        }
        double gN = 0.0, dot = 0.0;
        if constexpr (has_anm1<i>()) gN = bM_ * RDC(i, RD_ANM1);
        auto slot = [&](auto spc, const double gv) PJR_INL {
            constexpr int sp = decltype(spc)::value;
            if constexpr (sp == LAST) gN += gv;
            else if constexpr (sp != ONE) dot += gv * VS(sp + 1);
        };
        slot(std::integral_constant<int, pjs::RI[i][RI_R0]>{}, gkf * (a1 * a2));
        slot(std::integral_constant<int, pjs::RI[i][RI_R1]>{}, gkf * (a0 * a2));
        slot(std::integral_constant<int, pjs::RI[i][RI_R2]>{}, gkf * (a0 * a1));
        if constexpr ((FL & F_REV) != 0) {
            slot(std::integral_constant<int, pjs::RI[i][RI_P0]>{}, -gkr * (b1 * b2));
            slot(std::integral_constant<int, pjs::RI[i][RI_P1]>{}, -gkr * (b0 * b2));
            slot(std::integral_constant<int, pjs::RI[i][RI_P2]>{}, -gkr * (b0 * b1));
        }
        if constexpr (GNR + GNP > 0) {
            // one value per factor: c k nu C^(nu-1) prod_others (create_jacobian.py:400-448)
            double gcf[GNR + GNP > 0 ? GNR + GNP : 1], gpw[GNR + GNP > 0 ? GNR + GNP : 1];
            static_for<GNR + GNP>([&](auto fc) PJR_INL {
                constexpr int f = decltype(fc)::value;
                gcf[f] = CC(pjs::GEN_SP[GP + f][0]);
                gpw[f] = gen_pow<GP + f>(gcf[f]);
            });
            static_for<GNR + GNP>([&](auto fc) PJR_INL {
                constexpr int f = decltype(fc)::value;
                constexpr int f0 = f < GNR ? 0 : GNR, f1 = f < GNR ? GNR : GNR + GNP;
                double gv = (f < GNR ? gkf : -gkr) * gen_dpow<GP + f>(gcf[f]);
                static_range<f0, f1>([&](auto hc) PJR_INL { if constexpr (decltype(hc)::value != f) gv *= gpw[decltype(hc)::value]; });
                slot(std::integral_constant<int, pjs::GEN_SP[GP + f][0]>{}, gv);
            });
        }
        if constexpr ((FL & F_COLLIDER) != 0)
            slot(std::integral_constant<int, (pjs::RI[i][RI_COLLIDER] >= 0 ? pjs::RI[i][RI_COLLIDER] : ONE)>{}, bcol_);
        if constexpr ((FL & F_EFFTYPE) != 0) {
            static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJR_INL {
                constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                constexpr int es = pjs::EFF_SP[e][0];
                // the last species' enhanced efficiency is already in gN (RD_ANM1)
                if constexpr (es != LAST) slot(std::integral_constant<int, es>{}, EFC(e) * bM_);
            });
        }
        const double rq = rp + gN;
        const double del = DYDT_ ? q_ : theta * v0 + rp * SV1 - rq * SV0N + dot;
        static_for<ncnt>([&](auto qc) PJR_INL {
            constexpr int q = np0 + decltype(qc)::value;
            constexpr int k = pjs::NET_SP[q][0];
            constexpr double nu = pjs::NET_NU[q][0];
            acc[k] += nu * del;
            // reference quirk (create_jacobian.py:2786-2818): J_nplusone is assigned, not accumulated
            if constexpr (k == LAST) JTN += nu * theta;
            if constexpr (k == LAST && i == pjs::LASTQ) JTQ = nu * theta;
        });
        // (without a barrier nothing orders the reactions and the scheduler interleaves -- and keeps live -- far more of
        // them than the register file holds; a pair at a time fills the fp64 pipeline's dependent-issue bubbles)
        if constexpr (((i - R0_) / G_ + 1) % PJQ_JVD_SB == 0) PJQ_SCHED_BARRIER();
    };
    group_dispatch<0, G_>(grp, [&](auto gc) PJR_INL {
        static_for<PJQ_JVD_AHEAD>([&](auto ac) PJR_INL { fetch(std::integral_constant<int, R0_ + decltype(gc)::value + decltype(ac)::value * G_>{}); });
        static_range<R0_, R1_>([&](auto ic) PJR_INL {
            if constexpr ((decltype(ic)::value - R0_) % G_ == decltype(gc)::value) visit(ic);
        });
    });
    (void)jtq; (void)logp; (void)p; (void)hv; (void)ekc; (void)tdk;

    // mass-fraction weighted c_p sums from the concentrations (Y_k c_p,k = C_k R (a0 + ...) / rho) and the vector's share
    // sum_j c_p,j v_{j+1}: taken before the groups reuse the concentration columns
    double cpa = 0.0, dcpa = 0.0, cpN = 0.0, SCV = 0.0;
    PJQ_SCHED_BARRIER();
    fresh();
    if constexpr (LASTK_) {
        if (grp == 0)
            static_for<NSP>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                double hW, cpm, dcpm;
                nasa(kc, hW, cpm, dcpm);
                const double Ck = CC(k);
                cpa += Ck * cpm;
                dcpa += Ck * dcpm;
                const double cpk = (RU_ * pjs::SP[k][0]) * cpm;
                if constexpr (k == LAST) cpN = cpk;
                else SCV += cpk * (pjs::SP[k][1] * VS(k + 1));
            });
    }
#if PJQ_HALVES > 1
    // the groups' shares meet in group 0: D_k through the concentration columns (nobody reads them any more)
    static_range<1, G_>([&](auto gc) PJR_INL {
        constexpr int g = decltype(gc)::value;
        __syncthreads();
        if (grp == g) {
            static_for<NSP>([&](auto kc) PJR_INL { CL[decltype(kc)::value][tid] = acc[decltype(kc)::value]; });
            if constexpr (!DYDT_) { RED[0][tid] = JTN; RED[1][tid] = JTQ; RED[2][tid] = Hs; RED[3][tid] = SCP; }
        }
        __syncthreads();
        if (grp == 0) {
            static_for<NSP>([&](auto kc) PJR_INL { acc[decltype(kc)::value] += CL[decltype(kc)::value][tid]; });
            if constexpr (!DYDT_) { JTN += RED[0][tid]; JTQ += RED[1][tid]; Hs += RED[2][tid]; SCP += RED[3][tid]; }
        }
    });
    if (grp != 0) return;
#endif
    if constexpr (!LASTK_) {
        static_for<NSP>([&](auto kc) PJR_INL { A.sr[(SR_D + decltype(kc)::value) * A.sr_ld + s] = acc[decltype(kc)::value]; });
        A.sr[(SR_SC + 0) * A.sr_ld + s] = JTN; A.sr[(SR_SC + 1) * A.sr_ld + s] = JTQ;
        A.sr[(SR_SC + 2) * A.sr_ld + s] = Hs; A.sr[(SR_SC + 3) * A.sr_ld + s] = SCP;
    } else if constexpr (DYDT_) {
#if PJQ_JVD_DYDT
        // eval_spec_rates output (rate_subs.py:1297-1542) and dydt (rate_subs.py:2171-2335): k_rate's epilogue
        if (A.spec_rates)
            static_for<NSP>([&](auto kc) PJR_INL { PJQ_STORE(&OUT_(A.spec_rates, decltype(kc)::value), acc[decltype(kc)::value]); });
        if (A.dy) {
            PJQ_SCHED_BARRIER();
            fresh();
            const double cpavg = cpa * (RU_ * invrho);
            double Hd = 0.0;
            static_for<NSP>([&](auto kc) PJR_INL {
                constexpr int k = decltype(kc)::value;
                double hW, cpm, dcpm;
                nasa(kc, hW, cpm, dcpm);
                Hd += hW * acc[k];
                if constexpr (k < LAST) PJQ_STORE(&OUT_(A.dy, k + 1), acc[k] * pjs::SP[k][1] * invrho);
            });
            PJQ_STORE(&OUT_(A.dy, 0), -Hd / (rho * cpavg));
        }
#undef OUT_
#endif
    } else {
        double* wp = A.w + s * A.w_ss;
        double HD = 0.0, hWN = 0.0;
        PJQ_SCHED_BARRIER();
        fresh();
        static_for<NSP>([&](auto kc) PJR_INL {
            constexpr int k = decltype(kc)::value;
            double hW, cpm, dcpm;
            nasa(kc, hW, cpm, dcpm);
            HD += hW * acc[k];
            if constexpr (k == LAST) hWN = hW;
            else wp[(k + 1) * A.w_si] = pjs::SP[k][1] * acc[k];
        });
        const double cpavg = cpa * (RU_ * invrho), dcpavg = dcpa * (RU_ * invrho);
        const double icp = 1.0 / cpavg;
        const double quirk = A.sum_last ? 0.0 : hWN * (JTQ - JTN) * v0;
        wp[0] = -((SCP - (dcpavg * icp) * Hs) / (rho * cpavg)) * v0 - icp * (HD + quirk) +
                (SCV - cpN * SV0) * Hs * invrho * icp * icp;
    }
#undef CC
#undef VS
#undef SCR_ST
}

void launch_jvd(const PjqArgs& A, void* stream)
{
    const long blocks = (A.n + PJQ_BLOCK - 1) / PJQ_BLOCK;
    hipLaunchKernelGGL(k_jvd, dim3((unsigned)blocks), dim3(NTHR), 0, (hipStream_t)stream, A);
}
struct Reg { Reg() { pjq_register(PJQ_ID, PJQ_JVD_DYDT ? 10 : 9, launch_jvd); } } reg_;     // 9: w = J v, every reaction once; 10: its dydt build
#endif  // PJQ_PART == 5

#if PJQ_PART == 0
constexpr int MAXPARTS = 256;
pjq_launch_fn g_pre = nullptr, g_rows[MAXPARTS], g_rows_gen[MAXPARTS], g_rows_jv[MAXPARTS], g_timing[MAXPARTS];
pjq_launch_fn g_rate_lean[MAXPARTS], g_rate_full[MAXPARTS], g_jvd[MAXPARTS], g_rate_fast[MAXPARTS];
constexpr int MAXSTREAMS = 8;
// Everything a batch needs beyond the caller's arrays belongs to a CONTEXT: hand-over arrays, internal streams and
// events, the AoS staging block, the rate kernels' scratch, and the launch settings.  pj_api.hip creates one per
// mechanism handle (pj_spec_ctx_create at attach time), so two handles of one mechanism -- two Evaluators, two host
// threads -- run independently; callers of the plain entry points (tests, tools) share the library's default context.
// A context serves one batch at a time (its mutex serialises host threads; on the device a batch on another stream
// is ordered behind the previous one by enter_batch's event) and belongs to the device of its first batch.
struct Ctx {
    std::mutex batch_mutex;
    std::mutex aos_mutex;                // the staging block of the AoS path (taken before batch_mutex)
    double* scr[MAXSTREAMS] = {};
    long scr_ld[MAXSTREAMS] = {};
    hipStream_t streams[MAXSTREAMS] = {};
    hipEvent_t events[MAXSTREAMS + 1] = {};
    bool have_streams = false;
    int device = -1;
    int cus = 0;
    hipEvent_t last_event = nullptr;     // recorded after every batch, on the caller's stream
    void* last_stream = nullptr;
    double* aos_tmp = nullptr;
    long aos_tmp_states = 0;
    double* rate_scr = nullptr;          // omega_k between the rate kernels of a library that has several
    long rate_scr_ld = 0;
    double* rate_dummy = nullptr;        // one row that takes the per-reaction outputs the caller does not want
    long rate_dummy_ld = 0;
    double* jvd_scr[MAXSTREAMS] = {};    // D_k and four scalars between the k_jvd kernels of a library that has several
    long jvd_scr_ld[MAXSTREAMS] = {};
    int cfg_row_jv = 0;                  // w = J v through the row kernels' PJQ_JV builds (if the library has them)
    int cfg_rate_fast = 1;               // conc / spec_rates / dydt through k_jvd's dydt build (if the library has one) instead of k_rate
    long cfg_aos_chunk = 0;              // states per SoA staging block of the AoS path (0: 65536; PJ_RBLK_AOS_CHUNK)
    // launch settings: the environment is read ONCE, when the context is created (PJ_RBLK_STREAMS, PJ_RBLK_CHUNK,
    // PJ_RBLK_SPLIT, PJ_RBLK_AOS_DIRECT); pj_spec_ctx_config overrides them
    int cfg_streams = 0;                 // 0: the build's default (PJQ_STREAMS)
    long cfg_chunk = 0;                  // < 256: the build's default (PJQ_CHUNK)
    int cfg_split = 1;                   // two unequal parts on two streams for a partially filled last round
    int cfg_aos_direct = 0;              // AoS Jacobians by strided lane stores instead of SoA chunks + transpose
    int cfg_stagger = PJQ_STAGGER0;      // k_rblk: spread of the first round of workgroups (PjqArgs::stagger; PJ_RBLK_STAGGER)
    Ctx()
    {
        if (const char* e = getenv("PJ_RBLK_STAGGER")) cfg_stagger = atoi(e);
        if (const char* e = getenv("PJ_RBLK_STREAMS")) cfg_streams = atoi(e);
        if (const char* e = getenv("PJ_RBLK_CHUNK")) cfg_chunk = atol(e);
        if (const char* e = getenv("PJ_RBLK_SPLIT")) cfg_split = atoi(e) != 0;
        cfg_aos_direct = getenv("PJ_RBLK_AOS_DIRECT") != nullptr;
        cfg_row_jv = getenv("PJ_RBLK_ROW_JV") != nullptr;
        if (const char* e = getenv("PJ_RBLK_RATE_FAST")) cfg_rate_fast = atoi(e) != 0;
        if (const char* e = getenv("PJ_RBLK_AOS_CHUNK")) cfg_aos_chunk = atol(e);
    }
    void release()
    {
#ifndef PJR_HOST_EMU
        if (device >= 0) {
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != device) (void)hipSetDevice(device);
            (void)hipDeviceSynchronize();
            for (auto& p : scr) if (p) { (void)hipFree(p); p = nullptr; }
            for (auto& p : jvd_scr) if (p) { (void)hipFree(p); p = nullptr; }
            if (aos_tmp) (void)hipFree(aos_tmp);
            if (rate_scr) (void)hipFree(rate_scr);
            if (rate_dummy) (void)hipFree(rate_dummy);
            if (have_streams) {
                for (auto& st : streams) if (st) (void)hipStreamDestroy(st);
                for (auto& e : events) if (e) (void)hipEventDestroy(e);
            }
            if (last_event) (void)hipEventDestroy(last_event);
            if (cur != device && cur >= 0) (void)hipSetDevice(cur);
        }
#endif
    }
};
Ctx& default_ctx()
{
    static Ctx* c = new Ctx();           // never destroyed: the library stays resident (RTLD_NODELETE)
    return *c;
}
#endif

}  // namespace

#if PJQ_PART == 0
extern "C" {

void pjq_register(int id, int kind, pjq_launch_fn fn)
{
    if (kind == 1) g_pre = fn;
    else if (id >= 0 && id < MAXPARTS)
        (kind == 5 ? g_timing : kind == 4 ? g_rows_gen : kind == 6 ? g_rows_jv : kind == 7 ? g_rate_lean : kind == 8 ? g_rate_full : kind == 9 ? g_jvd : kind == 10 ? g_rate_fast : g_rows)[id] = fn;
}

// debug builds (-DPJQ_TIMING): cycles per phase of row kernel `part`, [5][1024 workgroups][4 wavefronts]
int pj_spec_debug_timing(int part, long long* out)
{
    if (part < 0 || part >= MAXPARTS || !g_timing[part]) return -1;
    PjqArgs A{};
    A.scr = (double*)out;
    g_timing[part](A, nullptr);
    return 0;
}

unsigned long long pj_spec_hash(void) { return PJS_HASH; }
int pj_spec_nsp(void) { return NSP; }
int pj_spec_kind(void) { return 4; }   // 1: pj_lane.hip, 4: pj_rblk.hip (2, 3: retired families)
long pj_spec_scratch_doubles_per_state(void) { return NSLOTS; }

// layouts as in include/pyjac_amd.h: element (i, s) at base[i*si + s*ss].  One batch at a time per
// context (its hand-over arrays are shared by its batches): calls on different streams are ordered.
//
// The batch runs in chunks, chunk c on internal stream c % S with its own hand-over array: all
// wavefronts of one launch move through "compute a block / store its rows" in step, so a single
// stream alternates between a busy memory system with idle SIMDs and the reverse; kernels of
// different chunks are out of step with each other.  The internal streams are forked from and joined
// to the caller's stream with events: the call is asynchronous and ordered like one kernel launch on
// `stream`.  streams = 1: everything on the caller's stream.
static int enter_batch(Ctx& C, void* stream)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return -3;
    if (C.device < 0) C.device = dev;
    else if (dev != C.device) return -6;
    if (!C.last_event) {
        if (hipEventCreateWithFlags(&C.last_event, hipEventDisableTiming) != hipSuccess) return -3;
    } else if (C.last_stream != stream) {
        (void)hipStreamWaitEvent((hipStream_t)stream, C.last_event, 0);
    }
    return 0;
}
static void leave_batch(Ctx& C, void* stream)
{
    (void)hipEventRecord(C.last_event, (hipStream_t)stream);
    C.last_stream = stream;
}

// a scratch array that only grows (geometrically: a caller that sweeps batch sizes upwards does not pay a
// device synchronisation + reallocation per call)
static int grow(double*& p, long& have, long want, size_t doubles_per_unit)
{
    if (have >= want) return 0;
    if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; }
    long cap = want;
    if (have > 0 && cap < have + have / 2) cap = have + have / 2;
    have = 0;
    if (hipMalloc((void**)&p, sizeof(double) * doubles_per_unit * (size_t)cap) != hipSuccess) {
        if (cap == want || hipMalloc((void**)&p, sizeof(double) * doubles_per_unit * (size_t)want) != hipSuccess) return -4;
        cap = want;
    }
    have = cap;
    return 0;
}

static int run_batch(Ctx& C, long n, const double* pres, const double* y, long y_si, long y_ss, double* jac, long j_si,
                     long j_ss, const double* v, long v_si, long v_ss, double* w, long w_si, long w_ss, int sum_last,
                     void* stream)
{
    if (n <= 0) return 0;
    const bool jv = w != nullptr;
    // w = J v: k_jvd (every reaction once) unless the library has only the row kernels' PJQ_JV builds or those are asked for
    // (the row kernels' product asked for -- PJ_RBLK_ROW_JV, pj_spec_ctx_row_jv -- in a library built without it: an error, not
    // a silent k_jvd; the build-time half of the switch is part of the library's digest, pyjac_amd/specbuild.py ENV)
    if (jv && C.cfg_row_jv && !g_rows_jv[0]) return -5;
    const bool jvd = jv && g_jvd[0] && !C.cfg_row_jv;
    if (jv && !jvd && !g_rows_jv[0]) return -5;
    int njvd = 0;
    while (jvd && njvd < MAXPARTS && g_jvd[njvd]) ++njvd;
    std::lock_guard<std::mutex> lock(C.batch_mutex);
    if (const int rc = enter_batch(C, stream)) return rc;
    int nstreams = C.cfg_streams > 0 ? C.cfg_streams : PJQ_STREAMS;
    if (nstreams > MAXSTREAMS) nstreams = MAXSTREAMS;
    const long chunk_cfg = C.cfg_chunk;
    // chunks: a multiple of the tile, at least 2 per stream when the batch fills the device several times
    long chunk = chunk_cfg >= 256 ? chunk_cfg : PJQ_CHUNK;
    // Every kernel of a step ends with a partially filled round of workgroups (one workgroup per CU is
    // resident; GRI-shaped 1e6 states = 15.26 rounds, USC-shaped 2e5 = 6.1), and kernels of one stream
    // do not overlap: 16 and 7 rounds are paid, per kernel.  Two unequal parts on two streams drift
    // apart, so one part's kernel fills the CUs the other part's last round leaves idle (measured:
    // 7.66 -> 7.37 ms and 10.56 -> 9.40 ms, tools/r02_tail.sh).  split = 0 switches it off;
    // explicit streams / chunk settings take precedence.
    if (C.cfg_streams <= 0 && chunk_cfg < 256 && PJQ_STREAMS == 1 && PJQ_SPLIT_TAIL && C.cfg_split) {
        if (!C.cus && hipDeviceGetAttribute(&C.cus, hipDeviceAttributeMultiprocessorCount, C.device) != hipSuccess) C.cus = 256;
        const long lds_wg = (long)NSP * PJQ_BLOCK * 8 * (PJQ_KCF ? 5 : 1) + 4096;   // (+ the factor columns)
        long per_cu = 256 / (PJQ_BLOCK * PJQ_HALVES);           // one wavefront per SIMD (512 registers)
        if (per_cu > (160L << 10) / lds_wg) per_cu = (160L << 10) / lds_wg;
        if (per_cu < 1) per_cu = 1;
        const long slots = (long)C.cus * per_cu, wgs = (n + PJQ_BLOCK - 1) / PJQ_BLOCK;
        const long rounds = (wgs + slots - 1) / slots;
        if (n <= PJQ_CHUNK && wgs >= 2 * slots && (double)(rounds * slots - wgs) > 0.02 * (double)(rounds * slots)) {
            chunk = (long)(0.525 * (double)n);
            nstreams = 2;
        }
    }
    chunk = (chunk + PJQ_TILE - 1) / PJQ_TILE * PJQ_TILE;
    if (chunk > n) chunk = (n + PJQ_TILE - 1) / PJQ_TILE * PJQ_TILE;
    const long nchunks = (n + chunk - 1) / chunk;
    const int S = (int)(nchunks < nstreams ? nchunks : nstreams);
    for (int b = 0; b < S; ++b) {
        if (jvd) { if (njvd > 1) if (const int rc = grow(C.jvd_scr[b], C.jvd_scr_ld[b], chunk, (size_t)NSP + 4)) return rc; }
        else if (const int rc = grow(C.scr[b], C.scr_ld[b], chunk, (size_t)NSLOTS)) return rc;
    }
    hipStream_t user = (hipStream_t)stream;
    if (S > 1) {
        if (!C.have_streams) {
            for (auto& st : C.streams)
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return -3;
            for (auto& e : C.events)
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -3;
            C.have_streams = true;
        }
        (void)hipEventRecord(C.events[MAXSTREAMS], user);
        for (int b = 0; b < S; ++b) (void)hipStreamWaitEvent(C.streams[b], C.events[MAXSTREAMS], 0);
    }
    // the pair-store kernels need lane-contiguous (SoA) output, whole workgroups and column offsets
    // that fit 32 bits; everything else goes to the general kernels of the library
    const bool have_fast = g_rows[0] != nullptr, have_gen = g_rows_gen[0] != nullptr;
    const bool fast_ok = !jv && have_fast && j_ss == 1 && (unsigned long)NSP * 8ul * (unsigned long)j_si < (1ul << 32);
    long c = 0;
    for (long s0 = 0; s0 < n; s0 += chunk, ++c) {
        const long m = s0 + chunk < n ? chunk : n - s0;
        const int b = (int)(c % S);
        void* st = S > 1 ? (void*)C.streams[b] : stream;
        PjqArgs A{m, pres + s0, y + s0 * y_ss, y_si, y_ss, jv ? nullptr : jac + s0 * j_ss, j_si, j_ss, C.scr[b], sum_last,
                  jv ? v + s0 * v_ss : nullptr, v_si, v_ss, jv ? w + s0 * w_ss : nullptr, w_si, w_ss};
        A.tile_rt = PJQ_TILE;
        A.stagger = C.cfg_stagger;
        const bool fast = fast_ok && m >= PJQ_BLOCK;
        if (!jv && !fast && !have_gen) return -5;
        if (jvd) {
            A.sr = C.jvd_scr[b]; A.sr_ld = C.jvd_scr_ld[b];
            for (int i = 0; i < njvd; ++i) g_jvd[i](A, st);
            continue;
        }
        if (g_pre) g_pre(A, st);
        pjq_launch_fn* rows = jv ? g_rows_jv : fast ? g_rows : g_rows_gen;
        for (int i = 0; i < MAXPARTS; ++i) if (rows[i]) rows[i](A, st);
    }
    if (S > 1)
        for (int b = 0; b < S; ++b) {
            (void)hipEventRecord(C.events[b], C.streams[b]);
            (void)hipStreamWaitEvent(user, C.events[b], 0);
        }
    leave_batch(C, stream);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// AoS Jacobians (pyJac's per-state C layout: state-major NSP x NSP blocks): a row block produces one row
// of every column, i.e. 8-byte pieces NSP doubles apart inside a state's block and NSP^2 apart between
// lanes -- nothing a wavefront could store contiguously.  So the row kernels write a chunk of SoA
// Jacobians (their native, pair-store layout) into a temporary block and this kernel transposes it:
// 64 x 64 tiles through LDS, 512-byte runs on both sides.
__global__ void __launch_bounds__(256) k_soa2aos(const double* __restrict__ src, long m, double* __restrict__ dst)
{
    __shared__ double tile[64][65];
    constexpr int NE = NSP * NSP;
    const long s0 = (long)blockIdx.x * 64;
    const int e0 = blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;         // 4 rows of 64 per pass
    for (int r = ly; r < 64; r += 4) {
        const int e = e0 + r;
        const long sidx = s0 + lx;
        if (e < NE && sidx < m) tile[r][lx] = __builtin_nontemporal_load(&src[(long)e * m + sidx]);
    }
    __syncthreads();
    for (int r = ly; r < 64; r += 4) {
        const long sidx = s0 + r;
        const int e = e0 + lx;
        if (sidx < m && e < NE) __builtin_nontemporal_store(tile[lx][r], &dst[sidx * NE + e]);
    }
}

int pj_spec_fast_aos(void) { return 1; }   // AoS Jacobians: SoA chunks + transpose, not strided lane stores

// contexts (see struct Ctx): created by pj_api.hip per mechanism handle; a null ctx means the library's default one
void* pj_spec_ctx_create(void) { return new (std::nothrow) Ctx(); }
void pj_spec_ctx_destroy(void* ctx)
{
    Ctx* C = (Ctx*)ctx;
    if (!C) return;
    { std::lock_guard<std::mutex> l1(C->aos_mutex); std::lock_guard<std::mutex> l2(C->batch_mutex); C->release(); }
    delete C;
}
// launch settings of a context; a negative value leaves a setting as it is.  streams: 0 = the build's default,
// chunk: states per chunk (< 256: the build's default), split: two unequal parts on two streams for batches whose
// last round of workgroups is partially filled, aos_direct: AoS Jacobians by strided lane stores
int pj_spec_ctx_config(void* ctx, int streams, long chunk, int split, int aos_direct)
{
    Ctx& C = ctx ? *(Ctx*)ctx : default_ctx();
    std::lock_guard<std::mutex> lock(C.batch_mutex);
    if (streams > MAXSTREAMS) return -1;
    if (streams >= 0) C.cfg_streams = streams;
    if (chunk >= 0) C.cfg_chunk = chunk;
    if (split >= 0) C.cfg_split = split != 0;
    if (aos_direct >= 0) C.cfg_aos_direct = aos_direct != 0;
    return 0;
}

// w = J v through the row kernels' PJQ_JV builds (on = 1; a library that has them) or through k_jvd (on = 0, the default);
// returns what is in force afterwards (also the environment: PJ_RBLK_ROW_JV), -5 if the row kernels' product is asked for and
// the library was built without it (specbuild: PJ_RBLK_ROW_JV at build time)
int pj_spec_ctx_row_jv(void* ctx, int on)
{
    Ctx& C = ctx ? *(Ctx*)ctx : default_ctx();
    std::lock_guard<std::mutex> lock(C.batch_mutex);
    if (on > 0 && !g_rows_jv[0]) return -5;
    if (on >= 0) C.cfg_row_jv = on != 0;
    return (C.cfg_row_jv && g_rows_jv[0]) || !g_jvd[0] ? 1 : 0;
}

// conc / spec_rates / dydt through k_jvd's dydt build (on = 1, the default where the library has one) or through k_rate's lean
// kernels (on = 0; also the environment: PJ_RBLK_RATE_FAST=0); returns what is in force afterwards
int pj_spec_ctx_rate_fast(void* ctx, int on)
{
    Ctx& C = ctx ? *(Ctx*)ctx : default_ctx();
    std::lock_guard<std::mutex> lock(C.batch_mutex);
    if (on >= 0) C.cfg_rate_fast = on != 0;
    return C.cfg_rate_fast && g_rate_fast[0] && !g_rate_fast[1] ? 1 : 0;
}

int pj_spec_jacobian_ctx(void* ctx, long n, const double* pres, const double* y, long y_si, long y_ss, double* jac,
                         long j_si, long j_ss, int sum_last, void* stream)
{
    Ctx& C = ctx ? *(Ctx*)ctx : default_ctx();
#ifndef PJR_HOST_EMU
    constexpr long NE = (long)NSP * NSP;
    if (n >= PJQ_BLOCK && j_si == 1 && j_ss == NE && g_rows[0] && !C.cfg_aos_direct) {
        std::lock_guard<std::mutex> lock(C.aos_mutex);
        // chunks that fill the device (one workgroup per CU, 256 CUs) a whole number of times: 65536 states
        long chunk = 256L * PJQ_BLOCK < 65536 ? 65536 : 256L * PJQ_BLOCK;
        if (C.cfg_aos_chunk >= PJQ_BLOCK) chunk = C.cfg_aos_chunk / PJQ_BLOCK * PJQ_BLOCK;
        if (chunk > n) chunk = n;
        if (const int rc = grow(C.aos_tmp, C.aos_tmp_states, chunk, (size_t)NE)) return rc;
        for (long s0 = 0; s0 < n; s0 += chunk) {
            long m = s0 + chunk < n ? chunk : n - s0;
            long sb = s0;
            if (m < PJQ_BLOCK) { sb = n - PJQ_BLOCK; m = PJQ_BLOCK; }      // short tail: redo a whole workgroup's worth
            const int rc = run_batch(C, m, pres + sb, y + sb * y_ss, y_si, y_ss, C.aos_tmp, m, 1, nullptr, 0, 0, nullptr, 0, 0,
                                     sum_last, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(k_soa2aos, dim3((unsigned)((m + 63) / 64), (unsigned)((NE + 63) / 64)), dim3(256), 0,
                               (hipStream_t)stream, (const double*)C.aos_tmp, m, jac + sb * NE);
        }
        (void)hipEventRecord(C.last_event, (hipStream_t)stream);      // the staging block is busy until here
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
#endif
    return run_batch(C, n, pres, y, y_si, y_ss, jac, j_si, j_ss, nullptr, 0, 0, nullptr, 0, 0, sum_last, stream);
}
int pj_spec_jacobian(long n, const double* pres, const double* y, long y_si, long y_ss, double* jac,
                     long j_si, long j_ss, int sum_last, void* stream)
{
    return pj_spec_jacobian_ctx(nullptr, n, pres, y, y_si, y_ss, jac, j_si, j_ss, sum_last, stream);
}

// w_s = J(Phi_s) v_s per state, the Jacobian consumed in registers (include/pyjac_amd.h:
// pj_eval_jacobian_vec_dev; pyJac's consumer: sparse_multiplier, create_jacobian.py:3301-3404)
int pj_spec_jacvec_ctx(void* ctx, long n, const double* pres, const double* y, long y_si, long y_ss, const double* v,
                       long v_si, long v_ss, double* w, long w_si, long w_ss, int sum_last, void* stream)
{
    if (!v || !w) return -1;
    return run_batch(ctx ? *(Ctx*)ctx : default_ctx(), n, pres, y, y_si, y_ss, nullptr, 0, 0, v, v_si, v_ss, w, w_si, w_ss,
                     sum_last, stream);
}
int pj_spec_jacvec(long n, const double* pres, const double* y, long y_si, long y_ss, const double* v,
                   long v_si, long v_ss, double* w, long w_si, long w_ss, int sum_last, void* stream)
{
    return pj_spec_jacvec_ctx(nullptr, n, pres, y, y_si, y_ss, v, v_si, v_ss, w, w_si, w_ss, sum_last, stream);
}

// Rate outputs of one pass (pyjacob.cu:18-35 k_dydt): conc, fwd, rev, pres_mod, spec_rates, dydt; any pointer
// may be null; SoA, leading dimension n.  One k_rate kernel per reaction range of the library (one for
// mechanisms whose K_c rows fit the LDS): omega_k stays in registers, or travels from kernel to kernel through
// the caller's spec_rates array (a chunk-sized scratch array when the caller does not want it).
int pj_spec_rates_ctx(void* ctx, long n, const double* pres, const double* y, long y_si, long y_ss, double* conc,
                      double* fwd, double* rev, double* pres_mod, double* spec_rates, double* dy, void* stream)
{
    if (n <= 0) return 0;
    Ctx& C = ctx ? *(Ctx*)ctx : default_ctx();
    const bool full = fwd || rev || pres_mod;
    // the lean outputs (conc / spec_rates / dydt): k_jvd's dydt build where the library has one (several lane groups on
    // shared concentration columns; PJ_RBLK_RATE_FAST=0: k_rate's lean kernels)
    const bool fast = !full && g_rate_fast[0] && !g_rate_fast[1] && C.cfg_rate_fast;
    pjq_launch_fn* parts = full ? g_rate_full : fast ? g_rate_fast : g_rate_lean;
    if (!parts[0]) return -5;
    std::lock_guard<std::mutex> lock(C.batch_mutex);
    if (const int rc = enter_batch(C, stream)) return rc;
    if (full && !(fwd && rev && pres_mod))
        if (const int rc = grow(C.rate_dummy, C.rate_dummy_ld, n, 1)) return rc;
    int nparts = 0;
    while (nparts < MAXPARTS && parts[nparts]) ++nparts;
    long chunk = n;
    if (nparts > 1 && !spec_rates) {
        chunk = n < 262144 ? n : 262144;
        if (const int rc = grow(C.rate_scr, C.rate_scr_ld, chunk, (size_t)NSP)) return rc;
    }
    for (long s0 = 0; s0 < n; s0 += chunk) {
        PjqArgs A{};
        A.n = s0 + chunk < n ? chunk : n - s0;
        A.pres = pres + s0; A.y = y + s0 * y_ss; A.y_si = y_si; A.y_ss = y_ss;
        A.conc = conc ? conc + s0 : nullptr; A.spec_rates = spec_rates ? spec_rates + s0 : nullptr;
        if (full) {
            A.fwd = fwd ? fwd + s0 : C.rate_dummy + s0; A.fwd_ld = fwd ? n : 0;
            A.rev = rev ? rev + s0 : C.rate_dummy + s0; A.rev_ld = rev ? n : 0;
            A.pres_mod = pres_mod ? pres_mod + s0 : C.rate_dummy + s0; A.pm_ld = pres_mod ? n : 0;
        }
        A.dy = dy ? dy + s0 : nullptr; A.o_ld = n;
        A.sr = spec_rates ? spec_rates + s0 : C.rate_scr; A.sr_ld = spec_rates ? n : C.rate_scr_ld;
        for (int i = 0; i < nparts; ++i) parts[i](A, stream);
    }
    leave_batch(C, stream);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int pj_spec_rates(long n, const double* pres, const double* y, long y_si, long y_ss, double* conc, double* fwd,
                  double* rev, double* pres_mod, double* spec_rates, double* dy, void* stream)
{
    return pj_spec_rates_ctx(nullptr, n, pres, y, y_si, y_ss, conc, fwd, rev, pres_mod, spec_rates, dy, stream);
}

}  // extern "C"
#endif
