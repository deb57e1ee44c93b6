// pj_lane.hip -- register-resident Jacobian kernel for SMALL mechanisms.
//
// One thermochemical state per lane; the mechanism is injected as constexpr
// tables (pj::emit_spec_header -> PJS_HEADER), every loop over species,
// reactions, molecule slots and Jacobian entries is fully unrolled, so all
// per-state arrays (C_k, omega_k, P_k, Q_k, the sparse S entries ...) have
// compile-time indices and live in VGPRs; stoichiometry, species indices and
// reaction kinds are folded into the instruction stream.  The real-valued
// coefficients (NASA rows, Arrhenius / falloff parameters, efficiencies) are
// staged in LDS once per workgroup and read with uniform ds_reads.  No
// divergence except the PLOG interval selects.  State loads and Jacobian
// stores are lane-contiguous (SoA) = fully coalesced.
//
// Same formulation as pj_kernel.h (which stays the path for mechanisms whose
// unrolled code or register footprint would be too large); reference
// emitters: pyjac/core/rate_subs.py:254-2335, pyjac/core/create_jacobian.py:2189-3298.
//
// Built per mechanism:  hipcc --offload-arch=gfx950 -O3 -DPJS_HEADER='"<hdr>"' -shared -fPIC pj_lane.hip
#ifdef PJL_HOST_EMU
#include "hip_shim.h"      // tests/emu: one lane per workgroup on the CPU (test infrastructure)
#else
#include <hip/hip_runtime.h>
#endif
#include <cstdint>
#include <type_traits>

#include "pj_tables.h"
#include PJS_HEADER

using namespace pj;

// SRI falloff and Chebyshev rate expressions live in pj_rate_pre.inc / pj_kernel.h only: such mechanisms
// are served by the row-block family (Evaluator.spec_kind) or the table-driven kernel
constexpr bool lane_supported()
{
    for (int i = 0; i < pjs::NRXN; ++i)
        if (pjs::RI[i][RI_FLAGS] & (F_SRI | F_CHEB | F_GEN)) return false;
    return true;
}
static_assert(lane_supported(), "pj_lane.hip: SRI / Chebyshev reactions and general stoichiometry are not implemented here; build kind 'rblk'");


namespace {

constexpr double RU_ = 8314.4621;
constexpr double INV_LN10 = 0.434294481903251828;
constexpr int NSP = pjs::NSP, NRXN = pjs::NRXN, LAST = pjs::NSP - 1, ONE = pjs::NSP;

// compile-time loop: f(std::integral_constant<int, I>) for I = 0..N-1, so every
// table index inside is a constant expression (guaranteed, not left to the unroller)
template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

#include "pj_math.h"
// 1 / u without the IEEE division sequence (div_scale x2, rcp, five FMAs, div_fmas, div_fixup: ~17 issue slots at one
// wavefront per SIMD): v_rcp_f64 (>= 24 bits) + two Newton steps, within an ulp of the correctly rounded value; the
// arguments here (T, rho, 1 + Pr, F_cent ...) are far from the range ends the sequence exists for.
#ifdef PJL_HOST_EMU
#define PJL_RCP(u) (1.0 / (u))
#else
__device__ __forceinline__ double pjl_rcp(const double u)
{
    double r = __builtin_amdgcn_rcp(u);
    r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-u, r, 1.0), r, r);
    return r;
}
#define PJL_RCP(u) pjl_rcp(u)
#endif

template <int J>
constexpr std::integral_constant<int, J + 1> jc_plus1(std::integral_constant<int, J>) { return {}; }

// does reaction i feed the last species' column (g_N != 0)?
constexpr bool has_gn(int i)
{
    const int fl = pjs::RI[i][RI_FLAGS];
    bool g = pjs::RD[i][RD_ANM1] != 0.0;
    for (int t = 0; t < 3; ++t) {
        g = g || pjs::RI[i][RI_R0 + t] == LAST;
        if (fl & F_REV) g = g || pjs::RI[i][RI_P0 + t] == LAST;
    }
    if (fl & F_COLLIDER) g = g || pjs::RI[i][RI_COLLIDER] == LAST;
    return g;
}
constexpr bool any_gn()
{
    for (int i = 0; i < NRXN; ++i) if (has_gn(i)) return true;
    return false;
}
constexpr bool ANY_GN = any_gn();      // false: Q_k == P_k, one accumulator set less

struct Args {
    long n;
    const double* pres;
    const double* y; long y_si, y_ss;
    double* jac; long j_si, j_ss;      // jac already points at state s0's block
    long s0;                           // first state of this launch chunk
    int sum_last;
    // fused Jacobian-vector product (k_lane<1, 0>): w_s = J(Phi_s) v_s, J never leaves the registers
    const double* v; long v_si, v_ss;
    double* w; long w_si, w_ss;
    // rate outputs (k_lane<2, .>), SoA with leading dimension o_ld, any may be null
    double *conc, *fwd, *rev, *pres_mod, *spec_rates, *dy; long o_ld;
};

#ifndef PJL_BLOCK
#define PJL_BLOCK 256    // one wavefront per SIMD; measured 64 / 128 / 256 x PJL_PERSIST 1 / 2 / 4
#endif

#ifndef PJL_PERSIST
#define PJL_PERSIST 1      // workgroups per resident slot; states are walked grid-stride
#endif

// MODE 0: Jacobian blocks to memory; 1: fused Jacobian-vector product; 2: rate outputs only
// (conc, fwd, rev, pres_mod, spec_rates, dy of pyjacob.cu's k_dydt pass -- the Jacobian
// accumulations are dead code there and the compiler drops them)
// ST (Jacobian store path, MODE 0): 0 strided plain stores, 1 lane-contiguous SoA (nontemporal),
// 3 the same as pair stores (two rows per 16-byte store, whole wavefronts only: see emit_col),
// 2 AoS (state-major NSP x NSP blocks, pyJac's per-state C layout) through a per-wavefront LDS
// transpose: a wavefront's 64 blocks are one contiguous region of memory, written in runs of whole
// columns instead of 8-byte stores that are NSP^2 doubles apart
template <int MODE, int ST>
__global__ void __launch_bounds__(PJL_BLOCK) k_lane(Args A)
{
    constexpr bool JV = MODE == 1;
    constexpr bool NT = ST == 1 || ST == 3;
    // AoS transpose tile: CC whole columns per flush, row stride padded to an odd number of doubles
    constexpr int CC = (48 / NSP) > 0 ? (48 / NSP) : 1, TW = CC * NSP, TWP = TW | 1;
    __shared__ double TL[(PJL_BLOCK + 63) / 64][ST == 2 ? 64 : 1][TWP];   // 1.3 KB placeholder when unused
    // NASA lo/hi coefficient rows live in LDS: one ds_read per coefficient pair at an
    // address picked by the range test, instead of a v_cndmask per 32-bit half
    // plus the real-valued coefficient tables (Arrhenius / falloff / Troe parameters,
    // efficiencies, molecular weights): uniform LDS reads where they are used, instead
    // of ~1000 64-bit literals that the persistent loop would hoist and spill
    __shared__ __attribute__((aligned(16))) double LT[pjs::LT_SIZE];
    __shared__ __attribute__((aligned(16))) double RDL[NRXN][RDW];
    __shared__ __attribute__((aligned(16))) double EFL[sizeof(pjs::EFFT) / 8][1];
    __shared__ __attribute__((aligned(16))) double SPL[NSP][4];
    for (int w = threadIdx.x; w < pjs::LT_SIZE; w += PJL_BLOCK) LT[w] = pjs::LTAB[w];
    for (int w = threadIdx.x; w < NRXN * RDW; w += PJL_BLOCK) (&RDL[0][0])[w] = (&pjs::RDT[0][0])[w];
    for (int w = threadIdx.x; w < (int)(sizeof(pjs::EFFT) / 8); w += PJL_BLOCK) EFL[w][0] = pjs::EFFT[w][0];
    for (int w = threadIdx.x; w < NSP * 4; w += PJL_BLOCK) (&SPL[0][0])[w] = (&pjs::SPT[0][0])[w];
    __syncthreads();
  // ST == 2: the trip count is workgroup-uniform (every lane takes part in the transpose; lanes past
  // the end repeat the last state); otherwise a lane simply stops at the end of the batch
  // The state of the NEXT grid-stride iteration is requested at the top of the current one: at one
  // wavefront per SIMD nothing else covers the loaded memory latency (several microseconds behind other
  // wavefronts' Jacobian stores), and these loads sit in front of this iteration's stores in the
  // in-order vmcnt queue.
#define PJL_INL __attribute__((always_inline))
  double nxT = 0.0, nxP = 0.0, nxY[LAST > 0 ? LAST : 1];
  auto fetch_state = [&](const long sl_) PJL_INL {
      const bool in = (ST == 2 ? sl_ - threadIdx.x : sl_) < A.n;
      if (in) {
          const long s_ = (ST == 2 && sl_ >= A.n) ? A.n - 1 : sl_;
          const double* y_ = A.y + s_ * A.y_ss;
          nxT = y_[0];
          nxP = A.pres[s_];
#pragma unroll
          for (int k = 0; k < LAST; ++k) nxY[k] = y_[(k + 1) * A.y_si];
      }
  };
  // pair stores (ST == 3): lanes 0..31 of a wavefront hold its even states, lanes 32..63 the odd ones, so
  // that one v_permlane32_swap per 32-bit half exchanges what a 16-byte store of two neighbouring states
  // needs (pair_store below); every other instance maps lane i to state i
  const unsigned ltid = ST == 3 ? (threadIdx.x & ~63u) + 2u * (threadIdx.x & 31u) + ((threadIdx.x >> 5) & 1u) : threadIdx.x;
  fetch_state(A.s0 + (long)blockIdx.x * PJL_BLOCK + ltid);
  for (long sl = A.s0 + (long)blockIdx.x * PJL_BLOCK + ltid; (ST == 2 ? sl - threadIdx.x : sl) < A.n;
       sl += (long)gridDim.x * PJL_BLOCK) {
    const long tb = sl - ltid;
    const long s = (ST == 2 && sl >= A.n) ? A.n - 1 : sl;
    // Without global stores in the loop body the optimiser treats the LDS tables as loop invariant
    // and hoists hundreds of coefficient reads out of the persistent loop (spills).  An opaque zero
    // offset per state stops that and keeps the pointers recognisable as LDS addresses (laundering
    // the pointers themselves turns every coefficient read into a flat_load).
    unsigned zoff = 0;
#ifndef PJL_HOST_EMU
    if constexpr (MODE != 0 || ST == 2) asm volatile("" : "+s"(zoff));
#endif
    const double (*RDT)[RDW] = (const double (*)[RDW])((const char*)RDL + zoff);
    const double (*EFFT)[1] = (const double (*)[1])((const char*)EFL + zoff);
    const double (*SPT)[4] = (const double (*)[4])((const char*)SPL + zoff);
    const double T = nxT;
    const double p = nxP;
    const double logT = log(T), invT = PJL_RCP(T), logp = log(p);

    // ---- eval_conc + NASA properties ----
    double C[NSP + 1], hW[NSP], cpk[NSP];
    double sumY = 0.0, sumYW = 0.0;
#pragma unroll
    for (int k = 0; k < LAST; ++k) {
        C[k] = nxY[k];
        sumY += C[k];
        sumYW += C[k] * SPT[k][0];
    }
    fetch_state(sl + (long)gridDim.x * PJL_BLOCK);
    const double yN = 1.0 - sumY;
    C[LAST] = yN;
    sumYW += yN * SPT[LAST][0];
    const double Wbar = PJL_RCP(sumYW);
    const double mconc = p * (invT * (1.0 / RU_));
    const double rho = mconc * Wbar, invrho = PJL_RCP(rho);
    double cpavg = 0.0, dcpavg = 0.0;
#pragma unroll
    for (int k = 0; k < NSP; ++k) {
        const double* a = LT + pjs::LT_SP + k * 16 + ((T <= SPT[k][2]) ? 0 : 8);
        hW[k] = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                       T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
        const double RW = RU_ * SPT[k][0];
        cpk[k] = RW * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
        const double dcp = RW * (a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T)));
        cpavg += C[k] * cpk[k];
        dcpavg += C[k] * dcp;
        C[k] = rho * C[k] * SPT[k][0];
    }
    C[ONE] = 1.0;

    double om[NSP], jt[NSP], P[NSP], Q[ANY_GN ? NSP : 1], S[pjs::NNZ];
    double ekc[pjs::NKCCLS], tdk[pjs::NKCCLS];
    double jtq = 0.0;
#pragma unroll
    for (int k = 0; k < NSP; ++k) { om[k] = 0.0; jt[k] = 0.0; P[k] = 0.0; if (ANY_GN) Q[k] = 0.0; }
#pragma unroll
    for (int e = 0; e < pjs::NNZ; ++e) S[e] = 0.0;

    // ---- reactions: compile-time loop, every table read is a constant expression ----
    static_for<NRXN>([&](auto ic) PJL_INL {
        constexpr int i = decltype(ic)::value;
        constexpr int fl = pjs::RI[i][RI_FLAGS];
        double lnk, dlnk;
        if constexpr ((fl & F_PLOG) != 0) {
            constexpr int pp = pjs::RI[i][RI_PLOG_PTR], np = pjs::RI[i][RI_PLOG_CNT];
            // interval select chain over the breakpoints (rate_subs.py:598-632)
            lnk = pjs::PLOG[pp][2] + pjs::PLOG[pp][3] * logT - pjs::PLOG[pp][4] * invT;
            dlnk = pjs::PLOG[pp][3] + pjs::PLOG[pp][4] * invT;
            static_for<np - 1>([&](auto qc) PJL_INL {
                constexpr int q = decltype(qc)::value + 1;
                constexpr double P1 = pjs::PLOG[pp + q - 1][0], L1 = pjs::PLOG[pp + q - 1][1],
                                 A1 = pjs::PLOG[pp + q - 1][2], B1 = pjs::PLOG[pp + q - 1][3],
                                 E1 = pjs::PLOG[pp + q - 1][4];
                constexpr double P2 = pjs::PLOG[pp + q][0], L2 = pjs::PLOG[pp + q][1], A2 = pjs::PLOG[pp + q][2],
                                 B2 = pjs::PLOG[pp + q][3], E2 = pjs::PLOG[pp + q][4];
                const double k1 = A1 + B1 * logT - E1 * invT;
                const double k2 = A2 + B2 * logT - E2 * invT;
                const double f = (logp - L1) * PJL_RCP(L2 - L1);
                const bool in = p > P1 && p <= P2;
                lnk = in ? k1 + (k2 - k1) * f : lnk;
                dlnk = in ? B1 + E1 * invT + ((B2 - B1) + (E2 - E1) * invT) * f : dlnk;
            });
            {
                constexpr double Pn = pjs::PLOG[pp + np - 1][0], An = pjs::PLOG[pp + np - 1][2],
                                 Bn = pjs::PLOG[pp + np - 1][3], En = pjs::PLOG[pp + np - 1][4];
                const bool hi = p > Pn;
                lnk = hi ? An + Bn * logT - En * invT : lnk;
                dlnk = hi ? Bn + En * invT : dlnk;
            }
        } else {
            lnk = RDT[i][RD_LNA] + RDT[i][RD_B] * logT - RDT[i][RD_TA] * invT;
            dlnk = RDT[i][RD_B] + RDT[i][RD_TA] * invT;
        }
        // k_f and, where this reaction is the first of its K_c class, exp(-ln K_c): side by side
        // (pj_math.h: two independent Horner chains fill each other's pipeline latency)
        double kf, kr = 0.0, TdlnKc = 0.0;
        if constexpr ((fl & F_REV) != 0) {
            constexpr int kcls = pjs::KC_CLASS[i][0];
            if constexpr (pjs::KC_FIRST[i][0] != 0) {
                double lnKc = RDT[i][RD_LNPREF], td = 0.0;
                static_for<pjs::RI[i][RI_KC_CNT]>([&](auto cc) PJL_INL {
                    constexpr int g = pjs::RI[i][RI_KC_PTR] + decltype(cc)::value;
                    const double* a = LT + pjs::LT_KC + g * 16 + ((T <= pjs::KCG[g][0]) ? 0 : 8);
                    lnKc += a[0] + a[1] * logT + T * (a[2] + T * (a[3] + T * (a[4] + a[5] * T))) - a[6] * invT;
                    td += a[1] + T * (a[2] + T * (2.0 * a[3] + T * (3.0 * a[4] + 4.0 * a[5] * T))) + a[6] * invT;
                });
                exp_pair(lnk, -lnKc, kf, ekc[kcls]);
                tdk[kcls] = td;
            } else {
                kf = exp_one(lnk);
            }
            if constexpr (pjs::RD[i][RD_SGN] < 0.0) kf = -kf;
            kr = kf * ekc[kcls];
            TdlnKc = tdk[kcls];
        } else {
            kf = exp_one(lnk);
            if constexpr (pjs::RD[i][RD_SGN] < 0.0) kf = -kf;
        }

        const double cr0 = C[pjs::RI[i][RI_R0]], cr1 = C[pjs::RI[i][RI_R1]], cr2 = C[pjs::RI[i][RI_R2]];
        const double cp0 = C[pjs::RI[i][RI_P0]], cp1 = C[pjs::RI[i][RI_P1]], cp2 = C[pjs::RI[i][RI_P2]];
        const double Rf = kf * (cr0 * cr1 * cr2);
        const double Rr = kr * (cp0 * cp1 * cp2);
        const double R = Rf - Rr;

        double c = 1.0, lead = 0.0, a_extra = 0.0, bM = 0.0, bcol = 0.0;
        if constexpr ((fl & (F_THD | F_PDEP)) != 0) {
            double Mc = mconc;
            static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJL_INL {
                constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                Mc += EFFT[e][0] * C[pjs::EFF_SP[e][0]];
            });
            if constexpr ((fl & F_THD) != 0) {
                c = Mc;
                lead = -c * R * invT;
                if constexpr ((fl & F_EFFTYPE) != 0) { bM = R; a_extra = c * R; }
            } else {
                constexpr int col = pjs::RI[i][RI_COLLIDER];
                double conc_temp = Mc;
                if constexpr (col >= 0) conc_temp = C[col >= 0 ? col : 0];
                const double e0T = RDT[i][RD_E0] * invT;
                const double k0kinf = exp(RDT[i][RD_LNAR] + RDT[i][RD_B0] * logT - e0T);
                const double Pr = conc_temp * k0kinf;
                const double i1Pr = PJL_RCP(1.0 + Pr);
                double F = 1.0, extra = 0.0, Xtroe = 0.0;
                if constexpr ((fl & F_TROE) != 0) {
                    const double ta = RDT[i][RD_TRA], T3 = RDT[i][RD_T3], T1 = RDT[i][RD_T1],
                                 T2 = RDT[i][RD_T2];
                    const double iT3 = PJL_RCP(T3), iT1 = PJL_RCP(T1);
                    const double e3 = exp(-T * iT3), e1 = exp(-T * iT1);
                    double Fcent = (1.0 - ta) * e3 + ta * e1;
                    double dF = -((1.0 - ta) * iT3) * e3 - (ta * iT1) * e1;
                    if constexpr ((fl & F_TROE4) != 0) {
                        const double e2 = exp(-T2 * invT);
                        Fcent += e2;
                        dF += T2 * invT * invT * e2;
                    }
                    const double lF = log(fmax(Fcent, 1.0e-300));
                    const double lgF = lF * INV_LN10;
                    const double lgPr = log(fmax(Pr, 1.0e-300)) * INV_LN10;
                    const double At = lgPr - 0.67 * lgF - 0.4;
                    const double Bt = 0.806 - 1.1762 * lgF - 0.14 * lgPr;
                    const double iB = PJL_RCP(Bt);
                    const double iden = PJL_RCP(1.0 + At * At * iB * iB);
                    F = exp(lF * iden);
                    const double lnF_AB = 2.0 * lF * At * iB * iB * iB * iden * iden;
                    const double iFc = PJL_RCP(Fcent);
                    Xtroe = lnF_AB * (INV_LN10 * Bt + (0.14 * INV_LN10) * At);
                    extra = (iFc * iden - lnF_AB * (-(0.67 * INV_LN10) * Bt + (1.1762 * INV_LN10) * At) * iFc) * dF -
                            Xtroe * (RDT[i][RD_B0] + e0T - 1.0) * invT;
                }
                double dpr = (RDT[i][RD_B04] + e0T - 1.0) * invT * i1Pr;
                double X;
                if constexpr ((fl & F_LOW) != 0) { c = F * Pr * i1Pr; X = i1Pr - Xtroe; }
                else { c = F * i1Pr; X = -Pr * i1Pr - Xtroe; dpr = -Pr * dpr; }
                lead = c * (dpr + extra) * R;
                if constexpr ((fl & (F_EFFTYPE | F_COLLIDER)) != 0) {
                    const double pmt = X * R;
                    a_extra = c * pmt;
                    const double bb = pmt * k0kinf * F * i1Pr;
                    if constexpr ((fl & F_COLLIDER) != 0) bcol = bb; else bM = bb;
                }
            }
        }

        constexpr double nr = pjs::RD[i][RD_NR], np_ = pjs::RD[i][RD_NP];
        double el = R * dlnk + Rf * (1.0 - nr);
        if constexpr ((fl & F_REV) != 0) el -= Rr * ((1.0 - np_) - TdlnKc);
        const double theta = (fl & F_NO_DT) ? 0.0 : (lead + c * invT * el) * invrho;
        const double a = c * (nr * Rf - ((fl & F_REV) ? np_ * Rr : 0.0)) + a_extra;
        const double ckf = c * kf, ckr = c * kr;

        // sparse values per molecule slot, accumulated straight into S (compile-time indices)
        double gN = 0.0;
        if constexpr (pjs::RD[i][RD_ANM1] != 0.0) gN = bM * RDT[i][RD_ANM1];
        constexpr int np0 = pjs::RI[i][RI_NET_PTR], ncnt = pjs::RI[i][RI_NET_CNT];
        auto slot = [&](auto spc, const double gv) PJL_INL {
            constexpr int sp = decltype(spc)::value;
            if constexpr (sp == LAST) gN += gv;
            else if constexpr (sp != ONE) {
                static_for<ncnt>([&](auto qc) PJL_INL {
                    constexpr int q = np0 + decltype(qc)::value;
                    constexpr int si = pjs::SIDX[pjs::NET_SP[q][0]][sp];
                    static_assert(si >= 0, "sparse pattern and program disagree");
                    S[si] += pjs::NET_NU[q][0] * gv;
                });
            }
        };
        slot(std::integral_constant<int, pjs::RI[i][RI_R0]>{}, ckf * (cr1 * cr2));
        slot(std::integral_constant<int, pjs::RI[i][RI_R1]>{}, ckf * (cr0 * cr2));
        slot(std::integral_constant<int, pjs::RI[i][RI_R2]>{}, ckf * (cr0 * cr1));
        if constexpr ((fl & F_REV) != 0) {
            slot(std::integral_constant<int, pjs::RI[i][RI_P0]>{}, -ckr * (cp1 * cp2));
            slot(std::integral_constant<int, pjs::RI[i][RI_P1]>{}, -ckr * (cp0 * cp2));
            slot(std::integral_constant<int, pjs::RI[i][RI_P2]>{}, -ckr * (cp0 * cp1));
        }
        if constexpr ((fl & F_COLLIDER) != 0)
            slot(std::integral_constant<int, (pjs::RI[i][RI_COLLIDER] >= 0 ? pjs::RI[i][RI_COLLIDER] : ONE)>{}, bcol);
        if constexpr ((fl & F_EFFTYPE) != 0) {
            static_for<pjs::RI[i][RI_EFF_CNT]>([&](auto ec) PJL_INL {
                constexpr int e = pjs::RI[i][RI_EFF_PTR] + decltype(ec)::value;
                constexpr int es = pjs::EFF_SP[e][0];
                // the last species' enhanced efficiency is already in gN (RD_ANM1)
                if constexpr (es != LAST) slot(std::integral_constant<int, es>{}, EFFT[e][0] * bM);
            });
        }

        const double q_ = c * R;
        if constexpr (MODE == 2) {
            // rate_subs.py:634-658, 811-840, 1076-1283: indices are positions in the mechanism file
            if (A.fwd) __builtin_nontemporal_store(Rf, &A.fwd[pjs::RI[i][RI_ORIG] * A.o_ld + s]);
            if constexpr (pjs::RI[i][RI_REV_IDX] >= 0) { if (A.rev) __builtin_nontemporal_store(Rr, &A.rev[pjs::RI[i][RI_REV_IDX] * A.o_ld + s]); }
            if constexpr (pjs::RI[i][RI_PRES_IDX] >= 0) {
                if (A.pres_mod) __builtin_nontemporal_store(c, &A.pres_mod[pjs::RI[i][RI_PRES_IDX] * A.o_ld + s]);
            }
        }
        const double rp = (Wbar * invrho) * (q_ - a) + bM;
        static_for<ncnt>([&](auto qc) PJL_INL {
            constexpr int q = np0 + decltype(qc)::value;
            constexpr int k = pjs::NET_SP[q][0];
            constexpr double nu = pjs::NET_NU[q][0];
            om[k] += nu * q_;
            jt[k] += nu * theta;
            P[k] += nu * rp;
            // (Q holds QN_k = sum nu gN, not Q_k = P_k + QN_k: pj_rblk.hip, near_last() -- for a column whose species weighs what
            // the last species weighs, P_k - w_j Q_k cancels to -QN_k, and formed from the two sums it carries P_k's rounding error)
            if constexpr (ANY_GN) Q[k] += nu * gN;
            if constexpr (k == LAST && i == pjs::LASTQ) jtq = nu * theta;
        });
    });
    if constexpr (MODE == 2) {
        // eval_spec_rates / dydt (rate_subs.py:1297-1542, 2171-2335)
        double Hs = 0.0;
#pragma unroll
        for (int k = 0; k < NSP; ++k) {
            Hs += hW[k] * om[k];
            if (A.conc) __builtin_nontemporal_store(C[k], &A.conc[k * A.o_ld + s]);
            if (A.spec_rates) __builtin_nontemporal_store(om[k], &A.spec_rates[k * A.o_ld + s]);
            if (A.dy && k < LAST) __builtin_nontemporal_store(om[k] * SPT[k][1] * invrho, &A.dy[(k + 1) * A.o_ld + s]);
        }
        if (A.dy) __builtin_nontemporal_store(-Hs * (invrho * PJL_RCP(cpavg)), &A.dy[s]);
        continue;
    }
    // reference quirk (create_jacobian.py:2786-2818), see pj_kernel.h phase 3
    if (!A.sum_last) jt[LAST] = jtq;

    // ---- Jacobian block: every entry written ----
    double H = 0.0, scp = 0.0, sjt = 0.0;
#pragma unroll
    for (int k = 0; k < NSP; ++k) {
        H += hW[k] * om[k];
        scp += om[k] * SPT[k][1] * cpk[k];
        sjt += hW[k] * jt[k];
    }
    double* const Jl = JV ? nullptr : A.jac + (s - A.s0) * A.j_ss;
    // JV: every Jacobian entry goes into w[row] += J(row, col) * v[col] instead of memory
    double vv[NSP], ww[NSP];
    if constexpr (JV) {
        const double* vs = A.v + s * A.v_ss;
#pragma unroll
        for (int k = 0; k < NSP; ++k) { vv[k] = vs[k * A.v_si]; ww[k] = 0.0; }
    }
// Jacobian entries are written once and not read back: nontemporal stores when a wavefront's
// store is one contiguous line (SoA, NT = true); with a state stride between lanes (AoS) they are
// partial-line writes and nontemporal is 7x slower than write-back caching
#define JMEM(e, val) do { if constexpr (ST == 2) TL[threadIdx.x / 64][threadIdx.x % 64][(e) - ((e) / NSP / CC) * TW] = (val); \
                          else if constexpr (NT) __builtin_nontemporal_store((val), &Jl[(long)(e) * A.j_si]); \
                          else Jl[(long)(e) * A.j_si] = (val); } while (0)
    // after column `col` is complete: write the tile's columns [c0, col] of this wavefront's states
    auto flush = [&](auto colc) PJL_INL {
        constexpr int col = decltype(colc)::value;
        if constexpr (ST == 2 && ((col + 1) % CC == 0 || col == NSP - 1)) {
            constexpr int c0 = col / CC * CC, w = (col + 1 - c0) * NSP;     // doubles per state in this flush
            const int wv = threadIdx.x / 64, ln = threadIdx.x % 64;
            const long wbase = tb + wv * 64;                                   // first state of the wavefront
            const long nvalid = A.n - wbase;                                   // states of it inside the batch
            double* const Jw = A.jac + (wbase - A.s0) * (long)(NSP * NSP) + c0 * NSP;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int i = 0; i < w; ++i) {
                const int idx = i * 64 + ln, st = idx / w, en = idx - st * w;
                if (st < nvalid) Jw[(long)st * (NSP * NSP) + en] = TL[wv][st][en];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };
    // SoA output from whole wavefronts goes out as PAIR stores (ST == 3): the halves of the wavefront
    // swap one value per two rows, then a lower lane writes two states of row r1, its upper partner the
    // same two states of row r2 > r1 -- 16 bytes per lane, half the store instructions in flight (a wavefront may
    // have 64 outstanding; the store phase is bound by that window, profiles/r02_micro_burst_bw.txt).
    // Rows are paired as they are produced: (1,2), (3,4), ... and the energy row 0, which needs the
    // whole column, with the last species row.
    [[maybe_unused]] const unsigned jlo = (ltid & 63u) * 8u;
    [[maybe_unused]] const bool upper = (threadIdx.x & 32u) != 0;
    // wavefront base: a scalar, so that a store is "SGPR base + 32-bit lane offset" with no 64-bit vector
    // address arithmetic; the strides are laundered per state (the optimiser would hoist every entry
    // offset e * j_si out of the persistent loop as an SGPR pair and spill them)
    [[maybe_unused]] double* Jw = nullptr;
    [[maybe_unused]] long jsi = A.j_si;
#ifndef PJL_HOST_EMU
    if constexpr (ST == 3 && !JV) {
        const long sw = sl - (ltid & 63u);
        const long swu = ((long)__builtin_amdgcn_readfirstlane((int)((unsigned long)sw >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sw);
        Jw = A.jac + (swu - A.s0) * A.j_ss;
        asm volatile("" : "+s"(jsi));
    }
#endif
    auto pair_store = [&](auto e1c, const double v1, auto e2c, const double v2) PJL_INL {
#ifndef PJL_HOST_EMU
        constexpr int e1 = decltype(e1c)::value, e2 = decltype(e2c)::value;
        static_assert(e2 > e1, "pair stores: second entry above the first");
        typedef double d2v __attribute__((ext_vector_type(2)));
        const unsigned long long u1 = __builtin_bit_cast(unsigned long long, v1), u2 = __builtin_bit_cast(unsigned long long, v2);
        // upper half of v1 <-> lower half of v2: the lower lanes then hold entry e1 of states (2 l, 2 l + 1),
        // the upper lanes entry e2 of the same two states
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u1, (unsigned)u2, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u1 >> 32), (unsigned)(u2 >> 32), false, false);
        d2v out;
        out.x = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
        out.y = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
        const unsigned off = upper ? jlo - 8u + (unsigned)(e2 - e1) * ((unsigned)jsi * 8u) : jlo;
        __builtin_nontemporal_store(out, (d2v*)((char*)(Jw + (long)e1 * jsi) + off));
#endif
    };
    auto single_store = [&](auto ec, const double v) PJL_INL {
        constexpr int e = decltype(ec)::value;
        if constexpr (JV) ww[e % NSP] += v * vv[e / NSP];
        else JMEM(e, v);
    };
#define PJL_E(e_) std::integral_constant<int, (e_)>{}
    const double icp = PJL_RCP(cpavg);
    // one column: val(k) -> row k + 1 for k = 0 .. LAST-1 (called in order), then row 0
    auto column = [&](auto colc, auto&& val, auto&& row0) PJL_INL {
        constexpr int col = decltype(colc)::value;
        if constexpr (ST == 3 && !JV) {
            double held = 0.0;      // odd-numbered species row waiting for its partner
            static_for<LAST>([&](auto kc) PJL_INL {
                constexpr int k = decltype(kc)::value;
                const double v = val(kc);
                if constexpr (k % 2 == 0) held = v;
                else pair_store(PJL_E(k + NSP * col), held, PJL_E(k + 1 + NSP * col), v);
            });
            const double v0 = row0();
            if constexpr (LAST % 2 == 1) pair_store(PJL_E(NSP * col), v0, PJL_E(LAST + NSP * col), held);
            else __builtin_nontemporal_store(v0, (double*)((char*)(Jw + (long)(NSP * col) * jsi) + jlo));
        } else {
            static_for<LAST>([&](auto kc) PJL_INL {
                constexpr int k = decltype(kc)::value;
                single_store(PJL_E(k + 1 + NSP * col), val(kc));
            });
            single_store(PJL_E(NSP * col), row0());
        }
        flush(colc);
    };
    column(PJL_E(0), [&](auto kc) PJL_INL { return SPT[decltype(kc)::value][1] * jt[decltype(kc)::value]; },
           [&]() PJL_INL { return -(scp - (dcpavg * icp) * H + rho * sjt) * (invrho * icp); });
    static_for<LAST>([&](auto jc) PJL_INL {
        constexpr int j = decltype(jc)::value;
        const double wj = SPT[j][3], iWj = SPT[j][0];
        const double omwj = 1.0 - wj;
        double tot = 0.0;
        auto mval = [&](auto kc) PJL_INL {
            constexpr int k = decltype(kc)::value;
            constexpr int si = pjs::SIDX[k][j];
            double m;
            if constexpr (ANY_GN) m = omwj * P[k] - wj * Q[k]; else m = omwj * P[k];
            if constexpr (si >= 0) m += S[si];
            tot += hW[k] * m;
            return m;
        };
        column(jc_plus1(jc), [&](auto kc) PJL_INL { return (SPT[decltype(kc)::value][1] * iWj) * mval(kc); },
               [&]() PJL_INL {
                   (void)mval(PJL_E(LAST));       // the last species has no row; its term enters the energy row
                   return -tot * iWj * icp + (cpk[j] - cpk[LAST]) * H * invrho * icp * icp;
               });
    });
#undef PJL_E
#undef JMEM
    if constexpr (JV) {
        double* ws = A.w + s * A.w_ss;
#pragma unroll
        for (int k = 0; k < NSP; ++k) ws[k * A.w_si] = ww[k];
    }
#undef PJL_INL
  }
}

}  // namespace

extern "C" {

unsigned long long pj_spec_hash(void) { return PJS_HASH; }
int pj_spec_nsp(void) { return NSP; }
int pj_spec_fast_aos(void) { return 1; }   // AoS Jacobians go through the LDS transpose (k_lane<0, 2>)

// layouts as in include/pyjac_amd.h: element (i, s) at base[i*si + s*ss]
int pj_spec_jacobian(long n, const double* pres, const double* y, long y_si, long y_ss, double* jac,
                     long j_si, long j_ss, int sum_last, void* stream)
{
    if (n <= 0) return 0;
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lane<0, 1>, PJL_BLOCK, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    // launch in chunks whose per-lane byte offset into the Jacobian fits 32 bits
    const long stride = j_ss > 0 ? j_ss : 1;
    long chunk = (long)(0xffffffffUL / (8UL * (unsigned long)stride)) - 64;
    if (chunk > n) chunk = n;
    for (long s0 = 0; s0 < n; s0 += chunk) {
        const long s1 = s0 + chunk < n ? s0 + chunk : n;
        Args A{s1, pres, y, y_si, y_ss, jac + s0 * j_ss, j_si, j_ss, s0, sum_last, nullptr, 0, 0, nullptr, 0, 0,
               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        long blocks = (s1 - s0 + PJL_BLOCK - 1) / PJL_BLOCK;
        if (blocks > resident * PJL_PERSIST) blocks = resident * PJL_PERSIST;
        if (j_ss == 1) {
#ifndef PJL_HOST_EMU
            // whole wavefronts through the pair-store kernel, the ragged tail (< 64 states) through the
            // scalar-store one
            const long s_main = s0 + (s1 - s0) / 64 * 64;
            const bool pair = s_main > s0 && (unsigned long)j_si * 8ul * (unsigned long)NSP < (1ul << 32);
            if (pair) {
                Args M = A;
                M.n = s_main;
                long mb = (s_main - s0 + PJL_BLOCK - 1) / PJL_BLOCK;
                if (mb > resident * PJL_PERSIST) mb = resident * PJL_PERSIST;
                hipLaunchKernelGGL((k_lane<0, 3>), dim3((unsigned)mb), dim3(PJL_BLOCK), 0, (hipStream_t)stream, M);
                if (s_main < s1) {
                    Args Tl = A;
                    Tl.s0 = s_main;
                    Tl.jac = jac + s_main * j_ss;
                    hipLaunchKernelGGL((k_lane<0, 1>), dim3(1), dim3(PJL_BLOCK), 0, (hipStream_t)stream, Tl);
                }
            } else
#endif
            hipLaunchKernelGGL((k_lane<0, 1>), dim3((unsigned)blocks), dim3(PJL_BLOCK), 0, (hipStream_t)stream, A);
        }
#ifndef PJL_HOST_EMU                  // the transpose needs whole wavefronts
        else if (j_si == 1 && j_ss == NSP * NSP)
            hipLaunchKernelGGL((k_lane<0, 2>), dim3((unsigned)blocks), dim3(PJL_BLOCK), 0, (hipStream_t)stream, A);
#endif
        else hipLaunchKernelGGL((k_lane<0, 0>), dim3((unsigned)blocks), dim3(PJL_BLOCK), 0, (hipStream_t)stream, A);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Fused Jacobian-vector product w_s = J(Phi_s) v_s (the consumer of pyJac's sparse_multiplier,
// create_jacobian.py:3301-3404, applied while the entries are still in registers): reads T, p, Y and v,
// writes NSP doubles per state.  Layouts as above.
int pj_spec_jacvec(long n, const double* pres, const double* y, long y_si, long y_ss, const double* v,
                   long v_si, long v_ss, double* w, long w_si, long w_ss, int sum_last, void* stream)
{
    if (n <= 0) return 0;
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lane<1, 0>, PJL_BLOCK, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    Args A{n, pres, y, y_si, y_ss, nullptr, 0, 0, 0, sum_last, v, v_si, v_ss, w, w_si, w_ss,
           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    long blocks = (n + PJL_BLOCK - 1) / PJL_BLOCK;
    if (blocks > resident * PJL_PERSIST) blocks = resident * PJL_PERSIST;
    hipLaunchKernelGGL((k_lane<1, 0>), dim3((unsigned)blocks), dim3(PJL_BLOCK), 0, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// Rate outputs of one pass (pyjacob.cu:18-35 k_dydt): any pointer may be null; SoA, leading dimension n.
int pj_spec_rates(long n, const double* pres, const double* y, long y_si, long y_ss, double* conc, double* fwd,
                  double* rev, double* pres_mod, double* spec_rates, double* dy, void* stream)
{
    if (n <= 0) return 0;
    static long resident = 0;
    if (!resident) {
        int dev = 0, cus = 256, per_cu = 1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lane<2, 0>, PJL_BLOCK, 0);
        resident = (long)cus * (per_cu > 0 ? per_cu : 1);
    }
    Args A{n, pres, y, y_si, y_ss, nullptr, 0, 0, 0, 0, nullptr, 0, 0, nullptr, 0, 0,
           conc, fwd, rev, pres_mod, spec_rates, dy, n};
    long blocks = (n + PJL_BLOCK - 1) / PJL_BLOCK;
    if (blocks > resident * PJL_PERSIST) blocks = resident * PJL_PERSIST;
    hipLaunchKernelGGL((k_lane<2, 0>), dim3((unsigned)blocks), dim3(PJL_BLOCK), 0, (hipStream_t)stream, A);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // extern "C"
