// pj_kernel.h -- device phases of the batched rate / analytical-Jacobian evaluator.
//
// Replaces pyJac's generated evaluator chain
//   eval_conc -> eval_rxn_rates -> get_rxn_pres_mod -> eval_spec_rates -> dydt / eval_jacob
// (emitters: pyjac/core/rate_subs.py:254-2335, pyjac/core/create_jacobian.py:2189-3298)
// with one table-driven kernel; the mechanism is data (pj_tables.h).
//
// Work mapping.  A workgroup of NT threads owns a tile of TS states.  Thread
// tid handles state s = tid % TS (fixed for the whole tile, so per-state
// scalars stay in registers) and walks work items u, u+NU, ... with
// u = tid / TS, NU = NT / TS.  With TS = 64 a wavefront is 64 states of one
// item (state-per-lane, wave-uniform tables); smaller TS trades states per
// wave for items per wave when the per-state working set is large.  All
// per-state arrays live in LDS as V[slot][TS] (state fastest -> conflict-free).
//
// Formulation.  pyJac emits, per reaction and per column, NSP-1 updates
//   J[k,j] += nu_k (W_k/W_j) (a_i (1 - W_j/W_N) + b_i (alpha_ij - alpha_iN W_j/W_N) + sparse_ij)
// (create_jacobian.py:341-489, 2850-2938).  Here the parts that are dense in j
// are accumulated once per species (P_k, Q_k) and only the truly sparse terms
// (columns of reactants / products / enhanced colliders) are gathered per
// entry, so J[k,j] = (W_k/W_j) (P_k - (W_j/W_N) Q_k + S_kj).  Same algebra,
// different rounding order.
#pragma once
#include "pj_tables.h"

#ifndef PJ_DEV
#define PJ_DEV __device__ __forceinline__
#endif

namespace pj {

constexpr double RU_ = 8314.4621;   // chem_utilities.py:16
constexpr double LN10 = 2.302585092994045684;
constexpr double INV_LN10 = 0.434294481903251828;

// C^nu of a general-stoichiometry factor: whole-number coefficients by repeated multiplication (as the
// reference emits them), fractional ones through pow() (rate_subs.py:634-658)
PJ_DEV double pj_cpow(const double C, const double nu)
{
    if (nu == floor(nu)) {
        double r = 1.0;
        for (int q = 0; q < (int)nu; ++q) r *= C;
        return r;
    }
    return pow(C, nu);
}

struct DevMech {
    int nsp, nrxn, ng, ne, nv;
    int lastq_rxn, sum_last;
    VMap v;
    const double* sp;
    const int32_t* ri;
    const double* rd;
    const uint16_t* smap;      // sparse-block slot map and per-column non-zero rows (pj_tables.h)
    const int32_t* ecol_ptr;
    const uint32_t* ecol;
    const int32_t* rti;        // field-major reaction tables (pj_tables.h)
    const double* rtd;
    int nrp;
    const int32_t* eff_sp;
    const double* eff_am1;
    const double* kcg;
    const double* plog;
    const double* sri;         // SRI parameter rows (pj_tables.h: SRW)
    const double* cheb;        // Chebyshev records
    const int32_t* net_sp;
    const double* net_nu;
    const int32_t* gen_sp;     // F_GEN reactions: (species, nu) factors (pj_tables.h)
    const double* gen_nu;
    double nutab[NUTAB_N];     // scatter nu codes 8..: fractional / large net coefficients
    const int32_t* sp_ptr;
    const int32_t* sp_rxn;
    const double* sp_nu;
    // scatter schedule (pj_tables.h: Schedule), built for (NT / 64) wavefronts x (64 / TS) item lanes
    const uint32_t* sched;
    int sched_off[16], sched_rounds[16], sched_rounds_dense[16];
    const int32_t* fin_tgt;
    const int32_t* fin_part;
    const int32_t* fin_cnt;
    int nfin;
};

// One launch's arguments.  Element (i, s) of a 2-D quantity lives at
// base[i * si + s * ss]; SoA (pyJac's batch layout, pyjacob.cu:139-187) is
// si = ld, ss = 1; AoS (pyJac's per-state C layout) is si = 1, ss = rows.
struct Batch {
    long n;
    const double* pres;
    const double* y; long y_si, y_ss;        // rows: T, Y_0 .. Y_{NSP-2}
    double* jac; long j_si, j_ss;            // rows: r + NSP*c (column-major per state)
    double* conc; double* fwd; double* rev; double* pres_mod; double* spec_rates; double* dy;
    long o_ld;                               // SoA leading dimension of the rate outputs
};

// species constants shared by all states of the workgroup, staged once:
// KC[0*nsp + k] = 1/W_k, KC[1*nsp + k] = W_k, KC[2*nsp + k] = W_k / W_N
template <int TS>
PJ_DEV double* lds_kc(const DevMech& M, double* V) { return V + (size_t)M.v.NSLOT * TS; }
template <int TS>
PJ_DEV const double* lds_kc(const DevMech& M, const double* V) { return V + (size_t)M.v.NSLOT * TS; }

template <int TS>
PJ_DEV void stage_consts(const DevMech& M, double* V, int tid, int NT)
{
    double* KC = lds_kc<TS>(M, V);
    for (int k = tid; k < M.nsp; k += NT) {
        KC[k] = M.sp[k * SPW + 0];
        KC[M.nsp + k] = M.sp[k * SPW + 1];
        KC[2 * M.nsp + k] = M.sp[k * SPW + 3];
    }
}

#ifndef PJ_LDS_ADD
// tile[x] += v without waiting for the old value: non-returning LDS atomic
// (ds_add_f64).  The schedule gives every target to one wavefront and LDS
// executes a wavefront's operations in order, so the sum order is fixed.
#define PJ_LDS_ADD(ptr, v) (void)__hip_atomic_fetch_add((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif

struct Lane {
    double T, logT, invT, p, logp, rho, invrho, Wbar, m, yN;
    double cpavg, dcp;             // per-state sums (phase 0b)
    long gs;
    int valid;
};

// ---------------------------------------------------------------- phase 0
// eval_conc (rate_subs.py:1625-1710) + eval_h / eval_cp (rate_subs.py:1806-2086).
// 0a: each species item stores Y_k and
// its NASA properties; 0b: every lane forms the per-state sums from LDS
// (Y_N, Wbar, rho, cp_avg, dcp_avg/dT); 0c: the item lanes turn Y_k into C_k = rho Y_k / W_k.
template <int TS>
PJ_DEV void phase0a(const DevMech& M, const Batch& B, double* V, int tid, int NT, long tile, Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp;
    long gs = tile * TS + s;
    L.valid = gs < B.n;
    if (!L.valid) gs = B.n - 1;
    L.gs = gs;
    const double* y = B.y + gs * B.y_ss;
    const double T = y[0];
    const double p = B.pres[gs];
    L.T = T; L.p = p;
    L.logT = log(T); L.invT = 1.0 / T; L.logp = log(p);
    L.m = p / (RU_ * T);
    for (int k = u; k < nsp; k += NU) {
        const double* sp = M.sp + k * SPW;
        const double* a = (T <= sp[2]) ? sp + 4 : sp + 11;
        const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                 T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
        const double RW = RU_ * sp[0];
        const double cp = RW * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
        const double dcp = RW * (a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T)));
        // the last species' mass fraction is known only after the sum (phase 0b):
        // its slots carry the unweighted values
        const double Yk = (k == nsp - 1) ? 1.0 : y[(k + 1) * B.y_si];
        V[(M.v.C + k) * TS + s] = Yk;
        V[(M.v.HW + k) * TS + s] = hW;
        V[(M.v.CP + k) * TS + s] = cp;
        V[(M.v.YC + k) * TS + s] = Yk * cp;
        V[(M.v.YD + k) * TS + s] = Yk * dcp;
    }
    if (u == 0) V[M.v.ONE * TS + s] = 1.0;
}

template <int TS>
PJ_DEV void phase0b(const DevMech& M, const Batch& B, double* V, int tid, int NT, Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp, last = nsp - 1;
    const double* KC = lds_kc<TS>(M, V);
    double sumY = 0.0, sumYW = 0.0, cpa = 0.0, dcp = 0.0;
#pragma unroll 8
    for (int k = 0; k < last; ++k) {
        const double Yk = V[(M.v.C + k) * TS + s];
        sumY += Yk;
        sumYW += Yk * KC[k];
        cpa += V[(M.v.YC + k) * TS + s];
        dcp += V[(M.v.YD + k) * TS + s];
    }
    const double yN = 1.0 - sumY;
    sumYW += yN * KC[last];
    cpa += yN * V[(M.v.YC + last) * TS + s];
    dcp += yN * V[(M.v.YD + last) * TS + s];
    L.yN = yN;
    L.Wbar = 1.0 / sumYW;
    L.rho = L.p * L.Wbar / (RU_ * L.T);
    L.invrho = 1.0 / L.rho;
    L.cpavg = cpa;
    L.dcp = dcp;
}

// second half of 0b, after every lane has read the unscaled slots
template <int TS>
PJ_DEV void phase0c_scale(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp, last = nsp - 1;
    for (int k = u; k < nsp; k += NU) {
        const double Yk = (k == last) ? L.yN : V[(M.v.C + k) * TS + s];
        const double Ck = L.rho * Yk * lds_kc<TS>(M, V)[k];
        V[(M.v.C + k) * TS + s] = Ck;
        if (B.conc && L.valid) B.conc[k * B.o_ld + L.gs] = Ck;
    }
}

// ---------------------------------------------------------------- phase 2
// eval_rxn_rates (rate_subs.py:254-876), get_rxn_pres_mod (rate_subs.py:879-1294)
// and the per-reaction derivative scalars of eval_jacob
// (create_jacobian.py:127-269 dR/dY, 953-1294 falloff, 1398-1529 d/dT, 1687-1850 PLOG).
template <int TS>
PJ_DEV void phase2(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const double T = L.T, logT = L.logT, invT = L.invT;
    const int last = M.nsp - 1;
    for (int i = u; i < M.nrxn; i += NU) {
        // field-major tables: lane i reads element i of every field -> coalesced, and all
        // loads of a reaction are issued together (one memory round trip)
        const int32_t* rti = M.rti + i;
        const double* rtd = M.rtd + i;
        const int NRP = M.nrp;
#define RI_(f) rti[(f) * NRP]
#define RD_(f) rtd[(f) * NRP]
        const int fl = RI_(RI_FLAGS);

        // ---- forward rate constant and T d(ln kf)/dT ----
        double lnk, dlnk;
        double kf_jac_ratio = 1.0;     // Chebyshev only: (k_f of eval_jacob's dR/dY_j terms) / (k_f of the rates)
        if (fl & F_PLOG) {
            // rate_subs.py:598-632; create_jacobian.py:1687-1850
            const double* P = M.plog + RI_(RI_PLOG_PTR) * PLW;
            const int np = RI_(RI_PLOG_CNT);
            int q = 0;
            while (q < np && L.p > P[q * PLW]) ++q;      // first breakpoint with p <= P_q
            if (q == 0 || q == np) {
                const double* r = P + (q == 0 ? 0 : np - 1) * PLW;
                lnk = r[2] + r[3] * logT - r[4] * invT;
                dlnk = r[3] + r[4] * invT;
            } else {
                const double* r1 = P + (q - 1) * PLW;
                const double* r2 = P + q * PLW;
                const double k1 = r1[2] + r1[3] * logT - r1[4] * invT;
                const double k2 = r2[2] + r2[3] * logT - r2[4] * invT;
                const double f = (L.logp - r1[1]) / (r2[1] - r1[1]);
                lnk = k1 + (k2 - k1) * f;
                dlnk = r1[3] + r1[4] * invT + ((r2[3] - r1[3]) + (r2[4] - r1[4]) * invT) * f;
            }
        } else if (fl & F_CHEB) {
            // rate_subs.py:149-251 ('{:.8e}' constants); create_jacobian.py:1532-1684 ('{:.16e}')
            const double* C = M.cheb + RI_(RI_PLOG_PTR);
            const int cn = RI_(RI_PLOG_CNT) >> 8, cm = RI_(RI_PLOG_CNT) & 255;
            const double lg10p = L.logp * INV_LN10;
            double dp[CHEB_MAXT];
            {
                const double Tred = (2.0 * invT - C[CH_TSUM8]) / C[CH_TSUB8];
                const double Pred = (2.0 * lg10p - C[CH_PSUM8]) / C[CH_PSUB8];
                const double* c = C + CH_COEF;
                for (int a = 0; a < cn; ++a) {
                    double acc = c[a * cm] + Pred * c[a * cm + 1];
                    double t0 = 1.0, t1 = Pred;
                    for (int j = 2; j < cm; ++j) {
                        const double tn = 2.0 * Pred * t1 - t0;
                        acc += c[a * cm + j] * tn;
                        t0 = t1; t1 = tn;
                    }
                    dp[a] = acc;
                }
                double kl = dp[0] + Tred * dp[1];
                double u0 = 1.0, u1 = Tred;
                for (int a = 2; a < cn; ++a) {
                    const double un = 2.0 * Tred * u1 - u0;
                    kl += dp[a] * un;
                    u0 = u1; u1 = un;
                }
                lnk = kl * LN10;
            }
            {
                const double Tred = (2.0 * invT - C[CH_TSUM16]) / C[CH_TSUB16];
                const double Pred = (2.0 * lg10p - C[CH_PSUM16]) / C[CH_PSUB16];
                {
                    // eval_jacob's own k_f for the dR/dY_j terms: the rate's coefficients with THESE reduced
                    // variables (get_cheb_rate(write_defns=False), create_jacobian.py:1647-1664)
                    const double* c8 = C + CH_COEF;
                    double kl = 0.0, u0 = 1.0, u1 = Tred;
                    for (int a = 0; a < cn; ++a) {
                        double acc = c8[a * cm] + Pred * c8[a * cm + 1];
                        double t0 = 1.0, t1 = Pred;
                        for (int j = 2; j < cm; ++j) {
                            const double tn = 2.0 * Pred * t1 - t0;
                            acc += c8[a * cm + j] * tn;
                            t0 = t1; t1 = tn;
                        }
                        if (a == 0) kl = acc;
                        else if (a == 1) kl += Tred * acc;
                        else { const double un = 2.0 * Tred * u1 - u0; kl += acc * un; u0 = u1; u1 = un; }
                    }
                    kf_jac_ratio = exp(kl * LN10 - lnk);
                }
                const double* c = C + CH_COEF + cn * cm;         // rows i = 1 .. cn-1 of i * c_ij
                for (int a = 1; a < cn; ++a) {
                    double acc = c[(a - 1) * cm] + Pred * c[(a - 1) * cm + 1];
                    double t0 = 1.0, t1 = Pred;
                    for (int j = 2; j < cm; ++j) {
                        const double tn = 2.0 * Pred * t1 - t0;
                        acc += c[(a - 1) * cm + j] * tn;
                        t0 = t1; t1 = tn;
                    }
                    dp[a] = acc;
                }
                double U = dp[1] + 2.0 * Tred * dp[2];
                double w0 = 1.0, w1 = 2.0 * Tred;
                for (int a = 3; a < cn; ++a) {
                    const double wn = 2.0 * Tred * w1 - w0;
                    U += dp[a] * wn;
                    w0 = w1; w1 = wn;
                }
                dlnk = U * C[CH_DFAC] * invT;
            }
        } else {
            lnk = RD_(RD_LNA) + RD_(RD_B) * logT - RD_(RD_TA) * invT;
            dlnk = RD_(RD_B) + RD_(RD_TA) * invT;
        }
        const double kf = RD_(RD_SGN) * exp(lnk);

        // ---- equilibrium constant (pre-summed NASA groups, rate_subs.py:660-809) ----
        double kr = 0.0, TdlnKc = 0.0;
        if (fl & F_REV) {
            double lnKc = RD_(RD_LNPREF);
            {   // group 0 sits inline in the field-major table
                const int o = (T <= RD_(RDW + 0)) ? RDW + 1 : RDW + 8;
                const double a0 = RD_(o), a1 = RD_(o + 1), a2 = RD_(o + 2), a3 = RD_(o + 3), a4 = RD_(o + 4),
                             a5 = RD_(o + 5), a6 = RD_(o + 6);
                lnKc += a0 + a1 * logT + T * (a2 + T * (a3 + T * (a4 + a5 * T))) - a6 * invT;
                TdlnKc += a1 + T * (a2 + T * (2.0 * a3 + T * (3.0 * a4 + 4.0 * a5 * T))) + a6 * invT;
            }
            const double* g = M.kcg + (RI_(RI_KC_PTR) + 1) * KCW;      // species with other T_mid: rare
            for (int c = 1; c < RI_(RI_KC_CNT); ++c, g += KCW) {
                const double* a = (T <= g[0]) ? g + 1 : g + 8;
                lnKc += a[0] + a[1] * logT + T * (a[2] + T * (a[3] + T * (a[4] + a[5] * T))) - a[6] * invT;
                TdlnKc += a[1] + T * (a[2] + T * (2.0 * a[3] + T * (3.0 * a[4] + 4.0 * a[5] * T))) + a[6] * invT;
            }
            kr = kf * exp(-lnKc);
        }

        // ---- concentration products ----
        const double cr0 = V[RI_(RI_R0) * TS + s], cr1 = V[RI_(RI_R1) * TS + s], cr2 = V[RI_(RI_R2) * TS + s];
        const double cp0 = V[RI_(RI_P0) * TS + s], cp1 = V[RI_(RI_P1) * TS + s], cp2 = V[RI_(RI_P2) * TS + s];
        double prodr = cr0 * cr1 * cr2, prodp = cp0 * cp1 * cp2;
        const int gp0 = RI_(RI_GEN_PTR), gnr = RI_(RI_GEN_NR), gnp = RI_(RI_GEN_NP);
        if (fl & F_GEN) {
            // general stoichiometry: C^nu by repeated multiplication for whole numbers, pow() otherwise
            // (rate_subs.py:634-658, 811-840)
            for (int f = 0; f < gnr; ++f) prodr *= pj_cpow(V[M.gen_sp[gp0 + f] * TS + s], M.gen_nu[gp0 + f]);
            for (int f = 0; f < gnp; ++f) prodp *= pj_cpow(V[M.gen_sp[gp0 + gnr + f] * TS + s], M.gen_nu[gp0 + gnr + f]);
        }
        const double Rf = kf * prodr;
        const double Rr = kr * prodp;
        const double R = Rf - Rr;

        // ---- pressure modification ----
        double c = 1.0, lead = 0.0, a_extra = 0.0, bM = 0.0, bcol = 0.0;
        if (fl & (F_THD | F_PDEP)) {
            double Mc = L.m;
            const int necnt = RI_(RI_EFF_CNT);
#pragma unroll
            for (int e = 0; e < EFF_INL; ++e)      // inline slots: padding is (ONE, 0.0)
                Mc += RD_(RDW + KCW + e) * V[RI_(RIW + e) * TS + s];
            for (int e = EFF_INL; e < necnt; ++e)
                Mc += M.eff_am1[RI_(RI_EFF_PTR) + e] * V[M.eff_sp[RI_(RI_EFF_PTR) + e] * TS + s];
            if (fl & F_THD) {
                c = Mc;
                lead = -c * R * invT;
                if (fl & F_EFFTYPE) { bM = R; a_extra = c * R; }
            } else {
                const int col = RI_(RI_COLLIDER);
                const double conc_temp = (col >= 0) ? V[col * TS + s] : Mc;
                const double e0T = RD_(RD_E0) * invT;
                const double k0kinf = exp(RD_(RD_LNAR) + RD_(RD_B0) * logT - e0T);
                const double Pr = conc_temp * k0kinf;
                const double i1Pr = 1.0 / (1.0 + Pr);
                double F = 1.0, extra = 0.0, Xtroe = 0.0;
                if (fl & F_TROE) {
                    // create_jacobian.py:1066-1111, 1240-1294
                    const double ta = RD_(RD_TRA);
                    const double e3 = exp(-T / RD_(RD_T3)), e1 = exp(-T / RD_(RD_T1));
                    double Fcent = (1.0 - ta) * e3 + ta * e1;
                    double dF = -((1.0 - ta) / RD_(RD_T3)) * e3 - (ta / RD_(RD_T1)) * e1;
                    if (fl & F_TROE4) {
                        const double e2 = exp(-RD_(RD_T2) * invT);
                        Fcent += e2;
                        dF += RD_(RD_T2) * invT * invT * e2;
                    }
                    const double lF = log(fmax(Fcent, 1.0e-300));
                    const double lgF = lF * INV_LN10;
                    const double lgPr = log(fmax(Pr, 1.0e-300)) * INV_LN10;
                    const double At = lgPr - 0.67 * lgF - 0.4;
                    const double Bt = 0.806 - 1.1762 * lgF - 0.14 * lgPr;
                    const double iB = 1.0 / Bt;
                    const double den = 1.0 + At * At * iB * iB;
                    const double iden = 1.0 / den;
                    F = exp(lF * iden);
                    const double lnF_AB = 2.0 * lF * At * iB * iB * iB * iden * iden;
                    const double iFc = 1.0 / Fcent;
                    Xtroe = lnF_AB * (INV_LN10 * Bt + (0.14 * INV_LN10) * At);
                    extra = (iFc * iden - lnF_AB * (-(0.67 * INV_LN10) * Bt + (1.1762 * INV_LN10) * At) * iFc) * dF -
                            Xtroe * (RD_(RD_B0) + e0T - 1.0) * invT;
                }
                if (fl & F_SRI) {
                    // rate_subs.py:1229-1256; create_jacobian.py:173-179, 249-266, 1194-1237: each emitter's
                    // parameter digits (pj_tables.h: SRW)
                    const double* Q = M.sri + RI_(RI_PLOG_PTR) * SRW;
                    const double lgPr = log(fmax(Pr, 1.0e-300)) * INV_LN10;
                    const double Xs = 1.0 / (1.0 + lgPr * lgPr);
                    const double S6 = Q[SR_A6] * exp(-Q[SR_B6] * invT) + exp(-T / Q[SR_C6]);
                    F = exp(Xs * log(S6));
                    if (Q[SR_USE_DE] != 0.0) F *= Q[SR_D8] * exp(Q[SR_E6] * logT);
                    const double S4 = Q[SR_A4] * exp(-Q[SR_B4] * invT) + exp(-T / Q[SR_C4]);
                    const double C2 = 0.8685889638065035;        // '{:.16}'.format(2 / ln 10)
                    Xtroe = Xs * Xs * C2 * lgPr * log(S4);
                    const double eb = exp(-Q[SR_B16] * invT), ec = exp(-T / Q[SR_C16]);
                    const double S16 = Q[SR_A16] * eb + ec;
                    const double dS = (Q[SR_AB16] * invT * invT) * eb - Q[SR_INVC16] * ec;
                    extra = Xs * (dS / S16 - Xs * C2 * lgPr * (RD_(RD_B0) + e0T - 1.0) * log(S16) * invT) + Q[SR_E16] * invT;
                }
                // get_pdep_dt (create_jacobian.py:1135-1191): beta difference as printed ('%.4e')
                double dpr = (RD_(RD_B04) + e0T - 1.0) * invT * i1Pr;
                double X;
                if (fl & F_LOW) { c = F * Pr * i1Pr; X = i1Pr - Xtroe; }
                else { c = F * i1Pr; X = -Pr * i1Pr - Xtroe; dpr = -Pr * dpr; }
                lead = c * (dpr + extra) * R;
                if (fl & (F_EFFTYPE | F_COLLIDER)) {
                    const double pmt = X * R;
                    a_extra = c * pmt;
                    const double bb = pmt * k0kinf * F * i1Pr;
                    if (fl & F_COLLIDER) bcol = bb; else bM = bb;
                }
            }
        }

        // ---- d/dT (create_jacobian.py:1398-1529) ----
        const double nr = RD_(RD_NR), np_ = RD_(RD_NP);
        double el = R * dlnk + Rf * (1.0 - nr);
        if (fl & F_REV) el -= Rr * ((1.0 - np_) - TdlnKc);
        const double theta = (fl & F_NO_DT) ? 0.0 : (lead + c * invT * el) * L.invrho;

        // ---- dense-in-j scalars (create_jacobian.py:127-269) ----
        // (q - a with a = c (nr R_f - np R_r) + a_extra is formed below so that nothing cancels when nr or np is 1)

        // ---- sparse column values g (one per molecule slot) ----
        const double ckf = c * kf * kf_jac_ratio, ckr = c * kr * kf_jac_ratio;
        int g = M.v.G + RI_(RI_GBASE);
        double gN = bM * RD_(RD_ANM1);
        #define PJ_GSLOT(spidx, val)                                        \
            if ((spidx) != M.v.ONE) {                                       \
                const double gv = (val);                                    \
                V[g * TS + s] = gv; ++g;                                    \
                if ((spidx) == last) gN += gv;                              \
            }
        PJ_GSLOT(RI_(RI_R0), ckf * (cr1 * cr2))
        PJ_GSLOT(RI_(RI_R1), ckf * (cr0 * cr2))
        PJ_GSLOT(RI_(RI_R2), ckf * (cr0 * cr1))
        if (fl & F_REV) {
            PJ_GSLOT(RI_(RI_P0), -ckr * (cp1 * cp2))
            PJ_GSLOT(RI_(RI_P1), -ckr * (cp0 * cp2))
            PJ_GSLOT(RI_(RI_P2), -ckr * (cp0 * cp1))
        }
        if (fl & F_GEN) {
            // one value per factor: c k nu C^(nu-1) prod_others (create_jacobian.py:400-448; the power of C_j
            // itself only "if (nu - 1) > 0": reference quirk kept for parity)
            for (int side = 0; side < ((fl & F_REV) ? 2 : 1); ++side) {
                const int f0 = gp0 + side * gnr, nf = side ? gnp : gnr;
                const double ck = side ? -ckr : ckf;
                for (int f = 0; f < nf; ++f) {
                    const double nuf = M.gen_nu[f0 + f];
                    double gv = ck * nuf;
                    if (nuf - 1.0 > 0.0) gv *= pj_cpow(V[M.gen_sp[f0 + f] * TS + s], nuf - 1.0);
                    for (int h = 0; h < nf; ++h)
                        if (h != f) gv *= pj_cpow(V[M.gen_sp[f0 + h] * TS + s], M.gen_nu[f0 + h]);
                    V[g * TS + s] = gv; ++g;
                    if (M.gen_sp[f0 + f] == last) gN += gv;
                }
            }
        }
        if (fl & F_COLLIDER) { PJ_GSLOT(RI_(RI_COLLIDER), bcol) }
        #undef PJ_GSLOT
        if (fl & F_EFFTYPE)      // (alpha_ij - 1) b_i for the enhanced-collider columns
            for (int e = 0; e < RI_(RI_EFF_CNT); ++e) {
                const int es = (e < EFF_INL) ? RI_(RIW + e) : M.eff_sp[RI_(RI_EFF_PTR) + e];
                const double am1 = (e < EFF_INL) ? RD_(RDW + KCW + e) : M.eff_am1[RI_(RI_EFF_PTR) + e];
                if (es != last) { V[g * TS + s] = am1 * bM; ++g; }
            }

        const double q = c * R;
        const double rp = (L.Wbar * L.invrho) * (c * ((1.0 - nr) * Rf - ((fl & F_REV) ? (1.0 - np_) * Rr : 0.0)) - a_extra) + bM;
        V[(M.v.RQ + i) * TS + s] = q;
        V[(M.v.RTH + i) * TS + s] = theta;
        V[(M.v.RP + i) * TS + s] = rp;
        // (the dense sums carry QN_k = sum nu gN, not Q_k = P_k + QN_k: for a column whose species weighs what the last species
        // weighs, P_k - w_j Q_k = -QN_k exactly, and formed from the two sums it carries the rounding error of P_k -- pj_rblk.hip,
        // near_last())
        V[(M.v.RQQ + i) * TS + s] = gN;

        if (L.valid) {
            if (B.fwd) B.fwd[RI_(RI_ORIG) * B.o_ld + L.gs] = Rf;
            if (B.rev && RI_(RI_REV_IDX) >= 0) B.rev[RI_(RI_REV_IDX) * B.o_ld + L.gs] = Rr;
            if (B.pres_mod && RI_(RI_PRES_IDX) >= 0) B.pres_mod[RI_(RI_PRES_IDX) * B.o_ld + L.gs] = c;
        }
    }
#undef RI_
#undef RD_
}

// ---------------------------------------------------------------- scatter
// eval_spec_rates (rate_subs.py:1297-1542) and every sum over reactions of the
// Jacobian in one pass: tile[target] += nu * V[source] for the terms listed in
// the schedule.  Targets: omega_k, sum_i nu_ki theta_i, P_k, Q_k and the sparse
// block S_kj.  Each wavefront owns its targets and its IL lanes touch distinct
// ones per round, so these are plain LDS read-modify-writes; the codes are read
// with coalesced loads, four rounds at a time.
template <int TS>
PJ_DEV void phase_zero_tile(const DevMech& M, double* V, int tid, int NT)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    for (int t = u; t < M.v.NTILE; t += NU) V[(M.v.TB + t) * TS + s] = 0.0;
}

template <int TS>
PJ_DEV void phase_scatter(const DevMech& M, double* V, int tid, int NT, bool dense_only)
{
    constexpr int IL = 64 / TS;
    const int lane = tid % 64, s = lane % TS, il = lane / TS, w = tid / 64;
    const uint32_t* sc = M.sched + M.sched_off[w] + il;
    const int nr = dense_only ? M.sched_rounds_dense[w] : M.sched_rounds[w];
    double* T = V + M.v.TB * TS + s;
    // The schedule guarantees that the 4 x IL targets of a group are distinct, so the
    // four read-modify-writes of a lane are independent: one round trip to LDS per group.
    // codes are prefetched two groups ahead (their L2 latency exceeds one group's work)
    uint32_t c[4], c1[4], c2[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        c[x] = (nr > 0) ? sc[x * IL] : (4u << 28);
        c1[x] = (nr > 4) ? sc[(4 + x) * IL] : (4u << 28);
    }
    for (int r = 0; r < nr; r += 4) {
        const bool more = r + 8 < nr;
#pragma unroll
        for (int x = 0; x < 4; ++x) c2[x] = more ? sc[(r + 8 + x) * IL] : (4u << 28);
        double v[4], t[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            v[x] = V[(c[x] & 8191u) * TS + s];
            t[x] = T[((c[x] >> 13) & 32767u) * TS];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            // nu code (pj_tables.h): 0..7 are the whole numbers -4..3, 8.. index the mechanism's table of
            // fractional / larger net coefficients
            const unsigned nc = c[x] >> 28;
            double nu = (double)((int)nc - 4);
            if (nc >= 8u) nu = M.nutab[nc - 8u];
            if (nc != 4u) T[((c[x] >> 13) & 32767u) * TS] = t[x] + nu * v[x];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) { c[x] = c1[x]; c1[x] = c2[x]; }
    }
}

// add the partial accumulators of split hub targets to their final slot
template <int TS>
PJ_DEV void phase_fin1(const DevMech& M, double* V, int tid, int NT)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    double* T = V + M.v.TB * TS + s;
    // one item lane per split target; its partials sit in consecutive slots
    for (int q = u; q < M.nfin; q += NU) {
        const int t = M.fin_tgt[q], p0 = M.fin_part[q], cnt = M.fin_cnt[q];
        double acc = T[t * TS];
#pragma unroll 4
        for (int z = 0; z < cnt; ++z) acc += T[(p0 + z) * TS];
        T[t * TS] = acc;
    }
}

// per-state scalars H = sum h_k W_k omega_k, HP, HQ (same with P_k, Q_k), SCP = sum omega_k W_k cp_k,
// SJT = sum h_k W_k (sum_i nu_ki theta_i): products per species (fin2a), then five
// row sums (fin2b); species rates / dydt outputs (rate_subs.py:2171-2335)
template <int TS>
PJ_DEV void phase_fin2a(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp, last = nsp - 1;
    const double* T = V + M.v.TB * TS + s;
    const double* KC = lds_kc<TS>(M, V);
    for (int k = u; k < nsp; k += NU) {
        const double hW = V[(M.v.HW + k) * TS + s];
        const double om = T[(M.v.T_OM + k) * TS];
        // reference quirk (create_jacobian.py:2786-2818): the last species contributes only
        // the d/dT of its last reaction unless sum_last is set
        const double jt = (k == last && !M.sum_last) ? T[M.v.T_JTQ * TS] : T[(M.v.T_JT + k) * TS];
        const double Wk = KC[nsp + k];
        V[(M.v.X + SC_H * nsp + k) * TS + s] = hW * om;
        V[(M.v.X + SC_HP * nsp + k) * TS + s] = hW * T[(M.v.T_P + k) * TS];
        V[(M.v.X + SC_HQ * nsp + k) * TS + s] = hW * T[(M.v.T_Q + k) * TS];
        V[(M.v.X + SC_SCP * nsp + k) * TS + s] = om * Wk * V[(M.v.CP + k) * TS + s];
        V[(M.v.X + SC_SJT * nsp + k) * TS + s] = hW * jt;
        if (L.valid) {
            if (B.spec_rates) B.spec_rates[k * B.o_ld + L.gs] = om;
            if (B.dy && k < last) B.dy[(k + 1) * B.o_ld + L.gs] = om * Wk * L.invrho;
        }
    }
}

template <int TS>
PJ_DEV void phase_fin2b(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp;
    for (int r = u; r < SC_COUNT; r += NU) {
        const double* X = V + (M.v.X + r * nsp) * TS + s;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < nsp; ++k) acc += X[k * TS];
        V[(M.v.SC + r) * TS + s] = acc;
        if (r == SC_H && B.dy && L.valid) B.dy[L.gs] = -acc / (L.rho * L.cpavg);
    }
}

// ---------------------------------------------------------------- output
// Jacobian entries in memory order (create_jacobian.py:2850-2938 species block,
// 2728-2845 d/dT column); every entry of the NSP x NSP block is written.
template <int TS>
PJ_DEV void phase_out_block(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp;
    const double* T = V + M.v.TB * TS + s;
    const double* KC = lds_kc<TS>(M, V);
#pragma unroll 4
    for (int e = u; e < nsp * nsp; e += NU) {
        const int col = e / nsp, row = e - col * nsp;
        const int k = row > 0 ? row - 1 : 0, j = col > 0 ? col - 1 : 0;
        const double Wk = KC[nsp + k];
        const unsigned si = M.smap[k + nsp * j];
        const double Skj = (si != 0xFFFFu) ? T[(M.v.T_S + (int)si) * TS] : 0.0;
        // d/dT column: W_k sum_i nu_ki theta_i ; species block: (W_k/W_j)((1 - w_j) P_k - w_j QN_k + S_kj), Q_k = P_k + QN_k
        const double wj = KC[2 * nsp + j];
        const double blk = (Wk * KC[j]) * (((1.0 - wj) * T[(M.v.T_P + k) * TS] - wj * T[(M.v.T_Q + k) * TS]) + Skj);
        const double val = (col == 0) ? Wk * T[(M.v.T_JT + k) * TS] : blk;
        if (row > 0 && L.valid) B.jac[e * B.j_si + L.gs * B.j_ss] = val;   // row 0: phase_out_energy
    }
}

// energy row (create_jacobian.py:3095-3234) and jac[0] (create_jacobian.py:1853-1905)
template <int TS>
PJ_DEV void phase_out_energy(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int nsp = M.nsp, last = nsp - 1;
    const double* T = V + M.v.TB * TS + s;
    const double H = V[(M.v.SC + SC_H) * TS + s];
    const double icp = 1.0 / L.cpavg;
    // walk the columns from the last item lanes so the long column sums do not
    // share a wavefront round with the first species-block entries
    for (int col = NU - 1 - u; col < nsp; col += NU) {
        double val;
        if (col == 0) {
            val = -(V[(M.v.SC + SC_SCP) * TS + s] - (L.dcp * icp) * H + L.rho * V[(M.v.SC + SC_SJT) * TS + s]) /
                  (L.rho * L.cpavg);
        } else {
            const int j = col - 1;
            const double* KC = lds_kc<TS>(M, V);
            const double spj[4] = {KC[j], 0.0, 0.0, KC[2 * nsp + j]};
            double hs = 0.0;
#pragma unroll 4
            for (int q = M.ecol_ptr[j]; q < M.ecol_ptr[j + 1]; ++q) {
                const uint32_t ks = M.ecol[q];
                hs += V[(M.v.HW + (int)(ks >> 16)) * TS + s] * T[(M.v.T_S + (int)(ks & 0xFFFFu)) * TS];
            }
            const double tot = ((1.0 - spj[3]) * V[(M.v.SC + SC_HP) * TS + s] - spj[3] * V[(M.v.SC + SC_HQ) * TS + s]) + hs;
            val = -tot * spj[0] * icp +
                  (V[(M.v.CP + j) * TS + s] - V[(M.v.CP + last) * TS + s]) * H * L.invrho * icp * icp;
        }
        if (L.valid) B.jac[(nsp * col) * B.j_si + L.gs * B.j_ss] = val;
    }
}

}  // namespace pj
