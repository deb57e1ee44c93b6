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

struct DevMech {
    int nsp, nrxn, ng, ne, nv;
    int lastq_rxn, sum_last;
    VMap v;
    const double* sp;
    const int32_t* ri;
    const double* rd;
    const int32_t* eff_sp;
    const double* eff_am1;
    const double* kcg;
    const double* plog;
    const int32_t* net_sp;
    const double* net_nu;
    const int32_t* sp_ptr;
    const int32_t* sp_rxn;
    const double* sp_nu;
    const uint32_t* prog;      // gather programs (global copy)
    int prog_words, p4en, p4c, p3en, p3c;
    int prog_in_lds;           // small programs are staged into LDS once per workgroup
};

// LDS layout of a workgroup (doubles): V[nv][TS] | RED[NT] | program (32-bit words)
template <int TS>
PJ_DEV const uint32_t* lds_prog(const DevMech& M, const double* V, int NT)
{
    return M.prog_in_lds ? reinterpret_cast<const uint32_t*>(V + (size_t)M.nv * TS + NT) : M.prog;
}

// copy the gather programs into LDS (once per workgroup)
template <int TS>
PJ_DEV void stage_prog(const DevMech& M, double* V, int tid, int NT)
{
    if (!M.prog_in_lds) return;
    uint32_t* dst = reinterpret_cast<uint32_t*>(V + (size_t)M.nv * TS + NT);
    for (int w = tid; w < M.prog_words; w += NT) dst[w] = M.prog[w];
}

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

struct Lane {
    double T, logT, invT, p, logp, rho, invrho, Wbar, m, yN;
    double cpavg, dcp, H, scp;     // per-state sums (phase 0b / 3c)
    long gs;
    int valid;
};

#ifndef PJ_WAVE_SYNC
// LDS hand-off between the lanes of ONE wavefront: LDS operations of a wave
// execute in program order, so only the compiler has to be kept from
// reordering the write and the dependent read.
#define PJ_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#endif

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
    double sumY = 0.0, sumYW = 0.0, cpa = 0.0, dcp = 0.0;
    for (int k = 0; k < last; ++k) {
        const double Yk = V[(M.v.C + k) * TS + s];
        sumY += Yk;
        sumYW += Yk * M.sp[k * SPW + 0];
        cpa += V[(M.v.YC + k) * TS + s];
        dcp += V[(M.v.YD + k) * TS + s];
    }
    const double yN = 1.0 - sumY;
    sumYW += yN * M.sp[last * SPW + 0];
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
        const double Ck = L.rho * Yk * M.sp[k * SPW + 0];
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
        const int32_t* ri = M.ri + i * RIW;
        const double* rd = M.rd + i * RDW;
        const int fl = ri[RI_FLAGS];

        // ---- forward rate constant and T d(ln kf)/dT ----
        double lnk, dlnk;
        if (fl & F_PLOG) {
            // rate_subs.py:598-632; create_jacobian.py:1687-1850
            const double* P = M.plog + ri[RI_PLOG_PTR] * PLW;
            const int np = ri[RI_PLOG_CNT];
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
        } else {
            lnk = rd[RD_LNA] + rd[RD_B] * logT - rd[RD_TA] * invT;
            dlnk = rd[RD_B] + rd[RD_TA] * invT;
        }
        const double kf = rd[RD_SGN] * exp(lnk);

        // ---- equilibrium constant (pre-summed NASA groups, rate_subs.py:660-809) ----
        double kr = 0.0, TdlnKc = 0.0;
        if (fl & F_REV) {
            double lnKc = rd[RD_LNPREF];
            const double* g = M.kcg + ri[RI_KC_PTR] * KCW;
            for (int c = 0; c < ri[RI_KC_CNT]; ++c, g += KCW) {
                const double* a = (T <= g[0]) ? g + 1 : g + 8;
                lnKc += a[0] + a[1] * logT + T * (a[2] + T * (a[3] + T * (a[4] + a[5] * T))) - a[6] * invT;
                TdlnKc += a[1] + T * (a[2] + T * (2.0 * a[3] + T * (3.0 * a[4] + 4.0 * a[5] * T))) + a[6] * invT;
            }
            kr = kf * exp(-lnKc);
        }

        // ---- concentration products ----
        const double cr0 = V[ri[RI_R0] * TS + s], cr1 = V[ri[RI_R1] * TS + s], cr2 = V[ri[RI_R2] * TS + s];
        const double cp0 = V[ri[RI_P0] * TS + s], cp1 = V[ri[RI_P1] * TS + s], cp2 = V[ri[RI_P2] * TS + s];
        const double Rf = kf * (cr0 * cr1 * cr2);
        const double Rr = kr * (cp0 * cp1 * cp2);
        const double R = Rf - Rr;

        // ---- pressure modification ----
        double c = 1.0, lead = 0.0, a_extra = 0.0, bM = 0.0, bcol = 0.0;
        if (fl & (F_THD | F_PDEP)) {
            double Mc = L.m;
            for (int e = 0; e < ri[RI_EFF_CNT]; ++e)
                Mc += M.eff_am1[ri[RI_EFF_PTR] + e] * V[M.eff_sp[ri[RI_EFF_PTR] + e] * TS + s];
            if (fl & F_THD) {
                c = Mc;
                lead = -c * R * invT;
                if (fl & F_EFFTYPE) { bM = R; a_extra = c * R; }
            } else {
                const int col = ri[RI_COLLIDER];
                const double conc_temp = (col >= 0) ? V[col * TS + s] : Mc;
                const double e0T = rd[RD_E0] * invT;
                const double k0kinf = exp(rd[RD_LNAR] + rd[RD_B0] * logT - e0T);
                const double Pr = conc_temp * k0kinf;
                const double i1Pr = 1.0 / (1.0 + Pr);
                double F = 1.0, extra = 0.0, Xtroe = 0.0;
                if (fl & F_TROE) {
                    // create_jacobian.py:1066-1111, 1240-1294
                    const double ta = rd[RD_TRA];
                    const double e3 = exp(-T / rd[RD_T3]), e1 = exp(-T / rd[RD_T1]);
                    double Fcent = (1.0 - ta) * e3 + ta * e1;
                    double dF = -((1.0 - ta) / rd[RD_T3]) * e3 - (ta / rd[RD_T1]) * e1;
                    if (fl & F_TROE4) {
                        const double e2 = exp(-rd[RD_T2] * invT);
                        Fcent += e2;
                        dF += rd[RD_T2] * invT * invT * e2;
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
                            Xtroe * (rd[RD_B0] + e0T - 1.0) * invT;
                }
                // get_pdep_dt (create_jacobian.py:1135-1191): beta difference as printed ('%.4e')
                double dpr = (rd[RD_B04] + e0T - 1.0) * invT * i1Pr;
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
        const double nr = rd[RD_NR], np_ = rd[RD_NP];
        double el = R * dlnk + Rf * (1.0 - nr);
        if (fl & F_REV) el -= Rr * ((1.0 - np_) - TdlnKc);
        const double theta = (fl & F_NO_DT) ? 0.0 : (lead + c * invT * el) * L.invrho;

        // ---- dense-in-j scalars (create_jacobian.py:127-269) ----
        const double a = c * (nr * Rf - ((fl & F_REV) ? np_ * Rr : 0.0)) + a_extra;

        // ---- sparse column values g (one per molecule slot) ----
        const double ckf = c * kf, ckr = c * kr;
        int g = M.v.G + ri[RI_GBASE];
        double gN = bM * rd[RD_ANM1];
        #define PJ_GSLOT(spidx, val)                                        \
            if ((spidx) != M.v.ONE) {                                       \
                const double gv = (val);                                    \
                V[g * TS + s] = gv; ++g;                                    \
                if ((spidx) == last) gN += gv;                              \
            }
        PJ_GSLOT(ri[RI_R0], ckf * (cr1 * cr2))
        PJ_GSLOT(ri[RI_R1], ckf * (cr0 * cr2))
        PJ_GSLOT(ri[RI_R2], ckf * (cr0 * cr1))
        if (fl & F_REV) {
            PJ_GSLOT(ri[RI_P0], -ckr * (cp1 * cp2))
            PJ_GSLOT(ri[RI_P1], -ckr * (cp0 * cp2))
            PJ_GSLOT(ri[RI_P2], -ckr * (cp0 * cp1))
        }
        if (fl & F_COLLIDER) { PJ_GSLOT(ri[RI_COLLIDER], bcol) }
        #undef PJ_GSLOT
        if (fl & F_EFFTYPE)      // (alpha_ij - 1) b_i for the enhanced-collider columns
            for (int e = 0; e < ri[RI_EFF_CNT]; ++e) {
                const int es = M.eff_sp[ri[RI_EFF_PTR] + e];
                if (es != last) { V[g * TS + s] = M.eff_am1[ri[RI_EFF_PTR] + e] * bM; ++g; }
            }

        const double q = c * R;
        const double rp = (L.Wbar * L.invrho) * (q - a) + bM;
        V[(M.v.RQ + i) * TS + s] = q;
        V[(M.v.RTH + i) * TS + s] = theta;
        V[(M.v.RP + i) * TS + s] = rp;
        V[(M.v.RQQ + i) * TS + s] = rp + gN;

        if (L.valid) {
            if (B.fwd) B.fwd[ri[RI_ORIG] * B.o_ld + L.gs] = Rf;
            if (B.rev && ri[RI_REV_IDX] >= 0) B.rev[ri[RI_REV_IDX] * B.o_ld + L.gs] = Rr;
            if (B.pres_mod && ri[RI_PRES_IDX] >= 0) B.pres_mod[ri[RI_PRES_IDX] * B.o_ld + L.gs] = c;
        }
    }
}

// ---------------------------------------------------------------- phase 3
// eval_spec_rates (rate_subs.py:1297-1542) and the per-species dense vectors
//   P_k = sum_i nu_ki [(Wbar/rho)(q_i - a_i) + bM_i],  Q_k = P_k + sum_i nu_ki gN_i.
template <int TS>
PJ_DEV void phase3(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    const int last = M.nsp - 1;
    for (int k = u; k < M.nsp; k += NU) {
        double om = 0.0, jt = 0.0, P = 0.0, Q = 0.0, jtq = 0.0;
        const uint32_t* PG = lds_prog<TS>(M, V, NT);
        const uint32_t en = PG[M.p3en + k];
        const uint2* cb = reinterpret_cast<const uint2*>(PG + M.p3c) + (en >> 8);
        for (int b = 0; b < (int)(en & 255u); ++b) {
            const uint2 cw = cb[b];
            const uint32_t c4[4] = {cw.x & 0xffffu, cw.x >> 16, cw.y & 0xffffu, cw.y >> 16};
            #pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int i = (int)(c4[x] >> 3);
                const int inu = (int)(c4[x] & 7u) - 4;
                if (inu == 0) continue;          // padding
                const double nu = (double)inu;
                const double th = V[(M.v.RTH + i) * TS + s];
                om += nu * V[(M.v.RQ + i) * TS + s];
                jt += nu * th;
                P += nu * V[(M.v.RP + i) * TS + s];
                Q += nu * V[(M.v.RQQ + i) * TS + s];
                if (i == M.lastq_rxn) jtq = nu * th;
            }
        }
        // Reference quirk kept for parity (create_jacobian.py:2786-2818): only the
        // last reaction's d/dT of the LAST species reaches jac[0].
        if (k == last && !M.sum_last) jt = jtq;
        const double Wk = M.sp[k * SPW + 1];
        V[(M.v.AP + k) * TS + s] = P;
        V[(M.v.AQ + k) * TS + s] = Q;
        V[(M.v.AJT + k) * TS + s] = jt;
        V[(M.v.AOM + k) * TS + s] = om;
        if (L.valid) {
            if (B.spec_rates) B.spec_rates[k * B.o_ld + L.gs] = om;
            if (B.dy && k < last) B.dy[(k + 1) * B.o_ld + L.gs] = om * Wk * L.invrho;
        }
    }
}

// per-state sums H = sum_k h_k W_k omega_k and SCP = sum_k omega_k W_k cp_k, needed by
// the lanes that finish the energy row (il == 0) and by dT/dt of dydt
// (rate_subs.py:2171-2335)
template <int TS>
PJ_DEV void phase3c(const DevMech& M, const Batch& B, double* V, int tid, int NT, Lane& L)
{
    const int s = tid % TS, il = (tid % 64) / TS, u = tid / TS;
    if (il != 0) return;
    double H = 0.0, scp = 0.0;
    for (int k = 0; k < M.nsp; ++k) {
        const double om = V[(M.v.AOM + k) * TS + s];
        H += V[(M.v.HW + k) * TS + s] * om;
        scp += om * M.sp[k * SPW + 1] * V[(M.v.CP + k) * TS + s];
    }
    L.H = H; L.scp = scp;
    if (u == 0 && B.dy && L.valid) B.dy[L.gs] = -H / (L.rho * L.cpavg);
}

// ---------------------------------------------------------------- phase 4
// Jacobian, one column per wavefront per round (create_jacobian.py:2850-2938
// species block, 3095-3234 energy row, 1853-1905 jac[0]).  Within a wavefront
// the IL = 64/TS item lanes of a state walk the NSP+1 "rows" of the column in
// chunks: row 0 is the energy entry, rows 1..NSP-1 the species rows, row NSP
// the eliminated last species (it enters the energy row only).  Each lane
// keeps its part of  sum_k h_k W_k M_kj  (phase4a) and the row-0 lane adds the
// IL parts up through the RED exchange area (phase4b).  Every entry of the
// NSP x NSP block is written.
template <int TS>
PJ_DEV int phase4_rounds(const DevMech& M, int NT) { return (M.nsp + NT / 64 - 1) / (NT / 64); }

template <int TS>
PJ_DEV void phase4a(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L, int round)
{
    constexpr int IL = 64 / TS;
    const int s = tid % TS, lane = tid % 64, il = lane / TS, w = tid / 64, NW = NT / 64;
    const int nsp = M.nsp, last = nsp - 1;
    const int col = round * NW + w;
    double part = 0.0;
    if (col == 0) {
        // d/dT column (create_jacobian.py:2728-2845): rows W_k * sum_i nu_ki theta_i
        for (int r = (il == 0 ? IL : il); r <= nsp; r += IL) {
            const int k = r - 1;
            const double jt = V[(M.v.AJT + k) * TS + s];
            part += V[(M.v.HW + k) * TS + s] * jt;
            if (k < last && L.valid) B.jac[r * B.j_si + L.gs * B.j_ss] = M.sp[k * SPW + 1] * jt;
        }
    } else if (col < nsp) {
        const int j = col - 1;
        const double wj = M.sp[j * SPW + 3];
        const double iWj = M.sp[j * SPW + 0];
        const uint32_t* PG = lds_prog<TS>(M, V, NT);
        const uint32_t* ep = PG + M.p4en + nsp * j;
        const uint2* cb = reinterpret_cast<const uint2*>(PG + M.p4c);
        for (int r = (il == 0 ? IL : il); r <= nsp; r += IL) {
            const int k = r - 1;
            const uint32_t en = ep[k];
            double sum = V[(M.v.AP + k) * TS + s] - wj * V[(M.v.AQ + k) * TS + s];
            const uint2* c = cb + (en >> 8);
            for (int b = 0; b < (int)(en & 255u); ++b) {
                const uint2 cw = c[b];
                const uint32_t c0 = cw.x & 0xffffu, c1 = cw.x >> 16, c2 = cw.y & 0xffffu, c3 = cw.y >> 16;
                sum += (double)((int)(c0 & 7u) - 4) * V[(c0 >> 3) * TS + s] +
                       (double)((int)(c1 & 7u) - 4) * V[(c1 >> 3) * TS + s] +
                       (double)((int)(c2 & 7u) - 4) * V[(c2 >> 3) * TS + s] +
                       (double)((int)(c3 & 7u) - 4) * V[(c3 >> 3) * TS + s];
            }
            part += V[(M.v.HW + k) * TS + s] * sum;
            if (k < last && L.valid)
                B.jac[(r + nsp * col) * B.j_si + L.gs * B.j_ss] = (M.sp[k * SPW + 1] * iWj) * sum;
        }
    }
    V[M.v.RED * TS + tid] = part;
}

template <int TS>
PJ_DEV void phase4b(const DevMech& M, const Batch& B, double* V, int tid, int NT, const Lane& L, int round)
{
    constexpr int IL = 64 / TS;
    const int s = tid % TS, lane = tid % 64, il = lane / TS, w = tid / 64, NW = NT / 64;
    const int nsp = M.nsp, last = nsp - 1;
    const int col = round * NW + w;
    if (col >= nsp || il != 0) return;
    double tot = 0.0;
    for (int x = 0; x < IL; ++x) tot += V[M.v.RED * TS + w * 64 + x * TS + s];
    double val;
    if (col == 0) {
        val = -(L.scp - (L.dcp / L.cpavg) * L.H + L.rho * tot) / (L.rho * L.cpavg);
    } else {
        const int j = col - 1;
        const double icp = 1.0 / L.cpavg;
        val = -tot * M.sp[j * SPW + 0] * icp +
              (V[(M.v.CP + j) * TS + s] - V[(M.v.CP + last) * TS + s]) * L.H * L.invrho * icp * icp;
    }
    if (L.valid) B.jac[(nsp * col) * B.j_si + L.gs * B.j_ss] = val;
}

}  // namespace pj
