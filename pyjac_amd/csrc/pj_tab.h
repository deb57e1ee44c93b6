// pj_tab.h -- device side of k_tab / k_tab_fin: the table-driven, state-per-lane Jacobian kernels that need no
// compilation per mechanism (program layout and design: pj_tabprog.h).
//
// Same formulation as pj_rblk.hip / pj_kernel.h (J(k,j) = (W_k/W_j)(P_k - w_j Q_k + S_kj), energy row from the
// same pieces); reference emitters: pyjac/core/rate_subs.py:254-2335, pyjac/core/create_jacobian.py:2189-3298.
// Written against PJ_DEV / PJ_UNIFORM so that tests/emu/emu.cpp can run the same text thread by thread on the host.
#pragma once
#include "pj_kernel.h"
#include "pj_tabprog.h"

#ifndef PJ_UNIFORM
// a value that is the same in every lane of the wavefront, moved to a scalar register so that everything
// derived from it (table addresses, loop bounds) is scalar too
#define PJ_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// Read-only tables are addressed through the constant address space on the device: a load at a wavefront-uniform
// address is then a scalar load (s_load) whatever stores the kernel has issued before -- as plain global pointers
// they turn into vector loads behind the Jacobian stores once the compiler sees a store that might alias them.
#ifndef PJT_CONST
#define PJT_CONST __attribute__((address_space(4)))
#endif

namespace pj {

typedef const PJT_CONST double* tab_cd;
typedef const PJT_CONST int32_t* tab_ci;
struct TabTabs {
    tab_ci gen_sp, I;
    tab_cd sp, plog, sri, cheb, gen_nu;
};

struct TabDev {
    int L, G, B, ZERO, TRASH;
    const int32_t* I;               // per lane group: first word, records, words of the first two records (TabProg::I)
    const double* D;                // record streams (TabProg::D)
    double* scr;                    // [nsp + 1][n]: omega_k of every species, then the last species' d/dT sum
    long scr_ld;
    int dbg;                        // timing experiments (PJ_TAB_DBG): 1 no reaction arithmetic, 2 no accumulation, 4 no output
};

PJ_DEV TabTabs tab_tabs(const DevMech& M, const TabDev& P)
{
    TabTabs X;
    X.gen_sp = (tab_ci)M.gen_sp; X.I = (tab_ci)P.I;
    X.sp = (tab_cd)M.sp; X.plog = (tab_cd)M.plog; X.sri = (tab_cd)M.sri; X.cheb = (tab_cd)M.cheb; X.gen_nu = (tab_cd)M.gen_nu;
    return X;
}

// exp(x) with the arithmetic of the device library's double-precision exp (same reduction and polynomial) minus
// its range selects -- ldexp saturates to 0 / inf by itself (pj_math.h has the two-argument form the compiled
// kernels use)
PJ_DEV double tab_exp(const double x)
{
    const double LOG2E = 0x1.71547652b82fep+0, NLN2H = -0x1.62e42fefa39efp-1, NLN2L = -0x1.abc9e3b39803fp-56;
    const double n = __builtin_rint(x * LOG2E);
    double r = __builtin_fma(n, NLN2H, x);
    r = __builtin_fma(n, NLN2L, r);
    double p = __builtin_fma(0x1.ade156a5dcb37p-26, r, 0x1.28af3fca7ab0cp-22);
    p = __builtin_fma(p, r, 0x1.71dee623fde64p-19);
    p = __builtin_fma(p, r, 0x1.a01997c89e6bp-16);
    p = __builtin_fma(p, r, 0x1.a01a014761f6ep-13);
    p = __builtin_fma(p, r, 0x1.6c16c1852b7bp-10);
    p = __builtin_fma(p, r, 0x1.1111111122322p-7);
    p = __builtin_fma(p, r, 0x1.55555555502a1p-5);
    p = __builtin_fma(p, r, 0x1.5555555555511p-3);
    p = __builtin_fma(p, r, 0x1.000000000000bp-1);
    p = __builtin_fma(r, p, 1.0);
    p = __builtin_fma(r, p, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// per-state scalars of a lane
struct TabLane {
    double T, logT, invT, p, logp, rho, invrho, Wbar, mconc, WR;
    long gs;
};

// what one visit of a reaction yields for the accumulate program
struct TabRx {
    double q, theta, rp, rq, bM, bcol, ckf, ckr;
    double g[TAB_NSLOT];            // molecule-slot values: R0 R1 R2 (c k_f prod others), P0 P1 P2 (-c k_r ...), collider
};

// ---- stage: T, p, Y of this lane's state -> per-state scalars; concentrations -> LDS columns CL[k][L] ----
// (all G groups hold the same L states; each group stores every G-th species' column)
PJ_DEV void tab_stage(const DevMech& M, const TabDev& P, const Batch& B, double* lds, int tid, long wg, TabLane& Ln)
{
    const TabTabs X = tab_tabs(M, P);
    const int L = P.L, nsp = M.nsp, last = nsp - 1;
    const int g = tid / L, lane = tid % L;
    long gs = wg * L + lane;
    if (gs >= B.n) gs = B.n - 1;          // lanes past the end repeat the last state (same values, same addresses)
    Ln.gs = gs;
    const double* y = B.y + gs * B.y_ss;
    const double T = y[0], p = B.pres[gs];
    double sumY = 0.0, sumYW = 0.0;
    for (int k = 0; k < last; ++k) {
        const double Yk = y[(long)(k + 1) * B.y_si];
        sumY += Yk;
        sumYW += Yk * X.sp[k * SPW];
    }
    const double yN = 1.0 - sumY;
    sumYW += yN * X.sp[last * SPW];
    Ln.T = T; Ln.p = p; Ln.logT = log(T); Ln.invT = 1.0 / T; Ln.logp = log(p);
    Ln.Wbar = 1.0 / sumYW;
    Ln.rho = p * Ln.Wbar / (RU_ * T);
    Ln.invrho = 1.0 / Ln.rho;
    Ln.mconc = p / (RU_ * T);
    Ln.WR = Ln.Wbar * Ln.invrho;
    double* CL = lds;
    for (int k = g; k < nsp; k += P.G) {
        const double Yk = (k == last) ? yN : y[(long)(k + 1) * B.y_si];
        CL[k * L + lane] = Ln.rho * Yk * X.sp[k * SPW];
    }
}

// ---- one reaction, every rate form, for one state (reaction index d is wavefront-uniform) ----
#define TAB_C(idx) (((idx) == nsp) ? 1.0 : CL[(idx) * L + lane])
// ri / rd: the reaction's records, effs / effa: its enhanced colliders and efficiencies - 1, kc: its K_c rows --
// all inline in the visit record, which sits in the wavefront's LDS ring (pj_tabprog.h): every lane reads the same
// address (a broadcast), the values arrive in vector registers and only what steers control flow is moved to
// scalar registers
PJ_DEV void tab_reaction(const DevMech& M, const TabTabs& X, const TabLane& Ln, const double* CL, const int L, const int lane,
                         const int32_t* ri, const double* rd, const double* kc, const int32_t* effs, const double* effa,
                         TabRx& R)
{
    const int nsp = M.nsp, last = nsp - 1;
    const int fl = PJ_UNIFORM(ri[RI_FLAGS]);
    const double T = Ln.T, logT = Ln.logT, invT = Ln.invT;
    // ---- forward rate constant and T dln k_f/dT ----
    double lnk, dlnk, kf_jac_ratio = 1.0;
    if (fl & F_PLOG) {
        // rate_subs.py:598-632 (breakpoints compared at their printed value); create_jacobian.py:1687-1850
        tab_cd Pl = X.plog + (long)PJ_UNIFORM(ri[RI_PLOG_PTR]) * PLW;
        const int np = PJ_UNIFORM(ri[RI_PLOG_CNT]);
        lnk = Pl[2] + Pl[3] * logT - Pl[4] * invT;
        dlnk = Pl[3] + Pl[4] * invT;
        for (int q = 1; q < np; ++q) {
            tab_cd r1 = Pl + (q - 1) * PLW;
            tab_cd r2 = Pl + q * PLW;
            const double k1 = r1[2] + r1[3] * logT - r1[4] * invT;
            const double k2 = r2[2] + r2[3] * logT - r2[4] * invT;
            const double f = (Ln.logp - r1[1]) / (r2[1] - r1[1]);
            const bool in = Ln.p > r1[0] && Ln.p <= r2[0];
            lnk = in ? k1 + (k2 - k1) * f : lnk;
            dlnk = in ? r1[3] + r1[4] * invT + ((r2[3] - r1[3]) + (r2[4] - r1[4]) * invT) * f : dlnk;
        }
        {
            tab_cd rn = Pl + (np - 1) * PLW;
            const bool hi = Ln.p > rn[0];
            lnk = hi ? rn[2] + rn[3] * logT - rn[4] * invT : lnk;
            dlnk = hi ? rn[3] + rn[4] * invT : dlnk;
        }
    } else if (fl & F_CHEB) {
        // rate_subs.py:149-251 ('{:.8e}' constants); create_jacobian.py:1532-1684 ('{:.16e}')
        tab_cd C = X.cheb + PJ_UNIFORM(ri[RI_PLOG_PTR]);
        const int cn = PJ_UNIFORM(ri[RI_PLOG_CNT]) >> 8, cm = PJ_UNIFORM(ri[RI_PLOG_CNT]) & 255;
        const double lg10p = Ln.logp * INV_LN10;
        auto cheb_sum = [&](tab_cd c, const double Tred, const double Pred) {
            double kl = 0.0, u0 = 1.0, u1 = Tred;
            for (int a = 0; a < cn; ++a) {
                double acc = c[a * cm] + Pred * c[a * cm + 1];
                double t0 = 1.0, t1 = Pred;
                for (int j = 2; j < cm; ++j) {
                    const double tn = 2.0 * Pred * t1 - t0;
                    acc += c[a * cm + j] * tn;
                    t0 = t1; t1 = tn;
                }
                if (a == 0) kl = acc;
                else if (a == 1) kl += Tred * acc;
                else { const double un = 2.0 * Tred * u1 - u0; kl += acc * un; u0 = u1; u1 = un; }
            }
            return kl;
        };
        lnk = cheb_sum(C + CH_COEF, (2.0 * invT - C[CH_TSUM8]) / C[CH_TSUB8], (2.0 * lg10p - C[CH_PSUM8]) / C[CH_PSUB8]) * LN10;
        const double Tred = (2.0 * invT - C[CH_TSUM16]) / C[CH_TSUB16];
        const double Pred = (2.0 * lg10p - C[CH_PSUM16]) / C[CH_PSUB16];
        // eval_jacob's own k_f for the dR/dY_j terms (get_cheb_rate(write_defns=False), create_jacobian.py:1647-1664)
        kf_jac_ratio = tab_exp(cheb_sum(C + CH_COEF, Tred, Pred) * LN10 - lnk);
        tab_cd c = C + CH_COEF + cn * cm;         // rows i = 1 .. cn-1 of i * c_ij
        double U = 0.0, w0 = 1.0, w1 = 2.0 * Tred;
        for (int a = 1; a < cn; ++a) {
            double acc = c[(a - 1) * cm] + Pred * c[(a - 1) * cm + 1];
            double t0 = 1.0, t1 = Pred;
            for (int j = 2; j < cm; ++j) {
                const double tn = 2.0 * Pred * t1 - t0;
                acc += c[(a - 1) * cm + j] * tn;
                t0 = t1; t1 = tn;
            }
            if (a == 1) U = acc;
            else if (a == 2) U += 2.0 * Tred * acc;
            else { const double wn = 2.0 * Tred * w1 - w0; U += acc * wn; w0 = w1; w1 = wn; }
        }
        dlnk = U * C[CH_DFAC] * invT;
    } else {
        lnk = rd[RD_LNA] + rd[RD_B] * logT - rd[RD_TA] * invT;
        dlnk = rd[RD_B] + rd[RD_TA] * invT;
    }
    const double kf = rd[RD_SGN] * tab_exp(lnk);
    // ---- equilibrium constant: pre-summed NASA groups (rate_subs.py:660-809); both ranges are evaluated with
    //      scalar coefficients and the lane picks one (the range is a per-state property) ----
    double kr = 0.0, TdlnKc = 0.0;
    if (fl & F_REV) {
        double lnKc = rd[RD_LNPREF];
        const double* g = kc;
        const double T2 = T * T, T3 = T2 * T, T4 = T2 * T2;
        const int nkc = PJ_UNIFORM(ri[RI_KC_CNT]);
        for (int c = 0; c < nkc; ++c, g += KCW) {
            // the range is a per-state property: each lane reads its own row of the pair (two addresses per
            // wavefront at most: no bank conflicts)
            const double* a = (T <= g[0]) ? g + 1 : g + 8;
            lnKc += a[0] + a[1] * logT + a[2] * T + a[3] * T2 + a[4] * T3 + a[5] * T4 - a[6] * invT;
            TdlnKc += a[1] + a[2] * T + 2.0 * a[3] * T2 + 3.0 * a[4] * T3 + 4.0 * a[5] * T4 + a[6] * invT;
        }
        kr = kf * tab_exp(-lnKc);
    }
    // ---- concentration products ----
    const int r0 = ri[RI_R0], r1 = ri[RI_R1], r2 = ri[RI_R2], p0 = ri[RI_P0], p1 = ri[RI_P1], p2 = ri[RI_P2];
    const double cr0 = TAB_C(r0), cr1 = TAB_C(r1), cr2 = TAB_C(r2);
    const double cp0 = TAB_C(p0), cp1 = TAB_C(p1), cp2 = TAB_C(p2);
    double prodr = cr0 * cr1 * cr2, prodp = cp0 * cp1 * cp2;
    const int gp0 = PJ_UNIFORM(ri[RI_GEN_PTR]), gnr = PJ_UNIFORM(ri[RI_GEN_NR]), gnp = PJ_UNIFORM(ri[RI_GEN_NP]);
    if (fl & F_GEN) {
        for (int f = 0; f < gnr; ++f) prodr *= pj_cpow(TAB_C(X.gen_sp[gp0 + f]), X.gen_nu[gp0 + f]);
        for (int f = 0; f < gnp; ++f) prodp *= pj_cpow(TAB_C(X.gen_sp[gp0 + gnr + f]), X.gen_nu[gp0 + gnr + f]);
    }
    const double Rf = kf * prodr, Rr = kr * prodp, Rn = Rf - Rr;
    // ---- pressure modification (rate_subs.py:879-1294; create_jacobian.py:953-1294) ----
    double c = 1.0, lead = 0.0, a_extra = 0.0, bM = 0.0, bcol = 0.0;
    if (fl & (F_THD | F_PDEP)) {
        double Mc = Ln.mconc;
        const int ne = PJ_UNIFORM(ri[RI_EFF_CNT]);
        for (int e = 0; e < ne; ++e) Mc += effa[e] * CL[effs[e] * L + lane];
        if (fl & F_THD) {
            c = Mc;
            lead = -c * Rn * invT;
            if (fl & F_EFFTYPE) { bM = Rn; a_extra = c * Rn; }
        } else {
            const int col = ri[RI_COLLIDER];
            const double conc_temp = (col >= 0) ? CL[col * L + lane] : Mc;
            const double e0T = rd[RD_E0] * invT;
            const double k0kinf = tab_exp(rd[RD_LNAR] + rd[RD_B0] * logT - e0T);
            const double Pr = conc_temp * k0kinf;
            const double i1Pr = 1.0 / (1.0 + Pr);
            double F = 1.0, extra = 0.0, Xtroe = 0.0;
            if (fl & F_TROE) {
                const double ta = rd[RD_TRA];
                const double e3 = tab_exp(-T / rd[RD_T3]), e1 = tab_exp(-T / rd[RD_T1]);
                double Fcent = (1.0 - ta) * e3 + ta * e1;
                double dF = -((1.0 - ta) / rd[RD_T3]) * e3 - (ta / rd[RD_T1]) * e1;
                if (fl & F_TROE4) {
                    const double e2 = tab_exp(-rd[RD_T2] * invT);
                    Fcent += e2;
                    dF += rd[RD_T2] * invT * invT * e2;
                }
                const double lF = log(fmax(Fcent, 1.0e-300));
                const double lgF = lF * INV_LN10;
                const double lgPr = log(fmax(Pr, 1.0e-300)) * INV_LN10;
                const double At = lgPr - 0.67 * lgF - 0.4;
                const double Bt = 0.806 - 1.1762 * lgF - 0.14 * lgPr;
                const double iB = 1.0 / Bt;
                const double iden = 1.0 / (1.0 + At * At * iB * iB);
                F = tab_exp(lF * iden);
                const double lnF_AB = 2.0 * lF * At * iB * iB * iB * iden * iden;
                const double iFc = 1.0 / Fcent;
                Xtroe = lnF_AB * (INV_LN10 * Bt + (0.14 * INV_LN10) * At);
                extra = (iFc * iden - lnF_AB * (-(0.67 * INV_LN10) * Bt + (1.1762 * INV_LN10) * At) * iFc) * dF -
                        Xtroe * (rd[RD_B0] + e0T - 1.0) * invT;
            }
            if (fl & F_SRI) {
                tab_cd Q = X.sri + (long)PJ_UNIFORM(ri[RI_PLOG_PTR]) * SRW;
                const double lgPr = log(fmax(Pr, 1.0e-300)) * INV_LN10;
                const double Xs = 1.0 / (1.0 + lgPr * lgPr);
                const double S6 = Q[SR_A6] * tab_exp(-Q[SR_B6] * invT) + tab_exp(-T / Q[SR_C6]);
                F = tab_exp(Xs * log(S6));
                if (Q[SR_USE_DE] != 0.0) F *= Q[SR_D8] * tab_exp(Q[SR_E6] * logT);
                const double S4 = Q[SR_A4] * tab_exp(-Q[SR_B4] * invT) + tab_exp(-T / Q[SR_C4]);
                const double C2 = 0.8685889638065035;        // '{:.16}'.format(2 / ln 10)
                Xtroe = Xs * Xs * C2 * lgPr * log(S4);
                const double eb = tab_exp(-Q[SR_B16] * invT), ec = tab_exp(-T / Q[SR_C16]);
                const double S16 = Q[SR_A16] * eb + ec;
                const double dS = (Q[SR_AB16] * invT * invT) * eb - Q[SR_INVC16] * ec;
                extra = Xs * (dS / S16 - Xs * C2 * lgPr * (rd[RD_B0] + e0T - 1.0) * log(S16) * invT) + Q[SR_E16] * invT;
            }
            double dpr = (rd[RD_B04] + e0T - 1.0) * invT * i1Pr;
            double X;
            if (fl & F_LOW) { c = F * Pr * i1Pr; X = i1Pr - Xtroe; }
            else { c = F * i1Pr; X = -Pr * i1Pr - Xtroe; dpr = -Pr * dpr; }
            lead = c * (dpr + extra) * Rn;
            if (fl & (F_EFFTYPE | F_COLLIDER)) {
                const double pmt = X * Rn;
                a_extra = c * pmt;
                const double bb = pmt * k0kinf * F * i1Pr;
                if (fl & F_COLLIDER) bcol = bb; else bM = bb;
            }
        }
    }
    // ---- d/dT and the dense-in-j scalars (create_jacobian.py:1398-1529, 127-269) ----
    const double nr = rd[RD_NR], np_ = rd[RD_NP];
    double el = Rn * dlnk + Rf * (1.0 - nr);
    if (fl & F_REV) el -= Rr * ((1.0 - np_) - TdlnKc);
    R.theta = (fl & F_NO_DT) ? 0.0 : (lead + c * invT * el) * Ln.invrho;
    // q - a with a = c (nr R_f - np R_r) + a_extra, written so that nothing cancels when nr or np is 1
    R.q = c * Rn;
    R.rp = Ln.WR * (c * ((1.0 - nr) * Rf - ((fl & F_REV) ? (1.0 - np_) * Rr : 0.0)) - a_extra) + bM;
    R.bM = bM; R.bcol = bcol;
    R.ckf = c * kf * kf_jac_ratio;
    R.ckr = c * kr * kf_jac_ratio;
    // ---- molecule-slot values and the last species' share gN ----
    R.g[0] = R.ckf * (cr1 * cr2); R.g[1] = R.ckf * (cr0 * cr2); R.g[2] = R.ckf * (cr0 * cr1);
    R.g[3] = -R.ckr * (cp1 * cp2); R.g[4] = -R.ckr * (cp0 * cp2); R.g[5] = -R.ckr * (cp0 * cp1);
    R.g[6] = bcol;
    double gN = bM * rd[RD_ANM1];
    gN += (r0 == last ? R.g[0] : 0.0) + (r1 == last ? R.g[1] : 0.0) + (r2 == last ? R.g[2] : 0.0);
    if (fl & F_REV) gN += (p0 == last ? R.g[3] : 0.0) + (p1 == last ? R.g[4] : 0.0) + (p2 == last ? R.g[5] : 0.0);
    if ((fl & F_COLLIDER) && ri[RI_COLLIDER] == last) gN += bcol;
    if (fl & F_GEN) {
        for (int side = 0; side < ((fl & F_REV) ? 2 : 1); ++side) {
            const int f0 = gp0 + side * gnr, nf = side ? gnp : gnr;
            for (int f = 0; f < nf; ++f) {
                if (X.gen_sp[f0 + f] != last) continue;
                const double nuf = X.gen_nu[f0 + f];
                double gv = (side ? -R.ckr : R.ckf) * nuf;
                if (nuf - 1.0 > 0.0) gv *= pj_cpow(CL[last * L + lane], nuf - 1.0);
                for (int h = 0; h < nf; ++h) if (h != f) gv *= pj_cpow(TAB_C(X.gen_sp[f0 + h]), X.gen_nu[f0 + h]);
                gN += gv;
            }
        }
    }
    R.rq = gN;      // (the blocks accumulate QN_k = sum nu gN; Q_k = P_k + QN_k: pj_rblk.hip, near_last())
}

// value of general-stoichiometry factor f (0-based: reactant factors, then product factors) of the reaction:
// c k nu C^(nu-1) prod_others, the power of C itself only "if (nu - 1) > 0" (create_jacobian.py:400-448)
PJ_DEV double tab_gen_value(const DevMech& M, const TabTabs& X, const double* CL, const int L, const int lane, const int32_t* ri,
                            const int f, const TabRx& R)
{
    const int nsp = M.nsp;
    const int gp0 = PJ_UNIFORM(ri[RI_GEN_PTR]), gnr = PJ_UNIFORM(ri[RI_GEN_NR]), gnp = PJ_UNIFORM(ri[RI_GEN_NP]);
    const bool prod = f >= gnr;
    const int f0 = prod ? gp0 + gnr : gp0, nf = prod ? gnp : gnr, fl_ = prod ? f - gnr : f;
    const double nuf = X.gen_nu[f0 + fl_];
    double gv = (prod ? -R.ckr : R.ckf) * nuf;
    if (nuf - 1.0 > 0.0) gv *= pj_cpow(TAB_C(X.gen_sp[f0 + fl_]), nuf - 1.0);
    for (int h = 0; h < nf; ++h) if (h != fl_) gv *= pj_cpow(TAB_C(X.gen_sp[f0 + h]), X.gen_nu[f0 + h]);
    return gv;
}
#undef TAB_C

// ---- the record stream of this thread's lane group (pj_tabprog.h), through the wavefront's LDS ring ----
// A dependent scalar load costs ~0.9 us here (the stream is read once per workgroup and misses the scalar cache; its
// counter, lgkmcnt, is the one the LDS reads wait on), so the program does not come through the scalar unit at all:
// the 64 lanes of a wavefront copy record r + 2 into the ring slot that record r leaves (512-byte vector loads
// issued when r starts, written to LDS when r is done), and every field is an LDS read at a uniform address.
#ifndef PJT_FETCH_ALL
#define PJT_FETCH_ALL 0         // 1 (host emulation, one thread at a time): a thread copies whole records itself
#endif
PJ_DEV void tab_blocks(const DevMech& M, const TabDev& P, const Batch& B, double* lds, int tid, const TabLane& Ln)
{
    const TabTabs X = tab_tabs(M, P);
    const int L = P.L, nsp = M.nsp, last = nsp - 1;
    const int g = PJ_UNIFORM(tid / L), lane = tid % L, wave = PJ_UNIFORM(tid / 64), wl = tid % 64;
    const double* CL = lds;
    double* ACC = lds + (long)nsp * L + (long)g * P.B * L + lane;       // this lane's column of the group's slots
    double* RING = lds + (long)nsp * L + 256L * P.B + (long)(PJT_FETCH_ALL ? tid : wave) * TAB_RING_WORDS;
#define A_(slot) ACC[(long)(slot) * L]
    const long gs = Ln.gs;
    A_(P.ZERO) = 0.0;
    A_(P.TRASH) = 0.0;
    const int nrec = X.I[4 * g + 1];
    const double* S = P.D + X.I[4 * g];
    // Ring of three slots: record r is in slot r % 3, record r + 1 in the next one; record r + 2 is in flight in
    // registers (requested when record r - 1 started, written to its slot when record r - 1 ... r ends), record
    // r + 3 is requested when record r starts: two requests are outstanding, i.e. a load has two records' time.
    long pos2 = 0;                  // word of record r + 2
    double p0 = 0.0, p1 = 0.0, p2 = 0.0;        // record r + 2 (requested one record ago)
    int pn = 0;                     // its words
    {
        const int nw0 = X.I[4 * g + 2], nw1 = X.I[4 * g + 3];
#if PJT_FETCH_ALL
        for (int w = 0; w < nw0; ++w) RING[w] = S[w];
        for (int w = 0; w < nw1; ++w) RING[TAB_RSZ + w] = S[nw0 + w];
#else
        for (int w = wl; w < nw0; w += 64) RING[w] = S[w];
        for (int w = wl; w < nw1; w += 64) RING[TAB_RSZ + w] = S[nw0 + w];
#endif
        pos2 = nw0 + nw1;
        // record 2: its size is in record 0's header (global copy)
        const int32_t* h0 = (const int32_t*)S;
        pn = nrec > 2 ? PJ_UNIFORM(h0[3]) : 0;
#if !PJT_FETCH_ALL
        if (pn > 0) p0 = S[pos2 + wl];
        if (pn > 64) p1 = S[pos2 + 64 + wl];
        if (pn > 128) p2 = S[pos2 + 128 + wl];
#endif
    }
    for (int r = 0; r < nrec; ++r) {
        double* rec = RING + (r % 3) * TAB_RSZ;
        const int32_t* hi = (const int32_t*)rec;
        const int type = PJ_UNIFORM(hi[0]) & 255, flags = PJ_UNIFORM(hi[0]) >> 8, a1 = PJ_UNIFORM(hi[1]);
        const int nw2 = PJ_UNIFORM(hi[3]), niw = PJ_UNIFORM(hi[4]), nw3 = PJ_UNIFORM(hi[5]);
        (void)nw2;
        // request record r + 3 (it follows record r + 2, whose position and size are known)
        const long pos3 = pos2 + pn;
        const double* nx = S + pos3;
        double f0 = 0.0, f1 = 0.0, f2 = 0.0;
#if !PJT_FETCH_ALL
        if (nw3 > 0) f0 = nx[wl];
        if (nw3 > 64) f1 = nx[64 + wl];
        if (nw3 > 128) f2 = nx[128 + wl];
#endif
        const int32_t* vi = hi + 6;
        const double* vd = rec + 3 + niw;
        if (type == TAB_T_VISIT) {
            const int nhit = a1;
            const int32_t* ri = vi;
            const int fl = PJ_UNIFORM(ri[RI_FLAGS]), necnt = PJ_UNIFORM(ri[RI_EFF_CNT]);
            const int kcw = (fl & F_REV) ? PJ_UNIFORM(ri[RI_KC_CNT]) * KCW : 0;
            TabRx R;
            if (P.dbg & 1) { R.q = R.theta = R.rp = R.rq = R.bM = R.bcol = R.ckf = R.ckr = Ln.T; for (int t = 0; t < TAB_NSLOT; ++t) R.g[t] = Ln.p; }
            else tab_reaction(M, X, Ln, CL, L, lane, ri, vd, vd + RDW, ri + RIW, vd + RDW + kcw, R);
            vi += RIW + necnt;
            vd += RDW + kcw + necnt;
            for (int h = 0; h < nhit; ++h) {
                const int neff = PJ_UNIFORM(vi[1 + TAB_NSLOT]), ngen = PJ_UNIFORM(vi[2 + TAB_NSLOT]), hfl = PJ_UNIFORM(vi[3 + TAB_NSLOT]);
                if (P.dbg & 2) { A_(P.TRASH) += R.q + R.rp + R.rq + R.theta + R.g[0] + R.g[1] + R.g[2] + R.g[3] + R.g[4] + R.g[5] + R.g[6]; vi += TAB_HIT_I + neff + 2 * ngen; vd += TAB_HIT_D + neff + ngen; continue; }
                const int base = vi[0];
                const double nu = vd[0];
                {   // dense sums and the reactant slots: read, update, write back (the slots of one batch are
                    // distinct, or TRASH)
                    double a0 = A_(base), a1_ = A_(base + 1), a2 = A_(base + 2), a3 = A_(base + 3);
                    double s0 = A_(vi[1]), s1 = A_(vi[2]), s2 = A_(vi[3]);
                    a0 += nu * R.q; a1_ += nu * R.rp; a2 += nu * R.rq; a3 += nu * R.theta;
                    s0 += vd[1] * R.g[0]; s1 += vd[2] * R.g[1]; s2 += vd[3] * R.g[2];
                    A_(base) = a0; A_(base + 1) = a1_; A_(base + 2) = a2; A_(base + 3) = a3;
                    A_(vi[1]) = s0; A_(vi[2]) = s1; A_(vi[3]) = s2;
                    // reference quirk (create_jacobian.py:2786-2818): J_nplusone is assigned, not accumulated
                    if (hfl & 1) A_(base + 4) = nu * R.theta;
                }
                if (fl & F_REV) {
                    double s3 = A_(vi[4]), s4 = A_(vi[5]), s5 = A_(vi[6]);
                    s3 += vd[4] * R.g[3]; s4 += vd[5] * R.g[4]; s5 += vd[6] * R.g[5];
                    A_(vi[4]) = s3; A_(vi[5]) = s4; A_(vi[6]) = s5;
                }
                if (fl & F_COLLIDER) A_(vi[7]) += vd[7] * R.g[6];
                vi += TAB_HIT_I; vd += TAB_HIT_D;
                for (int e = 0; e < neff; ++e) A_(vi[e]) += vd[e] * R.bM;
                vi += neff; vd += neff;
                for (int e = 0; e < ngen; ++e)
                    A_(vi[2 * e]) += vd[e] * tab_gen_value(M, X, CL, L, lane, ri, PJ_UNIFORM(vi[2 * e + 1]), R);
                vi += 2 * ngen; vd += ngen;
            }
        } else if (type == TAB_T_BEGIN) {
            for (int s_ = 0; s_ < a1; ++s_) A_(s_) = 0.0;
        } else if (!(P.dbg & 4)) {
            // ---- output record: a row of the block (or a column part of it) ----
            const int k = a1, first = flags & 1, base = PJ_UNIFORM(vi[0]), ne = PJ_UNIFORM(vi[1]);
            const int32_t* en = vi + 2;
            const double* ed = vd;
            const double om = A_(base), Pk = A_(base + 1), Qk = A_(base + 2), JT = A_(base + 3), JTQ = A_(base + 4);
            const double Wk = X.sp[k * SPW + 1];
            const long cs = (long)nsp * B.j_si;
            if (k < last) {
                // (Qk holds QN_k.  (1 / W_j - 1 / W_N) W_k P_k - W_k QN_k / W_N + (1 / W_j) W_k S_kj: for an isomer of the last
                // species the first coefficient is exactly zero, as the reference's a_i (1 - W_j / W_N) is)
                const double iWN = X.sp[last * SPW];
                const double WP = Wk * Pk, WCN = (Wk * iWN) * Qk;
                double* Jr = B.jac + gs * B.j_ss + (long)(k + 1) * B.j_si;
                if (first) {
                    Jr[0] = Wk * JT;                                   // d/dT column (create_jacobian.py:2786-2818)
                    P.scr[(long)k * P.scr_ld + gs] = om;
                }
                // TAB_EB entries at a time
                for (int e = 0; e < ne; e += TAB_EB) {
                    int w[TAB_EB];
                    double sv[TAB_EB];
                    for (int u = 0; u < TAB_EB; ++u) w[u] = en[e + u];
                    for (int u = 0; u < TAB_EB; ++u) sv[u] = A_(w[u] >> 16);
                    for (int u = 0; u < TAB_EB; ++u) Jr[cs * ((w[u] & 0xFFFF) + 1)] = ed[e + u] * (Wk * sv[u]) + ((ed[e + u] - iWN) * WP - WCN);
                }
            } else {
                // the last species has no row of its own: its terms open the energy row's column sums
                tab_cd sp = X.sp + (long)k * SPW;
                const double T = Ln.T;
                const bool lo = T <= sp[2];
                double a[6];
                for (int c = 0; c < 6; ++c) a[c] = lo ? sp[4 + c] : sp[11 + c];
                const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                         T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
                double* J0 = B.jac + gs * B.j_ss;
                if (first) {
                    P.scr[(long)k * P.scr_ld + gs] = om;
                    P.scr[(long)nsp * P.scr_ld + gs] = M.sum_last ? JT : JTQ;
                }
                for (int e = 0; e < ne; ++e) J0[cs * ((en[e] & 0xFFFF) + 1)] = hW * (((1.0 - ed[e]) * Pk - ed[e] * Qk) + A_(en[e] >> 16));
            }
        }
        // record r + 2 (requested a record ago) lands in the slot after record r + 1's; the request made above
        // stays in flight for another record
        {
            double* dst = RING + ((r + 2) % 3) * TAB_RSZ;
#if PJT_FETCH_ALL
            for (int w = 0; w < pn; ++w) dst[w] = S[pos2 + w];
#else
            if (pn > 0) dst[wl] = p0;
            if (pn > 64) dst[64 + wl] = p1;
            if (pn > 128) dst[128 + wl] = p2;
#endif
        }
        p0 = f0; p1 = f1; p2 = f2; pn = nw3; pos2 = pos3;
    }
#undef A_
}

// ---- k_tab_fin: the energy row from the finished species rows ----
// sum_k hW_k (P_k - w_j Q_k + S_kj) = sum_{k < N} (hW_k W_j / W_k) J(k, j) + (last species' terms, left in row 0 by
// tab_blocks).  A workgroup of 256 threads takes 64 states x 4 column lanes; hW_k / W_k per state in LDS.
// (rate_subs.py:2171-2335 dT/dt; create_jacobian.py:2940-3268 completion)
PJ_DEV void tab_fin_stage(const DevMech& M, const TabDev& P, const Batch& B, double* lds, int tid, long wg)
{
    const TabTabs X = tab_tabs(M, P);
    const int nsp = M.nsp, last = nsp - 1;
    const int s = tid % 64, c = tid / 64;
    long gs = wg * 64 + s;
    if (gs >= B.n) gs = B.n - 1;
    const double* y = B.y + gs * B.y_ss;
    const double T = y[0];
    double* HWK = lds;                       // [nsp][64]: hW_k / W_k (k < last), hW_last
    double* PART = lds + (long)nsp * 64;     // [4][7][64] partial sums
    double sumY = 0.0, sumYW = 0.0;
    for (int k = 0; k < last; ++k) { const double Yk = y[(long)(k + 1) * B.y_si]; sumY += Yk; sumYW += Yk * X.sp[k * SPW]; }
    const double yN = 1.0 - sumY;
    double cpa = 0.0, dcpa = 0.0, H = 0.0, SCP = 0.0, SJT = 0.0, cpN = 0.0;
    for (int k = c; k < nsp; k += 4) {
        tab_cd sp = X.sp + (long)k * SPW;
        const bool lo = T <= sp[2];
        double a[6];
        for (int q = 0; q < 6; ++q) a[q] = lo ? sp[4 + q] : sp[11 + q];
        const double hW = RU_ * (a[5] + T * (a[0] + T * (a[1] * (1.0 / 2.0) + T * (a[2] * (1.0 / 3.0) +
                                 T * (a[3] * (1.0 / 4.0) + a[4] * (1.0 / 5.0) * T)))));
        const double cpm = a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T)));
        const double dcpm = a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T));
        const double Yk = (k == last) ? yN : y[(long)(k + 1) * B.y_si];
        const double RW = RU_ * sp[0];
        cpa += Yk * RW * cpm;
        dcpa += Yk * RW * dcpm;
        const double om = P.scr[(long)k * P.scr_ld + gs];
        H += hW * om;
        SCP += om * sp[1] * (RW * cpm);
        if (k < last) {
            HWK[k * 64 + s] = hW * sp[0];
            SJT += (hW * sp[0]) * B.jac[gs * B.j_ss + (long)(k + 1) * B.j_si];     // hW_k JT_k = (hW_k / W_k) J(k, T)
        } else {
            HWK[k * 64 + s] = hW;
            SJT += hW * P.scr[(long)nsp * P.scr_ld + gs];
            cpN = RW * cpm;
        }
    }
    double* pp = PART + (long)c * 7 * 64 + s;
    pp[0] = cpa; pp[64] = dcpa; pp[128] = H; pp[192] = SCP; pp[256] = SJT; pp[320] = cpN; pp[384] = sumYW + yN * X.sp[last * SPW];
}

PJ_DEV void tab_fin_cols(const DevMech& M, const TabDev& P, const Batch& B, const double* lds, int tid, long wg)
{
    const TabTabs X = tab_tabs(M, P);
    const int nsp = M.nsp, last = nsp - 1;
    const int s = tid % 64, c = tid / 64;
    long gs = wg * 64 + s;
    if (gs >= B.n) gs = B.n - 1;
    const double* HWK = lds;
    const double* PART = lds + (long)nsp * 64;
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 4; ++q)
        for (int x = 0; x < 6; ++x) sum[x] += PART[((long)q * 7 + x) * 64 + s];
    const double cpavg = sum[0], dcpavg = sum[1], H = sum[2], SCP = sum[3], SJT = sum[4], cpN = sum[5];
    const double sumYW = PART[6 * 64 + s];          // every column lane computed the same full sum
    const double T = B.y[gs * B.y_ss], p = B.pres[gs];
    const double rho = p / (sumYW * RU_ * T), invrho = 1.0 / rho, icp = 1.0 / cpavg;
    double* J = B.jac + gs * B.j_ss;
    if (wg * 64 + s >= B.n) return;       // a lane past the end must not redo the read-modify-write of row 0
    if (c == 0) J[0] = -(SCP - (dcpavg * icp) * H + rho * SJT) / (rho * cpavg);
    for (int j = c; j < last; j += 4) {
        tab_cd sp = X.sp + (long)j * SPW;
        const double Wj = sp[1];
        double N = 0.0;
        double* col = J + (long)nsp * (j + 1) * B.j_si;
        for (int k = 0; k < last; ++k) N += HWK[k * 64 + s] * col[(long)(k + 1) * B.j_si];
        N = N * Wj + col[0];
        const bool lo = T <= sp[2];
        double a[5];
        for (int q = 0; q < 5; ++q) a[q] = lo ? sp[4 + q] : sp[11 + q];
        const double cpj = (RU_ * sp[0]) * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
        col[0] = -N * sp[0] * icp + (cpj - cpN) * H * invrho * icp * icp;
    }
}

}  // namespace pj
