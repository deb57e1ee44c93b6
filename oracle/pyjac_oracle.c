/*
 * pyjac_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the arithmetic pyJac's *generated C* performs
 * (eval_conc -> eval_rxn_rates -> get_rxn_pres_mod -> eval_spec_rates ->
 * dydt / eval_jacob), written table-driven so one binary serves every
 * mechanism.  It keeps pyJac's per-reaction ("verbatim") formulation and its
 * evaluation order, so that it agrees with the reference's generated C to
 * rounding; the HIP product path uses a different (restructured)
 * formulation and is checked AGAINST this file.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library (checker / reported baseline).  The product package
 * pyjac_amd never imports it.
 *
 * Parity pin: checked in this container against the reference's generated
 * code (oracle/build_ref.py -> oracle/_ref/libpyjac_ref_<mech>.so) on the
 * reference's own PaSR fixture and on synthetic mechanisms covering every
 * supported reaction type; see tests/test_oracle_golden.py and the
 * committed golden vectors under tests/golden/.
 *
 * Each function cites the reference emitter (file:line under /root/reference)
 * whose output it follows.  The table blob layout is pyjac_amd/tables.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HDR 96
#define MAGIC 0x314D4A50

enum { F_REV = 1, F_THD = 2, F_PDEP = 4, F_LOW = 8, F_HIGH = 16, F_TROE = 32,
       F_SRI = 64, F_PLOG = 128, F_TROE4 = 256, F_SRI5 = 512, F_HAS_EFF = 1024, F_CHEB = 32768 };
enum { IA_FLAGS, IA_REAC_PTR, IA_REAC_SP, IA_PROD_PTR, IA_PROD_SP, IA_NET_PTR,
       IA_NET_SP, IA_EFF_PTR, IA_EFF_SP, IA_PLOG_PTR, IA_KC_PTR, IA_PDEP_SP,
       IA_REV_IDX, IA_PRES_IDX, IA_SEEN, IA_CHEB_PTR };
enum { DA_MW, DA_TMID, DA_LO, DA_HI, DA_A, DA_B, DA_E, DA_REAC_NU, DA_PROD_NU,
       DA_NET_NU, DA_EFF, DA_PD, DA_TROE, DA_SRI, DA_PLOG, DA_KCG, DA_KCPREF,
       DA_INFS, DA_TROE8, DA_PLOG4, DA_SRIQ, DA_CHEB };

#define RU 8314.4621 /* chem_utilities.py:16, printed '%.8e' = 8.31446210e+03 */

typedef struct pjo_mech {
    int32_t *I;
    double *D;
    int nsp, nrxn, nrev, npres;
    const int32_t *flags, *reac_ptr, *reac_sp, *prod_ptr, *prod_sp, *net_ptr,
        *net_sp, *eff_ptr, *eff_sp, *plog_ptr, *kc_ptr, *pdep_sp, *rev_idx,
        *pres_idx, *seen;
    const double *mw, *tmid, *lo, *hi, *A, *b, *E, *reac_nu, *prod_nu,
        *net_nu, *eff, *pd, *troe, *sri, *plog, *kcg, *kcpref, *infs, *troe8,
        *plog4, *sriq, *cheb;
    const int32_t *cheb_ptr;
} pjo_mech;

pjo_mech *pjo_create(const int32_t *I, long nI, const double *D, long nD)
{
    if (nI < HDR || I[0] != MAGIC || I[1] != 2 || I[12] != nI || I[13] != nD)
        return NULL;
    pjo_mech *m = (pjo_mech *)calloc(1, sizeof(pjo_mech));
    m->I = (int32_t *)malloc(sizeof(int32_t) * nI);
    m->D = (double *)malloc(sizeof(double) * (nD > 0 ? nD : 1));
    memcpy(m->I, I, sizeof(int32_t) * nI);
    memcpy(m->D, D, sizeof(double) * nD);
    m->nsp = I[2]; m->nrxn = I[3]; m->nrev = I[4]; m->npres = I[5];
#define IP(j) (m->I + m->I[16 + (j)])
#define DP(j) (m->D + m->I[48 + (j)])
    m->flags = IP(IA_FLAGS); m->reac_ptr = IP(IA_REAC_PTR); m->reac_sp = IP(IA_REAC_SP);
    m->prod_ptr = IP(IA_PROD_PTR); m->prod_sp = IP(IA_PROD_SP); m->net_ptr = IP(IA_NET_PTR);
    m->net_sp = IP(IA_NET_SP); m->eff_ptr = IP(IA_EFF_PTR); m->eff_sp = IP(IA_EFF_SP);
    m->plog_ptr = IP(IA_PLOG_PTR); m->kc_ptr = IP(IA_KC_PTR); m->pdep_sp = IP(IA_PDEP_SP);
    m->rev_idx = IP(IA_REV_IDX); m->pres_idx = IP(IA_PRES_IDX); m->seen = IP(IA_SEEN);
    m->mw = DP(DA_MW); m->tmid = DP(DA_TMID); m->lo = DP(DA_LO); m->hi = DP(DA_HI);
    m->A = DP(DA_A); m->b = DP(DA_B); m->E = DP(DA_E); m->reac_nu = DP(DA_REAC_NU);
    m->prod_nu = DP(DA_PROD_NU); m->net_nu = DP(DA_NET_NU); m->eff = DP(DA_EFF);
    m->pd = DP(DA_PD); m->troe = DP(DA_TROE); m->sri = DP(DA_SRI); m->plog = DP(DA_PLOG);
    m->kcg = DP(DA_KCG); m->kcpref = DP(DA_KCPREF); m->infs = DP(DA_INFS);
    m->troe8 = DP(DA_TROE8); m->plog4 = DP(DA_PLOG4);
    m->sriq = DP(DA_SRIQ); m->cheb = DP(DA_CHEB); m->cheb_ptr = IP(IA_CHEB_PTR);
    return m;
}

void pjo_destroy(pjo_mech *m)
{
    if (!m) return;
    free(m->I); free(m->D); free(m);
}

int pjo_nsp(const pjo_mech *m) { return m->nsp; }
int pjo_nrxn(const pjo_mech *m) { return m->nrxn; }
int pjo_nrev(const pjo_mech *m) { return m->nrev; }
int pjo_npres(const pjo_mech *m) { return m->npres; }

/* rxn_rate_const, rate_subs.py:27-146: the FORM depends on exact zeros of b, E
 * and on the sign of A. */
static double rate_const(double A, double b, double E, double T, double logT)
{
    if (A > 0) {
        double logA = log(A);
        if (E == 0.0) {
            if (b == 0.0) return A;
            return exp(logA + b * logT);
        }
        if (b == 0.0) return exp(logA - (E / T));
        return exp(logA + b * logT - (E / T));
    }
    /* A < 0 (duplicate reactions): rate_subs.py:108-141 */
    if (E == 0.0) {
        if (b == 0.0) return A;
        if (b == floor(b)) { /* A * T * T ... (no factor at all for b < 0) */
            double k = A;
            for (int i = 0; i < (int)b; ++i) k *= T;
            return k;
        }
        return A * exp(b * logT);
    }
    if (b == 0.0) return A * exp(-(E / T));
    return A * exp(b * logT - (E / T));
}

/* PLOG forward rate constant, rate_subs.py:598-632 (breakpoints compared at
 * their printed '%.4e' value, log(p_i) at full precision). */
static double plog_kf(const pjo_mech *m, int i, double T, double logT, double pres)
{
    int p0 = m->plog_ptr[i], p1 = m->plog_ptr[i + 1];
    const double *P = m->plog, *P4 = m->plog4;
    if (pres <= P4[p0])
        return rate_const(P[4 * p0 + 1], P[4 * p0 + 2], P[4 * p0 + 3], T, logT);
    for (int q = p0; q < p1 - 1; ++q) {
        if (pres > P4[q] && pres <= P4[q + 1]) {
            double k1 = log(rate_const(P[4 * q + 1], P[4 * q + 2], P[4 * q + 3], T, logT));
            double k2 = log(rate_const(P[4 * q + 5], P[4 * q + 6], P[4 * q + 7], T, logT));
            double dl = log(P[4 * q + 4]) - log(P[4 * q]);
            return exp(k1 + (k2 - k1) * (log(pres) - log(P[4 * q])) / dl);
        }
    }
    int q = p1 - 1;
    if (pres > P4[q])
        return rate_const(P[4 * q + 1], P[4 * q + 2], P[4 * q + 3], T, logT);
    return 0.0; /* unreachable for ordered breakpoints */
}

/* Chebyshev record (pyjac_amd/tables.py): n, m, the reduced-variable constants as the rate emitter prints
 * them ('{:.8e}'), as the Jacobian emitter does ('{:.16e}'), -2 ln10 / (1/Tmax - 1/Tmin), n x m coefficients
 * ('{:.8e}'), (n-1) x m coefficients i * c_ij ('{:.16e}'). */
enum { CH_N, CH_M, CH_TSUM8, CH_TSUB8, CH_PSUM8, CH_PSUB8, CH_TSUM16, CH_TSUB16, CH_PSUM16, CH_PSUB16, CH_DFAC, CH_COEF };

/* get_cheb_rate, rate_subs.py:149-251 */
static double cheb_kf(const pjo_mech *m, int i, double T, double pres, int in_jacobian)
{
    const double *C = m->cheb + m->cheb_ptr[i];
    const int n = (int)C[CH_N], mm = (int)C[CH_M];
    const double *c = C + CH_COEF;
    double dot_prod[16];
    /* eval_jacob re-evaluates k_f for the dR/dY_j terms with get_cheb_rate(write_defns=False): the
     * '{:.8e}' coefficients, but the Tred / Pred of write_cheb_rxn_dt ('{:.16e}' constants,
     * create_jacobian.py:1647-1664) */
    double Tred = in_jacobian ? ((2.0 / T) - C[CH_TSUM16]) / C[CH_TSUB16] : ((2.0 / T) - C[CH_TSUM8]) / C[CH_TSUB8];
    double Pred = in_jacobian ? (2.0 * log10(pres) - C[CH_PSUM16]) / C[CH_PSUB16]
                              : (2.0 * log10(pres) - C[CH_PSUM8]) / C[CH_PSUB8];
    double ct[2] = {1, Pred};
    for (int a = 0; a < n; ++a) dot_prod[a] = c[a * mm] + Pred * c[a * mm + 1];
    int upd = 1;
    for (int j = 2; j < mm; ++j) {
        int nw = upd ? 1 : 0, old = upd ? 0 : 1;
        ct[old] = 2 * Pred * ct[nw] - ct[old];
        for (int a = 0; a < n; ++a) dot_prod[a] += c[a * mm + j] * ct[old];
        upd = !upd;
    }
    ct[0] = 1; ct[1] = Tred;
    double kf = dot_prod[0] + Tred * dot_prod[1];
    upd = 1;
    for (int a = 2; a < n; ++a) {
        int nw = upd ? 1 : 0, old = upd ? 0 : 1;
        ct[old] = 2 * Tred * ct[nw] - ct[old];
        kf += dot_prod[a] * ct[old];
        upd = !upd;
    }
    return pow(10.0, kf);
}

/* write_cheb_ut, create_jacobian.py:1532-1607: sum_i i c_ij T_j(Pred) U_{i-1}(Tred) */
static double cheb_ut(const pjo_mech *m, int i, double T, double pres)
{
    const double *C = m->cheb + m->cheb_ptr[i];
    const int n = (int)C[CH_N], mm = (int)C[CH_M];
    const double *c = C + CH_COEF + n * mm;      /* rows i = 1 .. n-1 */
    double dot_prod[16];
    double Tred = ((2.0 / T) - C[CH_TSUM16]) / C[CH_TSUB16];
    double Pred = (2.0 * log10(pres) - C[CH_PSUM16]) / C[CH_PSUB16];
    double ct[2] = {1, Pred};
    for (int a = 1; a < n; ++a) dot_prod[a] = c[(a - 1) * mm] + Pred * c[(a - 1) * mm + 1];
    int upd = 1;
    for (int j = 2; j < mm; ++j) {
        int nw = upd ? 1 : 0, old = upd ? 0 : 1;
        ct[old] = 2 * Pred * ct[nw] - ct[old];
        for (int a = 1; a < n; ++a) dot_prod[a] += c[(a - 1) * mm + j] * ct[old];
        upd = !upd;
    }
    ct[0] = 1.0; ct[1] = 2.0 * Tred;
    double kf = dot_prod[1] + 2.0 * Tred * dot_prod[2];
    upd = 1;
    for (int a = 3; a < n; ++a) {
        int nw = upd ? 1 : 0, old = upd ? 0 : 1;
        ct[old] = 2.0 * Tred * ct[nw] - ct[old];
        kf += dot_prod[a] * ct[old];
        upd = !upd;
    }
    return kf;
}

/* SRI parameter variants (pyjac_amd/tables.py DA_SRIQ): each emitter's print precision */
enum { SR_A6, SR_B6, SR_C6, SR_D8, SR_E6, SR_USE_DE, SR_A4, SR_B4, SR_C4, SR_A16, SR_B16, SR_C16, SR_AB16,
       SR_INVC16, SR_E16, SRW = 16 };

/* the F_i factor as get_rxn_pres_mod and write_dr_dy print it (rate_subs.py:1229-1256,
 * create_jacobian.py:249-266): '{:.6}' parameters, d as '{:.8e}' */
static double sri_F(const double *Q, double T, double X)
{
    double F = pow(Q[SR_A6] * exp(-Q[SR_B6] / T) + exp(-T / Q[SR_C6]), X);
    if (Q[SR_USE_DE] != 0.0) F = F * Q[SR_D8] * pow(T, Q[SR_E6]);
    return F;
}

static double fwd_kf(const pjo_mech *m, int i, double T, double logT, double pres)
{
    if (m->flags[i] & F_PLOG) return plog_kf(m, i, T, logT, pres);
    if (m->flags[i] & F_CHEB) return cheb_kf(m, i, T, pres, 0);
    return rate_const(m->A[i], m->b[i], m->E[i], T, logT);
}

/* ln-Kc polynomial + prefactor, rate_subs.py:660-809 */
static double eval_Kc(const pjo_mech *m, int i, double T, double logT)
{
    double Kc = 0.0;
    for (int g = m->kc_ptr[i]; g < m->kc_ptr[i + 1]; ++g) {
        const double *c = m->kcg + 15 * g;
        const double *a = (T <= c[0]) ? c + 1 : c + 8;
        Kc += (a[0] + a[1] * logT + T * (a[2] + T * (a[3] + T * (a[4] + a[5] * T))) - a[6] / T);
    }
    return m->kcpref[i] * exp(Kc);
}

/* concentration product of one reaction side, rate_subs.py:634-658 / 811-840: whole-number coefficients by
 * repeated multiplication, fractional ones through pow() */
static double conc_prod(const int32_t *sp, const double *nu, int p0, int p1, const double *C)
{
    double r = 1.0;
    for (int p = p0; p < p1; ++p) {
        if (nu[p] == floor(nu[p])) { for (int q = 0; q < (int)nu[p]; ++q) r *= C[sp[p]]; }
        else r *= pow(C[sp[p]], nu[p]);
    }
    return r;
}

/* eval_conc, rate_subs.py:1625-1710 */
void pjo_eval_conc(const pjo_mech *m, double T, double pres, const double *y,
                   double *y_N, double *mw_avg, double *rho, double *conc)
{
    int n = m->nsp;
    double s = 0.0;
    for (int k = 0; k < n - 1; ++k) s += y[k];
    *y_N = 1.0 - s;
    double w = 0.0;
    for (int k = 0; k < n - 1; ++k) w += y[k] * (1.0 / m->mw[k]);
    w += (*y_N) * (1.0 / m->mw[n - 1]);
    *mw_avg = 1.0 / w;
    *rho = pres * (*mw_avg) / (RU * T);
    for (int k = 0; k < n - 1; ++k) conc[k] = (*rho) * y[k] * (1.0 / m->mw[k]);
    conc[n - 1] = (*rho) * (*y_N) * (1.0 / m->mw[n - 1]);
}

/* eval_rxn_rates, rate_subs.py:254-876 */
void pjo_eval_rxn_rates(const pjo_mech *m, double T, double pres, const double *C,
                        double *fwd, double *rev)
{
    double logT = log(T);
    for (int i = 0; i < m->nrxn; ++i) {
        double kf = fwd_kf(m, i, T, logT, pres);
        fwd[i] = conc_prod(m->reac_sp, m->reac_nu, m->reac_ptr[i], m->reac_ptr[i + 1], C) * kf;
        if (m->flags[i] & F_REV) {
            double Kc = eval_Kc(m, i, T, logT);
            rev[m->rev_idx[i]] =
                conc_prod(m->prod_sp, m->prod_nu, m->prod_ptr[i], m->prod_ptr[i + 1], C) * kf / Kc;
        }
    }
}

static double third_body(const pjo_mech *m, int i, double mm, const double *C)
{
    double thd = mm;
    for (int e = m->eff_ptr[i]; e < m->eff_ptr[i + 1]; ++e) {
        double a = m->eff[e];
        if (a == 1.0) continue;
        if (a > 1.0) thd += (a - 1.0) * C[m->eff_sp[e]];
        else thd -= (1.0 - a) * C[m->eff_sp[e]];
    }
    return thd;
}

/* get_rxn_pres_mod, rate_subs.py:879-1294 */
void pjo_get_rxn_pres_mod(const pjo_mech *m, double T, double pres, const double *C,
                          double *pres_mod)
{
    double logT = log(T);
    double mm = pres / (RU * T);
    for (int i = 0; i < m->nrxn; ++i) {
        int fl = m->flags[i];
        int pi = m->pres_idx[i];
        if (pi < 0) continue;
        if (fl & F_THD) pres_mod[pi] = third_body(m, i, mm, C);
        if (fl & F_PDEP) {
            double thd = 0.0;
            if (m->pdep_sp[i] < 0) thd = third_body(m, i, mm, C);
            const double *pd = m->pd + 3 * i;
            double k0, kinf;
            if (fl & F_LOW) {
                k0 = rate_const(pd[0], pd[1], pd[2], T, logT);
                kinf = rate_const(m->A[i], m->b[i], m->E[i], T, logT);
            } else {
                k0 = rate_const(m->A[i], m->b[i], m->E[i], T, logT);
                kinf = rate_const(pd[0], pd[1], pd[2], T, logT);
            }
            double Pr = (m->pdep_sp[i] >= 0) ? k0 * C[m->pdep_sp[i]] / kinf : k0 * thd / kinf;
            double val;
            if (fl & F_TROE) {
                const double *t8 = m->troe8 + 5 * i;
                double Fc = t8[0] * exp(-T / t8[2]) + t8[1] * exp(-T / t8[3]);
                if (fl & F_TROE4) Fc += exp(-t8[4] / T);
                double logFcent = log10(fmax(Fc, 1.0e-300));
                double A = log10(fmax(Pr, 1.0e-300)) - 0.67 * logFcent - 0.4;
                double B = 0.806 - 1.1762 * logFcent - 0.14 * log10(fmax(Pr, 1.0e-300));
                val = pow(10.0, logFcent / (1.0 + A * A / (B * B)));
                if (fl & F_LOW) val = val * Pr / (1.0 + Pr);
                else val = val / (1.0 + Pr);
            } else if (fl & F_SRI) {
                /* rate_subs.py:1229-1256 */
                double X = 1.0 / (1.0 + log10(fmax(Pr, 1.0e-300)) * log10(fmax(Pr, 1.0e-300)));
                val = sri_F(m->sriq + SRW * i, T, X);
                if (fl & F_LOW) val = val * Pr / (1.0 + Pr);
                else val = val / (1.0 + Pr);
            } else {
                if (fl & F_LOW) val = Pr / (1.0 + Pr);
                else val = 1.0 / (1.0 + Pr);
            }
            pres_mod[pi] = val;
        }
    }
}

/* eval_spec_rates, rate_subs.py:1297-1542 */
void pjo_eval_spec_rates(const pjo_mech *m, const double *fwd, const double *rev,
                         const double *pres_mod, double *sp_rates, double *dy_N)
{
    int n = m->nsp;
    double last = 0.0;
    for (int k = 0; k < n - 1; ++k) sp_rates[k] = 0.0;
    for (int i = 0; i < m->nrxn; ++i) {
        double R = fwd[i];
        if (m->flags[i] & F_REV) R = fwd[i] - rev[m->rev_idx[i]];
        for (int p = m->net_ptr[i]; p < m->net_ptr[i + 1]; ++p) {
            double v = m->net_nu[p] * R;
            if (m->pres_idx[i] >= 0) v *= pres_mod[m->pres_idx[i]];
            if (m->net_sp[p] == n - 1) last += v;
            else sp_rates[m->net_sp[p]] += v;
        }
    }
    *dy_N = last;
}

/* eval_h / eval_cp, rate_subs.py:1806-2086 */
void pjo_eval_h(const pjo_mech *m, double T, double *h)
{
    for (int k = 0; k < m->nsp; ++k) {
        const double *a = (T <= m->tmid[k]) ? m->lo + 7 * k : m->hi + 7 * k;
        h[k] = (RU / m->mw[k]) *
               (a[5] + T * (a[0] + T * (a[1] / 2.0 + T * (a[2] / 3.0 + T * (a[3] / 4.0 + a[4] / 5.0 * T)))));
    }
}

void pjo_eval_cp(const pjo_mech *m, double T, double *cp)
{
    for (int k = 0; k < m->nsp; ++k) {
        const double *a = (T <= m->tmid[k]) ? m->lo + 7 * k : m->hi + 7 * k;
        cp[k] = (RU / m->mw[k]) * (a[0] + T * (a[1] + T * (a[2] + T * (a[3] + a[4] * T))));
    }
}

/* dydt (CONP), rate_subs.py:2171-2335.  y = [T, Y_0..Y_{NSP-2}], dy[NSP]. */
void pjo_dydt(const pjo_mech *m, double t, double pres, const double *y, double *dy)
{
    (void)t;
    int n = m->nsp;
    double *buf = (double *)malloc(sizeof(double) * (3 * n + 2 * m->nrxn + m->npres + 4));
    double *conc = buf, *cp = conc + n, *h = cp + n, *fwd = h + n, *rev = fwd + m->nrxn,
           *pm = rev + m->nrxn;
    double y_N, mw_avg, rho, dy_N;
    pjo_eval_conc(m, y[0], pres, y + 1, &y_N, &mw_avg, &rho, conc);
    pjo_eval_rxn_rates(m, y[0], pres, conc, fwd, rev);
    pjo_get_rxn_pres_mod(m, y[0], pres, conc, pm);
    pjo_eval_spec_rates(m, fwd, rev, pm, dy + 1, &dy_N);
    pjo_eval_cp(m, y[0], cp);
    double cp_avg = 0.0;
    for (int k = 0; k < n - 1; ++k) cp_avg += cp[k] * y[k + 1];
    cp_avg += cp[n - 1] * y_N;
    pjo_eval_h(m, y[0], h);
    double s = 0.0;
    for (int k = 0; k < n - 1; ++k)
        if (m->seen[k]) s += dy[k + 1] * h[k] * m->mw[k];
    if (m->seen[n - 1]) s += dy_N * h[n - 1] * m->mw[n - 1];
    dy[0] = (-1.0 / (rho * cp_avg)) * s;
    for (int k = 0; k < n - 1; ++k) dy[k + 1] *= (m->mw[k] / rho);
    free(buf);
}

static int g_sum_last = 0;
void pjo_set_sum_last_species(int on) { g_sum_last = on; }

static double sum_nu(const double *nu, int p0, int p1)
{
    double s = 0.0;
    for (int p = p0; p < p1; ++p) s += nu[p];
    return s;
}

/* d(conc product)/dC_j * k : "__get_s_term", create_jacobian.py:400-440 */
static double s_term(const int32_t *sp, const double *nu, int p0, int p1, int j,
                     const double *C, double k)
{
    int found = -1;
    for (int p = p0; p < p1; ++p) if (sp[p] == j) found = p;
    if (found < 0) return 0.0;
    double v = k;
    double n = nu[found];
    if (n != 1.0) v *= n;
    /* reference quirk kept for parity: the power of C_j is only written "if (nu - 1) > 0"
     * (create_jacobian.py:417-427), so a coefficient below one loses its C_j^(nu-1) factor */
    if (n - 1.0 > 0.0) {
        if (n == floor(n)) { for (int q = 0; q < (int)n - 1; ++q) v *= C[j]; }
        else v *= pow(C[j], n - 1.0);
    }
    for (int p = p0; p < p1; ++p) {
        if (p == found) continue;
        if (nu[p] == floor(nu[p])) { for (int q = 0; q < (int)nu[p]; ++q) v *= C[sp[p]]; }
        else v *= pow(C[sp[p]], nu[p]);
    }
    return v;
}

static int in_list(const int32_t *sp, int p0, int p1, int j)
{
    for (int p = p0; p < p1; ++p) if (sp[p] == j) return 1;
    return 0;
}

/* eval_jacob (CONP), create_jacobian.py:2189-3298.  y = [T, Y_0..Y_{NSP-2}];
 * jac[NSP*NSP] column-major, fully written (callers of the reference pre-zero
 * jac; this function does it itself). */
void pjo_eval_jacob(const pjo_mech *m, double t, double pres, const double *y, double *jac)
{
    (void)t;
    const int n = m->nsp, last = n - 1;
    const double T = y[0];
    double *buf = (double *)calloc(7 * n + 2 * m->nrxn + m->npres + 8, sizeof(double));
    double *conc = buf, *cp = conc + n, *h = cp + n, *sr = h + n, *dBdT = sr + n,
           *Jlast = dBdT + n /* J_nplusjplus[n] */, *fwd = Jlast + n, *rev = fwd + m->nrxn,
           *pm = rev + m->nrxn;
    double y_N, mw_avg, rho;
    memset(jac, 0, sizeof(double) * n * n);
    pjo_eval_conc(m, T, pres, y + 1, &y_N, &mw_avg, &rho, conc);
    pjo_eval_rxn_rates(m, T, pres, conc, fwd, rev);
    pjo_get_rxn_pres_mod(m, T, pres, conc, pm);
    pjo_eval_spec_rates(m, fwd, rev, pm, sr, &sr[last]);

    const double mm = pres / (RU * T);
    const double logT = log(T);
    const double rho_inv = 1.0 / rho;
    double J_nplusone = 0.0;
    /* dB/dT, create_jacobian.py:761-865 */
    for (int k = 0; k < n; ++k) {
        const double *a = (T <= m->tmid[k]) ? m->lo + 7 * k : m->hi + 7 * k;
        dBdT[k] = ((a[0] - 1.0) + a[5] / T) / T + a[1] / 2.0 +
                  T * (a[2] / 3.0 + T * (a[3] / 4.0 + a[4] / 5.0 * T));
    }

    for (int i = 0; i < m->nrxn; ++i) {
        const int fl = m->flags[i];
        const int r0 = m->reac_ptr[i], r1 = m->reac_ptr[i + 1];
        const int p0 = m->prod_ptr[i], p1 = m->prod_ptr[i + 1];
        const int isrev = (fl & F_REV) != 0;
        const int pi = m->pres_idx[i];
        const double Rf = fwd[i];
        const double Rr = isrev ? rev[m->rev_idx[i]] : 0.0;
        const double R = isrev ? (Rf - Rr) : Rf;
        const double nu_r = sum_nu(m->reac_nu, r0, r1);
        const double nu_p = sum_nu(m->prod_nu, p0, p1);
        double Pr = 0.0, Fcent = 0.0, A = 0.0, B = 0.0, lnF_AB = 0.0, X = 0.0;
        const double *Q = m->sriq + SRW * i;
        const double *inf = m->infs + 4 * i;

        /* ---------------- d/dT, create_jacobian.py:2728-2845 ---------------- */
        double lead = 0.0;   /* part of j_temp in front of "(pres_mod / T) * (...)" */
        double scale;        /* factor multiplying the elementary part */
        if (fl & F_PDEP) {
            /* write_pr, create_jacobian.py:953-1063 */
            double conc_temp;
            if (m->pdep_sp[i] >= 0) conc_temp = conc[m->pdep_sp[i]];
            else if (!(fl & F_HAS_EFF)) conc_temp = mm;
            else conc_temp = third_body(m, i, mm, conc);
            Pr = conc_temp * rate_const(inf[0], inf[1], inf[2], T, logT);
            double extra = 0.0;
            if (fl & F_TROE) {
                /* write_troe / write_troe_dt, create_jacobian.py:1066-1111, 1240-1294 */
                const double *tp = m->troe + 4 * i;
                Fcent = (1.0 - tp[0]) * exp(T / -tp[1]) + tp[0] * exp(T / -tp[2]);
                if (fl & F_TROE4) Fcent += exp(-tp[3] / T);
                A = log10(fmax(Pr, 1.0e-300)) - 0.67 * log10(fmax(Fcent, 1.0e-300)) - 0.4;
                B = 0.806 - 1.1762 * log10(fmax(Fcent, 1.0e-300)) - 0.14 * log10(fmax(Pr, 1.0e-300));
                lnF_AB = 2.0 * log(fmax(Fcent, 1.0e-300)) * A /
                         (B * B * B * (1.0 + A * A / (B * B)) * (1.0 + A * A / (B * B)));
                double dF = (-(1.0 - tp[0]) / tp[1]) * exp(T / -tp[1]) - (tp[0] / tp[2]) * exp(T / -tp[2]);
                if (fl & F_TROE4) dF += (tp[3] / (T * T)) * exp(-tp[3] / T);
                extra = (((1.0 / (Fcent * (1.0 + A * A / (B * B)))) -
                          lnF_AB * (-(0.67 / log(10.0)) * B + (1.1762 / log(10.0)) * A) / Fcent) * dF) -
                        lnF_AB * ((1.0 / log(10.0)) * B + (0.14 / log(10.0)) * A) *
                            (inf[1] + (inf[2] / T) - 1.0) / T;
            }
            if (fl & F_SRI) {
                /* write_sri / write_sri_dt, create_jacobian.py:1114-1132, 1194-1237 ('{:.16}' parameters) */
                X = 1.0 / (1.0 + log10(fmax(Pr, 1.0e-300)) * log10(fmax(Pr, 1.0e-300)));
                extra = X * (((Q[SR_AB16] / (T * T)) * exp(-Q[SR_B16] / T) - Q[SR_INVC16] * exp(T / -Q[SR_C16])) /
                                 (Q[SR_A16] * exp(-Q[SR_B16] / T) + exp(T / -Q[SR_C16])) -
                             X * 0.8685889638065035 * log10(fmax(Pr, 1.0e-300)) * (inf[1] + (inf[2] / T) - 1.0) *
                                 log(Q[SR_A16] * exp(-Q[SR_B16] / T) + exp(T / -Q[SR_C16])) / T);
                if (Q[SR_E16] != 0.0) extra += (Q[SR_E16] / T);
            }
            /* get_pdep_dt, create_jacobian.py:1135-1191 (beta difference '%.4e') */
            double dpr = (inf[3] + (inf[2] / T) - 1.0) / (T * (1.0 + Pr));
            if (fl & F_HIGH) dpr = -Pr * dpr;
            lead = pm[pi] * (dpr + extra) * R;
            scale = pm[pi] / T;
        } else if (fl & F_THD) {
            lead = -pm[pi] * R / T;
            scale = pm[pi] / T;
        } else {
            scale = 1.0 / T;
        }

        int doT = 1;
        double el = 0.0;
        if (fl & F_CHEB) {
            /* write_cheb_rxn_dt, create_jacobian.py:1610-1684 */
            const double *C = m->cheb + m->cheb_ptr[i];
            el = cheb_ut(m, i, T, pres) * (C[CH_DFAC] / T) * R;
            if (nu_r != 1.0) el += Rf * (1.0 - nu_r);
            if (isrev) {
                double db = 0.0;
                for (int p = m->net_ptr[i]; p < m->net_ptr[i + 1]; ++p) db += m->net_nu[p] * dBdT[m->net_sp[p]];
                el -= Rr * ((1.0 - nu_p) + (-T * db));
            }
        } else if (fl & F_PLOG) {
            /* write_plog_rxn_dt, create_jacobian.py:1687-1850 */
            int q0 = m->plog_ptr[i], q1 = m->plog_ptr[i + 1];
            const double *P = m->plog, *P4 = m->plog4;
            double dk = 0.0;
            int have = 0;
            if (pres <= P4[q0]) { dk = P[4 * q0 + 2] + P[4 * q0 + 3] / T; have = 1; }
            for (int q = q0; q < q1 - 1 && !have; ++q)
                if (pres > P4[q] && pres <= P4[q + 1]) {
                    dk = P[4 * q + 2] + P[4 * q + 3] / T +
                         ((P[4 * q + 6] - P[4 * q + 2]) + (P[4 * q + 7] - P[4 * q + 3]) / T) *
                             (log(pres) - log(P[4 * q])) / (log(P[4 * q + 4]) - log(P[4 * q]));
                    have = 1;
                }
            if (!have && pres > P4[q1 - 1]) { dk = P[4 * (q1 - 1) + 2] + P[4 * (q1 - 1) + 3] / T; have = 1; }
            el = dk * R;
            if (nu_r != 1.0) el += Rf * (1.0 - nu_r);
            if (isrev) {
                double db = 0.0;
                for (int p = m->net_ptr[i]; p < m->net_ptr[i + 1]; ++p) db += m->net_nu[p] * dBdT[m->net_sp[p]];
                el -= Rr * ((1.0 - nu_p) + (-T * db));
            }
        } else {
            /* get_elementary_rxn_dt, create_jacobian.py:1398-1529 */
            double bb = m->b[i], EE = m->E[i];
            int has_dk = (fabs(bb) > 1.0e-90) || (fabs(EE) > 1.0e-90);
            double dk = 0.0;
            if (fabs(bb) > 1.0e-90) dk += bb;
            if (fabs(EE) > 1.0e-90) dk += (EE / T);
            if (isrev) {
                int any = 0;
                if (has_dk) { el += R * dk; any = 1; }
                if (nu_r != 1.0) { el += Rf * (1.0 - nu_r); any = 1; }
                double db = 0.0;
                int has_db = 0;
                /* get_db_dt, create_jacobian.py:868-950: products first, then
                 * reactants not among the products */
                for (int p = p0; p < p1; ++p) {
                    int k = m->prod_sp[p];
                    double nu = m->prod_nu[p];
                    for (int r = r0; r < r1; ++r) if (m->reac_sp[r] == k) nu -= m->reac_nu[r];
                    if (nu == 0.0) continue;
                    db += nu * dBdT[k];
                    has_db = 1;
                }
                for (int r = r0; r < r1; ++r) {
                    int k = m->reac_sp[r];
                    if (in_list(m->prod_sp, p0, p1, k)) continue;
                    db -= m->reac_nu[r] * dBdT[k];
                    has_db = 1;
                }
                if (has_db || nu_p != 1.0) {
                    double inner = 0.0;
                    if (nu_p != 1.0) inner += (1.0 - nu_p);
                    if (has_db) inner += -T * db;
                    el -= Rr * inner;
                    any = 1;
                }
                doT = any;
            } else {
                if (has_dk || nu_r != 1.0) {
                    double inner = dk;
                    if (nu_r != 1.0) inner += (1.0 - nu_r);
                    el = Rf * inner;
                } else {
                    doT = 0; /* reference emits no d/dT line at all for this reaction */
                }
            }
        }
        if (doT) {
            double j_temp = (lead + scale * el) * rho_inv;
            for (int p = m->net_ptr[i]; p < m->net_ptr[i + 1]; ++p) {
                int k = m->net_sp[p];
                double v = j_temp * m->net_nu[p] * m->mw[k];
                /* Reference quirk kept for parity: the emitter tests
                 * touched[k_sp + 1] with k_sp + 1 == NSP (never set), so every
                 * reaction ASSIGNS J_nplusone ('=' not '+=',
                 * create_jacobian.py:2786-2793, 2817-2818): only the last
                 * reaction with net production of the last species survives
                 * into jac[0].  pjo_set_sum_last_species(1) restores the sum. */
                if (k == last) J_nplusone = g_sum_last ? J_nplusone + v : v;
                else jac[k + 1] += v;
            }
        }

        /* ---------------- d/dY_j, create_jacobian.py:2850-2938 ---------------- */
        double pres_mod_temp = 0.0;
        const int pdep_has = (fl & F_PDEP) && (m->pdep_sp[i] >= 0 || (fl & F_HAS_EFF));
        const int use_pmt = ((fl & F_PDEP) || (fl & F_THD)) && ((fl & F_HAS_EFF) || m->pdep_sp[i] >= 0);
        if (use_pmt) {
            /* write_dr_dy, create_jacobian.py:127-269 */
            if (fl & F_PDEP) {
                double x = (fl & F_LOW) ? (1.0 / (1.0 + Pr)) : (-Pr / (1.0 + Pr));
                if (fl & F_TROE)
                    x -= log(fmax(Fcent, 1.0e-300)) * 2.0 * A *
                         (B * (1.0 / log(10.0)) + A * (0.14 / log(10.0))) /
                         (B * B * B * (1.0 + A * A / (B * B)) * (1.0 + A * A / (B * B)));
                else if (fl & F_SRI)      /* create_jacobian.py:173-179 ('{:.4}' parameters) */
                    x -= X * X * 0.8685889638065035 * log10(fmax(Pr, 1.0e-300)) *
                         log(Q[SR_A4] * exp(-Q[SR_B4] / T) + exp(T / -Q[SR_C4]));
                pres_mod_temp = x * R;
            } else {
                pres_mod_temp = R;
            }
        }
        double rn = nu_r, pn = isrev ? nu_p : 0.0;
        if ((fl & F_HAS_EFF) && !(fl & F_PDEP)) { rn += 1.0; if (isrev) pn += 1.0; }
        double inner = 0.0;
        if (rn != 0.0) inner += rn * Rf;
        if (pn != 0.0) inner -= pn * Rr;
        if (pdep_has) inner += pres_mod_temp;
        double j_temp = -mw_avg * rho_inv * ((pi >= 0) ? pm[pi] * inner : inner);
        if (pdep_has) {
            pres_mod_temp *= rate_const(inf[0], inf[1], inf[2], T, logT);
            if (fl & F_TROE) pres_mod_temp *= pow(Fcent, 1.0 / (1 + A * A / (B * B)));
            else if (fl & F_SRI) pres_mod_temp *= sri_F(Q, T, X);      /* create_jacobian.py:249-266 */
            pres_mod_temp /= (1.0 + Pr);
        }
        double kf = (fl & F_CHEB) ? cheb_kf(m, i, T, pres, 1) : fwd_kf(m, i, T, logT, pres);
        double kr = 0.0;
        if (isrev) kr = kf / eval_Kc(m, i, T, logT);

        for (int j = 0; j < n - 1; ++j) {
            /* write_dr_dy_species, create_jacobian.py:341-489 */
            double mw_frac = m->mw[j] / m->mw[last];
            double v = j_temp * (1.0 - mw_frac);
            if ((((fl & F_PDEP) && m->pdep_sp[i] < 0) || (fl & F_THD)) && (fl & F_HAS_EFF)) {
                double aij = 1.0, aiN = 1.0;
                for (int e = m->eff_ptr[i]; e < m->eff_ptr[i + 1]; ++e) {
                    if (m->eff_sp[e] == j && aij == 1.0) aij = m->eff[e];
                    if (m->eff_sp[e] == last && aiN == 1.0) aiN = m->eff[e];
                }
                if (aiN != 0.0) aij -= aiN * mw_frac;
                if (aij != 0.0) v += aij * pres_mod_temp;
            } else if ((fl & F_PDEP) && m->pdep_sp[i] >= 0 &&
                       (m->pdep_sp[i] == j || m->pdep_sp[i] == last)) {
                if (m->pdep_sp[i] == j) v += pres_mod_temp;
                else v -= pres_mod_temp * (m->mw[j] / m->mw[m->pdep_sp[i]]);
            }
            double s = 0.0;
            s += s_term(m->reac_sp, m->reac_nu, r0, r1, j, conc, kf);
            if (isrev) s -= s_term(m->prod_sp, m->prod_nu, p0, p1, j, conc, kr);
            double sl = 0.0;
            sl += s_term(m->reac_sp, m->reac_nu, r0, r1, last, conc, kf);
            if (isrev) sl -= s_term(m->prod_sp, m->prod_nu, p0, p1, last, conc, kr);
            s -= mw_frac * sl;
            if (pi >= 0) v += pm[pi] * s;
            else v += s;
            for (int p = m->net_ptr[i]; p < m->net_ptr[i + 1]; ++p) {
                int k = m->net_sp[p];
                double f = (m->mw[k] / m->mw[j]) * m->net_nu[p];
                if (k == last) Jlast[j] += f * v;
                else jac[k + 1 + n * (j + 1)] += f * v;
            }
        }
    }

    /* ---------------- completion, create_jacobian.py:3040-3268 ---------------- */
    pjo_eval_h(m, T, h);
    pjo_eval_cp(m, T, cp);
    double cp_avg = 0.0;
    for (int k = 0; k < n - 1; ++k) cp_avg += y[k + 1] * cp[k];
    cp_avg += y_N * cp[last];
    jac[0] = 0.0;
    double working_temp = 1.0 / cp_avg;
    double j_temp = 1.0 / (rho * cp_avg * cp_avg);
    for (int k = 0; k < n; ++k) {
        for (int j = 0; j < n - 1; ++j) {
            double f = (m->mw[k] / m->mw[j]) * (1.0 - m->mw[j] / m->mw[last]);
            double *e = (k == last) ? &Jlast[j] : &jac[k + 1 + n * (j + 1)];
            if (m->seen[k]) *e += (sr[k] * mw_avg * f * rho_inv);
            if (m->seen[k])
                jac[n * (j + 1)] -= h[k] * (working_temp * (*e) - (j_temp * (cp[j] - cp[last]) * sr[k] * m->mw[k]));
        }
    }
    /* write_dcp_dt, create_jacobian.py:1297-1395 */
    working_temp = 0.0;
    for (int k = 0; k < n; ++k) {
        const double *a = (T <= m->tmid[k]) ? m->lo + 7 * k : m->hi + 7 * k;
        double yk = (k == last) ? y_N : y[k + 1];
        working_temp += yk * (RU / m->mw[k]) * (a[1] + T * (2.0 * a[2] + T * (3.0 * a[3] + 4.0 * a[4] * T)));
    }
    /* write_dt_completion, create_jacobian.py:1853-1905 */
    double s = 0.0;
    for (int k = 0; k < n; ++k) {
        s += sr[k] * m->mw[k] * (-working_temp * h[k] / cp_avg + cp[k]);
        if (k < last) s += jac[k + 1] * h[k] * rho;
        else s += J_nplusone * h[k] * rho;
    }
    jac[0] = -s / (rho * cp_avg);
    free(buf);
}

/* Finite-difference Jacobian of dydt, pyjac/performance_tester/fd_jacob.c:10-113
 * (FD_ORD 1, CVODE-style increment).  jac[NSP*NSP] column-major. */
#include <float.h>
void pjo_fd_jacob(const pjo_mech *m, double t, double pres, const double *cy, double *jac)
{
    const int n = m->nsp;
    const double ATOL = 1e-15, RTOL = 1e-8;
    double *y = (double *)malloc(sizeof(double) * 4 * n);
    double *dy = y + n, *ewt = dy + n, *ftemp = ewt + n;
    memcpy(y, cy, n * sizeof(double));
    pjo_dydt(m, t, pres, y, dy);
    for (int i = 0; i < n; ++i) ewt[i] = ATOL + (RTOL * fabs(y[i]));
    const double srur = sqrt(DBL_EPSILON);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += (ewt[i] * dy[i]) * (ewt[i] * dy[i]);
    const double fac = sqrt(sum / ((double)(n)));
    const double r0 = 1000.0 * RTOL * DBL_EPSILON * ((double)(n)) * fac;
    for (int j = 0; j < n; ++j) {
        const double yj_orig = y[j];
        const double r = fmax(srur * fabs(yj_orig), r0 / ewt[j]);
        y[j] = yj_orig + r;
        pjo_dydt(m, t, pres, y, ftemp);
        for (int i = 0; i < n; ++i) jac[i + n * j] = (ftemp[i] - dy[i]) / r;
        y[j] = yj_orig;
    }
    free(y);
}

/* ---- batch drivers (state-major AoS in/out; OpenMP over states) ----
 * Mirrors the reference speed test's protocol
 * (pyjac/performance_tester/tester.c.in:23-31): one parallel-for over states,
 * each writing its own NSP*NSP block.  y_aos[s*NSP .. ] = [T, Y_0..Y_{NSP-2}]. */
void pjo_batch_jacob(const pjo_mech *m, long num, const double *pres, const double *y_aos,
                     double *jac_aos, int nthreads)
{
    const long n = m->nsp;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (long s = 0; s < num; ++s)
        pjo_eval_jacob(m, 0.0, pres[s], y_aos + s * n, jac_aos + s * n * n);
}

void pjo_batch_dydt(const pjo_mech *m, long num, const double *pres, const double *y_aos,
                    double *dy_aos, int nthreads)
{
    const long n = m->nsp;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (long s = 0; s < num; ++s)
        pjo_dydt(m, 0.0, pres[s], y_aos + s * n, dy_aos + s * n);
}
