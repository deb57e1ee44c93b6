// pj_tables.cpp -- canonical mechanism blob -> device programs (host, C++).
#include "pj_tables.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <cstdio>
#include <cstring>

namespace pj {

namespace {
struct Blob {
    const int32_t* I;
    const double* D;
    const int32_t* ia(int j) const { return I + I[16 + j]; }
    const double* da(int j) const { return D + I[48 + j]; }
};

int kind_rank(int fl)
{
    if ((fl & F_PDEP) && (fl & F_TROE)) return 0;
    if (fl & F_PDEP) return 1;
    if (fl & F_THD) return 2;
    if (fl & (F_PLOG | F_CHEB)) return 3;
    if (fl & F_REV) return 4;
    return 5;
}
}  // namespace

// Structural validation of a table blob before anything indexes through it: a corrupt or truncated
// .pjtab must be refused, not read out of bounds.  Array j of the blob extends to the start of array
// j + 1 (pyjac_amd/tables.py writes them back to back).
static bool validate_blob(const int32_t* I, long nI, const double* D, long nD, std::string& err)
{
    auto bad = [&](const char* what) { err = std::string("malformed mechanism table blob: ") + what; return false; };
    if (nI < HDR || I[0] != MAGIC) return bad("header");
    if (I[1] != BLOB_VERSION) { err = "mechanism table blob: version " + std::to_string(I[1]) + ", this library reads version " + std::to_string(BLOB_VERSION) + " (write the table file again with this release)"; return false; }
    if (I[12] != nI || I[13] != nD) return bad("header");
    const int nsp = I[2], nrxn = I[3];
    if (nsp < 1 || nsp > 4096 || nrxn < 0 || nrxn > 8191 || I[4] < 0 || I[4] > nrxn || I[5] < 0 || I[5] > nrxn)
        return bad("sizes");
    constexpr int NIA = IA_COUNT, NDA = DA_COUNT;
    long ilen[NIA], dlen[NDA];
    for (int j = 0; j < NIA; ++j) {
        const long off = I[16 + j], nxt = j + 1 < NIA ? I[16 + j + 1] : nI;
        if (off < HDR || off > nI || nxt < off || nxt > nI) return bad("integer array offsets");
        ilen[j] = nxt - off;
    }
    for (int j = 0; j < NDA; ++j) {
        const long off = I[48 + j], nxt = j + 1 < NDA ? I[48 + j + 1] : nD;
        if (off < 0 || off > nD || nxt < off || nxt > nD) return bad("real array offsets");
        dlen[j] = nxt - off;
    }
    auto ia = [&](int j) { return I + I[16 + j]; };
    // per-reaction arrays
    for (int j : {(int)IA_FLAGS, (int)IA_PDEP_SP, (int)IA_REV_IDX, (int)IA_PRES_IDX})
        if (ilen[j] < nrxn) return bad("per-reaction integer array too short");
    for (int j : {(int)IA_REAC_PTR, (int)IA_PROD_PTR, (int)IA_NET_PTR, (int)IA_EFF_PTR, (int)IA_PLOG_PTR, (int)IA_KC_PTR,
                  (int)IA_CHEB_PTR})
        if (ilen[j] < nrxn + 1) return bad("CSR pointer array too short");
    if (ilen[IA_SEEN] < nsp) return bad("species array too short");
    for (int j : {(int)DA_A, (int)DA_B, (int)DA_E, (int)DA_KCPREF})
        if (dlen[j] < nrxn) return bad("per-reaction real array too short");
    if (dlen[DA_MW] < nsp || dlen[DA_TMID] < nsp || dlen[DA_LO] < 7L * nsp || dlen[DA_HI] < 7L * nsp) return bad("species tables");
    if (dlen[DA_PD] < 3L * nrxn || dlen[DA_TROE] < 4L * nrxn || dlen[DA_SRI] < 5L * nrxn || dlen[DA_INFS] < 4L * nrxn ||
        dlen[DA_TROE8] < 5L * nrxn || dlen[DA_SRIQ] < (long)SRW * nrxn)
        return bad("falloff tables");
    {
        // Chebyshev records: CH_COEF + n*m + (n-1)*m doubles each, inside the array
        const int32_t* cp = ia(IA_CHEB_PTR);
        const int32_t* fl = ia(IA_FLAGS);
        const double* ch = D + I[48 + DA_CHEB];
        if (cp[0] != 0) return bad("Chebyshev pointer does not start at 0");
        for (int i = 0; i < nrxn; ++i) {
            if (cp[i + 1] < cp[i] || cp[i + 1] > dlen[DA_CHEB]) return bad("Chebyshev pointer");
            if (!(fl[i] & F_CHEB)) continue;
            if (cp[i + 1] - cp[i] < CH_COEF) return bad("Chebyshev record too short");
            const double n = ch[cp[i] + CH_N], m = ch[cp[i] + CH_M];
            if (!(n >= 3 && n <= CHEB_MAXT && m >= 2 && m <= CHEB_MAXP) || n != std::floor(n) || m != std::floor(m))
                return bad("Chebyshev dimensions (3..12 x 2..12 supported)");
            if (cp[i + 1] - cp[i] != CH_COEF + (long)(n * m) + (long)((n - 1) * m)) return bad("Chebyshev record size");
        }
    }
    // CSR lists: monotone pointers inside their payload arrays, species indices in range
    struct Csr { int ptr, sp, nu; long per; };
    const Csr lists[] = {{IA_REAC_PTR, IA_REAC_SP, DA_REAC_NU, 1}, {IA_PROD_PTR, IA_PROD_SP, DA_PROD_NU, 1},
                         {IA_NET_PTR, IA_NET_SP, DA_NET_NU, 1}, {IA_EFF_PTR, IA_EFF_SP, DA_EFF, 1}};
    for (const Csr& c : lists) {
        const int32_t* ptr = ia(c.ptr);
        if (ptr[0] != 0) return bad("CSR pointer does not start at 0");
        for (int i = 0; i < nrxn; ++i) if (ptr[i + 1] < ptr[i]) return bad("CSR pointer not monotone");
        if (ptr[nrxn] > ilen[c.sp] || ptr[nrxn] > dlen[c.nu]) return bad("CSR payload too short");
        const int32_t* sp = ia(c.sp);
        for (int q = 0; q < ptr[nrxn]; ++q) if (sp[q] < 0 || sp[q] >= nsp) return bad("species index out of range");
    }
    {
        const int32_t *pp = ia(IA_PLOG_PTR), *kp = ia(IA_KC_PTR);
        if (pp[0] != 0 || kp[0] != 0) return bad("PLOG / K_c pointer does not start at 0");
        for (int i = 0; i < nrxn; ++i) if (pp[i + 1] < pp[i] || kp[i + 1] < kp[i]) return bad("PLOG / K_c pointer not monotone");
        if ((long)pp[nrxn] * 4 > dlen[DA_PLOG] || (long)pp[nrxn] > dlen[DA_PLOG4]) return bad("PLOG table too short");
        if ((long)kp[nrxn] * KCW > dlen[DA_KCG]) return bad("K_c group table too short");
    }
    {
        // flag combinations the kernels do not expect (a record field is shared between PLOG rows, SRI rows and
        // Chebyshev records: RI_PLOG_PTR / RI_PLOG_CNT)
        const int32_t* fl = ia(IA_FLAGS);
        for (int i = 0; i < nrxn; ++i) {
            if ((fl[i] & F_CHEB) && (fl[i] & (F_PLOG | F_PDEP | F_THD))) return bad("Chebyshev reaction combined with another pressure dependence");
            if ((fl[i] & F_SRI) && ((fl[i] & F_PLOG) || !(fl[i] & F_PDEP) || (fl[i] & F_TROE))) return bad("SRI parameters on a reaction that is not a plain falloff");
        }
    }
    const int32_t *pd = ia(IA_PDEP_SP), *ri = ia(IA_REV_IDX), *pi = ia(IA_PRES_IDX);
    for (int i = 0; i < nrxn; ++i)
        if (pd[i] < -1 || pd[i] >= nsp || ri[i] < -1 || ri[i] >= I[4] || pi[i] < -1 || pi[i] >= I[5])
            return bad("reaction index out of range");
    return true;
}

bool build_programs(const int32_t* I, long nI, const double* D, long nD, Programs& p)
{
    if (!validate_blob(I, nI, D, nD, p.error)) return false;
    Blob B{I, D};
    const int nsp = I[2], nrxn = I[3];
    p.nsp = nsp; p.nrxn = nrxn; p.nrev = I[4]; p.npres = I[5];
    const int last = nsp - 1;

    const int32_t *flags = B.ia(IA_FLAGS), *reac_ptr = B.ia(IA_REAC_PTR), *reac_sp = B.ia(IA_REAC_SP),
                  *prod_ptr = B.ia(IA_PROD_PTR), *prod_sp = B.ia(IA_PROD_SP), *net_ptr = B.ia(IA_NET_PTR),
                  *net_sp = B.ia(IA_NET_SP), *eff_ptr = B.ia(IA_EFF_PTR), *eff_sp = B.ia(IA_EFF_SP),
                  *plog_ptr = B.ia(IA_PLOG_PTR), *kc_ptr = B.ia(IA_KC_PTR), *pdep_sp = B.ia(IA_PDEP_SP),
                  *rev_idx = B.ia(IA_REV_IDX), *pres_idx = B.ia(IA_PRES_IDX);
    const double *mw = B.da(DA_MW), *tmid = B.da(DA_TMID), *lo = B.da(DA_LO), *hi = B.da(DA_HI),
                 *A = B.da(DA_A), *b = B.da(DA_B), *E = B.da(DA_E), *reac_nu = B.da(DA_REAC_NU),
                 *prod_nu = B.da(DA_PROD_NU), *net_nu = B.da(DA_NET_NU), *eff = B.da(DA_EFF),
                 *troe = B.da(DA_TROE), *plog = B.da(DA_PLOG), *kcg = B.da(DA_KCG),
                 *kcpref = B.da(DA_KCPREF), *infs = B.da(DA_INFS), *plog4 = B.da(DA_PLOG4);

    // ---- species ----
    p.sp.assign((size_t)nsp * SPW, 0.0);
    for (int k = 0; k < nsp; ++k) {
        double* s = &p.sp[(size_t)k * SPW];
        s[0] = 1.0 / mw[k]; s[1] = mw[k]; s[2] = tmid[k]; s[3] = mw[k] / mw[last];
        for (int c = 0; c < 7; ++c) { s[4 + c] = lo[7 * k + c]; s[11 + c] = hi[7 * k + c]; }
    }

    // ---- device order ----
    std::vector<int> order(nrxn);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int x, int y) { return kind_rank(flags[x]) < kind_rank(flags[y]); });
    std::vector<int> dev_of(nrxn);
    for (int d = 0; d < nrxn; ++d) dev_of[order[d]] = d;

    // quirk: which reaction's d/dT survives in J_nplusone (create_jacobian.py:2786-2818)
    auto no_dt = [&](int i) {
        const int fl = flags[i];
        if ((fl & F_REV) || (fl & (F_PLOG | F_CHEB))) return false;
        double nr = 0;
        for (int q = reac_ptr[i]; q < reac_ptr[i + 1]; ++q) nr += reac_nu[q];
        return std::fabs(b[i]) <= 1e-90 && std::fabs(E[i]) <= 1e-90 && nr == 1.0;
    };
    int lastq_orig = -1;
    for (int i = 0; i < nrxn; ++i) {
        if (no_dt(i)) continue;
        for (int q = net_ptr[i]; q < net_ptr[i + 1]; ++q)
            if (net_sp[q] == last && net_nu[q] != 0.0) lastq_orig = i;
    }

    const int ONE = nsp;
    p.ri.assign((size_t)nrxn * RIW, 0);
    p.rd.assign((size_t)nrxn * RDW, 0.0);
    int ng = 0;
    // per reaction (device order): species of each g slot, and alpha_ij - 1 list
    std::vector<std::vector<int>> gslot_sp(nrxn);
    std::vector<std::vector<std::pair<int, double>>> effs(nrxn);

    for (int d = 0; d < nrxn; ++d) {
        const int i = order[d];
        int fl = flags[i];
        int32_t* ri = &p.ri[(size_t)d * RIW];
        double* rd = &p.rd[(size_t)d * RDW];

        auto slots = [&](const int32_t* ptr, const int32_t* sp, const double* nu, int* out, double& total) {
            int n = 0;
            total = 0;
            for (int q = ptr[i]; q < ptr[i + 1]; ++q) {
                if (nu[q] != std::floor(nu[q]) || nu[q] < 0) return -1;
                total += nu[q];
                for (int r = 0; r < (int)nu[q]; ++r) {
                    if (n >= 3) return -1;
                    out[n++] = sp[q];
                }
            }
            for (int r = n; r < 3; ++r) out[r] = ONE;
            return n;
        };
        int rs[3], ps[3];
        double nr, np_;
        int nrs = slots(reac_ptr, reac_sp, reac_nu, rs, nr);
        int nps = slots(prod_ptr, prod_sp, prod_nu, ps, np_);
        // General stoichiometry (a fractional coefficient, or more than three molecules on a side): the molecule
        // slots stay empty (factor 1) and the reaction carries (species, nu) factor lists -- one derivative value
        // g per factor instead of one per molecule (mech_interpret.py:300-318, 398-416; rate_subs.py:634-658;
        // create_jacobian.py:400-448)
        ri[RI_GEN_PTR] = (int)p.gen_sp.size();
        if (nrs < 0 || nps < 0) {
            fl |= F_GEN;
            nr = np_ = 0.0;
            for (int q = reac_ptr[i]; q < reac_ptr[i + 1]; ++q) {
                if (!(reac_nu[q] > 0.0)) { p.error = "non-positive stoichiometric coefficient"; return false; }
                p.gen_sp.push_back(reac_sp[q]); p.gen_nu.push_back(reac_nu[q]); nr += reac_nu[q];
            }
            ri[RI_GEN_NR] = reac_ptr[i + 1] - reac_ptr[i];
            for (int q = prod_ptr[i]; q < prod_ptr[i + 1]; ++q) {
                if (!(prod_nu[q] > 0.0)) { p.error = "non-positive stoichiometric coefficient"; return false; }
                p.gen_sp.push_back(prod_sp[q]); p.gen_nu.push_back(prod_nu[q]); np_ += prod_nu[q];
            }
            ri[RI_GEN_NP] = prod_ptr[i + 1] - prod_ptr[i];
            for (int r = 0; r < 3; ++r) rs[r] = ps[r] = ONE;
            nrs = nps = 0;
        }
        for (int r = 0; r < 3; ++r) { ri[RI_R0 + r] = rs[r]; ri[RI_P0 + r] = ps[r]; }

        const int col = pdep_sp[i];
        if ((fl & F_PDEP) && col >= 0) fl |= F_COLLIDER;
        if (((fl & F_THD) || ((fl & F_PDEP) && col < 0)) && (fl & F_HAS_EFF)) fl |= F_EFFTYPE;
        if (no_dt(i)) fl |= F_NO_DT;
        if (i == lastq_orig) { fl |= F_LASTQ; p.lastq_rxn = d; }
        ri[RI_FLAGS] = fl;
        ri[RI_COLLIDER] = col;

        // enhanced colliders (alpha != 1), first occurrence wins
        ri[RI_EFF_PTR] = (int)p.eff_sp.size();
        double anm1 = 0.0;
        for (int q = eff_ptr[i]; q < eff_ptr[i + 1]; ++q) {
            if (eff[q] == 1.0) continue;
            bool dup = false;
            for (auto& e : effs[d]) dup |= (e.first == eff_sp[q]);
            if (dup) continue;
            effs[d].push_back({eff_sp[q], eff[q] - 1.0});
            p.eff_sp.push_back(eff_sp[q]);
            p.eff_am1.push_back(eff[q] - 1.0);
            if (eff_sp[q] == last) anm1 = eff[q] - 1.0;
        }
        ri[RI_EFF_CNT] = (int)effs[d].size();
        if (!(fl & F_EFFTYPE)) anm1 = 0.0;
        rd[RD_ANM1] = anm1;

        // rate constant (rate_subs.py:27-146); generic form sgn*exp(lnA + b logT - Ta/T)
        auto set_rate = [&](double Av, double bv, double Ev, double* lnA, double* bb, double* Ta, double* sgn) {
            if (Av == 0.0) return false;
            *sgn = Av > 0 ? 1.0 : -1.0;
            *lnA = std::log(std::fabs(Av));
            *bb = bv; *Ta = Ev;
            // reference quirk: A<0, E==0, negative integer b -> plain A
            if (Av < 0 && Ev == 0.0 && bv != 0.0 && bv == std::floor(bv) && bv < 0) *bb = 0.0;
            return true;
        };
        if (!(fl & (F_PLOG | F_CHEB))) {
            if (!set_rate(A[i], b[i], E[i], &rd[RD_LNA], &rd[RD_B], &rd[RD_TA], &rd[RD_SGN])) {
                p.error = "reaction with A == 0"; return false;
            }
        } else {
            rd[RD_SGN] = 1.0;
        }
        rd[RD_NR] = nr;
        rd[RD_NP] = np_;

        ri[RI_KC_PTR] = (int)(p.kcg.size() / KCW);
        if (fl & F_REV) {
            for (int g = kc_ptr[i]; g < kc_ptr[i + 1]; ++g)
                p.kcg.insert(p.kcg.end(), kcg + (size_t)KCW * g, kcg + (size_t)KCW * (g + 1));
            rd[RD_LNPREF] = std::log(kcpref[i]);
        }
        ri[RI_KC_CNT] = (int)(p.kcg.size() / KCW) - ri[RI_KC_PTR];

        ri[RI_PLOG_PTR] = (int)(p.plog.size() / PLW);
        if (fl & F_PLOG) {
            for (int q = plog_ptr[i]; q < plog_ptr[i + 1]; ++q) {
                const double* r = plog + 4 * (size_t)q;
                if (r[1] <= 0.0) { p.error = "PLOG with A <= 0"; return false; }
                p.plog.push_back(plog4[q]);
                p.plog.push_back(std::log(r[0]));
                p.plog.push_back(std::log(r[1]));
                p.plog.push_back(r[2]);
                p.plog.push_back(r[3]);
            }
        }
        ri[RI_PLOG_CNT] = (int)(p.plog.size() / PLW) - ri[RI_PLOG_PTR];
        if (fl & F_CHEB) {
            // Chebyshev record, copied as is (RI_PLOG_PTR / RI_PLOG_CNT are free: never PLOG as well)
            const int32_t* cp = B.ia(IA_CHEB_PTR);
            const double* ch = B.da(DA_CHEB);
            ri[RI_PLOG_PTR] = (int)p.cheb.size();
            ri[RI_PLOG_CNT] = (int)ch[cp[i] + CH_N] * 256 + (int)ch[cp[i] + CH_M];
            p.cheb.insert(p.cheb.end(), ch + cp[i], ch + cp[i + 1]);
        }
        if (fl & F_SRI) {
            if (!(fl & F_PDEP) || (fl & F_TROE)) { p.error = "SRI parameters on a reaction that is not a plain falloff"; return false; }
            ri[RI_PLOG_PTR] = (int)(p.sri.size() / SRW);
            const double* sq = B.da(DA_SRIQ) + (size_t)SRW * i;
            p.sri.insert(p.sri.end(), sq, sq + SRW);
        }

        if (fl & F_PDEP) {
            const double* in = infs + 4 * (size_t)i;
            if (in[0] <= 0.0) { p.error = "falloff with non-positive k0/kinf ratio"; return false; }
            rd[RD_LNAR] = std::log(in[0]); rd[RD_B0] = in[1]; rd[RD_E0] = in[2]; rd[RD_B04] = in[3];
            if (fl & F_TROE) {
                const double* t = troe + 4 * (size_t)i;
                rd[RD_TRA] = t[0]; rd[RD_T3] = t[1]; rd[RD_T1] = t[2]; rd[RD_T2] = t[3];
            }
        }

        // g slots: reactant molecules, product molecules (reversible), collider
        ri[RI_GBASE] = ng;
        for (int r = 0; r < nrs; ++r) gslot_sp[d].push_back(rs[r]);
        if (fl & F_REV) for (int r = 0; r < nps; ++r) gslot_sp[d].push_back(ps[r]);
        if (fl & F_GEN) {
            for (int f = 0; f < ri[RI_GEN_NR]; ++f) gslot_sp[d].push_back(p.gen_sp[ri[RI_GEN_PTR] + f]);
            if (fl & F_REV)
                for (int f = 0; f < ri[RI_GEN_NP]; ++f) gslot_sp[d].push_back(p.gen_sp[ri[RI_GEN_PTR] + ri[RI_GEN_NR] + f]);
        }
        if (fl & F_COLLIDER) gslot_sp[d].push_back(col);
        // enhanced colliders of an [M] with efficiencies: one slot each holding
        // (alpha_ij - 1) b_i, in eff-list order (the last species goes to gN instead)
        if (fl & F_EFFTYPE)
            for (auto& e : effs[d])
                if (e.first != last) gslot_sp[d].push_back(e.first);
        ng += (int)gslot_sp[d].size();

        ri[RI_NET_PTR] = (int)p.net_sp.size();
        for (int q = net_ptr[i]; q < net_ptr[i + 1]; ++q) {
            p.net_sp.push_back(net_sp[q]);
            p.net_nu.push_back(net_nu[q]);
        }
        ri[RI_NET_CNT] = (int)p.net_sp.size() - ri[RI_NET_PTR];
        ri[RI_ORIG] = i;
        ri[RI_REV_IDX] = rev_idx[i];
        ri[RI_PRES_IDX] = pres_idx[i];
    }
    p.ng = ng;

    // ---- V map ----
    VMap& v = p.vm;
    v.nsp = nsp; v.nrxn = nrxn; v.ng = ng;
    int o = 0;
    v.C = o; o += nsp; v.ONE = o; o += 1;
    v.HW = o; o += nsp; v.CP = o; o += nsp; v.YC = o; o += nsp; v.YD = o; o += nsp;
    v.RQ = o; o += nrxn; v.RTH = o; o += nrxn; v.RP = o; o += nrxn; v.RQQ = o; o += nrxn;
    v.G = o; o += ng;
    v.NV = o;
    v.TB = o;
    v.T_OM = 0; v.T_JT = nsp; v.T_P = 2 * nsp; v.T_Q = 3 * nsp; v.T_S = 4 * nsp;
    v.T_JTQ = v.T_S + nsp * (nsp - 1);
    v.T_PART = v.T_JTQ + 1;
    v.NTILE = v.T_PART;
    v.SC = v.TB + v.NTILE;
    v.X = v.SC + SC_COUNT;
    v.NSLOT = v.X + SC_COUNT * nsp;

    // ---- P3: per species gather (device reaction order) ----
    p.sp_ptr.assign(nsp + 1, 0);
    {
        std::vector<std::vector<std::pair<int, double>>> lists(nsp);
        for (int d = 0; d < nrxn; ++d) {
            const int32_t* ri = &p.ri[(size_t)d * RIW];
            for (int q = 0; q < ri[RI_NET_CNT]; ++q)
                lists[p.net_sp[ri[RI_NET_PTR] + q]].push_back({d, p.net_nu[ri[RI_NET_PTR] + q]});
        }
        for (int k = 0; k < nsp; ++k) {
            for (auto& e : lists[k]) { p.sp_rxn.push_back(e.first); p.sp_nu.push_back(e.second); }
            p.sp_ptr[k + 1] = (int)p.sp_rxn.size();
        }
    }

    if (v.NV >= 8192 || nrxn >= 8192) { p.error = "mechanism too large for the 13-bit program encoding"; return false; }

    p.ne = nsp * (nsp - 1);
    // ---- field-major reaction tables ----
    p.nrp = (nrxn + 63) / 64 * 64;
    p.rti.assign((size_t)(RIW + EFF_INL) * p.nrp, 0);
    p.rtd.assign((size_t)(RDW + KCW + EFF_INL) * p.nrp, 0.0);
    for (int d = 0; d < p.nrp; ++d) {
        const int src = d < nrxn ? d : 0;        // padding rows repeat reaction 0 (never stored)
        const int32_t* ri = &p.ri[(size_t)src * RIW];
        const double* rd = &p.rd[(size_t)src * RDW];
        for (int f = 0; f < RIW; ++f) p.rti[(size_t)f * p.nrp + d] = ri[f];
        for (int f = 0; f < RDW; ++f) p.rtd[(size_t)f * p.nrp + d] = rd[f];
        if (ri[RI_KC_CNT] > 0)
            for (int f = 0; f < KCW; ++f) p.rtd[(size_t)(RDW + f) * p.nrp + d] = p.kcg[(size_t)ri[RI_KC_PTR] * KCW + f];
        for (int e = 0; e < EFF_INL; ++e) {
            const bool has = e < ri[RI_EFF_CNT];
            p.rti[(size_t)(RIW + e) * p.nrp + d] = has ? p.eff_sp[ri[RI_EFF_PTR] + e] : ONE;
            p.rtd[(size_t)(RDW + KCW + e) * p.nrp + d] = has ? p.eff_am1[ri[RI_EFF_PTR] + e] : 0.0;
        }
    }

    // ---- scatter-phase terms ----
    for (int d = 0; d < nrxn; ++d) {
        const int32_t* ri = &p.ri[(size_t)d * RIW];
        const int gb = ri[RI_GBASE];
        for (int q = 0; q < ri[RI_NET_CNT]; ++q) {
            const int k = p.net_sp[ri[RI_NET_PTR] + q];
            const double nu = p.net_nu[ri[RI_NET_PTR] + q];
            p.contribs.push_back({v.RQ + d, v.T_OM + k, nu, true});
            p.contribs.push_back({v.RTH + d, v.T_JT + k, nu, true});
            p.contribs.push_back({v.RP + d, v.T_P + k, nu, true});
            p.contribs.push_back({v.RQQ + d, v.T_Q + k, nu, true});
            if (k == last && d == p.lastq_rxn) p.contribs.push_back({v.RTH + d, v.T_JTQ, nu, true});
        }
        for (size_t t = 0; t < gslot_sp[d].size(); ++t) {
            const int j = gslot_sp[d][t];
            if (j >= last) continue;
            for (int q = 0; q < ri[RI_NET_CNT]; ++q)
                p.contribs.push_back({v.G + gb + (int)t, v.T_S + p.net_sp[ri[RI_NET_PTR] + q] + nsp * j,
                                      p.net_nu[ri[RI_NET_PTR] + q], false});
        }
    }

    // ---- compact the sparse block: slots only for structurally non-zero S_kj ----
    {
        const int full = nsp * (nsp - 1);
        std::vector<char> used(full, 0);
        for (auto& c : p.contribs) if (!c.dense) used[c.tgt - v.T_S] = 1;
        p.smap.assign(full, 0xFFFF);
        p.ecol_ptr.assign(nsp, 0);
        int n = 0;
        for (int j = 0; j < nsp - 1; ++j) {
            for (int k = 0; k < nsp; ++k)
                if (used[k + nsp * j]) {
                    p.smap[k + nsp * j] = (uint16_t)n;
                    p.ecol.push_back(((uint32_t)k << 16) | (uint32_t)n);
                    ++n;
                }
            p.ecol_ptr[j + 1] = (int)p.ecol.size();
        }
        if (n >= 0xFFFF) { p.error = "sparse block too large"; return false; }
        p.nnz = n;
        for (auto& c : p.contribs) if (!c.dense) c.tgt = v.T_S + p.smap[c.tgt - v.T_S];
        v.T_JTQ = v.T_S + n;
        for (auto& c : p.contribs) if (c.dense && c.tgt == v.T_S + full) c.tgt = v.T_JTQ;
        v.T_PART = v.T_JTQ + 1;
        v.NTILE = v.T_PART;
        v.SC = v.TB + v.NTILE;
        v.X = v.SC + SC_COUNT;
        v.NSLOT = v.X + SC_COUNT * nsp;
    }

    // net coefficients the 4-bit scatter code cannot express as a whole number in -4..3
    p.n_nutab = 0;
    for (auto& c : p.contribs) {
        if (c.nu == std::floor(c.nu) && c.nu >= -4.0 && c.nu <= 3.0) continue;
        bool have = false;
        for (int t = 0; t < p.n_nutab; ++t) have |= (p.nutab[t] == c.nu);
        if (have) continue;
        if (p.n_nutab >= NUTAB_N) { p.error = "more than 8 distinct fractional / large net stoichiometric coefficients"; return false; }
        p.nutab[p.n_nutab++] = c.nu;
    }
    return true;
}

bool build_schedule(Programs& p, int NW, int IL, Schedule& out)
{
    if (NW < 1 || NW > 16 || IL < 1 || IL > 64) { p.error = "bad schedule geometry"; return false; }
    constexpr int GW = 4;            // rounds applied together by the kernel (independent read-modify-writes)
    VMap& v = p.vm;
    out = Schedule();
    out.NW = NW; out.IL = IL;
    const int nt0 = v.T_PART;
    std::vector<std::vector<int>> by_tgt(nt0);
    for (int c = 0; c < (int)p.contribs.size(); ++c) by_tgt[p.contribs[c].tgt].push_back(c);
    // A target may appear once per GROUP of GW rounds (so the GW updates of a lane and of
    // its neighbours never alias).  Targets with more terms than a balanced wavefront has
    // groups are split into partial accumulators that phase_fin1 adds up afterwards.
    long nd = 0, ns = 0;
    for (auto& c : p.contribs) (c.dense ? nd : ns)++;
    const int Rd = (int)std::max<long>(4, (nd + (long)NW * IL * GW - 1) / ((long)NW * IL * GW));
    const int Rs = (int)std::max<long>(4, (ns + (long)NW * IL * GW - 1) / ((long)NW * IL * GW));
    struct Tgt { int slot; std::vector<int> terms; bool dense; };
    std::vector<Tgt> tg;
    int nextp = v.T_PART;
    for (int t = 0; t < nt0; ++t) {
        auto& l = by_tgt[t];
        if (l.empty()) continue;
        const bool dense = p.contribs[l[0]].dense;
        const int R = dense ? Rd : Rs;
        if ((int)l.size() <= R) { tg.push_back({t, l, dense}); continue; }
        const int parts = ((int)l.size() + R - 1) / R;
        for (int q = 0; q < parts; ++q) {
            Tgt x{q == 0 ? t : nextp, {}, dense};
            for (int e = q; e < (int)l.size(); e += parts) x.terms.push_back(l[e]);
            if (q == 1) { out.fin_tgt.push_back(t); out.fin_part.push_back(nextp); out.fin_cnt.push_back(parts - 1); }
            if (q > 0) ++nextp;
            tg.push_back(std::move(x));
        }
    }
    v.NTILE = nextp;
    v.SC = v.TB + v.NTILE;
    v.X = v.SC + SC_COUNT;
    v.NSLOT = v.X + SC_COUNT * p.nsp;
    if (v.NTILE >= 32768 || v.NV >= 8192) { p.error = "mechanism too large for the scatter encoding"; return false; }
    // ownership: heaviest target to the least loaded wavefront; dense and sparse balanced separately
    std::vector<std::vector<int>> own_d(NW), own_s(NW);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<int> idx;
        for (int i = 0; i < (int)tg.size(); ++i) if (tg[i].dense == (pass == 0)) idx.push_back(i);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return tg[a].terms.size() > tg[b].terms.size(); });
        std::vector<long> load(NW, 0);
        for (int i : idx) {
            int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            (pass == 0 ? own_d : own_s)[w].push_back(i);
            load[w] += (long)tg[i].terms.size();
        }
    }
    // groups: up to GW*IL distinct targets per group (most remaining terms first), laid out as
    // GW rounds of IL codes
    auto emit = [&](const std::vector<int>& owned, std::vector<uint32_t>& codes) {
        std::vector<std::pair<int, int>> rem;
        std::vector<size_t> pos(owned.size(), 0);
        int rounds = 0;
        for (;;) {
            rem.clear();
            for (int i = 0; i < (int)owned.size(); ++i) {
                const int left = (int)tg[owned[i]].terms.size() - (int)pos[i];
                if (left > 0) rem.push_back({left, i});
            }
            if (rem.empty()) break;
            const size_t take = std::min<size_t>((size_t)GW * IL, rem.size());
            std::partial_sort(rem.begin(), rem.begin() + take, rem.end(),
                              [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
            for (int l = 0; l < GW * IL; ++l) {
                uint32_t code = 4u << 28;        // no-op (nu = 0)
                if (l < (int)take) {
                    const int i = rem[l].second;
                    const Contrib& c = p.contribs[tg[owned[i]].terms[pos[i]++]];
                    uint32_t nuc = 0;
                    if (c.nu == std::floor(c.nu) && c.nu >= -4.0 && c.nu <= 3.0) nuc = (uint32_t)((int)c.nu + 4);
                    else for (int t = 0; t < p.n_nutab; ++t) if (p.nutab[t] == c.nu) nuc = 8u + (uint32_t)t;
                    code = (uint32_t)c.src | ((uint32_t)tg[owned[i]].slot << 13) | (nuc << 28);
                }
                codes.push_back(code);
            }
            rounds += GW;
        }
        return rounds;
    };
    for (int w = 0; w < NW; ++w) {
        out.off[w] = (int)out.codes.size();
        out.rounds_dense[w] = emit(own_d[w], out.codes);
        out.rounds[w] = out.rounds_dense[w] + emit(own_s[w], out.codes);
    }
    return true;
}

uint64_t programs_hash(const Programs& p)
{
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](const void* d, size_t n) {
        const unsigned char* b = (const unsigned char*)d;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    };
    int hdr[4] = {p.nsp, p.nrxn, p.ng, p.lastq_rxn};
    mix(hdr, sizeof(hdr));
    mix(p.sp.data(), p.sp.size() * 8); mix(p.ri.data(), p.ri.size() * 4); mix(p.rd.data(), p.rd.size() * 8);
    mix(p.eff_sp.data(), p.eff_sp.size() * 4); mix(p.eff_am1.data(), p.eff_am1.size() * 8);
    mix(p.kcg.data(), p.kcg.size() * 8); mix(p.plog.data(), p.plog.size() * 8);
    if (!p.sri.empty()) mix(p.sri.data(), p.sri.size() * 8);        // (hashes of mechanisms without them unchanged)
    if (!p.cheb.empty()) mix(p.cheb.data(), p.cheb.size() * 8);
    mix(p.net_sp.data(), p.net_sp.size() * 4); mix(p.net_nu.data(), p.net_nu.size() * 8);
    mix(p.smap.data(), p.smap.size() * 2);
    if (!p.gen_sp.empty()) { mix(p.gen_sp.data(), p.gen_sp.size() * 4); mix(p.gen_nu.data(), p.gen_nu.size() * 8); }
    return h;
}

std::string emit_spec_header(const Programs& p)
{
    std::string o;
    char buf[64];
    auto d = [&](double x) { snprintf(buf, sizeof buf, "%a", x); o += buf; };
    auto arr_d = [&](const char* name, const std::vector<double>& v, int width) {
        const size_t rows = v.size() / width;
        o += "constexpr double "; o += name; o += "[" + std::to_string(rows ? rows : 1) + "][" + std::to_string(width) + "] = {";
        if (!rows) o += "{}";
        for (size_t r = 0; r < rows; ++r) {
            o += "{";
            for (int c = 0; c < width; ++c) { d(v[r * width + c]); o += ","; }
            o += "},\n";
        }
        o += "};\n";
    };
    auto arr_i = [&](const char* name, const std::vector<int32_t>& v, int width) {
        const size_t rows = v.size() / width;
        o += "constexpr int "; o += name; o += "[" + std::to_string(rows ? rows : 1) + "][" + std::to_string(width) + "] = {";
        if (!rows) o += "{}";
        for (size_t r = 0; r < rows; ++r) {
            o += "{";
            for (int c = 0; c < width; ++c) o += std::to_string(v[r * width + c]) + ",";
            o += "},\n";
        }
        o += "};\n";
    };
    const int nsp = p.nsp;
    snprintf(buf, sizeof buf, "0x%016llxULL", (unsigned long long)programs_hash(p));
    o += "// generated by pj::emit_spec_header -- mechanism constants for pj_lane.hip\n#pragma once\n";
    o += std::string("#define PJS_HASH ") + buf + "\nnamespace pjs {\n";
    // sparse-entry index map: [k][j] -> slot or -1
    std::vector<int32_t> sidx((size_t)nsp * nsp, -1);
    int nnz = 0;
    for (int j = 0; j < nsp - 1; ++j)
        for (int k = 0; k < nsp; ++k)
            if (p.smap[k + nsp * j] != 0xFFFF) sidx[(size_t)k * nsp + j] = nnz++;
    o += "constexpr int NSP = " + std::to_string(nsp) + ", NRXN = " + std::to_string(p.nrxn) +
         ", NNZ = " + std::to_string(nnz ? nnz : 1) + ", LASTQ = " + std::to_string(p.lastq_rxn) + ";\n";
    arr_d("SP", p.sp, SPW);
    arr_i("RI", p.ri, RIW);
    arr_d("RD", p.rd, RDW);
    arr_i("EFF_SP", p.eff_sp, 1);
    arr_d("EFF_AM1", p.eff_am1, 1);
    arr_d("KCG", p.kcg, KCW);
    arr_d("PLOG", p.plog, PLW);
    arr_d("SRI", p.sri, SRW);
    arr_d("CHEB", p.cheb, 1);
    arr_i("NET_SP", p.net_sp, 1);
    arr_d("NET_NU", p.net_nu, 1);
    arr_i("GEN_SP", p.gen_sp, 1);      // F_GEN reactions: (species, nu) factors, RI_GEN_PTR / RI_GEN_NR / RI_GEN_NP
    arr_d("GEN_NU", p.gen_nu, 1);
    arr_i("SIDX", sidx, nsp);
    {
        // reactions with identical equilibrium constants (same net stoichiometry: third-body
        // variants, duplicates) share one class: exp(-ln Kc) is evaluated once per class
        std::vector<int32_t> cls(p.nrxn, -1), first(p.nrxn, 0);
        int ncls = 0;
        for (int i = 0; i < p.nrxn; ++i) {
            const int32_t* ri = &p.ri[(size_t)i * RIW];
            if (!(ri[RI_FLAGS] & F_REV)) continue;
            for (int h = 0; h < i && cls[i] < 0; ++h) {
                const int32_t* rh = &p.ri[(size_t)h * RIW];
                if (!(rh[RI_FLAGS] & F_REV) || rh[RI_KC_CNT] != ri[RI_KC_CNT]) continue;
                if (p.rd[(size_t)h * RDW + RD_LNPREF] != p.rd[(size_t)i * RDW + RD_LNPREF]) continue;
                if (memcmp(&p.kcg[(size_t)rh[RI_KC_PTR] * KCW], &p.kcg[(size_t)ri[RI_KC_PTR] * KCW],
                           sizeof(double) * KCW * ri[RI_KC_CNT]) == 0) cls[i] = cls[h];
            }
            if (cls[i] < 0) { cls[i] = ncls++; first[i] = 1; }
        }
        o += "constexpr int NKCCLS = " + std::to_string(ncls ? ncls : 1) + ";\n";
        arr_i("KC_CLASS", cls, 1);
        arr_i("KC_FIRST", first, 1);
    }
    // NASA range tables for LDS: one 16-double row per K_c group, then per species:
    // [lo0..lo6, 0, hi0..hi6, 0]; the kernel reads 7 doubles at row*16 + (T <= Tmid ? 0 : 8)
    const size_t nkc = p.kcg.size() / KCW;
    o += "constexpr int LT_KC = 0, LT_SP = " + std::to_string(nkc * 16) + ", LT_SIZE = " +
         std::to_string((nkc + nsp) * 16) + ";\n";
    // coefficient tables read through the scalar cache (s_load) instead of being
    // materialised as 64-bit literals: reaction doubles, efficiencies, species constants
    auto dev_arr = [&](const char* name, const std::vector<double>& v, int width) {
        const size_t rows = v.size() / width;
        o += "__constant__ const double "; o += name;
        o += "[" + std::to_string(rows ? rows : 1) + "][" + std::to_string(width) + "] = {";
        if (!rows) o += "{}";
        for (size_t r = 0; r < rows; ++r) {
            o += "{";
            for (int c = 0; c < width; ++c) { d(v[r * width + c]); o += ","; }
            o += "},\n";
        }
        o += "};\n";
    };
    o += "#ifdef __HIPCC__\n";
    dev_arr("RDT", p.rd, RDW);
    dev_arr("EFFT", p.eff_am1, 1);
    dev_arr("PLOGT", p.plog, PLW);
    {
        std::vector<double> spc;
        for (int k = 0; k < nsp; ++k) for (int c = 0; c < 4; ++c) spc.push_back(p.sp[(size_t)k * SPW + c]);
        dev_arr("SPT", spc, 4);
        dev_arr("SPF", p.sp, SPW);      // full species records (kept for tools)
        std::vector<double> invw;       // 1 / W_j, contiguous: the per-column constant of the row kernels' output phase
        for (int k = 0; k < nsp; ++k) invw.push_back(p.sp[(size_t)k * SPW]);
        dev_arr("INVWT", invw, 1);
    }
    o += "#endif\n";
    o += "#ifdef __HIPCC__\n__device__ __attribute__((aligned(16))) const double LTAB[LT_SIZE] = {\n";
    auto row = [&](const double* lo, const double* hi) {
        for (int c = 0; c < 7; ++c) { d(lo[c]); o += ","; }
        o += "0,";
        for (int c = 0; c < 7; ++c) { d(hi[c]); o += ","; }
        o += "0,\n";
    };
    for (size_t g = 0; g < nkc; ++g) row(&p.kcg[g * KCW + 1], &p.kcg[g * KCW + 8]);
    for (int k = 0; k < nsp; ++k) row(&p.sp[(size_t)k * SPW + 4], &p.sp[(size_t)k * SPW + 11]);
    o += "};\n#endif\n";
    o += "}  // namespace pjs\n";
    return o;
}

// Row-block partition for pj_rblk.hip (state-per-lane kernels for mechanisms whose sparse
// block does not fit the register file at once): species rows are grouped greedily so that a
// group's accumulators (4 dense + its structurally non-zero S entries per row) fit `budget`
// doubles, preferring rows that share reactions (each group visits every reaction that
// touches one of its rows).  Also numbers the per-reaction values the rate kernel hands to
// the row kernels through the HBM scratch array.
std::string emit_rows_tables(const Programs& p, int budget, const RblkPlanOpts* plan_opts, RblkPlan* plan_out)
{
    const int nsp = p.nsp, nrxn = p.nrxn;
    std::string o;
    auto arr_i = [&](const char* name, const std::vector<int32_t>& v, int width) {
        const size_t rows = v.size() / width;
        o += "constexpr int "; o += name; o += "[" + std::to_string(rows ? rows : 1) + "][" + std::to_string(width) + "] = {";
        if (!rows) o += "{}";
        for (size_t r = 0; r < rows; ++r) {
            o += "{";
            for (int c = 0; c < width; ++c) o += std::to_string(v[r * width + c]) + ",";
            o += "},\n";
        }
        o += "};\n";
    };
    // structural pattern of S and the reactions touching each row
    std::vector<int> nnz_row(nsp, 0);
    for (int j = 0; j < nsp - 1; ++j)
        for (int k = 0; k < nsp; ++k)
            if (p.smap[k + nsp * j] != 0xFFFF) ++nnz_row[k];
    std::vector<std::vector<int>> rx_of(nsp);
    for (int i = 0; i < nrxn; ++i) {
        const int32_t* ri = &p.ri[(size_t)i * RIW];
        for (int q = 0; q < ri[RI_NET_CNT]; ++q) rx_of[p.net_sp[ri[RI_NET_PTR] + q]].push_back(i);
    }
    const int DENSE = 4;      // omega_k, P_k, Q_k and (pj_rblk.hip) sum nu theta per row
    std::vector<int> blk_of(nsp, -1);
    std::vector<std::vector<int>> blocks;
    std::vector<std::vector<char>> blk_rx;
    int left = nsp;
    while (left > 0) {
        int seed = -1;
        for (int k = 0; k < nsp; ++k)
            if (blk_of[k] < 0 && (seed < 0 || rx_of[k].size() > rx_of[seed].size())) seed = k;
        std::vector<int> blk{seed};
        std::vector<char> rxs(nrxn, 0);
        for (int i : rx_of[seed]) rxs[i] = 1;
        blk_of[seed] = (int)blocks.size(); --left;
        int cost = nnz_row[seed] + DENSE;
        for (;;) {
            int best = -1; double bscore = -1e300;
            for (int k = 0; k < nsp; ++k) {
                if (blk_of[k] >= 0 || cost + nnz_row[k] + DENSE > budget) continue;
                int shared = 0;
                for (int i : rx_of[k]) shared += rxs[i];
                const double score = shared - 0.3 * ((int)rx_of[k].size() - shared);
                if (score > bscore) { bscore = score; best = k; }
            }
            if (best < 0) break;
            blk.push_back(best); blk_of[best] = (int)blocks.size(); --left;
            cost += nnz_row[best] + DENSE;
            for (int i : rx_of[best]) rxs[i] = 1;
        }
        std::sort(blk.begin(), blk.end());
        blocks.push_back(blk);
        blk_rx.push_back(rxs);
    }
    const int nblk = (int)blocks.size();
    std::vector<int32_t> row_blk(nsp), rowloc(nsp), brow_ptr{0}, brows, brx_ptr{0}, brx, bnnz(nblk, 0);
    std::vector<int32_t> sloc((size_t)nsp * nsp, -1);
    int maxrows = 0, maxnnz = 1;
    for (int b = 0; b < nblk; ++b) {
        int loc = 0;
        for (int k : blocks[b]) {
            row_blk[k] = b; rowloc[k] = loc++;
            brows.push_back(k);
            for (int j = 0; j < nsp - 1; ++j)
                if (p.smap[k + nsp * j] != 0xFFFF) sloc[(size_t)k * nsp + j] = bnnz[b]++;
        }
        brow_ptr.push_back((int32_t)brows.size());
        // falloff / PLOG reactions last: pj_rblk.hip reads their hand-over values from memory, and a
        // load issued at the top of a block sits behind the previous block's Jacobian stores
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < nrxn; ++i)
                if (blk_rx[b][i] && ((p.ri[(size_t)i * RIW + RI_FLAGS] & (F_PDEP | F_PLOG | F_CHEB)) != 0) == (pass == 1))
                    brx.push_back(i);
        brx_ptr.push_back((int32_t)brx.size());
        maxrows = std::max(maxrows, loc);
        maxnnz = std::max(maxnnz, (int)bnnz[b]);
    }
    // scratch slots per reaction: (theta: stays in the rate kernel), c*kf, c*kr, rp, bM, bcol
    // (-1: not stored)
    std::vector<int32_t> scr((size_t)nrxn * 6, -1);
    int nscr = 0;
    for (int i = 0; i < nrxn; ++i) {
        const int fl = p.ri[(size_t)i * RIW + RI_FLAGS];
        scr[(size_t)i * 6 + 1] = nscr++;
        if (fl & F_REV) scr[(size_t)i * 6 + 2] = nscr++;
        if (fl & (F_THD | F_PDEP)) scr[(size_t)i * 6 + 3] = nscr++;
        if (fl & F_EFFTYPE) scr[(size_t)i * 6 + 4] = nscr++;
        if (fl & F_COLLIDER) scr[(size_t)i * 6 + 5] = nscr++;
    }
    // pj_rblk.hip: only falloff / PLOG reactions are handed over (a pre-pass evaluates them once per
    // state); everything else is rebuilt from T and the concentrations at every visit.
    // slots: theta, c*kf, (c*kr: never, rebuilt from K_c; Chebyshev: eval_jacob's second k_f), rp, bM, bcol
    std::vector<int32_t> scq((size_t)nrxn * 6, -1);
    int nscq = 0, npre = 0;
    for (int i = 0; i < nrxn; ++i) {
        const int fl = p.ri[(size_t)i * RIW + RI_FLAGS];
        if (!(fl & (F_PDEP | F_PLOG | F_CHEB))) continue;
        ++npre;
        scq[(size_t)i * 6 + 0] = nscq++;
        scq[(size_t)i * 6 + 1] = nscq++;
        if (fl & F_CHEB) scq[(size_t)i * 6 + 2] = nscq++;      // the Jacobian's own k_f (pj_rate_pre.inc)
        if (fl & (F_THD | F_PDEP)) scq[(size_t)i * 6 + 3] = nscq++;
        if (fl & F_EFFTYPE) scq[(size_t)i * 6 + 4] = nscq++;
        if (fl & F_COLLIDER) scq[(size_t)i * 6 + 5] = nscq++;
    }
    // fused kernel: row blocks dealt to 4 wavefronts, longest-processing-time first, with the
    // measured cost model (cycles): ~770 per reaction visit, ~14000 per row of Jacobian stores
    const int NARM = 4;
    std::vector<std::vector<int>> arm_blks(NARM);
    {
        std::vector<long> cost(nblk), load(NARM, 0);
        std::vector<int> order(nblk);
        for (int b = 0; b < nblk; ++b) {
            cost[b] = 770L * (brx_ptr[b + 1] - brx_ptr[b]) + 14000L * (long)blocks[b].size() + 2000;
            order[b] = b;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        for (int b : order) {
            const int a = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            arm_blks[a].push_back(b);
            load[a] += cost[b];
        }
    }
    std::vector<int32_t> arm_ptr{0}, arm_list;
    for (int a = 0; a < NARM; ++a) {
        std::sort(arm_blks[a].begin(), arm_blks[a].end());
        for (int b : arm_blks[a]) arm_list.push_back(b);
        arm_ptr.push_back((int32_t)arm_list.size());
    }
    o += "namespace pjs {\n";
    o += "constexpr int NBLK = " + std::to_string(nblk) + ", BLK_MAXROWS = " + std::to_string(maxrows) +
         ", BLK_MAXNNZ = " + std::to_string(maxnnz) + ", NSCR = " + std::to_string(nscr) +
         ", NVISIT = " + std::to_string(brx.size()) + ";\n";
    arr_i("ROW_BLK", row_blk, 1);
    arr_i("ROWLOC", rowloc, 1);
    arr_i("BLK_ROW_PTR", brow_ptr, 1);
    arr_i("BLK_ROWS", brows, 1);
    arr_i("BLK_RX_PTR", brx_ptr, 1);
    arr_i("BLK_RX", brx, 1);
    arr_i("BLK_NNZ", bnnz, 1);
    arr_i("SLOC", sloc, nsp);
    arr_i("SCR", scr, 6);
    o += "constexpr int NSCQ = " + std::to_string(nscq) + ", NPRE = " + std::to_string(npre) + ";\n";
    arr_i("SCQ", scq, 6);
    arr_i("ARM_BLK_PTR", arm_ptr, 1);
    arr_i("ARM_BLKS", arm_list, 1);
    {
        // equilibrium constants from per-species factors (PJQ_KCF kernels): shifted rows of ln X_k and the
        // per-reaction constant (p_atm / R_u)^(-sum nu) = exp(-RD_LNPREF)
        const bool kcf = p.kcf.size() == (size_t)nsp * KCW;
        char buf[64];
        o += std::string("constexpr int KCF_OK = ") + (kcf ? "1" : "0") + ";\n";
        o += "constexpr double KCF_ROW[" + std::to_string(nsp) + "][" + std::to_string(KCW) + "] = {";
        for (int k = 0; k < nsp; ++k) {
            o += "{";
            for (int c = 0; c < KCW; ++c) { snprintf(buf, sizeof buf, "%a", kcf ? p.kcf[(size_t)k * KCW + c] : 0.0); o += buf; o += ","; }
            o += "},\n";
        }
        o += "};\nconstexpr double KCF_PREFINV[" + std::to_string(nrxn ? nrxn : 1) + "][1] = {";
        if (!nrxn) o += "{}";
        for (int i = 0; i < nrxn; ++i) {
            snprintf(buf, sizeof buf, "%a", std::exp(-p.rd[(size_t)i * RDW + RD_LNPREF]));
            o += std::string("{") + buf + "},";
        }
        o += "};\n";
    }
    if (plan_opts) {
        // ---- kernel plan of a pj_rblk.hip library (which row blocks / reactions each kernel takes) ----
        const RblkPlanOpts& O = *plan_opts;
        const int halves = (O.halves == 2 || O.halves == 4 || O.halves == 8) ? O.halves : 1, fuse = O.fuse > 0 ? O.fuse : 13;
        // K_c groups a row block / a reaction needs (a kernel stages their polynomial rows in LDS, 128 bytes each)
        auto groups_of_rxn = [&](int i, std::vector<char>& g) {
            const int32_t* ri = &p.ri[(size_t)i * RIW];
            if (ri[RI_FLAGS] & F_REV)
                for (int c = 0; c < ri[RI_KC_CNT]; ++c) g[ri[RI_KC_PTR] + c] = 1;
        };
        const int ngrp = (int)(p.kcg.size() / KCW);
        auto count = [](const std::vector<char>& g) { int c = 0; for (char x : g) c += x; return c; };
        std::vector<int32_t> kb{0}, km;
        if (O.single) {
            // one row kernel: state read once, no hand-over of the energy-row sums, one prologue
            kb.push_back(nblk);
        } else if (halves == 1) {
            // row kernels of (nearly) equal block counts, at most `fuse` blocks each
            const int nker = (nblk + fuse - 1) / fuse;
            kb.clear();
            for (int i = 0; i <= nker; ++i) kb.push_back((int32_t)((long)nblk * i / nker));
        } else {
            // several lane groups per workgroup: a kernel spans up to halves * fuse blocks, and the K_c rows of all of
            // them sit next to the concentration columns -- kernels are cut where one more block's rows would not fit
            const long limit = (160L * 1024 - (long)nsp * O.block * 8) / 128 - 2;
            std::vector<char> cur(ngrp + 1, 0);
            for (int b = 0; b < nblk; ++b) {
                std::vector<char> both = cur;
                for (int v = brx_ptr[b]; v < brx_ptr[b + 1]; ++v) groups_of_rxn(brx[v], both);
                if (b > kb.back() && (count(both) > limit || b - kb.back() >= halves * fuse)) {
                    kb.push_back(b);
                    std::fill(cur.begin(), cur.end(), 0);
                    for (int v = brx_ptr[b]; v < brx_ptr[b + 1]; ++v) groups_of_rxn(brx[v], cur);
                } else {
                    cur = both;
                }
            }
            if (nblk - kb.back() < halves && kb.size() > 1) kb.pop_back();      // a kernel needs a block per lane group
            kb.push_back(nblk);
        }
        const int nker = (int)kb.size() - 1;
        // lane groups: the blocks of a kernel are cut where the groups' estimated times meet
        // (a visit ~ cost_visit, a Jacobian entry of the output phase ~ cost_entry).  KER_BM: the one boundary of a
        // two-group kernel (kept for those builds); KER_GB: [kernel][group] first block, halves + 1 entries per kernel
        std::vector<int32_t> kgb;
        for (int i = 0; i < nker; ++i) {
            const int b0 = kb[i], b1 = kb[i + 1];
            std::vector<double> cost;
            double tot = 0.0;
            for (int b = b0; b < b1; ++b) {
                cost.push_back(O.cost_visit * (brx_ptr[b + 1] - brx_ptr[b]) + O.cost_entry * nsp * (brow_ptr[b + 1] - brow_ptr[b]));
                tot += cost.back();
            }
            std::vector<int> cut{b0};
            double acc = 0.0;
            int b = b0;
            for (int g = 1; g < halves; ++g) {
                const double want = tot * g / halves;
                // first block of group g: where the running cost passes g / halves of the total (the closer side)
                while (b < b1 && acc + cost[b - b0] <= want) acc += cost[b++ - b0];
                if (b < b1 && want - acc > 0.5 * cost[b - b0]) acc += cost[b++ - b0];
                int c = std::max(b, cut.back() + (b1 - b0 >= halves ? 1 : 0));
                c = std::min(c, b1 - (b1 - b0 >= halves ? (halves - g) : 0));
                while (b < c) acc += cost[b++ - b0];
                cut.push_back(c);
            }
            cut.push_back(b1);
            for (int c : cut) kgb.push_back(c);
            km.push_back(halves == 2 ? cut[1] : b1);
        }
        // rate kernels: reaction ranges whose K_c rows fit the LDS (next to the concentration columns, if those
        // are in LDS), at most rate_groups groups each
        long rlimit = (160L * 1024 - (O.rate_c_lds ? (long)nsp * O.rate_block * 8 : 0) - 2048) / 128;
        if (O.rate_groups > 0 && O.rate_groups < rlimit) rlimit = O.rate_groups;
        std::vector<int32_t> rb{0};
        {
            std::vector<char> cur(ngrp + 1, 0);
            for (int i = 0; i < nrxn; ++i) {
                std::vector<char> both = cur;
                groups_of_rxn(i, both);
                if (i > rb.back() && count(both) > rlimit) {
                    rb.push_back(i);
                    std::fill(cur.begin(), cur.end(), 0);
                    groups_of_rxn(i, cur);
                } else {
                    cur = both;
                }
            }
            rb.push_back(nrxn);
        }
        o += "constexpr int NKER = " + std::to_string(nker) + ", NRATE = " + std::to_string(rb.size() - 1) + ";\n";
        arr_i("KER_B", kb, 1);
        arr_i("KER_BM", km, 1);
        o += "constexpr int KER_NG = " + std::to_string(halves) + ";\n";
        arr_i("KER_GB", kgb, halves + 1);
        arr_i("RATE_R", rb, 1);
        if (plan_out) { plan_out->n_row_kernels = nker; plan_out->n_rate_kernels = (int)rb.size() - 1; plan_out->n_pre = npre; plan_out->n_blocks = nblk; plan_out->n_visits = (int)brx.size(); }
    }
    o += "}  // namespace pjs\n";
    return o;
}

}  // namespace pj
