// pj_tabprog.cpp -- host builder of the k_tab program (pj_tabprog.h).
#include "pj_tabprog.h"

#include <algorithm>
#include <cstring>
#include <numeric>

namespace pj {

namespace {
constexpr int DENSE = TAB_DENSE;      // omega_k, P_k, Q_k, sum nu theta, and the J_nplusone quirk value per row
struct Part { int k; std::vector<int> cols; bool first; };           // a row or a column part of a row
struct Block { std::vector<Part> parts; std::vector<int> rx; long cost = 0; };
}  // namespace

bool build_tab_program(const Programs& p, size_t lds_avail, TabProg& out, int force_L)
{
    out = TabProg();
    const int nsp = p.nsp, nrxn = p.nrxn, last = nsp - 1, ONE = nsp;
    auto fail = [&](const char* w) { out.error = w; return false; };
    if (nsp < 2) return fail("k_tab: at least two species");
    // ---- geometry: L states per workgroup, B accumulator slots per lane group ----
    int L = 0, B = 0;
    const long ring = 4L * TAB_RING_WORDS * 8;       // one ring per wavefront
    for (int cand : {256, 128, 64}) {
        const long room = (long)lds_avail - (long)nsp * cand * 8 - ring;
        const int b = room > 0 ? (int)(room / (256L * 8)) : 0;
        // (measured, GRI-shaped: 64 states x 4 one-wavefront groups with 55 slots 52 ms, 128 x 2 with 41 slots 72 ms)
        if (force_L ? cand == force_L : (b >= 56 || cand == 64)) { L = cand; B = b; break; }
    }
    if (!L) return fail("k_tab: states per workgroup must be 256, 128 or 64");
    if (B > 96) B = 96;
    if (B < DENSE + 3 + 2) return fail("k_tab: the concentration columns of this mechanism leave no room for accumulators in LDS");
    const int cap = B - 2;                       // slots a block may use (ZERO and TRASH are the last two)
    out.L = L; out.G = 256 / L; out.B = B; out.ZERO = B - 2; out.TRASH = B - 1;
    out.lds_bytes = (size_t)nsp * L * 8 + (size_t)256 * B * 8 + (size_t)ring;

    // ---- structural pattern, reactions per row ----
    auto nz = [&](int k, int j) { return p.smap[(size_t)k + (size_t)nsp * j] != 0xFFFF; };
    std::vector<std::vector<int>> cols_of(nsp), rx_of(nsp);
    for (int k = 0; k < nsp; ++k)
        for (int j = 0; j < last; ++j) if (nz(k, j)) cols_of[k].push_back(j);
    for (int d = 0; d < nrxn; ++d) {
        const int32_t* ri = &p.ri[(size_t)d * RIW];
        for (int q = 0; q < ri[RI_NET_CNT]; ++q) rx_of[p.net_sp[ri[RI_NET_PTR] + q]].push_back(d);
    }
    // ---- partition: rows that do not fit are split into column parts, the rest grouped greedily by shared
    //      reactions (as pj::emit_rows_tables does for pj_rblk.hip) ----
    std::vector<Block> blocks;
    std::vector<char> done(nsp, 0);
    for (int k = 0; k < nsp; ++k) {
        if (DENSE + (int)cols_of[k].size() <= cap) continue;
        const int per = cap - DENSE;
        for (size_t c0 = 0; c0 < cols_of[k].size(); c0 += per) {
            Block b;
            Part pt{k, {}, c0 == 0};
            for (size_t c = c0; c < std::min(cols_of[k].size(), c0 + per); ++c) pt.cols.push_back(cols_of[k][c]);
            b.parts.push_back(pt);
            b.rx = rx_of[k];
            blocks.push_back(b);
        }
        done[k] = 1;
    }
    for (;;) {
        int seed = -1;
        for (int k = 0; k < nsp; ++k)
            if (!done[k] && (seed < 0 || rx_of[k].size() > rx_of[seed].size())) seed = k;
        if (seed < 0) break;
        Block b;
        std::vector<char> rxs(nrxn, 0);
        auto add = [&](int k) {
            b.parts.push_back(Part{k, cols_of[k], true});
            for (int d : rx_of[k]) rxs[d] = 1;
            done[k] = 1;
        };
        add(seed);
        int used = DENSE + (int)cols_of[seed].size();
        for (;;) {
            int best = -1; double bscore = -1e300;
            for (int k = 0; k < nsp; ++k) {
                if (done[k] || used + DENSE + (int)cols_of[k].size() > cap) continue;
                int shared = 0;
                for (int d : rx_of[k]) shared += rxs[d];
                const double score = shared - 0.3 * ((int)rx_of[k].size() - shared);
                if (score > bscore) { bscore = score; best = k; }
            }
            if (best < 0) break;
            used += DENSE + (int)cols_of[best].size();
            add(best);
        }
        std::sort(b.parts.begin(), b.parts.end(), [](const Part& x, const Part& y) { return x.k < y.k; });
        for (int d = 0; d < nrxn; ++d) if (rxs[d]) b.rx.push_back(d);
        blocks.push_back(b);
    }
    for (auto& b : blocks) b.cost = 220L * (long)b.rx.size() + 8L * nsp * (long)b.parts.size() + 400;
    const int nblk = (int)blocks.size();
    out.nblk = nblk;

    // ---- groups: longest processing time first ----
    std::vector<std::vector<int>> gb(out.G);
    {
        std::vector<int> order(nblk);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return blocks[a].cost > blocks[c].cost; });
        std::vector<long> load(out.G, 0);
        for (int b : order) {
            const int g = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            gb[g].push_back(b);
            load[g] += blocks[b].cost;
        }
        for (auto& l : gb) std::sort(l.begin(), l.end());
    }

    // ---- emit: one record stream per lane group, in execution order ----
    // (8-byte words; a record is header[3] + integer words + doubles, at most TAB_RSZ words: a wavefront moves it
    // into its LDS ring with one to three 512-byte loads, two records ahead of use)
    struct Rec { int type = 0, flags = 0, a1 = 0; std::vector<int32_t> iw; std::vector<double> dw; };
    std::vector<double>& S = out.D;
    auto words_of = [](const Rec& r) { return 3 + (int)((r.iw.size() + 1) / 2) + (int)r.dw.size(); };
    auto put_rec = [&](std::vector<Rec>& list, Rec&& r) -> bool {
        if (words_of(r) > TAB_RSZ) return false;
        list.push_back(std::move(r));
        return true;
    };
    std::vector<int32_t> grp;        // per group: first word, records, words of the first and of the second record
    for (int g = 0; g < out.G; ++g) {
        std::vector<Rec> recs;
        for (int b : gb[g]) {
            const Block& bl = blocks[b];
            // slots: DENSE per part, then the parts' S columns
            std::vector<int> base(bl.parts.size());
            std::vector<std::vector<int>> slot_of(bl.parts.size(), std::vector<int>(nsp, -1));
            int next = DENSE * (int)bl.parts.size();
            for (size_t r = 0; r < bl.parts.size(); ++r) {
                base[r] = DENSE * (int)r;
                for (int j : bl.parts[r].cols) slot_of[r][j] = next++;
            }
            if (next > cap) return fail("k_tab: internal error, block exceeds the accumulator budget");
            { Rec r; r.type = TAB_T_BEGIN; r.a1 = next; put_rec(recs, std::move(r)); }
            for (int d : bl.rx) {
                const int32_t* ri = &p.ri[(size_t)d * RIW];
                const int fl = ri[RI_FLAGS];
                std::vector<std::pair<int, double>> hits;      // (part index, nu_k)
                for (size_t r = 0; r < bl.parts.size(); ++r)
                    for (int q = 0; q < ri[RI_NET_CNT]; ++q)
                        if (p.net_sp[ri[RI_NET_PTR] + q] == bl.parts[r].k) hits.push_back({(int)r, p.net_nu[ri[RI_NET_PTR] + q]});
                int sp[TAB_NSLOT];
                for (int t = 0; t < 3; ++t) { sp[t] = ri[RI_R0 + t]; sp[3 + t] = (fl & F_REV) ? ri[RI_P0 + t] : ONE; }
                sp[6] = (fl & F_COLLIDER) ? ri[RI_COLLIDER] : ONE;
                // a visit carries its reaction's records inline: integer record + enhanced colliders, real record +
                // K_c polynomial rows + efficiencies, then the accumulate program of every row of the block it changes
                // (a visit that would not fit a ring slot is cut into several visits of the same reaction)
                size_t h0 = 0;
                while (h0 < hits.size() || h0 == 0) {
                    Rec v;
                    v.type = TAB_T_VISIT;
                    for (int f = 0; f < RIW; ++f) v.iw.push_back(ri[f]);
                    for (int e = 0; e < ri[RI_EFF_CNT]; ++e) v.iw.push_back(p.eff_sp[ri[RI_EFF_PTR] + e]);
                    for (int f = 0; f < RDW; ++f) v.dw.push_back(p.rd[(size_t)d * RDW + f]);
                    if (fl & F_REV)
                        for (int c = 0; c < ri[RI_KC_CNT] * KCW; ++c) v.dw.push_back(p.kcg[(size_t)ri[RI_KC_PTR] * KCW + c]);
                    for (int e = 0; e < ri[RI_EFF_CNT]; ++e) v.dw.push_back(p.eff_am1[ri[RI_EFF_PTR] + e]);
                    if (words_of(v) > TAB_RSZ) return fail("k_tab: a reaction record exceeds the ring slot (too many K_c groups / colliders)");
                    int nh = 0;
                    for (; h0 < hits.size(); ++h0) {
                        const int r = hits[h0].first;
                        const double nu = hits[h0].second;
                        int sl[TAB_NSLOT];
                        double mult[TAB_NSLOT];
                        for (int t = 0; t < TAB_NSLOT; ++t) {
                            sl[t] = out.TRASH; mult[t] = 0.0;
                            const int j = sp[t];
                            if (j == ONE || j >= last || j < 0) continue;
                            // the first position of a species within its side takes the multiplicity
                            const int t0 = t < 3 ? 0 : t < 6 ? 3 : 6, t1 = t < 3 ? 3 : t < 6 ? 6 : 7;
                            bool first = true; int cnt = 0;
                            for (int u = t0; u < t1; ++u) if (sp[u] == j) { if (u < t) first = false; ++cnt; }
                            if (!first || slot_of[r][j] < 0) continue;
                            sl[t] = slot_of[r][j]; mult[t] = cnt;
                        }
                        // enhanced colliders: (alpha - 1) b_M into the collider's column (the last species goes to gN)
                        std::vector<std::pair<int, double>> effl;
                        if (fl & F_EFFTYPE)
                            for (int e = 0; e < ri[RI_EFF_CNT]; ++e) {
                                const int es = p.eff_sp[ri[RI_EFF_PTR] + e];
                                if (es != last && slot_of[r][es] >= 0) effl.push_back({slot_of[r][es], nu * p.eff_am1[ri[RI_EFF_PTR] + e]});
                            }
                        // general stoichiometry: one value per factor
                        std::vector<std::pair<int, int>> genl;
                        if (fl & F_GEN) {
                            const int nf = ri[RI_GEN_NR] + ((fl & F_REV) ? ri[RI_GEN_NP] : 0);
                            for (int f = 0; f < nf; ++f) {
                                const int j = p.gen_sp[ri[RI_GEN_PTR] + f];
                                if (j < last && slot_of[r][j] >= 0) genl.push_back({slot_of[r][j], f});
                            }
                        }
                        Rec t = v;          // tentatively with this hit
                        t.iw.push_back(base[r]);
                        for (int q = 0; q < TAB_NSLOT; ++q) t.iw.push_back(sl[q]);
                        t.iw.push_back((int32_t)effl.size());
                        t.iw.push_back((int32_t)genl.size());
                        t.iw.push_back((bl.parts[r].k == last && d == p.lastq_rxn) ? 1 : 0);
                        for (auto& e : effl) t.iw.push_back(e.first);
                        for (auto& q : genl) { t.iw.push_back(q.first); t.iw.push_back(q.second); }
                        t.dw.push_back(nu);
                        for (int q = 0; q < TAB_NSLOT; ++q) t.dw.push_back(nu * mult[q]);
                        for (auto& e : effl) t.dw.push_back(e.second);
                        for (size_t q = 0; q < genl.size(); ++q) t.dw.push_back(nu);
                        if (words_of(t) > TAB_RSZ) {
                            if (nh == 0) return fail("k_tab: one row's share of a visit exceeds the ring slot");
                            break;
                        }
                        v = std::move(t);
                        ++nh;
                    }
                    v.a1 = nh;
                    put_rec(recs, std::move(v));
                    ++out.nvisit;
                    if (hits.empty()) break;
                }
            }
            // output records: a row (or column part of a row), at most TAB_EMAX entries each
            for (size_t r = 0; r < bl.parts.size(); ++r) {
                const Part& pt = bl.parts[r];
                std::vector<std::pair<int32_t, double>> en;
                for (int j = 0; j < last; ++j) {
                    const bool mine = slot_of[r][j] >= 0;
                    if (!mine && !(pt.first && !nz(pt.k, j))) continue;     // other part's column
                    // the entry's column constant next to it: 1 / W_j (W_j / W_N for the last species' pseudo-row)
                    en.push_back({(int32_t)j | ((int32_t)(mine ? slot_of[r][j] : out.ZERO) << 16),
                                  pt.k == last ? p.sp[(size_t)j * SPW + 3] : p.sp[(size_t)j * SPW]});
                }
                bool first = pt.first;
                for (size_t e0 = 0; e0 < en.size() || e0 == 0; e0 += TAB_EMAX) {
                    Rec o;
                    o.type = TAB_T_ROW;
                    o.flags = first ? 1 : 0;
                    o.a1 = pt.k;
                    const size_t e1 = std::min(en.size(), e0 + TAB_EMAX);
                    size_t cnt = e1 > e0 ? e1 - e0 : 0;
                    // padded to whole batches of TAB_EB entries by repeating the last one (the same value to the same
                    // address once more), plus one batch of look-ahead
                    const size_t padded = (cnt + TAB_EB - 1) / TAB_EB * TAB_EB;
                    o.iw.push_back(base[r]);
                    o.iw.push_back((int32_t)padded);
                    for (size_t e = 0; e < padded + TAB_EB; ++e) {
                        const auto& x = cnt ? en[e0 + std::min(e, cnt - 1)] : std::pair<int32_t, double>{out.ZERO << 16, 0.0};
                        o.iw.push_back(x.first);
                        o.dw.push_back(x.second);
                    }
                    if (!cnt) { o.iw[1] = 0; }
                    if (!put_rec(recs, std::move(o))) return fail("k_tab: internal error, output record exceeds the ring slot");
                    first = false;
                    if (en.empty()) break;
                }
            }
        }
        // serialise the group's records; every header names the sizes of the next two records
        grp.push_back((int32_t)S.size());
        grp.push_back((int32_t)recs.size());
        grp.push_back(recs.size() > 0 ? words_of(recs[0]) : 0);
        grp.push_back(recs.size() > 1 ? words_of(recs[1]) : 0);
        for (size_t q = 0; q < recs.size(); ++q) {
            const Rec& r = recs[q];
            const int niw = (int)((r.iw.size() + 1) / 2);
            auto pair = [&](int32_t lo, int32_t hi) { double w; int32_t v[2] = {lo, hi}; memcpy(&w, v, 8); S.push_back(w); };
            pair(r.type | (r.flags << 8), r.a1);
            pair(q + 1 < recs.size() ? words_of(recs[q + 1]) : 0, q + 2 < recs.size() ? words_of(recs[q + 2]) : 0);
            pair(niw, q + 3 < recs.size() ? words_of(recs[q + 3]) : 0);
            for (int w = 0; w < niw; ++w) pair(r.iw[2 * w], 2 * w + 1 < (int)r.iw.size() ? r.iw[2 * w + 1] : 0);
            for (double x : r.dw) S.push_back(x);
        }
        if (S.size() >= (1u << 30)) return fail("k_tab: program too large");
    }
    for (int q = 0; q < TAB_RSZ; ++q) S.push_back(0.0);          // the last fetches read a whole slot
    out.I = grp;
    out.ok = true;
    return true;
}

}  // namespace pj
