// pj_tabprog.cpp -- host builder of the k_tab program (pj_tabprog.h).
#include "pj_tabprog.h"

#include <algorithm>
#include <numeric>

namespace pj {

namespace {
constexpr int DENSE = TAB_DENSE;      // omega_k, P_k, Q_k, sum nu theta, and the J_nplusone quirk value per row
struct Part { int k; std::vector<int> cols; bool first; };           // a row or a column part of a row
struct Block { std::vector<Part> parts; std::vector<int> rx; long cost = 0; };
}  // namespace

bool build_tab_program(const Programs& p, size_t lds_avail, TabProg& out)
{
    out = TabProg();
    const int nsp = p.nsp, nrxn = p.nrxn, last = nsp - 1, ONE = nsp;
    auto fail = [&](const char* w) { out.error = w; return false; };
    if (nsp < 2) return fail("k_tab: at least two species");
    // ---- geometry: L states per workgroup, B accumulator slots per lane group ----
    int L = 0, B = 0;
    for (int cand : {256, 128, 64}) {
        const long room = (long)lds_avail - (long)nsp * cand * 8;
        const int b = room > 0 ? (int)(room / (256L * 8)) : 0;
        if (b >= 46 || cand == 64) { L = cand; B = b; break; }
    }
    if (B > 96) B = 96;
    if (B < DENSE + 3 + 2) return fail("k_tab: the concentration columns of this mechanism leave no room for accumulators in LDS");
    const int cap = B - 2;                       // slots a block may use (ZERO and TRASH are the last two)
    out.L = L; out.G = 256 / L; out.B = B; out.ZERO = B - 2; out.TRASH = B - 1;
    out.lds_bytes = (size_t)nsp * L * 8 + (size_t)256 * B * 8;

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

    // ---- emit ----
    std::vector<int32_t> grp_ptr{0}, grp_blk, blk, row, ent, vi;
    std::vector<double>& vd = out.D;
    for (int g = 0; g < out.G; ++g) {
        for (int b : gb[g]) grp_blk.push_back(b);
        grp_ptr.push_back((int32_t)grp_blk.size());
    }
    for (int b = 0; b < nblk; ++b) {
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
        blk.push_back((int32_t)vi.size());
        blk.push_back((int32_t)bl.rx.size());
        blk.push_back((int32_t)(row.size() / TAB_ROW));
        blk.push_back((int32_t)bl.parts.size());
        blk.push_back((int32_t)vd.size());
        blk.push_back((int32_t)next);
        for (size_t r = 0; r < bl.parts.size(); ++r) {
            const Part& pt = bl.parts[r];
            row.push_back(pt.k);
            row.push_back(base[r]);
            row.push_back(pt.first ? 1 : 0);
            row.push_back((int32_t)ent.size());
            int cnt = 0;
            for (int j = 0; j < last; ++j) {
                const bool mine = slot_of[r][j] >= 0;
                if (!mine && !(pt.first && !nz(pt.k, j))) continue;     // other part's column
                ent.push_back((int32_t)j | ((int32_t)(mine ? slot_of[r][j] : out.ZERO) << 16));
                // the entry's column constant next to it: 1 / W_j (W_j / W_N for the last species' pseudo-row)
                out.E.push_back(pt.k == last ? p.sp[(size_t)j * SPW + 3] : p.sp[(size_t)j * SPW]);
                ++cnt;
            }
            // padded to whole batches of TAB_EB entries by repeating the last one (the same value to the same
            // address once more): the output loop has no tail and reads one batch ahead
            while (cnt % TAB_EB) { ent.push_back(ent.back()); out.E.push_back(out.E.back()); ++cnt; }
            row.push_back(cnt);
            row.push_back(0);
        }
        // visits
        for (int d : bl.rx) {
            const int32_t* ri = &p.ri[(size_t)d * RIW];
            const int fl = ri[RI_FLAGS];
            std::vector<std::pair<int, double>> hits;      // (part index, nu_k)
            for (size_t r = 0; r < bl.parts.size(); ++r)
                for (int q = 0; q < ri[RI_NET_CNT]; ++q)
                    if (p.net_sp[ri[RI_NET_PTR] + q] == bl.parts[r].k) hits.push_back({(int)r, p.net_nu[ri[RI_NET_PTR] + q]});
            // a visit carries its reaction's records inline (integer record + enhanced colliders; real record + K_c
            // polynomial rows + efficiencies): everything a visit reads sits at fixed offsets from the two stream
            // pointers, i.e. one round of scalar loads instead of a chain of dependent ones
            const size_t vi0 = vi.size(), vd0 = vd.size();
            vi.push_back(d);
            vi.push_back((int32_t)hits.size());
            vi.push_back(0);        // ints of this visit (filled in below): the next visit's records are
            vi.push_back(0);        // doubles of this visit            requested while this one is computed
            for (int f = 0; f < RIW; ++f) vi.push_back(ri[f]);
            for (int e = 0; e < ri[RI_EFF_CNT]; ++e) vi.push_back(p.eff_sp[ri[RI_EFF_PTR] + e]);
            for (int f = 0; f < RDW; ++f) vd.push_back(p.rd[(size_t)d * RDW + f]);
            if (fl & F_REV)
                for (int c = 0; c < ri[RI_KC_CNT] * KCW; ++c) vd.push_back(p.kcg[(size_t)ri[RI_KC_PTR] * KCW + c]);
            for (int e = 0; e < ri[RI_EFF_CNT]; ++e) vd.push_back(p.eff_am1[ri[RI_EFF_PTR] + e]);
            int sp[TAB_NSLOT];
            for (int t = 0; t < 3; ++t) { sp[t] = ri[RI_R0 + t]; sp[3 + t] = (fl & F_REV) ? ri[RI_P0 + t] : ONE; }
            sp[6] = (fl & F_COLLIDER) ? ri[RI_COLLIDER] : ONE;
            for (auto& h : hits) {
                const int r = h.first;
                const double nu = h.second;
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
                vi.push_back(base[r]);
                for (int t = 0; t < TAB_NSLOT; ++t) vi.push_back(sl[t]);
                vi.push_back((int32_t)effl.size());
                vi.push_back((int32_t)genl.size());
                vi.push_back((bl.parts[r].k == last && d == p.lastq_rxn) ? 1 : 0);
                vd.push_back(nu);
                for (int t = 0; t < TAB_NSLOT; ++t) vd.push_back(nu * mult[t]);
                for (auto& e : effl) { vi.push_back(e.first); vd.push_back(e.second); }
                for (auto& g : genl) { vi.push_back(g.first); vi.push_back(g.second); vd.push_back(nu); }
            }
            vi[vi0 + 2] = (int32_t)(vi.size() - vi0);
            vi[vi0 + 3] = (int32_t)(vd.size() - vd0);
            ++out.nvisit;
        }
    }
    for (int q = 0; q < TAB_EB; ++q) { ent.push_back(out.ZERO << 16); out.E.push_back(0.0); }      // look-ahead of the last batch
    if (ent.size() >= (1u << 30) || vi.size() >= (1u << 30)) return fail("k_tab: program too large");
    auto put = [&](const std::vector<int32_t>& v) { const int o = (int)out.I.size(); out.I.insert(out.I.end(), v.begin(), v.end()); return o; };
    out.o_grp_ptr = put(grp_ptr); out.o_grp_blk = put(grp_blk); out.o_blk = put(blk); out.o_row = put(row);
    out.o_ent = put(ent); out.o_vi = put(vi);
    out.ok = true;
    return true;
}

}  // namespace pj
