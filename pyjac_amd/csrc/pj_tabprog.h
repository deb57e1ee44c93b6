// pj_tabprog.h -- the run-time "program" of the table-driven state-per-lane Jacobian kernel k_tab (pj_tab.h).
//
// k_tab is the path a user WITHOUT a compiler gets: no header emission, no hipcc -- the mechanism tables
// (pj_tables.h: Programs) are turned into a block / visit program on the host at mechanism-load time, and ONE
// precompiled kernel walks it with wavefront-uniform control flow (every lane of a wavefront is a different
// state working on the same reaction; reaction records and program words are scalar loads).
//
// It is the row-block formulation of pj_rblk.hip with run-time tables:
//   * a workgroup of 256 threads owns L states (L = 256 / G): G groups of L lanes on the SAME states share one
//     set of concentration columns in LDS and run different row blocks;
//   * a row block = the accumulators of one or more species rows (omega_k, P_k, Q_k, sum nu theta and the
//     structurally non-zero S_kj) as LDS columns ACC[slot][lane] of the group; a row whose non-zero pattern
//     exceeds the budget is split into column parts (each part re-visits the row's reactions);
//   * a visit of reaction i rebuilds its rate (every rate form, in line: there is no pre-pass) and applies the
//     block's accumulate program: per row of the block that the reaction changes, 4 dense + 7 molecule-slot
//     read-modify-writes at slots the program names (TRASH for "not here"), then efficiency / general-
//     stoichiometry lists;
//   * the output phase of a block writes its rows' Jacobian entries; the energy row is finished by a second
//     kernel (k_tab_fin) from the stored species rows: sum_k hW_k (P_k - w_j Q_k + S_kj) is a weighted column
//     sum of the finished block, so no per-column accumulators are needed at all.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "pj_tables.h"

namespace pj {

constexpr int TAB_NSLOT = 7;          // molecule-slot positions of a reaction: R0 R1 R2 P0 P1 P2 collider
constexpr int TAB_HIT_I = 1 + TAB_NSLOT + 3;   // ints per hit: dense base, 7 slots, n_eff, n_gen, flags (1: J_nplusone quirk value)
constexpr int TAB_HIT_D = 1 + TAB_NSLOT;       // doubles per hit: nu_k, nu_k * multiplicity per slot
constexpr int TAB_BLK = 6, TAB_ROW = 6, TAB_DENSE = 5;
constexpr int TAB_EB = 8;             // output entries per batch (a row's entry list is padded to whole batches)

struct TabProg {
    bool ok = false;
    int L = 0, G = 0;                 // states per workgroup, lane groups per workgroup (L * G = 256)
    int B = 0;                        // accumulator slots per group (incl. the two special ones below)
    int ZERO = 0, TRASH = 0;          // slot that always reads 0 / slot that takes updates nobody wants
    int nblk = 0, nvisit = 0;
    size_t lds_bytes = 0;
    // integer program, one array with offsets (scalar loads):
    //   grp_ptr[G + 1]                 blocks of group g: grp_blk[grp_ptr[g] .. grp_ptr[g + 1])
    //   blk[b * TAB_BLK ..]            visit offset (into vi), visit count, first row, row count, offset into D, slots used
    //   row[r * 6 ..]                  species k, dense base slot, flags (1: writes omega_k / d/dT column),
    //                                  entry offset (into ent), entry count, reserved
    //   ent[e]                         column j | slot << 16   (slot == ZERO: structurally zero)
    //   vi[...]                        visits: rxn, nhit, ints and doubles of this visit, the reaction's integer record (RIW) and enhanced-collider
    //                                  species inline, then per hit TAB_HIT_I ints + n_eff slots + n_gen (slot, factor)
    // doubles vd[...]: per visit the reaction's real record (RDW), its K_c rows (KCW each) and efficiencies - 1
    //                  inline, then per hit TAB_HIT_D doubles + n_eff coefficients + n_gen nu_k
    std::vector<int32_t> I;
    std::vector<double> D;
    std::vector<double> E;             // per output entry: 1 / W_j (W_j / W_N in the last species' pseudo-row)
    int o_grp_ptr = 0, o_grp_blk = 0, o_blk = 0, o_row = 0, o_ent = 0, o_vi = 0;
    std::string error;
};

// Chooses L (the largest of 256 / 128 / 64 that leaves at least 44 accumulator slots next to the concentration
// columns in `lds_avail` bytes), partitions the rows, assigns blocks to groups (longest processing time first)
// and writes the program.  false + error: the mechanism does not fit (a row needs more than the budget even
// when split, which cannot happen, or more than 32767 slots / entries).
bool build_tab_program(const Programs& p, size_t lds_avail, TabProg& out);

}  // namespace pj
