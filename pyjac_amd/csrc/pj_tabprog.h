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
constexpr int TAB_DENSE = 5;          // omega_k, P_k, Q_k, sum nu theta, J_nplusone quirk value
constexpr int TAB_EB = 8;             // output entries per batch (a record's entry list is padded to whole batches)
constexpr int TAB_RSZ = 192;          // 8-byte words of a ring slot = largest record
constexpr int TAB_EMAX = 112;         // output entries per record at most
constexpr int TAB_RING_WORDS = 3 * TAB_RSZ;   // per wavefront: the record in use and the next two
enum { TAB_T_VISIT = 0, TAB_T_ROW = 1, TAB_T_BEGIN = 2 };

struct TabProg {
    bool ok = false;
    int L = 0, G = 0;                 // states per workgroup, lane groups per workgroup (L * G = 256)
    int B = 0;                        // accumulator slots per group (incl. the two special ones below)
    int ZERO = 0, TRASH = 0;          // slot that always reads 0 / slot that takes updates nobody wants
    int nblk = 0, nvisit = 0;
    size_t lds_bytes = 0;
    // D: the record streams of the lane groups, 8-byte words.  A record:
    //   word 0   int32 pair { type | flags << 8, a1 }     VISIT: a1 = rows of the block this visit changes ("hits");
    //                                                     BEGIN: a1 = accumulator slots to clear; ROW: a1 = species k,
    //                                                     flags & 1: this part also writes omega_k / the d/dT column
    //   word 1   int32 pair { words of the next record, words of the one after }
    //   word 2   int32 pair { integer words that follow, words of the third record from here }
    //   then the integer words (int32 pairs), then doubles:
    //   VISIT    ints: the reaction's integer record (RIW), its enhanced-collider species, then per hit TAB_HIT_I
    //            ints + n_eff slots + n_gen (slot, factor);  doubles: the real record (RDW), its K_c rows (KCW each),
    //            efficiencies - 1, then per hit TAB_HIT_D doubles + n_eff coefficients + n_gen nu_k
    //   ROW      ints: dense base slot, entries (padded to batches of TAB_EB, plus one batch), then
    //            column j | slot << 16 each (slot == ZERO: structurally zero);  doubles: 1 / W_j per entry
    //            (W_j / W_N in the last species' pseudo-row)
    // I: per group { first word, records, words of the first record, words of the second }
    std::vector<int32_t> I;
    std::vector<double> D;
    std::string error;
};

// Chooses L (the largest of 256 / 128 / 64 that leaves at least 56 accumulator slots next to the concentration
// columns and the wavefronts' rings in `lds_avail` bytes; force_L: that value), partitions the rows, assigns blocks
// to groups (longest processing time first) and writes the record streams.  false + error: the mechanism does not
// fit (a reaction record larger than a ring slot, no room for accumulators).
bool build_tab_program(const Programs& p, size_t lds_avail, TabProg& out, int force_L = 0);

}  // namespace pj
