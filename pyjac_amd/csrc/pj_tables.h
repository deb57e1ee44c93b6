// pj_tables.h -- device-side mechanism "programs" and their host builder.
//
// The canonical mechanism blob (pyjac_amd/tables.py) is what crosses the C-ABI.
// build_programs() turns it into the flat arrays the HIP kernels walk:
// reaction records sorted by kind, per-species gather lists and per-Jacobian-
// entry gather lists.  Everything here is plain C++ (no HIP types) so the same
// code serves the device path and the thread-emulation test harness.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pj {

// canonical blob constants (pyjac_amd/tables.py)
constexpr int HDR = 96;
constexpr int32_t MAGIC = 0x314D4A50;
constexpr int32_t BLOB_VERSION = 2;   // 2: 16 integer / 22 real arrays (Chebyshev, per-emitter SRI rows)
enum { F_REV = 1, F_THD = 2, F_PDEP = 4, F_LOW = 8, F_HIGH = 16, F_TROE = 32,
       F_SRI = 64, F_PLOG = 128, F_TROE4 = 256, F_SRI5 = 512, F_HAS_EFF = 1024,
       // derived on the host:
       F_NO_DT = 2048,     // reference emits no d/dT line (create_jacobian.py:1512-1529)
       F_EFFTYPE = 4096,   // [M] carries enhanced efficiencies -> b_i acts on every column
       F_COLLIDER = 8192,  // falloff with a specific collider species
       F_LASTQ = 16384,    // the one reaction whose d/dT survives in J_nplusone (quirk)
       F_CHEB = 32768,     // blob flag again: Chebyshev rate expression (rate_subs.py:149-251)
       // derived on the host: general stoichiometry -- a fractional coefficient (mech_interpret.py:300-318,
       // 398-416; rate form pow(C, nu), rate_subs.py:634-658) or more than three molecules on a side: the
       // reaction carries (species, nu) factor lists instead of molecule slots
       F_GEN = 65536 };
enum { IA_FLAGS, IA_REAC_PTR, IA_REAC_SP, IA_PROD_PTR, IA_PROD_SP, IA_NET_PTR,
       IA_NET_SP, IA_EFF_PTR, IA_EFF_SP, IA_PLOG_PTR, IA_KC_PTR, IA_PDEP_SP,
       IA_REV_IDX, IA_PRES_IDX, IA_SEEN, IA_CHEB_PTR, IA_COUNT };
enum { DA_MW, DA_TMID, DA_LO, DA_HI, DA_A, DA_B, DA_E, DA_REAC_NU, DA_PROD_NU,
       DA_NET_NU, DA_EFF, DA_PD, DA_TROE, DA_SRI, DA_PLOG, DA_KCG, DA_KCPREF,
       DA_INFS, DA_TROE8, DA_PLOG4, DA_SRIQ, DA_CHEB, DA_COUNT };

// ---- record widths ----
constexpr int SPW = 18;   // species: invW, W, tmid, w=W/W_N, lo[7], hi[7]
constexpr int RIW = 23;   // reaction ints
constexpr int RDW = 16;   // reaction doubles
enum { RI_FLAGS, RI_R0, RI_R1, RI_R2, RI_P0, RI_P1, RI_P2, RI_EFF_PTR, RI_EFF_CNT,
       RI_COLLIDER, RI_KC_PTR, RI_KC_CNT, RI_PLOG_PTR, RI_PLOG_CNT, RI_GBASE,
       RI_NET_PTR, RI_NET_CNT, RI_ORIG, RI_REV_IDX, RI_PRES_IDX,
       RI_GEN_PTR, RI_GEN_NR, RI_GEN_NP };   // F_GEN: factors gen_sp/gen_nu[GEN_PTR ..], NR reactant then NP product factors
enum { RD_LNA, RD_B, RD_TA, RD_SGN, RD_NR, RD_NP, RD_LNPREF, RD_LNAR, RD_B0, RD_E0,
       RD_B04, RD_TRA, RD_T3, RD_T1, RD_T2, RD_ANM1 };
constexpr int EFF_INL = 8;  // enhanced colliders held inline in the field-major tables
constexpr int PLW = 5;    // plog row: P ('%.4e'), lnP, lnA, b, Ta
constexpr int KCW = 15;   // kc group: tmid, lo[7], hi[7]
// SRI falloff (F_SRI): RI_PLOG_PTR = row of the SRI table.  The reference prints the parameters
// a, b, c, d, e with a different number of digits in each emitter; the table carries each variant:
//   '{:.6}' (d: '{:.8e}')  get_rxn_pres_mod and the F_i factor    rate_subs.py:1229-1256, create_jacobian.py:249-266
//   '{:.4}'                the dPr/dY_j term                       create_jacobian.py:173-179
//   '{:.16}'               the d/dT term                           create_jacobian.py:1194-1237
constexpr int SRW = 16;
enum { SR_A6, SR_B6, SR_C6, SR_D8, SR_E6, SR_USE_DE, SR_A4, SR_B4, SR_C4, SR_A16, SR_B16, SR_C16, SR_AB16,
       SR_INVC16, SR_E16 };
// Chebyshev (F_CHEB): RI_PLOG_PTR = offset of the reaction's record in the Chebyshev array, RI_PLOG_CNT =
// n_temp * 256 + n_pres.  Record: n_temp, n_pres, the reduced-variable constants as the rate emitter
// prints them ('{:.8e}': 1/Tmin+1/Tmax, 1/Tmax-1/Tmin, lg Pmin+lg Pmax, lg Pmax-lg Pmin) and as the
// Jacobian emitter does ('{:.16e}'), -2 ln10 / (1/Tmax - 1/Tmin), the n x m coefficients ('{:.8e}'),
// then (n-1) x m coefficients i * c_ij ('{:.16e}') of the d/dT sum.
enum { CH_N, CH_M, CH_TSUM8, CH_TSUB8, CH_PSUM8, CH_PSUB8, CH_TSUM16, CH_TSUB16, CH_PSUM16, CH_PSUB16, CH_DFAC, CH_COEF };
constexpr int CHEB_MAXT = 12, CHEB_MAXP = 12;

// V-array slot map (per-state working set, doubles)
struct VMap {
    int nsp = 0, nrxn = 0, ng = 0;
    int C = 0, ONE = 0, HW = 0, CP = 0, YC = 0, YD = 0;
    int RQ = 0, RTH = 0, RP = 0, RQQ = 0;                    // [nrxn] each
    int G = 0;                                               // [ng]
    int NV = 0;                                              // slots written by phases 0-2
    // accumulation tile (targets of the scatter phase), slot = TB + target:
    //   T_OM + k, T_JT + k, T_P + k, T_Q + k   dense vectors (omega_k, sum nu theta, P_k, Q_k)
    //   T_S + smap[k + nsp*j]                  sparse block S_kj (k < nsp incl. last species, j < nsp-1):
    //                                          only structurally non-zero entries own a slot
    //   T_JTQ                                  d/dT of the last species from the one reaction the
    //                                          reference keeps (create_jacobian.py:2786-2818)
    //   T_PART ...                             partial accumulators of split hub targets
    int TB = 0, T_OM = 0, T_JT = 0, T_P = 0, T_Q = 0, T_S = 0, T_JTQ = 0, T_PART = 0;
    int NTILE = 0;                                           // incl. partials (set by build_schedule)
    int SC = 0;                                              // 5 per-state scalars after the tile
    int X = 0;                                               // [5][nsp] products feeding those scalars
    int NSLOT = 0;                                           // total slots per state
};
enum { SC_H, SC_HP, SC_HQ, SC_SCP, SC_SJT, SC_COUNT };

// one term of the scatter phase: tile[tgt] += nu * V[src]
struct Contrib { int src, tgt; double nu; bool dense; };
// nu of a scatter term as a 4-bit code: whole numbers -4..3 are codes 0..7 (value = code - 4), anything else
// (fractional or larger net coefficients) is one of up to NUTAB_N table entries, codes 8..15
constexpr int NUTAB_N = 8;

// Conflict-free scatter schedule for NW wavefronts x IL item lanes: every tile
// target is owned by one wavefront; within a round the IL lanes of a wavefront
// update distinct targets, so plain LDS read-modify-writes need no atomics and
// no barrier.  Dense-vector terms come first (rates-only launches stop there).
struct Schedule {
    int NW = 0, IL = 0;
    std::vector<uint32_t> codes;       // [wave][round][il]: src | tgt << 13 | nu code << 28
    int off[16] = {0}, rounds[16] = {0}, rounds_dense[16] = {0};
    // split targets: final slot, first partial slot, number of partials (consecutive slots)
    std::vector<int32_t> fin_tgt, fin_part, fin_cnt;
};

struct Programs {
    int nsp = 0, nrxn = 0, nrev = 0, npres = 0, ng = 0, ne = 0;
    VMap vm;
    std::vector<double> sp;        // [nsp*SPW]
    std::vector<int32_t> ri;       // [nrxn*RIW]  (device order)
    std::vector<double> rd;        // [nrxn*RDW]
    std::vector<int32_t> eff_sp;   // enhanced colliders (alpha != 1 only)
    std::vector<double> eff_am1;   // alpha - 1
    std::vector<double> kcg;       // [n*KCW]
    std::vector<double> plog;      // [n*PLW]
    std::vector<double> sri;       // [n*SRW]
    std::vector<double> cheb;      // Chebyshev records, back to back
    std::vector<int32_t> net_sp;   // per-reaction net list
    std::vector<double> net_nu;
    std::vector<int32_t> gen_sp;   // F_GEN reactions: (species, nu) factors, reactants then products
    std::vector<double> gen_nu;
    double nutab[NUTAB_N] = {0, 0, 0, 0, 0, 0, 0, 0};   // net coefficients outside -4..3 / fractional (scatter codes 8..)
    int n_nutab = 0;
    // P3: per species gather over reactions (global copy: k_spec_rates)
    std::vector<int32_t> sp_ptr, sp_rxn;
    std::vector<double> sp_nu;
    int lastq_rxn = -1;            // device index of the F_LASTQ reaction
    std::vector<Contrib> contribs;
    // field-major ("SoA") copies of the reaction records for the table-driven kernel:
    // consecutive item lanes hold consecutive reactions, so rti[f*nrp + i] / rtd[f*nrp + i]
    // are coalesced.  rtd also carries K_c group 0 (KCW fields) and up to EFF_INL
    // enhanced colliders per reaction inline (rti: species or ONE, rtd: alpha - 1).
    // sparse-block slot map: smap[k + nsp*j] = slot relative to T_S or 0xFFFF (structural zero);
    // per column j the non-zero rows as (k << 16 | slot), CSR by column (energy-row sums)
    std::vector<uint16_t> smap;
    std::vector<int32_t> ecol_ptr;
    std::vector<uint32_t> ecol;
    int nnz = 0;
    // Equilibrium constants from per-species factors (pj_rblk.hip, PJQ_KCF): [nsp][KCW] rows (T_mid, lo[7], hi[7]) of
    // the SHIFTED ln X_k in the K_c polynomial form, 1 / K_c,i = (p_atm / R_u)^(-sum nu) prod_k X_k^(-nu_ki); computed by
    // the host front end from the stoichiometry (pyjac_amd/kcfactors.py, set through pj_mech_set_kc_factors); empty:
    // the mechanism keeps the per-reaction polynomial form.  Not part of programs_hash (derived data).
    std::vector<double> kcf;
    int nrp = 0;                   // nrxn padded to a multiple of 64
    std::vector<int32_t> rti;      // [(RIW + EFF_INL) * nrp]
    std::vector<double> rtd;       // [(RDW + KCW + EFF_INL) * nrp]
    std::string error;
};

// (re)builds the schedule and fixes vm.NTILE / vm.SC / vm.NSLOT for this geometry
bool build_schedule(Programs& p, int NW, int IL, Schedule& out);

// 64-bit FNV-1a of the device programs: identifies a mechanism for the
// register-resident specialisation (pj_lane.hip).
uint64_t programs_hash(const Programs& p);
// Mechanism constants as a C++ header (constexpr arrays) for pj_lane.hip.
std::string emit_spec_header(const Programs& p);
// Row-block partition + hand-over numbering for pj_rblk.hip, appended to that header.
// plan_opts: also append the kernel plan of a pj_rblk.hip library (NKER, KER_B, KER_BM, NRATE, RATE_R): which
// row blocks each row kernel takes (and where its two lane groups meet), which reactions each rate kernel takes.
struct RblkPlanOpts {
    int fuse = 13;              // row blocks per kernel and lane group at most
    int block = 256;            // states per workgroup of the row kernels
    int halves = 1;             // lane groups per workgroup (2 / 4: on the same states, different row blocks)
    int single = 0;             // 1: ONE row kernel takes every row block (its lane groups split them)
    int rate_block = 256;       // states per workgroup of the rate kernels
    int rate_c_lds = 0;         // rate kernels keep the concentrations in LDS columns
    int rate_groups = 0;        // K_c groups per rate kernel at most (0: whatever fits the LDS)
    double cost_visit = 0.24, cost_entry = 0.06;    // halves balance: time of a visit / of a Jacobian entry (us)
};
struct RblkPlan { int n_row_kernels = 0, n_rate_kernels = 0, n_pre = 0, n_blocks = 0, n_visits = 0; };
std::string emit_rows_tables(const Programs& p, int budget, const RblkPlanOpts* plan_opts = nullptr, RblkPlan* plan_out = nullptr);

// Returns false (and sets p.error) when the blob is malformed or uses a
// feature outside the hot-path scope.
bool build_programs(const int32_t* I, long nI, const double* D, long nD, Programs& p);

}  // namespace pj
