/*
 * pyjac_amd.h -- C ABI of libpyjac_hip.so: MI355X-native batched species-rate
 * and analytical-Jacobian evaluation behind pyJac's pywrap boundary.
 *
 * What this replaces (paths relative to the reference tree):
 *   - the per-mechanism generated C evaluator that pyjac/pywrap/pyjacob_wrapper.pyx:4-16
 *     binds (dydt, eval_jacob, eval_rxn_rates, eval_spec_rates, get_rxn_pres_mod,
 *     eval_conc)                                   -> pj_dydt ... pj_eval_conc
 *   - the CUDA batch driver pyjac/pywrap/pyjacob.cuh:6-10 / pyjacob.cu:84-188
 *     (init / run / cleanup)                       -> pj_init / pj_run / pj_cleanup
 *   - the CUDA speed-test inner loop pyjac/performance_tester/tester.cu.in:109-156
 *     (device-resident evaluation)                 -> pj_eval_*_dev, pj_time_jacobian_dev
 * pyJac compiles one library per mechanism (NSP etc. are macros); here the
 * mechanism is data, so every entry point takes a mechanism handle first.
 * Everything else (argument order, units, layouts, side effects) is pyJac's.
 *
 * Conventions
 *   - fp64 everywhere, SI units (Pa, K, kmol, m^3, s, J) as docs/faqs.rst:92-103.
 *   - State vector y = [T, Y_0 .. Y_{NSP-2}] (last species eliminated).
 *   - Jacobian block is NSP x NSP, column-major per state: entry (r, c) at r + NSP*c
 *     (create_jacobian.py:2880, docs/faqs.rst:82-87).  Every entry is written; the
 *     caller does NOT need to pre-zero (pyJac's callers must, tester.c.in:27).
 *   - PJ_LAYOUT_SOA: element (i, s) at base[i*n + s]  (pyJac's batch layout,
 *     pyjacob.cu:139-187, test.py:655-660).  PJ_LAYOUT_AOS: base[s*rows + i]
 *     (pyJac's per-state C layout repeated per state).
 *   - All functions return 0 on success, a negative PJ_E* code otherwise;
 *     pj_last_error() describes the last failure of the calling thread.
 *     (pyJac's own C/CUDA code has no error channel and exit()s:
 *      mech_auxiliary.py:425-436, pyjacob.cu:108-112.)
 *   - No CPU fallback: if no HIP device is usable every evaluation call fails
 *     with PJ_ENODEV.
 */
#ifndef PYJAC_AMD_H
#define PYJAC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pj_mech pj_mech;

enum { PJ_LAYOUT_SOA = 0, PJ_LAYOUT_AOS = 1 };
enum { PJ_OK = 0, PJ_EINVAL = -1, PJ_ENODEV = -2, PJ_EHIP = -3, PJ_ENOMEM = -4,
       PJ_EUNSUPPORTED = -5, PJ_EIO = -6 };

const char* pj_last_error(void);
const char* pj_version(void);

/* ---- mechanism (replaces compile-time mechanism.h: NSP, FWD_RATES, REV_RATES,
 *      PRES_MOD_RATES; mech_auxiliary.py:136-161) ---- */
/* I/D: the table blob of pyjac_amd/tables.py (int32 header+arrays, fp64 arrays). */
int pj_mech_create(const int32_t* I, long nI, const double* D, long nD, pj_mech** out);
/* file written by MechTables.save(): [u64 nI][u64 nD][I][D], little endian */
int pj_mech_load(const char* path, pj_mech** out);
void pj_mech_destroy(pj_mech* m);
int pj_mech_nsp(const pj_mech* m);
int pj_mech_fwd_rates(const pj_mech* m);
int pj_mech_rev_rates(const pj_mech* m);
int pj_mech_pres_mod_rates(const pj_mech* m);
/* 0 (default): keep the reference's J_nplusone assignment quirk
 * (create_jacobian.py:2786-2818) so jac[0] equals pyJac's; 1: sum all reactions. */
int pj_mech_set_sum_last_species(pj_mech* m, int on);
/* 1: pj_eval_jacobian_dev / pj_eval_jacobian_vec_dev / pj_eval_rates_dev first verify the preconditions the
 * reference leaves to its caller (T > 0 -- log T is taken --, p > 0, every input finite: docs/faqs.rst:92-103) and
 * return PJ_EINVAL naming the first offending state; costs one pass over the inputs and a stream
 * synchronisation.  0 (default): no check, as in the reference (undefined results for such states). */
int pj_mech_set_check_inputs(pj_mech* m, int on);
/* Everything the six per-state functions below return, for ONE state y = [T, Y_0 .. Y_{NSP-2}], in one upload, one
 * set of launches and one download (any output may be null).  The reference's functional tester calls the six
 * functions on the same state one after the other (functional_tester/test.py:1299-1327); pyjac_amd/pyjacob.py fills
 * a one-state cache through this entry point so that the sequence costs one evaluation. */
int pj_eval_state(pj_mech* m, double pres, const double* y, double* conc, double* fwd, double* rev, double* pres_mod,
                  double* spec_rates, double* dy, double* jac);
/* Which kernel evaluates Jacobians when no mechanism-specific library is attached (or pj_mech_use_spec(m, 0)):
 * 2 k_tab -- table-driven, one state per lane, row blocks with accumulators in LDS, wavefront-uniform control
 * flow over a program built at load time (csrc/pj_tab.h; the formulation of the compiled row-block kernels: every
 * entry within rtol 1e-6 of the exact value); 0 k_eval -- the cooperative kernel (a workgroup per tile of states),
 * which also serves every rate output; 1 (default) k_tab for SoA Jacobians, k_eval for AoS ones (each one's
 * faster layout).  No compiler is needed for either.  PJ_EUNSUPPORTED if k_tab cannot hold the mechanism. */
int pj_mech_set_generic_kernel(pj_mech* m, int which);
/* launch tuning: states per workgroup tile (power of two <= 64, 0 = auto),
 * threads per workgroup (multiple of 64, 0 = auto) */
int pj_mech_set_launch(pj_mech* m, int tile_states, int threads);
int pj_mech_get_launch(const pj_mech* m, int* tile_states, int* threads, int* lds_bytes);

/* ---- register-resident specialisation for small mechanisms (pj_lane.hip) ----
 * pyJac compiles every mechanism (python -m pyjac + libgen); here compilation is
 * optional: the table-driven kernel serves any mechanism, and a mechanism-
 * specific build of ONE hand-written kernel (constexpr tables, everything in
 * registers) can be attached for speed.  The two paths agree to rounding. */
unsigned long long pj_mech_spec_hash(const pj_mech* m);
/* write the constexpr header consumed by pj_lane.hip (-DPJS_HEADER='"path"') */
int pj_mech_emit_spec(const pj_mech* m, const char* header_path);
/* same header plus the row-block partition consumed by pj_rblk.hip: the state-per-lane
 * kernels for mechanisms whose sparse Jacobian block exceeds the register file; acc_budget =
 * accumulator doubles a row block may hold in registers */
int pj_mech_emit_rows_spec(const pj_mech* m, const char* header_path, int acc_budget);
/* ... plus the kernel plan of a pj_rblk.hip library (NKER / KER_B / KER_BM: row blocks [B0, B1) of each row kernel
 * and where its two lane groups meet; NRATE / RATE_R: reactions of each rate kernel).  fuse: row blocks per
 * kernel and lane group at most; block, halves: states per workgroup and lane groups (1 | 2 | 4, on the same states)
 * of the row kernels, KER_GB: first row block of each lane group; single != 0: ONE row kernel takes every row block
 * (the state is read once, no hand-over of the energy-row sums between kernels); rate_block, rate_c_lds: states per workgroup of the rate kernels and whether they keep the
 * concentrations in LDS; rate_groups: K_c groups per rate kernel at most (0: what fits the LDS); cost_visit /
 * cost_entry (<= 0: defaults): balance of the two lane groups.  counts (may be null): [0] row kernels,
 * [1] rate kernels, [2] reactions evaluated by the pre-pass, [3] row blocks, [4] reaction visits.
 * (pyJac's counterpart: the generation step, python -m pyjac; libgen/libgen.py:330-420 compiles its output) */
int pj_mech_emit_rblk_spec(const pj_mech* m, const char* header_path, int acc_budget, int fuse, int block, int halves,
                           int single, int rate_block, int rate_c_lds, int rate_groups, double cost_visit,
                           double cost_entry, int* counts);
/* Equilibrium constants from per-species factors for the pj_rblk.hip row kernels (PJQ_KCF): rows = [nsp][15] doubles
 * (T_mid, lo[7], hi[7]) of the shifted ln X_k in pyJac's K_c polynomial form (rate_subs.py:540-558, 660-809), with
 * 1 / K_c,i = (p_atm / R_u)^(-sum nu) prod_k X_k^(-nu_ki) -- one exp per species and state instead of one per
 * reaction visit; computed by the host front end from the stoichiometry (pyjac_amd/kcfactors.py).  n = 15 * nsp, or
 * 0 / rows = NULL to go back to the per-reaction polynomial form.  Only shapes the headers the emit functions write. */
int pj_mech_set_kc_factors(pj_mech* m, const double* rows, long n);
/* dlopen a library built from pj_lane.hip / pj_rblk.hip for this mechanism (hash-checked) and
 * route pj_eval_jacobian_dev / pj_run / pj_eval_jacob through it */
int pj_mech_attach_spec(pj_mech* m, const char* library_path);
/* 1 if a specialised kernel is attached */
int pj_mech_has_spec(const pj_mech* m);
/* 0: table-driven kernel even if a specialisation is attached; 1 (default): use it for SoA Jacobians
 * and, if it transposes through LDS (pj_lane.hip), for AoS ones (otherwise AoS output is the
 * cooperative kernel's native layout and goes there); 2: use it for every layout */
int pj_mech_use_spec(pj_mech* m, int on);
/* Launch settings of an attached pj_rblk.hip library, per handle (every handle owns its hand-over arrays, internal
 * streams and staging blocks: two handles of one mechanism run independently).  A negative value leaves a setting
 * unchanged.  streams: internal streams the chunks of a batch are dealt to (0: the build's default, 1: everything
 * on the caller's stream, at most 8); chunk_states: states per chunk (< 256: the build's default); split_tail:
 * batches whose last round of workgroups is partially filled run as two unequal parts on two streams (default 1);
 * aos_direct: AoS Jacobians by strided lane stores instead of SoA chunks + LDS-tiled transpose (default 0).  The
 * defaults come from the environment, read once when the library is attached (PJ_RBLK_STREAMS, PJ_RBLK_CHUNK,
 * PJ_RBLK_SPLIT, PJ_RBLK_AOS_DIRECT).  PJ_EINVAL without such a library.  (pyJac's counterpart: none -- its batch
 * path owns one set of module-level device buffers, pyjacob.cu:65-80) */
int pj_mech_set_spec_launch(pj_mech* m, int streams, long chunk_states, int split_tail, int aos_direct);

/* ---- device-resident batch evaluation (pointers are device pointers on the
 *      current HIP device; stream is a hipStream_t or NULL) ---- */
int pj_eval_jacobian_dev(pj_mech* m, long n, const double* d_pres, const double* d_y,
                         int y_layout, double* d_jac, int jac_layout, void* stream);
/* Fused consumer: w_s = J(Phi_s) v_s for every state, v and w shaped like y (NSP rows, layout
 * vw_layout).  What an implicit / exponential integrator does with pyJac's output through
 * sparse_multiplier(A, Vm, w) (create_jacobian.py:3301-3404), without the NSP^2 doubles per state ever
 * reaching memory when a mechanism-specific library is attached (reads 8(2 NSP + 1), writes 8 NSP bytes per
 * state: pj_lane.hip consumes the Jacobian in registers, pj_rblk.hip evaluates the product as a directional
 * derivative with every reaction visited once -- k_jvd); otherwise Jacobians are evaluated chunk-wise into a
 * temporary block and multiplied there. */
int pj_eval_jacobian_vec_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                             const double* d_v, double* d_w, int vw_layout, void* stream);
/* any output pointer may be NULL; outputs are SoA with leading dimension n:
 * conc[NSP], fwd[FWD_RATES], rev[REV_RATES], pres_mod[PRES_MOD_RATES],
 * spec_rates[NSP], dy[NSP] = [dT/dt, dY_0/dt ..] */
int pj_eval_rates_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                      double* d_conc, double* d_fwd, double* d_rev, double* d_pres_mod,
                      double* d_spec_rates, double* d_dy, void* stream);
/* Finite-difference Jacobian of dydt: the reference's comparison arm
 * (pyjac/performance_tester/fd_jacob.c:10-113, fd_jacob.cu:23-96: first order, CVODE-style
 * increment r_j = max(sqrt(eps)|y_j|, r0/ewt_j)); NSP+1 dydt launches.  y must be SoA. */
int pj_eval_fd_jacobian_dev(pj_mech* m, long n, const double* d_pres, const double* d_y,
                            double* d_jac, int jac_layout, void* stream);
/* ---- batched LU and Newton solves on per-state NSP x NSP blocks (the "batched LU" consumer).
 * pyJac hands one state's Jacobian to the caller's dense solver (docs/examples.rst:106-170: the per-state
 * integrator loop); there is no batched form in the reference -- these entry points take the blocks where
 * pj_eval_jacobian_dev leaves them: a_layout = PJ_LAYOUT_AOS: state-major, each block column-major
 * (a[s*NSP*NSP + r + NSP*c], pyJac's per-state C layout); PJ_LAYOUT_SOA: a[(r + NSP*c)*n + s], pyJac's batch layout
 * (the one the row-block kernels write at full speed).  Factors (d_lu, d_perm) are always per state.  NSP <= 16: four blocks per wavefront; NSP <= 64: one
 * wavefront per block -- a lane per row, the block in registers; 65 <= NSP <= 128: four wavefronts per block, the block in
 * their registers (factorisation and the fused solve; a solve from stored factors takes the LDS kernel); 129 <= NSP <= 140: one
 * workgroup per block, the block in LDS (PJ_EUNSUPPORTED beyond).  gamma != 0: the matrix factored is I - gamma * A (the Newton matrix of an implicit
 * step); gamma == 0: A itself.  Partial pivoting: a row of maximum magnitude, as LAPACK dgetf2; on an EXACT tie the
 * LDS-resident kernel (NSP > 128, rows exchanged physically) takes dgetf2's row -- the first in the current order --, the
 * register-resident ones (NSP <= 128, rows never exchanged) the lowest original row, which differs only when the tie involves a row
 * that an earlier step displaced; the
 * result is P A = L U with L unit lower triangular below the diagonal of d_lu, U on and above it, and
 * d_perm[s*NSP + k] = the row of A that became row k.  A singular block yields non-finite factors (no info
 * array).  d_lu may alias d_a (per-state layout only).  Device pointers, asynchronous on `stream`. */
int pj_lu_factor_dev(int nsp, long n, const double* d_a, int a_layout, double gamma, double* d_lu, int* d_perm, void* stream);
/* x_s = A_s^-1 b_s from the factors: d_b, d_x in vec_layout ([n][NSP] per state, or [NSP][n] state-fastest like y);
 * d_x may alias d_b */
int pj_lu_solve_dev(int nsp, long n, const double* d_lu, const int* d_perm, const double* d_b, double* d_x, int vec_layout,
                    void* stream);
/* factor and solve in one pass over the blocks: x_s = (I - gamma A_s)^-1 b_s (or A_s^-1 b_s); the factors stay in
 * registers and are written only if d_lu / d_perm are given (both or neither).  With a_layout = vec_layout =
 * PJ_LAYOUT_SOA this is the Newton step straight from what pj_eval_jacobian_dev(..., PJ_LAYOUT_SOA) wrote -- no
 * transposed copy of the Jacobians is ever made. */
int pj_newton_solve_dev(int nsp, long n, const double* d_a, int a_layout, double gamma, const double* d_b, double* d_x,
                        int vec_layout, double* d_lu, int* d_perm, void* stream);

/* Launch the Jacobian kernel `iters` times on `stream` bracketed by HIP events
 * recorded on that stream; *ms_per_launch receives the average. */
int pj_time_jacobian_dev(pj_mech* m, long n, const double* d_pres, const double* d_y,
                         int y_layout, double* d_jac, int jac_layout, void* stream,
                         int iters, double* ms_per_launch);

#ifdef PJ_TIMING
/* debug builds only (-DPJ_TIMING: libpyjac_hip_timing.so, tools/phase_cycles.py): per-phase cycle
 * counts of the table-driven kernel's first 64 workgroups (d_dbg: 640 doubles) */
int pj_debug_phase_cycles(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                          double* d_jac, int jac_layout, double* d_dbg);
#endif

/* ---- host-pointer batch driver: pyjacob.cuh:6-10 init / run / cleanup ---- */
/* returns padded (>= 1, multiple of 64; may be < num when device memory is
 * short, the caller then chunks as test.py:709-714 does) or a negative code */
int pj_init(pj_mech* m, int num);
/* pyjacob.cu:134-188: all arrays SoA with leading dimension num.  Blocking like the reference's run(), but it copies,
 * launches and copies back on a stream that belongs to the handle and returns after a wait for THAT stream: no device-wide
 * synchronisation, other streams of the process keep running (the per-state functions below do the same). */
int pj_run(pj_mech* m, int num, int padded, const double* pres, const double* y,
           double* conc, double* fwd_rxn_rates, double* rev_rxn_rates, double* pres_mod,
           double* spec_rates, double* dy, double* jac);
int pj_cleanup(pj_mech* m);

/* ---- per-state host functions: the prototypes of pyjacob_wrapper.pyx:4-16 ---- */
int pj_dydt(pj_mech* m, double t, double pres, const double* y, double* dy);
int pj_eval_jacob(pj_mech* m, double t, double pres, const double* y, double* jac);
int pj_eval_rxn_rates(pj_mech* m, double T, double pres, const double* C, double* fwd, double* rev);
int pj_eval_spec_rates(pj_mech* m, const double* fwd, const double* rev, const double* pres_mod,
                       double* sp_rates, double* dy_N);
int pj_get_rxn_pres_mod(pj_mech* m, double T, double pres, const double* C, double* pres_mod);
int pj_eval_conc(pj_mech* m, double T, double pres, const double* mass_frac, double* y_N,
                 double* mw_avg, double* rho, double* conc);

#ifdef __cplusplus
}
#endif
#endif
