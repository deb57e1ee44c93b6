/*
 * pyjac_oracle_quad.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Extended-precision "truth" for conditioning studies: the very text of pyjac_oracle.c (pyJac's
 * per-reaction formulation and evaluation order, see that file's header) compiled with every
 * `double` turned into `__float128` and every libm call into its libquadmath counterpart.  Tables
 * and inputs are the same binary64 numbers (converted exactly); every intermediate carries 113
 * mantissa bits; results are rounded to binary64 once, at the very end.  So |x - truth| measures
 * the rounding error a binary64 evaluation ORDER accumulates -- the reference's (pyjac_oracle.c,
 * oracle/_ref) or the HIP kernels' regrouped one -- and nothing else.
 *
 * Only tests/ may load this library (tests/test_conditioning.py).  CPU only, slow (software
 * binary128): a 53-species Jacobian takes ~10 ms.
 */
#include <float.h>
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double f64;

#define double __float128
#define exp expq
#define log logq
#define log10 log10q
#define pow powq
#define fmax fmaxq
#define fabs fabsq
#define sqrt sqrtq
#define floor floorq
/* internal (binary128) entry points get their own names */
#define pjo_mech pjq_mech
#define pjo_create pjq_create_
#define pjo_destroy pjq_destroy_
#define pjo_nsp pjq_nsp_
#define pjo_nrxn pjq_nrxn_
#define pjo_nrev pjq_nrev_
#define pjo_npres pjq_npres_
#define pjo_eval_conc pjq_eval_conc_
#define pjo_eval_rxn_rates pjq_eval_rxn_rates_
#define pjo_get_rxn_pres_mod pjq_get_rxn_pres_mod_
#define pjo_eval_spec_rates pjq_eval_spec_rates_
#define pjo_eval_h pjq_eval_h_
#define pjo_eval_cp pjq_eval_cp_
#define pjo_dydt pjq_dydt_
#define pjo_set_sum_last_species pjq_set_sum_last_species
#define pjo_eval_jacob pjq_eval_jacob_
#define pjo_fd_jacob pjq_fd_jacob_
#define pjo_batch_jacob pjq_batch_jacob_
#define pjo_batch_dydt pjq_batch_dydt_
#include "pyjac_oracle.c"
#undef double

/* ---- binary64 interface ---- */
pjq_mech *pjq_create(const int32_t *I, long nI, const f64 *D, long nD)
{
    __float128 *Q = (__float128 *)malloc(sizeof(__float128) * (nD > 0 ? nD : 1));
    for (long q = 0; q < nD; ++q) Q[q] = D[q];
    pjq_mech *m = pjq_create_(I, nI, Q, nD);
    free(Q);
    return m;
}
void pjq_destroy(pjq_mech *m) { pjq_destroy_(m); }

/* everything the functional tester looks at, for one state y = [T, Y_0..Y_{NSP-2}] */
void pjq_eval_all(const pjq_mech *m, f64 pres, const f64 *y, f64 *conc, f64 *fwd, f64 *rev, f64 *pres_mod,
                  f64 *spec_rates, f64 *dy, f64 *jac)
{
    const int n = m->nsp, R = m->nrxn;
    __float128 *b = (__float128 *)calloc((size_t)(4 * n + 3 * R + n * n + 8), sizeof(__float128));
    __float128 *yq = b, *cq = yq + n, *fq = cq + n, *rq = fq + R, *pq = rq + R, *sq = pq + R, *dq = sq + n,
               *jq = dq + n;
    for (int k = 0; k < n; ++k) yq[k] = y[k];
    __float128 yN, mw, rho;
    pjq_eval_conc_(m, yq[0], pres, yq + 1, &yN, &mw, &rho, cq);
    pjq_eval_rxn_rates_(m, yq[0], pres, cq, fq, rq);
    pjq_get_rxn_pres_mod_(m, yq[0], pres, cq, pq);
    pjq_eval_spec_rates_(m, fq, rq, pq, sq, &sq[n - 1]);
    pjq_dydt_(m, 0.0, pres, yq, dq);
    pjq_eval_jacob_(m, 0.0, pres, yq, jq);
    for (int k = 0; k < n; ++k) {
        if (conc) conc[k] = (f64)cq[k];
        if (spec_rates) spec_rates[k] = (f64)sq[k];
        if (dy) dy[k] = (f64)dq[k];
    }
    for (int i = 0; i < R; ++i) if (fwd) fwd[i] = (f64)fq[i];
    for (int i = 0; i < m->nrev; ++i) if (rev) rev[i] = (f64)rq[i];
    for (int i = 0; i < m->npres; ++i) if (pres_mod) pres_mod[i] = (f64)pq[i];
    if (jac) for (int e = 0; e < n * n; ++e) jac[e] = (f64)jq[e];
    free(b);
}

/* state-major batch of Jacobians, OpenMP over states */
void pjq_batch_jacob(const pjq_mech *m, long num, const f64 *pres, const f64 *y_aos, f64 *jac_aos, int nthreads)
{
    const long n = m->nsp;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic)
#endif
    for (long s = 0; s < num; ++s)
        pjq_eval_all(m, pres[s], y_aos + s * n, NULL, NULL, NULL, NULL, NULL, NULL, jac_aos + s * n * n);
}

void pjq_batch_dydt(const pjq_mech *m, long num, const f64 *pres, const f64 *y_aos, f64 *dy_aos, int nthreads)
{
    const long n = m->nsp;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic)
#endif
    for (long s = 0; s < num; ++s) {
        __float128 *b = (__float128 *)malloc(sizeof(__float128) * 2 * n);
        for (int k = 0; k < n; ++k) b[k] = y_aos[s * n + k];
        pjq_dydt_(m, 0.0, pres[s], b, b + n);
        for (int k = 0; k < n; ++k) dy_aos[s * n + k] = (f64)b[n + k];
        free(b);
    }
}
