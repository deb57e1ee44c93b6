/*
 * ref_driver.c -- TEST INFRASTRUCTURE.  Batch loop around the REFERENCE's
 * generated eval_jacob / dydt, linked into oracle/_ref/libpyjac_ref_<mech>.so
 * by oracle/build_ref.py.  Protocol of the reference speed test
 * (pyjac/performance_tester/tester.c.in:23-31): one OpenMP parallel-for over
 * states, each iteration evaluating into a zeroed NSP*NSP block.
 */
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "mechanism.h"

void eval_jacob(const double t, const double pres, const double *y, double *jac);
void dydt(const double t, const double pres, const double *y, double *dy);

int ref_nsp(void) { return NSP; }
int ref_fwd_rates(void) { return FWD_RATES; }
int ref_rev_rates(void) { return REV_RATES; }
int ref_pres_mod_rates(void) { return PRES_MOD_RATES; }

void ref_batch_jacob(long num, const double *pres, const double *y_aos, double *jac_aos, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (long s = 0; s < num; ++s) {
        double *jac = jac_aos + s * NSP * NSP;
        memset(jac, 0, sizeof(double) * NSP * NSP);
        eval_jacob(0.0, pres[s], y_aos + s * NSP, jac);
    }
}

void ref_batch_dydt(long num, const double *pres, const double *y_aos, double *dy_aos, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (long s = 0; s < num; ++s)
        dydt(0.0, pres[s], y_aos + s * NSP, dy_aos + s * NSP);
}
