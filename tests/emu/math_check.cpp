// Host build of csrc/pj_math.h (test infrastructure): exp_one / exp_pair, the table-driven exp_tab and log_lean against
// long-double libm over the argument ranges the kernels use.  Prints "name max_relative_error" lines.
#include <cmath>
#include <cstdio>
#include <type_traits>
#include <utility>
#define __device__
#define __forceinline__ inline
#define PJM_INL
template <int N, class F> inline void static_for(F&& f) { if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); } }
#include "pj_math.h"
int main()
{
    double w_exp = 0, w_tab = 0, w_log = 0, w_pair = 0;
    for (long i = 0; i < 3000000; ++i) {
        const double u = (double)((i * 2654435761ul) % 1000003ul) / 1000003.0;
        const double x = -700.0 + 1400.0 * u;
        const long double e = expl((long double)x);
        const double a = exp_one(x), b = exp_tab(x, PJM_EXPT);
        double p0, p1;
        exp_pair(x, -x, p0, p1);
        w_exp = fmax(w_exp, fabs((double)((a - e) / e)));
        w_tab = fmax(w_tab, fabs((double)((b - e) / e)));
        w_pair = fmax(w_pair, fmax(fabs((double)((p0 - e) / e)), fabs((double)((p1 - 1.0L / e) * e))));
        const double y = (i % 3 == 0) ? exp(-690.0 + 1380.0 * u) : (i % 3 == 1) ? 0.5 + 1.5 * u : 1.0 + (u - 0.5) * 1e-3;
        const long double l = logl((long double)y);
        const double c = log_lean(y);
        w_log = fmax(w_log, l == 0 ? fabs(c) : fabs((double)((c - l) / l)));
    }
    printf("exp_one %.3g\nexp_pair %.3g\nexp_tab %.3g\nlog_lean %.3g\n", w_exp, w_pair, w_tab, w_log);
    printf("edge %g %g %g %.17g\n", exp_one(-800.0), exp_one(800.0), exp_tab(0.0, PJM_EXPT), log_lean(1e-300));
    return 0;
}
