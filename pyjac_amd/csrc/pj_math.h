// pj_math.h -- exponentials for the state-per-lane kernels (pj_lane.hip, pj_rblk.hip).
// Included inside their anonymous namespaces after static_for / PJR_INL (or PJL_INL) are defined.
#pragma once
#ifndef PJM_INL
#define PJM_INL __attribute__((always_inline))
#endif
// exp(x) with the arithmetic of the device library's double-precision exp (same reduction and
// polynomial) minus its range selects: ldexp saturates to 0 / inf by itself.  Two arguments at once,
// statement by statement: the two Horner chains are independent and a lane at one wavefront per
// SIMD has nothing else to fill the fp64 pipeline latency with.
__device__ __forceinline__ void exp_pair(const double x0, const double x1, double& y0, double& y1)
{
    constexpr double LOG2E = 0x1.71547652b82fep+0, NLN2H = -0x1.62e42fefa39efp-1, NLN2L = -0x1.abc9e3b39803fp-56;
    constexpr double C[10] = {0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22, 0x1.71dee623fde64p-19, 0x1.a01997c89e6bp-16,
                              0x1.a01a014761f6ep-13, 0x1.6c16c1852b7bp-10, 0x1.1111111122322p-7, 0x1.55555555502a1p-5,
                              0x1.5555555555511p-3, 0x1.000000000000bp-1};
    const double n0 = __builtin_rint(x0 * LOG2E), n1 = __builtin_rint(x1 * LOG2E);
    double r0 = __builtin_fma(n0, NLN2H, x0), r1 = __builtin_fma(n1, NLN2H, x1);
    r0 = __builtin_fma(n0, NLN2L, r0); r1 = __builtin_fma(n1, NLN2L, r1);
    double p0 = __builtin_fma(C[0], r0, C[1]), p1 = __builtin_fma(C[0], r1, C[1]);
    static_for<8>([&](auto cc) PJM_INL {
        constexpr int c = decltype(cc)::value + 2;
        p0 = __builtin_fma(p0, r0, C[c]); p1 = __builtin_fma(p1, r1, C[c]);
    });
    p0 = __builtin_fma(r0, p0, 1.0); p1 = __builtin_fma(r1, p1, 1.0);
    p0 = __builtin_fma(r0, p0, 1.0); p1 = __builtin_fma(r1, p1, 1.0);
    y0 = __builtin_ldexp(p0, (int)n0); y1 = __builtin_ldexp(p1, (int)n1);
}
__device__ __forceinline__ double exp_one(const double x)
{
    double y0, y1;
    exp_pair(x, x, y0, y1);
    (void)y1;
    return y0;
}

