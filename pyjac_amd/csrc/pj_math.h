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
#ifdef PJM_FAKE_EXP
    // timing experiment (results wrong): what would the kernels gain if an exponential cost two instructions?
    y0 = __builtin_fma(x0, 0x1.0000001p-30, 1.0); y1 = __builtin_fma(x1, 0x1.0000001p-30, 1.0);
    return;
#endif
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


// Table-driven exponential (Tang's scheme): x = (64 k + j) ln2 / 64 + r, |r| <= ln2 / 128, exp(x) = 2^k T_j (1 + P(r)) with
// T_j = 2^(j/64) from a 512-byte table and a degree-5 P (truncation r^6 / 720 < 3.5e-17): 11 fp64 instructions + 3 integer ones
// + one table read instead of 17 -- the row kernels evaluate 1.2 k exponentials per state, a quarter of their fp64 work
// (round 6).  The rounding of x * 64 / ln2 is the 1.5 * 2^52 trick: the low dword of the sum is n.  Error: the table entry's
// half ulp + the final fused multiply-add's: < 1.5 ulp (the polynomial form above: < 1).  No range selects: ldexp saturates.
constexpr double PJM_EXPT[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
// argument reduction: r and n (k = n >> 6, j = n & 63)
__device__ __forceinline__ void expt_reduce(const double x, double& r, int& n)
{
    constexpr double INV = 0x1.71547652b82fep+6, MAGIC = 0x1.8p52, NCH = -0x1.62e42fef80000p-7, NCL = -0x1.1cf79abc9e3b4p-42;
    const double t = __builtin_fma(x, INV, MAGIC);
    n = (int)(unsigned)__builtin_bit_cast(unsigned long long, t);
    const double nd = t - MAGIC;
    r = __builtin_fma(nd, NCL, __builtin_fma(nd, NCH, x));
}
__device__ __forceinline__ double expt_finish(const double r, const int n, const double tj)
{
    const double q = __builtin_fma(__builtin_fma(__builtin_fma(0x1.1111111111111p-7, r, 0x1.5555555555555p-5), r, 0x1.5555555555555p-3), r, 0.5);
    const double e = __builtin_fma(r * r, q, r);
    return __builtin_ldexp(__builtin_fma(tj, e, tj), n >> 6);
}
// tab: the table in LDS (a per-lane read: lanes differ in j)
__device__ __forceinline__ double exp_tab(const double x, const double* tab)
{
    double r;
    int n;
    expt_reduce(x, r, n);
    return expt_finish(r, n, tab[n & 63]);
}
__device__ __forceinline__ void exp_tab_pair(const double x0, const double x1, double& y0, double& y1, const double* tab)
{
    double r0, r1;
    int n0, n1;
    expt_reduce(x0, r0, n0);
    expt_reduce(x1, r1, n1);
    const double t0 = tab[n0 & 63], t1 = tab[n1 & 63];
    y0 = expt_finish(r0, n0, t0);
    y1 = expt_finish(r1, n1, t1);
}

// Natural logarithm of a positive NORMAL double (callers clamp with fmax(x, 1e-300)): fdlibm's __ieee754_log without its
// special cases (< 1 ulp) -- mantissa in [sqrt(1/2), sqrt(2)), s = f / (2 + f), a 7-term polynomial in s^2, the exponent's
// share of ln 2 in two pieces.  ~40 instructions; the device library's log() is 92 and its exp() 36: a Troe reaction of the
// pre-pass calls them twice / seven times, which was ALL of that kernel's work (round 6).
__device__ __forceinline__ double log_lean(const double x)
{
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01, LG3 = 2.857142874366239149e-01,
                     LG4 = 2.222219843214978396e-01, LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
                     LG7 = 1.479819860511658591e-01;
    int e;
    double m = __builtin_frexp(x, &e);              // [0.5, 1)
    const bool lo = m < 0x1.6a09e667f3bcdp-1;       // sqrt(1/2)
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    // (a reciprocal and a product: under -freciprocal-math v_rcp_f64 + two Newton steps, where f / (2 + f) is a 14-instruction
    // IEEE division sequence)
    const double s = f * (1.0 / (2.0 + f));
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, LG6, LG4), LG2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, LG7, LG5), LG3), LG1);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    return __builtin_fma(dk, LN2_HI, -((hfsq - __builtin_fma(s, hfsq + R, dk * LN2_LO)) - f));
}
