// rocco_amd/csrc/log2_cr.h -- log2 of a positive double, correctly rounded (double-double evaluation), gfx950.
//
// Used for the count path's log scale, log2(max(x, 0) + pseudocount) (rocco/inference.py:40-47).  The reference calls
// np.log2, which is NOT one function: NumPy dispatches to an AVX-512 SVML routine or to libm's log2, neither is
// correctly rounded, and they differ from each other (in this image: 0.03 % of integer counts are one ulp off the
// correctly rounded value).  The correctly rounded value is the only host-independent target; this evaluates it:
//   t = m 2^e, m in [0.75, 1.5);  c = 0.75 + i / 128 nearest to m;  r = (m - c) / c as a double-double, |r| <= 1/192;
//   log2 t = e + log2 c (table, double-double) + (r - r^2/2 + ... - r^14/14) / ln 2, the first seven terms in
//   double-double arithmetic (error ~2^-100 relative: a wrong rounding needs the true value that close to a
//   rounding boundary); powers of two come out exact.
#pragma once

#include "log2_table.h"

namespace rocco {

struct dd {
    double hi, lo;
};

__device__ __forceinline__ dd dd_quick(double a, double b)  // |a| >= |b|
{
    const double s = a + b;
    return {s, b - (s - a)};
}

__device__ __forceinline__ dd dd_two_sum(double a, double b)
{
    const double s = a + b;
    const double bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}

__device__ __forceinline__ dd dd_add(dd a, dd b)
{
    dd s = dd_two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return dd_quick(s.hi, s.lo);
}

__device__ __forceinline__ dd dd_mul(dd a, dd b)
{
    const double p = a.hi * b.hi;
    double e = fma(a.hi, b.hi, -p);
    e += a.hi * b.lo + a.lo * b.hi;
    return dd_quick(p, e);
}

// The full evaluation: error ~2^-100 relative (a wrong rounding needs the true value that close to a rounding boundary).
__device__ __forceinline__ double log2_cr_full(double t)
{
    // (callers pass finite t > 0)
    long long bits = __double_as_longlong(t);
    int e = (int)((bits >> 52) & 0x7FF);
    if (e == 0) {  // subnormal: scale up first
        t *= 0x1p54;
        bits = __double_as_longlong(t);
        e = (int)((bits >> 52) & 0x7FF) - 54;
    }
    e -= 1023;
    double m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFLL) | 0x3FF0000000000000LL);  // [1, 2)
    if (m >= 1.5) {
        m *= 0.5;
        e += 1;
    }
    const int i = (int)((m - 0.75) * 128.0 + 0.5);  // nearest c = 0.75 + i / 128, 0 <= i <= 96
    const double c = 0.75 + (double)i * 0.0078125;
    const double z = m - c;                           // exact
    dd r;
    r.hi = z / c;
    r.lo = fma(-r.hi, c, z) / c;                      // the division's remainder, exactly
    // series: s = r (1 - r/2 + r^2/3 - ... ) ; inner Horner from the small end: terms 1/8 .. 1/14 in plain doubles
    double tail = -1.0 / 14.0;
    tail = fma(tail, r.hi, 1.0 / 13.0);
    tail = fma(tail, r.hi, -1.0 / 12.0);
    tail = fma(tail, r.hi, 1.0 / 11.0);
    tail = fma(tail, r.hi, -1.0 / 10.0);
    tail = fma(tail, r.hi, 1.0 / 9.0);
    dd acc = dd_add({-kRecip[6][0], -kRecip[6][1]}, dd_mul(r, {tail, 0.0}));            // -1/8 + r (1/9 - ...)
    acc = dd_add({kRecip[5][0], kRecip[5][1]}, dd_mul(r, acc));                          // 1/7
    acc = dd_add({-kRecip[4][0], -kRecip[4][1]}, dd_mul(r, acc));                        // -1/6
    acc = dd_add({kRecip[3][0], kRecip[3][1]}, dd_mul(r, acc));                          // 1/5
    acc = dd_add({-kRecip[2][0], -kRecip[2][1]}, dd_mul(r, acc));                        // -1/4
    acc = dd_add({kRecip[1][0], kRecip[1][1]}, dd_mul(r, acc));                          // 1/3
    acc = dd_add({-kRecip[0][0], -kRecip[0][1]}, dd_mul(r, acc));                        // -1/2
    acc = dd_add({1.0, 0.0}, dd_mul(r, acc));                                            // 1 - r/2 + ...
    dd ln1p = dd_mul(r, acc);
    dd l2 = dd_mul(ln1p, {kInvLn2Hi, kInvLn2Lo});
    // e + log2(c) + log2(1 + r): the table entry and e are added as double-doubles (no cancellation beyond one bit:
    // for e >= 1 the table entry is >= -0.415, for e == 0 and c == 1 it is 0)
    dd sum = dd_add({kLog2C[i][0], kLog2C[i][1]}, l2);
    sum = dd_add({(double)e, 0.0}, sum);
    return sum.hi;
}

// Two stages (Ziv): the same reduction, then  log1p(r) = r - r^2/2 + r^3 (1/3 - r/4 + ... - r^7/10)  with r and r^2 as
// double-doubles and the bracket in plain doubles.  Error of this stage: the bracket's rounding, <= ~3 ulp of r^3 / 3 =
// 2^-53 |r|^3 <= 2^-68 |r| (|r| <= 2^-7.58); truncation |r|^11 / 11 <= 2^-79 |r|; everything else ~2^-104.  Relative to the
// result y = e + log2(c) + log2(1 + r): y is at least 0.41 in magnitude unless e = 0 and c = 1, where y = log2(1 + r) itself,
// so the error is below 2^-67 |y| in every case.  The result (hi, lo) is accepted when hi survives an error of 2^-64 |hi|
// on either side (8x the bound); otherwise -- about one value in two thousand -- the full evaluation decides.
__device__ __forceinline__ double log2_correctly_rounded(double t)
{
    // (callers pass finite t > 0)
#ifdef ROCCO_LOG2_FULL_ONLY  // (timing experiments only)
    return log2_cr_full(t);
#endif
    const double t_in = t;
    long long bits = __double_as_longlong(t);
    int e = (int)((bits >> 52) & 0x7FF);
    if (e == 0) {  // subnormal: scale up first
        t *= 0x1p54;
        bits = __double_as_longlong(t);
        e = (int)((bits >> 52) & 0x7FF) - 54;
    }
    e -= 1023;
    double m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFLL) | 0x3FF0000000000000LL);  // [1, 2)
    if (m >= 1.5) {
        m *= 0.5;
        e += 1;
    }
    const int i = (int)((m - 0.75) * 128.0 + 0.5);  // nearest c = 0.75 + i / 128, 0 <= i <= 96
    const double c = 0.75 + (double)i * 0.0078125;
    const double z = m - c;                           // exact
    const double h = z / c;
    const double l = fma(-h, c, z) / c;               // r = h + l to ~2^-106
    // r^2 as a double-double
    const double q_hi = h * h;
    const double q_lo = fma(h, h, -q_hi) + 2.0 * (h * l);
    // r^3 (1/3 - r/4 + r^2/5 - r^3/6 + r^4/7 - r^5/8 + r^6/9 - r^7/10)
    double p = -1.0 / 10.0;
    p = fma(p, h, 1.0 / 9.0);
    p = fma(p, h, -1.0 / 8.0);
    p = fma(p, h, 1.0 / 7.0);
    p = fma(p, h, -1.0 / 6.0);
    p = fma(p, h, 1.0 / 5.0);
    p = fma(p, h, -1.0 / 4.0);
    p = fma(p, h, 1.0 / 3.0);
    const double cube = (q_hi * h) * p;
    // log1p(r) = (h - q_hi / 2) + (l - q_lo / 2 + cube)
    dd s = dd_two_sum(h, -0.5 * q_hi);
    s.lo += (l - 0.5 * q_lo) + cube;
    const dd ln1p = dd_quick(s.hi, s.lo);
    const dd l2 = dd_mul(ln1p, {kInvLn2Hi, kInvLn2Lo});
    dd sum = dd_add({kLog2C[i][0], kLog2C[i][1]}, l2);
    sum = dd_add({(double)e, 0.0}, sum);
    const double slack = fabs(sum.hi) * 0x1p-64;
    if (sum.hi + (sum.lo + slack) == sum.hi && sum.hi + (sum.lo - slack) == sum.hi) {
        return sum.hi;
    }
    return log2_cr_full(t_in);
}

}  // namespace rocco
