// x^p for x > 0 in float64, for the colour kernels (dwt.hip).  (The CPU checker oracle/color_oracle.c does not include this
// file: it uses the C library's pow(); tests/native/spow_probe.c measures this function against 60-digit arithmetic.)
//
// The colour model change (RGB <-> IPT: spiht/color_models.py:6-13 -> colour-science) takes three signed powers per
// pixel and is arithmetic-bound: the device library's pow() costs about 250 float64 instruction slots (it carries the
// logarithm in double-double to stay under one ulp), exp(p * log(x)) about 200.  This one costs about 40:
//   log2 x = e + log2 c_i + log2(1 + r),   x = 2^e m,  c_i the centre of the 1/64-wide interval m falls in,
//            r = m / c_i - 1 by ONE fma with the tabulated reciprocal (|r| <= 2^-7), degree-8 Taylor polynomial;
//   2^y    = 2^q 2^(j/64) e^u,  y = (64 q + j) / 64 + t,  u = t ln 2 (|u| <= 0.0055), degree-6 Taylor polynomial.
// Error: under 4 units in the last place whatever the magnitudes (the integer part of p log2 x is handled exactly;
// checked against 60-digit arithmetic in tests/test_oracle.py); numpy's pow, which colour-science calls, stays under 1.  The reference's own arithmetic for this step cannot be
// consulted (colour-science is not installable here): colour parity is unpinned with any power function.
#ifndef SPIHT_SPOW_H
#define SPIHT_SPOW_H
#include <stdint.h>
#include <string.h>

#ifndef SPOW_FN
#define SPOW_FN static inline
#endif
#ifndef SPOW_FMA
#include <math.h>
#define SPOW_FMA(a, b, c) fma((a), (b), (c))
#define SPOW_RINT(a) rint(a)
#define SPOW_LDEXP(a, n) ldexp((a), (n))
#endif

// inv / log2c / exp2t: the three tables of spow_tables.h, wherever the caller keeps them (LDS on the device)
SPOW_FN double spow_pos(double ax, double p, const double *inv, const double *log2c, const double *exp2t) {
    int eadj = 0;
    if (ax < 0x1p-1022) { ax = ax * 0x1p54; eadj = -54; }  // subnormal
    uint64_t b;
    memcpy(&b, &ax, 8);
    const int e = (int)(b >> 52) - 1023 + eadj;
    const uint64_t mant = b & 0xFFFFFFFFFFFFFull;
    const int i = (int)(mant >> 46);
    const uint64_t mb = mant | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &mb, 8);
    const double r = SPOW_FMA(m, inv[i], -1.0);
    double pl = SPOW_L8;
    pl = SPOW_FMA(pl, r, SPOW_L7);
    pl = SPOW_FMA(pl, r, SPOW_L6);
    pl = SPOW_FMA(pl, r, SPOW_L5);
    pl = SPOW_FMA(pl, r, SPOW_L4);
    pl = SPOW_FMA(pl, r, SPOW_L3);
    pl = SPOW_FMA(pl, r, SPOW_L2);
    pl = SPOW_FMA(pl, r, SPOW_L1);
    // log2 x = ed + f, f in [0, 1].  The product p * log2 x is never formed as one rounded number (its rounding error
    // would grow with |p log2 x|: 13 ulp in the result at x = 1e-3, p = 1/0.43): n / 64 is taken off the EXACT product
    // p * ed inside one fma, and p * f -- small -- is added to the small remainder.
    const double ed = (double)e, f = log2c[i] + pl * r;
    const double y0 = p * (ed + f);
    if (y0 >= 1025.0) return 1.0 / 0.0;
    if (y0 <= -1080.0) return 0.0;
    const double n = SPOW_RINT(y0 * 64.0);
    const double t = SPOW_FMA(p, ed, n * -0x1p-6) + p * f;
    const double u = t * SPOW_LN2;
    double pe = SPOW_E6;
    pe = SPOW_FMA(pe, u, SPOW_E5);
    pe = SPOW_FMA(pe, u, SPOW_E4);
    pe = SPOW_FMA(pe, u, SPOW_E3);
    pe = SPOW_FMA(pe, u, SPOW_E2);
    pe = SPOW_FMA(pe, u, 1.0);
    const double w = pe * u;  // e^u - 1
    const int ni = (int)n;
    const double c = exp2t[ni & 63];
    return SPOW_LDEXP(SPOW_FMA(c, w, c), ni >> 6);
}

// sign(x) |x|^p, spow(0) = 0 (colour-science's spow); infinities and NaN pass through pow's rules loosely (not pixel data)
SPOW_FN double spow_signed(double x, double p, const double *inv, const double *log2c, const double *exp2t) {
    const double ax = x < 0.0 ? -x : x;
    if (!(ax > 0.0)) return x != x ? x : 0.0;
    if (ax > 0x1.fffffffffffffp+1023) return x;
    const double m = spow_pos(ax, p, inv, log2c, exp2t);
    return x < 0.0 ? -m : m;
}
#endif
