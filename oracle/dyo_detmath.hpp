// ORACLE — TEST INFRASTRUCTURE ONLY (see dyo_formats.hpp header).
// Fixed IEEE-754 evaluation of 10^x, e^x and erfc(x): only +, -, *, / in double, so the CPU oracle and the device code
// (which carries its own copy of the same formulas, dream_yara_amd/csrc/dy_detmath.h) produce bit-identical results.
// The reference calls std::pow / std::erfc (src/bits_matches.h:898-966); these agree with libm to a few ulp, which only
// matters for exact ties of pair weights.  Paired-end output is unpinned by any reference fixture anyway.
#pragma once
#include <cstdint>
#include <cstring>

namespace dyo {

inline double detScale2(double v, int n)      // v * 2^n, n within the normal range
{
    uint64_t bits = (uint64_t)(1023 + n) << 52;
    double s; memcpy(&s, &bits, 8);
    return v * s;
}

inline double detExp2(double y)
{
    if (y < -1000.0) return 0.0;
    if (y > 1000.0) y = 1000.0;
    double fl = (double)(long long)y;             // trunc
    if (fl > y) fl -= 1.0;                        // floor
    double f = y - fl;                            // [0,1)
    int n = (int)fl;
    if (f > 0.5) { f -= 1.0; n += 1; }            // [-0.5, 0.5]
    const double z = f * 0.6931471805599453094;   // ln 2
    // e^z, |z| <= 0.347: Taylor to z^17
    double term = 1.0, sum = 1.0;
    for (int i = 1; i <= 17; ++i) { term = term * z / (double)i; sum = sum + term; }
    return detScale2(sum, n);
}

inline double detPow10(double x) { return detExp2(x * 3.3219280948873623479); }     // log2(10)
inline double detExp(double x)   { return detExp2(x * 1.4426950408889634074); }     // log2(e)

// complementary error function for x >= 0 (Chebyshev fit, Numerical Recipes 3rd ed. erfccheb, ~1.2e-16 relative)
inline double detErfc(double x)
{
    static const double cof[28] = {-1.3026537197817094, 6.4196979235649026e-1, 1.9476473204185836e-2, -9.561514786808631e-3,
        -9.46595344482036e-4, 3.66839497852761e-4, 4.2523324806907e-5, -2.0278578112534e-5, -1.624290004647e-6, 1.303655835580e-6,
        1.5626441722e-8, -8.5238095915e-8, 6.529054439e-9, 5.059343495e-9, -9.91364156e-10, -2.27365122e-10, 9.6467911e-11,
        2.394038e-12, -6.886027e-12, 8.94487e-13, 3.13092e-13, -1.12708e-13, 3.81e-16, 7.106e-15, -1.523e-15, -9.4e-17, 1.21e-16, -2.8e-17};
    const bool neg = x < 0.0;
    const double z = neg ? -x : x;
    const double t = 2.0 / (2.0 + z);
    const double ty = 4.0 * t - 2.0;
    double d = 0.0, dd = 0.0;
    for (int j = 27; j > 0; --j) { const double tmp = d; d = ty * d - dd + cof[j]; dd = tmp; }
    const double r = t * detExp(-z * z + 0.5 * (cof[0] + ty * d) - dd);
    return neg ? 2.0 - r : r;
}

} // namespace dyo
