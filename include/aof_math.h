/*
 * aof_math.h -- the one transcendental of the path, pixel flow -> angular flow
 * (atan2(flow_px, focal_px), DESIGN.md "Facade semantics"), as ONE fixed sequence of IEEE-754 double
 * operations, so that the C++ facade on the host, any CPU checker of it and the device pipeline
 * (aof_sequence_device) return the same float bit for bit.  libm's atan2f is not specified to the last
 * bit (glibc's and the device library's differ), which is why neither is used.
 *
 * Accuracy: the argument is reduced to |u| <= tan(pi/8) and atan(u) summed as its Taylor series to u^29
 * in double (truncation < 1e-12 relative), so the double result is within a few double ulps of atan2 and
 * the float it rounds to is the correctly rounded one except when atan2 falls within ~1e-12 (relative) of
 * a float rounding boundary: <= 1 float ulp always (tests/test_math.py checks against math.atan2).
 *
 * Plain C99 / C++11 / HIP.  No fused multiply-adds: contraction is switched off for this function (an
 * fma rounds once where mul + add round twice, and x86-64 hosts without FMA could not follow).
 */
#ifndef AOF_MATH_H
#define AOF_MATH_H

#if defined(__HIPCC__)
#define AOF_MATH_FN __host__ __device__ static inline
#else
#define AOF_MATH_FN static inline
#endif

#if defined(__GNUC__) && !defined(__clang__)
#define AOF_MATH_NO_CONTRACT __attribute__((optimize("fp-contract=off")))
#else
#define AOF_MATH_NO_CONTRACT
#endif

AOF_MATH_NO_CONTRACT AOF_MATH_FN float aof_atan2f(float yf, float xf)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double pi = 3.14159265358979323846, half_pi = 1.57079632679489661923, quarter_pi = 0.78539816339744830962;
    const double tan_pi_8 = 0.41421356237309504880;
    double y = (double)yf, x = (double)xf;
    double ax, ay, lo, hi, t, u, z, p, base, r;
    int k;
    if (x != x || y != y) return xf + yf; /* NaN in, NaN out */
    ax = x < 0.0 ? -x : x;
    ay = y < 0.0 ? -y : y;
    lo = ax < ay ? ax : ay;
    hi = ax < ay ? ay : ax;
    if (hi == 0.0) {
        t = 0.0; /* atan2(+-0, +-0): the angle of the x sign alone */
    } else if (lo == hi) {
        t = 1.0; /* also inf / inf */
    } else {
        t = lo / hi; /* in [0, 1) */
    }
    if (t > tan_pi_8) {
        u = (t - 1.0) / (t + 1.0); /* atan(t) = pi/4 + atan((t - 1) / (t + 1)), |u| <= tan(pi/8) */
        base = quarter_pi;
    } else {
        u = t;
        base = 0.0;
    }
    z = u * u;
    /* atan(u) = u (1 - z/3 + z^2/5 - ... + z^14/29), Horner from the last term, mul and add rounded apart */
    p = 1.0 / 29.0;
    for (k = 27; k >= 1; k -= 2) {
        p = p * z;
        p = 1.0 / (double)k - p;
    }
    r = u * p;
    r = base + r;
    if (ay > ax) r = half_pi - r;
    /* the sign bit of x, -0.0 included: 1 / -0.0 = -inf */
    if (x < 0.0 || (x == 0.0 && 1.0 / x < 0.0)) r = pi - r;
    if (y < 0.0 || (y == 0.0 && 1.0 / y < 0.0)) r = -r;
    return (float)r;
}

#endif
