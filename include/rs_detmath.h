/* rs_detmath.h -- deterministic double-precision elementary functions.
 *
 * The RAN-slice step turns floating-point values into integer decisions (RB counts,
 * reception outcomes, SLA violation counts).  For the HIP path and the CPU oracle to
 * agree bit-for-bit on those integers BY CONSTRUCTION, every transcendental on the path
 * is evaluated with the routines below, which use only IEEE-754 correctly rounded
 * primitives (+, -, *, /, sqrt, fma) in a fixed order.  The same text compiles as C
 * (gcc, oracle) and as HIP device code (hipcc, gfx950); both translation units are built
 * with -ffp-contract=off so that no fused multiply-add is introduced or removed behind
 * our back.  Accuracy is ~1-2 ulp, far inside the 1e-12 relative tolerance at which the
 * oracle is pinned against the numpy reference (tests/golden, fixture G2).
 *
 * Where the reference calls these: np.exp/np.log in sigmoid/inv_sigmoid
 * (reference channel_models.py:35-41), np.log10/np.arccos/np.sqrt in macro_cell/location
 * (channel_models.py:62-68, 84-97).
 */
#ifndef RS_DETMATH_H
#define RS_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define RS_HD __host__ __device__ static inline
#define RS_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define RS_SQRT(a) __builtin_sqrt(a)
#define RS_RINT(a) __builtin_rint(a)
#else
#include <math.h>
#define RS_HD static inline
#define RS_FMA(a, b, c) fma((a), (b), (c))
#define RS_SQRT(a) sqrt(a)
#define RS_RINT(a) rint(a)
#endif

#define RS_PI 3.141592653589793
#define RS_PI_2 1.5707963267948966
#define RS_RAD2DEG 57.29577951308232
#define RS_LOG10E 0.43429448190325176
#define RS_INV_LN2 1.4426950408889634
#define RS_LN2_HI 6.93147180369123816490e-01 /* 0x1.62e42fee00000p-1, low 21 bits zero */
#define RS_LN2_LO 1.90821492927058770002e-10 /* 0x1.a39ef35793c76p-33 */

RS_HD uint64_t rs_d2u(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}

RS_HD double rs_u2d(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}

RS_HD double rs_inf(void) { return rs_u2d(0x7ff0000000000000ull); }
RS_HD double rs_nan(void) { return rs_u2d(0x7ff8000000000000ull); }

/* 2^k for k in [-1022, 1023] */
RS_HD double rs_pow2i(int k) { return rs_u2d((uint64_t)(k + 1023) << 52); }

/* exp(x): Cody-Waite reduction x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor in Horner
 * form (truncation 4e-18), scaled by 2^k in two exact steps. */
/* the arithmetic of rs_exp for finite x in [-745.2, 709.78]: no branches, so independent calls interleave */
RS_HD double rs_exp_core(double x) {
    double kd = RS_RINT(x * RS_INV_LN2);
    double r = RS_FMA(-kd, RS_LN2_HI, x);
    r = RS_FMA(-kd, RS_LN2_LO, r);
    double p = 1.6059043836821613e-10;
    p = RS_FMA(p, r, 2.08767569878681e-09);
    p = RS_FMA(p, r, 2.505210838544172e-08);
    p = RS_FMA(p, r, 2.755731922398589e-07);
    p = RS_FMA(p, r, 2.7557319223985893e-06);
    p = RS_FMA(p, r, 2.48015873015873e-05);
    p = RS_FMA(p, r, 0.0001984126984126984);
    p = RS_FMA(p, r, 0.001388888888888889);
    p = RS_FMA(p, r, 0.008333333333333333);
    p = RS_FMA(p, r, 0.041666666666666664);
    p = RS_FMA(p, r, 0.16666666666666666);
    p = RS_FMA(p, r, 0.5);
    p = RS_FMA(p, r, 1.0);
    p = RS_FMA(p, r, 1.0);
    int k = (int)kd;
    int k1 = k >> 1;
    int k2 = k - k1;
    return (p * rs_pow2i(k1)) * rs_pow2i(k2);
}

RS_HD double rs_exp(double x) {
    if (x != x) return x;
    if (x > 709.782712893384) return rs_inf();
    if (x < -745.2) return 0.0;
    return rs_exp_core(x);
}

/* rs_exp for x <= 0 (not NaN) without control flow: identical values */
RS_HD double rs_exp_nonpos(double x) {
    const double v = rs_exp_core(x < -745.2 ? -745.2 : x);
    return x < -745.2 ? 0.0 : v;
}

/* log(x): x = 2^e m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1),
 * odd series to s^25 (|s| <= 0.1716, truncation 6e-19). */
RS_HD double rs_log(double x) {
    if (x != x) return x;
    if (x < 0.0) return rs_nan();
    if (x == 0.0) return -rs_inf();
    uint64_t u = rs_d2u(x);
    if (u == 0x7ff0000000000000ull) return x;
    int e = 0;
    if ((u >> 52) == 0) { /* subnormal */
        x = x * 18014398509481984.0; /* 2^54 */
        u = rs_d2u(x);
        e = -54;
    }
    e += (int)(u >> 52) - 1023;
    uint64_t mant = u & 0x000fffffffffffffull;
    if (mant > 0x6a09e667f3bccull) { /* m > sqrt(2) */
        e += 1;
        u = mant | 0x3fe0000000000000ull; /* m/2 */
    } else {
        u = mant | 0x3ff0000000000000ull;
    }
    double m = rs_u2d(u);
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double q = 0.04;
    q = RS_FMA(q, z, 0.043478260869565216);
    q = RS_FMA(q, z, 0.047619047619047616);
    q = RS_FMA(q, z, 0.05263157894736842);
    q = RS_FMA(q, z, 0.058823529411764705);
    q = RS_FMA(q, z, 0.06666666666666667);
    q = RS_FMA(q, z, 0.07692307692307693);
    q = RS_FMA(q, z, 0.09090909090909091);
    q = RS_FMA(q, z, 0.1111111111111111);
    q = RS_FMA(q, z, 0.14285714285714285);
    q = RS_FMA(q, z, 0.2);
    q = RS_FMA(q, z, 0.3333333333333333);
    q = q * z;
    double s2 = 2.0 * s;
    double lm = RS_FMA(s2, q, s2);
    double ed = (double)e;
    return RS_FMA(ed, RS_LN2_HI, lm + ed * RS_LN2_LO);
}

RS_HD double rs_log10(double x) { return rs_log(x) * RS_LOG10E; }

/* asin on |x| <= 0.5 given z = x*x: x + x z P(z), 28 Taylor terms (truncation 3e-20) */
RS_HD double rs_asin_core(double x, double z) {
    double p = 0.0018622264064031275;
    p = RS_FMA(p, z, 0.0019650336162772837);
    p = RS_FMA(p, z, 0.0020776610325181676);
    p = RS_FMA(p, z, 0.0022014739737101384);
    p = RS_FMA(p, z, 0.002338091892111975);
    p = RS_FMA(p, z, 0.0024894486782468836);
    p = RS_FMA(p, z, 0.00265787063820729);
    p = RS_FMA(p, z, 0.002846178401108942);
    p = RS_FMA(p, z, 0.0030578216492580306);
    p = RS_FMA(p, z, 0.003297059503473485);
    p = RS_FMA(p, z, 0.0035692053938259347);
    p = RS_FMA(p, z, 0.003880964558837669);
    p = RS_FMA(p, z, 0.004240907093679363);
    p = RS_FMA(p, z, 0.004660143486915096);
    p = RS_FMA(p, z, 0.005153309682319905);
    p = RS_FMA(p, z, 0.005740037670841924);
    p = RS_FMA(p, z, 0.006447210311889649);
    p = RS_FMA(p, z, 0.0073125258735988454);
    p = RS_FMA(p, z, 0.008390335809616815);
    p = RS_FMA(p, z, 0.009761609529194078);
    p = RS_FMA(p, z, 0.011551800896139705);
    p = RS_FMA(p, z, 0.01396484375);
    p = RS_FMA(p, z, 0.017352764423076924);
    p = RS_FMA(p, z, 0.022372159090909092);
    p = RS_FMA(p, z, 0.030381944444444444);
    p = RS_FMA(p, z, 0.044642857142857144);
    p = RS_FMA(p, z, 0.075);
    p = RS_FMA(p, z, 0.16666666666666666);
    return RS_FMA(x * z, p, x);
}

RS_HD double rs_acos(double x) {
    if (x != x) return x;
    if (x >= 1.0) return x > 1.0 ? rs_nan() : 0.0;
    if (x <= -1.0) return x < -1.0 ? rs_nan() : RS_PI;
    if (x > 0.5) {
        double z = (1.0 - x) * 0.5;
        double s = RS_SQRT(z);
        return 2.0 * rs_asin_core(s, z);
    }
    if (x < -0.5) {
        double z = (1.0 + x) * 0.5;
        double s = RS_SQRT(z);
        return RS_PI - 2.0 * rs_asin_core(s, z);
    }
    return RS_PI_2 - rs_asin_core(x, x * x);
}

/* logistic 1/(1+exp(-k (x - x0))) with the reference's operation order
 * (channel_models.py:35-37): y = 1 / (1 + exp((-k) * (x - x0))) */
/* RS_EXP_CALL / RS_LOG_CALL let a translation unit route the calls below through out-of-line copies
 * (the HIP kernels do: keeping the polynomial constants out of the caller's loop-long live ranges is
 * worth more registers than the call costs); the arithmetic is the same function either way. */
#ifndef RS_EXP_CALL
#define RS_EXP_CALL rs_exp
#endif
#ifndef RS_LOG_CALL
#define RS_LOG_CALL rs_log
#endif

RS_HD double rs_sigmoid(double x, double x0, double k) {
    return 1.0 / (1.0 + RS_EXP_CALL((-k) * (x - x0)));
}

/* channel_models.py:39-41: x = -(1/k) * log(1/y - 1) + x0 */
RS_HD double rs_inv_sigmoid(double y, double x0, double k) {
    return (-(1.0 / k)) * RS_LOG_CALL(1.0 / y - 1.0) + x0;
}

#endif /* RS_DETMATH_H */
