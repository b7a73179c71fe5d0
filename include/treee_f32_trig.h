/*
 * treee_f32_trig.h -- the two transcendental functions on the hot path, defined to the last bit.
 *
 * fast_eigenvalues (treee/src/calculations.rs:467-498) calls f32::acos once and f32::cos twice per point; everything
 * else in the curvature loop is + - * / sqrt, which IEEE 754 fixes.  Rust forwards acos / cos to the platform's libm,
 * HIP to OCML, and the two disagree in the last bit of a few per cent of the arguments -- which used to force a
 * tolerance on `curve`.  This header is ONE self-contained implementation, compiled unchanged for the host (gcc, the
 * oracle) and the device (hipcc, the kernels): evaluate in f64 with only + - * / sqrt and explicit constants, round to
 * f32 ONCE.  The f64 value is accurate to < 2 ulp(f64), so the f32 result is the correctly rounded one except when the
 * exact value lies within ~2^-52 (relative) of a rounding boundary (about 1 argument in 2^27) -- i.e. it equals what a
 * correctly rounding libm (glibc >= 2.41's CORE-MATH acosf / cosf) returns, and differs from any libm by at most 1 ulp.
 * tests/test_oracle.py compares it with the host libm over dense argument grids.
 *
 * Needs -ffp-contract=off on both sides (the library and the oracle are built that way): no fused multiply-add.
 * The approximations are the classical fdlibm ones (e_asin.c, k_sin.c, k_cos.c, Sun Microsystems 1993: "Permission to
 * use, copy, modify, and distribute this software is freely granted, provided that this notice is preserved") --
 * rational asin on [0, 0.5] with the 2 asin(sqrt((1 - x) / 2)) identity beyond, degree-13 / -14 minimax sin / cos on
 * [-pi/4, pi/4] with Cody-Waite reduction by pi/2.
 */
#ifndef TREEE_F32_TRIG_H
#define TREEE_F32_TRIG_H

#if defined(__HIPCC__) || defined(__HIP__)
#define TREEE_TRIG_FN __host__ __device__ inline
#define TREEE_TRIG_SQRT(x) __builtin_sqrt(x)
#else
#include <math.h>
#define TREEE_TRIG_FN static inline
#define TREEE_TRIG_SQRT(x) sqrt(x)
#endif

/* No fused multiply-add inside these functions, whatever the includer's flags: clang / hipcc -- where the
 * device always has an FMA to contract into -- switch contraction off per function body (the setting is recorded on every
 * floating-point instruction and survives inlining); other compilers are refused on a target that has FMA instructions
 * unless the build says it passes -ffp-contract=off (-DTREEE_TRIG_CONTRACT_OFF; oracle/Makefile does both). */
#if defined(__clang__)
#define TREEE_TRIG_NOFMA _Pragma("clang fp contract(off)")
#else
#define TREEE_TRIG_NOFMA
#if defined(__FMA__) && !defined(TREEE_TRIG_CONTRACT_OFF)
#error "treee_f32_trig.h: this target has FMA instructions -- compile with -ffp-contract=off and -DTREEE_TRIG_CONTRACT_OFF"
#endif
#endif

/* (asin(x) - x) / x as a function of z = x * x, |x| <= 0.5 */
TREEE_TRIG_FN double treee_trig_asin_r(double z) {
	TREEE_TRIG_NOFMA
	const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
	             pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
	             qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
	             qS4 = 7.70381505559019352791e-02;
	const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
	const double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
	return p / q;
}

/* acos for an f32 argument, rounded once.  |x| > 1 or NaN -> NaN (the caller clamps: calculations.rs:487-493). */
TREEE_TRIG_FN float treee_acos_f32(float xf) {
	TREEE_TRIG_NOFMA
	const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
	const double x = (double)xf;
	if (!(x >= -1.0 && x <= 1.0)) return (float)((x - x) / (x - x));
	if (x == 1.0) return 0.0f;
	if (x == -1.0) return (float)(pi + 2.0 * pio2_lo);
	const double ax = x < 0.0 ? -x : x;
	if (ax <= 0.5) { /* pi/2 - (x + x * z * R(z)) */
		const double z = x * x;
		return (float)(pio2_hi - (x - (pio2_lo - x * treee_trig_asin_r(z))));
	}
	const double z = (1.0 - ax) * 0.5, s = TREEE_TRIG_SQRT(z), r = treee_trig_asin_r(z);
	if (x < 0.0) return (float)(pi - 2.0 * (s + (r * s - pio2_lo))); /* pi - 2 asin(sqrt((1 + x) / 2)) */
	return (float)(2.0 * (s + r * s));                                  /* 2 asin(sqrt((1 - x) / 2)) */
}

TREEE_TRIG_FN double treee_trig_ksin(double x) { /* |x| <= pi/4 */
	TREEE_TRIG_NOFMA
	const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
	             S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
	const double z = x * x, v = z * x;
	const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
	return x + v * (S1 + z * r);
}
TREEE_TRIG_FN double treee_trig_kcos(double x) { /* |x| <= pi/4 */
	TREEE_TRIG_NOFMA
	const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
	             C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
	const double z = x * x;
	const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
	const double hz = 0.5 * z, w = 1.0 - hz;
	return w + (((1.0 - w) - hz) + z * r);
}

/* cos for an f32 argument, rounded once.  The path only ever passes [0, pi + 2^-21]; the reduction is exact enough for
 * correct rounding up to |x| = 1e5 (three-piece pi/2); beyond that, NaN and inf the result is NaN. */
TREEE_TRIG_FN float treee_cos_f32(float xf) {
	TREEE_TRIG_NOFMA
	const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, /* first 33 bits of pi/2 */
	             pio2_2 = 6.07710050630396597660e-11,                                        /* next 33 bits */
	             pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;  /* the rest */
	double x = (double)xf;
	if (!(x - x == 0.0)) return (float)(x - x); /* NaN, inf */
	if (x < 0.0) x = -x;
	if (x <= 0.78539816339744830962) return (float)treee_trig_kcos(x);
	if (x > 1.0e5) return (float)((x - x) / (x - x)); /* outside the contract (see above): NaN, on both sides */
	/* n = nearest integer to x * 2/pi (round half up: any consistent choice works), r = x - n * pi/2 in three exact-product steps */
	const double fn = (double)(long long)(x * invpio2 + 0.5);
	const double r = (((x - fn * pio2_1) - fn * pio2_2) - fn * pio2_3) - fn * pio2_3t;
	const long long n = (long long)fn;
	switch ((int)(n & 3)) {
	case 0: return (float)treee_trig_kcos(r);
	case 1: return (float)(-treee_trig_ksin(r));
	case 2: return (float)(-treee_trig_kcos(r));
	default: return (float)treee_trig_ksin(r);
	}
}

#endif
