/*
 * oracle/portable_math.h -- TEST INFRASTRUCTURE (part of the CPU oracle; never linked by the product).
 *
 * Bit-reproducible replacements for the libm calls on the reference hot path
 * (expf / sinf / cosf / acosf / atan2f: DepthMap.h:547 GetWeight, DepthMap.cpp:609-613 smoothness,
 * Util.inl:614-626 Normal2Dir/Dir2Normal, DepthMap.h:629-634 CorrectNormal).
 *
 * They are built ONLY from IEEE-754 double add/mul/div/sqrt/fma/floor, so the same operation sequence
 * gives the same bits on the host CPU and on gfx950 (the HIP side carries its own copy of this
 * sequence in hc-mvs_amd/csrc/pm_math.h).  Results are evaluated in double and rounded once to float:
 * they agree with glibc's correctly-rounded-in-practice float functions except for rare 1-ulp cases
 * (checked in tests/test_oracle_math.py).  Used when arith_mode == HCOR_ARITH_DEVICE; the
 * reference-faithful mode (HCOR_ARITH_REFERENCE) calls libm exactly like the reference does.
 */
#ifndef HCOR_PORTABLE_MATH_H
#define HCOR_PORTABLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define PM_PI      0x1.921fb54442d18p+1
#define PM_PIO2    0x1.921fb54442d18p+0
#define PM_PIO2_HI 0x1.921fb50000000p+0
#define PM_PIO2_LO 0x1.110b4611a6263p-26
#define PM_2OPI    0x1.45f306dc9c883p-1 /* 2/pi */
#define PM_LN2_HI  0x1.62e42f8000000p-1
#define PM_LN2_LO  0x1.be8e7bcd5e4f2p-27
#define PM_INVLN2  0x1.71547652b82fep+0

static inline double pm_pow2i(int k) { /* 2^k, -1022 <= k <= 1023 */
	uint64_t b = (uint64_t)(k + 1023) << 52;
	double d;
	memcpy(&d, &b, sizeof d);
	return d;
}

/* e^x rounded to float; x float */
static inline float pm_expf(float x) {
	if (!(x > -104.0f)) return (x != x) ? x : 0.0f;
	if (x > 88.75f) return INFINITY;
	const double xd = (double)x;
	const double kd = floor(fma(xd, PM_INVLN2, 0.5));
	double r = fma(-kd, PM_LN2_HI, xd);
	r = fma(-kd, PM_LN2_LO, r); /* |r| <= 0.3466 */
	/* Taylor, degree 11 (remainder < 1e-15 relative) */
	double p = 1.0 / 39916800.0;
	p = fma(p, r, 1.0 / 3628800.0);
	p = fma(p, r, 1.0 / 362880.0);
	p = fma(p, r, 1.0 / 40320.0);
	p = fma(p, r, 1.0 / 5040.0);
	p = fma(p, r, 1.0 / 720.0);
	p = fma(p, r, 1.0 / 120.0);
	p = fma(p, r, 1.0 / 24.0);
	p = fma(p, r, 1.0 / 6.0);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	return (float)(p * pm_pow2i((int)kd));
}

/* sin and cos of a float angle (|x| well below 2^20), each rounded to float */
static inline void pm_sincosf(float x, float* s, float* c) {
	const double xd = (double)x;
	const double kd = floor(fma(xd, PM_2OPI, 0.5));
	double r = fma(-kd, PM_PIO2_HI, xd);
	r = fma(-kd, PM_PIO2_LO, r); /* |r| <= pi/4 */
	const double r2 = r * r;
	/* sin(r) = r + r^3 * S(r^2), Taylor through r^15 */
	double ps = -1.0 / 1307674368000.0;
	ps = fma(ps, r2, 1.0 / 6227020800.0);
	ps = fma(ps, r2, -1.0 / 39916800.0);
	ps = fma(ps, r2, 1.0 / 362880.0);
	ps = fma(ps, r2, -1.0 / 5040.0);
	ps = fma(ps, r2, 1.0 / 120.0);
	ps = fma(ps, r2, -1.0 / 6.0);
	const double sr = fma(ps * r2, r, r);
	/* cos(r) = 1 + r^2 * C(r^2), Taylor through r^16 */
	double pc = 1.0 / 20922789888000.0;
	pc = fma(pc, r2, -1.0 / 87178291200.0);
	pc = fma(pc, r2, 1.0 / 479001600.0);
	pc = fma(pc, r2, -1.0 / 3628800.0);
	pc = fma(pc, r2, 1.0 / 40320.0);
	pc = fma(pc, r2, -1.0 / 720.0);
	pc = fma(pc, r2, 1.0 / 24.0);
	pc = fma(pc, r2, -0.5);
	const double cr = fma(pc, r2, 1.0);
	const int q = (int)kd & 3;
	double sv, cv;
	if (q == 0) { sv = sr; cv = cr; }
	else if (q == 1) { sv = cr; cv = -sr; }
	else if (q == 2) { sv = -sr; cv = -cr; }
	else { sv = -cr; cv = sr; }
	*s = (float)sv;
	*c = (float)cv;
}
static inline float pm_sinf(float x) { float s, c; pm_sincosf(x, &s, &c); return s; }
static inline float pm_cosf(float x) { float s, c; pm_sincosf(x, &s, &c); return c; }

/* atan(t) for t >= 0 (double in, double out, ~1e-16) */
static inline double pm_atan_pos(double t) {
	int inv = 0;
	if (t > 1.0) { t = 1.0 / t; inv = 1; }
	/* nearest of c = i/8, i = 0..8 ; atan(t) = atan(c) + atan((t-c)/(1+t*c)) */
	const int i = (int)floor(fma(t, 8.0, 0.5));
	static const double ATAN_C[9] = {
		0.0,
		0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3, 0x1.6f61941e4def1p-2, 0x1.dac670561bb4fp-2,
		0x1.1e00babdefeb4p-1, 0x1.4978fa3269ee1p-1, 0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1};
	const double c = (double)i * 0.125;
	const double u = (t - c) / fma(t, c, 1.0); /* |u| <= 1/16 */
	const double u2 = u * u;
	double p = 1.0 / 15.0;
	p = fma(p, u2, -1.0 / 13.0);
	p = fma(p, u2, 1.0 / 11.0);
	p = fma(p, u2, -1.0 / 9.0);
	p = fma(p, u2, 1.0 / 7.0);
	p = fma(p, u2, -1.0 / 5.0);
	p = fma(p, u2, 1.0 / 3.0);
	p = -p; /* atan(u) = u - u^3/3 + u^5/5 ... = u + u^3 * (-(1/3 - u^2/5 + ...)) */
	const double au = fma(p * u2, u, u);
	const double a = ATAN_C[i] + au;
	return inv ? (PM_PIO2 - a) : a;
}

/* atan2(y, x) rounded to float */
static inline float pm_atan2f(float y, float x) {
	const double yd = (double)y, xd = (double)x;
	if (x != x || y != y) return x + y;
	if (yd == 0.0) {
		if (xd > 0.0 || (xd == 0.0 && !signbit(x))) return y;               /* +-0 */
		return signbit(y) ? (float)-PM_PI : (float)PM_PI;
	}
	if (xd == 0.0) return yd > 0.0 ? (float)PM_PIO2 : (float)-PM_PIO2;
	const double ay = fabs(yd), ax = fabs(xd);
	double a = pm_atan_pos(ay / ax);
	if (xd < 0.0) a = PM_PI - a;
	return (float)(yd < 0.0 ? -a : a);
}

/* acos(x) rounded to float; |x| >= 1 clamps to the end points (libm would return NaN beyond them) */
static inline float pm_acosf(float x) {
	if (x != x) return x;
	if (x >= 1.0f) return 0.0f;
	if (x <= -1.0f) return (float)PM_PI;
	const double xd = (double)x;
	/* acos(x) = 2 atan( sqrt((1-x)/(1+x)) ) ; 1-x and 1+x are exact in double */
	const double t = sqrt((1.0 - xd) / (1.0 + xd));
	return (float)(2.0 * pm_atan_pos(t));
}

#endif
