/*
 * hc-mvs_amd/csrc/pm_math.h -- device transcendental functions used on the PatchMatch path
 * (expf / sincosf / acosf / atan2f: DepthMap.h:547 GetWeight, DepthMap.cpp:609-613 smoothness,
 * Util.inl:614-626 Normal2Dir/Dir2Normal, DepthMap.h:629-634 CorrectNormal in the reference).
 *
 * Built only from IEEE-754 binary32 add/mul/div/sqrt/fmaf/floorf: no dependence on the device math
 * library, ~2 ulp from libm over the ranges used, and reproducible bit for bit on a host CPU (the test
 * oracle carries an independent copy of the same operation sequence).  Compile with -ffp-contract=off:
 * only the explicit fmaf() calls fuse.
 */
#ifndef HCMVS_PM_MATH_H
#define HCMVS_PM_MATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_PI_F      0x1.921fb6p+1f
#define PM_PI_LO_F   (-0x1.777a5cp-24f)  /* pi - (float)pi */
#define PM_PIO2_F    0x1.921fb6p+0f
#define PM_PIO2_LO_F (-0x1.777a5cp-25f)
#define PM_PIO4_F    0x1.921fb6p-1f
#define PM_2OPI_F    0x1.45f306p-1f
#define PM_PIO2_HI_F 0x1.92p+0f         /* 3-part Cody-Waite split of pi/2 */
#define PM_PIO2_MI_F 0x1.fb4p-12f
#define PM_PIO2_LO3_F 0x1.4442d2p-24f
#define PM_LN2_HI_F  0x1.62ep-1f
#define PM_LN2_LO_F  0x1.0bfbe8p-15f
#define PM_INVLN2_F  0x1.715476p+0f
#define PM_TANPIO8_F 0x1.a8279ap-2f

static __device__ __forceinline__ float pm_bits2f(uint32_t b) { return __uint_as_float(b); }

/* e^x, x float.  Results below 2^-126 flush to 0 (inputs below -87).  Straight-line: out-of-range inputs are
 * clamped for the evaluation and fixed up by selects at the end. */
static __device__ __forceinline__ float pm_expf(float x) {
	const float xc = fminf(fmaxf(x, -87.0f), 88.0f);
	const float kf = floorf(fmaf(xc, PM_INVLN2_F, 0.5f));
	float r = fmaf(-kf, PM_LN2_HI_F, xc);
	r = fmaf(-kf, PM_LN2_LO_F, r); /* |r| <= 0.3466 */
	float p = 1.0f / 5040.0f;
	p = fmaf(p, r, 1.0f / 720.0f);
	p = fmaf(p, r, 1.0f / 120.0f);
	p = fmaf(p, r, 1.0f / 24.0f);
	p = fmaf(p, r, 1.0f / 6.0f);
	p = fmaf(p, r, 0.5f);
	p = fmaf(p, r, 1.0f);
	p = fmaf(p, r, 1.0f);
	float e = p * pm_bits2f((uint32_t)((int)kf + 127) << 23);
	e = x > -87.0f ? e : 0.0f;
	e = x > 88.0f ? __builtin_huge_valf() : e;
	return x != x ? x : e;
}

/* sin and cos of a float angle, |x| < ~100 */
static __device__ __forceinline__ void pm_sincosf(float x, float* s, float* c) {
	const float kf = floorf(fmaf(x, PM_2OPI_F, 0.5f));
	float r = fmaf(-kf, PM_PIO2_HI_F, x);
	r = fmaf(-kf, PM_PIO2_MI_F, r);
	r = fmaf(-kf, PM_PIO2_LO3_F, r); /* |r| <= pi/4 */
	const float r2 = r * r;
	float ps = 1.0f / 362880.0f;
	ps = fmaf(ps, r2, -1.0f / 5040.0f);
	ps = fmaf(ps, r2, 1.0f / 120.0f);
	ps = fmaf(ps, r2, -1.0f / 6.0f);
	const float sr = fmaf(ps * r2, r, r);
	float pc = -1.0f / 3628800.0f;
	pc = fmaf(pc, r2, 1.0f / 40320.0f);
	pc = fmaf(pc, r2, -1.0f / 720.0f);
	pc = fmaf(pc, r2, 1.0f / 24.0f);
	pc = fmaf(pc, r2, -0.5f);
	const float cr = fmaf(pc, r2, 1.0f);
	const int q = (int)kf & 3;
	float sv, cv;
	if (q == 0) { sv = sr; cv = cr; }
	else if (q == 1) { sv = cr; cv = -sr; }
	else if (q == 2) { sv = -sr; cv = -cr; }
	else { sv = -cr; cv = sr; }
	*s = sv;
	*c = cv;
}
static __device__ __forceinline__ float pm_sinf(float x) { float s, c; pm_sincosf(x, &s, &c); return s; }
static __device__ __forceinline__ float pm_cosf(float x) { float s, c; pm_sincosf(x, &s, &c); return c; }

/* asin(sqrt(z))/sqrt(z) = 1 + z*R(z) on z in [0, 0.25] */
static __device__ __forceinline__ float pm_asin_r(float z) {
	float p = 0x1.14efcap-5f;
	p = fmaf(p, z, 0x1.17cdbcp-6f);
	p = fmaf(p, z, 0x1.fdce9ep-6f);
	p = fmaf(p, z, 0x1.6d58dap-5f);
	p = fmaf(p, z, 0x1.33343cp-4f);
	p = fmaf(p, z, 0x1.555554p-3f);
	return p;
}
/* acos(x); |x| >= 1 clamps to the end points (libm returns NaN beyond them).  Straight-line: one polynomial
 * evaluation serves both the |x| <= 0.5 and the |x| > 0.5 form. */
static __device__ __forceinline__ float pm_acosf(float x) {
	const float ax = fabsf(x);
	const int small = ax <= 0.5f;
	const float z = small ? x * x : (1.0f - ax) * 0.5f;
	const float s = small ? x : sqrtf(z);
	const float t = fmaf(s * z, pm_asin_r(z), s); /* asin(x) resp. asin(sqrt(z)) = acos(|x|)/2 */
	const float a_small = (PM_PIO2_F - t) + PM_PIO2_LO_F;
	const float a_big = x > 0.0f ? 2.0f * t : (PM_PI_F - 2.0f * t) + PM_PI_LO_F;
	float a = small ? a_small : a_big;
	a = ax >= 1.0f ? (x > 0.0f ? 0.0f : PM_PI_F) : a;
	return x != x ? x : a;
}

/* atan2(y, x) */
static __device__ __forceinline__ float pm_atan2f(float y, float x) {
	if (x != x || y != y) return x + y;
	const float ax = fabsf(x), ay = fabsf(y);
	if (ay == 0.0f) {
		if (x > 0.0f || (x == 0.0f && !__builtin_signbit(x))) return y; /* +-0 */
		return __builtin_signbit(y) ? -PM_PI_F : PM_PI_F;
	}
	if (ax == 0.0f) return y > 0.0f ? PM_PIO2_F : -PM_PIO2_F;
	const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
	const float t = mn / mx; /* [0,1] */
	float u = t, base = 0.0f;
	if (t > PM_TANPIO8_F) { u = (t - 1.0f) / (t + 1.0f); base = PM_PIO4_F; }
	const float w = u * u;
	float p = -0x1.089378p-4f;
	p = fmaf(p, w, 0x1.b82b1cp-4f);
	p = fmaf(p, w, -0x1.2421a8p-3f);
	p = fmaf(p, w, 0x1.99973ep-3f);
	p = fmaf(p, w, -0x1.555554p-2f);
	float a = fmaf(u * w, p, u) + base; /* atan(mn/mx) in [0, pi/4] */
	if (ay > ax) a = PM_PIO2_F - a;
	if (x < 0.0f) a = PM_PI_F - a;
	return y < 0.0f ? -a : a;
}

#endif
