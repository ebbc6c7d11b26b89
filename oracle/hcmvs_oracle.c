/*
 * oracle/hcmvs_oracle.c -- TEST INFRASTRUCTURE (see hcmvs_oracle.h: PARITY UNPINNED).
 *
 * CPU restatement of the PatchMatch depth-map estimation path of HC-MVS:
 *   DM.cpp:354-381 (visiting order), 386-439 (constants), 442-519 (patch + bilateral weights),
 *   522-616 + 890-893 (ScorePixelImage), 987-1046 (ScorePixel), 1050-1501 (ProcessPixel),
 *   1671-1738 (InterpolatePixel, InitPlane); DM.h:537-548, 565-574, 618-634;
 *   SD.cpp:581-595 (gradient map), 649-744 (passes), 758-1056 (driver), 783-808 (splat init).
 *
 * Two arithmetic modes compute the same algorithm:
 *   HCOR_ARITH_REFERENCE follows the reference operation by operation (libm, sequential tap sums,
 *     incremental warp, H = (Hl + Hm n^T/(n.X0 d)) Hr in double);
 *   HCOR_ARITH_DEVICE evaluates the same quantities in the association the gfx950 kernels use
 *     (per-segment partial sums + xor butterfly, direct per-tap warp with fmaf, one IEEE reciprocal
 *     per tap, H = Hl Hr + Hm (Hr^T n)^T/(n.X0 d) in float, portable_math.h transcendental functions) so that
 *     the GPU result can be compared BIT FOR BIT.
 *
 * Deliberate, documented departures from the reference (it is nondeterministic / undefined there):
 *   D1 RNG: counter-based hash keyed by (seed, pixel, pass, draw) instead of a per-thread mt19937
 *      seeded from std::random_device (DM.cpp:395-397, Random.h:105).
 *   D2 the smoothness plane is initialised from the pixel's current estimate at ProcessPixel entry;
 *      the reference leaves whatever plane the thread's previous pixel set (DM.cpp:1450-1463).
 *   D3 CorrectNormal with a degenerate rotation axis (normal parallel to the ray) leaves the normal
 *      unchanged instead of producing NaN (Rotation.inl:707-733).
 */
#include "hcmvs_oracle.h"
#include "portable_math.h"

#include <float.h>
#include <math.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SQ(x) ((x) * (x))

/* ------------------------------------------------------------------------------------------------ */
/* small utilities                                                                                  */

void hcor_default_params(hcor_params* p) {
	memset(p, 0, sizeof *p);
	p->adapthalfwin = 5;          /* DPC.cpp:163 */
	p->n_estimation_iters = 1;    /* DPC.cpp */
	p->it_external = 0;
	p->n_external_iters = 1;
	p->propagate_halfwin = 1;
	p->propagate_step = 4;
	p->n_random_iters = 6;        /* DM.cpp:120 */
	p->ncc_threshold_keep = 0.55f;/* DM.cpp:117 */
	p->random_depth_ratio = 0.003f;
	p->random_angle1_deg = 16.f;
	p->random_angle2_deg = 10.f;
	p->random_smooth_depth = 0.02f;
	p->random_smooth_normal_deg = 13.f;
	p->random_smooth_bonus = 0.93f;
	p->photometric_flow = 0.f;
	p->seed = 1234;
	p->arith_mode = HCOR_ARITH_REFERENCE;
	p->order = HCOR_ORDER_ZIGZAG;
	p->n_threads = 1;
	p->median_blur = 1;
}

static inline uint32_t fmix32(uint32_t h) {
	h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
	return h;
}
uint32_t hcor_rand_u32(uint32_t seed, uint32_t pix, uint32_t stream, uint32_t ctr) {
	uint32_t h = fmix32(seed ^ 0x9e3779b9u);
	h = fmix32(h ^ (pix * 0x9e3779b1u));
	h = fmix32(h ^ (stream * 0x85ebca77u));
	h = fmix32(h ^ (ctr * 0xc2b2ae3du));
	return h;
}
/* Random.h:113-115: (float)random()/(float)max() */
static inline float rand_unit(uint32_t r) { return (float)r / (float)4294967295u; }

#define FPI_F ((float)3.1415926535897932384626433832795)
static inline float fd2r(float d) { return d * (FPI_F / 180.f); } /* Types.h:566 */

typedef struct {
	float (*expf_)(float);
	float (*sinf_)(float);
	float (*cosf_)(float);
	float (*acosf_)(float);
	float (*atan2f_)(float, float);
} mathtab;
static float w_pm_expf(float x) { return pm_expf(x); }
static float w_pm_sinf(float x) { return pm_sinf(x); }
static float w_pm_cosf(float x) { return pm_cosf(x); }
static float w_pm_acosf(float x) { return pm_acosf(x); }
static float w_pm_atan2f(float y, float x) { return pm_atan2f(y, x); }
static const mathtab MT_LIBM = {expf, sinf, cosf, acosf, atan2f};
static const mathtab MT_PM = {w_pm_expf, w_pm_sinf, w_pm_cosf, w_pm_acosf, w_pm_atan2f};
static inline const mathtab* mt_of(int mode) { return mode == HCOR_ARITH_DEVICE ? &MT_PM : &MT_LIBM; }

float hcor_pm_expf(float x) { return pm_expf(x); }
float hcor_pm_sinf(float x) { return pm_sinf(x); }
float hcor_pm_cosf(float x) { return pm_cosf(x); }
float hcor_pm_acosf(float x) { return pm_acosf(x); }
float hcor_pm_atan2f(float y, float x) { return pm_atan2f(y, x); }

static void mat3_mul(const double* a, const double* b, double* c) { /* cv::Matx product, k ascending */
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double s = 0;
			for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
			c[i * 3 + j] = s;
		}
}
static void mat3_mul_bt(const double* a, const double* b, double* c) { /* a * b^T */
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double s = 0;
			for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[j * 3 + k];
			c[i * 3 + j] = s;
		}
}
static void mat3_inv(const double* m, double* r) { /* adjugate / determinant (cv::Matx33 inv) */
	const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
	                 m[2] * (m[3] * m[7] - m[4] * m[6]);
	const double id = 1.0 / d;
	r[0] = (m[4] * m[8] - m[5] * m[7]) * id;
	r[1] = (m[2] * m[7] - m[1] * m[8]) * id;
	r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
	r[3] = (m[5] * m[6] - m[3] * m[8]) * id;
	r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
	r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
	r[6] = (m[3] * m[7] - m[4] * m[6]) * id;
	r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
	r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* ------------------------------------------------------------------------------------------------ */
/* DM.cpp:354-381 MapMatrix2ZigzagIdx                                                               */

int hcor_zigzag_coords(int w, int hgt, int raw_stride, uint16_t* coords) {
	const int w1 = w - 1;
	int n = 0;
	for (int dy = 0, h = raw_stride; dy < hgt; dy += h) {
		if (h * 2 > hgt - dy) h = hgt - dy;
		int lastX = 0;
		int xx = 0, xy = 0;
		for (int i = 0, ei = w * h; i < ei; ++i) {
			coords[2 * n] = (uint16_t)xx;
			coords[2 * n + 1] = (uint16_t)(xy + dy);
			++n;
			const int wasZero = (xx == 0);
			--xx;
			if (wasZero || ++xy == h) {
				if (++lastX < w) {
					xx = lastX;
					xy = 0;
				} else {
					xx = w1;
					xy = lastX - w1;
				}
			}
		}
	}
	return n;
}

/* ------------------------------------------------------------------------------------------------ */
/* OpenCV pieces restated from their published algorithms (OpenCV 4.2 is what the reference's
 * Dockerfile installs; it is not present here -> parity unpinned for these)                         */

void hcor_bgr2gray_u8(const uint8_t* bgr, int w, int h, uint8_t* gray) {
	/* cv::cvtColor BGR2GRAY 8u: (B*1868 + G*9617 + R*4899 + (1<<13)) >> 14 */
	for (long i = 0; i < (long)w * h; ++i)
		gray[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}
void hcor_gray_f32_to_u8(const float* g, int w, int h, uint8_t* out) {
	for (long i = 0; i < (long)w * h; ++i) {
		float v = g[i] * 255.f;
		int r = (int)lrintf(v);
		out[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
	}
}
static inline int reflect101(int i, int n) {
	if (n == 1) return 0;
	while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
	return i;
}
void hcor_gradient_map(const uint8_t* g, int w, int h, uint8_t* gra) {
	/* SD.cpp:586-595: Sobel 3x3 -> CV_16S, convertScaleAbs, addWeighted(0.5, 0.5) */
	for (int y = 0; y < h; ++y) {
		const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
		for (int x = 0; x < w; ++x) {
			const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
			const int a00 = g[ym * w + xm], a01 = g[ym * w + x], a02 = g[ym * w + xp];
			const int a10 = g[y * w + xm], a12 = g[y * w + xp];
			const int a20 = g[yp * w + xm], a21 = g[yp * w + x], a22 = g[yp * w + xp];
			const int gx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
			const int gy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
			int ax = abs(gx), ay = abs(gy);
			if (ax > 255) ax = 255;
			if (ay > 255) ay = 255;
			/* saturate_cast<uchar>(cvRound(ax*0.5 + ay*0.5)), cvRound = round half to even */
			const double s = ax * 0.5 + ay * 0.5;
			int r = (int)lrint(s);
			gra[y * w + x] = (uint8_t)(r > 255 ? 255 : r);
		}
	}
}
void hcor_median3(const float* in, int w, int h, float* out) {
	for (int y = 0; y < h; ++y)
		for (int x = 0; x < w; ++x) {
			float v[9];
			int n = 0;
			for (int dy = -1; dy <= 1; ++dy) {
				int yy = y + dy; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
				for (int dx = -1; dx <= 1; ++dx) {
					int xx = x + dx; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
					v[n++] = in[yy * w + xx];
				}
			}
			/* insertion sort of 9, take the middle */
			for (int i = 1; i < 9; ++i) {
				float t = v[i]; int j = i - 1;
				while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
				v[j + 1] = t;
			}
			out[y * w + x] = v[4];
		}
}

/* cv::resize(src, dst, Size(), scale, scale, scale > 1 ? INTER_CUBIC : INTER_AREA) on an f32 image, as
 * DepthData::ViewData::ScaleImage calls it (DM.h:233-238).  OpenCV 4.2 imgproc/src/resize.cpp restated:
 *   dsize = (cvRound(w*scale), cvRound(h*scale)); scale_x = scale_y = 1/scale (double);
 *   INTER_AREA, integer factor -> ResizeAreaFast: block sum * (1/area);
 *   INTER_AREA otherwise -> computeResizeAreaTab (per axis: leading partial cell if sx1 - fsx1 > 1e-3, whole cells with
 *     alpha = 1/cellWidth, trailing partial cell if fsx2 - sx2 > 1e-3) and ResizeArea_Invoker: per source row the x table
 *     accumulates into buf from 0, rows combine as sum = beta*buf for the first and sum += beta*buf for the rest;
 *   INTER_CUBIC -> fx = (dx+0.5)*scale_x - 0.5, Keys kernel A = -0.75 (interpolateCubic), taps clamped to the image,
 *     horizontal sums first, then the vertical one, left to right. */
void hcor_resize_size(int w, int h, float scale, int* dw, int* dh) {
	*dw = (int)lrint((double)w * (double)scale);
	*dh = (int)lrint((double)h * (double)scale);
}
typedef struct { int idx[8]; float alpha[8]; int n; } area_tab; /* one destination cell; factor < 6 -> at most 8 entries */
static void area_cell(int d, double scale, int ssize, int* idx, float* alpha, int* n, int cap) {
	const double fsx1 = d * scale, fsx2 = fsx1 + scale;
	const double cellWidth = fmin(scale, ssize - fsx1);
	int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2), k = 0;
	if (sx2 > ssize - 1) sx2 = ssize - 1;
	if (sx1 > sx2) sx1 = sx2;
	if (sx1 - fsx1 > 1e-3 && k < cap) { idx[k] = sx1 - 1; alpha[k++] = (float)((sx1 - fsx1) / cellWidth); }
	for (int sx = sx1; sx < sx2 && k < cap; ++sx) { idx[k] = sx; alpha[k++] = (float)(1.0 / cellWidth); }
	if (fsx2 - sx2 > 1e-3 && k < cap) { idx[k] = sx2; alpha[k++] = (float)(fmin(fmin(fsx2 - sx2, 1.), cellWidth) / cellWidth); }
	*n = k;
}
static void cubic_w(float x, float* c) {
	const float A = -0.75f;
	c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
	c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
	c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
	c[3] = 1.f - c[0] - c[1] - c[2];
}
/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_AREA) when the destination is LARGER than the source, as the fork's `restore`
 * variant up-samples the previous level's depth and normal maps (restore/libs/MVS/SceneDensify.cpp:523-524).  OpenCV 4.2
 * (imgproc/src/resize.cpp, cv::resize): INTER_AREA with scale < 1 runs the bilinear kernel with "area mode" coefficients --
 *   sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) / scale, fx = fx <= 0 ? 0 : fx - floor(fx)
 * (a destination pixel that lies inside one source pixel copies it; one that straddles two mixes them by the overlap), the last
 * source column / row has no right / lower partner; horizontal pass first (HResizeLinear: S[sx] * (1 - fx) + S[sx + 1] * fx), then
 * the vertical one (VResizeLinear: R0 * (1 - fy) + R1 * fy).  Multiplies and adds are not fused (the C source's expression).
 * ch interleaved channels.  OpenCV itself is absent: restated from its published source (parity unpinned). */
static void area_up_coef(int d, int ssize, int dsize, int columns, int* s0, float* f) {
	const double scale = (double)ssize / (double)dsize, inv_scale = (double)dsize / (double)ssize;
	int sx = (int)floor(d * scale);
	float fx = (float)((d + 1) - (sx + 1) * inv_scale);
	fx = fx <= 0 ? 0.f : fx - floorf(fx);
	if (columns) { /* the x table resets the weight at the borders; the y table does not, its rows are clipped where they are used */
		if (sx < 0) { fx = 0; sx = 0; }
		if (sx >= ssize - 1) { fx = 0; sx = ssize - 1; }
	}
	*s0 = sx; *f = fx;
}
void hcor_resize_area_up(const float* src, int sw, int sh, int ch, float* dst, int dw, int dh) {
	for (int y = 0; y < dh; ++y) {
		int sy; float fy;
		area_up_coef(y, sh, dh, 0, &sy, &fy);
		const int sy1 = sy + 1 < sh ? (sy + 1 > 0 ? sy + 1 : 0) : sh - 1;
		sy = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
		const float b0 = 1.f - fy, b1 = fy;
		for (int x = 0; x < dw; ++x) {
			int sx; float fx;
			area_up_coef(x, sw, dw, 1, &sx, &fx);
			const int last = sx + 1 >= sw; /* dx >= xmax: the left sample alone, times ONE */
			const float a0 = 1.f - fx, a1 = fx;
			for (int c = 0; c < ch; ++c) {
				const float* r0 = src + ((size_t)sy * sw) * ch + c;
				const float* r1 = src + ((size_t)sy1 * sw) * ch + c;
				const float h0 = last ? r0[(size_t)sx * ch] : r0[(size_t)sx * ch] * a0 + r0[(size_t)(sx + 1) * ch] * a1;
				const float h1 = last ? r1[(size_t)sx * ch] : r1[(size_t)sx * ch] * a0 + r1[(size_t)(sx + 1) * ch] * a1;
				dst[((size_t)y * dw + x) * ch + c] = h0 * b0 + h1 * b1;
			}
		}
	}
}
void hcor_resize_gray(const float* src, int sw, int sh, float scale, float* dst, int dw, int dh) {
	const double sc = 1.0 / (double)scale;
	if (scale > 1.f) {
		for (int y = 0; y < dh; ++y) {
			float fy = (float)((y + 0.5) * sc - 0.5);
			const int sy = (int)floorf(fy);
			fy -= sy;
			float cy[4]; cubic_w(fy, cy);
			for (int x = 0; x < dw; ++x) {
				float fx = (float)((x + 0.5) * sc - 0.5);
				const int sx = (int)floorf(fx);
				fx -= sx;
				float cx[4]; cubic_w(fx, cx);
				float rows[4];
				for (int j = 0; j < 4; ++j) {
					int yy = sy - 1 + j; if (yy < 0) yy = 0; if (yy > sh - 1) yy = sh - 1;
					float r = 0;
					for (int i = 0; i < 4; ++i) {
						int xx = sx - 1 + i; if (xx < 0) xx = 0; if (xx > sw - 1) xx = sw - 1;
						const float t = src[(size_t)yy * sw + xx] * cx[i];
						r = i ? r + t : t;
					}
					rows[j] = r;
				}
				dst[(size_t)y * dw + x] = ((rows[0] * cy[0] + rows[1] * cy[1]) + rows[2] * cy[2]) + rows[3] * cy[3];
			}
		}
		return;
	}
	const int isc = (int)sc;
	if ((double)isc == sc && isc >= 1 && dw * isc <= sw && dh * isc <= sh) {
		const float inv = 1.f / (float)(isc * isc);
		for (int y = 0; y < dh; ++y)
			for (int x = 0; x < dw; ++x) {
				float sum = 0;
				for (int j = 0; j < isc; ++j)
					for (int i = 0; i < isc; ++i) sum += src[(size_t)(y * isc + j) * sw + x * isc + i];
				dst[(size_t)y * dw + x] = sum * inv;
			}
		return;
	}
	for (int y = 0; y < dh; ++y) {
		int yi[16], ny; float yb[16];
		area_cell(y, sc, sh, yi, yb, &ny, 16);
		for (int x = 0; x < dw; ++x) {
			int xi[16], nx; float xa[16];
			area_cell(x, sc, sw, xi, xa, &nx, 16);
			float sum = 0;
			for (int j = 0; j < ny; ++j) {
				float buf = 0;
				for (int i = 0; i < nx; ++i) buf += src[(size_t)yi[j] * sw + xi[i]] * xa[i];
				if (j == 0) sum = yb[j] * buf; else sum += yb[j] * buf;
			}
			dst[(size_t)y * dw + x] = sum;
		}
	}
}

void hcor_splat_init(const hcor_view* ref, const float* pts, int n, float* depth, float* normal,
                     float* d_min, float* d_max) {
	/* SD.cpp:783-808 */
	const int W = ref->width, H = ref->height;
	memset(depth, 0, sizeof(float) * (size_t)W * H);
	float dmin = FLT_MAX, dmax = 0;
	for (int i = 0; i < n; ++i) {
		const double X[3] = {pts[3 * i] - ref->C[0], pts[3 * i + 1] - ref->C[1], pts[3 * i + 2] - ref->C[2]};
		double c[3];
		for (int r = 0; r < 3; ++r) c[r] = ref->R[r * 3] * X[0] + ref->R[r * 3 + 1] * X[1] + ref->R[r * 3 + 2] * X[2];
		const double px = ref->K[2] + ref->K[0] * (c[0] / c[2]);
		const double py = ref->K[5] + ref->K[4] * (c[1] / c[2]);
		const int x = (int)floor(px + .5), y = (int)floor(py + .5);
		const float d = (float)c[2];
		const int sx = x - 2 > 0 ? x - 2 : 0, sy = y - 2 > 0 ? y - 2 : 0;
		const int ex = x + 2 < W - 1 ? x + 2 : W - 1, ey = y + 2 < H - 1 ? y + 2 : H - 1;
		for (int yy = sy; yy <= ey; ++yy)
			for (int xx = sx; xx <= ex; ++xx) {
				depth[yy * W + xx] = d;
				normal[3 * (yy * W + xx)] = normal[3 * (yy * W + xx) + 1] = normal[3 * (yy * W + xx) + 2] = 0;
			}
		if (dmin > d) dmin = d;
		if (dmax < d) dmax = d;
	}
	*d_min = dmin * 0.9f;
	*d_max = dmax * 1.1f;
}

/* ------------------------------------------------------------------------------------------------ */
/* estimator context (DM.cpp:386-439 constructor constants)                                         */

typedef struct {
	const hcor_view* ref;
	const hcor_view* srcs;
	int V;
	const uint8_t* gra;
	hcor_params p;
	const mathtab* mt;
	double Hl[HCOR_MAX_VIEWS][9], Hm[HCOR_MAX_VIEWS][3], Hr[9]; /* DM.h:412-444 */
	double A[HCOR_MAX_VIEWS][9];                                 /* Hl*Hr (device association) */
	float Af[HCOR_MAX_VIEWS][9], Hmf[HCOR_MAX_VIEWS][3], Hrf[9]; /* ... rounded to float: what the kernels hold */
	double ifx, ify;                                             /* device association multiplies by 1/f */
	int S;                                                       /* device: segments per view */
	float dMin, dMax, dMinSqr, dMaxSqr;
	float smoothBonusDepth, smoothBonusNormal, smoothSigmaDepth, smoothSigmaNormal;
	float angle1Range, angle2Range;
	float thConfSmall, thConfBig, thConfRand, thRobust;
	uint64_t evals;
} est_ctx;

static int device_segments(int V) {
	(void)V;
	return 8; /* the kernels run one lane layout for every view count: 8 tap segments per view (9..16 views: two sets of eight
	           * view groups; the per-view arithmetic does not depend on which lane group evaluates it) */
}

static void ctx_init(est_ctx* c, const hcor_view* ref, const hcor_view* srcs, int V, const uint8_t* gra,
                     const hcor_params* p, float dMin, float dMax) {
	memset(c, 0, sizeof *c);
	c->ref = ref; c->srcs = srcs; c->V = V; c->gra = gra; c->p = *p;
	c->mt = mt_of(p->arith_mode);
	mat3_inv(ref->K, c->Hr);
	for (int v = 0; v < V; ++v) {
		double KR[9];
		mat3_mul(srcs[v].K, srcs[v].R, KR);
		mat3_mul_bt(KR, ref->R, c->Hl[v]);
		const double dC[3] = {ref->C[0] - srcs[v].C[0], ref->C[1] - srcs[v].C[1], ref->C[2] - srcs[v].C[2]};
		for (int i = 0; i < 3; ++i) c->Hm[v][i] = KR[i * 3] * dC[0] + KR[i * 3 + 1] * dC[1] + KR[i * 3 + 2] * dC[2];
		mat3_mul(c->Hl[v], c->Hr, c->A[v]);
		for (int i = 0; i < 9; ++i) c->Af[v][i] = (float)c->A[v][i];
		for (int i = 0; i < 3; ++i) c->Hmf[v][i] = (float)c->Hm[v][i];
	}
	for (int i = 0; i < 9; ++i) c->Hrf[i] = (float)c->Hr[i];
	c->ifx = 1.0 / ref->K[0]; c->ify = 1.0 / ref->K[4];
	c->S = device_segments(V);
	c->dMin = dMin; c->dMax = dMax;
	c->dMinSqr = sqrtf(dMin); c->dMaxSqr = sqrtf(dMax);
	c->smoothBonusDepth = 1.f - p->random_smooth_bonus;
	c->smoothBonusNormal = (1.f - p->random_smooth_bonus) * 0.96f;
	c->smoothSigmaDepth = -1.f / (2.f * SQ(p->random_smooth_depth));
	c->smoothSigmaNormal = -1.f / (2.f * SQ(fd2r(p->random_smooth_normal_deg)));
	c->angle1Range = fd2r(p->random_angle1_deg);
	c->angle2Range = fd2r(p->random_angle2_deg);
	c->thConfSmall = p->ncc_threshold_keep * 0.2f;
	c->thConfBig = p->ncc_threshold_keep * 0.4f;
	c->thConfRand = p->ncc_threshold_keep * 0.9f;
	c->thRobust = p->ncc_threshold_keep * 1.2f;
}

/* per-pixel state (DM.h:460-476) */
typedef struct {
	int x, y;
	int a, nside, ntaps;
	float w[HCOR_MAX_TAPS], tw[HCOR_MAX_TAPS];
	float sumW, normSq0;
	double X0[3];
	float viewDir[3];
	int nClose;
	float cDepth[HCOR_MAX_NEIGHBORS];
	float cNormal[HCOR_MAX_NEIGHBORS][3];
	float cX[HCOR_MAX_NEIGHBORS][3];
	int cSlot[HCOR_MAX_NEIGHBORS]; /* position in the candidate list (= lane of the slot on the device) */
	float planeN[3], planeD;
} pix_state;

/* nSizeHalfWindow generalised (hcmvs_oracle.h): 7 for every patch the reference supports, adapthalfwin beyond that */
static inline int border_of(const hcor_params* p) { return p->adapthalfwin > HCOR_HALF_WINDOW ? p->adapthalfwin : HCOR_HALF_WINDOW; }
static inline int border_ok(const est_ctx* c, int x, int y) { /* DM.cpp:442-447 */
	const int b = border_of(&c->p);
	return x - b >= 0 && y - b >= 0 && x + b < c->ref->width && y + b < c->ref->height;
}

/* device association: tap handled by segment seg at step m (same mapping as the kernels' tap_offset) */
static int dev_tap(int S, int a, int seg, int m, int* ti, int* tj) {
	const int nside = a + 1;
	int row, col;
	if (a > HCOR_HALF_WINDOW) {
		/* patches beyond the reference's 64 taps: the taps in the reference's order (rows outer, columns inner,
		 * DM.cpp:486-494) are dealt round-robin to the S lanes of a view group, tap k = m * S + seg */
		const int k = m * S + seg, nt = nside * nside;
		const int kk = k < nt ? k : nt - 1;
		*ti = kk / nside; *tj = kk % nside;
		return k < nt;
	}
	col = seg & 7; /* 8 tap segments per view: a segment owns one patch column and walks down its rows */
	row = m;
	const int valid = row < nside && col < nside;
	*ti = row < nside ? row : nside - 1;
	*tj = col < nside ? col : nside - 1;
	return valid;
}

/* butterfly-ordered sum of S partials (device association): p[s] += p[s^step], step = 1,2,4,.. */
static float butterfly_sum(float* p, int S) {
	float t[64];
	for (int step = 1; step < S; step <<= 1) {
		for (int s = 0; s < S; ++s) t[s] = p[s] + p[s ^ step];
		memcpy(p, t, sizeof(float) * S);
	}
	return p[0];
}

/* DM.cpp:450-519 */
static void fill_patch(const est_ctx* c, pix_state* ps, int x, int y) {
	const hcor_view* ref = c->ref;
	const int W = ref->width;
	ps->x = x; ps->y = y;
	const float tx = (float)c->gra[y * W + x];
	ps->a = tx > 100 ? 5 : c->p.adapthalfwin;
	const int a = ps->a;
	ps->nside = a + 1;                  /* ((2a+2)/2) */
	ps->ntaps = ps->nside * ps->nside;
	const float colCenter = ref->gray[y * W + x];
	const float sigmaColor = -1.f / (2.f * SQ(0.2f));
	const float sigmaSpatial = -1.f / (2.f * (float)SQ(a));
	float I[HCOR_MAX_TAPS];
	int n = 0;
	for (int i = -a; i <= a; i += 2)
		for (int j = -a; j <= a; j += 2) {
			I[n] = ref->gray[(y + i) * W + (x + j)];
			const float wColor = SQ(I[n] - colCenter) * sigmaColor;
			const float wSpatial = (float)(SQ(j) + SQ(i)) * sigmaSpatial;
			ps->w[n] = c->mt->expf_(wColor + wSpatial);
			++n;
		}
	if (c->p.arith_mode == HCOR_ARITH_REFERENCE) {
		float nrm = 0, sw = 0;
		for (int k = 0; k < n; ++k) { nrm += I[k] * ps->w[k]; sw += ps->w[k]; }
		const float tm = nrm / sw;
		nrm = 0;
		for (int k = 0; k < n; ++k) {
			const float t = I[k] - tm;
			ps->tw[k] = ps->w[k] * t;
			nrm += ps->tw[k] * t;
		}
		ps->sumW = sw; ps->normSq0 = nrm;
	} else {
		const int S = c->S, big = a > HCOR_HALF_WINDOW, MAXM = big ? (HCOR_MAX_TAPS + S - 1) / S : 64 / S, nside = ps->nside;
		float pa[64], pb[64];
		if (!big) {
			/* up to 64 taps: the kernels give every tap of the patch its own lane (lane = 8 * row + column, lanes past
			 * the patch add exactly +0) and sum the 64 lanes with the xor butterfly */
			for (int l = 0; l < 64; ++l) {
				const int row = l >> 3, col = l & 7;
				pa[l] = pb[l] = 0.f;
				if (row >= nside || col >= nside) continue;
				const int k = row * nside + col;
				pa[l] = I[k] * ps->w[k]; pb[l] = ps->w[k];
			}
			const float swi = butterfly_sum(pa, 64), sw = butterfly_sum(pb, 64);
			const float tm = swi / sw;
			for (int l = 0; l < 64; ++l) {
				const int row = l >> 3, col = l & 7;
				pa[l] = 0.f;
				if (row >= nside || col >= nside) continue;
				const int k = row * nside + col;
				const float t = I[k] - tm;
				ps->tw[k] = ps->w[k] * t;
				pa[l] = ps->tw[k] * t;
			}
			ps->sumW = sw; ps->normSq0 = butterfly_sum(pa, 64);
		} else { /* patches beyond 64 taps: the taps are dealt round-robin to the S lanes of a view group (dev_tap) */
		for (int s = 0; s < S; ++s) {
			float sa = 0, sb = 0;
			for (int m = 0; m < MAXM; ++m) {
				int ti, tj;
				if (!dev_tap(S, a, s, m, &ti, &tj)) continue;
				const int k = ti * nside + tj;
				sa = fmaf(I[k], ps->w[k], sa); sb = sb + ps->w[k];
			}
			pa[s] = sa; pb[s] = sb;
		}
		const float swi = butterfly_sum(pa, S), sw = butterfly_sum(pb, S);
		const float tm = swi / sw;
		for (int s = 0; s < S; ++s) {
			float sa = 0;
			for (int m = 0; m < MAXM; ++m) {
				int ti, tj;
				if (!dev_tap(S, a, s, m, &ti, &tj)) continue;
				const int k = ti * nside + tj;
				const float t = I[k] - tm;
				ps->tw[k] = ps->w[k] * t;
				sa = fmaf(ps->tw[k], t, sa);
			}
			pa[s] = sa;
		}
		ps->sumW = sw; ps->normSq0 = butterfly_sum(pa, S);
		}
	}

	/* DM.cpp:517, Camera.h:299-304 */
	if (c->p.arith_mode == HCOR_ARITH_DEVICE) {
		ps->X0[0] = ((double)x - ref->K[2]) * c->ifx;
		ps->X0[1] = ((double)y - ref->K[5]) * c->ify;
	} else {
		ps->X0[0] = ((double)x - ref->K[2]) / ref->K[0];
		ps->X0[1] = ((double)y - ref->K[5]) / ref->K[4];
	}
	ps->X0[2] = 1.0;
	for (int i = 0; i < 3; ++i) ps->viewDir[i] = (float)ps->X0[i];
	ps->nClose = 0;
}

static inline float dot3f(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Types.inl:2250-2258 */
static inline float sample_ref(const hcor_view* im, float px, float py) {
	const int lx = (int)px, ly = (int)py;
	const float x = px - lx, x1 = 1.f - x;
	const float y = py - ly, y1 = 1.f - y;
	const float* r0 = im->gray + (size_t)ly * im->width + lx;
	const float* r1 = r0 + im->width;
	return (r0[0] * x1 + r0[1] * x) * y1 + (r1[0] * x1 + r1[1] * x) * y;
}
static inline float sample_dev(const hcor_view* im, float px, float py) {
	const int lx = (int)px, ly = (int)py;
	const float x = px - (float)lx;
	const float y = py - (float)ly;
	const float* r0 = im->gray + (size_t)ly * im->width + lx;
	const float* r1 = r0 + im->width;
	const float t = fmaf(x, r0[1] - r0[0], r0[0]);
	const float b = fmaf(x, r1[1] - r1[0], r1[0]);
	return fmaf(y, b - t, t);
}
static inline int inside_border1(const hcor_view* im, float px, float py) { /* Types.h:1633-1635 */
	return px >= 1.f && py >= 1.f && px <= (float)(im->width - 2) && py <= (float)(im->height - 2);
}

/* smoothness factor product terms (DM.cpp:607-615); f[k] = (1-bD*fd)(1-bN*fn) */
static int smooth_factors(const est_ctx* c, const pix_state* ps, float depth, const float* normal, float* f) {
	for (int k = 0; k < ps->nClose; ++k) {
		const float dist = dot3f(ps->planeN, ps->cX[k]) + ps->planeD; /* Planef::Distance */
		const float fd = c->mt->expf_(SQ(dist / depth) * c->smoothSigmaDepth);
		/* Util.inl:417-420 ComputeAngle */
		const float* nb = ps->cNormal[k];
		float ca = c->p.arith_mode == HCOR_ARITH_DEVICE ? dot3f(normal, nb) /* unit normals: no re-normalisation */
		                                                : dot3f(normal, nb) / sqrtf(dot3f(normal, normal) * dot3f(nb, nb));
		ca = ca < -1.f ? -1.f : (ca > 1.f ? 1.f : ca);
		const float ang = c->mt->acosf_(ca);
		const float fn = c->mt->expf_(SQ(ang) * c->smoothSigmaNormal);
		f[k] = (1.f - c->smoothBonusDepth * fd) * (1.f - c->smoothBonusNormal * fn);
	}
	return ps->nClose;
}

/* DM.cpp:522-616, 890-893: reference association */
static float score_view_ref(const est_ctx* c, const pix_state* ps, int v, float depth, const float* normal,
                            const float* sf, int nsf) {
	const hcor_view* im = &c->srcs[v];
	/* DM.h:565-574 */
	const double n[3] = {normal[0], normal[1], normal[2]};
	const double inv = 1.0 / ((n[0] * ps->X0[0] + n[1] * ps->X0[1] + n[2] * ps->X0[2]) * (double)depth);
	double M[9], Hd[9];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) M[i * 3 + j] = c->Hl[v][i * 3 + j] + c->Hm[v][i] * (n[j] * inv);
	mat3_mul(M, c->Hr, Hd);
	float H[9];
	for (int i = 0; i < 9; ++i) H[i] = (float)Hd[i];
	const int a = ps->a;
	const float bx = (float)(ps->x - a), by = (float)(ps->y - a);
	float X[3] = {H[0] * bx + H[1] * by + H[2], H[3] * bx + H[4] * by + H[5], H[6] * bx + H[7] * by + H[8]};
	float B[3] = {X[0], X[1], X[2]};
	for (int i = 0; i < 9; ++i) H[i] *= 2.f; /* nSizeStep */
	int k = 0;
	float sum = 0, sumSq = 0, num = 0;
	for (int i = -a; i <= a; i += 2) {
		for (int j = -a; j <= a; j += 2) {
			const float px = X[0] / X[2], py = X[1] / X[2];
			if (!inside_border1(im, px, py)) return c->thRobust;
			const float val = sample_ref(im, px, py);
			const float vw = val * ps->w[k];
			sum += vw;
			sumSq += val * vw;
			num += val * ps->tw[k];
			++k;
			X[0] += H[0]; X[1] += H[3]; X[2] += H[6];
		}
		B[0] += H[1]; B[1] += H[4]; B[2] += H[7];
		X[0] = B[0]; X[1] = B[1]; X[2] = B[2];
	}
	const float normSq1 = sumSq - SQ(sum) / ps->sumW;
	const float nrmSq = ps->normSq0 * normSq1;
	if (nrmSq <= 0.f) return c->thRobust;
	float ncc = num / sqrtf(nrmSq);
	ncc = ncc < -1.f ? -1.f : (ncc > 1.f ? 1.f : ncc);
	float s = 1.f - ncc;
	for (int q = 0; q < nsf; ++q) s *= sf[q];
	return (1.f - c->p.photometric_flow) * s + c->p.photometric_flow * 0.f;
}

/* statistics of the device association's inside test (see score_view_dev) */
static unsigned long long g_inside_cols, g_inside_differ;
void hcor_inside_rule_stats(uint64_t* columns, uint64_t* differ, int reset) {
	if (columns) *columns = __atomic_load_n(&g_inside_cols, __ATOMIC_RELAXED);
	if (differ) *differ = __atomic_load_n(&g_inside_differ, __ATOMIC_RELAXED);
	if (reset) { __atomic_store_n(&g_inside_cols, 0ull, __ATOMIC_RELAXED); __atomic_store_n(&g_inside_differ, 0ull, __ATOMIC_RELAXED); }
}

/* device association of the same computation */
static void device_H(const est_ctx* c, const pix_state* ps, int v, float depth, const float* normal, float* H) {
	const float n0 = normal[0], n1 = normal[1], n2 = normal[2];
	const float nx0 = fmaf(n2, 1.0f, fmaf(n1, ps->viewDir[1], n0 * ps->viewDir[0]));
	const float inv = 1.0f / (nx0 * depth);
	float q[3];
	for (int j = 0; j < 3; ++j) q[j] = fmaf(n2, c->Hrf[6 + j], fmaf(n1, c->Hrf[3 + j], n0 * c->Hrf[j])) * inv;
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) H[i * 3 + j] = fmaf(c->Hmf[v][i], q[j], c->Af[v][i * 3 + j]);
}
static float score_view_dev(const est_ctx* c, const pix_state* ps, int v, float depth, const float* normal,
                            const float* sf, int nsf) {
	const hcor_view* im = &c->srcs[v];
	float H[9];
	device_H(c, ps, v, depth, normal, H);
	const int a = ps->a, nside = ps->nside, S = c->S, big = a > HCOR_HALF_WINDOW;
	const int MAXM = big ? (HCOR_MAX_TAPS + S - 1) / S : 64 / S; /* taps per lane; lanes past the patch repeat the last tap with zero weight */
	float p0[64], p1[64], p2[64];
	int ok = 1;
	for (int s = 0; s < S; ++s) {
		float Xx[HCOR_MAX_TAPS], Xy[HCOR_MAX_TAPS], Xz[HCOR_MAX_TAPS], iz[HCOR_MAX_TAPS];
		int kk[HCOR_MAX_TAPS];
		int vv[HCOR_MAX_TAPS];
		for (int m = 0; m < MAXM; ++m) {
			int ti, tj;
			vv[m] = dev_tap(S, a, s, m, &ti, &tj);
			kk[m] = ti * nside + tj;
			const float px = (float)(ps->x - a + 2 * tj), py = (float)(ps->y - a + 2 * ti);
			Xx[m] = fmaf(H[1], py, fmaf(H[0], px, H[2])); /* the column term is formed first */
			Xy[m] = fmaf(H[4], py, fmaf(H[3], px, H[5]));
			Xz[m] = fmaf(H[7], py, fmaf(H[6], px, H[8])); /* steps past the patch repeat the clamped tap */
		}
		if (big) { /* beyond 64 taps: one IEEE reciprocal per tap */
			for (int m = 0; m < MAXM; ++m) iz[m] = 1.0f / Xz[m];
		} else { /* one IEEE reciprocal for the eight steps of a lane (steps past the patch repeat the last row) */
			const float q01 = Xz[0] * Xz[1], q23 = Xz[2] * Xz[3], q45 = Xz[4] * Xz[5], q67 = Xz[6] * Xz[7];
			const float qa = q01 * q23, qb = q45 * q67;
			const float r = 1.0f / (qa * qb);
			const float ra = r * qb, rb = r * qa;
			const float r01 = ra * q23, r23 = ra * q01, r45 = rb * q67, r67 = rb * q45;
			iz[0] = r01 * Xz[1]; iz[1] = r01 * Xz[0]; iz[2] = r23 * Xz[3]; iz[3] = r23 * Xz[2];
			iz[4] = r45 * Xz[5]; iz[5] = r45 * Xz[4]; iz[6] = r67 * Xz[7]; iz[7] = r67 * Xz[6];
		}
		float sum = 0, sumSq = 0, num = 0;
		int colOk = 1;
		if (!big) {
			/* device association of the inside test: a lane holds one patch COLUMN; a homography maps that segment to a segment
			 * as long as z keeps its sign along it (z is affine in the step: the two ends decide), and both coordinates are
			 * monotone along it, so the two end taps are tested instead of every tap (the reference: every tap, DM.cpp:566) */
			const int e = nside - 1; /* the column's last REAL row (steps past it repeat it; their reciprocals come out of another branch of the product tree) */
			colOk = inside_border1(im, Xx[0] * iz[0], Xy[0] * iz[0]) && inside_border1(im, Xx[e] * iz[e], Xy[e] * iz[e]) && Xz[0] * Xz[e] > 0.f;
			{ /* how often the end-tap rule and the reference's every-tap rule disagree (hcor_inside_rule_stats, tests only) */
				int all = 1;
				for (int m = 0; m <= e; ++m) all = all && inside_border1(im, Xx[m] * iz[m], Xy[m] * iz[m]);
				__atomic_fetch_add(&g_inside_cols, 1ull, __ATOMIC_RELAXED);
				if (all != colOk) __atomic_fetch_add(&g_inside_differ, 1ull, __ATOMIC_RELAXED);
			}
			if (!colOk) { ok = 0; p0[s] = p1[s] = p2[s] = 0; continue; }
		}
		for (int m = 0; m < MAXM; ++m) {
			const float qx = Xx[m] * iz[m], qy = Xy[m] * iz[m];
			if (big && !inside_border1(im, qx, qy)) { ok = 0; continue; }
			if (!vv[m]) continue; /* clamped duplicate of a real tap: zero weight */
			const int k = kk[m];
			const float val = sample_dev(im, qx, qy);
			const float vw = val * ps->w[k];
			sum = sum + vw;
			sumSq = fmaf(val, vw, sumSq);
			num = fmaf(val, ps->tw[k], num);
		}
		p0[s] = sum; p1[s] = sumSq; p2[s] = num;
	}
	if (!ok) return c->thRobust;
	const float sum = butterfly_sum(p0, S), sumSq = butterfly_sum(p1, S), num = butterfly_sum(p2, S);
	const float normSq1 = sumSq - SQ(sum) * (1.0f / ps->sumW);
	const float nrmSq = ps->normSq0 * normSq1;
	if (!(nrmSq > 0.f)) return c->thRobust;
	float ncc = num / sqrtf(nrmSq);
	ncc = ncc < -1.f ? -1.f : (ncc > 1.f ? 1.f : ncc);
	float s = 1.f - ncc;
	for (int q = 0; q < nsf; ++q) s *= sf[q];
	return (1.f - c->p.photometric_flow) * s;
}

/* DM.cpp:987-1046 (DENSE_AGGNCC_MINMEAN) */
static float score_pixel(est_ctx* c, const pix_state* ps, float depth, const float* normal) {
	float sf[HCOR_MAX_NEIGHBORS];
	int nsf = smooth_factors(c, ps, depth, normal, sf);
	if (c->p.arith_mode == HCOR_ARITH_DEVICE) {
		/* device association: the factors are multiplied by a butterfly over 8-slot chunks (slots without a
		 * neighbour count as 1), chunk after chunk, and the score is multiplied once by the product */
		float bySlot[HCOR_MAX_NEIGHBORS];
		int top = 0;
		for (int k = 0; k < HCOR_MAX_NEIGHBORS; ++k) bySlot[k] = 1.f;
		for (int k = 0; k < nsf; ++k) { bySlot[ps->cSlot[k]] = sf[k]; if (ps->cSlot[k] + 1 > top) top = ps->cSlot[k] + 1; }
		float F = 1.f;
		for (int ch = 0; ch * 8 < top; ++ch) {
			const float* f = bySlot + ch * 8;
			const float t = ((f[0] * f[1]) * (f[2] * f[3])) * ((f[4] * f[5]) * (f[6] * f[7]));
			F = ch == 0 ? t : F * t;
		}
		sf[0] = F;
		nsf = 1;
	}
	float s0 = FLT_MAX, s1 = FLT_MAX; /* two smallest */
	for (int v = 0; v < c->V; ++v) {
		const float s = c->p.arith_mode == HCOR_ARITH_DEVICE ? score_view_dev(c, ps, v, depth, normal, sf, nsf)
		                                                      : score_view_ref(c, ps, v, depth, normal, sf, nsf);
		if (s < s0) { s1 = s0; s0 = s; }
		else if (s < s1) s1 = s;
	}
	c->evals++;
	if (c->V <= 1) return s0;
	if (s1 >= c->thRobust) return s0;
	return (s0 + s1) / 2;
}

/* Util.inl:614-626 */
static void normal2dir(const mathtab* mt, const float* d, float* p) {
	p[0] = mt->atan2f_(d[1], d[0]);
	p[1] = mt->acosf_(d[2]);
}
static void dir2normal(const mathtab* mt, const float* p, float* d) {
	const float siny = mt->sinf_(p[1]);
	d[0] = mt->cosf_(p[0]) * siny;
	d[1] = mt->sinf_(p[0]) * siny;
	d[2] = mt->cosf_(p[1]);
}
void hcor_normal2dir(const float n[3], float p[2], int mode) { normal2dir(mt_of(mode), n, p); }
void hcor_dir2normal(const float p[2], float n[3], int mode) { dir2normal(mt_of(mode), p, n); }

/* DM.h:629-634 + Rotation.inl:707-733 */
static void correct_normal(const mathtab* mt, const float* viewDir, float* n) {
	const float cosAngLen = dot3f(n, viewDir);
	if (!(cosAngLen >= 0)) return;
	const float axis[3] = {n[1] * viewDir[2] - n[2] * viewDir[1], n[2] * viewDir[0] - n[0] * viewDir[2],
	                       n[0] * viewDir[1] - n[1] * viewDir[0]};
	const float vlen = sqrtf(dot3f(viewDir, viewDir));
	float phi = (mt->acosf_(cosAngLen / vlen) - fd2r(90.f)) * 1.01f;
	if (!(phi < -0.001f)) phi = -0.001f; /* MINF(phi, -0.001f) */
	const float wnorm = sqrtf(dot3f(axis, axis));
	if (!(wnorm >= FLT_EPSILON)) return; /* D3 */
	const float iw = 1.f / wnorm;
	const float w[3] = {axis[0] * iw, axis[1] * iw, axis[2] * iw};
	const float O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
	const float sp = mt->sinf_(phi), cp1 = 1.f - mt->cosf_(phi);
	float R[9];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			float s = 0;
			for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
			R[i * 3 + j] = ((i == j ? 1.f : 0.f) + O[i * 3 + j] * sp) + s * cp1;
		}
	const float r[3] = {R[0] * n[0] + R[1] * n[1] + R[2] * n[2], R[3] * n[0] + R[4] * n[1] + R[5] * n[2],
	                    R[6] * n[0] + R[7] * n[1] + R[8] * n[2]};
	n[0] = r[0]; n[1] = r[1]; n[2] = r[2];
}

/* DM.cpp:1671-1726 (ray-plane branch) */
static float interpolate_pixel(const est_ctx* c, const pix_state* ps, int nx, int ny, float depth, const float* normal) {
	const hcor_view* ref = c->ref;
	const double pn[3] = {normal[0], normal[1], normal[2]};
	const double z = depth;
	double P[3] = {((double)nx - ref->K[2]) * z / ref->K[0], ((double)ny - ref->K[5]) * z / ref->K[4], z};
	if (c->p.arith_mode == HCOR_ARITH_DEVICE) {
		P[0] = ((double)nx - ref->K[2]) * z * c->ifx;
		P[1] = ((double)ny - ref->K[5]) * z * c->ify;
	}
	const double planeD = pn[0] * P[0] + pn[1] * P[1] + pn[2] * P[2];
	const float dn = (float)(planeD / (pn[0] * ps->X0[0] + pn[1] * ps->X0[1] + pn[2] * ps->X0[2]));
	return (c->dMin <= dn && dn < c->dMax) ? dn : depth;
}
static inline void init_plane(pix_state* ps, float depth, const float* normal) { /* DM.cpp:1730-1738 */
	ps->planeN[0] = normal[0]; ps->planeN[1] = normal[1]; ps->planeN[2] = normal[2];
	ps->planeD = -depth * dot3f(normal, ps->viewDir);
}

static inline float random_depth(const est_ctx* c, float u) { /* DM.h:618-621 */
	const float r = c->dMinSqr + (c->dMaxSqr - c->dMinSqr) * u;
	return r * r;
}
static void random_normal(const est_ctx* c, const float* viewRay, float u1, float u2, float* n) { /* DM.h:622-626 */
	const float p[2] = {fd2r(0.f) + (fd2r(180.f) - fd2r(0.f)) * u1, fd2r(90.f) + (fd2r(180.f) - fd2r(90.f)) * u2};
	dir2normal(c->mt, p, n);
	if (dot3f(n, viewRay) > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
}

/* public single-piece wrappers for known-answer tests */
int hcor_fill_patch(const hcor_view* ref, const uint8_t* gra, const hcor_params* p, int n_src, int x, int y,
                    float* weight, float* temp_weight, float* sum_weights, float* norm_sq0) {
	est_ctx c; pix_state ps;
	ctx_init(&c, ref, ref, 0, gra, p, 1.f, 2.f);
	c.S = device_segments(n_src);
	fill_patch(&c, &ps, x, y);
	memcpy(weight, ps.w, sizeof(float) * ps.ntaps);
	memcpy(temp_weight, ps.tw, sizeof(float) * ps.ntaps);
	*sum_weights = ps.sumW; *norm_sq0 = ps.normSq0;
	return ps.ntaps;
}
float hcor_score_view(const hcor_view* ref, const hcor_view* src, const uint8_t* gra, const hcor_params* p, int x,
                      int y, float depth, const float normal[3]) {
	est_ctx c; pix_state ps;
	ctx_init(&c, ref, src, 1, gra, p, 1.f, 2.f);
	fill_patch(&c, &ps, x, y);
	return p->arith_mode == HCOR_ARITH_DEVICE ? score_view_dev(&c, &ps, 0, depth, normal, NULL, 0)
	                                          : score_view_ref(&c, &ps, 0, depth, normal, NULL, 0);
}
float hcor_score_pixel(const hcor_view* ref, const hcor_view* srcs, int n_src, const uint8_t* gra,
                       const hcor_params* p, int x, int y, float depth, const float normal[3]) {
	est_ctx c; pix_state ps;
	ctx_init(&c, ref, srcs, n_src, gra, p, 1.f, 2.f);
	fill_patch(&c, &ps, x, y);
	return score_pixel(&c, &ps, depth, normal);
}
void hcor_correct_normal(const hcor_view* ref, int x, int y, float n[3], int mode) {
	const float vd[3] = {(float)(((double)x - ref->K[2]) / ref->K[0]), (float)(((double)y - ref->K[5]) / ref->K[4]), 1.f};
	correct_normal(mt_of(mode), vd, n);
}
float hcor_interpolate_pixel(const hcor_view* ref, int x, int y, int nx, int ny, float depth, const float normal[3],
                             float d_min, float d_max) {
	est_ctx c; pix_state ps;
	hcor_params p; hcor_default_params(&p);
	ctx_init(&c, ref, ref, 0, NULL, &p, d_min, d_max);
	ps.X0[0] = ((double)x - ref->K[2]) / ref->K[0];
	ps.X0[1] = ((double)y - ref->K[5]) / ref->K[4];
	ps.X0[2] = 1.0;
	return interpolate_pixel(&c, &ps, nx, ny, depth, normal);
}

/* ------------------------------------------------------------------------------------------------ */
/* passes                                                                                           */

#define STREAM_SCORE(it_ext) ((uint32_t)(it_ext) * 64u)
#define STREAM_SWEEP(it_ext, iter) ((uint32_t)(it_ext) * 64u + 1u + (uint32_t)(iter))

/* SD.cpp:649-675 ScoreDepthMapTmp body for one pixel */
static void score_one(est_ctx* c, int x, int y, float* depth, float* normal, float* conf) {
	const int W = c->ref->width;
	const int idx = y * W + x;
	if (!border_ok(c, x, y)) {
		depth[idx] = 0; normal[3 * idx] = normal[3 * idx + 1] = normal[3 * idx + 2] = 0; conf[idx] = 2.f;
		return;
	}
	pix_state ps;
	fill_patch(c, &ps, x, y);
	float d = depth[idx];
	float n[3] = {normal[3 * idx], normal[3 * idx + 1], normal[3 * idx + 2]};
	const uint32_t st = STREAM_SCORE(c->p.it_external);
	if (!(c->dMin <= d && d < c->dMax)) {
		d = random_depth(c, rand_unit(hcor_rand_u32(c->p.seed, (uint32_t)idx, st, 0)));
		random_normal(c, ps.viewDir, rand_unit(hcor_rand_u32(c->p.seed, (uint32_t)idx, st, 1)),
		              rand_unit(hcor_rand_u32(c->p.seed, (uint32_t)idx, st, 2)), n);
	} else if (dot3f(n, ps.viewDir) >= 0) {
		random_normal(c, ps.viewDir, rand_unit(hcor_rand_u32(c->p.seed, (uint32_t)idx, st, 1)),
		              rand_unit(hcor_rand_u32(c->p.seed, (uint32_t)idx, st, 2)), n);
	}
	depth[idx] = d; normal[3 * idx] = n[0]; normal[3 * idx + 1] = n[1]; normal[3 * idx + 2] = n[2];
	conf[idx] = score_pixel(c, &ps, d, n);
}

static void add_close(const est_ctx* c, pix_state* ps, int nx, int ny, float nd, const float* nmap, int slot) {
	const hcor_view* ref = c->ref;
	const int k = ps->nClose++;
	ps->cSlot[k] = slot;
	const int nidx = ny * ref->width + nx;
	ps->cDepth[k] = nd;
	ps->cNormal[k][0] = nmap[3 * nidx]; ps->cNormal[k][1] = nmap[3 * nidx + 1]; ps->cNormal[k][2] = nmap[3 * nidx + 2];
	/* Cast<float>(camera.TransformPointI2C(Point3(nx, ndepth))), Camera.h:306-312 */
	const double z = nd;
	if (c->p.arith_mode == HCOR_ARITH_DEVICE) {
		ps->cX[k][0] = (float)(((double)nx - ref->K[2]) * z * c->ifx);
		ps->cX[k][1] = (float)(((double)ny - ref->K[5]) * z * c->ify);
	} else {
		ps->cX[k][0] = (float)(((double)nx - ref->K[2]) * z / ref->K[0]);
		ps->cX[k][1] = (float)(((double)ny - ref->K[5]) * z / ref->K[4]);
	}
	ps->cX[k][2] = (float)z;
}

/* DM.cpp:1050-1501 ProcessPixel (DENSE_REFINE_ITER, DENSE_SMOOTHNESS_PLANE) */
static void process_pixel(est_ctx* c, int x, int y, int iter, float* depthMap, float* normalMap, float* confMap) {
	const hcor_view* ref = c->ref;
	const int W = ref->width, H = ref->height, hw7 = border_of(&c->p);
	if (!border_ok(c, x, y)) return;
	pix_state ps;
	fill_patch(c, &ps, x, y);
	const int rev = (iter % 2) != 0; /* dir = RB2LT for odd iterations, DM.cpp:418 */
	int nbx[HCOR_MAX_NEIGHBORS], nby[HCOR_MAX_NEIGHBORS], nbc[HCOR_MAX_NEIGHBORS];
	int nNb = 0;
	if (c->p.it_external >= 1) {
		/* DM.cpp:1064-1274: cross pattern, identical for both directions */
		const float tx = (float)c->gra[y * W + x];
		int phw = tx > 150 ? 5 : c->p.propagate_halfwin;
		if (phw > 7) phw = 7;
		const int step = c->p.propagate_step > 0 ? c->p.propagate_step : 1;
		int cx[HCOR_MAX_NEIGHBORS], cy[HCOR_MAX_NEIGHBORS], nc = 0;
		if (x > phw && y > phw && x < W - phw && y < H - phw) {
			for (int i = 1; i <= phw; i += step) {
				cx[nc] = x; cy[nc++] = y - i;
				cx[nc] = x; cy[nc++] = y + i;
				cx[nc] = x - i; cy[nc++] = y;
				cx[nc] = x + i; cy[nc++] = y;
			}
		} else if (x > hw7 && y > hw7 && x < W - hw7 && y < H - hw7) {
			cx[nc] = x; cy[nc++] = y - 1;
			cx[nc] = x; cy[nc++] = y + 1;
			cx[nc] = x - 1; cy[nc++] = y;
			cx[nc] = x + 1; cy[nc++] = y;
		}
		for (int k = 0; k < nc; ++k) {
			const float nd = depthMap[cy[k] * W + cx[k]];
			if (nd > 0) {
				nbx[nNb] = cx[k]; nby[nNb] = cy[k]; nbc[nNb] = ps.nClose; ++nNb;
				add_close(c, &ps, cx[k], cy[k], nd, normalMap, k);
			}
		}
	} else {
		/* DM.cpp:1275-1391 */
		const int px[4] = {x - 1, x, x + 1, x}, py[4] = {y, y - 1, y, y + 1};
		const int valid[4] = {x > hw7, y > hw7, x < W - hw7, y < H - hw7};
		const int order_fwd[4] = {0, 1, 2, 3}, order_rev[4] = {2, 3, 0, 1};
		const int* ord = rev ? order_rev : order_fwd;
		for (int q = 0; q < 4; ++q) {
			const int k = ord[q];
			if (!valid[k]) continue;
			const float nd = depthMap[py[k] * W + px[k]];
			if (nd > 0) {
				if (q < 2) { nbx[nNb] = px[k]; nby[nNb] = py[k]; nbc[nNb] = ps.nClose; ++nNb; }
				add_close(c, &ps, px[k], py[k], nd, normalMap, q);
			}
		}
	}
	const int idx = y * W + x;
	float conf = confMap[idx], depth = depthMap[idx];
	float normal[3] = {normalMap[3 * idx], normalMap[3 * idx + 1], normalMap[3 * idx + 2]};
	init_plane(&ps, depth, normal); /* D2 */
	/* propagation, DM.cpp:1406-1440 */
	for (int q = 0; q < nNb; ++q) {
		if (confMap[nby[q] * W + nbx[q]] >= c->p.ncc_threshold_keep) continue;
		const int k = nbc[q];
		ps.cDepth[k] = interpolate_pixel(c, &ps, nbx[q], nby[q], ps.cDepth[k], ps.cNormal[k]);
		correct_normal(c->mt, ps.viewDir, ps.cNormal[k]);
		init_plane(&ps, ps.cDepth[k], ps.cNormal[k]);
		const float nconf = score_pixel(c, &ps, ps.cDepth[k], ps.cNormal[k]);
		if (conf > nconf) {
			conf = nconf; depth = ps.cDepth[k];
			normal[0] = ps.cNormal[k][0]; normal[1] = ps.cNormal[k][1]; normal[2] = ps.cNormal[k][2];
		}
	}
	/* refinement, DM.cpp:1442-1501 */
	const uint32_t st = STREAM_SWEEP(c->p.it_external, iter);
	const uint32_t seed = c->p.seed;
	unsigned idxScaleRange = 0;
	int done = 0;
	for (;;) { /* RefineIters: */
		if (conf <= c->thConfSmall) idxScaleRange = 2;
		else if (conf <= c->thConfBig) idxScaleRange = 1;
		else if (conf >= c->thConfRand) {
			int again = 0;
			for (int it = 0; it < c->p.n_random_iters; ++it) {
				const float nd = random_depth(c, rand_unit(hcor_rand_u32(seed, (uint32_t)idx, st, 3u * it)));
				float nn[3];
				random_normal(c, ps.viewDir, rand_unit(hcor_rand_u32(seed, (uint32_t)idx, st, 3u * it + 1)),
				              rand_unit(hcor_rand_u32(seed, (uint32_t)idx, st, 3u * it + 2)), nn);
				const float nconf = score_pixel(c, &ps, nd, nn);
				if (conf > nconf) {
					conf = nconf; depth = nd; normal[0] = nn[0]; normal[1] = nn[1]; normal[2] = nn[2];
					if (conf < c->thConfRand) { again = 1; break; }
				}
			}
			if (again) continue;
			done = 1;
		}
		break;
	}
	if (!done) {
		float scaleRange = 1.f / (float)(1u << idxScaleRange);
		const float depthRange = depth * c->p.random_depth_ratio; /* Util.inl:650-656 */
		float p[2];
		normal2dir(c->mt, normal, p);
		for (int it = 0; it < c->p.n_random_iters; ++it) {
			const uint32_t cb = 64u + 3u * it;
			const float nd = depth + (depthRange * scaleRange) * (2.f * rand_unit(hcor_rand_u32(seed, (uint32_t)idx, st, cb)) - 1.f);
			if (!(c->dMin <= nd && nd < c->dMax)) continue;
			const float np[2] = {
				p[0] + (c->angle1Range * scaleRange) * (2.f * rand_unit(hcor_rand_u32(seed, (uint32_t)idx, st, cb + 1)) - 1.f),
				p[1] + (c->angle2Range * scaleRange) * (2.f * rand_unit(hcor_rand_u32(seed, (uint32_t)idx, st, cb + 2)) - 1.f)};
			float nn[3];
			dir2normal(c->mt, np, nn);
			if (dot3f(nn, ps.viewDir) >= 0) continue;
			init_plane(&ps, nd, nn);
			const float nconf = score_pixel(c, &ps, nd, nn);
			if (conf > nconf) {
				conf = nconf; depth = nd; normal[0] = nn[0]; normal[1] = nn[1]; normal[2] = nn[2];
				p[0] = np[0]; p[1] = np[1];
				++idxScaleRange;
				scaleRange = 1.f / (float)(1u << idxScaleRange); /* scaleRanges[], DM.cpp:384 */
			}
		}
	}
	/* restore/libs/MVS/DepthMap.cpp:1527-1549: last sweep of the last outer iteration */
	if (c->p.hint_depth && c->p.hint_normal && c->p.it_external == c->p.n_external_iters - 1 && iter == c->p.n_estimation_iters - 1 &&
	    c->p.hint_depth[idx] > 0) {
		float nn[3] = {c->p.hint_normal[3 * idx], c->p.hint_normal[3 * idx + 1], c->p.hint_normal[3 * idx + 2]};
		const float nd = interpolate_pixel(c, &ps, x, y, c->p.hint_depth[idx], nn);
		correct_normal(c->mt, ps.viewDir, nn);
		init_plane(&ps, nd, nn);
		const float nconf = score_pixel(c, &ps, nd, nn);
		if (conf > nconf - 0.1f) { conf = nconf; depth = nd; normal[0] = nn[0]; normal[1] = nn[1]; normal[2] = nn[2]; }
	}
	confMap[idx] = conf; depthMap[idx] = depth;
	normalMap[3 * idx] = normal[0]; normalMap[3 * idx + 1] = normal[1]; normalMap[3 * idx + 2] = normal[2];
}

void hcor_pass_score(const hcor_view* ref, const hcor_view* srcs, int V, const uint8_t* gra, const hcor_params* p,
                     float dMin, float dMax, float* depth, float* normal, float* conf, uint64_t* evals) {
	const int W = ref->width, H = ref->height;
	uint64_t total = 0;
	const int nt = p->n_threads > 0 ? p->n_threads : 1;
	(void)nt;
#pragma omp parallel num_threads(nt) reduction(+ : total)
	{
		est_ctx c;
		ctx_init(&c, ref, srcs, V, gra, p, dMin, dMax);
#pragma omp for schedule(dynamic, 4)
		for (int y = 0; y < H; ++y)
			for (int x = 0; x < W; ++x) score_one(&c, x, y, depth, normal, conf);
		total += c.evals;
	}
	if (evals) *evals += total;
}

void hcor_pass_sweep(const hcor_view* ref, const hcor_view* srcs, int V, const uint8_t* gra, const hcor_params* p,
                     int iter, float dMin, float dMax, float* depth, float* normal, float* conf, uint64_t* evals) {
	const int W = ref->width, H = ref->height;
	const int rev = (iter % 2) != 0;
	if (p->order == HCOR_ORDER_ZIGZAG) {
		/* SD.cpp:677-686 with one thread: coords forward (LT2RB) or reversed (RB2LT), DM.cpp:1054 */
		est_ctx c;
		ctx_init(&c, ref, srcs, V, gra, p, dMin, dMax);
		uint16_t* coords = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)W * H);
		const int stride = 8 * p->n_threads > 64 ? 8 * p->n_threads : 64; /* SD.cpp:835 */
		const int n = hcor_zigzag_coords(W, H, stride, coords);
		for (int i = 0; i < n; ++i) {
			const int k = rev ? n - 1 - i : i;
			process_pixel(&c, coords[2 * k], coords[2 * k + 1], iter, depth, normal, conf);
		}
		free(coords);
		if (evals) *evals += c.evals;
		return;
	}
	/* HCOR_ORDER_ROWS: row-pipelined wavefront.  A pixel needs its left/up neighbours (right/down when
	 * reversed) already updated and the opposite ones not yet updated; any order with that property --
	 * the zig-zag above, or rows advancing with a one-pixel lag -- produces identical maps because the
	 * RNG is keyed by pixel, not by visiting sequence. */
	const int nt = p->n_threads > 0 ? p->n_threads : 1;
	atomic_int* progress = (atomic_int*)calloc((size_t)H, sizeof(atomic_int)); /* pixels finished per logical row */
	uint64_t total = 0;
#pragma omp parallel num_threads(nt) reduction(+ : total)
	{
		est_ctx c;
		ctx_init(&c, ref, srcs, V, gra, p, dMin, dMax);
		int tid = 0, nth = 1;
#ifdef _OPENMP
		tid = omp_get_thread_num(); nth = omp_get_num_threads();
#endif
		for (int r = tid; r < H; r += nth) { /* logical row r; thread t owns rows t, t+nth, ... in order */
			const int y = rev ? H - 1 - r : r;
			for (int q = 0; q < W; ++q) {
				if (r > 0) { /* wait until the previous logical row has finished column q */
					while (atomic_load_explicit(&progress[r - 1], memory_order_acquire) < q + 1) {
					}
				}
				const int x = rev ? W - 1 - q : q;
				process_pixel(&c, x, y, iter, depth, normal, conf);
atomic_store_explicit(&progress[r], q + 1, memory_order_release);
			}
		}
		total += c.evals;
	}
	free(progress);
	if (evals) *evals += total;
}

void hcor_pass_end(const hcor_params* p, int W, int H, float* depth, float* normal, float* conf) {
	/* SD.cpp:688-744 */
	for (long i = 0; i < (long)W * H; ++i) {
		if (depth[i] <= 0 || conf[i] >= p->ncc_threshold_keep) {
			conf[i] = 0; normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0; depth[i] = 0;
		} else {
			conf[i] = conf[i] >= 1.f ? 0.f : 1.f - conf[i];
		}
	}
}

int hcor_estimate(const hcor_view* ref, const hcor_view* srcs, int V, const uint8_t* gra, const hcor_params* p,
                  float dMin, float dMax, float* depth, float* normal, float* conf, uint64_t* evals) {
	if (V < 1 || V > HCOR_MAX_VIEWS) return 1;
	if (p->adapthalfwin < 1 || p->adapthalfwin > HCOR_MAX_HALF_WINDOW) return 1;
	const int W = ref->width, H = ref->height;
	if (evals) *evals = 0;
	if (p->median_blur) {
		float* tmp = (float*)malloc(sizeof(float) * (size_t)W * H);
		hcor_median3(depth, W, H, tmp);
		memcpy(depth, tmp, sizeof(float) * (size_t)W * H);
		free(tmp);
	}
	hcor_pass_score(ref, srcs, V, gra, p, dMin, dMax, depth, normal, conf, evals);
	for (int iter = 0; iter < p->n_estimation_iters; ++iter)
		hcor_pass_sweep(ref, srcs, V, gra, p, iter, dMin, dMax, depth, normal, conf, evals);
	if (p->it_external == p->n_external_iters - 1) hcor_pass_end(p, W, H, depth, normal, conf);
	return 0;
}
