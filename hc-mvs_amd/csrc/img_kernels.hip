/*
 * hc-mvs_amd/csrc/img_kernels.hip -- image resampling on the device for the rescaled-neighbour path.
 *
 * Reference: DepthData::ViewData::ScaleImage (frame_main/libs/MVS/DepthMap.h:233-238): a source view whose average
 * footprint scale differs from the reference image's by 15 % or more is resampled with
 *     cv::resize(image, imageScaled, cv::Size(), scale, scale, scale > 1 ? cv::INTER_CUBIC : cv::INTER_AREA)
 * on its f32 gray image (SceneDensify.cpp:372-374).  OpenCV is absent here: both interpolations are restated from
 * OpenCV's published algorithm (imgproc/src/resize.cpp: computeResizeAreaTab + ResizeArea_Invoker for INTER_AREA with a
 * non-integer factor, ResizeAreaFast for an integer factor, the Keys kernel with A = -0.75 and replicated borders for
 * INTER_CUBIC) in one specified float operation order; the test suite's CPU checker restates the same specification
 * independently.  Parity unpinned (no fixture in the reference).
 *
 * One thread per destination pixel; the source taps of neighbouring threads overlap, so the reads are served by L2.
 * -ffp-contract=off: every multiply and add below is its own IEEE operation.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hcmvs {

// the x (or y) table of computeResizeAreaTab for destination index d: up to a leading partial cell, whole cells, a trailing
// partial cell.  scale = source pixels per destination pixel (double, as OpenCV holds it).
struct AreaSpan { int s1, s2; float aLead, aFull, aTrail; bool lead, trail; };
__device__ __forceinline__ AreaSpan area_span(int d, double scale, int ssize) {
	AreaSpan a;
	const double fs1 = d * scale, fs2 = fs1 + scale;
	const double cell = fmin(scale, (double)ssize - fs1);
	int s1 = (int)ceil(fs1), s2 = (int)floor(fs2);
	s2 = s2 < ssize - 1 ? s2 : ssize - 1;
	s1 = s1 < s2 ? s1 : s2;
	a.s1 = s1; a.s2 = s2;
	a.lead = (double)s1 - fs1 > 1e-3;
	a.aLead = (float)(((double)s1 - fs1) / cell);
	a.aFull = (float)(1.0 / cell);
	a.trail = fs2 - (double)s2 > 1e-3;
	a.aTrail = (float)(fmin(fmin(fs2 - (double)s2, 1.0), cell) / cell);
	return a;
}

__global__ void resize_area_kernel(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst, int dw, int dh, double scale) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= dw || y >= dh) return;
	const AreaSpan ax = area_span(x, scale, sw), ay = area_span(y, scale, sh);
	// one source row of the cell: buf = sum over the x table, in table order, starting from 0
	auto row = [&](int sy) {
		const float* S = src + (size_t)sy * sw;
		float buf = 0.f;
		if (ax.lead) buf = buf + S[ax.s1 - 1] * ax.aLead;
		for (int sx = ax.s1; sx < ax.s2; ++sx) buf = buf + S[sx] * ax.aFull;
		if (ax.trail) buf = buf + S[ax.s2] * ax.aTrail;
		return buf;
	};
	float sum = 0.f;
	bool first = true;
	auto acc = [&](int sy, float beta) {
		const float b = beta * row(sy);
		sum = first ? b : sum + b;
		first = false;
	};
	if (ay.lead) acc(ay.s1 - 1, ay.aLead);
	for (int sy = ay.s1; sy < ay.s2; ++sy) acc(sy, ay.aFull);
	if (ay.trail) acc(ay.s2, ay.aTrail);
	dst[(size_t)y * dw + x] = sum;
}

// integer factor: ResizeAreaFast -- the sum of the factor x factor block (rows outer, columns inner) times 1 / area
__global__ void resize_area_fast_kernel(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst, int dw, int dh, int factor) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= dw || y >= dh) return;
	float sum = 0.f;
	for (int j = 0; j < factor; ++j)
		for (int i = 0; i < factor; ++i) sum = sum + src[(size_t)(y * factor + j) * sw + (x * factor + i)];
	dst[(size_t)y * dw + x] = sum * (1.f / (float)(factor * factor));
}

__device__ __forceinline__ void cubic_coeffs(float t, float* w) { // interpolateCubic, A = -0.75
	const float A = -0.75f;
	w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
	w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
	w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
	w[3] = 1.f - w[0] - w[1] - w[2];
}
__global__ void resize_cubic_kernel(const float* __restrict__ src, int sw, int sh, float* __restrict__ dst, int dw, int dh, double scale) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= dw || y >= dh) return;
	float fx = (float)(((double)x + 0.5) * scale - 0.5), fy = (float)(((double)y + 0.5) * scale - 0.5);
	const int ix = (int)floorf(fx), iy = (int)floorf(fy);
	fx -= (float)ix; fy -= (float)iy;
	float wx[4], wy[4];
	cubic_coeffs(fx, wx); cubic_coeffs(fy, wy);
	float acc = 0.f;
	for (int j = 0; j < 4; ++j) {
		int yy = iy - 1 + j; yy = yy < 0 ? 0 : (yy > sh - 1 ? sh - 1 : yy);
		const float* S = src + (size_t)yy * sw;
		float r = 0.f;
		for (int i = 0; i < 4; ++i) {
			int xx = ix - 1 + i; xx = xx < 0 ? 0 : (xx > sw - 1 ? sw - 1 : xx);
			const float t = S[xx] * wx[i];
			r = i == 0 ? t : r + t;
		}
		const float t = r * wy[j];
		acc = j == 0 ? t : acc + t;
	}
	dst[(size_t)y * dw + x] = acc;
}

// cv::resize(src, dst, Size(), scale, scale, scale > 1 ? INTER_CUBIC : INTER_AREA) on an f32 image; dw/dh = cvRound(size * scale)
void launch_resize_gray(const float* src, int sw, int sh, float* dst, int dw, int dh, float scaleParam, hipStream_t s) {
	const dim3 block(64, 4), grid((dw + 63) / 64, (dh + 3) / 4);
	const double inv = (double)scaleParam;       // inv_scale_x = fx
	const double scale = 1.0 / inv;              // scale_x = 1 / inv_scale_x: source pixels per destination pixel
	if (scaleParam > 1.f) { hipLaunchKernelGGL(resize_cubic_kernel, grid, block, 0, s, src, sw, sh, dst, dw, dh, scale); return; }
	const int iscale = (int)scale;               // saturate_cast<int>(scale_x) of an exact integer
	if ((double)iscale == scale && iscale >= 1 && dw * iscale <= sw && dh * iscale <= sh)
		hipLaunchKernelGGL(resize_area_fast_kernel, grid, block, 0, s, src, sw, sh, dst, dw, dh, iscale);
	else
		hipLaunchKernelGGL(resize_area_kernel, grid, block, 0, s, src, sw, sh, dst, dw, dh, scale);
}

} // namespace hcmvs
