/*
 * hc-mvs_amd/csrc/fuse_device.h -- small device helpers shared by the filter / fuse kernels (fuse_kernels.hip) and the incremental
 * fusion of the post-filter chain (pf_kernels.hip): camera transforms in the reference's double precision (Camera.h:276-367), the
 * similarity test and weights of FuseDepthMaps (SceneDensify.cpp:154-156, Util.inl:658-669), agent-scope accessors.
 */
#ifndef HCMVS_FUSE_DEVICE_H
#define HCMVS_FUSE_DEVICE_H

#include "fuse_common.h"

namespace hcmvs {

__device__ __forceinline__ void i2w(const DevMap& m, double x, double y, double z, double* X) { // Camera.h:306-320
	const double c0 = (x - m.K[2]) * z / m.K[0], c1 = (y - m.K[5]) * z / m.K[4], c2 = z;
#pragma unroll
	for (int i = 0; i < 3; ++i) X[i] = (m.R[0 * 3 + i] * c0 + m.R[1 * 3 + i] * c1 + m.R[2 * 3 + i] * c2) + m.C[i];
}
__device__ __forceinline__ void w2c(const DevMap& m, const double* X, double* c) { // Camera.h:357-359
	const double d0 = X[0] - m.C[0], d1 = X[1] - m.C[1], d2 = X[2] - m.C[2];
#pragma unroll
	for (int i = 0; i < 3; ++i) c[i] = m.R[i * 3] * d0 + m.R[i * 3 + 1] * d1 + m.R[i * 3 + 2] * d2;
}
__device__ __forceinline__ bool is_depth_similar(float d0, float d1, float th) { return fabsf(d0 - d1) / d0 < th; }
__device__ __forceinline__ float conf2weight(float conf, float depth) { // SceneDensify.cpp:154-156
	const float a = 1.f - conf;
	return 1.f / ((a > 0.03f ? a : 0.03f) * depth * depth);
}

// target of pixel `point` in neighbour map m: returns false when it projects behind / outside (SceneDensify.cpp:3387-3393)
__device__ __forceinline__ bool project_target(const double* p, int w, int h, const float* point, float& ptz, int& ib, int& xB, int& yB) {
	const float ptx = (float)(p[0] * (double)point[0] + p[1] * (double)point[1] + p[2] * (double)point[2] + p[3]);
	const float pty = (float)(p[4] * (double)point[0] + p[5] * (double)point[1] + p[6] * (double)point[2] + p[7]);
	ptz = (float)(p[8] * (double)point[0] + p[9] * (double)point[1] + p[10] * (double)point[2] + p[11]);
	if (ptz <= 0.f) return false;
	xB = (int)floorf(ptx / ptz + .5f); yB = (int)floorf(pty / ptz + .5f);
	if (xB < 0 || yB < 0 || xB >= w || yB >= h) return false;
	ib = yB * w + xB;
	return true;
}
__device__ __forceinline__ bool project_target(const DevMap& m, const float* point, float& ptz, int& ib, int& xB, int& yB) {
	return project_target(m.P, m.w, m.h, point, ptz, ib, xB, yB);
}
__device__ __forceinline__ void pixel_point(const DevMap& A, int idx, float depth, float* point) {
	double Xw[3];
	i2w(A, (double)(idx % A.w), (double)(idx / A.w), (double)depth, Xw);
	point[0] = (float)Xw[0]; point[1] = (float)Xw[1]; point[2] = (float)Xw[2];
}

#define FS_SCOPE __HIP_MEMORY_SCOPE_AGENT
#define FS_NOT_DONE 0xFFFFFFFFu
typedef __attribute__((address_space(1))) uint32_t* g_u32p;
__device__ __forceinline__ uint32_t ld_u32(const uint32_t* p) { return __hip_atomic_load((g_u32p)p, __ATOMIC_RELAXED, FS_SCOPE); }
__device__ __forceinline__ void st_u32(uint32_t* p, uint32_t v) { __hip_atomic_store((g_u32p)p, v, __ATOMIC_RELAXED, FS_SCOPE); }
__device__ __forceinline__ uint8_t ld_u8(const uint8_t* p) { return __hip_atomic_load((__attribute__((address_space(1))) uint8_t*)p, __ATOMIC_RELAXED, FS_SCOPE); }
__device__ __forceinline__ void st_u8(uint8_t* p, uint8_t v) { __hip_atomic_store((__attribute__((address_space(1))) uint8_t*)p, v, __ATOMIC_RELAXED, FS_SCOPE); }
__device__ __forceinline__ float ld_f32(const float* p) { return __uint_as_float(ld_u32((const uint32_t*)p)); }
__device__ __forceinline__ void st_f32(float* p, float v) { st_u32((uint32_t*)p, __float_as_uint(v)); }

__device__ __forceinline__ void list_append(bool pred, int value, uint32_t* list, uint32_t* count) { // one atomic per wave
	const unsigned long long m = __ballot(pred);
	if (!pred) return;
	const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
	uint32_t base = 0;
	if (lane == leader) base = atomicAdd(count, (uint32_t)__builtin_popcountll(m));
	base = __shfl(base, leader, 64);
	st_u32(&list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))], (uint32_t)value);
}

__device__ __forceinline__ void rotate_normal(const double* R, const float* nm, float* out) { // camera -> world, SceneDensify.cpp:3384
#pragma unroll
	for (int k = 0; k < 3; ++k) out[k] = (float)(R[0 * 3 + k] * (double)nm[0] + R[1 * 3 + k] * (double)nm[1] + R[2 * 3 + k] * (double)nm[2]);
}
__device__ __forceinline__ void rotate_normal(const DevMap& M, const float* nm, float* out) { rotate_normal(M.R, nm, out); }


} // namespace hcmvs
#endif
