/*
 * hc-mvs_amd/csrc/cloud_kernels.hip -- post-processing of the fused cloud on the device.
 *
 * MVS::EstimatePointNormals (frame_main/libs/MVS/DepthMap.cpp:2221-2269, --estimate-normals 1) calls
 * CGAL::pca_estimate_normals(points, k = 16): for every point the k nearest points (the point itself among them) are fitted
 * with a plane by principal component analysis and the plane normal becomes the point normal; the reference then flips it
 * towards the camera centre of the point's first view.  CGAL is absent: the k-nearest search and the PCA (covariance about
 * the centroid, smallest eigenvector by Jacobi rotations, double precision) are restated here; parity unpinned, the result
 * is defined up to the eigen-solver's rounding and ties in the k-th distance.
 *
 * Layout: the points are binned into a uniform grid (cell edge ~ sqrt(k/2 * surface area / n): a cloud is a surface, so about
 * k/2 points per occupied cell and the k nearest of a surface point lie within one cell of it) and sorted by the Morton code
 * of their cell (hipcub radix sort): a cell of ANY octree level -- 2^L cells on a side -- is then one contiguous range of the
 * sorted array, found by two binary searches.  One thread per point, in sorted order so that neighbouring threads search the
 * same ranges: level 0 looks at the 3 x 3 x 3 cells around the point, and as long as the k-th nearest candidate is farther
 * than one cell edge of the level (the margin inside which the block is complete) the search moves one level up -- exact
 * k-NN for every point, surface or outlier, at a cost that grows with the log of the distance to the k-th neighbour instead
 * of its cube.  The k best are kept sorted in registers.  HBM/L2-bound on the gathers of the candidate points.
 */
#include "cloud_kernels.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace hcmvs {

namespace {

constexpr int kMaxK = 32;
static const dim3 kGrid(2048), kBlock(256);

struct Grid {
	float lo[3];
	double cell;
	long long dim[3];
};

__device__ __forceinline__ void cell_of(const Grid& g, const float* p, long long* c) {
#pragma unroll
	for (int q = 0; q < 3; ++q) {
		long long v = (long long)floor(((double)p[q] - (double)g.lo[q]) / g.cell);
		c[q] = v < 0 ? 0 : (v >= g.dim[q] ? g.dim[q] - 1 : v);
	}
}
__device__ __forceinline__ unsigned long long spread3(unsigned long long v) { // 21 bits -> every third bit
	v &= 0x1fffffull;
	v = (v | v << 32) & 0x1f00000000ffffull;
	v = (v | v << 16) & 0x1f0000ff0000ffull;
	v = (v | v << 8) & 0x100f00f00f00f00full;
	v = (v | v << 4) & 0x10c30c30c30c30c3ull;
	v = (v | v << 2) & 0x1249249249249249ull;
	return v;
}
__device__ __forceinline__ unsigned long long morton3(long long x, long long y, long long z) {
	return spread3((unsigned long long)x) | spread3((unsigned long long)y) << 1 | spread3((unsigned long long)z) << 2;
}
__device__ __forceinline__ unsigned long long lower_bound(const unsigned long long* keys, unsigned long long n, unsigned long long v) {
	unsigned long long lo = 0, hi = n;
	while (lo < hi) { const unsigned long long mid = (lo + hi) >> 1; if (keys[mid] < v) lo = mid + 1; else hi = mid; }
	return lo;
}

// ordered-int encoding of a float for atomicMin / atomicMax
__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }

__global__ void bbox_kernel(unsigned long long n, const float* xyz, int* box) { // box: min x y z | max x y z, ordered ints
	int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
#pragma unroll
		for (int q = 0; q < 3; ++q) { const int v = f2ord(xyz[3 * i + q]); lo[q] = v < lo[q] ? v : lo[q]; hi[q] = v > hi[q] ? v : hi[q]; }
#pragma unroll
	for (int q = 0; q < 3; ++q) {
		for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(lo[q], o, 64), b = __shfl_xor(hi[q], o, 64); lo[q] = a < lo[q] ? a : lo[q]; hi[q] = b > hi[q] ? b : hi[q]; }
		if ((threadIdx.x & 63) == 0) { atomicMin(&box[q], lo[q]); atomicMax(&box[3 + q], hi[q]); }
	}
}
__global__ void keys_kernel(unsigned long long n, const float* xyz, Grid g, unsigned long long* keys, uint32_t* idx) {
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
		long long c[3];
		cell_of(g, xyz + 3 * i, c);
		keys[i] = morton3(c[0], c[1], c[2]);
		idx[i] = (uint32_t)i;
	}
}
__global__ void gather_kernel(unsigned long long n, const float* xyz, const uint32_t* idx, float4* sorted) {
	for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * blockDim.x) {
		const uint32_t i = idx[j];
		sorted[j] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __uint_as_float(i));
	}
}
__device__ void smallest_eigenvector(const double cov[6], double* v) { // symmetric 3x3: xx xy xz yy yz zz
	double a[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
	double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	for (int sweep = 0; sweep < 32; ++sweep) {
		const double offd = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
		if (offd < 1e-300) break;
#pragma unroll
		for (int p = 0; p < 2; ++p)
#pragma unroll
			for (int q = p + 1; q < 3; ++q) {
				if (fabs(a[p][q]) < 1e-300) continue;
				const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
				const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
				for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq; }
#pragma unroll
				for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk; }
#pragma unroll
				for (int k = 0; k < 3; ++k) { const double ekp = e[k][p], ekq = e[k][q]; e[k][p] = cs * ekp - sn * ekq; e[k][q] = sn * ekp + cs * ekq; }
			}
	}
	const int m = a[1][1] < a[0][0] ? (a[2][2] < a[1][1] ? 2 : 1) : (a[2][2] < a[0][0] ? 2 : 0);
#pragma unroll
	for (int k = 0; k < 3; ++k) v[k] = m == 0 ? e[k][0] : (m == 1 ? e[k][1] : e[k][2]);
}

// K: compile-time capacity of the candidate list (k <= K)
template <int K>
__global__ __launch_bounds__(256) void pca_normals_kernel(unsigned long long n, const float4* sorted, const unsigned long long* keys, const uint32_t* posOf, Grid g,
                                                          int k, const uint32_t* firstView, const double* viewC, float* normal) {
	for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * blockDim.x) {
		const float4 me = sorted[j];
		const float p[3] = {me.x, me.y, me.z};
		const uint32_t self = __float_as_uint(me.w); // original index
		long long c[3];
		cell_of(g, p, c);
		double bd[K]; uint32_t bi[K]; // the best candidates so far, ascending (distance, index)
		int cnt = 0;
		for (int L = 0;; ++L) { // octree level: cells of 2^L grid cells on a side
			cnt = 0;
#pragma unroll
			for (int t = 0; t < K; ++t) { bd[t] = 1e300; bi[t] = 0xFFFFFFFFu; }
			const long long X = c[0] >> L, Y = c[1] >> L, Z = c[2] >> L;
			const long long DX = (g.dim[0] - 1) >> L, DY = (g.dim[1] - 1) >> L, DZ = (g.dim[2] - 1) >> L; // last cell of the level per axis
			const long long x0 = X > 0 ? X - 1 : 0, x1 = X < DX ? X + 1 : DX, y0 = Y > 0 ? Y - 1 : 0, y1 = Y < DY ? Y + 1 : DY, z0 = Z > 0 ? Z - 1 : 0, z1 = Z < DZ ? Z + 1 : DZ;
			for (long long z = z0; z <= z1; ++z)
				for (long long y = y0; y <= y1; ++y)
					for (long long x = x0; x <= x1; ++x) {
						const unsigned long long mc = morton3(x, y, z);
						const unsigned long long s0 = lower_bound(keys, n, mc << (3 * L)), s1 = lower_bound(keys, n, (mc + 1ull) << (3 * L));
						for (unsigned long long s = s0; s < s1; ++s) {
							const float4 q = sorted[s];
							const double dx = (double)q.x - p[0], dy = (double)q.y - p[1], dz = (double)q.z - p[2];
							const double d = dx * dx + dy * dy + dz * dz;
							const uint32_t qi = __float_as_uint(q.w);
							++cnt;
							if (!(d < bd[K - 1] || (d == bd[K - 1] && qi < bi[K - 1]))) continue;
							// sorted insertion with static indices: the newcomer bubbles up from the last place
							bool placed = false;
#pragma unroll
							for (int t = K - 1; t >= 0; --t) {
								const bool before = t > 0 && (d < bd[t > 0 ? t - 1 : 0] || (d == bd[t > 0 ? t - 1 : 0] && qi < bi[t > 0 ? t - 1 : 0]));
								if (!placed) {
									if (before) { bd[t] = bd[t > 0 ? t - 1 : 0]; bi[t] = bi[t > 0 ? t - 1 : 0]; }
									else { bd[t] = d; bi[t] = qi; placed = true; }
								}
							}
						}
					}
			if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == DX && y1 == DY && z1 == DZ) break; // the block was the whole grid
			if (cnt >= k) {
				double kth = bd[0]; // the k-th nearest (k <= K): bd is ascending
#pragma unroll
				for (int t = 0; t < K; ++t) if (t == k - 1) kth = bd[t];
				const double margin = g.cell * (double)(1ll << L); // everything closer than one cell of this level is inside the block
				if (kth <= margin * margin) break;
			}
		}
		const int m = cnt < k ? cnt : k;
		// centroid and covariance of the m nearest, summed in ascending (distance, index) order; bi holds original indices
		// (the tie-break of the order), posOf maps them to positions in the sorted array
		double mean[3] = {0, 0, 0}, cov[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
		for (int t = 0; t < K; ++t) if (t < m) { const float4 q = sorted[posOf[bi[t]]]; mean[0] += (double)q.x; mean[1] += (double)q.y; mean[2] += (double)q.z; }
		for (int q = 0; q < 3; ++q) mean[q] /= (double)m;
#pragma unroll
		for (int t = 0; t < K; ++t) if (t < m) {
			const float4 q = sorted[posOf[bi[t]]];
			const double d0 = (double)q.x - mean[0], d1 = (double)q.y - mean[1], d2 = (double)q.z - mean[2];
			cov[0] += d0 * d0; cov[1] += d0 * d1; cov[2] += d0 * d2; cov[3] += d1 * d1; cov[4] += d1 * d2; cov[5] += d2 * d2;
		}
		double v[3];
		smallest_eigenvector(cov, v);
		float nn[3] = {(float)v[0], (float)v[1], (float)v[2]};
		// correct the orientation: towards the camera of the first view (DepthMap.cpp:2262-2265)
		const double* C = viewC + 3 * (size_t)firstView[self];
		const float tc[3] = {(float)C[0] - p[0], (float)C[1] - p[1], (float)C[2] - p[2]};
		if (nn[0] * tc[0] + nn[1] * tc[1] + nn[2] * tc[2] < 0) { nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2]; }
		normal[3 * (size_t)self] = nn[0]; normal[3 * (size_t)self + 1] = nn[1]; normal[3 * (size_t)self + 2] = nn[2];
	}
}
__global__ void positions_kernel(unsigned long long n, const uint32_t* idx, uint32_t* posOf) {
	for (unsigned long long j = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * blockDim.x) posOf[idx[j]] = (uint32_t)j;
}
} // namespace

int pca_normals_device(unsigned long long n, const float* hXyz, const uint32_t* hFirstView, const double* hViewC, size_t nViews, int k,
                       float* hNormal, hipStream_t s, std::string& err) {
	if (n == 0) return 0;
	if (k > kMaxK) { err = "estimate_point_normals: k above 32"; return 1; }
	if (n >= 0x7FFFFFFFull) { err = "estimate_point_normals: more than 2^31 - 1 points"; return 1; }
	size_t sortBytes = 0;
	(void)hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
	                                         (uint32_t*)nullptr, (int)n);
	size_t off = 0;
	auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
	const size_t oXyz = carve(n * 12), oKeys = carve(n * 8), oKeys2 = carve(n * 8), oIdx = carve(n * 4), oIdx2 = carve(n * 4), oSorted = carve(n * 16),
	             oPos = carve(n * 4), oFirst = carve(n * 4), oC = carve(nViews * 24), oN = carve(n * 12), oBox = carve(64),
	             oSort = carve(sortBytes);
	char* b = nullptr;
	if (hipMalloc(&b, off) != hipSuccess) { err = "estimate_point_normals: out of device memory"; return 2; }
	auto fail = [&](const char* what) { err = what; (void)hipFree(b); return 2; };
	float* dXyz = (float*)(b + oXyz);
	unsigned long long *keys = (unsigned long long*)(b + oKeys), *keys2 = (unsigned long long*)(b + oKeys2);
	uint32_t *idx = (uint32_t*)(b + oIdx), *idx2 = (uint32_t*)(b + oIdx2), *first = (uint32_t*)(b + oFirst);
	float4* sorted = (float4*)(b + oSorted);
	uint32_t* posOf = (uint32_t*)(b + oPos);
	double* dC = (double*)(b + oC);
	float* dN = (float*)(b + oN);
	int* box = (int*)(b + oBox);
	if (hipMemcpyAsync(dXyz, hXyz, n * 12, hipMemcpyHostToDevice, s) != hipSuccess || hipMemcpyAsync(first, hFirstView, n * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
	    hipMemcpyAsync(dC, hViewC, nViews * 24, hipMemcpyHostToDevice, s) != hipSuccess)
		return fail("estimate_point_normals: upload failed");
	const int boxInit[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
	if (hipMemcpyAsync(box, boxInit, sizeof boxInit, hipMemcpyHostToDevice, s) != hipSuccess) return fail("estimate_point_normals: upload failed");
	hipLaunchKernelGGL(bbox_kernel, kGrid, kBlock, 0, s, n, dXyz, box);
	int hb[6];
	if (hipMemcpyAsync(hb, box, sizeof hb, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail("estimate_point_normals: bounding box failed");
	auto ord2f = [](int v) { const int i = v >= 0 ? v : v ^ 0x7fffffff; float f; memcpy(&f, &i, 4); return f; };
	Grid g;
	double ext[3];
	for (int q = 0; q < 3; ++q) { g.lo[q] = ord2f(hb[q]); ext[q] = (double)ord2f(hb[3 + q]) - (double)g.lo[q]; }
	// the cloud is a surface: ~k/2 points per occupied cell when the cell edge is about sqrt(k/2 * area / n); area from the box
	const double area = std::max({ext[0] * ext[1], ext[0] * ext[2], ext[1] * ext[2], 1e-30});
	g.cell = std::max(std::sqrt(0.5 * (double)k * area / (double)n), 1e-12);
	const double longest = std::max({ext[0], ext[1], ext[2]});
	if (longest / g.cell > 1048575.0) g.cell = longest / 1048575.0; // 20 bits per axis in the cell key
	for (int q = 0; q < 3; ++q) g.dim[q] = (long long)std::floor(ext[q] / g.cell) + 1;
	hipLaunchKernelGGL(keys_kernel, kGrid, kBlock, 0, s, n, dXyz, g, keys, idx);
	if (hipcub::DeviceRadixSort::SortPairs(b + oSort, sortBytes, keys, keys2, idx, idx2, (int)n, 0, 64, s) != hipSuccess) return fail("estimate_point_normals: sort failed");
	hipLaunchKernelGGL(gather_kernel, kGrid, kBlock, 0, s, n, dXyz, idx2, sorted);
	hipLaunchKernelGGL(positions_kernel, kGrid, kBlock, 0, s, n, idx2, posOf);
	if (k <= 16) hipLaunchKernelGGL(pca_normals_kernel<16>, kGrid, kBlock, 0, s, n, sorted, keys2, posOf, g, k, first, dC, dN);
	else hipLaunchKernelGGL(pca_normals_kernel<32>, kGrid, kBlock, 0, s, n, sorted, keys2, posOf, g, k, first, dC, dN);
	if (hipGetLastError() != hipSuccess) return fail("estimate_point_normals: launch failed");
	if (hipMemcpyAsync(hNormal, dN, n * 12, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail("estimate_point_normals: device failure");
	(void)hipFree(b);
	return 0;
}

} // namespace hcmvs
