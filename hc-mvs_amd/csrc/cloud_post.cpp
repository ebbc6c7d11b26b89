/*
 * hc-mvs_amd/csrc/cloud_post.cpp -- host-side post-processing of the fused cloud.
 *
 * MVS::EstimatePointNormals (frame_main/libs/MVS/DepthMap.cpp:2221-2269, --estimate-normals 1) calls
 * CGAL::pca_estimate_normals(points, k = 16): for every point the k nearest points (the point itself among them) are fitted
 * with a plane by principal component analysis and the plane normal becomes the point normal; the reference then flips it
 * towards the camera centre of the point's first view.  CGAL is absent: the k-nearest search (uniform grid, exact) and the
 * PCA (covariance about the centroid, smallest eigenvector by Jacobi rotations) are restated here; parity unpinned, the
 * result is defined up to the eigen-solver's rounding.  Out of the GPU hot path (the default --estimate-normals 2 takes the
 * normals from fusion): OpenMP over the points like the reference's other per-point loops.
 */
#include "cloud_post.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace hcmvs {

static void smallest_eigenvector(const double cov[6], double* v) { // symmetric 3x3: xx xy xz yy yz zz
	double a[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
	double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	for (int sweep = 0; sweep < 32; ++sweep) {
		const double offd = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
		if (offd < 1e-300) break;
		for (int p = 0; p < 2; ++p)
			for (int q = p + 1; q < 3; ++q) {
				if (std::fabs(a[p][q]) < 1e-300) continue;
				const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
				const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
				for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = cs * akp - sn * akq; a[k][q] = sn * akp + cs * akq; }
				for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = cs * apk - sn * aqk; a[q][k] = sn * apk + cs * aqk; }
				for (int k = 0; k < 3; ++k) { const double ekp = e[k][p], ekq = e[k][q]; e[k][p] = cs * ekp - sn * ekq; e[k][q] = sn * ekp + cs * ekq; }
			}
	}
	int m = 0;
	for (int i = 1; i < 3; ++i) if (a[i][i] < a[m][m]) m = i;
	for (int k = 0; k < 3; ++k) v[k] = e[k][m];
}

void pca_normals(uint64_t n, const float* xyz, const double* viewC, int k, float* normal) {
	if (n == 0) return;
	// uniform grid sized for ~2 points per cell
	float lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
	for (uint64_t i = 1; i < n; ++i)
		for (int q = 0; q < 3; ++q) { lo[q] = std::min(lo[q], xyz[3 * i + q]); hi[q] = std::max(hi[q], xyz[3 * i + q]); }
	// the cloud is a surface: occupied cells ~ n / 2 when the cell edge is about sqrt(2 * area / n); estimate the area from the box
	const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
	const double area = std::max({ex * ey, ex * ez, ey * ez, 1e-30});
	const double cell = std::max(std::sqrt(2.0 * area / (double)n), 1e-12);
	auto cellOf = [&](const float* p, long long* c) { for (int q = 0; q < 3; ++q) c[q] = (long long)std::floor((p[q] - lo[q]) / cell); };
	auto keyOf = [](long long x, long long y, long long z) { return (uint64_t)(x * 73856093LL) ^ (uint64_t)(y * 19349663LL) ^ (uint64_t)(z * 83492791LL); };
	std::unordered_map<uint64_t, std::vector<uint32_t>> grid;
	grid.reserve(n);
	for (uint64_t i = 0; i < n; ++i) { long long c[3]; cellOf(xyz + 3 * i, c); grid[keyOf(c[0], c[1], c[2])].push_back((uint32_t)i); }
	const int K = (int)std::min<uint64_t>((uint64_t)k, n);
#pragma omp parallel
	{
		std::vector<std::pair<double, uint32_t>> cand;
#pragma omp for schedule(dynamic, 256)
		for (long long i = 0; i < (long long)n; ++i) {
			const float* p = xyz + 3 * i;
			long long c[3];
			cellOf(p, c);
			// grow the searched cube until the K-th nearest candidate is closer than the cube's inner margin (exact k-NN)
			for (int r = 1;; ++r) {
				cand.clear();
				for (long long z = c[2] - r; z <= c[2] + r; ++z)
					for (long long y = c[1] - r; y <= c[1] + r; ++y)
						for (long long x = c[0] - r; x <= c[0] + r; ++x) {
							auto it = grid.find(keyOf(x, y, z));
							if (it == grid.end()) continue;
							for (uint32_t j : it->second) {
								const float* q = xyz + 3 * (size_t)j;
								long long cq[3];
								cellOf(q, cq);
								if (cq[0] != x || cq[1] != y || cq[2] != z) continue; // hash collision of another cell
								const double dx = (double)q[0] - p[0], dy = (double)q[1] - p[1], dz = (double)q[2] - p[2];
								cand.emplace_back(dx * dx + dy * dy + dz * dz, j);
							}
						}
				if ((int)cand.size() >= K) {
					std::nth_element(cand.begin(), cand.begin() + (K - 1), cand.end());
					const double margin = (double)r * cell; // everything closer than r cells is inside the cube
					if (cand[K - 1].first <= margin * margin || r > 64) break;
				} else if (r > 64) break;
			}
			const int m = (int)std::min<size_t>((size_t)K, cand.size());
			std::partial_sort(cand.begin(), cand.begin() + m, cand.end());
			double mean[3] = {0, 0, 0};
			for (int t = 0; t < m; ++t) for (int q = 0; q < 3; ++q) mean[q] += xyz[3 * (size_t)cand[t].second + q];
			for (int q = 0; q < 3; ++q) mean[q] /= m;
			double cov[6] = {0, 0, 0, 0, 0, 0};
			for (int t = 0; t < m; ++t) {
				const float* q = xyz + 3 * (size_t)cand[t].second;
				const double d0 = q[0] - mean[0], d1 = q[1] - mean[1], d2 = q[2] - mean[2];
				cov[0] += d0 * d0; cov[1] += d0 * d1; cov[2] += d0 * d2; cov[3] += d1 * d1; cov[4] += d1 * d2; cov[5] += d2 * d2;
			}
			double v[3];
			smallest_eigenvector(cov, v);
			float nn[3] = {(float)v[0], (float)v[1], (float)v[2]};
			// correct the orientation: towards the camera of the first view (DepthMap.cpp:2262-2265)
			const float tc[3] = {(float)viewC[3 * i] - p[0], (float)viewC[3 * i + 1] - p[1], (float)viewC[3 * i + 2] - p[2]};
			if (nn[0] * tc[0] + nn[1] * tc[1] + nn[2] * tc[2] < 0) { nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2]; }
			normal[3 * i] = nn[0]; normal[3 * i + 1] = nn[1]; normal[3 * i + 2] = nn[2];
		}
	}
}

} // namespace hcmvs
