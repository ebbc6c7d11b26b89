/*
 * oracle/hcmvs_fuse.c -- TEST INFRASTRUCTURE (see hcmvs_oracle.h: PARITY UNPINNED).
 *
 * CPU restatement of the depth-map filter and fusion passes of HC-MVS:
 *   SD.cpp:3006-3259 DepthMapsData::FilterDepthMap   (splat neighbours into the reference view + vote)
 *   SD.cpp:3265-3495 DepthMapsData::FuseDepthMaps    (greedy, order dependent merge into a point cloud)
 *   SD.cpp:154-156 Conf2Weight, Util.inl:658-669 IsDepthSimilar, Camera.h:276-367 transforms.
 *
 * Every floating-point step is spelled out (no fused contraction, -ffp-contract=off) so that the gfx950
 * kernels, which perform the same operations in the same order, can be compared bit for bit.
 */
#include "hcmvs_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NO_ID 0xFFFFFFFFu
#define FUSE_MAX_VIEWS 32

/* Camera.h:306-320: pixel + depth -> world */
static void i2w(const hcor_depthmap* m, double x, double y, double z, double* X) {
	const double c[3] = {(x - m->K[2]) * z / m->K[0], (y - m->K[5]) * z / m->K[4], z};
	for (int i = 0; i < 3; ++i) X[i] = (m->R[0 * 3 + i] * c[0] + m->R[1 * 3 + i] * c[1] + m->R[2 * 3 + i] * c[2]) + m->C[i];
}
/* Camera.h:357-359: world -> camera */
static void w2c(const hcor_depthmap* m, const double* X, double* c) {
	const double d[3] = {X[0] - m->C[0], X[1] - m->C[1], X[2] - m->C[2]};
	for (int i = 0; i < 3; ++i) c[i] = m->R[i * 3] * d[0] + m->R[i * 3 + 1] * d[1] + m->R[i * 3 + 2] * d[2];
}
static inline int is_depth_similar(float d0, float d1, float th) { return fabsf(d0 - d1) / d0 < th; } /* Util.inl:658-669 */
static inline float conf2weight(float conf, float depth) { /* SD.cpp:154-156 */
	const float a = 1.f - conf;
	return 1.f / ((a > 0.03f ? a : 0.03f) * depth * depth);
}

/* ------------------------------------------------------------------------------------------------ */

int hcor_filter_depthmap(const hcor_depthmap* maps, uint32_t ref_id, const uint32_t* nbr, int N, int adjust,
                         int nMinViews, int nMinViewsAdjust, float fDepthDiffThreshold, float* newDepth,
                         float* newConf, uint64_t* n_processed, uint64_t* n_discarded) {
	if (N < nMinViews || N < nMinViewsAdjust) return 0; /* SD.cpp:3016-3019 */
	const hcor_depthmap* ref = &maps[ref_id];
	const int W = ref->width, H = ref->height;
	const size_t area = (size_t)W * H;
	float* dm = (float*)calloc(area * N, sizeof(float)); /* projected neighbour depths */
	float* cm = (float*)calloc(area * N, sizeof(float)); /* ... and their confidences */
	/* SD.cpp:3027-3089: project every neighbour map into the reference view, 4-pixel footprint, z-buffer */
	for (int n = 0; n < N; ++n) {
		const hcor_depthmap* nb = &maps[nbr[n]];
		float* D = dm + area * n;
		float* Cf = cm + area * n;
		for (int i = 0; i < nb->height; ++i)
			for (int j = 0; j < nb->width; ++j) {
				const float depth = nb->depth[(size_t)i * nb->width + j];
				if (depth == 0) continue;
				double X[3], c[3];
				i2w(nb, (double)j, (double)i, (double)depth, X);
				w2c(ref, X, c);
				if (c[2] <= 0) continue;
				const double ix = ref->K[2] + ref->K[0] * (c[0] / c[2]), iy = ref->K[5] + ref->K[4] * (c[1] / c[2]);
				const int fx = (int)floor(ix), fy = (int)floor(iy), cx = (int)ceil(ix), cy = (int)ceil(iy);
				const int xs[4] = {fx, fx, cx, cx}, ys[4] = {fy, cy, fy, cy};
				const float z = (float)c[2];
				for (int p = 0; p < 4; ++p) {
					if (xs[p] < 0 || ys[p] < 0 || xs[p] >= W || ys[p] >= H) continue;
					float* dr = &D[(size_t)ys[p] * W + xs[p]];
					if (*dr != 0 && *dr < z) continue;
					*dr = z;
					if (adjust) Cf[(size_t)ys[p] * W + xs[p]] = nb->conf[(size_t)i * nb->width + j];
				}
			}
	}
	const float thDepthDiff = fDepthDiffThreshold * 1.2f;
	uint64_t nProc = 0, nDisc = 0;
	if (adjust) {
		/* SD.cpp:3097-3170 */
		for (int i = 0; i < H; ++i)
			for (int j = 0; j < W; ++j) {
				const size_t idx = (size_t)i * W + j;
				const float depth = ref->depth[idx];
				if (depth == 0) { newDepth[idx] = 0; newConf[idx] = 0; continue; }
				++nProc;
				float posConf = ref->conf[idx], negConf = 0;
				float avgDepth = depth * posConf;
				unsigned nPos = 0, nNeg = 0;
				int n = N, discard = 0;
				do {
					const float d = dm[area * (--n) + idx];
					if (d == 0) {
						if (nPos + nNeg + (unsigned)n < (unsigned)nMinViews) { discard = 1; break; }
						continue;
					}
					if (is_depth_similar(depth, d, 0.12f)) { /* hard-coded in the reference, SD.cpp:3127 */
						const float c = cm[area * n + idx];
						avgDepth += d * c;
						posConf += c;
						++nPos;
					} else {
						if (depth > d) {
							negConf += cm[area * n + idx]; /* occlusion */
						} else { /* free-space violation */
							const hcor_depthmap* nb = &maps[nbr[n]];
							double X[3], c[3];
							i2w(ref, (double)j, (double)i, (double)depth, X);
							w2c(nb, X, c);
							const int x = (int)floor(nb->K[2] + nb->K[0] * (c[0] / c[2]) + .5);
							const int y = (int)floor(nb->K[5] + nb->K[4] * (c[1] / c[2]) + .5);
							if (x >= 0 && y >= 0 && x < nb->width && y < nb->height) {
								const float cc = nb->conf[(size_t)y * nb->width + x];
								negConf += (cc > 0 ? cc : cm[area * n + idx]);
							} else
								negConf += cm[area * n + idx];
						}
						++nNeg;
					}
				} while (n);
				if (!discard && nPos >= (unsigned)nMinViewsAdjust && posConf > negConf) {
					avgDepth /= posConf;
					if (ref->d_min <= avgDepth && avgDepth < ref->d_max) {
						newDepth[idx] = avgDepth; newConf[idx] = posConf - negConf;
						continue;
					}
				}
				newDepth[idx] = 0; newConf[idx] = 0; ++nDisc;
			}
	} else {
		/* SD.cpp:3171-3249 */
		const float thStrict = fDepthDiffThreshold * 0.8f;
		const unsigned nMinViewsDelta = (unsigned)nMinViews * 2; /* nMinViews*(nDeltas-2) */
		const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
		for (int i = 0; i < H; ++i)
			for (int j = 0; j < W; ++j) {
				const size_t idx = (size_t)i * W + j;
				const float depth = ref->depth[idx];
				if (depth == 0) { newDepth[idx] = 0; newConf[idx] = 0; continue; }
				++nProc;
				unsigned good = 0, views = 0;
				for (int n = N; n-- > 0;) {
					const float d = dm[area * n + idx];
					if (d > 0) { ++views; if (is_depth_similar(depth, d, thStrict)) ++good; }
				}
				if (good < (unsigned)nMinViews || good < views * 75 / 100) { newDepth[idx] = 0; newConf[idx] = 0; ++nDisc; continue; }
				good = views = 0;
				for (int q = 0; q < 4; ++q) {
					const int xx = j + dx[q], yy = i + dy[q];
					if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue; /* the reference reads out of bounds here */
					for (int n = N; n-- > 0;) {
						const float d = dm[area * n + (size_t)yy * W + xx];
						if (d > 0) { ++views; if (is_depth_similar(depth, d, thDepthDiff)) ++good; }
					}
				}
				if (good < nMinViewsDelta || good < views * 65 / 100) { newDepth[idx] = 0; newConf[idx] = 0; ++nDisc; continue; }
				newDepth[idx] = depth; newConf[idx] = ref->conf[idx];
			}
	}
	free(dm); free(cm);
	if (n_processed) *n_processed = nProc;
	if (n_discarded) *n_discarded = nDisc;
	return 1;
}

/* ------------------------------------------------------------------------------------------------ */

/* order in which the pixels of an image are visited: 0 = raster order (the reference's, SD.cpp:3355-3358), 1 = the C-ABI's
 * hcmvs_set_fuse_order(ctx, 1): ascending idx * 0x9E3779B1 (mod 2^32), a bijection of the raster index.  Test infrastructure for that
 * option; the points then come out in visiting order, not in raster order as the device writes them. */
static int g_fuse_pixel_order = 0;
void hcor_set_fuse_pixel_order(int mode) { g_fuse_pixel_order = mode; }
static const uint32_t* g_sort_key;
static int cmp_by_key(const void* a, const void* b) {
	const uint32_t ka = g_sort_key[*(const uint32_t*)a], kb = g_sort_key[*(const uint32_t*)b];
	return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

int hcor_fuse_depthmaps(hcor_depthmap* maps, int n_maps, const uint32_t* order, int n_order, int nMinViewsFuse,
                        float fDepthDiffThreshold, float fNormalDiffDeg, float depthweight, float normalweight,
                        hcor_cloud* cloud) {
	/* claim maps (SD.cpp:3313-3351 arrDepthIdx) */
	uint32_t** claim = (uint32_t**)calloc((size_t)n_maps, sizeof(uint32_t*));
	double (*P)[12] = (double (*)[12])calloc((size_t)n_maps, sizeof(double[12])); /* P = K [R | -R C] */
	for (int m = 0; m < n_maps; ++m) {
		if (!maps[m].depth) continue;
		const size_t area = (size_t)maps[m].width * maps[m].height;
		claim[m] = (uint32_t*)malloc(area * sizeof(uint32_t));
		memset(claim[m], 0xFF, area * sizeof(uint32_t));
		const double* K = maps[m].K; const double* R = maps[m].R; const double* Cc = maps[m].C;
		double t[3];
		for (int i = 0; i < 3; ++i) t[i] = -(R[i * 3] * Cc[0] + R[i * 3 + 1] * Cc[1] + R[i * 3 + 2] * Cc[2]);
		for (int i = 0; i < 3; ++i) {
			for (int j = 0; j < 3; ++j) P[m][i * 4 + j] = K[i * 3] * R[j] + K[i * 3 + 1] * R[3 + j] + K[i * 3 + 2] * R[6 + j];
			P[m][i * 4 + 3] = K[i * 3] * t[0] + K[i * 3 + 1] * t[1] + K[i * 3 + 2] * t[2];
		}
	}
	const float normalError = cosf(fNormalDiffDeg * normalweight * ((float)3.1415926535897932384626433832795 / 180.f));
	const float thDepth = fDepthDiffThreshold * depthweight;
	cloud->n_points = 0; cloud->n_depths = 0; cloud->n_view_entries = 0;
	int rc = 0;
	for (int oi = 0; oi < n_order && !rc; ++oi) {
		const uint32_t A = order[oi];
		hcor_depthmap* dA = &maps[A];
		const int W = dA->width, H = dA->height;
		const size_t areaA = (size_t)W * H;
		uint32_t* perm = NULL;
		if (g_fuse_pixel_order == 1) { /* not thread-safe (file-static key): the tests call it from one thread */
			perm = (uint32_t*)malloc(areaA * sizeof(uint32_t));
			uint32_t* key = (uint32_t*)malloc(areaA * sizeof(uint32_t));
			for (size_t k = 0; k < areaA; ++k) { perm[k] = (uint32_t)k; key[k] = (uint32_t)k * 0x9E3779B1u; }
			g_sort_key = key;
			qsort(perm, areaA, sizeof(uint32_t), cmp_by_key);
			free(key);
		}
		for (size_t kk = 0; kk < areaA && !rc; ++kk) {
			{
				const size_t idx = perm ? perm[kk] : kk;
				const int i = (int)(idx / (size_t)W), j = (int)(idx % (size_t)W);
				const float depth = dA->depth[idx];
				if (depth == 0) continue;
				++cloud->n_depths;
				if (claim[A][idx] != NO_ID) continue;
				if (cloud->xyz && cloud->n_points >= cloud->capacity) { rc = 1; break; }
				const uint32_t idPoint = (uint32_t)cloud->n_points;
				claim[A][idx] = idPoint;
				double Xw[3];
				i2w(dA, (double)j, (double)i, (double)depth, Xw);
				const float point[3] = {(float)Xw[0], (float)Xw[1], (float)Xw[2]};
				uint32_t views[FUSE_MAX_VIEWS]; size_t vpix[FUSE_MAX_VIEWS]; float vwt[FUSE_MAX_VIEWS]; int nv = 0;
				vwt[nv] = conf2weight(dA->conf[idx], depth); /* weights.emplace_back(Conf2Weight(...)), SD.cpp:3378 */
				views[nv] = A; vpix[nv] = idx; ++nv;
				double confidence = (double)vwt[0];
				float normal[3] = {0, 0, -1};
				if (dA->normal) {
					const float* nm = dA->normal + 3 * idx;
					for (int k = 0; k < 3; ++k) normal[k] = (float)(dA->R[0 * 3 + k] * (double)nm[0] + dA->R[1 * 3 + k] * (double)nm[1] + dA->R[2 * 3 + k] * (double)nm[2]);
				}
				double X[3] = {(double)point[0] * confidence, (double)point[1] * confidence, (double)point[2] * confidence};
				float Cc[3] = {0, 0, 0}, Nn[3];
				if (dA->bgr) for (int k = 0; k < 3; ++k) Cc[k] = (float)dA->bgr[3 * idx + k] * (float)confidence;
				for (int k = 0; k < 3; ++k) Nn[k] = normal[k] * (float)confidence;
				float* invalid[FUSE_MAX_VIEWS]; int ninv = 0;
				for (int q = 0; q < dA->n_neighbors; ++q) {
					const uint32_t B = dA->neighbors[q];
					if ((int)B >= n_maps || !maps[B].depth) continue;
					hcor_depthmap* dB = &maps[B];
					const double* p = P[B];
					const float ptx = (float)(p[0] * (double)point[0] + p[1] * (double)point[1] + p[2] * (double)point[2] + p[3]);
					const float pty = (float)(p[4] * (double)point[0] + p[5] * (double)point[1] + p[6] * (double)point[2] + p[7]);
					const float ptz = (float)(p[8] * (double)point[0] + p[9] * (double)point[1] + p[10] * (double)point[2] + p[11]);
					if (ptz <= 0) continue;
					const int xB = (int)floorf(ptx / ptz + .5f), yB = (int)floorf(pty / ptz + .5f);
					if (xB < 0 || yB < 0 || xB >= dB->width || yB >= dB->height) continue;
					const size_t ib = (size_t)yB * dB->width + xB;
					float* depthB = &dB->depth[ib];
					if (*depthB == 0) continue;
					if (claim[B][ib] != NO_ID) continue;
					if (is_depth_similar(ptz, *depthB, thDepth)) {
						float normalB[3] = {0, 0, -1};
						if (dB->normal) {
							const float* nm = dB->normal + 3 * ib;
							for (int k = 0; k < 3; ++k) normalB[k] = (float)(dB->R[0 * 3 + k] * (double)nm[0] + dB->R[1 * 3 + k] * (double)nm[1] + dB->R[2 * 3 + k] * (double)nm[2]);
						}
						if (normal[0] * normalB[0] + normal[1] * normalB[1] + normal[2] * normalB[2] > normalError) {
							const float confB = conf2weight(dB->conf[ib], *depthB);
							/* views.InsertSort(idxImageB): keep the list sorted by image id */
							int pos = nv;
							while (pos > 0 && views[pos - 1] > B) { views[pos] = views[pos - 1]; vpix[pos] = vpix[pos - 1]; vwt[pos] = vwt[pos - 1]; --pos; }
							views[pos] = B; vpix[pos] = ib; vwt[pos] = confB; ++nv; /* weights.InsertAt(idx, confidenceB), SD.cpp:3410 */
							claim[B][ib] = idPoint;
							double XB[3];
							i2w(dB, (double)xB, (double)yB, (double)*depthB, XB);
							for (int k = 0; k < 3; ++k) X[k] += XB[k] * (double)confB;
							if (dB->bgr) for (int k = 0; k < 3; ++k) Cc[k] += (float)dB->bgr[3 * ib + k] * confB;
							for (int k = 0; k < 3; ++k) Nn[k] += normalB[k] * confB;
							confidence += (double)confB;
							continue;
						}
					}
					if (ptz < *depthB) invalid[ninv++] = depthB; /* this estimate blocks the view of the point */
				}
				if (nv < nMinViewsFuse) {
					for (int v = 0; v < nv; ++v) claim[views[v]][vpix[v]] = NO_ID; /* SD.cpp:3426-3437 */
				} else {
					const double nrm = 1.0 / confidence;
					if (cloud->xyz) {
						float* out = cloud->xyz + 3 * cloud->n_points;
						for (int k = 0; k < 3; ++k) out[k] = (float)(X[k] * nrm);
					}
					if (cloud->bgr) for (int k = 0; k < 3; ++k) {
						const int r = (int)floorf(Cc[k] * (float)nrm + .5f);
						cloud->bgr[3 * cloud->n_points + k] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
					}
					if (cloud->normal) {
						const float n0 = Nn[0] * (float)nrm, n1 = Nn[1] * (float)nrm, n2 = Nn[2] * (float)nrm;
						const float len = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
						float* on = cloud->normal + 3 * cloud->n_points;
						on[0] = n0 / len; on[1] = n1 / len; on[2] = n2 / len;
					}
					if (cloud->n_views) cloud->n_views[cloud->n_points] = (uint32_t)nv;
					if (cloud->view_ids && cloud->n_view_entries + (uint64_t)nv <= cloud->views_capacity)
						for (int v = 0; v < nv; ++v) {
							cloud->view_ids[cloud->n_view_entries + v] = views[v];
							if (cloud->view_weights) cloud->view_weights[cloud->n_view_entries + v] = vwt[v];
						}
					cloud->n_view_entries += (uint64_t)nv;
					++cloud->n_points;
					for (int v = 0; v < ninv; ++v) *invalid[v] = 0; /* SD.cpp:3447-3449 */
				}
			}
		}
		free(perm);
	}
	if (cloud->claim_mask && (int)cloud->claim_image < n_maps && claim[cloud->claim_image]) {
		const size_t area = (size_t)maps[cloud->claim_image].width * maps[cloud->claim_image].height;
		for (size_t i = 0; i < area; ++i) cloud->claim_mask[i] = claim[cloud->claim_image][i] != NO_ID;
	}
	for (int m = 0; m < n_maps; ++m) free(claim[m]);
	free(claim); free(P);
	return rc;
}

/* MVS::EstimatePointColors, DM.cpp:2125-2161 */
void hcor_estimate_point_colors(const hcor_depthmap* maps, int n_maps, uint64_t n_points, const float* xyz, const uint32_t* n_views,
                                const uint32_t* view_ids, uint8_t* bgr) {
	uint64_t off = 0;
	for (uint64_t i = 0; i < n_points; ++i) {
		const float* X = xyz + 3 * i;
		double best = FLT_MAX;
		int img = -1;
		double Pb[12];
		for (uint32_t v = 0; v < n_views[i]; ++v) {
			const uint32_t id = view_ids[off + v];
			if ((int)id >= n_maps || !maps[id].bgr) continue;
			const hcor_depthmap* m = &maps[id];
			double t[3], P[12];
			for (int r = 0; r < 3; ++r) t[r] = -(m->R[r * 3] * m->C[0] + m->R[r * 3 + 1] * m->C[1] + m->R[r * 3 + 2] * m->C[2]);
			for (int r = 0; r < 3; ++r) {
				for (int c = 0; c < 3; ++c) P[r * 4 + c] = m->K[r * 3] * m->R[c] + m->K[r * 3 + 1] * m->R[3 + c] + m->K[r * 3 + 2] * m->R[6 + c];
				P[r * 4 + 3] = m->K[r * 3] * t[0] + m->K[r * 3 + 1] * t[1] + m->K[r * 3 + 2] * t[2];
			}
			const double d = P[8] * X[0] + P[9] * X[1] + P[10] * X[2] + P[11]; /* Camera::PointDepth */
			if (best > d) { best = d; img = (int)id; memcpy(Pb, P, sizeof Pb); }
		}
		off += n_views[i];
		uint8_t* c = bgr + 3 * i;
		c[0] = c[1] = c[2] = 255; /* Pixel8U::WHITE */
		if (img < 0) continue;
		const hcor_depthmap* m = &maps[img];
		const float qx = (float)(Pb[0] * X[0] + Pb[1] * X[1] + Pb[2] * X[2] + Pb[3]), qy = (float)(Pb[4] * X[0] + Pb[5] * X[1] + Pb[6] * X[2] + Pb[7]),
		            qz = (float)(Pb[8] * X[0] + Pb[9] * X[1] + Pb[10] * X[2] + Pb[11]);
		const float invZ = 1.f / qz;
		const float px = qx * invZ, py = qy * invZ;
		if (!(px >= 1.f && py >= 1.f && px <= (float)(m->width - 2) && py <= (float)(m->height - 2))) continue;
		const int lx = (int)px, ly = (int)py;
		const float x = px - lx, x1 = 1.f - x, y = py - ly, y1 = 1.f - y;
		const uint8_t* r0 = m->bgr + 3 * ((size_t)ly * m->width + lx);
		const uint8_t* r1 = r0 + 3 * (size_t)m->width;
		for (int k = 0; k < 3; ++k) { /* (p00*x1 + p01*x)*y1 + (p10*x1 + p11*x)*y over TPixel<uint8_t> */
			const uint8_t a = (uint8_t)((uint8_t)(x1 * r0[k]) + (uint8_t)(x * r0[3 + k]));
			const uint8_t b = (uint8_t)((uint8_t)(x1 * r1[k]) + (uint8_t)(x * r1[3 + k]));
			c[k] = (uint8_t)((uint8_t)(a * y1) + (uint8_t)(b * y));
		}
	}
}

/* ------------------------------------------------------------------------------------------------ */
/* SD.cpp:3939-3958: RemoveSmallSegments (fork version) + GapInterpolation                           */
#include "portable_math.h"
typedef struct { float (*atan2f_)(float, float); float (*acosf_)(float); float (*sinf_)(float); float (*cosf_)(float); } pf_math;
static float pf_pm_atan2(float y, float x) { return pm_atan2f(y, x); }
static float pf_pm_acos(float x) { return pm_acosf(x); }
static float pf_pm_sin(float x) { return pm_sinf(x); }
static float pf_pm_cos(float x) { return pm_cosf(x); }

/* one line (row or column) of GapInterpolation: element i of the line lives at index base + i*stride */
static uint64_t gap_line(float* dF, float* nF, float* conf, const uint8_t* gra, size_t base, size_t stride, int len, int gap, float thr,
                         const pf_math* M) {
	uint64_t filled = 0;
	unsigned count = 0;
	for (int u = 0; u < len; ++u) {
		const size_t iu = base + (size_t)u * stride;
		const float depth = dF[iu];
		if (depth <= 0) { ++count; continue; }
		if (count == 0) continue;
		if ((unsigned)u > count) {
			const int u_first = u - (int)count - 1;
			const size_t i0 = base + (size_t)u_first * stride;
			const float depthFirst = dF[i0];
			int fill = 0;
			if (count <= (unsigned)gap) {
				fill = is_depth_similar(depthFirst, depth, thr);                       /* SD.cpp:2320 */
			} else {
				const float texture0 = (float)gra[i0], texture1 = (float)gra[iu];      /* SD.cpp:2383-2388 */
				const float texture_ratio = (texture1 - texture0) / texture0;
				fill = texture_ratio <= 0.1 || is_depth_similar(depthFirst, depth, thr);
			}
			if (fill) {
				const float diff = (depth - depthFirst) / (float)(count + 1);
				float d = depthFirst;
				const float c = conf[i0] < conf[iu] ? conf[i0] : conf[iu];              /* MINF */
				float dir1[2] = {M->atan2f_(nF[3 * i0 + 1], nF[3 * i0]), M->acosf_(nF[3 * i0 + 2])};      /* Normal2Dir */
				const float dir2[2] = {M->atan2f_(nF[3 * iu + 1], nF[3 * iu]), M->acosf_(nF[3 * iu + 2])};
				const float dd[2] = {(dir2[0] - dir1[0]) / (float)(count + 1), (dir2[1] - dir1[1]) / (float)(count + 1)};
				for (int uc = u - (int)count; uc < u; ++uc) {
					const size_t ic = base + (size_t)uc * stride;
					d += diff;
					dF[ic] = d;
					dir1[0] += dd[0]; dir1[1] += dd[1];
					const float siny = M->sinf_(dir1[1]);                                /* Dir2Normal */
					nF[3 * ic] = M->cosf_(dir1[0]) * siny; nF[3 * ic + 1] = M->sinf_(dir1[0]) * siny; nF[3 * ic + 2] = M->cosf_(dir1[1]);
					conf[ic] = c;
					++filled;
				}
			}
		}
		count = 0;
	}
	return filled;
}

int hcor_postfilter(hcor_depthmap* maps, int n_maps, uint32_t id, const uint8_t* gra, const uint32_t* order, int n_order, int nMinViewsFuse,
                    float fDepthDiffThreshold, float fNormalDiffDeg, int gap, int mode, uint64_t* n_filled) {
	if ((int)id >= n_maps || !maps[id].depth || !maps[id].normal) return 1;
	hcor_depthmap* A = &maps[id];
	const int W = A->width, H = A->height;
	const size_t area = (size_t)W * H;
	const pf_math libm = {atan2f, acosf, sinf, cosf}, pm = {pf_pm_atan2, pf_pm_acos, pf_pm_sin, pf_pm_cos};
	const pf_math* M = mode == HCOR_ARITH_DEVICE ? &pm : &libm;
	/* RemoveSmallSegments, fork version: the whole fusion, then the mask of this image (SD.cpp:2048-2275) */
	hcor_cloud cl;
	memset(&cl, 0, sizeof cl);
	cl.claim_image = id;
	cl.claim_mask = (uint8_t*)calloc(area, 1);
	hcor_fuse_depthmaps(maps, n_maps, order, n_order, nMinViewsFuse, fDepthDiffThreshold, fNormalDiffDeg, 1.f, 1.f, &cl); /* SD.cpp:2083, 2177: unweighted */
	float* dF = (float*)malloc(area * sizeof(float));
	float* nF = (float*)malloc(area * 3 * sizeof(float));
	for (size_t i = 0; i < area; ++i) {
		const int on = cl.claim_mask[i];
		dF[i] = on ? A->depth[i] : 0.f;
		for (int k = 0; k < 3; ++k) nF[3 * i + k] = on ? A->normal[3 * i + k] : 0.f;
	}
	free(cl.claim_mask);
	/* GapInterpolation (SD.cpp:2280-3001): rows, then columns */
	const float thr = fDepthDiffThreshold * 2.5f;
	uint64_t filled = 0;
	for (int v = 0; v < H; ++v) filled += gap_line(dF, nF, (float*)A->conf, gra, (size_t)v * W, 1, W, gap, thr, M);
	for (int u = 0; u < W; ++u) filled += gap_line(dF, nF, (float*)A->conf, gra, (size_t)u, (size_t)W, H, gap, thr, M);
	/* SD.cpp:2989-3000 */
	for (size_t i = 0; i < area; ++i) {
		if (dF[i] > 0) A->depth[i] = dF[i];
		if (nF[3 * i] != 0 || nF[3 * i + 1] != 0 || nF[3 * i + 2] != 0) { float* n = (float*)A->normal + 3 * i; n[0] = nF[3 * i]; n[1] = nF[3 * i + 1]; n[2] = nF[3 * i + 2]; }
	}
	free(dF); free(nF);
	if (n_filled) *n_filled = filled;
	return 0;
}
