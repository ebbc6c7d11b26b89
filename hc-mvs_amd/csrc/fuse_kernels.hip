/*
 * hc-mvs_amd/csrc/fuse_kernels.hip -- gfx950 kernels of the depth-map filter and fusion passes.
 *
 * What the reference computes (paths under /root/reference/frame_main/libs/MVS/):
 *   FilterDepthMap  SceneDensify.cpp:3006-3259  splat every neighbour map into the reference view
 *                                               (4-pixel footprint, z-buffer), then a per-pixel vote
 *   FuseDepthMaps   SceneDensify.cpp:3265-3495  single-threaded greedy merge: images in order, pixels in
 *                                               raster order; a pixel claims agreeing pixels of its
 *                                               neighbour maps and zeroes the estimates it occludes
 *
 * How it is mapped to the GPU:
 *   - Filter: the z-buffer is a 64-bit atomicMin on (depth bits, ~sequence number), which reproduces the
 *     sequential rule "nearest wins, the later writer wins ties" exactly; the vote is embarrassingly
 *     parallel.  HBM-bound integer/float work, one thread per pixel, coalesced over the reference map.
 *   - Fuse: a pixel of image A interacts with other pixels of A only through the neighbour pixels it
 *     projects onto (its targets).  The pixels that project onto one neighbour pixel are listed per target
 *     (a linked list of "bidders", pushed with one atomicExch as fuse_begin_kernel classifies the pair).  Whether a pixel becomes a point depends on which of its targets the
 *     pixels BEFORE it (raster order) have left available -- a well-founded recursion with exactly one
 *     solution, which is found by iteration: every pixel evaluated in parallel under the assumption that all
 *     the others become points, then only the pixels whose inputs changed, until nothing changes (the settle
 *     iteration below; a handful of parallel steps where a walk of the dependence graph takes hundreds to
 *     thousands of sequential hops).  The decisions are the sequential algorithm's, so the cloud is identical
 *     to the sequential one, point order included (ordered compaction).
 *     "Claimed" (SceneDensify.cpp:3313 arrDepthIdx != NO_ID) is the SIGN of the depth while a fusion runs: an estimate that became
 *     part of a point holds -depth, an invalidated one 0, a free one +depth -- one 4-byte load tells a pixel everything about a
 *     target, and there are no claim maps to allocate and reset.  launch_unclaim() takes the signs off again when the fusion is over.
 *
 * Double precision follows the reference (Point3 = double); no contraction (-ffp-contract=off).
 */
#include "fuse_common.h"
#include "fuse_device.h"
#include "pm_math.h"

#include <cstdlib>

#include <hipcub/hipcub.hpp>

namespace hcmvs {

static const dim3 kGrid(2048), kBlock(256);
#define NO_ID 0xFFFFFFFFu

// ------------------------------------------------------------------------------------------------------
// filter

__global__ void filter_splat_kernel(DevMap ref, DevMap nb, unsigned long long* key) {
	const int n = nb.w * nb.h;
	for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
		const float depth = nb.depth[s];
		if (depth == 0.f) continue;
		const int j = s % nb.w, i = s / nb.w;
		double X[3], c[3];
		i2w(nb, (double)j, (double)i, (double)depth, X);
		w2c(ref, X, c);
		if (c[2] <= 0) continue;
		const double ix = ref.K[2] + ref.K[0] * (c[0] / c[2]), iy = ref.K[5] + ref.K[4] * (c[1] / c[2]);
		const int fx = (int)floor(ix), fy = (int)floor(iy), cx = (int)ceil(ix), cy = (int)ceil(iy);
		const int xs[4] = {fx, fx, cx, cx}, ys[4] = {fy, cy, fy, cy};
		const float z = (float)c[2];
#pragma unroll
		for (int p = 0; p < 4; ++p) {
			if (xs[p] < 0 || ys[p] < 0 || xs[p] >= ref.w || ys[p] >= ref.h) continue;
			// nearest depth wins; among equal depths the later (source raster, footprint) writer wins
			const unsigned long long k = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(0xFFFFFFFFu - ((unsigned)s * 4u + (unsigned)p));
			atomicMin(&key[(size_t)ys[p] * ref.w + xs[p]], k);
		}
	}
}

__device__ __forceinline__ float key_depth(unsigned long long k) { return k == ~0ull ? 0.f : __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ float key_conf(unsigned long long k, const float* conf) {
	return k == ~0ull ? 0.f : conf[(0xFFFFFFFFu - (unsigned)k) >> 2];
}

__global__ void filter_vote_kernel(DevMap ref, const DevMap* nbs, int N, const unsigned long long* keys, int adjust, int nMinViews,
                                   int nMinViewsAdjust, float fDepthDiffThreshold, float* newDepth, float* newConf,
                                   unsigned long long* counters) {
	const int W = ref.w, H = ref.h;
	const size_t area = (size_t)W * H;
	const float thDepthDiff = fDepthDiffThreshold * 1.2f;
	unsigned nProc = 0, nDisc = 0;
	for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < W * H; idx += gridDim.x * blockDim.x) {
		const int j = idx % W, i = idx / W;
		const float depth = ref.depth[idx];
		if (depth == 0.f) { newDepth[idx] = 0.f; newConf[idx] = 0.f; continue; }
		++nProc;
		if (adjust) { // SceneDensify.cpp:3097-3170
			float posConf = ref.conf[idx], negConf = 0.f;
			float avgDepth = depth * posConf;
			unsigned nPos = 0, nNeg = 0;
			int n = N;
			bool discard = false;
			do {
				--n;
				const unsigned long long k = keys[area * n + idx];
				const float d = key_depth(k);
				if (d == 0.f) {
					if (nPos + nNeg + (unsigned)n < (unsigned)nMinViews) { discard = true; break; }
					continue;
				}
				const float cproj = key_conf(k, nbs[n].conf);
				if (is_depth_similar(depth, d, 0.12f)) {
					avgDepth += d * cproj;
					posConf += cproj;
					++nPos;
				} else {
					if (depth > d) {
						negConf += cproj;
					} else {
						const DevMap& nb = nbs[n];
						double X[3], c[3];
						i2w(ref, (double)j, (double)i, (double)depth, X);
						w2c(nb, X, c);
						const int x = (int)floor(nb.K[2] + nb.K[0] * (c[0] / c[2]) + .5);
						const int y = (int)floor(nb.K[5] + nb.K[4] * (c[1] / c[2]) + .5);
						if (x >= 0 && y >= 0 && x < nb.w && y < nb.h) {
							const float cc = nb.conf[(size_t)y * nb.w + x];
							negConf += (cc > 0.f ? cc : cproj);
						} else
							negConf += cproj;
					}
					++nNeg;
				}
			} while (n);
			bool keep = false;
			if (!discard && nPos >= (unsigned)nMinViewsAdjust && posConf > negConf) {
				avgDepth /= posConf;
				if (ref.dMin <= avgDepth && avgDepth < ref.dMax) { newDepth[idx] = avgDepth; newConf[idx] = posConf - negConf; keep = true; }
			}
			if (!keep) { newDepth[idx] = 0.f; newConf[idx] = 0.f; ++nDisc; }
		} else { // SceneDensify.cpp:3171-3249
			const float thStrict = fDepthDiffThreshold * 0.8f;
			const unsigned nMinViewsDelta = (unsigned)nMinViews * 2u;
			unsigned good = 0, views = 0;
			for (int n = N; n-- > 0;) {
				const float d = key_depth(keys[area * n + idx]);
				if (d > 0.f) { ++views; if (is_depth_similar(depth, d, thStrict)) ++good; }
			}
			if (good < (unsigned)nMinViews || good < views * 75u / 100u) { newDepth[idx] = 0.f; newConf[idx] = 0.f; ++nDisc; continue; }
			good = views = 0;
			const int dx[4] = {-1, 1, 0, 0}, dy[4] = {0, 0, -1, 1};
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int xx = j + dx[q], yy = i + dy[q];
				if (xx < 0 || yy < 0 || xx >= W || yy >= H) continue;
				for (int n = N; n-- > 0;) {
					const float d = key_depth(keys[area * n + (size_t)yy * W + xx]);
					if (d > 0.f) { ++views; if (is_depth_similar(depth, d, thDepthDiff)) ++good; }
				}
			}
			if (good < nMinViewsDelta || good < views * 65u / 100u) { newDepth[idx] = 0.f; newConf[idx] = 0.f; ++nDisc; continue; }
			newDepth[idx] = depth; newConf[idx] = ref.conf[idx];
		}
	}
	if (nProc) atomicAdd(&counters[0], (unsigned long long)nProc);
	if (nDisc) atomicAdd(&counters[1], (unsigned long long)nDisc);
}

// ------------------------------------------------------------------------------------------------------
// fuse

// ---- the image pass -----------------------------------------------------------------------------------------------
// Everything another workgroup may have written in this launch (round stamps, claims, depths, the lists) is accessed
// with agent-scope (sc1) atomics, which reach past the per-XCD L2; every wave drains its stores before the barrier.

// A stored target: the neighbour's pixel index in the low bits, above them what the pixel can do to it.  A target a pixel
// can neither merge with nor invalidate (projects behind it, or the neighbour's pixel is empty or already part of a point
// when the pass begins -- neither ever changes back) is not stored at all: it cannot influence anybody.
constexpr int kTargetShift = 29;
constexpr int32_t kTargetIndexMask = (1 << kTargetShift) - 1;
constexpr int kTargetMerge = 1, kTargetInFront = 2;

// pending pixels of A (valid depth, not yet claimed by an earlier image) -> first candidate list, their targets and the
// per-target counts.  Everything about a (pixel, target) pair that does not depend on the order of the pass is settled
// here, in parallel: where the pixel projects, and whether it would merge with the neighbour's estimate there (similar
// depth and normal, SceneDensify.cpp:3400-3404) or lies in front of it (:3418).  The pass itself only has to look at what
// is left of the target when the pixel's turn comes.
__global__ void fuse_begin_kernel(DevMap A, const DevMap* maps, FuseTables tb, uint32_t* pending, uint32_t* roundCnt, uint8_t* flag,
                                  unsigned long long* counters, float thDepth, float normalError) {
	const int n = A.w * A.h;
	// the neighbours' cameras and map pointers, once per workgroup (every thread needs all of them for every pixel)
	__shared__ double sP[kFuseMaxViews - 1][12], sR[kFuseMaxViews - 1][9];
	__shared__ const float* sDepth[kFuseMaxViews - 1];
	__shared__ const float* sNormal[kFuseMaxViews - 1];
	__shared__ int sW[kFuseMaxViews - 1], sH[kFuseMaxViews - 1];
	for (int i = threadIdx.x; i < A.nNeighbors * 12; i += blockDim.x) sP[i / 12][i % 12] = maps[A.neighbors[i / 12]].P[i % 12];
	for (int i = threadIdx.x; i < A.nNeighbors * 9; i += blockDim.x) sR[i / 9][i % 9] = maps[A.neighbors[i / 9]].R[i % 9];
	for (int i = threadIdx.x; i < A.nNeighbors; i += blockDim.x) {
		const DevMap& B = maps[A.neighbors[i]];
		sDepth[i] = B.depth; sNormal[i] = B.normal; sW[i] = B.w; sH[i] = B.h;
	}
	__syncthreads();
	unsigned nd = 0;
	const int nPad = (n + 63) & ~63; // whole waves take part in list_append
	for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nPad; idx += gridDim.x * blockDim.x) {
		bool pend = false;
		if (idx < n) {
			if (A.depth[idx] != 0.f) {
				++nd;
				pend = A.depth[idx] > 0.f; // valid and not part of a point yet
			}
			flag[idx] = 0;
		}
		if (pend) {
			float point[3], normal[3] = {0.f, 0.f, -1.f};
			pixel_point(A, idx, A.depth[idx], point);
			if (A.normal) rotate_normal(A, A.normal + 3 * (size_t)idx, normal);
			for (int q = 0; q < A.nNeighbors; ++q) {
				float ptz; int ib = -1, xB, yB;
				int32_t t = -1;
				if (sDepth[q] && project_target(sP[q], sW[q], sH[q], point, ptz, ib, xB, yB)) {
					const float depthB = sDepth[q][ib];
					if (depthB > 0.f) { // valid and free
						int cls = 0;
						if (is_depth_similar(ptz, depthB, thDepth)) {
							float normalB[3] = {0.f, 0.f, -1.f};
							if (sNormal[q]) rotate_normal(sR[q], sNormal[q] + 3 * (size_t)ib, normalB);
							if (normal[0] * normalB[0] + normal[1] * normalB[1] + normal[2] * normalB[2] > normalError) cls = kTargetMerge;
						}
						if (!cls && ptz < depthB) cls = kTargetInFront;
						if (cls) t = ib | (cls << kTargetShift);
					}
				}
				tb.targets[(size_t)idx * A.nNeighbors + q] = t;
				if (t >= 0) { // pushed onto the target's list of bidders; list entry = q * pixels(A) + pixel
					const uint32_t e = (uint32_t)q * (uint32_t)n + (uint32_t)idx;
					tb.next[e] = atomicExch(&tb.head[q * tb.stride + ib], e);
				}
			}
		}
		list_append(pend, idx, pending, roundCnt + 1);
	}
	if (nd) atomicAdd(&counters[0], (unsigned long long)nd); // valid depths visited (SceneDensify.cpp:3359)
}
// order: 0 = raster order (the reference's), 1 = a fixed pseudo-random order (a bijective hash of the raster index)
__device__ __forceinline__ uint32_t fuse_prio(uint32_t idx, int order) { return order ? idx * 0x9E3779B1u : idx; }

struct FuseOut { // per pixel of the current image, compacted in raster order afterwards
	float* xyz; float* normal; uint8_t* bgr; uint32_t* nviews; uint8_t* flag;
	uint32_t* views; float* weights; int vstride; // optional: the point's view list (image ids ascending) and weights, vstride per pixel
};
// after an image's compaction: the image's counts (counters[0] depths, [3] points, [4] view entries) join the running totals the
// next image's compaction starts from; status[3] says that the cloud (1) or the view lists (2) do not fit
__global__ void fuse_advance_kernel(const unsigned long long* counters, unsigned long long* totals, unsigned long long capacity,
                                    unsigned long long viewCapacity, uint32_t* status) {
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	if (capacity && totals[0] + counters[3] > capacity) status[3] = 1u;
	if (viewCapacity && totals[1] + counters[4] > viewCapacity) status[3] = 2u;
	totals[0] += counters[3]; totals[1] += counters[4]; totals[2] += counters[0];
}
// the end of a fusion: the claim marks (negative depths) come off every map
__global__ void unclaim_kernel(const DevMap* maps, int nMaps) {
	for (int m = blockIdx.y; m < nMaps; m += gridDim.y) {
		float* d = maps[m].depth;
		if (!d) continue;
		const size_t n = (size_t)maps[m].w * maps[m].h;
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
			const float v = d[i];
			if (v < 0.f) d[i] = -v;
		}
	}
}
// ---- the image pass without a dependence chain -----------------------------------------------------------------------------
// The sequential rule (SceneDensify.cpp:3395-3449) in closed form: pixel p becomes a point iff 1 + the number of its merge-class
// targets that are still AVAILABLE reaches nMinViewsFuse, and a target is available to p iff no bidder of that target that comes
// before p in the order of the pass has become a point (such a pixel has claimed the target or, lying in front of it, removed it).
// That is a well-founded recursion over the order of the pass, so it has exactly one solution, and any iteration that keeps
// re-evaluating the pixels whose inputs changed arrives at it: start from "every pixel is a point" (true for > 99 % of the pixels
// of an estimated map), evaluate everybody in parallel, and then only the later bidders of the targets of whoever changed, until
// nothing changes.  The steps are ordinary parallel kernels; their number is the length of the longest chain of CHANGES (a handful),
// not of the dependence graph (hundreds to thousands of hops when it is walked one hop at a time, as rounds 1-3 did).
struct FuseSettle {
	FuseTables tb;
	uint8_t* acc;       // [w*h] 1: the pixel is a point (as far as the iteration knows)
	uint32_t* stamp;    // [w*h] last step a pixel was put on a work list in (one entry per pixel and step)
	uint32_t* work[2];  // work lists: step s reads work[(s - 1) & 1] and writes work[s & 1]
	uint32_t* ctl;      // kCtlWork + s: length of the list step s wrote
	int nNb, nMinViewsFuse, order;
	uint32_t nA;        // pixels of A: list entry e of neighbour q stands for pixel e - q * nA
	uint32_t* mergeMask; // [w*h] bit q: the pixel would merge its target in neighbour q / lies in front of it and would remove it -- as of the
	uint32_t* frontMask; // pixel's last evaluation
};
constexpr uint32_t kListEnd = 0xFFFFFFFFu;
// is p a point, given what acc says about the pixels before it?  (a target's bidders are a linked list built by fuse_begin_kernel, a handful of pixels long)
__device__ __forceinline__ bool settle_eval(const FuseSettle& S, uint32_t p, uint32_t* mergeOut, uint32_t* inFrontOut) {
	const uint32_t pp = fuse_prio(p, S.order);
	uint32_t merge = 0u, inFront = 0u;
	for (int q = 0; q < S.nNb; ++q) {
		const int32_t tg = S.tb.targets[(size_t)p * S.nNb + q];
		if (tg < 0) continue;
		bool avail = true;
		for (uint32_t e = S.tb.head[q * S.tb.stride + (size_t)(tg & kTargetIndexMask)]; e != kListEnd && avail; e = S.tb.next[e]) {
			const uint32_t b = e - (uint32_t)q * S.nA; // the pixel of list entry e
			if (b != p && fuse_prio(b, S.order) < pp && ld_u8(&S.acc[b]) != 0) avail = false;
		}
		if (!avail) continue;
		if ((tg >> kTargetShift) == kTargetMerge) merge |= 1u << q; else inFront |= 1u << q;
	}
	if (mergeOut) { *mergeOut = merge; *inFrontOut = inFront; }
	return 1 + __builtin_popcount(merge) >= S.nMinViewsFuse;
}
// one pixel of a step: re-evaluate; when the answer changed, the later bidders of its targets go on the next list (once per step)
template <class APPEND>
__device__ __forceinline__ void settle_pixel(const FuseSettle& S, uint32_t p, uint32_t step, APPEND append) {
	uint32_t merge, inFront;
	const bool now = settle_eval(S, p, &merge, &inFront);
	// what the pixel would merge and remove: kept from its LAST evaluation, whose inputs are final (a later change of any of them would
	// have put the pixel on a work list again), for fuse_settle_apply_kernel
	S.mergeMask[p] = merge; S.frontMask[p] = inFront;
	if ((ld_u8(&S.acc[p]) != 0) == now) return;
	st_u8(&S.acc[p], now ? 1 : 0);
	const uint32_t pp = fuse_prio(p, S.order);
	for (int q = 0; q < S.nNb; ++q) {
		const int32_t tg = S.tb.targets[(size_t)p * S.nNb + q];
		if (tg < 0) continue;
		for (uint32_t e = S.tb.head[q * S.tb.stride + (size_t)(tg & kTargetIndexMask)]; e != kListEnd; e = S.tb.next[e]) {
			const uint32_t b = e - (uint32_t)q * S.nA;
			if (fuse_prio(b, S.order) > pp && atomicExch(&S.stamp[b], step + 1u) != step + 1u) append(b);
		}
	}
}
// step 0 evaluates every pending pixel, step s > 0 the list step s - 1 left; an empty list costs an empty launch
__global__ void fuse_settle_step_kernel(FuseSettle S, const uint32_t* pending, uint32_t step) {
	const uint32_t n = step == 0u ? S.ctl[kCtlPending] : S.ctl[kCtlWork + step - 1u];
	if (n == 0u) return;
	const uint32_t* list = step == 0u ? pending : S.work[(step - 1u) & 1u];
	uint32_t* next = S.work[step & 1u];
	uint32_t* cnt = S.ctl + kCtlWork + step;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		settle_pixel(S, list[i], step, [&](uint32_t b) { next[atomicAdd(cnt, 1u)] = b; });
}
// whatever the kSettleSteps launches left: one workgroup goes on until the list is empty (no host in the loop, no bound on the
// number of steps other than the safety limit)
__global__ __launch_bounds__(1024) void fuse_settle_rest_kernel(FuseSettle S, uint32_t* status) {
	__shared__ uint32_t nNext;
	uint32_t n = S.ctl[kCtlWork + kSettleSteps];
	uint32_t step = (uint32_t)kSettleSteps + 1u;
	for (; n != 0u; ++step) {
		if (step > (1u << 20)) { if (threadIdx.x == 0) { S.ctl[kCtlErr] = 1u; status[0] = 1u; } break; } // never expected: bounded all the same
		if (threadIdx.x == 0) nNext = 0u;
		__syncthreads();
		const uint32_t* list = S.work[(step - 1u) & 1u];
		uint32_t* next = S.work[step & 1u];
		for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
			settle_pixel(S, ld_u32(&list[i]), step, [&](uint32_t b) { st_u32(&next[atomicAdd(&nNext, 1u)], b); });
		__syncthreads(); // also drains every wave's stores (s_waitcnt vmcnt(0) precedes the barrier)
		n = nNext;
		__syncthreads();
	}
	if (threadIdx.x == 0) S.ctl[kCtlSteps] = step - 1u; // diagnostic
}
// the decisions are final: the points claim the estimates they merge and remove the ones they lie in front of.  No two points
// touch the same estimate (the earlier one made it unavailable to the later one).
__global__ void fuse_settle_apply_kernel(DevMap A, const DevMap* maps, FuseSettle S, const uint32_t* pending, uint32_t* merged, FuseOut out,
                                         unsigned long long* counters) {
	const uint32_t n = S.ctl[kCtlPending];
	unsigned accepted = 0, viewEntries = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const uint32_t p = pending[i];
		if (!S.acc[p]) continue;
		const uint32_t merge = S.mergeMask[p], inFront = S.frontMask[p];
		A.depth[p] = -A.depth[p]; // the claim mark (launch_unclaim restores the sign)
		for (int q = 0; q < S.nNb; ++q) {
			if (!((merge | inFront) >> q & 1u)) continue;
			float* d = maps[A.neighbors[q]].depth + (S.tb.targets[(size_t)p * S.nNb + q] & kTargetIndexMask);
			*d = (merge >> q & 1u) ? -*d : 0.f;
		}
		const int nv = 1 + __builtin_popcount(merge);
		out.nviews[p] = (uint32_t)nv;
		merged[p] = merge;
		out.flag[p] = 1;
		++accepted;
		viewEntries += (unsigned)nv;
	}
	if (accepted) { atomicAdd(&counters[3], (unsigned long long)accepted); atomicAdd(&counters[4], (unsigned long long)viewEntries); }
}

// The points of the pass: position = confidence-weighted mean of the merged estimates, colour, normal, view list
// (SceneDensify.cpp:3371-3386, 3405-3416, 3425-3446), summed in the neighbour order of the reference loop.  Merged
// estimates are claimed, so nobody has changed them since the pass looked at them.
template <int MAXV>
__global__ void fuse_points_kernel(DevMap A, const DevMap* maps, FuseTables tb, const uint32_t* merged, FuseOut out, const uint32_t* pending,
                                   const uint32_t* roundCnt) {
	const int nNb = A.nNeighbors;
	const int nPending = (int)roundCnt[1];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nPending; i += gridDim.x * blockDim.x) {
		const int idx = (int)pending[i];
		if (!out.flag[idx]) continue;
		const uint32_t merge = merged[idx];
		const float depth = fabsf(A.depth[idx]); // the pixels of a point carry the claim mark (negative depth) until the fusion ends
		float point[3];
		pixel_point(A, idx, depth, point);
		uint32_t vimg[MAXV]; float vwt[MAXV]; int nv = 0;
		vwt[nv] = conf2weight(A.conf[idx], depth); // PointCloud::WeightArr (float), SceneDensify.cpp:3378
		vimg[nv] = A.id; ++nv;
		double confidence = (double)vwt[0];
		float normal[3] = {0.f, 0.f, -1.f};
		if (A.normal) rotate_normal(A, A.normal + 3 * (size_t)idx, normal);
		double X[3] = {(double)point[0] * confidence, (double)point[1] * confidence, (double)point[2] * confidence};
		float Cc[3] = {0.f, 0.f, 0.f}, Nn[3];
		if (A.bgr) for (int k = 0; k < 3; ++k) Cc[k] = (float)A.bgr[3 * (size_t)idx + k] * (float)confidence;
		for (int k = 0; k < 3; ++k) Nn[k] = normal[k] * (float)confidence;
		for (int q = 0; q < nNb; ++q) {
			if (!(merge >> q & 1u)) continue;
			const DevMap& B = maps[A.neighbors[q]];
			const int ib = tb.targets[(size_t)idx * nNb + q] & kTargetIndexMask;
			const int yB = ib / B.w, xB = ib - yB * B.w;
			const float depthB = fabsf(B.depth[ib]);
			float normalB[3] = {0.f, 0.f, -1.f};
			if (B.normal) rotate_normal(B, B.normal + 3 * (size_t)ib, normalB);
			const float confB = conf2weight(B.conf[ib], depthB);
			int pos = nv;
			while (pos > 0 && vimg[pos - 1] > B.id) { vimg[pos] = vimg[pos - 1]; vwt[pos] = vwt[pos - 1]; --pos; }
			vimg[pos] = B.id; vwt[pos] = confB; ++nv;
			double XB[3];
			i2w(B, (double)xB, (double)yB, (double)depthB, XB);
			for (int k = 0; k < 3; ++k) X[k] += XB[k] * (double)confB;
			if (B.bgr) for (int k = 0; k < 3; ++k) Cc[k] += (float)B.bgr[3 * (size_t)ib + k] * confB;
			for (int k = 0; k < 3; ++k) Nn[k] += normalB[k] * confB;
			confidence += (double)confB;
		}
		const double nrm = 1.0 / confidence;
		for (int k = 0; k < 3; ++k) out.xyz[3 * (size_t)idx + k] = (float)(X[k] * nrm);
		if (out.bgr) for (int k = 0; k < 3; ++k) {
			const int c8 = (int)floorf(Cc[k] * (float)nrm + .5f);
			out.bgr[3 * (size_t)idx + k] = (uint8_t)(c8 < 0 ? 0 : (c8 > 255 ? 255 : c8));
		}
		if (out.normal) {
			const float n0 = Nn[0] * (float)nrm, n1 = Nn[1] * (float)nrm, n2 = Nn[2] * (float)nrm;
			const float len = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
			out.normal[3 * (size_t)idx] = n0 / len; out.normal[3 * (size_t)idx + 1] = n1 / len; out.normal[3 * (size_t)idx + 2] = n2 / len;
		}
		if (out.views)
			for (int v = 0; v < nv; ++v) { out.views[(size_t)idx * out.vstride + v] = vimg[v]; out.weights[(size_t)idx * out.vstride + v] = vwt[v]; }
	}
}

// ordered compaction of the accepted pixels of one pass into the cloud
// bases: null, or the running totals of an unsynchronised fusion on the device ([0] points, [1] view entries before this image)
__global__ void fuse_gather_kernel(int n, const uint8_t* flag, const uint32_t* pos, FuseOut out, unsigned long long base,
                                   unsigned long long capacity, float* xyz, float* normal, uint8_t* bgr, uint32_t* nviews,
                                   const uint32_t* voff, unsigned long long viewBase, unsigned long long viewCapacity, uint32_t* cviews, float* cweights,
                                   const unsigned long long* bases) {
	if (bases) { base = bases[0]; viewBase = bases[1]; }
	for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
		if (!flag[idx]) continue;
		const unsigned long long o = base + pos[idx];
		if (o >= capacity) continue;
		if (cviews) {
			const unsigned long long vo = viewBase + voff[idx];
			const uint32_t nv = out.nviews[idx];
			if (vo + nv <= viewCapacity)
				for (uint32_t v = 0; v < nv; ++v) { cviews[vo + v] = out.views[(size_t)idx * out.vstride + v]; cweights[vo + v] = out.weights[(size_t)idx * out.vstride + v]; }
		}
		for (int k = 0; k < 3; ++k) xyz[3 * o + k] = out.xyz[3 * (size_t)idx + k];
		if (normal) for (int k = 0; k < 3; ++k) normal[3 * o + k] = out.normal[3 * (size_t)idx + k];
		if (bgr) for (int k = 0; k < 3; ++k) bgr[3 * o + k] = out.bgr[3 * (size_t)idx + k];
		if (nviews) nviews[o] = out.nviews[idx];
	}
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void fill_u64_kernel(unsigned long long* p, unsigned long long v, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void flag_to_u32_kernel(const uint8_t* f, uint32_t* o, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] = f[i];
}
__global__ void flag_nviews_kernel(const uint8_t* f, const uint32_t* nv, uint32_t* o, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] = f[i] ? nv[i] : 0u;
}

// MVS::EstimatePointColors (DepthMap.cpp:2125-2161): per point the colour of the closest of its views (smallest
// Camera::PointDepth), sampled bilinearly the way TImage<Pixel8U>::sample does it -- every product and sum is truncated
// back to 8 bits (Types.inl:2250-2258 over TPixel<uint8_t>::operator*, Types.h:1931) -- or white outside the image
__global__ void point_colors_kernel(unsigned long long n, const float* xyz, const unsigned long long* voff, const uint32_t* views, const DevMap* maps, uint8_t* bgr) {
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
		const float X[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
		double best = 3.402823466e+38; // FLT_MAX
		int bestImg = -1;
		for (unsigned long long v = voff[i]; v < voff[i + 1]; ++v) {
			const DevMap& m = maps[views[v]];
			if (!m.bgr) continue;
			const double d = m.P[8] * (double)X[0] + m.P[9] * (double)X[1] + m.P[10] * (double)X[2] + m.P[11]; // Camera::PointDepth
			if (best > d) { best = d; bestImg = (int)views[v]; }
		}
		uint8_t c[3] = {255, 255, 255};
		if (bestImg >= 0) {
			const DevMap& m = maps[bestImg];
			const double* p = m.P;
			const float qx = (float)(p[0] * X[0] + p[1] * X[1] + p[2] * X[2] + p[3]), qy = (float)(p[4] * X[0] + p[5] * X[1] + p[6] * X[2] + p[7]),
			            qz = (float)(p[8] * X[0] + p[9] * X[1] + p[10] * X[2] + p[11]);
			const float iz = 1.f / qz;
			const float px = qx * iz, py = qy * iz;
			if (px >= 1.f && py >= 1.f && px <= (float)(m.w - 2) && py <= (float)(m.h - 2)) { // isInsideWithBorder<float,1>
				const int lx = (int)px, ly = (int)py;
				const float x = px - (float)lx, x1 = 1.f - x, y = py - (float)ly, y1 = 1.f - y;
				const uint8_t* r0 = m.bgr + 3 * ((size_t)ly * m.w + lx);
				const uint8_t* r1 = r0 + 3 * (size_t)m.w;
				for (int k = 0; k < 3; ++k) {
					const uint8_t top = (uint8_t)(((uint8_t)((uint8_t)(x1 * (float)r0[k]) + (uint8_t)(x * (float)r0[3 + k]))) * y1);
					const uint8_t bot = (uint8_t)(((uint8_t)((uint8_t)(x1 * (float)r1[k]) + (uint8_t)(x * (float)r1[3 + k]))) * y);
					c[k] = (uint8_t)(top + bot);
				}
			}
		}
		bgr[3 * i] = c[0]; bgr[3 * i + 1] = c[1]; bgr[3 * i + 2] = c[2];
	}
}
// ------------------------------------------------------------------------------------------------------
// the fork's depth-map post-filters, applied after outer iterations 1 and 2 (SceneDensify.cpp:3939-3958):
// RemoveSmallSegments as the fork rewrote it (SceneDensify.cpp:2048-2275) = a complete fusion pass, after which depthMap_fuse /
// normalMap_fuse are the image's maps restricted to the pixels that ended up in a fused point (postfilter_mask_kernel, from
// the claim map the fuse pass leaves); GapInterpolation (SceneDensify.cpp:2280-3001): gaps along rows, then along columns
// (gap_lines_kernel: the reference walks a line left to right with a running counter; the gaps of a line turn out to be
// independent of each other, so every gap is found and filled by a thread of its own), then the merge (SceneDensify.cpp:2989-3000).  The third, per-pixel pass of the
// reference (SceneDensify.cpp:2717-2983) reads uninitialised variables and is not reproduced.

// runs BEFORE the claim marks come off: a negative depth = the pixel ended up in a fused point
__global__ void postfilter_mask_kernel(int n, const float* depth, const float* normal, float* dF, float* nF) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const bool on = depth[i] < 0.f;
		dF[i] = on ? -depth[i] : 0.f;
		for (int k = 0; k < 3; ++k) nF[3 * i + k] = on ? normal[3 * i + k] : 0.f;
	}
}
// One thread per PIXEL: the thread of a valid pixel that ends a run of empty ones measures the run and, when the rule says so,
// fills it (left to right, with the running sums of the reference).  Gaps do not interact: the two ends of a gap are pixels that were
// valid before the pass, and a pass only writes pixels that were empty, so every gap sees exactly what the sequential scan of its line
// would have seen.  To keep it so while other threads are filling their gaps, the depths are READ from dIn, which nobody writes during
// the pass, and WRITTEN to dOut (a copy of dIn when the pass begins); normals and confidences are only read at the ends of a gap,
// which are never written.
__global__ void gap_lines_kernel(const float* dIn, float* dOut, float* nF, float* conf, const uint8_t* gra, int nLines, int len, size_t lineStride,
                                 size_t stride, int gap, float thr, unsigned long long* filledOut) {
	const size_t total = (size_t)nLines * len;
	unsigned long long filled = 0;
	for (size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += (size_t)gridDim.x * blockDim.x) {
		// neighbouring threads touch neighbouring addresses in either direction of the pass
		const int line = stride == 1 ? (int)(tid / len) : (int)(tid % nLines);
		const int u = stride == 1 ? (int)(tid % len) : (int)(tid / nLines);
		const size_t base = (size_t)line * lineStride;
		const size_t iu = base + (size_t)u * stride;
		const float depth = dIn[iu];
		// "empty" is `depth <= 0` as in the reference's scan (SceneDensify.cpp:2305 rows, :2636 columns)
		if (depth <= 0.f || u == 0 || !(dIn[iu - stride] <= 0.f)) continue; // not the end of a run of empty pixels
		int count = 1;
		while (count < u && dIn[iu - (size_t)(count + 1) * stride] <= 0.f) ++count;
		if (count >= u) continue; // the run begins at the start of the line: nothing to interpolate from
		const size_t i0 = iu - (size_t)(count + 1) * stride;
		const float depthFirst = dIn[i0];
		bool fill;
		if (count <= gap) fill = is_depth_similar(depthFirst, depth, thr);
		else {
			const float t0 = (float)gra[i0], t1 = (float)gra[iu];
			const float ratio = (t1 - t0) / t0;
			// the reference compares the float against the DOUBLE literal 0.1 (SceneDensify.cpp:2390 rows, :2720 columns): a ratio that
			// rounds to 0.1f -- gradient pairs (10, 11), (20, 22) ... -- is above it and does not fill
			fill = (double)ratio <= 0.1 || is_depth_similar(depthFirst, depth, thr);
		}
		if (!fill) continue;
		const float cnt1 = (float)(count + 1);
		const float diff = (depth - depthFirst) / cnt1;
		float d = depthFirst;
		const float c = conf[i0] < conf[iu] ? conf[i0] : conf[iu];
		float p0 = pm_atan2f(nF[3 * i0 + 1], nF[3 * i0]), p1 = pm_acosf(nF[3 * i0 + 2]);               // Normal2Dir, Util.inl:614-618
		const float q0 = pm_atan2f(nF[3 * iu + 1], nF[3 * iu]), q1 = pm_acosf(nF[3 * iu + 2]);
		const float dd0 = (q0 - p0) / cnt1, dd1 = (q1 - p1) / cnt1;
		for (int k = 1; k <= count; ++k) {
			const size_t ic = i0 + (size_t)k * stride;
			d += diff;
			dOut[ic] = d;
			p0 += dd0; p1 += dd1;
			float s0, c0, s1, c1;
			pm_sincosf(p0, &s0, &c0); pm_sincosf(p1, &s1, &c1);                                       // Dir2Normal, Util.inl:620-625
			nF[3 * ic] = c0 * s1; nF[3 * ic + 1] = s0 * s1; nF[3 * ic + 2] = c1;
			conf[ic] = c;
			++filled;
		}
	}
	if (filled) atomicAdd(filledOut, filled);
}
__global__ void postfilter_merge_kernel(int n, float* depth, float* normal, const float* dF, const float* nF) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		if (dF[i] > 0.f) depth[i] = dF[i];
		const float a = nF[3 * i], b = nF[3 * i + 1], c = nF[3 * i + 2];
		if (a != 0.f || b != 0.f || c != 0.f) { normal[3 * i] = a; normal[3 * i + 1] = b; normal[3 * i + 2] = c; }
	}
}
// the image's own maps still carry the claim marks of the fusion that has just run; `maps` = all maps, unmarked after the mask is taken
void launch_postfilter(int w, int h, float* depth, float* normal, float* conf, const DevMap* maps, int nMaps, const uint8_t* gra, float* dF, float* dF2,
                       float* nF, int gap, float thr, unsigned long long* filled, hipStream_t s) {
	const int n = w * h;
	hipLaunchKernelGGL(postfilter_mask_kernel, kGrid, kBlock, 0, s, n, depth, normal, dF, nF);
	launch_unclaim(maps, nMaps, s);
	(void)hipMemcpyAsync(dF2, dF, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
	hipLaunchKernelGGL(gap_lines_kernel, kGrid, kBlock, 0, s, dF, dF2, nF, conf, gra, h, w, (size_t)w, (size_t)1, gap, thr, filled);   // rows
	(void)hipMemcpyAsync(dF, dF2, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
	hipLaunchKernelGGL(gap_lines_kernel, kGrid, kBlock, 0, s, dF2, dF, nF, conf, gra, w, h, (size_t)1, (size_t)w, gap, thr, filled);   // columns
	hipLaunchKernelGGL(postfilter_merge_kernel, kGrid, kBlock, 0, s, n, depth, normal, dF, nF);
}
void launch_gap_lines(const float* dIn, float* dOut, float* nF, float* conf, const uint8_t* gra, int nLines, int len, size_t lineStride, size_t stride, int gap,
                      float thr, unsigned long long* filled, hipStream_t s) {
	hipLaunchKernelGGL(gap_lines_kernel, kGrid, kBlock, 0, s, dIn, dOut, nF, conf, gra, nLines, len, lineStride, stride, gap, thr, filled);
}
void launch_unclaim(const DevMap* maps, int nMaps, hipStream_t s) {
	hipLaunchKernelGGL(unclaim_kernel, dim3(256, nMaps < 1024 ? (nMaps > 0 ? nMaps : 1) : 1024), dim3(256), 0, s, maps, nMaps);
}

void launch_point_colors(unsigned long long n, const float* xyz, const unsigned long long* voff, const uint32_t* views, const DevMap* maps, uint8_t* bgr, hipStream_t s) {
	hipLaunchKernelGGL(point_colors_kernel, dim3(2048), dim3(256), 0, s, n, xyz, voff, views, maps, bgr);
}

// ------------------------------------------------------------------------------------------------------
// launch wrappers


void launch_fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s) { hipLaunchKernelGGL(fill_u32_kernel, kGrid, kBlock, 0, s, p, v, n); }
void launch_fill_u64(unsigned long long* p, unsigned long long v, size_t n, hipStream_t s) { hipLaunchKernelGGL(fill_u64_kernel, kGrid, kBlock, 0, s, p, v, n); }

void launch_filter_splat(const DevMap& ref, const DevMap& nb, unsigned long long* key, hipStream_t s) {
	hipLaunchKernelGGL(filter_splat_kernel, kGrid, kBlock, 0, s, ref, nb, key);
}
void launch_filter_vote(const DevMap& ref, const DevMap* nbs, int N, const unsigned long long* keys, int adjust, int nMinViews,
                        int nMinViewsAdjust, float thr, float* newDepth, float* newConf, unsigned long long* counters, hipStream_t s) {
	hipLaunchKernelGGL(filter_vote_kernel, kGrid, kBlock, 0, s, ref, nbs, N, keys, adjust, nMinViews, nMinViewsAdjust, thr, newDepth, newConf, counters);
}
FuseTables fuse_tables(int32_t* targets, uint32_t* head, uint32_t* next, size_t stride) {
	FuseTables tb;
	tb.targets = targets; tb.head = head; tb.next = next; tb.stride = stride;
	return tb;
}
// begin of an image pass: the pending list (its length in ctl[kCtlPending]), the pixels' targets and the per-target bidder lists
// (tb.head must be all ones)
void launch_fuse_begin(const DevMap& A, const DevMap* maps, const FuseTables& tb, uint32_t* pending, uint32_t* ctl, uint8_t* flag,
                       unsigned long long* counters, float thDepth, float normalError, hipStream_t s) {
	hipLaunchKernelGGL(fuse_begin_kernel, kGrid, kBlock, 0, s, A, maps, tb, pending, ctl + kCtlPending - 1, flag, counters, thDepth, normalError); // roundCnt[1] == ctl[kCtlPending]
}
// the image pass: which pending pixels become points (the settle iteration), their claims, then the points themselves.
// settle: scratch of fuse_settle_bytes(pixels of A)
size_t fuse_settle_bytes(size_t pixels) { return ((pixels * 4 + 255) & ~(size_t)255) * 3 + pixels; }
void launch_fuse_pass(const DevMap& A, const DevMap* maps, const FuseTables& tb, const uint32_t* pending, void* settle, uint32_t* ctl,
                      float* oxyz, float* onormal, uint8_t* obgr, uint32_t* onv, uint8_t* oflag, uint32_t* oviews, float* oweights, int vstride,
                      uint32_t* merged, int nMinViewsFuse, int order, unsigned long long* counters, uint32_t* status, bool wantPoints, hipStream_t s) {
	FuseOut out{oxyz, onormal, obgr, onv, oflag, oviews, oweights, vstride};
	const size_t n = (size_t)A.w * A.h, words = ((n * 4 + 255) & ~(size_t)255) / 4;
	FuseSettle S;
	S.tb = tb; S.work[0] = (uint32_t*)settle; S.work[1] = S.work[0] + words; S.stamp = S.work[1] + words; S.acc = (uint8_t*)(S.stamp + words);
	S.ctl = ctl; S.nNb = A.nNeighbors; S.nMinViewsFuse = nMinViewsFuse; S.order = order;
	S.nA = (uint32_t)n;
	S.mergeMask = merged; S.frontMask = onv; // onv: the points' view counts, written by the apply kernel after it has read the mask
	(void)hipMemsetAsync(S.stamp, 0, n * 4, s);
	(void)hipMemsetAsync(S.acc, 1, n, s);
	for (int step = 0; step <= kSettleSteps; ++step) hipLaunchKernelGGL(fuse_settle_step_kernel, kGrid, kBlock, 0, s, S, pending, (uint32_t)step);
	hipLaunchKernelGGL(fuse_settle_rest_kernel, dim3(1), dim3(1024), 0, s, S, status);
	hipLaunchKernelGGL(fuse_settle_apply_kernel, kGrid, kBlock, 0, s, A, maps, S, pending, merged, out, counters);
	if (!wantPoints) return;
	if (A.nNeighbors < 16) hipLaunchKernelGGL(fuse_points_kernel<16>, kGrid, kBlock, 0, s, A, maps, tb, merged, out, pending, ctl + kCtlPending - 1);
	else hipLaunchKernelGGL(fuse_points_kernel<32>, kGrid, kBlock, 0, s, A, maps, tb, merged, out, pending, ctl + kCtlPending - 1);
}
size_t fuse_scan_temp_bytes(int n) {
	size_t bytes = 0;
	(void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, n);
	return bytes;
}
void launch_fuse_compact(int n, const uint8_t* flag, uint32_t* flag32, uint32_t* pos, void* temp, size_t tempBytes, float* oxyz,
                         float* onormal, uint8_t* obgr, uint32_t* onv, unsigned long long base, unsigned long long capacity, float* xyz,
                         float* normal, uint8_t* bgr, uint32_t* nviews, uint32_t* oviews, float* oweights, int vstride, uint32_t* voff,
                         unsigned long long viewBase, unsigned long long viewCapacity, uint32_t* cviews, float* cweights, const unsigned long long* bases, hipStream_t s) {
	hipLaunchKernelGGL(flag_to_u32_kernel, kGrid, kBlock, 0, s, flag, flag32, n);
	(void)hipcub::DeviceScan::ExclusiveSum(temp, tempBytes, flag32, pos, n, s);
	if (cviews) { // offsets of the accepted pixels' view lists inside this image's part of the CSR arrays
		hipLaunchKernelGGL(flag_nviews_kernel, kGrid, kBlock, 0, s, flag, onv, flag32, n);
		(void)hipcub::DeviceScan::ExclusiveSum(temp, tempBytes, flag32, voff, n, s);
	}
	FuseOut out{oxyz, onormal, obgr, onv, const_cast<uint8_t*>(flag), oviews, oweights, vstride};
	hipLaunchKernelGGL(fuse_gather_kernel, kGrid, kBlock, 0, s, n, flag, pos, out, base, capacity, xyz, normal, bgr, nviews, voff, viewBase, viewCapacity,
	                   cviews, cweights, bases);
}
void launch_fuse_advance(const unsigned long long* counters, unsigned long long* totals, unsigned long long capacity, unsigned long long viewCapacity,
                         uint32_t* status, hipStream_t s) {
	hipLaunchKernelGGL(fuse_advance_kernel, dim3(1), dim3(64), 0, s, counters, totals, capacity, viewCapacity, status);
}

} // namespace hcmvs
