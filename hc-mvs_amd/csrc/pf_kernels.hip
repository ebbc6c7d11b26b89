/*
 * hc-mvs_amd/csrc/pf_kernels.hip -- the fusions of the post-filter chain, incrementally.
 *
 * The fork's RemoveSmallSegments (frame_main/libs/MVS/SceneDensify.cpp:2048-2275) is a complete FuseDepthMaps pass over the maps of ALL
 * images, run once per image k of an outer iteration (SceneDensify.cpp:3939-3958): n whole-scene fusions per filtered iteration.  Two
 * facts make all but the first of them cheap without changing a single decision:
 *   (1) A fusion repeated on its own output repeats its decisions: the estimates it invalidated are simply absent the second time, every
 *       other pair of estimates compares as before (the claims are per fusion, only the zeroed depths persist).
 *   (2) Between the fusion of image k and that of image k + 1 only image k changes -- the pixels GapInterpolation filled.
 * So fusion k + 1 differs from fusion k only in what depends on those pixels.  The sequential rule (SceneDensify.cpp:3353-3449) in closed
 * form, over ALL images: a pixel p of image A (the i-th of the fusion order) becomes a point iff it is still free when its turn comes
 * and 1 + the number of its merge-class targets that are still available then reaches nMinViewsFuse; an estimate is taken (claimed or
 * zeroed) by the FIRST point, in the order (image, raster index), that merges it or lies in front of it.  That is a well-founded
 * recursion with one solution; fuse_kernels.hip solves it per image pass from scratch, here the solution of the previous fusion is KEPT
 * (per image: the targets of every pixel, the lists of bidders per target, the decisions, and per estimate the pass that owns it) and
 * only the pixels whose inputs changed are evaluated again, then whoever depends on a changed decision, until nothing changes:
 *     pf_delta_kernel   finds the pixels of image A whose own estimate or one of whose targets changed (value: filled / zeroed; status:
 *                       claimed or released by an earlier pass of THIS fusion), recomputes their projections and classes
 *     pf_settle_*       the settle iteration of fuse_kernels.hip over that set (plus whoever its changes reach)
 *     pf_apply_*        the differences between old and new decisions: estimates released, claimed (possibly taken from a LATER pass,
 *                       whose bidders then find the estimate changed when their pass comes), zeroed
 * "Claimed" is the index of the owning pass per estimate (own[]), not the sign of the depth: a pass must see the estimates a later pass
 * claimed in the previous fusion as free.  The first fusion of a chain evaluates everything (same kernels, `first`).  The result is the
 * sequential algorithm's, fusion by fusion: tests/test_gpu_fuse.py checks the chain image after image against the oracle, bit for bit.
 */
#include "fuse_common.h"
#include "fuse_device.h"
#include "pf_chain.h"

namespace hcmvs {

static const dim3 kPfGrid(2048), kPfBlock(256);

// a stored target (PfImage::tgt): where pixel p of A projects in neighbour q and what it could do there
constexpr uint32_t kPfIdxMask = 0x03FFFFFFu, kPfNoTarget = 0x03FFFFFFu; // bits 0-25: pixel index in the neighbour (2^26 - 1 = projects nowhere)
constexpr int kPfClsShift = 26;                                         // bits 26-27: 0 nothing, 1 merge (similar depth and normal), 2 in front
constexpr int kPfPushShift = 28;                                        // bits 28-29: how often the pair was linked into a list (at most twice)
constexpr uint32_t kPfLinked = 1u << 30, kPfBank = 1u << 31;            // the pair's entry of bank (bit 31) is linked into the list of target idx
constexpr int kPfMerge = 1, kPfFront = 2;
constexpr uint32_t kPfEnd = 0xFFFFFFFFu;

__device__ __forceinline__ uint16_t ld_u16(const uint16_t* p) { return __hip_atomic_load((__attribute__((address_space(1))) uint16_t*)p, __ATOMIC_RELAXED, FS_SCOPE); }
__device__ __forceinline__ void st_u16(uint16_t* p, uint16_t v) { __hip_atomic_store((__attribute__((address_space(1))) uint16_t*)p, v, __ATOMIC_RELAXED, FS_SCOPE); }

// is the estimate of pixel p of an image free for pass i?  (unclaimed, or claimed by pass i itself / a later pass of the previous fusion)
__device__ __forceinline__ bool pf_free(uint16_t own, int i) { return own == kPfNone || (int)own >= i; }

// ---- which pixels of image A have to be evaluated again, with fresh projections and classes ------------------------------------------
__global__ void pf_delta_kernel(DevMap A, int i, const DevMap* maps, const PfImage* pf, int first, float thDepth, float normalError, uint32_t* delta,
                                uint32_t* ctl) {
	const int n = A.w * A.h, nNb = A.nNeighbors;
	const PfImage me = pf[A.id];
	__shared__ double sP[kFuseMaxViews - 1][12], sR[kFuseMaxViews - 1][9];
	__shared__ const float* sDepth[kFuseMaxViews - 1];
	__shared__ const float* sNormal[kFuseMaxViews - 1];
	__shared__ const uint8_t* sChg[kFuseMaxViews - 1];
	__shared__ int sW[kFuseMaxViews - 1], sH[kFuseMaxViews - 1];
	for (int k = threadIdx.x; k < nNb * 12; k += blockDim.x) sP[k / 12][k % 12] = maps[A.neighbors[k / 12]].P[k % 12];
	for (int k = threadIdx.x; k < nNb * 9; k += blockDim.x) sR[k / 9][k % 9] = maps[A.neighbors[k / 9]].R[k % 9];
	for (int k = threadIdx.x; k < nNb; k += blockDim.x) {
		const DevMap& B = maps[A.neighbors[k]];
		sDepth[k] = B.depth; sNormal[k] = B.normal; sW[k] = B.w; sH[k] = B.h; sChg[k] = pf[A.neighbors[k]].chgNow;
	}
	__shared__ uint32_t sAny;
	if (threadIdx.x == 0) {
		uint32_t a = first ? 1u : *me.anyChg;
		for (int k = 0; k < nNb && !a; ++k) if (pf[A.neighbors[k]].anyChg) a = *pf[A.neighbors[k]].anyChg;
		sAny = a;
	}
	__syncthreads();
	if (!sAny) return; // neither the image nor a neighbour holds a changed estimate: every decision of the pass stands
	const int nPad = (n + 63) & ~63; // whole waves take part in list_append
	const size_t bankSize = (size_t)nNb * (size_t)n;
	for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nPad; idx += gridDim.x * blockDim.x) {
		bool inDelta = false;
		if (idx < n) {
			const float d = A.depth[idx];
			const bool pend = d > 0.f && pf_free(me.own[idx], i);
			const bool wasPoint = !first && me.acc[idx] != 0;
			bool any = first || me.chgNow[idx] != 0;
			if (!any && (pend || wasPoint))
				for (int q = 0; q < nNb && !any; ++q) {
					const uint32_t t = me.tgt[(size_t)idx * nNb + q];
					if ((t & kPfIdxMask) != kPfNoTarget && sDepth[q]) any = sChg[q][t & kPfIdxMask] != 0;
				}
			inDelta = any && (pend || wasPoint);
			if (first) me.acc[idx] = pend ? 1 : 0; // the iteration starts from "every pending pixel is a point"
			if (inDelta && pend) {
				float point[3], normal[3] = {0.f, 0.f, -1.f};
				pixel_point(A, idx, d, point);
				if (A.normal) rotate_normal(A, A.normal + 3 * (size_t)idx, normal);
				for (int q = 0; q < nNb; ++q) {
					uint32_t t = me.tgt[(size_t)idx * nNb + q];
					float ptz; int ib = -1, xB, yB;
					uint32_t newIdx = kPfNoTarget, cls = 0;
					if (sDepth[q] && project_target(sP[q], sW[q], sH[q], point, ptz, ib, xB, yB)) {
						newIdx = (uint32_t)ib;
						const float depthB = sDepth[q][ib];
						if (depthB > 0.f) {
							if (is_depth_similar(ptz, depthB, thDepth)) {
								float normalB[3] = {0.f, 0.f, -1.f};
								if (sNormal[q]) rotate_normal(sR[q], sNormal[q] + 3 * (size_t)ib, normalB);
								if (normal[0] * normalB[0] + normal[1] * normalB[1] + normal[2] * normalB[2] > normalError) cls = kPfMerge;
							}
							if (!cls && ptz < depthB) cls = kPfFront;
						}
					}
					uint32_t linked = t & kPfLinked, bank = t & kPfBank, pushes = (t >> kPfPushShift) & 3u;
					if (newIdx != (t & kPfIdxMask) && linked) { bank ^= kPfBank; linked = 0; } // my estimate moved: the old link is dead (it stays in its list, recognised by its bank)
					if (cls && !linked) {
						if (pushes >= 2u) ctl[kCtlErr] = 1u; // a third link of one pair: can not happen while every image is filled once per chain
						else {
							const uint32_t e = (uint32_t)((bank ? bankSize : 0) + (size_t)q * n + idx);
							me.next[e] = atomicExch(&me.head[(size_t)q * me.stride + newIdx], e);
							linked = kPfLinked; ++pushes;
						}
					}
					me.tgt[(size_t)idx * nNb + q] = newIdx | (cls << kPfClsShift) | (pushes << kPfPushShift) | linked | bank;
				}
			}
		}
		list_append(inDelta, idx, delta, ctl + kCtlPending);
	}
}

// ---- the settle iteration over the delta set -------------------------------------------------------------------------------------
struct PfSettle {
	PfImage me;            // tables of image A
	const DevMap* maps; const PfImage* pf;
	const uint32_t* nbIds; // A.neighbors
	const float* depthA;
	int i, nNb, nMinViewsFuse;
	uint32_t nA;
	uint32_t* work[2];
	uint32_t* touched;     // the pixels evaluated in this pass (once each), with what they were before
	uint8_t* oldAcc; uint32_t *oldMM, *oldFM;
	uint32_t* ctl;
	uint32_t tag;          // unique per (fusion, pass): marks `touch`
	uint32_t stampBase;    // unique per (fusion, pass): stamp = stampBase + step
	int first;
};
// is list entry e (of bank, neighbour q, pixel b) alive and a bid on target y?
__device__ __forceinline__ bool pf_entry_bids(const PfSettle& S, uint32_t e, int q, uint32_t y, uint32_t& b) {
	const size_t bankSize = (size_t)S.nNb * S.nA;
	const uint32_t bankBit = e >= bankSize ? kPfBank : 0u;
	b = (uint32_t)((e - (bankBit ? bankSize : 0)) - (size_t)q * S.nA);
	const uint32_t tb = S.me.tgt[(size_t)b * S.nNb + q];
	return (tb & kPfLinked) && (tb & kPfBank) == bankBit && (tb & kPfIdxMask) == y && ((tb >> kPfClsShift) & 3u) != 0u;
}
__device__ __forceinline__ bool pf_eval(const PfSettle& S, uint32_t p, uint32_t* mergeOut, uint32_t* frontOut) {
	uint32_t merge = 0u, front = 0u;
	const float d = S.depthA[p];
	const bool pend = d > 0.f && pf_free(ld_u16(&S.me.own[p]), S.i);
	if (pend)
		for (int q = 0; q < S.nNb; ++q) {
			const uint32_t t = S.me.tgt[(size_t)p * S.nNb + q];
			const uint32_t cls = (t >> kPfClsShift) & 3u;
			if (!cls) continue;
			const uint32_t y = t & kPfIdxMask;
			const uint32_t B = S.nbIds[q];
			if (!(S.maps[B].depth[y] > 0.f)) continue;                 // zeroed since the class was taken
			if (!pf_free(S.pf[B].own[y], S.i)) continue;                // belongs to a point of an earlier pass
			bool avail = true;
			for (uint32_t e = S.me.head[(size_t)q * S.me.stride + y]; e != kPfEnd && avail; e = S.me.next[e]) {
				uint32_t b;
				if (pf_entry_bids(S, e, q, y, b) && b < p && ld_u8(&S.me.acc[b]) != 0) avail = false;
			}
			if (!avail) continue;
			if (cls == (uint32_t)kPfMerge) merge |= 1u << q; else front |= 1u << q;
		}
	*mergeOut = merge; *frontOut = front;
	return pend && 1 + __builtin_popcount(merge) >= S.nMinViewsFuse;
}
template <class APPEND>
__device__ __forceinline__ void pf_settle_pixel(const PfSettle& S, uint32_t p, uint32_t step, APPEND append) {
	if (atomicExch(&S.me.touch[p], S.tag) != S.tag) { // first evaluation in this pass: remember what the pixel was
		const uint32_t k = atomicAdd(S.ctl + kCtlTouched, 1u);
		S.touched[k] = p;
		S.oldAcc[p] = S.first ? 0 : S.me.acc[p]; S.oldMM[p] = S.first ? 0u : S.me.mm[p]; S.oldFM[p] = S.first ? 0u : S.me.fm[p];
	}
	uint32_t merge, front;
	const bool now = pf_eval(S, p, &merge, &front);
	S.me.mm[p] = merge; S.me.fm[p] = front; // kept from the LAST evaluation, whose inputs are final
	if ((ld_u8(&S.me.acc[p]) != 0) == now) return;
	st_u8(&S.me.acc[p], now ? 1 : 0);
	for (int q = 0; q < S.nNb; ++q) { // the later bidders of my targets see another world now
		const uint32_t t = S.me.tgt[(size_t)p * S.nNb + q];
		if (!((t >> kPfClsShift) & 3u)) continue;
		const uint32_t y = t & kPfIdxMask;
		for (uint32_t e = S.me.head[(size_t)q * S.me.stride + y]; e != kPfEnd; e = S.me.next[e]) {
			uint32_t b;
			if (pf_entry_bids(S, e, q, y, b) && b > p && atomicExch(&S.me.stamp[b], S.stampBase + step + 1u) != S.stampBase + step + 1u) append(b);
		}
	}
}
__global__ void pf_settle_step_kernel(PfSettle S, const uint32_t* delta, uint32_t step) {
	const uint32_t n = step == 0u ? S.ctl[kCtlPending] : S.ctl[kCtlWork + step - 1u];
	if (n == 0u) return;
	const uint32_t* list = step == 0u ? delta : S.work[(step - 1u) & 1u];
	uint32_t* next = S.work[step & 1u];
	uint32_t* cnt = S.ctl + kCtlWork + step;
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
		pf_settle_pixel(S, list[k], step, [&](uint32_t b) { next[atomicAdd(cnt, 1u)] = b; });
}
__global__ __launch_bounds__(1024) void pf_settle_rest_kernel(PfSettle S, uint32_t* status) {
	__shared__ uint32_t nNext;
	uint32_t n = S.ctl[kCtlWork + kSettleSteps];
	uint32_t step = (uint32_t)kSettleSteps + 1u;
	for (; n != 0u; ++step) {
		if (step > (1u << 20)) { if (threadIdx.x == 0) { S.ctl[kCtlErr] = 1u; status[0] = 1u; } break; } // never expected: bounded all the same
		if (threadIdx.x == 0) nNext = 0u;
		__syncthreads();
		const uint32_t* list = S.work[(step - 1u) & 1u];
		uint32_t* next = S.work[step & 1u];
		for (uint32_t k = threadIdx.x; k < n; k += blockDim.x)
			pf_settle_pixel(S, ld_u32(&list[k]), step, [&](uint32_t b) { st_u32(&next[atomicAdd(&nNext, 1u)], b); });
		__syncthreads();
		n = nNext;
		__syncthreads();
	}
	if (threadIdx.x == 0) { S.ctl[kCtlSteps] = step - 1u; if (S.ctl[kCtlErr]) status[0] = 1u; }
}

// ---- what changed: releases first, then claims and invalidations (a released estimate may be claimed by another pixel of the pass) ----
// Which changes later passes must hear of (chgNow + the image's flag).  An estimate that is FREE when a point takes it -- claims it,
// zeroes it, or is itself the pixel that becomes a point -- was free at the end of the previous fusion too (or was released in this
// one, which marked it): no point of a later pass can have had it as an available merge or in-front target, for that point would have
// claimed or zeroed it then.  Its later bidders are pixels that failed to become points, and losing a target does not make a failing
// pixel succeed; whoever of them is evaluated again for another reason reads the owner map as it stands.  So taking a free estimate
// changes no later decision and is not announced.  What is: an estimate RELEASED (its later bidders may now merge it), one TAKEN FROM A
// LATER PASS's point of the previous fusion (that point loses a view), and -- by pf_merge_kernel, for the next fusion -- a changed value.
__global__ void pf_apply_kernel(DevMap A, PfSettle S, int phase) {
	const uint32_t n = S.ctl[kCtlTouched];
	for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const uint32_t p = S.touched[k];
		const bool was = S.oldAcc[p] != 0, now = S.me.acc[p] != 0;
		const uint32_t oldM = was ? S.oldMM[p] : 0u, newM = now ? S.me.mm[p] : 0u, newF = now ? S.me.fm[p] : 0u;
		const uint16_t me16 = (uint16_t)S.i;
		if (phase == 0) {
			for (uint32_t bits = oldM & ~newM; bits; bits &= bits - 1u) {
				const int q = __builtin_ctz(bits);
				const uint32_t y = S.me.tgt[(size_t)p * S.nNb + q] & kPfIdxMask;
				const PfImage& tb = S.pf[S.nbIds[q]];
				if (y != kPfNoTarget && tb.own[y] == me16) { tb.own[y] = kPfNone; tb.chgNow[y] = 1; *tb.anyChg = 1u; }
			}
			if (was && !now && S.me.own[p] == me16) { S.me.own[p] = kPfNone; S.me.chgNow[p] = 1; *S.me.anyChg = 1u; }
		} else {
			for (uint32_t bits = newM & ~oldM; bits; bits &= bits - 1u) {
				const int q = __builtin_ctz(bits);
				const uint32_t y = S.me.tgt[(size_t)p * S.nNb + q] & kPfIdxMask;
				const PfImage& tb = S.pf[S.nbIds[q]];
				const uint16_t prev = tb.own[y];
				tb.own[y] = me16;
				if (prev != kPfNone && prev != me16) { tb.chgNow[y] = 1; *tb.anyChg = 1u; } // taken from a later pass of the previous fusion
			}
			for (uint32_t bits = newF; bits; bits &= bits - 1u) {
				const int q = __builtin_ctz(bits);
				const uint32_t y = S.me.tgt[(size_t)p * S.nNb + q] & kPfIdxMask;
				const uint32_t B = S.nbIds[q];
				float* dB = S.maps[B].depth;
				if (dB[y] != 0.f) { // SceneDensify.cpp:3447-3449
					dB[y] = 0.f;
					const PfImage& tb = S.pf[B];
					const uint16_t prev = tb.own[y];
					if (prev != kPfNone) { // it belonged to a point of this pass (whose pixel the settle iteration has re-evaluated) or of a later one
						tb.own[y] = kPfNone;
						if (prev != me16) { tb.chgNow[y] = 1; *tb.anyChg = 1u; }
					}
				}
			}
			if (now && !was) {
				const uint16_t prev = S.me.own[p];
				S.me.own[p] = me16;
				if (prev != kPfNone && prev != me16) { S.me.chgNow[p] = 1; *S.me.anyChg = 1u; } // it was a merged view of a later pass's point
			}
		}
	}
	(void)A;
}

// the chain moves on to its next fusion: what changed VALUE since the last one (filled, zeroed) is what every pass must look at
__global__ void pf_roll_kernel(const DevMap* maps, const PfImage* pf, int nMaps) {
	for (int m = blockIdx.y; m < nMaps; m += gridDim.y) {
		if (!maps[m].depth || !pf[m].chgNow) continue;
		const size_t n = (size_t)maps[m].w * maps[m].h;
		uint8_t *c = pf[m].chgNow, *v = pf[m].valNext;
		bool any = false;
		for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) { const uint8_t x = v[k]; c[k] = x; v[k] = 0; any = any || x; }
		if (any) *pf[m].anyChg = 1u; // (the words are zeroed before the launch)
	}
}

// depthMap_fuse / normalMap_fuse of the image (SceneDensify.cpp:2232-2275): its maps restricted to the estimates that belong to a point
__global__ void pf_mask_kernel(int n, const float* depth, const float* normal, const uint16_t* own, float* dF, float* nF) {
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		const bool on = own[k] != kPfNone && depth[k] > 0.f;
		dF[k] = on ? depth[k] : 0.f;
		for (int c = 0; c < 3; ++c) nF[3 * k + c] = on ? normal[3 * k + c] : 0.f;
	}
}
// SceneDensify.cpp:2989-3000 + the note for the next fusion: which estimates of the image changed
__global__ void pf_merge_kernel(int n, float* depth, float* normal, const float* dF, const float* nF, uint8_t* valNext) {
	for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
		bool changed = false;
		if (dF[k] > 0.f && depth[k] != dF[k]) { depth[k] = dF[k]; changed = true; }
		const float a = nF[3 * k], b = nF[3 * k + 1], c = nF[3 * k + 2];
		if (a != 0.f || b != 0.f || c != 0.f) {
			if (normal[3 * k] != a || normal[3 * k + 1] != b || normal[3 * k + 2] != c) changed = true;
			normal[3 * k] = a; normal[3 * k + 1] = b; normal[3 * k + 2] = c;
		}
		if (changed) valNext[k] = 1;
	}
}

// ------------------------------------------------------------------------------------------------------
// launch wrappers

size_t pf_pass_scratch_bytes(size_t pixels) { return ((pixels * 4 + 255) & ~(size_t)255) * 6 + ((pixels + 255) & ~(size_t)255); }

void launch_pf_roll(const DevMap* maps, const PfImage* pf, int nMaps, uint32_t* anyChgWords, hipStream_t s) {
	(void)hipMemsetAsync(anyChgWords, 0, (size_t)nMaps * 4, s);
	hipLaunchKernelGGL(pf_roll_kernel, dim3(64, nMaps < 1024 ? (nMaps > 0 ? nMaps : 1) : 1024), dim3(256), 0, s, maps, pf, nMaps);
}
// one image pass of one fusion of the chain.  scratch: pf_pass_scratch_bytes(pixels of A); ctl: kCtlBytes (zeroed here)
void launch_pf_pass(const DevMap& A, int i, const DevMap* maps, const PfImage* dPf, const PfImage& hostPfA, bool first, float thDepth, float normalError,
                    int nMinViewsFuse, uint32_t fusionIndex, void* scratch, uint32_t* ctl, uint32_t* status, hipStream_t s) {
	const size_t n = (size_t)A.w * A.h, words = ((n * 4 + 255) & ~(size_t)255) / 4;
	uint32_t* base = (uint32_t*)scratch;
	uint32_t* delta = base;
	PfSettle S;
	S.me = hostPfA; S.maps = maps; S.pf = dPf; S.nbIds = A.neighbors; S.depthA = A.depth;
	S.i = i; S.nNb = A.nNeighbors; S.nMinViewsFuse = nMinViewsFuse; S.nA = (uint32_t)n;
	S.work[0] = base + words; S.work[1] = base + 2 * words; S.touched = base + 3 * words; S.oldMM = base + 4 * words; S.oldFM = base + 5 * words;
	S.oldAcc = (uint8_t*)(base + 6 * words);
	S.ctl = ctl; S.first = first ? 1 : 0;
	S.tag = fusionIndex + 1u;                      // `touch` is per image, an image has one pass per fusion
	S.stampBase = fusionIndex << 21;               // steps stay below 2^20 (pf_settle_rest_kernel gives up there)
	(void)hipMemsetAsync(ctl, 0, kCtlBytes, s);
	// (smaller grids for the later fusions of a chain were tried -- 512 / 128 workgroups: 1.09 instead of 0.88 s per filtered iteration on
	// 64 x 1080p; a pass re-evaluates some 10^5 pixels, it is not launch-bound)
	const dim3 gridScan = kPfGrid, gridList = kPfGrid;
	hipLaunchKernelGGL(pf_delta_kernel, gridScan, kPfBlock, 0, s, A, i, maps, dPf, first ? 1 : 0, thDepth, normalError, delta, ctl);
	for (int step = 0; step <= kSettleSteps; ++step) hipLaunchKernelGGL(pf_settle_step_kernel, gridList, kPfBlock, 0, s, S, delta, (uint32_t)step);
	hipLaunchKernelGGL(pf_settle_rest_kernel, dim3(1), dim3(1024), 0, s, S, status);
	hipLaunchKernelGGL(pf_apply_kernel, gridList, kPfBlock, 0, s, A, S, 0);
	hipLaunchKernelGGL(pf_apply_kernel, gridList, kPfBlock, 0, s, A, S, 1);
}
// mask -> gap interpolation along rows, then columns -> merge, for the image whose turn it is (launch_postfilter of fuse_kernels.hip with
// the owner map in place of the claim marks)
void launch_gap_lines(const float* dIn, float* dOut, float* nF, float* conf, const uint8_t* gra, int nLines, int len, size_t lineStride, size_t stride, int gap,
                      float thr, unsigned long long* filled, hipStream_t s); // fuse_kernels.hip
void launch_pf_filter(int w, int h, float* depth, float* normal, float* conf, const PfImage& hostPf, const uint8_t* gra, float* dF, float* dF2, float* nF, int gap,
                      float thr, unsigned long long* filled, hipStream_t s) {
	const int n = w * h;
	hipLaunchKernelGGL(pf_mask_kernel, kPfGrid, kPfBlock, 0, s, n, depth, normal, hostPf.own, dF, nF);
	(void)hipMemcpyAsync(dF2, dF, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
	launch_gap_lines(dF, dF2, nF, conf, gra, h, w, (size_t)w, (size_t)1, gap, thr, filled, s);   // rows
	(void)hipMemcpyAsync(dF, dF2, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
	launch_gap_lines(dF2, dF, nF, conf, gra, w, h, (size_t)1, (size_t)w, gap, thr, filled, s);   // columns
	hipLaunchKernelGGL(pf_merge_kernel, kPfGrid, kPfBlock, 0, s, n, depth, normal, dF, nF, hostPf.valNext);
}

} // namespace hcmvs
