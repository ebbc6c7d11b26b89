/*
 * hc-mvs_amd/csrc/fuse_common.h -- structures shared by the host API and the filter / fuse kernels.
 */
#ifndef HCMVS_FUSE_COMMON_H
#define HCMVS_FUSE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hcmvs {

constexpr int kFuseMaxViews = 32; // views merged into one point / estimates one point can invalidate

// one image's estimated maps + camera as the filter / fuse kernels see it (DepthMap.h:214-262 DepthData)
struct DevMap {
	uint32_t id;
	int32_t w, h, nNeighbors;
	double K[9], R[9], C[3], P[12]; // P = K [R | -R C] (Camera.h:276-282)
	float* depth;            // mutated by fusion (SceneDensify.cpp:3447-3449); while a fusion runs a NEGATIVE depth marks an estimate that is
	                         // part of a point (SceneDensify.cpp:3313 arrDepthIdx); launch_unclaim() restores the sign
	const float* normal;     // camera space, may be null
	const float* conf;
	const uint8_t* bgr;      // may be null
	const uint32_t* neighbors; // device array of image ids, decreasing importance
	float dMin, dMax;
};

void launch_fill_u32(uint32_t* p, uint32_t v, size_t n, hipStream_t s);
void launch_fill_u64(unsigned long long* p, unsigned long long v, size_t n, hipStream_t s);
void launch_filter_splat(const DevMap& ref, const DevMap& nb, unsigned long long* key, hipStream_t s);
void launch_filter_vote(const DevMap& ref, const DevMap* nbs, int N, const unsigned long long* keys, int adjust, int nMinViews,
                        int nMinViewsAdjust, float thr, float* newDepth, float* newConf, unsigned long long* counters, hipStream_t s);
// per-pass tables of image A (sized for the largest image / neighbour count of the call)
struct FuseTables {
	int32_t* targets;    // [w*h][nNeighbors]: pixel index A's pixel projects onto in neighbour q (SceneDensify.cpp:3387-3393) + what it can do there
	                     // (bits 29-30: merge / in front); -1 when it can do nothing
	uint32_t* head;      // [nNeighbors][stride]: per neighbour pixel, the first entry of the list of A's pending pixels that project onto it
	                     // (its "bidders"), 0xFFFFFFFF = none.  List entry e = q * (w*h) + pixel
	uint32_t* next;      // [nNeighbors][w*h]: the entry after entry e in its target's list, 0xFFFFFFFF = end
	size_t stride;       // pixels reserved per neighbour map in head
};

FuseTables fuse_tables(int32_t* targets, uint32_t* head, uint32_t* next, size_t stride);
// ctl: kCtlBytes, zero before the pass: [kCtlPending] pending pixels, [kCtlErr] the settle iteration gave up (never expected),
// [kCtlSteps] steps it took, [kCtlWork + s] length of the work list step s wrote (diagnostics)
constexpr int kSettleSteps = 1; // full-grid steps of the settle iteration after step 0; a single workgroup finishes what they leave
constexpr int kCtlPending = 4, kCtlSteps = 6, kCtlWork = 16, kCtlErr = 32; // word indices
constexpr size_t kCtlBytes = 256;
void launch_fuse_begin(const DevMap& A, const DevMap* maps, const FuseTables& tb, uint32_t* pending, uint32_t* ctl, uint8_t* flag,
                       unsigned long long* counters, float thDepth, float normalError, hipStream_t s);
// status words of a fusion enqueued without host synchronisation: [0] a pass gave up (never expected), [3] = 1 / 2: the cloud / the view
// lists overflowed their capacity
void launch_unclaim(const DevMap* maps, int nMaps, hipStream_t s); // takes the claim marks (negative depths) off every map
size_t fuse_settle_bytes(size_t pixels); // scratch of launch_fuse_pass for an image of that many pixels
void launch_fuse_pass(const DevMap& A, const DevMap* maps, const FuseTables& tb, const uint32_t* pending, void* settle, uint32_t* ctl,
                      float* oxyz, float* onormal, uint8_t* obgr, uint32_t* onv, uint8_t* oflag, uint32_t* oviews, float* oweights, int vstride,
                      uint32_t* merged, int nMinViewsFuse, int order, unsigned long long* counters, uint32_t* status, bool wantPoints, hipStream_t s);
// dF, dF2: w*h floats each, nF: 3*w*h floats of scratch
void launch_postfilter(int w, int h, float* depth, float* normal, float* conf, const DevMap* maps, int nMaps, const uint8_t* gra, float* dF, float* dF2,
                       float* nF, int gap, float thr, unsigned long long* filled, hipStream_t s);
void launch_point_colors(unsigned long long n, const float* xyz, const unsigned long long* voff, const uint32_t* views, const DevMap* maps, uint8_t* bgr, hipStream_t s);
size_t fuse_scan_temp_bytes(int n);
// bases: device words [0] points, [1] view entries written by the images before this one (null: base / viewBase)
void launch_fuse_compact(int n, const uint8_t* flag, uint32_t* flag32, uint32_t* pos, void* temp, size_t tempBytes, float* oxyz,
                         float* onormal, uint8_t* obgr, uint32_t* onv, unsigned long long base, unsigned long long capacity, float* xyz,
                         float* normal, uint8_t* bgr, uint32_t* nviews, uint32_t* oviews, float* oweights, int vstride, uint32_t* voff,
                         unsigned long long viewBase, unsigned long long viewCapacity, uint32_t* cviews, float* cweights, const unsigned long long* bases,
                         hipStream_t s);
// after an image's compaction: totals [0] points, [1] view entries, [2] depths so far; status[3] = 1 / 2 when the cloud / the view lists overflow
void launch_fuse_advance(const unsigned long long* counters, unsigned long long* totals, unsigned long long capacity, unsigned long long viewCapacity,
                         uint32_t* status, hipStream_t s);

} // namespace hcmvs
#endif
