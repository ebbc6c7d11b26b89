/*
 * hc-mvs_amd/csrc/pf_chain.h -- the post-filter chain's incremental fusion (pf_kernels.hip): state kept from one whole-scene fusion to
 * the next.
 */
#ifndef HCMVS_PF_CHAIN_H
#define HCMVS_PF_CHAIN_H

#include "fuse_common.h"

namespace hcmvs {

constexpr uint16_t kPfNone = 0xFFFFu; // own[]: the estimate belongs to no point
constexpr int kCtlTouched = 8;        // ctl word: pixels evaluated in the pass

// per image, indexed by image id like the DevMap table.  Target side (every image with maps): own / chgNow / valNext.  Seed side (the
// images of the fusion order): the rest, null otherwise.
struct PfImage {
	uint16_t* own;     // [w*h] index (position in the fusion order) of the pass whose point the estimate belongs to, kPfNone = free
	uint8_t* chgNow;   // [w*h] the estimate changed (value or ownership) since the pass now running looked at it in the previous fusion
	uint8_t* valNext;  // [w*h] its VALUE changed during this fusion / the gap interpolation after it: what the next fusion starts from
	uint32_t* anyChg;  // one word: some chgNow byte of the image is set (a pass none of whose images has one has nothing to do)
	uint32_t* tgt;     // [w*h][nNeighbors] where the pixel projects in neighbour q, class, list bookkeeping (pf_kernels.hip)
	uint32_t* head;    // [nNeighbors][stride] first entry of the list of bidders per neighbour pixel, 0xFFFFFFFF = none
	uint32_t* next;    // [2][nNeighbors][w*h] list links; entry e = bank * nNeighbors * w*h + q * w*h + pixel
	uint8_t* acc;      // [w*h] the pixel is a point (its last evaluation)
	uint32_t *mm, *fm; // [w*h] bit q: it merges / lies in front of its target in neighbour q (its last evaluation)
	uint32_t* stamp;   // [w*h] work-list membership of the settle iteration
	uint32_t* touch;   // [w*h] fusion in which the pixel was last evaluated
	size_t stride;     // pixels reserved per neighbour in head
};

size_t pf_pass_scratch_bytes(size_t pixels);
void launch_pf_roll(const DevMap* maps, const PfImage* pf, int nMaps, uint32_t* anyChgWords, hipStream_t s); // anyChgWords: the nMaps words PfImage::anyChg point into
void launch_pf_pass(const DevMap& A, int i, const DevMap* maps, const PfImage* dPf, const PfImage& hostPfA, bool first, float thDepth, float normalError,
                    int nMinViewsFuse, uint32_t fusionIndex, void* scratch, uint32_t* ctl, uint32_t* status, hipStream_t s);
void launch_pf_filter(int w, int h, float* depth, float* normal, float* conf, const PfImage& hostPf, const uint8_t* gra, float* dF, float* dF2, float* nF, int gap,
                      float thr, unsigned long long* filled, hipStream_t s);

} // namespace hcmvs
#endif
