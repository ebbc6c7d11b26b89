/*
 * hc-mvs_amd/csrc/pm_common.h -- structures shared by the host API and the gfx950 kernels.
 */
#ifndef HCMVS_PM_COMMON_H
#define HCMVS_PM_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hcmvs {

constexpr int kHalfWindow = 7;      // nSizeHalfWindow, DepthMap.h:354 (fixed border, DepthMap.cpp:442-447)
// The reference fixes nSizeHalfWindow = 7 / nTexels = 64 at compile time (DepthMap.h:354-358).  Patches beyond that
// (BASELINE.json configs[4]: 11 x 11 taps, adapthalfwin 10) generalise the two constants: the border every pass keeps
// clear follows the half window, EstConst::border = max(7, adapthalfwin).
constexpr int kMaxHalfWindow = 10;
constexpr int kBigTaps = (kMaxHalfWindow + 1) * (kMaxHalfWindow + 1); // 121
constexpr int kBigSlots = 128;      // tap slots of the big-patch tables (taps dealt round-robin to the lanes of a view group)
constexpr int kMaxViews = 16;
constexpr int kMaxSlots = 32;       // neighbour slots of one pixel (cross pattern: 4 * ceil(halfwin/step))
constexpr int kProgressStride = 16; // ints between the progress words of consecutive rows (64 B)

// per source view constants, one lane group reads its own view's entry (DepthMap.h:412-444 ViewData)
struct DevView {
	uint32_t byteOff; // start of the view's footprint image (launch_quads) relative to EstConst::imgBase
	int32_t w, h;
	float A[9];  // Hl * Hr = Kj Rj Ri^T Ki^-1 (computed in double on the host, held as float)
	float Hm[3]; // Kj Rj (Ci - Cj)
};

// uniform constants of one EstimateDepthMap call (DepthMap.cpp:386-439 DepthEstimator ctor)
struct EstConst {
	int32_t W, H, V;
	int32_t border;         // max(kHalfWindow, adapthalfwin): pixels closer than this to an edge are not estimated
	int32_t adapthalfwin, nRandomIters, itExternal, propHalfwin, propStep;
	const float* ref;
	const uint8_t* gra;
	const DevView* views;
	const char* imgBase;    // lowest source footprint-image address of this call (32-bit offsets from here)
	float Hr[9];            // Ki^-1 (double on the host, held as float)
	double cx, cy, ifx, ify; // reference principal point and 1/focal (Camera.h:299-312)
	float dMin, dMax, dMinSqr, dMaxSqr;
	float smoothBonusDepth, smoothBonusNormal, smoothSigmaDepth, smoothSigmaNormal;
	float angle1Range, angle2Range;
	float thConfSmall, thConfBig, thConfRand, thRobust, thKeep;
	float depthRatio, pfScale;
	uint32_t seed;
	// working state: (depth, nx, ny, nz) per pixel + score per pixel
	float4* dn;
	float* conf;
	int32_t* progress; // [rows * kProgressStride] pixels finished per logical row of this image (sweeps)
	// restore-variant extra hypothesis (restore/libs/MVS/DepthMap.cpp:1527-1549): in sweep number hintIter every pixel also tries
	// the estimate of the up-sampled coarser level, with a 0.1 bonus; null / -1 when not in use
	const float* hintDepth;
	const float* hintNormal;
	int32_t hintIter;
};

struct SweepSync {
	int32_t* ticket;   // [kMaxBatch] next row of every image of the batch, counted through the sweeps of the launch (row + sweep * rows)
	int32_t* rowsDone; // [kMaxBatch] rows every image has finished, counted through the sweeps of the launch
	int32_t* error;    // set non-zero when a worker times out
	unsigned long long* evals;  // [0] ScorePixel calls of the sequential algorithm, [1] evaluations issued (incl. speculative), [2] patch taps of [0]
};

// lane layout class for V source views (the items of a batch must share it): 8 for up to 8 views, 4 for 9..16 views, which run
// the 8 x 8 lane layout twice (two sets of eight view groups)
int segments_for(int V);

// launch wrappers (pm_kernels.hip)
void launch_gray_to_u8(const float* gray, uint8_t* out, int n, hipStream_t s);
void launch_bgr_to_u8(const uint8_t* bgr, uint8_t* out, int n, hipStream_t s);
void launch_gradient_map(const uint8_t* g8, uint8_t* gra, int W, int H, hipStream_t s);
void launch_median3(const float* in, float* out, int W, int H, hipStream_t s);
void launch_quads(const float* gray, float4* out, int W, int H, hipStream_t s); // 2 x 2 footprint layout of a source view
void launch_score_pass(const EstConst& c, const float* depthIn, const float* normalIn, unsigned long long* evals,
                       hipStream_t s);
void launch_sweep(const EstConst* dItems, int nItems, int maxRows, int totalRows, int V, bool bigPatch, bool hint, const SweepSync& sync, int iter, int nSweeps,
                  int lag, int wavesPerRow, int affinity, int segLen, hipStream_t s); // segLen > 0: tickets are stretches of segLen columns of a row
void launch_end_pass(const EstConst& c, int finalPass, float* depth, float* normal, float* conf, hipStream_t s);

} // namespace hcmvs
#endif
