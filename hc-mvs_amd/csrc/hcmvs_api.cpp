/*
 * hc-mvs_amd/csrc/hcmvs_api.cpp -- host side of the C-ABI declared in include/hcmvs_hip.h.
 *
 * Holds the per-device context (views resident in HBM, working maps, row-progress words), derives the
 * per-call constants the way DepthEstimator's constructor does (DepthMap.cpp:386-439, DepthMap.h:412-444)
 * and enqueues the kernels of pm_kernels.hip.  No compute happens on the host; without a usable HIP
 * device every entry point fails.
 */
#include "../../include/hcmvs_hip.h"
#include "pm_common.h"
#include "fuse_common.h"
#include "pf_chain.h"
#include "tri_init.h"
#include "cloud_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <algorithm>
#include <chrono>
#include <vector>

using namespace hcmvs;

namespace {

constexpr int kMaxBatch = HCMVS_MAX_BATCH; // reference images estimated by one call

struct View {
	int w = 0, h = 0;
	float* gray = nullptr;   // device
	uint8_t* bgr = nullptr;  // device or null
	uint8_t* gra = nullptr;  // device gradient map (lazy)
	float4* quads = nullptr; // device 2 x 2 footprint layout of the gray image, built when the view first serves as a source view
	bool owned = false;
	double K[9], R[9], C[3];
	// estimated maps registered for filter / fuse
	float *mDepth = nullptr, *mNormal = nullptr, *mConf = nullptr;
	bool mapsOwned = false;
	float dMin = 0.f, dMax = 0.f;
	uint32_t* dNeighbors = nullptr;
	std::vector<uint32_t> neighbors;
};

} // namespace


struct hcmvs_ctx {
	int device = 0;
	hipStream_t ownStream = nullptr;
	hipStream_t stream = nullptr;
	std::string err;
	std::map<uint32_t, View> views;
	// working buffers of the batch items (grown on demand)
	struct Slot {
		size_t capPixels = 0; float4* dn = nullptr; float* conf = nullptr; float* tmpDepth = nullptr;
		size_t capRows = 0; int32_t* progress = nullptr;
		size_t capSlab = 0; char* srcSlab = nullptr; // compact copy of an item's source images when they are > 4 GiB apart
	};
	std::vector<Slot> slots;
	size_t capPixels = 0;
	uint8_t* tmpU8 = nullptr; // gradient-map staging
	// host-path staging
	size_t capStage = 0;
	float *sDepth = nullptr, *sNormal = nullptr, *sConf = nullptr;
	DevView* dViews = nullptr;           // [kMaxBatch][kMaxViews]
	std::vector<DevView> hViews;          // host copy handed to hipMemcpyAsync (must outlive the call)
	EstConst* dItems = nullptr;           // [kMaxBatch]
	std::vector<EstConst> hItems;
	int32_t* sync = nullptr;              // [0] unused, [1] error word, [16 .. 16 + kMaxBatch) row tickets of the batch items, then kMaxBatch rows-done counters
	int sweepPerLaunch = 0;               // HCMVS_SWEEP_LAUNCHES: 0 automatic (one launch for all sweeps from 16 images on), 1 per-sweep, 2 one
	unsigned long long* evals = nullptr;
	hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
	int lastSweeps = 0, lastSweepLaunches = 0;
	bool haveStats = false;
	int sweepLag = 1;
	int sweepSegment = -1; // columns per ticket of the sweep worker: -1 = automatic (see the launch of the sweeps), 0 = whole rows
	int fuseOrder = 0; // hcmvs_set_fuse_order
	int xcdAffinity = 1; // rows of an image prefer the workgroups of one XCD (HCMVS_XCD_AFFINITY=0 turns it off)
	// filter / fuse scratch
	DevMap* dMaps = nullptr; size_t capMaps = 0;
	unsigned long long* counters = nullptr;
	void* fuseScratch = nullptr; size_t capFuseScratch = 0;
	char* passScratch = nullptr; size_t capPass = 0; // per-pass tables of the fusion (hcmvs_fuse_cloud, hcmvs_postfilter_sequence)
	char* pfState = nullptr; size_t capPf = 0;       // the post-filter chain's state kept from fusion to fusion (pf_kernels.hip)
	bool errPending = false; // an estimate was enqueued since the error word was last read
	int nCU = 0;          // compute units of the device: 4 SIMDs x 3 sweep workers each (the waves-per-row policy)
	hipEvent_t upEv[2] = {nullptr, nullptr};
	char* pinned = nullptr; size_t capPinned = 0; // page-locked staging of the host-buffer uploads (a pageable hipMemcpy crawls at ~1.3 GB/s here)
	int wavesPerRow = 0; // 0 = automatic: 3 waves per row for one image, 2 for two (latency), 1 when >= 3 images fill the chip
};

static int fail(hcmvs_ctx* c, int code, const char* fmt, ...) {
	if (c) {
		char buf[512];
		va_list ap;
		va_start(ap, fmt);
		vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		c->err = buf;
	}
	return code;
}
#define HIPCHK(c, call)                                                                             \
	do {                                                                                            \
		hipError_t e_ = (call);                                                                     \
		if (e_ != hipSuccess) return fail(c, HCMVS_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
	} while (0)

static void mat3_mul(const double* a, const double* b, double* c) {
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double s = 0;
			for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
			c[i * 3 + j] = s;
		}
}
static void mat3_mul_bt(const double* a, const double* b, double* c) {
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			double s = 0;
			for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[j * 3 + k];
			c[i * 3 + j] = s;
		}
}
static void mat3_inv(const double* m, double* r) {
	const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
	                 m[2] * (m[3] * m[7] - m[4] * m[6]);
	const double id = 1.0 / d;
	r[0] = (m[4] * m[8] - m[5] * m[7]) * id;
	r[1] = (m[2] * m[7] - m[1] * m[8]) * id;
	r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
	r[3] = (m[5] * m[6] - m[3] * m[8]) * id;
	r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
	r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
	r[6] = (m[3] * m[7] - m[4] * m[6]) * id;
	r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
	r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}
static inline float fd2r(float d) { return d * (3.14159274101257324f / 180.f); }

#ifdef HCMVS_STAMPS
namespace hcmvs { void debug_read_stamps(unsigned long long* out, int reset); }
#endif
namespace hcmvs { void launch_resize_gray(const float* src, int sw, int sh, float* dst, int dw, int dh, float scaleParam, hipStream_t s); } // img_kernels.hip

extern "C" {

void hcmvs_default_params(hcmvs_params* p) {
	if (!p) return;
	memset(p, 0, sizeof *p);
	p->adapthalfwin = 5;           // DensifyPointCloud.cpp:163
	p->n_estimation_iters = 1;
	p->it_external = 0;
	p->n_external_iters = 1;
	p->propagate_halfwin = 1;
	p->propagate_step = 4;
	p->n_random_iters = 6;         // DepthMap.cpp:120
	p->ncc_threshold_keep = 0.55f; // DepthMap.cpp:117
	p->random_depth_ratio = 0.003f;
	p->random_angle1_deg = 16.f;
	p->random_angle2_deg = 10.f;
	p->random_smooth_depth = 0.02f;
	p->random_smooth_normal_deg = 13.f;
	p->random_smooth_bonus = 0.93f;
	p->photometric_flow = 0.f;
	p->seed = 1234;
	p->median_blur = 1;
}

int hcmvs_create(int device, hcmvs_ctx** out) {
	if (!out) return HCMVS_ERR_INVALID;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return HCMVS_ERR_NO_DEVICE;
	if (hipSetDevice(device) != hipSuccess) return HCMVS_ERR_NO_DEVICE;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HCMVS_ERR_NO_DEVICE;
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HCMVS_ERR_NO_DEVICE; // kernels are built for gfx950 only
	hcmvs_ctx* c = new hcmvs_ctx();
	c->device = device;
	c->nCU = prop.multiProcessorCount;
	if (hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess) { delete c; return HCMVS_ERR_NO_DEVICE; }
	c->stream = c->ownStream;
	for (auto& e : c->ev)
		if (hipEventCreate(&e) != hipSuccess) { delete c; return HCMVS_ERR_NO_DEVICE; }
	c->hViews.resize((size_t)kMaxBatch * kMaxViews); c->hItems.resize(kMaxBatch);
	if (hipMalloc(&c->dViews, sizeof(DevView) * kMaxViews * kMaxBatch) != hipSuccess || hipMalloc(&c->evals, 32) != hipSuccess ||
	    hipMalloc(&c->dItems, sizeof(EstConst) * kMaxBatch) != hipSuccess || hipMalloc(&c->sync, 64 + sizeof(int32_t) * 2 * kMaxBatch) != hipSuccess) {
		delete c;
		return HCMVS_ERR_NO_DEVICE;
	}
	const char* lag = getenv("HCMVS_SWEEP_LAG");
	if (lag && atoi(lag) >= 1) c->sweepLag = atoi(lag);
	const char* aff = getenv("HCMVS_XCD_AFFINITY");
	if (aff) c->xcdAffinity = atoi(aff) != 0;
	const char* spl = getenv("HCMVS_SWEEP_LAUNCHES");
	if (spl && strcmp(spl, "per-sweep") == 0) c->sweepPerLaunch = 1;
	if (spl && strcmp(spl, "one") == 0) c->sweepPerLaunch = 2;
	const char* sgl = getenv("HCMVS_SWEEP_SEGMENT"); // tuning knob: columns per ticket (0 = whole rows)
	if (sgl && atoi(sgl) >= 0) c->sweepSegment = atoi(sgl);
	const char* wpr = getenv("HCMVS_WAVES_PER_ROW"); // tuning knob: 1, 2, 3 or 4 waves cooperate on one image row
	if (wpr && (atoi(wpr) == 1 || atoi(wpr) == 2 || atoi(wpr) == 3 || atoi(wpr) == 4)) c->wavesPerRow = atoi(wpr);
	*out = c;
	return HCMVS_OK;
}

static void free_maps(View& v) {
	if (v.mapsOwned) for (void* p : {(void*)v.mDepth, (void*)v.mNormal, (void*)v.mConf}) if (p) (void)hipFree(p);
	v.mDepth = v.mNormal = v.mConf = nullptr; v.mapsOwned = false;
}
static void free_view(View& v) {
	if (v.owned) { if (v.gray) (void)hipFree(v.gray); if (v.bgr) (void)hipFree(v.bgr); }
	if (v.gra) (void)hipFree(v.gra);
	if (v.quads) (void)hipFree(v.quads);
	if (v.dNeighbors) (void)hipFree(v.dNeighbors);
	free_maps(v);
	v = View();
}

void hcmvs_destroy(hcmvs_ctx* c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (auto& kv : c->views) free_view(kv.second);
	for (auto& sl : c->slots) for (void* p : {(void*)sl.dn, (void*)sl.conf, (void*)sl.tmpDepth, (void*)sl.progress, (void*)sl.srcSlab}) if (p) (void)hipFree(p);
	for (void* p : {(void*)c->tmpU8, (void*)c->sDepth, (void*)c->sNormal, (void*)c->sConf, (void*)c->dViews, (void*)c->dItems, (void*)c->sync,
	                (void*)c->evals, (void*)c->dMaps, (void*)c->counters, c->fuseScratch})
		if (p) (void)hipFree(p);
	if (c->passScratch) (void)hipFree(c->passScratch);
	if (c->pfState) (void)hipFree(c->pfState);
	for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
	if (c->pinned) (void)hipHostFree(c->pinned);
	for (auto& e : c->upEv) if (e) (void)hipEventDestroy(e);
	if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
	delete c;
}

const char* hcmvs_last_error(const hcmvs_ctx* c) { return c ? c->err.c_str() : "null context"; }

// A sweep worker that gave up waiting on its predecessor row leaves the error word set (pm_kernels.hip wait_progress):
// every synchronising entry point reports it, so a caller of the asynchronous *_device entries learns of it without a
// separate hcmvs_get_stats.  The stream must be idle.
static int check_sweep_error(hcmvs_ctx* c) {
	if (!c->errPending) return HCMVS_OK;
	int32_t flags[2] = {0, 0};
	HIPCHK(c, hipMemcpy(flags, c->sync, sizeof flags, hipMemcpyDeviceToHost));
	c->errPending = false;
	if (flags[1] != 0) return fail(c, HCMVS_ERR_TIMEOUT, "sweep worker timed out waiting for its predecessor row; the maps of the last estimate are incomplete");
	return HCMVS_OK;
}

int hcmvs_set_stream(hcmvs_ctx* c, void* stream) {
	if (!c) return HCMVS_ERR_INVALID;
	c->stream = stream ? (hipStream_t)stream : c->ownStream;
	return HCMVS_OK;
}
int hcmvs_synchronize(hcmvs_ctx* c) {
	if (!c) return HCMVS_ERR_INVALID;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return check_sweep_error(c);
}

// host -> device through the context's page-locked staging buffer, in pieces: the caller's buffer is pageable, and a pageable
// hipMemcpy is an order of magnitude slower than memcpy + a pinned transfer
static int upload_staged(hcmvs_ctx* c, void* dst, const void* src, size_t bytes) {
	{ // a caller that already holds the data in page-locked memory (hipHostMalloc / hipHostRegister) needs no staging
		hipPointerAttribute_t at;
		if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost) {
			HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			return HCMVS_OK;
		}
		(void)hipGetLastError(); // pageable memory is "invalid value" to the query
	}
	const size_t piece = (size_t)32 << 20;
	if (c->capPinned < 2 * piece) {
		if (c->pinned) (void)hipHostFree(c->pinned);
		c->pinned = nullptr; c->capPinned = 0;
		if (hipHostMalloc((void**)&c->pinned, 2 * piece, hipHostMallocDefault) != hipSuccess) { // no pinned memory to be had: the plain copy
			(void)hipGetLastError();
			HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			return HCMVS_OK;
		}
		c->capPinned = 2 * piece;
	}
	for (auto& e : c->upEv) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
	hipEvent_t done[2] = {c->upEv[0], c->upEv[1]};
	bool used[2] = {false, false};
	int k = 0;
	for (size_t off = 0; off < bytes; off += piece, k ^= 1) {
		const size_t n = std::min(piece, bytes - off);
		if (used[k]) HIPCHK(c, hipEventSynchronize(done[k])); // the half I am about to overwrite has left
		memcpy(c->pinned + (size_t)k * piece, (const char*)src + off, n);
		HIPCHK(c, hipMemcpyAsync((char*)dst + off, c->pinned + (size_t)k * piece, n, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipEventRecord(done[k], c->stream));
		used[k] = true;
	}
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return HCMVS_OK;
}

static int set_view(hcmvs_ctx* c, uint32_t id, int w, int h, const float* gray, const uint8_t* bgr, const double* K,
                    const double* R, const double* C, bool copy) {
	if (!c) return HCMVS_ERR_INVALID;
	// gray may be null when a colour image is given: a view that is only fused / filtered (camera, colours, gradient map), never the
	// reference or a source view of an estimate -- what a rank of a multi-GPU job holds of the images other ranks estimate
	if ((!gray && !bgr) || !K || !R || !C || w < 2 * kHalfWindow + 2 || h < 2 * kHalfWindow + 2 || w > 32768 || h > 32768 || id >= 65536)
		return fail(c, HCMVS_ERR_INVALID, "upload_view: bad arguments (id %u, %dx%d)", id, w, h);
	HIPCHK(c, hipSetDevice(c->device));
	auto it = c->views.find(id);
	if (it != c->views.end()) { HIPCHK(c, hipStreamSynchronize(c->stream)); free_view(it->second); }
	View v;
	v.w = w; v.h = h; v.owned = copy;
	const size_t n = (size_t)w * h;
	if (copy) {
		if (gray) {
			HIPCHK(c, hipMalloc(&v.gray, n * sizeof(float)));
			const int rc = upload_staged(c, v.gray, gray, n * sizeof(float));
			if (rc) return rc;
		}
		if (bgr) {
			HIPCHK(c, hipMalloc(&v.bgr, n * 3));
			const int rc = upload_staged(c, v.bgr, bgr, n * 3);
			if (rc) return rc;
		} // (upload_staged has synchronised: the caller may free its host buffers on return)
	} else {
		v.gray = const_cast<float*>(gray);
		v.bgr = const_cast<uint8_t*>(bgr);
	}
	memcpy(v.K, K, sizeof v.K); memcpy(v.R, R, sizeof v.R); memcpy(v.C, C, sizeof v.C);
	c->views[id] = v;
	return HCMVS_OK;
}

int hcmvs_upload_view(hcmvs_ctx* c, uint32_t id, int32_t w, int32_t h, const float* gray, const uint8_t* bgr,
                      const double K[9], const double R[9], const double C[3]) {
	return set_view(c, id, w, h, gray, bgr, K, R, C, true);
}
int hcmvs_set_view_device(hcmvs_ctx* c, uint32_t id, int32_t w, int32_t h, const float* gray, const uint8_t* bgr,
                          const double K[9], const double R[9], const double C[3]) {
	return set_view(c, id, w, h, gray, bgr, K, R, C, false);
}
int hcmvs_rescale_view(hcmvs_ctx* c, uint32_t src_id, uint32_t dst_id, float scale) {
	if (!c) return HCMVS_ERR_INVALID;
	auto it = c->views.find(src_id);
	if (it == c->views.end() || dst_id >= 65536 || dst_id == src_id) return fail(c, HCMVS_ERR_INVALID, "rescale_view: bad view ids %u -> %u", src_id, dst_id);
	if (!(scale > 0.f) || fabsf(scale - 1.f) < 0.15f) return fail(c, HCMVS_ERR_INVALID, "rescale_view: scale %g is within 15 %% of 1 (DepthMap.h:234: not resampled)", scale);
	const View src = it->second;
	if (!src.gray) return fail(c, HCMVS_ERR_INVALID, "rescale_view: view %u has no gray image (fuse-only view)", src_id);
	// cv::resize with dsize empty: Size(saturate_cast<int>(w * fx), saturate_cast<int>(h * fy)), saturate_cast<int>(double) = cvRound
	const int nw = (int)lrint((double)src.w * (double)scale), nh = (int)lrint((double)src.h * (double)scale);
	if (nw < 2 * kHalfWindow + 2 || nh < 2 * kHalfWindow + 2 || nw > 32768 || nh > 32768) return fail(c, HCMVS_ERR_INVALID, "rescale_view: %dx%d x %g gives an unusable size", src.w, src.h, scale);
	HIPCHK(c, hipSetDevice(c->device));
	auto old = c->views.find(dst_id);
	if (old != c->views.end()) { HIPCHK(c, hipStreamSynchronize(c->stream)); free_view(old->second); c->views.erase(old); }
	View v;
	v.w = nw; v.h = nh; v.owned = true;
	HIPCHK(c, hipMalloc(&v.gray, (size_t)nw * nh * sizeof(float)));
	hcmvs::launch_resize_gray(src.gray, src.w, src.h, v.gray, nw, nh, scale, c->stream);
	HIPCHK(c, hipGetLastError());
	// Image::GetCamera(platforms, size): K normalised by max(w, h) of the image, scaled to max(w', h') (Camera.h:167-180)
	const double f = (double)std::max(nw, nh) / (double)std::max(src.w, src.h);
	memcpy(v.K, src.K, sizeof v.K); memcpy(v.R, src.R, sizeof v.R); memcpy(v.C, src.C, sizeof v.C);
	v.K[0] *= f; v.K[4] *= f; v.K[2] *= f; v.K[5] *= f;
	c->views[dst_id] = v;
	return HCMVS_OK;
}
int hcmvs_get_view_info(hcmvs_ctx* c, uint32_t id, int32_t* w, int32_t* h, double K[9]) {
	if (!c) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "get_view_info: unknown view %u", id);
	if (w) *w = it->second.w;
	if (h) *h = it->second.h;
	if (K) memcpy(K, it->second.K, sizeof(double) * 9);
	return HCMVS_OK;
}
int hcmvs_get_view_gray(hcmvs_ctx* c, uint32_t id, float* out) {
	if (!c || !out) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end() || !it->second.gray) return fail(c, HCMVS_ERR_INVALID, "get_view_gray: unknown view %u (or a fuse-only view)", id);
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemcpyAsync(out, it->second.gray, (size_t)it->second.w * it->second.h * sizeof(float), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return HCMVS_OK;
}

int hcmvs_release_view(hcmvs_ctx* c, uint32_t id) {
	if (!c) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "release_view: unknown view %u", id);
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	free_view(it->second);
	c->views.erase(it);
	return HCMVS_OK;
}

static int ensure_slot(hcmvs_ctx* c, int i, size_t n, int rows) {
	if ((int)c->slots.size() <= i) c->slots.resize((size_t)i + 1);
	hcmvs_ctx::Slot& sl = c->slots[i];
	if (n > sl.capPixels) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		for (void* p : {(void*)sl.dn, (void*)sl.conf, (void*)sl.tmpDepth}) if (p) (void)hipFree(p);
		sl.dn = nullptr; sl.conf = nullptr; sl.tmpDepth = nullptr; sl.capPixels = 0;
		HIPCHK(c, hipMalloc(&sl.dn, n * sizeof(float4)));
		HIPCHK(c, hipMalloc(&sl.conf, n * sizeof(float)));
		HIPCHK(c, hipMalloc(&sl.tmpDepth, n * sizeof(float)));
		sl.capPixels = n;
	}
	if ((size_t)rows > sl.capRows) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if (sl.progress) (void)hipFree(sl.progress);
		sl.progress = nullptr; sl.capRows = 0;
		HIPCHK(c, hipMalloc(&sl.progress, (size_t)rows * kProgressStride * sizeof(int32_t)));
		sl.capRows = (size_t)rows;
	}
	return HCMVS_OK;
}
static int ensure_u8(hcmvs_ctx* c, size_t n) {
	if (n > c->capPixels) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if (c->tmpU8) (void)hipFree(c->tmpU8);
		c->tmpU8 = nullptr; c->capPixels = 0;
		HIPCHK(c, hipMalloc(&c->tmpU8, n));
		c->capPixels = n;
	}
	return HCMVS_OK;
}

// SceneDensify.cpp:581-595 InitGraMap, computed once per view on the device
static int ensure_gradient(hcmvs_ctx* c, View& v) {
	if (v.gra) return HCMVS_OK;
	const int n = v.w * v.h;
	int rc = ensure_u8(c, (size_t)n);
	if (rc) return rc;
	HIPCHK(c, hipMalloc(&v.gra, (size_t)n));
	if (v.bgr) launch_bgr_to_u8(v.bgr, c->tmpU8, n, c->stream);
	else launch_gray_to_u8(v.gray, c->tmpU8, n, c->stream);
	launch_gradient_map(c->tmpU8, v.gra, v.w, v.h, c->stream);
	HIPCHK(c, hipGetLastError());
	return HCMVS_OK;
}

// the layout the scorer samples a source view from (pm_kernels.hip quad_kernel), built once per view
static int ensure_quads(hcmvs_ctx* c, View& v) {
	if (v.quads) return HCMVS_OK;
	HIPCHK(c, hipMalloc(&v.quads, (size_t)v.w * v.h * sizeof(float4)));
	launch_quads(v.gray, v.quads, v.w, v.h, c->stream);
	HIPCHK(c, hipGetLastError());
	return HCMVS_OK;
}

int hcmvs_get_gradient_map(hcmvs_ctx* c, uint32_t id, uint8_t* out) {
	if (!c || !out) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "get_gradient_map: unknown view %u", id);
	HIPCHK(c, hipSetDevice(c->device));
	int rc = ensure_gradient(c, it->second);
	if (rc) return rc;
	HIPCHK(c, hipMemcpyAsync(out, it->second.gra, (size_t)it->second.w * it->second.h, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return HCMVS_OK;
}

// per-item constants: DepthMap.cpp:386-439, DepthMap.h:412-444
static int build_item(hcmvs_ctx* c, int slot, const hcmvs_batch_item& it, const hcmvs_params* p, EstConst& k) {
	if (!it.src_ids || !it.d_depth || !it.d_normal || !it.d_conf) return fail(c, HCMVS_ERR_INVALID, "estimate: null argument");
	const int n_src = it.n_src;
	if (n_src < 1 || n_src > HCMVS_MAX_VIEWS) return fail(c, HCMVS_ERR_INVALID, "estimate: n_src %d not in 1..%d", n_src, HCMVS_MAX_VIEWS);
	// d_min == 0 is what the `restore` variant ends up with (its range takes the enlarged previous-level map in, zeros included,
	// restore/libs/MVS/SceneDensify.cpp:526-532)
	if (!(it.d_min >= 0.f) || !(it.d_max > it.d_min)) return fail(c, HCMVS_ERR_INVALID, "estimate: bad depth range [%g,%g)", it.d_min, it.d_max);
	auto rit = c->views.find(it.ref_id);
	if (rit == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "estimate: unknown reference view %u", it.ref_id);
	View& ref = rit->second;
	if (!ref.gray) return fail(c, HCMVS_ERR_INVALID, "estimate: view %u was registered without a gray image (fuse-only view)", it.ref_id);
	int rc = ensure_slot(c, slot, (size_t)ref.w * ref.h, ref.h);
	if (rc) return rc;
	rc = ensure_gradient(c, ref);
	if (rc) return rc;
	memset(&k, 0, sizeof k);
	k.W = ref.w; k.H = ref.h; k.V = n_src;
	k.border = p->adapthalfwin > kHalfWindow ? p->adapthalfwin : kHalfWindow; // nSizeHalfWindow generalised (pm_common.h)
	if (ref.w < 2 * k.border + 2 || ref.h < 2 * k.border + 2) return fail(c, HCMVS_ERR_INVALID, "estimate: view %u (%dx%d) has no pixel inside the %d px border", it.ref_id, ref.w, ref.h, k.border);
	k.adapthalfwin = p->adapthalfwin; k.nRandomIters = p->n_random_iters; k.itExternal = p->it_external;
	k.propHalfwin = p->propagate_halfwin; k.propStep = p->propagate_step;
	k.ref = ref.gray; k.gra = ref.gra; k.views = c->dViews + (size_t)slot * kMaxViews;
	double Hr[9];
	mat3_inv(ref.K, Hr);
	for (int i = 0; i < 9; ++i) k.Hr[i] = (float)Hr[i];
	k.ifx = 1.0 / ref.K[0]; k.ify = 1.0 / ref.K[4]; k.cx = ref.K[2]; k.cy = ref.K[5];
	DevView* hv = c->hViews.data() + (size_t)slot * kMaxViews;
	memset(hv, 0, sizeof(DevView) * kMaxViews);
	uintptr_t lo = ~(uintptr_t)0, hi = 0;
	for (int v = 0; v < n_src; ++v) {
		auto sit = c->views.find(it.src_ids[v]);
		if (sit == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "estimate: unknown source view %u", it.src_ids[v]);
		View& s = sit->second;
		if (!s.gray) return fail(c, HCMVS_ERR_INVALID, "estimate: source view %u was registered without a gray image (fuse-only view)", it.src_ids[v]);
		rc = ensure_quads(c, s);
		if (rc) return rc;
		double KR[9], Hl[9], A[9];
		mat3_mul(s.K, s.R, KR);
		mat3_mul_bt(KR, ref.R, Hl);
		const double dC[3] = {ref.C[0] - s.C[0], ref.C[1] - s.C[1], ref.C[2] - s.C[2]};
		for (int i = 0; i < 3; ++i) hv[v].Hm[i] = (float)(KR[i * 3] * dC[0] + KR[i * 3 + 1] * dC[1] + KR[i * 3 + 2] * dC[2]);
		mat3_mul(Hl, Hr, A);
		for (int i = 0; i < 9; ++i) hv[v].A[i] = (float)A[i];
		hv[v].w = s.w; hv[v].h = s.h;
		const uintptr_t b = (uintptr_t)s.quads, e = b + (size_t)s.w * s.h * sizeof(float4);
		if (b < lo) lo = b;
		if (e > hi) hi = e;
	}
	// 32-bit texel offsets from one scalar base: the source views of one item must lie within a 4 GiB window.
	// When the caller's images are further apart they are first copied (device to device, a few tens of MB) into a
	// compact slab owned by the context.
	if (hi - lo < 0xFFFF0000ull) {
		k.imgBase = (const char*)lo;
		for (int v = 0; v < n_src; ++v) hv[v].byteOff = (uint32_t)((uintptr_t)c->views.find(it.src_ids[v])->second.quads - lo);
	} else {
		size_t total = 0;
		for (int v = 0; v < n_src; ++v) { const View& s = c->views.find(it.src_ids[v])->second; total += ((size_t)s.w * s.h * sizeof(float4) + 255) & ~(size_t)255; }
		if (total >= 0xFFFF0000ull) return fail(c, HCMVS_ERR_INVALID, "estimate: the source images of one item exceed 4 GiB");
		hcmvs_ctx::Slot& sl = c->slots[slot];
		if (total > sl.capSlab) {
			HIPCHK(c, hipStreamSynchronize(c->stream));
			if (sl.srcSlab) (void)hipFree(sl.srcSlab);
			sl.srcSlab = nullptr; sl.capSlab = 0;
			HIPCHK(c, hipMalloc(&sl.srcSlab, total));
			sl.capSlab = total;
		}
		size_t off = 0;
		for (int v = 0; v < n_src; ++v) {
			const View& s = c->views.find(it.src_ids[v])->second;
			const size_t bytes = (size_t)s.w * s.h * sizeof(float4);
			HIPCHK(c, hipMemcpyAsync(sl.srcSlab + off, s.quads, bytes, hipMemcpyDeviceToDevice, c->stream));
			hv[v].byteOff = (uint32_t)off;
			off += (bytes + 255) & ~(size_t)255;
		}
		k.imgBase = sl.srcSlab;
	}
	k.dMin = it.d_min; k.dMax = it.d_max; k.dMinSqr = sqrtf(it.d_min); k.dMaxSqr = sqrtf(it.d_max);
	k.smoothBonusDepth = 1.f - p->random_smooth_bonus;
	k.smoothBonusNormal = (1.f - p->random_smooth_bonus) * 0.96f;
	k.smoothSigmaDepth = -1.f / (2.f * (p->random_smooth_depth * p->random_smooth_depth));
	{ const float r = fd2r(p->random_smooth_normal_deg); k.smoothSigmaNormal = -1.f / (2.f * (r * r)); }
	k.angle1Range = fd2r(p->random_angle1_deg);
	k.angle2Range = fd2r(p->random_angle2_deg);
	k.thConfSmall = p->ncc_threshold_keep * 0.2f;
	k.thConfBig = p->ncc_threshold_keep * 0.4f;
	k.thConfRand = p->ncc_threshold_keep * 0.9f;
	k.thRobust = p->ncc_threshold_keep * 1.2f;
	k.thKeep = p->ncc_threshold_keep;
	k.depthRatio = p->random_depth_ratio;
	k.pfScale = 1.f - p->photometric_flow;
	k.seed = p->seed + it.seed_offset;
	k.dn = c->slots[slot].dn; k.conf = c->slots[slot].conf; k.progress = c->slots[slot].progress;
	k.hintDepth = nullptr; k.hintNormal = nullptr; k.hintIter = -1;
	if (it.d_hint_depth && it.d_hint_normal && p->it_external == p->n_external_iters - 1) { // restore/libs/MVS/DepthMap.cpp:1527
		k.hintDepth = it.d_hint_depth; k.hintNormal = it.d_hint_normal; k.hintIter = p->n_estimation_iters - 1;
	}
	return HCMVS_OK;
}

int hcmvs_estimate_batch_device(hcmvs_ctx* c, const hcmvs_batch_item* items, int32_t n_items, const hcmvs_params* p) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!items || !p) return fail(c, HCMVS_ERR_INVALID, "estimate: null argument");
	if (n_items < 1 || n_items > kMaxBatch) return fail(c, HCMVS_ERR_INVALID, "estimate: batch size %d not in 1..%d", n_items, kMaxBatch);
	if (p->adapthalfwin < 1 || p->adapthalfwin > kMaxHalfWindow) return fail(c, HCMVS_ERR_INVALID, "estimate: adapthalfwin %d not in 1..%d", p->adapthalfwin, kMaxHalfWindow);
	if (p->n_estimation_iters < 0 || p->n_random_iters < 0 || p->n_random_iters > 20)
		return fail(c, HCMVS_ERR_INVALID, "estimate: bad iteration counts");
	HIPCHK(c, hipSetDevice(c->device));
	int maxRows = 0, totalRows = 0;
	for (int i = 0; i < n_items; ++i) {
		if (items[i].n_src < 1 || items[i].n_src > kMaxViews || hcmvs::segments_for(items[i].n_src) != hcmvs::segments_for(items[0].n_src))
			return fail(c, HCMVS_ERR_INVALID, "estimate: the items of a batch must use source-view counts of one class (1-8 or 9-16)");
		int rc = build_item(c, i, items[i], p, c->hItems[i]);
		if (rc) return rc;
		const int rows = c->hItems[i].H - 2 * c->hItems[i].border;
		if (rows > maxRows) maxRows = rows;
		totalRows += rows;
	}
	hipStream_t s = c->stream;
	// same stream => the previous call's kernels are done with these tables before the copies land
	HIPCHK(c, hipMemcpyAsync(c->dViews, c->hViews.data(), sizeof(DevView) * kMaxViews * n_items, hipMemcpyHostToDevice, s));
	HIPCHK(c, hipMemcpyAsync(c->dItems, c->hItems.data(), sizeof(EstConst) * n_items, hipMemcpyHostToDevice, s));
	HIPCHK(c, hipMemsetAsync(c->evals, 0, 32, s));
	HIPCHK(c, hipMemsetAsync(c->sync, 0, 64 + sizeof(int32_t) * 2 * kMaxBatch, s));

	HIPCHK(c, hipEventRecord(c->ev[0], s));
	for (int i = 0; i < n_items; ++i) {
		const EstConst& k = c->hItems[i];
		const float* depthIn = items[i].d_depth;
		if (p->median_blur) { // SceneDensify.cpp:859
			launch_median3(items[i].d_depth, c->slots[i].tmpDepth, k.W, k.H, s);
			depthIn = c->slots[i].tmpDepth;
		}
		launch_score_pass(k, depthIn, items[i].d_normal, c->evals, s);
	}
	HIPCHK(c, hipEventRecord(c->ev[1], s));
	SweepSync sy;
	sy.ticket = c->sync + 16; sy.rowsDone = c->sync + 16 + kMaxBatch; sy.error = c->sync + 1; sy.evals = c->evals;
	// A batch whose rows fill the chip four times over (12 images of 1080p or more) runs all its sweeps in ONE launch (round 4): the images go
	// through their sweeps independently of each other, without a chip-wide drain and refill between two sweeps (+5 % at 12, +7 % at 16 images,
	// +2.6 ... 3.5 % at 32; profiles/r04_launch_modes.txt).  Fewer images are bound by the latency of their row wavefronts, not by the
	// chip, and gain nothing from it (measured: 4 -> 8 % slower), so they keep one launch per sweep.  HCMVS_SWEEP_LAUNCHES=one | per-sweep
	// overrides.  The `restore` variant's extra hypothesis belongs to the last sweep of the last outer iteration: that sweep then
	// gets a launch of its own, with the kernel instance that knows the hint.
	{
		const int nSweeps = p->n_estimation_iters;
		bool hintLast = false;
		for (int i = 0; i < n_items; ++i) hintLast = hintLast || (c->hItems[i].hintDepth && c->hItems[i].hintIter == nSweeps - 1);
		// Waves per row, by how the batch's rows compare with the workers the chip holds at once (12 per CU; measured on 1080p images,
		// profiles/r04_launch_modes.txt): a batch of few rows is bound by the latency of one pixel along the (W + H) critical path of its row
		// wavefronts, and the scoring of a pixel's hypotheses divides among the waves of a row -- three waves for one image (1066 rows), two
		// for two or three (a lone 3840x2160 image has 2146 rows: two), one from there on, where the chip's throughput counts.  More waves
		// than workers are fine since the rows are handed out in stretches (below): a row no longer needs a worker of its own from its
		// first to its last column.
		const int slots = (c->nCU > 0 ? c->nCU : 256) * (p->adapthalfwin > kHalfWindow ? 8 : 12); // (the big-patch worker: two waves per SIMD)
		const int nw = c->wavesPerRow ? c->wavesPerRow : (10 * totalRows <= 4 * slots ? 3 : (10 * totalRows <= 11 * slots ? 2 : 1));
		// the launcher picks the kernel variant by view count: an item whose count leaves two or more view groups idle wants the
		// pair-packing variant, which is correct for the other items of its layout class too
		int vSel = items[0].n_src;
		for (int i = 0; i < n_items; ++i) if (items[i].n_src % 8 != 0 && items[i].n_src % 8 != 7) vSel = items[i].n_src;
		int first = 0, nLaunches = 0;
		c->lastSweepLaunches = 0;
		while (first < nSweeps) {
			const bool perSweep = c->sweepPerLaunch == 1 || (c->sweepPerLaunch == 0 && totalRows < 4 * slots);
			int count = perSweep ? 1 : nSweeps - first;
			bool hint = false;
			if (hintLast) { if (first == nSweeps - 1) hint = true; else if (first + count == nSweeps) --count; } // the hint sweep runs alone
			HIPCHK(c, hipMemsetAsync(c->sync + 16, 0, sizeof(int32_t) * 2 * kMaxBatch, s)); // tickets + rowsDone; the error word stays sticky
			for (int i = 0; i < n_items; ++i)
				HIPCHK(c, hipMemsetAsync(c->slots[i].progress, 0, (size_t)(c->hItems[i].H - 2 * c->hItems[i].border) * kProgressStride * sizeof(int32_t), s));
			// A launch of ONE sweep with more waves than the chip holds workers (and less than four times as many rows: from there on all
			// sweeps run in one launch) hands out stretches of 256 columns instead of whole rows: a worker's slot comes free after 256 pixels.
			// With whole rows the rows beyond the resident set begin only when row 0 has reached its end -- 8 images: 107.2 -> 100.3 ms per
			// sweep; it is also what lets one to three images have two or three waves per row (1: 35.6 -> 33.7 ms, 2: 46.1 -> 42.0,
			// 3: 59.9 -> 55.4); 4 and 5 images: 0 ... +2 % (profiles/r04_launch_modes.txt).  HCMVS_SWEEP_SEGMENT=N forces stretches of N
			// columns, 0 whole rows.
			int segLen = 0, tickets = totalRows;
			if (c->sweepSegment > 0 || (c->sweepSegment < 0 && count == 1 && nw * totalRows > slots && totalRows < 4 * slots)) {
				segLen = c->sweepSegment > 0 ? c->sweepSegment : 256;
				if (segLen < 32) segLen = 32; // (the ring of a row's latest results is re-read from memory at the start of a stretch)
				tickets = 0;
				for (int i = 0; i < n_items; ++i) {
					const int rows = c->hItems[i].H - 2 * c->hItems[i].border, cols = c->hItems[i].W - 2 * c->hItems[i].border;
					tickets += rows * ((cols + segLen - 1) / segLen);
				}
			}
			launch_sweep(c->dItems, n_items, maxRows, tickets, vSel, p->adapthalfwin > kHalfWindow, hint, sy, first, count, c->sweepLag, nw, c->xcdAffinity, segLen, s);
			first += count; ++nLaunches;
		}
		c->lastSweepLaunches = nLaunches;
	}
	HIPCHK(c, hipEventRecord(c->ev[2], s));
	for (int i = 0; i < n_items; ++i)
		launch_end_pass(c->hItems[i], p->it_external == p->n_external_iters - 1 ? 1 : 0, items[i].d_depth, items[i].d_normal, items[i].d_conf, s);
	HIPCHK(c, hipEventRecord(c->ev[3], s));
	HIPCHK(c, hipGetLastError());
	c->lastSweeps = p->n_estimation_iters;
	c->haveStats = true;
	c->errPending = true;
	return HCMVS_OK;
}

int hcmvs_estimate_device(hcmvs_ctx* c, uint32_t ref_id, const uint32_t* src_ids, int32_t n_src, const hcmvs_params* p,
                          float d_min, float d_max, float* d_depth, float* d_normal, float* d_conf) {
	hcmvs_batch_item it;
	memset(&it, 0, sizeof it);
	it.ref_id = ref_id; it.src_ids = src_ids; it.n_src = n_src; it.d_min = d_min; it.d_max = d_max;
	it.d_depth = d_depth; it.d_normal = d_normal; it.d_conf = d_conf; it.seed_offset = 0;
	return hcmvs_estimate_batch_device(c, &it, 1, p);
}

int hcmvs_get_stats(hcmvs_ctx* c, hcmvs_stats* out) {
	if (!c || !out) return HCMVS_ERR_INVALID;
	memset(out, 0, sizeof *out);
	if (!c->haveStats) return fail(c, HCMVS_ERR_INVALID, "get_stats: no estimate has run");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	unsigned long long ev[4] = {0, 0, 0, 0};
	HIPCHK(c, hipMemcpy(ev, c->evals, 32, hipMemcpyDeviceToHost));
	out->evals = ev[0];
	out->evals_issued = ev[1];
	out->tap_evals = ev[2];
	HIPCHK(c, hipEventElapsedTime(&out->ms_score, c->ev[0], c->ev[1]));
	HIPCHK(c, hipEventElapsedTime(&out->ms_sweeps, c->ev[1], c->ev[2]));
	HIPCHK(c, hipEventElapsedTime(&out->ms_end, c->ev[2], c->ev[3]));
	HIPCHK(c, hipEventElapsedTime(&out->ms_total, c->ev[0], c->ev[3]));
	out->n_sweeps = c->lastSweeps;
	out->n_sweep_launches = c->lastSweepLaunches;
	out->ms_sweep_avg = c->lastSweeps > 0 ? out->ms_sweeps / (float)c->lastSweeps : 0.f;
	c->errPending = true;
	return check_sweep_error(c);
}

int hcmvs_estimate(hcmvs_ctx* c, uint32_t ref_id, const uint32_t* src_ids, int32_t n_src, const hcmvs_params* p, float d_min,
                   float d_max, float* depth, float* normal, float* conf) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!depth || !normal || !conf) return fail(c, HCMVS_ERR_INVALID, "estimate: null map buffer");
	auto rit = c->views.find(ref_id);
	if (rit == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "estimate: unknown reference view %u", ref_id);
	HIPCHK(c, hipSetDevice(c->device));
	const size_t n = (size_t)rit->second.w * rit->second.h;
	if (n > c->capStage) {
		HIPCHK(c, hipStreamSynchronize(c->stream));
		for (void* q : {(void*)c->sDepth, (void*)c->sNormal, (void*)c->sConf}) if (q) (void)hipFree(q);
		c->sDepth = c->sNormal = c->sConf = nullptr; c->capStage = 0;
		HIPCHK(c, hipMalloc(&c->sDepth, n * 4));
		HIPCHK(c, hipMalloc(&c->sNormal, n * 12));
		HIPCHK(c, hipMalloc(&c->sConf, n * 4));
		c->capStage = n;
	}
	hipStream_t s = c->stream;
	HIPCHK(c, hipMemcpyAsync(c->sDepth, depth, n * 4, hipMemcpyHostToDevice, s));
	HIPCHK(c, hipMemcpyAsync(c->sNormal, normal, n * 12, hipMemcpyHostToDevice, s));
	HIPCHK(c, hipMemcpyAsync(c->sConf, conf, n * 4, hipMemcpyHostToDevice, s));
	int rc = hcmvs_estimate_device(c, ref_id, src_ids, n_src, p, d_min, d_max, c->sDepth, c->sNormal, c->sConf);
	if (rc) return rc;
	HIPCHK(c, hipMemcpyAsync(depth, c->sDepth, n * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipMemcpyAsync(normal, c->sNormal, n * 12, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipMemcpyAsync(conf, c->sConf, n * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipStreamSynchronize(s));
	hcmvs_stats st;
	return hcmvs_get_stats(c, &st); // surfaces a sweep timeout as an error
}

int hcmvs_splat_points(int32_t W, int32_t H, const double K[9], const double R[9], const double C[3], const float* pts, int32_t n, float* depth,
                       float* normal, float* d_min, float* d_max) {
	if (!K || !R || !C || !pts || !depth || !normal || !d_min || !d_max || n < 1 || W < 1 || H < 1) return HCMVS_ERR_INVALID;
	memset(depth, 0, sizeof(float) * (size_t)W * H);
	float dmin = 3.402823466e+38f, dmax = 0.f;
	for (int i = 0; i < n; ++i) { // SceneDensify.cpp:789-806
		const double X[3] = {pts[3 * i] - C[0], pts[3 * i + 1] - C[1], pts[3 * i + 2] - C[2]};
		double cam[3];
		for (int r = 0; r < 3; ++r) cam[r] = R[r * 3] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2];
		const int x = (int)std::floor(K[2] + K[0] * (cam[0] / cam[2]) + .5);
		const int y = (int)std::floor(K[5] + K[4] * (cam[1] / cam[2]) + .5);
		const float d = (float)cam[2];
		const int sx = x - 2 > 0 ? x - 2 : 0, sy = y - 2 > 0 ? y - 2 : 0;
		const int ex = x + 2 < W - 1 ? x + 2 : W - 1, ey = y + 2 < H - 1 ? y + 2 : H - 1;
		for (int yy = sy; yy <= ey; ++yy)
			for (int xx = sx; xx <= ex; ++xx) {
				depth[(size_t)yy * W + xx] = d;
				float* nn = normal + 3 * ((size_t)yy * W + xx);
				nn[0] = nn[1] = nn[2] = 0.f;
			}
		if (dmin > d) dmin = d;
		if (dmax < d) dmax = d;
	}
	*d_min = dmin * 0.9f;
	*d_max = dmax * 1.1f;
	return HCMVS_OK;
}
int hcmvs_splat_init(hcmvs_ctx* c, uint32_t id, const float* pts, int32_t n, float* depth, float* normal, float* d_min,
                     float* d_max) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!pts || !depth || !normal || !d_min || !d_max || n < 1) return fail(c, HCMVS_ERR_INVALID, "splat_init: bad arguments");
	auto it = c->views.find(id);
	if (it == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "splat_init: unknown view %u", id);
	const View& v = it->second;
	return hcmvs_splat_points(v.w, v.h, v.K, v.R, v.C, pts, n, depth, normal, d_min, d_max);
}

int hcmvs_triangulate_points(int32_t W, int32_t H, const double K[9], const double R[9], const double C[3], const float* pts,
                             int32_t n, float avg_depth, int32_t add_corners, float* depth, float* normal, float* d_min, float* d_max) {
	if (!K || !R || !C || !pts || !depth || !normal || !d_min || !d_max || n < 1 || W < 1 || H < 1) return HCMVS_ERR_INVALID;
	float lo = 0.f, hi = 0.f;
	if (!hcmvs::triangulate_init(W, H, K, R, C, pts, n, avg_depth, add_corners != 0, depth, normal, &lo, &hi)) return HCMVS_ERR_INVALID;
	*d_min = lo * 0.9f; // SceneDensify.cpp:524-525
	*d_max = hi * 1.1f;
	return HCMVS_OK;
}
int hcmvs_triangulate_init(hcmvs_ctx* c, uint32_t id, const float* pts, int32_t n, float avg_depth, int32_t add_corners, float* depth,
                           float* normal, float* d_min, float* d_max) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!pts || !depth || !normal || !d_min || !d_max || n < 1) return fail(c, HCMVS_ERR_INVALID, "triangulate_init: bad arguments");
	auto it = c->views.find(id);
	if (it == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "triangulate_init: unknown view %u", id);
	const View& v = it->second;
	if (hcmvs_triangulate_points(v.w, v.h, v.K, v.R, v.C, pts, n, avg_depth, add_corners, depth, normal, d_min, d_max) != HCMVS_OK)
		return fail(c, HCMVS_ERR_INVALID, "triangulate_init: no sparse point in front of view %u", id);
	return HCMVS_OK;
}

// cv::resize INTER_AREA, enlarging (OpenCV 4.2 imgproc/src/resize.cpp: the bilinear kernel driven by "area mode" tables):
// per destination column / row the left / upper source index and the weight of its right / lower partner
int hcmvs_resize_area_up(const float* src, int32_t sw, int32_t sh, int32_t ch, float* dst, int32_t dw, int32_t dh) {
	if (!src || !dst || sw < 1 || sh < 1 || ch < 1 || dw < sw || dh < sh) return HCMVS_ERR_INVALID;
	struct Tab { std::vector<int> ofs; std::vector<float> w; };
	auto table = [](int ssize, int dsize, bool columns) {
		Tab t; t.ofs.resize((size_t)dsize); t.w.resize((size_t)dsize);
		const double scale = (double)ssize / dsize, inv = (double)dsize / ssize;
		for (int d = 0; d < dsize; ++d) {
			int s = (int)std::floor(d * scale);
			float f = (float)((d + 1) - (s + 1) * inv);
			f = f <= 0 ? 0.f : f - std::floor(f);
			if (columns && s >= ssize - 1) { f = 0.f; s = ssize - 1; } // no right partner: the sample alone (HResizeLinear past xmax)
			t.ofs[(size_t)d] = s; t.w[(size_t)d] = f;
		}
		return t;
	};
	const Tab tx = table(sw, dw, true), ty = table(sh, dh, false);
	std::vector<float> row0((size_t)dw * ch), row1((size_t)dw * ch);
	auto hresize = [&](int sy, std::vector<float>& out) {
		const float* S = src + (size_t)sy * sw * ch;
		for (int x = 0; x < dw; ++x) {
			const int sx = tx.ofs[(size_t)x];
			const float a1 = tx.w[(size_t)x], a0 = 1.f - a1;
			for (int c = 0; c < ch; ++c)
				out[(size_t)x * ch + c] = sx + 1 >= sw ? S[(size_t)sx * ch + c] : S[(size_t)sx * ch + c] * a0 + S[(size_t)(sx + 1) * ch + c] * a1;
		}
	};
	int have0 = -1, have1 = -1;
	for (int y = 0; y < dh; ++y) {
		const int r0 = std::min(std::max(ty.ofs[(size_t)y], 0), sh - 1), r1 = std::min(std::max(ty.ofs[(size_t)y] + 1, 0), sh - 1);
		if (have0 != r0) { if (have1 == r0) { row0.swap(row1); std::swap(have0, have1); } else { hresize(r0, row0); have0 = r0; } }
		if (have1 != r1) { hresize(r1, row1); have1 = r1; }
		const float b1 = ty.w[(size_t)y], b0 = 1.f - b1;
		float* D = dst + (size_t)y * dw * ch;
		for (size_t k = 0; k < (size_t)dw * ch; ++k) D[k] = row0[k] * b0 + row1[k] * b1;
	}
	return HCMVS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// filter / fuse

static int set_maps(hcmvs_ctx* c, uint32_t id, const float* depth, const float* normal, const float* conf, float dmin, float dmax, bool copy) {
	if (!c) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end()) return fail(c, HCMVS_ERR_INVALID, "set_depthmap: unknown view %u", id);
	if (!depth || !conf) return fail(c, HCMVS_ERR_INVALID, "set_depthmap: null map");
	View& v = it->second;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	free_maps(v);
	const size_t n = (size_t)v.w * v.h;
	if (copy) {
		HIPCHK(c, hipMalloc(&v.mDepth, n * 4));
		HIPCHK(c, hipMalloc(&v.mConf, n * 4));
		int rc = upload_staged(c, v.mDepth, depth, n * 4);
		if (!rc) rc = upload_staged(c, v.mConf, conf, n * 4);
		if (rc) return rc;
		if (normal) {
			HIPCHK(c, hipMalloc(&v.mNormal, n * 12));
			rc = upload_staged(c, v.mNormal, normal, n * 12);
			if (rc) return rc;
		}
		v.mapsOwned = true;
	} else {
		v.mDepth = const_cast<float*>(depth); v.mNormal = const_cast<float*>(normal); v.mConf = const_cast<float*>(conf);
	}
	v.dMin = dmin; v.dMax = dmax;
	return HCMVS_OK;
}
int hcmvs_set_depthmap(hcmvs_ctx* c, uint32_t id, const float* depth, const float* normal, const float* conf, float d_min, float d_max) {
	return set_maps(c, id, depth, normal, conf, d_min, d_max, true);
}
int hcmvs_set_depthmap_device(hcmvs_ctx* c, uint32_t id, float* d_depth, const float* d_normal, const float* d_conf, float d_min, float d_max) {
	return set_maps(c, id, d_depth, d_normal, d_conf, d_min, d_max, false);
}
int hcmvs_get_depthmap(hcmvs_ctx* c, uint32_t id, float* depth, float* normal, float* conf) {
	if (!c) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end() || !it->second.mDepth) return fail(c, HCMVS_ERR_INVALID, "get_depthmap: view %u has no maps", id);
	const View& v = it->second;
	const size_t n = (size_t)v.w * v.h;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (depth) HIPCHK(c, hipMemcpy(depth, v.mDepth, n * 4, hipMemcpyDeviceToHost));
	if (normal && v.mNormal) HIPCHK(c, hipMemcpy(normal, v.mNormal, n * 12, hipMemcpyDeviceToHost));
	if (conf) HIPCHK(c, hipMemcpy(conf, v.mConf, n * 4, hipMemcpyDeviceToHost));
	return HCMVS_OK;
}
int hcmvs_set_neighbors(hcmvs_ctx* c, uint32_t id, const uint32_t* ids, int32_t n) {
	if (!c) return HCMVS_ERR_INVALID;
	auto it = c->views.find(id);
	if (it == c->views.end() || n < 0 || n > 64 || (n > 0 && !ids)) return fail(c, HCMVS_ERR_INVALID, "set_neighbors: bad arguments for view %u", id);
	View& v = it->second;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	if (v.dNeighbors) { (void)hipFree(v.dNeighbors); v.dNeighbors = nullptr; }
	v.neighbors.assign(ids, ids + n);
	if (n > 0) {
		HIPCHK(c, hipMalloc(&v.dNeighbors, (size_t)n * 4));
		HIPCHK(c, hipMemcpy(v.dNeighbors, ids, (size_t)n * 4, hipMemcpyHostToDevice));
	}
	return HCMVS_OK;
}

static void fill_devmap(uint32_t id, const View& v, DevMap& m) {
	memset(&m, 0, sizeof m);
	m.id = id; m.w = v.w; m.h = v.h; m.nNeighbors = (int)v.neighbors.size();
	memcpy(m.K, v.K, sizeof m.K); memcpy(m.R, v.R, sizeof m.R); memcpy(m.C, v.C, sizeof m.C);
	double t[3];
	for (int i = 0; i < 3; ++i) t[i] = -(v.R[i * 3] * v.C[0] + v.R[i * 3 + 1] * v.C[1] + v.R[i * 3 + 2] * v.C[2]);
	for (int i = 0; i < 3; ++i) { // P = K [R | -R C], Camera.h:276-282
		for (int j = 0; j < 3; ++j) m.P[i * 4 + j] = v.K[i * 3] * v.R[j] + v.K[i * 3 + 1] * v.R[3 + j] + v.K[i * 3 + 2] * v.R[6 + j];
		m.P[i * 4 + 3] = v.K[i * 3] * t[0] + v.K[i * 3 + 1] * t[1] + v.K[i * 3 + 2] * t[2];
	}
	m.depth = v.mDepth; m.normal = v.mNormal; m.conf = v.mConf; m.bgr = v.bgr;
	m.neighbors = v.dNeighbors; m.dMin = v.dMin; m.dMax = v.dMax;
}
// device table indexed by image id (views without maps have depth == null)
static int build_map_table(hcmvs_ctx* c, std::vector<DevMap>& host) {
	uint32_t maxId = 0;
	for (auto& kv : c->views) if (kv.first > maxId) maxId = kv.first;
	host.assign((size_t)maxId + 1, DevMap());
	for (auto& m : host) memset(&m, 0, sizeof m);
	for (auto& kv : c->views) fill_devmap(kv.first, kv.second, host[kv.first]);
	if (host.size() > c->capMaps) {
		if (c->dMaps) (void)hipFree(c->dMaps);
		c->dMaps = nullptr; c->capMaps = 0;
		HIPCHK(c, hipMalloc(&c->dMaps, host.size() * sizeof(DevMap)));
		c->capMaps = host.size();
	}
	HIPCHK(c, hipMemcpy(c->dMaps, host.data(), host.size() * sizeof(DevMap), hipMemcpyHostToDevice));
	if (!c->counters) HIPCHK(c, hipMalloc(&c->counters, 8 * sizeof(unsigned long long)));
	return HCMVS_OK;
}
// hipMalloc that gives the post-filter chain's state back to the device when memory is short (the chain allocates it again, or falls
// back to fusions from scratch, the next time it runs)
static hipError_t malloc_or_release_chain(hcmvs_ctx* c, void** p, size_t bytes) {
	hipError_t e = hipMalloc(p, bytes);
	if (e != hipSuccess && c->pfState) {
		(void)hipGetLastError();
		(void)hipStreamSynchronize(c->stream);
		(void)hipFree(c->pfState);
		c->pfState = nullptr; c->capPf = 0;
		e = hipMalloc(p, bytes);
	}
	return e;
}
static int ensure_scratch(hcmvs_ctx* c, size_t bytes) {
	if (bytes <= c->capFuseScratch) return HCMVS_OK;
	if (c->fuseScratch) (void)hipFree(c->fuseScratch);
	c->fuseScratch = nullptr; c->capFuseScratch = 0;
	HIPCHK(c, malloc_or_release_chain(c, &c->fuseScratch, bytes));
	c->capFuseScratch = bytes;
	return HCMVS_OK;
}

int hcmvs_filter(hcmvs_ctx* c, uint32_t ref_id, const uint32_t* nbr, int32_t N, int32_t adjust, int32_t n_min_views,
                 int32_t n_min_views_adjust, float depth_diff_threshold, float* out_depth, float* out_conf, uint64_t* n_processed,
                 uint64_t* n_discarded) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!nbr || !out_depth || !out_conf || N < 1 || N > 64) return fail(c, HCMVS_ERR_INVALID, "filter: bad arguments");
	if (N < n_min_views || N < n_min_views_adjust) return fail(c, HCMVS_ERR_INVALID, "filter: depth map %u can not be filtered (%d neighbours)", ref_id, N); // SceneDensify.cpp:3016-3019
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	std::vector<DevMap> host;
	int rc = build_map_table(c, host);
	if (rc) return rc;
	if (ref_id >= host.size() || !host[ref_id].depth) return fail(c, HCMVS_ERR_INVALID, "filter: view %u has no maps", ref_id);
	std::vector<DevMap> nbs((size_t)N);
	for (int n = 0; n < N; ++n) {
		if (nbr[n] >= host.size() || !host[nbr[n]].depth) return fail(c, HCMVS_ERR_INVALID, "filter: neighbour %u has no maps", nbr[n]);
		nbs[n] = host[nbr[n]];
	}
	const DevMap& ref = host[ref_id];
	const size_t area = (size_t)ref.w * ref.h;
	// scratch: keys [N*area u64] | neighbour table | new depth | new conf
	const size_t offNb = area * N * 8, offD = offNb + ((sizeof(DevMap) * N + 255) & ~(size_t)255), offC = offD + area * 4;
	rc = ensure_scratch(c, offC + area * 4);
	if (rc) return rc;
	char* base = (char*)c->fuseScratch;
	unsigned long long* keys = (unsigned long long*)base;
	DevMap* dNbs = (DevMap*)(base + offNb);
	float* dD = (float*)(base + offD); float* dC = (float*)(base + offC);
	hipStream_t s = c->stream;
	HIPCHK(c, hipMemcpyAsync(dNbs, nbs.data(), sizeof(DevMap) * N, hipMemcpyHostToDevice, s));
	HIPCHK(c, hipMemsetAsync(c->counters, 0, 64, s));
	launch_fill_u64(keys, ~0ull, area * N, s);
	for (int n = 0; n < N; ++n) launch_filter_splat(ref, nbs[n], keys + area * n, s);
	launch_filter_vote(ref, dNbs, N, keys, adjust, n_min_views, n_min_views_adjust, depth_diff_threshold, dD, dC, c->counters, s);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipMemcpyAsync(out_depth, dD, area * 4, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipMemcpyAsync(out_conf, dC, area * 4, hipMemcpyDeviceToHost, s));
	unsigned long long cnt[2] = {0, 0};
	HIPCHK(c, hipMemcpyAsync(cnt, c->counters, 16, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipStreamSynchronize(s));
	if (n_processed) *n_processed = cnt[0];
	if (n_discarded) *n_discarded = cnt[1];
	return HCMVS_OK;
}

int hcmvs_set_fuse_order(hcmvs_ctx* c, int32_t mode) {
	if (!c) return HCMVS_ERR_INVALID;
	if (mode != 0 && mode != 1) return fail(c, HCMVS_ERR_INVALID, "set_fuse_order: mode %d not in {0, 1}", mode);
	c->fuseOrder = mode;
	return HCMVS_OK;
}

int hcmvs_fuse_cloud(hcmvs_ctx* c, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse, float depth_diff_threshold,
                     float normal_diff_deg, float depthweight, float normalweight, hcmvs_cloud* cloud) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!cloud) return fail(c, HCMVS_ERR_INVALID, "fuse: null cloud");
	const uint64_t capacity = cloud->capacity, viewCapacity = cloud->view_ids ? cloud->views_capacity : 0;
	float* xyz = cloud->xyz; float* normal = cloud->normal; uint8_t* bgr = cloud->bgr; uint32_t* n_views = cloud->n_views;
	uint64_t* n_points = &cloud->n_points; uint64_t* n_depths = &cloud->n_depths;
	cloud->n_points = cloud->n_depths = cloud->n_view_entries = 0;
	if (cloud->view_ids && !cloud->view_weights) return fail(c, HCMVS_ERR_INVALID, "fuse: view_ids without view_weights");
	if (!c) return HCMVS_ERR_INVALID;
	if (!order || n_order < 1 || !n_points) return fail(c, HCMVS_ERR_INVALID, "fuse: bad arguments");
	const auto tCall = std::chrono::steady_clock::now();
	const bool wantCloud = xyz != nullptr; // without xyz only the fusion's side effects (claims, invalidated depths) and counts are produced
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	std::vector<DevMap> host;
	int rc = build_map_table(c, host);
	if (rc) return rc;
	size_t maxArea = 0;
	for (int i = 0; i < n_order; ++i) {
		if (order[i] >= host.size() || !host[order[i]].depth) return fail(c, HCMVS_ERR_INVALID, "fuse: view %u has no maps", order[i]);
		const size_t a = (size_t)host[order[i]].w * host[order[i]].h;
		if (a > maxArea) maxArea = a;
		if (host[order[i]].nNeighbors > kFuseMaxViews - 1) return fail(c, HCMVS_ERR_INVALID, "fuse: view %u has too many neighbours", order[i]);
	}
	hipStream_t s = c->stream;
	launch_unclaim(c->dMaps, (int)host.size(), s); // no claim mark may be left over from a fusion that failed half way
	int maxNb = 1;
	size_t stride = maxArea; // pixels reserved per neighbour map in the per-target tables
	for (int i = 0; i < n_order; ++i) maxNb = std::max(maxNb, (int)host[order[i]].nNeighbors);
	for (const auto& m : host) if (m.depth) stride = std::max(stride, (size_t)m.w * m.h);
	const size_t tblElems = stride * (size_t)maxNb;
	if (stride >= ((size_t)1 << 29)) return fail(c, HCMVS_ERR_CAPACITY, "fuse: maps of %zu pixels exceed the target tables (2^29 pixels)", stride);
	if (tblElems > 0x7FFFFFFFull) return fail(c, HCMVS_ERR_CAPACITY, "fuse: %d neighbours of %zu pixels exceed the per-pass tables", maxNb, stride);
	const size_t scanBytes = (fuse_scan_temp_bytes((int)maxArea) + 255) & ~(size_t)255;
	size_t off = 0;
	auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
	// the device cloud (context scratch) ...
	const size_t oCX = carve(wantCloud ? capacity * 12 : 0), oCN = carve(normal ? capacity * 12 : 0), oCB = carve(bgr ? capacity * 3 : 0),
	             oCV = carve(n_views || viewCapacity ? capacity * 4 : 0), oCVI = carve(viewCapacity * 4), oCVW = carve(viewCapacity * 4);
	rc = ensure_scratch(c, std::max(off, (size_t)256));
	if (rc) return rc;
	char* cb = (char*)c->fuseScratch;
	float* cX = (float*)(cb + oCX); float* cN = normal ? (float*)(cb + oCN) : nullptr; uint8_t* cB = bgr ? (uint8_t*)(cb + oCB) : nullptr;
	uint32_t* cV = n_views || viewCapacity ? (uint32_t*)(cb + oCV) : nullptr;
	uint32_t* cVI = viewCapacity ? (uint32_t*)(cb + oCVI) : nullptr; float* cVW = viewCapacity ? (float*)(cb + oCVW) : nullptr;
	// ... and the per-pass tables (sized for the largest image)
	off = 0;
	const size_t oPending = carve(maxArea * 4), oSettle = carve(fuse_settle_bytes(maxArea)), oTgt = carve(maxArea * 4 * (size_t)maxNb),
	             oHead = carve(tblElems * 4), oNext = carve(maxArea * 4 * (size_t)maxNb), oCtl = carve(kCtlBytes), oCounters = carve(64), oStatus = carve(64), oTotals = carve(64),
	             oMerged = carve(maxArea * 4), oFlag = carve(maxArea), oFlag32 = carve(maxArea * 4), oPos = carve(maxArea * 4), oScan = carve(scanBytes),
	             oXyz = carve(maxArea * 12), oNrm = carve(maxArea * 12), oBgr = carve(maxArea * 3), oNv = carve(maxArea * 4),
	             oPV = carve(viewCapacity ? maxArea * 4 * (size_t)(maxNb + 1) : 0), oPW = carve(viewCapacity ? maxArea * 4 * (size_t)(maxNb + 1) : 0),
	             oVoff = carve(viewCapacity ? maxArea * 4 : 0);
	if (c->capPass < off) {
		if (c->passScratch) (void)hipFree(c->passScratch);
		c->passScratch = nullptr; c->capPass = 0;
		HIPCHK(c, malloc_or_release_chain(c, (void**)&c->passScratch, off));
		c->capPass = off;
	}
	const int vstride = maxNb + 1;
	const float normalError = cosf(normal_diff_deg * normalweight * (3.14159274101257324f / 180.f)); // SceneDensify.cpp:3310
	const float thDepth = depth_diff_threshold * depthweight;                                       // SceneDensify.cpp:3400
	const bool debug = getenv("HCMVS_FUSE_DEBUG") != nullptr; // per image: a synchronisation and a line on stderr

	// Every pass of the fusion is enqueued on the context's stream, no host synchronisation between the images (stream order is the
	// order of the sequential loop, SceneDensify.cpp:3302).  What the host would have to know in between stays on the device: the
	// running point / view-entry totals an image's compaction starts from (fuse_advance_kernel), and the word that says a cloud or
	// view-list capacity was exceeded.
	char* b = c->passScratch;
	uint32_t* pendingList = (uint32_t*)(b + oPending); uint32_t* ctl = (uint32_t*)(b + oCtl);
	unsigned long long* counters = (unsigned long long*)(b + oCounters);
	uint32_t* status = (uint32_t*)(b + oStatus);
	unsigned long long* totals = (unsigned long long*)(b + oTotals);
	int32_t* targets = (int32_t*)(b + oTgt);
	uint32_t *head = (uint32_t*)(b + oHead), *next = (uint32_t*)(b + oNext);
	uint8_t* flag = (uint8_t*)(b + oFlag);
	uint32_t* flag32 = (uint32_t*)(b + oFlag32); uint32_t* pos = (uint32_t*)(b + oPos); uint32_t* merged = (uint32_t*)(b + oMerged);
	float* pxyz = (float*)(b + oXyz); float* pnrm = (float*)(b + oNrm); uint8_t* pbgr = (uint8_t*)(b + oBgr); uint32_t* pnv = (uint32_t*)(b + oNv);
	uint32_t* pviews = viewCapacity ? (uint32_t*)(b + oPV) : nullptr; float* pweights = viewCapacity ? (float*)(b + oPW) : nullptr;
	uint32_t* voff = viewCapacity ? (uint32_t*)(b + oVoff) : nullptr;
	const FuseTables tb = fuse_tables(targets, head, next, stride);
	const auto tPasses = std::chrono::steady_clock::now();
	HIPCHK(c, hipMemsetAsync(status, 0, 64, s));
	HIPCHK(c, hipMemsetAsync(totals, 0, 64, s));
	for (int oi = 0; oi < n_order; ++oi) {
		const DevMap& A = host[order[oi]];
		const int n = A.w * A.h;
		const auto tPass = std::chrono::steady_clock::now();
		HIPCHK(c, hipMemsetAsync(counters, 0, 64, s));
		HIPCHK(c, hipMemsetAsync(ctl, 0, kCtlBytes, s));
		HIPCHK(c, hipMemsetAsync(head, 0xFF, (size_t)A.nNeighbors * stride * 4, s)); // empty bidder lists
		launch_fuse_begin(A, c->dMaps, tb, pendingList, ctl, flag, counters, thDepth, normalError, s);
		launch_fuse_pass(A, c->dMaps, tb, pendingList, b + oSettle, ctl, pxyz, cN ? pnrm : nullptr, cB ? pbgr : nullptr, pnv, flag, pviews, pweights, vstride,
		                 merged, n_min_views_fuse, c->fuseOrder, counters, status, wantCloud, s);
		if (wantCloud)
			launch_fuse_compact(n, flag, flag32, pos, b + oScan, scanBytes, pxyz, pnrm, pbgr, pnv, 0, capacity, cX, cN, cB, cV, pviews, pweights, vstride,
			                    voff, 0, viewCapacity, cVI, cVW, totals, s);
		launch_fuse_advance(counters, totals, wantCloud ? capacity : 0, wantCloud ? viewCapacity : 0, status, s);
		if (debug) {
			unsigned long long cnt[5] = {0, 0, 0, 0, 0};
			uint32_t ctlWords[kCtlBytes / 4];
			HIPCHK(c, hipMemcpyAsync(cnt, counters, 40, hipMemcpyDeviceToHost, s));
			HIPCHK(c, hipMemcpyAsync(ctlWords, ctl, kCtlBytes, hipMemcpyDeviceToHost, s));
			HIPCHK(c, hipStreamSynchronize(s));
			fprintf(stderr, "fuse: image %u: %u pending pixels, %llu become points; settle iteration: %u steps, work lists", A.id, ctlWords[kCtlPending], cnt[3], ctlWords[kCtlSteps]);
			for (int k = 0; k <= kSettleSteps; ++k) fprintf(stderr, " %u", ctlWords[kCtlWork + k]);
			fprintf(stderr, "; %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tPass).count());
		}
	}
	launch_unclaim(c->dMaps, (int)host.size(), s); // the claim marks come off the depth maps
	HIPCHK(c, hipGetLastError());
	uint32_t st[4] = {0, 0, 0, 0};
	unsigned long long tot[3] = {0, 0, 0};
	HIPCHK(c, hipMemcpyAsync(st, status, 16, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipMemcpyAsync(tot, totals, 24, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipStreamSynchronize(s));
	if (st[0] != 0) return fail(c, HCMVS_ERR_TIMEOUT, "fuse: the settle iteration of a pass gave up; the registered depth maps are left partially fused");
	if (st[3] == 1) return fail(c, HCMVS_ERR_CAPACITY, "fuse: cloud capacity %llu exceeded", (unsigned long long)capacity);
	if (st[3] == 2) return fail(c, HCMVS_ERR_CAPACITY, "fuse: view-list capacity %llu exceeded", (unsigned long long)viewCapacity);
	const auto tCopy = std::chrono::steady_clock::now();
	const unsigned long long total = tot[0], viewTotal = viewCapacity && wantCloud ? tot[1] : 0;
	if (wantCloud) HIPCHK(c, hipMemcpyAsync(xyz, cX, total * 12, hipMemcpyDeviceToHost, s));
	if (normal) HIPCHK(c, hipMemcpyAsync(normal, cN, total * 12, hipMemcpyDeviceToHost, s));
	if (bgr) HIPCHK(c, hipMemcpyAsync(bgr, cB, total * 3, hipMemcpyDeviceToHost, s));
	if (n_views) HIPCHK(c, hipMemcpyAsync(n_views, cV, total * 4, hipMemcpyDeviceToHost, s));
	if (viewCapacity && wantCloud) {
		HIPCHK(c, hipMemcpyAsync(cloud->view_ids, cVI, viewTotal * 4, hipMemcpyDeviceToHost, s));
		HIPCHK(c, hipMemcpyAsync(cloud->view_weights, cVW, viewTotal * 4, hipMemcpyDeviceToHost, s));
	}
	HIPCHK(c, hipStreamSynchronize(s));
	if (debug) {
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "fuse: %d images, %llu points: setup %.1f ms, passes %.1f ms, copy of the cloud to the host %.1f ms\n", n_order, total,
		        std::chrono::duration<double, std::milli>(tPasses - tCall).count(), std::chrono::duration<double, std::milli>(tCopy - tPasses).count(),
		        std::chrono::duration<double, std::milli>(now - tCopy).count());
	}
	*n_points = total;
	if (n_depths) *n_depths = tot[2];
	cloud->n_view_entries = wantCloud ? viewTotal : tot[1]; // a counting call (no xyz) reports what the cloud's view lists would hold
	return HCMVS_OK;
}

int hcmvs_fuse(hcmvs_ctx* c, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse, float depth_diff_threshold,
               float normal_diff_deg, float depthweight, float normalweight, uint64_t capacity, float* xyz, float* normal,
               uint8_t* bgr, uint32_t* n_views, uint64_t* n_points, uint64_t* n_depths) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!n_points) return fail(c, HCMVS_ERR_INVALID, "fuse: bad arguments");
	hcmvs_cloud cl;
	memset(&cl, 0, sizeof cl);
	cl.capacity = capacity; cl.xyz = xyz; cl.normal = normal; cl.bgr = bgr; cl.n_views = n_views;
	const int rc = hcmvs_fuse_cloud(c, order, n_order, n_min_views_fuse, depth_diff_threshold, normal_diff_deg, depthweight, normalweight, &cl);
	*n_points = cl.n_points;
	if (n_depths) *n_depths = cl.n_depths;
	return rc;
}

// The post-filter chain with its fusions computed incrementally (pf_kernels.hip): the first fusion of the chain evaluates every pixel,
// each later one only what depends on the estimates that changed since (the pixels the previous image's gap interpolation filled, the
// estimates the previous fusion zeroed).  Same decisions as a fusion from scratch, fusion after fusion.  Returns -1 when the chain can
// not run this way (state does not fit the device, an image occurs twice, too many images) -- the caller then fuses from scratch.
static int postfilter_chain_incremental(hcmvs_ctx* c, const std::vector<DevMap>& host, const uint32_t* ids, int32_t n_ids, const uint32_t* order, int32_t n_order,
                                        int32_t n_min_views_fuse, float depth_diff_threshold, float normal_diff_deg, int32_t gap_size, uint64_t* n_filled) {
	if (n_ids < 2 || n_ids > 2000 || n_order > 60000) return -1;
	{ // an image is filled once per chain (a pair of estimates is linked into a bidder list at most twice)
		std::vector<uint32_t> u(ids, ids + n_ids);
		std::sort(u.begin(), u.end());
		if (std::adjacent_find(u.begin(), u.end()) != u.end()) return -1;
	}
	std::vector<int> orderIndex(host.size(), -1);
	for (int i = 0; i < n_order; ++i) { if (orderIndex[order[i]] >= 0) return -1; orderIndex[order[i]] = i; }
	// layout of the state
	size_t off = 0, maxArea = 0, maxIdArea = 0;
	auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
	struct Lay { size_t own, chg, val, tgt, head, next, acc, mm, fm, stamp, touch, stride; };
	std::vector<Lay> lay(host.size());
	for (size_t id = 0; id < host.size(); ++id) {
		const DevMap& m = host[id];
		if (!m.depth) continue;
		const size_t n = (size_t)m.w * m.h;
		if (n >= ((size_t)1 << 26) - 1) return -1;
		Lay& L = lay[id];
		L.own = carve(n * 2); L.chg = carve(n); L.val = carve(n);
		if (orderIndex[id] < 0) continue;
		size_t stride = 1;
		for (int q = 0; q < m.nNeighbors; ++q) {
			const uint32_t nb = c->views.find((uint32_t)id)->second.neighbors[(size_t)q];
			if (nb < host.size() && host[nb].depth) stride = std::max(stride, (size_t)host[nb].w * host[nb].h);
		}
		const size_t nNb = (size_t)std::max(m.nNeighbors, 1);
		if (2 * nNb * n >= 0xFFFFFFFFull || nNb * stride >= 0xFFFFFFFFull) return -1;
		L.stride = stride;
		L.tgt = carve(n * nNb * 4); L.head = carve(nNb * stride * 4); L.next = carve(2 * nNb * n * 4);
		L.acc = carve(n); L.mm = carve(n * 4); L.fm = carve(n * 4); L.stamp = carve(n * 4); L.touch = carve(n * 4);
		maxArea = std::max(maxArea, n);
	}
	for (int k = 0; k < n_ids; ++k) { const View& v = c->views.find(ids[k])->second; maxIdArea = std::max(maxIdArea, (size_t)v.w * v.h); }
	const size_t oAny = carve(host.size() * 4);
	const size_t oTable = carve(host.size() * sizeof(PfImage)), oScratch = carve(pf_pass_scratch_bytes(maxArea)), oCtl = carve(kCtlBytes), oCounters = carve(64),
	             oStatus = carve(64), oDF = carve(maxIdArea * 4), oDF2 = carve(maxIdArea * 4), oNF = carve(maxIdArea * 12);
	if (c->capPf < off) {
		size_t freeB = 0, totalB = 0;
		if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) return -1;
		if (off > freeB + c->capPf || off > (freeB + c->capPf) / 10 * 9) { (void)hipGetLastError(); return -1; } // does not fit (with a margin): fuse from scratch
		if (c->pfState) (void)hipFree(c->pfState);
		c->pfState = nullptr; c->capPf = 0;
		if (hipMalloc(&c->pfState, off) != hipSuccess) { (void)hipGetLastError(); return -1; }
		c->capPf = off;
	}
	if (getenv("HCMVS_FUSE_DEBUG")) fprintf(stderr, "postfilter: chain of %d images, fusions computed incrementally (%.1f MB of state kept between them)\n", n_ids, off / 1048576.0);
	char* b = c->pfState;
	hipStream_t s = c->stream;
	std::vector<PfImage> pf(host.size());
	memset(pf.data(), 0, pf.size() * sizeof(PfImage));
	for (size_t id = 0; id < host.size(); ++id) {
		const DevMap& m = host[id];
		if (!m.depth) continue;
		const size_t n = (size_t)m.w * m.h;
		const Lay& L = lay[id];
		PfImage& P = pf[id];
		P.own = (uint16_t*)(b + L.own); P.chgNow = (uint8_t*)(b + L.chg); P.valNext = (uint8_t*)(b + L.val); P.anyChg = (uint32_t*)(b + oAny) + id;
		HIPCHK(c, hipMemsetAsync(P.own, 0xFF, n * 2, s));
		HIPCHK(c, hipMemsetAsync(P.chgNow, 0, n, s));
		HIPCHK(c, hipMemsetAsync(P.valNext, 0, n, s));
		if (orderIndex[id] < 0) continue;
		const size_t nNb = (size_t)std::max(m.nNeighbors, 1);
		P.tgt = (uint32_t*)(b + L.tgt); P.head = (uint32_t*)(b + L.head); P.next = (uint32_t*)(b + L.next); P.acc = (uint8_t*)(b + L.acc);
		P.mm = (uint32_t*)(b + L.mm); P.fm = (uint32_t*)(b + L.fm); P.stamp = (uint32_t*)(b + L.stamp); P.touch = (uint32_t*)(b + L.touch); P.stride = L.stride;
		launch_fill_u32(P.tgt, 0x03FFFFFFu, n * nNb, s); // "projects nowhere", never linked
		HIPCHK(c, hipMemsetAsync(P.head, 0xFF, nNb * L.stride * 4, s));
		HIPCHK(c, hipMemsetAsync(P.acc, 0, n, s));
		HIPCHK(c, hipMemsetAsync(P.mm, 0, n * 4, s)); HIPCHK(c, hipMemsetAsync(P.fm, 0, n * 4, s));
		HIPCHK(c, hipMemsetAsync(P.stamp, 0, n * 4, s)); HIPCHK(c, hipMemsetAsync(P.touch, 0, n * 4, s));
	}
	HIPCHK(c, hipMemsetAsync(b + oAny, 0, host.size() * 4, s));
	PfImage* dPf = (PfImage*)(b + oTable);
	HIPCHK(c, hipMemcpyAsync(dPf, pf.data(), pf.size() * sizeof(PfImage), hipMemcpyHostToDevice, s));
	uint32_t* ctl = (uint32_t*)(b + oCtl); unsigned long long* counters = (unsigned long long*)(b + oCounters); uint32_t* status = (uint32_t*)(b + oStatus);
	float* dF = (float*)(b + oDF); float* dF2 = (float*)(b + oDF2); float* nF = (float*)(b + oNF);
	HIPCHK(c, hipMemsetAsync(status, 0, 64, s));
	HIPCHK(c, hipMemsetAsync(counters, 0, 64, s));
	const float normalError = cosf(normal_diff_deg * (3.14159274101257324f / 180.f)); // plain thresholds (SceneDensify.cpp:2083, 2177)
	for (int k = 0; k < n_ids; ++k) {
		if (k > 0) launch_pf_roll(c->dMaps, dPf, (int)host.size(), (uint32_t*)(b + oAny), s);
		for (int oi = 0; oi < n_order; ++oi)
			launch_pf_pass(host[order[oi]], oi, c->dMaps, dPf, pf[order[oi]], k == 0, depth_diff_threshold, normalError, n_min_views_fuse, (uint32_t)k, b + oScratch, ctl, status, s);
		View& v = c->views.find(ids[k])->second;
		launch_pf_filter(v.w, v.h, v.mDepth, v.mNormal, v.mConf, pf[ids[k]], v.gra, dF, dF2, nF, gap_size, depth_diff_threshold * 2.5f, counters + 5, s);
	}
	HIPCHK(c, hipGetLastError());
	uint32_t st[4] = {0, 0, 0, 0};
	unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};
	HIPCHK(c, hipMemcpyAsync(st, status, 16, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipMemcpyAsync(cnt, counters, 48, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipStreamSynchronize(s));
	if (st[0] != 0) return fail(c, HCMVS_ERR_TIMEOUT, "postfilter: the settle iteration of a fusion pass gave up; the registered depth maps are left partially fused");
	if (n_filled) *n_filled = cnt[5];
	return HCMVS_OK;
}

// The post-filters of one outer iteration, image after image.  Every image costs a complete fusion over all maps (that IS the
// fork's RemoveSmallSegments, SceneDensify.cpp:2048-2275), so the fusion here is the one thing that must be cheap: it produces no
// cloud, and all its passes are enqueued on the context's stream WITHOUT host synchronisation -- stream order is the image order
// of the sequential algorithm.  One synchronisation at the end of the sequence.
int hcmvs_postfilter_sequence(hcmvs_ctx* c, const uint32_t* ids, int32_t n_ids, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse,
                              float depth_diff_threshold, float normal_diff_deg, int32_t gap_size, uint64_t* n_filled) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!ids || n_ids < 1 || !order || n_order < 1 || gap_size < 0) return fail(c, HCMVS_ERR_INVALID, "postfilter: bad arguments");
	for (int k = 0; k < n_ids; ++k) {
		auto it = c->views.find(ids[k]);
		if (it == c->views.end() || !it->second.mDepth || !it->second.mNormal) return fail(c, HCMVS_ERR_INVALID, "postfilter: view %u has no registered depth + normal maps", ids[k]);
	}
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	std::vector<DevMap> host;
	int rc = build_map_table(c, host);
	if (rc) return rc;
	size_t maxArea = 0, stride = 0, maxIdArea = 0;
	int maxNb = 1;
	for (int i = 0; i < n_order; ++i) {
		if (order[i] >= host.size() || !host[order[i]].depth) return fail(c, HCMVS_ERR_INVALID, "postfilter: view %u has no maps", order[i]);
		maxArea = std::max(maxArea, (size_t)host[order[i]].w * host[order[i]].h);
		if (host[order[i]].nNeighbors > kFuseMaxViews - 1) return fail(c, HCMVS_ERR_INVALID, "postfilter: view %u has too many neighbours", order[i]);
		maxNb = std::max(maxNb, (int)host[order[i]].nNeighbors);
	}
	stride = maxArea;
	for (const auto& m : host) if (m.depth) stride = std::max(stride, (size_t)m.w * m.h);
	for (int k = 0; k < n_ids; ++k) { View& v = c->views.find(ids[k])->second; maxIdArea = std::max(maxIdArea, (size_t)v.w * v.h); rc = ensure_gradient(c, v); if (rc) return rc; }
	const size_t tblElems = stride * (size_t)maxNb;
	if (stride >= ((size_t)1 << 29) || tblElems > 0x7FFFFFFFull) return fail(c, HCMVS_ERR_CAPACITY, "postfilter: %d neighbours of %zu pixels exceed the per-pass tables", maxNb, stride);
	// a chain of two or more images: every fusion after the first is computed incrementally when the state fits the device
	// (HCMVS_PF_FULL=1: every fusion from scratch, the path of rounds 1-3, kept as the fall-back and for comparison)
	if (n_ids >= 2 && !getenv("HCMVS_PF_FULL")) {
		launch_unclaim(c->dMaps, (int)host.size(), c->stream); // no claim mark may be left over from a fusion that failed half way
		rc = postfilter_chain_incremental(c, host, ids, n_ids, order, n_order, n_min_views_fuse, depth_diff_threshold, normal_diff_deg, gap_size, n_filled);
		if (rc >= 0) return rc;
	}
	size_t off = 0;
	auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
	const size_t oPending = carve(maxArea * 4), oSettle = carve(fuse_settle_bytes(maxArea)), oTgt = carve(maxArea * 4 * (size_t)maxNb),
	             oHead = carve(tblElems * 4), oNext = carve(maxArea * 4 * (size_t)maxNb), oCtl = carve(kCtlBytes), oCounters = carve(64), oStatus = carve(64),
	             oMerged = carve(maxArea * 4), oFlag = carve(maxArea), oNv = carve(maxArea * 4), oDF = carve(maxIdArea * 4), oDF2 = carve(maxIdArea * 4), oNF = carve(maxIdArea * 12);
	if (c->capPass < off) {
		if (c->passScratch) (void)hipFree(c->passScratch);
		c->passScratch = nullptr; c->capPass = 0;
		HIPCHK(c, malloc_or_release_chain(c, (void**)&c->passScratch, off));
		c->capPass = off;
	}
	char* b = c->passScratch;
	uint32_t* pendingList = (uint32_t*)(b + oPending); uint32_t* ctl = (uint32_t*)(b + oCtl);
	unsigned long long* counters = (unsigned long long*)(b + oCounters);
	uint32_t* status = (uint32_t*)(b + oStatus);
	int32_t* targets = (int32_t*)(b + oTgt);
	uint32_t *head = (uint32_t*)(b + oHead), *next = (uint32_t*)(b + oNext);
	uint8_t* flag = (uint8_t*)(b + oFlag); uint32_t* merged = (uint32_t*)(b + oMerged); uint32_t* pnv = (uint32_t*)(b + oNv);
	float* dF = (float*)(b + oDF); float* dF2 = (float*)(b + oDF2); float* nF = (float*)(b + oNF);
	const float normalError = cosf(normal_diff_deg * (3.14159274101257324f / 180.f)); // plain thresholds (SceneDensify.cpp:2083, 2177)
	const float thDepth = depth_diff_threshold;
	const FuseTables tb = fuse_tables(targets, head, next, stride);
	hipStream_t s = c->stream;
	launch_unclaim(c->dMaps, (int)host.size(), s); // no claim mark may be left over from a fusion or post-filter that failed half way
	HIPCHK(c, hipMemsetAsync(status, 0, 64, s));
	HIPCHK(c, hipMemsetAsync(counters, 0, 64, s)); // counters[5]: pixels filled, summed over the sequence
	for (int k = 0; k < n_ids; ++k) {
		View& v = c->views.find(ids[k])->second;
		for (int oi = 0; oi < n_order; ++oi) {
			const DevMap& A = host[order[oi]];
			HIPCHK(c, hipMemsetAsync(ctl, 0, kCtlBytes, s));
			HIPCHK(c, hipMemsetAsync(head, 0xFF, (size_t)A.nNeighbors * stride * 4, s)); // empty bidder lists
			launch_fuse_begin(A, c->dMaps, tb, pendingList, ctl, flag, counters, thDepth, normalError, s);
			// the fork's RemoveSmallSegments visits the pixels in raster order (SceneDensify.cpp:2130-2131), whatever hcmvs_set_fuse_order says
			// about FuseDepthMaps
			launch_fuse_pass(A, c->dMaps, tb, pendingList, b + oSettle, ctl, nullptr, nullptr, nullptr, pnv, flag, nullptr, nullptr, maxNb + 1, merged, n_min_views_fuse,
			                 0, counters, status, false, s);
		}
		launch_postfilter(v.w, v.h, v.mDepth, v.mNormal, v.mConf, c->dMaps, (int)host.size(), v.gra, dF, dF2, nF, gap_size, depth_diff_threshold * 2.5f, counters + 5, s);
	}
	HIPCHK(c, hipGetLastError());
	uint32_t st[4] = {0, 0, 0, 0};
	unsigned long long cnt[6] = {0, 0, 0, 0, 0, 0};
	HIPCHK(c, hipMemcpyAsync(st, status, 16, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipMemcpyAsync(cnt, counters, 48, hipMemcpyDeviceToHost, s));
	HIPCHK(c, hipStreamSynchronize(s));
	if (st[0] != 0) {
		launch_unclaim(c->dMaps, (int)host.size(), s); // an estimate or a saved map must never see a claim mark (negative depth)
		(void)hipStreamSynchronize(s);
		return fail(c, HCMVS_ERR_TIMEOUT, "postfilter: the settle iteration of a fusion pass gave up; the registered depth maps are left partially fused");
	}
	if (n_filled) *n_filled = cnt[5];
	return HCMVS_OK;
}

int hcmvs_postfilter(hcmvs_ctx* c, uint32_t id, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse, float depth_diff_threshold,
                     float normal_diff_deg, int32_t gap_size, uint64_t* n_filled) {
	return hcmvs_postfilter_sequence(c, &id, 1, order, n_order, n_min_views_fuse, depth_diff_threshold, normal_diff_deg, gap_size, n_filled);
}

int hcmvs_estimate_point_colors(hcmvs_ctx* c, uint64_t n, const float* xyz, const uint32_t* n_views, const uint32_t* view_ids, uint8_t* bgr) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!xyz || !n_views || !view_ids || !bgr) return fail(c, HCMVS_ERR_INVALID, "estimate_point_colors: null argument");
	if (n == 0) return HCMVS_OK;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	std::vector<DevMap> host;
	int rc = build_map_table(c, host);
	if (rc) return rc;
	std::vector<unsigned long long> off(n + 1, 0);
	for (uint64_t i = 0; i < n; ++i) off[i + 1] = off[i] + n_views[i];
	for (unsigned long long k = 0; k < off[n]; ++k)
		if (view_ids[k] >= host.size() || !c->views.count(view_ids[k])) return fail(c, HCMVS_ERR_INVALID, "estimate_point_colors: unknown view %u", view_ids[k]);
	float* dX = nullptr; unsigned long long* dOff = nullptr; uint32_t* dV = nullptr; uint8_t* dC = nullptr;
	auto freeAll = [&]() { for (void* p : {(void*)dX, (void*)dOff, (void*)dV, (void*)dC}) if (p) (void)hipFree(p); };
	if (hipMalloc(&dX, n * 12) != hipSuccess || hipMalloc(&dOff, (n + 1) * 8) != hipSuccess || hipMalloc(&dV, std::max<unsigned long long>(off[n], 1) * 4) != hipSuccess ||
	    hipMalloc(&dC, n * 3) != hipSuccess) { freeAll(); return fail(c, HCMVS_ERR_HIP, "estimate_point_colors: out of device memory"); }
	hipStream_t s = c->stream;
	(void)hipMemcpyAsync(dX, xyz, n * 12, hipMemcpyHostToDevice, s);
	(void)hipMemcpyAsync(dOff, off.data(), (n + 1) * 8, hipMemcpyHostToDevice, s);
	(void)hipMemcpyAsync(dV, view_ids, off[n] * 4, hipMemcpyHostToDevice, s);
	launch_point_colors(n, dX, dOff, dV, c->dMaps, dC, s);
	const hipError_t e1 = hipMemcpyAsync(bgr, dC, n * 3, hipMemcpyDeviceToHost, s);
	const hipError_t e2 = hipStreamSynchronize(s);
	freeAll();
	if (e1 != hipSuccess || e2 != hipSuccess || hipGetLastError() != hipSuccess) return fail(c, HCMVS_ERR_HIP, "estimate_point_colors: device failure");
	return HCMVS_OK;
}

int hcmvs_estimate_point_normals(hcmvs_ctx* c, uint64_t n, const float* xyz, const uint32_t* n_views, const uint32_t* view_ids, int32_t k, float* normal) {
	if (!c) return HCMVS_ERR_INVALID;
	if (!xyz || !n_views || !view_ids || !normal || k < 3 || k > 32) return fail(c, HCMVS_ERR_INVALID, "estimate_point_normals: bad arguments (3 <= k <= 32)");
	if (n == 0) return HCMVS_OK;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	uint32_t maxId = 0;
	for (auto& kv : c->views) maxId = std::max(maxId, kv.first);
	std::vector<double> centres(3 * ((size_t)maxId + 1), 0.0); // camera centres by view id
	for (auto& kv : c->views) for (int q = 0; q < 3; ++q) centres[3 * (size_t)kv.first + q] = kv.second.C[q];
	std::vector<uint32_t> first(n);
	unsigned long long off = 0;
	for (uint64_t i = 0; i < n; ++i) {
		if (n_views[i] < 1) return fail(c, HCMVS_ERR_INVALID, "estimate_point_normals: point %llu has no view", (unsigned long long)i);
		if (!c->views.count(view_ids[off])) return fail(c, HCMVS_ERR_INVALID, "estimate_point_normals: unknown view %u", view_ids[off]);
		first[i] = view_ids[off];
		off += n_views[i];
	}
	std::string err;
	const int rc = hcmvs::pca_normals_device(n, xyz, first.data(), centres.data(), (size_t)maxId + 1, k, normal, c->stream, err);
	if (rc) return fail(c, rc == 1 ? HCMVS_ERR_INVALID : HCMVS_ERR_HIP, "%s", err.c_str());
	return HCMVS_OK;
}

#ifdef HCMVS_STAMPS
// diagnostic build only: per-phase cycle totals of the sweep workers (not part of the public header)
int hcmvs_debug_stamps(hcmvs_ctx* c, unsigned long long* out, int reset) {
	if (!c || !out) return HCMVS_ERR_INVALID;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	hcmvs::debug_read_stamps(out, reset);
	return HCMVS_OK;
}
#endif

} // extern "C"
