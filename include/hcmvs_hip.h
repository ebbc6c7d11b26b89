/*
 * include/hcmvs_hip.h -- C-ABI of the MI355X-native PatchMatch densifier (libhcmvs_hip.so).
 *
 * The reference (Liaoyongjian1/HC-MVS, an OpenMVS v1.1.1 fork) has no FFI for this path: it is private
 * C++ inside libMVS.  Each entry point below replaces one of its seams; paths are
 * under /root/reference/frame_main/libs/MVS/:
 *
 *   hcmvs_upload_view / hcmvs_set_view_device
 *        <- DepthMapsData::InitViews (SceneDensify.cpp:336-397): per view gray f32 image + camera
 *   hcmvs_estimate / hcmvs_estimate_device
 *        <- bool DepthMapsData::EstimateDepthMap(int it_external, IIndex idxImage)
 *           (SceneDensify.h:63, SceneDensify.cpp:758-1072) with DepthEstimator (DepthMap.cpp:386-1738)
 *   hcmvs_filter   <- bool DepthMapsData::FilterDepthMap(DepthData&, const IIndexArr&, bool bAdjust)
 *                     (SceneDensify.h:69, SceneDensify.cpp:3006-3259)
 *   hcmvs_fuse     <- void DepthMapsData::FuseDepthMaps(PointCloud&, bool, bool)
 *                     (SceneDensify.h:70, SceneDensify.cpp:3265-3495)
 *   hcmvs_params   <- the OPTDENSE::* globals (DepthMap.cpp:67-143) this path reads
 *
 * Conventions: plain pointers and sizes; caller owns every buffer; int status (0 = ok) instead of bool;
 * no exceptions cross the boundary; hcmvs_last_error() gives the message of the last failure on a
 * context.  One estimate may be in flight per context (the reference serialises EstimateDepthMap under
 * a semaphore of count 1, SceneDensify.cpp:3895); contexts on different devices run concurrently.
 * There is NO CPU fallback: every compute entry fails with HCMVS_ERR_NO_DEVICE when no gfx950 device is
 * usable.
 *
 * Layouts: images/maps row-major; gray f32 in [0,1]; depth f32 (0 = invalid); normal 3 x f32 interleaved,
 * camera space, unit, facing the camera; conf f32 (score in [0,2] between outer iterations, confidence in
 * [0,1] after the last one); K, R 3x3 row-major f64, C 3 x f64, x_cam = R (X - C), pixel = K x_cam / z
 * with pixel centres at integer coordinates (Camera.h:150-151).
 */
#ifndef HCMVS_HIP_H
#define HCMVS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCMVS_MAX_VIEWS 16 /* source views per estimate (reference cap nMaxViews = 12, DepthMap.cpp:73) */

enum {
	HCMVS_OK = 0,
	HCMVS_ERR_NO_DEVICE = 1,   /* no usable gfx950 device / HIP runtime failure at create */
	HCMVS_ERR_INVALID = 2,     /* bad argument (null pointer, unknown view id, size mismatch, V out of range) */
	HCMVS_ERR_HIP = 3,         /* a HIP call failed */
	HCMVS_ERR_TIMEOUT = 4,     /* a sweep worker gave up waiting on its predecessor row (never expected) */
	HCMVS_ERR_CAPACITY = 5     /* output buffer too small (fuse) */
};

typedef struct hcmvs_ctx hcmvs_ctx;

/* replaces the OPTDENSE globals read on this path; hcmvs_default_params() fills the reference defaults */
typedef struct {
	int32_t adapthalfwin;           /* --n-adapthalfwin 1..10, patch half window (DepthMap.cpp:455-461; beyond 7 = more than the reference's 64 taps) */
	int32_t n_estimation_iters;     /* --n-EstimationIters, inner sweeps (SceneDensify.cpp:949) */
	int32_t it_external;            /* outer iteration index of this call (SceneDensify.cpp:758) */
	int32_t n_external_iters;       /* --n-EstimationIters-external; the end pass runs on the last one */
	int32_t propagate_halfwin;      /* --n-propagatehalfwin (DepthMap.cpp:1071), <= 7 */
	int32_t propagate_step;         /* --n-propagatestep (DepthMap.cpp:1072), >= 1 */
	int32_t n_random_iters;         /* nRandomIters (DepthMap.cpp:120) */
	float ncc_threshold_keep;       /* fNCCThresholdKeep (DepthMap.cpp:117) */
	float random_depth_ratio;       /* fRandomDepthRatio */
	float random_angle1_deg;        /* fRandomAngle1Range */
	float random_angle2_deg;        /* fRandomAngle2Range */
	float random_smooth_depth;      /* fRandomSmoothDepth */
	float random_smooth_normal_deg; /* fRandomSmoothNormal */
	float random_smooth_bonus;      /* fRandomSmoothBonus */
	float photometric_flow;         /* --n-photometric_flow: every score is scaled by (1-pf) (DepthMap.cpp:892) */
	uint32_t seed;                  /* counter-based RNG seed; same seed => same maps on any schedule */
	int32_t median_blur;            /* 1: 3x3 median on the depth map first (SceneDensify.cpp:859) */
} hcmvs_params;

typedef struct {
	uint64_t evals;       /* ScorePixel evaluations of the sequential algorithm (each covers all source views) */
	uint64_t evals_issued;/* evaluations the device executed, including discarded speculative ones */
	uint64_t tap_evals;   /* sum over `evals` of the patch taps each one samples per source view (36/49/64) */
	float ms_score;       /* device time of the init-score pass (HIP events on the context's stream) */
	float ms_sweeps;      /* device time of all propagate/refine sweeps */
	float ms_sweep_avg;   /* ms_sweeps / n_sweeps: average duration of one sweep */
	float ms_end;         /* device time of median/end/export kernels */
	float ms_total;
	int32_t n_sweeps;
	int32_t n_sweep_launches; /* kernel launches the n_sweeps sweeps ran in: a batch whose rows fill the chip four times over (12 or more images of 1080p) runs all its sweeps in one launch */
} hcmvs_stats;

void hcmvs_default_params(hcmvs_params* p);

/* device = HIP device ordinal.  Fails with HCMVS_ERR_NO_DEVICE when it is not a usable GPU. */
int hcmvs_create(int device, hcmvs_ctx** out);
void hcmvs_destroy(hcmvs_ctx* ctx);
const char* hcmvs_last_error(const hcmvs_ctx* ctx);
/* stream: a hipStream_t owned by the caller (NULL = the context's own stream).  All work is enqueued
 * on it; the *_device entry points return without synchronising. */
int hcmvs_set_stream(hcmvs_ctx* ctx, void* stream);
int hcmvs_synchronize(hcmvs_ctx* ctx);

/* register view `id` (any number < 65536): host buffers are copied to the device.  bgr (B,G,R u8) is
 * optional: it feeds the gradient map exactly as the reference does (SceneDensify.cpp:586) and the fused
 * colours; without it the gradient map is taken from round(gray*255).  gray may be NULL when bgr is given: such a view can be
 * filtered and fused (camera, colours, gradient map) but not estimated or matched against -- what a rank of a multi-GPU job
 * holds of the images the other ranks estimate. */
int hcmvs_upload_view(hcmvs_ctx* ctx, uint32_t id, int32_t width, int32_t height, const float* gray,
                      const uint8_t* bgr_or_null, const double K[9], const double R[9], const double C[3]);
/* same, but gray/bgr already live in device memory and stay owned by the caller */
int hcmvs_set_view_device(hcmvs_ctx* ctx, uint32_t id, int32_t width, int32_t height, const float* d_gray,
                          const uint8_t* d_bgr_or_null, const double K[9], const double R[9], const double C[3]);
/* DepthData::ViewData::ScaleImage (DepthMap.h:233-238) + Image::GetCamera for the new size (SceneDensify.cpp:372-374,
 * Image.cpp:194-209): register view dst_id as view src_id resampled by `scale` on the device -- cv::resize(image, Size(),
 * scale, scale, scale > 1 ? INTER_CUBIC : INTER_AREA) on the f32 gray image, new size cvRound(w * scale) x cvRound(h * scale),
 * K rescaled by max(w', h') / max(w, h) like the reference's normalised K.  The reference resamples a source view when its
 * average footprint scale differs from the reference image's by 15 % or more; |scale - 1| < 0.15 is refused as it is there.
 * The new view has no colour image (source views are only matched against). */
int hcmvs_rescale_view(hcmvs_ctx* ctx, uint32_t src_id, uint32_t dst_id, float scale);
/* size and camera of a registered view (any pointer may be NULL) */
int hcmvs_get_view_info(hcmvs_ctx* ctx, uint32_t id, int32_t* width, int32_t* height, double K[9]);
/* copy the f32 gray image of a registered view to a host buffer of w*h floats */
int hcmvs_get_view_gray(hcmvs_ctx* ctx, uint32_t id, float* out);
int hcmvs_release_view(hcmvs_ctx* ctx, uint32_t id);
/* copy the u8 gradient map of a view (SceneDensify.cpp:581-595 InitGraMap) to a host buffer of w*h bytes */
int hcmvs_get_gradient_map(hcmvs_ctx* ctx, uint32_t id, uint8_t* out);

/* One EstimateDepthMap call for reference view ref_id against src_ids[0..n_src): [median] -> init score
 * -> n_estimation_iters sweeps -> end pass if it_external == n_external_iters-1.
 * depth (w*h), normal (w*h*3), conf (w*h) are in/out HOST buffers: on entry the initial maps (zeros +
 * splatted sparse depths at it_external 0, SceneDensify.cpp:783-808; the previous maps later).  Blocking. */
int hcmvs_estimate(hcmvs_ctx* ctx, uint32_t ref_id, const uint32_t* src_ids, int32_t n_src,
                   const hcmvs_params* params, float d_min, float d_max, float* depth, float* normal, float* conf);
/* the same on DEVICE buffers, asynchronous on the context's stream */
int hcmvs_estimate_device(hcmvs_ctx* ctx, uint32_t ref_id, const uint32_t* src_ids, int32_t n_src,
                          const hcmvs_params* params, float d_min, float d_max, float* d_depth, float* d_normal,
                          float* d_conf);
/* A batch of independent EstimateDepthMap calls (different reference images, the same options, source-view counts
 * of one class: up to 8, or 9-16) in ONE set of launches: the sweep kernel interleaves the rows of all items, so the images fill each
 * other's wavefront ramps (the reference overlaps images with two worker threads, SceneDensify.cpp:3699).
 * Every item produces exactly the maps hcmvs_estimate_device would produce for it with seed + seed_offset.
 * 1 <= n_items <= HCMVS_MAX_BATCH. */
#define HCMVS_MAX_BATCH 64
typedef struct {
	uint32_t ref_id;
	const uint32_t* src_ids;
	int32_t n_src;
	uint32_t seed_offset;  /* added to params->seed for this item */
	float d_min, d_max;
	float *d_depth, *d_normal, *d_conf; /* DEVICE in/out maps of this item */
	/* optional (NULL = none): DEVICE maps (w*h depth, w*h*3 normal) of the up-sampled coarser level.  With them, the last
	 * sweep of the last outer iteration also tries that estimate at every pixel, accepting it when it is at most 0.1 worse
	 * than the current score -- the extra hypothesis of the fork's `restore` variant (restore/libs/MVS/DepthMap.cpp:1527-1549) */
	const float *d_hint_depth, *d_hint_normal;
} hcmvs_batch_item;
int hcmvs_estimate_batch_device(hcmvs_ctx* ctx, const hcmvs_batch_item* items, int32_t n_items, const hcmvs_params* params);
/* synchronises, then reports counters/timings of the last estimate (summed over the items of a batch) */
int hcmvs_get_stats(hcmvs_ctx* ctx, hcmvs_stats* out);

/* SceneDensify.cpp:783-808: splat n_points sparse world points (xyz f32) seen by view `id` as 5x5 blocks
 * into host maps depth (w*h) / normal (w*h*3); returns the depth range (min*0.9, max*1.1). Host-side helper. */
int hcmvs_splat_init(hcmvs_ctx* ctx, uint32_t id, const float* points_xyz, int32_t n_points, float* depth,
                     float* normal, float* d_min, float* d_max);
/* the same without a context (size and camera given): pure host code, callable from any thread */
int hcmvs_splat_points(int32_t width, int32_t height, const double K[9], const double R[9], const double C[3],
                       const float* points_xyz, int32_t n_points, float* depth, float* normal, float* d_min, float* d_max);

/* DepthMap.cpp:1796-1936 TriangulatePoints2DepthMap (the default initialisation, nMinViewsTrustPoint >= 2) as
 * DepthMapsData::InitDepthMap uses it (SceneDensify.cpp:522-526): Delaunay-triangulate the projections of the sparse
 * points seen by view `id`, add the four image corners as support points when add_corners != 0 (OPTDENSE::bAddCorners;
 * their depth = distance-weighted mean of the three closest faces), rasterise one plane per face into host maps
 * depth (w*h) / normal (w*h*3, camera space); returns the depth range (min*0.9, max*1.1).  avg_depth <= 0: mean depth
 * of the points (Scene.cpp:565-603).  Host-side helper; hcmvs_triangulate_points is the same without a context. */
int hcmvs_triangulate_init(hcmvs_ctx* ctx, uint32_t id, const float* points_xyz, int32_t n_points, float avg_depth,
                           int32_t add_corners, float* depth, float* normal, float* d_min, float* d_max);
int hcmvs_triangulate_points(int32_t width, int32_t height, const double K[9], const double R[9], const double C[3],
                             const float* points_xyz, int32_t n_points, float avg_depth, int32_t add_corners, float* depth,
                             float* normal, float* d_min, float* d_max);

/* cv::resize(src, dst, Size(dst_w, dst_h), 0, 0, INTER_AREA) for an ENLARGING resize of an f32 image with `channels` interleaved
 * channels -- how the fork's `restore` variant brings the previous pyramid level's depth and normal maps to the current size before
 * offering them as the extra hypothesis (restore/libs/MVS/SceneDensify.cpp:523-524; d_hint_depth / d_hint_normal above).  OpenCV's
 * INTER_AREA enlarges with the bilinear kernel and "area mode" coefficients (a destination pixel inside one source pixel copies it,
 * one that straddles two mixes them by the overlap).  Host buffers, host code, no context (once per image and level).  dst_w >= src_w
 * and dst_h >= src_h, else HCMVS_ERR_INVALID. */
int hcmvs_resize_area_up(const float* src, int32_t src_w, int32_t src_h, int32_t channels, float* dst, int32_t dst_w, int32_t dst_h);

/* ---- filter and fuse: work on the estimated maps registered per view ------------------------------------ */

/* register the maps of view `id` (host buffers are copied).  normal may be NULL.  d_min/d_max: the depth range
 * FilterDepthMap clips the adjusted depth to (SceneDensify.cpp:3156). */
int hcmvs_set_depthmap(hcmvs_ctx* ctx, uint32_t id, const float* depth, const float* normal_or_null,
                       const float* conf, float d_min, float d_max);
/* same for maps that already live in device memory; d_depth is MUTATED by hcmvs_fuse like the reference does
 * (SceneDensify.cpp:3447-3449).  Depths must be >= 0 (0 = no estimate): while a fusion or a post-filter runs, the sign of a depth
 * marks the estimates that already belong to a point; it is restored before the call returns. */
int hcmvs_set_depthmap_device(hcmvs_ctx* ctx, uint32_t id, float* d_depth, const float* d_normal_or_null,
                              const float* d_conf, float d_min, float d_max);
/* read the (possibly mutated) maps back; any pointer may be NULL */
int hcmvs_get_depthmap(hcmvs_ctx* ctx, uint32_t id, float* depth, float* normal, float* conf);
/* DepthData::neighbors of view `id`: image ids in decreasing importance (Scene.cpp:545-678), at most 31 for fuse */
int hcmvs_set_neighbors(hcmvs_ctx* ctx, uint32_t id, const uint32_t* ids, int32_t n);

/* bool DepthMapsData::FilterDepthMap(DepthData&, const IIndexArr&, bool bAdjust) (SceneDensify.cpp:3006-3259).
 * out_depth / out_conf: w*h host buffers (the reference writes them to filtered.dmap/.cmap).
 * Fails with HCMVS_ERR_INVALID when there are fewer neighbours than n_min_views (the reference returns false). */
int hcmvs_filter(hcmvs_ctx* ctx, uint32_t ref_id, const uint32_t* neighbor_ids, int32_t n_neighbors, int32_t adjust,
                 int32_t n_min_views, int32_t n_min_views_adjust, float depth_diff_threshold, float* out_depth,
                 float* out_conf, uint64_t* n_processed, uint64_t* n_discarded);

/* Order in which the pixels of ONE image are visited by hcmvs_fuse (the image order is always the caller's).
 * 0 (default): raster order, the reference's (SceneDensify.cpp:3355-3358); the cloud equals the sequential one bit for bit.
 * 1: a fixed pseudo-random order (a bijective hash of the raster index): the same greedy rule visits the pixels in
 *    another order, and the cloud differs from the reference's within the tolerance the north star states (point count
 *    within 1 %); the output is still written in raster order and deterministic.  (The option dates from the rounds in
 *    which the order decided how long the fusion took; it no longer does.) */
int hcmvs_set_fuse_order(hcmvs_ctx* ctx, int32_t mode);

/* void DepthMapsData::FuseDepthMaps(PointCloud&, bool, bool) (SceneDensify.cpp:3265-3495).  order: image ids,
 * best connected first (SceneDensify.cpp:3302).  Output host buffers hold `capacity` points: xyz 3 f32, normal
 * 3 f32 or NULL, bgr 3 u8 (B,G,R) or NULL, n_views u32 or NULL.  *n_points / *n_depths are the numbers the
 * reference logs (SceneDensify.cpp:3461).  Points come out in the reference's order. */
int hcmvs_fuse(hcmvs_ctx* ctx, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse,
               float depth_diff_threshold, float normal_diff_deg, float depthweight, float normalweight,
               uint64_t capacity, float* xyz, float* normal_or_null, uint8_t* bgr_or_null, uint32_t* n_views_or_null,
               uint64_t* n_points, uint64_t* n_depths);

/* The fork's depth-map post-filters, which the reference applies to every image after outer iterations 1 and 2
 * (SceneDensify.cpp:3939-3958), on the registered DEVICE maps of view `id` (hcmvs_set_depthmap_device; depth, normal and conf
 * are updated in place):
 *   RemoveSmallSegments as the fork rewrote it (SceneDensify.cpp:2048-2275): a complete FuseDepthMaps pass over the maps of
 *   `order` -- it zeroes the estimates that fused points occlude, in every image, exactly like hcmvs_fuse -- which leaves
 *   depthMap_fuse / normalMap_fuse = the image's maps restricted to the pixels that ended up in a fused point;
 *   GapInterpolation (SceneDensify.cpp:2280-3001): gaps of at most gap_size (nIpolGapSize = 7) pixels along rows, then
 *   columns, whose ends agree within 2.5 x depth_diff_threshold -- or longer ones whose ends agree or whose gradient-map values
 *   differ by at most 10 % -- are filled by linear interpolation of depth and normal direction; finally the maps take the
 *   fused-and-filled values where those are valid.
 * RemoveSmallSegments compares with the plain thresholds (SceneDensify.cpp:2083, 2177): --depthweight / --normalweight only scale the
 * thresholds of FuseDepthMaps (SceneDensify.cpp:3310, 3400), so this entry takes no weights; and it visits the pixels of an image in
 * raster order (SceneDensify.cpp:2130-2131) whatever hcmvs_set_fuse_order says about FuseDepthMaps.
 * The reference's third, per-pixel pass (SceneDensify.cpp:2790-2983) reads uninitialised variables and is not reproduced.
 * n_filled: pixels the two interpolation passes wrote. */
int hcmvs_postfilter(hcmvs_ctx* ctx, uint32_t id, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse, float depth_diff_threshold,
                     float normal_diff_deg, int32_t gap_size, uint64_t* n_filled);
/* The same for the images ids[0 .. n_ids) one after the other: identical in effect to n_ids calls of hcmvs_postfilter in that order
 * (every image's fusion sees the maps the images before it left), with the per-call set-up paid once and every fusion after the first
 * computed INCREMENTALLY -- only what depends on the estimates that changed since the previous fusion (the pixels the previous image's
 * gap interpolation filled, the estimates that fusion zeroed) is evaluated again; same decisions, fusion after fusion (the state kept
 * between the fusions is about 360 B per pixel of the scene; when it does not fit the device, or an image occurs twice in ids, every
 * fusion runs from scratch).  n_filled: total over the images.
 * NOTE on the order: the reference filters image k right after image k's own estimate (single-thread event order,
 * SceneDensify.cpp:3889-3965), i.e. hcmvs_estimate(k), hcmvs_postfilter(k), hcmvs_estimate(k + 1), ...  Calling this entry AFTER all
 * estimates of an outer iteration is the batch schedule of this repo's driver -- a documented departure. */
int hcmvs_postfilter_sequence(hcmvs_ctx* ctx, const uint32_t* ids, int32_t n_ids, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse,
                              float depth_diff_threshold, float normal_diff_deg, int32_t gap_size, uint64_t* n_filled);

/* The fused cloud with everything PointCloud holds (PointCloud.h: points, pointViews, pointWeights, colors, normals).
 * Host buffers owned by the caller; any optional pointer may be NULL.  The view lists are stored back to back (CSR): point p's
 * n_views[p] entries follow those of point p-1, image ids ascending like PointCloud::ViewArr (SceneDensify.cpp:3376-3379,
 * 3408-3411); a point merges at most one depth per image, so views_capacity = the number of valid depths is always enough. */
typedef struct {
	uint64_t capacity;        /* in: room for this many points */
	float* xyz;               /* capacity * 3 */
	float* normal;            /* capacity * 3 or NULL */
	uint8_t* bgr;             /* capacity * 3 (B,G,R) or NULL */
	uint32_t* n_views;        /* capacity or NULL */
	uint64_t views_capacity;  /* in: room for this many view entries (0 with NULL lists) */
	uint32_t* view_ids;       /* views_capacity or NULL */
	float* view_weights;      /* views_capacity or NULL: Conf2Weight of the merged estimates (SceneDensify.cpp:154-156) */
	uint64_t n_points, n_depths, n_view_entries; /* out */
} hcmvs_cloud;
/* hcmvs_fuse with the complete cloud.  With cloud->xyz == NULL the call only COUNTS: the fusion runs (with its side effect, the
 * invalidated depths) and n_points / n_depths / n_view_entries come back, no point is produced.  A fusion repeated on the maps a fusion
 * has left makes the same decisions (what it invalidated is simply absent the second time), so a counting call followed by a call with
 * buffers of exactly that size yields the cloud a single call with large enough buffers would have produced -- how a caller avoids
 * reserving the worst case (half a point per pixel of the scene). */
int hcmvs_fuse_cloud(hcmvs_ctx* ctx, const uint32_t* order, int32_t n_order, int32_t n_min_views_fuse, float depth_diff_threshold,
                     float normal_diff_deg, float depthweight, float normalweight, hcmvs_cloud* cloud);

/* MVS::EstimatePointColors (DepthMap.cpp:2125-2161; --estimate-colors 1): per point the colour of the closest of its views,
 * sampled bilinearly from that view's colour image (white when it projects outside).  Views must have been registered with a
 * colour image.  Host arrays: xyz n*3, n_views n, view_ids (CSR as in hcmvs_cloud), out bgr n*3. */
int hcmvs_estimate_point_colors(hcmvs_ctx* ctx, uint64_t n_points, const float* xyz, const uint32_t* n_views, const uint32_t* view_ids,
                                uint8_t* bgr);
/* MVS::EstimatePointNormals (DepthMap.cpp:2221-2269; --estimate-normals 1): normal of the plane fitted by PCA to the
 * n_neighbors nearest points of every point (the reference calls CGAL::pca_estimate_normals with 16), oriented towards the first
 * view of the point.  Computed on the device (exact k-nearest search + PCA, 3 <= n_neighbors <= 32, fewer than 2^31 points);
 * CGAL is absent, so the PCA is restated (parity unpinned).  Host arrays in and out. */
int hcmvs_estimate_point_normals(hcmvs_ctx* ctx, uint64_t n_points, const float* xyz, const uint32_t* n_views, const uint32_t* view_ids,
                                 int32_t n_neighbors, float* normal);

#ifdef __cplusplus
}
#endif
#endif
