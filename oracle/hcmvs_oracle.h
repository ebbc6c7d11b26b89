/*
 * oracle/hcmvs_oracle.h -- TEST INFRASTRUCTURE.  CPU restatement of the HC-MVS (OpenMVS v1.1.1 fork)
 * PatchMatch depth-map estimation, geometric filter and fusion path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library; the
 * product (hc-mvs_amd/, include/) never includes, links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path and cannot be
 * built in this environment (needs OpenCV, Eigen, Boost, CGAL), so this
 * restatement is checked against hand-derived known-answer vectors only (tests/test_oracle_*.py).
 *
 * Citation shorthand (all under /root/reference/frame_main/libs/):
 *   DM.cpp = MVS/DepthMap.cpp, DM.h = MVS/DepthMap.h, SD.cpp = MVS/SceneDensify.cpp,
 *   Util.inl / Types.h / Types.inl / Random.h = Common/...
 *
 * "Defined subset" (SURVEY.md Appendix A): opticalflow=0, use-semantic=0, viewspread=0, nOptimize=0;
 * photometric_flow is honoured as the plain (1-pf) scale it amounts to (DM.cpp:892,931).
 */
#ifndef HCMVS_ORACLE_H
#define HCMVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCOR_MAX_VIEWS 16
/* The reference fixes nSizeHalfWindow = 7 and nTexels = 64 at compile time (DM.h:354-358), i.e. --n-adapthalfwin <= 7.
 * BASELINE.json configs[4] asks for an 11x11 patch (adapthalfwin 10, 121 taps): SURVEY.md 8d item 5 has the restatement
 * generalise the two constants.  The patch loops (DM.cpp:486-494, 554-577) are already written for any `a`; what follows the
 * constant is the border every pass keeps clear (DM.cpp:442-447 uses nSizeHalfWindow): max(7, adapthalfwin). */
#define HCOR_MAX_HALF_WINDOW 10
#define HCOR_MAX_TAPS 121    /* (HCOR_MAX_HALF_WINDOW + 1)^2; the reference's nTexels is 64, DM.h:358 */
#define HCOR_HALF_WINDOW 7   /* nSizeHalfWindow, DM.h:354: the border for adapthalfwin <= 7 */
#define HCOR_MAX_NEIGHBORS 32

enum { HCOR_ARITH_REFERENCE = 0, HCOR_ARITH_DEVICE = 1 };
enum { HCOR_ORDER_ZIGZAG = 0, HCOR_ORDER_ROWS = 1 };

/* one calibrated view: x_cam = R (X - C), pixel = K x_cam / z, pixel centres at integers (Camera.h) */
typedef struct {
	int width, height;
	const float* gray;   /* H*W row-major, [0,1] (Types.inl:2354-2400 toGray) */
	const uint8_t* bgr;  /* optional H*W*3 (B,G,R) working-resolution colour image, may be NULL */
	double K[9], R[9], C[3];
} hcor_view;

/* replaces the OPTDENSE globals (DM.cpp:67-143) for this path */
typedef struct {
	int adapthalfwin;           /* --n-adapthalfwin (DM.cpp:455-461); reference <= 7, generalised to <= HCOR_MAX_HALF_WINDOW */
	int n_estimation_iters;     /* --n-EstimationIters: inner sweeps (SD.cpp:949) */
	int it_external;            /* outer iteration this call is (SD.cpp:758) */
	int n_external_iters;       /* --n-EstimationIters-external: pass C runs when it_external == n-1 */
	int propagate_halfwin;      /* --n-propagatehalfwin (DM.cpp:1071) */
	int propagate_step;         /* --n-propagatestep (DM.cpp:1072) */
	int n_random_iters;         /* nRandomIters = 6 (DM.cpp:120) */
	float ncc_threshold_keep;   /* fNCCThresholdKeep = 0.55 (DM.cpp:117) */
	float random_depth_ratio;   /* fRandomDepthRatio = 0.003 */
	float random_angle1_deg;    /* fRandomAngle1Range = 16 */
	float random_angle2_deg;    /* fRandomAngle2Range = 10 */
	float random_smooth_depth;  /* fRandomSmoothDepth = 0.02 */
	float random_smooth_normal_deg; /* fRandomSmoothNormal = 13 */
	float random_smooth_bonus;  /* fRandomSmoothBonus = 0.93 */
	float photometric_flow;     /* --n-photometric_flow: scores scaled by (1-pf) (DM.cpp:892) */
	uint32_t seed;              /* counter-based RNG seed (replaces mt19937, Random.h:102) */
	int arith_mode;             /* HCOR_ARITH_* */
	int order;                  /* HCOR_ORDER_*: pixel visiting order of the sweeps */
	int n_threads;              /* zig-zag band height = max(64, 8*n_threads) (SD.cpp:835); ROWS order
	                               uses n_threads OpenMP threads and gives the same result for any count */
	int median_blur;            /* 1 = cv::medianBlur(depth,3) at the start of the call (SD.cpp:859) */
	/* restore variant (restore/libs/MVS/DepthMap.cpp:1527-1549): maps of the up-sampled coarser level (W*H depth, W*H*3 normal)
	 * tried as one more hypothesis, with a 0.1 bonus, in the last sweep of the last outer iteration; NULL = frame_main behaviour */
	const float* hint_depth;
	const float* hint_normal;
} hcor_params;

void hcor_default_params(hcor_params* p);

/* DepthData::ViewData::ScaleImage (DM.h:233-238): cv::resize(..., Size(), scale, scale, scale>1 ? INTER_CUBIC : INTER_AREA)
 * on an f32 image, restated from OpenCV 4.2's published algorithm (OpenCV is absent: parity unpinned) */
void hcor_resize_size(int w, int h, float scale, int* dw, int* dh);
void hcor_resize_gray(const float* src, int sw, int sh, float scale, float* dst, int dw, int dh);
/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_AREA) for dw >= sw, dh >= sh (INTER_AREA enlarging = bilinear kernel with area-mode
 * coefficients), ch interleaved f32 channels: the `restore` variant's up-sampling of the previous level's maps
 * (restore/libs/MVS/SceneDensify.cpp:523-524).  Restated from OpenCV 4.2's published source (parity unpinned). */
void hcor_resize_area_up(const float* src, int sw, int sh, int ch, float* dst, int dw, int dh);

/* The device association tests the two END taps of a patch column for "inside the image" (plus: z keeps its sign between them) where
 * the reference tests every tap (DM.cpp:566).  Counters of the device-mode evaluations since the last reset: patch columns tested,
 * and columns on which the two rules disagree. */
void hcor_inside_rule_stats(uint64_t* columns, uint64_t* differ, int reset);

/* ---- small pieces, exposed for known-answer tests ------------------------------------------- */

/* DM.cpp:354-381 MapMatrix2ZigzagIdx (no mask). coords_xy: 2*w*h uint16 (x,y pairs). returns count */
int hcor_zigzag_coords(int w, int h, int raw_stride, uint16_t* coords_xy);

/* counter-based RNG: uniform u32 for (seed, pixel index, stream, counter) */
uint32_t hcor_rand_u32(uint32_t seed, uint32_t pix, uint32_t stream, uint32_t ctr);

/* OpenCV-compatible helpers restated from their published algorithms (OpenCV itself is absent):
 * BGR->gray 8u fixed point, Sobel 3x3 (BORDER_REFLECT_101) + convertScaleAbs + addWeighted(.5,.5)
 * (SD.cpp:581-595 InitGraMap) and 3x3 median with BORDER_REPLICATE (SD.cpp:859). */
void hcor_bgr2gray_u8(const uint8_t* bgr, int w, int h, uint8_t* gray);
void hcor_gray_f32_to_u8(const float* gray, int w, int h, uint8_t* out);
void hcor_gradient_map(const uint8_t* gray, int w, int h, uint8_t* gra);
void hcor_median3(const float* in, int w, int h, float* out);

/* SD.cpp:783-808: splat sparse points (world XYZ, n of them) as 5x5 blocks; returns dMin/dMax */
void hcor_splat_init(const hcor_view* ref, const float* points_xyz, int n_points, float* depth,
                     float* normal, float* d_min, float* d_max);

/* portable math entry points (tests compare them with libm) */
float hcor_pm_expf(float x);
float hcor_pm_sinf(float x);
float hcor_pm_cosf(float x);
float hcor_pm_acosf(float x);
float hcor_pm_atan2f(float y, float x);

/* DM.cpp:450-519 FillPixelPatch for pixel (x,y): fills weight[n], temp_weight[n] (n = (a+1)^2 taps),
 * returns the number of taps; *sum_weights, *norm_sq0 as cached by the reference.  n_src only selects the
 * device-mode summation layout (segments per view) */
int hcor_fill_patch(const hcor_view* ref, const uint8_t* gra, const hcor_params* p, int n_src, int x, int y,
                    float* weight, float* temp_weight, float* sum_weights, float* norm_sq0);

/* DM.cpp:522-616 + 890-893 ScorePixelImage for one source view, no smoothness neighbours */
float hcor_score_view(const hcor_view* ref, const hcor_view* src, const uint8_t* gra,
                      const hcor_params* p, int x, int y, float depth, const float normal[3]);

/* DM.cpp:987-1046 ScorePixel over all views, no smoothness neighbours */
float hcor_score_pixel(const hcor_view* ref, const hcor_view* srcs, int n_src, const uint8_t* gra,
                       const hcor_params* p, int x, int y, float depth, const float normal[3]);

/* Util.inl:614-626 */
void hcor_normal2dir(const float n[3], float p[2], int arith_mode);
void hcor_dir2normal(const float p[2], float n[3], int arith_mode);
/* DM.h:629-634 CorrectNormal for the pixel ray X0 = ((x-cx)/fx,(y-cy)/fy,1) */
void hcor_correct_normal(const hcor_view* ref, int x, int y, float n[3], int arith_mode);
/* DM.cpp:1671-1726 InterpolatePixel: depth at (x,y) of the plane through neighbour (nx,ny,depth,normal) */
float hcor_interpolate_pixel(const hcor_view* ref, int x, int y, int nx, int ny, float depth,
                             const float normal[3], float d_min, float d_max);

/* ---- the path -------------------------------------------------------------------------------- */

/* SD.cpp:758-1072 EstimateDepthMap for one reference view (defined subset).
 * depth/normal/conf: H*W, H*W*3, H*W in/out.  gra: H*W gradient map (SD.cpp:581-595).
 * Runs: [median blur] -> pass A (SD.cpp:649-675) -> n_estimation_iters sweeps (SD.cpp:677-686,
 * DM.cpp:1050-1501) -> pass C when it_external == n_external_iters-1 (SD.cpp:688-744).
 * eval_count (optional): receives the number of ScorePixel evaluations performed. */
int hcor_estimate(const hcor_view* ref, const hcor_view* srcs, int n_src, const uint8_t* gra,
                  const hcor_params* p, float d_min, float d_max, float* depth, float* normal,
                  float* conf, uint64_t* eval_count);

/* the three passes individually (same arguments), for per-pass parity tests */
void hcor_pass_score(const hcor_view* ref, const hcor_view* srcs, int n_src, const uint8_t* gra,
                     const hcor_params* p, float d_min, float d_max, float* depth, float* normal,
                     float* conf, uint64_t* eval_count);
void hcor_pass_sweep(const hcor_view* ref, const hcor_view* srcs, int n_src, const uint8_t* gra,
                     const hcor_params* p, int iter, float d_min, float d_max, float* depth,
                     float* normal, float* conf, uint64_t* eval_count);
void hcor_pass_end(const hcor_params* p, int w, int h, float* depth, float* normal, float* conf);

/* ---- filter and fuse (hcmvs_fuse.c) ----------------------------------------------------------- */

typedef struct {
	int width, height;
	double K[9], R[9], C[3];
	float* depth;        /* H*W, mutated by fusion (SD.cpp:3448-3449) */
	const float* normal; /* H*W*3 camera space, may be NULL */
	const float* conf;   /* H*W */
	const uint8_t* bgr;  /* H*W*3 or NULL */
	float d_min, d_max;
	int n_neighbors;
	const uint32_t* neighbors; /* image ids, decreasing importance */
} hcor_depthmap;

/* SD.cpp:3006-3259 FilterDepthMap(bAdjust).  neighbor_ids index into maps[].  out_depth/out_conf: H*W.
 * returns 0 if too few neighbours (SD.cpp:3016-3019), 1 otherwise; n_discarded and n_processed as logged */
int hcor_filter_depthmap(const hcor_depthmap* maps, uint32_t ref_id, const uint32_t* neighbor_ids,
                         int n_neighbors, int adjust, int n_min_views, int n_min_views_adjust,
                         float depth_diff_threshold, float* out_depth, float* out_conf,
                         uint64_t* n_processed, uint64_t* n_discarded);

typedef struct {
	uint64_t n_points, capacity;
	float* xyz;       /* capacity*3 */
	float* normal;    /* capacity*3 or NULL */
	uint8_t* bgr;     /* capacity*3 or NULL (stored B,G,R like Interface.h:369) */
	uint32_t* n_views;/* capacity: number of views merged into each point */
	uint64_t n_depths;/* valid depths visited (SD.cpp:3359) */
	/* optional: PointCloud::pointViews / pointWeights stored back to back (point p's n_views[p] entries follow those of p-1) */
	uint64_t views_capacity, n_view_entries;
	uint32_t* view_ids;
	float* view_weights;
	/* optional: which pixels of image claim_image ended up in a fused point (arrDepthIdx != NO_ID), W*H bytes */
	uint32_t claim_image;
	uint8_t* claim_mask;
} hcor_cloud;

/* MVS::EstimatePointColors (DM.cpp:2125-2161): colour of the closest view of each point, bilinear over 8-bit pixels with every
 * product and sum truncated to 8 bits (Types.inl:2250-2258, Types.h:1931), white outside.  views: CSR like hcor_cloud */
void hcor_estimate_point_colors(const hcor_depthmap* maps, int n_maps, uint64_t n_points, const float* xyz, const uint32_t* n_views,
                                const uint32_t* view_ids, uint8_t* bgr);

/* SD.cpp:3265-3495 FuseDepthMaps.  order: image ids sorted by #neighbours descending (SD.cpp:3302;
 * ties broken by ascending id here -- std::sort leaves them unspecified).  returns 0 ok, 1 = capacity */
/* pixel visiting order of hcor_fuse_depthmaps / hcor_postfilter: 0 raster (the reference's), 1 the hashed order of the C-ABI's
 * hcmvs_set_fuse_order(ctx, 1) */
void hcor_set_fuse_pixel_order(int mode);
int hcor_fuse_depthmaps(hcor_depthmap* maps, int n_maps, const uint32_t* order, int n_order,
                        int n_min_views_fuse, float depth_diff_threshold, float normal_diff_deg,
                        float depthweight, float normalweight, hcor_cloud* cloud);

/* The fork's depth-map post-filters, applied to every image after outer iterations 1 and 2 (SD.cpp:3939-3958):
 *   RemoveSmallSegments as the fork rewrote it (SD.cpp:2048-2275): a complete FuseDepthMaps pass over the current maps of all
 *     images (it zeroes the estimates fused points occlude, in every image, like the final fusion), after which
 *     depthMap_fuse / normalMap_fuse = this image's maps restricted to the pixels that ended up in a fused point;
 *   GapInterpolation (SD.cpp:2280-3001): gaps of depthMap_fuse along rows, then along columns, are filled by linear
 *     interpolation of depth and of the normal's (atan2, acos) direction when they are at most nIpolGapSize = 7 pixels long and
 *     their ends agree within 2.5 x fDepthDiffThreshold, or longer and either that or the gradient-map values at the ends differ
 *     by at most 10 %; the confidence of a filled pixel is the smaller of the two ends'.  Finally depthMap / normalMap take the
 *     fused-and-filled values wherever those are valid (SD.cpp:2989-3000).
 * Restated: the row pass SD.cpp:2293-2447 and the column pass SD.cpp:2560-2713 (their live branches; `u == size.x` and
 * `(u-count) == 0` can not be true inside the loops).  NOT restated: the third, per-pixel pass (SD.cpp:2717-2983): it reads
 * variables that are never initialised (dir1, dirDiffsum, x1_demin ... at SD.cpp:2745-2760, 2782-2791) and divides by counters
 * that may be zero, so it has no defined result to match.  gra: the image's u8 gradient map.  mode: HCOR_ARITH_*.
 * maps[id]'s depth / normal / conf are updated in place; other images' depths may be zeroed by the fusion.  The fusion inside
 * RemoveSmallSegments uses the plain thresholds (SD.cpp:2083, 2177), not the --depthweight / --normalweight ones of FuseDepthMaps. */
int hcor_postfilter(hcor_depthmap* maps, int n_maps, uint32_t id, const uint8_t* gra, const uint32_t* order, int n_order, int n_min_views_fuse,
                    float depth_diff_threshold, float normal_diff_deg, int gap_size, int mode, uint64_t* n_filled);

#ifdef __cplusplus
}
#endif
#endif
