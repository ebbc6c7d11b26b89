/*
 * hc-mvs_amd/csrc/pm_kernels.hip -- gfx950 kernels of the PatchMatch depth-map estimation path.
 *
 * What the reference computes (paths under /root/reference/frame_main/libs/MVS/):
 *   FillPixelPatch      DepthMap.cpp:450-519   bilateral patch weights of the reference pixel
 *   ScorePixelImage     DepthMap.cpp:522-616   homography warp + weighted ZNCC + plane smoothness
 *   ScorePixel          DepthMap.cpp:987-1046  mean of the two best views
 *   ProcessPixel        DepthMap.cpp:1050-1501 propagate from neighbours + random refinement
 *   ScoreDepthMapTmp / EstimateDepthMapTmp / EndDepthMapTmp   SceneDensify.cpp:649-744
 *
 * How it is mapped to CDNA4 (not a translation of the reference's pthread loop):
 *   - ONE PERSISTENT WORKER (NW waves, one workgroup) PER IMAGE ROW.  The reference sweeps pixels sequentially
 *     (Gauss-Seidel): a pixel must see its left/up neighbours already updated and its right/down neighbours not
 *     yet updated.  Rows advancing left-to-right with row y one pixel behind row y-1 satisfy exactly that
 *     dependence, so all rows run concurrently and hand results down through L2 with agent-scope (sc1) stores
 *     + a per-row progress word.  The maps are identical to the sequential sweep.  The rows of several
 *     independent reference images share one launch (ticket t -> row t / nItems of item t % nItems) so the ramp
 *     of one image's wavefront is filled by the others.
 *   - INSIDE A WAVE the 64 lanes are 8 view groups x 8 tap segments: lane = 8 * view + seg, for every number of source
 *     views (9..16 views: two sets of eight groups, TWO; view counts that leave groups idle let them score (hypothesis,
 *     view) pairs of their own, PACK).  A segment owns one patch column and walks down its rows; partial sums are
 *     combined with DPP butterflies inside the group, the per-view ZNCC epilogue runs lane-parallel over (hypothesis,
 *     view) pairs and the "two best views" selection is a second butterfly across groups.  Hypothesis generation and the
 *     plane-smoothness terms are lane-parallel too.  LDS holds the per-wave tables (patch weights, view constants,
 *     neighbour slots, chunk homographies), the row's own recent results (hist ring) and, for NW > 1, the score exchange.
 *   - Rows are handed out by an atomic ticket in dependency order, so a waiting worker always waits on one
 *     that is already running: no deadlock for any grid size or dispatch order.  Every spin is bounded.
 *
 * Arithmetic is an explicitly specified IEEE sequence (explicit fmaf, one IEEE reciprocal per lane and evaluation,
 * pm_math.h transcendental functions); compile with -ffp-contract=off.
 */
#include "pm_common.h"
#include "pm_math.h"

#include <float.h>
#include <type_traits>

namespace hcmvs {

#define HC_SQ(x) ((x) * (x))

// diagnostic build only (-DHCMVS_STAMPS): per-phase cycle accounting of wave 0 of every row worker
#ifdef HCMVS_STAMPS
__device__ unsigned long long g_stamps[32]; // [0, 16) cycles per phase, [16, 32) executions of the blocks of BLOCK()
#define STAMP_DECL unsigned long long st_last = __builtin_amdgcn_s_memtime(), st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned st_cnt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; }
#define STAMP_ARGS , unsigned long long& st_last, unsigned long long (&st_acc)[16], unsigned (&st_cnt)[16]
#define STAMP_PASS , st_last, st_acc, st_cnt
#define STAMP_FLUSH if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_stamps[i_], st_acc[i_]); for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_stamps[16 + i_], (unsigned long long)st_cnt[i_]); }
#define SUBMARK(name)
#define BLOCK(name, i) ++st_cnt[i];
#define BLOCKN(name, i, n) st_cnt[i] += (unsigned)(n);
#elif defined(HCMVS_MARK)
// diagnostic ISA listing only (-DHCMVS_MARK -S, tools/isa_budget.py): the phase boundaries of the stamps build as comments in the
// assembly, plus sub-phases of the scorer, so that the static instruction mix can be attributed to the source phases
#define STAMP_DECL
#define STAMP(i) asm volatile("; HCMARK " #i);
#define SUBMARK(name) asm volatile("; HCMARK " #name);
#define BLOCK(name, i) asm volatile("; HCMARK " #name);
#define BLOCKN(name, i, n) asm volatile("; HCMARK " #name);
#define STAMP_ARGS
#define STAMP_PASS
#define STAMP_FLUSH
#else
#define STAMP_DECL
#define STAMP(i)
#define SUBMARK(name)
#define BLOCK(name, i)
#define BLOCKN(name, i, n)
#define STAMP_ARGS
#define STAMP_PASS
#define STAMP_FLUSH
#endif
#define HC_SCOPE __HIP_MEMORY_SCOPE_AGENT

// ------------------------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
	h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
	return h;
}
// counter-based RNG keyed by (seed, pixel, pass, draw): replaces the per-thread mt19937 (Random.h:102).
// rand_key() hashes the per-pixel part once; rand_unit() finishes one draw.
__device__ __forceinline__ uint32_t rand_key(uint32_t seed, uint32_t pix, uint32_t stream) {
	uint32_t h = fmix32(seed ^ 0x9e3779b9u);
	h = fmix32(h ^ (pix * 0x9e3779b1u));
	return fmix32(h ^ (stream * 0x85ebca77u));
}
__device__ __forceinline__ float rand_unit(uint32_t key, uint32_t ctr) {
	return (float)fmix32(key ^ (ctr * 0xc2b2ae3du)) / 4294967296.0f; // (float)max() == 2^32 (Random.h:113-115)
}
__device__ __forceinline__ float fd2r(float d) { return d * (3.14159274101257324f / 180.f); } // Types.h:566

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
	return a0 * b0 + a1 * b1 + a2 * b2;
}

// Every pointer held in an EstConst addresses device global memory.  Where the struct is read from memory (batched
// sweep) the compiler only sees generic pointers and would emit flat_* instructions with 64-bit VGPR addresses; the
// explicit global address space keeps them global_* (scalar base + 32-bit offset).
#define HC_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ HC_GLOBAL T* as_global(T* p) { return (HC_GLOBAL T*)p; }
// One byte of a map that no kernel of this launch writes (the gradient map), at a wave-uniform index: read through the
// constant address space it becomes a scalar load -- it lands in an SGPR behind lgkmcnt and never makes the wave drain
// its vector-memory queue (a vector load of a uniform byte is turned into v_readfirstlane behind `s_waitcnt vmcnt(0)`,
// i.e. a full memory round trip per pixel with every prefetch of the pixel just issued in front of it).
__device__ __forceinline__ uint8_t uniform_byte(const uint8_t* base, int idx) {
	typedef __attribute__((address_space(4))) const uint32_t* cu32p;
	const uintptr_t a = (uintptr_t)base + (uintptr_t)(unsigned)idx;
	const uint32_t w = *(cu32p)(a & ~(uintptr_t)3);
	return (uint8_t)(w >> (8u * (unsigned)(a & 3u)));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// agent-scope (sc1) accesses for everything another row's wave may have written in this launch
__device__ __forceinline__ float4 load_dn(const float4* p) {
	HC_GLOBAL unsigned long long* q = (HC_GLOBAL unsigned long long*)p;
	const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, HC_SCOPE);
	const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, HC_SCOPE);
	return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b),
	                   __uint_as_float((uint32_t)(b >> 32)));
}
__device__ __forceinline__ void store_dn(float4* p, float d, float n0, float n1, float n2) {
	HC_GLOBAL unsigned long long* q = (HC_GLOBAL unsigned long long*)p;
	__hip_atomic_store(q, (unsigned long long)__float_as_uint(d) | ((unsigned long long)__float_as_uint(n0) << 32),
	                   __ATOMIC_RELAXED, HC_SCOPE);
	__hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(n1) | ((unsigned long long)__float_as_uint(n2) << 32),
	                   __ATOMIC_RELAXED, HC_SCOPE);
}
__device__ __forceinline__ float load_f(const float* p) {
	return __uint_as_float(__hip_atomic_load((HC_GLOBAL uint32_t*)p, __ATOMIC_RELAXED, HC_SCOPE));
}
__device__ __forceinline__ void store_f(float* p, float v) {
	__hip_atomic_store((HC_GLOBAL uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, HC_SCOPE);
}

__device__ __forceinline__ float rlf(float v, int lane) { // value of a (wave-uniform index) lane
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int rli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// value of lane (l ^ STEP).  Steps 1,2 are quad permutes; 4 and 8 use the DPP half-row / row mirrors, which
// coincide with the xor partner's VALUE because the butterflies below make every 4- resp. 8-lane block
// uniform before those steps; 16 is a swizzle inside 32 lanes; 32 goes through the LDS crossbar.
template <int STEP>
__device__ __forceinline__ float lane_xor(float v) {
	if (STEP == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)); // quad_perm [1,0,3,2]
	if (STEP == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)); // quad_perm [2,3,0,1]
	if (STEP == 4) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)); // row_half_mirror
	if (STEP == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)); // row_mirror
	if (STEP == 16) return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));                    // xor 0x10
	return __shfl_xor(v, 32, 64);
}
template <int S, int STEP = 1>
__device__ __forceinline__ float group_sum(float v) { // xor butterfly inside a view group of S lanes
	if constexpr (STEP < S) {
		v = v + lane_xor<STEP>(v);
		return group_sum<S, STEP * 2>(v);
	} else {
		return v;
	}
}
// two smallest of the per-group scores, butterfly across the view groups (every group is uniform inside)
template <int STEP>
__device__ __forceinline__ void min2_across(float& m1, float& m2) {
	if constexpr (STEP < 64) {
		const float o1 = lane_xor<STEP>(m1), o2 = lane_xor<STEP>(m2);
		const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
		m2 = fminf(hi, fminf(m2, o2));
		m1 = lo;
		min2_across<STEP * 2>(m1, m2);
	}
}

// the same inside aligned groups of NV lanes (one lane per view)
template <int NV, int STEP = 1>
__device__ __forceinline__ void min2_group(float& m1, float& m2) {
	if constexpr (STEP < NV) {
		const float o1 = lane_xor<STEP>(m1), o2 = lane_xor<STEP>(m2);
		const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
		m2 = fminf(hi, fminf(m2, o2));
		m1 = lo;
		min2_group<NV, STEP * 2>(m1, m2);
	}
}

// ------------------------------------------------------------------------------------------------------
// per-lane context

typedef const __attribute__((address_space(1))) float* gcfptr; // global-address-space loads, not flat

template <int S>
struct LaneCtx {
	static constexpr int MAXM = 64 / S; // taps per lane
	int lane, view, seg;
	int vloc;         // view group of the lane inside the wave (= view, except in the second view set of a 9..16-view estimate)
	bool vact;        // lane's view group exists
	unsigned imgOff;  // byte offset of my source view's footprint image from EstConst::imgBase (all views of a call lie within 4 GiB)
	int iw, ixmax, iymax; // row pitch in pixels; largest top-left texel column / row of a bilinear footprint
	float wmax, hmax; // inside-with-border-1 limits of my view
	unsigned long long groupMask;
};

template <int S>
__device__ __forceinline__ void lane_init(const EstConst& c, LaneCtx<S>& L, int vbase = 0) {
	L.lane = threadIdx.x & 63;
	L.vloc = L.lane / S;
	L.view = vbase + L.vloc;
	L.seg = L.lane % S;
	L.vact = L.view < c.V;
	const HC_GLOBAL DevView* dv = as_global(c.views) + (L.vact ? L.view : 0); // idle groups mirror view 0 (results masked)
	L.imgOff = dv->byteOff; L.iw = dv->w; L.ixmax = dv->w - 2; L.iymax = dv->h - 2;
	L.wmax = (float)(dv->w - 2); L.hmax = (float)(dv->h - 2);
	L.groupMask = (S == 64 ? ~0ull : ((1ull << S) - 1ull)) << (L.vloc * S);
}

template <int S>
struct Patch { // DepthMap.h:202-212 WeightedPatchFix, spread over the lanes of a group
	static constexpr int MAXM = 64 / S;
	float px0;        // image column of the lane's taps (a segment owns one patch column)
	float sumW, invSumW, normSq0;
	int x, y, a;
};

// Where the scorer's per-lane tables live between evaluations: the per-tap arrays of the patch (image row py, weight
// w, centred weighted reference texel tw; DepthMap.h:202-212) and the homography constants of the lane's view
// (DepthMap.h:412-444).  The init-score pass keeps them in registers; the sweep parks them in LDS, because there
// the state of the pixel's hypothesis rounds is live across every evaluation and registers decide the occupancy.
template <int S>
struct RegStore {
	static constexpr int MAXM = 64 / S;
	float A[9], Hm[3];
	float py[MAXM], w[MAXM], tw[MAXM];
	float (*stage)[8][8]; // [3][column][row] LDS of this wave, the hand-over between the patch and the scorer lane layouts
	float (*bw)[kBigSlots]; // big-patch kernels: [4][slot] px | py | w | tw of this wave (LDS), see fill_patch_big
	int seg;
	// the lane that computed tap (row, col) hands it to the scorer lanes of column col
	__device__ __forceinline__ void put_patch64(int row, int col, float py_, float w_, float tw_) {
		stage[0][col][row] = py_; stage[1][col][row] = w_; stage[2][col][row] = tw_; // same-wave LDS accesses are ordered
#pragma unroll
		for (int m = 0; m < MAXM; ++m) { py[m] = stage[0][seg][m]; w[m] = stage[1][seg][m]; tw[m] = stage[2][seg][m]; }
	}
	__device__ __forceinline__ void put_view(const HC_GLOBAL DevView* dv, int) {
#pragma unroll
		for (int i = 0; i < 9; ++i) A[i] = dv->A[i];
#pragma unroll
		for (int i = 0; i < 3; ++i) Hm[i] = dv->Hm[i];
	}
	__device__ __forceinline__ void copy_patch(const RegStore& o) {
#pragma unroll
		for (int m = 0; m < MAXM; ++m) { py[m] = o.py[m]; w[m] = o.w[m]; tw[m] = o.tw[m]; }
	}
	__device__ __forceinline__ void get_view(float (&A_)[9], float (&Hm_)[3]) const {
#pragma unroll
		for (int i = 0; i < 9; ++i) A_[i] = A[i];
#pragma unroll
		for (int i = 0; i < 3; ++i) Hm_[i] = Hm[i];
	}
	__device__ __forceinline__ void get_py(float (&o)[MAXM]) const {
#pragma unroll
		for (int m = 0; m < MAXM; ++m) o[m] = py[m];
	}
	__device__ __forceinline__ void get_w(float (&w_)[MAXM], float (&tw_)[MAXM]) const {
#pragma unroll
		for (int m = 0; m < MAXM; ++m) { w_[m] = w[m]; tw_[m] = tw[m]; }
	}
};
template <int S>
struct WavePark { // LDS of one wave of a row worker
	static constexpr int MAXM = 64 / S;
	float4 vh[16][3];         // per view: A[0..8], Hm[0..2] (two sets of eight views, see TWO in the kernels)
	float ps[3][8][8];        // py | w | tw as [column][row], read by every view group
	float cl[9][kMaxSlots];   // smoothness neighbours, slot k at [.][k]: X0 X1 X2 | n0 n1 n2 | k0 k1 k2 (see Close)
	float4 hl[8][64 / S][3];  // homographies of the (hypothesis, view) pairs of the current chunk of eight hypotheses
#ifdef HCMVS_PARTHALF
	float part[3][8][64 / S][S / 2]; // their ZNCC sums (sum | sumSq | num), neighbouring lane pairs of the view group already added (the butterfly's first step)
#else
	float part[3][8][64 / S][S]; // their ZNCC sums (sum | sumSq | num) as the S lanes of the view group left them: added up per chunk, not per evaluation
#endif
};
// park the lane's partial sums of pair (g, v) for the chunk epilogue
template <int S>
__device__ __forceinline__ void park_partials(WavePark<S>* pk, int g, int v, int seg, bool valid, float sum, float sumSq, float num) {
#ifdef HCMVS_PARTHALF
	sum = sum + lane_xor<1>(sum); sumSq = sumSq + lane_xor<1>(sumSq); num = num + lane_xor<1>(num);
	if (valid && !(seg & 1)) { pk->part[0][g][v][seg >> 1] = sum; pk->part[1][g][v][seg >> 1] = sumSq; pk->part[2][g][v][seg >> 1] = num; }
#else
	if (valid) { pk->part[0][g][v][seg] = sum; pk->part[1][g][v][seg] = sumSq; pk->part[2][g][v][seg] = num; }
#endif
}
template <int S>
struct LdsStore {
	static constexpr int MAXM = 64 / S;
	WavePark<S>* pk;
	float (*bw)[kBigSlots]; // big-patch kernels: [4][slot] px | py | w | tw of this wave (LDS), see fill_patch_big
	int lane, view;
	__device__ __forceinline__ void put_view(const HC_GLOBAL DevView* dv, int seg) {
		if (seg == 0) {
			pk->vh[view][0] = make_float4(dv->A[0], dv->A[1], dv->A[2], dv->A[3]);
			pk->vh[view][1] = make_float4(dv->A[4], dv->A[5], dv->A[6], dv->A[7]);
			pk->vh[view][2] = make_float4(dv->A[8], dv->Hm[0], dv->Hm[1], dv->Hm[2]);
		}
	}
	__device__ __forceinline__ void put_patch64(int row, int col, float py_, float w_, float tw_) {
		pk->ps[0][col][row] = py_; pk->ps[1][col][row] = w_; pk->ps[2][col][row] = tw_;
	}
	__device__ __forceinline__ void get_col(int k, int os, float (&v)[MAXM]) const { // the eight rows of my column
		const float4* q = (const float4*)&pk->ps[k][os][0];
		const float4 a = q[0], b = q[1];
		v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
	}
	// the index is laundered so that the reads stay inside the evaluation (hoisting them would undo the parking)
	__device__ __forceinline__ int opaque(int v) const { asm volatile("" : "+v"(v)); return v; }
	__device__ __forceinline__ void get_view(float (&A_)[9], float (&Hm_)[3]) const {
		const int ov = opaque(view);
		const float4 a = pk->vh[ov][0], b = pk->vh[ov][1], cc = pk->vh[ov][2];
		A_[0] = a.x; A_[1] = a.y; A_[2] = a.z; A_[3] = a.w; A_[4] = b.x; A_[5] = b.y; A_[6] = b.z; A_[7] = b.w; A_[8] = cc.x;
		Hm_[0] = cc.y; Hm_[1] = cc.z; Hm_[2] = cc.w;
	}
	__device__ __forceinline__ void get_py(float (&o)[MAXM]) const {
		get_col(0, opaque(lane & 7), o);
	}
	__device__ __forceinline__ void get_w(float (&w_)[MAXM], float (&tw_)[MAXM]) const {
		const int os = opaque(lane & 7);
		get_col(1, os, w_); get_col(2, os, tw_);
	}
};

// everything one pixel reads from memory; the sweep loads it one pixel ahead (software pipeline)
template <int S>
struct PixIn {
	static constexpr int MAXM = 64 / S;
	float I[MAXM];  // reference-image taps of this lane (DepthMap.cpp:486-494)
	float center;   // I0(x,y)
	float tx;       // gradient-map value of (x,y) (DepthMap.cpp:455)
	float4 cur;     // previous estimate of (x,y): depth, normal
	float curConf;
	// neighbour slot held by this lane (slot k lives in lane k)
	int nx, ny;
	bool slot, sprop;
	int back;       // > 0: the slot is `back` columns behind me in my own row -> comes from the row ring
	bool isUp;      // the slot lies in an earlier logical row -> needs that row's progress hand-off
	bool loaded;    // payload below is valid
	float4 ndn;
	float nconf;
};

__device__ __forceinline__ int patch_halfwin(const EstConst& c, float tx) { return tx > 100.f ? 5 : c.adapthalfwin; } // DepthMap.cpp:455-461

// The scorer's lanes are 8 view groups x 8 tap segments: a segment owns ONE patch column and walks down its rows, so (a) the
// lanes of a view group read the same source-image row in each load instruction -- a handful of cache lines per wave-instruction
// instead of 64 -- and (b) a lane's warped position is affine in the step.  (Round 2 started with one layout per view-count class;
// every view count now runs this one.)
template <int S>
__device__ __forceinline__ void load_patch_inputs(const EstConst& c, const LaneCtx<S>& L, int x, int y, PixIn<S>& in) {
	static_assert(S == 8, "one lane layout: 8 view groups x 8 tap segments");
	gcfptr ref = (gcfptr)c.ref;
	const int a = patch_halfwin(c, in.tx);
	in.center = ref[y * c.W + x];
	// one tap of the patch per lane: lane = 8 * row + column (lanes past the patch repeat its last row / column)
	const int row = L.lane >> 3, col = L.lane & 7;
	const int i = -a + 2 * (row < a ? row : a), j = -a + 2 * (col < a ? col : a);
	in.I[0] = ref[__mul24(y + i, c.W) + (x + j)];
}

// DepthMap.cpp:450-519 FillPixelPatch + DepthMap.h:537-548 GetWeight.
// Lanes whose tap index is past the patch repeat the last tap with zero weights: they add exactly +0.
// Every tap of the (a + 1)^2 patch gets its own lane for the weights (lane = 8 * row + column;
// lanes past the patch add exactly +0), the three sums are 64-lane butterflies, and the results are handed to the
// scorer's layout (a segment = one column, all view groups alike) through 768 bytes of LDS.
template <class ST>
__device__ __forceinline__ void fill_patch_64(const EstConst& c, const LaneCtx<8>& L, int x, int y, int a, const PixIn<8>& in, Patch<8>& P, ST& st) {
	const int row = L.lane >> 3, col = L.lane & 7;
	const bool valid = row <= a && col <= a;
	const int i = -a + 2 * (row < a ? row : a), j = -a + 2 * (col < a ? col : a);
	const float sigmaColor = -1.f / (2.f * HC_SQ(0.2f));
	const float sigmaSpatial = -1.f / (2.f * (float)HC_SQ(a));
	const float I = in.I[0];
	const float wColor = HC_SQ(I - in.center) * sigmaColor;
	const float wSpatial = (float)(HC_SQ(j) + HC_SQ(i)) * sigmaSpatial;
	const float we = pm_expf(wColor + wSpatial);
	const float w = valid ? we : 0.f;
	const float swi = group_sum<64>(I * w), sw = group_sum<64>(w);
	const float tm = swi / sw;
	const float t = I - tm;
	const float tw = w * t;
	P.sumW = sw;
	P.invSumW = 1.0f / sw;
	P.normSq0 = group_sum<64>(tw * t);
	P.x = x; P.y = y; P.a = a;
	P.px0 = (float)(x - a + 2 * (L.seg < a ? L.seg : a));
	st.put_patch64(row, col, (float)(y + i), w, tw);
}
// Patches beyond the reference's 64 taps (a > 7; DepthMap.h:354-358 generalised, see pm_common.h): the (a + 1)^2 taps, in
// the reference's order (rows outer, columns inner, DepthMap.cpp:486-494), are dealt round-robin to the S lanes of a view
// group -- tap k = m * S + seg -- and every view group computes the same weights.  The per-tap tables (tap position,
// weight, centred weighted texel) go to LDS, slot k; slots past the patch repeat its last tap with zero weights.
template <int S, class ST>
__device__ __forceinline__ void fill_patch_big(const EstConst& c, const LaneCtx<S>& L, int x, int y, int a, float center, Patch<S>& P, ST& st) {
	constexpr int MB = (kBigTaps + S - 1) / S;
	const int nside = a + 1, nt = nside * nside;
	gcfptr ref = (gcfptr)c.ref;
	const float sigmaColor = -1.f / (2.f * HC_SQ(0.2f));
	const float sigmaSpatial = -1.f / (2.f * (float)HC_SQ(a));
	float (*bw)[kBigSlots] = st.bw;
	float sa = 0.f, sb = 0.f;
	for (int m = 0; m < MB; ++m) {
		const int k = m * S + L.seg;
		const bool valid = k < nt;
		const int kk = valid ? k : nt - 1;
		const int row = kk / nside, col = kk - row * nside;
		const int i = -a + 2 * row, j = -a + 2 * col;
		const float I = ref[__mul24(y + i, c.W) + (x + j)];
		const float wColor = HC_SQ(I - center) * sigmaColor;
		const float wSpatial = (float)(HC_SQ(j) + HC_SQ(i)) * sigmaSpatial;
		const float we = pm_expf(wColor + wSpatial);
		const float w = valid ? we : 0.f;
		sa = fmaf(I, w, sa);
		sb = sb + w;
		bw[0][k] = (float)(x + j); bw[1][k] = (float)(y + i); bw[2][k] = w; bw[3][k] = I; // I parked until the mean is known
	}
	const float swi = group_sum<S>(sa), sw = group_sum<S>(sb);
	const float tm = swi / sw;
	sa = 0.f;
	for (int m = 0; m < MB; ++m) {
		const int k = m * S + L.seg;
		const float t = bw[3][k] - tm;
		const float tw = bw[2][k] * t;
		sa = fmaf(tw, t, sa);
		bw[3][k] = tw;
	}
	P.sumW = sw;
	P.invSumW = 1.0f / sw;
	P.normSq0 = group_sum<S>(sa);
	P.x = x; P.y = y; P.a = a;
	P.px0 = 0.f;
}
template <int S, bool BIG, class ST>
__device__ __forceinline__ void fill_patch(const EstConst& c, const LaneCtx<S>& L, int x, int y, const PixIn<S>& in, Patch<S>& P, ST& st) {
	const int a = patch_halfwin(c, in.tx);
	if constexpr (BIG) {
		if (a > kHalfWindow) { fill_patch_big<S>(c, L, x, y, a, in.center, P, st); return; }
	}
	fill_patch_64(c, L, x, y, a, in, P, st);
}

// smoothness neighbours (DepthMap.h:376-382 NeighborEstimate): slot k lives in lane k
struct Close {
	float d, n0, n1, n2;     // neighbour estimate as loaded
	float kd, k0, k1, k2;    // interpolated depth + corrected normal (propagation candidates, DepthMap.cpp:1417-1418)
	float X0, X1, X2, conf;
	int nx, ny;
	unsigned long long closeMask, eligMask;
};

struct PixelGeom {
	double X0x, X0y;          // pixel ray (z = 1), Camera.h:299-304
	float v0, v1;             // (float) of it; v2 == 1
	float pn0, pn1, pn2, pd;  // smoothness plane (DepthMap.cpp:1730-1738)
};

// DepthMap.cpp:987-1046 ScorePixel over DepthMap.cpp:522-616 ScorePixelImage, all views at once, for NH
// hypotheses in ONE instruction stream: the NH evaluations are independent, so writing every phase as a loop over
// them lets the scheduler overlap one hypothesis' dependent chains and load latency with the other's arithmetic.
// smoothF: product of the plane-smoothness factors of the hypothesis (DepthMap.cpp:607-615), see smooth_pass().
// homography of one (hypothesis, view) pair (DepthMap.h:565-574), association H = A + Hm (Hr^T n)^T / (n.X0 d)
__device__ __forceinline__ void make_homography(const EstConst& c, const float (&vA)[9], const float (&vHm)[3], float v0, float v1,
                                                float depth, float n0, float n1, float n2, float (&H)[9]) {
	const float nx0 = fmaf(n2, 1.0f, fmaf(n1, v1, n0 * v0));
	const float inv = 1.0f / (nx0 * depth);
	float q[3];
#pragma unroll
	for (int j = 0; j < 3; ++j) q[j] = fmaf(n2, c.Hr[6 + j], fmaf(n1, c.Hr[3 + j], n0 * c.Hr[j])) * inv;
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j) H[i * 3 + j] = fmaf(vHm[i], q[j], vA[i * 3 + j]);
}

// DepthMap.cpp:522-606 ScorePixelImage up to the ZNCC sums, all views at once: every lane warps and samples its taps of
// its own view through H (the homography of the lane's view), the partial sums are combined inside the view group.
// (1) warp every tap, (2) issue all loads, (3) interpolate + accumulate.  The inside-the-image test (Types.h:1633-1635)
// is done on the two end taps of the lane's column (see below); a lane that fails it skips (2) and (3) altogether.
// NR: as in fill_patch_n -- steps >= NR are zero-weight repeats of step NR - 1 in every lane and are skipped; the grouped
// reciprocal still multiplies the repeated denominators, so every remaining tap gets the same bits as with all steps.
// REDUCE = false: the lane's own partial sums are returned (the sweep parks them in LDS and adds them up once per chunk of
// hypotheses, in the butterfly's association: see score_chunk)
template <int S, int NR, class ST, bool REDUCE = true>
__device__ __forceinline__ void score_taps(const EstConst& c, const LaneCtx<S>& L, const Patch<S>& P, const ST& st, const float (&H)[9],
                                           float& sum, float& sumSq, float& num, bool& viewBad) {
	constexpr int MAXM = NR;
	SUBMARK(tap_warp)
	float Ppy[64 / S];
	st.get_py(Ppy);
	float qx[MAXM], qy[MAXM];
	bool bad;
	{
		float Xx[MAXM], Xy[MAXM], Xz[MAXM], iz[MAXM];
		// a segment walks down one patch column: the column term of the warp is hoisted.  Steps past the patch repeat the
		// last row (zero weights), so they change neither the sums nor the inside test.
		const float bx = fmaf(H[0], P.px0, H[2]), by = fmaf(H[3], P.px0, H[5]), bz = fmaf(H[6], P.px0, H[8]);
#pragma unroll
		for (int m = 0; m < MAXM; ++m) {
			const float py = Ppy[m];
			Xx[m] = fmaf(H[1], py, bx); Xy[m] = fmaf(H[4], py, by); Xz[m] = fmaf(H[7], py, bz);
		}
		bool nan = false;
		// perspective divide
		{ // ONE IEEE reciprocal for the (up to) eight steps of the lane, 1/z_i = (1/prod z) * prod_{j!=i} z_j; steps past the patch repeat the last row
			constexpr int last = MAXM - 1;
			const float z4 = Xz[4 < last ? 4 : last], z5 = Xz[5 < last ? 5 : last], z6 = Xz[6 < last ? 6 : last], z7 = Xz[last];
			const float p01 = Xz[0] * Xz[1], p23 = Xz[2] * Xz[3], p45 = z4 * z5, p67 = z6 * z7;
			const float pa = p01 * p23, pb = p45 * p67;
			const float r = 1.0f / (pa * pb);
			nan = nan || !(fabsf(r) < __builtin_huge_valf()); // a zero / non-finite denominator poisons the lane
			const float ra = r * pb, rb = r * pa;
			const float r01 = ra * p23, r23 = ra * p01, r45 = rb * p67, r67 = rb * p45;
			iz[0] = r01 * Xz[1]; iz[1] = r01 * Xz[0]; iz[2] = r23 * Xz[3]; iz[3] = r23 * Xz[2];
			iz[4] = r45 * z5; iz[5] = r45 * z4;
			if constexpr (MAXM > 6) iz[6] = r67 * z7;
			if constexpr (MAXM > 7) iz[7] = r67 * z6;
		}
#pragma unroll
		for (int m = 0; m < MAXM; ++m) { qx[m] = Xx[m] * iz[m]; qy[m] = Xy[m] * iz[m]; }
		// The inside-the-image test (Types.h:1633-1635) on the two END taps of the lane's column only: the column is a straight
		// segment of the reference patch, a homography maps it to a straight segment as long as the depth z keeps its sign along
		// it (z is affine in the step, so the two ends decide that too), and both image coordinates are monotone along such a
		// segment -- if the ends lie inside the (convex) image, so does everything between them.  A column whose ends see z of
		// opposite signs is treated as leaving the image.  Device association; the reference tests every tap.
		constexpr int e = MAXM - 1;
		const bool ends = qx[0] >= 1.f && qy[0] >= 1.f && qx[0] <= L.wmax && qy[0] <= L.hmax &&
		                  qx[e] >= 1.f && qy[e] >= 1.f && qx[e] <= L.wmax && qy[e] <= L.hmax;
		bad = nan || !ends || !(Xz[0] * Xz[e] > 0.f);
	}
	// Only lanes whose taps all lie inside the image (Types.h:1633-1635) sample it: their texel addresses need no clamping, and
	// a lane with a tap outside contributes nothing but the NaN that turns its view's score into thRobust below.
	float a = 0.f, b2 = 0.f, cnum = 0.f;
	SUBMARK(tap_sample)
	if (!bad) {
		const HC_GLOBAL char* imgBase = as_global(c.imgBase);
		float Pw[64 / S], Ptw[64 / S];
#ifdef HCMVS_TAPHALF
		// diagnostic variant (profiles/r04_sweep_levers.md): the rows in two halves -- half the gathered footprints live at a time
		constexpr int HALF = (MAXM + 1) / 2;
		st.get_w(Pw, Ptw);
#pragma unroll
		for (int h0 = 0; h0 < MAXM; h0 += HALF) {
			float2 top[HALF], bot[HALF];
			float fx[HALF], fy[HALF];
#pragma unroll
			for (int u = 0; u < HALF; ++u) {
				const int m = h0 + u < MAXM ? h0 + u : MAXM - 1;
				const int lx = (int)qx[m], ly = (int)qy[m];
				fx[u] = __builtin_amdgcn_fractf(qx[m]);
				fy[u] = __builtin_amdgcn_fractf(qy[m]);
				const unsigned off = L.imgOff + ((unsigned)(__mul24(ly, L.iw) + lx) << 4);
				const f32x4 fp = *(const HC_GLOBAL f32x4*)(imgBase + off);
				top[u] = make_float2(fp.x, fp.y);
				bot[u] = make_float2(fp.z, fp.w);
			}
#pragma unroll
			for (int u = 0; u < HALF; ++u) {
				if (h0 + u >= MAXM) continue;
				const int m = h0 + u;
				const float t = fmaf(fx[u], top[u].y, top[u].x);
				const float b = fmaf(fx[u], bot[u].y, bot[u].x);
				const float val = fmaf(fy[u], b - t, t);
				const float vw = val * Pw[m];
				a = a + vw;
				b2 = fmaf(val, vw, b2);
				cnum = fmaf(val, Ptw[m], cnum);
			}
			asm volatile("" ::: "memory"); // the second half's gathers are not hoisted above the first half's arithmetic
		}
#else
		float2 top[MAXM], bot[MAXM];
		float fx[MAXM], fy[MAXM];
#pragma unroll
		for (int m = 0; m < MAXM; ++m) {
			const int lx = (int)qx[m], ly = (int)qy[m];
			fx[m] = __builtin_amdgcn_fractf(qx[m]);
			fy[m] = __builtin_amdgcn_fractf(qy[m]);
			// the source views are held as 2 x 2 footprints (quad_kernel): ONE 16-byte gather per bilinear sample
			const unsigned off = L.imgOff + ((unsigned)(__mul24(ly, L.iw) + lx) << 4);
#if defined(HCMVS_ABL) && HCMVS_ABL == 1 /* diagnostic ablation: no gather loads (results are wrong) */
			const float fake = (float)(off & 255u) * (1.f / 255.f);
			top[m] = make_float2(fake, fake * -0.1f);
			bot[m] = make_float2(fake * 0.8f, fake * -0.1f);
#else
			const f32x4 fp = *(const HC_GLOBAL f32x4*)(imgBase + off); // (I00, I10 - I00, I01, I11 - I01), see quad_kernel
			top[m] = make_float2(fp.x, fp.y);
			bot[m] = make_float2(fp.z, fp.w);
#endif
		}
		st.get_w(Pw, Ptw);
#pragma unroll
		for (int m = 0; m < MAXM; ++m) {
			// bilinear sample (Types.inl:2250-2258) in lerp form; the horizontal differences come ready from the footprint layout
			const float t = fmaf(fx[m], top[m].y, top[m].x);
			const float b = fmaf(fx[m], bot[m].y, bot[m].x);
			const float val = fmaf(fy[m], b - t, t);
			const float vw = val * Pw[m];
			a = a + vw;
			b2 = fmaf(val, vw, b2);
			cnum = fmaf(val, Ptw[m], cnum);
		}
#endif
	}
	// a tap outside the image (or a degenerate warp) must turn the view's score into thRobust: the lane poisons its
	// partial sum, the NaN survives the butterfly and fails the `nrmSq > 0` test of view_score
	viewBad = false;
	SUBMARK(tap_reduce)
	if constexpr (REDUCE) { sum = group_sum<S>(bad ? __builtin_nanf("") : a); sumSq = group_sum<S>(b2); num = group_sum<S>(cnum); }
	else { sum = bad ? __builtin_nanf("") : a; sumSq = b2; num = cnum; }
	SUBMARK(tap_end)
}

// score_taps for the patches of fill_patch_big: tap k = m * S + seg from the LDS tables, one IEEE reciprocal per tap, four
// taps in flight at a time; same poisoning of the partial sum when a tap leaves the image
template <int S, class ST, bool REDUCE = true>
__device__ __forceinline__ void score_taps_big(const EstConst& c, const LaneCtx<S>& L, const ST& st, const float (&H)[9], float& sum, float& sumSq,
                                               float& num, bool& viewBad) {
	constexpr int MB = (kBigTaps + S - 1) / S, CH = 4;
	const float (*bw)[kBigSlots] = st.bw;
	const HC_GLOBAL char* imgBase = as_global(c.imgBase);
	float a = 0.f, b2 = 0.f, cnum = 0.f;
	bool bad = false;
	for (int m0 = 0; m0 < MB; m0 += CH) {
		float fx[CH], fy[CH];
		unsigned off[CH];
#pragma unroll
		for (int u = 0; u < CH; ++u) {
			const int m = m0 + u < MB ? m0 + u : MB - 1; // MB % CH != 0: the spare steps repeat the last one and are not accumulated
			const int k = m * S + L.seg;
			const float px = bw[0][k], py = bw[1][k];
			const float Xx = fmaf(H[1], py, fmaf(H[0], px, H[2]));
			const float Xy = fmaf(H[4], py, fmaf(H[3], px, H[5]));
			const float Xz = fmaf(H[7], py, fmaf(H[6], px, H[8]));
			const float iz = 1.0f / Xz;
			const float qx = Xx * iz, qy = Xy * iz;
			bad = bad || !(qx >= 1.f && qy >= 1.f && qx <= L.wmax && qy <= L.hmax); // Types.h:1633-1635; a NaN fails it
			const int lx = (int)__builtin_amdgcn_fmed3f(qx, 0.f, L.wmax), ly = (int)__builtin_amdgcn_fmed3f(qy, 0.f, L.hmax);
			fx[u] = __builtin_amdgcn_fractf(qx);
			fy[u] = __builtin_amdgcn_fractf(qy);
			off[u] = L.imgOff + ((unsigned)(__mul24(ly, L.iw) + lx) << 4);
		}
		f32x2 tv[CH], bv[CH];
#pragma unroll
		for (int u = 0; u < CH; ++u) {
			const f32x4 fp = *(const HC_GLOBAL f32x4*)(imgBase + off[u]);
			tv[u].x = fp.x; tv[u].y = fp.y; bv[u].x = fp.z; bv[u].y = fp.w;
		}
#pragma unroll
		for (int u = 0; u < CH; ++u) {
			if (m0 + u >= MB) continue;
			const int k = (m0 + u) * S + L.seg;
			const float t = fmaf(fx[u], tv[u].y, tv[u].x);
			const float b = fmaf(fx[u], bv[u].y, bv[u].x);
			const float val = fmaf(fy[u], b - t, t);
			const float vw = val * bw[2][k];
			a = a + vw;
			b2 = fmaf(val, vw, b2);
			cnum = fmaf(val, bw[3][k], cnum);
		}
	}
	viewBad = false;
	if constexpr (REDUCE) { sum = group_sum<S>(bad ? __builtin_nanf("") : a); sumSq = group_sum<S>(b2); num = group_sum<S>(cnum); }
	else { sum = bad ? __builtin_nanf("") : a; sumSq = b2; num = cnum; }
}

// DepthMap.cpp:597-615, 890-893: score of one view from its ZNCC sums, times the smoothness factor of the hypothesis
__device__ __forceinline__ float view_score(const EstConst& c, float sum, float sumSq, float num, bool viewBad, float invSumW,
                                            float normSq0, float smoothF) {
	const float normSq1 = sumSq - HC_SQ(sum) * invSumW;
	const float nrmSq = normSq0 * normSq1;
	float ncc = num / sqrtf(nrmSq);
	ncc = ncc < -1.f ? -1.f : (ncc > 1.f ? 1.f : ncc);
	float s = (1.f - ncc) * smoothF;
	s = c.pfScale * s;
	if (viewBad || !(nrmSq > 0.f)) s = c.thRobust;
	return s;
}
// DepthMap.cpp:987-1046 ScorePixel: mean of the two best views (m1 <= m2 are the two smallest view scores)
__device__ __forceinline__ float two_best(const EstConst& c, float m1, float m2) {
	return c.V <= 1 ? m1 : (m2 >= c.thRobust ? m1 : (m1 + m2) / 2.f);
}

// the two smallest of {a1 <= a2} and {b1 <= b2}
__device__ __forceinline__ void min2_merge(float& a1, float& a2, float b1, float b2) {
	const float lo = fminf(a1, b1), hi = fmaxf(a1, b1);
	a2 = fminf(hi, fminf(a2, b2));
	a1 = lo;
}

// one hypothesis, everything in one go (init-score pass): the two best view scores of the views of L's set are merged into
// (r1, r2); two_best() of those is the pixel's score (an estimate with 9..16 source views runs a second set of eight)
template <int S, bool BIG, class ST>
__device__ __forceinline__ void score_pixel(const EstConst& c, const LaneCtx<S>& L, const Patch<S>& P, const ST& st, float v0, float v1,
                                            float smoothF, float depth, float n0, float n1, float n2, float& r1, float& r2) {
	float vA[9], vHm[3], H[9];
	st.get_view(vA, vHm);
	make_homography(c, vA, vHm, v0, v1, depth, n0, n1, n2, H);
	float sum, sumSq, num;
	bool viewBad;
	if (BIG && P.a > kHalfWindow) {
		score_taps_big<S>(c, L, st, H, sum, sumSq, num, viewBad);
	} else {
		if (P.a == 6) score_taps<S, 7>(c, L, P, st, H, sum, sumSq, num, viewBad);
		else if (P.a == 5) score_taps<S, 6>(c, L, P, st, H, sum, sumSq, num, viewBad);
		else score_taps<S, 8>(c, L, P, st, H, sum, sumSq, num, viewBad);
	}
	const float s = view_score(c, sum, sumSq, num, viewBad, P.invSumW, P.normSq0, smoothF);
	float m1 = L.vact ? s : __builtin_huge_valf(), m2 = __builtin_huge_valf();
	min2_across<S>(m1, m2);
	min2_merge(r1, r2, m1, m2);
}

// Score one chunk of up to eight hypotheses of a round (bits of `todo`, all in [base, base + 8); lane t holds hypothesis t
// in hd/h0/h1/h2; F: smoothness factors in the layout of smooth_pass, hypothesis base + g in lanes 8g..8g+7).
//   (1) the homographies of all (hypothesis, view) pairs of the chunk, one pair per lane -> LDS
//   (2) per hypothesis: the tap sums of all views (score_taps), one lane per view parks them in LDS
//   (3) the per-view scores and the two-best-views means of the whole chunk, one (hypothesis, view) pair per lane
// (1) and (3) cost one instruction stream per chunk instead of one per hypothesis.  Lane t gets the two best view scores of
// hypothesis t merged into (r1, r2); two_best() of those is the hypothesis' score.  L selects the view set (views
// L.view - L.vloc ... + 7): an estimate with 9..16 source views calls this twice per chunk.
template <int S, bool BIG, bool PACK = false>
__device__ __forceinline__ void score_chunk(const EstConst& c, const LaneCtx<S>& L, const Patch<S>& P, const LdsStore<S>& st, float v0, float v1,
                                            float F, float hd, float h0, float h1, float h2, unsigned long long todoIn, int baseIn,
                                            int fallbackIn, float& r1, float& r2, unsigned& issued) {
	// the list of hypotheses is the same in every lane: keep it (and the loop over it) on the scalar unit
	const unsigned long long todo = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(todoIn >> 32)) << 32) |
	                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)todoIn);
	const int base = __builtin_amdgcn_readfirstlane(baseIn), fallback = __builtin_amdgcn_readfirstlane(fallbackIn);
	constexpr int NV = 64 / S;               // views (lane groups) of a wave
	constexpr int HP = 8;                    // hypotheses per pair-pass (lane = g * NV + v)
	WavePark<S>* pk = st.pk;
	SUBMARK(sc_hom)
	const int lane = L.lane, pv = lane % NV, pg = lane / NV;
	const int vbase = __builtin_amdgcn_readfirstlane(L.view - L.vloc); // first view of the set
	float vA[9], vHm[3];
	{
		const float4 a = pk->vh[vbase + pv][0], b = pk->vh[vbase + pv][1], cc = pk->vh[vbase + pv][2];
		vA[0] = a.x; vA[1] = a.y; vA[2] = a.z; vA[3] = a.w; vA[4] = b.x; vA[5] = b.y; vA[6] = b.z; vA[7] = b.w; vA[8] = cc.x;
		vHm[0] = cc.y; vHm[1] = cc.z; vHm[2] = cc.w;
	}
#pragma unroll
	for (int p0 = 0; p0 < 8; p0 += HP) {
		if ((todo >> (base + p0)) == 0ull) break;
		const int g = p0 + (pg < HP ? pg : 0);
		const int src = ((todo >> (base + g)) & 1ull) ? base + g : fallback;
		const float gd = __shfl(hd, src, 64), g0 = __shfl(h0, src, 64), g1 = __shfl(h1, src, 64), g2 = __shfl(h2, src, 64);
		if (pg < HP) {
			float H[9];
			make_homography(c, vA, vHm, v0, v1, gd, g0, g1, g2, H);
			pk->hl[g][pv][0] = make_float4(H[0], H[1], H[2], H[3]);
			pk->hl[g][pv][1] = make_float4(H[4], H[5], H[6], H[7]);
			pk->hl[g][pv][2] = make_float4(H[8], 0.f, 0.f, 0.f);
		}
	}
	// views of this set that exist: with fewer than eight, the idle view groups take (hypothesis, view) pairs of their own --
	// pair p = 8 * pass + group -> hypothesis p / nAct, view p % nAct -- so a chunk costs ceil(n * nAct / 8) tap passes instead
	// of n.  Every pair is evaluated exactly as in the plain loop (same lanes-per-view layout), only by another group.
	const int nAct = __builtin_amdgcn_readfirstlane(c.V - vbase < NV ? c.V - vbase : NV);
	SUBMARK(sc_taps)
	auto taps_of = [&](auto nr) {
		constexpr int NR = decltype(nr)::value;
		if constexpr (PACK) { // instantiated for view counts that leave groups idle (not 7, 8, 15 or 16 views)
			if (nAct < NV) {
				uint32_t slots = 0; // nibble j: slot (hypothesis index - base) of the j-th hypothesis of the chunk
				int n = 0;
				for (unsigned long long td = todo; td; td &= td - 1ull, ++n) slots |= (uint32_t)(__builtin_ctzll(td) - base) << (4 * n);
				const int total = n * nAct;
				const uint32_t inv = (65536u + (uint32_t)nAct - 1u) / (uint32_t)nAct; // p / nAct == (p * inv) >> 16 for p < 128
				for (int k0 = 0; k0 < total; k0 += NV) {
					const int p = k0 + L.vloc;
					const bool valid = p < total;
					const int pp = valid ? p : 0;
					const int j = (int)(((uint32_t)pp * inv) >> 16), v = pp - j * nAct;
					const int g = (int)((slots >> (4 * j)) & 15u);
					LaneCtx<S> Lp = L; // the lane works for view vbase + v in this pass: that view's image constants
					Lp.imgOff = (unsigned)__shfl((int)L.imgOff, v * S, 64); Lp.iw = __shfl(L.iw, v * S, 64);
					Lp.wmax = __shfl(L.wmax, v * S, 64); Lp.hmax = __shfl(L.hmax, v * S, 64);
					float H[9];
					{
						const float4 a = pk->hl[g][v][0], b = pk->hl[g][v][1], cc = pk->hl[g][v][2];
						H[0] = a.x; H[1] = a.y; H[2] = a.z; H[3] = a.w; H[4] = b.x; H[5] = b.y; H[6] = b.z; H[7] = b.w; H[8] = cc.x;
					}
					float sum, sumSq, num;
					bool viewBad;
					if constexpr (NR == 0) score_taps_big<S, LdsStore<S>, false>(c, Lp, st, H, sum, sumSq, num, viewBad);
					else score_taps<S, NR, LdsStore<S>, false>(c, Lp, P, st, H, sum, sumSq, num, viewBad);
					park_partials<S>(pk, g, v, L.seg, valid, sum, sumSq, num);
				}
				issued += (unsigned)n;
				return;
			}
		}
		for (unsigned long long td = todo; td; td &= td - 1ull) {
			const int g = __builtin_ctzll(td) - base;
			float H[9];
			{
				const float4 a = pk->hl[g][L.vloc][0], b = pk->hl[g][L.vloc][1], cc = pk->hl[g][L.vloc][2];
				H[0] = a.x; H[1] = a.y; H[2] = a.z; H[3] = a.w; H[4] = b.x; H[5] = b.y; H[6] = b.z; H[7] = b.w; H[8] = cc.x;
			}
			float sum, sumSq, num;
			bool viewBad;
			if constexpr (NR == 0) score_taps_big<S, LdsStore<S>, false>(c, L, st, H, sum, sumSq, num, viewBad);
			else score_taps<S, NR, LdsStore<S>, false>(c, L, P, st, H, sum, sumSq, num, viewBad);
			++issued;
			park_partials<S>(pk, g, L.vloc, L.seg, true, sum, sumSq, num);
		}
	};
	if (BIG && P.a > kHalfWindow) {
		taps_of(std::integral_constant<int, 0>()); // the big-patch scorer
	} else {
		if (P.a == 6) taps_of(std::integral_constant<int, 7>());
		else if (P.a == 5) taps_of(std::integral_constant<int, 6>());
		else taps_of(std::integral_constant<int, 8>());
	}
	SUBMARK(sc_epi)
#pragma unroll
	for (int p0 = 0; p0 < 8; p0 += HP) {
		if ((todo >> (base + p0)) == 0ull) break;
		const int g = p0 + (pg < HP ? pg : 0);
		// the sums over the S lanes of the pair's view group, in the association of the xor butterfly (group_sum):
		// ((l0 + l1) + (l2 + l3)) + ((l4 + l5) + (l6 + l7)) -- the same bits, one instruction stream per chunk instead of per evaluation
		float rs[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			static_assert(S == 8, "tree below is written for 8 lanes per view group");
#ifdef HCMVS_PARTHALF
			const float4 a = *(const float4*)&pk->part[k][g][pv][0]; // (l0 + l1, l2 + l3, l4 + l5, l6 + l7)
			rs[k] = (a.x + a.y) + (a.z + a.w);
#else
			const float4* q = (const float4*)&pk->part[k][g][pv][0];
			const float4 a = q[0], b = q[1];
			rs[k] = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
#endif
		}
		const float Fg = __shfl(F, g * 8, 64);
		const float s = view_score(c, rs[0], rs[1], rs[2], false, P.invSumW, P.normSq0, Fg);
		float m1 = vbase + pv < c.V ? s : __builtin_huge_valf(), m2 = __builtin_huge_valf();
		min2_group<NV>(m1, m2);
		// hypothesis base + p0 + k sits in lanes k * NV ... of (m1, m2)
		const int k = lane - base - p0;
		const int from = (k >= 0 && k < HP ? k : 0) * NV;
		const float g1 = __shfl(m1, from, 64), g2 = __shfl(m2, from, 64);
		if (k >= 0 && k < HP && ((todo >> lane) & 1ull)) min2_merge(r1, r2, g1, g2);
	}
	SUBMARK(sc_end)
}

// Util.inl:614-626
__device__ __forceinline__ void dir2normal(float p0, float p1, float& n0, float& n1, float& n2) {
	float s0, c0, s1, c1;
	pm_sincosf(p0, &s0, &c0);
	pm_sincosf(p1, &s1, &c1);
	n0 = c0 * s1; n1 = s0 * s1; n2 = c1;
}
// DepthMap.h:618-626
__device__ __forceinline__ float random_depth(const EstConst& c, float u) {
	const float r = c.dMinSqr + (c.dMaxSqr - c.dMinSqr) * u;
	return r * r;
}
__device__ __forceinline__ void random_normal(const PixelGeom& G, float u1, float u2, float& n0, float& n1, float& n2) {
	const float p0 = fd2r(0.f) + (fd2r(180.f) - fd2r(0.f)) * u1;
	const float p1 = fd2r(90.f) + (fd2r(180.f) - fd2r(90.f)) * u2;
	dir2normal(p0, p1, n0, n1, n2);
	if (dot3(n0, n1, n2, G.v0, G.v1, 1.f) > 0.f) { n0 = -n0; n1 = -n1; n2 = -n2; }
}
// DepthMap.h:629-634 CorrectNormal + Rotation.inl:707-733 (Rodrigues).  The rotation runs only for a normal that faces away from the
// camera -- a wave sees one in well under 1 % of the pixels and branches around the block otherwise (s_cbranch_execz).  Tried in
// round 4 as a real function (noinline): the call needs a stack (80-144 bytes of scratch per lane) and the caller-saved registers
// around it spill 4-15 VGPRs in the two-wave / TWO / PACK instances (profiles/r04_sweep_levers.md), so it stays inline
__device__ __forceinline__ void correct_normal_rotate(float v0, float v1, float cosAngLen, float& n0, float& n1, float& n2) {
	const float v2 = 1.f;
	const float a0 = n1 * v2 - n2 * v1, a1 = n2 * v0 - n0 * v2, a2 = n0 * v1 - n1 * v0;
	const float vlen = sqrtf(dot3(v0, v1, v2, v0, v1, v2));
	float phi = (pm_acosf(cosAngLen / vlen) - fd2r(90.f)) * 1.01f;
	if (!(phi < -0.001f)) phi = -0.001f;
	const float wnorm = sqrtf(dot3(a0, a1, a2, a0, a1, a2));
	if (!(wnorm >= FLT_EPSILON)) return;
	const float iw = 1.f / wnorm;
	const float w0 = a0 * iw, w1 = a1 * iw, w2 = a2 * iw;
	const float O[9] = {0.f, -w2, w1, w2, 0.f, -w0, -w1, w0, 0.f};
	float sp, cp;
	pm_sincosf(phi, &sp, &cp);
	const float cp1 = 1.f - cp;
	float R[9];
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j) {
			float s = 0.f;
#pragma unroll
			for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
			R[i * 3 + j] = ((i == j ? 1.f : 0.f) + O[i * 3 + j] * sp) + s * cp1;
		}
	const float r0 = R[0] * n0 + R[1] * n1 + R[2] * n2, r1 = R[3] * n0 + R[4] * n1 + R[5] * n2,
	            r2 = R[6] * n0 + R[7] * n1 + R[8] * n2;
	n0 = r0; n1 = r1; n2 = r2;
}
__device__ __forceinline__ void correct_normal(const PixelGeom& G, float& n0, float& n1, float& n2) {
	const float cosAngLen = dot3(n0, n1, n2, G.v0, G.v1, 1.f);
	if (cosAngLen >= 0.f) correct_normal_rotate(G.v0, G.v1, cosAngLen, n0, n1, n2);
}
// DepthMap.cpp:1671-1726 InterpolatePixel (ray-plane form)
__device__ __forceinline__ float interpolate_pixel(const EstConst& c, const PixelGeom& G, int nx, int ny, float depth,
                                                   float n0, float n1, float n2) {
	const double p0 = n0, p1 = n1, p2 = n2, z = depth;
	const double P0 = ((double)nx - c.cx) * z * c.ifx, P1 = ((double)ny - c.cy) * z * c.ify;
	const double planeD = p0 * P0 + p1 * P1 + p2 * z;
	const float dn = (float)(planeD / (p0 * G.X0x + p1 * G.X0y + p2 * 1.0));
	return (c.dMin <= dn && dn < c.dMax) ? dn : depth;
}
__device__ __forceinline__ void init_plane(PixelGeom& G, float depth, float n0, float n1, float n2) {
	G.pn0 = n0; G.pn1 = n1; G.pn2 = n2;
	G.pd = -depth * dot3(n0, n1, n2, G.v0, G.v1, 1.f);
}
__device__ __forceinline__ void pixel_geom(const EstConst& c, int x, int y, PixelGeom& G) {
	G.X0x = ((double)x - c.cx) * c.ifx;
	G.X0y = ((double)y - c.cy) * c.ify;
	G.v0 = (float)G.X0x; G.v1 = (float)G.X0y;
	G.pn0 = G.pn1 = G.pn2 = G.pd = 0.f;
}

// Plane-smoothness factors (DepthMap.cpp:607-615) of up to EIGHT hypotheses in one lane-parallel pass:
// lane = g*8 + j handles hypothesis g and neighbour slot (chunk*8 + j).  The per-slot factors
// (1 - bD e^{sD (dist/d)^2}) (1 - bN e^{sN acos^2}) are multiplied by a butterfly inside each 8-lane group (slots
// without a neighbour contribute exactly 1), chunk after chunk.  hd..hpd: the lane's own group's hypothesis
// (depth, normal, plane normal, plane offset); limit: last slot whose corrected normal is already in effect.
__device__ __forceinline__ float smooth_pass(const EstConst& c, const float (*cl)[kMaxSlots], unsigned long long closeMask,
                                             unsigned long long eligMask, int lane, float hd, float h0, float h1, float h2,
                                             float hp0, float hp1, float hp2, float hpd, int limit) {
	float F = 1.f;
	if (closeMask == 0ull) return F;
	const int nChunks = (64 - __builtin_clzll(closeMask) + 7) >> 3;
	const int j = lane & 7;
	for (int ch = 0; ch < nChunks; ++ch) {
		SUBMARK(blk_smooth_chunk)
		const int slot = ch * 8 + j;
		const float X0 = cl[0][slot], X1 = cl[1][slot], X2 = cl[2][slot];
		const float o0 = cl[3][slot], o1 = cl[4][slot], o2 = cl[5][slot];
		const float k0 = cl[6][slot], k1 = cl[7][slot], k2 = cl[8][slot];
		const bool valid = (closeMask >> slot) & 1ull;
		const bool corr = ((eligMask >> slot) & 1ull) && slot <= limit;
		const float c0 = corr ? k0 : o0, c1 = corr ? k1 : o1, c2 = corr ? k2 : o2;
		const float dist = dot3(hp0, hp1, hp2, X0, X1, X2) + hpd; // Planef::Distance
		const float fd = pm_expf(HC_SQ(dist / hd) * c.smoothSigmaDepth);
		float ca = dot3(h0, h1, h2, c0, c1, c2); // Util.inl:417-420 for unit normals (every normal on this path is unit)
		ca = ca < -1.f ? -1.f : (ca > 1.f ? 1.f : ca);
		const float ang = pm_acosf(ca);
		const float fn = pm_expf(HC_SQ(ang) * c.smoothSigmaNormal);
		float f = (1.f - c.smoothBonusDepth * fd) * (1.f - c.smoothBonusNormal * fn);
		f = valid ? f : 1.f;
		f = f * lane_xor<1>(f);
		f = f * lane_xor<2>(f);
		f = f * lane_xor<4>(f);
		F = ch == 0 ? f : F * f;
	}
	SUBMARK(blk_smooth_end)
	return F;
}

// ------------------------------------------------------------------------------------------------------
// sweep: NW waves cooperate on one image row.  Every wave carries the complete (identical) pixel state;
// in each round wave w scores hypothesis number w of the batch and the scores are exchanged through LDS,
// after which every wave replays the reference's sequential accept logic (DepthMap.cpp:1425, 1455, 1484).
// Propagation candidates and full-random hypotheses do not depend on earlier accepts, so a batch is
// exact; refinement trials do (they perturb the current estimate), so the trials after an accepted one
// are discarded and re-issued from the new state.  The maps are identical to the sequential sweep.

constexpr int kHist = 16; // ring of the row's own latest results (neighbours behind in the same row)

template <int NW>
struct RowShared {
	float sc[2][32];   // per round parity: score of hypothesis t (propagation candidate or trial number)
	float hist[kHist][6];
	int row, item;
};

// started: the value the awaited word reaches once its owner is AT WORK on what I wait for (a row stores its sweep tag in its progress
// word when it begins; the rows-done counter of an image is two rows short of the end of its sweep).  Below it the waiter sleeps long
// between polls (thousands of row workers wait for their turn while a wavefront builds up; polling every 64 cycles
// they take issue slots from the rows that are running on the same SIMDs and load the L2 for nothing) -- only the next in line polls fast.
__device__ __forceinline__ int wait_progress(HC_GLOBAL int32_t* word, int need, HC_GLOBAL int32_t* err, int started = -0x7fffffff) {
#if defined(HCMVS_ABL) && HCMVS_ABL == 3 /* diagnostic ablation: rows do not wait for each other (results are wrong) */
	return need;
#endif
	int v;
	unsigned spins = 0;
	unsigned long long t0 = 0;
	while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, HC_SCOPE)) < need) {
		if (v < started) __builtin_amdgcn_s_sleep(32); // ~2000 cycles
		else __builtin_amdgcn_s_sleep(1);
		if ((++spins & 127u) == 0u) {
			if (__hip_atomic_load(err, __ATOMIC_RELAXED, HC_SCOPE) != 0) return -1;
			// bounded in TIME (the 100 MHz constant clock), not in polls: give up after 20 s instead of hanging the device -- long enough
			// for a device shared with other processes, whose kernels may hold the chip for whole sweeps
			const unsigned long long now = __builtin_amdgcn_s_memrealtime();
			if (t0 == 0) t0 = now;
			else if (now - t0 > 2000000000ull) {
				__hip_atomic_store(err, 1, __ATOMIC_RELAXED, HC_SCOPE);
				return -1;
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler-only: keep payload loads below the poll
	return v;
}

// state of one row worker that lives across pixels
template <int S>
struct RowPipe {
	uint8_t tx1;      // gradient-map byte of the next column (kept raw so that nothing waits on the load)
	uint8_t tx2;      // ... and of the one after it
	float nI, nC;     // reference-image tap / centre of the NEXT pixel, loaded a whole pixel ahead
	int known;        // columns the previous logical row is known to have finished
	int poll;         // progress value of an in-flight poll
	int pendingPub;   // > 0: results up to this column are stored but not yet published
	HC_GLOBAL int32_t *upWord, *myWord, *err;
	int base;         // progress words count columns of sweep k of the launch as (k << 16) + columns: the words are not reset between the sweeps
	int r, y;
	bool rev, fail;
};

// neighbour slot of this lane for pixel (x,y), logical column q (DepthMap.cpp:1064-1391)
template <int S>
__device__ __forceinline__ void slot_setup(const EstConst& c, int lane, int x, int y, bool rev, PixIn<S>& in) {
	const int W = c.W, H = c.H;
	int nx = x, ny = y;
	bool slot = false, sprop = false;
	if (c.itExternal >= 1) {
		// cross pattern, the same set for both sweep directions
		int phw = in.tx > 150.f ? 5 : c.propHalfwin;
		if (phw > 7) phw = 7;
		const int step = c.propStep > 0 ? c.propStep : 1;
		int i = 0;
		if (x > phw && y > phw && x < W - phw && y < H - phw) {
			const int ni = phw >= 1 ? (phw - 1) / step + 1 : 0;
			if (lane < 4 * ni) { i = 1 + (lane >> 2) * step; slot = true; }
		} else if (x > c.border && y > c.border && x < W - c.border && y < H - c.border) {
			if (lane < 4) { i = 1; slot = true; }
		}
		const int t = lane & 3;
		nx = x + (t == 2 ? -i : (t == 3 ? i : 0));
		ny = y + (t == 0 ? -i : (t == 1 ? i : 0));
		sprop = slot;
	} else {
		// 2 causal neighbours propagate, all 4 smooth
		if (lane < 4) {
			const int d = rev ? ((lane + 2) & 3) : lane; // 0 left, 1 up, 2 right, 3 down
			nx = x + (d == 0 ? -1 : (d == 2 ? 1 : 0));
			ny = y + (d == 1 ? -1 : (d == 3 ? 1 : 0));
			slot = d == 0 ? x > c.border : (d == 1 ? y > c.border : (d == 2 ? x < W - c.border : y < H - c.border));
			sprop = lane < 2;
		}
	}
	in.nx = nx; in.ny = ny; in.slot = slot; in.sprop = sprop;
	in.back = ny == y ? (rev ? nx - x : x - nx) : 0;
	in.isUp = slot && (rev ? ny > y : ny < y);
	in.loaded = false;
	in.ndn = make_float4(0.f, 0.f, 0.f, 0.f);
	in.nconf = 0.f;
}

// issue the loads of pixel (x,y) that do not depend on other rows' progress in this sweep
template <int S>
__device__ __forceinline__ void prefetch_static(const EstConst& c, const LaneCtx<S>& L, int x, int y, int q, bool rev, bool upKnown, PixIn<S>& in) {
	slot_setup<S>(c, L.lane, x, y, rev, in);
	const int idx = y * c.W + x;
	in.cur = load_dn(&c.dn[idx]);
	in.curConf = load_f(&c.conf[idx]);
	// slots ahead of me in my row, in later rows, or left of the swept range still hold last pass's values.  The load is
	// unconditional (lanes without such a slot read the pixel itself and never look at it): behind a branch the number of
	// loads in flight is unknown to the compiler, and the first `s_waitcnt` after it -- the hand-over of the patch inputs
	// loaded a pixel ago -- would be set for the shorter path, i.e. wait for the loads issued just above.
	// upKnown: the row above is already known to have passed this column, so its slots go out in the same batch (a second
	// load into the same registers would have to wait for the first)
	const bool want = in.slot && (!in.isUp || upKnown) && !(in.back > 0 && in.back <= q);
	const int nidx = want ? in.ny * c.W + in.nx : idx;
	in.ndn = load_dn(&c.dn[nidx]);
	in.nconf = load_f(&c.conf[nidx]);
	in.loaded = want;
}
template <int S>
__device__ __forceinline__ void prefetch_up(const EstConst& c, PixIn<S>& in) {
	if (in.isUp && !in.loaded) {
		in.ndn = load_dn(&c.dn[in.ny * c.W + in.nx]);
		in.nconf = load_f(&c.conf[in.ny * c.W + in.nx]);
		in.loaded = true;
	}
}

// scores of hypotheses [0, n) held in lane t; each wave evaluated [lo, hi) -> everybody gets all of them
template <int NW>
__device__ __forceinline__ float share_scores(RowShared<NW>& sh, int& par, int lane, int lo, int hi, float mine) {
	if (NW == 1) return mine;
	if (lane >= lo && lane < hi) sh.sc[par][lane] = mine;
	__syncthreads();
	const float all = sh.sc[par][lane & 31];
	par ^= 1;
	return all;
}

// DepthMap.cpp:1050-1501 ProcessPixel for logical column q of the row, pixel (x,y).
// Per phase the hypotheses are generated lane-parallel (lane t = hypothesis t) by every wave, the smoothness
// factors of a wave's share come from one smooth_pass, the share is scored hypothesis by hypothesis, the scores are
// exchanged through LDS and every wave replays the reference's sequential accept logic (DepthMap.cpp:1425, 1455,
// 1484).  Refinement trials depend on earlier accepts: after an accepted trial the later ones are regenerated.
// TWO: the estimate has 9..16 source views: a second set of eight view groups (L1: views 8..15 in the same lane layout) is
// scored after the first, and the two best views are taken over both
// HINT: the launch is the sweep in which the `restore` variant offers the up-sampled coarser level as one more hypothesis
// (EstConst::hintDepth / hintIter); every other launch -- all of BASELINE configs[1] -- runs the instance without that code
template <int S, int NW, bool BIG, bool TWO, bool PACK, bool HINT>
__device__ __forceinline__ void process_pixel(const EstConst& c, const LaneCtx<S>& L, const LaneCtx<S>& L1, RowShared<NW>& sh, int& par, int wv,
                                              int x, int y, int q, int iter, const PixIn<S>& in, const Patch<S>& P,
                                              const LdsStore<S>& st, RowPipe<S>& pp, unsigned& evals, unsigned& issued STAMP_ARGS) {
	const int W = c.W, lane = L.lane;
	PixelGeom G;
	pixel_geom(c, x, y, G);

	// neighbour slots: slot k in lane k
	Close C;
	const int nx = in.nx, ny = in.ny;
	C.d = in.ndn.x; C.n0 = in.ndn.y; C.n1 = in.ndn.z; C.n2 = in.ndn.w; C.conf = in.nconf; C.nx = nx; C.ny = ny;
	if (in.slot && in.back > 0 && in.back <= q) { // finished by this row earlier in this sweep: the row ring has it
		const float* hrec = sh.hist[(q - in.back) & (kHist - 1)];
		C.d = hrec[0]; C.n0 = hrec[1]; C.n1 = hrec[2]; C.n2 = hrec[3]; C.conf = hrec[4];
	}
	C.kd = C.d; C.k0 = C.n0; C.k1 = C.n1; C.k2 = C.n2;
	const bool closeV = in.slot && C.d > 0.f;
	C.closeMask = __ballot(closeV);
	{ // Cast<float>(camera.TransformPointI2C(Point3(nx, ndepth))), Camera.h:306-312
		const double z = C.d;
		C.X0 = (float)(((double)nx - c.cx) * z * c.ifx);
		C.X1 = (float)(((double)ny - c.cy) * z * c.ify);
		C.X2 = (float)z;
	}
	// propagation candidates (DepthMap.cpp:1412-1418), lane-parallel: ray-plane depth + corrected normal
	const bool elig = closeV && in.sprop && !(C.conf >= c.thKeep);
	C.eligMask = __ballot(elig);
	if (elig) {
		C.kd = interpolate_pixel(c, G, nx, ny, C.d, C.n0, C.n1, C.n2);
		correct_normal(G, C.k0, C.k1, C.k2);
	}
	const unsigned long long closeMask = C.closeMask, eligMask = C.eligMask;
	{ // park the slots for the smoothness passes (LDS of this wave; same-wave LDS accesses are ordered)
		float (*cl)[kMaxSlots] = st.pk->cl;
		if (lane < kMaxSlots) {
			cl[0][lane] = C.X0; cl[1][lane] = C.X1; cl[2][lane] = C.X2;
			cl[3][lane] = C.n0; cl[4][lane] = C.n1; cl[5][lane] = C.n2;
			cl[6][lane] = C.k0; cl[7][lane] = C.k1; cl[8][lane] = C.k2;
		}
	}

	const int idx = y * W + x;
	float conf = in.curConf;
	float depth = in.cur.x, n0 = in.cur.y, n1 = in.cur.z, n2 = in.cur.w;
	init_plane(G, depth, n0, n1, n2);
	STAMP(2)

	// ---- one state machine for the three kinds of hypothesis rounds (a single inlined scorer keeps the code small) ----
	//   PROP   propagation candidates, DepthMap.cpp:1406-1440: candidate i (slot order) sees slots <= its own corrected
	//   RAND   completely random hypotheses, DepthMap.cpp:1448-1465 (independent of the estimate: a batch is exact)
	//   REFINE perturbations of the current estimate, DepthMap.cpp:1466-1501 (after an accept the later trials are redone)
	enum { PH_PROP, PH_PICK, PH_RAND, PH_REFINE, PH_DONE };
	const int nc = __builtin_popcountll(eligMask);
	int candSlot = 0; // lane i <- slot of the i-th candidate
	{
		int i = 0;
		for (unsigned long long mk = eligMask; mk; mk &= mk - 1ull, ++i)
			if (lane == i) candSlot = __builtin_ctzll(mk);
	}
	// candidate i's estimate, brought from lane candSlot(i) to lane i
	const float qd = __shfl(C.kd, candSlot, 64), q0 = __shfl(C.k0, candSlot, 64), q1 = __shfl(C.k1, candSlot, 64), q2 = __shfl(C.k2, candSlot, 64);
	const uint32_t rk = rand_key(c.seed, (uint32_t)idx, (uint32_t)c.itExternal * 64u + 1u + (uint32_t)iter);
	const int nR = c.nRandomIters;
	int phase = nc > 0 ? PH_PROP : PH_PICK;
	int cb = 0, t0 = 0;
	unsigned idxScaleRange = 0;
	float scaleRange = 1.f, depthRange = 0.f, p0 = 0.f, p1 = 0.f;
	bool hooked = false, pollTaken = false;
	for (;;) {
		if (phase == PH_PICK) { // the RefineIters label, DepthMap.cpp:1443-1448
			BLOCK(blk_pick, 0)
			if (!hooked) {
				// pipeline hook: publish the previous column (its stores were issued a whole scoring round ago, so this
				// wait is free) and refresh the progress of the row above with an asynchronous poll
				hooked = true;
				STAMP(5)
				if (pp.pendingPub > 0) { // NW == 1 only (see the end of process_pixel)
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					if (lane == 0) __hip_atomic_store(pp.myWord, pp.base + pp.pendingPub, __ATOMIC_RELAXED, HC_SCOPE);
					pp.pendingPub = 0;
				}
				if (pp.r > 0) pp.poll = __hip_atomic_load(pp.upWord, __ATOMIC_RELAXED, HC_SCOPE) - pp.base;
				STAMP(6)
			}
			if (conf <= c.thConfSmall) idxScaleRange = 2;
			else if (conf <= c.thConfBig) idxScaleRange = 1;
			if (conf > c.thConfBig && conf >= c.thConfRand) {
				phase = PH_RAND;
			} else {
				phase = PH_REFINE;
				scaleRange = 1.f / (float)(1u << idxScaleRange);
				depthRange = depth * c.depthRatio;
				p0 = pm_atan2f(n1, n0); p1 = pm_acosf(n2); // Normal2Dir
				t0 = 0;
			}
		}
		// ---- hypotheses of this round, lane t = hypothesis t ----
		float hd = 0.f, h0 = 0.f, h1 = 0.f, h2 = -1.f, hq0 = 0.f, hq1 = 0.f;
		int hlimit = 63, r0, r1;
		bool hv = false, planeOwn = true;
		if (phase == PH_PROP) {
			BLOCK(blk_gen_prop, 1)
			r0 = cb; r1 = cb + 8 * NW < nc ? cb + 8 * NW : nc;
			hd = qd; h0 = q0; h1 = q1; h2 = q2; hlimit = candSlot;
			hv = lane >= r0 && lane < r1;
		} else if (phase == PH_RAND) {
			BLOCK(blk_gen_rand, 2)
			r0 = 0; r1 = nR; planeOwn = false; // the smoothness plane is whatever the propagation left (DepthMap.cpp:1450-1463)
			if (lane < nR) {
				// few pixels ever get here; the laundered key keeps this (loop-invariant, speculatable) arithmetic from being
				// hoisted in front of the rounds of every pixel
				uint32_t rkr = rk;
				asm volatile("" : "+v"(rkr));
				hd = random_depth(c, rand_unit(rkr, 3u * lane));
				random_normal(G, rand_unit(rkr, 3u * lane + 1u), rand_unit(rkr, 3u * lane + 2u), h0, h1, h2);
				hv = true;
			}
		} else {
			BLOCK(blk_gen_refine, 3)
			r0 = t0; r1 = nR;
			if (lane >= t0 && lane < nR) {
				const uint32_t cbase = 64u + 3u * lane;
				hd = depth + (depthRange * scaleRange) * (2.f * rand_unit(rk, cbase) - 1.f);
				if (c.dMin <= hd && hd < c.dMax) {
					hq0 = p0 + (c.angle1Range * scaleRange) * (2.f * rand_unit(rk, cbase + 1u) - 1.f);
					hq1 = p1 + (c.angle2Range * scaleRange) * (2.f * rand_unit(rk, cbase + 2u) - 1.f);
					dir2normal(hq0, hq1, h0, h1, h2);
					hv = !(dot3(h0, h1, h2, G.v0, G.v1, 1.f) >= 0.f); // out-of-range depth / back-facing normal: not scored
				}
			}
		}
		const unsigned long long vmask = __ballot(hv);
		STAMP(3)
		// ---- my share of the round: one smoothness pass per eight hypotheses, then the scorer ----
		const int cnt = r1 - r0, per = (cnt + NW - 1) / NW, lo = r0 + wv * per, hi = (lo + per < r1 ? lo + per : r1);
		float best1 = __builtin_huge_valf(), best2 = __builtin_huge_valf(); // lane t: the two best view scores of hypothesis t
		for (int base = lo; base < hi; base += 8) {
			BLOCK(blk_chunk, 4)
			const int g = base + (lane >> 3);
			const int src = g < r1 ? g : r0;
			const float gd = __shfl(hd, src, 64), g0 = __shfl(h0, src, 64), g1 = __shfl(h1, src, 64), g2 = __shfl(h2, src, 64);
			const int glimit = __shfl(hlimit, src, 64);
			const float gp0 = planeOwn ? g0 : G.pn0, gp1 = planeOwn ? g1 : G.pn1, gp2 = planeOwn ? g2 : G.pn2;
			const float gpd = planeOwn ? -gd * dot3(g0, g1, g2, G.v0, G.v1, 1.f) : G.pd; // InitPlane
#if defined(HCMVS_ABL) && HCMVS_ABL == 2 /* diagnostic ablation: no smoothness pass */
			const float F = 1.f + 0.f * (gd + g0 + g1 + g2 + gp0 + gp1 + gp2 + gpd + (float)glimit);
#else
			BLOCKN(blk_smooth_chunks, 10, closeMask ? (64 - __builtin_clzll(closeMask) + 7) >> 3 : 0)
			const float F = smooth_pass(c, st.pk->cl, closeMask, eligMask, lane, gd, g0, g1, g2, gp0, gp1, gp2, gpd, glimit);
#endif
			const int top = base + 8 < hi ? base + 8 : hi;
			const unsigned long long todo = vmask & ((1ull << top) - 1ull) & ~((1ull << base) - 1ull); // top <= 32
			STAMP(4)
			if (todo) {
				BLOCK(blk_score_chunk, 5)
				score_chunk<S, BIG, PACK>(c, L, P, st, G.v0, G.v1, F, hd, h0, h1, h2, todo, base, __builtin_ctzll(todo), best1, best2, issued);
				if constexpr (TWO) {
					unsigned again = 0;
					score_chunk<S, BIG, PACK>(c, L1, P, st, G.v0, G.v1, F, hd, h0, h1, h2, todo, base, __builtin_ctzll(todo), best1, best2, again);
				}
			}
		}
		const float mine = two_best(c, best1, best2); // lanes without a scored hypothesis: two_best(inf, inf) = inf
		STAMP(7)
		// the poll of the row above, issued by the hook in front of this round's evaluations, has come back with their gathers (vector
		// memory returns in order): taking it here costs no wait.  Taken at the end of the pixel it was a load in flight across the loop's
		// back-edge, and the compiler guards such a value with a full `s_waitcnt vmcnt(0)` there -- which also drained the pixel's stores
		// (NW = 1) on the spot, the very wait the deferred publication is there to avoid
		if (hooked && !pollTaken) { pollTaken = true; if (pp.r > 0 && pp.poll > pp.known) pp.known = pp.poll; }
		const float all = share_scores<NW>(sh, par, lane, lo, hi, mine);
		STAMP(8)
		// ---- every wave replays the sequential accept logic ----
		if (phase == PH_PROP) {
			BLOCK(blk_acc_prop, 6)
			for (int i = r0; i < r1; ++i) {
				BLOCK(blk_acc_prop_cand, 7)
				const float nconf = rlf(all, i);
				++evals;
				if (conf > nconf) { conf = nconf; depth = rlf(hd, i); n0 = rlf(h0, i); n1 = rlf(h1, i); n2 = rlf(h2, i); }
				if (i == nc - 1) init_plane(G, rlf(hd, i), rlf(h0, i), rlf(h1, i), rlf(h2, i)); // the last candidate's plane stays
			}
			cb = r1;
			if (cb >= nc) phase = PH_PICK;
		} else if (phase == PH_RAND) {
			BLOCK(blk_acc_rand, 8)
			bool again = false;
			for (int t = 0; t < nR && !again; ++t) {
				const float nconf = rlf(all, t);
				++evals;
				if (conf > nconf) {
					conf = nconf; depth = rlf(hd, t); n0 = rlf(h0, t); n1 = rlf(h1, t); n2 = rlf(h2, t);
					if (conf < c.thConfRand) again = true;
				}
			}
			phase = again ? PH_PICK : PH_DONE;
		} else {
			// the sequential scan stops at the first valid trial that beats the estimate (DepthMap.cpp:1484): found with
			// one ballot; the trials before it were scored and rejected, the ones behind it are regenerated
			BLOCK(blk_acc_refine, 9)
			int tnext = nR;
			const unsigned long long range = ((1ull << nR) - 1ull) & ~((1ull << t0) - 1ull);
			const unsigned long long live = vmask & range;
			const unsigned long long better = __ballot(conf > all) & live;
			if (better) {
				const int t = __builtin_ctzll(better);
				evals += (unsigned)__builtin_popcountll(live & ((2ull << t) - 1ull)); // the valid trials the sequential algorithm reaches
				conf = rlf(all, t); depth = rlf(hd, t); n0 = rlf(h0, t); n1 = rlf(h1, t); n2 = rlf(h2, t);
				p0 = rlf(hq0, t); p1 = rlf(hq1, t);
				++idxScaleRange;
				scaleRange = 1.f / (float)(1u << idxScaleRange);
				tnext = t + 1; // later trials used the old estimate: regenerate them
			} else {
				evals += (unsigned)__builtin_popcountll(live);
			}
			t0 = tnext;
			if (t0 >= nR) phase = PH_DONE;
		}
		STAMP(11)
		if (phase == PH_DONE) break;
	}
	SUBMARK(blk_pixel_tail)
	if constexpr (HINT) if (c.hintDepth && iter == c.hintIter) {
		// restore variant, last sweep of the last outer iteration (restore/libs/MVS/DepthMap.cpp:1527-1549): the estimate of the
		// up-sampled coarser level is one more hypothesis; it wins even when up to 0.1 worse.  Every wave evaluates it itself.
		const float hdep = as_global(c.hintDepth)[idx];
		if (hdep > 0.f) {
			float h0 = as_global(c.hintNormal)[3 * idx], h1 = as_global(c.hintNormal)[3 * idx + 1], h2 = as_global(c.hintNormal)[3 * idx + 2];
			const float hd = interpolate_pixel(c, G, x, y, hdep, h0, h1, h2);
			correct_normal(G, h0, h1, h2);
			const float hpd = -hd * dot3(h0, h1, h2, G.v0, G.v1, 1.f); // InitPlane
			const float F = smooth_pass(c, st.pk->cl, closeMask, eligMask, lane, hd, h0, h1, h2, h0, h1, h2, hpd, 63);
			float s1 = __builtin_huge_valf(), s2 = __builtin_huge_valf();
			score_chunk<S, BIG, PACK>(c, L, P, st, G.v0, G.v1, F, hd, h0, h1, h2, 1ull, 0, 0, s1, s2, issued);
			if constexpr (TWO) {
				unsigned again = 0;
				score_chunk<S, BIG, PACK>(c, L1, P, st, G.v0, G.v1, F, hd, h0, h1, h2, 1ull, 0, 0, s1, s2, again);
			}
			const float nconf = rlf(two_best(c, s1, s2), 0);
			++evals;
			if (conf > nconf - 0.1f) { conf = nconf; depth = hd; n0 = h0; n1 = h1; n2 = h2; }
		}
	}
	if (hooked && !pollTaken && pp.r > 0 && pp.poll > pp.known) pp.known = pp.poll; // (a pixel whose last round scored nothing: rare)
	if (lane == 0) {
		float* hrec = sh.hist[q & (kHist - 1)];
		hrec[0] = depth; hrec[1] = n0; hrec[2] = n1; hrec[3] = n2; hrec[4] = conf;
		if (wv == NW - 1) { // every wave holds the result; the last one stores it
			store_dn(&c.dn[idx], depth, n0, n1, n2);
			store_f(&c.conf[idx], conf);
		}
	}
	if constexpr (NW == 1) {
		pp.pendingPub = q + 1; // published by the hook of the next pixel, a scoring round later: the drain of the stores is then free
	} else if (wv == NW - 1) {
		// several waves per row (one or two images alone on the chip: the row wavefront's critical path is what counts): the column is
		// published at once by the row's last wave.  Deferring it to the next pixel's hook as with one wave per row was measured again in
		// round 4, without the compiler's own drain at the loop's back-edge in the way: 41.3 instead of 35.4 ms per sweep of a lone 1080p
		// image -- every row then trails the one above by a scoring round more (profiles/r04_launch_modes.txt)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (lane == 0) __hip_atomic_store(pp.myWord, pp.base + q + 1, __ATOMIC_RELAXED, HC_SCOPE);
	}
	STAMP(15)

}

// ------------------------------------------------------------------------------------------------------
// sweep kernel: persistent row workers (SceneDensify.cpp:677-686 EstimateDepthMapTmp)

// One launch sweeps a BATCH of independent reference images: ticket t -> (row t / nItems of image t % nItems), so the
// rows of every image are still handed out in dependence order while the images fill each other's wavefront ramps.
#ifndef HCMVS_OCC
#define HCMVS_OCC 3 // waves per SIMD the register allocation of the 5..8-view sweep worker is held to (diagnostic builds vary it)
#endif
template <int S, int NW, bool BIG, bool TWO = false, bool PACK = false, bool HINT = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(!BIG ? HCMVS_OCC : 1, !BIG ? HCMVS_OCC : 2))) void sweep_kernel(const EstConst* __restrict__ items, int nItems, int maxRows, SweepSync sy,
                                                        int iter0, int nSweeps, int lag, int affinity, int segLen) {
	__shared__ RowShared<NW> sh;
	__shared__ WavePark<S> park[NW];
	__shared__ float bigTab[BIG ? NW : 1][4][BIG ? kBigSlots : 1]; // big-patch kernels: per-tap tables of fill_patch_big
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform, and the compiler should know it
	unsigned evals = 0, issued = 0;
	unsigned long long taps = 0; // patch taps of the sequential algorithm's evaluations (per source view)
	int par = 0, rot = (int)blockIdx.x;
	STAMP_DECL
	RowPipe<S> pp;
	pp.err = as_global(sy.error);
	for (;;) {
		// Rows are handed out per image in dependence order (one counter per image): whoever holds row r-1 of an image
		// is already running, so a waiting worker always waits on a running one.  Which image a worker serves is a
		// matter of speed only: with affinity on, a workgroup prefers the images whose index is congruent to the id of
		// its XCD (the rows of one image then share one L2, which holds the source lines the row above has just
		// fetched) and only takes rows of the other images when its own are used up.
		// ONE launch runs all nSweeps sweeps (round 4): an image's counter goes on through its sweeps (ticket t = row t % rows of sweep
		// t / rows), so there is no chip-wide drain and refill between two sweeps -- an image that has finished a sweep starts its next one
		// while the others are still in theirs.  The first row of a sweep waits until every row of the image's previous sweep is done (a
		// sweep begins at the pixel the one before it ended with, DepthMap.cpp:418); a worker looks for a row that can run first.
		if (NW > 1) __syncthreads(); // everyone is done with the previous row's shared state
		if (threadIdx.x == 0) {
			int item = -1, row = 0;
			const int G = nItems < 8 ? nItems : 8;
			const int home = affinity ? (int)(__builtin_amdgcn_s_getreg(6164) & 7u) % G : 0; // HW_REG_XCC_ID[3:0]
			const int nHome = affinity ? (nItems - home + G - 1) / G : nItems;
			// stretches: no image's tickets run out before the end of the sweep, so the preference for the home images would leave an image
			// with fewer home XCDs than the others short of workers all sweep long -- a worker keeps to its home images only while they
			// are not ahead of the image that is furthest behind (by more than 1 / 64 of a sweep)
			float behind = 2.f;
			if (segLen > 0 && affinity)
				for (int k = 0; k < nItems; ++k) {
					const int nrows_ = items[k].H - 2 * items[k].border;
					const int per_ = nrows_ * ((items[k].W - 2 * items[k].border + segLen - 1) / segLen);
					const int t_ = __hip_atomic_load(as_global(sy.ticket) + k, __ATOMIC_RELAXED, HC_SCOPE);
					if (t_ < per_ * nSweeps) { const float f = (float)t_ / (float)per_; behind = f < behind ? f : behind; }
				}
			for (int pass = 0; pass < 2 && item < 0; ++pass) // pass 0: only rows that need not wait for a sweep to end
				for (int k = 0; k < nItems && item < 0; ++k) {
					int cand;
					if (!affinity) cand = (rot + k) % nItems;
					else if (k < nHome) cand = home + ((rot + k) % nHome) * G;
					else { // the other images, in index order
						cand = k - nHome;
						cand += cand / (G - 1) + (cand % (G - 1) >= home ? 1 : 0); // skip the indices congruent to home
					}
					const int nrows_ = items[cand].H - 2 * items[cand].border;
					const int per_ = nrows_ * (segLen > 0 ? (items[cand].W - 2 * items[cand].border + segLen - 1) / segLen : 1); // tickets of a sweep
					HC_GLOBAL int32_t* tk = as_global(sy.ticket) + cand;
					const int t_ = __hip_atomic_load(tk, __ATOMIC_RELAXED, HC_SCOPE);
					if (t_ >= per_ * nSweeps) continue;
					if (pass == 0 && segLen > 0 && affinity && (float)t_ / (float)per_ > behind + 1.f / 64.f) continue;
					if (pass == 0 && t_ > 0 && t_ % per_ == 0 &&
					    __hip_atomic_load(as_global(sy.rowsDone) + cand, __ATOMIC_RELAXED, HC_SCOPE) < (t_ / per_) * nrows_) continue;
					const int r_ = atomicAdd(sy.ticket + cand, 1);
					if (r_ < per_ * nSweeps) { item = cand; row = r_; }
				}
			++rot;
			sh.row = row; sh.item = item;
		}
		__syncthreads();
		const int itemIdx = __builtin_amdgcn_readfirstlane(sh.item);
		if (itemIdx < 0) break;
		const int ticket = __builtin_amdgcn_readfirstlane(sh.row);
		const EstConst& c = items[itemIdx];
		const int bd = c.border;
		const int nrows = c.H - 2 * bd, ncols = c.W - 2 * bd;
		// SEGMENTS (segLen > 0; batches whose rows do not all fit the chip at once): a ticket is a stretch of segLen columns of a row, so
		// that a worker's slot comes free after segLen pixels instead of a whole row -- with whole rows the rows beyond the resident set
		// begin only when row 0 has reached its END, W - (resident rows per image) pixel periods late.  Tickets are numbered by the time
		// r + seg * segLen at which a stretch can begin at the earliest (its upper neighbour is one pixel ahead, its left neighbour done):
		// whatever a stretch waits for has a smaller ticket, i.e. is running or finished.
		const int nseg = segLen > 0 ? (ncols + segLen - 1) / segLen : 1;
		const int perSweep = nrows * nseg;
		const int sweep = ticket / perSweep;
		int r = ticket - sweep * perSweep, seg = 0;
		if (nseg > 1) {
			const int tt = r;
			auto before = [&](int k) { // stretches that can begin before time k
				int n = 0;
				for (int g = 0; g < nseg; ++g) { const int v = k - g * segLen; n += v < 0 ? 0 : (v > nrows ? nrows : v); }
				return n;
			};
			int lo = 0, hi = nrows + (nseg - 1) * segLen; // before(lo) <= tt < before(hi)
			while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (before(mid) <= tt) lo = mid; else hi = mid; }
			const int j = tt - before(lo);               // the j-th stretch of time lo, in the order of the segments
			const int over = lo - nrows + 1;             // segments g with lo - g * segLen < nrows: g >= over / segLen, rounded up
			const int g0 = over > 0 ? (over + segLen - 1) / segLen : 0;
			seg = g0 + j;
			r = lo - seg * segLen;
		}
		const int q0 = seg * segLen, q1 = nseg > 1 && q0 + segLen < ncols ? q0 + segLen : ncols;
		const int iter = iter0 + sweep;
		const bool rev = (iter & 1) != 0; // dir = RB2LT on odd iterations, DepthMap.cpp:418
		pp.rev = rev; pp.base = sweep << 16;
		LaneCtx<S> L;
		lane_init<S>(c, L);
		LdsStore<S> st;
		st.pk = &park[wv]; st.lane = L.lane; st.view = L.view;
		st.bw = (float (*)[kBigSlots])(BIG ? &bigTab[BIG ? wv : 0][0][0] : nullptr);
		st.put_view(as_global(c.views) + (L.vact ? L.view : 0), L.seg);
		LaneCtx<S> L1 = L; // TWO: the same lanes as views 8..15
		if constexpr (TWO) {
			lane_init<S>(c, L1, 64 / S);
			LdsStore<S> s1 = st;
			s1.view = L1.view;
			s1.put_view(as_global(c.views) + (L1.vact ? L1.view : 0), L1.seg);
		}
		const int y = rev ? c.H - 1 - bd - r : bd + r;
		pp.r = r; pp.y = y;
		pp.upWord = as_global(c.progress) + (size_t)(r > 0 ? r - 1 : 0) * kProgressStride;
		pp.myWord = as_global(c.progress) + (size_t)r * kProgressStride;
		pp.known = r > 0 ? 0 : 0x7fffffff; // columns the previous logical row has finished
		pp.poll = 0; pp.pendingPub = 0; pp.fail = false;
		STAMP(10) // (ticket)
		if (q0 > 0) { // the stretch to my left must be finished (and published: its results are in memory)
			pp.fail = wait_progress(pp.myWord, pp.base + q0, pp.err, pp.base + q0 - 4) < 0;
		}
		STAMP(14) // (wait for the stretch to my left)
		if (r > 0 && !pp.fail) {
			const int need = q0 + lag < q1 ? q0 + lag : q1; // (never beyond my own stretch: the stretch above-right has a later ticket)
			pp.known = wait_progress(pp.upWord, pp.base + need, pp.err, pp.base + q0); // (the row above stores pp.base when it begins)
			pp.fail = pp.known < 0;
			pp.known -= pp.base;
		} else if (sweep > 0 && q0 == 0) { // the image's previous sweep must be over, every row of it (fast polling for its last few rows only)
			pp.fail = wait_progress(as_global(sy.rowsDone) + itemIdx, sweep * nrows, pp.err, sweep * nrows - 2) < 0;
		}
		if (!pp.fail && (threadIdx.x == 0) && ncols > 0 && q0 == 0) __hip_atomic_store(pp.myWord, pp.base, __ATOMIC_RELAXED, HC_SCOPE); // "this row is at work"
		if (q0 > 0 && !pp.fail) {
			// the ring of the row's latest results (the neighbours behind me in my row) as the stretch to my left stored them
			const int lane = (int)(threadIdx.x & 63);
			const int col = q0 - 1 - lane;
			if (lane < kHist && col >= 0) {
				const int hx = rev ? c.W - 1 - bd - col : bd + col;
				const float4 hdn = load_dn(&c.dn[y * c.W + hx]);
				const float hcf = load_f(&c.conf[y * c.W + hx]);
				float* hrec = sh.hist[col & (kHist - 1)];
				hrec[0] = hdn.x; hrec[1] = hdn.y; hrec[2] = hdn.z; hrec[3] = hdn.w; hrec[4] = hcf; // (every wave of the row writes the same values)
			}
			if (NW > 1) __syncthreads();
		}

		const int xs = rev ? c.W - 1 - bd - q0 : bd + q0;
		const int dx = rev ? -1 : 1;
		pp.tx1 = uniform_byte(c.gra, y * c.W + xs);
		{ // the patch inputs travel one pixel ahead of the pixel being processed, the gradient byte two
			PixIn<S> first;
			first.tx = (float)pp.tx1;
			load_patch_inputs<S>(c, L, xs, y, first);
			pp.nI = first.I[0]; pp.nC = first.center;
			pp.tx2 = q0 + 1 < q1 ? uniform_byte(c.gra, y * c.W + xs + dx) : (uint8_t)0;
		}
		STAMP(9) // (the wait until the row above has begun: the ramp of the row wavefront)
		for (int q = q0; q < q1 && !pp.fail; ++q) {
			const int x = rev ? c.W - 1 - bd - q : bd + q;
			STAMP(13)
			// all loads of this pixel that do not depend on other rows go out in one batch ...
			PixIn<S> in;
			in.tx = (float)pp.tx1;
			prefetch_static<S>(c, L, x, y, q, rev, pp.known >= q + 1, in);
			in.I[0] = pp.nI; in.center = pp.nC;
			if (q + 1 < q1) { // next pixel's patch inputs (its gradient byte arrived a pixel ago)
				PixIn<S> nxt;
				nxt.tx = (float)pp.tx2;
				load_patch_inputs<S>(c, L, x + dx, y, nxt);
				pp.nI = nxt.I[0]; pp.nC = nxt.center;
			}
			pp.tx1 = pp.tx2;
			if (q + 2 < q1) pp.tx2 = uniform_byte(c.gra, y * c.W + x + 2 * dx);
			STAMP(12)
			// ... and the patch weights are computed while they (and the previous row) arrive
			Patch<S> P;
			fill_patch<S, BIG>(c, L, x, y, in, P, st);
			STAMP(1)
			if (pp.known < q + 1) { // the previous row must have finished this column
				pp.known = wait_progress(pp.upWord, pp.base + q + 1, pp.err);
				if (pp.known < 0) { // (the same hand-over as below, so that the two ways out of the pixel meet with nothing in flight)
					pp.fail = true;
					asm volatile("s_waitcnt vmcnt(0)" : "+v"(pp.nI), "+v"(pp.nC) :: "memory");
					break;
				}
				pp.known -= pp.base;
				prefetch_up<S>(c, in);
			}
			// The next pixel's patch inputs (issued above, a pixel ahead) and this pixel's own loads are all back before the pixel's first
			// arithmetic needs the latter: wait for everything HERE, where it is free, and hand pp.nI / pp.nC on as plain registers -- a
			// load still "in flight" at the loop's back-edge would make the compiler put a full `s_waitcnt vmcnt(0)` there (see the hook's poll)
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(pp.nI), "+v"(pp.nC) :: "memory");
			STAMP(0)
			const unsigned e0 = evals;
			process_pixel<S, NW, BIG, TWO, PACK, HINT>(c, L, L1, sh, par, wv, x, y, q, iter, in, P, st, pp, evals, issued STAMP_PASS);
			taps += (unsigned long long)(evals - e0) * (unsigned)((P.a + 1) * (P.a + 1));
			STAMP(5) // (the pixel loop's back-edge is stamp 13's alone)
		}
		if (pp.fail) break;
		if (pp.pendingPub > 0) { // last column of the row (NW == 1)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if ((threadIdx.x & 63) == 0) __hip_atomic_store(pp.myWord, pp.base + pp.pendingPub, __ATOMIC_RELAXED, HC_SCOPE);
			pp.pendingPub = 0;
		}
		// the row is done and published (with several waves per row: by its last wave, whose stores were drained before it published):
		// one more row of this sweep of the image
		if (NW > 1) __syncthreads();
		if (threadIdx.x == 0 && q1 == c.W - 2 * bd) __hip_atomic_fetch_add(as_global(sy.rowsDone) + itemIdx, 1, __ATOMIC_RELAXED, HC_SCOPE);
	}
	STAMP(10)
	STAMP_FLUSH
	if ((threadIdx.x & 63) == 0) {
		if (wv == 0 && evals) { atomicAdd(sy.evals, (unsigned long long)evals); atomicAdd(sy.evals + 2, taps); }
		if (issued) atomicAdd(sy.evals + 1, (unsigned long long)issued);
	}
}

// ------------------------------------------------------------------------------------------------------
// init-score pass (SceneDensify.cpp:649-675 ScoreDepthMapTmp): no inter-pixel dependence

__global__ void import_kernel(EstConst c, const float* depthIn, const float* normalIn) {
	const int n = c.W * c.H;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int x = i % c.W, y = i / c.W;
		const bool inb = x >= c.border && y >= c.border && x < c.W - c.border && y < c.H - c.border;
		if (inb) {
			c.dn[i] = make_float4(depthIn[i], normalIn[3 * i], normalIn[3 * i + 1], normalIn[3 * i + 2]);
			c.conf[i] = 2.f;
		} else {
			c.dn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
			c.conf[i] = 2.f;
		}
	}
}

template <int S, bool BIG, bool TWO = false>
__global__ __launch_bounds__(256) void score_kernel(EstConst c, unsigned long long* evalsOut) {
	LaneCtx<S> L;
	lane_init<S>(c, L);
	__shared__ float stage[4][3][8][8]; // one hand-over area per wave (launch_bounds 256)
	__shared__ float bigTab[BIG ? 4 : 1][4][BIG ? kBigSlots : 1];
	RegStore<S> st;
	st.stage = stage[threadIdx.x >> 6]; st.seg = L.seg;
	st.bw = (float (*)[kBigSlots])(BIG ? &bigTab[BIG ? (threadIdx.x >> 6) : 0][0][0] : nullptr);
	st.put_view(as_global(c.views) + (L.vact ? L.view : 0), L.seg);
	LaneCtx<S> L1 = L;    // TWO (9..16 source views): the same lanes as views 8..15, with their own view constants
	RegStore<S> st1 = st;
	if constexpr (TWO) {
		lane_init<S>(c, L1, 64 / S);
		st1.put_view(as_global(c.views) + (L1.vact ? L1.view : 0), L1.seg);
	}
	const int nrows = c.H - 2 * c.border, ncols = c.W - 2 * c.border;
	const int total = nrows * ncols;
	const int wavesPerBlock = blockDim.x >> 6;
	const int gw = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6), nw = gridDim.x * wavesPerBlock;
	unsigned evals = 0;
	unsigned long long taps = 0;
	const uint32_t stream0 = (uint32_t)c.itExternal * 64u;
	for (int p = gw; p < total; p += nw) {
		const int x = c.border + p % ncols, y = c.border + p / ncols;
		const int idx = y * c.W + x;
		const uint32_t rk = rand_key(c.seed, (uint32_t)idx, stream0);
		PixIn<S> in;
		in.tx = (float)c.gra[idx];
		load_patch_inputs<S>(c, L, x, y, in);
		Patch<S> P;
		fill_patch<S, BIG>(c, L, x, y, in, P, st);
		PixelGeom G;
		pixel_geom(c, x, y, G);
		const float4 cur = c.dn[idx];
		float d = cur.x, n0 = cur.y, n1 = cur.z, n2 = cur.w;
		if (!(c.dMin <= d && d < c.dMax)) {
			d = random_depth(c, rand_unit(rk, 0u));
			random_normal(G, rand_unit(rk, 1u), rand_unit(rk, 2u), n0, n1, n2);
		} else if (dot3(n0, n1, n2, G.v0, G.v1, 1.f) >= 0.f) {
			random_normal(G, rand_unit(rk, 1u), rand_unit(rk, 2u), n0, n1, n2);
		}
		float m1 = __builtin_huge_valf(), m2 = __builtin_huge_valf();
		score_pixel<S, BIG>(c, L, P, st, G.v0, G.v1, 1.f, d, n0, n1, n2, m1, m2);
		if constexpr (TWO) {
			st1.copy_patch(st); // the pixel's patch tables (registers) with the second set's view constants
			score_pixel<S, BIG>(c, L1, P, st1, G.v0, G.v1, 1.f, d, n0, n1, n2, m1, m2);
		}
		const float s = two_best(c, m1, m2);
		++evals;
		taps += (unsigned)((P.a + 1) * (P.a + 1));
		if (L.lane == 0) {
			c.dn[idx] = make_float4(d, n0, n1, n2);
			c.conf[idx] = s;
		}
	}
	if (L.lane == 0 && evals) {
		atomicAdd(evalsOut, (unsigned long long)evals);
		atomicAdd(evalsOut + 1, (unsigned long long)evals);
		atomicAdd(evalsOut + 2, taps);
	}
}

#ifdef HCMVS_COUNT
// instruction-count probes (never built into the library): the scorer and the smoothness pass in isolation
template <int S>
__global__ void probe_score_kernel(EstConst c, Patch<S> P, RegStore<S> st, float F, float d, float n0, float n1, float n2, float* out) {
	LaneCtx<S> L;
	lane_init<S>(c, L);
	float m1 = __builtin_huge_valf(), m2 = __builtin_huge_valf();
	score_pixel<S, false>(c, L, P, st, 0.1f, 0.2f, F, d, n0, n1, n2, m1, m2);
	out[threadIdx.x] = two_best(c, m1, m2);
}
template __global__ void probe_score_kernel<8>(EstConst, Patch<8>, RegStore<8>, float, float, float, float, float, float*);
__global__ void probe_smooth_kernel(EstConst c, Close C, float* out) {
	__shared__ float cl[9][kMaxSlots];
	for (int k = 0; k < 9; ++k) cl[k][threadIdx.x & 31] = out[k * 64 + threadIdx.x];
	out[threadIdx.x] = smooth_pass(c, cl, C.closeMask, C.eligMask, threadIdx.x & 63, out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], 63);
}
template <int S>
__global__ void probe_patch_kernel(EstConst c, PixIn<S> in, Patch<S>* out, RegStore<S>* so) {
	LaneCtx<S> L;
	lane_init<S>(c, L);
	Patch<S> P;
	__shared__ float stage[3][8][8];
	RegStore<S> st;
	st.stage = stage; st.seg = L.seg;
	fill_patch<S, false>(c, L, 100, 100, in, P, st);
	out[threadIdx.x] = P;
	so[threadIdx.x] = st;
}
template __global__ void probe_patch_kernel<8>(EstConst, PixIn<8>, Patch<8>*, RegStore<8>*);
#endif

// SceneDensify.cpp:688-744 EndDepthMapTmp (finalPass) or plain export of the working state
__global__ void end_kernel(EstConst c, int finalPass, float* depth, float* normal, float* conf) {
	const int n = c.W * c.H;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		float4 v = c.dn[i];
		float cf = c.conf[i];
		if (finalPass) {
			if (v.x <= 0.f || cf >= c.thKeep) { v = make_float4(0.f, 0.f, 0.f, 0.f); cf = 0.f; }
			else cf = cf >= 1.f ? 0.f : 1.f - cf;
		}
		depth[i] = v.x;
		normal[3 * i] = v.y; normal[3 * i + 1] = v.z; normal[3 * i + 2] = v.w;
		conf[i] = cf;
	}
}

// ------------------------------------------------------------------------------------------------------
// image helpers (SceneDensify.cpp:581-595 InitGraMap, :859 medianBlur)

__global__ void gray_to_u8_kernel(const float* g, uint8_t* out, int n) {
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int r = (int)rintf(g[i] * 255.f);
		out[i] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
	}
}
__global__ void bgr_to_u8_kernel(const uint8_t* bgr, uint8_t* out, int n) {
	// cv::cvtColor BGR2GRAY for 8-bit: fixed point, 14 fractional bits
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		out[i] = (uint8_t)((bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + 8192) >> 14);
}
__device__ __forceinline__ int reflect101(int i, int n) {
	if (n == 1) return 0;
	while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
	return i;
}
__global__ void gradient_kernel(const uint8_t* g, uint8_t* gra, int W, int H) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= W || y >= H) return;
	const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W), ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
	const int a00 = g[ym * W + xm], a01 = g[ym * W + x], a02 = g[ym * W + xp];
	const int a10 = g[y * W + xm], a12 = g[y * W + xp];
	const int a20 = g[yp * W + xm], a21 = g[yp * W + x], a22 = g[yp * W + xp];
	int ax = abs((a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20));
	int ay = abs((a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02));
	ax = ax > 255 ? 255 : ax; ay = ay > 255 ? 255 : ay;
	// addWeighted(0.5, 0.5) + saturate_cast<uchar>(cvRound()): round half to even
	const int s2 = ax + ay; // 2 * result
	int r = s2 >> 1;
	if ((s2 & 1) && (r & 1)) r += 1;
	gra[y * W + x] = (uint8_t)(r > 255 ? 255 : r);
}
// The layout the scorer gathers from: per pixel the 2 x 2 footprint of a bilinear sample (Types.inl:2250-2258 reads I(x,y), I(x+1,y),
// I(x,y+1), I(x+1,y+1)) as (I00, I10 - I00, I01, I11 - I01): one 16-byte load serves a sample, and the two horizontal differences
// of the lerp form are taken here, once per texel, instead of once per sample -- the same IEEE subtraction of the same operands,
// so the sample's bits do not change.  The last column / row repeat (never sampled: positions are inside the image with a border
// of 1, Types.h:1633-1635).  Four times the bytes of the image -- HBM is sized for it -- and the same number of cache lines per
// wave-instruction as two 8-byte gathers from two rows.
__global__ void quad_kernel(const float* g, float4* out, int W, int H) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= W || y >= H) return;
	const int x1 = x + 1 < W ? x + 1 : W - 1, y1 = y + 1 < H ? y + 1 : H - 1;
	const float i00 = g[(size_t)y * W + x], i10 = g[(size_t)y * W + x1], i01 = g[(size_t)y1 * W + x], i11 = g[(size_t)y1 * W + x1];
	out[(size_t)y * W + x] = make_float4(i00, i10 - i00, i01, i11 - i01);
}
__global__ void median3_kernel(const float* in, float* out, int W, int H) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
	if (x >= W || y >= H) return;
	float v[9];
	int n = 0;
#pragma unroll
	for (int dy = -1; dy <= 1; ++dy) {
		int yy = y + dy; yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
		for (int dx = -1; dx <= 1; ++dx) {
			int xx = x + dx; xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
			v[n++] = in[yy * W + xx];
		}
	}
	// median of 9 by a fixed exchange network (Paeth)
#define HC_SORT2(a, b) { const float lo = fminf(v[a], v[b]), hi = fmaxf(v[a], v[b]); v[a] = lo; v[b] = hi; }
	HC_SORT2(1, 2) HC_SORT2(4, 5) HC_SORT2(7, 8) HC_SORT2(0, 1) HC_SORT2(3, 4) HC_SORT2(6, 7) HC_SORT2(1, 2) HC_SORT2(4, 5)
	HC_SORT2(7, 8) HC_SORT2(0, 3) HC_SORT2(5, 8) HC_SORT2(4, 7) HC_SORT2(3, 6) HC_SORT2(1, 4) HC_SORT2(2, 5) HC_SORT2(4, 7)
	HC_SORT2(4, 2) HC_SORT2(6, 4) HC_SORT2(4, 2)
#undef HC_SORT2
	out[y * W + x] = v[4];
}

// ------------------------------------------------------------------------------------------------------
// launch wrappers

#ifdef HCMVS_STAMPS
void debug_read_stamps(unsigned long long* out, int reset) {
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
	if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z); }
}
#endif

// lane layout class of an estimate with V source views.  Every estimate runs the 8 view groups x 8 tap segments layout: once
// for up to 8 views (class 8), twice for 9..16 (class 4: two sets of eight view groups, TWO in the kernels); view counts that
// leave groups idle let them work on (hypothesis, view) pairs of their own (PACK)
int segments_for(int V) { return V <= 8 ? 8 : 4; }

void launch_gray_to_u8(const float* gray, uint8_t* out, int n, hipStream_t s) {
	hipLaunchKernelGGL(gray_to_u8_kernel, dim3(2048), dim3(256), 0, s, gray, out, n);
}
void launch_bgr_to_u8(const uint8_t* bgr, uint8_t* out, int n, hipStream_t s) {
	hipLaunchKernelGGL(bgr_to_u8_kernel, dim3(2048), dim3(256), 0, s, bgr, out, n);
}
void launch_gradient_map(const uint8_t* g8, uint8_t* gra, int W, int H, hipStream_t s) {
	hipLaunchKernelGGL(gradient_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, g8, gra, W, H);
}
void launch_quads(const float* gray, float4* out, int W, int H, hipStream_t s) {
	hipLaunchKernelGGL(quad_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, gray, out, W, H);
}
void launch_median3(const float* in, float* out, int W, int H, hipStream_t s) {
	hipLaunchKernelGGL(median3_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, in, out, W, H);
}

template <bool BIG>
static void launch_score_big(const EstConst& c, unsigned long long* evals, hipStream_t s) {
	const dim3 grid(4096), block(256);
	if (c.V <= 8) hipLaunchKernelGGL((score_kernel<8, BIG>), grid, block, 0, s, c, evals);
	else hipLaunchKernelGGL((score_kernel<8, BIG, true>), grid, block, 0, s, c, evals); // 9..16 views: two sets of eight

}
void launch_score_pass(const EstConst& c, const float* depthIn, const float* normalIn, unsigned long long* evals,
                       hipStream_t s) {
	hipLaunchKernelGGL(import_kernel, dim3(2048), dim3(256), 0, s, c, depthIn, normalIn);
	if (c.adapthalfwin > kHalfWindow) launch_score_big<true>(c, evals, s);
	else launch_score_big<false>(c, evals, s);
}

template <int NW, bool BIG, bool HINT>
static void launch_sweep_nw(const EstConst* dItems, int nItems, int maxRows, int totalRows, int V, const SweepSync& sync, int iter, int nSweeps, int lag,
                            int affinity, int segLen, hipStream_t s) {
	// one workgroup per row; rows beyond the resident set are picked up through the ticket
	int grid = totalRows < 8192 ? totalRows : 8192;
	if (grid < 1) return;
	const dim3 g(grid), b(64 * NW);
	// a view count that leaves two or more of a set's eight view groups idle takes the variant whose idle groups work on
	// (hypothesis, view) pairs of their own (score_chunk PACK; with one idle group it costs more than it saves)
	const bool pack = V % 8 != 0 && V % 8 != 7;
	if (V <= 8) {
		if (!pack) hipLaunchKernelGGL((sweep_kernel<8, NW, BIG, false, false, HINT>), g, b, 0, s, dItems, nItems, maxRows, sync, iter, nSweeps, lag, affinity, segLen);
		else hipLaunchKernelGGL((sweep_kernel<8, NW, BIG, false, true, HINT>), g, b, 0, s, dItems, nItems, maxRows, sync, iter, nSweeps, lag, affinity, segLen);
	} else {
		if (!pack) hipLaunchKernelGGL((sweep_kernel<8, NW, BIG, true, false, HINT>), g, b, 0, s, dItems, nItems, maxRows, sync, iter, nSweeps, lag, affinity, segLen);
		else hipLaunchKernelGGL((sweep_kernel<8, NW, BIG, true, true, HINT>), g, b, 0, s, dItems, nItems, maxRows, sync, iter, nSweeps, lag, affinity, segLen);
	}

}
template <bool HINT>
static void launch_sweep_hint(const EstConst* dItems, int nItems, int maxRows, int totalRows, int V, bool bigPatch, const SweepSync& sync, int iter, int nSweeps, int lag,
                              int wavesPerRow, int affinity, int segLen, hipStream_t s) {
	if (bigPatch) { // patches beyond 64 taps: one or two waves per row
		if (wavesPerRow >= 2) launch_sweep_nw<2, true, HINT>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s);
		else launch_sweep_nw<1, true, HINT>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s);
		return;
	}
	if constexpr (HINT) { // the one sweep of a run that carries the hint: one or two waves per row
		if (wavesPerRow >= 2) launch_sweep_nw<2, false, true>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s);
		else launch_sweep_nw<1, false, true>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s);
		return;
	}
	switch (wavesPerRow) {
	case 1: launch_sweep_nw<1, false, HINT>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s); break;
	case 3: launch_sweep_nw<3, false, HINT>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s); break;
	case 4: launch_sweep_nw<4, false, HINT>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s); break;
	default: launch_sweep_nw<2, false, HINT>(dItems, nItems, maxRows, totalRows, V, sync, iter, nSweeps, lag, affinity, segLen, s); break;
	}
}
// One launch for the sweeps iter .. iter + nSweeps - 1 of every item (tickets, rowsDone and the progress words of the items must be
// zero).  hint: some item of the batch offers the `restore` variant's extra hypothesis in one of these sweeps (EstConst::hintDepth,
// hintIter): the instance that knows the hint.
void launch_sweep(const EstConst* dItems, int nItems, int maxRows, int totalRows, int V, bool bigPatch, bool hint, const SweepSync& sync, int iter, int nSweeps,
                  int lag, int wavesPerRow, int affinity, int segLen, hipStream_t s) {
	if (hint) launch_sweep_hint<true>(dItems, nItems, maxRows, totalRows, V, bigPatch, sync, iter, nSweeps, lag, wavesPerRow, affinity, segLen, s);
	else launch_sweep_hint<false>(dItems, nItems, maxRows, totalRows, V, bigPatch, sync, iter, nSweeps, lag, wavesPerRow, affinity, segLen, s);
}

void launch_end_pass(const EstConst& c, int finalPass, float* depth, float* normal, float* conf, hipStream_t s) {
	hipLaunchKernelGGL(end_kernel, dim3(2048), dim3(256), 0, s, c, finalPass, depth, normal, conf);
}

} // namespace hcmvs
