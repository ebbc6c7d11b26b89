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
 *   - ONE WAVEFRONT PER IMAGE ROW.  The reference sweeps pixels sequentially (Gauss-Seidel): a pixel must
 *     see its left/up neighbours already updated and its right/down neighbours not yet updated.  Rows
 *     advancing left-to-right with row y one pixel behind row y-1 satisfy exactly that dependence, so all
 *     rows run concurrently as persistent waves that hand results down through HBM/L2 with agent-scope
 *     (sc1) stores + a per-row progress word.  The maps are identical to the sequential sweep.
 *   - INSIDE A WAVE the 64 lanes are (view group) x (tap segment): lane = view*S + seg.  Each lane warps
 *     and bilinearly samples its ~T*T/S taps of its own source view, partial sums are combined with an
 *     xor butterfly inside the group, the per-view ZNCC epilogue runs lane-parallel over views and the
 *     "two best views" selection is a second butterfly across groups.  No LDS, no block barriers.
 *   - Rows are handed out by an atomic ticket in dependency order, so a waiting wave always waits on a
 *     wave that is already running: no deadlock for any grid size or dispatch order.  Every spin is bounded.
 *
 * Arithmetic is an explicitly specified IEEE sequence (explicit fmaf, one IEEE reciprocal per tap,
 * pm_math.h transcendental functions); compile with -ffp-contract=off.
 */
#include "pm_common.h"
#include "pm_math.h"

#include <float.h>

namespace hcmvs {

#define HC_SQ(x) ((x) * (x))
#define HC_SCOPE __HIP_MEMORY_SCOPE_AGENT

// ------------------------------------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
	h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
	return h;
}
// counter-based RNG keyed by (seed, pixel, pass, draw): replaces the per-thread mt19937 (Random.h:102)
__device__ __forceinline__ float rand_unit(uint32_t seed, uint32_t pix, uint32_t stream, uint32_t ctr) {
	uint32_t h = fmix32(seed ^ 0x9e3779b9u);
	h = fmix32(h ^ (pix * 0x9e3779b1u));
	h = fmix32(h ^ (stream * 0x85ebca77u));
	h = fmix32(h ^ (ctr * 0xc2b2ae3du));
	return (float)h / 4294967296.0f; // (float)max() == 2^32 (Random.h:113-115)
}
__device__ __forceinline__ float fd2r(float d) { return d * (3.14159274101257324f / 180.f); } // Types.h:566

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
	return a0 * b0 + a1 * b1 + a2 * b2;
}

// agent-scope (sc1) accesses for everything another row's wave may have written in this launch
__device__ __forceinline__ float4 load_dn(const float4* p) {
	unsigned long long* q = (unsigned long long*)p;
	const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, HC_SCOPE);
	const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, HC_SCOPE);
	return make_float4(__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)(a >> 32)), __uint_as_float((uint32_t)b),
	                   __uint_as_float((uint32_t)(b >> 32)));
}
__device__ __forceinline__ void store_dn(float4* p, float d, float n0, float n1, float n2) {
	unsigned long long* q = (unsigned long long*)p;
	__hip_atomic_store(q, (unsigned long long)__float_as_uint(d) | ((unsigned long long)__float_as_uint(n0) << 32),
	                   __ATOMIC_RELAXED, HC_SCOPE);
	__hip_atomic_store(q + 1, (unsigned long long)__float_as_uint(n1) | ((unsigned long long)__float_as_uint(n2) << 32),
	                   __ATOMIC_RELAXED, HC_SCOPE);
}
__device__ __forceinline__ float load_f(const float* p) {
	return __uint_as_float(__hip_atomic_load((uint32_t*)p, __ATOMIC_RELAXED, HC_SCOPE));
}
__device__ __forceinline__ void store_f(float* p, float v) {
	__hip_atomic_store((uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, HC_SCOPE);
}

__device__ __forceinline__ float rlf(float v, int lane) { // value of a (wave-uniform index) lane
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int rli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

template <int S>
__device__ __forceinline__ float group_sum(float v) { // xor butterfly inside a view group
#pragma unroll
	for (int step = 1; step < S; step <<= 1) v = v + __shfl_xor(v, step, 64);
	return v;
}

// ------------------------------------------------------------------------------------------------------
// per-lane context

template <int S>
struct LaneCtx {
	static constexpr int MAXM = 64 / S; // taps per lane
	int lane, view, seg;
	bool vact;           // lane's view group exists
	const float* img;    // my source view
	int iw, ih;
	double A[9], Hm[3];
	unsigned long long groupMask;
};

template <int S>
__device__ __forceinline__ void lane_init(const EstConst& c, LaneCtx<S>& L) {
	L.lane = threadIdx.x & 63;
	L.view = L.lane / S;
	L.seg = L.lane % S;
	L.vact = L.view < c.V;
	const DevView* dv = &c.views[L.vact ? L.view : 0];
	L.img = dv->img; L.iw = dv->w; L.ih = dv->h;
#pragma unroll
	for (int i = 0; i < 9; ++i) L.A[i] = dv->A[i];
#pragma unroll
	for (int i = 0; i < 3; ++i) L.Hm[i] = dv->Hm[i];
	L.groupMask = (S == 64 ? ~0ull : ((1ull << S) - 1ull)) << (L.view * S);
}

template <int S>
struct Patch { // DepthMap.h:202-212 WeightedPatchFix, spread over the lanes of a group
	static constexpr int MAXM = 64 / S;
	float w[MAXM], tw[MAXM], px[MAXM], py[MAXM];
	int ntaps;
	float sumW, normSq0;
};

// DepthMap.cpp:450-519 FillPixelPatch + DepthMap.h:537-548 GetWeight
template <int S>
__device__ __forceinline__ void fill_patch(const EstConst& c, const LaneCtx<S>& L, int x, int y, Patch<S>& P) {
	constexpr int MAXM = 64 / S;
	const int W = c.W;
	const float tx = (float)c.gra[y * W + x];
	const int a = tx > 100.f ? 5 : c.adapthalfwin;
	const int nside = a + 1;
	const int ntaps = nside * nside;
	const int magic = (1024 + nside - 1) / nside; // (k*magic)>>10 == k/nside for k < 64, nside 2..8
	P.ntaps = ntaps;
	const float colCenter = c.ref[y * W + x];
	const float sigmaColor = -1.f / (2.f * HC_SQ(0.2f));
	const float sigmaSpatial = -1.f / (2.f * (float)HC_SQ(a));
	float I[MAXM];
	float sa = 0.f, sb = 0.f;
#pragma unroll
	for (int m = 0; m < MAXM; ++m) {
		const int k = m * S + L.seg;
		I[m] = 0.f; P.w[m] = 0.f; P.tw[m] = 0.f; P.px[m] = 0.f; P.py[m] = 0.f;
		if (k < ntaps) {
			const int ti = (k * magic) >> 10, tj = k - ti * nside;
			const int i = -a + 2 * ti, j = -a + 2 * tj;
			P.px[m] = (float)(x + j);
			P.py[m] = (float)(y + i);
			I[m] = c.ref[(y + i) * W + (x + j)];
			const float wColor = HC_SQ(I[m] - colCenter) * sigmaColor;
			const float wSpatial = (float)(HC_SQ(j) + HC_SQ(i)) * sigmaSpatial;
			P.w[m] = pm_expf(wColor + wSpatial);
			sa = fmaf(I[m], P.w[m], sa);
			sb = sb + P.w[m];
		}
	}
	const float swi = group_sum<S>(sa), sw = group_sum<S>(sb);
	const float tm = swi / sw;
	sa = 0.f;
#pragma unroll
	for (int m = 0; m < MAXM; ++m) {
		const int k = m * S + L.seg;
		if (k < ntaps) {
			const float t = I[m] - tm;
			P.tw[m] = P.w[m] * t;
			sa = fmaf(P.tw[m], t, sa);
		}
	}
	P.sumW = sw;
	P.normSq0 = group_sum<S>(sa);
}

// smoothness neighbours (DepthMap.h:376-382 NeighborEstimate): slot k lives in lane k
struct Close {
	float d, n0, n1, n2, X0, X1, X2, conf;
	int nx, ny;
	unsigned long long closeMask, propMask;
};

struct PixelGeom {
	double X0x, X0y;          // pixel ray (z = 1), Camera.h:299-304
	float v0, v1;             // (float) of it; v2 == 1
	float pn0, pn1, pn2, pd;  // smoothness plane (DepthMap.cpp:1730-1738)
};

// DepthMap.cpp:987-1046 ScorePixel over DepthMap.cpp:522-616 ScorePixelImage, all views at once
template <int S>
__device__ __forceinline__ float score_pixel(const EstConst& c, const LaneCtx<S>& L, const Patch<S>& P, const Close& C,
                                             const PixelGeom& G, float depth, float n0, float n1, float n2) {
	constexpr int MAXM = 64 / S;
	// smoothness factor of my slot (DepthMap.cpp:607-615); slot k is lane k
	float f = 1.f;
	if ((C.closeMask >> L.lane) & 1ull) {
		const float dist = dot3(G.pn0, G.pn1, G.pn2, C.X0, C.X1, C.X2) + G.pd;
		const float fd = pm_expf(HC_SQ(dist / depth) * c.smoothSigmaDepth);
		float ca = dot3(n0, n1, n2, C.n0, C.n1, C.n2) / sqrtf(dot3(n0, n1, n2, n0, n1, n2) * dot3(C.n0, C.n1, C.n2, C.n0, C.n1, C.n2));
		ca = ca < -1.f ? -1.f : (ca > 1.f ? 1.f : ca);
		const float ang = pm_acosf(ca);
		const float fn = pm_expf(HC_SQ(ang) * c.smoothSigmaNormal);
		f = (1.f - c.smoothBonusDepth * fd) * (1.f - c.smoothBonusNormal * fn);
	}
	// homography of my view (DepthMap.h:565-574), association H = A + Hm (Hr^T n)^T / (n.X0 d)
	float H[9];
	{
		const double d0 = n0, d1 = n1, d2 = n2;
		const double nx0 = fma(d2, 1.0, fma(d1, G.X0y, d0 * G.X0x));
		const double inv = 1.0 / (nx0 * (double)depth);
		double q[3];
#pragma unroll
		for (int j = 0; j < 3; ++j) q[j] = fma(d2, c.Hr[6 + j], fma(d1, c.Hr[3 + j], d0 * c.Hr[j])) * inv;
#pragma unroll
		for (int i = 0; i < 3; ++i)
#pragma unroll
			for (int j = 0; j < 3; ++j) H[i * 3 + j] = (float)fma(L.Hm[i], q[j], L.A[i * 3 + j]);
	}
	float sum = 0.f, sumSq = 0.f, num = 0.f;
	bool bad = false;
	const float wmax = (float)(L.iw - 2), hmax = (float)(L.ih - 2);
#pragma unroll
	for (int m = 0; m < MAXM; ++m) {
		const int k = m * S + L.seg;
		if (k < P.ntaps && L.vact) {
			const float px = P.px[m], py = P.py[m];
			const float Xx = fmaf(H[0], px, fmaf(H[1], py, H[2]));
			const float Xy = fmaf(H[3], px, fmaf(H[4], py, H[5]));
			const float Xz = fmaf(H[6], px, fmaf(H[7], py, H[8]));
			const float iz = 1.0f / Xz;
			const float qx = Xx * iz, qy = Xy * iz;
			if (qx >= 1.f && qy >= 1.f && qx <= wmax && qy <= hmax) { // Types.h:1633-1635
				const int lx = (int)qx, ly = (int)qy;
				const float fx = qx - (float)lx, fx1 = 1.f - fx;
				const float fy = qy - (float)ly, fy1 = 1.f - fy;
				const float* r0 = L.img + (size_t)ly * L.iw + lx;
				const float* r1 = r0 + L.iw;
				const float i00 = r0[0], i01 = r0[1], i10 = r1[0], i11 = r1[1];
				float t = i00 * fx1; t = fmaf(i01, fx, t);
				float b = i10 * fx1; b = fmaf(i11, fx, b);
				float val = t * fy1; val = fmaf(b, fy, val); // Types.inl:2250-2258
				const float vw = val * P.w[m];
				sum = sum + vw;
				sumSq = fmaf(val, vw, sumSq);
				num = fmaf(val, P.tw[m], num);
			} else {
				bad = true;
			}
		}
	}
	const bool viewBad = (__ballot(bad) & L.groupMask) != 0ull;
	sum = group_sum<S>(sum); sumSq = group_sum<S>(sumSq); num = group_sum<S>(num);
	const float normSq1 = sumSq - HC_SQ(sum) / P.sumW;
	const float nrmSq = P.normSq0 * normSq1;
	float ncc = num / sqrtf(nrmSq);
	ncc = ncc < -1.f ? -1.f : (ncc > 1.f ? 1.f : ncc);
	float s = 1.f - ncc;
	for (unsigned long long mk = C.closeMask; mk;) {
		const int k = __builtin_ctzll(mk);
		mk &= mk - 1ull;
		s *= rlf(f, k);
	}
	s = c.pfScale * s;
	if (viewBad || !(nrmSq > 0.f)) s = c.thRobust;
	float m1 = L.vact ? s : __builtin_huge_valf(), m2 = __builtin_huge_valf();
#pragma unroll
	for (int step = S; step < 64; step <<= 1) { // two smallest scores across view groups
		const float o1 = __shfl_xor(m1, step, 64), o2 = __shfl_xor(m2, step, 64);
		const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
		m2 = fminf(hi, fminf(m2, o2));
		m1 = lo;
	}
	if (c.V <= 1) return m1;
	return m2 >= c.thRobust ? m1 : (m1 + m2) / 2.f;
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
// DepthMap.h:629-634 CorrectNormal + Rotation.inl:707-733 (Rodrigues)
__device__ __forceinline__ void correct_normal(const PixelGeom& G, float& n0, float& n1, float& n2) {
	const float v0 = G.v0, v1 = G.v1, v2 = 1.f;
	const float cosAngLen = dot3(n0, n1, n2, v0, v1, v2);
	if (!(cosAngLen >= 0.f)) return;
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
// DepthMap.cpp:1671-1726 InterpolatePixel (ray-plane form)
__device__ __forceinline__ float interpolate_pixel(const EstConst& c, const PixelGeom& G, int nx, int ny, float depth,
                                                   float n0, float n1, float n2) {
	const double p0 = n0, p1 = n1, p2 = n2, z = depth;
	const double P0 = ((double)nx - c.cx) * z / c.fx, P1 = ((double)ny - c.cy) * z / c.fy;
	const double planeD = p0 * P0 + p1 * P1 + p2 * z;
	const float dn = (float)(planeD / (p0 * G.X0x + p1 * G.X0y + p2 * 1.0));
	return (c.dMin <= dn && dn < c.dMax) ? dn : depth;
}
__device__ __forceinline__ void init_plane(PixelGeom& G, float depth, float n0, float n1, float n2) {
	G.pn0 = n0; G.pn1 = n1; G.pn2 = n2;
	G.pd = -depth * dot3(n0, n1, n2, G.v0, G.v1, 1.f);
}
__device__ __forceinline__ void pixel_geom(const EstConst& c, int x, int y, PixelGeom& G) {
	G.X0x = ((double)x - c.cx) / c.fx;
	G.X0y = ((double)y - c.cy) / c.fy;
	G.v0 = (float)G.X0x; G.v1 = (float)G.X0y;
	G.pn0 = G.pn1 = G.pn2 = G.pd = 0.f;
}

// ------------------------------------------------------------------------------------------------------
// DepthMap.cpp:1050-1501 ProcessPixel for pixel (x,y), executed by one wave

template <int S>
__device__ __forceinline__ void process_pixel(const EstConst& c, const LaneCtx<S>& L, int x, int y, int iter, bool rev,
                                              unsigned& evals) {
	const int W = c.W, H = c.H, lane = L.lane;
	Patch<S> P;
	fill_patch<S>(c, L, x, y, P);
	PixelGeom G;
	pixel_geom(c, x, y, G);

	// neighbour slots: slot k in lane k
	Close C;
	int nx = x, ny = y;
	bool slot = false, sprop = false;
	if (c.itExternal >= 1) {
		// DepthMap.cpp:1064-1274: cross pattern (same set for both sweep directions)
		const float tx = (float)c.gra[y * W + x];
		int phw = tx > 150.f ? 5 : c.propHalfwin;
		if (phw > 7) phw = 7;
		const int step = c.propStep > 0 ? c.propStep : 1;
		int i = 0;
		if (x > phw && y > phw && x < W - phw && y < H - phw) {
			const int ni = phw >= 1 ? (phw - 1) / step + 1 : 0;
			if (lane < 4 * ni) { i = 1 + (lane >> 2) * step; slot = true; }
		} else if (x > kHalfWindow && y > kHalfWindow && x < W - kHalfWindow && y < H - kHalfWindow) {
			if (lane < 4) { i = 1; slot = true; }
		}
		const int t = lane & 3;
		nx = x + (t == 2 ? -i : (t == 3 ? i : 0));
		ny = y + (t == 0 ? -i : (t == 1 ? i : 0));
		sprop = slot;
	} else {
		// DepthMap.cpp:1275-1391: 2 causal neighbours propagate, all 4 smooth
		if (lane < 4) {
			const int d = rev ? ((lane + 2) & 3) : lane; // 0 left, 1 up, 2 right, 3 down
			nx = x + (d == 0 ? -1 : (d == 2 ? 1 : 0));
			ny = y + (d == 1 ? -1 : (d == 3 ? 1 : 0));
			slot = d == 0 ? x > kHalfWindow : (d == 1 ? y > kHalfWindow : (d == 2 ? x < W - kHalfWindow : y < H - kHalfWindow));
			sprop = lane < 2;
		}
	}
	C.d = 0.f; C.n0 = C.n1 = C.n2 = 0.f; C.conf = 0.f; C.nx = nx; C.ny = ny;
	if (slot) {
		const float4 v = load_dn(&c.dn[ny * W + nx]);
		C.d = v.x; C.n0 = v.y; C.n1 = v.z; C.n2 = v.w;
		C.conf = load_f(&c.conf[ny * W + nx]);
	}
	const bool closeV = slot && C.d > 0.f;
	C.closeMask = __ballot(closeV);
	C.propMask = __ballot(closeV && sprop);
	{ // Cast<float>(camera.TransformPointI2C(Point3(nx, ndepth))), Camera.h:306-312
		const double z = C.d;
		C.X0 = (float)(((double)nx - c.cx) * z / c.fx);
		C.X1 = (float)(((double)ny - c.cy) * z / c.fy);
		C.X2 = (float)z;
	}

	const int idx = y * W + x;
	const float4 cur = load_dn(&c.dn[idx]);
	float conf = load_f(&c.conf[idx]);
	float depth = cur.x, n0 = cur.y, n1 = cur.z, n2 = cur.w;
	init_plane(G, depth, n0, n1, n2);

	// propagation, DepthMap.cpp:1406-1440
	for (unsigned long long mk = C.propMask; mk;) {
		const int k = __builtin_ctzll(mk);
		mk &= mk - 1ull;
		if (rlf(C.conf, k) >= c.thKeep) continue;
		float kd = rlf(C.d, k), k0 = rlf(C.n0, k), k1 = rlf(C.n1, k), k2 = rlf(C.n2, k);
		const int knx = rli(C.nx, k), kny = rli(C.ny, k);
		kd = interpolate_pixel(c, G, knx, kny, kd, k0, k1, k2);
		correct_normal(G, k0, k1, k2);
		if (lane == k) { C.d = kd; C.n0 = k0; C.n1 = k1; C.n2 = k2; }
		init_plane(G, kd, k0, k1, k2);
		const float nconf = score_pixel<S>(c, L, P, C, G, kd, k0, k1, k2);
		++evals;
		if (conf > nconf) { conf = nconf; depth = kd; n0 = k0; n1 = k1; n2 = k2; }
	}

	// refinement, DepthMap.cpp:1442-1501
	const uint32_t st = (uint32_t)c.itExternal * 64u + 1u + (uint32_t)iter;
	unsigned idxScaleRange = 0;
	bool done = false;
	for (;;) {
		if (conf <= c.thConfSmall) idxScaleRange = 2;
		else if (conf <= c.thConfBig) idxScaleRange = 1;
		else if (conf >= c.thConfRand) {
			bool again = false;
			for (int it = 0; it < c.nRandomIters; ++it) {
				const float nd = random_depth(c, rand_unit(c.seed, (uint32_t)idx, st, 3u * it));
				float r0, r1, r2;
				random_normal(G, rand_unit(c.seed, (uint32_t)idx, st, 3u * it + 1u), rand_unit(c.seed, (uint32_t)idx, st, 3u * it + 2u), r0, r1, r2);
				const float nconf = score_pixel<S>(c, L, P, C, G, nd, r0, r1, r2);
				++evals;
				if (conf > nconf) {
					conf = nconf; depth = nd; n0 = r0; n1 = r1; n2 = r2;
					if (conf < c.thConfRand) { again = true; break; }
				}
			}
			if (again) continue;
			done = true;
		}
		break;
	}
	if (!done) {
		float scaleRange = 1.f / (float)(1u << idxScaleRange);
		const float depthRange = depth * c.depthRatio;
		float p0 = pm_atan2f(n1, n0), p1 = pm_acosf(n2); // Normal2Dir
		for (int it = 0; it < c.nRandomIters; ++it) {
			const uint32_t cb = 64u + 3u * it;
			const float nd = depth + (depthRange * scaleRange) * (2.f * rand_unit(c.seed, (uint32_t)idx, st, cb) - 1.f);
			if (!(c.dMin <= nd && nd < c.dMax)) continue;
			const float np0 = p0 + (c.angle1Range * scaleRange) * (2.f * rand_unit(c.seed, (uint32_t)idx, st, cb + 1u) - 1.f);
			const float np1 = p1 + (c.angle2Range * scaleRange) * (2.f * rand_unit(c.seed, (uint32_t)idx, st, cb + 2u) - 1.f);
			float r0, r1, r2;
			dir2normal(np0, np1, r0, r1, r2);
			if (dot3(r0, r1, r2, G.v0, G.v1, 1.f) >= 0.f) continue;
			init_plane(G, nd, r0, r1, r2);
			const float nconf = score_pixel<S>(c, L, P, C, G, nd, r0, r1, r2);
			++evals;
			if (conf > nconf) {
				conf = nconf; depth = nd; n0 = r0; n1 = r1; n2 = r2;
				p0 = np0; p1 = np1;
				++idxScaleRange;
				scaleRange = 1.f / (float)(1u << idxScaleRange);
			}
		}
	}
	if (lane == 0) {
		store_dn(&c.dn[idx], depth, n0, n1, n2);
		store_f(&c.conf[idx], conf);
	}
}

// ------------------------------------------------------------------------------------------------------
// sweep kernel: persistent row workers (SceneDensify.cpp:677-686 EstimateDepthMapTmp)

__device__ __forceinline__ int wait_progress(int32_t* word, int need, int32_t* err) {
	int v;
	unsigned spins = 0;
	while ((v = __hip_atomic_load(word, __ATOMIC_RELAXED, HC_SCOPE)) < need) {
		__builtin_amdgcn_s_sleep(4);
		++spins;
		if ((spins & 255u) == 0u) {
			if (__hip_atomic_load(err, __ATOMIC_RELAXED, HC_SCOPE) != 0) return -1;
			if (spins > (1u << 22)) { // bounded: give up instead of hanging the device
				__hip_atomic_store(err, 1, __ATOMIC_RELAXED, HC_SCOPE);
				return -1;
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // compiler-only: keep payload loads below the poll
	return v;
}

template <int S>
__global__ __launch_bounds__(64) void sweep_kernel(EstConst c, SweepSync sy, int iter, int lag) {
	LaneCtx<S> L;
	lane_init<S>(c, L);
	const bool rev = (iter & 1) != 0; // dir = RB2LT on odd iterations, DepthMap.cpp:418
	const int nrows = c.H - 2 * kHalfWindow, ncols = c.W - 2 * kHalfWindow;
	unsigned evals = 0;
	for (;;) {
		int r = 0;
		if (L.lane == 0) r = atomicAdd(sy.ticket, 1);
		r = __builtin_amdgcn_readfirstlane(r);
		if (r >= nrows) break;
		const int y = rev ? c.H - 1 - kHalfWindow - r : kHalfWindow + r;
		int32_t* upWord = sy.progress + (size_t)(r > 0 ? r - 1 : 0) * kProgressStride;
		int32_t* myWord = sy.progress + (size_t)r * kProgressStride;
		int known = r > 0 ? 0 : 0x7fffffff; // columns the previous logical row has finished
		bool fail = false;
		if (r > 0) { // start `lag` columns behind so that later polls rarely have to wait
			const int need = lag < ncols ? lag : ncols;
			known = wait_progress(upWord, need, sy.error);
			fail = known < 0;
		}
		for (int q = 0; q < ncols && !fail; ++q) {
			if (known < q + 1) {
				known = wait_progress(upWord, q + 1, sy.error);
				if (known < 0) { fail = true; break; }
			}
			const int x = rev ? c.W - 1 - kHalfWindow - q : kHalfWindow + q;
			process_pixel<S>(c, L, x, y, iter, rev, evals);
			// publish: results must have left the wave before the progress word moves
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (L.lane == 0) __hip_atomic_store(myWord, q + 1, __ATOMIC_RELAXED, HC_SCOPE);
		}
		if (fail) break;
	}
	if (L.lane == 0 && evals) atomicAdd(sy.evals, (unsigned long long)evals);
}

// ------------------------------------------------------------------------------------------------------
// init-score pass (SceneDensify.cpp:649-675 ScoreDepthMapTmp): no inter-pixel dependence

__global__ void import_kernel(EstConst c, const float* depthIn, const float* normalIn) {
	const int n = c.W * c.H;
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const int x = i % c.W, y = i / c.W;
		const bool inb = x >= kHalfWindow && y >= kHalfWindow && x < c.W - kHalfWindow && y < c.H - kHalfWindow;
		if (inb) {
			c.dn[i] = make_float4(depthIn[i], normalIn[3 * i], normalIn[3 * i + 1], normalIn[3 * i + 2]);
			c.conf[i] = 2.f;
		} else {
			c.dn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
			c.conf[i] = 2.f;
		}
	}
}

template <int S>
__global__ __launch_bounds__(256) void score_kernel(EstConst c, unsigned long long* evalsOut) {
	LaneCtx<S> L;
	lane_init<S>(c, L);
	const int nrows = c.H - 2 * kHalfWindow, ncols = c.W - 2 * kHalfWindow;
	const int total = nrows * ncols;
	const int wavesPerBlock = blockDim.x >> 6;
	const int gw = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6), nw = gridDim.x * wavesPerBlock;
	unsigned evals = 0;
	const uint32_t st = (uint32_t)c.itExternal * 64u;
	for (int p = gw; p < total; p += nw) {
		const int x = kHalfWindow + p % ncols, y = kHalfWindow + p / ncols;
		const int idx = y * c.W + x;
		Patch<S> P;
		fill_patch<S>(c, L, x, y, P);
		PixelGeom G;
		pixel_geom(c, x, y, G);
		Close C;
		C.closeMask = 0ull; C.propMask = 0ull;
		C.d = C.n0 = C.n1 = C.n2 = C.X0 = C.X1 = C.X2 = C.conf = 0.f; C.nx = C.ny = 0;
		const float4 cur = c.dn[idx];
		float d = cur.x, n0 = cur.y, n1 = cur.z, n2 = cur.w;
		if (!(c.dMin <= d && d < c.dMax)) {
			d = random_depth(c, rand_unit(c.seed, (uint32_t)idx, st, 0u));
			random_normal(G, rand_unit(c.seed, (uint32_t)idx, st, 1u), rand_unit(c.seed, (uint32_t)idx, st, 2u), n0, n1, n2);
		} else if (dot3(n0, n1, n2, G.v0, G.v1, 1.f) >= 0.f) {
			random_normal(G, rand_unit(c.seed, (uint32_t)idx, st, 1u), rand_unit(c.seed, (uint32_t)idx, st, 2u), n0, n1, n2);
		}
		const float s = score_pixel<S>(c, L, P, C, G, d, n0, n1, n2);
		++evals;
		if (L.lane == 0) {
			c.dn[idx] = make_float4(d, n0, n1, n2);
			c.conf[idx] = s;
		}
	}
	if (L.lane == 0 && evals) atomicAdd(evalsOut, (unsigned long long)evals);
}

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

static inline int segments_for(int V) { return V <= 1 ? 64 : (V <= 2 ? 32 : (V <= 4 ? 16 : (V <= 8 ? 8 : 4))); }

void launch_gray_to_u8(const float* gray, uint8_t* out, int n, hipStream_t s) {
	hipLaunchKernelGGL(gray_to_u8_kernel, dim3(2048), dim3(256), 0, s, gray, out, n);
}
void launch_bgr_to_u8(const uint8_t* bgr, uint8_t* out, int n, hipStream_t s) {
	hipLaunchKernelGGL(bgr_to_u8_kernel, dim3(2048), dim3(256), 0, s, bgr, out, n);
}
void launch_gradient_map(const uint8_t* g8, uint8_t* gra, int W, int H, hipStream_t s) {
	hipLaunchKernelGGL(gradient_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, g8, gra, W, H);
}
void launch_median3(const float* in, float* out, int W, int H, hipStream_t s) {
	hipLaunchKernelGGL(median3_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, in, out, W, H);
}

void launch_score_pass(const EstConst& c, const float* depthIn, const float* normalIn, unsigned long long* evals,
                       hipStream_t s) {
	hipLaunchKernelGGL(import_kernel, dim3(2048), dim3(256), 0, s, c, depthIn, normalIn);
	const dim3 grid(4096), block(256);
	switch (segments_for(c.V)) {
	case 64: hipLaunchKernelGGL(score_kernel<64>, grid, block, 0, s, c, evals); break;
	case 32: hipLaunchKernelGGL(score_kernel<32>, grid, block, 0, s, c, evals); break;
	case 16: hipLaunchKernelGGL(score_kernel<16>, grid, block, 0, s, c, evals); break;
	case 8: hipLaunchKernelGGL(score_kernel<8>, grid, block, 0, s, c, evals); break;
	default: hipLaunchKernelGGL(score_kernel<4>, grid, block, 0, s, c, evals); break;
	}
}

void launch_sweep(const EstConst& c, const SweepSync& sync, int iter, int lag, hipStream_t s) {
	const int nrows = c.H - 2 * kHalfWindow;
	// one wave per workgroup; rows beyond the resident set are picked up through the ticket
	int grid = nrows < 8192 ? nrows : 8192;
	if (grid < 1) return;
	const dim3 g(grid), b(64);
	switch (segments_for(c.V)) {
	case 64: hipLaunchKernelGGL(sweep_kernel<64>, g, b, 0, s, c, sync, iter, lag); break;
	case 32: hipLaunchKernelGGL(sweep_kernel<32>, g, b, 0, s, c, sync, iter, lag); break;
	case 16: hipLaunchKernelGGL(sweep_kernel<16>, g, b, 0, s, c, sync, iter, lag); break;
	case 8: hipLaunchKernelGGL(sweep_kernel<8>, g, b, 0, s, c, sync, iter, lag); break;
	default: hipLaunchKernelGGL(sweep_kernel<4>, g, b, 0, s, c, sync, iter, lag); break;
	}
}

void launch_end_pass(const EstConst& c, int finalPass, float* depth, float* normal, float* conf, hipStream_t s) {
	hipLaunchKernelGGL(end_kernel, dim3(2048), dim3(256), 0, s, c, finalPass, depth, normal, conf);
}

} // namespace hcmvs
