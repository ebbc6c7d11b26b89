/*
 * hc-mvs_amd/csrc/tri_init.cpp -- rough depth / normal maps from the sparse points by Delaunay triangulation
 * (reference: TriangulatePointsDelaunay + TriangulatePoints2DepthMap, frame_main/libs/MVS/DepthMap.cpp:1796-1936;
 * rasteriser TImage::RasterizeTriangle, frame_main/libs/Common/Types.inl:2474-2606).  Host code: the reference does
 * this step with CGAL on the CPU once per image, before the first sweep; here the triangulation is a plain
 * Bowyer-Watson insertion that starts from the image rectangle (the four corner support points of
 * OPTDENSE::bAddCorners are its convex hull), so no CGAL and no points at infinity are needed.
 */
#include "tri_init.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

namespace hcmvs {

namespace {

struct P3 { double x, y, z; };
struct Tri { int v[3]; };

inline double orient(const P3& a, const P3& b, const P3& c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

// > 0 when d lies inside the circumcircle of the counter-clockwise triangle a b c
inline long double in_circle(const P3& a, const P3& b, const P3& c, const P3& d) {
	const long double ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
	return (ax * ax + ay * ay) * (bx * cy - cx * by) - (bx * bx + by * by) * (ax * cy - cx * ay) + (cx * cx + cy * cy) * (ax * by - bx * ay);
}

// Delaunay triangulation of pts[4..) inside the rectangle spanned by pts[0..4) (already in the list, counter-clockwise).
// Incremental Bowyer-Watson: the points go in along a serpentine grid order, the triangle holding a point is found by
// a visibility walk from the last triangle made, the cavity by a flood fill over edge neighbours.
void bowyer_watson(const std::vector<P3>& pts, std::vector<Tri>& out) {
	struct T3 { int v[3]; bool alive; };
	std::vector<T3> tris;
	std::unordered_map<uint64_t, int> owner; // directed edge a->b  ->  the (counter-clockwise) triangle that has it
	auto key = [](int a, int b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; };
	auto add = [&](int a, int b, int c) {
		const int t = (int)tris.size();
		tris.push_back({{a, b, c}, true});
		owner[key(a, b)] = t; owner[key(b, c)] = t; owner[key(c, a)] = t;
		return t;
	};
	add(0, 1, 2); add(0, 2, 3);
	const int n = (int)pts.size();
	std::vector<int> order;
	for (int i = 4; i < n; ++i) order.push_back(i);
	{ // serpentine order over a grid of ~sqrt(n) cells per side keeps consecutive insertions close to each other
		const double x0 = pts[0].x, y0 = pts[0].y, w = pts[2].x - pts[0].x, h = pts[2].y - pts[0].y;
		const int g = std::max(1, (int)std::sqrt((double)n / 2));
		auto cell = [&](int i) {
			const int cx = std::min(g - 1, std::max(0, (int)((pts[i].x - x0) / w * g))), cy = std::min(g - 1, std::max(0, (int)((pts[i].y - y0) / h * g)));
			return cy * g + ((cy & 1) ? g - 1 - cx : cx);
		};
		std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return cell(l) < cell(r); });
	}
	std::vector<int> cavity, stack;
	std::vector<std::pair<int, int>> edges;
	std::vector<char> bad;
	int last = 1;
	for (int p : order) {
		// visibility walk to the triangle that holds p
		int t = last;
		while (!tris[t].alive) --t;
		for (int guard = 0; guard < (int)tris.size() + 8; ++guard) {
			bool moved = false;
			for (int e = 0; e < 3 && !moved; ++e) {
				const int a = tris[t].v[e], b = tris[t].v[(e + 1) % 3];
				if (orient(pts[a], pts[b], pts[p]) < 0) {
					auto it = owner.find(key(b, a));
					if (it != owner.end()) { t = it->second; moved = true; }
				}
			}
			if (!moved) break;
		}
		if (!(in_circle(pts[tris[t].v[0]], pts[tris[t].v[1]], pts[tris[t].v[2]], pts[p]) > 0)) continue; // numerically degenerate: leave it out
		// flood fill the cavity: triangles whose circumcircle holds p
		bad.assign(tris.size(), 0);
		cavity.clear(); stack.clear(); edges.clear();
		stack.push_back(t); bad[t] = 1;
		while (!stack.empty()) {
			const int u = stack.back(); stack.pop_back();
			cavity.push_back(u);
			for (int e = 0; e < 3; ++e) {
				const int a = tris[u].v[e], b = tris[u].v[(e + 1) % 3];
				auto it = owner.find(key(b, a));
				if (it == owner.end()) { edges.push_back({a, b}); continue; } // hull edge
				const int g = it->second;
				if (bad[g] == 1) continue;
				if (bad[g] == 0 && in_circle(pts[tris[g].v[0]], pts[tris[g].v[1]], pts[tris[g].v[2]], pts[p]) > 0) { bad[g] = 1; stack.push_back(g); }
				else { bad[g] = 2; edges.push_back({a, b}); }
			}
		}
		for (int u : cavity) {
			tris[u].alive = false;
			for (int e = 0; e < 3; ++e) owner.erase(key(tris[u].v[e], tris[u].v[(e + 1) % 3]));
		}
		for (const auto& e : edges)
			if (orient(pts[e.first], pts[e.second], pts[p]) > 0) last = add(e.first, e.second, p); // skips flat fans on the border
	}
	out.clear();
	for (const T3& t : tris) if (t.alive) out.push_back({{t.v[0], t.v[1], t.v[2]}});
}

inline void i2c(const double* K, double x, double y, double z, double* o) { // Camera.h:306-312
	o[0] = (x - K[2]) * z / K[0]; o[1] = (y - K[5]) * z / K[4]; o[2] = z;
}

} // namespace

bool triangulate_init(int W, int H, const double* K, const double* R, const double* C, const float* xyz, int n, float avgDepth,
                      bool addCorners, float* depth, float* normal, float* dMin, float* dMax) {
	// DepthMap.cpp:1796-1808: project, keep (x/z, y/z, z); CGAL keeps the first of two points at the same position
	double Pm[12];
	for (int r = 0; r < 3; ++r) {
		double KR[3];
		for (int c = 0; c < 3; ++c) KR[c] = K[r * 3] * R[c] + K[r * 3 + 1] * R[3 + c] + K[r * 3 + 2] * R[6 + c];
		for (int c = 0; c < 3; ++c) Pm[r * 4 + c] = KR[c];
		Pm[r * 4 + 3] = -(KR[0] * C[0] + KR[1] * C[1] + KR[2] * C[2]);
	}
	// scaffold: the image rectangle (support points of bAddCorners), or a much larger one whose triangles are dropped
	const double m = addCorners ? 0.0 : 16.0 * std::max(W, H);
	std::vector<P3> pts = {{-m, -m, 0}, {W + m, -m, 0}, {W + m, H + m, 0}, {-m, H + m, 0}};
	float lo = FLT_MAX, hi = 0.f;
	double sumDepth = 0;
	std::map<std::pair<double, double>, int> seen;
	for (int i = 0; i < n; ++i) {
		const double X = xyz[3 * i], Y = xyz[3 * i + 1], Z = xyz[3 * i + 2];
		const float px = (float)(Pm[0] * X + Pm[1] * Y + Pm[2] * Z + Pm[3]), py = (float)(Pm[4] * X + Pm[5] * Y + Pm[6] * Z + Pm[7]),
		            pz = (float)(Pm[8] * X + Pm[9] * Y + Pm[10] * Z + Pm[11]);
		if (!(pz > 0.f)) continue;
		const P3 q = {(double)(px / pz), (double)(py / pz), (double)pz};
		if (lo > pz) lo = pz;
		if (hi < pz) hi = pz;
		sumDepth += pz;
		if (addCorners && (q.x < 0 || q.y < 0 || q.x > W || q.y > H)) continue; // outside the hull of the support points
		if (!seen.insert({{q.x, q.y}, (int)pts.size()}).second) continue;
		pts.push_back(q);
	}
	const int nUsed = (int)pts.size() - 4;
	if (nUsed < 1) return false;
	if (!(avgDepth > 0.f)) avgDepth = (float)(sumDepth / std::max(1, n)); // Scene.cpp:565-603 (mean depth of the image's points)
	for (int k = 0; k < 4; ++k) pts[k].z = avgDepth;
	std::vector<Tri> tris;
	bowyer_watson(pts, tris);

	if (addCorners) { // DepthMap.cpp:1810-1876: corner depth from the three closest faces behind its incident faces
		std::map<std::pair<int, int>, std::vector<int>> edgeFaces;
		for (int t = 0; t < (int)tris.size(); ++t)
			for (int e = 0; e < 3; ++e) {
				const int a = tris[t].v[e], b = tris[t].v[(e + 1) % 3];
				edgeFaces[{std::min(a, b), std::max(a, b)}].push_back(t);
			}
		for (int corner = 0; corner < 4; ++corner) {
			const P3 A = pts[corner];
			double rayA[3];
			i2c(K, A.x, A.y, A.z, rayA);
			const double len = std::sqrt(rayA[0] * rayA[0] + rayA[1] * rayA[1] + rayA[2] * rayA[2]);
			for (double& r : rayA) r /= len;
			std::vector<std::pair<float, float>> cand; // (1 / distance, depth)
			for (int t = 0; t < (int)tris.size(); ++t) {
				int ci = -1;
				for (int e = 0; e < 3; ++e) if (tris[t].v[e] == corner) ci = e;
				if (ci < 0) continue;
				const int a = tris[t].v[(ci + 1) % 3], b = tris[t].v[(ci + 2) % 3];
				const auto& fs = edgeFaces[{std::min(a, b), std::max(a, b)}];
				int g = -1;
				for (int f : fs) if (f != t) g = f;
				if (g < 0) continue; // hull edge: the infinite face
				const Tri& G = tris[g];
				if (G.v[0] < 4 || G.v[1] < 4 || G.v[2] < 4) continue; // faces touching a corner are not used
				double c0[3], c1[3], c2[3];
				i2c(K, pts[G.v[0]].x, pts[G.v[0]].y, pts[G.v[0]].z, c0);
				i2c(K, pts[G.v[1]].x, pts[G.v[1]].y, pts[G.v[1]].z, c1);
				i2c(K, pts[G.v[2]].x, pts[G.v[2]].y, pts[G.v[2]].z, c2);
				const double e1[3] = {c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2]}, e2[3] = {c2[0] - c0[0], c2[1] - c0[1], c2[2] - c0[2]};
				double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
				const double nl = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
				for (double& v : nn) v /= nl;
				const double denom = nn[0] * rayA[0] + nn[1] * rayA[1] + nn[2] * rayA[2];
				const double tt = (nn[0] * c0[0] + nn[1] * c0[1] + nn[2] * c0[2]) / denom; // ray from the origin
				const double z = rayA[2] * tt;
				if (!(z > 0)) continue;
				const double bx = (pts[G.v[0]].x + pts[G.v[1]].x + pts[G.v[2]].x) / 3.f, by = (pts[G.v[0]].y + pts[G.v[1]].y + pts[G.v[2]].y) / 3.f;
				const double dist = std::sqrt((bx - A.x) * (bx - A.x) + (by - A.y) * (by - A.y));
				const float zc = std::min(std::max((float)z, lo), hi);
				cand.push_back({1.f / (float)dist, zc});
			}
			if (cand.size() < 3) continue; // "normally this should never happen": the corner keeps the average depth
			std::stable_sort(cand.begin(), cand.end(), [](const std::pair<float, float>& l, const std::pair<float, float>& r) { return l.first > r.first; });
			const float s = cand[0].first + cand[1].first + cand[2].first;
			const float inv = 1.f / s;
			float w[3] = {cand[0].first * inv, cand[1].first * inv, cand[2].first * inv};
			pts[corner].z = (double)(cand[0].second * w[0] + cand[1].second * w[1] + cand[2].second * w[2]);
		}
	}

	// DepthMap.cpp:1886-1931: one plane per face, rasterised with the 28.4 fixed-point half-space rule
	std::fill(depth, depth + (size_t)W * H, 0.f);
	std::fill(normal, normal + (size_t)W * H * 3, 0.f);
	const float fx = (float)K[0], fy = (float)K[4], cx = (float)K[2], cy = (float)K[5];
	for (const Tri& T : tris) {
		if (!addCorners && (T.v[0] < 4 || T.v[1] < 4 || T.v[2] < 4)) continue;
		float c[3][3];
		for (int k = 0; k < 3; ++k) { // Point3f(i_k) then TransformPointI2C
			const float x = (float)pts[T.v[k]].x, y = (float)pts[T.v[k]].y, z = (float)pts[T.v[k]].z;
			c[k][0] = (float)(((double)x - K[2]) * (double)z / K[0]); c[k][1] = (float)(((double)y - K[5]) * (double)z / K[4]); c[k][2] = z;
		}
		const float e1[3] = {c[1][0] - c[0][0], c[1][1] - c[0][1], c[1][2] - c[0][2]}, e2[3] = {c[2][0] - c[0][0], c[2][1] - c[0][1], c[2][2] - c[0][2]};
		float nn[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]}; // edge2 x edge1
		const float nl = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
		if (!(nl > 0.f)) continue;
		for (float& v : nn) v /= nl;
		const float dn = 1.f / (nn[0] * c[0][0] + nn[1] * c[0][1] + nn[2] * c[0][2]);
		const float np[3] = {nn[0] * dn, nn[1] * dn, nn[2] * dn};
		// the reference passes the counter-clockwise face reversed: (v2, v1, v0)
		const P3 &v1 = pts[T.v[2]], &v2 = pts[T.v[1]], &v3 = pts[T.v[0]];
		auto r16 = [](double v) { return (int64_t)std::floor(16.0 * (double)(float)v + 0.5); }; // ROUND2INT(16 * v), v as float
		const int64_t Y1 = r16(v1.y), Y2 = r16(v2.y), Y3 = r16(v3.y), X1 = r16(v1.x), X2 = r16(v2.x), X3 = r16(v3.x);
		const int64_t DX12 = X1 - X2, DX23 = X2 - X3, DX31 = X3 - X1, DY12 = Y1 - Y2, DY23 = Y2 - Y3, DY31 = Y3 - Y1;
		int minx = (int)((std::min({X1, X2, X3}) + 0xF) >> 4), maxx = (int)((std::max({X1, X2, X3}) + 0xF) >> 4);
		int miny = (int)((std::min({Y1, Y2, Y3}) + 0xF) >> 4), maxy = (int)((std::max({Y1, Y2, Y3}) + 0xF) >> 4);
		minx &= ~7; miny &= ~7;
		int64_t C1 = DY12 * X1 - DX12 * Y1, C2 = DY23 * X2 - DX23 * Y2, C3 = DY31 * X3 - DX31 * Y3;
		if (DY12 < 0 || (DY12 == 0 && DX12 > 0)) C1++;
		if (DY23 < 0 || (DY23 == 0 && DX23 > 0)) C2++;
		if (DY31 < 0 || (DY31 == 0 && DX31 > 0)) C3++;
		const int yEnd = miny + ((maxy - miny + 7) / 8) * 8, xEnd = minx + ((maxx - minx + 7) / 8) * 8; // whole 8x8 blocks
		for (int y = std::max(miny, 0); y < std::min(yEnd, H); ++y)
			for (int x = std::max(minx, 0); x < std::min(xEnd, W); ++x) {
				const int64_t xs = (int64_t)x << 4, ys = (int64_t)y << 4;
				if (!(C1 + DX12 * ys - DY12 * xs > 0 && C2 + DX23 * ys - DY23 * xs > 0 && C3 + DX31 * ys - DY31 * xs > 0)) continue;
				const float X0x = ((float)x - cx) / fx, X0y = ((float)y - cy) / fy; // TransformPointI2C(Point2f)
				const float z = 1.f / (np[0] * X0x + np[1] * X0y + np[2] * 1.f);
				if (!(z > 0.f)) continue; // "due to numerical instability"
				depth[(size_t)y * W + x] = z;
				float* o = normal + 3 * ((size_t)y * W + x);
				o[0] = nn[0]; o[1] = nn[1]; o[2] = nn[2];
			}
	}
	*dMin = lo; *dMax = hi;
	return true;
}

} // namespace hcmvs
