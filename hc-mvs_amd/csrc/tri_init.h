/* hc-mvs_amd/csrc/tri_init.h -- see tri_init.cpp */
#ifndef HCMVS_TRI_INIT_H
#define HCMVS_TRI_INIT_H
namespace hcmvs {
// depth[W*H], normal[W*H*3] (camera space); dMin/dMax: depth range of the projected points (not yet widened).
// avgDepth <= 0: use the mean depth of the points.  Returns false when no point projects in front of the camera.
bool triangulate_init(int W, int H, const double* K, const double* R, const double* C, const float* xyz, int n, float avgDepth,
                      bool addCorners, float* depth, float* normal, float* dMin, float* dMax);
}
#endif
