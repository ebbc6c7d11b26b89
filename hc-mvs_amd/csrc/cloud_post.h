/* hc-mvs_amd/csrc/cloud_post.h -- host-side post-processing of the fused cloud (cloud_post.cpp) */
#ifndef HCMVS_CLOUD_POST_H
#define HCMVS_CLOUD_POST_H
#include <stdint.h>
namespace hcmvs {
// MVS::EstimatePointNormals (DepthMap.cpp:2221-2269): PCA plane normal of the k nearest points, flipped towards viewC (3 doubles per point)
void pca_normals(uint64_t n, const float* xyz, const double* viewC, int k, float* normal);
}
#endif
