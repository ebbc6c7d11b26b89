/* hc-mvs_amd/csrc/cloud_kernels.h -- post-processing of the fused cloud on the device (cloud_kernels.hip) */
#ifndef HCMVS_CLOUD_KERNELS_H
#define HCMVS_CLOUD_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
namespace hcmvs {
// MVS::EstimatePointNormals (DepthMap.cpp:2221-2269): PCA plane normal of the k nearest points, flipped towards the camera centre of
// the point's first view (firstView[i] indexes viewC, 3 doubles per view).  Host buffers in and out; 0 = ok, 1 = bad argument,
// 2 = device failure (err says which)
int pca_normals_device(unsigned long long n, const float* xyz, const uint32_t* firstView, const double* viewC, size_t nViews, int k, float* normal,
                       hipStream_t s, std::string& err);
}
#endif
