// stats.hip -- per-Gaussian densification statistics, one launch each instead of strings of element-wise torch kernels.
//
//  visibility_stats : seen = radii > 0 ; max_2D_radius = max(max_2D_radius, radii)
//                     (src/mapper/splatam/splatam.py:296-298: `seen = radius > 0`, `max_2D_radius[seen] = max(radius[seen], ...)`;
//                      a radius of 0 never raises the maximum, so the masked and the unmasked update are the same)
//  accumulate_grad2d: means2D_gradient_accum[seen] += || means2D.grad[seen, :2] || ; denom[seen] += 1
//                     (src/mapper/splatam/utils/slam_external.py:100-108)
// Pure streaming, 4-16 B per Gaussian.
#include "gs_common.h"

namespace gs {

__global__ __launch_bounds__(kBlock) void visibility_stats_kernel(int P, const int32_t* __restrict__ radii, uint8_t* __restrict__ seen,
                                                                   float* __restrict__ max_radius)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const int32_t r = radii[i];
    if (seen) seen[i] = r > 0 ? 1 : 0;
    if (max_radius) max_radius[i] = fmaxf(max_radius[i], (float)r);
}

__global__ __launch_bounds__(kBlock) void accumulate_grad2d_kernel(int P, const float* __restrict__ grad, const uint8_t* __restrict__ seen,
                                                                    float* __restrict__ accum, float* __restrict__ denom)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P || !seen[i]) return;
    const float gx = grad[3 * (size_t)i], gy = grad[3 * (size_t)i + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}

hipError_t launch_visibility_stats(int P, const int32_t* radii, uint8_t* seen, float* max_radius, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(visibility_stats_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, st, P, radii, seen, max_radius);
    return hipGetLastError();
}

hipError_t launch_accumulate_grad2d(int P, const float* grad, const uint8_t* seen, float* accum, float* denom, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(accumulate_grad2d_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, st, P, grad, seen, accum, denom);
    return hipGetLastError();
}

}  // namespace gs
