// adam.hip -- fused dense Adam step over one flat fp32 parameter tensor.
//
// Replaces the multi-kernel torch.optim.Adam foreach step the reference runs per parameter group
// (src/mapper/splatam/splatam.py:118-124: Adam(param_groups, lr=0.0, eps=1e-15), betas (0.9,0.999),
// no weight decay, non-amsgrad; stepped at src/mapper/splatam/__init__.py:479).  Arithmetic follows
// torch's single-tensor path: m.lerp_(g, 1-b1); v = b2 v + (1-b2) g g;
// p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
// Pure HBM streaming: 16 B/lane loads, 28 B of traffic per element (read p,g,m,v; write p,m,v).
#include "gs_common.h"

namespace gs {

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float one_m_b1, float b2, float one_m_b2,
                                          float step_size, float inv_bc2s, float eps)
{
    m = m + one_m_b1 * (g - m);
    v = b2 * v + one_m_b2 * g * g;
    const float denom = sqrtf(v) * inv_bc2s + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(kBlock) void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, float one_m_b1,
                                                       float b2, float one_m_b2, float step_size, float inv_bc2s, float eps)
{
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adam_elem(pp.x, gg.x, mm.x, vv.x, one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps);
        adam_elem(pp.y, gg.y, mm.y, vv.y, one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps);
        adam_elem(pp.z, gg.z, mm.z, vv.z, one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps);
        adam_elem(pp.w, gg.w, mm.w, vv.w, one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n not a multiple of 4)
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x == 0 && t < n) adam_elem(p[t], g[t], m[t], v[t], one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps);
}

hipError_t launch_adam(int64_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2,
                       double eps, int step, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    // bias corrections on the host in double, as torch does for python-scalar steps
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float inv_bc2s = (float)(1.0 / sqrt(bc2));
    int64_t nb = ((n >> 2) + kBlock - 1) / kBlock;
    if (nb < 1) nb = 1;
    if (nb > 256 * 8) nb = 256 * 8;          // grid-stride: 8 workgroups per CU
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(kBlock), 0, st, n, p, g, m, v, (float)(1.0 - b1), (float)b2,
                       (float)(1.0 - b2), step_size, inv_bc2s, (float)eps);
    return hipGetLastError();
}

}  // namespace gs
