// activate.hip -- fused frame transform + activations (forward and backward), one launch each.
//
// Replaces the ~15 elementwise / matmul torch kernels (and as many autograd nodes) the reference issues per
// render to build the rasteriser inputs:
//   transform_to_frame            src/mapper/splatam/utils/slam_helpers.py:252-304
//       means_cam = R(q_cam) p + t_cam ;  rot = quat_mult(q_cam, normalize(q))   (anisotropic; isotropic keeps q)
//   transformed_params2rendervar  slam_helpers.py:124-139
//       rotations = normalize(rot) ; opacities = sigmoid(logit) ; scales = exp(log_scales) (tiled x3 if isotropic)
// q_cam is the already-normalised camera quaternion (w,x,y,z); camera gradients are not produced (the mapper
// runs with camera_grad=False, slam_helpers.py:270-271).  HBM-bound streaming: 14 floats in, 11 out per Gaussian.
#include "gs_common.h"

namespace gs {

struct Pose { float q[4]; float t[3]; };

__global__ __launch_bounds__(kBlock) void activate_forward_kernel(int P, int iso, Pose pose, const float* __restrict__ means3D,
                                                                  const float* __restrict__ rots, const float* __restrict__ logit_op,
                                                                  const float* __restrict__ log_scales, float* __restrict__ o_means,
                                                                  float* __restrict__ o_rots, float* __restrict__ o_op,
                                                                  float* __restrict__ o_scales)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    float R[3][3];
    quat_to_rot(pose.q, R);
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    o_means[3 * i] = R[0][0] * px + R[0][1] * py + R[0][2] * pz + pose.t[0];
    o_means[3 * i + 1] = R[1][0] * px + R[1][1] * py + R[1][2] * pz + pose.t[1];
    o_means[3 * i + 2] = R[2][0] * px + R[2][1] * py + R[2][2] * pz + pose.t[2];
    const float4 q4 = reinterpret_cast<const float4*>(rots)[i];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    float out[4];
    if (iso) {
        const float inv = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
        for (int k = 0; k < 4; k++) out[k] = q[k] * inv;
    } else {
        const float inv = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
        const float u[4] = {q[0] * inv, q[1] * inv, q[2] * inv, q[3] * inv};
        float m[4];
        qmul(pose.q, u, m);
        const float invm = 1.0f / fmaxf(sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3]), 1e-12f);
        for (int k = 0; k < 4; k++) out[k] = m[k] * invm;
    }
    reinterpret_cast<float4*>(o_rots)[i] = make_float4(out[0], out[1], out[2], out[3]);
    o_op[i] = 1.0f / (1.0f + __expf(-logit_op[i]));
    if (iso) {
        const float s = __expf(log_scales[i]);
        o_scales[3 * i] = s; o_scales[3 * i + 1] = s; o_scales[3 * i + 2] = s;
    } else {
        o_scales[3 * i] = __expf(log_scales[3 * i]); o_scales[3 * i + 1] = __expf(log_scales[3 * i + 1]);
        o_scales[3 * i + 2] = __expf(log_scales[3 * i + 2]);
    }
}

// ACC: the four outputs are ADDED to what d_* already hold (gradient accumulation over the keyframes of a batch in the kernel that produces
// the gradient: the separate `grad += new` passes of autograd move 3 x 44 bytes per Gaussian and keyframe)
template <bool ACC>
__global__ __launch_bounds__(kBlock) void activate_backward_kernel(int P, int iso, Pose pose, const float* __restrict__ rots,
                                                                   const float* __restrict__ o_op, const float* __restrict__ o_scales,
                                                                   const float* __restrict__ g_means, const float* __restrict__ g_rots,
                                                                   const float* __restrict__ g_op, const float* __restrict__ g_scales,
                                                                   float* __restrict__ d_means, float* __restrict__ d_rots,
                                                                   float* __restrict__ d_logit, float* __restrict__ d_logs)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    float R[3][3];
    quat_to_rot(pose.q, R);
    const float gx = g_means ? g_means[3 * i] : 0.f, gy = g_means ? g_means[3 * i + 1] : 0.f, gz = g_means ? g_means[3 * i + 2] : 0.f;
    const float m0 = R[0][0] * gx + R[1][0] * gy + R[2][0] * gz, m1 = R[0][1] * gx + R[1][1] * gy + R[2][1] * gz,
                m2 = R[0][2] * gx + R[1][2] * gy + R[2][2] * gz;
    d_means[3 * i] = ACC ? d_means[3 * i] + m0 : m0;
    d_means[3 * i + 1] = ACC ? d_means[3 * i + 1] + m1 : m1;
    d_means[3 * i + 2] = ACC ? d_means[3 * i + 2] + m2 : m2;
    // rotations
    const float4 q4 = reinterpret_cast<const float4*>(rots)[i];
    const float q[4] = {q4.x, q4.y, q4.z, q4.w};
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (g_rots) { const float4 t = reinterpret_cast<const float4*>(g_rots)[i]; g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w; }
    const float nq = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    const float u[4] = {q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq};
    float du[4];
    if (iso) {
        for (int k = 0; k < 4; k++) du[k] = g[k];
    } else {
        float m[4];
        qmul(pose.q, u, m);
        const float nm = fmaxf(sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3]), 1e-12f);
        const float r[4] = {m[0] / nm, m[1] / nm, m[2] / nm, m[3] / nm};
        const float dot = r[0] * g[0] + r[1] * g[1] + r[2] * g[2] + r[3] * g[3];
        float dm[4];
        for (int k = 0; k < 4; k++) dm[k] = (g[k] - r[k] * dot) / nm;
        qmul_bwd_rhs(pose.q, dm, du);
    }
    const float dotu = u[0] * du[0] + u[1] * du[1] + u[2] * du[2] + u[3] * du[3];
    float4 dq = make_float4((du[0] - u[0] * dotu) / nq, (du[1] - u[1] * dotu) / nq, (du[2] - u[2] * dotu) / nq, (du[3] - u[3] * dotu) / nq);
    if (ACC) { const float4 old = reinterpret_cast<const float4*>(d_rots)[i]; dq.x += old.x; dq.y += old.y; dq.z += old.z; dq.w += old.w; }
    reinterpret_cast<float4*>(d_rots)[i] = dq;
    const float o = o_op[i];
    const float dlg = (g_op ? g_op[i] : 0.f) * o * (1.0f - o);
    d_logit[i] = ACC ? d_logit[i] + dlg : dlg;
    const float s0 = g_scales ? g_scales[3 * i] * o_scales[3 * i] : 0.f, s1 = g_scales ? g_scales[3 * i + 1] * o_scales[3 * i + 1] : 0.f,
                s2 = g_scales ? g_scales[3 * i + 2] * o_scales[3 * i + 2] : 0.f;
    if (iso) d_logs[i] = ACC ? d_logs[i] + (s0 + s1 + s2) : s0 + s1 + s2;
    else {
        d_logs[3 * i] = ACC ? d_logs[3 * i] + s0 : s0; d_logs[3 * i + 1] = ACC ? d_logs[3 * i + 1] + s1 : s1;
        d_logs[3 * i + 2] = ACC ? d_logs[3 * i + 2] + s2 : s2;
    }
}

hipError_t launch_activate_forward(int P, int iso, const float* pose7, const float* means3D, const float* rots, const float* logit_op,
                                   const float* log_scales, float* o_means, float* o_rots, float* o_op, float* o_scales, hipStream_t st)
{
    Pose p; for (int k = 0; k < 4; k++) p.q[k] = pose7[k]; for (int k = 0; k < 3; k++) p.t[k] = pose7[4 + k];
    const int nb = (P + kBlock - 1) / kBlock;
    if (nb > 0) hipLaunchKernelGGL(activate_forward_kernel, dim3(nb), dim3(kBlock), 0, st, P, iso, p, means3D, rots, logit_op, log_scales,
                                   o_means, o_rots, o_op, o_scales);
    return hipGetLastError();
}

hipError_t launch_activate_backward(int P, int iso, const float* pose7, const float* rots, const float* o_op, const float* o_scales,
                                    const float* g_means, const float* g_rots, const float* g_op, const float* g_scales, float* d_means,
                                    float* d_rots, float* d_logit, float* d_logs, int accumulate, hipStream_t st)
{
    Pose p; for (int k = 0; k < 4; k++) p.q[k] = pose7[k]; for (int k = 0; k < 3; k++) p.t[k] = pose7[4 + k];
    const int nb = (P + kBlock - 1) / kBlock;
    if (nb > 0 && accumulate)
        hipLaunchKernelGGL(activate_backward_kernel<true>, dim3(nb), dim3(kBlock), 0, st, P, iso, p, rots, o_op, o_scales, g_means, g_rots,
                           g_op, g_scales, d_means, d_rots, d_logit, d_logs);
    else if (nb > 0)
        hipLaunchKernelGGL(activate_backward_kernel<false>, dim3(nb), dim3(kBlock), 0, st, P, iso, p, rots, o_op, o_scales, g_means, g_rots,
                           g_op, g_scales, d_means, d_rots, d_logit, d_logs);
    return hipGetLastError();
}

}  // namespace gs
