// densify.hip -- the decision and fix-up halves of the mapper's densify / prune step.
//
// Replaces the mask algebra of src/mapper/splatam/utils/slam_external.py:171-192 (prune_gaussians) and :195-247 (densify):
// there every event is ~40 element-wise torch launches, three torch.cat per tensor and two boolean-index compactions of every
// parameter AND both Adam moments.  Here ONE kernel classifies every Gaussian (clone / split / cull, and whether its clone and
// its split children survive the cull that follows -- they inherit the parent's opacity; the children's scale is the
// parent's / (0.8 n)), the three masks go through the ballot/popcount compaction of compact.hip, and each tensor is then
// produced by ONE row gather (compact.hip, new rows' moments zero-filled by the same launch).  The split children -- a
// contiguous tail of the output -- get their offset R(q) sample and shrunk scale from densify_children_kernel.
//
// Decision rules, exactly the reference's (same fp32 expressions, same comparison directions):
//   grads     = accum / denom, NaN -> 0                                  slam_external.py:204-205
//   clone     = grads >= grad_thresh and max exp(log_scale) <= 0.01 R    :207-208
//   split     = grads >= grad_thresh and max exp(log_scale) >  0.01 R    :219-220 (clones are appended with gradient 0: never split)
//   cull      = sigmoid(logit_opacity) < thr  or (remove_big and max exp(log_scale) > 0.1 R)     :237-243, :177-183
// Output order of the reference: surviving originals (split parents removed), surviving clones, then n blocks of surviving
// children (block c holds child c of every surviving split parent, parents ascending) -- :216, :232, :236.
#include "gs_common.h"

namespace gs {

__global__ __launch_bounds__(kBlock) void densify_classify_kernel(
    int N, int scale_dim, const float* __restrict__ log_scales, const float* __restrict__ logit_op,
    const float* __restrict__ accum, const float* __restrict__ denom, const float* __restrict__ scene_radius, float grad_thresh,
    float opacity_thresh, int remove_big, int n_split, uint8_t* __restrict__ keep_orig, uint8_t* __restrict__ keep_clone,
    uint8_t* __restrict__ keep_child, uint8_t* __restrict__ split_mask)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const float R = scene_radius[0];
    float smax = expf(log_scales[(size_t)i * scale_dim]);
    for (int c = 1; c < scale_dim; c++) smax = fmaxf(smax, expf(log_scales[(size_t)i * scale_dim + c]));
    const float op = 1.0f / (1.0f + expf(-logit_op[i]));
    const bool cull = op < opacity_thresh || (remove_big && smax > 0.1f * R);
    bool clone = false, split = false, child_ok = false;
    if (accum) {                                   // densify; accum == NULL: prune only
        float g = accum[i] / denom[i];
        if (g != g) g = 0.0f;
        const bool hot = g >= grad_thresh;
        clone = hot && smax <= 0.01f * R;
        split = hot && smax > 0.01f * R;
        // a child keeps the parent's opacity; its scale is exp(log(s / (0.8 n))) in the reference -- the same value up to rounding
        const float cs = expf(logf(smax / (0.8f * (float)n_split)));
        child_ok = split && !(op < opacity_thresh || (remove_big && cs > 0.1f * R));
    }
    keep_orig[i] = (!split && !cull) ? 1 : 0;
    if (keep_clone) keep_clone[i] = (clone && !cull) ? 1 : 0;
    if (keep_child) keep_child[i] = child_ok ? 1 : 0;
    if (split_mask) split_mask[i] = split ? 1 : 0;
}

// counter-based generator of the split offsets: splitmix64 of (seed, child row, draw) -> uniforms -> Box-Muller.  Stateless: a child's
// sample depends on the event's seed and its row only (no generator state on the device, no extra launches)
__device__ __forceinline__ uint32_t mix_u32(uint64_t seed, uint64_t ctr)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
__device__ __forceinline__ void normal_pair(uint64_t seed, uint64_t ctr, float& n0, float& n1)
{
    const float u1 = ((float)mix_u32(seed, 2 * ctr) + 1.0f) * 2.3283064365386963e-10f;          // (0, 1]
    const float u2 = (float)mix_u32(seed, 2 * ctr + 1) * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * logf(u1)), ang = 6.283185307179586f * u2;
    n0 = rad * cosf(ang); n1 = rad * sinf(ang);
}

// children rows [0, n_child) of the output tail: means3D += R(unnorm_rotation) * sample, log_scale = log(exp(log_scale) / (0.8 n))
// (slam_external.py:224-230; the quaternion is normalised inside build_rotation, slam_helpers.py:41-60).  samples == NULL: the
// N(0, scale) offsets (slam_external.py:221-224: torch.normal(mean = 0, std = the parent's scale)) are drawn here.
__global__ __launch_bounds__(kBlock) void densify_children_kernel(int n_child, int scale_dim, int n_split, const float* __restrict__ rots,
                                                                   const float* __restrict__ samples, unsigned long long seed,
                                                                   float* __restrict__ means3D, float* __restrict__ log_scales)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_child) return;
    const float4 q = reinterpret_cast<const float4*>(rots)[i];
    const float inv = 1.0f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float r = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
    float sx, sy, sz;
    if (samples) {
        sx = samples[3 * i]; sy = samples[3 * i + 1]; sz = samples[3 * i + 2];
    } else {
        float n0, n1, n2, n3;
        normal_pair(seed, 2ull * (unsigned)i, n0, n1);
        normal_pair(seed, 2ull * (unsigned)i + 1ull, n2, n3);
        const float s0 = expf(log_scales[(size_t)i * scale_dim]);
        const float s1 = scale_dim == 3 ? expf(log_scales[(size_t)i * scale_dim + 1]) : s0;       // anisotropic: per-axis scales
        const float s2 = scale_dim == 3 ? expf(log_scales[(size_t)i * scale_dim + 2]) : s0;
        sx = n0 * s0; sy = n1 * s1; sz = n2 * s2;
    }
    const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    means3D[3 * i] += R00 * sx + R01 * sy + R02 * sz;
    means3D[3 * i + 1] += R10 * sx + R11 * sy + R12 * sz;
    means3D[3 * i + 2] += R20 * sx + R21 * sy + R22 * sz;
    for (int c = 0; c < scale_dim; c++) {
        float* ls = log_scales + (size_t)i * scale_dim + c;
        *ls = logf(expf(*ls) / (0.8f * (float)n_split));
    }
}

hipError_t launch_densify_classify(int N, int scale_dim, const float* log_scales, const float* logit_op, const float* accum,
                                   const float* denom, const float* scene_radius, float grad_thresh, float opacity_thresh,
                                   int remove_big, int n_split, uint8_t* keep_orig, uint8_t* keep_clone, uint8_t* keep_child,
                                   uint8_t* split_mask, hipStream_t st)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(densify_classify_kernel, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, st, N, scale_dim, log_scales, logit_op,
                       accum, denom, scene_radius, grad_thresh, opacity_thresh, remove_big, n_split, keep_orig, keep_clone, keep_child,
                       split_mask);
    return hipGetLastError();
}

hipError_t launch_densify_children(int n_child, int scale_dim, int n_split, const float* rots, const float* samples, uint64_t seed,
                                   float* means3D, float* log_scales, hipStream_t st)
{
    if (n_child <= 0) return hipSuccess;
    hipLaunchKernelGGL(densify_children_kernel, dim3((n_child + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n_child, scale_dim, n_split, rots,
                       samples, (unsigned long long)seed, means3D, log_scales);
    return hipGetLastError();
}

}  // namespace gs
