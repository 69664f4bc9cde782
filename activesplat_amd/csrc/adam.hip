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

__global__ __launch_bounds__(kBlock) void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, float one_m_b1,
                                                       float b2, float one_m_b2, float step_size, float inv_bc2s, float eps,
                                                       const uint32_t* __restrict__ fail)
{
    if (chain_failed(fail)) return;            // a backward in front of this step timed out (NaN gradients): leave parameters and moments alone
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];     // the gradient is read once: keep it out of L2
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
                       (float)(1.0 - b2), step_size, inv_bc2s, (float)eps, (const uint32_t*)chain_fail_word());
    return hipGetLastError();
}

// ---- multi-tensor variant: all parameter groups of the mapper in ONE launch ------------------------------------
// The five per-Gaussian groups (means3D, rgb, rotations, opacities, scales) are 0.5-2 M elements each: separately
// they are five ~8 us launches that cannot fill the chip; together one launch with a workgroup range per tensor.
struct AdamSlot {
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps;
    unsigned block_begin;            // first workgroup of this tensor
    unsigned blocks;                 // workgroups assigned to it
    int stream;                      // tensor larger than the last-level cache: non-temporal loads / stores
};
constexpr int64_t kAdamStreamBytes = 256ll << 20;     // 256 MiB last-level (Infinity) cache: beyond it, re-use is impossible anyway
struct AdamBatch { AdamSlot s[kAdamMaxTensors]; int count; const uint32_t* fail; };

__global__ __launch_bounds__(kBlock) void adam_multi_kernel(AdamBatch b)
{
    if (chain_failed(b.fail)) return;          // (see adam_kernel)
    int t = 0;
#pragma unroll
    for (int i = 1; i < kAdamMaxTensors; ++i)
        if (i < b.count && blockIdx.x >= b.s[i].block_begin) t = i;
    const AdamSlot& a = b.s[t];
    const unsigned lb = blockIdx.x - a.block_begin;
    const int64_t n4 = a.n >> 2;
    const int64_t stride = (int64_t)a.blocks * kBlock;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
#define GS_ADAM4(pp, gg, mm, vv)                                                                                 \
    do {                                                                                                        \
        adam_elem(pp.x, gg.x, mm.x, vv.x, a.one_m_b1, a.b2, a.one_m_b2, a.step_size, a.inv_bc2s, a.eps);        \
        adam_elem(pp.y, gg.y, mm.y, vv.y, a.one_m_b1, a.b2, a.one_m_b2, a.step_size, a.inv_bc2s, a.eps);        \
        adam_elem(pp.z, gg.z, mm.z, vv.z, a.one_m_b1, a.b2, a.one_m_b2, a.step_size, a.inv_bc2s, a.eps);        \
        adam_elem(pp.w, gg.w, mm.w, vv.w, a.one_m_b1, a.b2, a.one_m_b2, a.step_size, a.inv_bc2s, a.eps);        \
    } while (0)
    // TWO 16-byte pieces per stream and trip: eight loads in flight per lane before the first update (one piece per trip left the kernel
    // at 5.0-5.6 TB/s of its 28 B per element)
    if (a.stream) {
        for (int64_t i = (int64_t)lb * kBlock + threadIdx.x; i < n4; i += 2 * stride) {
            const int64_t j = i + stride;
            const bool two = j < n4;
            float4 pp = load_stream(&p4[i]), gg = load_stream(&g4[i]), mm = load_stream(&m4[i]), vv = load_stream(&v4[i]);
            float4 pq = pp, gq = gg, mq = mm, vq = vv;
            if (two) { pq = load_stream(&p4[j]); gq = load_stream(&g4[j]); mq = load_stream(&m4[j]); vq = load_stream(&v4[j]); }
            GS_ADAM4(pp, gg, mm, vv);
            store_stream(&p4[i], pp); store_stream(&m4[i], mm); store_stream(&v4[i], vv);
            if (two) {
                GS_ADAM4(pq, gq, mq, vq);
                store_stream(&p4[j], pq); store_stream(&m4[j], mq); store_stream(&v4[j], vq);
            }
        }
    } else {
        for (int64_t i = (int64_t)lb * kBlock + threadIdx.x; i < n4; i += 2 * stride) {
            const int64_t j = i + stride;
            const bool two = j < n4;
            float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
            float4 pq = pp, gq = gg, mq = mm, vq = vv;
            if (two) { pq = p4[j]; gq = g4[j]; mq = m4[j]; vq = v4[j]; }
            GS_ADAM4(pp, gg, mm, vv);
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
            if (two) {
                GS_ADAM4(pq, gq, mq, vq);
                p4[j] = pq; m4[j] = mq; v4[j] = vq;
            }
        }
    }
#undef GS_ADAM4
    const int64_t tl = (n4 << 2) + threadIdx.x;
    if (lb == 0 && tl < a.n) adam_elem(a.p[tl], a.g[tl], a.m[tl], a.v[tl], a.one_m_b1, a.b2, a.one_m_b2, a.step_size, a.inv_bc2s, a.eps);
}

hipError_t launch_adam_multi(int count, const GsAdamTensor* t, hipStream_t st)
{
    for (int base = 0; base < count; base += kAdamMaxTensors) {
        AdamBatch b{};
        b.fail = chain_fail_word();
        unsigned next = 0;
        for (int i = base; i < count && i < base + kAdamMaxTensors; ++i) {
            if (t[i].n <= 0) continue;
            AdamSlot& a = b.s[b.count++];
            a.p = t[i].param; a.g = t[i].grad; a.m = t[i].exp_avg; a.v = t[i].exp_avg_sq; a.n = t[i].n;
            const double bc1 = 1.0 - pow(t[i].beta1, (double)t[i].step);
            const double bc2 = 1.0 - pow(t[i].beta2, (double)t[i].step);
            a.one_m_b1 = (float)(1.0 - t[i].beta1); a.b2 = (float)t[i].beta2; a.one_m_b2 = (float)(1.0 - t[i].beta2);
            a.step_size = (float)(t[i].lr / bc1); a.inv_bc2s = (float)(1.0 / sqrt(bc2)); a.eps = (float)t[i].eps;
            int64_t nb = ((t[i].n >> 2) + kBlock - 1) / kBlock;
            if (nb < 1) nb = 1;
            if (nb > 256 * 8) nb = 256 * 8;
            a.block_begin = next; a.blocks = (unsigned)nb;
            a.stream = t[i].n * 28 > kAdamStreamBytes ? 1 : 0;
            next += (unsigned)nb;
        }
        if (b.count == 0) continue;
        hipLaunchKernelGGL(adam_multi_kernel, dim3(next), dim3(kBlock), 0, st, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace gs
