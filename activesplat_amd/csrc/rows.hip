// rows.hip -- the keyframe-sharded optimiser step's glue between the per-key tensors and ONE flat [rows, G] fp32 buffer
// (activesplat_amd/parallel.py; SURVEY.md section 8e -- a new capability: the reference is single-GPU and steps one keyframe at a
// time, src/mapper/splatam/__init__.py:450-480).
//
//   pack   : the K per-key gradients ([N, w_k] each) -> flat[N_padded][G], G = sum w_k, padding rows zero: the send buffer of the
//            RCCL reduce-scatter / all-reduce, written by ONE launch (was: zero-fill + K strided copies + a padded copy);
//   adam   : fused Adam on this rank's row block of every key straight from the reduce-scattered [rows, G] shard; the updated
//            rows are also written into out[rows][G], the send buffer of the all-gather (was: K contiguous-slice copies, K Adam
//            launches, K strided copies);
//   unpack : flat[N][G] (all ranks' updated rows) -> the K parameter tensors, one launch.
// Element-wise, HBM-bound: one thread per flat element, consecutive lanes on consecutive columns of a row (the flat side is fully
// coalesced, the per-key side in runs of w_k floats).
#include "gs_common.h"

namespace gs {

struct RowBatch {
    float* p[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    const float* g[kAdamMaxTensors];
    int width[kAdamMaxTensors];
    float one_m_b1[kAdamMaxTensors], b2[kAdamMaxTensors], one_m_b2[kAdamMaxTensors], step_size[kAdamMaxTensors],
        inv_bc2s[kAdamMaxTensors], eps[kAdamMaxTensors];
    unsigned char col_tensor[64], col_off[64];      // flat column -> (tensor, column inside it)
    int G;
};

// MODE 0: pack, 1: unpack, 2: adam
template <int MODE>
__global__ __launch_bounds__(kBlock) void rows_kernel(RowBatch b, int64_t row0, int64_t n_valid, int64_t n_rows, const float* __restrict__ in,
                                                       float* __restrict__ out)
{
    // (row, column) of this thread's element advance by the grid stride without a division per element
    const int64_t total = n_rows * b.G, stride = (int64_t)gridDim.x * kBlock;
    const int64_t e0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t dr = stride / b.G;
    const int dc = (int)(stride - dr * b.G);
    int64_t r = e0 / b.G;
    int col = (int)(e0 - r * b.G);
    for (int64_t e = e0; e < total; e += stride, r += dr, col += dc) {
        if (col >= b.G) { col -= b.G; ++r; }
        const int t = b.col_tensor[col], c = b.col_off[col];
        const int64_t k = (row0 + r) * b.width[t] + c;            // element of tensor t
        if (MODE == 0) {
            out[e] = (r < n_valid && b.g[t]) ? b.g[t][k] : 0.0f;
        } else if (MODE == 1) {
            if (r < n_valid) b.p[t][k] = in[e];
        } else {
            float pv = 0.0f;
            if (r < n_valid) {
                pv = b.p[t][k];
                float mv = b.m[t][k], vv = b.v[t][k];
                adam_elem(pv, in[e], mv, vv, b.one_m_b1[t], b.b2[t], b.one_m_b2[t], b.step_size[t], b.inv_bc2s[t], b.eps[t]);
                b.p[t][k] = pv; b.m[t][k] = mv; b.v[t][k] = vv;
            }
            if (out) out[e] = pv;
        }
    }
}

hipError_t launch_rows(int mode, int count, const GsRowTensor* t, int64_t row0, int64_t n_valid, int64_t n_rows, const float* in, float* out,
                       hipStream_t st)
{
    RowBatch b{};
    int G = 0;
    for (int i = 0; i < count; ++i) {
        b.p[i] = t[i].param; b.m[i] = t[i].exp_avg; b.v[i] = t[i].exp_avg_sq; b.g[i] = t[i].grad; b.width[i] = t[i].width;
        if (mode == 2) {
            const double bc1 = 1.0 - pow(t[i].beta1, (double)t[i].step), bc2 = 1.0 - pow(t[i].beta2, (double)t[i].step);
            b.one_m_b1[i] = (float)(1.0 - t[i].beta1); b.b2[i] = (float)t[i].beta2; b.one_m_b2[i] = (float)(1.0 - t[i].beta2);
            b.step_size[i] = (float)(t[i].lr / bc1); b.inv_bc2s[i] = (float)(1.0 / sqrt(bc2)); b.eps[i] = (float)t[i].eps;
        }
        for (int c = 0; c < t[i].width; ++c) { b.col_tensor[G] = (unsigned char)i; b.col_off[G] = (unsigned char)c; ++G; }
    }
    b.G = G;
    const int64_t total = n_rows * G;
    if (total <= 0) return hipSuccess;
    int64_t nb = (total + kBlock - 1) / kBlock;
    if (nb > 256 * 16) nb = 256 * 16;
    if (mode == 0) hipLaunchKernelGGL(rows_kernel<0>, dim3((unsigned)nb), dim3(kBlock), 0, st, b, row0, n_valid, n_rows, in, out);
    else if (mode == 1) hipLaunchKernelGGL(rows_kernel<1>, dim3((unsigned)nb), dim3(kBlock), 0, st, b, row0, n_valid, n_rows, in, out);
    else hipLaunchKernelGGL(rows_kernel<2>, dim3((unsigned)nb), dim3(kBlock), 0, st, b, row0, n_valid, n_rows, in, out);
    return hipGetLastError();
}

}  // namespace gs
