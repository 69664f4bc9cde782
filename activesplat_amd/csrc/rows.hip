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
// Element-wise, HBM-bound.  Round 5: a workgroup takes 256 rows and walks the keys one after the other; for key k its 256 x w_k floats are ONE
// contiguous run of the per-key tensor (thread j takes element j of the run: fully coalesced on the side that has five to six separate tensors, for
// p / m / v of the Adam mode as well), the flat side is touched in runs of w_k floats per row, and the rows of the block are completed by the passes
// of the other keys while their lines are still in the L2.  No per-element table lookups: the key loop is uniform, so widths and base pointers sit in
// scalar registers.  (Rounds 3-4: one thread per FLAT element with byte tables indexed per lane from the kernel-argument struct -- five dependent
// vector-memory instructions of bookkeeping per element: pack 130 us, unpack 99 us, Adam on all rows 465 us at [2 M, 14], i.e. 1.7-2.3 TB/s.)
#include "gs_common.h"

namespace gs {

struct RowBatch {
    float* p[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    const float* g[kAdamMaxTensors];
    int width[kAdamMaxTensors];
    float inv_width[kAdamMaxTensors];
    float one_m_b1[kAdamMaxTensors], b2[kAdamMaxTensors], one_m_b2[kAdamMaxTensors], step_size[kAdamMaxTensors],
        inv_bc2s[kAdamMaxTensors], eps[kAdamMaxTensors];
    int count, G;
    const uint32_t* fail;      // Cam::chain_fail (MODE 2): set -> rows pass through unstepped
};

// MODE 0: pack, 1: unpack, 2: adam.  ROWS x GMAX floats of LDS hold the block's tile of the flat buffer: BOTH sides of every copy are contiguous runs in
// global memory (the flat side: ROWS x G floats read / written in one sweep; the per-key side: ROWS x w_k floats per key), the transposition between
// the two happens in LDS.  (ROWS, GMAX) = (256, 16) for the reference's 14 floats per Gaussian, (64, 64) for rows with SH coefficients (59).
template <int MODE, int ROWS, int GMAX>
__global__ __launch_bounds__(kBlock) void rows_kernel(RowBatch b, int64_t row0, int64_t n_valid, int64_t n_rows, const float* __restrict__ in,
                                                       float* __restrict__ out)
{
    __shared__ float tile[ROWS * GMAX];
    const int tid = threadIdx.x;
    const int G = b.G;
    const bool step_ok = MODE != 2 || !chain_failed(b.fail);   // (a backward in front of this step timed out: the rows pass through unstepped)
    const int64_t nblocks = (n_rows + ROWS - 1) / ROWS;
    for (int64_t rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
        const int64_t r0 = rb * ROWS;
        const int nr = (int)min((int64_t)ROWS, n_rows - r0);
        const int nflat = nr * G;
        const int live = (int)max((int64_t)0, min((int64_t)nr, n_valid - r0));      // rows of this block that exist in the tensors (the rest: padding)
        if (MODE != 0) {                                           // unpack / adam: the block's tile of the flat input, one contiguous sweep
            const float* src = in + r0 * G;
            for (int j = tid; j < nflat; j += kBlock) tile[j] = src[j];
        }
        __syncthreads();
        int off = 0;
        for (int t = 0; t < b.count; ++t) {                       // uniform: widths and pointers are scalars
            const int w = b.width[t];
            const float iw = b.inv_width[t];
            const int n = live * w;                                // the key's run for this block: contiguous in its tensor
            const int64_t kbase = (row0 + r0) * w;
            for (int j = tid; j < nr * w; j += kBlock) {
                const int row = (int)(((float)j + 0.5f) * iw);     // j / w for j < 256 * 64 (exact: the product is at least 0.5 / w away from an integer)
                const int c = j - row * w;
                float* cell = tile + row * G + off + c;
                const int64_t k = kbase + j;
                if (MODE == 0) {
                    *cell = (j < n && b.g[t]) ? b.g[t][k] : 0.0f;
                } else if (MODE == 1) {
                    if (j < n) b.p[t][k] = *cell;
                } else {
                    float pv = 0.0f;
                    if (j < n) {
                        pv = b.p[t][k];
                        if (step_ok) {
                            float mv = b.m[t][k], vv = b.v[t][k];
                            adam_elem(pv, *cell, mv, vv, b.one_m_b1[t], b.b2[t], b.one_m_b2[t], b.step_size[t], b.inv_bc2s[t], b.eps[t]);
                            b.p[t][k] = pv; b.m[t][k] = mv; b.v[t][k] = vv;
                        }
                    }
                    *cell = pv;                                    // (each cell is read and rewritten by the one thread that owns it)
                }
            }
            off += w;
        }
        __syncthreads();
        if (MODE == 0 || (MODE == 2 && out)) {                     // pack / adam: the tile leaves as one contiguous sweep
            float* dst = out + r0 * G;
            for (int j = tid; j < nflat; j += kBlock) dst[j] = tile[j];
        }
        __syncthreads();
    }
}

hipError_t launch_rows(int mode, int count, const GsRowTensor* t, int64_t row0, int64_t n_valid, int64_t n_rows, const float* in, float* out,
                       hipStream_t st)
{
    RowBatch b{};
    int G = 0;
    for (int i = 0; i < count; ++i) {
        b.p[i] = t[i].param; b.m[i] = t[i].exp_avg; b.v[i] = t[i].exp_avg_sq; b.g[i] = t[i].grad; b.width[i] = t[i].width;
        b.inv_width[i] = 1.0f / (float)t[i].width;
        if (mode == 2) {
            const double bc1 = 1.0 - pow(t[i].beta1, (double)t[i].step), bc2 = 1.0 - pow(t[i].beta2, (double)t[i].step);
            b.one_m_b1[i] = (float)(1.0 - t[i].beta1); b.b2[i] = (float)t[i].beta2; b.one_m_b2[i] = (float)(1.0 - t[i].beta2);
            b.step_size[i] = (float)(t[i].lr / bc1); b.inv_bc2s[i] = (float)(1.0 / sqrt(bc2)); b.eps[i] = (float)t[i].eps;
        }
        G += t[i].width;
    }
    b.count = count; b.G = G; b.fail = mode == 2 ? chain_fail_word() : nullptr;
    if (n_rows <= 0 || G <= 0) return hipSuccess;
    const bool wide = G > 16;
    if (G > 64) return hipErrorInvalidValue;                        // (gs_pack_columns refuses more than 64 floats per Gaussian before it gets here)
    const int rows = wide ? 64 : 256;
    int64_t nb = (n_rows + rows - 1) / rows;
    if (nb > 256 * 16) nb = 256 * 16;
#define GS_ROWS(M)                                                                                                                              \
    do {                                                                                                                                        \
        if (wide) hipLaunchKernelGGL((rows_kernel<M, 64, 64>), dim3((unsigned)nb), dim3(kBlock), 0, st, b, row0, n_valid, n_rows, in, out);      \
        else hipLaunchKernelGGL((rows_kernel<M, 256, 16>), dim3((unsigned)nb), dim3(kBlock), 0, st, b, row0, n_valid, n_rows, in, out);          \
    } while (0)
    if (mode == 0) GS_ROWS(0); else if (mode == 1) GS_ROWS(1); else GS_ROWS(2);
#undef GS_ROWS
    return hipGetLastError();
}

}  // namespace gs
