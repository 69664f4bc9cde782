// loss.hip -- fused mapping loss, forward AND backward in two launches.
//
// Replaces the ~40 torch kernels the reference's get_loss issues per iteration for
//     loss = w_depth * mean_{gt_depth > 0} |gt_depth - depth|
//          + w_im    * ( 0.8 * mean |im - gt_im|  +  0.2 * (1 - mean SSIM(im, gt_im)) )
// (src/mapper/splatam/splatam.py:213-249; SSIM = src/mapper/splatam/utils/slam_external.py:54-97: 11x11 Gaussian
// window sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2, five depthwise convolutions + their autograd) and
// produces dL/dim and dL/ddepth directly for the rasteriser's backward.  On MI355X the depthwise 11x11
// convolutions alone cost ~1.7 ms per iteration through MIOpen; here every 16x16 pixel tile stages the rendered
// and target tiles (+5 px halo) in LDS and runs the separable window there.
//
//   loss_stats_kernel : per tile and channel, mu1, mu2, E[x^2], E[y^2], E[xy] by separable 11-tap passes in LDS ->
//                       SSIM value and its partials w.r.t. (mu1, E[x^2], E[xy]) per pixel; block-reduced sums of
//                       SSIM, |x-y|, masked |depth error| and the mask count go to 4 device accumulators.
//   loss_grad_kernel  : convolves the three partial maps with the (symmetric) window, combines them into dL/dim,
//                       adds the L1 terms, writes dL/ddepth (needs the mask count of pass 1 -- read from device
//                       memory, no host sync) and the three loss scalars.
#include "gs_common.h"

namespace gs {

constexpr int kLT = 16;              // output tile edge
constexpr int kLH = 5;               // window half-width
constexpr int kLP = kLT + 2 * kLH;   // 26: tile + halo

// normalised 1-D Gaussian window, sigma 1.5 (exp(-(i-5)^2 / 4.5) / sum)
__device__ __constant__ float kWin[11] = {0.00102838f, 0.00759876f, 0.03600077f, 0.10936069f, 0.21300554f, 0.26601172f,
                                          0.21300554f, 0.10936069f, 0.03600077f, 0.00759876f, 0.00102838f};

__device__ __forceinline__ float block_sum(float v, float* s_red, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// Accumulators: kAccSlots copies of {sum SSIM, sum |im - gt|, sum masked |gt_depth - depth|, mask count}, one 64-byte line
// each; a block adds to slot (block index mod kAccSlots).  With a single copy the 4 x 1200 same-line device atomics of a
// 640x480 frame serialise at the memory side and cost ~70 us -- more than all the arithmetic of the loss.
// (round 3: 256 copies, and a block's four sums leave as ONE request -- four lanes of one atomic instruction on one line -- instead of four:
// 3600 blocks on 64 lines were 4 x 56 same-line atomics in a row)
constexpr int kAccSlots = kLossAccSlots;
constexpr int kAccFloats = kAccSlots * 16;
__device__ __forceinline__ void acc_totals(const float* __restrict__ acc, float* s_tot, int tid)
{
    if (tid < kWave) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < kAccSlots / kWave; k++) {
            const float4 u = *reinterpret_cast<const float4*>(acc + (k * kWave + tid) * 16);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const float a = wave_sum(v.x), b = wave_sum(v.y), c = wave_sum(v.z), d = wave_sum(v.w);
        if (tid == 0) { s_tot[0] = a; s_tot[1] = b; s_tot[2] = c; s_tot[3] = d; }
    }
    __syncthreads();
}
__global__ __launch_bounds__(kBlock) void loss_stats_kernel(int W, int H, const float* __restrict__ im,
                                                            const float* __restrict__ gt, const float* __restrict__ depth,
                                                            const float* __restrict__ depth_sq,
                                                            const float* __restrict__ gt_depth, float* __restrict__ partials,
                                                            float* __restrict__ acc)
{
    __shared__ float s_x[kLP][kLP + 1];
    __shared__ float s_y[kLP][kLP + 1];
    __shared__ float s_h[5][kLP][kLT + 1];          // horizontal-pass results of x, y, xx, yy, xy
    __shared__ float s_red[16];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)W * H;
    float sum_ssim = 0.f, sum_l1 = 0.f;
    const int ch = blockIdx.z;                  // one colour channel per workgroup: 3x the workgroups, a third of the serial chain
    {
        // the tile + halo: 676 values of each image, three per thread.  ALL loads are issued before the first LDS store: as a loop with a store
        // behind each load the staging was three dependent memory round trips -- 5 of a wavefront's 6.8 us (round 6, rocprofv3 SQ_WAVE_CYCLES)
        constexpr int kStage = (kLP * kLP + kBlock - 1) / kBlock;
        float vx[kStage], vy[kStage];
#pragma unroll
        for (int it = 0; it < kStage; it++) {
            const int e = tid + it * kBlock;
            const int r = e / kLP, c = e - r * kLP;
            const int gx = x0 + c - kLH, gy = y0 + r - kLH;
            const bool in = e < kLP * kLP && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t o = ch * HW + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
            vx[it] = im[o]; vy[it] = gt[o];
            if (!in) { vx[it] = 0.f; vy[it] = 0.f; }
        }
#pragma unroll
        for (int it = 0; it < kStage; it++) {
            const int e = tid + it * kBlock;
            if (e < kLP * kLP) { const int r = e / kLP, c = e - r * kLP; s_x[r][c] = vx[it]; s_y[r][c] = vy[it]; }
        }
        __syncthreads();
        for (int e = tid; e < kLP * kLT; e += kBlock) {       // horizontal pass: 26 rows x 16 columns
            const int r = e / kLT, c = e - r * kLT;
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kWin[k], xv = s_x[r][c + k], yv = s_y[r][c + k];
                a += w * xv; b += w * yv; aa += w * xv * xv; bb += w * yv * yv; ab += w * xv * yv;
            }
            s_h[0][r][c] = a; s_h[1][r][c] = b; s_h[2][r][c] = aa; s_h[3][r][c] = bb; s_h[4][r][c] = ab;
        }
        __syncthreads();
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {                         // vertical pass
            const float w = kWin[k];
            m1 += w * s_h[0][ty + k][tx]; m2 += w * s_h[1][ty + k][tx]; e11 += w * s_h[2][ty + k][tx];
            e22 += w * s_h[3][ty + k][tx]; e12 += w * s_h[4][ty + k][tx];
        }
        if (inside) {
            const float c1 = 0.0001f, c2 = 0.0009f;
            const float A1 = 2.f * m1 * m2 + c1, A2 = 2.f * (e12 - m1 * m2) + c2;
            const float B1 = m1 * m1 + m2 * m2 + c1, B2 = (e11 - m1 * m1) + (e22 - m2 * m2) + c2;
            const float inv = 1.0f / (B1 * B2);
            const float S = A1 * A2 * inv;
            sum_ssim += S;
            const size_t o = ch * HW + (size_t)py * W + px;
            partials[o] = 2.f * m2 * (A2 - A1) * inv - 2.f * m1 * S * (1.0f / B1 - 1.0f / B2);   // dS/dmu1
            partials[3 * HW + o] = -S / B2;                                                      // dS/dE[x^2]
            partials[6 * HW + o] = 2.f * A1 * inv;                                               // dS/dE[xy]
            sum_l1 += fabsf(s_x[ty + kLH][tx + kLH] - s_y[ty + kLH][tx + kLH]);
        }
    }
    float sum_d = 0.f, cnt = 0.f;
    if (inside && ch == 0) {                    // the depth term rides with the channel-0 workgroups
        const size_t o = (size_t)py * W + px;
        const float d = depth[o], g = gt_depth[o];
        const float unc = depth_sq ? depth_sq[o] - d * d : 0.f;
        if (g > 0.f && d == d && unc == unc) { sum_d = fabsf(g - d); cnt = 1.f; }
    }
    // the four sums of the workgroup in ONE reduction (two barriers instead of eight): wave sums -> LDS -> four lanes add the four waves' values
    sum_ssim = wave_sum(sum_ssim); sum_l1 = wave_sum(sum_l1); sum_d = wave_sum(sum_d); cnt = wave_sum(cnt);
    __syncthreads();
    if ((tid & 63) == 0) { float* r = s_red + (tid >> 6) * 4; r[0] = sum_ssim; r[1] = sum_l1; r[2] = sum_d; r[3] = cnt; }
    __syncthreads();
    if (tid < 4) {
        float* a = acc + (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & (kAccSlots - 1)) * 16;
        atomicAdd(a + tid, (s_red[tid] + s_red[4 + tid]) + (s_red[8 + tid] + s_red[12 + tid]));
    }
}

__global__ __launch_bounds__(kBlock) void loss_grad_kernel(int W, int H, const float* __restrict__ im,
                                                           const float* __restrict__ gt, const float* __restrict__ depth,
                                                           const float* __restrict__ depth_sq,
                                                           const float* __restrict__ gt_depth, const float* __restrict__ partials,
                                                           const float* __restrict__ acc, float w_im, float w_depth,
                                                           float* __restrict__ dL_dim, float* __restrict__ dL_ddepth,
                                                           float* __restrict__ losses, float* __restrict__ acc_other)
{
    __shared__ float s_p[3][kLP][kLP + 1];
    __shared__ float s_h[3][kLP][kLT + 1];
    __shared__ float s_tot[4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    acc_totals(acc, s_tot, tid);
    // persistent scratch: the accumulator set of the NEXT call is zeroed here (nobody touches it during this call), so that no memset is needed
    if (acc_other && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int e = tid; e < kAccFloats; e += kBlock) acc_other[e] = 0.0f;
    const int x0 = blockIdx.x * kLT, y0 = blockIdx.y * kLT;
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)W * H;
    const float n3 = 3.0f * (float)HW;
    const float k_ssim = -0.2f * w_im / n3, k_l1 = 0.8f * w_im / n3;
    const int ch = blockIdx.z;
    {
        // (all nine loads of a thread -- and the centre pixel's im / gt of the combine below -- in flight before the first LDS store: see loss_stats_kernel)
        constexpr int kStage = (kLP * kLP + kBlock - 1) / kBlock;
        float v0[kStage], v1[kStage], v2[kStage];
#pragma unroll
        for (int it = 0; it < kStage; it++) {
            const int e = tid + it * kBlock;
            const int r = e / kLP, c = e - r * kLP;
            const int gx = x0 + c - kLH, gy = y0 + r - kLH;
            const bool in = e < kLP * kLP && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t o = ch * HW + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
            v0[it] = partials[o]; v1[it] = partials[3 * HW + o]; v2[it] = partials[6 * HW + o];
            if (!in) { v0[it] = 0.f; v1[it] = 0.f; v2[it] = 0.f; }
        }
#pragma unroll
        for (int it = 0; it < kStage; it++) {
            const int e = tid + it * kBlock;
            if (e < kLP * kLP) { const int r = e / kLP, c = e - r * kLP; s_p[0][r][c] = v0[it]; s_p[1][r][c] = v1[it]; s_p[2][r][c] = v2[it]; }
        }
        __syncthreads();
        for (int e = tid; e < kLP * kLT; e += kBlock) {
            const int r = e / kLT, c = e - r * kLT;
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = kWin[k];
                a += w * s_p[0][r][c + k]; b += w * s_p[1][r][c + k]; d += w * s_p[2][r][c + k];
            }
            s_h[0][r][c] = a; s_h[1][r][c] = b; s_h[2][r][c] = d;
        }
        __syncthreads();
        float g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = kWin[k];
            g1 += w * s_h[0][ty + k][tx]; g2 += w * s_h[1][ty + k][tx]; g3 += w * s_h[2][ty + k][tx];
        }
        if (inside) {
            const size_t o = ch * HW + (size_t)py * W + px;
            const float x = im[o], y = gt[o];
            const float diff = x - y;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            dL_dim[o] = k_ssim * (g1 + 2.f * x * g2 + y * g3) + k_l1 * sgn;
        }
    }
    const float cnt = s_tot[3];
    if (inside && ch == 0) {
        const size_t o = (size_t)py * W + px;
        const float d = depth[o], g = gt_depth[o];
        const float unc = depth_sq ? depth_sq[o] - d * d : 0.f;
        const bool m = g > 0.f && d == d && unc == unc;
        const float diff = d - g;
        const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
        dL_ddepth[o] = m ? w_depth * sgn / cnt : 0.f;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
        const float l_im = w_im * (0.8f * s_tot[1] / n3 + 0.2f * (1.0f - s_tot[0] / n3));
        const float l_depth = w_depth * s_tot[2] / cnt;
        losses[0] = l_im + l_depth; losses[1] = l_im; losses[2] = l_depth;
        losses[3] = l_im + l_depth;      // second copy: the host side hands out [0..2] as the report and [3] as the loss value
    }
}

hipError_t launch_mapping_loss(int W, int H, const float* im, const float* gt, const float* depth, const float* depth_sq,
                               const float* gt_depth, float w_im, float w_depth, float* losses, float* dL_dim,
                               float* dL_ddepth, float* scratch, int64_t persistent_call, hipStream_t st)
{
    // two sets of kAccSlots x 4 accumulators (one 64-byte line each), then 9 partial maps.  persistent_call = 0: any scratch, set 0 is
    // memset here.  persistent_call = k >= 1: the k-th call on a scratch its owner zeroed ONCE and keeps for this stream -- the call uses
    // set k & 1 and its second kernel zeroes the other one for call k + 1 (no memset launch per call)
    const int set = persistent_call > 0 ? (int)(persistent_call & 1) : 0;
    float* acc = scratch + set * kAccFloats;
    float* acc_other = persistent_call > 0 ? scratch + (1 - set) * kAccFloats : nullptr;
    float* partials = scratch + 2 * kAccFloats;
    if (persistent_call <= 0) {
        hipError_t e = hipMemsetAsync(acc, 0, kAccFloats * sizeof(float), st);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((W + kLT - 1) / kLT, (H + kLT - 1) / kLT, 3);
    hipLaunchKernelGGL(loss_stats_kernel, grid, dim3(kBlock), 0, st, W, H, im, gt, depth, depth_sq, gt_depth, partials, acc);
    hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(kBlock), 0, st, W, H, im, gt, depth, depth_sq, gt_depth, partials, acc, w_im,
                       w_depth, dL_dim, dL_ddepth, losses, acc_other);
    return hipGetLastError();
}

}  // namespace gs
