// grow.hip -- map growth and keyframe-overlap scoring, the two per-map-frame steps either side of the optimise loop
// (SURVEY.md section 8f-3).
//
// 1. add_new_gaussians (src/mapper/splatam/splatam.py:332-379, helpers get_pointcloud :25-75 and
//    initialize_new_params :304-329): from the rendered depth + silhouette of the current map and the sensor frame,
//      err          = |gt - render| * (gt > 0)
//      non_presence = sil < thr  |  (render > gt  &  err > 2 median(err)  &  sil > thr  &  gt < 5)
//    every non_presence pixel with valid depth becomes a Gaussian: mean = c2w * ((u-cx)/fx z, (v-cy)/fy z, z), colour =
//    pixel RGB, rotation (1,0,0,0), logit opacity 0, log scale = log(sqrt((z / ((fx+fy)/2))^2)).  The reference does this
//    with ~40 torch launches, a device-wide sort for the median and boolean-mask gathers; here: one single-workgroup radix
//    select (the image is only H*W <= a few 100k values), one mask kernel, the ordered compaction of compact.hip and one
//    row-emitting kernel.  Row order = row-major pixel order, as boolean-mask indexing yields.
// 2. keyframe_selection_overlap's scoring loop (src/mapper/splatam/utils/keyframe_selection.py:62-86): for every keyframe,
//    the number of sampled world points that project inside its image with a 20 px border -- one workgroup per keyframe
//    instead of ~12 torch launches and a blocking .sum() each.
#include "gs_common.h"

namespace gs {

constexpr int kSelectThreads = 1024;
constexpr int kSelectBins = 2048;

__device__ __forceinline__ float depth_error(const float* __restrict__ gt, const float* __restrict__ rd, int64_t i)
{
    const float g = gt[i];
    return fabsf(g - rd[i]) * (g > 0.0f ? 1.0f : 0.0f);
}

// torch.median of a flat tensor = the LOWER median, element (n-1)/2 of the sorted values.  err >= 0, so the order of the
// floats is the order of their bit patterns: three histogram passes (11 + 11 + 10 bits) pin the value exactly.
__global__ __launch_bounds__(kSelectThreads) void grow_median_kernel(int64_t n, const float* __restrict__ gt,
                                                                      const float* __restrict__ rd, float* __restrict__ d_median)
{
    __shared__ uint32_t s_hist[kSelectBins];
    __shared__ uint32_t s_prefix, s_mask;
    __shared__ uint64_t s_k;
    const int tid = threadIdx.x;
    if (tid == 0) { s_prefix = 0u; s_mask = 0u; s_k = (uint64_t)((n - 1) / 2); }
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        for (int b = tid; b < kSelectBins; b += kSelectThreads) s_hist[b] = 0u;
        __syncthreads();
        const uint32_t prefix = s_prefix, mask = s_mask;
        const int sh = shifts[pass];
        const uint32_t bm = (1u << widths[pass]) - 1u;
        for (int64_t i = tid; i < n; i += kSelectThreads) {
            const uint32_t bits = __float_as_uint(depth_error(gt, rd, i));
            if ((bits & mask) == prefix) atomicAdd(&s_hist[(bits >> sh) & bm], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint64_t k = s_k;
            uint32_t b = 0;
            for (; b < bm; ++b) {
                const uint32_t c = s_hist[b];
                if (k < c) break;
                k -= c;
            }
            s_k = k;
            s_prefix = prefix | (b << sh);
            s_mask = mask | (bm << sh);
        }
        __syncthreads();
    }
    if (tid == 0) *d_median = __uint_as_float(s_prefix);
}

__global__ __launch_bounds__(kBlock) void grow_mask_kernel(int64_t n, const float* __restrict__ gt, const float* __restrict__ rd,
                                                            const float* __restrict__ sil, const float* __restrict__ d_median,
                                                            float sil_thres, uint8_t* __restrict__ keep, uint32_t* __restrict__ d_candidates)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool cand = false, take = false;
    if (i < n) {
        const float g = gt[i], r = rd[i], s = sil[i];
        const float err = fabsf(g - r) * (g > 0.0f ? 1.0f : 0.0f);
        const bool behind = (r > g) && (err > 2.0f * d_median[0]);
        cand = (s < sil_thres) || (behind && (s > sil_thres) && (g < 5.0f));
        take = cand && (g > 0.0f);
        keep[i] = take ? 1 : 0;
    }
    // candidate count: 64 accumulator lines (thousands of same-address device atomics would serialise for ~70 us)
    const unsigned long long m = __ballot(cand);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(d_candidates + (blockIdx.x & 63) * 16, (uint32_t)__popcll(m));
}

__global__ __launch_bounds__(kWave) void grow_sum_slots_kernel(const uint32_t* __restrict__ slots, uint32_t* __restrict__ out)
{
    const uint32_t v = wave_sum_u32(slots[threadIdx.x * 16]);
    if (threadIdx.x == 0) *out = v;
}

struct GrowCam { float fx, fy, cx, cy; float c2w[12]; int W; int isotropic; };

__global__ __launch_bounds__(kBlock) void grow_rows_kernel(GrowCam c, int64_t npix, const uint32_t* __restrict__ d_count,
                                                            const uint32_t* __restrict__ index, const float* __restrict__ gt,
                                                            const float* __restrict__ color, float* __restrict__ means3D,
                                                            float* __restrict__ rgb, float* __restrict__ rot, float* __restrict__ logit,
                                                            float* __restrict__ log_scales)
{
    const int64_t count = (int64_t)*d_count;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < count; r += stride) {
        const uint32_t pix = index[r];
        const int u = (int)(pix % (uint32_t)c.W), v = (int)(pix / (uint32_t)c.W);
        const float z = gt[pix];
        const float x = ((float)u - c.cx) / c.fx * z;
        const float y = ((float)v - c.cy) / c.fy * z;
        means3D[3 * r + 0] = c.c2w[0] * x + c.c2w[1] * y + c.c2w[2] * z + c.c2w[3];
        means3D[3 * r + 1] = c.c2w[4] * x + c.c2w[5] * y + c.c2w[6] * z + c.c2w[7];
        means3D[3 * r + 2] = c.c2w[8] * x + c.c2w[9] * y + c.c2w[10] * z + c.c2w[11];
        rgb[3 * r + 0] = color[pix];
        rgb[3 * r + 1] = color[npix + pix];
        rgb[3 * r + 2] = color[2 * npix + pix];
        rot[4 * r + 0] = 1.0f; rot[4 * r + 1] = 0.0f; rot[4 * r + 2] = 0.0f; rot[4 * r + 3] = 0.0f;
        logit[r] = 0.0f;
        const float sd = z / ((c.fx + c.fy) / 2.0f);
        const float ls = logf(sqrtf(sd * sd));
        if (c.isotropic) log_scales[r] = ls;
        else { log_scales[3 * r + 0] = ls; log_scales[3 * r + 1] = ls; log_scales[3 * r + 2] = ls; }
    }
}

uint64_t grow_scratch_bytes(int64_t npix)
{   // keep mask | index list | median | compaction block sums
    return (uint64_t)((npix + 255) / 256 * 256) + (uint64_t)npix * 4 + 256 + 64 * 64 + compact_scratch_bytes(npix) + 256;
}

hipError_t launch_grow(int W, int H, const float* rd, const float* sil, const float* gt, const float* color, const float* k4,
                       const float* c2w12, float sil_thres, int isotropic, float* means3D, float* rgb, float* rot, float* logit,
                       float* log_scales, uint32_t* d_counts, void* scratch, hipStream_t st)
{
    const int64_t n = (int64_t)W * H;
    uint8_t* keep = (uint8_t*)scratch;
    uint32_t* index = (uint32_t*)(keep + (n + 255) / 256 * 256);
    float* med = (float*)(index + n);
    uint32_t* slots = (uint32_t*)(med + 64);                 // 64 lines of candidate counters
    void* cscr = (void*)(slots + 64 * 16);
    hipError_t e = hipMemsetAsync(slots, 0, 64 * 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(grow_median_kernel, dim3(1), dim3(kSelectThreads), 0, st, n, gt, rd, med);
    const int nb = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(grow_mask_kernel, dim3(nb), dim3(kBlock), 0, st, n, gt, rd, sil, (const float*)med, sil_thres, keep, slots);
    hipLaunchKernelGGL(grow_sum_slots_kernel, dim3(1), dim3(kWave), 0, st, (const uint32_t*)slots, d_counts);
    e = launch_compact_index(n, keep, index, d_counts + 1, cscr, st);
    if (e != hipSuccess) return e;
    GrowCam c;
    c.fx = k4[0]; c.fy = k4[1]; c.cx = k4[2]; c.cy = k4[3]; c.W = W; c.isotropic = isotropic;
    for (int i = 0; i < 12; ++i) c.c2w[i] = c2w12[i];
    int nbr = nb > 256 * 4 ? 256 * 4 : nb;
    hipLaunchKernelGGL(grow_rows_kernel, dim3(nbr), dim3(kBlock), 0, st, c, n, (const uint32_t*)(d_counts + 1), (const uint32_t*)index, gt,
                       color, means3D, rgb, rot, logit, log_scales);
    return hipGetLastError();
}

// ---- keyframe overlap ------------------------------------------------------------------------------------------------
struct OverlapCam { float k[9]; float w, h, edge; };

__global__ __launch_bounds__(kBlock) void keyframe_overlap_kernel(OverlapCam c, int n_pts, const float* __restrict__ pts,
                                                                   const float* __restrict__ w2c, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t s_sum;
    const float* m = w2c + 16 * (int64_t)blockIdx.x;            // row-major 4x4 estimated w2c of keyframe blockIdx.x
    if (threadIdx.x == 0) s_sum = 0u;
    __syncthreads();
    uint32_t mine = 0;
    for (int i = threadIdx.x; i < n_pts; i += kBlock) {
        const float X = pts[3 * i], Y = pts[3 * i + 1], Z = pts[3 * i + 2];
        const float x = m[0] * X + m[1] * Y + m[2] * Z + m[3];
        const float y = m[4] * X + m[5] * Y + m[6] * Z + m[7];
        const float z = m[8] * X + m[9] * Y + m[10] * Z + m[11];
        const float px = c.k[0] * x + c.k[1] * y + c.k[2] * z;
        const float py = c.k[3] * x + c.k[4] * y + c.k[5] * z;
        const float pz = (c.k[6] * x + c.k[7] * y + c.k[8] * z) + 1e-5f;
        const float u = px / pz, v = py / pz;
        if (u < c.w - c.edge && u > c.edge && v < c.h - c.edge && v > c.edge && pz > 0.0f) ++mine;
    }
    const unsigned long long any = __ballot(mine != 0);
    if (any) atomicAdd(&s_sum, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_sum;
}

hipError_t launch_keyframe_overlap(int n_pts, const float* pts, int n_kf, const float* w2c, const float* k9, int W, int H, int edge,
                                   uint32_t* counts, hipStream_t st)
{
    if (n_kf <= 0) return hipSuccess;
    OverlapCam c;
    for (int i = 0; i < 9; ++i) c.k[i] = k9[i];
    c.w = (float)W; c.h = (float)H; c.edge = (float)edge;
    hipLaunchKernelGGL(keyframe_overlap_kernel, dim3(n_kf), dim3(kBlock), 0, st, c, n_pts, pts, w2c, counts);
    return hipGetLastError();
}

}  // namespace gs
