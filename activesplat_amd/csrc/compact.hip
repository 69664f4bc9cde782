// compact.hip -- stream compaction of per-Gaussian rows (prune / densify surgery).
//
// Replaces the boolean-mask indexing + torch.cat + Parameter re-creation the reference performs on every
// per-Gaussian tensor AND its Adam moments (src/mapper/splatam/utils/slam_external.py:143-164 remove_points,
// :126-140 cat_params_to_optimizer, :171-247 prune_gaussians / densify).  Two primitives:
//   compact_index : keep-mask -> ordered list of kept row indices (+ count), wavefront ballot/popcount scan
//   gather_rows   : dst[r][:] = src[index[r]][:] for any row width (params 3/4/1 floats, moments, statistics)
// HBM-bound streaming; one index build serves every tensor of the surgery.
#include "gs_common.h"

namespace gs {

constexpr int kCompactBlock = 1024;

__global__ __launch_bounds__(kCompactBlock) void compact_count_kernel(int64_t n, const uint8_t* __restrict__ keep,
                                                                      uint32_t* __restrict__ block_counts)
{
    __shared__ uint32_t s_w[kCompactBlock / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCompactBlock + tid;
    const unsigned long long m = __ballot(i < n && keep[i] != 0);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0;
        for (int w = 0; w < kCompactBlock / kWave; w++) c += s_w[w];
        block_counts[blockIdx.x] = c;
    }
}

// (workgroup b scans list b of a multi-list compaction: its nb counts start at block_counts + b * nb, its total goes to d_count[b])
__global__ __launch_bounds__(1024) void compact_scan_kernel(uint32_t* __restrict__ block_counts, int nb, uint32_t* __restrict__ d_count)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    block_counts += (size_t)blockIdx.x * nb; d_count += blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < nb ? block_counts[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(v, lane);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t wprefix = 0;
        for (int w = 0; w < wave; w++) wprefix += s_w[w];
        const uint32_t carry = s_carry;
        if (i < nb) block_counts[i] = carry + wprefix + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wprefix + inc;
        __syncthreads();
    }
    if (tid == 0) *d_count = s_carry;
}

__global__ __launch_bounds__(kCompactBlock) void compact_write_kernel(int64_t n, const uint8_t* __restrict__ keep,
                                                                      const uint32_t* __restrict__ block_offsets,
                                                                      uint32_t* __restrict__ src_index)
{
    __shared__ uint32_t s_w[kCompactBlock / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCompactBlock + tid;
    const bool k = i < n && keep[i] != 0;
    const unsigned long long m = __ballot(k);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x];
    for (int w = 0; w < wave; w++) off += s_w[w];
    if (k) src_index[off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
}

// ---- three masks -> ONE index list in one count / scan / write sequence (the densify event: surviving originals | surviving clones |
// n_rep blocks of surviving split parents = the rows of the children, slam_external.py:216, :232, :236).  The three-list layout needs the
// totals of the first two lists before anything can be written, so the scan kernel leaves them in d_counts and the write kernel reads
// them there; the host reads d_counts once, afterwards, to size the new tensors.
__global__ __launch_bounds__(kCompactBlock) void compact3_count_kernel(int64_t n, const uint8_t* __restrict__ ka, const uint8_t* __restrict__ kb,
                                                                       const uint8_t* __restrict__ kc, uint32_t* __restrict__ block_counts, int nb)
{
    __shared__ uint32_t s_w[3][kCompactBlock / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCompactBlock + tid;
    const unsigned long long ma = __ballot(i < n && ka[i] != 0), mb = __ballot(i < n && kb[i] != 0), mc = __ballot(i < n && kc[i] != 0);
    if (lane == 0) { s_w[0][wave] = (uint32_t)__popcll(ma); s_w[1][wave] = (uint32_t)__popcll(mb); s_w[2][wave] = (uint32_t)__popcll(mc); }
    __syncthreads();
    if (tid < 3) {
        uint32_t c = 0;
        for (int w = 0; w < kCompactBlock / kWave; w++) c += s_w[tid][w];
        block_counts[(size_t)tid * nb + blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(kCompactBlock) void compact3_write_kernel(int64_t n, const uint8_t* __restrict__ ka, const uint8_t* __restrict__ kb,
                                                                       const uint8_t* __restrict__ kc, const uint32_t* __restrict__ block_offsets, int nb,
                                                                       const uint32_t* __restrict__ d_counts, int n_rep, uint32_t* __restrict__ src_index)
{
    __shared__ uint32_t s_w[3][kCompactBlock / kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * kCompactBlock + tid;
    const bool a = i < n && ka[i] != 0, b = i < n && kb[i] != 0, c = i < n && kc[i] != 0;
    const unsigned long long ma = __ballot(a), mb = __ballot(b), mc = __ballot(c);
    if (lane == 0) { s_w[0][wave] = (uint32_t)__popcll(ma); s_w[1][wave] = (uint32_t)__popcll(mb); s_w[2][wave] = (uint32_t)__popcll(mc); }
    __syncthreads();
    uint32_t oa = block_offsets[blockIdx.x], ob = block_offsets[(size_t)nb + blockIdx.x], oc = block_offsets[2 * (size_t)nb + blockIdx.x];
    for (int w = 0; w < wave; w++) { oa += s_w[0][w]; ob += s_w[1][w]; oc += s_w[2][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t na = d_counts[0], nbb = d_counts[1], nc = d_counts[2];
    if (a) src_index[oa + (uint32_t)__popcll(ma & below)] = (uint32_t)i;
    if (b) src_index[na + ob + (uint32_t)__popcll(mb & below)] = (uint32_t)i;
    if (c) {
        const uint32_t r = oc + (uint32_t)__popcll(mc & below);
        for (int k = 0; k < n_rep; k++) src_index[(size_t)na + nbb + (size_t)k * nc + r] = (uint32_t)i;
    }
}

// rows [0, n_copy) are gathered, rows [n_copy, n_out) zero-filled (the Adam moments of appended Gaussians start from zero:
// slam_external.py:131-134)
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(int64_t total, int row_floats, const uint32_t* __restrict__ src_index,
                                                              const float* __restrict__ src, float* __restrict__ dst, int64_t n_copy)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / row_floats;
        const int c = (int)(e - r * row_floats);
        dst[e] = r < n_copy ? src[(int64_t)src_index[r] * row_floats + c] : 0.0f;
    }
}

// rows of a multiple of 4 floats (quaternions, SH coefficient rows and their Adam moments): 16 B per lane, 32-bit index math
__global__ __launch_bounds__(kBlock) void gather_rows_vec4_kernel(uint32_t total4, uint32_t row4, const uint32_t* __restrict__ src_index,
                                                                   const float4* __restrict__ src, float4* __restrict__ dst, uint32_t n_copy)
{
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < total4; e += stride) {
        const uint32_t r = e / row4, c = e - r * row4;
        dst[e] = r < n_copy ? src[(size_t)src_index[r] * row4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

uint64_t compact_scratch_bytes(int64_t n) { return (uint64_t)((n + kCompactBlock - 1) / kCompactBlock + 1) * 4; }

hipError_t launch_compact_index(int64_t n, const uint8_t* keep, uint32_t* src_index, uint32_t* d_count, void* scratch, hipStream_t st)
{
    const int nb = (int)((n + kCompactBlock - 1) / kCompactBlock);
    uint32_t* bc = (uint32_t*)scratch;
    if (nb > 0) hipLaunchKernelGGL(compact_count_kernel, dim3(nb), dim3(kCompactBlock), 0, st, n, keep, bc);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, bc, nb, d_count);
    if (nb > 0) hipLaunchKernelGGL(compact_write_kernel, dim3(nb), dim3(kCompactBlock), 0, st, n, keep, bc, src_index);
    return hipGetLastError();
}

uint64_t compact3_scratch_bytes(int64_t n) { return 3 * compact_scratch_bytes(n); }

hipError_t launch_compact_index3(int64_t n, const uint8_t* ka, const uint8_t* kb, const uint8_t* kc, int n_rep, uint32_t* src_index,
                                 uint32_t* d_counts, void* scratch, hipStream_t st)
{
    const int nb = (int)((n + kCompactBlock - 1) / kCompactBlock);
    uint32_t* bc = (uint32_t*)scratch;
    if (nb > 0) hipLaunchKernelGGL(compact3_count_kernel, dim3(nb), dim3(kCompactBlock), 0, st, n, ka, kb, kc, bc, nb);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(3), dim3(1024), 0, st, bc, nb, d_counts);
    if (nb > 0) hipLaunchKernelGGL(compact3_write_kernel, dim3(nb), dim3(kCompactBlock), 0, st, n, ka, kb, kc, bc, nb, d_counts, n_rep, src_index);
    return hipGetLastError();
}

hipError_t launch_gather_rows(int64_t n_out, int row_floats, const uint32_t* src_index, const float* src, float* dst, int64_t n_copy, hipStream_t st)
{
    const int64_t total = n_out * row_floats;
    if (total <= 0) return hipSuccess;
    if ((row_floats & 3) == 0 && (total >> 2) < ((int64_t)1 << 32) && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const uint32_t total4 = (uint32_t)(total >> 2);
        uint32_t nb4 = (total4 + kBlock - 1) / kBlock;
        if (nb4 > 256 * 16) nb4 = 256 * 16;
        hipLaunchKernelGGL(gather_rows_vec4_kernel, dim3(nb4), dim3(kBlock), 0, st, total4, (uint32_t)(row_floats >> 2), src_index,
                           reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), (uint32_t)n_copy);
        return hipGetLastError();
    }
    int64_t nb = (total + kBlock - 1) / kBlock;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)nb), dim3(kBlock), 0, st, total, row_floats, src_index, src, dst, n_copy);
    return hipGetLastError();
}

}  // namespace gs
