// binning.hip -- tile-instance emission ("duplicate with keys") and per-tile range detection.
//
// Replaces the inclusive-scan / duplicateWithKeys / identifyTileRanges work items of the reference's
// absent CUDA extension.  MI355X design: the scan is a wavefront prefix sum fused
// into the emitter, and each wavefront writes ITS 64 Gaussians' instances cooperatively -- lane l
// writes instance (t + l) of the wave's contiguous output range, finding the owning Gaussian by a
// 6-step binary search over the wave's prefix array in LDS -- so the (key,value) stores are perfectly
// coalesced 8 B / 4 B per lane and a Gaussian covering thousands of tiles costs no more per instance
// than one covering a single tile.
//
// key = (tile_id << 32) | float_bits(view-space depth); value = Gaussian index.  Integer work: bit-exact
// against the oracle.
#include "gs_common.h"

namespace gs {

__global__ __launch_bounds__(kBlock) void emit_kernel(Cam cam, int P, GeomPtrs gp, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals)
{
    __shared__ uint32_t s_incl[kBlock];       // per-wave inclusive prefix of tiles_touched
    __shared__ uint32_t s_x0w[kBlock];        // xmin | (width << 16)
    __shared__ uint32_t s_y0[kBlock];
    __shared__ uint32_t s_depth[kBlock];
    __shared__ uint32_t s_wtot[kBlock / kWave];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * kBlock + tid;
    uint32_t n = 0;
    if (i < P) {
        n = gp.tiles[i];
        const uint2 rc = gp.rect[i];
        const uint32_t x0 = rc.x & 0xffffu, x1 = rc.x >> 16;
        s_x0w[tid] = x0 | ((x1 - x0) << 16);
        s_y0[tid] = rc.y & 0xffffu;
        s_depth[tid] = gp.depth_bits[i];
    }
    const uint32_t incl = wave_inclusive_scan(n, lane);
    s_incl[tid] = incl;
    if (lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t wbase = gp.block_sums[blockIdx.x];
    for (int w = 0; w < wave; w++) wbase += s_wtot[w];
    if (i < P) gp.offsets[i] = wbase + incl;
    const uint32_t total = s_wtot[wave];
    const uint32_t* incl_w = s_incl + wave * kWave;
    for (uint32_t t0 = 0; t0 < total; t0 += kWave) {
        const uint32_t t = t0 + lane;
        if (t < total) {
            // smallest j with incl_w[j] > t
            int lo = 0, hi = 63;
#pragma unroll
            for (int s = 0; s < 6; s++) {
                const int mid = (lo + hi) >> 1;
                if (incl_w[mid] > t) hi = mid; else lo = mid + 1;
            }
            const int j = wave * kWave + lo;
            const uint32_t k = t - (lo ? incl_w[lo - 1] : 0u);
            const uint32_t xw = s_x0w[j];
            const uint32_t w = xw >> 16, x0 = xw & 0xffffu;
            const uint32_t ty = s_y0[j] + k / w, tx = x0 + k % w;
            const uint64_t key = ((uint64_t)(ty * (uint32_t)cam.gx + tx) << 32) | s_depth[j];
            keys[wbase + t] = key;
            vals[wbase + t] = (uint32_t)(blockIdx.x * kBlock + j);
        }
    }
}

__global__ __launch_bounds__(kBlock) void ranges_kernel(int64_t D, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges)
{
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= D) return;
    uint32_t* r = reinterpret_cast<uint32_t*>(ranges);
    const uint32_t tile = (uint32_t)(keys[idx] >> 32);
    if (idx == 0) r[2 * tile] = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (prev != tile) { r[2 * prev + 1] = (uint32_t)idx; r[2 * tile] = (uint32_t)idx; }
    }
    if (idx == D - 1) r[2 * tile + 1] = (uint32_t)D;
}

hipError_t launch_emit(const Cam& cam, int P, GeomPtrs gp, uint64_t* keys, uint32_t* vals, hipStream_t st)
{
    const int nb = (P + kBlock - 1) / kBlock;
    if (nb > 0) hipLaunchKernelGGL(emit_kernel, dim3(nb), dim3(kBlock), 0, st, cam, P, gp, keys, vals);
    return hipGetLastError();
}

hipError_t launch_ranges(int64_t D, const uint64_t* keys_sorted, uint2* ranges, hipStream_t st)
{
    const int nb = (int)((D + kBlock - 1) / kBlock);
    if (nb > 0) hipLaunchKernelGGL(ranges_kernel, dim3(nb), dim3(kBlock), 0, st, D, keys_sorted, ranges);
    return hipGetLastError();
}

}  // namespace gs
