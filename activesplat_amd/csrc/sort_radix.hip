// sort_radix.hip -- stable (key64, value32) LSD radix sort, 8 bits per pass, restricted to the significant key bits
// (32 depth bits + ceil(log2(#tiles)) tile bits).
//
// Backend of the RADIX binning path only: images of more than 8192 tiles (the LDS tile histogram of tilebin.hip holds no more) or
// gs_set_sort_path(GS_SORT_RADIX) in the tests.  Replaces the device-wide radix sort of the reference's absent CUDA extension
//  -- and, since round 6, the rocPRIM call that stood here in rounds 1-5: nothing on the product path is a library now.
// It is the plain three-kernel pass (no decoupled look-back): at 640 x 480 this path never runs, and the path that does (tilebin.hip) exists
// because a global sort of ~5 M pairs is six passes over 60 MB each whatever the kernel.
//
//   radix_hist_kernel    : one WAVEFRONT per 2048 keys counts its 256 digit values in LDS -> hist[digit][block] (digit-major)
//   radix_rowsum / rowscan: exclusive scan of that matrix in reading order (one workgroup per digit row): start of every (digit, block) run in the output
//   radix_scatter_kernel : the wavefront walks its 2048 keys again, 64 at a time IN INDEX ORDER; lanes with the same digit find each other with
//                          eight ballots, rank = population count of the lower lanes of the match mask, the run's cursor advances by the match
//                          count -> equal digits keep their input order inside the block, the digit-major scan keeps it across blocks: STABLE
//                          (the tie-break of the keys is the Gaussian index the emitter wrote them in)
#include "gs_common.h"

namespace gs {

constexpr int kRadixBits = 8, kRadix = 1 << kRadixBits;
constexpr int kRadixRounds = 32;                         // 64-key rounds per wavefront
constexpr int kRadixChunk = kWave * kRadixRounds;        // 2048 keys per wavefront ("run block"): the unit of the digit-major count matrix
constexpr int kRadixWaves = kBlock / kWave;              // four independent wavefronts per workgroup (occupancy; no workgroup barrier anywhere)
constexpr int kRadixBatch = 8;                           // rounds whose keys are requested together (one memory round trip per 512 keys)
static_assert(kBlock == kRadix, "radix_rowscan_kernel reads one row sum per thread");

__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t D, int shift, int nblocks,
                                                             uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[kRadixWaves][kRadix];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vb = blockIdx.x * kRadixWaves + wave;      // this wavefront's run block
    if (vb >= nblocks) return;                           // (wave-uniform; the kernel has no workgroup barrier)
    uint32_t* h = s_h[wave];
    for (int d = lane; d < kRadix; d += kWave) h[d] = 0u;
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)vb * kRadixChunk;
    for (int r0 = 0; r0 < kRadixRounds; r0 += kRadixBatch) {
        uint64_t k[kRadixBatch];
#pragma unroll
        for (int u = 0; u < kRadixBatch; u++) { const int64_t i = base + (r0 + u) * kWave + lane; k[u] = i < D ? keys[i] : 0ull; }
#pragma unroll
        for (int u = 0; u < kRadixBatch; u++)
            if (base + (r0 + u) * kWave + lane < D) atomicAdd(&h[(uint32_t)(k[u] >> shift) & (kRadix - 1)], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < kRadix; d += kWave) hist[(size_t)d * nblocks + vb] = h[d];
}

// The digit-major count matrix [256][nblocks] -> exclusive scan in reading order, two launches of 256 workgroups (one per digit row):
// row sums, then every row adds the sums of the rows before it and scans itself in coalesced 256-entry pieces.
__global__ __launch_bounds__(kBlock) void radix_rowsum_kernel(const uint32_t* __restrict__ hist, int nblocks, uint32_t* __restrict__ rowsum)
{
    __shared__ uint32_t s_w[kRadixWaves];
    const int tid = threadIdx.x;
    const uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t s = 0;
    for (int i = tid; i < nblocks; i += kBlock) s += row[i];
    s = wave_sum_u32(s);
    if ((tid & 63) == 0) s_w[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) rowsum[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(kBlock) void radix_rowscan_kernel(uint32_t* __restrict__ hist, int nblocks, const uint32_t* __restrict__ rowsum)
{
    __shared__ uint32_t s_w[kRadixWaves];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // start of this digit's rows: the counts of all smaller digits
    uint32_t before = tid < (int)blockIdx.x ? rowsum[tid] : 0u;         // (kBlock == kRadix: one row sum per thread)
    before = wave_sum_u32(before);
    if (lane == 0) s_w[wave] = before;
    __syncthreads();
    uint32_t carry = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    for (int i0 = 0; i0 < nblocks; i0 += kBlock) {
        const int i = i0 + tid;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v, lane);
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        uint32_t off = carry;
        for (int w = 0; w < wave; w++) off += s_w[w];
        if (i < nblocks) row[i] = off + incl - v;
        if (tid == kBlock - 1) s_carry = off + incl;
        __syncthreads();
        carry = s_carry;
    }
}

__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int64_t D,
                                                                int shift, int nblocks, const uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_c[kRadixWaves][kRadix];        // next output position of every digit's run of this run block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vb = blockIdx.x * kRadixWaves + wave;
    if (vb >= nblocks) return;
    uint32_t* s_cur = s_c[wave];
    for (int d = lane; d < kRadix; d += kWave) s_cur[d] = hist[(size_t)d * nblocks + vb];
    __builtin_amdgcn_wave_barrier();
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int64_t base = (int64_t)vb * kRadixChunk;
    for (int r0 = 0; r0 < kRadixRounds; r0 += kRadixBatch) {
        if (base + r0 * kWave >= D) break;               // (uniform)
        uint64_t k[kRadixBatch];
        uint32_t v[kRadixBatch];
#pragma unroll
        for (int u = 0; u < kRadixBatch; u++) {
            const int64_t i = base + (r0 + u) * kWave + lane;
            k[u] = i < D ? keys_in[i] : 0ull; v[u] = i < D ? vals_in[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kRadixBatch; u++) {          // IN INDEX ORDER: round after round, lane after lane
            const bool valid = base + (r0 + u) * kWave + lane < D;
            const uint32_t d = (uint32_t)(k[u] >> shift) & (kRadix - 1);
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < kRadixBits; b++) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                same &= bit ? m : ~m;
            }
            const uint32_t rank = (uint32_t)__popcll(same & lower);
            const uint32_t start = s_cur[d];
            __builtin_amdgcn_wave_barrier();              // every lane has read its run's cursor ...
            if (valid && rank == 0u) s_cur[d] = start + (uint32_t)__popcll(same);      // ... before the run's first lane moves it on
            __builtin_amdgcn_wave_barrier();
            if (valid) { keys_out[start + rank] = k[u]; vals_out[start + rank] = v[u]; }
        }
    }
}

static inline size_t radix_align(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int radix_blocks(int64_t D) { return (int)((D + kRadixChunk - 1) / kRadixChunk); }

size_t sort_temp_bytes(int64_t D, int end_bit)
{
    (void)end_bit;
    const int64_t n = D > 0 ? D : 1;
    return radix_align((size_t)n * 8) + radix_align((size_t)n * 4) + radix_align((size_t)kRadix * radix_blocks(n) * 4) + radix_align(kRadix * 4);
}

hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, int64_t D, int end_bit, hipStream_t st)
{
    if (D <= 0) return hipSuccess;
    if (temp_bytes < sort_temp_bytes(D, end_bit)) return hipErrorInvalidValue;
    char* t = (char*)temp;
    uint64_t* ktmp = (uint64_t*)t; t += radix_align((size_t)D * 8);
    uint32_t* vtmp = (uint32_t*)t; t += radix_align((size_t)D * 4);
    uint32_t* hist = (uint32_t*)t; t += radix_align((size_t)kRadix * radix_blocks(D) * 4);
    uint32_t* rowsum = (uint32_t*)t;
    const int nblocks = radix_blocks(D), passes = (max(end_bit, 0) + kRadixBits - 1) / kRadixBits;
    if (passes == 0) {
        hipError_t e = hipMemcpyAsync(keys_out, keys_in, (size_t)D * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(vals_out, vals_in, (size_t)D * 4, hipMemcpyDeviceToDevice, st);
        return e;
    }
    const uint64_t* ks = keys_in; const uint32_t* vs = vals_in;
    for (int p = 0; p < passes; p++) {
        const bool to_out = ((passes - p) & 1) != 0;     // the last pass writes the caller's output buffers
        uint64_t* kd = to_out ? keys_out : ktmp; uint32_t* vd = to_out ? vals_out : vtmp;
        const int shift = p * kRadixBits;
        const int wgs = (nblocks + kRadixWaves - 1) / kRadixWaves;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(wgs), dim3(kBlock), 0, st, ks, D, shift, nblocks, hist);
        hipLaunchKernelGGL(radix_rowsum_kernel, dim3(kRadix), dim3(kBlock), 0, st, (const uint32_t*)hist, nblocks, rowsum);
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kRadix), dim3(kBlock), 0, st, hist, nblocks, (const uint32_t*)rowsum);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(wgs), dim3(kBlock), 0, st, ks, vs, kd, vd, D, shift, nblocks, (const uint32_t*)hist);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        ks = kd; vs = vd;
    }
    return hipSuccess;
}

}  // namespace gs
