// tilebin.hip -- tile binning without a global sort: count -> scan -> scatter -> per-tile LDS sort.
//
// Replaces the duplicateWithKeys + 64-bit device radix sort + identifyTileRanges work items of the
// reference's absent CUDA extension when every tile list fits LDS; the result
// (point_list, ranges) is identical: per tile, instances in ascending (view depth bits, Gaussian index).
//
// Why not a device-wide radix sort on MI355X: for ~1M (key,value) pairs the 6-pass sort is launch/latency
// bound (198 us measured, 1.8 % of HBM peak).  The instances only need to be GROUPED by tile and ordered
// by depth WITHIN a tile, so:
//   tile_count   : workgroups of 2048 Gaussians walk their rects and count into an LDS-private histogram (LDS integer
//                  atomics); the histogram becomes the workgroup's row of a [chunks][tiles] matrix, whose column-wise
//                  exclusive scan (tile_colscan) gives every (workgroup, tile) the start of its slice of the tile segment.
//                  (Direct global atomics per instance measured 61-140 us for 1.2M instances on 1200 counters.)
//   tile_scan    : exclusive scan of the per-tile totals -> ranges[tile], D, max instances per tile.
//   tile_scatter : same expansion; LDS returning atomics hand out slots inside the reserved slices;
//                  one 8-byte (depth_bits<<32 | id) store per instance.
//   per-tile sort, chosen PER TILE on the device (keys are unique, so the result does not depend on the atomics' arrival order):
//     tile_bucket_sort        : lists of at most 5632 keys (every list of a 640 x 480 frame up to ~2.3 M Gaussians): ONE 512-thread workgroup per
//                               tile, keys in registers; a tile's depths lie in a narrow interval, so a LINEAR map onto up to 2048 bins is a
//                               monotone coarse key: min / max -> one returning LDS atomic per key (bin count + arrival index) -> scan -> scatter
//                               by bin -> rank inside the bin's handful of keys; writes ONLY the ids (point_list).  <BIG>: up to 11264 keys in a
//                               1024-thread workgroup for images of at most 512 tiles (a 1 M-Gaussian map in the reference's 256 x 256 frame).
//                               A tile whose fullest bin exceeds 64 keys (clustered depths + an outlier) takes the comparison network instead,
//                               from the same registers.
//     tile_sort + tile_merge  : longer lists (the planner's multi-view atlas: up to 78 k per tile): register-blocked bitonic run sort of
//                               2048-key chunks, rank merge in LDS up to 16384 keys, pairwise merge passes through global memory beyond.
#include "gs_common.h"


namespace gs {

// one wavefront expands its 64 Gaussians' rects; f(tile, owner_slot, k) is called once per instance
template <class F>
__device__ __forceinline__ void expand_wave(const uint32_t* incl_w, const uint32_t* s_x0w, const uint32_t* s_y0,
                                            int wave, int lane, uint32_t total, int gx, F f)
{
    for (uint32_t t0 = 0; t0 < total; t0 += kWave) {
        const uint32_t t = t0 + lane;
        if (t < total) {
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
            const uint32_t tile = (s_y0[j] + k / w) * (uint32_t)gx + x0 + k % w;
            f(tile, j);
        }
    }
}

template <bool SCATTER, int THREADS>
__global__ __launch_bounds__(THREADS) void tile_bin_kernel(Cam cam, int P, GeomPtrs gp, int tiles, int chunk,
                                                          uint32_t* __restrict__ tile_total,
                                                          uint32_t* __restrict__ tile_base,
                                                          const uint2* __restrict__ ranges,
                                                          unsigned long long* __restrict__ pairs, uint32_t cap, int nchunks)
{
    __shared__ uint32_t s_hist[kMaxLdsTiles];      // count pass: histogram; scatter pass: cursors
    __shared__ uint32_t s_incl[THREADS];
    __shared__ uint32_t s_x0w[THREADS];
    __shared__ uint32_t s_y0[THREADS];
    __shared__ uint32_t s_depth[THREADS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Scatter pass: workgroup b runs on XCD b & 7 and takes chunk (b >> 3) of that XCD's contiguous band of chunks.  The 8-byte pair
  // stores of a chunk go to ~1000 tile segments, ~4 instances (32 bytes) each; the slices that CONSECUTIVE chunks write into a tile's
  // segment are adjacent in memory, so this way the ~4 slices of a 128-byte line meet in ONE write-back L2 and leave it as a whole line,
  // instead of leaving four L2s as four partial lines (2 M Gaussians: 55 -> 33 us; an LDS-staged variant that ordered a chunk's
  // instances by tile before writing them reached 42 us in plain chunk order and nothing on top of this one: removed).  The count pass
  // writes whole matrix rows: plain order (measured faster).
  const int cper = (nchunks + 7) >> 3;
  for (int b = blockIdx.x; b < 8 * cper; b += gridDim.x) {
    const int cid = SCATTER ? (b & 7) * cper + (b >> 3) : b;
    if (cid >= nchunks) continue;
    uint32_t* my_base = tile_base + (size_t)cid * tiles;
    __syncthreads();
    for (int t = tid; t < tiles; t += THREADS) s_hist[t] = SCATTER ? ranges[t].x + my_base[t] : 0u;
    __syncthreads();
    const int cbase = cid * chunk;
    for (int r = 0; r < chunk / THREADS; r++) {
        const int i = cbase + r * THREADS + tid;
        if (cbase + r * THREADS >= P) break;                      // uniform
        uint32_t n = 0, x0 = 0, w = 1, y0 = 0, dbits = 0;
        if (i < P) {
            n = gp.tiles[i];
            if (n) {
                const uint2 rc = gp.rect[i];
                x0 = rc.x & 0xffffu; w = (rc.x >> 16) - x0; y0 = rc.y & 0xffffu;
                if (SCATTER) dbits = gp.depth_bits[i];
            }
        }
        const int gbase = cbase + r * THREADS;
        auto place = [&](uint32_t tile, uint32_t depth_bits, uint32_t id) {
            if (SCATTER) {
                const uint32_t slot = atomicAdd(&s_hist[tile], 1u);
                // cap = capacity of `pairs`: an optimistic launch (workspace sized from the previous frame, true D still in
                // flight) must never write past it -- the host discards and repeats such a frame
                if (slot < cap) pairs[slot] = ((unsigned long long)depth_bits << 32) | id;
            } else {
                atomicAdd(&s_hist[tile], 1u);
            }
        };
        // Small rects (every Gaussian of the wavefront covers at most 16 tiles -- the usual case: 2.4 tiles on average in the bench
        // scenes): every lane walks its OWN rect, one LDS integer atomic (4 clk per 64-lane instruction) per step and no
        // search.  A wavefront that holds a large rect uses the balanced emitter below, whose cost per instance does not depend on
        // the rect (6-step LDS binary search + an integer division per instance: 23 of the count pass's 36 us at 2 M went there).
        if (!__any(n > 16u)) {
            uint32_t kx = 0, ky = 0;
            for (uint32_t k = 0; __any(k < n); k++) {
                if (k < n) {
                    place((y0 + ky) * (uint32_t)cam.gx + x0 + kx, dbits, (uint32_t)i);
                    if (++kx == w) { kx = 0; ++ky; }
                }
            }
            continue;                                             // (uniform per wavefront; the barriers below belong to the emitter)
        }
        if (n) {
            s_x0w[tid] = x0 | (w << 16);
            s_y0[tid] = y0;
            if (SCATTER) s_depth[tid] = dbits;
        }
        const uint32_t incl = wave_inclusive_scan(n, lane);
        s_incl[tid] = incl;
        const uint32_t total = __shfl(incl, 63);
        __builtin_amdgcn_wave_barrier();
        expand_wave(s_incl + wave * kWave, s_x0w, s_y0, wave, lane, total, cam.gx, [&](uint32_t tile, int j) {
            place(tile, SCATTER ? s_depth[j] : 0u, (uint32_t)(gbase + j));
        });
        __builtin_amdgcn_wave_barrier();
    }
    if (!SCATTER) {
        // this chunk's row of the [chunks][tiles] count matrix; tile_colscan_kernel turns the columns into slice bases.  (One global
        // returning atomic per (chunk, tile) on the tile's total -- round 1 -- chains ~1000 same-address atomics per tile at 2 M
        // Gaussians: 13 of the count pass's 31 us.)
        __syncthreads();
        for (int t = tid; t < tiles; t += THREADS) my_base[t] = s_hist[t];
    }
  }
}

// Column-wise exclusive scan of the [chunks][tiles] count matrix, in place: base[c][t] = instances of tile t in the chunks before c;
// tile_total[t] = the column sum.  TW tiles x (1024 / TW) row segments per workgroup (the matrix was just written and sits in L2, so
// short row pieces are fine; measured at 2 M Gaussians / 1200 tiles: TW = 32: 16.3 us, 16: 11.4, 8: 12.0, 4: 18.7).
template <int TW>
__global__ __launch_bounds__(1024) void tile_colscan_kernel(uint32_t* __restrict__ base, int chunks, int tiles, uint32_t* __restrict__ tile_total)
{
    constexpr int SEG = 1024 / TW;
    __shared__ uint32_t s_sum[SEG][TW + 1];
    const int l = threadIdx.x % TW, sgm = threadIdx.x / TW;
    // XCD-banded column groups: the two 64-byte halves of a 128-byte matrix line are rewritten through the same L2
    const int ngroups = (tiles + TW - 1) / TW, gper = (ngroups + 7) >> 3;
    const int grp = (int)(blockIdx.x & 7) * gper + (int)(blockIdx.x >> 3);
    const int t = ((int)(blockIdx.x >> 3) < gper && grp < ngroups) ? grp * TW + l : tiles;
    const int L = (chunks + SEG - 1) / SEG, r0 = min(chunks, sgm * L), r1 = min(chunks, r0 + L);
    uint32_t sum = 0;
    if (t < tiles) {
#pragma unroll 8
        for (int r = r0; r < r1; r++) sum += base[(size_t)r * tiles + t];
    }
    s_sum[sgm][l] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int q = 0; q < sgm; q++) run += s_sum[q][l];
    if (t < tiles) {
#pragma unroll 8
        for (int r = r0; r < r1; r++) {
            const size_t idx = (size_t)r * tiles + t;
            const uint32_t v = base[idx];
            base[idx] = run;
            run += v;
        }
        if (sgm == SEG - 1) tile_total[t] = run;     // the last segment ends with the column sum (empty segments pass it through)
    }
}

// exclusive scan of the per-tile totals: ranges[t] = [start, start+count); counts[0] = D, counts[1] = max.  One workgroup, ONE pass: a thread
// takes ceil(tiles / 1024) consecutive tiles (its own little serial scan), the workgroup scans the thread sums (two barriers in all --
// the loop over 1024-tile slabs this replaces took three per slab)
constexpr int kScanPerThreadMax = 8;                 // tiles <= kMaxLdsTiles = 8192 on this path
__global__ __launch_bounds__(1024) void tile_scan_kernel(const uint32_t* __restrict__ tile_total, int tiles,
                                                         uint2* __restrict__ ranges, uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ host_counts)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_m[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (tiles + 1023) >> 10, t0 = tid * per;
    uint32_t v[kScanPerThreadMax];
    uint32_t sum = 0, vmax = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThreadMax; k++) {
        v[k] = (k < per && t0 + k < tiles) ? tile_total[t0 + k] : 0u;
        sum += v[k];
        vmax = max(vmax, v[k]);
    }
    const uint32_t inc = wave_inclusive_scan(sum, lane);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor(vmax, m));
    if (lane == 63) { s_w[wave] = inc; s_m[wave] = vmax; }
    __syncthreads();
    uint32_t start = inc - sum, total = 0, mx = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) start += s_w[w];
        total += s_w[w];
        mx = max(mx, s_m[w]);
    }
#pragma unroll
    for (int k = 0; k < kScanPerThreadMax; k++) {
        if (k < per && t0 + k < tiles) ranges[t0 + k] = make_uint2(start, start + v[k]);
        start += v[k];
    }
    if (tid == 0) {
        counts[0] = total; counts[1] = mx;
        // mapped pinned host memory: the two counters land on the host without a separate D2H copy in the stream (the kernel's end is the
        // release the host's event wait pairs with: no fence here -- a system-scope fence costs this one-workgroup kernel a PCIe round trip)
        if (host_counts) { host_counts[0] = total; host_counts[1] = mx; }
    }
}

// One workgroup per (tile, chunk): bitonic sort of (depth_bits<<32 | id), REGISTER-BLOCKED: every thread owns E consecutive
// keys.  Steps with stride < E (and the whole first log2(E) levels) are compare-exchanges between a thread's own registers; a
// step with stride >= E pairs each thread with ONE partner thread (t ^ stride/E), whose E keys it needs.
//
// What the first version of this kernel was bound by (2 M Gaussians: 76 us) is the LDS pipe: 36 partner steps, each moving
// every key through LDS (8 x ds_write_b64 + 8 x ds_read_b64 per thread), and a 64-bit compare + four 32-bit selects per
// compare-exchange.  Here
//   * the keys are handled as DOUBLES: the high word is the bit pattern of a positive finite float (the view depth, > 0.2), so
//     the 64-bit pattern is a positive normal double and doubles order exactly like the unsigned keys; a compare-exchange is
//     v_min_f64 + v_max_f64 (full rate on this chip) instead of v_cmp_lt_u64 + 4 x v_cndmask.  Padding = the largest finite
//     double (0x7fefffff ffffffff), which no key reaches;
//   * a thread that has to keep the LARGER key of each pair holds its keys NEGATED for that step: both partners then do the
//     same thing, v = min(v, -partner), and the per-thread direction of a level turns into one sign flip (v_xor on the high
//     word) per key where it changes;
//   * the partner's keys of the 26 steps with strides 1, 2, 4, 8 come by DPP (quad_perm, row_shl / row_shr with bank masks,
//     row_ror:8: one or two moves per word); the 7 steps with strides 16 and 32 go through LDS without a workgroup barrier (same
//     wavefront), only the 3 steps whose partner sits in another wavefront need barriers.
// (inline assembly: __builtin_fmin / fmax add a v_max_f64 x, x canonicalisation in front of every operand; the host emulator of the
// tests defines the two macros as std::fmin / std::fmax)
#ifndef GS_MIN_F64
#define GS_MIN_F64(a, b) ({ double r_; asm("v_min_f64 %0, %1, %2" : "=v"(r_) : "v"(a), "v"(b)); r_; })
#define GS_MAX_F64(a, b) ({ double r_; asm("v_max_f64 %0, %1, %2" : "=v"(r_) : "v"(a), "v"(b)); r_; })
#endif
constexpr unsigned long long kPadKey = 0x7fefffffffffffffull;

__device__ __forceinline__ double key_to_f64(unsigned long long k) { return __builtin_bit_cast(double, k); }
__device__ __forceinline__ unsigned long long f64_to_key(double v) { return __builtin_bit_cast(unsigned long long, v); }
// v with its sign flipped where mask = 0x80000000 (0: unchanged)
__device__ __forceinline__ double f64_flip(double v, uint32_t mask)
{
    return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, v) ^ ((unsigned long long)mask << 32));
}

// the value lane (lane ^ M) holds, M in {1, 2, 4, 8} (inside a row of 16 lanes): DPP, no LDS
template <int M>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x)
{
    const int xi = (int)x;
    // (mov_dpp: no "old" operand to keep alive -- every lane is written, so the result needs no copy of x in front of it)
    if (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp(xi, 0xB1, 0xf, 0xf, true);                    // quad_perm [1,0,3,2]
    if (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp(xi, 0x4E, 0xf, 0xf, true);                    // quad_perm [2,3,0,1]
    if (M == 4) {
        const int t = __builtin_amdgcn_mov_dpp(xi, 0x104, 0xf, 0x5, true);                              // banks 0, 2 read lane + 4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, xi, 0x114, 0xf, 0xA, false);                    // banks 1, 3 read lane - 4
    }
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "in-row strides only");
    return (uint32_t)__builtin_amdgcn_mov_dpp(xi, 0x128, 0xf, 0xf, true);                                // row_ror:8
}
template <int M>
__device__ __forceinline__ double lane_xor_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = lane_xor_u32<M>((uint32_t)b), hi = lane_xor_u32<M>((uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int E, int THREADS>
__device__ __forceinline__ void bitonic_sort_block(double (&v)[E], double* buf, int tid)
{
    const uint32_t base = (uint32_t)tid * E;
    uint32_t neg = 0u;                                            // 0x80000000 while v holds the negated keys
#define GS_DOMAIN(want_neg)                                                      \
    {                                                                            \
        const uint32_t w_ = (want_neg) ? 0x80000000u : 0u, m_ = w_ ^ neg;        \
        _Pragma("unroll") for (int i = 0; i < E; i++) v[i] = f64_flip(v[i], m_); \
        neg = w_;                                                                \
    }
#define GS_SORT2(a, b)                                      \
    {                                                       \
        const double mn_ = GS_MIN_F64(a, b), mx_ = GS_MAX_F64(a, b); \
        a = mn_; b = mx_;                                   \
    }
    // levels that fit inside one thread: the direction of pair (i, i | j) at level kk is ((base + i) & kk) == 0 -- a constant of
    // the unrolled code for kk < E, per thread for kk == E (negated domain for descending threads)
#pragma unroll
    for (int kk = 2; kk <= E; kk <<= 1) {
        if (kk == E) GS_DOMAIN((base & (uint32_t)E) != 0u)
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < E; i++) {
                if ((i & j) != 0) continue;
                if (kk == E || (i & kk) == 0) GS_SORT2(v[i], v[i | j])
                else GS_SORT2(v[i | j], v[i])
            }
        }
    }
    for (uint32_t kk = 2 * E; kk <= (uint32_t)E * THREADS; kk <<= 1) {
        const bool up = (base & kk) == 0;                         // same for all E keys of a thread (kk >= 2E)
        for (uint32_t j = kk >> 1; j >= (uint32_t)E; j >>= 1) {
            const int m = (int)(j / E);                           // partner thread = tid ^ m
            const bool lower = (tid & m) == 0;
            GS_DOMAIN(lower != up)                                // natural domain <=> this thread keeps the smaller key of each pair
            double p[E];
            switch (m) {
#define GS_CASE(M) case M: _Pragma("unroll") for (int i = 0; i < E; i++) p[i] = lane_xor_f64<M>(v[i]); break;
                GS_CASE(1) GS_CASE(2) GS_CASE(4) GS_CASE(8)
#undef GS_CASE
                case 16: case 32:
                    // same wavefront, other row / half: through the wave's own slots of the exchange buffer (LDS serves a wave in
                    // order: no workgroup barrier).  The swap instructions would cost two register copies and a select per word
                    // on the VALU, which is this kernel's bound; the LDS pipe is idle.
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < E; i++) buf[i * THREADS + tid] = v[i];
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < E; i++) p[i] = buf[i * THREADS + (tid ^ m)];
                    __builtin_amdgcn_wave_barrier();
                    break;
                default:                                          // the partner is in another wavefront: through LDS, transposed
                    __syncthreads();                              // (lane-contiguous 8-byte accesses, no bank conflicts)
#pragma unroll
                    for (int i = 0; i < E; i++) buf[i * THREADS + tid] = v[i];
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < E; i++) p[i] = buf[i * THREADS + (tid ^ m)];
                    __syncthreads();                              // the slots are rewritten by wave-local steps that follow
            }
            // the partner is in the opposite domain: -p is its key in this thread's domain
#pragma unroll
            for (int i = 0; i < E; i++) v[i] = GS_MIN_F64(v[i], f64_flip(p[i], 0x80000000u));
        }
        GS_DOMAIN(!up)
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < E; i++) {
                if ((i & j) == 0) GS_SORT2(v[i], v[i | j])
            }
        }
    }
    GS_DOMAIN(false)
#undef GS_DOMAIN
#undef GS_SORT2
}

template <int E, int THREADS>
__device__ __forceinline__ void tile_sort_impl(const uint2 range, uint32_t n, const unsigned long long* pairs,
                                               unsigned long long* pairs_out, uint32_t* __restrict__ point_list,
                                               unsigned long long* buf, int tid)
{
    double k[E];
    const uint32_t base = (uint32_t)tid * E;
    // coalesced global reads straight into the registers: which unsorted key starts in which (thread, slot) does not matter
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t e = (uint32_t)i * THREADS + tid;
        k[i] = key_to_f64(e < n ? pairs[range.x + e] : kPadKey);
    }
    bitonic_sort_block<E, THREADS>(k, reinterpret_cast<double*>(buf), tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; i++) buf[base + i] = f64_to_key(k[i]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t e = (uint32_t)i * THREADS + tid;
        if (e < n) {
            const unsigned long long v = buf[e];
            pairs_out[range.x + e] = v;
            point_list[range.x + e] = (uint32_t)v;
        }
    }
}

// grid = (tiles, chunks): block (t, c) sorts keys [c*CAP, min(n, (c+1)*CAP)) of tile t (pairs -> pairs_out, which may be the same
// array) and writes their ids.  A tile with n <= CAP is finished by this kernel alone; longer lists are completed by
// tile_merge_kernel (up to kSortCapMax keys, in LDS) or by tile_merge_pass_kernel (any length, run by run through global memory).
template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void tile_sort_kernel(const uint2* __restrict__ ranges,
                                                             const unsigned long long* pairs, unsigned long long* pairs_out,
                                                             uint32_t* __restrict__ point_list, uint32_t cap, uint32_t skip_le)
{
    __shared__ unsigned long long s_a[CAP];
    const int tid = threadIdx.x;
    uint2 range = ranges[blockIdx.x];
    range.x = min(range.x, cap); range.y = min(range.y, cap);   // workspace capacity (see tile_bin_kernel)
    const uint32_t start = blockIdx.y * (uint32_t)CAP;
    if (range.y - range.x <= start || range.y - range.x <= skip_le) return;        // uniform (also n == 0; skip_le: lists the bucket kernel sorts)
    range.x += start;
    const uint32_t n = min(range.y - range.x, (uint32_t)CAP);
    constexpr int EMAX = CAP / THREADS;                       // 8
    if (n <= (uint32_t)THREADS * (EMAX / 4)) tile_sort_impl<EMAX / 4, THREADS>(range, n, pairs, pairs_out, point_list, s_a, tid);
    else if (n <= (uint32_t)THREADS * (EMAX / 2)) tile_sort_impl<EMAX / 2, THREADS>(range, n, pairs, pairs_out, point_list, s_a, tid);
    else tile_sort_impl<EMAX, THREADS>(range, n, pairs, pairs_out, point_list, s_a, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Bucket sort of a whole tile list of at most kBucketCap keys in ONE workgroup (lists beyond that: the run sort + merges above).
//
// The keys are (view depth bits << 32 | id); a tile's depths lie in a narrow interval [zlo, zhi], so a LINEAR map of the depth onto
// kBuckets bins is a monotone coarse key for free (three VALU operations per key -- no comparison network):
//   1. every thread loads its E keys into registers; workgroup min / max of the depths;
//   2. bin = floor((z - zlo) * scale): one returning LDS integer atomic per key counts the bin and hands the key its arrival index
//      inside the bin (LDS integer atomics: 4-5 clk per 64-lane instruction);
//   3. exclusive scan of the kBuckets counts -> every bin's slice of the sorted list;
//   4. the keys are scattered into LDS, grouped by bin (bins in depth order, arrival order inside a bin);
//   5. thread p takes position p of that array: its bin's slice holds a handful of keys (n / kBuckets ~ 1 ... 5 on average), the key's
//      rank inside the slice is the number of smaller keys in it (keys are unique), and the key's ID leaves for its final position in
//      point_list (the sorted 64-bit pairs themselves are not written back: nothing downstream reads them -- 20 -> 12 bytes per key).
//      The lanes of a wavefront sit in the same or neighbouring bins: the LDS reads are broadcasts and the stores land in the same few
//      lines.
// Work per key: ~25 VALU + ~10 LDS operations, against 78 compare-exchange stages of the bitonic network (2 M Gaussians, lists of ~4000:
// sort + merge 69 us -> see profiles/README.md).  The result is THE sorted order (unique keys): identical to the network's.
// Clustered depths (a wall at constant depth and one far outlier in the same tile) crowd a bin and the quadratic rank step with it:
// a tile whose fullest bin exceeds kBucketRankMax keys is sorted by the bitonic network instead -- same kernel, keys already in
// registers: runs of 4096 (and the rest) + a rank merge in LDS -- i.e. never slower than the path above by more than the counting.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int kBucketCap = 5632;          // keys per tile list the bucket kernel takes (44 KB of LDS: three workgroups per CU)
constexpr int kBuckets = 2048;           // 16-bit counters, two per LDS word (a list holds < 65536 keys)
constexpr int kBucketThreads = 512;
constexpr int kBucketRankMax = 64;
constexpr int kBucketThreadsBig = 1024;  // lists of kBucketCap < n <= kBucketCapBig keys: eleven keys per thread of a 1024-thread workgroup
constexpr int kBucketCapBig = 11 * kBucketThreadsBig;
constexpr int kBucketsBig = 4096;

template <int E, int T, int NB, int CAP>
__device__ __forceinline__ void bucket_sort_impl(const uint2 range, uint32_t n, const unsigned long long* __restrict__ pairs,
                                                 uint32_t* __restrict__ point_list, unsigned long long* s_keys, uint32_t* s_cnt, uint16_t* s_off,
                                                 float* s_redf, uint32_t* s_redu, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long k[E];
    float zmin = 3.0e38f, zmax = 0.0f;
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t e = (uint32_t)i * T + tid;
        k[i] = e < n ? pairs[range.x + e] : kPadKey;
        if (e < n) {
            const float z = __uint_as_float((uint32_t)(k[i] >> 32));
            zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
        }
    }
    // short lists use fewer bins (fixed cost of zeroing and scanning them): ~2 keys per bin either way
    const int nbins = n <= (uint32_t)NB / 2 ? NB / 4 : (n <= (uint32_t)(NB + NB / 4) ? NB / 2 : NB);
    for (int b = tid; b < nbins / 2; b += T) s_cnt[b] = 0u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { zmin = fminf(zmin, __shfl_xor(zmin, m)); zmax = fmaxf(zmax, __shfl_xor(zmax, m)); }
    if (lane == 0) { s_redf[wave] = zmin; s_redf[T / kWave + wave] = zmax; }
    __syncthreads();
    float zlo = s_redf[0], zhi = s_redf[T / kWave];
#pragma unroll
    for (int w = 1; w < T / kWave; w++) { zlo = fminf(zlo, s_redf[w]); zhi = fmaxf(zhi, s_redf[T / kWave + w]); }
    // bin = min(nbins - 1, (uint)((z - zlo) * scale)): monotone in z whatever the rounding (subtraction, product and conversion are)
    const float span = zhi - zlo;
    const float scale = span > 0.0f ? (float)nbins / span : 0.0f;
    const uint32_t last_bin = (uint32_t)nbins - 1u;
#define GS_BIN(key) min(last_bin, (uint32_t)((__uint_as_float((uint32_t)((key) >> 32)) - zlo) * scale))
    uint32_t arr[(E + 1) / 2];                                  // arrival index inside the bin, two 16-bit values per register (the bin is
                                                                // recomputed from the key: three VALU operations against a register per key)
#pragma unroll
    for (int i = 0; i < E; i++) {
        const uint32_t e = (uint32_t)i * T + tid;
        if ((i & 1) == 0) arr[i >> 1] = 0u;
        if (e < n) {
            const uint32_t b = GS_BIN(k[i]);
            const uint32_t sh = (b & 1u) * 16u;
            arr[i >> 1] |= ((atomicAdd(&s_cnt[b >> 1], 1u << sh) >> sh) & 0xffffu) << (16 * (i & 1));
        }
    }
    __syncthreads();
    // exclusive scan of the counts (kBuckets / T consecutive bins per thread) and the fullest bin
    constexpr int PB = NB / T;
    static_assert(PB % 2 == 0, "a thread scans whole counter words");
    uint32_t c[PB], sum = 0, cmax = 0;
#pragma unroll
    for (int j = 0; j < PB; j++) { c[j] = tid * PB < nbins ? (s_cnt[(tid * PB + j) >> 1] >> (16 * (j & 1))) & 0xffffu : 0u; sum += c[j]; cmax = max(cmax, c[j]); }
    const uint32_t incl = wave_inclusive_scan(sum, lane);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor(cmax, m));
    if (lane == 63) s_redu[wave] = incl;
    if (lane == 0) s_redu[T / kWave + wave] = cmax;
    __syncthreads();
    uint32_t run = incl - sum, fullest = 0;
#pragma unroll
    for (int w = 0; w < T / kWave; w++) { if (w < wave) run += s_redu[w]; fullest = max(fullest, s_redu[T / kWave + w]); }
    if (tid * PB < nbins) {
#pragma unroll
        for (int j = 0; j < PB; j++) { s_off[tid * PB + j] = (uint16_t)run; run += c[j]; }
        if (tid * PB + PB == nbins) s_off[nbins] = (uint16_t)run;
    }
    __syncthreads();
    if (fullest <= (uint32_t)kBucketRankMax) {
#pragma unroll
        for (int i = 0; i < E; i++) {
            const uint32_t e = (uint32_t)i * T + tid;
            if (e < n) s_keys[s_off[GS_BIN(k[i])] + ((arr[i >> 1] >> (16 * (i & 1))) & 0xffffu)] = k[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; i++) {
            const uint32_t p = (uint32_t)i * T + tid;
            if (p < n) {
                const unsigned long long key = s_keys[p];
                const uint32_t b = GS_BIN(key);
                const uint32_t lo = s_off[b], hi = s_off[b + 1];
                uint32_t rank = lo;
                for (uint32_t j = lo; j < hi; j++) rank += s_keys[j] < key ? 1u : 0u;
                point_list[range.x + rank] = (uint32_t)key;
            }
        }
        return;
    }
#undef GS_BIN
    // ---- crowded bins: the comparison network on the keys in the registers (uniform branch) ----
    constexpr int EA = E > 8 ? 8 : E;                              // run A: slots [0, EA * T) of the load order; run B: the rest (E = 11 only)
    double va[EA];
#pragma unroll
    for (int i = 0; i < EA; i++) va[i] = key_to_f64(k[i]);
    bitonic_sort_block<EA, T>(va, reinterpret_cast<double*>(s_keys), tid);
    const uint32_t nA = min(n, (uint32_t)(EA * T));
    if (E <= 8) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EA; i++) s_keys[(uint32_t)tid * EA + i] = f64_to_key(va[i]);
        __syncthreads();
        for (uint32_t e = tid; e < n; e += T) {
            const unsigned long long v = s_keys[e];
            point_list[range.x + e] = (uint32_t)v;
        }
        return;
    }
    constexpr int EB = 4;                                          // (kBucketCap - 8 T) / T = 3 keys per thread, as a power of two
    static_assert(CAP <= (8 + EB) * T && E <= 8 + EB, "run B holds the keys beyond run A");
    double vb[EB];
#pragma unroll
    for (int i = 0; i < EB; i++) vb[i] = key_to_f64((EA + i) < E ? k[(EA + i) < E ? (EA + i) : 0] : kPadKey);
    __syncthreads();                                               // run A is in the registers: its exchange buffer is free again
    bitonic_sort_block<EB, T>(vb, reinterpret_cast<double*>(s_keys), tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EA; i++) s_keys[(uint32_t)tid * EA + i] = f64_to_key(va[i]);
#pragma unroll
    for (int i = 0; i < EB; i++) if ((uint32_t)(EA * T + tid * EB + i) < (uint32_t)CAP) s_keys[EA * T + (uint32_t)tid * EB + i] = f64_to_key(vb[i]);
    __syncthreads();
    // rank merge of the two sorted runs A = s_keys[0, nA), B = s_keys[EA T, EA T + nB): keys are unique
    const uint32_t nB = n - nA;
    for (uint32_t e = tid; e < n; e += T) {
        const bool in_a = e < nA;
        const unsigned long long key = in_a ? s_keys[e] : s_keys[EA * T + (e - nA)];
        const uint32_t base = in_a ? (uint32_t)(EA * T) : 0u, len = in_a ? nB : nA;
        uint32_t lo = 0, hi = len;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_keys[base + mid] < key) lo = mid + 1; else hi = mid;
        }
        const uint32_t rank = (in_a ? e : e - nA) + lo;
        point_list[range.x + rank] = (uint32_t)key;
    }
}

// (launch bound: THREE workgroups per CU -- 24 wavefronts, 80 registers; the natural allocation of 92 admits two, i.e. 512 slots for the 1200 tiles of
// a 640 x 480 image: 28.7 -> 25.4 us at 2 M Gaussians.  Two tiles per workgroup with both lists requested up front: 123 registers, 34.7 us.)
// BIG (round 4): lists of kBucketCap < n <= kBucketCapBig keys -- a 1 M-Gaussian map in the reference's 256 x 256 frame: ~9.4 k per tile -- in a
// 1024-thread workgroup (eleven keys per thread, 4096 bins, 106 KB of LDS: one workgroup per CU, which is what a 256-tile image has) instead
// of 4096-key bitonic runs + the LDS rank merge.
template <bool BIG>
__global__ __launch_bounds__(BIG ? kBucketThreadsBig : kBucketThreads, BIG ? 4 : 6) void tile_bucket_sort_kernel(
    const uint2* __restrict__ ranges, const unsigned long long* __restrict__ pairs, uint32_t* __restrict__ point_list, uint32_t cap)
{
    constexpr int T = BIG ? kBucketThreadsBig : kBucketThreads, NB = BIG ? kBucketsBig : kBuckets, CAP = BIG ? kBucketCapBig : kBucketCap;
    __shared__ unsigned long long s_keys[CAP];
    __shared__ uint32_t s_cnt[NB / 2];
    __shared__ uint16_t s_off[NB + 2];
    __shared__ float s_redf[2 * T / kWave];
    __shared__ uint32_t s_redu[2 * T / kWave];
    const int tid = threadIdx.x;
    uint2 range = ranges[blockIdx.x];
    range.x = min(range.x, cap); range.y = min(range.y, cap);   // workspace capacity (see tile_bin_kernel)
    const uint32_t n = range.y - range.x;
    if (n == 0 || n > (uint32_t)CAP || (BIG && n <= (uint32_t)kBucketCap)) return;             // uniform (BIG: the shorter lists are the other instantiation's)
    if (BIG) { bucket_sort_impl<11, T, NB, CAP>(range, n, pairs, point_list, s_keys, s_cnt, s_off, s_redf, s_redu, tid); return; }
    if (n <= 2u * T) bucket_sort_impl<2, T, NB, CAP>(range, n, pairs, point_list, s_keys, s_cnt, s_off, s_redf, s_redu, tid);
    else if (n <= 4u * T) bucket_sort_impl<4, T, NB, CAP>(range, n, pairs, point_list, s_keys, s_cnt, s_off, s_redf, s_redu, tid);
    else if (n <= 8u * T) bucket_sort_impl<8, T, NB, CAP>(range, n, pairs, point_list, s_keys, s_cnt, s_off, s_redf, s_redu, tid);
    else bucket_sort_impl<11, T, NB, CAP>(range, n, pairs, point_list, s_keys, s_cnt, s_off, s_redf, s_redu, tid);
}

// Tiles with CHUNK < n <= CAP: the CHUNK-sized sorted runs left by tile_sort_kernel are merged by RANK: the whole
// list sits in LDS, every key adds to its index inside its own run the number of smaller keys in each other run
// (binary search in LDS; keys are unique) and is written straight to its final position.
template <int CAP, int CHUNK, int THREADS>
__global__ __launch_bounds__(THREADS) void tile_merge_kernel(const uint2* __restrict__ ranges,
                                                              unsigned long long* __restrict__ pairs,
                                                              uint32_t* __restrict__ point_list, uint32_t cap, uint32_t skip_le)
{
    __shared__ unsigned long long s_k[CAP];
    const int tid = threadIdx.x;
    uint2 range = ranges[blockIdx.x];
    range.x = min(range.x, cap); range.y = min(range.y, cap);
    const uint32_t n = range.y - range.x;
    if (n <= (uint32_t)CHUNK || n > (uint32_t)CAP || n <= skip_le) return;     // uniform: already final / longer than the caller assumed / bucket-sorted
    for (uint32_t i = tid; i < n; i += THREADS) s_k[i] = pairs[range.x + i];
    __syncthreads();
    const uint32_t nruns = (n + CHUNK - 1) / CHUNK;
    for (uint32_t i = tid; i < n; i += THREADS) {
        const unsigned long long key = s_k[i];
        const uint32_t own = i / CHUNK;
        uint32_t rank = i - own * CHUNK;
        for (uint32_t r = 0; r < nruns; r++) {
            if (r == own) continue;
            const uint32_t b0 = r * CHUNK, len = min((uint32_t)CHUNK, n - b0);
            uint32_t lo = 0, hi = len;                          // first position with s_k[b0 + pos] > key
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_k[b0 + mid] < key) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        pairs[range.x + rank] = key;
        point_list[range.x + rank] = (uint32_t)key;
    }
}

// Lists longer than kSortCapMax (the planner's 120 x 150 views of a large map: tens of thousands of records per tile): the sorted
// runs are merged PAIRWISE, pass by pass, through global memory (src -> dst, run length doubling).  grid = (tiles, output blocks of
// OUT keys); OUT divides the run length, so a block lies inside one pair of runs (A, B).  Two merge-path searches (first wavefront:
// start diagonal, second: end diagonal) give the stretches of A and B that make up the block's OUT outputs; they are loaded into LDS,
// every key finds its rank by a binary search in the other stretch (keys are unique), the block is permuted inside LDS and leaves
// in list order.  A pair whose B is empty (odd run out, or a tile that was finished passes ago) is copied.
template <int OUT, int THREADS>
__global__ __launch_bounds__(THREADS) void tile_merge_pass_kernel(const uint2* __restrict__ ranges, const unsigned long long* __restrict__ src,
                                                                   unsigned long long* __restrict__ dst, uint32_t* __restrict__ point_list,
                                                                   uint32_t run, uint32_t cap)
{
    __shared__ unsigned long long s_in[OUT];
    __shared__ unsigned long long s_out[OUT];
    __shared__ uint32_t s_split[2];
    const int tid = threadIdx.x;
    uint2 range = ranges[blockIdx.x];
    range.x = min(range.x, cap); range.y = min(range.y, cap);
    const uint32_t n = range.y - range.x;
    const uint32_t o0 = blockIdx.y * (uint32_t)OUT;
    if (o0 >= n) return;                                        // uniform
    const uint32_t o1 = min(n, o0 + (uint32_t)OUT);
    const uint32_t pair0 = o0 / (2u * run) * (2u * run);
    const uint32_t lenA = min(run, n - pair0), lenB = min(run, n - pair0 - lenA);
    const unsigned long long* A = src + range.x + pair0;
    const unsigned long long* B = A + lenA;
    if (tid < 2 * kWave) {
        // merge path: how many of the first d outputs of merge(A, B) come from A = the smallest i in [lo, hi] for which
        // A[i] < B[d-1-i] no longer holds (it holds for small i, then never again).  64 probes per round instead of one: the
        // interval shrinks 64-fold per global-memory round trip (a 128 k-key run: 3 rounds instead of 17)
        const int lane = tid & (kWave - 1);
        const uint32_t d = (tid < kWave ? o0 : o1) - pair0;
        uint32_t lo = d > lenB ? d - lenB : 0u, hi = min(d, lenA);
        while (lo < hi) {                                       // wave-uniform
            const uint32_t span = hi - lo;
            const uint32_t q = lo + (uint32_t)(((uint64_t)span * (uint32_t)lane) >> 6);      // lo <= q <= hi - 1, non-decreasing in lane
            const int c = (int)__popcll(__ballot(A[q] < B[d - 1 - q]));                       // the first c probes hold
            const uint32_t q_prev = lo + (uint32_t)(((uint64_t)span * (uint32_t)max(c - 1, 0)) >> 6);
            const uint32_t q_c = lo + (uint32_t)(((uint64_t)span * (uint32_t)min(c, kWave - 1)) >> 6);
            if (c == 0) hi = lo;
            else if (c == kWave) lo = q_prev + 1;
            else { lo = q_prev + 1; hi = q_c; }
        }
        if (lane == 0) s_split[tid / kWave] = lo;
    }
    __syncthreads();
    const uint32_t d0 = o0 - pair0, d1 = o1 - pair0;
    const uint32_t a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
    const uint32_t na = a1 - a0, nb = b1 - b0, m = na + nb;     // m = o1 - o0
    for (uint32_t i = tid; i < m; i += THREADS) s_in[i] = i < na ? A[a0 + i] : B[b0 + (i - na)];
    __syncthreads();
    for (uint32_t i = tid; i < m; i += THREADS) {
        const unsigned long long key = s_in[i];
        const bool in_a = i < na;
        const uint32_t base = in_a ? na : 0u, len = in_a ? nb : na;
        uint32_t lo = 0, hi = len;                              // keys of the other stretch below this one
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_in[base + mid] < key) lo = mid + 1; else hi = mid;
        }
        s_out[(in_a ? i : i - na) + lo] = key;
    }
    __syncthreads();
    for (uint32_t i = tid; i < m; i += THREADS) {
        const unsigned long long key = s_out[i];
        dst[range.x + o0 + i] = key;
        if (point_list) point_list[range.x + o0 + i] = (uint32_t)key;
    }
}

// 1024 threads x kBinChunk Gaussians per workgroup (measured best of 256/512/1024 threads x 1024..4096 Gaussians)
static void bin_config(int P, int& threads, int& chunk)
{
    // (4096 / 8192 Gaussians per workgroup: the count pass gains a third at 2 M, the scatter pass loses as much -- fewer workgroups)
    (void)P;
    threads = 1024; chunk = kBinChunk;
}

template <bool SCATTER>
static void launch_bin(int threads, int nb, hipStream_t st, Cam cam, int P, GeomPtrs gp, int tiles, int chunk,
                       uint32_t* tile_total, uint32_t* tile_base, const uint2* ranges, unsigned long long* pairs, uint32_t cap)
{
    const int nchunks = nb;
    nb = 8 * ((nb + 7) / 8);                              // XCD bands of chunks
    hipLaunchKernelGGL((tile_bin_kernel<SCATTER, 1024>), dim3(nb), dim3(1024), 0, st, cam, P, gp, tiles, chunk, tile_total, tile_base, ranges, pairs, cap, nchunks);
}

hipError_t launch_tile_count(const Cam& cam, int P, GeomPtrs gp, uint32_t* tile_total, uint32_t* tile_base,
                             uint2* ranges, uint32_t* d_counts, uint32_t* host_counts, hipStream_t st)
{
    const int tiles = cam.gx * cam.gy;       // tile_total was zeroed by the preprocess stage
    int threads, chunk; bin_config(P, threads, chunk);
    const int nb = (P + chunk - 1) / chunk;
    if (nb > 0) launch_bin<false>(threads, nb, st, cam, P, gp, tiles, chunk, tile_total, tile_base, nullptr, nullptr, 0u);
    hipLaunchKernelGGL(tile_colscan_kernel<16>, dim3(8 * (((tiles + 15) / 16 + 7) / 8)), dim3(1024), 0, st, tile_base, nb, tiles, tile_total);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, tile_total, tiles, ranges, d_counts, host_counts);
    return hipGetLastError();
}

hipError_t launch_tile_scatter_sort(const Cam& cam, int P, GeomPtrs gp, uint32_t* tile_base, const uint2* ranges,
                                    uint32_t max_tile_instances, unsigned long long* pairs, unsigned long long* pairs_alt,
                                    uint32_t* point_list, uint32_t cap, hipStream_t st)
{
    const int tiles = cam.gx * cam.gy;
    int threads, chunk; bin_config(P, threads, chunk);
    const int nb = (P + chunk - 1) / chunk;
    if (nb > 0) launch_bin<true>(threads, nb, st, cam, P, gp, tiles, chunk, nullptr, tile_base, ranges, pairs, cap);
    // Lists of at most 2048 keys: one 2048-key sort per tile.  Longer lists: 4096-key runs sorted with 16 keys per thread (the same 36
    // partner steps as a 2048-key sort; measured at 2 M Gaussians, lists of ~4000: 53 us + 15 us of merging against 45 + 31 us with
    // 2048-key runs), then the runs of a tile are merged -- by rank inside LDS up to kSortCapMax keys, pass by pass through global
    // memory beyond (`pairs` <-> `pairs_alt`; the run sort writes into whichever of the two leaves the final pass's output in `pairs`).
    constexpr int kBigChunk = 2 * kSortChunk;
    // Lists of at most kBucketCap keys: the bucket sort, one workgroup per tile.  The choice is made PER TILE on the device (the caller's
    // max_tile_instances may be the capacity guess of an optimistic launch): when longer lists are possible the run sort + merges below
    // are launched as well and skip the tiles the bucket kernel took.
    // (lists of up to kBucketCapBig keys in images of at most 512 tiles -- one 1024-thread workgroup per CU, two rounds at most --: the big instantiation)
    const bool big = max_tile_instances > (uint32_t)kBucketCap && max_tile_instances <= (uint32_t)kSortCapMax && tiles <= 512;
    const uint32_t skip_le = max_tile_instances <= (uint32_t)kSortCapMax ? (uint32_t)(big ? kBucketCapBig : kBucketCap) : 0u;
    if (skip_le) {
        hipLaunchKernelGGL(tile_bucket_sort_kernel<false>, dim3(tiles), dim3(kBucketThreads), 0, st, ranges, pairs, point_list, cap);
        if (max_tile_instances <= (uint32_t)kBucketCap) return hipGetLastError();
        if (big) {
            hipLaunchKernelGGL(tile_bucket_sort_kernel<true>, dim3(tiles), dim3(kBucketThreadsBig), 0, st, ranges, pairs, point_list, cap);
            if (max_tile_instances <= (uint32_t)kBucketCapBig) return hipGetLastError();
        }
    }
    if (max_tile_instances <= (uint32_t)kSortChunk) {
        hipLaunchKernelGGL((tile_sort_kernel<kSortChunk, 256>), dim3(tiles, 1), dim3(256), 0, st, ranges, pairs, pairs, point_list, cap, 0u);
        return hipGetLastError();
    }
    const unsigned chunks = (max_tile_instances + kBigChunk - 1) / kBigChunk;
    if (max_tile_instances <= (uint32_t)kSortCapMax) {
        hipLaunchKernelGGL((tile_sort_kernel<kBigChunk, 256>), dim3(tiles, chunks), dim3(256), 0, st, ranges, pairs, pairs, point_list, cap, skip_le);
        // the list sits in LDS: the smaller capacity keeps two workgroups resident per CU
        if (max_tile_instances > 8192)
            hipLaunchKernelGGL((tile_merge_kernel<kSortCapMax, kBigChunk, 1024>), dim3(tiles), dim3(1024), 0, st, ranges, pairs, point_list, cap, skip_le);
        else if (max_tile_instances > (uint32_t)kBigChunk)
            hipLaunchKernelGGL((tile_merge_kernel<8192, kBigChunk, 1024>), dim3(tiles), dim3(1024), 0, st, ranges, pairs, point_list, cap, skip_le);
        return hipGetLastError();
    }
    constexpr int kOut = kSortChunk;
    int passes = 0;
    for (uint64_t r = kBigChunk; r < max_tile_instances; r <<= 1) passes++;
    unsigned long long* cur = (passes & 1) ? pairs_alt : pairs;
    hipLaunchKernelGGL((tile_sort_kernel<kBigChunk, 256>), dim3(tiles, chunks), dim3(256), 0, st, ranges, pairs, cur, point_list, cap, 0u);
    const unsigned blocks = (max_tile_instances + kOut - 1) / kOut;
    uint64_t r = kBigChunk;
    for (int p = 0; p < passes; p++, r <<= 1) {
        unsigned long long* nxt = cur == pairs ? pairs_alt : pairs;
        hipLaunchKernelGGL((tile_merge_pass_kernel<kOut, 256>), dim3(tiles, blocks), dim3(256), 0, st, ranges, cur, nxt,
                           p == passes - 1 ? point_list : nullptr, (uint32_t)r, cap);
        cur = nxt;
    }
    return hipGetLastError();
}

}  // namespace gs
