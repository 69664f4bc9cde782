// blend.hip -- per-tile front-to-back alpha blend (forward) and back-to-front gradient replay (backward).
//
// Replaces the "render" forward/backward work items of the reference's absent CUDA extension
// (contract SURVEY App. A.2; consumers src/mapper/splatam/splatam.py:208-212,430-431).
//
// MI355X design (wave-64 first, not a 32-wide warp tiling):
//   * one 16x16 tile per 256-thread workgroup, but the 4 wavefronts are INDEPENDENT: each owns one 8x8
//     pixel quadrant (lane = pixel), walks the tile's depth-sorted instance list on its own and never
//     meets a workgroup barrier -- a quadrant that saturates (T < 1e-4) or whose contributors end early
//     simply finishes.  Sharing a workgroup keeps the four walkers of a tile on one CU, so the 48-byte
//     record gathers of three of them hit that CU's L1.
//   * 64 records at a time: every lane gathers one record, stages it in the wave's private LDS slice (3 KiB)
//     and tests ITS alpha>=1/255 bounding box against the four 4x4-pixel blocks of the quadrant (scan_blocks:
//     the tests are wave masks in scalar registers, no branches); per block an mbcnt rank of its mask turns
//     the hits into a LIST of record slots in LDS.  The 16 lanes of a block (one DPP row) then walk only their
//     own list -- four record streams per wavefront, no scalar bookkeeping in the loop.  Skipped records can
//     never pass the alpha>=1/255 test, so the result is identical to evaluating all of them.  (The backward
//     stages only the records that hit and runs its phases on nearly full staging: see there.)
//   * the loop is software-pipelined: ids two chunks ahead, records one chunk ahead.
//   * blockIdx -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8, so XCD x is
//     given the contiguous band of tiles [x*ceil(T/8), (x+1)*ceil(T/8)) and neighbouring tiles (which
//     share Gaussians) reuse records in one 4 MiB L2.
#include "gs_common.h"
#include <atomic>

namespace gs {

constexpr int kFwdStreams = 4;        // record streams per wavefront in the forward (8 streams of 4x2 pixels measured the same)

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTmin = 0.0001f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr uint32_t kNoId = 0xffffffffu;

struct TileCtx {
    int tile, tx, ty, px, py, quad, seg;
    float qx0, qy0, pxf, pyf;
    bool inside;
};

// Workgroup b runs on XCD b & 7 (round-robin dispatch) and takes entry (b >> 3) of that XCD's contiguous band of tiles (neighbouring
// tiles share Gaussians: their records are reused in one 4 MiB L2); a tile is NW-wavefront workgroups of quadrant walkers -- four
// 8 x 8 quadrants, or four quadrants x `segments` list segments (the few-tile backward: 2 or 3).
template <int NW>
__device__ __forceinline__ bool tile_ctx_at(const Cam& cam, unsigned block, int wave, int lane, TileCtx& c, int segments = 0)
{
    const int G = (segments > 1 ? 4 * segments : 4) / NW;    // workgroups per tile
    const int ntiles = cam.gx * cam.gy, per = (ntiles + 7) >> 3;
    const int idx = (int)(block >> 3);
    c.tile = (int)(block & 7) * per + idx / G;
    if (idx / G >= per || c.tile >= ntiles) return false;
    int quad = (idx % G) * NW + wave;
    c.seg = segments > 1 ? quad >> 2 : 0;                    // (list segments: walkers 0-3 = front segment of the four quadrants, 4-7 = the next, ...)
    if (segments > 1) quad &= 3;
    c.tx = c.tile % cam.gx; c.ty = c.tile / cam.gx; c.quad = quad;
    const int qx = c.tx * kTile + (quad & 1) * kQuad, qy = c.ty * kTile + (quad >> 1) * kQuad;
    c.px = qx + (lane & 7); c.py = qy + (lane >> 3);
    c.qx0 = (float)qx; c.qy0 = (float)qy; c.pxf = (float)c.px; c.pyf = (float)c.py;
    c.inside = c.px < cam.W && c.py < cam.H;
    return true;
}

template <int NW>
__device__ __forceinline__ bool tile_ctx_nw(const Cam& cam, int wave, int lane, TileCtx& c, int segments = 0)
{
    return tile_ctx_at<NW>(cam, blockIdx.x, wave, lane, c, segments);
}

// does the record's alpha-visible box overlap the 8x8 quadrant at pixel origin (qx0,qy0)?
__device__ __forceinline__ bool quadrant_hit(const float4& q0, const float4& q2, float qx0, float qy0)
{
    const float ex = q2.z, ey = q2.w;
    return ex >= 0.0f && (q0.x + ex >= qx0) && (q0.x - ex <= qx0 + 7.0f) && (q0.y + ey >= qy0) && (q0.y - ey <= qy0 + 7.0f);
}

// LDS staging form: the conic pre-scaled so that the inner loop is  p = (A dx + B dy) dx + C dy dy ; G = 2^p
__device__ __forceinline__ void stage_record(float4* s0, float4* s1, float4* s2, int lane, const float4& q0,
                                             const float4& q1, const float4& q2, uint32_t id)
{
    s0[lane] = make_float4(q0.x, q0.y, -0.5f * kLog2e * q0.z, -kLog2e * q0.w);
    s1[lane] = make_float4(-0.5f * kLog2e * q1.x, q1.y, q1.z, q1.w);
    s2[lane] = make_float4(q2.x, q2.y, __uint_as_float(id), 0.0f);
}

// does the record's alpha-visible box overlap the 4x4 sub-block at pixel origin (x0,y0)?
__device__ __forceinline__ bool subblock_hit(const float4& q0, const float4& q2, float x0, float y0)
{
    const float ex = q2.z, ey = q2.w;
    return ex >= 0.0f && (q0.x + ex >= x0) && (q0.x - ex <= x0 + 3.0f) && (q0.y + ey >= y0) && (q0.y - ey <= y0 + 3.0f);
}

// pop the lowest / highest set bit of a wave-uniform mask; returns its index -- or kWave, the SENTINEL slot of the
// staging arrays, when the mask is empty: that slot holds a record with opacity 0, whose alpha fails the 1/255 test
// for every pixel, so a stream that has run dry needs no validity flag in the inner loop
__device__ __forceinline__ int pop_low(unsigned long long& m)
{
    const int j = m ? __ffsll(m) - 1 : kWave;
    m &= m - 1;
    return j;
}
__device__ __forceinline__ int pop_high(unsigned long long& m)
{
    const int j = m ? 63 - __clzll(m) : kWave;
    m &= ~(1ull << (j & 63));
    return j;
}
__device__ __forceinline__ void write_sentinel(float4* s0, float4* s1, float4* s2, int lane)
{
    if (lane == 0) {
        s0[kWave] = make_float4(0.f, 0.f, 0.f, 0.f);
        s1[kWave] = make_float4(0.f, 0.f, 0.f, 0.f);       // .y = opacity 0
        s2[kWave] = make_float4(0.f, 0.f, __uint_as_float(0u), 0.f);
    }
    __builtin_amdgcn_wave_barrier();
}

// SCAN of a 64-record chunk (one record per lane): which of the quadrant's four 4x4 blocks does the record's alpha-visible box meet?  g0..g3
// (wave-uniform) switch a block off (forward: all its pixels have stopped; backward: no contributor of the block lies this deep).  The
// tests are wave masks in scalar registers -- the blocks share their column / row halves -- and only the lane's four flag bits are vector
// work.  Returns the flags (bit r: block r) and the mask of lanes with any.
__device__ __forceinline__ unsigned scan_blocks(const float4& q0, const float4& q2, bool valid, float qx0, float qy0, bool g0, bool g1, bool g2, bool g3,
                                                unsigned long long& m_any, unsigned long long* masks = nullptr)
{
    const float ex = q2.z, ey = q2.w;
    const float xl = q0.x - ex, xh = q0.x + ex, yl = q0.y - ey, yh = q0.y + ey;
    const unsigned long long vis = __ballot(valid) & __ballot(ex >= 0.0f);
    const unsigned long long cx0 = __ballot(xh >= qx0) & __ballot(xl <= qx0 + 3.0f), cx1 = __ballot(xh >= qx0 + 4.0f) & __ballot(xl <= qx0 + 4.0f + 3.0f);
    const unsigned long long cy0 = vis & __ballot(yh >= qy0) & __ballot(yl <= qy0 + 3.0f), cy1 = vis & __ballot(yh >= qy0 + 4.0f) & __ballot(yl <= qy0 + 4.0f + 3.0f);
    const unsigned long long b0 = g0 ? cx0 & cy0 : 0ull, b1 = g1 ? cx1 & cy0 : 0ull, b2 = g2 ? cx0 & cy1 : 0ull, b3 = g3 ? cx1 & cy1 : 0ull;
    m_any = (b0 | b1) | (b2 | b3);
    if (masks) { masks[0] = b0; masks[1] = b1; masks[2] = b2; masks[3] = b3; }
    return (__builtin_amdgcn_inverse_ballot_w64(b0) ? 1u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b1) ? 2u : 0u) |
           (__builtin_amdgcn_inverse_ballot_w64(b2) ? 4u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b3) ? 8u : 0u);
}
constexpr int kRoundSlack = 4;         // a round is closed after a chunk of n hit records if cnt + n + this many would not fit in the 64 staging slots

// ---------------------------------------------------------------------------------------------------
// Forward: the quadrant's 64 lanes form NS streams of 64/NS lanes (NS = 4: 4x4 pixel blocks,
// NS = 8: 4x2 blocks), every stream walks ITS OWN list of the staged records whose alpha-visible box overlaps its
// block.  With sigma ~ 1 px splats a record touches 2.2 of the 4 (3.0 of the 8) blocks of a quadrant, so the loop
// makes 0.66x (0.59x) the trips of the one-record-per-wave walk (simulated on BASELINE configs[1]).
// The per-stream lists live in LDS: while staging a chunk, each lane whose record hits stream s writes its lane
// index at position mbcnt(ballot_s) of list s (lists pre-filled with the sentinel slot); in the loop a lane reads
// two list entries with ONE 16-bit LDS read -- no scalar pop sequences, which is what made a four-stream forward
// lose before (four s_ff1/s_andn2/s_cselect chains per trip for ~22 VALU of blending).
// ---------------------------------------------------------------------------------------------------
// SEG (segmented compositing for images of a few tiles with very long lists -- the planner's 120 x 150 views over a million
// Gaussians are 80 tiles of up to ~80 k records: 320 wavefronts on a 5120-wavefront machine).  Alpha compositing is
// associative, so a tile's list is cut into gridDim.y segments that run in parallel:
//   SEG = 1: every (tile, segment) workgroup multiplies up its segment's transmittance T_seg per pixel (alpha tests only);
//   SEG = 2: it composites its segment starting from T_in = product of the earlier segments' T_seg, with the normal stop
//            rule (a pixel has stopped before this segment  <=>  T_in < 1e-4, because up to the stop the running T IS that
//            product) and adds its colour / depth sums to the zero-initialised images with atomics; the last segment of a
//            pixel that is not yet stopped at entry writes final_T, opacity and T*bg; n_contrib is an atomic max.
//   SEG = 0: the whole list in one workgroup (the normal path; its code is untouched by the other two).
// (images of at most 256 tiles take blend_forward_pc_kernel below, which also records the states the segmented backward resumes from)
template <bool DEPTH_SQ, int NS, int SEG, int NW>     // DEPTH_SQ: also accumulate sum z^2 alpha T (third channel of the reference's depth/silhouette pass)
__global__ __launch_bounds__(NW * kWave) void blend_forward_streams_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ geom, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_depth_sq, uint32_t cap, float* __restrict__ seg_T, float* __restrict__ split_state, uint32_t P,
    float4* __restrict__ zero_fill)
{
    constexpr int LS = kWave / NS;          // lanes per stream
    constexpr int BH = LS / 4;              // block = 4 x BH pixels
    __shared__ float4 s_rec[NW][3][kWave + 1];           // + the sentinel slot
    // NS rows of 64 entries per wave (+16 bytes so that the look-ahead read behind the last row stays inside the wave's slab)
    __shared__ __attribute__((aligned(16))) uint8_t s_list[NW][NS * kWave + 16];
    static_assert(NS == 4 && kWave / NS == 16, "the scan tests the four 4x4 blocks of a quadrant");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // side job (SEG = 0 only): this workgroup's slice of the backward's gradient records is zero-filled here instead of by a fill launch
    // in front of the backward -- plain fire-and-forget 16-byte stores, kFillPerStep of them per trip of the chunk loop below.  (Issued
    // all at once at the top of the kernel -- the first version -- they are a 128 MB memset at 2 M Gaussians that every CU's store
    // queue has to drain BEFORE its wavefronts get to blend: ~25 us of the forward; interleaved they ride along.)  What is left when a
    // wavefront's walk ends is stored then.
    constexpr int kFillPerStep = 2;
    size_t zf = 0, zf_end = 0;
    if (SEG == 0 && zero_fill) {
        const size_t total = (size_t)P * (kGradStride / 4), per = (total + gridDim.x - 1) / gridDim.x;
        zf = (size_t)blockIdx.x * per + tid; zf_end = min(total, (size_t)blockIdx.x * per + per);
    }
#define GS_FILL_REST() if (SEG == 0) { for (; zf < zf_end; zf += NW * kWave) zero_fill[zf] = make_float4(0.f, 0.f, 0.f, 0.f); }
    TileCtx c;
    // chained backward walks: the ticket counters (behind the hand-over flags) start from zero with every forward
    if (SEG == 0 && cam.chain > 1 && split_state && blockIdx.x == 0 && tid < 8)
        reinterpret_cast<uint32_t*>(split_state)[(size_t)cam.gx * cam.gy * 4 * (kChainStateFloats + kChainPieces - 1) + (size_t)tid * kChainTicketStride] = 0u;
    if (!tile_ctx_nw<NW>(cam, wave, lane, c)) { GS_FILL_REST(); return; }
    // chained backward walks: the quadrant's hand-over flags start from zero with every forward (the backward's epoch is never zero)
    if (SEG == 0 && cam.chain > 1 && split_state && lane < kChainPieces - 1)
        reinterpret_cast<uint32_t*>(split_state)[(size_t)cam.gx * cam.gy * 4 * kChainStateFloats + (size_t)(c.tile * 4 + c.quad) * (kChainPieces - 1) + lane] = 0u;
    // lane -> pixel: stream sid owns block (sid & 1, sid >> 1) of the quadrant
    const int sid = lane / LS, l = lane % LS;
    const int px = (int)c.qx0 + (sid & 1) * 4 + (l & 3), py = (int)c.qy0 + (sid >> 1) * BH + (l >> 2);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    float4* s0 = s_rec[wave][0]; float4* s1 = s_rec[wave][1]; float4* s2 = s_rec[wave][2];
    write_sentinel(s0, s1, s2, lane);
    uint8_t* my_list = s_list[wave] + sid * kWave;
    uint2 range = ranges[c.tile];
    range.x = min(range.x, cap); range.y = min(range.y, cap);
    const uint32_t n_all = range.y - range.x;
    const uint32_t* list = point_list + range.x;
    // segment [first, n) of the list (SEG = 0: everything)
    uint32_t first = 0, n = n_all;
    float T = 1.0f;
    bool next_stopped = true;                   // SEG = 2: the following segment finds this pixel stopped at its entry
    float* my_seg_T = nullptr;
    uint32_t* seg_flag = nullptr;
    if (SEG != 0) {
        const uint32_t S = gridDim.y, seg = blockIdx.y;
        const uint32_t L = (((n_all + S - 1) / S + kWave - 1) / kWave) * kWave;       // chunk-aligned segment length
        first = min(n_all, seg * L);
        n = min(n_all, first + L);
        float* tile_T = seg_T + ((size_t)c.tile * S) * kBlock + c.quad * kWave + lane;   // [tile][segment][quadrant][lane]
        my_seg_T = tile_T + (size_t)seg * kBlock;
        if (SEG == 1) {
            // an EARLIER segment that takes every pixel of this quadrant below the stop threshold on its own makes this one
            // irrelevant (all its pixels are stopped at entry whatever it holds): such segments leave a bit in the quadrant's
            // flag word (behind the transmittances).  Segments are dispatched in order, so the bit is usually there in time; if
            // it is not, the segment simply does its work -- the result does not depend on the timing.
            seg_flag = reinterpret_cast<uint32_t*>(seg_T + (size_t)cam.gx * cam.gy * S * kBlock) + c.tile * 4 + c.quad;
            const uint32_t f = __hip_atomic_load(seg_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f & ((1u << seg) - 1u)) { my_seg_T[0] = 1.0f; return; }
        }
        if (SEG == 2) {
            for (uint32_t q = 0; q < seg; q++) T *= tile_T[(size_t)q * kBlock];
            next_stopped = seg + 1 == S ? true : (T * my_seg_T[0] < kTmin);
        }
    }
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Dq = 0.f;
    uint32_t last = 0;
    const bool stopped_at_entry = SEG == 2 && T < kTmin;
    bool done = !inside || stopped_at_entry;

    if (SEG == 1 || !__all(done)) {
        // An optimistic launch (rasterizer.py) may run with a tile-list capacity below the true list length: the sort then
        // leaves the tail of point_list unwritten (uninitialised memory).  Such a frame is discarded and re-rendered, but it
        // must not fault: an id that is not a Gaussian index is treated as "no record" (ids >= P; kNoId is one of them).
        uint32_t id_next = first + (uint32_t)lane < n ? list[first + lane] : kNoId;
        uint32_t id_next2 = first + (uint32_t)lane + 64u < n ? list[first + lane + 64] : kNoId;
        if (id_next >= P) id_next = kNoId;
        if (id_next2 >= P) id_next2 = kNoId;
        // (unconditional record loads: a lane without a list entry fetches record 0 -- there is one: the list is not empty -- and is masked
        // by its id in the scan; a conditional load keeps the old registers alive for the masked lanes: twelve moves and a branch per chunk)
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = make_float4(0.f, 0.f, -1.f, -1.f);
        if (first < n) {
            const float4* gr = geom + (size_t)(id_next != kNoId ? id_next : 0u) * 3;
            r0 = gr[0]; r1 = gr[1]; r2 = gr[2];
        }
        for (uint32_t base = first; base < n; base += kWave) {
            if (SEG == 0) {
#pragma unroll
                for (int f = 0; f < kFillPerStep; f++)
                    if (zf < zf_end) { zero_fill[zf] = make_float4(0.f, 0.f, 0.f, 0.f); zf += NW * kWave; }
            }
            const float4 q0 = r0, q1 = r1, q2 = r2;
            const uint32_t id_cur = id_next;
            id_next = id_next2;
            id_next2 = base + 128u + (uint32_t)lane < n ? list[base + 128u + lane] : kNoId;
            if (id_next2 >= P) id_next2 = kNoId;
            {
                const float4* gr = geom + (size_t)(id_next != kNoId ? id_next : 0u) * 3;
                r0 = gr[0]; r1 = gr[1]; r2 = gr[2];
            }

            // a stream whose pixels have ALL stopped gets an empty list: the walk's trip count is the longest list of the streams that
            // still blend (pixels of a quadrant saturate at different depths: at 2 M Gaussians the mean stop is at position ~900, the
            // last pixel of a quadrant stops around 1400).  The four block tests are wave masks in scalar registers (scan_blocks): no
            // branches, the lane's rank in a list is one mbcnt of the block's mask
            const unsigned long long going = SEG == 1 ? ~0ull : ~__ballot(done);
            unsigned long long m_any, mb[4];
            scan_blocks(q0, q2, id_cur != kNoId, c.qx0, c.qy0, (going & 0xffffull) != 0ull, (going & 0xffff0000ull) != 0ull,
                        (going & 0xffff00000000ull) != 0ull, (going >> 48) != 0ull, m_any, mb);
            if (m_any == 0ull) continue;
            stage_record(s0, s1, s2, lane, q0, q1, q2, id_cur);
            // per-stream lists: sentinel fill (one store per lane covers NS x 64 bytes), then every hit lane drops its index
            {
                uint32_t* fill = reinterpret_cast<uint32_t*>(s_list[wave]);
                constexpr int kWords = NS * kWave / 4;
                for (int w = lane; w < kWords; w += kWave) fill[w] = 0x40404040u;
            }
            __builtin_amdgcn_wave_barrier();
            int ntrips = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const unsigned long long m = mb[s];
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (__builtin_amdgcn_inverse_ballot_w64(m)) s_list[wave][s * kWave + rank] = (uint8_t)lane;
                ntrips = max(ntrips, (int)__popcll(m));
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t jj2_next = *reinterpret_cast<const uint16_t*>(my_list);                // two list entries per read
            int lj = -1;                             // staged slot of this chunk's last contributor (slots ascend along a stream's list)
            for (int t = 0; t < ntrips; t += 2) {
                const uint32_t jj2 = jj2_next;
                jj2_next = *reinterpret_cast<const uint16_t*>(my_list + t + 2);           // next pair: off the critical path
                const int jj[2] = {(int)(jj2 & 0xffu), (int)(jj2 >> 8)};
                float4 a0[2], a1[2], a2[2];
#pragma unroll
                for (int u = 0; u < 2; u++) { a0[u] = s0[jj[u]]; a1[u] = s1[jj[u]]; a2[u] = s2[jj[u]]; }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const float dx = a0[u].x - pxf, dy = a0[u].y - pyf;
                    const float p = (a0[u].z * dx + a0[u].w * dy) * dx + (a1[u].x * dy) * dy;
                    const float alpha = fminf(0.99f, a1[u].y * __builtin_amdgcn_exp2f(p));
                    if (SEG == 1) {                                                  // transmittance of the segment only
                        T = (p <= 0.0f && alpha >= kAlphaMin) ? T * (1.0f - alpha) : T;
                        continue;
                    }
                    const float test_T = T * (1.0f - alpha);
                    const bool vis = !done && p <= 0.0f && alpha >= kAlphaMin;     // sentinel: alpha = 0
                    const bool ok = vis && test_T >= kTmin;
                    done = done || (vis != ok);                                    // (ok implies vis: one mask operation, no second compare)
                    const float w = ok ? alpha * T : 0.0f;
                    C0 += a1[u].z * w; C1 += a1[u].w * w; C2 += a2[u].x * w; Dp += a2[u].y * w;
                    if (DEPTH_SQ) Dq += a2[u].y * a2[u].y * w;
                    T = ok ? test_T : T;
                    lj = ok ? jj[u] : lj;
                }
            }
            last = lj >= 0 ? base + (uint32_t)lj + 1u : last;
            __builtin_amdgcn_wave_barrier();
            if (SEG != 1 && __all(done)) break;
            if (SEG == 1 && __all(!inside || T < kTmin)) break;      // every pixel is below the stop threshold: the rest cannot matter
        }
    }
    GS_FILL_REST();
    if (SEG == 1) {
        my_seg_T[0] = T;
        if (__all(!inside || T < kTmin) && lane == 0) atomicOr(seg_flag, 1u << blockIdx.y);
        return;
    }
    if (SEG == 2) {
        if (inside && !stopped_at_entry) {
            const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
            const float tb = next_stopped ? T : 0.0f;            // exactly one segment per pixel owns final_T and the background
            atomicAdd(out_color + pix, C0 + tb * cam.bg[0]);
            atomicAdd(out_color + HW + pix, C1 + tb * cam.bg[1]);
            atomicAdd(out_color + 2 * HW + pix, C2 + tb * cam.bg[2]);
            atomicAdd(out_depth + pix, Dp);
            if (DEPTH_SQ) atomicAdd(out_depth_sq + pix, Dq);
            if (last) atomicMax(n_contrib + pix, last);
            if (next_stopped) { final_T[pix] = T; out_opacity[pix] = 1.0f - T; }
        }
        return;
    }
    if (inside) {
        const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * cam.bg[0];
        out_color[HW + pix] = C1 + T * cam.bg[1];
        out_color[2 * HW + pix] = C2 + T * cam.bg[2];
        out_depth[pix] = Dp;
        out_opacity[pix] = 1.0f - T;
        if (DEPTH_SQ) out_depth_sq[pix] = Dq;
    }
}

#undef GS_FILL_REST

// ---------------------------------------------------------------------------------------------------
// Forward for images of FEW tiles (at most 256: the reference's 256 x 256 frames, the planner's views): a three-stage pipeline per quadrant.
// 256 tiles x 4 quadrants are one walker per SIMD of this chip, and a walker's trip through a 64-record chunk is one serial chain --
// gather + stage + lists, then alpha (12 VALU + exp) and the compositing step (T, stop test, four sums: 12 more) for every list entry.
// The stages do not depend on each other in the same way: alpha needs nothing from the walk, the compositing step needs alpha, the
// lists need neither.  So a tile is ONE 12-wavefront workgroup, one per CU, with three wavefronts per quadrant:
//   LISTER   : fetches the tile's records (ids four chunks ahead, records three; the four listers share the gather: 16 records each, one
//              gather per tile instead of four), stages chunk i+2, and builds the per-stream lists of chunk i+1;
//   PRODUCER : alpha of every list entry of chunk i -> an LDS plane [trip][lane];
//   CONSUMER : composites chunk i-1 from the plane (and records the states the segmented backward resumes from).
// One workgroup barrier per chunk; alpha planes double-buffered, lists triple-buffered, staged records in four buffers (written at step
// i-2, read by the lister at i-1, the producer at i, the consumer at i+1).  Same arithmetic per entry as blend_forward_streams_kernel:
// identical images.  LDS: 4 x 2 x 16 KB planes + lists + 4 x 3 KB records = 147 KB: one workgroup per CU, which is what a 256-tile image
// offers anyway.  Measured (256 x 256, 200 k Gaussians): one walker per (half) quadrant 80 us; here 65 us (1 M Gaussians: 87 -> 73 us).  What is
// left is instruction issue: the three wavefronts of a quadrant share one SIMD (256 tiles x 4 quadrants = the chip's 1024 SIMDs), ~1500
// instructions per 64-record chunk between them -- splitting the stages hides their latencies, it does not add issue slots (a two-stage
// version, and 64-bit / 128-bit variants of the LDS traffic, measured the same 65 us: profiles/README.md).
// ---------------------------------------------------------------------------------------------------
constexpr int kPcWaves = 12;
template <bool DEPTH_SQ>
__global__ __launch_bounds__(kPcWaves * kWave) void blend_forward_pc_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ geom,
    float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_opacity, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, float* __restrict__ out_depth_sq, uint32_t cap, float* __restrict__ split_state, uint32_t P,
    float4* __restrict__ zero_fill)
{
    // staged records, split by reader: [0] = (x, y, A', B') and [1] = (C', opacity, ext_x, ext_y) for the lister and the producer,
    // [2] = (r, g, b, depth) for the consumer: one 128-bit read per entry there, 128 + 64 bits for alpha (the kernel is bound by the CU's
    // LDS pipe: one workgroup, twelve wavefronts, every list entry read by two of them)
    __shared__ float4 s_rec[4][3][kWave + 1];                                         // [buffer][part][slot]; slot 64 = sentinel
    __shared__ float4 s_alpha[4][2][(kWave / 4) * kWave];                             // [pair][buffer][(trip / 4) * 64 + lane]: four trips' alpha
    __shared__ __attribute__((aligned(16))) uint8_t s_list[4][3][4 * kWave + 16];     // [pair][buffer][stream * 64 + position]
    __shared__ int s_ntrips[4][3];
    __shared__ unsigned long long s_going[4];                                         // lanes of the quadrant that still blend
    __shared__ int s_done[2][4];                                                      // [step parity][quadrant]: every pixel has stopped
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave & 3, role = wave >> 2;                                      // role 0: lister, 1: producer, 2: consumer
    const int ntiles = cam.gx * cam.gy, per = (ntiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    // the backward's gradient records: this workgroup's slice, a few 16-byte stores per step of the loop below
    size_t zf = 0, zf_end = 0;
    if (zero_fill) {
        const size_t total = (size_t)P * (kGradStride / 4), pw = (total + gridDim.x - 1) / gridDim.x;
        zf = (size_t)blockIdx.x * pw + tid; zf_end = min(total, (size_t)blockIdx.x * pw + pw);
    }
    const size_t HWs = (size_t)cam.W * cam.H;
    // (the recorded planes and their "recorded" word exist only in the workspace of an image of at most kFewTiles tiles; gs_set_half_quadrants
    // may send a larger image here, whose workspace holds the chained backward's hand-over state at that offset instead)
    const bool few = ntiles <= kFewTiles;
    const bool record = few && cam.split != 0 && zero_fill != nullptr && split_state != nullptr;
    if (few && split_state && blockIdx.x == 0 && tid == 0) reinterpret_cast<uint32_t*>(split_state + (kCutLevels * 5 + 4) * HWs)[0] = record ? 1u : 0u;
    if ((int)(blockIdx.x >> 3) >= per || tile >= ntiles) {                             // (uniform for the workgroup)
        for (; zf < zf_end; zf += kPcWaves * kWave) zero_fill[zf] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const float qx0 = (float)(tx * kTile + (pair & 1) * kQuad), qy0 = (float)(ty * kTile + (pair >> 1) * kQuad);
    const int sid = lane >> 4, l = lane & 15;
    const int px = (int)qx0 + (sid & 1) * 4 + (l & 3), py = (int)qy0 + (sid >> 1) * 4 + (l >> 2);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    uint2 range = ranges[tile];
    range.x = min(range.x, cap); range.y = min(range.y, cap);
    const uint32_t n = range.y - range.x;
    const uint32_t* list = point_list + range.x;
    const int nchunks = (int)((n + kWave - 1) / kWave);
    if (tid < 4) {
        s_rec[tid][0][kWave] = make_float4(0.f, 0.f, 0.f, 0.f); s_rec[tid][1][kWave] = make_float4(0.f, 0.f, -1.f, -1.f); s_rec[tid][2][kWave] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_going[tid] = ~0ull; s_done[0][tid] = 0; s_done[1][tid] = 0; s_ntrips[tid][0] = 0; s_ntrips[tid][1] = 0; s_ntrips[tid][2] = 0;
    }

    // Record `lane + 16 * pair` of a chunk is fetched by lanes 0-15 of lister `pair` and staged in the blend form (conic pre-scaled for
    // 2^p: see stage_record; the last word pair holds the alpha-visible half extents here).  The fetch is a chain of two dependent memory
    // reads (id, then record), so it runs ahead of the walk.
    auto fetch_id = [&](int chunk) -> uint32_t {
        const uint32_t e = (uint32_t)chunk * kWave + (uint32_t)(pair * 16 + lane);
        uint32_t id = kNoId;
        if (lane < 16 && chunk < nchunks && e < n) id = list[e];
        return id < P ? id : kNoId;                                                   // (an id that is no Gaussian index: no record)
    };
    auto fetch_rec = [&](uint32_t id, float4 (&g)[3]) {
        g[0] = make_float4(0.f, 0.f, 0.f, 0.f); g[1] = g[0]; g[2] = make_float4(0.f, 0.f, -1.f, -1.f);
        if (id != kNoId) { g[0] = geom[(size_t)id * 3]; g[1] = geom[(size_t)id * 3 + 1]; g[2] = geom[(size_t)id * 3 + 2]; }
    };
    auto stage = [&](int chunk, const float4 (&g)[3]) {
        if (lane < 16 && chunk < nchunks) {
            const int b = chunk & 3, slot = pair * 16 + lane;
            s_rec[b][0][slot] = make_float4(g[0].x, g[0].y, -0.5f * kLog2e * g[0].z, -kLog2e * g[0].w);
            s_rec[b][1][slot] = make_float4(-0.5f * kLog2e * g[1].x, g[1].y, g[2].z, g[2].w);
            s_rec[b][2][slot] = make_float4(g[1].z, g[1].w, g[2].x, g[2].y);
        }
    };
    // the per-stream lists of one chunk (lane = staged record): hits of the alpha-visible box on the quadrant's four 4x4 blocks
    auto build_lists = [&](int chunk) {
        if (chunk >= nchunks) return;
        const int lb = chunk % 3, rb = chunk & 3;
        const float2 q0 = *reinterpret_cast<const float2*>(&s_rec[rb][0][lane]);                       // (x, y)
        const float2 q2 = *(reinterpret_cast<const float2*>(&s_rec[rb][1][lane]) + 1);                 // (ext_x, ext_y)
        int ntrips = 0;
        // (the four block tests as wave masks in scalar registers: scan_blocks; a record slot behind the list's end was staged with extent -1)
        const unsigned long long going = s_going[pair];
        unsigned long long m_any, mb[4];
        scan_blocks(make_float4(q0.x, q0.y, 0.f, 0.f), make_float4(0.f, 0.f, q2.x, q2.y), true, qx0, qy0, (going & 0xffffull) != 0ull,
                    (going & 0xffff0000ull) != 0ull, (going & 0xffff00000000ull) != 0ull, (going >> 48) != 0ull, m_any, mb);
        if (m_any != 0ull) {
            reinterpret_cast<uint32_t*>(s_list[pair][lb])[lane] = 0x40404040u;        // sentinel fill: 4 x 64 bytes
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const unsigned long long m = mb[st];
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (__builtin_amdgcn_inverse_ballot_w64(m)) s_list[pair][lb][st * kWave + rank] = (uint8_t)lane;
                ntrips = max(ntrips, (int)__popcll(m));
            }
        }
        if (lane == 0) s_ntrips[pair][lb] = ntrips;
    };
    float4 held[3], flying[3];                       // lister: records of chunk it + 2 (arrived, staged during step it) / of chunk it + 3 (in flight)
    uint32_t id_ahead = kNoId;                       // lister: this lane's id of chunk it + 3 at the top of step it
    __syncthreads();                                 // (sentinels and flags)
    if (role == 0) {
        fetch_rec(fetch_id(0), held); stage(0, held);
        fetch_rec(fetch_id(1), held); stage(1, held);
        fetch_rec(fetch_id(2), held);
        id_ahead = fetch_id(3);
    }
    __syncthreads();
    if (role == 0) build_lists(0);
    __syncthreads();

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, Dq = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    for (int it = 0; it <= nchunks; it++) {
        for (int f = 0; f < 2; f++)
            if (zf < zf_end) { zero_fill[zf] = make_float4(0.f, 0.f, 0.f, 0.f); zf += kPcWaves * kWave; }
        const bool quad_done = s_done[(it + 1) & 1][pair] != 0;                         // as of the end of step it - 1
        if (role == 0) {
            // ---- lister: records of chunk it + 3 requested, lists of chunk it + 1, chunk it + 2 staged ----
            const uint32_t id_next = fetch_id(it + 4);
            fetch_rec(id_ahead, flying);
            if (!quad_done) build_lists(it + 1);
            stage(it + 2, held);
#pragma unroll
            for (int q = 0; q < 3; q++) held[q] = flying[q];
            id_ahead = id_next;
        } else if (role == 1) {
            // ---- producer: alpha of chunk it ----
            if (it < nchunks && !quad_done) {
                const int ab = it & 1, lb = it % 3, rb = it & 3;
                const int ntrips = s_ntrips[pair][lb];
                const uint8_t* my_list = s_list[pair][lb] + sid * kWave;
                float4* plane = s_alpha[pair][ab] + lane;
                const float4* s0 = s_rec[rb][0]; const float4* s1 = s_rec[rb][1];
                // four list entries per trip (one 32-bit list read, fetched a trip ahead): the entries are independent here, so the
                // eight record reads of a trip are in flight together (entries behind the list's end are the sentinel: alpha 0)
                uint32_t jj4_next = ntrips > 0 ? *reinterpret_cast<const uint32_t*>(my_list) : 0x40404040u;
                for (int t = 0; t < ntrips; t += 4) {
                    const uint32_t jj4 = jj4_next;
                    jj4_next = *reinterpret_cast<const uint32_t*>(my_list + ((t + 4) & 63));
                    float4 a0[4];
                    float2 a1[4];
                    float al[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int j = (int)((jj4 >> (8 * u)) & 0xffu); a0[u] = s0[j]; a1[u] = *reinterpret_cast<const float2*>(&s1[j]); }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float dx = a0[u].x - pxf, dy = a0[u].y - pyf;
                        const float p = (a0[u].z * dx + a0[u].w * dy) * dx + (a1[u].x * dy) * dy;
                        const float alpha = fminf(0.99f, a1[u].y * __builtin_amdgcn_exp2f(p));
                        al[u] = (p <= 0.0f && alpha >= kAlphaMin) ? alpha : 0.0f;                        // 0 = this pixel does not see the record
                    }
                    plane[(t >> 2) * kWave] = make_float4(al[0], al[1], al[2], al[3]);
                }
            }
        } else {
            // ---- consumer: composite chunk it - 1 ----
            if (it >= 1 && !__all(done)) {
                const int ab = (it - 1) & 1, lb = (it - 1) % 3, rb = (it - 1) & 3;
                const uint32_t base = (uint32_t)(it - 1) * kWave;
                const int k = record ? cut_level(base) : -1;                                   // (base is wave-uniform)
                if (k >= 0 && inside) {
                    float* stt = split_state + (size_t)k * 5 * HWs + (size_t)py * cam.W + px;
                    stt[0] = T; stt[HWs] = C0; stt[2 * HWs] = C1; stt[3 * HWs] = C2; stt[4 * HWs] = Dp;
                }
                const int ntrips = s_ntrips[pair][lb];
                const uint8_t* my_list = s_list[pair][lb] + sid * kWave;
                const float4* plane = s_alpha[pair][ab] + lane;
                const float4* s2 = s_rec[rb][2];
                uint32_t jj4_next = ntrips > 0 ? *reinterpret_cast<const uint32_t*>(my_list) : 0x40404040u;
                int lj = -1;                         // staged slot of this chunk's last contributor (slots ascend along a stream's list)
                for (int t = 0; t < ntrips; t += 4) {
                    const uint32_t jj4 = jj4_next;
                    jj4_next = *reinterpret_cast<const uint32_t*>(my_list + ((t + 4) & 63));
                    const float4 al4 = plane[(t >> 2) * kWave];
                    const float al[4] = {al4.x, al4.y, al4.z, al4.w};
                    float4 b2[4];
                    int jj[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { jj[u] = (int)((jj4 >> (8 * u)) & 0xffu); b2[u] = s2[jj[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float alpha = al[u];
                        const float4 cc = b2[u];                                 // (r, g, b, depth)
                        const float test_T = T * (1.0f - alpha);
                        const bool vis = !done && alpha > 0.0f;
                        const bool ok = vis && test_T >= kTmin;
                        done = done || (vis != ok);                              // (ok implies vis)
                        const float w = ok ? alpha * T : 0.0f;
                        C0 += cc.x * w; C1 += cc.y * w; C2 += cc.z * w; Dp += cc.w * w;
                        if (DEPTH_SQ) Dq += cc.w * cc.w * w;
                        T = ok ? test_T : T;
                        lj = ok ? jj[u] : lj;
                    }
                }
                last = lj >= 0 ? base + (uint32_t)lj + 1u : last;
                const unsigned long long going = ~__ballot(done);
                if (lane == 0) s_going[pair] = going;
            }
            const bool all_done = __all(done);                     // (a wave vote: outside of the one-lane store below)
            if (lane == 0) s_done[it & 1][pair] = all_done ? 1 : 0;
        }
        __syncthreads();
        if (s_done[it & 1][0] && s_done[it & 1][1] && s_done[it & 1][2] && s_done[it & 1][3]) break;      // (uniform: written before the barrier)
    }
    for (; zf < zf_end; zf += kPcWaves * kWave) zero_fill[zf] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (role == 2 && inside) {
        const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * cam.bg[0];
        out_color[HW + pix] = C1 + T * cam.bg[1];
        out_color[2 * HW + pix] = C2 + T * cam.bg[2];
        out_depth[pix] = Dp;
        out_opacity[pix] = 1.0f - T;
        if (DEPTH_SQ) out_depth_sq[pix] = Dq;
        if (record) {
            float* tot = split_state + (size_t)kCutLevels * 5 * HW + pix;
            tot[0] = C0; tot[HW] = C1; tot[2 * HW] = C2; tot[3 * HW] = Dp;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward: back-to-front replay in two phases per batch of list positions.  Same independent-quadrant walk and the same four
// record streams as the forward (one per 16-lane row = 4x4 pixel block of the quadrant), lists built deepest-first.
//
// Per-Gaussian 2-D gradient record accumulated here (raw moments; the conic algebra is finished per Gaussian in
// preprocess_bwd.hip):  with Z = G dL/dG, d = mean - pixel
//   0: sum Z dx   1: sum Z dy   2: sum Z dx dx   3: sum Z dx dy   4: sum Z dy dy   5: sum G dL/dalpha   6..8: sum w dL/dC
//   9: sum w dL/d(depth)   (DEPTH_GRAD: the depth output is one more blended channel whose per-Gaussian value is the
//      view-space z; this is the fused replacement of the reference's second, [z,1,z^2] raster pass)
//
// The expensive part of a one-phase walk (round 1: 213 us on BASELINE configs[1]) was summing those ten moments over the 16 pixels
// of a row for every (row, record) pair: a transposed DPP butterfly (35 DPP adds + selects per two-record iteration, each a 4.2-cycle
// issue slot against 2.7 for a plain add -- scripts/exp/valu_issue.hip) plus LDS float atomics (3 clk per lane) wherever two rows
// held the same record.  Here the replay (lane = pixel) only produces the two per-pixel SCALARS every moment is built from --
// GdA = G dL/dalpha (Z = opacity x GdA) and w = alpha T -- and hands them through a per-wave LDS exchange to a second phase in
// which a lane is a (row, list position) PAIR that owns all 16 pixels of its block:
//   phase A (up to 16 list positions per batch, two per iteration): alpha, branch-free replay step (T divided back, one
//           behind-colour accumulator, dL/dalpha), 2 x ds_write_b32 per record, lane-contiguous;
//   phase B (once per batch): lane (position t, row r) reads its block's 2 x 16 scalars with eight conflict-free ds_read_b128
//           (plane rows are 68 floats apart), forms the ten moments with plain FMAs against ITS pixels' coordinates and dL/dcolour
//           (48 registers, loaded once per kernel), factored into column / row sums; writes them as the pair's 12-float slot;
//   gather : lane = staged record j knows its position in each row's list (its own mbcnt rank from the list build), so it reads the
//           slots of its <= 4 pairs of this batch and adds them to ten REGISTER accumulators -- no LDS read-modify-write, no atomics;
//   flush  : once per ROUND the staged records put their sums into the idle exchange planes and leave with one global fp32 atomic
//           request per (quadrant, record), 6 records x 10 components per instruction, into 64-byte-aligned gradient records (a
//           device-scope atomic is a read-modify-write of a whole line at the memory side on this multi-XCD part).
// A ROUND is not a list chunk (round 4): a 64-record chunk is only SCANNED (scan_blocks: four block tests as wave masks, one ballot) and the
// ~21 records that meet the quadrant are staged -- with list position and block flags -- behind those of the chunks before; lists, phases,
// gather and flush run when the 64 staging slots are (nearly) full.  Chunk by chunk, three of ten phase-B batches ran nearly empty and the
// per-chunk set-up was paid per ~14 list positions; rounds hold ~45 records / ~30 positions (scripts/exp/bwd_work.py replays the control
// flow from a frame's integer artefacts and prices it with the ISA's instruction counts: 91 M -> 78 M vector instructions at 2 M Gaussians).
// Measured (profiles/README.md): round 2: 184 us on configs[1], 262 us at 2 M Gaussians; round 3 (chained pieces): 157 / 210; round 4
// (rounds on compacted staging): 141 / 188.
// LDS 12.1 KB per wave; the wavefronts share nothing, so a workgroup is ONE wavefront (NW = 1): LDS and CU slots are handed out at
// that granularity and 4800 small workgroups drain more evenly than 1200 whole-tile ones.
// ---------------------------------------------------------------------------------------------------
constexpr int kBT = 16;                // list positions per batch: 16 x 4 rows = one (row, position) pair per lane in phase B
constexpr int kMT = kWave + 4;         // floats per position in an exchange plane: +4 makes phase B's b128 reads conflict-free
constexpr int kPairStride = 12;        // floats per pair / record slot in the sum exchanges: components 0-4 at [0,5), 5-9 at [6,11)
constexpr int kMPlane = (kBT - 1) * kMT + kWave;   // floats of one exchange plane

template <bool DEPTH_GRAD, int NW, bool FEW = false>        // FEW: two or three list segments per quadrant (images of few tiles)
__global__ __launch_bounds__(NW * kWave) __attribute__((amdgpu_waves_per_eu(3, 3))) void blend_backward_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ geom, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, float* __restrict__ grad2d,
    const float* __restrict__ split_state)
{
    __shared__ float4 s_rec[NW][3][kWave + 1];           // + the sentinel slot
    __shared__ __attribute__((aligned(16))) uint8_t s_list[NW][4 * kWave + 16];
    __shared__ __attribute__((aligned(16))) float s_m[NW][2][kMPlane];
    __shared__ uint8_t s_flag[NW][kWave];                // per staging slot: which of the four blocks the record's box meets
    __shared__ __attribute__((aligned(16))) float s_ez[DEPTH_GRAD ? NW : 1][DEPTH_GRAD ? kWave : 4];      // dL/ddepth of the quadrant, (block, pixel) order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx c;
    const int nseg = FEW ? cam.split : 0;                    // 0: one walker per quadrant; 2 / 3: list segments (walkers) per quadrant
    // Chained walks (images of more quadrants than resident walkers; never together with list segments): the grid is cam.chain groups of
    // workgroups, group p walks piece p of every quadrant (piece 0 = the deepest third of the chunks).  A quadrant's pieces form a serial
    // chain -- piece p starts from the (T, S) piece p - 1 ends with -- but 3 x 4800 short items pack the chip's 3072 walker slots far
    // better than 4800 long ones: a wavefront's pace does not depend on how many share its SIMD (the walk is a latency chain), so with one
    // walker per quadrant the kernel lasts two full walks (3072 + 1728 walkers) for 1.56 walks' worth of work.  Workgroups are dispatched in
    // index order and a piece only ever waits for a LOWER index (same XCD: the group size is a multiple of 8), so the wait cannot deadlock.
    const int pieces = (!FEW && cam.chain > 1) ? cam.chain : 1;
    const unsigned group = gridDim.x / (unsigned)pieces;
    // ORDERED TICKETS (cam.chain_ticket): the workgroup's place in the chain order is the ticket it draws NOW, from the counter of its index class
    // (blockIdx & 7 -- on this part the XCD it runs on), so "the piece in front" is by construction a workgroup that drew earlier: running or
    // done.  The index order of the dispatcher is then an optimisation (the chains start in list order), no longer a correctness assumption.
    // The last drawer of a class puts its counter back to zero for the next launch on this workspace.
    unsigned vb = blockIdx.x;
    if (pieces > 1 && cam.chain_ticket) {
        uint32_t* ctr = reinterpret_cast<uint32_t*>(const_cast<float*>(split_state)) + (size_t)cam.gx * cam.gy * 4 * (kChainStateFloats + kChainPieces - 1) +
                        (size_t)(blockIdx.x & 7u) * kChainTicketStride;
        unsigned t = 0u;
        if (lane == 0) t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t == (gridDim.x >> 3) - 1u && lane == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vb = (t << 3) | (blockIdx.x & 7u);
    }
    const int piece = pieces > 1 ? (int)(vb / group) : 0;
    if (!tile_ctx_at<NW>(cam, pieces > 1 ? vb - (unsigned)piece * group : vb, wave, lane, c, nseg)) return;
    // phase A role: row (lane>>4) = 4x4 sub-block, (lane&15) = pixel inside it
    const int row = lane >> 4, l16 = lane & 15;
    const int px = (int)c.qx0 + (row & 1) * 4 + (l16 & 3), py = (int)c.qy0 + (row >> 1) * 4 + (l16 >> 2);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    float4* s0 = s_rec[wave][0]; float4* s1 = s_rec[wave][1]; float4* s2 = s_rec[wave][2];
    write_sentinel(s0, s1, s2, lane);
    const uint8_t* my_list = s_list[wave] + row * kWave;
    float* pair_lds = s_m[wave][0];                          // the pair sums overwrite the scalars phase B has consumed
    float* m1p = s_m[wave][0]; float* m2p = s_m[wave][1];
    const int fl_rec = lane / 10, fl_comp = lane - fl_rec * 10;                  // flush mapping: 6 records x 10 components per instruction
    const int fl_off = fl_comp < 5 ? fl_comp : fl_comp + 1;
    const uint2 range = ranges[c.tile];
    const uint32_t* list = point_list + range.x;
    const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;

    const float Tf = inside ? final_T[pix] : 0.f;
    uint32_t last = inside ? n_contrib[pix] : 0u;
    // List segments (cam.split = 2 or 3, and the forward recorded): the quadrant's walk, wmax positions deep, is cut at the recorded
    // positions nearest to wmax / nseg, 2 wmax / nseg.  Walker s takes the positions [cut s, cut s+1): the LAST one replays from the
    // final state exactly as the one-walker kernel would; the others take the pixels that contributed past their upper cut from the state
    // recorded there -- T in front of that record, behind-colour = (totals - sums up to the cut) / T -- and everything else as usual.
    uint32_t wmax = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor(wmax, m));
    if (wmax == 0) return;
    uint32_t seg_lo = 0u, seg_hi = wmax;
    int k_hi = -1;                                           // level of the recorded state this walker resumes from (-1: the final state)
    if (nseg > 1) {
        const bool recorded = split_state && reinterpret_cast<const uint32_t*>(split_state + (kCutLevels * 5 + 4) * HW)[0] == 1u;
        // (named scalars, no indexed local: kFewSegmentsMax = 3)
        uint32_t cut1 = recorded ? cut_nearest((uint32_t)((unsigned long long)wmax / (unsigned)nseg)) : 0u;
        if (cut1 >= wmax) cut1 = 0u;
        uint32_t cut2 = (recorded && nseg > 2) ? cut_nearest((uint32_t)((unsigned long long)wmax * 2u / (unsigned)nseg)) : cut1;
        if (cut2 >= wmax || cut2 <= cut1) cut2 = cut1;       // (no usable position there: an empty segment)
        // walker 0: [0, cut1), walker 1: [cut1, cut2) (two segments: [cut1, wmax)), walker 2: [cut2, wmax)
        const int s_ = c.seg;
        seg_lo = s_ == 0 ? 0u : (s_ == 1 ? cut1 : cut2);
        seg_hi = s_ >= nseg - 1 ? wmax : (s_ == 0 ? cut1 : cut2);
        if (s_ >= nseg || seg_hi <= seg_lo) return;          // nothing in this segment: a neighbour covers it
        if (seg_hi < wmax) k_hi = cut_level(seg_hi);
    }
    const bool resumed = k_hi >= 0 && last > seg_hi;
    if (k_hi >= 0) { last = min(last, seg_hi); wmax = min(wmax, seg_hi); }
    const float d0 = inside ? dL_dcolor[pix] : 0.f, d1 = inside ? dL_dcolor[HW + pix] : 0.f,
                d2 = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float dz_ = (DEPTH_GRAD && inside) ? dL_ddepth[pix] : 0.f;
    const float tfbg = Tf * (cam.bg[0] * d0 + cam.bg[1] * d1 + cam.bg[2] * d2);    // background term of dL/dalpha
    // The colour blended BEHIND the current record enters the replay only through its product with this pixel's dL/dcolour (and
    // dL/ddepth): one scalar S = behind . dL instead of a three- (four-) component accumulator -- dot = c . dL - S, S += alpha dot
    // (four vector operations per record less than the component-wise form)
    float T = Tf, S = 0.f;
    if (resumed) {
        const float* st = split_state + (size_t)k_hi * 5 * HW + pix;
        const float* tot = split_state + (size_t)kCutLevels * 5 * HW + pix;
        T = st[0];
        const float it = 1.0f / T;                               // (T in front of a record that contributed later is >= 1e-4)
        S = ((tot[0] - st[HW]) * it) * d0 + ((tot[HW] - st[2 * HW]) * it) * d1 + ((tot[2 * HW] - st[3 * HW]) * it) * d2;
        if (DEPTH_GRAD) S += ((tot[3 * HW] - st[4 * HW]) * it) * dz_;
    }
    int cmin = (int)(seg_lo / kWave);                            // a walker stops at its lower cut
    int ctop = (int)((wmax - 1) / kWave);                        // first (deepest) chunk of this walker
    // (wave-uniform values, kept in scalar registers: the kernel sits at its register budget)
    const int chain_q = __builtin_amdgcn_readfirstlane(c.tile * 4 + c.quad);
    float* const chain_st = const_cast<float*>(split_state) + (size_t)chain_q * kChainStateFloats;
    uint32_t* const chain_fl = reinterpret_cast<uint32_t*>(const_cast<float*>(split_state) + (size_t)cam.gx * cam.gy * 4 * kChainStateFloats) +
                               (size_t)chain_q * (kChainPieces - 1);
    if (pieces > 1) {
        // piece p of n chunks: chunks [n - b(p+1), n - b(p)), b(p) = n p / pieces (the quadrant's pieces all derive this from the same n_contrib)
        const int n = __builtin_amdgcn_readfirstlane(ctop + 1);
        ctop = n - (n * piece) / pieces - 1; cmin = n - (n * (piece + 1)) / pieces;
    }
    // the deepest contributing position of every 4x4 block (row of 16 lanes): the walk starts at the quadrant's deepest contributor, but a
    // block takes part only from its own -- above that its list stays empty and the trip count is set by the blocks that do contribute
    uint32_t rmax = last;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) rmax = max(rmax, (uint32_t)__shfl_xor(rmax, m));
    const uint32_t rm0 = (uint32_t)__builtin_amdgcn_readlane((int)rmax, 0), rm1 = (uint32_t)__builtin_amdgcn_readlane((int)rmax, 16);
    const uint32_t rm2 = (uint32_t)__builtin_amdgcn_readlane((int)rmax, 32), rm3 = (uint32_t)__builtin_amdgcn_readlane((int)rmax, 48);

    // phase B role: the block of row rb at list position tb of the batch; its 16 pixels' dL/dcolour stay in registers
    const int tb = lane >> 2, rb = lane & 3;
    const int bxb = (int)c.qx0 + (rb & 1) * 4, byb = (int)c.qy0 + (rb >> 1) * 4;
    float e0[16], e1[16], e2[16];
    {
        // the quadrant's dL/dcolour is already in the wavefront, one pixel per lane in (block, pixel) order: it goes through the still idle
        // exchange planes instead of 48 more global loads with their address arithmetic per lane (the prologue is paid per chained piece).
        // The fused RGB-D backward's fourth channel, dL/ddepth, stays in LDS (256 bytes per wavefront) and is read per batch in phase B:
        // sixteen more registers would cost the kernel its third wavefront per SIMD.
        float* xs = s_m[wave][0];
        xs[lane] = d0; xs[kWave + lane] = d1; xs[2 * kWave + lane] = d2;
        if (DEPTH_GRAD) s_ez[wave][lane] = dz_;
        __builtin_amdgcn_wave_barrier();
        const float4* x4 = reinterpret_cast<const float4*>(xs + rb * 16);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float4 a = x4[v], b = x4[kWave / 4 + v], cc = x4[2 * (kWave / 4) + v];
            e0[4 * v] = a.x; e0[4 * v + 1] = a.y; e0[4 * v + 2] = a.z; e0[4 * v + 3] = a.w;
            e1[4 * v] = b.x; e1[4 * v + 1] = b.y; e1[4 * v + 2] = b.z; e1[4 * v + 3] = b.w;
            e2[4 * v] = cc.x; e2[4 * v + 1] = cc.y; e2[4 * v + 2] = cc.z; e2[4 * v + 3] = cc.w;
        }
        __builtin_amdgcn_wave_barrier();           // (the planes are written again in the first batch)
    }
    const float bxf = (float)bxb, byf = (float)byb;
    const int m_rd = tb * kMT + rb * 16;                         // phase B read offset inside a plane
    const uint8_t* list_b = s_list[wave] + rb * kWave + tb;

    const int cmax = ctop;
    uint32_t id_next = (cmax >= cmin && (uint32_t)cmax * kWave + lane < wmax) ? list[cmax * kWave + lane] : kNoId;
    uint32_t id_next2 = cmax >= cmin + 1 ? list[(cmax - 1) * kWave + lane] : kNoId;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = make_float4(0.f, 0.f, -1.f, -1.f);
    if (id_next != kNoId) { r0 = geom[(size_t)id_next * 3]; r1 = geom[(size_t)id_next * 3 + 1]; r2 = geom[(size_t)id_next * 3 + 2]; }
    if (pieces > 1 && piece > 0) {
        // (as late as possible: this walker's own loads are in flight while the piece in front of it finishes)
        // (state and flag travel as device-scope atomics, which bypass the non-coherent cache levels: no acquire on the poll -- a cache
        // invalidate per poll of thousands of waiting wavefronts costs everybody's record gathers their hits)
        // With ordered tickets (gs_set_backward_chain_tickets(1)) the piece in front drew its ticket before this one: it is resident or done.
        // Without them (the default: the ticket costs 2.5-5 % of this kernel) that rests on workgroups starting in index order.  The poll is BOUNDED either way (~0.3 s): a wait
        // that runs out raises the host-visible status bit and the walk goes on with NaN state -- NaN gradients for this quadrant's
        // Gaussians, loud in every consumer -- instead of hanging the device.
        int polls = 0;
        bool handed = cam.chain_polls >= 0;                 // (negative: give up without looking -- the tests' way to take the timeout path)
        while (handed && __hip_atomic_load(chain_fl + piece - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != cam.chain_epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (++polls > cam.chain_polls) handed = false;
        }
        if (!handed && lane == 0) {
            // host-mapped word: rasterizer.py reads it before the next launch and stops chaining.  A plain system-scope STORE: every writer writes
            // the same 1, and a read-modify-write on host memory would need PCIe atomics, which not every platform routes
            if (cam.async_status) __hip_atomic_store(cam.async_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // device word: the optimiser kernels behind this backward on the stream read it and skip their step (Cam::chain_fail)
            if (cam.chain_fail) __hip_atomic_store(cam.chain_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        GS_WAIT_VMEM();
        const float* in = chain_st + (piece - 1) * 2 * kWave;
        T = __hip_atomic_load(in + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S = __hip_atomic_load(in + kWave + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!handed) T = __builtin_nanf("");
    }
    // ---- the walk: SCAN list chunks, STAGE only the records that hit the quadrant, PROCESS a round when the staging is (nearly) full ----
    // A 64-record chunk of the tile's list holds ~21 records whose alpha-visible box meets this quadrant and its longest block list is ~14
    // positions: processed chunk by chunk (rounds 1-4), the per-chunk work (lists, ranks, flush set-up) was spread over a third of the
    // lanes and three of ten batches of phase B ran nearly empty (scripts/exp/bwd_work.py: 1.25 batches per chunk, 29 % of them full).
    // Now a chunk is only scanned -- four block tests, one ballot -- and its hit records go to the next free staging slots (deepest first,
    // with their list position and block flags); a ROUND (lists, phases A / B, gather, flush -- lane = staging slot) runs on ~45 records:
    // two to three batches, nearly full, one flush of dense slots.  A round is closed after a chunk if another chunk like it would not fit;
    // a chunk that does not fit all the same is SPLIT: its deepest hits complete the round and the chunk is fetched again for the rest.
    int cnt = 0;                                                 // records staged (wave-uniform)
    int upto = kWave;                                            // split chunk: its lanes from `upto` on (the deeper ones) are done
    const int cmin_u = __builtin_amdgcn_readfirstlane(cmin);     // (wave-uniform values the compiler cannot see as such: scalar registers)
    for (int ch = __builtin_amdgcn_readfirstlane(cmax); ch >= cmin_u;) {
        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint32_t id_cur = id_next;
        id_next = id_next2;
        id_next2 = ch >= cmin_u + 2 ? list[(ch - 2) * kWave + lane] : kNoId;
        {   // (unconditional loads: a lane beyond the walker's range fetches record 0 and is masked by its id in the scan)
            const float4* gr = geom + (size_t)(id_next != kNoId ? id_next : 0u) * 3;
            r0 = gr[0]; r1 = gr[1]; r2 = gr[2];
        }

        // scan: which of the four blocks does the record's alpha-visible box meet (and has the block contributors this deep)?  Wave masks
        // in scalar registers; the block tests share their column / row halves
        const uint32_t cpos = (uint32_t)ch * kWave;                  // first list position of this chunk
        unsigned long long m_any;
        const unsigned hf = scan_blocks(q0, q2, id_cur != kNoId && lane < upto, c.qx0, c.qy0, cpos < rm0, cpos < rm1, cpos < rm2, cpos < rm3, m_any);
        const int n_hit = (int)__popcll(m_any);
        bool split = false;
        if (m_any != 0ull) {
            const int room = kWave - cnt;
            split = n_hit > room;
            const int n_stage = split ? room : n_hit;
            // deepest first (the higher lane is the deeper list position)
            const int rel = n_hit - 1 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m_any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_any, 0u));
            const bool stage = hf != 0u && rel < n_stage;
            if (stage) {
                const int slot = cnt + rel;
                s0[slot] = make_float4(q0.x, q0.y, -0.5f * kLog2e * q0.z, -kLog2e * q0.w);
                s1[slot] = make_float4(-0.5f * kLog2e * q1.x, q1.y, q1.z, q1.w);
                s2[slot] = make_float4(q2.x, __uint_as_float(cpos + (uint32_t)lane), q2.y, __uint_as_float(id_cur));
                s_flag[wave][slot] = (uint8_t)hf;
            }
            cnt = __builtin_amdgcn_readfirstlane(cnt + n_stage);
            if (split) upto = (int)__ffsll((unsigned long long)__ballot(stage)) - 1;           // (a split chunk is taken up again below its last staged lane)
        }
        if (!split) upto = kWave;
        if (split) {
            // (rare) the chunk goes back into the prefetch registers and is scanned again after the round for the rest of its hits; the
            // records prefetched for the next chunk are dropped and fetched again
            r0 = q0; r1 = q1; r2 = q2; id_next2 = id_next; id_next = id_cur;
        } else {
            ch--;
        }
        if (cnt == 0 || !(split || ch < cmin_u || (m_any != 0ull && cnt + n_hit + kRoundSlack > kWave))) continue;

        // ---- a round: lane = staging slot ----
        __builtin_amdgcn_wave_barrier();
        const unsigned fl = lane < cnt ? (unsigned)s_flag[wave][lane] : 0u;
        const bool h0 = (fl & 1u) != 0u, h1 = (fl & 2u) != 0u, h2 = (fl & 4u) != 0u, h3 = (fl & 8u) != 0u;
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
        reinterpret_cast<uint32_t*>(s_list[wave])[lane] = 0x40404040u;          // sentinel fill: one store per lane = 4 x 64 bytes
        __builtin_amdgcn_wave_barrier();
        const int n0 = (int)__popcll(m0), n1 = (int)__popcll(m1), n2 = (int)__popcll(m2), n3 = (int)__popcll(m3);
#define GS_RANK(m) ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u)))
        // slots are in walk order (deepest first); the lane (= record) remembers where it sits in every row's list (0xff: not in that row's list)
        const int p0 = h0 ? GS_RANK(m0) : 0xff, p1 = h1 ? GS_RANK(m1) : 0xff;
        const int p2 = h2 ? GS_RANK(m2) : 0xff, p3 = h3 ? GS_RANK(m3) : 0xff;
        if (h0) s_list[wave][0 * kWave + p0] = (uint8_t)lane;
        if (h1) s_list[wave][1 * kWave + p1] = (uint8_t)lane;
        if (h2) s_list[wave][2 * kWave + p2] = (uint8_t)lane;
        if (h3) s_list[wave][3 * kWave + p3] = (uint8_t)lane;
#undef GS_RANK
        float racc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // this record's moments, summed over rows and batches
        const int ntrips = max(max(n0, n1), max(n2, n3));
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < ntrips; t0 += kBT) {
            // ---- phase A: up to kBT list positions, two per iteration ----
            const int tend = min(kBT, ntrips - t0);
            uint32_t jj2_next = *reinterpret_cast<const uint16_t*>(my_list + t0);
            for (int t = 0; t < tend; t += 2) {
                const uint32_t jj2 = jj2_next;
                jj2_next = *reinterpret_cast<const uint16_t*>(my_list + t0 + t + 2);
                const int jj[2] = {(int)(jj2 & 0xffu), (int)(jj2 >> 8)};
                float4 a0[2], a1[2], a2[2];
                float dx[2], dy[2], G[2], alpha[2];
                bool ok[2];
#pragma unroll
                for (int u = 0; u < 2; u++) { a0[u] = s0[jj[u]]; a1[u] = s1[jj[u]]; a2[u] = s2[jj[u]]; }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    dx[u] = a0[u].x - pxf; dy[u] = a0[u].y - pyf;
                    const float p = (a0[u].z * dx[u] + a0[u].w * dy[u]) * dx[u] + (a1[u].x * dy[u]) * dy[u];
                    G[u] = __builtin_amdgcn_exp2f(p);
                    alpha[u] = fminf(0.99f, a1[u].y * G[u]);
                    ok[u] = __float_as_uint(a2[u].y) < last && p <= 0.0f && alpha[u] >= kAlphaMin;     // (list position below this pixel's last contributor)
                }
                float* w1 = m1p + (t * kMT + lane); float* w2 = m2p + (t * kMT + lane);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    // branch-free replay step (deeper record first); lanes that do not contribute run with alpha = 0, G = 0
                    const float a_eff = ok[u] ? alpha[u] : 0.0f;
                    const float G_eff = ok[u] ? G[u] : 0.0f;
                    const float rcp = __builtin_amdgcn_rcpf(1.0f - a_eff);
                    T = T * rcp;                                      // transmittance in front of this record
                    float cd = a1[u].z * d0 + a1[u].w * d1 + a2[u].x * d2;
                    if (DEPTH_GRAD) cd += a2[u].z * dz_;
                    const float dot = cd - S;                         // (colour of this record - colour behind it) . dL
                    const float dL_dalpha = dot * T - tfbg * rcp;
                    S += a_eff * dot;
                    const float GdA = G_eff * dL_dalpha;
                    w1[u * kMT] = GdA;                                // (Z = G dL/dG = opacity x GdA is formed in phase B)
                    w2[u * kMT] = a_eff * T;                          // blend weight
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- phase B: lane = (half hb, position tb, row rb) ----
            {
                const int jjb = (int)list_b[t0];
                const float4 rc = s0[jjb];
                const float opb = s1[jjb].y;
                const float4* gp4 = reinterpret_cast<const float4*>(m1p + m_rd);
                const float4* wp4 = reinterpret_cast<const float4*>(m2p + m_rd);
                const float4 g0 = gp4[0], g1 = gp4[1], g2 = gp4[2], g3 = gp4[3];
                const float4 v0 = wp4[0], v1 = wp4[1], v2 = wp4[2], v3 = wp4[3];
                const float gg[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
                const float ww[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
                const float ddx[4] = {rc.x - bxf, rc.x - (bxf + 1.0f), rc.x - (bxf + 2.0f), rc.x - (bxf + 3.0f)};
                const float ddy[4] = {rc.y - byf, rc.y - (byf + 1.0f), rc.y - (byf + 2.0f), rc.y - (byf + 3.0f)};
                // moments of the 4x4 block, factored: column sums for the x moments, row sums for the y moments, row-wise x-weighted
                // sums for the cross moment; Z = G dL/dG = opacity x GdA is applied once per sum
                float gc[4], gr[4], rx[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    gc[q] = (gg[q] + gg[4 + q]) + (gg[8 + q] + gg[12 + q]);
                    gr[q] = (gg[4 * q] + gg[4 * q + 1]) + (gg[4 * q + 2] + gg[4 * q + 3]);
                    rx[q] = fmaf(gg[4 * q + 3], ddx[3], fmaf(gg[4 * q + 2], ddx[2], fmaf(gg[4 * q + 1], ddx[1], gg[4 * q] * ddx[0])));
                }
                const float t0x = gc[0] * ddx[0], t1x = gc[1] * ddx[1], t2x = gc[2] * ddx[2], t3x = gc[3] * ddx[3];
                const float u0 = gr[0] * ddy[0], u1 = gr[1] * ddy[1], u2 = gr[2] * ddy[2], u3 = gr[3] * ddy[3];
                float sm[10];
                sm[0] = opb * ((rx[0] + rx[1]) + (rx[2] + rx[3]));
                sm[1] = opb * ((u0 + u1) + (u2 + u3));
                sm[2] = opb * fmaf(t3x, ddx[3], fmaf(t2x, ddx[2], fmaf(t1x, ddx[1], t0x * ddx[0])));
                sm[3] = opb * fmaf(rx[3], ddy[3], fmaf(rx[2], ddy[2], fmaf(rx[1], ddy[1], rx[0] * ddy[0])));
                sm[4] = opb * fmaf(u3, ddy[3], fmaf(u2, ddy[2], fmaf(u1, ddy[1], u0 * ddy[0])));
                sm[5] = (gr[0] + gr[1]) + (gr[2] + gr[3]);
                sm[6] = ww[0] * e0[0]; sm[7] = ww[0] * e1[0]; sm[8] = ww[0] * e2[0]; sm[9] = 0.0f;
#pragma unroll
                for (int i = 1; i < 16; i++) {
                    sm[6] = fmaf(ww[i], e0[i], sm[6]); sm[7] = fmaf(ww[i], e1[i], sm[7]); sm[8] = fmaf(ww[i], e2[i], sm[8]);
                }
                if (DEPTH_GRAD) {
                    const float4* z4 = reinterpret_cast<const float4*>(s_ez[wave] + rb * 16);
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const float4 z = z4[v];
                        sm[9] = fmaf(ww[4 * v + 3], z.w, fmaf(ww[4 * v + 2], z.z, fmaf(ww[4 * v + 1], z.y, fmaf(ww[4 * v], z.x, sm[9]))));
                    }
                }
                // pair sums -> LDS (over the scalars just consumed: every lane has issued its reads; LDS serves a wave in order),
                // then every RECORD lane picks up the pairs of its record
                __builtin_amdgcn_wave_barrier();
                float4* ps = reinterpret_cast<float4*>(pair_lds + lane * kPairStride);
                ps[0] = make_float4(sm[0], sm[1], sm[2], sm[3]);
                ps[1] = make_float4(sm[4], 0.f, sm[5], sm[6]);
                ps[2] = make_float4(sm[7], sm[8], sm[9], 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            {
                const int pr[4] = {p0 - t0, p1 - t0, p2 - t0, p3 - t0};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if ((unsigned)pr[r] < (unsigned)kBT) {
                        const float4* q = reinterpret_cast<const float4*>(pair_lds + (pr[r] * 4 + r) * kPairStride);
                        const float4 qa = q[0], qb = q[1], qc = q[2];
                        racc[0] += qa.x; racc[1] += qa.y; racc[2] += qa.z; racc[3] += qa.w; racc[4] += qb.x;
                        racc[5] += qb.z; racc[6] += qb.w; racc[7] += qc.x; racc[8] += qc.y; racc[9] += qc.z;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // flush: the staged records (dense slots: every one of them hit some row) put their ten sums into the (now idle) exchange planes;
        // then 6 records x 10 components per global atomic instruction, software-pipelined
        __builtin_amdgcn_wave_barrier();
        {
            float* fls = s_m[wave][0];                          // 64 x 12 floats <= 2 planes
            if (lane < cnt) {
                float4* f4 = reinterpret_cast<float4*>(fls + lane * kPairStride);
                f4[0] = make_float4(racc[0], racc[1], racc[2], racc[3]);
                f4[1] = make_float4(racc[4], 0.f, racc[5], racc[6]);
                f4[2] = make_float4(racc[7], racc[8], racc[9], 0.f);
            }
            __builtin_amdgcn_wave_barrier();
            const int krec = min(fl_rec, 5);                   // lanes 60..63 idle along with record 5's mapping (never valid)
            // four groups per trip: eight unconditional LDS reads first (a slot behind the last staged record is clamped to the sentinel
            // slot), then the atomics, each under one predicate -- the conditional reads of the first version were a taken branch each and an
            // LDS round trip per group
            for (int g = 0; g < cnt; g += 24) {
                float val[4];
                uint32_t id[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int slot = g + 6 * u + krec, sl = min(slot, kWave);
                    const float v = fls[sl * kPairStride + fl_off];
                    id[u] = __float_as_uint(s2[sl].w);
                    val[u] = (lane < 60 && slot < cnt) ? v : 0.0f;
                }
                asm volatile("" ::: "memory");                 // (the reads stay in front of the atomics)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (val[u] != 0.0f) atomicAdd(grad2d + (size_t)id[u] * kGradStride + fl_comp, val[u]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        cnt = 0;
    }
    if (pieces > 1 && piece < pieces - 1) {            // hand the state on (no exit between the range set-up above and this point)
        float* out = chain_st + piece * 2 * kWave;
        __hip_atomic_store(out + lane, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(out + kWave + lane, S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        GS_WAIT_VMEM();                                // the state has arrived before the flag leaves
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(chain_fl + piece, cam.chain_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// images of at most this many tiles (gs_set_half_quadrants; the name dates from the first few-tile variant) take the few-tile kernels:
// 256 tiles x 4 quadrants are one walker per SIMD of this chip and every walker's chunk-by-chunk chain is exposed.  Forward: producer /
// consumer workgroups (blend_forward_pc_kernel: 256 x 256, 200 k Gaussians 80 -> 65 us, 1 M 87 -> 74 us; 120 x 150 76 -> 68 us); backward:
// three list segments per quadrant from the states that forward records (134 -> 88 us with two segments, -> 65 us with three).  Above 256 tiles the plain kernels win (400 tiles:
// backward 134 -> 172 us with the few-tile variant): hence 256
std::atomic<int> g_half_quadrant_tiles{256};
// pieces of a chained backward walk (images of more than kChainMinTiles tiles); 1 switches the chaining off (tests, A/B measurements)
std::atomic<int> g_chain_pieces{kChainPieces};
std::atomic<int> g_chain_min_tiles{kChainMinTiles};
// ordered tickets for the chained walks (gs_set_backward_chain_tickets): OFF by default -- the ticket is a fourth dependent round trip in front of a
// walker's prologue and measured +4-6 us on the 2 M frame's 182 us, +7 us on configs[1]'s 135 us (profiles/r05_ab_tickets.txt); without them the
// chain order is the workgroup index, which the dispatcher hands out in order.  Either way a wait is bounded and a timeout is reported.
std::atomic<int> g_chain_tickets{0};
std::atomic<int> g_chain_polls{kChainPollsDefault};      // bound of a piece's wait for the piece in front (gs_set_backward_chain_polls: tests)
uint32_t* g_async_status_dev = nullptr;      // device view of the host-mapped status word (api.hip: gs_async_status_word)
uint32_t* g_chain_fail_dev = nullptr;        // its sticky device-memory twin (Cam::chain_fail)
int g_chain_fail_device = -1;
// list segments (walkers) per quadrant in the few-tile backward: 3 x 256 tiles x 4 quadrants = the chip's 3072 walker slots (gs_set_backward_segments)
std::atomic<int> g_few_segments{kFewSegmentsMax};

hipError_t launch_blend_forward(const Cam& cam_in, const uint2* ranges, const uint32_t* point_list, const float4* geom,
                                float* out_color, float* out_depth, float* out_opacity, float* final_T,
                                uint32_t* n_contrib, float* out_depth_sq, uint32_t cap, int segments, float* seg_T, float* split_state, uint32_t P,
                                float* zero_fill, hipStream_t st)
{
    // whole-tile workgroups (NW = 4) here: one-wavefront workgroups measured 85 vs 80 us on configs[1] and the same at 2 M -- the
    // four walkers of a tile gather the same records, and on one CU three of them hit its L1
    Cam cam = cam_in;
    // (the development knobs are atomics read ONCE per launch: a knob changed on another thread -- the reference runs a visualiser thread next to the
    // mapper -- takes effect at a launch boundary, never inside one decision)
    const int k_half = g_half_quadrant_tiles.load(std::memory_order_relaxed), k_fseg = g_few_segments.load(std::memory_order_relaxed);
    cam.half = (segments <= 1 || !seg_T) && cam.gx * cam.gy <= k_half;          // few tiles: the producer / consumer forward
    cam.split = (cam.V == 1 && split_state && cam.gx * cam.gy <= min(k_half, kFewTiles)) && k_fseg > 1 ? k_fseg : 0;      // (the backward refuses atlases; > 0 makes the forward record)
    // (the forward's only use of `chain` is to zero the hand-over flags of the chained backward walks.  It does so for EVERY image whose
    // workspace holds them -- more than kFewTiles tiles -- whatever gs_set_backward_chain says at this moment: the backward takes its own
    // decision from the knob when IT is launched, and must find zeroed flags even if the knob changed in between)
    cam.chain = (cam.V == 1 && split_state && cam.gx * cam.gy > kFewTiles) ? kChainPieces : 0;
    const int nb = ((cam.gx * cam.gy + 7) >> 3) << 3;
#define GS_FWD(DSQ, SEG, GRID)                                                                                                     \
    hipLaunchKernelGGL((blend_forward_streams_kernel<DSQ, kFwdStreams, SEG, 4>), GRID, dim3(kBlock), 0, st, cam, ranges, point_list, geom, \
                       out_color, out_depth, out_opacity, final_T, n_contrib, out_depth_sq, cap, seg_T, split_state, P, (float4*)zero_fill)
    const size_t HW = (size_t)cam.W * cam.H;
    if (segments > 1 && seg_T) {
        // segmented compositing: the sums are added with atomics, so the images start from zero
        hipError_t e = zero_fill ? hipMemsetAsync(zero_fill, 0, (size_t)P * kGradStride * sizeof(float), st) : hipSuccess;
        if (e == hipSuccess)       // the quadrants' "an earlier segment saturates" flag words behind the transmittances
            e = hipMemsetAsync(seg_T + (size_t)cam.gx * cam.gy * segments * kBlock, 0, (size_t)cam.gx * cam.gy * 4 * sizeof(uint32_t), st);
        if (e == hipSuccess) e = hipMemsetAsync(out_color, 0, 3 * HW * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(out_depth, 0, HW * sizeof(float), st);
        if (e == hipSuccess && out_depth_sq) e = hipMemsetAsync(out_depth_sq, 0, HW * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(n_contrib, 0, HW * sizeof(uint32_t), st);
        if (e == hipSuccess && split_state && cam.gx * cam.gy <= kFewTiles)
            e = hipMemsetAsync(split_state + (kCutLevels * 5 + 4) * HW, 0, sizeof(uint32_t), st);    // nothing recorded
        if (e == hipSuccess && cam.chain > 1)      // (the segmented kernels do not clear the chained backward's hand-over flags)
            e = hipMemsetAsync(split_state + (size_t)cam.gx * cam.gy * 4 * kChainStateFloats, 0, chain_flag_words((size_t)cam.gx * cam.gy) * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
        const dim3 grid(nb, segments);
        if (out_depth_sq) { GS_FWD(true, 1, grid); GS_FWD(true, 2, grid); }
        else { GS_FWD(false, 1, grid); GS_FWD(false, 2, grid); }
    } else if (cam.half || cam.split) {
        // images of few tiles: one producer / consumer workgroup per tile
        if (cam.chain > 1) {      // (an image of more than kFewTiles tiles sent here by gs_set_half_quadrants: this kernel does not clear the hand-over flags)
            hipError_t e = hipMemsetAsync(split_state + (size_t)cam.gx * cam.gy * 4 * kChainStateFloats, 0, chain_flag_words((size_t)cam.gx * cam.gy) * sizeof(uint32_t), st);
            if (e != hipSuccess) return e;
        }
        const dim3 grid(((cam.gx * cam.gy + 7) >> 3) << 3), block(kPcWaves * kWave);
        if (out_depth_sq)
            hipLaunchKernelGGL((blend_forward_pc_kernel<true>), grid, block, 0, st, cam, ranges, point_list, geom, out_color, out_depth, out_opacity,
                               final_T, n_contrib, out_depth_sq, cap, split_state, P, (float4*)zero_fill);
        else
            hipLaunchKernelGGL((blend_forward_pc_kernel<false>), grid, block, 0, st, cam, ranges, point_list, geom, out_color, out_depth, out_opacity,
                               final_T, n_contrib, out_depth_sq, cap, split_state, P, (float4*)zero_fill);
    } else {
        if (split_state && cam.gx * cam.gy <= kFewTiles) {          // a small image with the few-tile paths switched off: "nothing recorded"
            hipError_t e = hipMemsetAsync(split_state + (kCutLevels * 5 + 4) * HW, 0, sizeof(uint32_t), st);
            if (e != hipSuccess) return e;
        }
        if (out_depth_sq) GS_FWD(true, 0, dim3(nb)); else GS_FWD(false, 0, dim3(nb));
    }
#undef GS_FWD
    return hipGetLastError();
}

hipError_t launch_blend_backward(const Cam& cam_in, const uint2* ranges, const uint32_t* point_list, const float4* geom, const float* split_state,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 const float* dL_ddepth, float* grad2d, hipStream_t st)
{
    // one-wavefront workgroups (see blend_backward_kernel): per XCD band ceil(tiles/8) tiles x 4 quadrants; images of few tiles: x 2 list
    // segments (the forward of such an image has recorded the state at the cut; if it has not, the back walkers exit at once)
    Cam cam = cam_in;
    cam.half = 0;
    const int k_half = g_half_quadrant_tiles.load(std::memory_order_relaxed), k_fseg = g_few_segments.load(std::memory_order_relaxed);
    const int k_pieces = g_chain_pieces.load(std::memory_order_relaxed), k_min = g_chain_min_tiles.load(std::memory_order_relaxed);
    cam.split = (split_state != nullptr && cam.gx * cam.gy <= min(k_half, kFewTiles) && k_fseg > 1) ? k_fseg : 0;
    cam.chain = (cam.V == 1 && split_state && cam.gx * cam.gy > max(k_min, kFewTiles) && k_pieces > 1) ? k_pieces : 0;
    static std::atomic<unsigned> epoch{0};
    do { cam.chain_epoch = ++epoch; } while (cam.chain_epoch == 0u);          // (the forward leaves zero in the hand-over flags)
    cam.chain_ticket = g_chain_tickets.load(std::memory_order_relaxed); cam.chain_polls = g_chain_polls.load(std::memory_order_relaxed);
    cam.async_status = g_async_status_dev;
    cam.chain_fail = chain_fail_word();
    const int per = ((cam.gx * cam.gy + 7) >> 3) * (cam.split ? cam.split : cam.chain > 1 ? cam.chain : 1);
#define GS_BWD(DG, FEW)                                                                                                          \
    hipLaunchKernelGGL((blend_backward_kernel<DG, 1, FEW>), dim3(per * 8 * 4), dim3(kWave), 0, st, cam, ranges, point_list, geom, final_T, \
                       n_contrib, dL_dcolor, dL_ddepth, grad2d, split_state)
    if (cam.split) { if (dL_ddepth) GS_BWD(true, true); else GS_BWD(false, true); }
    else { if (dL_ddepth) GS_BWD(true, false); else GS_BWD(false, false); }
#undef GS_BWD
    return hipGetLastError();
}

}  // namespace gs
