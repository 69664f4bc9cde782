// gs_common.h -- shared declarations of the gfx950 rasteriser kernels (internal; the public
// boundary is include/gsplat_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include <atomic>

#include "../../include/gsplat_hip.h"

namespace gs {

constexpr int kTile = GS_TILE;      // 16x16 pixel tile = one 256-thread workgroup = 4 wavefronts
constexpr int kBlock = 256;
constexpr int kWave = 64;           // CDNA wavefront
constexpr int kQuad = 8;            // each wavefront owns one 8x8 pixel quadrant of the tile
#ifndef GS_BIN_CHUNK
#define GS_BIN_CHUNK 2048           // (A/B knob: scripts/exp/build_variant.sh c4096 -DGS_BIN_CHUNK=4096)
#endif
constexpr int kBinChunk = GS_BIN_CHUNK;     // Gaussians per tile-binning workgroup (LDS-private histogram)
constexpr int kMaxLdsTiles = 8192;  // tile-binning path needs the tile histogram in LDS (32 KiB)
constexpr int kSortChunk = 2048;    // keys one workgroup bitonic-sorts in LDS
constexpr int kSortCapMax = 16384;  // largest per-tile list whose sorted chunks are rank-merged in LDS (128 KiB); beyond -> pairwise merge passes

// By-value kernel argument; matrices stay in device memory exactly where the caller's settings
// tensors put them (uniform loads -> scalar cache).
struct Cam {
    int W, H, gx, gy;
    // multi-view atlas (planner panoramas): V views of Wv x H pixels side by side, view v in tile columns [v gxv, (v+1) gxv);
    // the per-Gaussian stage runs over V x Ppad VIRTUAL Gaussians (view-major, nbv 256-row blocks per view), every later stage
    // sees one image of W = V gxv 16 pixels.  V = 1: Wv = W, gxv = gx.
    int V, Wv, gxv, nbv;
    float tanfovx, tanfovy, fx, fy, mod;
    int sh_degree, sh_coeffs;
    const float* bg;
    const float* view;
    const float* proj;
    const float* campos;
    // forward blend, images of few tiles: the producer / consumer kernel (set by the blend launchers)
    int half;
    // backward blend, images of few tiles: every tile list is walked in `split` (2 or 3) segments by as many wavefronts per quadrant; the front
    // one starts from the per-pixel state the forward left at the boundary (set by the blend launchers)
    int split;
    // backward blend, images of MORE quadrants than the chip holds walkers (3 wavefronts x 1024 SIMDs): every quadrant's walk is cut into
    // `chain` consecutive pieces run by `chain` workgroups in dispatch order, the running state handed on through memory (set by the blend
    // launchers; 0 / 1: one walker per quadrant)
    int chain;
    unsigned chain_epoch;
    // chained walks, robustness: chain_ticket != 0: a workgroup's (quadrant, piece) comes from an ORDERED TICKET it draws when it starts (one counter
    // per blockIdx & 7), not from its index -- a piece then only ever waits for a workgroup that is already running or done, whatever order the
    // hardware starts workgroups in; chain_polls bounds the wait all the same; a walker whose wait runs out sets bit 0 of *async_status
    // (host-mapped, gs_async_status) and goes on with NaN state
    int chain_ticket, chain_polls;
    uint32_t* async_status;
    // the same event for the DEVICE: a sticky word in device memory that the timed-out walker sets next to the host-visible one.  Every kernel that
    // applies an optimiser step (adam.hip, rows.hip, the Adam inside the per-Gaussian backward) reads it first and leaves parameters and moments
    // untouched while it is set -- the backward is asynchronous, the host learns of the timeout a render later, and by then a step on NaN gradients
    // would have destroyed the map in place.  Cleared by the host when it has reported the event (gs_async_status_clear)
    uint32_t* chain_fail;
    // raw-parameter mode of the per-Gaussian kernels (gs_preprocess_forward_raw / gs_render_backward_raw): the inputs are the mapper's
    // PARAMETERS -- world-frame means, unnormalised quaternions, logit opacities, log scales ([P,1] when act_iso) -- and the frame transform
    // + activations of slam_helpers.py:252-304,124-139 (activate.hip) happen inside the kernels; act_accumulate: the backward ADDS its
    // parameter gradients to what the output buffers hold
    int act, act_iso, act_accumulate;
    float act_q[4], act_t[3];
};

// frame transform + activation algebra shared by activate.hip and the raw-parameter mode of the per-Gaussian kernels
__device__ __forceinline__ void quat_to_rot(const float* q, float (&R)[3][3])
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}
// m = a (x) b  (Hamilton product, w first)
__device__ __forceinline__ void qmul(const float* a, const float* b, float* m)
{
    m[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    m[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    m[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    m[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
// du = L(a)^T dm  where m = a (x) u is linear in u
__device__ __forceinline__ void qmul_bwd_rhs(const float* a, const float* dm, float* du)
{
    du[0] = a[0] * dm[0] + a[1] * dm[1] + a[2] * dm[2] + a[3] * dm[3];
    du[1] = -a[1] * dm[0] + a[0] * dm[1] + a[3] * dm[2] - a[2] * dm[3];
    du[2] = -a[2] * dm[0] - a[3] * dm[1] + a[0] * dm[2] + a[1] * dm[3];
    du[3] = -a[3] * dm[0] + a[2] * dm[1] - a[1] * dm[2] + a[0] * dm[3];
}
// rotation the rasteriser sees for the parameter quaternion q (w first): normalize(q) (isotropic map), else normalize(q_cam (x) normalize(q))
__device__ __forceinline__ void activate_rotation(const float* cam_q, int iso, const float* q, float* out)
{
    const float inv = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    const float u[4] = {q[0] * inv, q[1] * inv, q[2] * inv, q[3] * inv};
    if (iso) { for (int k = 0; k < 4; k++) out[k] = u[k]; return; }
    float m[4];
    qmul(cam_q, u, m);
    const float invm = 1.0f / fmaxf(sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3]), 1e-12f);
    for (int k = 0; k < 4; k++) out[k] = m[k] * invm;
}
// gradient w.r.t. the parameter quaternion q of a gradient g w.r.t. activate_rotation's output
__device__ __forceinline__ void activate_rotation_bwd(const float* cam_q, int iso, const float* q, const float* g, float* dq)
{
    const float nq = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    const float u[4] = {q[0] / nq, q[1] / nq, q[2] / nq, q[3] / nq};
    float du[4];
    if (iso) {
        for (int k = 0; k < 4; k++) du[k] = g[k];
    } else {
        float m[4];
        qmul(cam_q, u, m);
        const float nm = fmaxf(sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2] + m[3] * m[3]), 1e-12f);
        const float r[4] = {m[0] / nm, m[1] / nm, m[2] / nm, m[3] / nm};
        const float dot = r[0] * g[0] + r[1] * g[1] + r[2] * g[2] + r[3] * g[3];
        float dm[4];
        for (int k = 0; k < 4; k++) dm[k] = (g[k] - r[k] * dot) / nm;
        qmul_bwd_rhs(cam_q, dm, du);
    }
    const float dotu = u[0] * du[0] + u[1] * du[1] + u[2] * du[2] + u[3] * du[3];
    for (int k = 0; k < 4; k++) dq[k] = (du[k] - u[k] * dotu) / nq;
}

// The optimiser step INSIDE the raw-parameter backward (gs_render_backward_raw_adam; single-keyframe steps -- the reference's loop,
// src/mapper/splatam/__init__.py:470-480, and BASELINE configs[2]'s): the thread that forms a Gaussian's parameter gradient applies
// Adam to the parameter and its two moments in place -- no gradient tensor is written and read back.  Tensor order: 0 means3D,
// 1 logit opacities, 2 log scales, 3 unnormalised rotations, 4 colours or 16-coefficient SH rows.
struct AdamCoef { float one_m_b1, b2, one_m_b2, step_size, inv_bc2s, eps; };
struct FusedAdam {
    float* p[5];
    float* m[5];
    float* v[5];
    AdamCoef c[5];
    const uint32_t* fail;      // Cam::chain_fail: set -> the step is skipped
};

// Per-Gaussian screen-space record, 3 x float4 = 48 B, one gather per tile instance in the blend.
//   q0 = (x, y, conic_a, conic_b)   q1 = (conic_c, opacity, r, g)   q2 = (b, depth, ext_x, ext_y)
// ext_x/ext_y: half-extent (pixels) of the axis-aligned box outside of which alpha < 1/255 is
// guaranteed; negative when the Gaussian can never reach 1/255.  Used only to skip work.
struct GeomPtrs {
    float4* geom;
    uint2* rect;
    uint32_t* tiles;
    uint32_t* offsets;
    uint32_t* block_sums;
    uint32_t* clamped;   // uchar4 packed
    uint32_t* tile_total;  // [tiles]
    uint32_t* tile_base;   // [ceil(P/kBinChunk)][tiles]
    uint32_t* depth_bits;  // [P]: bit pattern of the view-space depth (the binning key), compact copy of geom[.][9] for coalesced reads
    float2* sh_jac;        // [P][5]: d(rgb before the clamp)/d(unit view direction), 3x3 row-major, and the colour clamp flags in the tenth word
                           // (SH inputs with a backward to follow): 40 B per Gaussian
    // raw-parameter mode only (caller's tensors, not part of the workspace): the mapper's visibility statistics of this render --
    // vis_max[i] = max(vis_max[i], radius), vis_seen[i] = radius > 0 (splatam.py:296-298) -- written where the radius is computed
    float* vis_max;
    uint8_t* vis_seen;
};

// wait for every outstanding vector-memory operation of this wavefront: orders a device-scope atomic store behind earlier ones without a
// release fence (which writes the L2 back: ~20 ns per workgroup when thousands issue them, profiles/README.md).  The host emulator defines it empty.
#ifndef GS_WAIT_VMEM
#define GS_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// ---- wave-64 helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// inclusive prefix sum across the 64 lanes of a wavefront
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Coalesced staging of `nrows` rows of K floats (array-of-structs in HBM) into LDS: the full-block
// case moves 16 B per lane per instruction; rows are then read back at stride K (conflict-free for
// K = 3, and as one ds_read_b128 for K = 4).
template <int K>
__device__ __forceinline__ void stage_rows(float* lds, const float* __restrict__ src, int base, int nrows, int tid)
{
    const float* s = src + (size_t)base * K;
    if (nrows == kBlock) {
        const float4* s4 = reinterpret_cast<const float4*>(s);
        float4* d4 = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < K * (kBlock / 4); i += kBlock) d4[i] = s4[i];
    } else {
        for (int i = tid; i < nrows * K; i += kBlock) lds[i] = s[i];
    }
}

// SH coefficient rows ([P][M][3] floats, M*3 = K floats per Gaussian, 192 B at degree 3) are moved between HBM
// and a wave's threads through an LDS slab with an ODD row stride (conflict-free lane = row access): one
// wavefront's 64 rows at a time, all 256 threads of the workgroup doing the coalesced global side.
// streaming (non-temporal) 16-byte load: data that is read exactly once should not evict the L2's working set
__device__ __forceinline__ float4 load_stream(const float4* p)
{
    typedef float v4f __attribute__((vector_size(16)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store_stream(float4* p, const float4& x)
{
    typedef float v4f __attribute__((vector_size(16)));
    const v4f v = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}

constexpr int kShPad = 49;          // LDS row stride for up to 48 floats (16 coefficients x RGB)
__device__ __forceinline__ int sh_row_stride(int K) { return K | 1; }
// global rows [row0, row0+nrows) -> LDS (padded); nrows <= 64
__device__ __forceinline__ void sh_rows_to_lds(float* lds, const float* __restrict__ src, int row0, int nrows, int K, int tid)
{
    const int stride = sh_row_stride(K);
    const float* s = src + (size_t)row0 * K;
    for (int e = tid; e < nrows * K; e += kBlock) {
        const int r = e / K;
        lds[r * stride + (e - r * K)] = s[e];
    }
}
// LDS (padded) -> global rows
__device__ __forceinline__ void sh_rows_from_lds(const float* lds, float* __restrict__ dst, int row0, int nrows, int K, int tid)
{
    const int stride = sh_row_stride(K);
    float* d = dst + (size_t)row0 * K;
    for (int e = tid; e < nrows * K; e += kBlock) {
        const int r = e / K;
        d[e] = lds[r * stride + (e - r * K)];
    }
}

// Wavefront-private variants: one wave moves `nrows` (<= kShHalf) rows between global memory and ITS OWN padded slab,
// 16 B per lane per step, no workgroup barrier -- the four waves of a workgroup overlap their load / compute / store
// phases instead of taking turns.  row0 * K * 4 bytes is 16-byte aligned because row0 is a multiple of 32.
constexpr int kShHalf = 32;
// row_mask: bit r set = row r is needed (rows of culled Gaussians are not fetched; their slab contents stay undefined)
// KC > 0: the row width is a compile-time constant (16 coefficients: KC = 48) -- the element -> (row, column) divisions of the
// copy loops fold into multiplies; KC = 0: runtime K
template <int KC = 0>
__device__ __forceinline__ void sh_wave_rows_to_lds(float* slab, const float* __restrict__ src, int row0, int nrows, int Krt, int lane,
                                                    uint32_t row_mask = 0xffffffffu)
{
    const int K = KC > 0 ? KC : Krt;
    const int stride = sh_row_stride(K), total = nrows * K, total4 = total >> 2;
    const float* s = src + (size_t)row0 * K;
    const float4* s4 = reinterpret_cast<const float4*>(s);
    for (int q = lane; q < total4; q += kWave) {
        int e = q << 2, r = e / K, c = e - r * K;
        const int r_last = (e + 3) / K;                    // a 16-byte piece may straddle two rows
        if (!(((row_mask >> r) | (row_mask >> min(r_last, 31))) & 1u)) continue;
        const float4 v = load_stream(&s4[q]);              // coefficient rows are read once per pass
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            slab[r * stride + c] = vv[t];
            if (++c == K) { c = 0; ++r; }
        }
    }
    for (int e = (total4 << 2) + lane; e < total; e += kWave) {
        const int r = e / K;
        if ((row_mask >> r) & 1u) slab[r * stride + (e - r * K)] = s[e];
    }
}
// The same copy split in two for full 32-row halves of 48-float rows (six 16-byte pieces per lane): the loads of BOTH halves of a
// wavefront's 64 rows are issued before the first half is consumed (twice the bytes in flight per wave).
__device__ __forceinline__ void sh48_half_load(float4 (&v)[6], const float* __restrict__ src, int row0, int lane)
{
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)row0 * 48);
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] = load_stream(&s4[lane + kWave * j]);
}
// 16-coefficient rows (48 floats) in slabs with a 52-float row stride: rows stay 16-byte aligned, so a 16-byte piece of a row moves with ONE
// ds_write_b128 / ds_read_b128 (the 49-float stride needs four scalar LDS operations per piece), and lane = row accesses are conflict-free
// (52 r mod 64 takes 16 distinct multiples of 4 over the lanes a 128-bit LDS access serves together).
constexpr int kShPad4 = 52;
__device__ __forceinline__ void sh48_half_to_lds4(float* slab, const float4 (&v)[6], int lane)
{
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int q = lane + kWave * j, r = q / 12, c = q - 12 * r;
        *reinterpret_cast<float4*>(slab + r * kShPad4 + 4 * c) = v[j];
    }
}
// any number of rows (<= kShHalf) with a row mask: the last wavefront of the array, and rows that are not needed
__device__ __forceinline__ void sh48_rows_to_lds4(float* slab, const float* __restrict__ src, int row0, int nrows, int lane, uint32_t row_mask = 0xffffffffu)
{
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)row0 * 48);
    for (int q = lane; q < nrows * 12; q += kWave) {
        const int r = q / 12, c = q - 12 * r;
        if ((row_mask >> r) & 1u) *reinterpret_cast<float4*>(slab + r * kShPad4 + 4 * c) = load_stream(&s4[q]);
    }
}
template <int KC = 0>
__device__ __forceinline__ void sh_wave_rows_from_lds(const float* slab, float* __restrict__ dst, int row0, int nrows, int Krt, int lane)
{
    const int K = KC > 0 ? KC : Krt;
    const int stride = sh_row_stride(K), total = nrows * K, total4 = total >> 2;
    float* d = dst + (size_t)row0 * K;
    float4* d4 = reinterpret_cast<float4*>(d);
    for (int q = lane; q < total4; q += kWave) {
        int e = q << 2, r = e / K, c = e - r * K;
        float vv[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            vv[t] = slab[r * stride + c];
            if (++c == K) { c = 0; ++r; }
        }
        store_stream(&d4[q], make_float4(vv[0], vv[1], vv[2], vv[3]));
    }
    for (int e = (total4 << 2) + lane; e < total; e += kWave) {
        const int r = e / K;
        d[e] = slab[r * stride + (e - r * K)];
    }
}

// real-SH basis of the 3DGS family and its gradient w.r.t. the unit direction
__device__ __forceinline__ void sh_basis_and_grad(int deg, float x, float y, float z, float* b, float* bx, float* by, float* bz)
{
    for (int k = 0; k < 16; k++) { b[k] = 0.f; bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; }
    const float C1 = 0.4886025119029199f;
    b[0] = 0.28209479177387814f;
    if (deg > 0) {
        b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
        by[1] = -C1; bz[2] = C1; bx[3] = -C1;
        if (deg > 1) {
            const float c20 = 1.0925484305920792f, c21 = -1.0925484305920792f, c22 = 0.31539156525252005f,
                        c23 = -1.0925484305920792f, c24 = 0.5462742152960396f;
            const float xx = x * x, yy = y * y, zz = z * z;
            b[4] = c20 * x * y; b[5] = c21 * y * z; b[6] = c22 * (2.f * zz - xx - yy); b[7] = c23 * x * z; b[8] = c24 * (xx - yy);
            bx[4] = c20 * y; by[4] = c20 * x;
            by[5] = c21 * z; bz[5] = c21 * y;
            bx[6] = c22 * (-2.f * x); by[6] = c22 * (-2.f * y); bz[6] = c22 * (4.f * z);
            bx[7] = c23 * z; bz[7] = c23 * x;
            bx[8] = c24 * (2.f * x); by[8] = c24 * (-2.f * y);
            if (deg > 2) {
                const float c30 = -0.5900435899266435f, c31 = 2.890611442640554f, c32 = -0.4570457994644658f,
                            c33 = 0.3731763325901154f, c34 = -0.4570457994644658f, c35 = 1.445305721320277f,
                            c36 = -0.5900435899266435f;
                b[9] = c30 * y * (3.f * xx - yy); b[10] = c31 * x * y * z; b[11] = c32 * y * (4.f * zz - xx - yy);
                b[12] = c33 * z * (2.f * zz - 3.f * xx - 3.f * yy); b[13] = c34 * x * (4.f * zz - xx - yy);
                b[14] = c35 * z * (xx - yy); b[15] = c36 * x * (xx - 3.f * yy);
                bx[9] = c30 * (6.f * x * y); by[9] = c30 * (3.f * xx - 3.f * yy);
                bx[10] = c31 * y * z; by[10] = c31 * x * z; bz[10] = c31 * x * y;
                bx[11] = c32 * (-2.f * x * y); by[11] = c32 * (4.f * zz - xx - 3.f * yy); bz[11] = c32 * (8.f * y * z);
                bx[12] = c33 * (-6.f * x * z); by[12] = c33 * (-6.f * y * z); bz[12] = c33 * (6.f * zz - 3.f * xx - 3.f * yy);
                bx[13] = c34 * (4.f * zz - 3.f * xx - yy); by[13] = c34 * (-2.f * x * y); bz[13] = c34 * (8.f * x * z);
                bx[14] = c35 * (2.f * x * z); by[14] = c35 * (-2.f * y * z); bz[14] = c35 * (xx - yy);
                bx[15] = c36 * (3.f * xx - 3.f * yy); by[15] = c36 * (-6.f * x * y);
            }
        }
    }
}


// J[ch][xyz] = sum_k coef[k][ch] * grad b_k(x, y, z) for the active degree, WITHOUT materialising the 48 gradient values: one
// coefficient at a time, three FMAs per channel (rows `sh` hold coef[k][ch] at 3k + ch).  Used by the forward to spare the backward a
// second read of the coefficient rows.
#define GS_SHJ(k, gx, gy, gz)                                                                                                  \
    do {                                                                                                                        \
        const float c0_ = sh[3 * (k)], c1_ = sh[3 * (k) + 1], c2_ = sh[3 * (k) + 2];                                            \
        const float gx_ = (gx), gy_ = (gy), gz_ = (gz);                                                                         \
        J[0] = fmaf(c0_, gx_, J[0]); J[1] = fmaf(c0_, gy_, J[1]); J[2] = fmaf(c0_, gz_, J[2]);                                  \
        J[3] = fmaf(c1_, gx_, J[3]); J[4] = fmaf(c1_, gy_, J[4]); J[5] = fmaf(c1_, gz_, J[5]);                                  \
        J[6] = fmaf(c2_, gx_, J[6]); J[7] = fmaf(c2_, gy_, J[7]); J[8] = fmaf(c2_, gz_, J[8]);                                  \
    } while (0)
__device__ __forceinline__ void sh_direction_jacobian(int deg, float x, float y, float z, const float* sh, float (&J)[9])
{
#pragma unroll
    for (int q = 0; q < 9; q++) J[q] = 0.0f;
    if (deg < 1) return;
    const float C1 = 0.4886025119029199f;
    GS_SHJ(1, 0.f, -C1, 0.f); GS_SHJ(2, 0.f, 0.f, C1); GS_SHJ(3, -C1, 0.f, 0.f);
    if (deg < 2) return;
    const float c20 = 1.0925484305920792f, c21 = -1.0925484305920792f, c22 = 0.31539156525252005f, c23 = -1.0925484305920792f,
                c24 = 0.5462742152960396f;
    GS_SHJ(4, c20 * y, c20 * x, 0.f);
    GS_SHJ(5, 0.f, c21 * z, c21 * y);
    GS_SHJ(6, c22 * (-2.f * x), c22 * (-2.f * y), c22 * (4.f * z));
    GS_SHJ(7, c23 * z, 0.f, c23 * x);
    GS_SHJ(8, c24 * (2.f * x), c24 * (-2.f * y), 0.f);
    if (deg < 3) return;
    const float c30 = -0.5900435899266435f, c31 = 2.890611442640554f, c32 = -0.4570457994644658f, c33 = 0.3731763325901154f,
                c34 = -0.4570457994644658f, c35 = 1.445305721320277f, c36 = -0.5900435899266435f;
    const float xx = x * x, yy = y * y, zz = z * z;
    GS_SHJ(9, c30 * (6.f * x * y), c30 * (3.f * xx - 3.f * yy), 0.f);
    GS_SHJ(10, c31 * y * z, c31 * x * z, c31 * x * y);
    GS_SHJ(11, c32 * (-2.f * x * y), c32 * (4.f * zz - xx - 3.f * yy), c32 * (8.f * y * z));
    GS_SHJ(12, c33 * (-6.f * x * z), c33 * (-6.f * y * z), c33 * (6.f * zz - 3.f * xx - 3.f * yy));
    GS_SHJ(13, c34 * (4.f * zz - 3.f * xx - yy), c34 * (-2.f * x * y), c34 * (8.f * x * z));
    GS_SHJ(14, c35 * (2.f * x * z), c35 * (-2.f * y * z), c35 * (xx - yy));
    GS_SHJ(15, c36 * (3.f * xx - 3.f * yy), c36 * (-6.f * x * y), 0.f);
}
#undef GS_SHJ

// Has a chained backward walk of this process timed out since the host last cleared the word (Cam::chain_fail)?  One cached 4-byte load per workgroup.
__device__ __forceinline__ bool chain_failed(const uint32_t* w)
{
    return w != nullptr && __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// One Adam update (torch's single-tensor arithmetic; adam.hip and rows.hip share it so that the keyframe-sharded step is the full step to the bit)
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float one_m_b1, float b2, float one_m_b2,
                                          float step_size, float inv_bc2s, float eps)
{
    // no FMA contraction here: which products the compiler fuses would otherwise depend on the kernel the function is inlined into
#pragma clang fp contract(off)
    m = m + one_m_b1 * (g - m);
    v = b2 * v + (one_m_b2 * g) * g;
    const float denom = sqrtf(v) * inv_bc2s + eps;
    p = p - step_size * (m / denom);
}

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamCoef& c)
{
    adam_elem(p, g, m, v, c.one_m_b1, c.b2, c.one_m_b2, c.step_size, c.inv_bc2s, c.eps);
}
// host side: torch's bias corrections, in double, as launch_adam_multi forms them
inline AdamCoef adam_coef(double lr, double beta1, double beta2, double eps, int step)
{
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamCoef c;
    c.one_m_b1 = (float)(1.0 - beta1); c.b2 = (float)beta2; c.one_m_b2 = (float)(1.0 - beta2);
    c.step_size = (float)(lr / bc1); c.inv_bc2s = (float)(1.0 / sqrt(bc2)); c.eps = (float)eps;
    return c;
}

// Colour sums and (optionally) the direction Jacobian of ONE 16-coefficient row that sits 16-byte aligned in LDS: the row is read four
// coefficients (three ds_read_b128) at a time and consumed at once, in the order of the scalar formulation above (colour: k ascending, one
// FMA per channel; Jacobian: sh_direction_jacobian's sequence) -- same results to the bit, a dozen live registers instead of 48.
#define GS_SHC(j, k)                                                                                                           \
    do {                                                                                                                        \
        acc[0] = fmaf(b[k], f[3 * (j)], acc[0]); acc[1] = fmaf(b[k], f[3 * (j) + 1], acc[1]); acc[2] = fmaf(b[k], f[3 * (j) + 2], acc[2]);    \
    } while (0)
#define GS_SHJ4(j, gx, gy, gz)                                                                                                 \
    do {                                                                                                                        \
        const float c0_ = f[3 * (j)], c1_ = f[3 * (j) + 1], c2_ = f[3 * (j) + 2];                                               \
        const float gx_ = (gx), gy_ = (gy), gz_ = (gz);                                                                         \
        J[0] = fmaf(c0_, gx_, J[0]); J[1] = fmaf(c0_, gy_, J[1]); J[2] = fmaf(c0_, gz_, J[2]);                                  \
        J[3] = fmaf(c1_, gx_, J[3]); J[4] = fmaf(c1_, gy_, J[4]); J[5] = fmaf(c1_, gz_, J[5]);                                  \
        J[6] = fmaf(c2_, gx_, J[6]); J[7] = fmaf(c2_, gy_, J[7]); J[8] = fmaf(c2_, gz_, J[8]);                                  \
    } while (0)
#define GS_SHLOAD(g)                                                                                                           \
    const float4 q0_##g = row4[3 * (g)], q1_##g = row4[3 * (g) + 1], q2_##g = row4[3 * (g) + 2];                              \
    const float f[12] = {q0_##g.x, q0_##g.y, q0_##g.z, q0_##g.w, q1_##g.x, q1_##g.y, q1_##g.z, q1_##g.w, q2_##g.x, q2_##g.y, q2_##g.z, q2_##g.w}
__device__ __forceinline__ void sh48_color_and_jacobian(int deg, float x, float y, float z, const float (&b)[16], const float* row, bool want_j,
                                                        float (&acc)[3], float (&J)[9])
{
    const float4* row4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int q = 0; q < 9; q++) J[q] = 0.0f;
    acc[0] = acc[1] = acc[2] = 0.0f;
    const float C1 = 0.4886025119029199f;
    const float c20 = 1.0925484305920792f, c21 = -1.0925484305920792f, c22 = 0.31539156525252005f, c23 = -1.0925484305920792f,
                c24 = 0.5462742152960396f;
    const float c30 = -0.5900435899266435f, c31 = 2.890611442640554f, c32 = -0.4570457994644658f, c33 = 0.3731763325901154f,
                c34 = -0.4570457994644658f, c35 = 1.445305721320277f, c36 = -0.5900435899266435f;
    const float xx = x * x, yy = y * y, zz = z * z;
    {
        GS_SHLOAD(0);
        GS_SHC(0, 0);
        if (deg < 1) return;
        GS_SHC(1, 1); GS_SHC(2, 2); GS_SHC(3, 3);
        if (want_j) { GS_SHJ4(1, 0.f, -C1, 0.f); GS_SHJ4(2, 0.f, 0.f, C1); GS_SHJ4(3, -C1, 0.f, 0.f); }
    }
    if (deg < 2) return;
    {
        GS_SHLOAD(1);
        GS_SHC(0, 4); GS_SHC(1, 5); GS_SHC(2, 6); GS_SHC(3, 7);
        if (want_j) {
            GS_SHJ4(0, c20 * y, c20 * x, 0.f);
            GS_SHJ4(1, 0.f, c21 * z, c21 * y);
            GS_SHJ4(2, c22 * (-2.f * x), c22 * (-2.f * y), c22 * (4.f * z));
            GS_SHJ4(3, c23 * z, 0.f, c23 * x);
        }
    }
    {
        GS_SHLOAD(2);
        GS_SHC(0, 8);
        if (want_j) GS_SHJ4(0, c24 * (2.f * x), c24 * (-2.f * y), 0.f);
        if (deg >= 3) {
            GS_SHC(1, 9); GS_SHC(2, 10); GS_SHC(3, 11);
            if (want_j) {
                GS_SHJ4(1, c30 * (6.f * x * y), c30 * (3.f * xx - 3.f * yy), 0.f);
                GS_SHJ4(2, c31 * y * z, c31 * x * z, c31 * x * y);
                GS_SHJ4(3, c32 * (-2.f * x * y), c32 * (4.f * zz - xx - 3.f * yy), c32 * (8.f * y * z));
            }
        }
    }
    if (deg < 3) return;
    {
        GS_SHLOAD(3);
        GS_SHC(0, 12); GS_SHC(1, 13); GS_SHC(2, 14); GS_SHC(3, 15);
        if (want_j) {
            GS_SHJ4(0, c33 * (-6.f * x * z), c33 * (-6.f * y * z), c33 * (6.f * zz - 3.f * xx - 3.f * yy));
            GS_SHJ4(1, c34 * (4.f * zz - 3.f * xx - yy), c34 * (-2.f * x * y), c34 * (8.f * x * z));
            GS_SHJ4(2, c35 * (2.f * x * z), c35 * (-2.f * y * z), c35 * (xx - yy));
            GS_SHJ4(3, c36 * (3.f * xx - 3.f * yy), c36 * (-6.f * x * y), 0.f);
        }
    }
}
#undef GS_SHC
#undef GS_SHJ4
#undef GS_SHLOAD

// the saved SH Jacobian record: nine floats + the clamp flags (what `clamped` holds), five 8-byte pieces
constexpr int kShJacFloats = 10;
__device__ __forceinline__ void store_sh_jac(float2* base, size_t i, const float (&J)[9], uint32_t clamp_bits)
{
    float2* o = base + i * (kShJacFloats / 2);
    o[0] = make_float2(J[0], J[1]); o[1] = make_float2(J[2], J[3]); o[2] = make_float2(J[4], J[5]); o[3] = make_float2(J[6], J[7]);
    o[4] = make_float2(J[8], __uint_as_float(clamp_bits));
}
__device__ __forceinline__ void load_sh_jac(const float2* base, size_t i, float (&J)[9], uint32_t& clamp_bits)
{
    const float2* o = base + i * (kShJacFloats / 2);
    const float2 a = o[0], b = o[1], c = o[2], d = o[3], e = o[4];
    J[0] = a.x; J[1] = a.y; J[2] = b.x; J[3] = b.y; J[4] = c.x; J[5] = c.y; J[6] = d.x; J[7] = d.y; J[8] = e.x;
    clamp_bits = __float_as_uint(e.y);
}

// ---- launchers implemented in the individual translation units -------------------------------------
hipError_t launch_preprocess_forward(const Cam& cam, int P, const float* means3D, const float* shs,
                                     const float* colors, const float* opac, const float* scales,
                                     const float* rots, const float* cov3Dp, int32_t* radii, GeomPtrs gp,
                                     uint32_t* d_num_rendered, hipStream_t st);
hipError_t launch_scan_block_sums(int P, GeomPtrs gp, uint32_t* d_total, hipStream_t st);
hipError_t launch_preprocess_backward(const Cam& cam, int P, const float* means3D, const float* shs,
                                      const float* scales, const float* rots, const float* cov3Dp,
                                      const int32_t* radii, const uint32_t* clamped, const float2* sh_jac, const float* grad2d,
                                      float* dmeans2D, float* dmeans3D, float* dopac, float* dcolors, float* dshs,
                                      float* dscales, float* drots, float* dcov3D, const float* logit, const FusedAdam* adam, hipStream_t st);
hipError_t launch_tile_count(const Cam& cam, int P, GeomPtrs gp, uint32_t* tile_total, uint32_t* tile_base,
                             uint2* ranges, uint32_t* d_counts, uint32_t* host_counts, hipStream_t st);
hipError_t launch_tile_scatter_sort(const Cam& cam, int P, GeomPtrs gp, uint32_t* tile_base, const uint2* ranges,
                                    uint32_t max_tile_instances, unsigned long long* pairs, unsigned long long* pairs_alt,
                                    uint32_t* point_list, uint32_t cap, hipStream_t st);
extern std::atomic<int> g_half_quadrant_tiles;
extern std::atomic<int> g_chain_pieces;
extern std::atomic<int> g_chain_min_tiles;
extern std::atomic<int> g_chain_tickets;
extern std::atomic<int> g_chain_polls;
extern uint32_t* g_async_status_dev;
extern uint32_t* g_chain_fail_dev;      // device-memory twin of the status word (Cam::chain_fail); null until gs_async_status_word has been called
extern int g_chain_fail_device;         // the device it lives on
// the word for a launch on the CURRENT device: one process per GPU is this library's layout, but a launch on another device of the same process must
// not be handed a pointer into the first one's memory (it then runs unprotected, as before round 6)
inline uint32_t* chain_fail_word()
{
    if (!g_chain_fail_dev) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return dev == g_chain_fail_device ? g_chain_fail_dev : nullptr;
}
extern std::atomic<int> g_few_segments;
// images of few tiles (at most kFewTiles; the knob above can only lower the limit): the forward records every pixel's running state
// at the recorded list positions (cut_level below: every 256th up to 4096, then powers of two) for the segmented backward.  Planes of H*W floats: [level][T, C0, C1, C2, D], then the
// four totals, then one word "recorded"
constexpr int kFewTiles = 256;
// chained backward walks (images of more than kChainMinTiles tiles): pieces per quadrant, and what the image workspace holds for them behind
// `split_state`'s offset -- [quadrant][piece boundary][T, S][64 lanes] floats, then one flag word per (quadrant, piece boundary)
constexpr int kChainPieces = 3;
constexpr int kChainMinTiles = 768;                       // default threshold: 3072 resident walkers / 4 quadrants (gs_set_backward_chain lowers it for tests)
constexpr int kChainStateFloats = (kChainPieces - 1) * 2 * kWave;
constexpr int kChainTicketStride = 32;                    // words between the eight ticket counters: one 128-byte line each
constexpr int kChainTicketWords = 8 * kChainTicketStride; // ... behind the hand-over flags (zeroed with them by every forward)
inline size_t chain_flag_words(size_t tiles) { return tiles * 4 * (kChainPieces - 1) + kChainTicketWords; }
inline size_t chain_state_words(size_t tiles) { return tiles * 4 * kChainStateFloats + chain_flag_words(tiles); }
constexpr int kChainPollsDefault = 1 << 21;               // ~0.3 s of polling
// Recorded list positions (all multiples of the 64-record chunk): every kCutStep-th position up to kCutLinear (levels 0 .. kCutLinear /
// kCutStep - 1), then the powers of two up to kCutLast.  (Round 3 recorded 128 * 2^k only: a list of 1900 could be cut at 1024 and nowhere
// near a third or two thirds of it.)
constexpr int kCutStep = 256;
constexpr int kCutLinear = 4096;
constexpr int kCutLast = 131072;
constexpr int kCutLevels = kCutLinear / kCutStep + 5;          // 16 + {8192, 16384, 32768, 65536, 131072}
constexpr int kFewSegmentsMax = 3;                             // list segments (walkers) per quadrant in the few-tile backward
// level of a recorded position, or -1
__host__ __device__ inline int cut_level(uint32_t pos)
{
    if (pos >= (uint32_t)kCutStep && pos <= (uint32_t)kCutLinear) return pos % kCutStep == 0 ? (int)(pos / kCutStep) - 1 : -1;
    if (pos > (uint32_t)kCutLinear && pos <= (uint32_t)kCutLast && (pos & (pos - 1u)) == 0u) {
        int l = kCutLinear / kCutStep - 1;
        for (uint32_t q = kCutLinear; q < pos; q <<= 1) ++l;
        return l;
    }
    return -1;
}
// the recorded position nearest to `target` (absolute distance below kCutLinear, ratio above); 0 if the target is below half the first one
__host__ __device__ inline uint32_t cut_nearest(uint32_t target)
{
    if (target < (uint32_t)kCutStep / 2) return 0u;
    if (target <= (uint32_t)kCutLinear + kCutStep / 2) {
        uint32_t k = (target + kCutStep / 2) / kCutStep;
        if (k < 1u) k = 1u;
        if (k > (uint32_t)(kCutLinear / kCutStep)) k = kCutLinear / kCutStep;
        return k * kCutStep;
    }
    uint32_t pos = kCutLinear;
    // the next power of two is nearer (in ratio) once target >= pos * sqrt(2): 46341 / 65536 = 1 / sqrt 2
    // (round 4 shifted by 17: targets up to 2.83 pos stayed at pos, so that cuts above 4096 came out one level low -- results correct, walkers unbalanced)
    while (pos < (uint32_t)kCutLast && (unsigned long long)target * 46341ull >= ((unsigned long long)pos << 16)) pos <<= 1;
    return pos;
}
hipError_t launch_emit(const Cam& cam, int P, GeomPtrs gp, uint64_t* keys, uint32_t* vals, hipStream_t st);
hipError_t launch_ranges(int64_t D, const uint64_t* keys_sorted, uint2* ranges, hipStream_t st);
hipError_t launch_blend_forward(const Cam& cam, const uint2* ranges, const uint32_t* point_list, const float4* geom,
                                float* out_color, float* out_depth, float* out_opacity, float* final_T,
                                uint32_t* n_contrib, float* out_depth_sq, uint32_t cap, int segments, float* seg_T, float* split_state, uint32_t P, float* zero_fill,
                                hipStream_t st);
hipError_t launch_blend_backward(const Cam& cam, const uint2* ranges, const uint32_t* point_list, const float4* geom, const float* split_state,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 const float* dL_ddepth, float* grad2d, hipStream_t st);
hipError_t launch_adam_multi(int count, const GsAdamTensor* tensors, hipStream_t st);
hipError_t launch_rows(int mode, int count, const GsRowTensor* t, int64_t row0, int64_t n_valid, int64_t n_rows, const float* in, float* out,
                       hipStream_t st);
hipError_t launch_adam(int64_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2,
                       double eps, int step, hipStream_t st);

hipError_t launch_activate_forward(int P, int iso, const float* pose7, const float* means3D, const float* rots, const float* logit_op,
                                   const float* log_scales, float* o_means, float* o_rots, float* o_op, float* o_scales, hipStream_t st);
hipError_t launch_activate_backward(int P, int iso, const float* pose7, const float* rots, const float* o_op, const float* o_scales,
                                    const float* g_means, const float* g_rots, const float* g_op, const float* g_scales, float* d_means,
                                    float* d_rots, float* d_logit, float* d_logs, int accumulate, hipStream_t st);
constexpr int kLossAccSlots = 256;          // 64-byte accumulator lines at the head of the mapping loss' scratch (loss.hip)
hipError_t launch_mapping_loss(int W, int H, const float* im, const float* gt, const float* depth, const float* depth_sq,
                               const float* gt_depth, float w_im, float w_depth, float* losses, float* dL_dim,
                               float* dL_ddepth, float* scratch, int64_t persistent_call, hipStream_t st);
hipError_t launch_visibility_stats(int P, const int32_t* radii, uint8_t* seen, float* max_radius, hipStream_t st);
hipError_t launch_accumulate_grad2d(int P, const float* grad, const uint8_t* seen, float* accum, float* denom, hipStream_t st);
uint64_t grow_scratch_bytes(int64_t npix);
hipError_t launch_grow(int W, int H, const float* rd, const float* sil, const float* gt, const float* color, const float* k4,
                       const float* c2w12, float sil_thres, int isotropic, float* means3D, float* rgb, float* rot, float* logit,
                       float* log_scales, uint32_t* d_counts, void* scratch, hipStream_t st);
hipError_t launch_keyframe_overlap(int n_pts, const float* pts, int n_kf, const float* w2c, const float* k9, int W, int H, int edge,
                                   uint32_t* counts, hipStream_t st);
uint64_t compact_scratch_bytes(int64_t n);
hipError_t launch_compact_index(int64_t n, const uint8_t* keep, uint32_t* src_index, uint32_t* d_count, void* scratch, hipStream_t st);
hipError_t launch_gather_rows(int64_t n_out, int row_floats, const uint32_t* src_index, const float* src, float* dst, int64_t n_copy, hipStream_t st);
hipError_t launch_densify_classify(int N, int scale_dim, const float* log_scales, const float* logit_op, const float* accum,
                                   const float* denom, const float* scene_radius, float grad_thresh, float opacity_thresh,
                                   int remove_big, int n_split, uint8_t* keep_orig, uint8_t* keep_clone, uint8_t* keep_child,
                                   uint8_t* split_mask, hipStream_t st);
hipError_t launch_densify_children(int n_child, int scale_dim, int n_split, const float* rots, const float* samples, uint64_t seed,
                                   float* means3D, float* log_scales, hipStream_t st);
uint64_t compact3_scratch_bytes(int64_t n);
hipError_t launch_compact_index3(int64_t n, const uint8_t* ka, const uint8_t* kb, const uint8_t* kc, int n_rep, uint32_t* src_index,
                                 uint32_t* d_counts, void* scratch, hipStream_t st);

// sort backend (sort_radix.hip: hand-written LSD radix sort, 8 bits per pass): stable ascending sort of (key64, val32) pairs on bits [0,end_bit)
size_t sort_temp_bytes(int64_t D, int end_bit);
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, int64_t D, int end_bit, hipStream_t st);

constexpr int kWaveSlots = 256 * 4 * 5;   // CUs x SIMDs x the five wavefronts per SIMD the blend kernels are sized for
constexpr int kAdamMaxTensors = 8; // parameter tensors per multi-tensor Adam launch
constexpr int kGradStride = 16;   // floats per Gaussian in the 2-D gradient record (64 B = one cache line, so an
                                  // atomic flush of the 9 components is ONE memory-side read-modify-write):
                                  // raw moments, see blend.hip: 0..4 geometry, 5 opacity, 6..8 colour, 9 view depth
}  // namespace gs
