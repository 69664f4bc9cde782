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
//   * 64 records at a time: every lane gathers one record and tests ITS alpha>=1/255 bounding box
//     against the wave's quadrant; one 64-bit __ballot gives the hit mask.  Hit records are staged into
//     the wave's private LDS slice (3 KiB) and the wave then walks only the set bits with scalar
//     find-first-set -- a wave-uniform loop, LDS broadcast reads, no divergence.  Skipped records can
//     never pass the alpha>=1/255 test, so the result is identical to evaluating all of them.
//   * the loop is software-pipelined: ids two chunks ahead, records one chunk ahead.
//   * blockIdx -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8, so XCD x is
//     given the contiguous band of tiles [x*ceil(T/8), (x+1)*ceil(T/8)) and neighbouring tiles (which
//     share Gaussians) reuse records in one 4 MiB L2.
#include "gs_common.h"

namespace gs {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTmin = 0.0001f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr uint32_t kNoId = 0xffffffffu;

struct TileCtx {
    int tile, tx, ty, px, py;
    float qx0, qy0, pxf, pyf;
    bool inside;
};

__device__ __forceinline__ bool tile_ctx(const Cam& cam, int wave, int lane, TileCtx& c)
{
    const int ntiles = cam.gx * cam.gy, per = (ntiles + 7) >> 3;
    c.tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);          // XCD-aware band mapping
    if ((int)(blockIdx.x >> 3) >= per || c.tile >= ntiles) return false;
    c.tx = c.tile % cam.gx; c.ty = c.tile / cam.gx;
    const int qx = c.tx * kTile + (wave & 1) * kQuad, qy = c.ty * kTile + (wave >> 1) * kQuad;
    c.px = qx + (lane & 7); c.py = qy + (lane >> 3);
    c.qx0 = (float)qx; c.qy0 = (float)qy; c.pxf = (float)c.px; c.pyf = (float)c.py;
    c.inside = c.px < cam.W && c.py < cam.H;
    return true;
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

__global__ __launch_bounds__(kBlock) void blend_forward_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ geom, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    __shared__ float4 s_rec[kBlock / kWave][3][kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx c;
    if (!tile_ctx(cam, wave, lane, c)) return;
    float4* s0 = s_rec[wave][0]; float4* s1 = s_rec[wave][1]; float4* s2 = s_rec[wave][2];
    const uint2 range = ranges[c.tile];
    const uint32_t n = range.y - range.x;
    const uint32_t* list = point_list + range.x;

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = !c.inside;

    if (!__all(done)) {
        // pipeline prologue: ids of chunks 0 and 1, records of chunk 0
        uint32_t id_next = (uint32_t)lane < n ? list[lane] : kNoId;
        uint32_t id_next2 = (uint32_t)lane + 64u < n ? list[lane + 64] : kNoId;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = make_float4(0.f, 0.f, -1.f, -1.f);
        if (id_next != kNoId) { r0 = geom[(size_t)id_next * 3]; r1 = geom[(size_t)id_next * 3 + 1]; r2 = geom[(size_t)id_next * 3 + 2]; }
        for (uint32_t base = 0; base < n; base += kWave) {
            const float4 q0 = r0, q1 = r1, q2 = r2;
            const uint32_t id_cur = id_next;
            // issue the next chunk's record gather and the ids two chunks ahead
            id_next = id_next2;
            id_next2 = base + 128u + (uint32_t)lane < n ? list[base + 128u + lane] : kNoId;
            r2 = make_float4(0.f, 0.f, -1.f, -1.f);
            if (id_next != kNoId) { r0 = geom[(size_t)id_next * 3]; r1 = geom[(size_t)id_next * 3 + 1]; r2 = geom[(size_t)id_next * 3 + 2]; }

            const bool hit = id_cur != kNoId && quadrant_hit(q0, q2, c.qx0, c.qy0);
            unsigned long long m = __ballot(hit);
            if (m == 0ull) continue;
            stage_record(s0, s1, s2, lane, q0, q1, q2, id_cur);
            __builtin_amdgcn_wave_barrier();
            // two hit records per LDS wait; the blend itself is branch-free (predicated weights)
            while (m) {
                const int j1 = __ffsll(m) - 1;
                m &= m - 1;
                const bool two = m != 0ull;
                const int j2 = two ? __ffsll(m) - 1 : j1;
                m &= m - 1;
                const float4 a0 = s0[j1], a1 = s1[j1], a2 = s2[j1];
                const float4 b0 = s0[j2], b1 = s1[j2], b2 = s2[j2];
                {
                    const float dx = a0.x - c.pxf, dy = a0.y - c.pyf;
                    const float p = (a0.z * dx + a0.w * dy) * dx + (a1.x * dy) * dy;
                    const float alpha = fminf(0.99f, a1.y * __builtin_amdgcn_exp2f(p));
                    const float test_T = T * (1.0f - alpha);
                    const bool vis = !done && p <= 0.0f && alpha >= kAlphaMin;
                    const bool ok = vis && test_T >= kTmin;
                    done = done || (vis && !ok);
                    const float w = ok ? alpha * T : 0.0f;
                    C0 += a1.z * w; C1 += a1.w * w; C2 += a2.x * w; Dp += a2.y * w;
                    T = ok ? test_T : T;
                    last = ok ? base + (uint32_t)j1 + 1u : last;
                }
                {
                    const float dx = b0.x - c.pxf, dy = b0.y - c.pyf;
                    const float p = (b0.z * dx + b0.w * dy) * dx + (b1.x * dy) * dy;
                    const float alpha = fminf(0.99f, b1.y * __builtin_amdgcn_exp2f(p));
                    const float test_T = T * (1.0f - alpha);
                    const bool vis = two && !done && p <= 0.0f && alpha >= kAlphaMin;
                    const bool ok = vis && test_T >= kTmin;
                    done = done || (vis && !ok);
                    const float w = ok ? alpha * T : 0.0f;
                    C0 += b1.z * w; C1 += b1.w * w; C2 += b2.x * w; Dp += b2.y * w;
                    T = ok ? test_T : T;
                    last = ok ? base + (uint32_t)j2 + 1u : last;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (__all(done)) break;
        }
    }
    if (c.inside) {
        const size_t pix = (size_t)c.py * cam.W + c.px, HW = (size_t)cam.H * cam.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * cam.bg[0];
        out_color[HW + pix] = C1 + T * cam.bg[1];
        out_color[2 * HW + pix] = C2 + T * cam.bg[2];
        out_depth[pix] = Dp;
        out_opacity[pix] = 1.0f - T;
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward: back-to-front replay, same independent-quadrant walk as the forward.  Per (wavefront, record)
// the nine per-pixel partials are reduced across the 64 lanes and accumulated with one hardware fp32
// atomic instruction (9 lanes, one component each) into the 48-byte per-Gaussian gradient record
// (device-scope atomics: correct across the 8 XCDs' private L2s).
//
// Wave-64 reduction of 9 values in ~32 VALU instead of 9 x 6 shuffle+add: a TRANSPOSED butterfly.
//   level 32: v_permlane32_swap pairs two values -> one register whose halves hold one value each
//   level 16: v_permlane16_swap pairs two such registers -> one register whose 4 rows hold 4 values
//   in-row  : 4 DPP adds (quad_perm xor1, xor2, row_half_mirror, row_mirror) finish 4 values at once
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_add(float x, const int ctrl_tag)
{
    // ctrl must be a literal: dispatch on the four controls used
    if (ctrl_tag == 0) return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
    if (ctrl_tag == 1) return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x4E, 0xf, 0xf, true));   // quad_perm:[2,3,0,1]
    if (ctrl_tag == 2) return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x140, 0xf, 0xf, true));                     // row_mirror
}
__device__ __forceinline__ float row_sum16(float x)
{
    x = dpp_add(x, 0); x = dpp_add(x, 1); x = dpp_add(x, 2); x = dpp_add(x, 3);
    return x;     // every lane of a 16-lane row holds the row total
}
__device__ __forceinline__ float fold32(float a, float b)     // lanes 0-31: a[l]+a[l+32], lanes 32-63: b[l-32]+b[l]
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold16(float a, float b)     // rows (0,2): a.r+a.(r+1), rows (1,3): b.(r-1)+b.r
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// In: 9 per-lane partials.  Out: the wave total of component c in the lane `red9_lane(c)`;
// returns this lane's value and writes the component it carries (or -1) to comp.
__device__ __forceinline__ float wave_reduce9(const float (&v)[9], int lane, int& comp)
{
    const float q0123 = row_sum16(fold16(fold32(v[0], v[1]), fold32(v[2], v[3])));   // rows: V0, V2, V1, V3
    const float q4567 = row_sum16(fold16(fold32(v[4], v[5]), fold32(v[6], v[7])));   // rows: V4, V6, V5, V7
    const float h8 = fold32(v[8], v[8]);
    const float q8 = row_sum16(fold16(h8, h8));                                       // every lane: V8
    const int row = lane >> 4, c = lane & 15;
    const int rowcomp = ((row & 1) << 1) | (row >> 1);                                // 0,2,1,3
    comp = c == 0 ? rowcomp : (c == 1 ? 4 + rowcomp : (lane == 2 ? 8 : -1));
    return c == 0 ? q0123 : (c == 1 ? q4567 : q8);
}


// Per-Gaussian 2-D gradient record accumulated by the backward blend (raw moments; the conic algebra is
// finished per Gaussian in preprocess_bwd.hip):  with Z = G dL/dG, d = mean - pixel
//   0: sum Z dx   1: sum Z dy   2: sum Z dx dx   3: sum Z dx dy   4: sum Z dy dy   5: sum G dL/dalpha   6..8: sum w dL/dC
__global__ __launch_bounds__(kBlock) void blend_backward_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ geom, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, float* __restrict__ grad2d)
{
    __shared__ float4 s_rec[kBlock / kWave][3][kWave];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx c;
    if (!tile_ctx(cam, wave, lane, c)) return;
    float4* s0 = s_rec[wave][0]; float4* s1 = s_rec[wave][1]; float4* s2 = s_rec[wave][2];
    const uint2 range = ranges[c.tile];
    const uint32_t* list = point_list + range.x;
    const size_t pix = (size_t)c.py * cam.W + c.px, HW = (size_t)cam.H * cam.W;

    const float Tf = c.inside ? final_T[pix] : 0.f;
    const uint32_t last = c.inside ? n_contrib[pix] : 0u;
    const float d0 = c.inside ? dL_dcolor[pix] : 0.f, d1 = c.inside ? dL_dcolor[HW + pix] : 0.f,
                d2 = c.inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float bgdot = cam.bg[0] * d0 + cam.bg[1] * d1 + cam.bg[2] * d2;
    float T = Tf, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;

    // the deepest contributor of any pixel of this quadrant bounds the replay
    uint32_t wmax = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor(wmax, m));
    if (wmax == 0) return;

    const int cmax = (int)((wmax - 1) / kWave);
    // pipeline prologue (walking chunks downwards): ids of chunks cmax and cmax-1, records of chunk cmax
    uint32_t id_next = (uint32_t)cmax * kWave + lane < wmax ? list[cmax * kWave + lane] : kNoId;
    uint32_t id_next2 = cmax >= 1 ? list[(cmax - 1) * kWave + lane] : kNoId;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = make_float4(0.f, 0.f, -1.f, -1.f);
    if (id_next != kNoId) { r0 = geom[(size_t)id_next * 3]; r1 = geom[(size_t)id_next * 3 + 1]; r2 = geom[(size_t)id_next * 3 + 2]; }
    for (int ch = cmax; ch >= 0; ch--) {
        const float4 q0 = r0, q1 = r1, q2 = r2;
        const uint32_t id_cur = id_next;
        id_next = id_next2;
        id_next2 = ch >= 2 ? list[(ch - 2) * kWave + lane] : kNoId;
        r2 = make_float4(0.f, 0.f, -1.f, -1.f);
        if (id_next != kNoId) { r0 = geom[(size_t)id_next * 3]; r1 = geom[(size_t)id_next * 3 + 1]; r2 = geom[(size_t)id_next * 3 + 2]; }

        const bool hit = id_cur != kNoId && quadrant_hit(q0, q2, c.qx0, c.qy0);
        unsigned long long m = __ballot(hit);
        if (m == 0ull) continue;
        stage_record(s0, s1, s2, lane, q0, q1, q2, id_cur);
        __builtin_amdgcn_wave_barrier();
        while (m) {
            const int j = 63 - __clzll(m);
            m &= ~(1ull << j);
            const uint32_t pos = (uint32_t)ch * kWave + (uint32_t)j;          // 0-based position in the tile list
            const float4 a0 = s0[j], a1 = s1[j], a2 = s2[j];
            const float dx = a0.x - c.pxf, dy = a0.y - c.pyf;
            const float p = (a0.z * dx + a0.w * dy) * dx + (a1.x * dy) * dy;
            const float G = __builtin_amdgcn_exp2f(p);
            const float alpha = fminf(0.99f, a1.y * G);
            const bool ok = pos < last && p <= 0.0f && alpha >= kAlphaMin;
            if (!__any(ok)) continue;
            float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float rcp = __builtin_amdgcn_rcpf(1.0f - alpha);
                T = T * rcp;
                const float w = alpha * T;
                acc0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0;
                acc1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1;
                acc2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2;
                lc0 = a1.z; lc1 = a1.w; lc2 = a2.x;
                float dL_dalpha = ((lc0 - acc0) * d0 + (lc1 - acc1) * d1 + (lc2 - acc2) * d2) * T;
                v[6] = w * d0; v[7] = w * d1; v[8] = w * d2;
                last_alpha = alpha;
                dL_dalpha -= Tf * rcp * bgdot;
                const float GdA = G * dL_dalpha;
                const float Z = a1.y * GdA;                 // G dL/dG
                const float zx = Z * dx, zy = Z * dy;
                v[0] = zx; v[1] = zy; v[2] = zx * dx; v[3] = zx * dy; v[4] = zy * dy; v[5] = GdA;
            }
            int comp;
            const float x = wave_reduce9(v, lane, comp);
            if (comp >= 0) atomicAdd(grad2d + (size_t)__float_as_uint(a2.z) * kGradStride + comp, x);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_blend_forward(const Cam& cam, const uint2* ranges, const uint32_t* point_list, const float4* geom,
                                float* out_color, float* out_depth, float* out_opacity, float* final_T,
                                uint32_t* n_contrib, hipStream_t st)
{
    const int nb = ((cam.gx * cam.gy + 7) >> 3) << 3;
    hipLaunchKernelGGL(blend_forward_kernel, dim3(nb), dim3(kBlock), 0, st, cam, ranges, point_list, geom,
                       out_color, out_depth, out_opacity, final_T, n_contrib);
    return hipGetLastError();
}

hipError_t launch_blend_backward(const Cam& cam, const uint2* ranges, const uint32_t* point_list, const float4* geom,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 float* grad2d, hipStream_t st)
{
    const int nb = ((cam.gx * cam.gy + 7) >> 3) << 3;
    hipLaunchKernelGGL(blend_backward_kernel, dim3(nb), dim3(kBlock), 0, st, cam, ranges, point_list, geom,
                       final_T, n_contrib, dL_dcolor, grad2d);
    return hipGetLastError();
}

}  // namespace gs
