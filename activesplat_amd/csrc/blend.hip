// blend.hip -- per-tile front-to-back alpha blend (forward) and back-to-front gradient replay (backward).
//
// Replaces the "render" forward/backward work items of the reference's absent CUDA extension
// (contract SURVEY App. A.2; consumers src/mapper/splatam/splatam.py:208-212,430-431).
//
// MI355X design (wave-64 first, not a 32-wide warp tiling):
//   * one 16x16 tile per 256-thread workgroup; each of the 4 wavefronts owns one 8x8 pixel QUADRANT
//     (lane = pixel), so a wavefront is the unit of work skipping;
//   * the tile's depth-sorted instance list is staged through LDS 256 records at a time (one 48-byte
//     record gather per lane);
//   * while staging, every lane tests ITS record's alpha>=1/255 bounding box against the four quadrants;
//     64-bit __ballot masks (one per producer wave x consumer quadrant) go to LDS, and each wavefront
//     then walks only the set bits of its own masks with scalar find-first-set -- a wave-uniform loop,
//     no divergence, LDS broadcast reads.  Skipped records can never pass the alpha>=1/255 test, so the
//     result is identical to evaluating all of them.
#include <stdlib.h>

#include "gs_common.h"

namespace gs {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTmin = 0.0001f;

// bit q set <=> the record's alpha-visible box overlaps quadrant q of the tile at pixel origin (ox,oy)
__device__ __forceinline__ uint32_t quadrant_bits(const float4& q0, const float4& q2, float ox, float oy)
{
    const float ex = q2.z, ey = q2.w;
    if (!(ex >= 0.0f)) return 0u;
    const float xlo = q0.x - ex, xhi = q0.x + ex, ylo = q0.y - ey, yhi = q0.y + ey;
    const bool cx0 = (xhi >= ox) && (xlo <= ox + 7.0f);
    const bool cx1 = (xhi >= ox + 8.0f) && (xlo <= ox + 15.0f);
    const bool cy0 = (yhi >= oy) && (ylo <= oy + 7.0f);
    const bool cy1 = (yhi >= oy + 8.0f) && (ylo <= oy + 15.0f);
    return (cx0 && cy0 ? 1u : 0u) | (cx1 && cy0 ? 2u : 0u) | (cx0 && cy1 ? 4u : 0u) | (cx1 && cy1 ? 8u : 0u);
}

__global__ __launch_bounds__(kBlock) void blend_forward_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ geom, float* __restrict__ out_color, float* __restrict__ out_depth,
    float* __restrict__ out_opacity, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
{
    __shared__ float4 s_q0[kBlock];
    __shared__ float4 s_q1[kBlock];
    __shared__ float4 s_q2[kBlock];
    __shared__ unsigned long long s_mask[4][4];      // [consumer quadrant][producer wave]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const float ox = (float)(tx * kTile), oy = (float)(ty * kTile);
    const int px = tx * kTile + (wave & 1) * kQuad + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * kQuad + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];

    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t base = range.x; base < range.y; base += kBlock) {
        if (__syncthreads_and(done)) break;          // also fences the previous batch's LDS reads
        const uint32_t idx = base + tid;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        uint32_t bits = 0;
        if (idx < range.y) {
            const uint32_t g = point_list[idx];
            q0 = geom[(size_t)g * 3]; q1 = geom[(size_t)g * 3 + 1]; q2 = geom[(size_t)g * 3 + 2];
            bits = quadrant_bits(q0, q2, ox, oy);
        }
        s_q0[tid] = q0; s_q1[tid] = q1; s_q2[tid] = q2;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long m = __ballot((bits >> q) & 1u);
            if (lane == 0) s_mask[q][wave] = m;
        }
        __syncthreads();
        if (__all(done)) continue;                    // this quadrant is finished; keep staging for the others
        for (int s = 0; s < 4; s++) {
            const unsigned long long mv = s_mask[wave][s];
            uint32_t mlo = __builtin_amdgcn_readfirstlane((uint32_t)mv);
            uint32_t mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mv >> 32));
            unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
            while (m) {
                const int j = s * kWave + (__ffsll(m) - 1);
                m &= m - 1;
                const float4 a0 = s_q0[j], a1 = s_q1[j], a2 = s_q2[j];
                const float dx = a0.x - pxf, dy = a0.y - pyf;
                const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                const float alpha = fminf(0.99f, a1.y * __expf(power));
                bool ok = !done && power <= 0.0f && alpha >= kAlphaMin;
                const float test_T = T * (1.0f - alpha);
                if (ok && test_T < kTmin) { done = true; ok = false; }
                if (ok) {
                    const float w = alpha * T;
                    C0 += a1.z * w; C1 += a1.w * w; C2 += a2.x * w; Dp += a2.y * w;
                    T = test_T;
                    last = (base - range.x) + (uint32_t)j + 1u;
                }
            }
        }
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
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward: back-to-front replay.  Per (wavefront, record) the nine per-pixel partials are reduced
// across the 64 lanes and accumulated with one hardware fp32 atomic instruction (9 lanes, one
// component each) into the 48-byte per-Gaussian gradient record (device-scope atomics: correct across
// the 8 XCDs' private L2s).
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

template <int VARIANT>   // 0 = shipped (transposed reduce + atomics); 1..3 = ablations (see launch_blend_backward)
__global__ __launch_bounds__(kBlock) void blend_backward_kernel(
    Cam cam, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ geom, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, float* __restrict__ grad2d)
{
    __shared__ float4 s_q0[kBlock];
    __shared__ float4 s_q1[kBlock];
    __shared__ float4 s_q2[kBlock];
    __shared__ uint32_t s_id[kBlock];
    __shared__ unsigned long long s_mask[4][4];
    __shared__ uint32_t s_wmax[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int tx = tile % cam.gx, ty = tile / cam.gx;
    const float ox = (float)(tx * kTile), oy = (float)(ty * kTile);
    const int px = tx * kTile + (wave & 1) * kQuad + (lane & 7);
    const int py = ty * kTile + (wave >> 1) * kQuad + (lane >> 3);
    const bool inside = px < cam.W && py < cam.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = ranges[tile];
    const size_t pix = (size_t)py * cam.W + px, HW = (size_t)cam.H * cam.W;

    const float Tf = inside ? final_T[pix] : 0.f;
    const uint32_t last = inside ? n_contrib[pix] : 0u;
    const float d0 = inside ? dL_dcolor[pix] : 0.f, d1 = inside ? dL_dcolor[HW + pix] : 0.f,
                d2 = inside ? dL_dcolor[2 * HW + pix] : 0.f;
    const float bgdot = cam.bg[0] * d0 + cam.bg[1] * d1 + cam.bg[2] * d2;
    float T = Tf, acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;

    // the deepest contributor of any pixel of this tile bounds the replay
    uint32_t wm = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wm = max(wm, (uint32_t)__shfl_xor(wm, m));
    if (lane == 0) s_wmax[wave] = wm;
    __syncthreads();
    const uint32_t kmax = max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3]));
    if (kmax == 0) return;
    const uint32_t wave_max = s_wmax[wave];

    for (int b = (int)((kmax - 1) / kBlock); b >= 0; b--) {
        __syncthreads();                                // previous batch fully consumed
        const uint32_t pos = (uint32_t)b * kBlock + tid;
        const uint32_t idx = range.x + pos;
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
        uint32_t bits = 0, g = 0;
        if (pos < kmax && idx < range.y) {
            g = point_list[idx];
            q0 = geom[(size_t)g * 3]; q1 = geom[(size_t)g * 3 + 1]; q2 = geom[(size_t)g * 3 + 2];
            bits = quadrant_bits(q0, q2, ox, oy);
        }
        s_q0[tid] = q0; s_q1[tid] = q1; s_q2[tid] = q2; s_id[tid] = g;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long m = __ballot((bits >> q) & 1u);
            if (lane == 0) s_mask[q][wave] = m;
        }
        __syncthreads();
        if ((uint32_t)b * kBlock >= wave_max) continue;   // nothing of this batch reaches this quadrant
        for (int s = 3; s >= 0; s--) {
            const unsigned long long mv = s_mask[wave][s];
            uint32_t mlo = __builtin_amdgcn_readfirstlane((uint32_t)mv);
            uint32_t mhi = __builtin_amdgcn_readfirstlane((uint32_t)(mv >> 32));
            unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
            while (m) {
                const int bit = 63 - __clzll(m);
                m &= ~(1ull << bit);
                const int j = s * kWave + bit;
                const uint32_t p = (uint32_t)b * kBlock + (uint32_t)j;        // 0-based position in the tile list
                const float4 a0 = s_q0[j], a1 = s_q1[j], a2 = s_q2[j];
                const float dx = a0.x - pxf, dy = a0.y - pyf;
                const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(0.99f, a1.y * G);
                const bool ok = p < last && power <= 0.0f && alpha >= kAlphaMin;
                if (!__any(ok)) continue;
                float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // gx gy ga gb gc go gr gg gbl
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
                    const float dL_dG = a1.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    v[0] = dL_dG * (-gdx * a0.z - gdy * a0.w);
                    v[1] = dL_dG * (-gdy * a1.x - gdx * a0.w);
                    v[2] = -0.5f * gdx * dx * dL_dG;
                    v[3] = -gdx * dy * dL_dG;
                    v[4] = -0.5f * gdy * dy * dL_dG;
                    v[5] = G * dL_dalpha;
                }
                if (VARIANT == 3) {            // ablation: no reduction, no atomics (keep the partials live)
#pragma unroll
                    for (int k = 0; k < 9; k++) asm volatile("" ::"v"(v[k]));
                    continue;
                }
                if (VARIANT == 1) {            // ablation: round-1 reduction (9 x 6 shuffles)
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = wave_sum(v[k]);
                    if (lane < 9) {
                        float x = v[0];
#pragma unroll
                        for (int k = 1; k < 9; k++) x = lane == k ? v[k] : x;
                        atomicAdd(grad2d + (size_t)s_id[j] * kGradStride + lane, x);
                    }
                    continue;
                }
                int comp;
                const float x = wave_reduce9(v, lane, comp);
                if (VARIANT == 2) { asm volatile("" ::"v"(x)); continue; }   // ablation: reduce, no atomics
                if (comp >= 0) atomicAdd(grad2d + (size_t)s_id[j] * kGradStride + comp, x);
            }
        }
    }
}

hipError_t launch_blend_forward(const Cam& cam, const uint2* ranges, const uint32_t* point_list, const float4* geom,
                                float* out_color, float* out_depth, float* out_opacity, float* final_T,
                                uint32_t* n_contrib, hipStream_t st)
{
    hipLaunchKernelGGL(blend_forward_kernel, dim3(cam.gx * cam.gy), dim3(kBlock), 0, st, cam, ranges, point_list, geom,
                       out_color, out_depth, out_opacity, final_T, n_contrib);
    return hipGetLastError();
}

hipError_t launch_blend_backward(const Cam& cam, const uint2* ranges, const uint32_t* point_list, const float4* geom,
                                 const float* final_T, const uint32_t* n_contrib, const float* dL_dcolor,
                                 float* grad2d, hipStream_t st)
{
    // GS_BWD_VARIANT (development only): 1 = shuffle reduction, 2 = no atomics, 3 = no reduction/atomics
    const char* ev = getenv("GS_BWD_VARIANT");
    const int variant = ev ? atoi(ev) : 0;
    const dim3 grid(cam.gx * cam.gy), block(kBlock);
    if (variant == 1) hipLaunchKernelGGL(blend_backward_kernel<1>, grid, block, 0, st, cam, ranges, point_list, geom, final_T, n_contrib, dL_dcolor, grad2d);
    else if (variant == 2) hipLaunchKernelGGL(blend_backward_kernel<2>, grid, block, 0, st, cam, ranges, point_list, geom, final_T, n_contrib, dL_dcolor, grad2d);
    else if (variant == 3) hipLaunchKernelGGL(blend_backward_kernel<3>, grid, block, 0, st, cam, ranges, point_list, geom, final_T, n_contrib, dL_dcolor, grad2d);
    else hipLaunchKernelGGL(blend_backward_kernel<0>, grid, block, 0, st, cam, ranges, point_list, geom, final_T, n_contrib, dL_dcolor, grad2d);
    return hipGetLastError();
}

}  // namespace gs
