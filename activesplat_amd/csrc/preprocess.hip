// preprocess.hip -- per-Gaussian forward stage (+ tile-count block sums and their scan).
//
// Replaces the "preprocess" + "inclusive scan" work items of the reference's absent CUDA extension
// (contract: SURVEY App. A.2, call sites src/mapper/splatam/splatam.py:208,212).
// HBM-bound streaming kernel: one Gaussian per lane, array-of-struct inputs staged through LDS with
// 16-byte-per-lane coalesced loads, one 48-byte screen-space record written per Gaussian.
//
// Bit-level spec: this translation unit is compiled with -ffp-contract=off and every expression is
// parenthesised exactly as DESIGN.md section 3 states, so radii, tile rects, tile counts and depth-key bits
// are reproducible to the bit by the CPU oracle.
#include "gs_common.h"

#pragma clang fp contract(off)

namespace gs {

__device__ __forceinline__ int clamp_tile(float v, int hi)
{
    v = fminf(fmaxf(v, -1048576.0f), 1048576.0f);
    int i = (int)v;
    return min(hi, max(0, i));
}

// real-SH basis of the 3DGS family
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b)
{
    b[0] = 0.28209479177387814f;
    if (deg > 0) {
        b[1] = -0.4886025119029199f * y; b[2] = 0.4886025119029199f * z; b[3] = -0.4886025119029199f * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = 1.0925484305920792f * xy;
            b[5] = -1.0925484305920792f * yz;
            b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            b[7] = -1.0925484305920792f * xz;
            b[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                b[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
                b[10] = 2.890611442640554f * xy * z;
                b[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
                b[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
                b[14] = 1.445305721320277f * z * (xx - yy);
                b[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
            }
        }
    }
}

// The per-Gaussian geometry of one lane (bit-level spec): projection, EWA covariance, radius, tile rect, the 48-byte
// record.  Inputs come from the workgroup's LDS staging (means; scales + rotations or the covariance; colours for the no-SH case).
// Returns the number of tiles the Gaussian touches.
template <bool HAS_SH>
__device__ __forceinline__ uint32_t project_gaussian(const Cam& cam, int P, int view, int i, int io, int tid, const float* s_mean,
                                                     const float* s_scale, const float* s_rot, const float* s_cov, const float* s_col,
                                                     bool has_cov, float o, const float (&sh_rgb)[3], uint32_t sh_clamp,
                                                     int32_t* __restrict__ radii, const GeomPtrs& gp, int* radius_out = nullptr)
{
    uint32_t ntiles = 0;
    if (radius_out) *radius_out = 0;
    if (i < P) {
        const float* m = cam.view + 16 * view;
        const float* q = cam.proj + 16 * view;
        const float px = s_mean[tid * 3], py = s_mean[tid * 3 + 1], pz = s_mean[tid * 3 + 2];
        const float tx = ((m[0] * px + m[4] * py) + m[8] * pz) + m[12];
        const float ty = ((m[1] * px + m[5] * py) + m[9] * pz) + m[13];
        const float tz = ((m[2] * px + m[6] * py) + m[10] * pz) + m[14];
        int radius = 0;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
        uint2 rc = make_uint2(0u, 0u);
        uint32_t clampbits = 0;
        if (tz > 0.2f) {                              // near cull only -- no far plane
            const float hx = ((q[0] * px + q[4] * py) + q[8] * pz) + q[12];
            const float hy = ((q[1] * px + q[5] * py) + q[9] * pz) + q[13];
            const float hw = ((q[3] * px + q[7] * py) + q[11] * pz) + q[15];
            const float pw = 1.0f / (hw + 1e-7f);
            const float ndcx = hx * pw, ndcy = hy * pw;
            // EWA projection of the covariance: cov2D = T Sigma T^T + 0.3 I with T = J W (2 x 3)
            const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
            const float txtz = tx / tz, tytz = ty / tz;
            const float cx_ = fminf(limx, fmaxf(-limx, txtz)) * tz;
            const float cy_ = fminf(limy, fmaxf(-limy, tytz)) * tz;
            const float J00 = cam.fx / tz, J02 = -(cam.fx * cx_) / (tz * tz);
            const float J11 = cam.fy / tz, J12 = -(cam.fy * cy_) / (tz * tz);
            const float T00 = J00 * m[0] + J02 * m[2], T01 = J00 * m[4] + J02 * m[6], T02 = J00 * m[8] + J02 * m[10];
            const float T10 = J11 * m[1] + J12 * m[2], T11 = J11 * m[5] + J12 * m[6], T12 = J11 * m[9] + J12 * m[10];
            float k00, k01, k11, det;
            if (has_cov) {
                // a covariance handed in as six numbers: the quadratic form as it stands, det = k00 k11 - k01^2
                const float c0 = s_cov[tid * 6], c1 = s_cov[tid * 6 + 1], c2 = s_cov[tid * 6 + 2];
                const float c3 = s_cov[tid * 6 + 3], c4 = s_cov[tid * 6 + 4], c5 = s_cov[tid * 6 + 5];
                const float v00 = (c0 * T00 + c1 * T01) + c2 * T02;
                const float v01 = (c1 * T00 + c3 * T01) + c4 * T02;
                const float v02 = (c2 * T00 + c4 * T01) + c5 * T02;
                const float v10 = (c0 * T10 + c1 * T11) + c2 * T12;
                const float v11 = (c1 * T10 + c3 * T11) + c4 * T12;
                const float v12 = (c2 * T10 + c4 * T11) + c5 * T12;
                k00 = ((T00 * v00 + T01 * v01) + T02 * v02) + 0.3f;
                k01 = (T10 * v00 + T11 * v01) + T12 * v02;
                k11 = ((T10 * v10 + T11 * v11) + T12 * v12) + 0.3f;
                det = k00 * k11 - k01 * k01;
            } else {
                // Scale + rotation: Sigma = M M^T with M = R diag(s), so cov2D = A A^T + 0.3 I with A = T M (rows a1, a2), and (Lagrange)
                //   det = |a1 x a2|^2 + 0.3 (|a1|^2 + |a2|^2) + 0.09 :
                // a sum of non-negative terms.  (Rounds 1-5 formed Sigma, then T Sigma T^T, then k00 k11 - k01^2: for an elongated splat -- a 240 : 1
                // needle of radius 3973 px in front of the near plane, seed 180021 of the s = 1.2 sweep -- that determinant is 3e-5 of either
                // product, and the fp32 rounding of the ENTRIES alone put 2e-3 of relative error on the conic: forward images 1e-3 off, dL/dmeans2D
                // 1.1e-3 -- in the fp32 oracle just as in the kernels.  The cross product's error is eps / sin(theta) instead of eps / sin^2:
                // 1e-5 on that splat.  Same mathematics as the published T Sigma T^T; profiles/README.md, "The determinant without its cancellation".)
                const float sx = cam.mod * s_scale[tid * 3], sy = cam.mod * s_scale[tid * 3 + 1], sz = cam.mod * s_scale[tid * 3 + 2];
                const float4 rq = reinterpret_cast<const float4*>(s_rot)[tid];
                const float r = rq.x, x = rq.y, y = rq.z, z = rq.w;
                const float R00 = 1.0f - 2.0f * (y * y + z * z), R01 = 2.0f * (x * y - r * z), R02 = 2.0f * (x * z + r * y);
                const float R10 = 2.0f * (x * y + r * z), R11 = 1.0f - 2.0f * (x * x + z * z), R12 = 2.0f * (y * z - r * x);
                const float R20 = 2.0f * (x * z - r * y), R21 = 2.0f * (y * z + r * x), R22 = 1.0f - 2.0f * (x * x + y * y);
                const float M00 = R00 * sx, M01 = R01 * sy, M02 = R02 * sz;
                const float M10 = R10 * sx, M11 = R11 * sy, M12 = R12 * sz;
                const float M20 = R20 * sx, M21 = R21 * sy, M22 = R22 * sz;
                const float a10 = (T00 * M00 + T01 * M10) + T02 * M20, a11 = (T00 * M01 + T01 * M11) + T02 * M21, a12 = (T00 * M02 + T01 * M12) + T02 * M22;
                const float a20 = (T10 * M00 + T11 * M10) + T12 * M20, a21 = (T10 * M01 + T11 * M11) + T12 * M21, a22 = (T10 * M02 + T11 * M12) + T12 * M22;
                const float n1 = (a10 * a10 + a11 * a11) + a12 * a12, n2 = (a20 * a20 + a21 * a21) + a22 * a22;
                const float xc = a11 * a22 - a12 * a21, yc = a12 * a20 - a10 * a22, zc = a10 * a21 - a11 * a20;
                k00 = n1 + 0.3f;
                k01 = (a10 * a20 + a11 * a21) + a12 * a22;
                k11 = n2 + 0.3f;
                det = (((xc * xc + yc * yc) + zc * zc) + 0.3f * (n1 + n2)) + 0.09f;
            }
            if (det > 0.0f) {
                const float det_inv = 1.0f / det;
                const float mid = 0.5f * (k00 + k11);
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lam = fmaxf(mid + sq, mid - sq);
                float rf = ceilf(3.0f * sqrtf(lam));
                rf = fminf(rf, 16777216.0f);
                const float pxv = ((ndcx + 1.0f) * (float)cam.Wv - 1.0f) * 0.5f;     // pixel x inside this view
                const float pyy = ((ndcy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
                // the tile rect is clamped to the view's own columns, then moved to the view's slot of the atlas
                const int xs = view * cam.gxv;
                const int x0 = clamp_tile((pxv - rf) / 16.0f, cam.gxv) + xs, x1 = clamp_tile(((pxv + rf) + 15.0f) / 16.0f, cam.gxv) + xs;
                const float pxx = cam.V > 1 ? pxv + (float)(xs * kTile) : pxv;
                const int y0 = clamp_tile((pyy - rf) / 16.0f, cam.gy), y1 = clamp_tile(((pyy + rf) + 15.0f) / 16.0f, cam.gy);
                const int area = (x1 - x0) * (y1 - y0);
                float cr, cg, cb;
                if (HAS_SH) { cr = sh_rgb[0]; cg = sh_rgb[1]; cb = sh_rgb[2]; }
                else { cr = s_col[tid * 3]; cg = s_col[tid * 3 + 1]; cb = s_col[tid * 3 + 2]; }
                const float con_a = k11 * det_inv, con_b = -k01 * det_inv, con_c = k00 * det_inv;
                // NON-FINITE RULE: a Gaussian whose screen-space record -- pixel mean, conic, opacity, colour, depth -- holds a
                // NaN or an infinity is CULLED like one behind the camera (radius 0, no tile instance, zero gradients).  Whatever made it so (a NaN /
                // inf mean, scale, quaternion, opacity, colour or SH coefficient of an active degree: a diverged optimiser step) then cannot reach
                // a pixel -- the blend kernels weight a skipped record's colour with 0 instead of branching, and NaN x 0 would poison every pixel
                // of the tiles the record is staged for.  Finite but out-of-range values (opacity 2, negative scales) are taken as they are.
                const bool finite = __builtin_isfinite(pxx) && __builtin_isfinite(pyy) && __builtin_isfinite(con_a) && __builtin_isfinite(con_b) &&
                                    __builtin_isfinite(con_c) && __builtin_isfinite(o) && __builtin_isfinite(cr) && __builtin_isfinite(cg) &&
                                    __builtin_isfinite(cb) && __builtin_isfinite(tz);
                if (area > 0 && finite) {
                    radius = (int)rf;
                    ntiles = (uint32_t)area;
                    rc = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16));
                    if (HAS_SH) clampbits = sh_clamp;
                    // work-skipping extents: alpha >= 1/255 needs power >= -ln(255 o); the ellipse
                    // {d : d^T conic d <= 2 tau} has half-extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy).
                    // tau carries a 0.02 slack (>> any fp32 rounding of conic / exp); o*255 <= 1 -> never visible.
                    float ex = -1.0f, ey = -1.0f;
                    const float o255 = o * 255.0f;
                    if (o255 > 1.0f) {
                        const float tau2 = 2.0f * (__logf(o255) + 0.02f);
                        ex = sqrtf(tau2 * k00) + 0.01f;
                        ey = sqrtf(tau2 * k11) + 0.01f;
                    }
                    g0 = make_float4(pxx, pyy, con_a, con_b);
                    g1 = make_float4(con_c, o, cr, cg);
                    g2 = make_float4(cb, tz, ex, ey);
                }
            }
        }
        radii[io] = radius;
        if (radius_out) *radius_out = radius;
        // a culled Gaussian leaves radius 0 and tile count 0 behind and nothing else: its 48-byte record, rect and depth bits are only ever read
        // through a tile instance (binning reads rect / depth bits behind `tiles > 0`, the blend gathers records by the ids of the tile lists).  A
        // mapper's keyframe sees a fraction of the map: 60 of the 68 bytes a culled Gaussian used to write
        if (ntiles) {
            gp.geom[(size_t)io * 3] = g0; gp.geom[(size_t)io * 3 + 1] = g1; gp.geom[(size_t)io * 3 + 2] = g2;
            gp.rect[io] = rc;
            gp.depth_bits[io] = __float_as_uint(g2.y);
        }
        gp.tiles[io] = ntiles;
        if (HAS_SH && !gp.sh_jac) gp.clamped[io] = clampbits;     // (with a saved Jacobian the flags travel in its record)
    } else if (cam.V > 1) {                            // padding rows of a view's last block: never visible
        radii[io] = 0;
        gp.geom[(size_t)io * 3] = gp.geom[(size_t)io * 3 + 1] = gp.geom[(size_t)io * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        gp.rect[io] = make_uint2(0u, 0u);
        gp.tiles[io] = 0u;
        gp.depth_bits[io] = 0u;
        if (HAS_SH) gp.clamped[io] = 0u;
    }
    return ntiles;
}

// Raw-parameter mode (Cam::act): the staged rows are the mapper's parameters; every thread turns ITS rows into what the rasteriser takes --
// means into the camera frame, exp of the log scales (one per Gaussian when act_iso: staged [kBlock] wide), the activated rotation -- in place,
// and returns the opacity sigmoid(logit).  The same formulas as activate.hip (slam_helpers.py:252-304,124-139).
__device__ __forceinline__ float activate_staged_rows(const Cam& cam, int tid, bool row, float* s_mean, float* s_scale, float* s_rot, float logit)
{
    const float ls0 = s_scale[cam.act_iso ? tid : tid * 3];
    if (cam.act_iso) __syncthreads();                     // (every thread has its [kBlock]-staged value before the 3-wide rows are written)
    if (row) {
        float R[3][3];
        quat_to_rot(cam.act_q, R);
        const float px = s_mean[tid * 3], py = s_mean[tid * 3 + 1], pz = s_mean[tid * 3 + 2];
        s_mean[tid * 3] = R[0][0] * px + R[0][1] * py + R[0][2] * pz + cam.act_t[0];
        s_mean[tid * 3 + 1] = R[1][0] * px + R[1][1] * py + R[1][2] * pz + cam.act_t[1];
        s_mean[tid * 3 + 2] = R[2][0] * px + R[2][1] * py + R[2][2] * pz + cam.act_t[2];
        if (cam.act_iso) {
            const float e = __expf(ls0);
            s_scale[tid * 3] = e; s_scale[tid * 3 + 1] = e; s_scale[tid * 3 + 2] = e;
        } else {
            s_scale[tid * 3] = __expf(ls0); s_scale[tid * 3 + 1] = __expf(s_scale[tid * 3 + 1]); s_scale[tid * 3 + 2] = __expf(s_scale[tid * 3 + 2]);
        }
        const float4 q4 = reinterpret_cast<const float4*>(s_rot)[tid];
        const float q[4] = {q4.x, q4.y, q4.z, q4.w};
        float out[4];
        activate_rotation(cam.act_q, cam.act_iso, q, out);
        reinterpret_cast<float4*>(s_rot)[tid] = make_float4(out[0], out[1], out[2], out[3]);
    }
    return 1.0f / (1.0f + __expf(-logit));
}

// SH: 0 = colours given, 1 = coefficient rows of any width through per-wave LDS slabs (16-coefficient rows, the layout every caller of
// the reference uses, take preprocess_forward_sh48_kernel below)
template <int SH, bool ACT = false>        // ACT: raw-parameter mode (colours given or 16-coefficient rows only)
__global__ __launch_bounds__(kBlock) void preprocess_forward_kernel(
    Cam cam, int P, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ colors, const float* __restrict__ opac, const float* __restrict__ scales,
    const float* __restrict__ rots, const float* __restrict__ cov3Dp, int32_t* __restrict__ radii, GeomPtrs gp)
{
    // LDS: means | one pool that holds, in turn, the per-wave SH slabs (HAS_SH, first phase) and the staged
    // scale+rotation (or covariance) rows and colours (second phase)
    constexpr bool HAS_SH = SH != 0;
    constexpr bool SLAB = SH == 1;
    constexpr int kGeoFloats = kBlock * 7, kColFloats = SLAB ? 0 : kBlock * 3;
    constexpr int kSlabFloats = SLAB ? (kBlock / kWave) * kShHalf * kShPad : 0;
    constexpr int kPoolFloats = (kGeoFloats + kColFloats) > kSlabFloats ? (kGeoFloats + kColFloats) : kSlabFloats;
    __shared__ __attribute__((aligned(16))) float s_mean[kBlock * 3];
    __shared__ __attribute__((aligned(16))) float s_pool[kPoolFloats];
    __shared__ uint32_t s_wsum[kBlock / kWave];
    float* const s_scale = s_pool;                       // [kBlock*3]  (scales + rotations) ...
    float* const s_rot = s_pool + kBlock * 3;            // [kBlock*4]
    float* const s_cov = s_pool;                         // [kBlock*6]  ... or the precomputed covariance
    float* const s_col = s_pool + kGeoFloats;            // [kBlock*3]  colours (no-SH instantiation only)
    float* const s_sh = s_pool;

    const int tid = threadIdx.x;
    const int vbase = blockIdx.x * kBlock;                     // first VIRTUAL Gaussian of this block (= output row)
    const int view = cam.V > 1 ? (int)blockIdx.x / cam.nbv : 0;
    const int base = (cam.V > 1 ? (int)blockIdx.x - view * cam.nbv : (int)blockIdx.x) * kBlock;   // first INPUT row
    const int nrows = max(0, min(kBlock, P - base));
    // the per-tile instance counters of the binning stage are zeroed here (saves a memset launch)
    if (vbase + tid < cam.gx * cam.gy) gp.tile_total[vbase + tid] = 0u;
    const int i = base + tid, io = vbase + tid;
    float o_in = i < P ? opac[i] : 0.0f;                       // requested up front: not a dependent load inside the visible branch
    stage_rows<3>(s_mean, means3D, base, nrows, tid);
    if (!SLAB) {
        if (cov3Dp) stage_rows<6>(s_cov, cov3Dp, base, nrows, tid);
        else {
            if (ACT && cam.act_iso) stage_rows<1>(s_scale, scales, base, nrows, tid); else stage_rows<3>(s_scale, scales, base, nrows, tid);
            stage_rows<4>(s_rot, rots, base, nrows, tid);
        }
        if (SH == 0) stage_rows<3>(s_col, colors, base, nrows, tid);
    }
    __syncthreads();
    if (ACT) o_in = activate_staged_rows(cam, tid, tid < nrows, s_mean, s_scale, s_rot, o_in);

    // SH -> RGB for every Gaussian: each wavefront streams its own 64 coefficient rows through a private padded LDS slab,
    // 32 rows at a time (coalesced 16 B/lane global reads; lane = row reads at an odd stride are conflict-free)
    float sh_rgb[3] = {0.f, 0.f, 0.f};
    uint32_t sh_clamp = 0;
    if (SLAB) {
        const int M = cam.sh_coeffs, K = M * 3, nbasis = (cam.sh_degree + 1) * (cam.sh_degree + 1);
        const int stride = sh_row_stride(K);
        const int lane = tid & 63, wave = tid >> 6;
        float* slab = s_sh + wave * kShHalf * kShPad;
        for (int h = 0; h < kWave / kShHalf; h++) {
            const int row0 = base + wave * kWave + h * kShHalf;
            if (row0 >= P) break;                                  // wave-uniform
            __builtin_amdgcn_wave_barrier();
            sh_wave_rows_to_lds<0>(slab, shs, row0, min(kShHalf, P - row0), K, lane);
            __builtin_amdgcn_wave_barrier();
            if ((lane >> 5) == h && base + tid < P) {
                const float* cp = cam.campos + 3 * view;
                const float dx = s_mean[tid * 3] - cp[0], dy = s_mean[tid * 3 + 1] - cp[1], dz = s_mean[tid * 3 + 2] - cp[2];
                const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
                float b[16];
                sh_basis(cam.sh_degree, dx * inv, dy * inv, dz * inv, b);
                const float* sh = slab + (lane & 31) * stride;
                float acc[3] = {0.f, 0.f, 0.f};
                // (the colour is not part of the bit-level spec of this translation unit: fused multiply-adds here)
                for (int k = 0; k < nbasis; k++) {
                    acc[0] = fmaf(b[k], sh[3 * k], acc[0]); acc[1] = fmaf(b[k], sh[3 * k + 1], acc[1]); acc[2] = fmaf(b[k], sh[3 * k + 2], acc[2]);
                }
                float J[9];
                if (gp.sh_jac) sh_direction_jacobian(cam.sh_degree, dx * inv, dy * inv, dz * inv, sh, J);   // for the backward: it need not read the coefficient rows again
                acc[0] += 0.5f; acc[1] += 0.5f; acc[2] += 0.5f;
                sh_clamp = (acc[0] < 0.f ? 1u : 0u) | (acc[1] < 0.f ? 0x100u : 0u) | (acc[2] < 0.f ? 0x10000u : 0u);
                // (a select, not fmaxf: a NaN colour must stay NaN -- the non-finite rule of project_gaussian culls it -- where maxNum would turn it into 0)
                sh_rgb[0] = acc[0] < 0.f ? 0.f : acc[0]; sh_rgb[1] = acc[1] < 0.f ? 0.f : acc[1]; sh_rgb[2] = acc[2] < 0.f ? 0.f : acc[2];
                if (gp.sh_jac) store_sh_jac(gp.sh_jac, (size_t)io, J, sh_clamp);
            }
        }
    }

    if (SLAB) {                                          // the slabs are done: reuse the pool for the geometry rows
        __syncthreads();
        if (cov3Dp) stage_rows<6>(s_cov, cov3Dp, base, nrows, tid);
        else { stage_rows<3>(s_scale, scales, base, nrows, tid); stage_rows<4>(s_rot, rots, base, nrows, tid); }
        __syncthreads();
    }
    int radius = 0;
    const uint32_t ntiles = project_gaussian<HAS_SH>(cam, P, view, i, io, tid, s_mean, s_scale, s_rot, s_cov, s_col, cov3Dp != nullptr, o_in,
                                                     sh_rgb, sh_clamp, radii, gp, ACT ? &radius : nullptr);
    if (ACT && i < P) {                                   // the mapper's visibility statistics of this render (raw-parameter mode)
        if (gp.vis_max) gp.vis_max[i] = fmaxf(gp.vis_max[i], (float)radius);
        if (gp.vis_seen) gp.vis_seen[i] = radius > 0 ? 1 : 0;
    }
    // per-block tile count -> block_sums (scanned by scan_block_sums_kernel)
    const uint32_t ws = wave_sum_u32(ntiles);
    if ((tid & 63) == 0) s_wsum[tid >> 6] = ws;
    __syncthreads();
    if (tid == 0) gp.block_sums[blockIdx.x] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
}

// 16-coefficient SH rows: ONE workgroup barrier instead of the generic kernel's three and one memory round trip in front of it --
// the first half's coefficient loads, the opacity and the staging of means / scales / rotations are all issued before anything waits; the
// second half's loads are issued as soon as the first half sits in the slab and fly during its arithmetic (the same 24 registers: 128 VGPRs,
// four workgroups per CU instead of three); slabs with 16-byte aligned rows (gs_common.h: kShPad4), i.e. 128-bit LDS accesses.
// Measured at 2 M Gaussians against the generic design with compile-time row width (three barriers, both halves' loads held in 48
// registers: 167 VGPRs, three workgroups per CU, 49-float slab rows): 155 -> 131 us; same arithmetic, same results to the bit.
template <bool ACT>
__global__ __launch_bounds__(kBlock, 4) void preprocess_forward_sh48_kernel(
    Cam cam, int P, const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ opac,
    const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ cov3Dp, int32_t* __restrict__ radii, GeomPtrs gp)
{
    __shared__ __attribute__((aligned(16))) float s_mean[kBlock * 3];
    __shared__ __attribute__((aligned(16))) float s_geo[kBlock * 7];
    __shared__ __attribute__((aligned(16))) float s_sh[(kBlock / kWave) * kShHalf * kShPad4];
    __shared__ uint32_t s_wsum[kBlock / kWave];
    float* const s_scale = s_geo;
    float* const s_rot = s_geo + kBlock * 3;
    float* const s_cov = s_geo;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vbase = blockIdx.x * kBlock;
    const int view = cam.V > 1 ? (int)blockIdx.x / cam.nbv : 0;
    const int base = (cam.V > 1 ? (int)blockIdx.x - view * cam.nbv : (int)blockIdx.x) * kBlock;
    const int nrows = max(0, min(kBlock, P - base));
    if (vbase + tid < cam.gx * cam.gy) gp.tile_total[vbase + tid] = 0u;
    const int i = base + tid, io = vbase + tid;
    const int row_w = base + wave * kWave;
    const bool full = row_w + kWave <= P;
    float4 va[6];
    if (full) sh48_half_load(va, shs, row_w, lane);
    float o_in = i < P ? opac[i] : 0.0f;
    stage_rows<3>(s_mean, means3D, base, nrows, tid);
    if (cov3Dp) stage_rows<6>(s_cov, cov3Dp, base, nrows, tid);
    else {
        if (ACT && cam.act_iso) stage_rows<1>(s_scale, scales, base, nrows, tid); else stage_rows<3>(s_scale, scales, base, nrows, tid);
        stage_rows<4>(s_rot, rots, base, nrows, tid);
    }
    __syncthreads();
    if (ACT) o_in = activate_staged_rows(cam, tid, tid < nrows, s_mean, s_scale, s_rot, o_in);

    float sh_rgb[3] = {0.f, 0.f, 0.f};
    uint32_t sh_clamp = 0;
    float* slab = s_sh + wave * kShHalf * kShPad4;
    for (int h = 0; h < kWave / kShHalf; h++) {
        const int row0 = row_w + h * kShHalf;
        if (row0 >= P) break;                                  // wave-uniform
        __builtin_amdgcn_wave_barrier();
        if (full) {
            sh48_half_to_lds4(slab, va, lane);
            if (h == 0) sh48_half_load(va, shs, row_w + kShHalf, lane);
        } else {
            sh48_rows_to_lds4(slab, shs, row0, min(kShHalf, P - row0), lane);
        }
        __builtin_amdgcn_wave_barrier();
        if ((lane >> 5) == h && i < P) {
            const float* cp = cam.campos + 3 * view;
            const float dx = s_mean[tid * 3] - cp[0], dy = s_mean[tid * 3 + 1] - cp[1], dz = s_mean[tid * 3 + 2] - cp[2];
            const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
            float b[16];
            sh_basis(cam.sh_degree, dx * inv, dy * inv, dz * inv, b);
            float acc[3], J[9];
            sh48_color_and_jacobian(cam.sh_degree, dx * inv, dy * inv, dz * inv, b, slab + (lane & 31) * kShPad4, gp.sh_jac != nullptr, acc, J);
            acc[0] += 0.5f; acc[1] += 0.5f; acc[2] += 0.5f;
            sh_clamp = (acc[0] < 0.f ? 1u : 0u) | (acc[1] < 0.f ? 0x100u : 0u) | (acc[2] < 0.f ? 0x10000u : 0u);
            if (gp.sh_jac) store_sh_jac(gp.sh_jac, (size_t)io, J, sh_clamp);
            // (a select, not fmaxf: a NaN colour must stay NaN -- the non-finite rule of project_gaussian culls it -- where maxNum would turn it into 0)
                sh_rgb[0] = acc[0] < 0.f ? 0.f : acc[0]; sh_rgb[1] = acc[1] < 0.f ? 0.f : acc[1]; sh_rgb[2] = acc[2] < 0.f ? 0.f : acc[2];
        }
    }
    int radius = 0;
    const uint32_t ntiles = project_gaussian<true>(cam, P, view, i, io, tid, s_mean, s_scale, s_rot, s_cov, nullptr, cov3Dp != nullptr, o_in,
                                                   sh_rgb, sh_clamp, radii, gp, ACT ? &radius : nullptr);
    if (ACT && i < P) {
        if (gp.vis_max) gp.vis_max[i] = fmaxf(gp.vis_max[i], (float)radius);
        if (gp.vis_seen) gp.vis_seen[i] = radius > 0 ? 1 : 0;
    }
    const uint32_t ws = wave_sum_u32(ntiles);
    if (lane == 0) s_wsum[wave] = ws;
    __syncthreads();
    if (tid == 0) gp.block_sums[blockIdx.x] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
}

// Exclusive scan (in place) of the per-block tile counts; total -> block_sums[n] and *d_total.
__global__ __launch_bounds__(1024) void scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int n, uint32_t* __restrict__ d_total)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < n ? block_sums[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(v, lane);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint32_t wprefix = 0;
        for (int w = 0; w < wave; w++) wprefix += s_w[w];
        const uint32_t carry = s_carry;
        if (i < n) block_sums[i] = carry + wprefix + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wprefix + inc;
        __syncthreads();
    }
    if (tid == 0) { block_sums[n] = s_carry; *d_total = s_carry; }
}

hipError_t launch_preprocess_forward(const Cam& cam, int P, const float* means3D, const float* shs,
                                     const float* colors, const float* opac, const float* scales,
                                     const float* rots, const float* cov3Dp, int32_t* radii, GeomPtrs gp,
                                     uint32_t* d_num_rendered, hipStream_t st)
{
    const int nb = cam.V > 1 ? cam.V * cam.nbv : (P + kBlock - 1) / kBlock;
    if (cam.act && (cov3Dp || (shs && cam.sh_coeffs != 16))) return hipErrorInvalidValue;      // (api.hip refuses these before)
    if (nb > 0 && shs && cam.sh_coeffs == 16) {
        if (cam.act) hipLaunchKernelGGL(preprocess_forward_sh48_kernel<true>, dim3(nb), dim3(kBlock), 0, st, cam, P, means3D, shs, opac, scales, rots, cov3Dp, radii, gp);
        else hipLaunchKernelGGL(preprocess_forward_sh48_kernel<false>, dim3(nb), dim3(kBlock), 0, st, cam, P, means3D, shs, opac, scales, rots, cov3Dp, radii, gp);
    } else if (nb > 0 && shs)
        hipLaunchKernelGGL(preprocess_forward_kernel<1>, dim3(nb), dim3(kBlock), 0, st, cam, P, means3D, shs, colors, opac,
                           scales, rots, cov3Dp, radii, gp);
    else if (nb > 0 && cam.act)
        hipLaunchKernelGGL((preprocess_forward_kernel<0, true>), dim3(nb), dim3(kBlock), 0, st, cam, P, means3D, shs, colors, opac,
                           scales, rots, cov3Dp, radii, gp);
    else if (nb > 0)
        hipLaunchKernelGGL(preprocess_forward_kernel<0>, dim3(nb), dim3(kBlock), 0, st, cam, P, means3D, shs, colors, opac,
                           scales, rots, cov3Dp, radii, gp);
    // tile_total must be zero even when the grid above does not cover every tile (tiny P, many tiles)
    if ((size_t)nb * kBlock < (size_t)cam.gx * cam.gy) {
        hipError_t e = hipMemsetAsync(gp.tile_total, 0, (size_t)cam.gx * cam.gy * 4, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// per-Gaussian offsets (radix path only): exclusive scan of the per-block tile counts, total -> *d_total
hipError_t launch_scan_block_sums(int P, GeomPtrs gp, uint32_t* d_total, hipStream_t st)
{
    const int nb = (P + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, st, gp.block_sums, nb, d_total);
    return hipGetLastError();
}

}  // namespace gs
