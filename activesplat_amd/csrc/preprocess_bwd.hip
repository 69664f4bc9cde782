// preprocess_bwd.hip -- per-Gaussian backward stage: 2-D gradient record (pixel mean, conic, opacity,
// colour) -> gradients of means3D / scales / rotations / cov3D / opacities / colours / SH / means2D.
//
// Replaces the "bwd preprocess" work item of the reference's absent CUDA extension.
// Contract: SURVEY App. A.2 -- clamped tx/tz, ty/tz are constants for the gradient; dL/dmeans2D is
// reported in NDC units (pixel gradient x 0.5W, 0.5H), which is what the reference's densifier
// thresholds (src/mapper/splatam/utils/slam_external.py:100-108).
// HBM-bound streaming kernel, one Gaussian per lane.
#include "gs_common.h"

namespace gs {

// SH: 0 = colours given, 1 = coefficient rows of any width through per-wave LDS slabs, 3 = the same for 16-coefficient rows
// (row width 48 known at compile time).  (Slabs with 16-byte aligned rows and 128-bit LDS
// accesses, as the forward uses them, measured slower here: the wider stores' registers spill next to the hoisted loads.)
// ACT: raw-parameter mode (Cam::act; colours given or 16-coefficient rows): means3D / scales / rots / logit are the mapper's PARAMETERS, the
// frame transform + activations are recomputed here and the four gradients leave w.r.t. the parameters (activate.hip's backward inline);
// with Cam::act_accumulate they -- and dcolors -- are ADDED to the output buffers (rows of Gaussians that were not rendered stay untouched).
// ADAM (with ACT, never with accumulation): the optimiser step rides along -- every Gaussian's thread applies Adam (adam_elem: the arithmetic of
// adam.hip, contraction-free, so the result is the separate step's to the bit) to its parameters and moments in place with the gradient it has
// just formed (zero for a Gaussian that was not rendered: its moments still decay and its parameter still moves); the SH gradient rows go
// from the slab straight into a coalesced p / m / v update instead of out to HBM and back (2 x 192 B per Gaussian and step less).  The
// parameter gradients are NOT written; dmeans2D is.
// workgroups per CU the register budget is sized for: 4 (128 VGPRs); the raw-parameter SH variants -- which carry the activation algebra and,
// with ADAM, the update's operands on top of the slab traffic -- are the tightest: 32 bytes of scratch per lane at 4, none at 3 (A/B knob)
#ifndef GS_PBWD_RAW_SH_WGS
#define GS_PBWD_RAW_SH_WGS 4
#endif
// The parameter inputs are `__restrict__` only where nothing writes them: in the ADAM instantiation they ALIAS ad.p[0..4], which the same kernel
// updates in place (every read of an element precedes its write through a data dependence, but a no-alias promise there would be a false one).
template <bool NOALIAS> struct ParamPtr { typedef const float* __restrict__ type; };
template <> struct ParamPtr<false> { typedef const float* type; };

template <int SH, bool ACT = false, bool ADAM = false>
__global__ __launch_bounds__(kBlock, (ACT && SH == 3) ? GS_PBWD_RAW_SH_WGS : 4) void preprocess_backward_kernel(
    Cam cam, int P, typename ParamPtr<!ADAM>::type means3D, typename ParamPtr<!ADAM>::type shs,
    typename ParamPtr<!ADAM>::type scales, typename ParamPtr<!ADAM>::type rots, const float* __restrict__ cov3Dp,
    const int32_t* __restrict__ radii, const uint32_t* __restrict__ clamped, const float2* __restrict__ sh_jac,
    const float* __restrict__ grad2d, float* __restrict__ dmeans2D, float* __restrict__ dmeans3D, float* __restrict__ dopac,
    float* __restrict__ dcolors, float* __restrict__ dshs, float* __restrict__ dscales,
    float* __restrict__ drots, float* __restrict__ dcov3D, typename ParamPtr<!ADAM>::type logit, FusedAdam ad)
{
    // per-wave slabs: 32 coefficient rows in, their gradients written back IN PLACE (each element is read before it is overwritten)
    constexpr bool HAS_SH = SH != 0;
    constexpr bool SLAB = SH == 1 || SH == 3;
    __shared__ __attribute__((aligned(16))) float s_sh[SLAB ? (kBlock / kWave) * kShHalf * kShPad : 1];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kBlock + tid;
    const bool in_range = i < P;
    if (ADAM && chain_failed(ad.fail)) {
        // this backward's blend timed out in a chained walk (NaN in the gradient records): no step -- parameters and moments stay as they are, the
        // screen-space gradient leaves as zeros; the host reports the event in front of the next render (Cam::chain_fail)
        if (in_range) { dmeans2D[3 * i] = 0.f; dmeans2D[3 * i + 1] = 0.f; dmeans2D[3 * i + 2] = 0.f; }
        return;
    }
    if (!HAS_SH && !in_range) return;
    const int ic = in_range ? i : P - 1;                       // clamped index: out-of-range lanes only help with the SH slabs
    const bool live = in_range && radii[ic] > 0;
    const float4* gr4 = reinterpret_cast<const float4*>(grad2d + (size_t)ic * kGradStride);
    // Keyframe batches (raw parameters, gradients ADDED to the buffers: rows of Gaussians that were not rendered stay untouched): a keyframe sees a
    // fraction of the map, and a Gaussian it does not see needs none of this kernel's reads.  There -- and only there -- the radius is waited for
    // first: a wavefront none of whose Gaussians was rendered leaves at once (a map grown frame by frame is coherent in memory), and the 64-byte
    // gradient records -- half of what the kernel reads -- are requested by the rendered lanes only (at the 27 % visibility of configs[3]'s random
    // shell scene about half of their lines).  Everywhere else every request stays up front in one round trip.
    const bool gate = ACT && !ADAM && !SLAB && cam.act_accumulate;
    float4 ga, gb, gc;
    if (gate) {
        if (!__any(live)) {
            // dL/dmeans2D is NOT accumulated: the header promises that every row is written, and the host hands the buffer out uninitialised
            // (torch.empty).  12 B per row against the 64-byte record read the gate saves.
            if (in_range) { dmeans2D[3 * i] = 0.f; dmeans2D[3 * i + 1] = 0.f; dmeans2D[3 * i + 2] = 0.f; }
            return;
        }
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        ga = z4; gb = z4; gc = z4;
        if (live) { ga = gr4[0]; gb = gr4[1]; gc = gr4[2]; }
    } else {
        ga = gr4[0]; gb = gr4[1]; gc = gr4[2];
    }
    float dmean[3] = {0.f, 0.f, 0.f};
    // record (raw moments from blend_backward_kernel, Z = G dL/dG, d = mean - pixel):
    //   ga = (sum Z dx, sum Z dy, sum Z dx dx, sum Z dx dy)  gb = (sum Z dy dy, sum G dL/dalpha, dr, dg)  gc = (db, -, -, -)
    float o_m2d[3] = {0.f, 0.f, 0.f}, o_sc[3] = {0.f, 0.f, 0.f}, o_rot[4] = {0.f, 0.f, 0.f, 0.f};
    float o_cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float px = means3D[3 * ic], py = means3D[3 * ic + 1], pz = means3D[3 * ic + 2];
    float actR[3][3];                                          // (ACT) rotation of the frame transform
    if (ACT) {
        quat_to_rot(cam.act_q, actR);
        const float wx = px, wy = py, wz = pz;
        px = actR[0][0] * wx + actR[0][1] * wy + actR[0][2] * wz + cam.act_t[0];
        py = actR[1][0] * wx + actR[1][1] * wy + actR[1][2] * wz + cam.act_t[1];
        pz = actR[2][0] * wx + actR[2][1] * wy + actR[2][2] * wz + cam.act_t[2];
    }
    const float drgb[3] = {gb.z, gb.w, gc.x};
    // every other per-Gaussian input is requested here as well, whether or not the lane turns out to need it: ONE memory round trip
    // per wavefront instead of three dependent ones (the compiler does not move loads out of the `live` branches below)
    uint32_t cl_in = 0u;
    float jac[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (HAS_SH && sh_jac) load_sh_jac(sh_jac, (size_t)ic, jac, cl_in);       // (the clamp flags travel in the Jacobian record)
    else if (HAS_SH) cl_in = clamped[ic];
    float sc_in[3] = {0.f, 0.f, 0.f};
    float4 rq_in = make_float4(0.f, 0.f, 0.f, 0.f), rq_raw = rq_in;
    float lg_in = 0.f;
    float cov_in[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cov3Dp) {
        for (int c = 0; c < 6; c++) cov_in[c] = cov3Dp[(size_t)ic * 6 + c];
    } else if (ACT) {
        if (cam.act_iso) { sc_in[0] = sc_in[1] = sc_in[2] = __expf(scales[ic]); }
        else { sc_in[0] = __expf(scales[3 * ic]); sc_in[1] = __expf(scales[3 * ic + 1]); sc_in[2] = __expf(scales[3 * ic + 2]); }
        rq_raw = reinterpret_cast<const float4*>(rots)[ic];
        lg_in = logit[ic];
        if (!live) {
            // A Gaussian that was not rendered leaves with ZERO parameter gradients.  The chain below multiplies its zero 2-D gradients with the
            // activations' derivatives -- which are NaN / inf for the very parameters the non-finite rule may have culled it for (0 x NaN): such a
            // row runs the chain on harmless stand-ins instead.  The Adam operands of the ADAM instantiation are loaded
            // separately from ad.p: the parameters themselves are stepped as they are.
            sc_in[0] = sc_in[1] = sc_in[2] = 0.f;
            rq_raw = make_float4(1.f, 0.f, 0.f, 0.f);
            lg_in = 0.f;
        }
        const float qr[4] = {rq_raw.x, rq_raw.y, rq_raw.z, rq_raw.w};
        float qa[4];
        activate_rotation(cam.act_q, cam.act_iso, qr, qa);
        rq_in = make_float4(qa[0], qa[1], qa[2], qa[3]);
    } else {
        sc_in[0] = scales[3 * ic]; sc_in[1] = scales[3 * ic + 1]; sc_in[2] = scales[3 * ic + 2];
        rq_in = reinterpret_cast<const float4*>(rots)[ic];
    }
    // ---- colour / SH: one wavefront's 64 coefficient rows per LDS pass, coalesced global traffic by all threads ----
    if (SLAB) {
        constexpr int KC = SH == 3 ? 48 : 0;
        const int deg = cam.sh_degree, nb = (deg + 1) * (deg + 1), M = KC ? 16 : cam.sh_coeffs, K = M * 3;
        const int stride = sh_row_stride(K);
        const int lane = tid & 63, wave = tid >> 6;
        float* slab = s_sh + wave * kShHalf * kShPad;
        for (int h = 0; h < kWave / kShHalf; h++) {
            const int row0 = blockIdx.x * kBlock + wave * kWave + h * kShHalf;
            if (row0 >= P) break;                                  // wave-uniform
            const int nrows = min(kShHalf, P - row0);
            __builtin_amdgcn_wave_barrier();
            // the colour's dependence on the mean (through the view direction) needs sum_k coef[k][ch] grad b_k: the forward saved that
            // 3x3 block per Gaussian (sh_jac), so the coefficient rows -- 192 B per Gaussian at 16 coefficients -- are NOT read again;
            // without it (a forward that was not told a backward follows) only live Gaussians read their rows
            const uint32_t live_rows = (uint32_t)(__ballot(live) >> (h * kShHalf));
            if (!sh_jac) sh_wave_rows_to_lds<KC>(slab, shs, row0, nrows, K, lane, live_rows);
            __builtin_amdgcn_wave_barrier();
            if ((lane >> 5) == h && in_range) {
                float* dsh = slab + (lane & 31) * stride;
                if (live) {
                    const float dx = px - cam.campos[0], dy = py - cam.campos[1], dz = pz - cam.campos[2];
                    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                    const float ux = dx * inv, uy = dy * inv, uz = dz * inv;
                    float b[16], bx[16], by[16], bz[16];
                    sh_basis_and_grad(deg, ux, uy, uz, b, bx, by, bz);
                    const uint32_t cl = cl_in;
                    float du[3] = {0.f, 0.f, 0.f};
                    // (k loops over the full 16 coefficients with predicates: constant indices keep b[] / bx[] ... in registers -- a loop bound
                    // of nb made them indexed locals in scratch memory in the ADAM instantiation)
                    if (ADAM || sh_jac) {                              // (ADAM: the API guarantees the saved Jacobian)
                        const float jr[3][3] = {{jac[0], jac[1], jac[2]}, {jac[3], jac[4], jac[5]}, {jac[6], jac[7], jac[8]}};
                        for (int ch = 0; ch < 3; ch++) {
                            const float g = ((cl >> (8 * ch)) & 1u) ? 0.f : drgb[ch];
                            du[0] += g * jr[ch][0]; du[1] += g * jr[ch][1]; du[2] += g * jr[ch][2];
#pragma unroll
                            for (int k = 0; k < 16; k++)
                                if (k < M) dsh[3 * k + ch] = k < nb ? g * b[k] : 0.f;
                        }
                    } else {
                        for (int ch = 0; ch < 3; ch++) {
                            const float g = ((cl >> (8 * ch)) & 1u) ? 0.f : drgb[ch];
#pragma unroll
                            for (int k = 0; k < 16; k++) {
                                if (k < nb) {
                                    const float coef = dsh[3 * k + ch];
                                    dsh[3 * k + ch] = g * b[k];
                                    du[0] += g * coef * bx[k]; du[1] += g * coef * by[k]; du[2] += g * coef * bz[k];
                                } else if (k < M) {
                                    dsh[3 * k + ch] = 0.f;
                                }
                            }
                        }
                    }
                    const float dot = ux * du[0] + uy * du[1] + uz * du[2];
                    dmean[0] += (du[0] - ux * dot) * inv; dmean[1] += (du[1] - uy * dot) * inv; dmean[2] += (du[2] - uz * dot) * inv;
                } else {
                    for (int k = 0; k < K; k++) dsh[k] = 0.f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (ADAM) {
                // the wavefront's (up to) 32 gradient rows against the same rows of the coefficients and their two moments: 16 B per lane,
                // coalesced, non-temporal (each is touched once per step); K = 48 is a multiple of 4, so a piece never straddles two rows
                float4* p4 = reinterpret_cast<float4*>(ad.p[4] + (size_t)row0 * K);
                float4* m4 = reinterpret_cast<float4*>(ad.m[4] + (size_t)row0 * K);
                float4* v4 = reinterpret_cast<float4*>(ad.v[4] + (size_t)row0 * K);
                const int total4 = (nrows * K) >> 2;
                for (int q4 = lane; q4 < total4; q4 += kWave) {
                    float4 pp = load_stream(&p4[q4]), mm = load_stream(&m4[q4]), vv = load_stream(&v4[q4]);
                    const int e = q4 << 2, rr = e / K, cc = e - rr * K;
                    const float* gs_ = slab + rr * stride + cc;
                    adam_elem(pp.x, gs_[0], mm.x, vv.x, ad.c[4]); adam_elem(pp.y, gs_[1], mm.y, vv.y, ad.c[4]);
                    adam_elem(pp.z, gs_[2], mm.z, vv.z, ad.c[4]); adam_elem(pp.w, gs_[3], mm.w, vv.w, ad.c[4]);
                    store_stream(&p4[q4], pp); store_stream(&m4[q4], mm); store_stream(&v4[q4], vv);
                }
            } else {
                sh_wave_rows_from_lds<KC>(slab, dshs, row0, nrows, K, lane);
            }
        }
        if (!in_range) return;
    }
    if (live) {
        const float* m = cam.view;
        const float* q = cam.proj;
        // ---- recompute forward intermediates ----
        const float tx = m[0] * px + m[4] * py + m[8] * pz + m[12];
        const float ty = m[1] * px + m[5] * py + m[9] * pz + m[13];
        const float tz = m[2] * px + m[6] * py + m[10] * pz + m[14];
        const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
        const float txtz = tx / tz, tytz = ty / tz;
        const bool okx = !(txtz < -limx || txtz > limx), oky = !(tytz < -limy || tytz > limy);
        const float cx_ = fminf(limx, fmaxf(-limx, txtz)) * tz, cy_ = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float J00 = cam.fx * itz, J02 = -(cam.fx * cx_) * itz2, J11 = cam.fy * itz, J12 = -(cam.fy * cy_) * itz2;
        const float W0[3] = {m[0], m[4], m[8]}, W1[3] = {m[1], m[5], m[9]}, W2[3] = {m[2], m[6], m[10]};
        float T0[3], T1[3];
        for (int c = 0; c < 3; c++) { T0[c] = J00 * W0[c] + J02 * W2[c]; T1[c] = J11 * W1[c] + J12 * W2[c]; }
        // 2-D covariance (with the 0.3 low-pass) p, q, r, its determinant, and Sigma T0^T, Sigma T1^T for the chain to T
        float ST0[3], ST1[3];
        float Rm[3][3], s3[3] = {0.f, 0.f, 0.f};
        float r = 0.f, x = 0.f, y = 0.f, z = 0.f;
        double p_d, q_d, r_d, det_d;
        // The ONE ill-conditioned block of the chain, in fp64: for an elongated splat the determinant of the 2-D covariance lies orders of magnitude
        // below its entries (seed 160050 of the round-5 sweep: p, q, r = 74, -208, 590, det = 396), and the conic-to-covariance gradient divides
        // by its square -- in fp32 this block alone put 3.5e-3 of relative error on that Gaussian's rotation gradient (the fp32 oracle: 3.4e-4); with
        // the block in fp64 the row is at 2.8e-5.  ~25 double-precision operations per Gaussian in an HBM-bound kernel: no measurable time
        // (profiles/README.md).
        if (cov3Dp) {
            const float* c = cov_in;
            const float S[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
            for (int a = 0; a < 3; a++) {
                ST0[a] = S[a][0] * T0[0] + S[a][1] * T0[1] + S[a][2] * T0[2];
                ST1[a] = S[a][0] * T1[0] + S[a][1] * T1[1] + S[a][2] * T1[2];
            }
            p_d = (double)T0[0] * ST0[0] + (double)T0[1] * ST0[1] + (double)T0[2] * ST0[2] + 0.3;
            q_d = (double)T1[0] * ST0[0] + (double)T1[1] * ST0[1] + (double)T1[2] * ST0[2];
            r_d = (double)T1[0] * ST1[0] + (double)T1[1] * ST1[1] + (double)T1[2] * ST1[2] + 0.3;
            det_d = p_d * r_d - q_d * q_d;
        } else {
            // scale + rotation: the forward's factorised form (preprocess.hip): A = T M with M = R diag(s), cov2D = A A^T + 0.3 I,
            // det = |a1 x a2|^2 + 0.3 (|a1|^2 + |a2|^2) + 0.09 -- round 6: with p r - q^2 even fp64 arithmetic inherits eps32 / sin^2(theta) from
            // the fp32 rounding of p, q, r themselves (2e-3 on the 240 : 1 needle of seed 180021); the cross product of the fp32 rows: 1e-5
            s3[0] = cam.mod * sc_in[0]; s3[1] = cam.mod * sc_in[1]; s3[2] = cam.mod * sc_in[2];
            const float4 rq = rq_in;
            r = rq.x; x = rq.y; y = rq.z; z = rq.w;
            Rm[0][0] = 1.f - 2.f * (y * y + z * z); Rm[0][1] = 2.f * (x * y - r * z); Rm[0][2] = 2.f * (x * z + r * y);
            Rm[1][0] = 2.f * (x * y + r * z); Rm[1][1] = 1.f - 2.f * (x * x + z * z); Rm[1][2] = 2.f * (y * z - r * x);
            Rm[2][0] = 2.f * (x * z - r * y); Rm[2][1] = 2.f * (y * z + r * x); Rm[2][2] = 1.f - 2.f * (x * x + y * y);
            float a1[3], a2[3];
            for (int j = 0; j < 3; j++) {
                a1[j] = (T0[0] * Rm[0][j] + T0[1] * Rm[1][j] + T0[2] * Rm[2][j]) * s3[j];
                a2[j] = (T1[0] * Rm[0][j] + T1[1] * Rm[1][j] + T1[2] * Rm[2][j]) * s3[j];
            }
            for (int a = 0; a < 3; a++) {                       // Sigma T^T = M (M^T T^T) = M a^T
                ST0[a] = Rm[a][0] * s3[0] * a1[0] + Rm[a][1] * s3[1] * a1[1] + Rm[a][2] * s3[2] * a1[2];
                ST1[a] = Rm[a][0] * s3[0] * a2[0] + Rm[a][1] * s3[1] * a2[1] + Rm[a][2] * s3[2] * a2[2];
            }
            const double n1 = (double)a1[0] * a1[0] + (double)a1[1] * a1[1] + (double)a1[2] * a1[2];
            const double n2 = (double)a2[0] * a2[0] + (double)a2[1] * a2[1] + (double)a2[2] * a2[2];
            const double xc = (double)a1[1] * a2[2] - (double)a1[2] * a2[1], yc = (double)a1[2] * a2[0] - (double)a1[0] * a2[2],
                         zc = (double)a1[0] * a2[1] - (double)a1[1] * a2[0];
            p_d = n1 + 0.3;
            q_d = (double)a1[0] * a2[0] + (double)a1[1] * a2[1] + (double)a1[2] * a2[2];
            r_d = n2 + 0.3;
            det_d = (xc * xc + yc * yc + zc * zc) + 0.3 * (n1 + n2) + 0.09;
        }
        const double d2_d = 1.0 / (det_d * det_d);
        const double dA = -0.5 * ga.z, dB = -(double)ga.w, dC = -0.5 * gb.x;      // true partials w.r.t. conic (a, b, c)
        const float dp = (float)((-r_d * r_d * dA + q_d * r_d * dB - q_d * q_d * dC) * d2_d);
        const float dq = (float)((2.0 * q_d * r_d * dA - (p_d * r_d + q_d * q_d) * dB + 2.0 * p_d * q_d * dC) * d2_d);
        const float dr = (float)((-q_d * q_d * dA + p_d * q_d * dB - p_d * p_d * dC) * d2_d);
        const float p_ = (float)p_d, q_ = (float)q_d, r_ = (float)r_d, det = (float)det_d;
        const float h = 0.5f * dq;
        // dL/dSigma = T^T G2 T,  G2 = [[dp,h],[h,dr]]
        float dS[3][3];
        for (int a = 0; a < 3; a++)
            for (int b2 = 0; b2 < 3; b2++)
                dS[a][b2] = T0[a] * (dp * T0[b2] + h * T1[b2]) + T1[a] * (h * T0[b2] + dr * T1[b2]);
        o_cov[0] = dS[0][0]; o_cov[3] = dS[1][1]; o_cov[5] = dS[2][2];
        o_cov[1] = 2.f * dS[0][1]; o_cov[2] = 2.f * dS[0][2]; o_cov[4] = 2.f * dS[1][2];
        // dL/dT = 2 G2 T Sigma
        float dT0[3], dT1[3];
        for (int c = 0; c < 3; c++) {
            dT0[c] = 2.f * (dp * ST0[c] + h * ST1[c]);
            dT1[c] = 2.f * (h * ST0[c] + dr * ST1[c]);
        }
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
        for (int c = 0; c < 3; c++) { dJ00 += dT0[c] * W0[c]; dJ02 += dT0[c] * W2[c]; dJ11 += dT1[c] * W1[c]; dJ12 += dT1[c] * W2[c]; }
        const float dtx = okx ? -cam.fx * itz2 * dJ02 : 0.f;
        const float dty = oky ? -cam.fy * itz2 * dJ12 : 0.f;
        const float dtz = -cam.fx * itz2 * dJ00 - cam.fy * itz2 * dJ11 + 2.f * cam.fx * cx_ * itz3 * dJ02 + 2.f * cam.fy * cy_ * itz3 * dJ12;
        for (int c = 0; c < 3; c++) dmean[c] += dtx * W0[c] + dty * W1[c] + dtz * W2[c];
        // gradient of the blended depth output w.r.t. this Gaussian's view depth z = W2 . p + t_z (zero unless the
        // fused RGB-D backward ran)
        for (int c = 0; c < 3; c++) dmean[c] += gc.y * W2[c];
        // ---- mean2D -> mean3D through the projective divide ----
        // dL/dmean2D (pixels) = -(conic d-moments): (-(a M1x + b M1y), -(c M1y + b M1x)); conic = 1/det (r, -q, p)
        const float idet = 1.0f / det;
        const float ca = r_ * idet, cb = -q_ * idet, cc = p_ * idet;
        const float gxp = -(ca * ga.x + cb * ga.y), gyp = -(cc * ga.y + cb * ga.x);
        const float gxn = gxp * 0.5f * (float)cam.W, gyn = gyp * 0.5f * (float)cam.H;
        o_m2d[0] = gxn; o_m2d[1] = gyn;
        const float hx = q[0] * px + q[4] * py + q[8] * pz + q[12];
        const float hy = q[1] * px + q[5] * py + q[9] * pz + q[13];
        const float hw = q[3] * px + q[7] * py + q[11] * pz + q[15];
        const float pw = 1.0f / (hw + 1e-7f);
        const float dhx = gxn * pw, dhy = gyn * pw, dhw = -(gxn * hx + gyn * hy) * pw * pw;
        for (int c = 0; c < 3; c++) dmean[c] += dhx * q[4 * c] + dhy * q[4 * c + 1] + dhw * q[4 * c + 3];
        // ---- Sigma -> scale, quaternion ----
        if (!cov3Dp) {
            float dR[3][3];
            for (int j = 0; j < 3; j++) {
                float accs = 0.f;
                for (int a = 0; a < 3; a++) {
                    float acc = 0.f;
                    for (int b2 = 0; b2 < 3; b2++) acc += dS[a][b2] * Rm[b2][j];
                    const float dM = 2.f * acc * s3[j];          // dL/dM_aj, M = R diag(s)
                    accs += dM * Rm[a][j];
                    dR[a][j] = dM * s3[j];
                }
                o_sc[j] = accs * cam.mod;
            }
            o_rot[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            o_rot[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
            o_rot[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
            o_rot[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
    }
    if (ACT) {
        // gradients w.r.t. the parameters (activate.hip's backward): means through the frame rotation, rotation through the normalisations and
        // the camera quaternion, opacity through the sigmoid, scales through the exponential
        for (int c = 0; c < 3; c++) dmeans2D[3 * i + c] = o_m2d[c];
        const bool acc = !ADAM && cam.act_accumulate != 0;
        if (acc && !live) return;                                   // nothing to add
        const float dm[3] = {actR[0][0] * dmean[0] + actR[1][0] * dmean[1] + actR[2][0] * dmean[2],
                             actR[0][1] * dmean[0] + actR[1][1] * dmean[1] + actR[2][1] * dmean[2],
                             actR[0][2] * dmean[0] + actR[1][2] * dmean[1] + actR[2][2] * dmean[2]};
        if (ADAM) {
            // the step of this Gaussian's 11 (isotropic: 9) scalar parameters (+ 3 colours), in place; gradients exactly as the plain path
            // would have stored them
            const float o = 1.0f / (1.0f + __expf(-lg_in));
            const float dl = (live ? gb.y : 0.f) * o * (1.0f - o);
            const float ds[3] = {o_sc[0] * sc_in[0], o_sc[1] * sc_in[1], o_sc[2] * sc_in[2]};
            const float qr[4] = {rq_raw.x, rq_raw.y, rq_raw.z, rq_raw.w};
            float dq[4];
            activate_rotation_bwd(cam.act_q, cam.act_iso, qr, o_rot, dq);
            // every operand this thread needs is requested before the first update (one round trip; named scalars: no indexed locals)
            const int ns = cam.act_iso ? 1 : 3;
            const size_t i3 = (size_t)3 * i, is = (size_t)ns * i;
#define GS_LD3(T, j) float p##T##j = ad.p[T][(T == 2 ? is : i3) + j], m##T##j = ad.m[T][(T == 2 ? is : i3) + j], v##T##j = ad.v[T][(T == 2 ? is : i3) + j]
#define GS_ST3(T, j, g) do { adam_elem(p##T##j, (g), m##T##j, v##T##j, ad.c[T]);                                                  \
                             ad.p[T][(T == 2 ? is : i3) + j] = p##T##j; ad.m[T][(T == 2 ? is : i3) + j] = m##T##j; ad.v[T][(T == 2 ? is : i3) + j] = v##T##j; } while (0)
            GS_LD3(0, 0); GS_LD3(0, 1); GS_LD3(0, 2);
            float lp = ad.p[1][i], lm = ad.m[1][i], lv = ad.v[1][i];
            GS_LD3(2, 0);
            float p21 = 0.f, m21 = 0.f, v21 = 0.f, p22 = 0.f, m22 = 0.f, v22 = 0.f;
            if (!cam.act_iso) { p21 = ad.p[2][is + 1]; m21 = ad.m[2][is + 1]; v21 = ad.v[2][is + 1]; p22 = ad.p[2][is + 2]; m22 = ad.m[2][is + 2]; v22 = ad.v[2][is + 2]; }
            float4 rp = reinterpret_cast<const float4*>(ad.p[3])[i], rm = reinterpret_cast<const float4*>(ad.m[3])[i],
                   rv = reinterpret_cast<const float4*>(ad.v[3])[i];
            float p40 = 0.f, m40 = 0.f, v40 = 0.f, p41 = 0.f, m41 = 0.f, v41 = 0.f, p42 = 0.f, m42 = 0.f, v42 = 0.f;
            if (!HAS_SH) {
                p40 = ad.p[4][i3]; m40 = ad.m[4][i3]; v40 = ad.v[4][i3]; p41 = ad.p[4][i3 + 1]; m41 = ad.m[4][i3 + 1]; v41 = ad.v[4][i3 + 1];
                p42 = ad.p[4][i3 + 2]; m42 = ad.m[4][i3 + 2]; v42 = ad.v[4][i3 + 2];
            }
            GS_ST3(0, 0, dm[0]); GS_ST3(0, 1, dm[1]); GS_ST3(0, 2, dm[2]);
            adam_elem(lp, dl, lm, lv, ad.c[1]);
            ad.p[1][i] = lp; ad.m[1][i] = lm; ad.v[1][i] = lv;
            if (cam.act_iso) {
                GS_ST3(2, 0, ds[0] + ds[1] + ds[2]);
            } else {
                GS_ST3(2, 0, ds[0]); GS_ST3(2, 1, ds[1]); GS_ST3(2, 2, ds[2]);
            }
            adam_elem(rp.x, dq[0], rm.x, rv.x, ad.c[3]); adam_elem(rp.y, dq[1], rm.y, rv.y, ad.c[3]);
            adam_elem(rp.z, dq[2], rm.z, rv.z, ad.c[3]); adam_elem(rp.w, dq[3], rm.w, rv.w, ad.c[3]);
            reinterpret_cast<float4*>(ad.p[3])[i] = rp; reinterpret_cast<float4*>(ad.m[3])[i] = rm; reinterpret_cast<float4*>(ad.v[3])[i] = rv;
            if (!HAS_SH) {
                GS_ST3(4, 0, live ? drgb[0] : 0.f); GS_ST3(4, 1, live ? drgb[1] : 0.f); GS_ST3(4, 2, live ? drgb[2] : 0.f);
            }
#undef GS_LD3
#undef GS_ST3
            return;
        }
        for (int c = 0; c < 3; c++) dmeans3D[3 * i + c] = acc ? dmeans3D[3 * i + c] + dm[c] : dm[c];
        const float o = 1.0f / (1.0f + __expf(-lg_in));
        const float dl = (live ? gb.y : 0.f) * o * (1.0f - o);
        dopac[i] = acc ? dopac[i] + dl : dl;
        if (dcolors) for (int c = 0; c < 3; c++) { const float v = live ? drgb[c] : 0.f; dcolors[3 * i + c] = acc ? dcolors[3 * i + c] + v : v; }
        const float ds[3] = {o_sc[0] * sc_in[0], o_sc[1] * sc_in[1], o_sc[2] * sc_in[2]};
        if (cam.act_iso) { const float v = ds[0] + ds[1] + ds[2]; dscales[i] = acc ? dscales[i] + v : v; }
        else for (int c = 0; c < 3; c++) dscales[3 * i + c] = acc ? dscales[3 * i + c] + ds[c] : ds[c];
        const float qr[4] = {rq_raw.x, rq_raw.y, rq_raw.z, rq_raw.w};
        float dq[4];
        activate_rotation_bwd(cam.act_q, cam.act_iso, qr, o_rot, dq);
        float4 out = make_float4(dq[0], dq[1], dq[2], dq[3]);
        if (acc) { const float4 old = reinterpret_cast<const float4*>(drots)[i]; out.x += old.x; out.y += old.y; out.z += old.z; out.w += old.w; }
        reinterpret_cast<float4*>(drots)[i] = out;
        return;
    }
    for (int c = 0; c < 3; c++) { dmeans2D[3 * i + c] = o_m2d[c]; dmeans3D[3 * i + c] = dmean[c]; }
    dopac[i] = live ? gb.y : 0.f;
    if (dcolors) for (int c = 0; c < 3; c++) dcolors[3 * i + c] = live ? drgb[c] : 0.f;
    if (dscales) for (int c = 0; c < 3; c++) dscales[3 * i + c] = o_sc[c];
    if (drots) reinterpret_cast<float4*>(drots)[i] = make_float4(o_rot[0], o_rot[1], o_rot[2], o_rot[3]);
    if (dcov3D) for (int c = 0; c < 6; c++) dcov3D[6 * i + c] = o_cov[c];
}

hipError_t launch_preprocess_backward(const Cam& cam, int P, const float* means3D, const float* shs,
                                      const float* scales, const float* rots, const float* cov3Dp,
                                      const int32_t* radii, const uint32_t* clamped, const float2* sh_jac, const float* grad2d,
                                      float* dmeans2D, float* dmeans3D, float* dopac, float* dcolors, float* dshs,
                                      float* dscales, float* drots, float* dcov3D, const float* logit, const FusedAdam* adam, hipStream_t st)
{
    const int nb = (P + kBlock - 1) / kBlock;
    if (cam.act && (cov3Dp || !logit || !scales || !rots || (!adam && (!dscales || !drots)) || (shs && cam.sh_coeffs != 16))) return hipErrorInvalidValue;
    if (adam && (!cam.act || cam.act_accumulate)) return hipErrorInvalidValue;
    const FusedAdam ad = adam ? *adam : FusedAdam{};
#define GS_PBWD(...) hipLaunchKernelGGL((preprocess_backward_kernel<__VA_ARGS__>), dim3(nb), dim3(kBlock), 0, st, cam, P, means3D, shs, scales, rots, \
                                        cov3Dp, radii, clamped, sh_jac, grad2d, dmeans2D, dmeans3D, dopac, dcolors, dshs, dscales, drots, dcov3D, logit, ad)
    if (nb > 0 && shs && cam.sh_coeffs == 16) { if (adam) GS_PBWD(3, true, true); else if (cam.act) GS_PBWD(3, true); else GS_PBWD(3, false); }
    else if (nb > 0 && shs) GS_PBWD(1, false);
    else if (nb > 0) { if (adam) GS_PBWD(0, true, true); else if (cam.act) GS_PBWD(0, true); else GS_PBWD(0, false); }
#undef GS_PBWD
    return hipGetLastError();
}

}  // namespace gs
