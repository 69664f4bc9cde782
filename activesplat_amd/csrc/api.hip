// api.hip -- the C ABI of libgsplat_hip.so (declared in include/gsplat_hip.h).
// Plain pointers and sizes in, launches on the caller's stream out; no torch types, no hidden sync.
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "gs_common.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}

uint64_t align_up(uint64_t v, uint64_t a = 256) { return (v + a - 1) / a * a; }

// ---- optional per-stage timing with HIP events on the caller's stream (bench.py roofline leg) ----
enum Stage { ST_PREPROCESS = 0, ST_TILE_COUNT, ST_EMIT, ST_SORT, ST_RANGES, ST_TILE_SCATTER_SORT, ST_BLEND_FWD, ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_ADAM, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"preprocess_forward+scan", "tile_count+scan", "emit", "sort", "ranges",
                                     "tile_scatter+sort", "blend_forward", "blend_backward", "preprocess_backward", "adam"};
std::atomic<int> g_sort_path{GS_SORT_AUTO};          // (development knobs: atomics, read once per decision -- see the header's note on threads)
std::atomic<bool> g_segments_enabled{true};

int choose_path(int tiles, uint32_t max_tile_instances)
{
    // any list length a grid's y dimension can index in 2048-key blocks (runs beyond the LDS merge are merged pass by pass
    // through global memory); 0xffffffff = counts unknown (more tiles than the LDS histogram holds)
    const bool fits = tiles <= gs::kMaxLdsTiles && max_tile_instances <= 65535u * (uint32_t)gs::kSortChunk;
    if (g_sort_path == GS_SORT_RADIX) return GS_SORT_RADIX;
    return fits ? GS_SORT_TILE_LDS : GS_SORT_RADIX;
}
constexpr int kMaxPairs = 8192;
struct Prof {
    bool on = false;
    int n = 0;
    hipEvent_t ev[kMaxPairs][2];
    int stage[kMaxPairs];
    int created = 0;
} g_prof;

struct ScopedStage {
    int idx = -1;
    hipStream_t st;
    ScopedStage(int stage, hipStream_t s) : st(s)
    {
        if (!g_prof.on || g_prof.n >= kMaxPairs) return;
        idx = g_prof.n++;
        if (idx >= g_prof.created) { (void)hipEventCreate(&g_prof.ev[idx][0]); (void)hipEventCreate(&g_prof.ev[idx][1]); g_prof.created = idx + 1; }
        g_prof.stage[idx] = stage;
        (void)hipEventRecord(g_prof.ev[idx][0], st);
    }
    ~ScopedStage() { if (idx >= 0) (void)hipEventRecord(g_prof.ev[idx][1], st); }
};

int tile_bits(int tiles)
{
    int b = 1;
    while ((1 << b) < tiles) b++;
    return b;
}

bool make_cam(const GsCamera* c, gs::Cam& k)
{
    if (!c || c->image_width <= 0 || c->image_height <= 0 || !c->bg || !c->viewmatrix || !c->projmatrix) return false;
    if (!(c->tanfovx > 0.f) || !(c->tanfovy > 0.f)) return false;
    if (c->num_views < 0 || c->num_views > 64) return false;
    k.V = c->num_views > 1 ? c->num_views : 1;
    k.Wv = c->image_width; k.H = c->image_height;
    k.gxv = (k.Wv + gs::kTile - 1) / gs::kTile;
    k.gx = k.V * k.gxv; k.gy = (k.H + gs::kTile - 1) / gs::kTile;
    k.W = k.V > 1 ? k.gx * gs::kTile : k.Wv;               // atlas: every view padded to whole tiles
    k.nbv = 0;                                             // set by virtual_count()
    if (k.gx >= 65536 || k.gy >= 65536) return false;
    k.tanfovx = c->tanfovx; k.tanfovy = c->tanfovy;
    k.fx = (float)k.Wv / (2.0f * c->tanfovx); k.fy = (float)k.H / (2.0f * c->tanfovy);
    k.mod = c->scale_modifier;
    k.sh_degree = c->sh_degree; k.sh_coeffs = c->sh_coeffs;
    k.bg = c->bg; k.view = c->viewmatrix; k.proj = c->projmatrix; k.campos = c->campos;
    k.half = 0; k.split = 0;                               // decided by the blend launchers
    k.act = k.act_iso = k.act_accumulate = 0;
    return true;
}

// rows of the per-Gaussian state: P for one view; V x (P rounded up to whole 256-row blocks) virtual Gaussians for an atlas
int32_t virtual_count(gs::Cam& k, int32_t P)
{
    k.nbv = (P + gs::kBlock - 1) / gs::kBlock;
    return k.V > 1 ? k.V * k.nbv * gs::kBlock : P;
}

gs::GeomPtrs carve_geom(void* base, int32_t P, const gs::Cam& k)
{
    GsGeomLayout L;
    gs_geom_layout(P, k.W, k.H, &L);
    char* b = (char*)base;
    gs::GeomPtrs g;
    g.geom = (float4*)(b + L.geom); g.rect = (uint2*)(b + L.rect); g.tiles = (uint32_t*)(b + L.tiles_touched);
    g.offsets = (uint32_t*)(b + L.offsets); g.block_sums = (uint32_t*)(b + L.block_sums);
    g.clamped = (uint32_t*)(b + L.clamped);
    g.tile_total = (uint32_t*)(b + L.tile_total); g.tile_base = (uint32_t*)(b + L.tile_base);
    g.sh_jac = (float2*)(b + L.sh_jac);
    g.depth_bits = (uint32_t*)(b + L.depth_bits);
    g.vis_max = nullptr; g.vis_seen = nullptr;
    return g;
}

}  // namespace

extern "C" {

const char* gs_last_error(void) { return g_err; }
const char* gs_version(void) { return "activesplat_amd gsplat_hip 0.4 (gfx950)"; }
int32_t gs_abi_version(void) { return GS_ABI_VERSION; }

int gs_profile_enable(int32_t on)
{
    g_prof.on = on != 0;
    g_prof.n = 0;
    return GS_OK;
}

int32_t gs_profile_stage_count(void) { return ST_COUNT; }

const char* gs_profile_stage_name(int32_t stage) { return stage >= 0 && stage < ST_COUNT ? kStageNames[stage] : ""; }

int gs_profile_collect(float* ms_sum, int32_t* calls, int32_t n_stages)
{
    if (!ms_sum || !calls || n_stages < ST_COUNT) return fail(GS_EINVAL, "gs_profile_collect: bad argument");
    for (int i = 0; i < n_stages; i++) { ms_sum[i] = 0.f; calls[i] = 0; }
    for (int i = 0; i < g_prof.n; i++) {
        if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess) return fail(GS_ELAUNCH, "gs_profile_collect: event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) return fail(GS_ELAUNCH, "gs_profile_collect: elapsed failed");
        ms_sum[g_prof.stage[i]] += ms; calls[g_prof.stage[i]]++;
    }
    g_prof.n = 0;
    return GS_OK;
}

int gs_set_sort_path(int32_t path)
{
    if (path < GS_SORT_AUTO || path > GS_SORT_RADIX) return fail(GS_EINVAL, "gs_set_sort_path: bad path");
    g_sort_path = path;
    return GS_OK;
}

int gs_set_half_quadrants(int32_t max_tiles)
{
    gs::g_half_quadrant_tiles = max_tiles < 0 ? 0 : max_tiles;
    return GS_OK;
}

int gs_set_backward_chain(int32_t pieces, int32_t min_tiles)
{
    if (pieces < 1 || pieces > gs::kChainPieces) return fail(GS_EINVAL, "gs_set_backward_chain: pieces out of range");
    gs::g_chain_pieces = pieces;
    gs::g_chain_min_tiles = min_tiles < 0 ? gs::kChainMinTiles : min_tiles;
    return GS_OK;
}

int gs_set_backward_chain_tickets(int32_t on)
{
    gs::g_chain_tickets = on != 0;
    return GS_OK;
}

int gs_set_backward_chain_polls(int32_t polls)
{
    gs::g_chain_polls = polls == -1 ? gs::kChainPollsDefault : polls;      // (below -1: every waiting piece gives up at once -- tests)
    return GS_OK;
}

int gs_async_status_word(uint32_t** host_word)
{
    // one host-mapped word per process (portable: every device can raise it); plain host reads see it once the raising kernel has ended
    static uint32_t* word = nullptr;
    static std::mutex mu;                               // (both hosts above the ABI may ask for it first, from different threads)
    std::lock_guard<std::mutex> lock(mu);
    if (!word) {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            return fail(GS_ELAUNCH, "gs_async_status_word: hipHostMalloc failed");
        }
        memset(h, 0, 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); return fail(GS_ELAUNCH, "gs_async_status_word: no device view of the host word"); }
        // the sticky device-memory twin the optimiser kernels read (Cam::chain_fail).  Without it the library still works -- the step is then not
        // protected against a timed-out walk, as before round 6
        void* f = nullptr;
        if (hipMalloc(&f, 64) == hipSuccess && hipMemset(f, 0, 64) == hipSuccess && hipGetDevice(&gs::g_chain_fail_device) == hipSuccess) gs::g_chain_fail_dev = (uint32_t*)f;
        else (void)hipGetLastError();
        word = (uint32_t*)h;
        gs::g_async_status_dev = (uint32_t*)d;
    }
    if (host_word) *host_word = word;
    return GS_OK;
}

int gs_async_status_clear(void)
{
    // the host has seen and reported the event: optimiser steps run again.  Synchronous (the rare path): every launch enqueued so far has ended
    // when this returns, so no walker of an old launch can raise the word again behind the clear
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return fail(GS_ELAUNCH, "gs_async_status_clear: device synchronisation failed"); }
    if (gs::g_chain_fail_dev && hipMemset(gs::g_chain_fail_dev, 0, 4) != hipSuccess) { (void)hipGetLastError(); return fail(GS_ELAUNCH, "gs_async_status_clear: hipMemset failed"); }
    return GS_OK;
}

int gs_recorded_cut(uint32_t target, uint32_t* nearest, int32_t* level)
{
    const uint32_t pos = gs::cut_nearest(target);
    if (nearest) *nearest = pos;
    if (level) *level = pos ? gs::cut_level(pos) : -1;
    return GS_OK;
}

int gs_set_backward_segments(int32_t segments)
{
    if (segments < 1 || segments > gs::kFewSegmentsMax) return fail(GS_EINVAL, "gs_set_backward_segments: 1, 2 or 3");
    gs::g_few_segments = segments;
    return GS_OK;
}

int gs_set_forward_segments(int32_t on)
{
    g_segments_enabled = on != 0;
    return GS_OK;
}

int gs_atlas_layout(int32_t P, int32_t view_width, int32_t num_views, int32_t* virtual_P, int32_t* atlas_width, int32_t* view_stride)
{
    if (P < 0 || view_width <= 0 || num_views < 1 || num_views > 64) return fail(GS_EINVAL, "gs_atlas_layout: bad argument");
    const int32_t gxv = (view_width + gs::kTile - 1) / gs::kTile;
    const int32_t stride = num_views > 1 ? gxv * gs::kTile : view_width;
    if (virtual_P) *virtual_P = num_views > 1 ? num_views * ((P + gs::kBlock - 1) / gs::kBlock) * gs::kBlock : P;
    if (atlas_width) *atlas_width = num_views > 1 ? num_views * stride : view_width;
    if (view_stride) *view_stride = stride;
    return GS_OK;
}

int gs_geom_layout(int32_t P, int32_t width, int32_t height, GsGeomLayout* out)
{
    if (!out || P < 0 || width <= 0 || height <= 0) return fail(GS_EINVAL, "gs_geom_layout: bad argument");
    const uint64_t n = (uint64_t)(P > 0 ? P : 1);
    const uint64_t nb = (n + gs::kBlock - 1) / gs::kBlock;
    const uint64_t tiles = (uint64_t)((width + gs::kTile - 1) / gs::kTile) * ((height + gs::kTile - 1) / gs::kTile);
    const uint64_t rows = (n + 1023) / 1024;   // sized for the smallest binning chunk
    uint64_t o = 0;
    out->geom = o; o = align_up(o + n * GS_GEOM_FLOATS * 4);
    out->rect = o; o = align_up(o + n * 8);
    out->tiles_touched = o; o = align_up(o + n * 4);
    out->offsets = o; o = align_up(o + n * 4);
    out->block_sums = o; o = align_up(o + (nb + 1) * 4);
    out->clamped = o; o = align_up(o + n * 4);
    out->tile_total = o; o = align_up(o + tiles * 4);
    out->tile_base = o; o = align_up(o + (tiles <= (uint64_t)gs::kMaxLdsTiles ? rows * tiles * 4 : 4));
    out->sh_jac = o; o = align_up(o + n * gs::kShJacFloats * 4);
    out->depth_bits = o; o = align_up(o + n * 4);
    out->total_bytes = o;
    return GS_OK;
}

int gs_image_layout(int32_t width, int32_t height, GsImageLayout* out)
{
    if (!out || width <= 0 || height <= 0) return fail(GS_EINVAL, "gs_image_layout: bad argument");
    const uint64_t tiles = (uint64_t)((width + gs::kTile - 1) / gs::kTile) * ((height + gs::kTile - 1) / gs::kTile);
    const uint64_t hw = (uint64_t)width * height;
    uint64_t o = 0;
    out->ranges = o; o = align_up(o + tiles * 8);
    out->final_T = o; o = align_up(o + hw * 4);
    out->n_contrib = o; o = align_up(o + hw * 4);
    // (recorded only for images of few tiles: every pixel's state at the recorded list positions the segmented backward resumes from)
    // (images of many tiles: the hand-over state of the chained backward walks lives at the same offset)
    out->split_state = o; o = align_up(o + (tiles <= (uint64_t)gs::kFewTiles ? ((uint64_t)gs::kCutLevels * 5 + 4) * hw + 4
                                             : (uint64_t)gs::chain_state_words(tiles)) * 4);
    out->total_bytes = o;
    return GS_OK;
}

int gs_bin_layout(int64_t D, uint32_t max_tile_instances, int32_t width, int32_t height, GsBinLayout* out)
{
    if (!out || D < 0 || width <= 0 || height <= 0) return fail(GS_EINVAL, "gs_bin_layout: bad argument");
    const int tiles = ((width + gs::kTile - 1) / gs::kTile) * ((height + gs::kTile - 1) / gs::kTile);
    const uint64_t n = (uint64_t)(D > 0 ? D : 1);
    memset(out, 0, sizeof(*out));
    out->path = (uint64_t)choose_path(tiles, max_tile_instances);
    uint64_t o = 0;
    if (out->path == GS_SORT_TILE_LDS) {
        out->pairs = o; o = align_up(o + n * 8);
        if (max_tile_instances > (uint32_t)gs::kSortCapMax) { out->pairs_alt = o; o = align_up(o + n * 8); }
    } else {
        out->keys_unsorted = o; o = align_up(o + n * 8);
        out->vals_unsorted = o; o = align_up(o + n * 4);
        out->keys_sorted = o; o = align_up(o + n * 8);
        out->sort_temp = o; o = align_up(o + gs::sort_temp_bytes(D, 32 + tile_bits(tiles)));
    }
    // few tiles with very long lists (the planner's 120 x 150 views of a large map): cut every list into segments that are
    // composited in parallel -- enough of them to fill the 5120 wavefront slots, each at least 1024 records long
    out->segments = 1;
    // (pass 1 walks the WHOLE list, the normal walk stops where T saturates: at 256 tiles x 11 k records the normal path is 1.8x
    // faster, at 80 tiles x 78 k the segmented one 4.7x, at the 240 tiles x 78 k of a three-view atlas 2.5x -- so: lists of at least
    // 8192 in at most 160 tiles, or of at least 32768 as long as the tiles alone cannot fill the wavefront slots)
    const bool few_tiles = tiles * 4 * 8 <= gs::kWaveSlots && max_tile_instances >= 8192;
    const bool long_lists = tiles * 4 * 2 <= gs::kWaveSlots && max_tile_instances >= 32768;
    if (g_segments_enabled && max_tile_instances != 0xffffffffu && (few_tiles || long_lists)) {
        uint64_t S = 3 * (uint64_t)gs::kWaveSlots / ((uint64_t)tiles * 4);      // 3x oversubscribed: segments differ in work (early stop);
                                                                                // measured on the 240-tile atlas: 1x 1.10, 2x 1.06, 3x 0.95, 4x 0.96, 6x 1.05 ms
        const uint64_t by_len = max_tile_instances / 1024;
        if (S > by_len) S = by_len;
        if (S > 32) S = 32;
        if (S >= 2) { out->segments = S; out->seg_T = o; o = align_up(o + (uint64_t)tiles * S * gs::kBlock * 4 + (uint64_t)tiles * 16); }
    }
    out->total_bytes = o;
    return GS_OK;
}

uint64_t gs_backward_scratch_bytes(int32_t P) { return align_up((uint64_t)(P > 0 ? P : 1) * gs::kGradStride * 4); }

static bool set_input_activation(gs::Cam& k, const float* h_pose7, int32_t isotropic, int32_t accumulate)
{
    k.act = k.act_iso = k.act_accumulate = 0;
    if (!h_pose7) return true;
    k.act = 1; k.act_iso = isotropic != 0; k.act_accumulate = accumulate != 0;
    for (int c = 0; c < 4; c++) k.act_q[c] = h_pose7[c];
    for (int c = 0; c < 3; c++) k.act_t[c] = h_pose7[4 + c];
    return k.V == 1;
}

static int preprocess_forward_impl(const GsCamera* cam, int32_t P, const float* means3D, const float* shs,
                                   const float* colors_precomp, const float* opacities, const float* scales,
                                   const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_state,
                                   void* image_state, uint32_t* d_counts, uint32_t* h_counts, int32_t want_backward, gs_stream_t stream,
                                   const float* h_pose7, int32_t isotropic, float* max_2D_radius, uint8_t* seen)
{
    gs::Cam k;
    if (!make_cam(cam, k)) return fail(GS_EINVAL, "gs_preprocess_forward: invalid camera settings");
    if (!set_input_activation(k, h_pose7, isotropic, 0)) return fail(GS_EINVAL, "gs_preprocess_forward_raw: one view only");
    if (k.act && (cov3D_precomp || (shs && k.sh_coeffs != 16)))
        return fail(GS_EINVAL, "gs_preprocess_forward_raw: scale / rotation parameters with colours or 16-coefficient SH rows only");
    if (P < 0 || !geom_state || !image_state || !d_counts) return fail(GS_EINVAL, "gs_preprocess_forward: null state pointer");
    if (P > 0 && (!means3D || !opacities || !radii)) return fail(GS_EINVAL, "gs_preprocess_forward: null input pointer");
    if ((shs == nullptr) == (colors_precomp == nullptr) && P > 0)
        return fail(GS_EINVAL, "Please provide excatly one of either SHs or precomputed colors!");
    const bool have_sr = scales != nullptr && rotations != nullptr;
    if (P > 0 && (have_sr == (cov3D_precomp != nullptr) || ((scales != nullptr) != (rotations != nullptr))))
        return fail(GS_EINVAL, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (shs && (k.sh_degree < 0 || k.sh_degree > 3 || k.sh_coeffs < (k.sh_degree + 1) * (k.sh_degree + 1) || k.sh_coeffs > 16 || !k.campos))
        return fail(GS_EINVAL, "gs_preprocess_forward: sh_degree / sh_coeffs / campos inconsistent");
    hipStream_t st = (hipStream_t)stream;
    const int32_t Pv = virtual_count(k, P);
    gs::GeomPtrs gp = carve_geom(geom_state, Pv, k);
    if (!(want_backward && shs)) gp.sh_jac = nullptr;        // written only for SH inputs whose backward will follow
    gp.vis_max = k.act ? max_2D_radius : nullptr; gp.vis_seen = k.act ? seen : nullptr;
    GsImageLayout IL; gs_image_layout(k.W, k.H, &IL);
    uint2* ranges = (uint2*)((char*)image_state + IL.ranges);
    const int tiles = k.gx * k.gy;
    hipError_t e;
    {
        ScopedStage ps(ST_PREPROCESS, st);
        e = gs::launch_preprocess_forward(k, P, means3D, shs, colors_precomp, opacities, scales, rotations,
                                          cov3D_precomp, radii, gp, d_counts, st);
    }
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_preprocess_forward: %s", hipGetErrorString(e));
    bool mirrored = false;
    if (tiles <= gs::kMaxLdsTiles) {          // tile counting: ranges, D and the largest tile list
        // if h_counts is mapped pinned host memory the scan kernel stores the counters there itself (no copy engine hop)
        uint32_t* host_dev = nullptr;
        if (h_counts && hipHostGetDevicePointer((void**)&host_dev, h_counts, 0) != hipSuccess) { host_dev = nullptr; (void)hipGetLastError(); }
        mirrored = host_dev != nullptr;
        ScopedStage ps(ST_TILE_COUNT, st);
        e = gs::launch_tile_count(k, Pv, gp, gp.tile_total, gp.tile_base, ranges, d_counts, host_dev, st);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_preprocess_forward: tile count %s", hipGetErrorString(e));
    } else {                                   // too many tiles for the LDS histogram: radix path, counts = {D, 2^32-1}
        e = gs::launch_scan_block_sums(Pv, gp, d_counts, st);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_preprocess_forward: scan %s", hipGetErrorString(e));
        e = hipMemsetAsync(d_counts + 1, 0xff, sizeof(uint32_t), st);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_preprocess_forward: memset %s", hipGetErrorString(e));
    }
    if (h_counts && !mirrored) {
        e = hipMemcpyAsync(h_counts, d_counts, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_preprocess_forward: D2H %s", hipGetErrorString(e));
    }
    return GS_OK;
}

int gs_preprocess_forward(const GsCamera* cam, int32_t P, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_state,
                          void* image_state, uint32_t* d_counts, uint32_t* h_counts, int32_t want_backward, gs_stream_t stream)
{
    return preprocess_forward_impl(cam, P, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, geom_state,
                                   image_state, d_counts, h_counts, want_backward, stream, nullptr, 0, nullptr, nullptr);
}

int gs_preprocess_forward_raw(const GsCamera* cam, int32_t P, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* logit_opacities, const float* log_scales, const float* unnorm_rotations,
                              const float* h_pose7, int32_t isotropic, float* max_2D_radius, uint8_t* seen, int32_t* radii, void* geom_state,
                              void* image_state, uint32_t* d_counts, uint32_t* h_counts, int32_t want_backward, gs_stream_t stream)
{
    if (!h_pose7) return fail(GS_EINVAL, "gs_preprocess_forward_raw: null pose");
    return preprocess_forward_impl(cam, P, means3D, shs, colors_precomp, logit_opacities, log_scales, unnorm_rotations, nullptr, radii,
                                   geom_state, image_state, d_counts, h_counts, want_backward, stream, h_pose7, isotropic, max_2D_radius, seen);
}


int gs_render_forward(const GsCamera* cam, int32_t P, int64_t D, uint32_t max_tile_instances, void* geom_state,
                      void* bin_state, uint32_t* point_list, void* image_state, float* out_color, float* out_depth,
                      float* out_opacity, float* out_depth_sq, void* backward_scratch, gs_stream_t stream)
{
    gs::Cam k;
    if (!make_cam(cam, k)) return fail(GS_EINVAL, "gs_render_forward: invalid camera settings");
    if (P < 0 || D < 0 || !geom_state || !image_state || !out_color || !out_depth || !out_opacity)
        return fail(GS_EINVAL, "gs_render_forward: null pointer");
    if (D > 0 && (!bin_state || !point_list)) return fail(GS_EINVAL, "gs_render_forward: null binning workspace");
    if (D >= (int64_t)1 << 32) return fail(GS_ECAPACITY, "gs_render_forward: more than 2^32 tile instances");
    hipStream_t st = (hipStream_t)stream;
    const int32_t Pin = P;
    P = virtual_count(k, Pin);                             // every stage below works on the virtual Gaussians of the atlas
    gs::GeomPtrs gp = carve_geom(geom_state, P, k);
    GsImageLayout IL; gs_image_layout(k.W, k.H, &IL);
    char* ib = (char*)image_state;
    uint2* ranges = (uint2*)(ib + IL.ranges);
    GsBinLayout BL; gs_bin_layout(D, max_tile_instances, k.W, k.H, &BL);
    char* bb = (char*)bin_state;
    hipError_t e;
    if (BL.path == GS_SORT_TILE_LDS) {
        if (D > 0) {          // ranges were written by gs_preprocess_forward
            ScopedStage ps(ST_TILE_SCATTER_SORT, st);
            e = gs::launch_tile_scatter_sort(k, P, gp, gp.tile_base, ranges, max_tile_instances,
                                             (unsigned long long*)(bb + BL.pairs), (unsigned long long*)(bb + BL.pairs_alt), point_list, (uint32_t)D, st);
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: tile scatter/sort %s", hipGetErrorString(e));
        }
    } else {
        e = hipMemsetAsync(ranges, 0, (size_t)k.gx * k.gy * 8, st);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: memset %s", hipGetErrorString(e));
        if (k.gx * k.gy <= gs::kMaxLdsTiles) {   // per-Gaussian offsets were not needed before the sync: scan them now
            e = gs::launch_scan_block_sums(P, gp, gp.block_sums + (P + gs::kBlock - 1) / gs::kBlock, st);
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: scan %s", hipGetErrorString(e));
        }
        if (D == 0) {   // nothing visible: the emitter still writes the (all-zero) scan offsets
            e = gs::launch_emit(k, P, gp, nullptr, nullptr, st);
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: emit %s", hipGetErrorString(e));
        } else {
            uint64_t* ku = (uint64_t*)(bb + BL.keys_unsorted); uint32_t* vu = (uint32_t*)(bb + BL.vals_unsorted);
            uint64_t* ks = (uint64_t*)(bb + BL.keys_sorted);
            { ScopedStage ps(ST_EMIT, st); e = gs::launch_emit(k, P, gp, ku, vu, st); }
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: emit %s", hipGetErrorString(e));
            const int end_bit = 32 + tile_bits(k.gx * k.gy);
            {
                ScopedStage ps(ST_SORT, st);
                e = gs::sort_pairs(bb + BL.sort_temp, (size_t)(BL.total_bytes - BL.sort_temp), ku, ks, vu, point_list, D, end_bit, st);
            }
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: sort %s", hipGetErrorString(e));
            { ScopedStage ps(ST_RANGES, st); e = gs::launch_ranges(D, ks, ranges, st); }
            if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: ranges %s", hipGetErrorString(e));
        }
    }
    {
        ScopedStage ps(ST_BLEND_FWD, st);
        e = gs::launch_blend_forward(k, ranges, point_list, gp.geom, out_color, out_depth, out_opacity,
                                     (float*)(ib + IL.final_T), (uint32_t*)(ib + IL.n_contrib), out_depth_sq,
                                     BL.path == GS_SORT_TILE_LDS ? (uint32_t)D : 0xffffffffu, (int)BL.segments,
                                     BL.segments > 1 ? (float*)(bb + BL.seg_T) : nullptr,
                                     (float*)(ib + IL.split_state), (uint32_t)P,
                                     (float*)backward_scratch, st);
    }
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_forward: blend %s", hipGetErrorString(e));
    return GS_OK;
}

static int render_backward_impl(const GsCamera* cam, int32_t P, int64_t D, const float* means3D, const float* shs,
                                const float* colors_precomp, const float* scales, const float* rotations,
                                const float* cov3D_precomp, const int32_t* radii, const void* geom_state,
                                const uint32_t* point_list, const void* image_state, const float* dL_dcolor,
                                const float* dL_ddepth, float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dopacities, float* dL_dcolors_precomp,
                                float* dL_dshs, float* dL_dscales, float* dL_drotations, float* dL_dcov3D, void* scratch,
                                int32_t scratch_zeroed, int32_t have_sh_jacobian, gs_stream_t stream, const float* logit,
                                const float* h_pose7, int32_t isotropic, int32_t accumulate, const GsAdamTensor* adam5 = nullptr)
{
    gs::Cam k;
    if (!make_cam(cam, k)) return fail(GS_EINVAL, "gs_render_backward: invalid camera settings");
    if (!set_input_activation(k, h_pose7, isotropic, accumulate)) return fail(GS_EINVAL, "gs_render_backward_raw: one view only");
    if (k.act && (!logit || cov3D_precomp || (shs && k.sh_coeffs != 16)))
        return fail(GS_EINVAL, "gs_render_backward_raw: scale / rotation parameters with colours or 16-coefficient SH rows only");
    if (k.V > 1) return fail(GS_EINVAL, "gs_render_backward: multi-view atlas renders are forward-only");
    if (P < 0 || D < 0 || !geom_state || !image_state || !dL_dcolor || !scratch)
        return fail(GS_EINVAL, "gs_render_backward: null pointer");
    if (P == 0) return GS_OK;
    gs::FusedAdam fa{};
    if (adam5) {
        // the optimiser step inside the per-Gaussian kernel: the five descriptors must describe the very tensors this call reads
        if (!k.act || accumulate) return fail(GS_EINVAL, "gs_render_backward_raw_adam: raw-parameter mode without accumulation only");
        if (!means3D || !radii || !dL_dmeans2D || !scales || !rotations || (shs == nullptr) == (colors_precomp == nullptr))
            return fail(GS_EINVAL, "gs_render_backward_raw_adam: null input/output pointer");
        if (shs && !have_sh_jacobian) return fail(GS_EINVAL, "gs_render_backward_raw_adam: SH rows need the forward's saved Jacobian (have_sh_jacobian = 1)");
        const float* par[5] = {means3D, logit, scales, rotations, shs ? shs : colors_precomp};
        const int64_t width[5] = {3, 1, isotropic ? 1 : 3, 4, shs ? 48 : 3};
        for (int t = 0; t < 5; ++t) {
            const GsAdamTensor& a = adam5[t];
            if (a.param != par[t] || !a.exp_avg || !a.exp_avg_sq || a.step < 1 || a.n != width[t] * (int64_t)P)
                return fail(GS_EINVAL, "gs_render_backward_raw_adam: descriptor of %s does not describe the input tensor (param / moments / n / step)",
                            t == 0 ? "means3D" : t == 1 ? "logit_opacities" : t == 2 ? "log_scales" : t == 3 ? "unnorm_rotations" : "the colours");
            fa.p[t] = a.param; fa.m[t] = a.exp_avg; fa.v[t] = a.exp_avg_sq;
            fa.c[t] = gs::adam_coef(a.lr, a.beta1, a.beta2, a.eps, a.step);
        }
        fa.fail = gs::chain_fail_word();
    } else {
        if (!means3D || !radii || !dL_dmeans2D || !dL_dmeans3D || !dL_dopacities)
            return fail(GS_EINVAL, "gs_render_backward: null input/output pointer");
        if (shs ? !dL_dshs : !dL_dcolors_precomp) return fail(GS_EINVAL, "gs_render_backward: missing colour gradient output");
        if (cov3D_precomp ? !dL_dcov3D : (!scales || !rotations || !dL_dscales || !dL_drotations))
            return fail(GS_EINVAL, "gs_render_backward: missing covariance inputs/outputs");
    }
    hipStream_t st = (hipStream_t)stream;
    gs::GeomPtrs gp = carve_geom(const_cast<void*>(geom_state), P, k);
    GsImageLayout IL; gs_image_layout(k.W, k.H, &IL);
    const char* ib = (const char*)image_state;
    float* grad2d = (float*)scratch;
    hipError_t e = scratch_zeroed ? hipSuccess : hipMemsetAsync(grad2d, 0, (size_t)P * gs::kGradStride * 4, st);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_backward: memset %s", hipGetErrorString(e));
    if (D > 0) {
        ScopedStage ps(ST_BLEND_BWD, st);
        e = gs::launch_blend_backward(k, (const uint2*)(ib + IL.ranges), point_list, gp.geom,
                                      (const float*)(ib + IL.split_state),
                                      (const float*)(ib + IL.final_T),
                                      (const uint32_t*)(ib + IL.n_contrib), dL_dcolor, dL_ddepth, grad2d, st);
        if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_backward: blend %s", hipGetErrorString(e));
    }
    {
        ScopedStage ps(ST_PREPROCESS_BWD, st);
        e = gs::launch_preprocess_backward(k, P, means3D, shs, scales, rotations, cov3D_precomp, radii, gp.clamped, (shs && have_sh_jacobian) ? gp.sh_jac : nullptr, grad2d,
                                           dL_dmeans2D, dL_dmeans3D, dL_dopacities, dL_dcolors_precomp, dL_dshs, dL_dscales,
                                           dL_drotations, dL_dcov3D, logit, adam5 ? &fa : nullptr, st);
    }
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_render_backward: preprocess %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_render_backward(const GsCamera* cam, int32_t P, int64_t D, const float* means3D, const float* shs,
                       const float* colors_precomp, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const int32_t* radii, const void* geom_state,
                       const uint32_t* point_list, const void* image_state, const float* dL_dcolor,
                       const float* dL_ddepth, float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dopacities, float* dL_dcolors_precomp,
                       float* dL_dshs, float* dL_dscales, float* dL_drotations, float* dL_dcov3D, void* scratch,
                       int32_t scratch_zeroed, int32_t have_sh_jacobian, gs_stream_t stream)
{
    return render_backward_impl(cam, P, D, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, radii, geom_state, point_list,
                                image_state, dL_dcolor, dL_ddepth, dL_dmeans2D, dL_dmeans3D, dL_dopacities, dL_dcolors_precomp, dL_dshs,
                                dL_dscales, dL_drotations, dL_dcov3D, scratch, scratch_zeroed, have_sh_jacobian, stream, nullptr, nullptr, 0, 0);
}

int gs_render_backward_raw(const GsCamera* cam, int32_t P, int64_t D, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* logit_opacities, const float* log_scales, const float* unnorm_rotations, const float* h_pose7,
                           int32_t isotropic, int32_t accumulate, const int32_t* radii, const void* geom_state, const uint32_t* point_list,
                           const void* image_state, const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans2D, float* dL_dmeans3D,
                           float* dL_dlogit_opacities, float* dL_dcolors_precomp, float* dL_dshs, float* dL_dlog_scales,
                           float* dL_dunnorm_rotations, void* scratch, int32_t scratch_zeroed, int32_t have_sh_jacobian, gs_stream_t stream)
{
    if (!h_pose7 || (P > 0 && !logit_opacities)) return fail(GS_EINVAL, "gs_render_backward_raw: null pose / opacity parameters");
    return render_backward_impl(cam, P, D, means3D, shs, colors_precomp, log_scales, unnorm_rotations, nullptr, radii, geom_state, point_list,
                                image_state, dL_dcolor, dL_ddepth, dL_dmeans2D, dL_dmeans3D, dL_dlogit_opacities, dL_dcolors_precomp, dL_dshs,
                                dL_dlog_scales, dL_dunnorm_rotations, nullptr, scratch, scratch_zeroed, have_sh_jacobian, stream,
                                logit_opacities, h_pose7, isotropic, accumulate);
}

int gs_render_backward_raw_adam(const GsCamera* cam, int32_t P, int64_t D, const float* means3D, const float* shs, const float* colors_precomp,
                                const float* logit_opacities, const float* log_scales, const float* unnorm_rotations, const float* h_pose7,
                                int32_t isotropic, const int32_t* radii, const void* geom_state, const uint32_t* point_list,
                                const void* image_state, const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans2D, void* scratch,
                                int32_t scratch_zeroed, int32_t have_sh_jacobian, const GsAdamTensor* adam5, gs_stream_t stream)
{
    if (!h_pose7 || !adam5 || (P > 0 && !logit_opacities)) return fail(GS_EINVAL, "gs_render_backward_raw_adam: null pose / opacity parameters / descriptors");
    return render_backward_impl(cam, P, D, means3D, shs, colors_precomp, log_scales, unnorm_rotations, nullptr, radii, geom_state, point_list,
                                image_state, dL_dcolor, dL_ddepth, dL_dmeans2D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                scratch, scratch_zeroed, have_sh_jacobian, stream, logit_opacities, h_pose7, isotropic, 0, adam5);
}

int gs_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, double lr,
                 double beta1, double beta2, double eps, int32_t step, gs_stream_t stream)
{
    if (n < 0 || step < 1) return fail(GS_EINVAL, "gs_adam_step: bad n/step");
    if (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq)) return fail(GS_EINVAL, "gs_adam_step: null pointer");
    hipError_t e;
    {
        ScopedStage ps(ST_ADAM, (hipStream_t)stream);
        e = gs::launch_adam(n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_adam_step: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_adam_step_multi(int32_t count, const GsAdamTensor* tensors, gs_stream_t stream)
{
    if (count < 0 || (count > 0 && !tensors)) return fail(GS_EINVAL, "gs_adam_step_multi: bad count/tensors");
    for (int i = 0; i < count; ++i) {
        const GsAdamTensor& t = tensors[i];
        if (t.n < 0 || t.step < 1) return fail(GS_EINVAL, "gs_adam_step_multi: a tensor has bad n/step");
        if (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) return fail(GS_EINVAL, "gs_adam_step_multi: a tensor has a null pointer");
    }
    hipError_t e;
    {
        ScopedStage sc(ST_ADAM, (hipStream_t)stream);
        e = gs::launch_adam_multi(count, tensors, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_adam_step_multi: %s", hipGetErrorString(e));
    return GS_OK;
}

static int rows_args_ok(int32_t count, const GsRowTensor* t, int mode)
{
    if (count < 1 || count > gs::kAdamMaxTensors || !t) return 0;
    int G = 0;
    for (int i = 0; i < count; ++i) {
        if (t[i].width < 1 || t[i].width > 64) return 0;
        G += t[i].width;
        if (mode != 0 && !t[i].param) return 0;
        if (mode == 2 && (!t[i].exp_avg || !t[i].exp_avg_sq || t[i].step < 1)) return 0;
    }
    return G <= 64;
}

int gs_pack_columns(int32_t count, const GsRowTensor* tensors, int64_t n, int64_t n_padded, float* flat, gs_stream_t stream)
{
    if (!rows_args_ok(count, tensors, 0) || n < 0 || n_padded < n || (n_padded > 0 && !flat)) return fail(GS_EINVAL, "gs_pack_columns: bad argument");
    hipError_t e = gs::launch_rows(0, count, tensors, 0, n, n_padded, nullptr, flat, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_pack_columns: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_adam_rows(int32_t count, const GsRowTensor* tensors, int64_t row_lo, int64_t n_valid, int64_t n_rows, const float* grad_shard,
                 float* out_shard, gs_stream_t stream)
{
    if (!rows_args_ok(count, tensors, 2) || row_lo < 0 || n_valid < 0 || n_rows < n_valid || (n_rows > 0 && !grad_shard))
        return fail(GS_EINVAL, "gs_adam_rows: bad argument");
    hipError_t e;
    {
        ScopedStage sc(ST_ADAM, (hipStream_t)stream);
        e = gs::launch_rows(2, count, tensors, row_lo, n_valid, n_rows, grad_shard, out_shard, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_adam_rows: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_unpack_columns(int32_t count, const GsRowTensor* tensors, int64_t n, const float* flat, gs_stream_t stream)
{
    if (!rows_args_ok(count, tensors, 1) || n < 0 || (n > 0 && !flat)) return fail(GS_EINVAL, "gs_unpack_columns: bad argument");
    hipError_t e = gs::launch_rows(1, count, tensors, 0, n, n, flat, nullptr, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_unpack_columns: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_activate_forward(int32_t P, int32_t isotropic, const float* h_pose7, const float* means3D, const float* unnorm_rotations,
                        const float* logit_opacities, const float* log_scales, float* out_means3D, float* out_rotations,
                        float* out_opacities, float* out_scales, gs_stream_t stream)
{
    if (P < 0 || !h_pose7 || (P > 0 && (!means3D || !unnorm_rotations || !logit_opacities || !log_scales || !out_means3D ||
                                        !out_rotations || !out_opacities || !out_scales)))
        return fail(GS_EINVAL, "gs_activate_forward: bad argument");
    hipError_t e = gs::launch_activate_forward(P, isotropic, h_pose7, means3D, unnorm_rotations, logit_opacities, log_scales,
                                               out_means3D, out_rotations, out_opacities, out_scales, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_activate_forward: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_activate_backward(int32_t P, int32_t isotropic, const float* h_pose7, const float* unnorm_rotations, const float* out_opacities,
                         const float* out_scales, const float* g_means3D, const float* g_rotations, const float* g_opacities,
                         const float* g_scales, float* d_means3D, float* d_unnorm_rotations, float* d_logit_opacities,
                         float* d_log_scales, gs_stream_t stream)
{
    if (P < 0 || !h_pose7 || (P > 0 && (!unnorm_rotations || !out_opacities || !out_scales || !d_means3D || !d_unnorm_rotations ||
                                        !d_logit_opacities || !d_log_scales)))
        return fail(GS_EINVAL, "gs_activate_backward: bad argument");
    hipError_t e = gs::launch_activate_backward(P, isotropic, h_pose7, unnorm_rotations, out_opacities, out_scales, g_means3D, g_rotations,
                                                g_opacities, g_scales, d_means3D, d_unnorm_rotations, d_logit_opacities, d_log_scales, 0,
                                                (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_activate_backward: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_activate_backward_accumulate(int32_t P, int32_t isotropic, const float* h_pose7, const float* unnorm_rotations, const float* out_opacities,
                                    const float* out_scales, const float* g_means3D, const float* g_rotations, const float* g_opacities,
                                    const float* g_scales, float* d_means3D, float* d_unnorm_rotations, float* d_logit_opacities,
                                    float* d_log_scales, gs_stream_t stream)
{
    if (P < 0 || !h_pose7 || (P > 0 && (!unnorm_rotations || !out_opacities || !out_scales || !d_means3D || !d_unnorm_rotations ||
                                        !d_logit_opacities || !d_log_scales)))
        return fail(GS_EINVAL, "gs_activate_backward_accumulate: bad argument");
    hipError_t e = gs::launch_activate_backward(P, isotropic, h_pose7, unnorm_rotations, out_opacities, out_scales, g_means3D, g_rotations,
                                                g_opacities, g_scales, d_means3D, d_unnorm_rotations, d_logit_opacities, d_log_scales, 1,
                                                (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_activate_backward_accumulate: %s", hipGetErrorString(e));
    return GS_OK;
}

uint64_t gs_mapping_loss_scratch_bytes(int32_t width, int32_t height)
{
    return align_up((uint64_t)(2 * gs::kLossAccSlots * 16 + 9 * (uint64_t)(width > 0 ? width : 1) * (uint64_t)(height > 0 ? height : 1)) * 4);   // two sets of accumulator lines + 9 maps
}

int gs_mapping_loss(int32_t width, int32_t height, const float* im, const float* gt_im, const float* depth,
                    const float* depth_sq, const float* gt_depth, float w_im, float w_depth, float* losses, float* dL_dim,
                    float* dL_ddepth, void* scratch, int64_t persistent_call, gs_stream_t stream)
{
    if (width <= 0 || height <= 0 || !im || !gt_im || !depth || !gt_depth || !losses || !dL_dim || !dL_ddepth || !scratch || persistent_call < 0)
        return fail(GS_EINVAL, "gs_mapping_loss: bad argument");
    hipError_t e = gs::launch_mapping_loss(width, height, im, gt_im, depth, depth_sq, gt_depth, w_im, w_depth, losses, dL_dim,
                                           dL_ddepth, (float*)scratch, persistent_call, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_mapping_loss: %s", hipGetErrorString(e));
    return GS_OK;
}

uint64_t gs_compact_scratch_bytes(int64_t n) { return align_up(gs::compact_scratch_bytes(n > 0 ? n : 1)); }

int gs_compact_index(int64_t n, const uint8_t* keep, uint32_t* src_index, uint32_t* d_count, void* scratch, gs_stream_t stream)
{
    if (n < 0 || !d_count || !scratch || (n > 0 && (!keep || !src_index))) return fail(GS_EINVAL, "gs_compact_index: bad argument");
    if (n >= (int64_t)1 << 32) return fail(GS_ECAPACITY, "gs_compact_index: more than 2^32 rows");
    hipError_t e = gs::launch_compact_index(n, keep, src_index, d_count, scratch, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_compact_index: %s", hipGetErrorString(e));
    return GS_OK;
}

uint64_t gs_compact3_scratch_bytes(int64_t n) { return align_up(gs::compact3_scratch_bytes(n > 0 ? n : 1)); }

int gs_compact_index3(int64_t n, const uint8_t* keep_a, const uint8_t* keep_b, const uint8_t* keep_c, int32_t repeat_c, uint32_t* src_index,
                      uint32_t* d_counts, void* scratch, gs_stream_t stream)
{
    if (n < 0 || repeat_c < 1 || !d_counts || !scratch || (n > 0 && (!keep_a || !keep_b || !keep_c || !src_index)))
        return fail(GS_EINVAL, "gs_compact_index3: bad argument");
    if (n * (2 + (int64_t)repeat_c) >= (int64_t)1 << 32) return fail(GS_ECAPACITY, "gs_compact_index3: more than 2^32 rows");
    hipError_t e = gs::launch_compact_index3(n, keep_a, keep_b, keep_c, repeat_c, src_index, d_counts, scratch, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_compact_index3: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_gather_rows(int64_t n_out, int32_t row_floats, const uint32_t* src_index, const float* src, float* dst, gs_stream_t stream)
{
    if (n_out < 0 || row_floats <= 0 || (n_out > 0 && (!src_index || !src || !dst))) return fail(GS_EINVAL, "gs_gather_rows: bad argument");
    hipError_t e = gs::launch_gather_rows(n_out, row_floats, src_index, src, dst, n_out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_gather_rows: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_gather_rows_zero_tail(int64_t n_out, int64_t n_copy, int32_t row_floats, const uint32_t* src_index, const float* src, float* dst,
                             gs_stream_t stream)
{
    if (n_out < 0 || n_copy < 0 || n_copy > n_out || row_floats <= 0 || (n_out > 0 && !dst) || (n_copy > 0 && (!src_index || !src)))
        return fail(GS_EINVAL, "gs_gather_rows_zero_tail: bad argument");
    hipError_t e = gs::launch_gather_rows(n_out, row_floats, src_index, src, dst, n_copy, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_gather_rows_zero_tail: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_densify_classify(int32_t N, int32_t scale_dim, const float* log_scales, const float* logit_opacities, const float* grad_accum,
                        const float* denom, const float* d_scene_radius, float grad_thresh, float opacity_thresh, int32_t remove_big,
                        int32_t num_to_split_into, uint8_t* keep_orig, uint8_t* keep_clone, uint8_t* keep_child, uint8_t* split_mask,
                        gs_stream_t stream)
{
    if (N < 0 || (scale_dim != 1 && scale_dim != 3) || num_to_split_into < 1 || !d_scene_radius ||
        (N > 0 && (!log_scales || !logit_opacities || !keep_orig)) || ((grad_accum == nullptr) != (denom == nullptr)))
        return fail(GS_EINVAL, "gs_densify_classify: bad argument");
    hipError_t e = gs::launch_densify_classify(N, scale_dim, log_scales, logit_opacities, grad_accum, denom, d_scene_radius, grad_thresh,
                                               opacity_thresh, remove_big, num_to_split_into, keep_orig, keep_clone, keep_child, split_mask,
                                               (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_densify_classify: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_densify_children(int32_t n_child, int32_t scale_dim, int32_t num_to_split_into, const float* unnorm_rotations, const float* samples,
                        uint64_t seed, float* means3D, float* log_scales, gs_stream_t stream)
{
    if (n_child < 0 || (scale_dim != 1 && scale_dim != 3) || num_to_split_into < 1 ||
        (n_child > 0 && (!unnorm_rotations || !means3D || !log_scales)))
        return fail(GS_EINVAL, "gs_densify_children: bad argument");
    hipError_t e = gs::launch_densify_children(n_child, scale_dim, num_to_split_into, unnorm_rotations, samples, seed, means3D, log_scales,
                                               (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_densify_children: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_visibility_stats(int32_t P, const int32_t* radii, uint8_t* seen, float* max_2D_radius, gs_stream_t stream)
{
    if (P < 0 || (P > 0 && !radii)) return fail(GS_EINVAL, "gs_visibility_stats: bad argument");
    hipError_t e = gs::launch_visibility_stats(P, radii, seen, max_2D_radius, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_visibility_stats: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_accumulate_grad2d(int32_t P, const float* means2D_grad, const uint8_t* seen, float* grad_accum, float* denom, gs_stream_t stream)
{
    if (P < 0 || (P > 0 && (!means2D_grad || !seen || !grad_accum || !denom))) return fail(GS_EINVAL, "gs_accumulate_grad2d: bad argument");
    hipError_t e = gs::launch_accumulate_grad2d(P, means2D_grad, seen, grad_accum, denom, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_accumulate_grad2d: %s", hipGetErrorString(e));
    return GS_OK;
}

uint64_t gs_grow_scratch_bytes(int32_t width, int32_t height)
{
    return align_up(gs::grow_scratch_bytes((int64_t)(width > 0 ? width : 1) * (height > 0 ? height : 1)));
}

int gs_grow_gaussians(int32_t width, int32_t height, const float* render_depth, const float* silhouette, const float* gt_depth,
                      const float* color, const float* h_intrinsics4, const float* h_c2w12, float sil_thres, int32_t isotropic,
                      float* out_means3D, float* out_rgb_colors, float* out_unnorm_rotations, float* out_logit_opacities,
                      float* out_log_scales, uint32_t* d_counts, void* scratch, gs_stream_t stream)
{
    if (width <= 0 || height <= 0 || !render_depth || !silhouette || !gt_depth || !color || !h_intrinsics4 || !h_c2w12 ||
        !out_means3D || !out_rgb_colors || !out_unnorm_rotations || !out_logit_opacities || !out_log_scales || !d_counts || !scratch)
        return fail(GS_EINVAL, "gs_grow_gaussians: bad argument");
    hipError_t e = gs::launch_grow(width, height, render_depth, silhouette, gt_depth, color, h_intrinsics4, h_c2w12, sil_thres,
                                   isotropic != 0, out_means3D, out_rgb_colors, out_unnorm_rotations, out_logit_opacities,
                                   out_log_scales, d_counts, scratch, (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_grow_gaussians: %s", hipGetErrorString(e));
    return GS_OK;
}

int gs_keyframe_overlap(int32_t n_pts, const float* pts_world, int32_t n_keyframes, const float* w2c, const float* h_intrinsics9,
                        int32_t width, int32_t height, int32_t edge, uint32_t* counts, gs_stream_t stream)
{
    if (n_pts < 0 || n_keyframes < 0 || width <= 0 || height <= 0 || !h_intrinsics9 ||
        (n_keyframes > 0 && (!w2c || !counts)) || (n_pts > 0 && !pts_world))
        return fail(GS_EINVAL, "gs_keyframe_overlap: bad argument");
    hipError_t e = gs::launch_keyframe_overlap(n_pts, pts_world, n_keyframes, w2c, h_intrinsics9, width, height, edge, counts,
                                               (hipStream_t)stream);
    if (e != hipSuccess) return fail(GS_ELAUNCH, "gs_keyframe_overlap: %s", hipGetErrorString(e));
    return GS_OK;
}

}  // extern "C"
