// torch_frontend.cpp -- the drop-in GaussianRasterizer call as a C++ autograd function above the C ABI (include/gsplat_hip.h).
//
// What the reference reaches through `diff_gaussian_rasterization._C.rasterize_gaussians{,_backward}` (call sites
// src/mapper/splatam/splatam.py:208,212,338,430,431): the reference's binding is a C++ torch extension too.  This file is PLUMBING -- tensors in,
// device pointers to the library, tensors out -- and exists for one reason: at the reference's own frame sizes (256 x 256, ~200 k Gaussians) a
// forward + backward is ~175 us of GPU work, while the same plumbing written in Python (activesplat_amd/rasterizer.py: _RasterizeGaussians, kept as the
// readable twin and for the raw-parameter / capture paths) costs the host ~240 us -- torch's own floor for a Python autograd.Function of this
// signature is ~100 us (scripts/exp/host_breakdown.py).  In C++ the forward is ONE workspace allocation + the outputs, the backward runs on the
// autograd engine's thread without the interpreter.  Same library calls in the same order as the Python twin; no kernel lives here.
//
// Built by activesplat_amd/_frontend.py (g++, no device code) into activesplat_amd/_gs_frontend.so, in-tree.
#include <torch/extension.h>

#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime.h>

#include <map>
#include <atomic>
#include <mutex>
#include <tuple>
#include <unordered_map>

#include "../../include/gsplat_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc)
{
    if (rc != GS_OK) throw std::runtime_error(gs_last_error());
}

struct Layouts {
    GsGeomLayout gl;
    GsImageLayout il;
    uint64_t scratch;
};

std::mutex g_mu;
std::map<std::tuple<int, int, int>, Layouts> g_layouts;

Layouts layouts(int P, int W, int H)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(P, W, H);
    auto it = g_layouts.find(key);
    if (it != g_layouts.end()) return it->second;
    Layouts L;
    check(gs_geom_layout(P, W, H, &L.gl));
    check(gs_image_layout(W, H, &L.il));
    L.scratch = gs_backward_scratch_bytes(P);
    if (g_layouts.size() >= 64) g_layouts.erase(g_layouts.begin());      // P changes with every densify / growth step
    g_layouts[key] = L;
    return L;
}

// pinned host counter pair (D, largest tile list) + the event behind the counting kernels, one per (thread, device, stream): two forwards in
// flight (another stream of a keyframe batch, the visualiser thread of the reference's threaded layout) must never share one
struct PerStream {
    uint32_t* h_counts = nullptr;
    hipEvent_t ev = nullptr;
};
thread_local std::unordered_map<uint64_t, PerStream> tl_streams;

PerStream& per_stream(int device, hipStream_t st)
{
    const uint64_t key = ((uint64_t)(uintptr_t)st) ^ ((uint64_t)device << 56);
    auto it = tl_streams.find(key);
    if (it != tl_streams.end()) return it->second;
    if (tl_streams.size() >= 64) {                                        // short-lived streams: do not grow without bound
        auto v = tl_streams.begin();
        // a forward still in flight on the evicted stream writes these pinned counters from its counting kernels: wait for the event recorded
        // behind them before the memory goes back
        if (v->second.ev) (void)hipEventSynchronize(v->second.ev);
        if (v->second.h_counts) (void)hipHostFree(v->second.h_counts);
        if (v->second.ev) (void)hipEventDestroy(v->second.ev);
        tl_streams.erase(v);
    }
    PerStream ps;
    if (hipHostMalloc((void**)&ps.h_counts, 64, hipHostMallocMapped) != hipSuccess) throw std::runtime_error("frontend: hipHostMalloc of the counter pair failed");
    ps.h_counts[0] = ps.h_counts[1] = 0;
    if (hipEventCreateWithFlags(&ps.ev, hipEventDisableTiming) != hipSuccess) throw std::runtime_error("frontend: hipEventCreate failed");
    return tl_streams[key] = ps;
}

std::atomic<uint32_t*> g_status{nullptr};      // (double-checked under g_mu: the pointer itself is published with release / read with acquire)

void poll_async_status()
{
    uint32_t* w = g_status.load(std::memory_order_acquire);
    if (!w) {
        std::lock_guard<std::mutex> lk(g_mu);
        w = g_status.load(std::memory_order_relaxed);
        if (!w) {
            check(gs_async_status_word(&w));
            g_status.store(w, std::memory_order_release);
        }
    }
    if (*reinterpret_cast<volatile uint32_t*>(w)) {
        check(gs_set_backward_chain(1, -1));
        check(gs_async_status_clear());           // (synchronises: no walker of an old launch can raise the words again behind the clear)
        *reinterpret_cast<volatile uint32_t*>(w) = 0;
        throw std::runtime_error("activesplat_amd: a chained backward walk timed out waiting for the piece in front of it -- the gradients of the previous "
                                 "backward on this process are invalid (NaN); optimiser steps enqueued behind it were SKIPPED by their kernels (parameters and "
                                 "moments untouched, step counters one ahead).  Chained walks are now off (gs_set_backward_chain(1, -1)); render that frame again.");
    }
}

// the library takes contiguous fp32 tensors on the device with 16-byte aligned base pointers
Tensor f32(const Tensor& t, const at::Device& dev)
{
    if (!t.defined()) return t;
    Tensor r = t;
    if (r.scalar_type() != at::kFloat || r.device() != dev) r = r.to(dev, at::kFloat);
    if (!r.is_contiguous()) r = r.contiguous();
    if (reinterpret_cast<uintptr_t>(r.data_ptr()) % 16) r = r.clone();
    return r;
}

inline const float* fp(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline uint64_t up256(uint64_t v) { return (v + 255) / 256 * 256; }

// what a forward tells its Python caller besides the tensors (read right after the call, same thread)
thread_local int64_t tl_D = 0, tl_max_tile = 0;
thread_local int tl_hit = 0;          // 1: the optimistic launch stood; 0: rendered with exact sizes (no guess, or a miss)
// rasterizer.capture() (tests, bench.py): the state buffers of the forward and where the pieces sit in the workspace
thread_local bool tl_want_state = false;
thread_local std::vector<Tensor> tl_state;
thread_local std::vector<int64_t> tl_offsets;

struct Rasterize : public torch::autograd::Function<Rasterize> {
    // inputs 0..7 are the reference's tensors (undefined = not given), 8..11 the settings' tensors; the rest are plain numbers
    // (inputs that the caller may leave out travel as optionals: autograd's bookkeeping asks every TENSOR argument for its device)
    static variable_list forward(AutogradContext* ctx, Tensor means3D, std::optional<Tensor> means2D_, std::optional<Tensor> shs_, std::optional<Tensor> colors_,
                                 Tensor opacities, std::optional<Tensor> scales_, std::optional<Tensor> rotations_, std::optional<Tensor> cov3D_, Tensor bg,
                                 Tensor view, Tensor proj, Tensor campos, int64_t W, int64_t H, double tanfovx, double tanfovy, double scale_modifier,
                                 int64_t sh_degree, bool fused, int64_t guess_d, int64_t guess_tile, bool want_bwd)
    {
        const at::Device dev = means3D.device();
        const int P = (int)means3D.size(0);
        auto un = [](const std::optional<Tensor>& t) { return t.has_value() ? *t : Tensor(); };
        Tensor shs = un(shs_), colors = un(colors_), scales = un(scales_), rotations = un(rotations_), cov3D = un(cov3D_);
        means3D = f32(means3D, dev); shs = f32(shs, dev); colors = f32(colors, dev); opacities = f32(opacities, dev);
        scales = f32(scales, dev); rotations = f32(rotations, dev); cov3D = f32(cov3D, dev);
        bg = f32(bg, dev).reshape({-1}); view = f32(view, dev).reshape({-1}); proj = f32(proj, dev).reshape({-1}); campos = f32(campos, dev).reshape({-1});
        if (view.numel() != 16 || proj.numel() != 16 || bg.numel() != 3)
            throw std::runtime_error("GaussianRasterizationSettings: viewmatrix/projmatrix must hold 16 values, bg 3");
        const int M = shs.defined() ? (int)shs.size(1) : 0;
        GsCamera cam{(int32_t)W, (int32_t)H, (int32_t)sh_degree, (int32_t)M, (float)tanfovx, (float)tanfovy, (float)scale_modifier, 0,
                     fp(bg), fp(view), fp(proj), fp(campos)};
        const hipStream_t st = c10::hip::getCurrentHIPStream(dev.index()).stream();
        PerStream& ps = per_stream(dev.index(), st);
        const Layouts L = layouts(P, (int)W, (int)H);
        const bool need_scratch = want_bwd && P > 0;

        // ONE allocation for everything the backward needs again: geom state | image state | device counters | gradient records | (with a capacity
        // guess) point list.  The binning workspace (8 bytes per tile instance: 39 MB at 2 M Gaussians) is scratch of this forward only and goes back
        // to the allocator when it returns
        GsBinLayout bl{};
        bool optimistic = guess_d > 0;
        if (optimistic) {
            check(gs_bin_layout(guess_d, (uint32_t)guess_tile, (int32_t)W, (int32_t)H, &bl));
            // (never optimistic where the capacities would CHOOSE the algorithm: the segmented compositing of few-tile images is switched on, and its
            // segment count set, by the tile-list bound -- an inflated guess would make the image depend on the call history)
            optimistic = bl.path == GS_SORT_TILE_LDS && bl.segments <= 1;
        }
        const uint64_t o_geom = 0, o_image = up256(L.gl.total_bytes), o_num = o_image + up256(L.il.total_bytes), o_scratch = o_num + 256;
        const uint64_t o_plist = o_scratch + (need_scratch ? up256(L.scratch) : 0);
        const uint64_t total = o_plist + (optimistic ? up256(4 * (uint64_t)std::max<int64_t>(guess_d, 1)) : 0);
        auto bytes = at::TensorOptions().dtype(at::kByte).device(dev);
        Tensor ws = at::empty({(int64_t)total}, bytes);
        uint8_t* base = ws.data_ptr<uint8_t>();
        Tensor bin1 = optimistic ? at::empty({(int64_t)bl.total_bytes}, bytes) : Tensor();
        auto f_opts = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor radii = at::empty({P}, at::TensorOptions().dtype(at::kInt).device(dev));
        Tensor color = at::empty({3, H, W}, f_opts), depth = at::empty({1, H, W}, f_opts), opacity = at::empty({1, H, W}, f_opts);
        Tensor depth_sq = fused ? at::empty({1, H, W}, f_opts) : Tensor();

        check(gs_preprocess_forward(&cam, P, fp(means3D), fp(shs), fp(colors), fp(opacities), fp(scales), fp(rotations), fp(cov3D), radii.data_ptr<int32_t>(),
                                    base + o_geom, base + o_image, (uint32_t*)(base + o_num), ps.h_counts, want_bwd ? 1 : 0, st));
        // Optimistic launch (see rasterizer.py): the render is enqueued BEHIND the counting kernels with the previous frame's capacities, and only
        // then does the host wait for the two counters; a frame that outgrew them is rendered again with exact sizes.
        if (optimistic) {
            if (hipEventRecord(ps.ev, st) != hipSuccess) throw std::runtime_error("frontend: hipEventRecord failed");
            check(gs_render_forward(&cam, P, guess_d, (uint32_t)guess_tile, base + o_geom, bin1.data_ptr(), (uint32_t*)(base + o_plist), base + o_image,
                                    color.data_ptr<float>(), depth.data_ptr<float>(), opacity.data_ptr<float>(), fused ? depth_sq.data_ptr<float>() : nullptr,
                                    need_scratch ? base + o_scratch : nullptr, st));
            if (hipEventSynchronize(ps.ev) != hipSuccess) throw std::runtime_error("frontend: hipEventSynchronize failed");
        } else if (guess_d > 0) {
            if (hipEventRecord(ps.ev, st) != hipSuccess || hipEventSynchronize(ps.ev) != hipSuccess) throw std::runtime_error("frontend: event wait failed");
        } else {
            if (hipStreamSynchronize(st) != hipSuccess) throw std::runtime_error("frontend: hipStreamSynchronize failed");
        }
        const int64_t D = (int64_t)ps.h_counts[0], max_tile = (int64_t)ps.h_counts[1];
        Tensor bin2, plist2;
        const bool hit = optimistic && D <= guess_d && max_tile <= guess_tile;
        if (!hit) {
            GsBinLayout b2{};
            check(gs_bin_layout(D, (uint32_t)max_tile, (int32_t)W, (int32_t)H, &b2));
            bin2 = at::empty({(int64_t)b2.total_bytes}, bytes);
            plist2 = at::empty({std::max<int64_t>(D, 1)}, at::TensorOptions().dtype(at::kInt).device(dev));
            check(gs_render_forward(&cam, P, D, (uint32_t)max_tile, base + o_geom, bin2.data_ptr(), (uint32_t*)plist2.data_ptr<int32_t>(), base + o_image,
                                    color.data_ptr<float>(), depth.data_ptr<float>(), opacity.data_ptr<float>(), fused ? depth_sq.data_ptr<float>() : nullptr,
                                    need_scratch ? base + o_scratch : nullptr, st));
        }
        tl_D = D; tl_max_tile = max_tile; tl_hit = hit ? 1 : 0;
        if (tl_want_state) {
            tl_state = {ws, hit ? bin1 : bin2, plist2};
            tl_offsets = {(int64_t)o_geom, (int64_t)L.gl.total_bytes, (int64_t)o_image, (int64_t)L.il.total_bytes, (int64_t)o_plist, std::max<int64_t>(guess_d, 1)};
        }

        ctx->save_for_backward({means3D, shs, colors, scales, rotations, cov3D, radii, ws, plist2, bg, view, proj, campos});
        auto& sd = ctx->saved_data;
        sd["W"] = W; sd["H"] = H; sd["tanfovx"] = tanfovx; sd["tanfovy"] = tanfovy; sd["mod"] = scale_modifier; sd["sh_degree"] = sh_degree;
        sd["fused"] = fused; sd["D"] = D; sd["want_bwd"] = want_bwd; sd["clean"] = need_scratch;
        sd["o_image"] = (int64_t)o_image; sd["o_scratch"] = (int64_t)(need_scratch ? o_scratch : 0); sd["o_plist"] = (int64_t)o_plist;
        ctx->set_materialize_grads(false);
        if (fused) {
            ctx->mark_non_differentiable({radii, opacity, depth_sq});
            return {color, radii, depth, opacity, depth_sq};
        }
        ctx->mark_non_differentiable({radii, depth, opacity});
        return {color, radii, depth, opacity};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const Tensor &means3D = saved[0], &shs = saved[1], &colors = saved[2], &scales = saved[3], &rotations = saved[4], &cov3D = saved[5];
        const Tensor &radii = saved[6], &ws = saved[7], &plist2 = saved[8], &bg = saved[9], &view = saved[10], &proj = saved[11], &campos = saved[12];
        auto& sd = ctx->saved_data;
        const int64_t W = sd["W"].toInt(), H = sd["H"].toInt(), D = sd["D"].toInt();
        const bool fused = sd["fused"].toBool();
        const at::Device dev = means3D.device();
        const int P = (int)means3D.size(0), M = shs.defined() ? (int)shs.size(1) : 0;
        GsCamera cam{(int32_t)W, (int32_t)H, (int32_t)sd["sh_degree"].toInt(), (int32_t)M, (float)sd["tanfovx"].toDouble(), (float)sd["tanfovy"].toDouble(),
                     (float)sd["mod"].toDouble(), 0, fp(bg), fp(view), fp(proj), fp(campos)};
        auto f_opts = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor g_color = grads[0].defined() ? f32(grads[0], dev) : at::zeros({3, H, W}, f_opts);
        Tensor g_depth = (fused && grads.size() > 2 && grads[2].defined()) ? f32(grads[2], dev) : Tensor();
        uint8_t* base = ws.data_ptr<uint8_t>();
        const int64_t o_scratch = sd["o_scratch"].toInt();
        Tensor fresh;
        void* scratch = o_scratch ? (void*)(base + o_scratch) : nullptr;
        const bool clean = sd["clean"].toBool();
        if (!scratch) {                                        // (a forward that did not expect a backward: a fresh buffer, the library fills it)
            fresh = at::empty({(int64_t)gs_backward_scratch_bytes(P)}, at::TensorOptions().dtype(at::kByte).device(dev));
            scratch = fresh.data_ptr();
        }
        sd["clean"] = false;                                   // a second backward through the same graph finds the records dirty
        const uint32_t* plist = plist2.defined() ? (const uint32_t*)plist2.data_ptr<int32_t>() : (const uint32_t*)(base + sd["o_plist"].toInt());
        Tensor d_m2d = at::empty({P, 3}, f_opts), d_m3d = at::empty({P, 3}, f_opts), d_op = at::empty({P, 1}, f_opts);
        Tensor d_col = colors.defined() ? at::empty({P, 3}, f_opts) : Tensor(), d_shs = shs.defined() ? at::empty({P, M, 3}, f_opts) : Tensor();
        Tensor d_sc = scales.defined() ? at::empty({P, 3}, f_opts) : Tensor(), d_rot = rotations.defined() ? at::empty({P, 4}, f_opts) : Tensor();
        Tensor d_cov = cov3D.defined() ? at::empty({P, 6}, f_opts) : Tensor();
        const hipStream_t st = c10::hip::getCurrentHIPStream(dev.index()).stream();
        auto mp = [](Tensor& t) -> float* { return t.defined() ? t.data_ptr<float>() : nullptr; };
        check(gs_render_backward(&cam, P, D, fp(means3D), fp(shs), fp(colors), fp(scales), fp(rotations), fp(cov3D), radii.data_ptr<int32_t>(), base,
                                 plist, base + sd["o_image"].toInt(), fp(g_color), fp(g_depth), mp(d_m2d), mp(d_m3d), mp(d_op), mp(d_col), mp(d_shs), mp(d_sc),
                                 mp(d_rot), mp(d_cov), scratch, clean ? 1 : 0, sd["want_bwd"].toBool() ? 1 : 0, st));
        Tensor none;
        return {d_m3d, d_m2d, d_shs, d_col, d_op, d_sc, d_rot, d_cov, none, none, none, none, none, none, none, none, none, none, none, none, none, none};
    }
};

// -> (outputs, D, largest tile list, optimistic launch stood, [workspace, exact binning workspace, exact point list] and their offsets when want_state)
std::tuple<std::vector<Tensor>, int64_t, int64_t, bool, std::vector<Tensor>, std::vector<int64_t>> rasterize(
    const Tensor& means3D, const std::optional<Tensor>& means2D, const std::optional<Tensor>& shs, const std::optional<Tensor>& colors,
    const Tensor& opacities, const std::optional<Tensor>& scales, const std::optional<Tensor>& rotations, const std::optional<Tensor>& cov3D,
    const Tensor& bg, const Tensor& view, const Tensor& proj, const Tensor& campos, int64_t W, int64_t H, double tanfovx, double tanfovy, double scale_modifier,
    int64_t sh_degree, bool fused, int64_t guess_d, int64_t guess_tile, bool want_state)
{
    if (!means3D.is_cuda()) throw std::runtime_error("activesplat_amd rasteriser needs ROCm device tensors (no CPU fallback)");
    poll_async_status();
    auto opt = [](const std::optional<Tensor>& t) { return (t.has_value() && t->defined()) ? t : std::optional<Tensor>(); };
    const std::optional<Tensor> m2d = opt(means2D), s = opt(shs), c = opt(colors), sc = opt(scales), r = opt(rotations), cv = opt(cov3D);
    bool want_bwd = false;
    if (at::GradMode::is_enabled()) {
        want_bwd = means3D.requires_grad() || opacities.requires_grad();
        for (const std::optional<Tensor>* t : {&m2d, &s, &c, &sc, &r, &cv})
            if (t->has_value() && (*t)->requires_grad()) want_bwd = true;
    }
    c10::hip::HIPGuard guard(means3D.device().index());
    tl_want_state = want_state;
    tl_state.clear(); tl_offsets.clear();
    auto out = Rasterize::apply(means3D, m2d, s, c, opacities, sc, r, cv, bg, view, proj, campos, W, H, tanfovx, tanfovy, scale_modifier, sh_degree, fused,
                                guess_d, guess_tile, want_bwd);
    tl_want_state = false;
    std::vector<Tensor> state;
    state.swap(tl_state);
    return {out, tl_D, tl_max_tile, tl_hit != 0, state, tl_offsets};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "C++ autograd front-end of the drop-in GaussianRasterizer call (plumbing above libgsplat_hip.so's C ABI)";
    m.def("rasterize", &rasterize, pybind11::call_guard<pybind11::gil_scoped_release>());
    m.def("abi_version", []() { return (int)gs_abi_version(); });
}
