/*
 * gsplat_hip.h -- C ABI of libgsplat_hip.so, the MI355X (gfx950) native replacement for the
 * `_C.rasterize_gaussians{,_backward}` extension entry points that ActiveSplat reaches through
 * `diff_gaussian_rasterization.GaussianRasterizer`.
 *
 * Reference interface replaced (the implementation itself is an un-vendored submodule, reference
 * .gitmodules:1-3; these are the call sites that define the contract):
 *   settings      : src/mapper/splatam/utils/recon_helpers.py:14-27 (12-field settings tuple)
 *   forward call  : src/mapper/splatam/splatam.py:208,212,338,430,431 and
 *                   src/mapper/splatam/utils/slam_helpers.py:131-138 (keyword tensors)
 *   backward      : autograd of `color` incl. means2D.grad, src/mapper/splatam/splatam.py:207-209,
 *                   src/mapper/splatam/__init__.py:470, utils/slam_external.py:100-108
 *   Adam          : src/mapper/splatam/splatam.py:118-124, src/mapper/splatam/__init__.py:479-480
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_ (pinned/pageable host);
 *   - all float tensors are contiguous fp32, layouts exactly those of the Python API
 *     (means3D [P,3], shs [P,M,3], colors_precomp [P,3], opacities [P,1], scales [P,3],
 *      rotations [P,4] (w,x,y,z), cov3D_precomp [P,6]); base pointers 16-byte aligned;
 *   - `stream` is a hipStream_t; every entry point only enqueues work on it (no implicit sync);
 *   - return value 0 = success, otherwise a GS_E* code; gs_last_error() gives the message;
 *   - no torch types.  Process-wide state: the last-error string (thread-local), the host-mapped status word (gs_async_status_word) and the
 *     development knobs gs_set_sort_path, gs_set_forward_segments, gs_set_half_quadrants, gs_set_backward_chain[_tickets|_polls],
 *     gs_set_backward_segments (defaults: automatic path choice, segments on, few-tile kernels up to 256 tiles, three chained pieces above 768 tiles).
 *     The knobs are atomics that every launch reads ONCE: set from another thread (the reference runs a visualiser thread next to the mapper) a
 *     new value takes effect at a launch boundary of the other threads, never inside one launch's decisions; a forward and the backward that
 *     consumes its state tolerate a change in between (the forward clears the hand-over flags and records its "recorded" word whatever the
 *     knobs say).  The gs_profile_* event log (off by default) is not thread-safe: one profiling thread at a time.
 *
 * Call sequence for one forward:
 *     gs_preprocess_forward(...)            // per-Gaussian stage + tile counting; writes the counts
 *                                           // {D, max instances per tile} to d_counts / h_counts
 *     <caller synchronises `stream`, reads the counts, allocates binning workspace + point_list>
 *     gs_render_forward(...)                // scatter into tile segments, per-tile depth sort, alpha-blend
 * and for the backward:
 *     gs_render_backward(...)               // per-pixel replay -> per-Gaussian grads -> input grads
 * Optional: gs_render_forward may run optimistically right behind gs_preprocess_forward with capacities from the previous
 * frame (see its comment), and may zero-fill the backward's scratch as a side job (backward_scratch).
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK 0
#define GS_EINVAL 1      /* bad argument (null pointer, bad size, both/neither of an xor pair) */
#define GS_ELAUNCH 2     /* a HIP launch / runtime call failed */
#define GS_ECAPACITY 3   /* workspace too small for this call */

#define GS_TILE 16       /* tile edge in pixels (spec constant) */
#define GS_GEOM_FLOATS 12

typedef void* gs_stream_t;

/* Mirror of GaussianRasterizationSettings (recon_helpers.py:14-27). bg/viewmatrix/projmatrix/campos
 * stay device tensors exactly as the reference hands them over (transposed [1,4,4] matrices). */
typedef struct GsCamera {
    int32_t image_width;
    int32_t image_height;
    int32_t sh_degree;      /* active SH degree 0..3 (only read when shs != NULL) */
    int32_t sh_coeffs;      /* M = coefficients per Gaussian stored in shs; 0 when colors_precomp */
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t num_views;      /* 0/1: one view.  V > 1: multi-view ATLAS (forward only, see gs_atlas_layout): viewmatrix / projmatrix /
                             * campos hold V consecutive blocks; image_width is the width of ONE view */
    const float* bg;          /* [3] */
    const float* viewmatrix;  /* [16] = w2c^T row-major */
    const float* projmatrix;  /* [16] = (P w2c)^T row-major */
    const float* campos;      /* [3] */
} GsCamera;

/* Byte offsets of the arrays inside the caller-allocated state buffers, so that tests (and other
 * hosts) can inspect the integer artefacts without any private header. */
typedef struct GsGeomLayout {
    uint64_t total_bytes;
    uint64_t geom;          /* float [P][12]: x, y, conic_a, conic_b, conic_c, opacity, r, g, b, depth, ext_x, ext_y */
    uint64_t rect;          /* uint32 [P][2]: (xmin | xmax<<16), (ymin | ymax<<16) in tiles */
    uint64_t tiles_touched; /* uint32 [P] */
    uint64_t offsets;       /* uint32 [P]  inclusive scan of tiles_touched (radix-sort path only) */
    uint64_t block_sums;    /* uint32 [ceil(P/256)+1] exclusive scan of per-block tile counts */
    uint64_t clamped;       /* uint8  [P][4]: SH colour clamp flags (r,g,b,pad) */
    uint64_t tile_total;    /* uint32 [tiles]: instances per tile */
    uint64_t tile_base;     /* uint32 [ceil(P/4096)][tiles]: slice reserved by each binning workgroup */
    uint64_t sh_jac;        /* float [P][10]: 3x3 d(rgb)/d(view direction) of SH inputs + the colour clamp flags in the tenth word, written when
                             * want_backward (the `clamped` array is then not written) */
    uint64_t depth_bits;    /* uint32 [P]: float bits of the view-space depth (binning key) */
} GsGeomLayout;

typedef struct GsImageLayout {
    uint64_t total_bytes;
    uint64_t ranges;        /* uint32 [tiles][2] : [start,end) into point_list */
    uint64_t final_T;       /* float  [H*W] */
    uint64_t n_contrib;     /* uint32 [H*W] : 1-based position in the tile list of the last contributor */
    uint64_t split_state;   /* images of at most 256 tiles: float [12][5][H*W] + [4][H*W] + 1 word, per-pixel running state at the list
                             * positions 128 * 2^k, where the backward may cut a quadrant's walk in two */
} GsImageLayout;

#define GS_SORT_AUTO 0
#define GS_SORT_TILE_LDS 1   /* count/scatter into tile segments + per-tile sort in LDS / registers (bucket sort for lists of <= 5632 keys, else bitonic
                              * runs + merges); <= 8192 tiles */
#define GS_SORT_RADIX 2      /* duplicate with 64-bit keys + device radix sort (any image size) */

typedef struct GsBinLayout {
    uint64_t total_bytes;
    uint64_t path;          /* GS_SORT_TILE_LDS or GS_SORT_RADIX: what gs_render_forward will run */
    uint64_t pairs;         /* TILE_LDS: uint64 [D] : (float_bits(view depth) << 32) | Gaussian index, tile-major -- scratch: as the scatter pass
                             * left them for tile lists of <= 5632 keys (the bucket sort writes point_list only), sorted in place for longer lists */
    uint64_t keys_unsorted; /* RADIX: uint64 [D] : (tile << 32) | float_bits(view depth) */
    uint64_t vals_unsorted; /* RADIX: uint32 [D] : Gaussian index */
    uint64_t keys_sorted;   /* RADIX: uint64 [D] */
    uint64_t sort_temp;     /* RADIX: scratch of the sort */
    uint64_t segments;      /* > 1: gs_render_forward composites every tile list in this many parallel segments (few tiles, long lists) */
    uint64_t seg_T;         /* float [tiles][segments][256]: per-segment transmittance of the segmented forward, then uint32 [tiles][4] flag words */
    uint64_t pairs_alt;     /* TILE_LDS, tile lists longer than 16384: uint64 [D], the second buffer of the pairwise merge passes */
} GsBinLayout;

/* Multi-view atlas (the planner's look-around: V small views of one map, src/mapper/splatam/__init__.py:707-736): with
 * GsCamera.num_views = V > 1 the per-Gaussian stage runs once over V x P virtual Gaussians (view-major, each view's rows
 * padded to whole 256-row blocks) and binning, sorting and blending see ONE image of V view slots side by side, every slot
 * padded to whole tiles.  Sizes for the caller: virtual_P = rows of radii[] and the P to pass to gs_geom_layout; atlas_width =
 * the width to pass to the layout functions and the row length of the output images; view v occupies columns
 * [v * view_stride, v * view_stride + view_width).  gs_preprocess_forward / gs_render_forward still take the INPUT count P. */
int gs_atlas_layout(int32_t P, int32_t view_width, int32_t num_views, int32_t* virtual_P, int32_t* atlas_width, int32_t* view_stride);
int gs_geom_layout(int32_t P, int32_t width, int32_t height, GsGeomLayout* out);
int gs_image_layout(int32_t width, int32_t height, GsImageLayout* out);
int gs_bin_layout(int64_t D, uint32_t max_tile_instances, int32_t width, int32_t height, GsBinLayout* out);
/* Force a binning path (GS_SORT_*; default GS_SORT_AUTO picks TILE_LDS whenever the image has at most 8192 tiles). */
int gs_set_sort_path(int32_t path);
/* Segmented compositing of long tile lists in images of few tiles (GsBinLayout.segments > 1): on by default; 0 switches it
 * off (every list is then walked by one workgroup, bit-reproducible forward). */
int gs_set_forward_segments(int32_t on);
/* Images of at most max_tiles tiles (default 256; 0 = never) get twice the wavefronts: 256 tiles x 4 quadrants are one wavefront per SIMD of
 * an MI355X.  The forward's wavefronts take half an 8 x 8 quadrant each (half their lanes idle; results unchanged); the backward walks every
 * tile list in up to three segments (gs_set_backward_segments), the front ones from per-pixel states the forward recorded at the cuts
 * (gradients agree to rounding). */
int gs_set_half_quadrants(int32_t max_tiles);
/* Images of more than min_tiles tiles (default 768 = more quadrants than the chip holds backward walkers; negative: the default; values
 * below 256 act as 256): every quadrant's backward walk is cut into `pieces` consecutive pieces (default 3, the maximum; 1 = one walker per
 * quadrant) that run as separate workgroups in dispatch order and hand the per-pixel running state on through the image workspace.  The
 * arithmetic per pixel is the same sequence either way (gradients differ only by the order of the atomic sums). */
int gs_set_backward_chain(int32_t pieces, int32_t min_tiles);
/* Chained walks, robustness.  Ordered tickets (gs_set_backward_chain_tickets(1); default OFF: one more dependent round trip per walker, measured
 * +2.5-5 % of the backward blend): a workgroup's place in the chain order is a ticket it draws when it starts, so the piece it waits for is by
 * construction already running or done -- no assumption about the order in which the hardware starts workgroups.  Off: the place is the
 * workgroup index, which the dispatcher hands out in order (a piece only ever waits for a lower index of its own XCD class).
 * gs_set_backward_chain_polls: bound of a piece's wait, in polls (-1: the default, 2^21 ~ 0.3 s; below -1: every piece gives up without
 * looking -- the tests' way to take the timeout path).
 * A wait that runs out sets bit 0 of the process' host-mapped status word and the walk continues with NaN state (NaN gradients for that
 * quadrant's Gaussians).  gs_async_status_word returns the word's HOST address (created on first call; call it once before the first
 * backward): the caller reads it -- a plain load -- before its next launch; non-zero = the previous chained backward's gradients are invalid:
 * gs_set_backward_chain(1, -1), gs_async_status_clear(), clear the word, and render again (activesplat_amd/rasterizer.py does exactly that).
 * The backward is asynchronous: the host learns of the event a render later, by which time an optimiser step on the NaN gradients would have
 * been enqueued (optimizer.step(), gs_adam_rows, or the Adam inside gs_render_backward_raw_adam -- the same launch).  The timed-out walker
 * therefore also sets a sticky word in DEVICE memory (created together with the host word) that every optimiser kernel of this library reads
 * first: while it is set gs_adam_step / gs_adam_step_multi / gs_adam_rows / gs_render_backward_raw_adam leave parameters and moments
 * untouched (gs_adam_rows still forwards the unstepped rows; the raw_adam backward writes zeros to dL_dmeans2D).  gs_async_status_clear
 * synchronises the device and clears that word: steps run again.  Step counters the host advanced for skipped steps stay advanced. */
int gs_set_backward_chain_tickets(int32_t on);
int gs_set_backward_chain_polls(int32_t polls);
int gs_async_status_word(uint32_t** host_word);
int gs_async_status_clear(void);
/* Introspection of the few-tile backward's cuts: the recorded list position nearest to `target` (0: none below it) and its level (-1: none) --
 * every 256th position up to 4096, then the powers of two up to 131072, nearest in ratio above 4096. */
int gs_recorded_cut(uint32_t target, uint32_t* nearest, int32_t* level);
/* Test / tuning knob: list segments (walkers) per quadrant in the backward blend of images of few tiles (at most 256): 3 (default: 3 x 256
 * tiles x 4 quadrants = the chip's 3072 walker slots), 2, or 1 (one walker per quadrant; the forward then records nothing).  The walkers
 * of a quadrant resume from the per-pixel state the forward recorded at every 256th list position up to 4096 and at the powers of two
 * beyond. */
int gs_set_backward_segments(int32_t segments);
/* bytes of the scratch gs_render_backward needs (per-Gaussian 2-D gradient records) */
uint64_t gs_backward_scratch_bytes(int32_t P);

const char* gs_last_error(void);
const char* gs_version(void);
/* Integer version of THIS binary interface: bumped whenever an entry point's argument list or a published record layout changes (e.g.
 * the seed argument of gs_densify_children, the 40-byte SH Jacobian record).  A host binding compares it with the GS_ABI_VERSION it was
 * written against before the first call, so that a stale prebuilt library fails at load time instead of misreading its arguments. */
#define GS_ABI_VERSION 10
int32_t gs_abi_version(void);

/* Optional per-stage timing (hipEvents recorded on the caller's stream around each stage's launches).
 * Off by default; bench.py switches it on for its roofline leg.  gs_profile_collect synchronises the
 * recorded events, sums elapsed ms and call counts per stage into arrays of gs_profile_stage_count()
 * entries, and clears the log. */
int gs_profile_enable(int32_t on);
int32_t gs_profile_stage_count(void);
const char* gs_profile_stage_name(int32_t stage);
int gs_profile_collect(float* ms_sum, int32_t* calls, int32_t n_stages);

/* Stage 1: per-Gaussian preprocess (view/projective transform, near cull, 3-D -> 2-D covariance, conic,
 * radius, tile rect, SH -> RGB), tile counting and the tile-range scan.  Writes radii[P] (0 = culled),
 * the geom state and the tile ranges (image state); d_counts[0] = D (number of tile instances),
 * d_counts[1] = largest per-tile instance count; if h_counts != NULL both also reach that host buffer
 * asynchronously -- stored by the scan kernel itself when h_counts is mapped pinned memory (hipHostMalloc), by an
 * async copy otherwise (read them after synchronising `stream` or an event recorded behind this call). Exactly one of shs/colors_precomp and
 * exactly one of (scales,rotations)/cov3D_precomp must be given.
 * want_backward != 0 with shs: the stage also stores, per Gaussian, the 3x3 block sum_k coef[k] (x) grad b_k(direction) in the geom
 * state, so that gs_render_backward (have_sh_jacobian = 1) does not read the coefficient rows a second time.
 * PAIRING RULE: want_backward and have_sh_jacobian go together.  With want_backward != 0 the colour clamp flags travel in the Jacobian
 * record and the separate `clamped` array is NOT written: a backward over that geom state must pass have_sh_jacobian = 1 (with 0 it
 * would read an unwritten `clamped` array); with want_backward = 0 the backward must pass have_sh_jacobian = 0. */
int gs_preprocess_forward(const GsCamera* cam, int32_t P,
                          const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, const float* rotations,
                          const float* cov3D_precomp,
                          int32_t* radii, void* geom_state, void* image_state, uint32_t* d_counts,
                          uint32_t* h_counts, int32_t want_backward, gs_stream_t stream);

/* Stage 2: bin the instances by tile and depth-sort every tile list, then front-to-back alpha blend.
 * out_color [3,H,W], out_depth [1,H,W] (sum z*alpha*T), out_opacity [1,H,W] (1 - T_final).
 * point_list [D] receives the (tile, depth, index)-sorted Gaussian ids (kept for the backward).
 * out_depth_sq [1,H,W] (nullable) additionally receives sum z^2*alpha*T: with out_depth and out_opacity these are
 * the three channels of the reference's second, [z, 1, z^2] raster pass (slam_helpers.py:196-249), produced by
 * the SAME pass as the colour.
 * D and max_tile_instances may be UPPER BOUNDS (the capacity the caller sized bin_state / point_list for) as long as
 * gs_bin_layout reports GS_SORT_TILE_LDS for them: no kernel reads or writes past D, so a host may enqueue this call
 * right behind gs_preprocess_forward with bounds from the previous frame, read h_counts afterwards, and simply call it
 * again with the exact counts in the rare frame where they exceed the bounds (the call is idempotent; a frame whose
 * true counts exceed the bounds it was launched with has unspecified outputs).
 * backward_scratch (nullable, gs_backward_scratch_bytes(P) bytes): the gradient records gs_render_backward accumulates into
 * are zero-filled by the blend workgroups of THIS call (stores riding along an arithmetic-bound kernel) instead of by a
 * separate fill in front of the backward; pass the same buffer to gs_render_backward with scratch_zeroed = 1. */
int gs_render_forward(const GsCamera* cam, int32_t P, int64_t D, uint32_t max_tile_instances,
                      void* geom_state, void* bin_state, uint32_t* point_list, void* image_state,
                      float* out_color, float* out_depth, float* out_opacity, float* out_depth_sq,
                      void* backward_scratch, gs_stream_t stream);

/* Backward of `out_color` (and, when dL_ddepth [1,H,W] is non-NULL, of `out_depth`) w.r.t. every input.  Any dL_d* output pointer may be NULL if that input
 * was not given (shs vs colors_precomp, scales/rotations vs cov3D_precomp).
 * dL_dmeans2D [P,3] receives the NDC-scaled screen-space gradient (x*0.5W, y*0.5H, 0).
 * scratch: gs_backward_scratch_bytes(P) bytes; scratch_zeroed != 0 promises that it is all zero (see gs_render_forward's
 * backward_scratch) -- the call leaves it dirty either way.  have_sh_jacobian != 0: geom_state comes from a
 * gs_preprocess_forward(want_backward = 1) call. */
int gs_render_backward(const GsCamera* cam, int32_t P, int64_t D,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* scales, const float* rotations, const float* cov3D_precomp,
                       const int32_t* radii, const void* geom_state, const uint32_t* point_list,
                       const void* image_state, const float* dL_dcolor, const float* dL_ddepth,
                       float* dL_dmeans2D, float* dL_dmeans3D, float* dL_dopacities,
                       float* dL_dcolors_precomp, float* dL_dshs, float* dL_dscales,
                       float* dL_drotations, float* dL_dcov3D, void* scratch, int32_t scratch_zeroed,
                       int32_t have_sh_jacobian, gs_stream_t stream);

/* Raw-parameter mode of the two calls above ("fused pre-activations", one step further than gs_activate_*): the inputs are
 * the mapper's PARAMETERS -- world-frame means, logit opacities, log scales ([P,3], or [P,1] when isotropic != 0), unnormalised quaternions --
 * and the frame transform + activations of slam_helpers.py:252-304,124-139 run inside the per-Gaussian kernels (h_pose7: HOST array
 * {qw,qx,qy,qz,tx,ty,tz} of the frame's relative w2c, as for gs_activate_forward).  Colours given or 16-coefficient SH rows; no precomputed
 * covariance; one view.  The backward returns the gradients w.r.t. the parameters; accumulate != 0: dL_dmeans3D, dL_dlogit_opacities,
 * dL_dlog_scales, dL_dunnorm_rotations and dL_dcolors_precomp are ADDED to what the buffers hold (rows of Gaussians that were not rendered stay
 * untouched) -- the gradient accumulation over the keyframes of a batch without autograd's `grad += new` passes.
 * max_2D_radius / seen (both nullable): the mapper's visibility statistics of this render, as gs_visibility_stats would leave them --
 * max_2D_radius[i] = max(max_2D_radius[i], radii[i]) in place, seen[i] = radii[i] > 0 -- written by the forward kernel itself. */
int gs_preprocess_forward_raw(const GsCamera* cam, int32_t P, const float* means3D, const float* shs, const float* colors_precomp,
                              const float* logit_opacities, const float* log_scales, const float* unnorm_rotations,
                              const float* h_pose7, int32_t isotropic, float* max_2D_radius, uint8_t* seen, int32_t* radii, void* geom_state,
                              void* image_state, uint32_t* d_counts, uint32_t* h_counts, int32_t want_backward, gs_stream_t stream);
int gs_render_backward_raw(const GsCamera* cam, int32_t P, int64_t D, const float* means3D, const float* shs, const float* colors_precomp,
                           const float* logit_opacities, const float* log_scales, const float* unnorm_rotations, const float* h_pose7,
                           int32_t isotropic, int32_t accumulate, const int32_t* radii, const void* geom_state, const uint32_t* point_list,
                           const void* image_state, const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans2D, float* dL_dmeans3D,
                           float* dL_dlogit_opacities, float* dL_dcolors_precomp, float* dL_dshs, float* dL_dlog_scales,
                           float* dL_dunnorm_rotations, void* scratch, int32_t scratch_zeroed, int32_t have_sh_jacobian, gs_stream_t stream);

/* Fused dense Adam step over one flat parameter tensor with torch.optim.Adam semantics
 * (non-amsgrad, no weight decay): splatam.py:118-124 uses betas (0.9,0.999), eps 1e-15.
 * `step` is the 1-based step count of this tensor AFTER the increment.  Hyper-parameters are doubles (as the
 * Python floats torch receives): 1-beta, the bias corrections and lr/(1-beta1^t) are formed in double and
 * rounded to fp32 once, exactly like torch's scalar handling. */
int gs_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                 double lr, double beta1, double beta2, double eps, int32_t step, gs_stream_t stream);

/* The same step for several parameter tensors in one launch (the mapper's five per-Gaussian groups; each keeps its
 * own lr and step counter exactly like torch's per-group state).  `tensors` is a HOST array. */
typedef struct GsAdamTensor {
    int64_t n;
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    double lr, beta1, beta2, eps;
    int32_t step;
    int32_t reserved;       /* padding; set to 0 */
} GsAdamTensor;
int gs_adam_step_multi(int32_t count, const GsAdamTensor* tensors, gs_stream_t stream);

/* gs_render_backward_raw with the optimiser step INSIDE (single-keyframe steps: the reference's mapping loop -- loss.backward() followed by
 * optimizer.step() on the same keyframe's gradients, src/mapper/splatam/__init__.py:470-480 -- and BASELINE configs[2]'s loop): the
 * per-Gaussian backward kernel applies Adam to the five per-Gaussian parameter tensors and their moments in place with the gradient it
 * forms, instead of writing gradient tensors that gs_adam_step_multi reads back one launch later.  Same arithmetic as gs_adam_step*, so the
 * parameters and moments afterwards are those of gs_render_backward_raw + gs_adam_step_multi, bit for bit.
 * adam5: HOST array of exactly five descriptors in the order {means3D, logit_opacities, log_scales, unnorm_rotations, colours | SH rows};
 * .param must be the very tensors passed as inputs (they are updated in place), .grad is ignored, .n = elements of the tensor
 * (3P, P, 3P | P, 4P, 3P | 48P), .step >= 1 is the step number this call performs.  Gaussians that were not rendered are stepped with a zero
 * gradient (their moments decay), as a dense optimiser does.  dL_dmeans2D is written as usual; no parameter gradient is written, no
 * accumulation.  SH rows need have_sh_jacobian = 1 (the kernel overwrites the rows it would otherwise read). */
int gs_render_backward_raw_adam(const GsCamera* cam, int32_t P, int64_t D, const float* means3D, const float* shs, const float* colors_precomp,
                                const float* logit_opacities, const float* log_scales, const float* unnorm_rotations, const float* h_pose7,
                                int32_t isotropic, const int32_t* radii, const void* geom_state, const uint32_t* point_list,
                                const void* image_state, const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans2D, void* scratch,
                                int32_t scratch_zeroed, int32_t have_sh_jacobian, const GsAdamTensor* adam5, gs_stream_t stream);

/* Keyframe-sharded optimiser step (activesplat_amd/parallel.py; SURVEY.md section 8e -- new capability, the reference steps one
 * keyframe on one GPU, src/mapper/splatam/__init__.py:450-480): glue between the K per-key tensors ([N, width] fp32, K <= 8,
 * sum of widths <= 64) and ONE flat [rows, G] buffer, G = sum of widths.  `tensors` is a HOST array.
 *   gs_pack_columns  : flat[r][.] = the keys' `grad` rows (NULL grad = zeros) for r < n, zeros for n <= r < n_padded
 *                      (the send buffer of the reduce-scatter / all-reduce);
 *   gs_adam_rows     : Adam (arithmetic of gs_adam_step) on rows [row_lo, row_lo + n_valid) of every key, gradient from
 *                      grad_shard[r - row_lo][.]; the updated rows are also written to out_shard (may be NULL), rows
 *                      n_valid..n_rows-1 of it zero (the send buffer of the all-gather);
 *   gs_unpack_columns: the keys' `param` rows r < n = flat[r][.]. */
typedef struct GsRowTensor {
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    const float* grad;
    double lr, beta1, beta2, eps;
    int32_t width;
    int32_t step;
} GsRowTensor;
int gs_pack_columns(int32_t count, const GsRowTensor* tensors, int64_t n, int64_t n_padded, float* flat, gs_stream_t stream);
int gs_adam_rows(int32_t count, const GsRowTensor* tensors, int64_t row_lo, int64_t n_valid, int64_t n_rows, const float* grad_shard,
                 float* out_shard, gs_stream_t stream);
int gs_unpack_columns(int32_t count, const GsRowTensor* tensors, int64_t n, const float* flat, gs_stream_t stream);

/* Fused frame transform + activations (replaces transform_to_frame + transformed_params2rendervar,
 * src/mapper/splatam/utils/slam_helpers.py:252-304,124-139).  h_pose7 is a HOST array {qw,qx,qy,qz,tx,ty,tz}: the
 * normalised camera quaternion and translation of the relative w2c.  isotropic != 0: log_scales is [P,1] (tiled to 3,
 * rotations only normalised).  The backward takes the gradients w.r.t. the four outputs (any may be NULL = zero). */
int gs_activate_forward(int32_t P, int32_t isotropic, const float* h_pose7, const float* means3D, const float* unnorm_rotations,
                        const float* logit_opacities, const float* log_scales, float* out_means3D, float* out_rotations,
                        float* out_opacities, float* out_scales, gs_stream_t stream);
int gs_activate_backward(int32_t P, int32_t isotropic, const float* h_pose7, const float* unnorm_rotations,
                         const float* out_opacities, const float* out_scales, const float* g_means3D, const float* g_rotations,
                         const float* g_opacities, const float* g_scales, float* d_means3D, float* d_unnorm_rotations,
                         float* d_logit_opacities, float* d_log_scales, gs_stream_t stream);
/* The same, ADDED to what the four d_* buffers hold: the gradient accumulation over the keyframes of a batch (autograd's `grad += new`
 * passes) inside the kernel that produces the gradient. */
int gs_activate_backward_accumulate(int32_t P, int32_t isotropic, const float* h_pose7, const float* unnorm_rotations,
                                    const float* out_opacities, const float* out_scales, const float* g_means3D, const float* g_rotations,
                                    const float* g_opacities, const float* g_scales, float* d_means3D, float* d_unnorm_rotations,
                                    float* d_logit_opacities, float* d_log_scales, gs_stream_t stream);

/* Fused mapping loss, forward AND backward (replaces src/mapper/splatam/splatam.py:213-249 + the SSIM of
 * utils/slam_external.py:54-97 and their autograd):
 *   loss = w_depth * mean_{gt_depth>0, finite} |gt_depth - depth| + w_im * (0.8 * mean|im - gt_im| + 0.2 * (1 - SSIM))
 * im, gt_im [3,H,W]; depth, gt_depth [1,H,W]; depth_sq [1,H,W] nullable (only its NaN-ness enters the mask).
 * Writes losses[4] = {loss, weighted image term, weighted depth term, loss again} (device), dL_dim [3,H,W],
 * dL_ddepth [1,H,W]. */
/* persistent_call = 0: `scratch` is any buffer of gs_mapping_loss_scratch_bytes bytes (its accumulators are cleared by a memset in front of the
 * two kernels).  persistent_call = k >= 1: the k-th call (k counts up by one) on a scratch that its owner zeroed ONCE and keeps for this
 * stream: the call accumulates in one of two accumulator sets and its second kernel zeroes the other for call k + 1 -- no memset launch. */
uint64_t gs_mapping_loss_scratch_bytes(int32_t width, int32_t height);
int gs_mapping_loss(int32_t width, int32_t height, const float* im, const float* gt_im, const float* depth,
                    const float* depth_sq, const float* gt_depth, float w_im, float w_depth, float* losses,
                    float* dL_dim, float* dL_ddepth, void* scratch, int64_t persistent_call, gs_stream_t stream);


/* Stream compaction for prune / densify surgery (replaces the boolean-mask gathers of
 * src/mapper/splatam/utils/slam_external.py:143-164 remove_points and the torch.cat appends of
 * :126-140): gs_compact_index turns a keep mask [n] (uint8) into the ascending list of kept row indices
 * src_index[0..count) and writes count to *d_count; gs_gather_rows then copies dst[r][:] = src[src_index[r]][:]
 * for a tensor of `row_floats` floats per row (one index build serves params, Adam moments and statistics;
 * an index list with repeats implements clone / split). */
uint64_t gs_compact_scratch_bytes(int64_t n);
int gs_compact_index(int64_t n, const uint8_t* keep, uint32_t* src_index, uint32_t* d_count, void* scratch,
                     gs_stream_t stream);
int gs_gather_rows(int64_t n_out, int32_t row_floats, const uint32_t* src_index, const float* src, float* dst,
                   gs_stream_t stream);
/* The same gather for rows [0, n_copy); rows [n_copy, n_out) of dst are zero-filled by the same launch (Adam moments of appended
 * Gaussians, slam_external.py:131-134). */
int gs_gather_rows_zero_tail(int64_t n_out, int64_t n_copy, int32_t row_floats, const uint32_t* src_index, const float* src,
                             float* dst, gs_stream_t stream);

/* Densify / prune decisions of slam_external.py:171-247 in ONE launch.  Per Gaussian i of the N present before the event:
 *   keep_orig[i]  = 1 if it stays (not split, not culled)
 *   keep_clone[i] = 1 if it is cloned AND the clone survives the cull        (nullable; densify only)
 *   keep_child[i] = 1 if it is split AND its children survive the cull       (nullable; densify only)
 *   split_mask[i] = 1 if it is split (before the cull; indexes injected samples)   (nullable)
 * grad_accum/denom NULL = prune only (slam_external.py:171-192: keep_orig = not culled).  d_scene_radius: DEVICE pointer to
 * variables['scene_radius'] (no host read-back).  scale_dim = 1 (isotropic) or 3.  remove_big: the `iter >= remove_big_after`
 * clause (scale > 0.1 scene_radius). */
int gs_densify_classify(int32_t N, int32_t scale_dim, const float* log_scales, const float* logit_opacities,
                        const float* grad_accum, const float* denom, const float* d_scene_radius, float grad_thresh,
                        float opacity_thresh, int32_t remove_big, int32_t num_to_split_into, uint8_t* keep_orig,
                        uint8_t* keep_clone, uint8_t* keep_child, uint8_t* split_mask, gs_stream_t stream);
/* Split children (a contiguous block of rows): means3D += R(normalised unnorm_rotation) * sample, log_scale = log(exp(log_scale) /
 * (0.8 n)) in place (slam_external.py:224-230).  samples [n_child,3] are the N(0, scale) offsets (slam_external.py:221-224:
 * torch.normal(0, the parent's scale)); samples == NULL: they are drawn inside the kernel by a counter-based generator
 * (splitmix64 of (seed, child row, draw) -> Box-Muller; per-axis scales when scale_dim = 3): no sample tensor, no extra launches. */
int gs_densify_children(int32_t n_child, int32_t scale_dim, int32_t num_to_split_into, const float* unnorm_rotations,
                        const float* samples, uint64_t seed, float* means3D, float* log_scales, gs_stream_t stream);
/* The densify event's ONE index list from its three masks in one count / scan / write sequence:
 *   src_index = [ rows with keep_a | rows with keep_b | repeat_c blocks of the rows with keep_c ]   (each ascending)
 * d_counts[0..2] = the three totals (device; the host reads them once, afterwards, to size the new tensors).  src_index must hold
 * n * (2 + repeat_c) entries in the worst case.  Replaces three gs_compact_index passes, a torch.cat and a repeat. */
uint64_t gs_compact3_scratch_bytes(int64_t n);
int gs_compact_index3(int64_t n, const uint8_t* keep_a, const uint8_t* keep_b, const uint8_t* keep_c, int32_t repeat_c,
                      uint32_t* src_index, uint32_t* d_counts, void* scratch, gs_stream_t stream);

/* Densification statistics, one launch each.
 * gs_visibility_stats : seen[i] = radii[i] > 0 (uint8, nullable); max_2D_radius[i] = max(max_2D_radius[i], radii[i]) (nullable)
 *                       -- src/mapper/splatam/splatam.py:296-298.
 * gs_accumulate_grad2d: for seen rows, grad_accum += ||means2D_grad[i,:2]||, denom += 1; means2D_grad is [P,3]
 *                       -- src/mapper/splatam/utils/slam_external.py:100-108. */
int gs_visibility_stats(int32_t P, const int32_t* radii, uint8_t* seen, float* max_2D_radius, gs_stream_t stream);
int gs_accumulate_grad2d(int32_t P, const float* means2D_grad, const uint8_t* seen, float* grad_accum, float* denom,
                         gs_stream_t stream);

/* Map growth (replaces add_new_gaussians, src/mapper/splatam/splatam.py:332-379, with get_pointcloud :25-75 and
 * initialize_new_params :304-329).  render_depth / silhouette / gt_depth are [H*W] device images, color is [3,H*W];
 * h_intrinsics4 = HOST {fx,fy,cx,cy}; h_c2w12 = HOST row-major 3x4 camera-to-world of the frame.  Outputs must hold H*W
 * rows (the worst case); d_counts[0] = pixels flagged non-present BEFORE the valid-depth mask (the reference enters its
 * append branch, which also resets the densification statistics, iff this is > 0), d_counts[1] = rows written, in
 * row-major pixel order.  log_scales is [rows,1] when isotropic != 0, else [rows,3]. */
uint64_t gs_grow_scratch_bytes(int32_t width, int32_t height);
int gs_grow_gaussians(int32_t width, int32_t height, const float* render_depth, const float* silhouette,
                      const float* gt_depth, const float* color, const float* h_intrinsics4, const float* h_c2w12,
                      float sil_thres, int32_t isotropic, float* out_means3D, float* out_rgb_colors,
                      float* out_unnorm_rotations, float* out_logit_opacities, float* out_log_scales,
                      uint32_t* d_counts, void* scratch, gs_stream_t stream);

/* Keyframe overlap scores (the loop of keyframe_selection_overlap, src/mapper/splatam/utils/keyframe_selection.py:62-86):
 * counts[k] = number of the n_pts world points [n_pts,3] that project into keyframe k (row-major 4x4 w2c at
 * w2c[16k]) inside the image shrunk by `edge` pixels, with positive depth.  h_intrinsics9 = HOST row-major 3x3. */
int gs_keyframe_overlap(int32_t n_pts, const float* pts_world, int32_t n_keyframes, const float* w2c,
                        const float* h_intrinsics9, int32_t width, int32_t height, int32_t edge, uint32_t* counts,
                        gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
