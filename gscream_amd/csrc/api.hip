// api.hip -- the extern "C" surface declared in include/gsraster.h (host code only).
//
// Each entry validates its arguments, carves the caller-allocated workspaces, enqueues the stage
// kernels on the caller's stream and reports failures as an int code + thread-local message.
// Orchestration replaces CudaRasterizer::Rasterizer::{forward,backward,visible_filter,
// position2D_filter,markVisible} (DGR rasterizer_impl.cu:141-153,199-347,350-530,536-643).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "gsr_common.h"

static thread_local char g_err[512] = "";

static int gsr_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define GSR_HIP(expr, what)                                                                                     \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return gsr_fail(GSR_ERR_HIP, "%s: %s (%s)", what, hipGetErrorString(e_), #expr); \
    } while (0)

// ---- optional per-stage timing with HIP events on the caller's stream (bench / profiling only) ----
#include <mutex>
#include <vector>
static const char* const g_stage_names[GSR_NUM_STAGES] = { "preprocess", "count_scan", "scatter", "tile_sort",
                                                           "blend_forward", "blend_backward", "gauss_backward" };
struct GsrProfRec { int stage; hipEvent_t t0, t1; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static unsigned g_prof_mask = 0xffffffffu;
static unsigned g_prof_every = 1;                  // time every n-th invocation of a stage (gsr_profile_begin_sampled)
static unsigned g_prof_seen[GSR_NUM_STAGES] = {};  // invocations of each stage since gsr_profile_begin
static std::vector<GsrProfRec> g_prof;

struct GsrStageTimer {
    bool on = false;
    GsrProfRec rec{};
    hipStream_t stream;
    GsrStageTimer(int stage, hipStream_t s) : stream(s)
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_on || !((g_prof_mask >> stage) & 1u)) return;
        if (g_prof_seen[stage]++ % g_prof_every != 0u) return;  // sampled: the markers perturb the stream (a bubble on either side)
        if (hipEventCreate(&rec.t0) != hipSuccess || hipEventCreate(&rec.t1) != hipSuccess) return;
        rec.stage = stage;
        on = hipEventRecord(rec.t0, stream) == hipSuccess;
    }
    ~GsrStageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(rec.t1, stream);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(rec);
    }
};

// debug semantics of the reference's CHECK_CUDA (auxiliary.h:166-173): sync + check after a stage
#define GSR_STAGE(stage_id, expr, what)                              \
    do {                                                             \
        {                                                            \
            GsrStageTimer timer_(stage_id, stream);                  \
            GSR_HIP(expr, what);                                     \
        }                                                            \
        if (debug) GSR_HIP(hipStreamSynchronize(stream), what);      \
    } while (0)

extern "C" const char* gsr_stage_name(int stage)
{
    return (stage >= 0 && stage < GSR_NUM_STAGES) ? g_stage_names[stage] : "";
}

extern "C" int gsr_profile_begin_sampled(unsigned stage_mask, unsigned every)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = stage_mask ? stage_mask : 0xffffffffu;
    g_prof_every = every ? every : 1u;
    for (unsigned& n : g_prof_seen) n = 0u;
    for (auto& r : g_prof) { (void)hipEventDestroy(r.t0); (void)hipEventDestroy(r.t1); }
    g_prof.clear();
    g_prof_on = true;
    return GSR_OK;
}

extern "C" int gsr_profile_begin(unsigned stage_mask) { return gsr_profile_begin_sampled(stage_mask, 1u); }

extern "C" int gsr_profile_end(gsr_profile* out_host)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    if (!out_host) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "out_host is NULL");
    memset(out_host, 0, sizeof(*out_host));
    int rc = GSR_OK;
    for (auto& r : g_prof) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.t1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.t0, r.t1);
        if (e != hipSuccess) rc = gsr_fail(GSR_ERR_HIP, "profile: %s", hipGetErrorString(e));
        else { out_host->total_ms[r.stage] += ms; out_host->launches[r.stage] += 1; }
        (void)hipEventDestroy(r.t0); (void)hipEventDestroy(r.t1);
    }
    g_prof.clear();
    return rc;
}

extern "C" const char* gsr_version(void) { return "gsraster 0.4 (gfx950)"; }
extern "C" int gsr_abi_version(void) { return GSR_ABI_VERSION; }
extern "C" const char* gsr_last_error(void) { return g_err; }

extern "C" int gsr_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return gsr_fail(GSR_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" size_t gsr_geom_bytes(int P) { return gsr_carve_geom(nullptr, P).bytes; }
extern "C" size_t gsr_image_bytes(int P, int W, int H) { return gsr_carve_image(nullptr, P, W, H).bytes; }
extern "C" size_t gsr_binning_bytes(int R) { return gsr_carve_binning(nullptr, R).bytes; }
extern "C" size_t gsr_backward_scratch_bytes(int P, int num_slots)
{
    // slots[num_slots], 48 B each, + the per-Gaussian backward's list of heavy 64-Gaussian groups (count + indices)
    return gsr_align((size_t)(num_slots > 0 ? num_slots : 1) * GSR_SLOT_FLOATS * sizeof(float)) +
           gsr_align(((size_t)(P > 0 ? P : 1) / 64 + 18) * sizeof(uint32_t));
}

static bool gsr_partial_sort(const gsr_tuning* tuning) { return !(tuning && tuning->disable_partial_sort); }
static bool gsr_inference(const gsr_tuning* tuning) { return tuning && tuning->inference; }
static int gsr_forced_bands(const gsr_tuning* tuning) { return tuning ? tuning->scatter_bands : 0; }
// per-view walk depths of the forward blend (gsraster.h): the caller's array + whether it already holds a previous visit's
struct GsrWalkHint { uint32_t* depths; bool valid; };
static GsrWalkHint gsr_walk_hint(const gsr_tuning* tuning)
{
    GsrWalkHint h = { nullptr, false };
    if (tuning && tuning->walk_depths) { h.depths = reinterpret_cast<uint32_t*>((uintptr_t)tuning->walk_depths); h.valid = tuning->walk_depths_valid != 0; }
    return h;
}
// the occlusion cut-off works on the tile-cull masks and keeps its table in LDS beside the histogram
static bool gsr_occlusion(const gsr_tuning* tuning, int T)
{
    return tuning && tuning->occlusion_cut && !tuning->disable_tile_cull && T <= GSR_OCC_MAX_TILES;
}

static int gsr_make_cam(GsrCam& cam, int W, int H, const float* view_d, const float* proj_d, const float* campos_d,
                        float tan_fovx, float tan_fovy, float scale_modifier, hipStream_t stream)
{
    (void)stream;
    cam.view = view_d; cam.proj = proj_d; cam.campos = campos_d;
    cam.tan_fovx = tan_fovx; cam.tan_fovy = tan_fovy;
    cam.focal_y = H / (2.0f * tan_fovy);  // DGR rasterizer_impl.cu:226-227
    cam.focal_x = W / (2.0f * tan_fovx);
    cam.scale_modifier = scale_modifier;
    cam.W = W; cam.H = H;
    cam.gx = (W + GSR_TILE - 1) / GSR_TILE; cam.gy = (H + GSR_TILE - 1) / GSR_TILE;
    return GSR_OK;
}

static int gsr_check_dims(int P, int W, int H)
{
    if (P < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "P must be >= 0 (got %d)", P);
    if (W <= 0 || H <= 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "image size must be positive (got %dx%d)", W, H);
    const long gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    if (gx > 65535 || gy > 65535) return gsr_fail(GSR_ERR_UNSUPPORTED, "image too large: %ldx%ld tiles", gx, gy);
    // more than GSR_MAX_TILES_LDS tiles: supported through the global-counter binning fallback (binning.hip)
    if (gx * gy > (1L << 24)) return gsr_fail(GSR_ERR_UNSUPPORTED, "image too large: %ld tiles", gx * gy);
    return GSR_OK;
}

// Per host thread and device: the pinned, device-mapped words the tile-scan kernel writes {R, longest list} into (valid
// to read after the stream / event the caller waits on) and the event that marks "R is on the host".  A thread that
// alternates between devices finds each device's pair again (nothing is re-allocated on a switch); the pairs live as
// long as the thread and are released when it exits.
#define GSR_MAX_DEVICES 64
struct GsrThreadDeviceState {
    uint32_t* host[GSR_MAX_DEVICES] = {};
    uint32_t* dev[GSR_MAX_DEVICES] = {};
    hipEvent_t ev[GSR_MAX_DEVICES] = {};
    uint8_t heavy_mode[GSR_MAX_DEVICES] = {};   // per-Gaussian backward: 1 = launch the heavy-group kernel (gsr_heavy_groups_expected)
    uint8_t heavy_probe[GSR_MAX_DEVICES] = {};  // backwards to wait before the heavy kernel is launched again just to count the groups
    // partial sort of long tile lists, the bet and its feedback (gsr_partial_bet)
    uint32_t fwd_serial[GSR_MAX_DEVICES] = {};   // forwards of this thread on the device (what a resumed quadrant reports back)
    uint32_t ranoff_seen[GSR_MAX_DEVICES] = {};  // the last report acted upon
    uint8_t bet_hold[GSR_MAX_DEVICES] = {};      // forwards left during which lists up to GSR_SORT_CAP_SMALL are sorted completely at once
    ~GsrThreadDeviceState()
    {
        for (int d = 0; d < GSR_MAX_DEVICES; d++) {
            // kernels of this thread's last calls may still store into the block (the heavy-group hint is written without a
            // host wait): drain the owning device before the block goes away
            if (host[d] && hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
            if (host[d]) (void)hipHostFree(host[d]);
            if (ev[d]) (void)hipEventDestroy(ev[d]);
        }
    }
};
static thread_local GsrThreadDeviceState g_tds;

static int gsr_current_device(int* d)
{
    GSR_HIP(hipGetDevice(d), "hipGetDevice");
    if (*d < 0 || *d >= GSR_MAX_DEVICES) return gsr_fail(GSR_ERR_UNSUPPORTED, "device index %d out of range", *d);
    return GSR_OK;
}

// words of the pinned, device-mapped block (one per host thread and device): 0, 1 = {R, longest list} of stage 1, 4 = prefiltered
// trap, 8 = the per-Gaussian backward's report on heavy groups (0 = nothing new, 1 = the one-wave kernel met one, 2 + n = the
// heavy kernel was given n)
#define GSR_PINNED_HEAVY_SEEN 8
// 9 = the forward's fix-up pass: serial number of the last forward in which a quadrant ran off its tile's partially sorted prefix
#define GSR_PINNED_RANOFF 9
#define GSR_BET_HOLD 64  // forwards without the partial sort after a lost bet, then one forward probes again
#define GSR_HEAVY_MIN_GROUPS 192  // fewer heavy groups than this: the one-wave kernel does them itself (gauss_bwd.hip, launcher)
static int gsr_info_buffer(volatile uint32_t** host, uint32_t** dev)
{
    int d = 0;
    int rc = gsr_current_device(&d);
    if (rc) return rc;
    if (!g_tds.host[d]) {
        void* p = nullptr;
        GSR_HIP(hipHostMalloc(&p, 64, hipHostMallocMapped), "hipHostMalloc(info)");
        void* pd = nullptr;
        GSR_HIP(hipHostGetDevicePointer(&pd, p, 0), "hipHostGetDevicePointer(info)");
        memset(p, 0, 64);
        g_tds.host[d] = (uint32_t*)p; g_tds.dev[d] = (uint32_t*)pd;
    }
    *host = g_tds.host[d]; *dev = g_tds.dev[d];
    return GSR_OK;
}

// The state machine behind `heavy_expected` (see gsr_backward): report = the pinned word, mode / probe = the thread's state.
static bool gsr_heavy_groups_expected(uint32_t report, uint8_t* mode, uint8_t* probe)
{
    if (report >= 2u) {  // counted by the heavy kernel
        *mode = (report - 2u) >= GSR_HEAVY_MIN_GROUPS;
        *probe = (report - 2u) > 0u ? 64 : 0;  // none at all: the next sighting is counted at once
    } else if (report == 1u && !*mode) {  // met by the one-wave kernel while it was doing them itself
        if (*probe > 0) --*probe;
        else *mode = 1;
    }  // 0, or 1 in heavy mode (the heavy kernel of that call has not reported yet): nothing new
    return *mode != 0;
}

// Partial sort of long lists -- the bet and its feedback (round 6).  Lists beyond GSR_NEAR_CAP are depth-sorted only as far as the blend is
// expected to walk (binning.hip); a tile whose pixels are still blending at the end of that prefix is sorted completely and its
// quadrants resume in a SECOND forward-blend launch.  On frames that saturate the bet wins (large splats: tile sort 133 instead of
// 239 us).  On the frames of an initialised, untrained scene it loses on every long list -- nothing saturates -- and the second launch is
// ~140 us of a mostly idle GPU (init-state frame: 18 tiles of 2 049 - 3 238 entries, forward 2 x 169 us).  Which kind of frame a
// scene produces is a property of its training state, not of the view: a resumed quadrant stores the forward's serial number in a
// host-mapped word, and after such a report the next GSR_BET_HOLD forwards of this thread on this device sort lists that fit the
// small LDS class (<= GSR_SORT_CAP_SMALL entries) completely at once -- one launch, no fix-up -- then one forward bets again.  Longer
// lists always bet: a complete sort of 10 000-entry lists costs more than the fix-up.  Results never depend on any of this (the
// partial sort is bit-identical to a complete one, tests + tools/fuzz_parity.py).  -> the partial sort is worth betting on
static bool gsr_partial_bet(int devidx, const volatile uint32_t* pinned, int longest, bool longest_is_a_provision)
{
    const uint32_t rep = pinned[GSR_PINNED_RANOFF];
    if (rep != g_tds.ranoff_seen[devidx]) { g_tds.ranoff_seen[devidx] = rep; g_tds.bet_hold[devidx] = GSR_BET_HOLD; }
    const bool hold = g_tds.bet_hold[devidx] > 0;
    if (hold) --g_tds.bet_hold[devidx];
    // (a speculative caller provisions the longest list with slack over what its last frames had -- gsraster.h: 1.25 x: the call then
    // sorts completely up to GSR_SORT_CAP_SMALL and treats anything longer like any list beyond its provision: GSR_NEED_CAPACITY)
    const int limit = longest_is_a_provision ? GSR_SORT_CAP_SMALL + GSR_SORT_CAP_SMALL / 4 + 64 : GSR_SORT_CAP_SMALL;
    return !(hold && longest > GSR_NEAR_CAP && longest <= limit);
}

extern "C" void gsr_adaptive_reset(void)
{
    // forget what this thread's previous calls taught the launch heuristics (tests, gscream_amd.set_tuning): reports that are
    // still on their way count as seen
    for (int d = 0; d < GSR_MAX_DEVICES; d++) {
        g_tds.bet_hold[d] = 0;
        if (g_tds.host[d]) g_tds.ranoff_seen[d] = ((volatile uint32_t*)g_tds.host[d])[GSR_PINNED_RANOFF];
    }
}

static int gsr_info_event(hipEvent_t* ev)
{
    int d = 0;
    int rc = gsr_current_device(&d);
    if (rc) return rc;
    if (!g_tds.ev[d]) GSR_HIP(hipEventCreateWithFlags(&g_tds.ev[d], hipEventDisableTiming), "hipEventCreate");
    *ev = g_tds.ev[d];
    return GSR_OK;
}

// `prefiltered` (GaussianRasterizationSettings field 11).  The reference uses it for one thing: a point that fails the near
// plane although the caller declared the cloud pre-filtered prints "Point is filtered although prefiltered is set" and
// traps the kernel (DGR auxiliary.h:154-162) -- the process then dies at its next synchronisation.  Here the same
// condition is an ordinary error of the call, checked in debug mode (`debug=True` synchronises after every stage anyway);
// without debug the flag costs nothing and culled points are simply culled.
static int gsr_prefiltered_trap(int P, int prefiltered, int debug, const float* means3D, const float* viewmatrix, hipStream_t stream)
{
    if (!prefiltered || !debug || P <= 0) return GSR_OK;
    volatile uint32_t* host = nullptr;
    uint32_t* dev = nullptr;
    int rc = gsr_info_buffer(&host, &dev);
    if (rc) return rc;
    GSR_HIP(hipStreamSynchronize(stream), "prefiltered check");  // nothing in flight may still write the info words
    host[4] = 0u;
    GSR_HIP(gsr_launch_prefiltered_check(P, means3D, viewmatrix, dev + 4, stream), "prefiltered check");
    GSR_HIP(hipStreamSynchronize(stream), "prefiltered check");
    const uint32_t n = host[4];
    if (n) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "Point is filtered although prefiltered is set. This shouldn't happen! "
                                                    "(%u of %d points have view-space z <= 0.2)", n, P);
    return GSR_OK;
}

// Validates the stage-1 arguments and enqueues preprocess + counting + the 16-byte D2H copy of {R, max tile count}.
static int gsr_enqueue_stage1(int P, int D, int M, int W, int H, const float* means3D, const float* scales,
                              float scale_modifier, const float* rotations, const float* opacities,
                              const float* features, const float* shs, const float* cov3D_precomp,
                              const float* colors_precomp, const float* viewmatrix, const float* projmatrix,
                              const float* campos, float tan_fovx, float tan_fovy, void* geom_ws, void* image_ws,
                              int32_t* radii, volatile uint32_t** info_pinned, uint32_t** info_mapped_dev, bool defer_tile_scan,
                              const gsr_tuning* tuning, bool* ordered /* out (or NULL): the forward blend's dispatch order was computed
                              beside the column scan (gsr_tuning.walk_depths) */, int debug, hipStream_t stream)
{
    if (!means3D || !opacities || !features || !viewmatrix || !projmatrix || !geom_ws || !image_ws || !radii)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "a required pointer is NULL");
    if ((!scales || !rotations) == (cov3D_precomp == nullptr))  // DGR __init__.py:227-228
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "provide exactly one of scales+rotations or cov3D_precomp");
    if ((shs == nullptr) == (colors_precomp == nullptr))  // DGR __init__.py:224-225
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "provide exactly one of shs or colors_precomp");
    if (shs && (!campos || M <= 0)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "SH colours need campos and M > 0");
    GsrCam cam;
    int rc = gsr_make_cam(cam, W, H, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, scale_modifier, stream);
    if (rc) return rc;
    const GsrGeom geom = gsr_carve_geom(geom_ws, P);
    const GsrImage image = gsr_carve_image(image_ws, P, W, H);
    const int T = cam.gx * cam.gy;
    const bool occlusion = gsr_occlusion(tuning, T);
    if (occlusion) GSR_HIP(hipMemsetAsync(image.occ_mass, 0, (size_t)GSR_OCC_COPIES * T * GSR_OCC_BUCKETS * sizeof(uint32_t), stream), "clear occlusion masses");
    GSR_STAGE(GSR_STAGE_PREPROCESS, gsr_launch_preprocess(0, P, D, M, cam, means3D, scales, rotations, opacities, features, shs,
                                    cov3D_precomp, colors_precomp, &geom, radii, nullptr, nullptr,
                                    !(tuning && tuning->disable_tile_cull), occlusion ? image.occ_mass : nullptr, stream),
              "preprocess");
    // {R, longest list} reach the host through a pinned, device-mapped word pair the scan kernel stores into (one per
    // host thread and device, 8 bytes, kept for the life of the thread): no copy kernel between the scan and the scatter.
    uint32_t* mapped_dev = nullptr;
    rc = gsr_info_buffer(info_pinned, &mapped_dev);
    if (rc) return rc;
    if (info_mapped_dev) *info_mapped_dev = mapped_dev;
    const GsrWalkHint walk = gsr_walk_hint(tuning);
    GSR_STAGE(GSR_STAGE_COUNT_SCAN, gsr_launch_count(P, T, cam.gx, geom, image, mapped_dev, defer_tile_scan, occlusion,
                                                     (ordered && walk.valid) ? walk.depths : nullptr, ordered, stream), "tile count / scans");
    return GSR_OK;
}

static int gsr_publish_stage1(const uint32_t* info, gsr_stage1_result* out)
{
    if (info[0] > 0x7fffffffu) return gsr_fail(GSR_ERR_UNSUPPORTED, "instance count %u overflows int32", info[0]);
    out->num_rendered = (int32_t)info[0];
    out->max_tile_count = (int32_t)info[1];
    out->num_slots = (int32_t)info[0];  // one gradient slot per binned instance
    out->num_occluded = (int32_t)info[2];
    return GSR_OK;
}

extern "C" int gsr_forward_stage1(int P, int D, int M, int W, int H, const float* means3D, const float* scales,
                                  float scale_modifier, const float* rotations, const float* opacities,
                                  const float* features, const float* shs, const float* cov3D_precomp,
                                  const float* colors_precomp, const float* viewmatrix, const float* projmatrix,
                                  const float* campos, float tan_fovx, float tan_fovy, int prefiltered, void* geom_ws,
                                  void* image_ws, int32_t* radii, gsr_stage1_result* result_host,
                                  const gsr_tuning* tuning, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = gsr_check_dims(P, W, H);
    if (rc) return rc;
    if (!result_host) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "result_host is NULL");
    memset(result_host, 0, sizeof(*result_host));
    if (P == 0) return GSR_OK;  // DGR rasterize_points.cu:85
    if (prefiltered && debug && means3D && viewmatrix) {
        rc = gsr_prefiltered_trap(P, prefiltered, debug, means3D, viewmatrix, stream);
        if (rc) return rc;
    }
    volatile uint32_t* info = nullptr;  // pinned words the scan kernel writes {R, max} into
    rc = gsr_enqueue_stage1(P, D, M, W, H, means3D, scales, scale_modifier, rotations, opacities, features, shs,
                            cov3D_precomp, colors_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, geom_ws,
                            image_ws, radii, &info, nullptr, false, tuning, nullptr, debug, stream);
    if (rc) return rc;
    GSR_HIP(hipStreamSynchronize(stream), "read num_rendered");
    uint32_t got[3] = { info[0], info[1], info[2] };
    return gsr_publish_stage1(got, result_host);
}

// After a forward over partially sorted lists (binning.hip): tiles whose pixels were still blending at the end of their
// sorted prefix get a full sort and are blended again.  Nothing happens on the GPU beyond a few empty launches when no
// tile asked for it; not enqueued at all when no list was long enough to be partially sorted.
static int gsr_enqueue_fixup(int P, int W, int H, int capacity, int max_tile_count, const float* background, void* geom_ws,
                             void* image_ws, void* binning_ws, float* out_color, float* out_depth, float* out_feature,
                             bool inference, GsrWalkHint walk, int debug, hipStream_t stream)
{
    if (max_tile_count <= GSR_NEAR_CAP) return GSR_OK;
    // where a resumed quadrant reports the lost bet (gsr_partial_bet)
    volatile uint32_t* pinned = nullptr;
    uint32_t* pinned_dev = nullptr;
    int devidx = 0;
    int rc0 = gsr_info_buffer(&pinned, &pinned_dev);
    if (rc0) return rc0;
    rc0 = gsr_current_device(&devidx);
    if (rc0) return rc0;
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE, T = gx * gy;
    const GsrGeom geom = gsr_carve_geom(geom_ws, P);
    const GsrImage image = gsr_carve_image(image_ws, P, W, H);
    const GsrBinning bin = gsr_carve_binning(binning_ws, capacity);
    GSR_STAGE(GSR_STAGE_TILE_SORT, gsr_launch_sort_fixup(T, capacity, max_tile_count, image, bin, inference, stream), "tile sort (fix-up)");
    GSR_STAGE(GSR_STAGE_BLEND_FWD, gsr_launch_blend_forward(W, H, gx, T, background, geom, image, bin, out_color, out_depth,
                                                            out_feature, capacity, max_tile_count, true, inference, walk.depths, walk.valid, false,
                                                            pinned_dev + GSR_PINNED_RANOFF, g_tds.fwd_serial[devidx], stream),
              "forward blend (fix-up)");
    return GSR_OK;
}

static int gsr_enqueue_stage2(int P, int W, int H, int capacity, int max_tile_count, int partial /* gsr_launch_tile_sort */, bool inference, int forced_bands, const float* background,
                              void* geom_ws, void* image_ws, void* binning_ws, float* out_color, float* out_depth,
                              float* out_feature, GsrWalkHint walk, int debug, hipStream_t stream)
{
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE, T = gx * gy;
    const GsrGeom geom = gsr_carve_geom(geom_ws, P);
    const GsrImage image = gsr_carve_image(image_ws, P, W, H);
    const GsrBinning bin = gsr_carve_binning(binning_ws, capacity);
    // info[3] (quadrant walks that entered the second depth tier) is zeroed by the tile scan of stage 1 and counted up by the forward
    // blend: a stage 2 that REPEATS a speculative forward (GSR_NEED_CAPACITY) would count those walks twice and could flip the
    // backward's launch order on exactly the frames that outgrew their hint (results never depended on it)
    GSR_HIP(hipMemsetAsync(image.info + 3, 0, sizeof(uint32_t), stream), "reset the deep-walk counter");
    GSR_STAGE(GSR_STAGE_SCATTER, gsr_launch_scatter(P, T, gx, geom, image, bin, capacity, capacity, forced_bands, false, nullptr, inference, false, stream), "scatter");
    GSR_STAGE(GSR_STAGE_TILE_SORT, gsr_launch_tile_sort(T, capacity, max_tile_count, partial, false, inference, geom, image, bin, stream), "tile sort");
    GSR_STAGE(GSR_STAGE_BLEND_FWD, gsr_launch_blend_forward(W, H, gx, T, background, geom, image, bin, out_color, out_depth,
                                                            out_feature, capacity, max_tile_count, false, inference, walk.depths, walk.valid, false, nullptr, 0u, stream),
              "forward blend");
    // longest list known (two-stage form): the fix-up can follow at once; the one-call form enqueues it after the read-back
    if (partial == 1 && max_tile_count >= 0)
        return gsr_enqueue_fixup(P, W, H, capacity, max_tile_count, background, geom_ws, image_ws, binning_ws, out_color,
                                 out_depth, out_feature, inference, walk, debug, stream);
    return GSR_OK;
}

extern "C" int gsr_forward(int P, int D, int M, int W, int H, const float* means3D, const float* scales,
                           float scale_modifier, const float* rotations, const float* opacities, const float* features,
                           const float* shs, const float* cov3D_precomp, const float* colors_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                           float tan_fovy, int prefiltered, const float* background, void* geom_ws, void* image_ws,
                           void* binning_ws, int binning_capacity, int max_tile_count_hint, int32_t* radii,
                           float* out_color, float* out_depth, float* out_feature, gsr_stage1_result* result_host,
                           const gsr_tuning* tuning, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = gsr_check_dims(P, W, H);
    if (rc) return rc;
    if (!result_host) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "result_host is NULL");
    memset(result_host, 0, sizeof(*result_host));
    if (P == 0) return GSR_OK;
    if (prefiltered && debug && means3D && viewmatrix) {
        rc = gsr_prefiltered_trap(P, prefiltered, debug, means3D, viewmatrix, stream);
        if (rc) return rc;
    }
    if (!background || !binning_ws || !out_color || !out_depth || !out_feature || binning_capacity <= 0)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "a required pointer is NULL or the binning capacity is not positive");
    // One event per thread and device marks "R is on the host"; stage 2 is enqueued BEFORE we wait for it, so the GPU
    // never idles on the host round trip the reference pays at rasterizer_impl.cu:287.
    hipEvent_t ev = nullptr;
    rc = gsr_info_event(&ev);
    if (rc) return rc;
    volatile uint32_t* info = nullptr;
    uint32_t* info_dev = nullptr;
    // The tile scan (ranges, R, longest list) is folded into the scatter kernel here: nobody needs R before stage 2 is
    // enqueued, and the one-block launch cost as much as the whole column scan.  The event therefore follows the scatter.
    // the caller provisions the workspace with slack over what it expects (gsraster.h: "e.g. 1.25 x the previous frame's
    // num_rendered"): the scatter's staging (and its bands, binning.hip) is sized for the expectation, not for the provision
    const int expected_R = (int)(0.8 * (double)binning_capacity);
    const bool fold_tile_scan = true;
    bool ordered = false;
    rc = gsr_enqueue_stage1(P, D, M, W, H, means3D, scales, scale_modifier, rotations, opacities, features, shs,
                            cov3D_precomp, colors_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, geom_ws,
                            image_ws, radii, &info, &info_dev, fold_tile_scan, tuning, &ordered, debug, stream);
    if (rc) return rc;
    bool partial = gsr_partial_sort(tuning);
    int sort_mode = partial ? 1 : 0;
    const bool inference = gsr_inference(tuning);
    if (partial) {  // ... unless this thread's recent forwards lost the bet on lists like these (gsr_partial_bet)
        volatile uint32_t* pinned = nullptr;
        uint32_t* pinned_dev = nullptr;
        int devidx = 0;
        rc = gsr_info_buffer(&pinned, &pinned_dev);
        if (rc) return rc;
        rc = gsr_current_device(&devidx);
        if (rc) return rc;
        ++g_tds.fwd_serial[devidx];
        partial = gsr_partial_bet(devidx, pinned, max_tile_count_hint, true);
        if (!partial) {  // complete sorts, the few long lists in a launch of their own, nothing beyond the small LDS class
            sort_mode = 2;
            if (max_tile_count_hint > GSR_SORT_CAP_SMALL) max_tile_count_hint = GSR_SORT_CAP_SMALL;
        }
    }
    const GsrWalkHint walk = gsr_walk_hint(tuning);
    // (speculative: the sort variants are chosen from the hint; with partial sorting the hint only sizes the LDS of the
    // lists up to GSR_NEAR_CAP)
    {
        const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE, T = gx * gy;
        const GsrGeom geom = gsr_carve_geom(geom_ws, P);
        const GsrImage image = gsr_carve_image(image_ws, P, W, H);
        const GsrBinning bin = gsr_carve_binning(binning_ws, binning_capacity);
        const int hint = max_tile_count_hint > 0 ? max_tile_count_hint : -1;
        GSR_STAGE(GSR_STAGE_SCATTER, gsr_launch_scatter(P, T, gx, geom, image, bin, binning_capacity, expected_R, gsr_forced_bands(tuning), true, info_dev, inference,
                                                        gsr_occlusion(tuning, T), stream), "scatter");
        // (beyond the LDS tile limit stage 1 ran the stand-alone tile scan: the same words, written earlier)
        GSR_HIP(hipEventRecord(ev, stream), "record");
        // partial: lists beyond GSR_NEAR_CAP take the fixed-LDS prefix sort whatever the hint says
        GSR_STAGE(GSR_STAGE_TILE_SORT, gsr_launch_tile_sort(T, binning_capacity, hint, sort_mode, true, inference, geom, image, bin, stream),
                  "tile sort");
        GSR_STAGE(GSR_STAGE_BLEND_FWD, gsr_launch_blend_forward(W, H, gx, T, background, geom, image, bin, out_color, out_depth,
                                                                out_feature, binning_capacity, hint, false, inference, walk.depths, walk.valid, ordered, nullptr, 0u, stream),
                  "forward blend");
    }
    if (rc) return rc;
    GSR_HIP(hipEventSynchronize(ev), "read num_rendered");
    uint32_t got[3] = { info[0], info[1], info[2] };
    rc = gsr_publish_stage1(got, result_host);
    if (rc) return rc;
    // the guesses hold iff every list fitted the workspace AND the sort variants that were launched cover the longest list
    const bool ok = result_host->num_rendered <= binning_capacity &&
                    (max_tile_count_hint <= 0 || result_host->max_tile_count <= max_tile_count_hint ||
                     (partial && max_tile_count_hint > GSR_NEAR_CAP));  // lists beyond the cap do not depend on the hint
    // (strictly greater: gsr_launch_tile_sort starts the prefix-sort kernel only for a provision > GSR_NEAR_CAP)
    if (!ok) return GSR_NEED_CAPACITY;
    if (partial)
        return gsr_enqueue_fixup(P, W, H, binning_capacity, result_host->max_tile_count, background, geom_ws, image_ws,
                                 binning_ws, out_color, out_depth, out_feature, inference, walk, debug, stream);
    return GSR_OK;
}

extern "C" int gsr_forward_stage2(int P, int W, int H, int R, int max_tile_count, const float* background,
                                  void* geom_ws, void* image_ws, void* binning_ws, float* out_color, float* out_depth,
                                  float* out_feature, const gsr_tuning* tuning, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc = gsr_check_dims(P, W, H);
    if (rc) return rc;
    if (P == 0) return GSR_OK;  // outputs keep the caller's zero fill, like DGR rasterize_points.cu:69-85
    if (!background || !geom_ws || !image_ws || !binning_ws || !out_color || !out_depth || !out_feature)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "a required pointer is NULL");
    if (R < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "R must be >= 0");
    bool partial = gsr_partial_sort(tuning);
    if (partial) {  // (gsr_partial_bet: not when this thread's recent forwards lost the bet on lists like these)
        volatile uint32_t* pinned = nullptr;
        uint32_t* pinned_dev = nullptr;
        int devidx = 0;
        rc = gsr_info_buffer(&pinned, &pinned_dev);
        if (rc) return rc;
        rc = gsr_current_device(&devidx);
        if (rc) return rc;
        ++g_tds.fwd_serial[devidx];
        partial = gsr_partial_bet(devidx, pinned, max_tile_count, false);
    }
    return gsr_enqueue_stage2(P, W, H, R, max_tile_count, partial ? 1 : gsr_partial_sort(tuning) ? 2 : 0, gsr_inference(tuning), gsr_forced_bands(tuning), background, geom_ws, image_ws, binning_ws,
                              out_color, out_depth, out_feature, gsr_walk_hint(tuning), debug, stream);
}

extern "C" int gsr_backward(int P, int D, int M, int W, int H, int R, int binning_capacity, int max_tile_count, const float* background,
                            const float* means3D,
                            const int32_t* radii, const float* colors_precomp, const float* shs, const float* scales,
                            float scale_modifier, const float* rotations, const float* cov3D_precomp,
                            const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                            float tan_fovy, const float* dL_dout_color, const float* dL_dout_depth,
                            const float* dL_dout_feature, const void* geom_ws, const void* image_ws,
                            const void* binning_ws, void* scratch, float* dL_dmeans2D, float* dL_dcolors,
                            float* dL_dopacity, float* dL_dfeatures, float* dL_dmeans3D, float* dL_dcov3D,
                            float* dL_dsh, float* dL_dscales, float* dL_drotations, const gsr_tuning* tuning, int debug,
                            void* stream_)
{
    (void)colors_precomp;  // colours live in the geometry records written by the forward
    hipStream_t stream = (hipStream_t)stream_;
    int rc = gsr_check_dims(P, W, H);
    if (rc) return rc;
    if (P == 0) return GSR_OK;  // DGR rasterize_points.cu:172
    if (!background || !means3D || !radii || !viewmatrix || !projmatrix || !dL_dout_color || !geom_ws || !image_ws || !binning_ws || !scratch || !dL_dmeans2D || !dL_dcolors ||
        !dL_dopacity || !dL_dfeatures || !dL_dmeans3D)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "a required pointer is NULL");
    if ((!scales || !rotations) == (cov3D_precomp == nullptr))
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "provide exactly one of scales+rotations or cov3D_precomp");
    if (!cov3D_precomp && (!dL_dscales || !dL_drotations))
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "dL_dscales / dL_drotations are required with scales+rotations");
    if (shs && (!dL_dsh || !campos || M <= 0))
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "SH colours need dL_dsh, campos and M > 0");
    if (R < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "R must be >= 0");

    GsrCam cam;
    rc = gsr_make_cam(cam, W, H, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, scale_modifier, stream);
    if (rc) return rc;
    const int T = cam.gx * cam.gy;
    const GsrGeom geom = gsr_carve_geom(const_cast<void*>(geom_ws), P);
    const GsrImage image = gsr_carve_image(const_cast<void*>(image_ws), P, W, H);
    if (binning_capacity < R) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "binning_capacity < R");
    const GsrBinning bin = gsr_carve_binning(const_cast<void*>(binning_ws), binning_capacity);
    float* slots = (float*)scratch;
    // heavy[0] = number of heavy groups (zeroed by the backward blend, which always runs first), heavy[16..] = their indices
    uint32_t* heavy = (uint32_t*)((char*)scratch + gsr_align((size_t)(R > 0 ? R : 1) * GSR_SLOT_FLOATS * sizeof(float)));
    // written-slot flags: cleared by the forward's tile sort, set by the backward blend; the set of written slots is a function
    // of the forward state alone, so a second backward over the same forward state finds exactly the flags it would set
    uint8_t* slot_written = bin.slot_written;
    if (R > 0)
        GSR_STAGE(GSR_STAGE_BLEND_BWD, gsr_launch_blend_backward(W, H, cam.gx, T, background, geom, image, bin, dL_dout_color, dL_dout_depth,
                                            dL_dout_feature, slots, slot_written, heavy, max_tile_count > 0 ? max_tile_count : -1, stream),
                  "backward blend");
    else
        GSR_HIP(hipMemsetAsync(heavy, 0, 2 * sizeof(uint32_t), stream), "heavy-group counters");
    // Is the heavy-group kernel worth its launch?  Decided from what this thread's previous backwards (on this device) reported
    // through one word of the pinned block, read and cleared here without waiting for anything -- a hint only: both ways give the
    // same bits.  The heavy kernel reports how many groups it was given; below GSR_HEAVY_MIN_GROUPS the next calls go without it
    // (then only "met one" is known, so every 64th call launches it again to count).
    volatile uint32_t* pinned = nullptr;
    uint32_t* pinned_dev = nullptr;
    rc = gsr_info_buffer(&pinned, &pinned_dev);
    if (rc) return rc;
    int devidx = 0;
    rc = gsr_current_device(&devidx);
    if (rc) return rc;
    bool heavy_expected = gsr_heavy_groups_expected(pinned[GSR_PINNED_HEAVY_SEEN], &g_tds.heavy_mode[devidx], &g_tds.heavy_probe[devidx]);
    if (tuning && tuning->heavy_groups) heavy_expected = tuning->heavy_groups == 1;
    pinned[GSR_PINNED_HEAVY_SEEN] = 0u;
    GSR_STAGE(GSR_STAGE_GAUSS_BWD, gsr_launch_gauss_backward(P, D, M, cam, means3D, radii, shs, scales, rotations, cov3D_precomp, geom, slots, slot_written, heavy,
                                        pinned_dev + GSR_PINNED_HEAVY_SEEN, heavy_expected, R,
                                        dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dfeatures, dL_dmeans3D, dL_dcov3D,
                                        dL_dsh, dL_dscales, dL_drotations, stream),
              "per-Gaussian backward");
    return GSR_OK;
}

extern "C" int gsr_filter(int P, int W, int H, const float* means3D, const float* scales, float scale_modifier,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, float tan_fovx, float tan_fovy, int prefiltered, int32_t* radii,
                          float* px, float* py, int debug, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0 || W <= 0 || H <= 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "bad sizes P=%d W=%d H=%d", P, W, H);
    if (P == 0) return GSR_OK;
    if (!means3D || !viewmatrix || !projmatrix || !radii) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "a required pointer is NULL");
    {
        const int rc0 = gsr_prefiltered_trap(P, prefiltered, debug, means3D, viewmatrix, stream);
        if (rc0) return rc0;
    }
    if ((!scales || !rotations) == (cov3D_precomp == nullptr))
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "provide exactly one of scales+rotations or cov3D_precomp");
    if ((px == nullptr) != (py == nullptr)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "px and py go together");
    GsrCam cam;
    int rc = gsr_make_cam(cam, W, H, viewmatrix, projmatrix, nullptr, tan_fovx, tan_fovy, scale_modifier, stream);
    if (rc) return rc;
    GSR_STAGE(GSR_STAGE_PREPROCESS, gsr_launch_preprocess(px ? 2 : 1, P, 0, 0, cam, means3D, scales, rotations, nullptr, nullptr, nullptr,
                                    cov3D_precomp, nullptr, nullptr, radii, px, py, 0, nullptr, stream),
              "filter preprocess");
    return GSR_OK;
}

extern "C" int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                uint8_t* present, void* stream_)
{
    (void)projmatrix;  // unused by the reference's test as well (auxiliary.h:154 checks view-space z only)
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "P must be >= 0");
    if (P == 0) return GSR_OK;
    if (!means3D || !viewmatrix || !present) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "a required pointer is NULL");
    GSR_HIP(gsr_launch_mark_visible(P, means3D, viewmatrix, present, stream), "mark_visible");
    return GSR_OK;
}

// ---- image-space RGB loss (loss.hip) ----
extern "C" size_t gsr_loss_workspace_bytes(int C, int H, int W) { return gsl_workspace_bytes(C, H, W); }

static int gsr_check_loss_args(int C, int H, int W, const float* img, const float* gt, const void* workspace)
{
    if (C < 1 || H < 1 || W < 1) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "loss: C, H, W must be >= 1 (got %d, %d, %d)", C, H, W);
    if ((long long)C * H * W > 0x7fffffffLL) return gsr_fail(GSR_ERR_UNSUPPORTED, "loss: image too large");
    if (!img || !gt || !workspace) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "loss: a required pointer is NULL");
    return GSR_OK;
}

static int gsr_check_window(int window_size)
{
    if (window_size < 1 || window_size > 11 || (window_size & 1) == 0)
        return gsr_fail(GSR_ERR_UNSUPPORTED, "loss: window_size must be odd and in 1..11 (got %d)", window_size);
    return GSR_OK;
}

extern "C" int gsr_rgb_loss_forward_window(int C, int H, int W, const float* img, const float* gt, const float* weight,
                                           float a_l1, float a_ssim, int window_size, void* workspace, float* out3, int keep_state,
                                           void* stream)
{
    int rc = gsr_check_loss_args(C, H, W, img, gt, workspace);
    if (rc) return rc;
    rc = gsr_check_window(window_size);
    if (rc) return rc;
    if (!out3) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "loss: out3 is NULL");
    GSR_HIP(gsl_launch_forward(C, H, W, img, gt, weight, a_l1, a_ssim, workspace, out3, keep_state, window_size, (hipStream_t)stream),
            "rgb loss forward");
    return GSR_OK;
}

extern "C" int gsr_rgb_loss_backward_window(int C, int H, int W, const float* img, const float* gt, const float* weight,
                                            float a_l1, float a_ssim, int window_size, const void* workspace,
                                            const float* upstream, float* dL_dimg, void* stream)
{
    int rc = gsr_check_loss_args(C, H, W, img, gt, workspace);
    if (rc) return rc;
    rc = gsr_check_window(window_size);
    if (rc) return rc;
    if (!dL_dimg) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "loss: dL_dimg is NULL");
    GSR_HIP(gsl_launch_backward(C, H, W, img, gt, weight, a_l1, a_ssim, workspace, upstream, dL_dimg, window_size, (hipStream_t)stream),
            "rgb loss backward");
    return GSR_OK;
}

extern "C" int gsr_rgb_loss_forward(int C, int H, int W, const float* img, const float* gt, const float* weight,
                                    float a_l1, float a_ssim, void* workspace, float* out3, int keep_state, void* stream)
{
    return gsr_rgb_loss_forward_window(C, H, W, img, gt, weight, a_l1, a_ssim, 11, workspace, out3, keep_state, stream);
}

extern "C" int gsr_rgb_loss_backward(int C, int H, int W, const float* img, const float* gt, const float* weight,
                                     float a_l1, float a_ssim, const void* workspace, const float* upstream,
                                     float* dL_dimg, void* stream)
{
    return gsr_rgb_loss_backward_window(C, H, W, img, gt, weight, a_l1, a_ssim, 11, workspace, upstream, dL_dimg, stream);
}

// ---- simple_knn (knn.hip) ----
extern "C" size_t gsr_knn_workspace_bytes(int P) { return gsk_workspace_bytes(P); }

extern "C" int gsr_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* workspace, void* stream)
{
    if (P < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "knn: P must be >= 0 (got %d)", P);
    if (P == 0) return GSR_OK;
    if (!points || !mean_dist2 || !workspace) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "knn: a required pointer is NULL");
    const char* why = "";
    hipError_t e = gsk_launch(P, points, mean_dist2, workspace, (hipStream_t)stream, &why);
    if (e != hipSuccess) return gsr_fail(GSR_ERR_HIP, "knn: %s %s", hipGetErrorString(e), why);
    return GSR_OK;
}

// ---- neural-Gaussian decode (decode.hip) ----
static int gsr_check_decode(int N, int K, const float* const* weights)
{
    if (N < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: N must be >= 0 (got %d)", N);
    if (K < 1 || K > 10) return gsr_fail(GSR_ERR_UNSUPPORTED, "decode: n_offsets must be in 1..10 (got %d)", K);
    if (!weights) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: weights is NULL");
    for (int i = 0; i < 16; i++)
        if (!weights[i]) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: weights[%d] is NULL", i);
    return GSR_OK;
}

extern "C" int gsr_decode_visible_rows(int N, const uint8_t* visible_mask, int32_t* rows, uint32_t* count, uint32_t* block_scratch, void* stream)
{
    if (N < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: N must be >= 0 (got %d)", N);
    if (!count || (N > 0 && (!visible_mask || !rows || !block_scratch))) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: a required pointer is NULL");
    GSR_HIP(gsd_launch_visible_rows(N, visible_mask, rows, count, block_scratch, (hipStream_t)stream), "decode visible rows");
    return GSR_OK;
}

extern "C" int gsr_decode_count(int N, int K, const float* const* weights, const int32_t* visible, const uint32_t* visible_count, const float* feat, const float* anchor,
                                const float* campos, float* neural_opacity, uint8_t* mask, uint8_t* count, uint32_t* first,
                                uint32_t* total, uint32_t* block_scratch, void* stream)
{
    int rc = gsr_check_decode(N, K, weights);
    if (rc) return rc;
    if (!total) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: total is NULL");
    if (N == 0) { GSR_HIP(hipMemsetAsync(total, 0, 4, (hipStream_t)stream), "decode total"); return GSR_OK; }
    if (!feat || !anchor || !campos || !neural_opacity || !mask || !count || !first || !block_scratch)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: a required pointer is NULL");
    if (visible_count && !visible) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: visible_count without a row list");
    GSR_HIP(gsd_launch_count(N, K, weights, visible, visible_count, feat, anchor, campos, neural_opacity, mask, count, first, total, block_scratch,
                             (hipStream_t)stream),
            "decode count");
    return GSR_OK;
}

extern "C" int gsr_decode_emit(int N, int K, const float* const* weights, const int32_t* visible, const uint32_t* visible_count, const float* feat, const float* anchor,
                               const float* offsets, const float* grid_scaling, const float* campos,
                               const float* neural_opacity, const uint8_t* mask, const uint32_t* first, float* xyz, float* color, float* opacity, float* uncertainty,
                               float* scaling, float* rot, void* stream)
{
    int rc = gsr_check_decode(N, K, weights);
    if (rc) return rc;
    if (N == 0) return GSR_OK;
    if (!feat || !anchor || !offsets || !grid_scaling || !campos || !neural_opacity || !mask || !first)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: a required pointer is NULL");
    GSR_HIP(gsd_launch_emit(N, K, weights, visible, visible_count, feat, anchor, offsets, grid_scaling, campos, neural_opacity, mask, first, xyz, color, opacity,
                            uncertainty, scaling, rot, (hipStream_t)stream), "decode emit");
    return GSR_OK;
}

extern "C" int gsr_decode_backward(int N, int K, const float* const* weights, const int32_t* visible, const float* feat,
                                   const float* anchor, const float* offsets, const float* grid_scaling, const float* campos,
                                         const uint8_t* mask, const uint32_t* first, const float* g_xyz, const float* g_color,
                                         const float* g_opacity, const float* g_uncertainty, const float* g_scaling,
                                         const float* g_rot, float* d_feat, float* d_anchor, float* d_offsets,
                                         float* d_grid_scaling, void* workspace, float* const* grads16, void* stream)
{
    int rc = gsr_check_decode(N, K, weights);
    if (rc) return rc;
    if (!workspace || !grads16) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: workspace / grads16 is NULL");
    for (int i = 0; i < 16; i++)
        if (!grads16[i]) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: grads16[%d] is NULL", i);
    if (N > 0 && (!feat || !anchor || !offsets || !grid_scaling || !campos || !mask || !first || !d_feat || !d_anchor || !d_offsets ||
                  !d_grid_scaling))
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: a required pointer is NULL");
    if (!campos) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: campos is NULL");
    GSR_HIP(gsd_launch_backward(N, K, weights, visible, feat, anchor, offsets, grid_scaling, campos, mask, first, g_xyz, g_color,
                                      g_opacity, g_uncertainty, g_scaling, g_rot, d_feat, d_anchor, d_offsets, d_grid_scaling,
                                      workspace, grads16, (hipStream_t)stream), "decode backward");
    return GSR_OK;
}

extern "C" size_t gsr_decode_weight_grad_workspace_bytes(void) { return gsd_weight_grad_workspace_bytes(); }

extern "C" int gsr_decode_zero_hidden_rows(int N, int K, const uint8_t* visible_mask, float* d_feat, float* d_anchor, float* d_offsets,
                                           float* d_grid_scaling, void* stream)
{
    if (N < 0 || K < 1 || K > 10) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: bad sizes N=%d K=%d", N, K);
    if (N == 0) return GSR_OK;
    if (!visible_mask || !d_feat || !d_anchor || !d_offsets || !d_grid_scaling)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "decode: a required pointer is NULL");
    GSR_HIP(gsd_launch_zero_hidden(N, K, visible_mask, d_feat, d_anchor, d_offsets, d_grid_scaling, (hipStream_t)stream), "decode zero hidden rows");
    return GSR_OK;
}

// ---- depth loss (depth_loss.hip) ----
extern "C" size_t gsr_depth_loss_workspace_bytes(int H, int W) { return gdl_workspace_bytes(H, W); }

extern "C" int gsr_depth_loss_forward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                                      const float* l1_weight, const float* grad_mask, float lambda_l1, float lambda_smooth,
                                      void* workspace, float* out5, void* stream)
{
    if (H < 1 || W < 1 || (long long)H * W > 0x7fffffffLL) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "depth loss: bad size %dx%d", W, H);
    if (!depth || !target || !workspace || !out5) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "depth loss: a required pointer is NULL");
    GSR_HIP(gdl_launch_forward(H, W, depth, target, lsq_mask, l1_weight, grad_mask, lambda_l1, lambda_smooth, workspace, out5,
                               (hipStream_t)stream), "depth loss forward");
    return GSR_OK;
}

extern "C" int gsr_depth_loss_backward(int H, int W, const float* depth, const float* target, const float* lsq_mask,
                                       const void* workspace, const float* upstream, float* dL_ddepth, void* stream)
{
    if (H < 1 || W < 1 || (long long)H * W > 0x7fffffffLL) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "depth loss: bad size %dx%d", W, H);
    if (!depth || !target || !workspace || !dL_ddepth) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "depth loss: a required pointer is NULL");
    GSR_HIP(gdl_launch_backward(H, W, depth, target, lsq_mask, workspace, upstream, dL_ddepth, (hipStream_t)stream), "depth loss backward");
    return GSR_OK;
}

// ---- densification statistics (stats.hip) ----
extern "C" int gsr_training_stats(int Nv, int K, int M, const int32_t* visible, const float* neural_opacity, const uint8_t* selection,
                                  const uint32_t* first, const uint8_t* update_filter, const float* viewspace_grad,
                                  float* opacity_accum, float* anchor_demon, float* offset_gradient_accum,
                                  float* offset_denom, void* stream)
{
    if (Nv < 0 || K < 1 || M < 0) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "training stats: bad sizes Nv=%d K=%d M=%d", Nv, K, M);
    if (Nv == 0) return GSR_OK;
    if (!neural_opacity || !selection || !first || !opacity_accum || !anchor_demon || !offset_gradient_accum || !offset_denom)
        return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "training stats: a required pointer is NULL");
    // update_filter / viewspace_grad may be NULL only if no offset was selected (M = 0); the kernel then never reads them
    if (M > 0 && (!update_filter || !viewspace_grad)) return gsr_fail(GSR_ERR_INVALID_ARGUMENT, "training stats: update_filter / viewspace_grad is NULL with M = %d", M);
    GSR_HIP(gst_launch_training_stats(Nv, K, M, visible, neural_opacity, selection, first, update_filter, viewspace_grad,
                                      opacity_accum, anchor_demon, offset_gradient_accum, offset_denom, (hipStream_t)stream),
            "training stats");
    return GSR_OK;
}

