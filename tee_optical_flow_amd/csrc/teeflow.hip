// teeflow.hip -- host driver + C ABI (include/teeflow.h) of the MI355X DualTVL1 engine.
//
// Replaces, for the reference's hot path only, what cv2's DenseOpticalFlow object does behind
// /root/reference/optical_flow/calculate_optical_flow.py:564-600, 627-642.  No CPU fallback: without a
// gfx950 device tf_create fails with TF_ERR_NO_DEVICE.
//
// Execution model: a batch of B frame pairs advances in lock-step through
//   pyramid -> for level (coarse..fine): for warp: k_warp, then [median every `inner` iterations,
//   tvl1_iter] until each pair's own convergence test stops it.
// The stop decision lives on the device (per-pair error slots, blocks of stopped pairs exit at once);
// each tvl1_iter launch publishes "pairs still iterating" to a host-mapped word, which the host reads a few
// launches later to stop enqueuing a stage -- it never stalls the stream inside the iteration budget.
#include "teeflow_kernels.hip.h"
#include "teeflow_deepflow.hip.h"
#include "teeflow_sor_rt.hip.h"
#include "teeflow_analysis.hip.h"
#include "teeflow_wase.hip.h"
#include "teeflow_saliency.hip.h"
#include "../../include/teeflow.h"
#include <rccl/rccl.h>      // types and prototypes only: librccl is loaded with dlopen when a communicator is first asked for
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <deque>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <map>
#include <vector>

#define TF_API extern "C" __attribute__((visibility("default")))

namespace {

constexpr int MAXLEV = 64;
constexpr int DF_MAXLEV = 160;       // x0.95 pyramid: 60 levels at 512^2, 87 at 2048^2
constexpr int SLOT_RING = 1024;       // host-mapped words the tvl1_iter launches publish their active-pair count to
constexpr int DEFAULT_LAG = 1;        // the host enqueues at most this many launches beyond the last answer it has read
constexpr int DEFAULT_MAX_BATCH = 128;

thread_local std::string g_create_error;

struct ProfEv { hipEvent_t a, b; int level = 0, warp = 0, it = 0; float ms = 0.f; double work = 0.0; };

}  // namespace

// Implementation knobs (tf_set_tuning; results never depend on them).  One struct, so that a lane gets its engine's settings with ONE
// assignment (a knob missed in a field-by-field copy would silently make the lanes differ).
struct LanePool;
struct QJob;
struct TfKnobs {
    int iter_variant = 2;        // 0 = 64x16 tiles (k_iter), 1 = full-width row strips (k_iter_rows), 2 = row strips with TWO
                                 // iterations per launch (k_iter2_rows); 1 and 2 need W <= max_strip_width (2048) and enough rows*pairs
    int force_ry = 0;            // 0 = floor(256/QX) rows per step
    int adaptive_strips = 0;     // strip length from the known active-pair count: measured no gain
    int dynamic_strips = 1;      // strips sized on the device from the exact active-pair count (one round of resident blocks)
    int tile_max_w = 0;          // levels this narrow or narrower always take the tile kernels (experiment: see DESIGN section 8)
    int tile2 = 1;               // launches the row strips do not take (single pair, few pairs, > 2048 px wide) run two iterations per launch on tiles
    int max_strip_width = 2048;  // widest level the full-width strip kernels take (one quad per thread: 2048 px = 512-thread blocks).
                                 // 8 pairs: 1080x1920 57.7 vs 32.6 pairs/s with the tile kernel, 768x1100 184 vs 131, 720x1280 137 vs 148
    int sub_batches = 1;         // >1 cuts a host-pointer call that fits the capacity into that many sub-batches so the copy-out of
                                 // one overlaps the solve of the next; measured at 128 pairs @512^2: smaller batches cost more (2099 /
                                 // 2030 / 1886 / 1701 pairs/s for 1 / 2 / 3 / 4) than the 5 ms of D2H they hide.  Calls larger than
                                 // the capacity are cut anyway and do overlap.
    int sor_rt = 1;              // DeepFlow SOR: 1 = register-tile kernel k_df_sor_rt (teeflow_sor_rt.hip.h), 0 = one colour per launch (k_df_sor)
    int sor_plain_div = 0;       // tests: k_df_sor_rt takes its plain-IEEE-division path (what a block with out-of-range diagonals does)
    int sor_rt_shape = 3;        // k_df_sor_rt: 1 = 16 bands x 4 rows (1024 threads), 2 = 8 bands x 4 rows (128 x 32
                                 // regions, 512 threads), 3 = 1 or 2 per launch (launch_sor_rt)
    int df_fuse_ds = 2;          // DeepFlow: data term + smoothness contributions in one kernel (non-zero: k_df_data_smooth4, four pixels per thread,
                                 // 16-byte loads; 0: k_df_data then k_df_smooth, the plain form it is tested against)
    int sor_coop = 1;            // DeepFlow: all sweeps of a fixed-point iteration in one launch of co-resident regions (k_df_sor_rt_coop) where a
                                 // level needs more than one region and its regions fit the CUs this handle may use; 0 = always the tiled form,
                                 // 2 = 128 x 64 regions whatever the batch size and however full the launches (tests), 3 = the small-batch form
                                 // (128 x 32 regions) whenever the batch is small, sor_coop_small or not (tests)
    int sor_coop_min_util = 85;  // co-resident launches must be at least this full (per cent) RELATIVE to the tiled form's rounds, else the level runs
                                 // tiled (600x800 studies: 324 pairs/s always co-resident, 357 tiled, 359 with the rule)
    int sor_coop_small = 1;      // few pairs: co-resident 128 x 32 regions (0: the tiled form, as before)
    int sor_coop_s = 5;          // sweeps between two exchanges of (du, dv) in that kernel (the halo is 2 x this)
    int sor_fuse = 5;            // DeepFlow: complete red-black SOR sweeps per launch of k_df_sor_rt (0 = one colour per launch, in place).
                                 // 64 pairs @512^2: 466 / 534 / 562 / 548 / 567 pairs/s for 3 / 4 / 5 / 6 / 7; 5 divides the 25 sweeps evenly
    int warp_margin = 8;         // > 0: k_warp_lds<M> stages the I1 tile + margin in LDS (0: k_warp, 36 global gathers per pixel).  k_warp is
                                 // bound by the texture path (~7-10 cycles per scattered dword load and wave); from LDS the same taps cost
                                 // ~2.  128 pairs @512^2, warp stage per step: 5.0 ms gathers, 3.35 / 3.5 / 3.55 / 3.8 ms for M = 4 / 8 / 12 /
                                 // 16 (+3.4 % pairs/s).  A pixel displaced by more than M falls back to the gathers, so M only moves time.
    int min_rows_work = 8192;    // rows*pairs of a level below which the tile kernels are used (measured at 512^2 with k_iter2_tile: 16 pairs
                                 // 12.5 ms on tiles vs 13.9 ms on strips, 24 pairs 17.1 vs 17.3, 64 pairs 34.1 vs 29.5)
    int strip_blocks = 2048;     // target number of strip blocks per launch (sets rows per strip)
    int lag = DEFAULT_LAG;
    int slots_override = 0;      // resident blocks the strips are sized for (0 = what the occupancy query says)
    int lane_slots_pct = 67;     // queue lanes: per cent of the resident blocks a lane sizes its strips for.  Three lanes' launches share the GPU, so a
                                 // lane that cuts its level into one round of ALL resident blocks pays the 3 halo + 2 RY fill rows of short strips for
                                 // parallelism the other lanes already provide (queue form, 384 pairs per call: 100 % 2728-2745, 67 % 2769-2772,
                                 // 50 % 2767-2769, 33 % 2706-2711 pairs/s on one box)
    int coop_test_occ16 = -1, coop_test_occ8 = -1;   // tests: pretend the occupancy query answered this
    int coop_test_mute = 0;      // tests: block 0 of every co-resident launch never raises its flag -> its neighbours give up -> the call is repeated tiled
    int profile = 0;
};

struct tf_handle : TfKnobs {
    tf_params P;
    int dev = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    // geometry the buffers are allocated for
    int H = 0, W = 0, cap = 0, nlev = 0;
    double alloc_scale_step = 0; int alloc_nscales = 0;
    Geom lv[MAXLEV];
    float* pyr[MAXLEV] = {};
    float* gxl[MAXLEV] = {}; float* gyl[MAXLEV] = {};   // TF_VARIANT_CUDA: centred gradient of every frame, per level
    int alloc_variant = 0;
    float *cwx = nullptr, *cwy = nullptr, *crho = nullptr;
    StateBufs sb = {};
    PairCtl* ctl = nullptr;
    u64* errs = nullptr; int errstride = 0;
    int* iters_dev = nullptr; size_t iters_cap = 0;
    volatile int* slots_host = nullptr; int* slots_dev = nullptr;   // fine-grained pinned ring (SLOT_RING ints)
    unsigned launch_seq = 0;
    float* tab = nullptr;
    // staging for the host-pointer API
    hipStream_t copy_stream = nullptr;           // D2H of finished sub-batches overlaps the next solve (pinned destinations)
    hipEvent_t cev[4] = {nullptr, nullptr, nullptr, nullptr};   // solve done [2], copy done [2]
    uint8_t* st_u8 = nullptr; size_t st_u8_bytes = 0;
    int src_f32 = 0;                       // this call's frames are float32 in [0,1] (tf_calc_pairs_f32): 4 bytes per pixel
    float* st_flow = nullptr; size_t st_flow_bytes = 0;
    hipEvent_t ev[4] = {};
    // profiling of tvl1_iter launches
    std::deque<ProfEv> prof_pool; size_t prof_used = 0;      // deque: records keep their address while the pool grows
    // results of the last call
    std::vector<int> last_iters; int last_pairs = 0, last_nlev = 0, last_warps = 0;
    // per-call accumulators
    unsigned long long iter_launches = 0;
    double df_sor_bytes = 0;     // DeepFlow: algorithmic bytes of the SOR launches of the current call (40 B per pixel-sweep)
    double df_sor_px = 0;        // DeepFlow: pixels x pairs summed over the SOR launches (a launch's compulsory traffic is 40 B per pixel: 8 planes in, 2 out)
    // ---- DeepFlow (algo == TF_ALGO_DEEPFLOW) ----
    tf_deepflow_params DP = {};
    int dnlev = 0, dH = 0, dW = 0, dcap = 0;
    Geom dlv[DF_MAXLEV];
    float* dpyr_base = nullptr; size_t dpyr_off[DF_MAXLEV] = {};   // one allocation: level l of frame f at dpyr_base + off[l] + f*plane_l
    float* dtmp = nullptr;                                          // unblurred level-0 frames
    float* dplanes = nullptr;                                       // 21 state planes x cap pairs
    DfBufs df = {};
    // ---- frame preprocessing (conditioning, saliency): grow-only work buffers, freed with the handle ----
    struct GrowBuf { void* p = nullptr; size_t cap = 0; };
    enum { PRE_SRC, PRE_G0, PRE_G1, PRE_ION, PRE_IOFF, PRE_P, PRE_I, PRE_MON, PRE_MOFF, PRE_MX, PRE_OUT, PRE_COUNT };
    GrowBuf pre[PRE_COUNT];
    double pre_kernel_ms = 0;    // device time of the last saliency call's kernels (HIP events on the handle's stream)
    // ---- analysis session (row f1) ----
    double* an_rad = nullptr; double* an_lon = nullptr; int anN = 0, anH = 0, anW = 0;
    int lanes = 2;               // a batch of >= 32 pairs is split over this many independent (handle, stream, host thread) lanes:
                                 // while one lane runs the thin tail of a stage, the other fills the GPU.  Measured at 128 pairs
                                 // @512^2: 1 lane 2180, 2 lanes 2470, 3 lanes 2415, 4 lanes 2165 pairs/s (DeepFlow 377 vs 309)
    std::vector<tf_handle*> twins; bool is_twin = false;   // extra lanes (own stream, buffers, host thread each)
    int num_cus = 256;
    // WASE scratch (grown on demand): compacted products, block counts / offsets, piece sums, per-flow backgrounds
    float* wa = nullptr; size_t wa_cap = 0;
    unsigned* wcnt = nullptr; u64* woff = nullptr; size_t wcnt_cap = 0;
    float* wsum = nullptr; size_t wsum_cap = 0;
    float* wbg = nullptr; size_t wbg_cap = 0;
    std::map<size_t, int> slots_cache;      // resident k_iter2_rows blocks on the device, by (LDS bytes, waves per block)
    int coop_share = 0;          // CUs (= resident 1024-thread blocks) this handle may fill with such a launch; set per call (calc_entry)
    bool coop_disabled = false;  // a launch of this handle gave up waiting (foreign work on the GPU): tiled form until the back-off has run out
    int coop_backoff = 0;        // tiled solves (sub-batches) to sit out before the co-resident form is tried again; doubles with every abort
    int coop_cooldown = 0;       // ... of which this many are left
    int coop_rearms = 0;         // times the form was re-armed after a back-off
    int coop_occ16 = -1, coop_occ8 = -1;   // resident blocks per CU of k_df_sor_rt_coop<4,16> / <4,8> (hipOccupancyMaxActiveBlocksPerMultiprocessor), -1 = not asked yet
    int coop_asked16 = -2, coop_asked8 = -2;   // the test overrides (knobs) that answer was made with
    bool coop_used = false;      // this call launched k_df_sor_rt_coop
    int coop_aborts = 0;
    long long coop_launches = 0;
    unsigned coop_epoch = 0;     // flag value base of the next launch
    unsigned* coop_flags = nullptr;   // one 128-byte line per resident block + the abort word behind them
    int coop_flag_lines = 0;
    // RCCL: one communicator rank per handle, its own stream, a small ring of completion events
    ncclComm_t comm = nullptr; int comm_rank = 0, comm_size = 0;
    hipStream_t comm_stream = nullptr; hipEvent_t comm_ev[8] = {}; hipEvent_t comm_ready = nullptr; unsigned comm_tickets = 0;
    double warp_ms = 0, median_ms = 0;   // profiling: summed launch durations per stage of the last call
    // ---- engine lanes that pull whole sub-batches from a queue (calc_entry, tf_submit_*) ----
    int queue_lanes = -1;        // -1 = per algorithm (3 DualTVL1, 1 DeepFlow); 0 = never: every call is cut in contiguous parts that are joined at its end
    int queue_unit = 0;          // pairs per queue unit (0 = equal units of at most max_batch pairs, a multiple of the lane count of them)
    int queue_test_fail_unit = -1;   // tests: the lane that takes this unit of the next queued job reports a failure instead of solving it
    bool is_lane = false;        // this handle is a queue lane of another handle (an engine of its own: stream, buffers, host thread, its own twins)
    tf_handle* owner = nullptr;  // ... of this one
    LanePool* pool = nullptr;
    long long q_jobs = 0, q_units_done = 0, q_units_skipped = 0, q_units_failed = 0;
    std::map<int, QJob*> tickets; int next_ticket = 1;      // tf_submit_* jobs not yet waited for
    int stream_retries = 0;      // streams made and dropped while looking for lane / twin streams that run beside each other (give_concurrent_stream)
    int streams_serialised = 0;  // bit 0: a lane or twin had to keep a solve stream that shares a hardware queue with a sibling's; bit 1: a lane's copy stream shares one with a solve stream
};


// ---- lanes that pull whole sub-batches from a queue ---------------------------------------------------------------------------
// A call larger than one sub-batch (and every tf_submit_* job) becomes a QJob: it is cut into units of at most `unit` pairs, and the
// handle's lanes -- engines of their own: handle, stream, buffers, host thread -- take one unit at a time, oldest job first.  A lane
// that has finished a unit starts the next one at once, whichever job it belongs to, so one sub-batch's tail (few pairs still
// iterating, the fine pyramid levels done) runs under other sub-batches' full launches.  Rounds 1-4 cut such a call into L contiguous
// parts and joined them; bench.py reached the same overlap with three engines driven by Python threads (EnginePool).
struct QJob {
    int mode = 0; const uint8_t* in0 = nullptr; const uint8_t* in1 = nullptr; int n_pairs = 0, H = 0, W = 0; float scale = 1.f;
    float* out = nullptr; int device = 0; int src_f32 = 0;      // device: W_* bits
    tf_params P; tf_deepflow_params DP; TfKnobs knobs;      // the engine's settings when the job was queued
    int split_lanes = 1;                                    // lanes each unit is split over inside its queue lane (calc_split)
    int unit = 0, n_units = 0, next = 0, done = 0, fail_unit = -1;
    int rc = TF_OK; std::string err;
    tf_stats st; int merged = 0;
    std::vector<int> iters; size_t per_pair = 0; int nlev = 0, warps = 0;
    double t0 = 0;
    std::function<void(QJob*)> on_done;                     // run once, by the lane that finishes the last unit, before `finished`
    void* owned_dev = nullptr;                              // device memory that lives as long as the job (tf_submit_seq_rgb: the conditioned frames)
    bool finished = false;
    ~QJob() { if (owned_dev) (void)hipFree(owned_dev); }
};
struct LanePool {
    std::mutex m; std::condition_variable cv_work, cv_done;
    std::deque<QJob*> jobs;                                 // jobs that still have units to hand out, oldest first
    std::vector<tf_handle*> lanes; std::vector<std::thread> th;
    bool stop = false; int outstanding = 0;                 // jobs queued and not finished
};

TF_API int tf_create(const tf_params* p, int device_id, tf_handle** out);
TF_API int tf_create_deepflow(const tf_deepflow_params* p, int device_id, tf_handle** out);
TF_API const char* tf_last_error(tf_handle* h);
TF_API int tf_comm_destroy(tf_handle* h);

namespace {

int fail(tf_handle* h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPC(h, call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(h, e_ == hipErrorOutOfMemory ? TF_ERR_NOMEM : TF_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int cv_round_d(double v) { return (int)lrint(v); }   // saturate_cast<int>(double): nearest-even

Geom make_geom(int w, int h)
{
    Geom g; g.w = w; g.h = h; g.pitch = round_up(w, 32); g.plane = (long long)g.pitch * h; g.splane = g.plane;
    return g;
}

int validate_params(tf_handle* h, const tf_params& p)
{
    if (p.algo != TF_ALGO_TVL1) return fail(h, TF_ERR_UNSUPPORTED, "algo %d not implemented (only TF_ALGO_TVL1)", p.algo);
    if (p.nscales < 1 || p.nscales > MAXLEV) return fail(h, TF_ERR_INVALID_ARG, "nscales must be in [1,%d], got %d", MAXLEV, p.nscales);
    if (p.warps < 1) return fail(h, TF_ERR_INVALID_ARG, "warps must be >= 1, got %d", p.warps);
    if (p.inner_iterations < 1 || p.outer_iterations < 1)
        return fail(h, TF_ERR_INVALID_ARG, "inner/outer iterations must be >= 1, got %d/%d", p.inner_iterations, p.outer_iterations);
    if ((long long)p.inner_iterations * p.outer_iterations > 100000)
        return fail(h, TF_ERR_INVALID_ARG, "inner*outer iterations too large");
    if (p.median_filtering != 1 && p.median_filtering != 3 && p.median_filtering != 5)
        return fail(h, TF_ERR_UNSUPPORTED, "medianFiltering must be 1, 3 or 5 (cv::medianBlur on CV_32F), got %d", p.median_filtering);
    if (p.gamma != 0.0) return fail(h, TF_ERR_UNSUPPORTED, "gamma != 0 (illumination term u3) is not implemented");
    if (p.variant != TF_VARIANT_CPU && p.variant != TF_VARIANT_CUDA) return fail(h, TF_ERR_INVALID_ARG, "variant must be TF_VARIANT_CPU or TF_VARIANT_CUDA, got %d", p.variant);
    if (p.variant == TF_VARIANT_CUDA && (p.inner_iterations * p.outer_iterations) % 2 != 0)
        return fail(h, TF_ERR_UNSUPPORTED, "TF_VARIANT_CUDA needs an even iteration count (inner*outer), got %d", p.inner_iterations * p.outer_iterations);
    if (p.use_initial_flow) return fail(h, TF_ERR_UNSUPPORTED, "useInitialFlow is not implemented");
    if (!(p.scale_step > 0.0 && p.scale_step < 1.0)) return fail(h, TF_ERR_INVALID_ARG, "scaleStep must be in (0,1), got %g", p.scale_step);
    // cv::resize silently runs INTER_AREA instead of INTER_LINEAR when both scale factors are exactly 2 (imgproc/resize.cpp, "in
    // case of scale_x && scale_y is equal to 2"): same value mathematically, not the same float as the bilinear form k_pyr_down
    // computes.  That branch is not restated, so the one scaleStep that takes it is refused rather than silently different.
    if (p.scale_step == 0.5) return fail(h, TF_ERR_UNSUPPORTED, "scaleStep == 0.5 makes cv::resize take its INTER_AREA fast path for the pyramid, which is not implemented");
    if (!(p.theta > 0.0) || !(p.tau > 0.0) || !(p.lambda > 0.0) || !(p.epsilon >= 0.0))
        return fail(h, TF_ERR_INVALID_ARG, "tau, lambda, theta must be > 0 and epsilon >= 0");
    return TF_OK;
}

void free_buffers(tf_handle* h)
{
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    for (int l = 0; l < MAXLEV; ++l) { F(h->pyr[l]); F(h->gxl[l]); F(h->gyl[l]); }
    F(h->cwx); F(h->cwy); F(h->crho);
    for (int k = 0; k < 2; ++k) { F(h->sb.u1[k]); F(h->sb.u2[k]); F(h->sb.p11[k]); F(h->sb.p12[k]); F(h->sb.p21[k]); F(h->sb.p22[k]); }
    F(h->ctl); F(h->errs); F(h->iters_dev);
    F(h->st_u8); F(h->st_flow);
    F(h->dpyr_base); F(h->dtmp); F(h->dplanes); F(h->coop_flags); h->coop_flag_lines = 0;
    F(h->an_rad); F(h->an_lon); h->anN = 0;
    h->dnlev = h->dH = h->dW = h->dcap = 0;
    h->st_u8_bytes = h->st_flow_bytes = 0;
    h->H = h->W = h->cap = h->nlev = 0; h->iters_cap = 0;
}

// pyramid geometry of DualTVL1::calc: dsize = cvRound(size*scaleStep); stop before a level < 16 px
int compute_levels(const tf_params& P, int H, int W, Geom* lv)
{
    int n = 1;
    lv[0] = make_geom(W, H);
    for (int s = 1; s < P.nscales; ++s) {
        const int w = cv_round_d(lv[s - 1].w * P.scale_step), hh = cv_round_d(lv[s - 1].h * P.scale_step);
        if (w < 16 || hh < 16) break;
        lv[s] = make_geom(w, hh);
        lv[s].splane = lv[0].plane;
        n = s + 1;
    }
    return n;
}

int ensure_alloc(tf_handle* h, int H, int W, int B)
{
    const int want_cap = B < (h->P.max_batch > 0 ? h->P.max_batch : DEFAULT_MAX_BATCH) ? B : (h->P.max_batch > 0 ? h->P.max_batch : DEFAULT_MAX_BATCH);
    const int total = h->P.inner_iterations * h->P.outer_iterations;
    if (h->H == H && h->W == W && h->cap >= want_cap && h->alloc_scale_step == h->P.scale_step &&
        h->alloc_nscales == h->P.nscales && h->errstride >= total && h->alloc_variant == h->P.variant &&
        h->iters_cap >= (size_t)h->cap * (size_t)h->nlev * (size_t)h->P.warps * 2)
        return TF_OK;
    HIPC(h, hipStreamSynchronize(h->stream));
    free_buffers(h);
    h->nlev = compute_levels(h->P, H, W, h->lv);
    const size_t cap = (size_t)want_cap, fcap = 2 * cap;
    for (int l = 0; l < h->nlev; ++l) HIPC(h, hipMalloc(&h->pyr[l], fcap * h->lv[l].plane * sizeof(float)));
    if (h->P.variant == TF_VARIANT_CUDA)
        for (int l = 0; l < h->nlev; ++l) {
            HIPC(h, hipMalloc(&h->gxl[l], fcap * h->lv[l].plane * sizeof(float)));
            HIPC(h, hipMalloc(&h->gyl[l], fcap * h->lv[l].plane * sizeof(float)));
        }
    h->alloc_variant = h->P.variant;
    const size_t pl = (size_t)h->lv[0].plane * cap * sizeof(float);
    HIPC(h, hipMalloc(&h->cwx, pl)); HIPC(h, hipMalloc(&h->cwy, pl)); HIPC(h, hipMalloc(&h->crho, pl));
    for (int k = 0; k < 2; ++k) {
        HIPC(h, hipMalloc(&h->sb.u1[k], pl)); HIPC(h, hipMalloc(&h->sb.u2[k], pl));
        HIPC(h, hipMalloc(&h->sb.p11[k], pl)); HIPC(h, hipMalloc(&h->sb.p12[k], pl));
        HIPC(h, hipMalloc(&h->sb.p21[k], pl)); HIPC(h, hipMalloc(&h->sb.p22[k], pl));
    }
    HIPC(h, hipMalloc(&h->ctl, cap * sizeof(PairCtl)));
    h->errstride = total;
    HIPC(h, hipMalloc(&h->errs, cap * (size_t)h->errstride * sizeof(u64)));
    h->iters_cap = cap * (size_t)h->nlev * (size_t)h->P.warps * 2;
    HIPC(h, hipMalloc(&h->iters_dev, h->iters_cap * sizeof(int)));
    h->H = H; h->W = W; h->cap = want_cap;
    h->alloc_scale_step = h->P.scale_step; h->alloc_nscales = h->P.nscales;
    return TF_OK;
}

int ensure_staging(tf_handle* h, size_t u8_bytes, size_t flow_bytes)
{
    if (h->st_u8_bytes < u8_bytes) {
        if (h->st_u8) (void)hipFree(h->st_u8);
        h->st_u8 = nullptr; h->st_u8_bytes = 0;
        HIPC(h, hipMalloc(&h->st_u8, u8_bytes)); h->st_u8_bytes = u8_bytes;
    }
    if (h->st_flow_bytes < flow_bytes) {
        if (h->st_flow) (void)hipFree(h->st_flow);
        h->st_flow = nullptr; h->st_flow_bytes = 0;
        HIPC(h, hipMalloc(&h->st_flow, flow_bytes)); h->st_flow_bytes = flow_bytes;
    }
    return TF_OK;
}

inline dim3 grid64x4(const Geom& g, int z) { return dim3((g.w + 63) / 64, (g.h + 3) / 4, z); }

struct StageTotals {
    double iter_bytes = 0, total_bytes = 0;
};

// full-width strips need W <= max_strip_width (at most 2048: one quad per thread, 512 threads) and enough rows*pairs to fill 256 CUs; tiny launches (single-pair latency mode) keep the tiles
bool rows_ok(const tf_handle* h, const Geom& g, int B)
{
    return h->iter_variant >= 1 && g.w <= h->max_strip_width && g.w > h->tile_max_w && (long long)g.h * B >= h->min_rows_work;
}

// Block shape of the row-strip kernels: QX quads per row, RY = floor(256/QX) rows per step, 256 threads.
// (Measured on MI355X: shapes that fill more lanes with 320-512-thread blocks, or 1-row/128-thread blocks, are 10-35 %
// SLOWER -- more waves per barrier domain / fewer blocks per CU cost more than idle lanes; "force_ry" keeps the experiment.)
void strip_shape(const tf_handle* h, const Geom& g, int B, int* R, int* QX, int* RY, int* threads, bool two)
{
    (void)two;
    const int qx = (g.w + 3) / 4;
    int ry = 256 / qx;
    if (ry < 1) ry = 1;
    const int forced = h->force_ry;
    if (forced > 0 && qx * forced <= 512) ry = forced;
    *QX = qx; *RY = ry;
    *threads = (qx * ry <= 256 && forced <= 0) ? 256 : (qx * ry + 63) / 64 * 64;   // forced shapes: no idle waves
    long long n = (long long)g.h * B / ((long long)h->strip_blocks * ry);
    if (n < 2) n = 2;
    if (n > 16) n = 16;
    *R = ry * (int)n;
}

// launch one two-iteration tvl1_iter step (k_iter2_rows)
void launch_iter2(tf_handle* h, const Iter2Args& A, int B, hipStream_t s, int active_hint = 0)
{
    const Geom& g = A.a.g;
    if (!rows_ok(h, g, B)) {      // small launches and very wide levels: tiles
        hipLaunchKernelGGL(k_iter2_tile, dim3((g.w + T2_OW - 1) / T2_OW, (g.h + T2_OH - 1) / T2_OH, B), dim3(256), 0, s, A);
        return;
    }
    int R, QX, RY, threads;
    // rows per strip follow the number of pairs known to be still iterating: the thin tail launches of a stage get many
    // short strips (latency of a few steps) instead of a few long ones
    strip_shape(h, g, active_hint > 0 && h->adaptive_strips ? active_hint : B, &R, &QX, &RY, &threads, true);
    const int LW = QX * 4 + 4;
    const size_t shmem = (size_t)(32 + 8 * RY * LW + 2 * (RY + 1) * LW + 2 * RY * QX) * sizeof(float);
    if (h->dynamic_strips && B <= 1024) {
        // strips sized on the device from the exact number of pairs still iterating; the grid covers the largest item count
        int slots = h->slots_override;
        if (slots <= 0) {
            auto f = h->slots_cache.find(shmem * 1024 + (size_t)threads / 64);
            if (f == h->slots_cache.end()) {
                int per_cu = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_iter2_rows, threads, shmem);
                if (per_cu < 1) per_cu = 1;
                f = h->slots_cache.emplace(shmem * 1024 + (size_t)threads / 64, per_cu * h->num_cus).first;
            }
            slots = f->second;
            if (h->is_lane && h->lane_slots_pct > 0 && h->lane_slots_pct < 100) slots = slots * h->lane_slots_pct / 100;
        }
        int items = 1;
        for (int n = 1; n <= B; ++n) {
            int r, sn;
            strip_rule(n, g.h, RY, slots, &r, &sn);
            if (n * sn > items) items = n * sn;
        }
        hipLaunchKernelGGL(k_iter2_rows, dim3(items, 1, 1), dim3(threads), shmem, s, A, 0, QX, RY, slots);
        return;
    }
    hipLaunchKernelGGL(k_iter2_rows, dim3((g.h + R - 1) / R, 1, B), dim3(threads), shmem, s, A, R, QX, RY, 0);
}

// launch one tvl1_iter step for pairs [0,B) in the configured kernel form
void launch_iter(tf_handle* h, const IterArgs& ia, int B, hipStream_t s)
{
    const Geom& g = ia.g;
    if (rows_ok(h, g, B)) {
        int R, QX, RY, threads;
        strip_shape(h, g, B, &R, &QX, &RY, &threads, false);
        const int LW = QX * 4 + 4;
        const size_t shmem = (size_t)(32 + 8 * RY * LW + 2 * RY * QX) * sizeof(float);
        hipLaunchKernelGGL(k_iter_rows, dim3((g.h + R - 1) / R, 1, B), dim3(threads), shmem, s, ia, R, QX, RY);
    } else {
        const dim3 gi((g.w + IT_OW - 1) / IT_OW, (g.h + IT_OH - 1) / IT_OH, B);
        hipLaunchKernelGGL(k_iter, gi, dim3(256), 0, s, ia);
    }
}

// k_warp_lds is instantiated for a few margins (the staged width is a compile-time constant)
inline int warp_margin_class(int m) { return m <= 0 ? 0 : (m <= 4 ? 4 : (m <= 8 ? 8 : 16)); }
void launch_warp(tf_handle* h, const WarpArgs& wa, int B, hipStream_t s, const float* gx = nullptr, const float* gy = nullptr)
{
    const Geom& g = wa.g;
    if (h->P.variant == TF_VARIANT_CUDA) {
        WarpCudaArgs ca; ca.w = wa; ca.gx = gx; ca.gy = gy;
        hipLaunchKernelGGL(k_warp_cuda, dim3((g.w + 63) / 64, (g.h + 3) / 4, B), dim3(256), 0, s, ca);
        return;
    }
    const int M = warp_margin_class(h->warp_margin);
    const dim3 grid((g.w + WL_TW - 1) / WL_TW, (g.h + WL_TH - 1) / WL_TH, B);
    const size_t shm = (size_t)(128 + (WL_TW + 2 * (M + 4)) * (WL_TH + 2 * M + 7)) * sizeof(float);
    switch (M) {
        case 4: hipLaunchKernelGGL(k_warp_lds<4>, grid, dim3(256), shm, s, wa); break;
        case 8: hipLaunchKernelGGL(k_warp_lds<8>, grid, dim3(256), shm, s, wa); break;
        case 16: hipLaunchKernelGGL(k_warp_lds<16>, grid, dim3(256), shm, s, wa); break;
        default: hipLaunchKernelGGL(k_warp, dim3((g.w + 63) / 64, (g.h + 3) / 4, B), dim3(256), 0, s, wa); break;
    }
}

ProfEv* prof_next(tf_handle* h)
{
    if (h->prof_used == h->prof_pool.size()) {
        ProfEv pe;
        if (hipEventCreate(&pe.a) != hipSuccess || hipEventCreate(&pe.b) != hipSuccess) return nullptr;
        h->prof_pool.push_back(pe);
    }
    return &h->prof_pool[h->prof_used++];
}

// one (level, warp) stage for pairs [0,B)
int run_stage(tf_handle* h, int l, int wi, int B, int off0, int off1)
{
    const tf_params& P = h->P;
    const Geom g = h->lv[l];
    const int inner = P.inner_iterations, total = P.inner_iterations * P.outer_iterations;
    const float thr_f = (float)(P.epsilon * P.epsilon * (double)(g.w * g.h));
    const double thr_q = (double)thr_f * 1073741824.0;
    const double thr_d = P.epsilon * P.epsilon * (double)(g.w * g.h);     // TF_VARIANT_CUDA compares in double
    hipStream_t s = h->stream;

    WarpArgs wa;
    wa.pyr = h->pyr[l]; wa.off0 = off0; wa.off1 = off1; wa.sb = h->sb; wa.ctl = h->ctl; wa.tab = h->tab;
    wa.wx = h->cwx; wa.wy = h->cwy; wa.rho = h->crho; wa.g = g;
    if (h->profile) {
        ProfEv* pe = prof_next(h);
        if (!pe) return fail(h, TF_ERR_HIP, "hipEventCreate failed");
        pe->level = -4;
        HIPC(h, hipEventRecord(pe->a, s));
        launch_warp(h, wa, B, s, h->gxl[l], h->gyl[l]);
        HIPC(h, hipEventRecord(pe->b, s));
    } else launch_warp(h, wa, B, s, h->gxl[l], h->gyl[l]);
    HIPC(h, hipMemsetAsync(h->errs, 0, (size_t)B * h->errstride * sizeof(u64), s));

    IterArgs ia;
    ia.wx = h->cwx; ia.wy = h->cwy; ia.rho = h->crho; ia.sb = h->sb; ia.ctl = h->ctl; ia.err = h->errs;
    ia.errstride = h->errstride; ia.thr_q = thr_q; ia.g = g;
    ia.l_t = (float)(P.lambda * P.theta); ia.theta = (float)P.theta; ia.taut = (float)(P.tau / P.theta);
    ia.variant = P.variant; ia.thr_d = thr_d;
    const bool cuda_variant = P.variant == TF_VARIANT_CUDA;      // one loop, no median, stops only after odd iterations
    MedArgs ma;
    ma.sb = h->sb; ma.ctl = h->ctl; ma.err = h->errs; ma.errstride = h->errstride; ma.thr_q = thr_q; ma.g = g;

    const dim3 gm((g.w + 63) / 64, (g.h + 15) / 16, 2 * B);
    ia.B = B;
    const bool two = cuda_variant || (h->iter_variant >= 2 && (rows_ok(h, g, B) || h->tile2) && (inner % 2 == 0));
    if (two) {
        // two iterations per launch; launch index it = 0,2,..,total (the last one can only hold REPLAY blocks)
        int utog = 0, ptog = 0, utog_prev = 0, ptog_prev = 0, pzero_prev = 0;
        bool stop = false;
        int last_active = B;
        const unsigned seq0 = h->launch_seq;
        unsigned checked = seq0;
        for (int it = 0; it <= total && !stop; it += 2) {
            if (it < total && it % inner == 0 && P.median_filtering > 1 && !cuda_variant) {
                ma.it = it; ma.utog = utog;
                ProfEv* pm = h->profile ? prof_next(h) : nullptr;
                if (pm) { pm->level = -5; HIPC(h, hipEventRecord(pm->a, s)); }
                if (P.median_filtering == 5) hipLaunchKernelGGL(k_median2<5>, gm, dim3(256), 0, s, ma, total);
                else hipLaunchKernelGGL(k_median2<3>, gm, dim3(256), 0, s, ma, total);
                if (pm) HIPC(h, hipEventRecord(pm->b, s));
                ++utog;
            }
            const unsigned q = h->launch_seq++;
            h->slots_host[q % SLOT_RING] = -1;
            Iter2Args A2;
            A2.a = ia;
            A2.a.host_slot = h->slots_dev + q % SLOT_RING;
            A2.a.it = it; A2.a.utog = utog; A2.a.ptog = ptog; A2.a.pzero = (wi == 0 && it == 0) ? 1 : 0;
            A2.utog_prev = utog_prev; A2.ptog_prev = ptog_prev; A2.pzero_prev = pzero_prev; A2.total = total;
            if (h->profile) {
                if (h->prof_used == h->prof_pool.size()) {
                    ProfEv pe;
                    HIPC(h, hipEventCreate(&pe.a)); HIPC(h, hipEventCreate(&pe.b));
                    h->prof_pool.push_back(pe);
                }
                ProfEv& pe = h->prof_pool[h->prof_used++];
                pe.level = l; pe.warp = wi; pe.it = it;
                HIPC(h, hipEventRecord(pe.a, s));
                launch_iter2(h, A2, B, s, last_active);
                HIPC(h, hipEventRecord(pe.b, s));
            } else {
                launch_iter2(h, A2, B, s, last_active);
            }
            ++h->iter_launches;
            utog_prev = utog; ptog_prev = ptog; pzero_prev = A2.a.pzero;
            ++utog; ++ptog;
            while (checked <= q) {
                int v = h->slots_host[checked % SLOT_RING];
                if (v < 0) {
                    if (q - checked < (unsigned)h->lag) break;
                    const double t0 = now_ms();
                    while ((v = h->slots_host[checked % SLOT_RING]) < 0) {
                        if (now_ms() - t0 > 20000.0) return fail(h, TF_ERR_HIP, "tvl1_iter launch %u never reported (GPU hang?)", checked);
                        if (hipStreamQuery(s) == hipSuccess && h->slots_host[checked % SLOT_RING] < 0)
                            return fail(h, TF_ERR_HIP, "stream drained but launch %u did not report", checked);
                    }
                }
                ++checked;
                if (v == 0) { stop = true; break; }
                last_active = v;
            }
        }
        hipLaunchKernelGGL(k_stage_end2, dim3((B + 255) / 256), dim3(256), 0, s, h->errs, h->errstride, h->ctl, h->iters_dev, B,
                           total, inner, (P.median_filtering > 1 && !cuda_variant) ? 1 : 0, thr_q, l, wi, h->nlev, P.warps, P.variant, thr_d);
        return TF_OK;
    }
    int utog = 0, ptog = 0;
    bool stop = false;
    const unsigned seq0 = h->launch_seq;
    unsigned checked = seq0;          // launches [seq0, checked) have been read back
    for (int it = 0; it < total && !stop;) {
        if (it % inner == 0 && P.median_filtering > 1) {
            ma.it = it; ma.utog = utog;
            if (P.median_filtering == 5) hipLaunchKernelGGL(k_median<5>, gm, dim3(256), 0, s, ma);
            else hipLaunchKernelGGL(k_median<3>, gm, dim3(256), 0, s, ma);
            ++utog;
        }
        const unsigned q = h->launch_seq++;
        h->slots_host[q % SLOT_RING] = -1;
        ia.host_slot = h->slots_dev + q % SLOT_RING;
        ia.it = it; ia.utog = utog; ia.ptog = ptog; ia.pzero = (wi == 0 && it == 0) ? 1 : 0;
        if (h->profile) {
            if (h->prof_used == h->prof_pool.size()) {
                ProfEv pe;
                HIPC(h, hipEventCreate(&pe.a)); HIPC(h, hipEventCreate(&pe.b));
                h->prof_pool.push_back(pe);
            }
            ProfEv& pe = h->prof_pool[h->prof_used++];
            pe.level = l; pe.warp = wi; pe.it = it;
            HIPC(h, hipEventRecord(pe.a, s));
            launch_iter(h, ia, B, s);
            HIPC(h, hipEventRecord(pe.b, s));
        } else {
            launch_iter(h, ia, B, s);
        }
        ++h->iter_launches;
        ++utog; ++ptog; ++it;
        // read back what earlier launches of this stage published: never more than `lag` launches unread
        while (checked <= q) {
            int v = h->slots_host[checked % SLOT_RING];
            if (v < 0) {
                if (q - checked < (unsigned)h->lag) break;        // not there yet, and we may still run ahead
                const double t0 = now_ms();
                while ((v = h->slots_host[checked % SLOT_RING]) < 0) {
                    if (now_ms() - t0 > 20000.0) return fail(h, TF_ERR_HIP, "tvl1_iter launch %u never reported (GPU hang?)", checked);
                    if (hipStreamQuery(s) == hipSuccess && h->slots_host[checked % SLOT_RING] < 0)
                        return fail(h, TF_ERR_HIP, "stream drained but launch %u did not report", checked);
                }
            }
            ++checked;
            if (v == 0) { stop = true; break; }   // nobody entered that iteration active: the rest would be no-ops
        }
    }
    hipLaunchKernelGGL(k_stage_end, dim3((B + 255) / 256), dim3(256), 0, s, h->errs, h->errstride, h->ctl, h->iters_dev, B,
                       total, inner, P.median_filtering > 1 ? 1 : 0, thr_q, l, wi, h->nlev, P.warps);
    return TF_OK;
}

// Solve B pairs whose u8 frames are in device memory: frames[F][H][W], pair b = (off0+b, off1+b).
int solve_resident(tf_handle* h, const uint8_t* dframes, int F, int B, int off0, int off1, float scale, float* dflow)
{
    const tf_params& P = h->P;
    hipStream_t s = h->stream;
    const Geom g0 = h->lv[0];
    if (h->src_f32) hipLaunchKernelGGL(k_f32_to_level0, dim3((g0.w + 255) / 256, g0.h, F), dim3(256), 0, s, (const float*)dframes, h->pyr[0], g0, 1);
    else hipLaunchKernelGGL(k_u8_to_f32, dim3((g0.w + 255) / 256, g0.h, F), dim3(256), 0, s, dframes, h->pyr[0], g0);
    for (int l = 1; l < h->nlev; ++l) {
        const double sc = 1.0 / P.scale_step;   // resize(src, Size(), fx, fy): scale = 1/fx
        hipLaunchKernelGGL(k_pyr_down, grid64x4(h->lv[l], F), dim3(256), 0, s, h->pyr[l - 1], h->lv[l - 1], h->pyr[l], h->lv[l], sc, sc,
                           P.variant == TF_VARIANT_CUDA ? 1 : 0);
    }
    if (P.variant == TF_VARIANT_CUDA)
        for (int l = 0; l < h->nlev; ++l)
            hipLaunchKernelGGL(k_grad, grid64x4(h->lv[l], F), dim3(256), 0, s, h->pyr[l], h->gxl[l], h->gyl[l], h->lv[l]);
    const int L = h->nlev - 1;
    hipLaunchKernelGGL(k_ctl_set, dim3((B + 255) / 256), dim3(256), 0, s, h->ctl, B, 0);
    HIPC(h, hipMemset2DAsync(h->sb.u1[0], (size_t)h->lv[L].splane * sizeof(float), 0, (size_t)h->lv[L].plane * sizeof(float), B, s));
    HIPC(h, hipMemset2DAsync(h->sb.u2[0], (size_t)h->lv[L].splane * sizeof(float), 0, (size_t)h->lv[L].plane * sizeof(float), B, s));
    for (int l = L; l >= 0; --l) {
        for (int wi = 0; wi < P.warps; ++wi) {
            int rc = run_stage(h, l, wi, B, off0, off1);
            if (rc) return rc;
        }
        if (l == 0) break;
        const Geom gs = h->lv[l], gd = h->lv[l - 1];
        // resize(u, size(I0s[s-1])): inv_scale = dsize/ssize, scale = 1/inv_scale
        const double sx = 1.0 / ((double)gd.w / gs.w), sy = 1.0 / ((double)gd.h / gs.h);
        hipLaunchKernelGGL(k_flow_up, grid64x4(gd, B), dim3(256), 0, s, h->sb, h->ctl, gs, gd, sx, sy, (float)(1 / P.scale_step),
                           P.variant == TF_VARIANT_CUDA ? 1 : 0);
        hipLaunchKernelGGL(k_ctl_set, dim3((B + 255) / 256), dim3(256), 0, s, h->ctl, B, 1);
    }
    hipLaunchKernelGGL(k_output, grid64x4(g0, B), dim3(256), 0, s, h->sb, h->ctl, g0, scale, dflow);
    HIPC(h, hipGetLastError());
    return TF_OK;
}

// algorithmic (compulsory) HBM bytes of one solved pair from its executed iteration counts
void account_bytes(const tf_handle* h, const int* it /* [nlev][warps][2] */, double* iter_bytes, double* total_bytes,
                   unsigned long long* n_in, unsigned long long* n_out)
{
    const int warps = h->P.warps;
    double ib = 0, tb = 0;
    for (int l = 0; l < h->nlev; ++l) {
        const double px = (double)h->lv[l].w * h->lv[l].h;
        for (int w = 0; w < warps; ++w) {
            const int ni = it[(l * warps + w) * 2], no = it[(l * warps + w) * 2 + 1];
            *n_in += ni; *n_out += no;
            ib += px * 60.0 * ni;                                   // tvl1_iter: 9 reads + 6 writes
            tb += px * (16.0 * (h->P.median_filtering > 1 ? no : 0)  // median: read+write u1,u2
                        + 28.0);                                    // warp: read I0,I1,u1,u2; write I1wx,I1wy,rho_c
        }
        if (l > 0) tb += px * 8.0 + (double)h->lv[l - 1].w * h->lv[l - 1].h * 8.0;          // flow upsample
        if (l > 0) tb += 2.0 * (px * 4.0 + (double)h->lv[l - 1].w * h->lv[l - 1].h * 4.0);  // pyramid level (2 frames)
    }
    tb += (double)h->lv[0].w * h->lv[0].h * (2.0 * (1 + 4) + 8.0 + 8.0);  // u8->f32 of 2 frames, output interleave
    *iter_bytes += ib; *total_bytes += tb + ib;
}

// =================================================================================================
// DeepFlow host side
// =================================================================================================
int df_levels(const tf_deepflow_params& P, int H, int W, Geom* lv)
{
    int n = 1;
    lv[0] = make_geom(W, H);
    while (n < DF_MAXLEV) {
        // Size((int)(cols*downscaleFactor + 0.5f), (int)(rows*downscaleFactor + 0.5f)), float arithmetic
        const int nw = (int)(lv[n - 1].w * P.downscale_factor + 0.5f), nh = (int)(lv[n - 1].h * P.downscale_factor + 0.5f);
        if (nh <= P.min_size || nw <= P.min_size) break;
        lv[n] = make_geom(nw, nh);
        lv[n].splane = lv[0].plane;
        ++n;
    }
    return n;
}

int df_validate(tf_handle* h, const tf_deepflow_params& p)
{
    if (!(p.sigma > 0.f) || (int)floorf(3 * p.sigma) * 2 + 1 != 3)
        return fail(h, TF_ERR_UNSUPPORTED, "DeepFlow pre-blur: only the 3x3 kernel (1/3 <= sigma < 2/3) is implemented, sigma=%g", p.sigma);
    if (!(p.downscale_factor > 0.1f && p.downscale_factor < 1.f)) return fail(h, TF_ERR_INVALID_ARG, "downscaleFactor must be in (0.1,1)");
    if (p.min_size < 1 || p.fixed_point_iterations < 0 || p.sor_iterations < 0 || p.fixed_point_iterations > 1000 || p.sor_iterations > 10000)
        return fail(h, TF_ERR_INVALID_ARG, "bad DeepFlow iteration/size parameters");
    return TF_OK;
}

// one sub-batch solved: a handle that is sitting out an abort comes one step closer to trying the co-resident form again
void coop_tick(tf_handle* h)
{
    if (h->coop_disabled && h->coop_cooldown > 0 && --h->coop_cooldown == 0) { h->coop_disabled = false; ++h->coop_rearms; }
}
// The co-resident form counts on ONE 1024-thread block (128 x 64 regions) or TWO 512-thread blocks (128 x 32) per CU.  Ask the runtime
// instead of assuming it: a build whose register or LDS use has grown past that is refused the form (the tiled one does the same work).
void coop_query_occupancy(tf_handle* h)
{
    if (h->coop_occ16 >= 0 && h->coop_asked16 == h->coop_test_occ16 && h->coop_asked8 == h->coop_test_occ8) return;
    h->coop_asked16 = h->coop_test_occ16; h->coop_asked8 = h->coop_test_occ8;
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_df_sor_rt_coop<4, 16>, 1024, 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_df_sor_rt_coop<4, 8>, 512, 0) != hipSuccess) b = 0;
    (void)hipGetLastError();
    h->coop_occ16 = h->coop_test_occ16 >= 0 ? h->coop_test_occ16 : a;
    h->coop_occ8 = h->coop_test_occ8 >= 0 ? h->coop_test_occ8 : b;
}
// k_df_sor_rt_coop's meeting place: a flag line per block that can be resident (one per CU) + the abort word behind them
int coop_ensure(tf_handle* h)
{
    if (h->coop_flags) return TF_OK;
    h->coop_flag_lines = 2 * h->num_cus;          // 128 x 32 regions: two 512-thread blocks per CU
    HIPC(h, hipMalloc(&h->coop_flags, ((size_t)h->coop_flag_lines + 1) * 128));
    HIPC(h, hipMemsetAsync(h->coop_flags, 0, ((size_t)h->coop_flag_lines + 1) * 128, h->stream));
    h->coop_epoch = 0;
    coop_query_occupancy(h);
    return TF_OK;
}
// after the stream has drained: did a launch of this call give up waiting?  Then the results are void: the tiled form from now on.
int coop_aborted(tf_handle* h, bool* aborted)
{
    *aborted = false;
    if (!h->coop_used || !h->coop_flags) return TF_OK;
    h->coop_used = false;
    unsigned word = 0;
    HIPC(h, hipMemcpy(&word, h->coop_flags + (size_t)h->coop_flag_lines * 32, sizeof word, hipMemcpyDeviceToHost));
    if (!word) return TF_OK;
    // Back-off, not a verdict: whatever held the CUs (another process, another stream) is usually gone a few solves later.  Sit out
    // 16 tiled sub-batches, twice as many after every further abort (capped), then try the form again (coop_tick).
    h->coop_disabled = true;
    ++h->coop_aborts;
    h->coop_backoff = h->coop_backoff ? (h->coop_backoff < 4096 ? 2 * h->coop_backoff : 4096) : 16;
    h->coop_cooldown = h->coop_backoff;
    h->err = "co-resident SOR launch gave up waiting (foreign work on the GPU?): sub-batch repeated with the tiled form";   // readable through tf_last_error
    HIPC(h, hipMemset(h->coop_flags, 0, ((size_t)h->coop_flag_lines + 1) * 128));
    *aborted = true;
    return TF_OK;
}
// Two launches of co-resident regions that each count on the same CUs can wait for each other for ever (each holds CUs the other's
// last blocks need), so at most one call per device and process may use the form at a time; its lanes split the CUs between them.
static std::atomic<int> g_coop_busy[64];
struct CoopClaim {
    int dev; bool ok;
    explicit CoopClaim(int dev_) : dev(dev_), ok(false) { int z = 0; if (dev >= 0 && dev < 64) ok = g_coop_busy[dev].compare_exchange_strong(z, 1); }
    ~CoopClaim() { if (ok) g_coop_busy[dev].store(0); }
};

int df_ensure_alloc(tf_handle* h, int H, int W, int B)
{
    int rc_coop = TF_OK;
    const int mb = h->DP.max_batch > 0 ? h->DP.max_batch : DEFAULT_MAX_BATCH;
    const int want = B < mb ? B : mb;
    if (h->dH == H && h->dW == W && h->dcap >= want) return TF_OK;
    HIPC(h, hipStreamSynchronize(h->stream));
    free_buffers(h);
    h->dnlev = df_levels(h->DP, H, W, h->dlv);
    const size_t cap = (size_t)want, F = 2 * cap;
    size_t total = 0;
    for (int l = 0; l < h->dnlev; ++l) { h->dpyr_off[l] = total; total += F * (size_t)h->dlv[l].plane; }
    HIPC(h, hipMalloc(&h->dpyr_base, total * sizeof(float)));
    HIPC(h, hipMalloc(&h->dtmp, F * (size_t)h->dlv[0].plane * sizeof(float)));
    const size_t pl = (size_t)h->dlv[0].plane * cap;
    HIPC(h, hipMalloc(&h->dplanes, 23 * pl * sizeof(float)));
    float* p = h->dplanes;
    DfBufs& d = h->df;
    float** slots[] = {&d.avg, &d.Iz, &d.Ix, &d.Iy, &d.Ixx, &d.Ixy, &d.Iyy, &d.Ixz, &d.Iyz, &d.A11, &d.A12, &d.A22, &d.b1, &d.b2, &d.wg,
                       &d.du, &d.dv, &d.du2, &d.dv2, &d.Wu[0], &d.Wu[1], &d.Wv[0], &d.Wv[1]};
    for (auto s_ : slots) { *s_ = p; p += pl; }
    rc_coop = coop_ensure(h);
    if (rc_coop) return rc_coop;
    h->dH = H; h->dW = W; h->dcap = want;
    return TF_OK;
}

DfConst df_consts(const tf_deepflow_params& P)
{
    // OpticalFlowDeepFlow::calc: var->setAlpha(4*alpha); setDelta(delta/3); setGamma(gamma/3)
    const float alpha = 4 * P.alpha, delta = P.delta / 3, gamma = P.gamma / 3;
    DfConst c;
    c.zeta2 = P.zeta * P.zeta; c.eps2 = P.epsilon * P.epsilon;
    c.delta2 = delta / 2; c.gamma2 = gamma / 2; c.alpha2 = alpha / 2; c.omega = P.omega;
    return c;
}

void df_gauss3(float sigma, float* k0, float* k1)
{
    // getGaussianKernel(3, sigma, CV_32F): normalised in double, cast to float
    const double s2 = -0.5 / ((double)sigma * (double)sigma);
    const double t0 = exp(s2 * 1.0), t1 = exp(0.0);
    const double inv = 1.0 / (t0 + t1 + t0);
    *k0 = (float)(t1 * inv); *k1 = (float)(t0 * inv);
}

// register-tile SOR (teeflow_sor_rt.hip.h): `sweeps` sweeps per launch on 128 x (R*NB) regions with a halo of hl = 2 * sweeps
// (hl = 0: the region holds the whole level)
template <int R, int NB>
void launch_sor_rt_t(const DfBufs& d, const Geom& g, int B, float omega, int sweeps, int hl, hipStream_t s, int plain_div)
{
    constexpr int RW = 128, RH = R * NB;
    const int nx = g.w <= RW ? 1 : 1 + (g.w - RW + (RW - 2 * hl) - 1) / (RW - 2 * hl);
    const int ny = g.h <= RH ? 1 : 1 + (g.h - RH + (RH - 2 * hl) - 1) / (RH - 2 * hl);
    hipLaunchKernelGGL((k_df_sor_rt<R, NB>), dim3(nx, ny, B), dim3(64 * NB), 0, s, d, g, omega, sweeps, hl, plain_div);
}
// returns the number of sweeps it ran (all of `left` when the level fits one region).
// Region shapes: 128 x 64 held by 16 bands x 4 rows (1024 threads, one block per CU) is the throughput shape -- least halo.  When it
// would leave most of the chip idle (a single pair, or the small levels of a batch: fewer blocks than CUs) the same 4-row bands are
// stacked only 8 high: 128 x 32 regions, 512 threads, two blocks per CU, ~2.5x the blocks and half the sweep time per block -- as
// long as they all fit one round of resident blocks.  (64 pairs @512^2 are unaffected; single pair 27.3 -> see DESIGN.md.)
int launch_sor_rt(tf_handle* h, const DfBufs& d, const Geom& g, int B, float omega, int left, int fuse, hipStream_t s)
{
    auto tiles = [&](int RH, int hl) {
        const int nx = g.w <= 128 ? 1 : 1 + (g.w - 128 + (128 - 2 * hl) - 1) / (128 - 2 * hl);
        const int ny = g.h <= RH ? 1 : 1 + (g.h - RH + (RH - 2 * hl) - 1) / (RH - 2 * hl);
        return nx * ny;
    };
    int shape = h->sor_rt_shape;
    if (shape == 3 && g.w <= 62 && g.h <= 128) {
        // a level this narrow fills at most half a wave: two bands per wave (k_df_sor_rt<.., HALF>), all sweeps in one launch
        if (g.h <= 64) hipLaunchKernelGGL((k_df_sor_rt<4, 8, true>), dim3(1, 1, B), dim3(512), 0, s, d, g, omega, left, 0, h->sor_plain_div);
        else hipLaunchKernelGGL((k_df_sor_rt<4, 16, true>), dim3(1, 1, B), dim3(1024), 0, s, d, g, omega, left, 0, h->sor_plain_div);
        return left;
    }
    const bool whole64 = g.w <= 128 && g.h <= 64, whole32 = g.w <= 128 && g.h <= 32;
    if (shape == 3) {
        shape = 1;
        const int n5 = left < fuse ? left : fuse;
        if (whole32) shape = 2;                                            // fits 8 bands: half the waves, same sweeps
        else if (!whole64 && 32 - 4 * n5 >= 8 && tiles(64, 2 * n5) * B < h->num_cus && tiles(32, 2 * n5) * B <= 2 * h->num_cus) shape = 2;
    }
    const bool whole = shape == 2 ? whole32 : whole64;
    int n = whole ? left : (left < fuse ? left : fuse);
    if (!whole && shape == 2 && 32 - 4 * n < 4) n = 6;                     // 128 x 32 regions: at most 6 sweeps per launch (core of 8 rows)
    if (n > left) n = left;
    const int hl = whole ? 0 : 2 * n;
    if (shape == 2) launch_sor_rt_t<4, 8>(d, g, B, omega, n, hl, s, h->sor_plain_div);
    else launch_sor_rt_t<4, 16>(d, g, B, omega, n, hl, s, h->sor_plain_div);
    return n;
}

// Co-resident form (k_df_sor_rt_coop): regions of a level and how many pairs' worth of them this handle may keep resident at once
// (0: the level is one region, or its regions do not fit -- tiled / whole-level form)
int sor_coop_pairs(const tf_handle* h, const Geom& g, int B, int S, int* nx_, int* ny_, int* rows_)
{
    const int hl = 2 * S;
    *rows_ = 64;
    if (!h->sor_coop || h->coop_disabled || !h->coop_flags || h->sor_rt_shape != 3 || 64 - 2 * hl < 8 || 3 * hl > 64) return 0;
    if (h->coop_occ16 < 1) return 0;                        // the runtime does not promise a resident 1024-thread block per CU: no co-resident form
    const int nx = g.w <= 128 ? 1 : 1 + (g.w - 128 + (128 - 2 * hl) - 1) / (128 - 2 * hl);
    const int ny = g.h <= 64 ? 1 : 1 + (g.h - 64 + (64 - 2 * hl) - 1) / (64 - 2 * hl);
    const int share = h->coop_share < h->coop_flag_lines / 2 ? h->coop_share : h->coop_flag_lines / 2;
    if (nx * ny < 2 || nx * ny > share) return 0;
    // few pairs: 128 x 64 regions would leave most CUs idle for the whole fixed-point iteration.  Like the tiled form (launch_sor_rt) the
    // co-resident one then takes 128 x 32 regions: 512-thread blocks, two per CU, ~2.5 x the blocks and half the sweep time per block
    const int ny32 = g.h <= 32 ? 1 : 1 + (g.h - 32 + (32 - 2 * hl) - 1) / (32 - 2 * hl);
    // (a region waits for the 8 regions around it, so its halo must not reach past their cores: hl <= core, i.e. 3 hl <= 32 -- S <= 5;
    // the 64-row regions satisfy 3 hl <= 64 for every S the knob allows)
    if (h->sor_coop != 2 && 32 - 2 * hl >= 8 && 3 * hl <= 32 && nx * ny * B < h->num_cus && nx * ny32 * B <= 2 * h->num_cus) {
        if (h->sor_coop != 3 && !h->sor_coop_small) return 0;
        if (h->coop_occ8 < 2) return 0;                     // two resident 512-thread blocks per CU are what this form counts on
        if (nx * ny32 < 2 || nx * ny32 * B > 2 * share) return 0;
        *nx_ = nx; *ny_ = ny32; *rows_ = 32;
        return B;                                                       // all of them in one launch
    }
    // Whole pairs only: a batch goes through in ceil(B / cp) launches that each hold `share` CUs, the tiled form needs
    // ceil(B * regions / share) rounds of blocks.  Where whole pairs leave much of the share empty (one pair of 66 regions on 128 CUs)
    // the tiled form is quicker although it loads the system five times: co-resident only if its launches are nearly as full.
    const int cp = share / (nx * ny);
    const long long groups = (B + cp - 1) / cp, rounds_t = ((long long)B * nx * ny + share - 1) / share;
    if (h->sor_coop == 1 && rounds_t * 100 < (long long)h->sor_coop_min_util * groups) return 0;
    *nx_ = nx; *ny_ = ny;
    return cp;
}

// one cv::VariationalRefinement::calcUV for pairs [0,B) on level geometry g: W[cur] -> (avg, Iz) = W + dW
void df_refine_level(tf_handle* h, const float* pyr_l, int off0, int off1, const Geom& g, int cur, int B, hipStream_t s)
{
    DfBufs d = h->df;
    const DfConst c = df_consts(h->DP);
    const dim3 gr = grid64x4(g, B), bl(256);
    const dim3 gsor(((g.w + 1) / 2 + 63) / 64, (g.h + 3) / 4, B);
    hipLaunchKernelGGL(k_df_warp, gr, bl, 0, s, pyr_l, off0, off1, d, cur, g);
    hipLaunchKernelGGL(k_df_grad1, gr, bl, 0, s, d, g);
    hipLaunchKernelGGL(k_df_grad2, gr, bl, 0, s, d, g);
    const int fuse = h->sor_fuse < 0 ? 0 : h->sor_fuse;
    for (int fp = 0; fp < h->DP.fixed_point_iterations; ++fp) {
        if (h->df_fuse_ds) hipLaunchKernelGGL(k_df_data_smooth4, dim3((g.w + 255) / 256, (g.h + 3) / 4, B), bl, 0, s, d, cur, g, c);
        else {
            hipLaunchKernelGGL(k_df_data, gr, bl, 0, s, d, cur, g, c);
            hipLaunchKernelGGL(k_df_smooth, gr, bl, 0, s, d, cur, g);
        }
        int left = h->DP.sor_iterations;
        auto prof_begin = [&]() -> ProfEv* {
            ProfEv* pe = nullptr;
            if (h->profile) {
                if (h->prof_used == h->prof_pool.size()) {
                    ProfEv ne;
                    if (hipEventCreate(&ne.a) == hipSuccess && hipEventCreate(&ne.b) == hipSuccess) h->prof_pool.push_back(ne);
                }
                if (h->prof_used < h->prof_pool.size()) { pe = &h->prof_pool[h->prof_used++]; (void)hipEventRecord(pe->a, s); }
            }
            ++h->iter_launches;
            return pe;
        };
        int cnx = 0, cny = 0, crows = 64;
        const int S = h->sor_coop_s < 1 ? 1 : (h->sor_coop_s > 8 ? 8 : h->sor_coop_s);
        const int cpairs = h->sor_rt && fuse > 0 && left > S ? sor_coop_pairs(h, g, B, S, &cnx, &cny, &crows) : 0;
        if (cpairs > 0) {
            // all `left` sweeps in one launch per group of pairs; the result is in (du2, dv2) after an odd number of phases
            const int phases = (left + S - 1) / S;
            if (h->coop_epoch > (1u << 30)) {            // flags are compared as signed differences: start over long before a stale line could look ahead
                (void)hipMemsetAsync(h->coop_flags, 0, (size_t)h->coop_flag_lines * 128, s);
                h->coop_epoch = 0;
            }
            for (int b0 = 0; b0 < B; b0 += cpairs) {
                const int nb = B - b0 < cpairs ? B - b0 : cpairs;
                ProfEv* pe = prof_begin();
                if (crows == 32)
                    hipLaunchKernelGGL((k_df_sor_rt_coop<4, 8>), dim3(cnx, cny, nb), dim3(512), 0, s, d, g, c.omega, left, S, h->sor_plain_div | (h->coop_test_mute ? 2 : 0), b0,
                                       h->coop_flags, h->coop_epoch, h->coop_flags + (size_t)h->coop_flag_lines * 32);
                else
                    hipLaunchKernelGGL((k_df_sor_rt_coop<4, 16>), dim3(cnx, cny, nb), dim3(1024), 0, s, d, g, c.omega, left, S, h->sor_plain_div | (h->coop_test_mute ? 2 : 0), b0,
                                       h->coop_flags, h->coop_epoch, h->coop_flags + (size_t)h->coop_flag_lines * 32);
                h->coop_epoch += (unsigned)phases;
                ++h->coop_launches;
                h->df_sor_bytes += (double)left * g.w * g.h * nb * 40.0;
                h->df_sor_px += (double)g.w * g.h * nb;
                if (pe) (void)hipEventRecord(pe->b, s);
            }
            h->coop_used = true;
            if (phases & 1) { std::swap(d.du, d.du2); std::swap(d.dv, d.dv2); }
            left = 0;
        }
        while (h->sor_rt && fuse > 0 && left > 0) {
            ProfEv* pe = prof_begin();
            const int n = launch_sor_rt(h, d, g, B, c.omega, left, fuse > 8 ? 8 : fuse, s);
            h->df_sor_bytes += (double)n * g.w * g.h * B * 40.0;
            h->df_sor_px += (double)g.w * g.h * B;
            if (pe) (void)hipEventRecord(pe->b, s);
            std::swap(d.du, d.du2); std::swap(d.dv, d.dv2);
            left -= n;
        }
        while (left > 0) {     // sor_fuse = 0 (or sor_rt = 0): one colour per launch, in place -- the plain form the others are tested against
            hipLaunchKernelGGL(k_df_sor, gsor, bl, 0, s, d, g, 0, c.omega);
            hipLaunchKernelGGL(k_df_sor, gsor, bl, 0, s, d, g, 1, c.omega);
            h->df_sor_bytes += (double)g.w * g.h * B * 40.0;
            --left;
        }
    }
    hipLaunchKernelGGL(k_df_sum, gr, bl, 0, s, d, cur, g);
}

int df_solve_resident(tf_handle* h, const uint8_t* dframes, int F, int B, int off0, int off1, float scale, float* dflow)
{
    hipStream_t s = h->stream;
    const Geom g0 = h->dlv[0];
    float k0, k1;
    df_gauss3(h->DP.sigma, &k0, &k1);
    // convertTo(CV_32F) without a factor: uint8 frames keep 0..255, float frames (a saliency map in [0,1]) are taken as they are
    if (h->src_f32) hipLaunchKernelGGL(k_f32_to_level0, dim3((g0.w + 255) / 256, g0.h, F), dim3(256), 0, s, (const float*)dframes, h->dtmp, g0, 0);
    else hipLaunchKernelGGL(k_u8_to_f32, dim3((g0.w + 255) / 256, g0.h, F), dim3(256), 0, s, dframes, h->dtmp, g0);
    hipLaunchKernelGGL(k_df_blur, grid64x4(g0, F), dim3(256), 0, s, h->dtmp, h->dpyr_base + h->dpyr_off[0], g0, k0, k1);
    for (int l = 1; l < h->dnlev; ++l) {
        const Geom gs = h->dlv[l - 1], gd = h->dlv[l];
        const double sx = 1.0 / ((double)gd.w / gs.w), sy = 1.0 / ((double)gd.h / gs.h);
        hipLaunchKernelGGL(k_pyr_down, grid64x4(gd, F), dim3(256), 0, s, h->dpyr_base + h->dpyr_off[l - 1], gs, h->dpyr_base + h->dpyr_off[l], gd, sx, sy);
    }
    const int L = h->dnlev - 1;
    int cur = 0;
    HIPC(h, hipMemset2DAsync(h->df.Wu[0], (size_t)g0.plane * sizeof(float), 0, (size_t)h->dlv[L].plane * sizeof(float), B, s));
    HIPC(h, hipMemset2DAsync(h->df.Wv[0], (size_t)g0.plane * sizeof(float), 0, (size_t)h->dlv[L].plane * sizeof(float), B, s));
    const float mul = 1.0f / h->DP.downscale_factor;
    for (int l = L; l >= 0; --l) {
        const Geom g = h->dlv[l];
        df_refine_level(h, h->dpyr_base + h->dpyr_off[l], off0, off1, g, cur, B, s);
        if (l == 0) break;
        const Geom gd = h->dlv[l - 1];
        const double sx = 1.0 / ((double)gd.w / g.w), sy = 1.0 / ((double)gd.h / g.h);
        hipLaunchKernelGGL(k_df_up, grid64x4(gd, B), dim3(256), 0, s, h->df, cur, g, gd, sx, sy, mul);
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_df_out, grid64x4(g0, B), dim3(256), 0, s, h->df, g0, scale, dflow);
    HIPC(h, hipGetLastError());
    return TF_OK;
}

// algorithmic bytes of one DeepFlow pair (fp32 planes touched once per kernel)
double df_account_bytes(const tf_handle* h)
{
    double tb = 0;
    for (int l = 0; l < h->dnlev; ++l) {
        const double px = (double)h->dlv[l].w * h->dlv[l].h;
        const double per_fp = (10 + 3 + 6) * 4.0 /*data*/ + (4 + 3 + 4) * 4.0 /*smooth*/ + h->DP.sor_iterations * 2 * 10 * 4.0 /*SOR colour passes*/;
        tb += px * ((4 + 4) * 4.0 /*warp*/ + (2 + 4 + 2 + 3) * 4.0 /*grads*/ + h->DP.fixed_point_iterations * per_fp + 6 * 4.0 /*sum*/ + 4 * 4.0 /*up*/);
    }
    return tb;
}

enum Mode { MODE_PAIRS, MODE_SEQ };
// where a call's buffers live: bit 0 = the frames are device memory, bit 1 = the flow destination is
enum { W_HOST = 0, W_IN_DEV = 1, W_OUT_DEV = 2, W_DEV = 3 };

// common driver: device==true -> in/out pointers are device memory
int calc_common(tf_handle* h, Mode mode, const uint8_t* in0, const uint8_t* in1, int n_pairs, int H, int W, float scale,
                float* flow_out, int device, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (!in0 || (mode == MODE_PAIRS && !in1) || !flow_out) return fail(h, TF_ERR_INVALID_ARG, "null image/flow pointer");
    if (H < 1 || W < 1 || n_pairs < 1) return fail(h, TF_ERR_INVALID_ARG, "bad sizes: pairs=%d H=%d W=%d", n_pairs, H, W);
    if ((long long)H * W > (1LL << 24)) return fail(h, TF_ERR_UNSUPPORTED, "images above 2^24 pixels are not supported");
    const bool deep = h->P.algo == TF_ALGO_DEEPFLOW;
    int rc = deep ? df_validate(h, h->DP) : validate_params(h, h->P);
    if (rc) return rc;
    HIPC(h, hipSetDevice(h->dev));
    const double t0 = now_ms();
    rc = deep ? df_ensure_alloc(h, H, W, n_pairs) : ensure_alloc(h, H, W, n_pairs);
    if (rc) return rc;
    if (deep) h->cap = h->dcap;
    const size_t fpx = (size_t)H * W * (h->src_f32 ? 4 : 1);   // BYTES per frame (the flow offsets below use npx)
    const size_t npx = (size_t)H * W;
    h->last_iters.assign(deep ? 0 : (size_t)n_pairs * h->nlev * h->P.warps * 2, 0);
    h->last_pairs = n_pairs; h->last_nlev = deep ? h->dnlev : h->nlev; h->last_warps = deep ? 0 : h->P.warps;
    h->iter_launches = 0; h->prof_used = 0; h->df_sor_bytes = 0; h->df_sor_px = 0;
    float ms_h2d = 0, ms_dev = 0, ms_d2h = 0;
    // Host destinations that are pinned (tf_host_alloc, hipHostMalloc, hipHostRegister) take the overlapped path: each
    // sub-batch is solved into one half of a double staging buffer and copied out on a second stream while the next one
    // is being solved.  Pageable destinations keep the simple in-order path.
    bool overlap = false;
    const bool in_dev = device & W_IN_DEV, out_dev = device & W_OUT_DEV;
    if (!out_dev) {
        hipPointerAttribute_t pa;
        if (hipPointerGetAttributes(&pa, flow_out) == hipSuccess && pa.type == hipMemoryTypeHost) overlap = true;
        else (void)hipGetLastError();
    }
    int step = h->cap;
    if (overlap && n_pairs >= 48 && h->sub_batches > 1) {
        step = (n_pairs + h->sub_batches - 1) / h->sub_batches;
        if (step < 16) step = 16;
        if (step > h->cap) step = h->cap;
    }
    if (overlap && !h->copy_stream) HIPC(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (overlap && !h->cev[0])
        for (auto& e : h->cev) HIPC(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const size_t flow_half = (size_t)step * npx * 2;          // floats per staging half
    if (!in_dev || !out_dev) {
        rc = ensure_staging(h, in_dev ? 0 : 2 * (size_t)h->cap * fpx, out_dev ? 0 : (overlap ? 2 : 1) * flow_half * sizeof(float));
        if (rc) return rc;
    }
    int kb = 0;
    bool repeating = false;
    for (int c0 = 0; c0 < n_pairs; c0 += step, ++kb) {
        const int nb = n_pairs - c0 < step ? n_pairs - c0 : step;
        const uint8_t* dfr; int F, off0, off1;
        float* dfl;
        HIPC(h, hipEventRecord(h->ev[0], h->stream));
        if (mode == MODE_SEQ) {
            F = nb + 1; off0 = 0; off1 = 1;
            if (in_dev) dfr = in0 + (size_t)c0 * fpx;
            else { HIPC(h, hipMemcpyAsync(h->st_u8, in0 + (size_t)c0 * fpx, (size_t)F * fpx, hipMemcpyHostToDevice, h->stream)); dfr = h->st_u8; }
        } else {
            F = 2 * nb; off0 = 0; off1 = nb;
            if (in_dev && n_pairs <= h->cap && in1 == in0 + (size_t)n_pairs * fpx) dfr = in0;   // already [I0s|I1s] contiguous
            else {
                if (in_dev) { rc = ensure_staging(h, 2 * (size_t)h->cap * fpx, 0); if (rc) return rc; }
                const hipMemcpyKind k = in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
                HIPC(h, hipMemcpyAsync(h->st_u8, in0 + (size_t)c0 * fpx, (size_t)nb * fpx, k, h->stream));
                HIPC(h, hipMemcpyAsync(h->st_u8 + (size_t)nb * fpx, in1 + (size_t)c0 * fpx, (size_t)nb * fpx, k, h->stream));
                dfr = h->st_u8;
            }
        }
        dfl = out_dev ? flow_out + (size_t)c0 * npx * 2 : h->st_flow + (overlap ? (size_t)(kb & 1) * flow_half : 0);
        if (overlap && kb >= 2) HIPC(h, hipStreamWaitEvent(h->stream, h->cev[2 + (kb & 1)], 0));   // that half's last copy-out
        HIPC(h, hipEventRecord(h->ev[1], h->stream));
        // what a repeat of this sub-batch must not count twice (an aborted co-resident attempt is void)
        const unsigned long long snap_launches = h->iter_launches;
        const double snap_bytes = h->df_sor_bytes, snap_px = h->df_sor_px;
        const size_t snap_prof = h->prof_used;
        rc = deep ? df_solve_resident(h, dfr, F, nb, off0, off1, scale, dfl)
                  : solve_resident(h, dfr, F, nb, off0, off1, scale, dfl);
        if (rc) return rc;
        HIPC(h, hipEventRecord(h->ev[2], h->stream));
        if (overlap) {
            HIPC(h, hipEventRecord(h->cev[kb & 1], h->stream));
            HIPC(h, hipStreamWaitEvent(h->copy_stream, h->cev[kb & 1], 0));
            HIPC(h, hipMemcpyAsync(flow_out + (size_t)c0 * npx * 2, dfl, (size_t)nb * npx * 2 * sizeof(float), hipMemcpyDeviceToHost, h->copy_stream));
            HIPC(h, hipEventRecord(h->cev[2 + (kb & 1)], h->copy_stream));
        } else if (!out_dev)
            HIPC(h, hipMemcpyAsync(flow_out + (size_t)c0 * npx * 2, h->st_flow, (size_t)nb * npx * 2 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        if (!deep)
            HIPC(h, hipMemcpyAsync(h->last_iters.data() + (size_t)c0 * h->nlev * h->P.warps * 2, h->iters_dev,
                                   (size_t)nb * h->nlev * h->P.warps * 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipEventRecord(h->ev[3], h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        if (deep) {
            bool aborted = false;
            rc = coop_aborted(h, &aborted);
            if (rc) return rc;
            if (aborted) {                                       // solve this sub-batch again, tiled (what was copied out is overwritten)
                if (overlap) HIPC(h, hipStreamSynchronize(h->copy_stream));
                h->iter_launches = snap_launches; h->df_sor_bytes = snap_bytes; h->df_sor_px = snap_px; h->prof_used = snap_prof;
                c0 -= step; --kb;
                repeating = true;
                continue;
            }
            if (!repeating) coop_tick(h);                        // the repeat of an aborted sub-batch does not count towards the back-off
            repeating = false;
        }
        float t;
        HIPC(h, hipEventElapsedTime(&t, h->ev[0], h->ev[1])); ms_h2d += t;
        HIPC(h, hipEventElapsedTime(&t, h->ev[1], h->ev[2])); ms_dev += t;
        HIPC(h, hipEventElapsedTime(&t, h->ev[2], h->ev[3])); ms_d2h += t;
    }
    if (overlap) HIPC(h, hipStreamSynchronize(h->copy_stream));
    if (st) {
        memset(st, 0, sizeof *st);
        st->n_pairs = n_pairs; st->nscales_used = deep ? h->dnlev : h->nlev; st->warps = deep ? 0 : h->P.warps;
        st->ms_h2d = ms_h2d; st->ms_device = ms_dev; st->ms_d2h = ms_d2h;
        st->iter_launches = h->iter_launches;
        if (deep) { st->total_bytes = df_account_bytes(h) * n_pairs; st->iter_bytes = h->df_sor_bytes; st->iter_pair_steps = (unsigned long long)h->df_sor_px; }
        for (int b = 0; !deep && b < n_pairs; ++b)
            account_bytes(h, h->last_iters.data() + (size_t)b * h->nlev * h->P.warps * 2, &st->iter_bytes, &st->total_bytes,
                          &st->inner_iters_total, &st->outer_iters_total);
        if (!deep) st->iter_pair_steps = st->inner_iters_total;     // DeepFlow: pixels x pairs summed over the SOR launches (set above)
        double ims = 0;
        h->warp_ms = h->median_ms = 0;
        for (size_t i = 0; i < h->prof_used; ++i) {
            float t = 0;
            ProfEv& pe = h->prof_pool[i];
            HIPC(h, hipEventElapsedTime(&t, pe.a, pe.b));
            pe.ms = t;
            if (pe.level == -4) h->warp_ms += t;
            else if (pe.level == -5) h->median_ms += t;
            else ims += t;
        }
        st->iter_ms = ims;
        st->ms_warp = h->warp_ms; st->ms_median = h->median_ms; st->ms_misc = 0; st->ms_sched = 0;
        st->ms_total = now_ms() - t0;
    }
    return TF_OK;
}

// ---- do two streams run beside each other? --------------------------------------------------------------------------------------
// HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and work of two streams that share
// one is serialised.  A lane exists to run BESIDE the others: with two lanes' streams on one hardware queue the queue form measured
// 2580 instead of 2830 pairs/s (and the two-lane split 2170 instead of 2630 with GPU_MAX_HW_QUEUES=2).  Which queue a new stream lands
// on depends on every other stream the process holds (torch's, copy streams, idle handles), so it is probed, not assumed: a kernel on
// stream A waits (bounded: ~0.5 ms) for a flag that a kernel on stream B sets; it sees the flag only if B's kernel could start while
// A's was running.
__global__ void k_probe_wait(int* flag, int* seen, long long ticks)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();       // 100 MHz
    int v = 0;
    while (!(v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && (long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    *seen = v;
}
__global__ void k_probe_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 1 = b's kernel ran while a's was running, 0 = it did not (shared hardware queue), -1 = the probe itself failed
int streams_concurrent(hipStream_t a, hipStream_t b)
{
    if (a == b) return 0;
    int* d = nullptr;
    if (hipMalloc(&d, 2 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return -1; }
    int seen = -1;
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(int), a);
    if (e == hipSuccess) e = hipStreamSynchronize(a);
    if (e == hipSuccess) e = hipStreamSynchronize(b);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, d, d + 1, 50000LL);
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, d);
        e = hipStreamSynchronize(b);
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess) e = hipMemcpy(&seen, d + 1, sizeof(int), hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
    return seen == 1 ? 1 : 0;
}

// Give handle `t` a solve stream that runs beside every stream in `others` (new streams are tried until one does; the rejected ones are
// kept until the search is over, so that the next one lands elsewhere, then destroyed).  Returns true when t's stream is concurrent
// with all of them; false leaves the last stream tried (work still runs, serialised with one of the others).
bool give_concurrent_stream(tf_handle* t, const std::vector<hipStream_t>& others, int* retries)
{
    auto ok_with_all = [&](hipStream_t s) {
        for (hipStream_t o : others) if (streams_concurrent(o, s) == 0) return false;
        return true;
    };
    if (ok_with_all(t->own_stream)) return true;
    std::vector<hipStream_t> rejected;
    bool ok = false;
    for (int k = 0; k < 8 && !ok; ++k) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        if (retries) ++*retries;
        if (ok_with_all(s)) {
            (void)hipStreamSynchronize(t->own_stream);
            rejected.push_back(t->own_stream);
            if (t->stream == t->own_stream) t->stream = s;
            t->own_stream = s;
            ok = true;
        } else rejected.push_back(s);
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    return ok;
}

void merge_stats(tf_stats* st, const tf_stats& sb)
{
    st->ms_total = std::max(st->ms_total, sb.ms_total);
    st->ms_h2d = std::max(st->ms_h2d, sb.ms_h2d);
    st->ms_device = std::max(st->ms_device, sb.ms_device);
    st->ms_d2h = std::max(st->ms_d2h, sb.ms_d2h);
    st->iter_launches += sb.iter_launches; st->iter_pair_steps += sb.iter_pair_steps; st->iter_ms += sb.iter_ms;
    st->iter_bytes += sb.iter_bytes; st->total_bytes += sb.total_bytes;
    st->ms_warp += sb.ms_warp; st->ms_median += sb.ms_median; st->ms_misc += sb.ms_misc; st->ms_sched += sb.ms_sched;
    st->inner_iters_total += sb.inner_iters_total; st->outer_iters_total += sb.outer_iters_total;
}

// One sub-batch-sized call: optionally split over several lanes (this handle + twins with their own stream, buffers and host
// thread, joined at the end): the launch gaps and thin tail launches of one lane are filled by the others.
// Whatever went wrong, nothing of the failed call may still be in flight when the caller gets its buffers back (a D2H
// copy into flow_out on the copy stream, kernels writing the caller's device buffer): drain every stream of the handle.
int calc_common_guarded(tf_handle* h, Mode mode, const uint8_t* in0, const uint8_t* in1, int n_pairs, int H, int W, float scale,
                        float* flow_out, int device, tf_stats* st)
{
    const int rc = calc_common(h, mode, in0, in1, n_pairs, H, W, scale, flow_out, device, st);
    if (rc != TF_OK && h) {
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
        (void)hipGetLastError();
    }
    return rc;
}

int calc_split(tf_handle* h, Mode mode, const uint8_t* in0, const uint8_t* in1, int n_pairs, int H, int W, float scale,
               float* flow_out, int device, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    int L = h->lanes;
    while (L > 1 && n_pairs / L < 16) --L;                   // a lane needs a batch worth its launches
    CoopClaim claim(h->is_twin || h->P.algo != TF_ALGO_DEEPFLOW ? -1 : h->dev);
    if (!h->is_twin) h->coop_share = claim.ok ? h->num_cus : 0;
    if (L < 2 || h->is_twin || !in0 || !flow_out || (mode == MODE_PAIRS && !in1) || H < 1 || W < 1 || h->stream != h->own_stream)
        return calc_common_guarded(h, mode, in0, in1, n_pairs, H, W, scale, flow_out, device, st);
    while ((int)h->twins.size() < L - 1) {
        tf_handle* t = nullptr;
        int rc = h->P.algo == TF_ALGO_DEEPFLOW ? tf_create_deepflow(&h->DP, h->dev, &t) : tf_create(&h->P, h->dev, &t);
        if (rc) return fail(h, rc, "creating lane %d failed: %s", (int)h->twins.size() + 2, tf_last_error(nullptr));
        t->is_twin = true;
        std::vector<hipStream_t> others{h->own_stream};
        for (tf_handle* o : h->twins) others.push_back(o->own_stream);
        if (!give_concurrent_stream(t, others, &h->stream_retries)) h->streams_serialised |= 1;
        h->twins.push_back(t);
    }
    const size_t npx = (size_t)H * W, fpx = npx * (h->src_f32 ? 4 : 1);      // fpx in bytes
    std::vector<tf_stats> ss((size_t)L);
    std::vector<int> rcs((size_t)L, TF_OK), first((size_t)L + 1, 0);
    for (int k = 0; k <= L; ++k) first[k] = (int)((long long)n_pairs * k / L);
    std::vector<std::thread> th;
    for (int k = 1; k < L; ++k) {
        tf_handle* t = h->twins[k - 1];
        t->P = h->P; t->DP = h->DP; t->src_f32 = h->src_f32;
        static_cast<TfKnobs&>(*t) = static_cast<const TfKnobs&>(*h);          // every knob, one assignment
        if (t->coop_flags) coop_query_occupancy(t);                           // asks again only if the test overrides have changed
        t->coop_share = (claim.ok ? h->num_cus : 0) / L;
        // pairs [first[k], first[k+1]); in sequence mode the lane's frames start at its first pair (one frame of overlap)
        const uint8_t* b0 = in0 + (size_t)first[k] * fpx;
        const uint8_t* b1 = mode == MODE_SEQ ? nullptr : in1 + (size_t)first[k] * fpx;
        const int nb = first[k + 1] - first[k];
        float* fo = flow_out + (size_t)first[k] * npx * 2;
        th.emplace_back([=, &ss, &rcs] { rcs[k] = calc_common_guarded(t, mode, b0, b1, nb, H, W, scale, fo, device, &ss[k]); });
    }
    h->coop_share = (claim.ok ? h->num_cus : 0) / L;
    rcs[0] = calc_common_guarded(h, mode, in0, in1, first[1], H, W, scale, flow_out, device, &ss[0]);
    for (auto& x : th) x.join();
    if (rcs[0]) return rcs[0];
    for (int k = 1; k < L; ++k)
        if (rcs[k]) return fail(h, rcs[k], "%s", h->twins[k - 1]->err.c_str());
    for (int k = 1; k < L; ++k) h->last_iters.insert(h->last_iters.end(), h->twins[k - 1]->last_iters.begin(), h->twins[k - 1]->last_iters.end());
    h->last_pairs = n_pairs;
    if (st) {
        *st = ss[0];
        st->n_pairs = n_pairs;
        for (int k = 1; k < L; ++k) merge_stats(st, ss[k]);
    }
    return TF_OK;
}


// ---- the queue (structs above tf_create's prototype) ---------------------------------------------------------------------------
int queue_lane_count(const tf_handle* h)
{
    if (h->queue_lanes >= 0) return h->queue_lanes > 8 ? 8 : h->queue_lanes;
    return h->P.algo == TF_ALGO_DEEPFLOW ? 1 : 3;           // DeepFlow's co-resident SOR launches take every CU: one lane (its units are split over two twins)
}
int queue_unit_pairs(const tf_handle* h)
{
    const int mb = h->P.algo == TF_ALGO_DEEPFLOW ? h->DP.max_batch : h->P.max_batch;
    const int cap = mb > 0 ? mb : DEFAULT_MAX_BATCH;
    return h->queue_unit > 0 && h->queue_unit < cap ? h->queue_unit : cap;
}

void lane_worker(tf_handle* owner, LanePool* pool, tf_handle* lane)
{
    for (;;) {
        QJob* j; int u; bool skip;
        {
            std::unique_lock<std::mutex> lk(pool->m);
            pool->cv_work.wait(lk, [&] { return pool->stop || !pool->jobs.empty(); });
            if (pool->jobs.empty()) return;                  // stop was asked for and nothing is left to hand out
            j = pool->jobs.front();
            u = j->next++;
            if (j->next >= j->n_units) pool->jobs.pop_front();
            skip = j->rc != TF_OK;                           // a unit of this job has failed: the rest is not started
        }
        tf_stats us; memset(&us, 0, sizeof us);
        int rc = TF_OK;
        const int c0 = u * j->unit, nb = j->n_pairs - c0 < j->unit ? j->n_pairs - c0 : j->unit;
        if (!skip) {
            const size_t npx = (size_t)j->H * j->W, fpx = npx * (j->src_f32 ? 4 : 1);
            lane->P = j->P; lane->DP = j->DP; lane->src_f32 = j->src_f32; lane->lanes = j->split_lanes;
            static_cast<TfKnobs&>(*lane) = j->knobs;
            if (lane->coop_flags) coop_query_occupancy(lane);        // asks again only if the test overrides have changed
            if (u == j->fail_unit) rc = fail(lane, TF_ERR_HIP, "injected failure (queue_test_fail_unit)");
            else rc = calc_split(lane, (Mode)j->mode, j->in0 + (size_t)c0 * fpx, j->in1 ? j->in1 + (size_t)c0 * fpx : nullptr, nb, j->H, j->W, j->scale,
                                 j->out + (size_t)c0 * npx * 2, j->device, &us);
            lane->src_f32 = 0;
        }
        bool last;
        {
            std::lock_guard<std::mutex> lk(pool->m);
            if (skip) ++owner->q_units_skipped; else if (rc != TF_OK) ++owner->q_units_failed; else ++owner->q_units_done;
            if (!skip && rc != TF_OK) {
                if (j->rc == TF_OK) {
                    j->rc = rc;
                    char where[96]; snprintf(where, sizeof where, "sub-batch %d (pairs %d..%d): ", u, c0, c0 + nb - 1);
                    j->err = std::string(where) + lane->err;
                }
            } else if (!skip) {
                if (!j->merged++) { const double t = j->st.ms_total; j->st = us; j->st.ms_total = t; } else merge_stats(&j->st, us);
                if (j->per_pair && lane->last_iters.size() == (size_t)nb * j->per_pair)
                    memcpy(j->iters.data() + (size_t)c0 * j->per_pair, lane->last_iters.data(), (size_t)nb * j->per_pair * sizeof(int));
            }
            last = ++j->done == j->n_units;
        }
        if (last) {
            // every lane that worked for this job has drained its streams (a solve is host-synchronous, a failed one drains in
            // calc_common_guarded): nothing of the job is in flight any more
            if (j->on_done) j->on_done(j);
            std::lock_guard<std::mutex> lk(pool->m);
            j->finished = true;
            --pool->outstanding;
            pool->cv_done.notify_all();                      // (the waiter may free the job from here on)
        }
    }
}

void pool_destroy(tf_handle* h)
{
    LanePool* pool = h->pool;
    if (!pool) return;
    {
        std::lock_guard<std::mutex> lk(pool->m);
        pool->stop = true;                                   // the lanes first finish what is queued
    }
    pool->cv_work.notify_all();
    for (auto& t : pool->th) t.join();
    for (tf_handle* l : pool->lanes) tf_destroy(l);
    delete pool;
    h->pool = nullptr;
}

// the handle's lanes, made on first use (and again when the "queue_lanes" knob has changed and nothing is queued)
int pool_ensure(tf_handle* h)
{
    const int want = queue_lane_count(h);
    if (h->pool && (int)h->pool->lanes.size() != want) {
        bool idle;
        { std::lock_guard<std::mutex> lk(h->pool->m); idle = h->pool->outstanding == 0; }
        if (idle) pool_destroy(h);
    }
    if (h->pool) return TF_OK;
    LanePool* pool = new LanePool();
    for (int k = 0; k < want; ++k) {
        tf_handle* t = nullptr;
        const int rc = h->P.algo == TF_ALGO_DEEPFLOW ? tf_create_deepflow(&h->DP, h->dev, &t) : tf_create(&h->P, h->dev, &t);
        if (rc) {
            for (tf_handle* l : pool->lanes) tf_destroy(l);
            delete pool;
            return fail(h, rc, "creating queue lane %d failed: %s", k + 1, tf_last_error(nullptr));
        }
        t->is_lane = true; t->owner = h;
        std::vector<hipStream_t> others;
        for (tf_handle* o : pool->lanes) others.push_back(o->own_stream);
        if (!give_concurrent_stream(t, others, &h->stream_retries)) h->streams_serialised |= 1;
        pool->lanes.push_back(t);
    }
    // A lane's copy-out (pinned host destinations: D2H of one unit under the next unit's solve) must run beside EVERY lane's solve, its own
    // included: each lane gets a copy stream on a hardware queue none of the solve streams uses (all lanes are idle here; with HIP's four
    // hardware queues that is the fourth one, which the copy streams then share among themselves -- PCIe is one resource anyway).
    for (tf_handle* l : pool->lanes) {
        std::vector<hipStream_t> rejected;
        bool ok = false;
        for (int k = 0; k < 5 && !ok; ++k) {
            hipStream_t cs = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            ok = true;
            for (tf_handle* o : pool->lanes) if (streams_concurrent(o->own_stream, cs) == 0) { ok = false; break; }
            if (ok || k == 4) {
                l->copy_stream = cs;
                for (auto& e : l->cev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) (void)hipGetLastError();
                if (!ok) h->streams_serialised |= 2;         // bit 1: a copy stream shares a hardware queue with a solve stream
                break;
            }
            rejected.push_back(cs);
            ++h->stream_retries;
        }
        for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    }
    h->pool = pool;
    for (tf_handle* l : pool->lanes) pool->th.emplace_back(lane_worker, h, pool, l);
    return TF_OK;
}

// fills the job from the handle's current settings and hands it to the lanes
int queue_submit(tf_handle* h, QJob* j, Mode mode, const uint8_t* in0, const uint8_t* in1, int n_pairs, int H, int W, float scale, float* flow_out, int device,
                 bool balance)
{
    const bool deep = h->P.algo == TF_ALGO_DEEPFLOW;
    int rc = deep ? df_validate(h, h->DP) : validate_params(h, h->P);
    if (rc) return rc;
    if ((long long)H * W > (1LL << 24)) return fail(h, TF_ERR_UNSUPPORTED, "images above 2^24 pixels are not supported");
    rc = pool_ensure(h);
    if (rc) return rc;
    j->mode = mode; j->in0 = in0; j->in1 = in1; j->n_pairs = n_pairs; j->H = H; j->W = W; j->scale = scale; j->out = flow_out; j->device = device;
    j->src_f32 = h->src_f32; j->P = h->P; j->DP = h->DP; j->knobs = static_cast<const TfKnobs&>(*h);
    j->split_lanes = deep ? h->lanes : 1;
    j->unit = queue_unit_pairs(h);
    if (h->queue_unit <= 0 && balance) {
        // A synchronous call on an empty queue ends with its lanes draining.  Equal units, a multiple of the lane count of them: 512 pairs as 4 x 128 leave two lanes idle while the third solves its second
        // unit; as 6 x 86 every lane gets two.  200 pairs: 2629-2646 pairs/s as 128 + 72, 2772-2796 as 3 x 67; 448: 2734-2786 -> 2780-2824;
        // 640: 2867-2877 -> 2915-2922; 1024 (8 x 128 against 9 x 114): a tie (gpurun_out/r5o).  Smaller units cost a few per cent each
        // (6 x 64 for 384 pairs: - 5 %), which is why the count is the SMALLEST multiple of the lanes whose units fit a sub-batch.
        const int L = queue_lane_count(h) > 0 ? queue_lane_count(h) : 1;
        const int U = L * ((n_pairs + L * j->unit - 1) / (L * j->unit));
        j->unit = (n_pairs + U - 1) / U;
    }   // (a submitted job's last units run beside the next job's first: whole sub-batches, which are the more efficient units)
    j->n_units = (n_pairs + j->unit - 1) / j->unit;
    j->fail_unit = h->queue_test_fail_unit; h->queue_test_fail_unit = -1;
    memset(&j->st, 0, sizeof j->st);
    if (deep) {
        std::vector<Geom> lv(DF_MAXLEV);
        j->nlev = df_levels(h->DP, H, W, lv.data()); j->warps = 0; j->per_pair = 0;
    } else {
        Geom lv[MAXLEV];
        j->nlev = compute_levels(h->P, H, W, lv); j->warps = h->P.warps; j->per_pair = (size_t)j->nlev * j->warps * 2;
    }
    j->iters.assign((size_t)n_pairs * j->per_pair, 0);
    j->t0 = now_ms();
    {
        std::lock_guard<std::mutex> lk(h->pool->m);
        h->pool->jobs.push_back(j);
        ++h->pool->outstanding;
        ++h->q_jobs;
    }
    h->pool->cv_work.notify_all();
    return TF_OK;
}

// waits for the job and moves its results to where a synchronous call leaves them (tf_get_iters, tf_last_error, *st)
int queue_finish(tf_handle* h, QJob* j, tf_stats* st)
{
    if (h->pool) {
        std::unique_lock<std::mutex> lk(h->pool->m);
        h->pool->cv_done.wait(lk, [&] { return j->finished; });
    }
    h->last_iters = std::move(j->iters);
    h->last_pairs = j->n_pairs; h->last_nlev = j->nlev; h->last_warps = j->warps;
    if (j->rc != TF_OK) { h->err = j->err; return j->rc; }
    if (st) {
        *st = j->st;
        st->n_pairs = j->n_pairs; st->nscales_used = j->nlev; st->warps = j->warps;
        st->ms_total = now_ms() - j->t0;
    }
    return TF_OK;
}

// Entry used by the C ABI.  A call of at most one sub-batch is solved by this handle (calc_split: contiguous parts on its twins,
// joined).  A larger one -- or any call while tf_submit_* jobs are in flight -- goes to the queue: whole sub-batches, taken by the
// lanes as they come free; the call returns when every lane has finished its last unit of it (a failed unit stops the job's
// remaining units from starting; the units already running complete, so nothing of the call is in flight when it returns).
int calc_entry(tf_handle* h, Mode mode, const uint8_t* in0, const uint8_t* in1, int n_pairs, int H, int W, float scale,
               float* flow_out, int device, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    const bool can_queue = !h->is_twin && !h->is_lane && queue_lane_count(h) > 0 && in0 && flow_out && (mode != MODE_PAIRS || in1) && H >= 1 && W >= 1 &&
                           n_pairs >= 1 && h->stream == h->own_stream;
    bool busy = false;
    if (can_queue && h->pool) { std::lock_guard<std::mutex> lk(h->pool->m); busy = h->pool->outstanding > 0; }
    if (!can_queue || (!busy && n_pairs <= queue_unit_pairs(h)))
        return calc_split(h, mode, in0, in1, n_pairs, H, W, scale, flow_out, device, st);
    QJob j;
    int rc = queue_submit(h, &j, mode, in0, in1, n_pairs, H, W, scale, flow_out, device, !busy);
    if (rc) return rc;
    return queue_finish(h, &j, st);
}

// tf_submit_*: the same job, not waited for.  Returns a ticket for tf_wait.
int submit_entry(tf_handle* h, Mode mode, const uint8_t* in0, const uint8_t* in1, int n_pairs, int H, int W, float scale, float* flow_out, int device, int* ticket,
                 void* owned_dev = nullptr)
{
    struct Guard { void* p; ~Guard() { if (p) (void)hipFree(p); } } guard{owned_dev};
    if (!h || !ticket) return TF_ERR_INVALID_ARG;
    if (!in0 || (mode == MODE_PAIRS && !in1) || !flow_out) return fail(h, TF_ERR_INVALID_ARG, "null image/flow pointer");
    if (H < 1 || W < 1 || n_pairs < 1) return fail(h, TF_ERR_INVALID_ARG, "bad sizes: pairs=%d H=%d W=%d", n_pairs, H, W);
    if (h->is_twin || h->is_lane || h->stream != h->own_stream) return fail(h, TF_ERR_UNSUPPORTED, "tf_submit_* needs the handle's own stream");
    guard.p = nullptr;                                       // from here on the job owns it
    QJob* j = new QJob();
    j->owned_dev = owned_dev;                                // (freed with the job, whatever happens below)
    int rc;
    if (queue_lane_count(h) < 1) {                           // "queue_lanes" = 0: no lanes, the job is done when the call returns
        tf_stats st;
        rc = calc_split(h, mode, in0, in1, n_pairs, H, W, scale, flow_out, device, &st);
        if (rc) { delete j; return rc; }
        j->st = st; j->n_pairs = n_pairs; j->nlev = h->last_nlev; j->warps = h->last_warps; j->iters = h->last_iters; j->t0 = now_ms() - st.ms_total; j->finished = true;
    } else {
        rc = queue_submit(h, j, mode, in0, in1, n_pairs, H, W, scale, flow_out, device, false);
        if (rc) { delete j; return rc; }
    }
    *ticket = h->next_ticket++;
    h->tickets[*ticket] = j;
    return TF_OK;
}
// ---- small RAII device buffer for the tf_dbg_* hooks ---------------------------------------------
struct DBuf {
    float* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
};

int dbg_up(tf_handle* h, DBuf& d, const float* src, const Geom& g)
{
    HIPC(h, hipMalloc(&d.p, (size_t)g.plane * sizeof(float)));
    // stream-ordered on the handle's (non-blocking) stream: legacy-stream copies would race with its kernels
    HIPC(h, hipMemsetAsync(d.p, 0, (size_t)g.plane * sizeof(float), h->stream));
    if (src) HIPC(h, hipMemcpy2DAsync(d.p, (size_t)g.pitch * 4, src, (size_t)g.w * 4, (size_t)g.w * 4, g.h, hipMemcpyHostToDevice, h->stream));
    return TF_OK;
}
int dbg_down(tf_handle* h, float* dst, const float* d, const Geom& g)
{
    HIPC(h, hipMemcpy2DAsync(dst, (size_t)g.w * 4, d, (size_t)g.pitch * 4, (size_t)g.w * 4, g.h, hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return TF_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
TF_API int tf_abi_version(void) { return TF_ABI_VERSION; }

TF_API int tf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

TF_API int tf_default_params(tf_params* p)
{
    if (!p) return TF_ERR_INVALID_ARG;
    p->tau = 0.25; p->lambda = 0.15; p->theta = 0.3; p->epsilon = 0.01; p->scale_step = 0.8; p->gamma = 0.0;
    p->nscales = 5; p->warps = 5; p->inner_iterations = 30; p->outer_iterations = 10; p->median_filtering = 5;
    p->use_initial_flow = 0; p->algo = TF_ALGO_TVL1; p->max_batch = 0; p->variant = TF_VARIANT_CPU;
    return TF_OK;
}

TF_API int tf_default_deepflow_params(tf_deepflow_params* p)
{
    if (!p) return TF_ERR_INVALID_ARG;
    p->sigma = 0.6f; p->min_size = 25; p->downscale_factor = 0.95f; p->fixed_point_iterations = 5; p->sor_iterations = 25;
    p->alpha = 1.0f; p->delta = 0.5f; p->gamma = 5.0f; p->omega = 1.6f; p->zeta = 0.1f; p->epsilon = 0.001f; p->max_batch = 0;
    return TF_OK;
}

TF_API int tf_create_deepflow(const tf_deepflow_params* p, int device_id, tf_handle** out)
{
    tf_deepflow_params dp;
    if (p) dp = *p; else tf_default_deepflow_params(&dp);
    int rc = df_validate(nullptr, dp);
    if (rc) return rc;
    rc = tf_create(nullptr, device_id, out);
    if (rc) return rc;
    (*out)->P.algo = TF_ALGO_DEEPFLOW;
    (*out)->DP = dp;
    return TF_OK;
}

TF_API const char* tf_last_error(tf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

TF_API int tf_create(const tf_params* p, int device_id, tf_handle** out)
{
    if (!out) return fail(nullptr, TF_ERR_INVALID_ARG, "out == NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(nullptr, TF_ERR_NO_DEVICE, "no HIP device visible: libteeflow_hip has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(nullptr, TF_ERR_INVALID_ARG, "device_id %d out of range [0,%d)", device_id, n);
    tf_handle* h = new tf_handle();
    if (p) h->P = *p; else tf_default_params(&h->P);
    h->dev = device_id;
    int rc = validate_params(h, h->P);
    if (rc) { g_create_error = h->err; delete h; return rc; }
    auto bail = [&](hipError_t e, const char* what) {
        fail(nullptr, TF_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
        tf_destroy(h);
        return TF_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    h->stream = h->own_stream;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) h->num_cus = cus;
    }
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_iter2_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess)
        return bail(e, "hipFuncSetAttribute");
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_iter_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess)
        return bail(e, "hipFuncSetAttribute");
    for (auto& ev : h->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    {
        void* hp = nullptr; void* dp = nullptr;
        if ((e = hipHostMalloc(&hp, SLOT_RING * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e, "hipHostMalloc");
        h->slots_host = (volatile int*)hp;
        if ((e = hipHostGetDevicePointer(&dp, hp, 0)) != hipSuccess) return bail(e, "hipHostGetDevicePointer");
        h->slots_dev = (int*)dp;
        for (int i = 0; i < SLOT_RING; ++i) h->slots_host[i] = -1;
    }
    // bicubic coefficient table of cv::remap (interpolateCubic, A = -0.75, 1/32-px steps), float arithmetic
    float tab[128];
    {
        const float A = -0.75f, scale = 1.f / 32;
        for (int i = 0; i < 32; ++i) {
            const float x = i * scale;
            float* c = tab + i * 4;
            c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
            c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
            c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
            c[3] = 1.f - c[0] - c[1] - c[2];
        }
    }
    if ((e = hipMalloc(&h->tab, sizeof tab)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemcpyAsync(h->tab, tab, sizeof tab, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return bail(e, "hipMemcpy");
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = h;
    return TF_OK;
}

TF_API void tf_destroy(tf_handle* h)
{
    if (!h) return;
    pool_destroy(h);                                     // the lanes finish what is queued, then go
    for (auto& kv : h->tickets) delete kv.second;
    h->tickets.clear();
    for (tf_handle* t : h->twins) tf_destroy(t);
    h->twins.clear();
    (void)hipSetDevice(h->dev);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_buffers(h);
    if (h->tab) (void)hipFree(h->tab);
    if (h->slots_host) (void)hipHostFree((void*)h->slots_host);
    for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& pe : h->prof_pool) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    (void)tf_comm_destroy(h);
    for (auto& e : h->cev) if (e) (void)hipEventDestroy(e);
    for (auto& b : h->pre) if (b.p) (void)hipFree(b.p);
    if (h->wa) (void)hipFree(h->wa);
    if (h->wcnt) (void)hipFree(h->wcnt);
    if (h->woff) (void)hipFree(h->woff);
    if (h->wsum) (void)hipFree(h->wsum);
    if (h->wbg) (void)hipFree(h->wbg);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

TF_API int tf_set_param(tf_handle* h, int key, double v)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (h->P.algo == TF_ALGO_DEEPFLOW) return fail(h, TF_ERR_UNSUPPORTED, "DeepFlow handles have creation-time parameters only (as cv2's object)");
    tf_params p = h->P;
    switch (key) {
        case TF_PARAM_TAU: p.tau = v; break;
        case TF_PARAM_LAMBDA: p.lambda = v; break;
        case TF_PARAM_THETA: p.theta = v; break;
        case TF_PARAM_NSCALES: p.nscales = (int)v; break;
        case TF_PARAM_WARPS: p.warps = (int)v; break;
        case TF_PARAM_EPSILON: p.epsilon = v; break;
        case TF_PARAM_INNER_ITERATIONS: p.inner_iterations = (int)v; break;
        case TF_PARAM_OUTER_ITERATIONS: p.outer_iterations = (int)v; break;
        case TF_PARAM_SCALE_STEP: p.scale_step = v; break;
        case TF_PARAM_GAMMA: p.gamma = v; break;
        case TF_PARAM_MEDIAN_FILTERING: p.median_filtering = (int)v; break;
        case TF_PARAM_USE_INITIAL_FLOW: p.use_initial_flow = v != 0.0; break;
        default: return fail(h, TF_ERR_INVALID_ARG, "unknown parameter key %d", key);
    }
    int rc = validate_params(h, p);
    if (rc) return rc;
    h->P = p;
    return TF_OK;
}

TF_API int tf_get_param(tf_handle* h, int key, double* v)
{
    if (!h || !v) return TF_ERR_INVALID_ARG;
    const tf_params& p = h->P;
    switch (key) {
        case TF_PARAM_TAU: *v = p.tau; break;
        case TF_PARAM_LAMBDA: *v = p.lambda; break;
        case TF_PARAM_THETA: *v = p.theta; break;
        case TF_PARAM_NSCALES: *v = p.nscales; break;
        case TF_PARAM_WARPS: *v = p.warps; break;
        case TF_PARAM_EPSILON: *v = p.epsilon; break;
        case TF_PARAM_INNER_ITERATIONS: *v = p.inner_iterations; break;
        case TF_PARAM_OUTER_ITERATIONS: *v = p.outer_iterations; break;
        case TF_PARAM_SCALE_STEP: *v = p.scale_step; break;
        case TF_PARAM_GAMMA: *v = p.gamma; break;
        case TF_PARAM_MEDIAN_FILTERING: *v = p.median_filtering; break;
        case TF_PARAM_USE_INITIAL_FLOW: *v = p.use_initial_flow; break;
        default: return fail(h, TF_ERR_INVALID_ARG, "unknown parameter key %d", key);
    }
    return TF_OK;
}

TF_API int tf_set_stream(tf_handle* h, void* hip_stream, int use_external)
{
    if (!h) return TF_ERR_INVALID_ARG;
    HIPC(h, hipStreamSynchronize(h->stream));
    h->stream = use_external ? (hipStream_t)hip_stream : h->own_stream;
    return TF_OK;
}

TF_API int tf_set_tuning(tf_handle* h, const char* name, int value)
{
    if (!h || !name) return TF_ERR_INVALID_ARG;
    const std::string n(name);
    if (n == "iter_variant") {                      // 0 / 1 / 2 name a form of tvl1_iter (the one-wave-per-strip experiments 4 / 5 / 6 of round 4 are gone: DESIGN.md section 4c)
        if (value < 0 || value > 2) return fail(h, TF_ERR_UNSUPPORTED, "iter_variant must be 0 (64x16 tiles), 1 (row strips) or 2 (row strips, two iterations per launch), got %d", value);
        h->iter_variant = value;
    }
    else if (n == "strip_blocks") h->strip_blocks = value > 0 ? value : 2048;
    else if (n == "lag") h->lag = value < 0 ? DEFAULT_LAG : (value < SLOT_RING / 2 ? value : SLOT_RING / 2);   // 0 = wait for every launch's report (it is published at the launch's start); unread slots must never be overwritten
    else if (n == "min_rows_work") h->min_rows_work = value;
    else if (n == "force_ry") h->force_ry = value;
    else if (n == "adaptive_strips") h->adaptive_strips = value;
    else if (n == "dynamic_strips") h->dynamic_strips = value;
    else if (n == "slots") h->slots_override = value;
    else if (n == "lane_slots_pct") h->lane_slots_pct = value;
    else if (n == "tile_max_w") h->tile_max_w = value;
    else if (n == "sor_rt") h->sor_rt = value ? 1 : 0;
    else if (n == "sor_rt_shape") h->sor_rt_shape = value;
    else if (n == "sor_plain_div") h->sor_plain_div = value ? 1 : 0;
    else if (n == "tile2") h->tile2 = value;
    else if (n == "max_strip_width") h->max_strip_width = value < 4 ? 4 : (value > 2048 ? 2048 : value);
    else if (n == "sub_batches") h->sub_batches = value < 1 ? 1 : value;
    else if (n == "lanes") h->lanes = value < 1 ? 1 : (value > 8 ? 8 : value);
    else if (n == "queue_lanes") h->queue_lanes = value;          // -1 = per algorithm (3 DualTVL1, 1 DeepFlow), 0 = no queue: contiguous parts, joined
    else if (n == "queue_unit") h->queue_unit = value < 0 ? 0 : value;
    else if (n == "queue_test_fail_unit") h->queue_test_fail_unit = value;
    else if (n == "sor_fuse") h->sor_fuse = value;
    else if (n == "sor_coop") {                       // setting the knob re-arms the form at once and forgets the back-off
        h->sor_coop = value;
        if (value) {
            std::vector<tf_handle*> all{h};
            for (auto* t : h->twins) all.push_back(t);
            if (h->pool) for (auto* l : h->pool->lanes) { all.push_back(l); for (auto* t : l->twins) all.push_back(t); }
            for (auto* t : all) { t->coop_disabled = false; t->coop_backoff = t->coop_cooldown = 0; }
        }
    }
    else if (n == "sor_coop_s") h->sor_coop_s = value;
    else if (n == "sor_coop_small") h->sor_coop_small = value ? 1 : 0;
    else if (n == "sor_coop_min_util") h->sor_coop_min_util = value;
    else if (n == "coop_test_mute") h->coop_test_mute = value ? 1 : 0;
    else if (n == "coop_test_occ16") { h->coop_test_occ16 = value; h->coop_occ16 = -1; if (h->coop_flags) coop_query_occupancy(h); }
    else if (n == "coop_test_occ8") { h->coop_test_occ8 = value; h->coop_occ16 = -1; if (h->coop_flags) coop_query_occupancy(h); }
    else if (n == "coop_backoff") { h->coop_backoff = value < 0 ? 0 : value; }        // tests: next abort sits out 2 x this (0: the default 16)
    else if (n == "df_fuse_ds") h->df_fuse_ds = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "warp_margin") h->warp_margin = value < 0 ? 0 : (value > 40 ? 40 : value);
    else return fail(h, TF_ERR_INVALID_ARG, "unknown tuning knob %s", name);
    return TF_OK;
}

#ifdef TF_COOP_TIMING
extern "C" __attribute__((visibility("default"))) int tf_dbg_coop_times(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_t), sizeof(unsigned long long) * 4 * 64);
}
#endif
TF_API long long tf_dbg_counter(tf_handle* h, const char* name)
{
    if (!h || !name) return -1;
    const std::string n(name);
    long long v = -1;
    std::vector<tf_handle*> all{h};                              // this handle, its twins, its queue lanes and theirs
    for (auto* t : h->twins) all.push_back(t);
    if (h->pool) for (auto* l : h->pool->lanes) { all.push_back(l); for (auto* t : l->twins) all.push_back(t); }
    if (n == "coop_launches") { v = 0; for (auto* t : all) v += t->coop_launches; }
    else if (n == "coop_aborts") { v = 0; for (auto* t : all) v += t->coop_aborts; }
    else if (n == "coop_disabled") { v = 0; for (auto* t : all) v |= (long long)t->coop_disabled; }
    else if (n == "coop_rearms") { v = 0; for (auto* t : all) v += t->coop_rearms; }
    else if (n == "coop_cooldown") { v = 0; for (auto* t : all) v = v > t->coop_cooldown ? v : t->coop_cooldown; }
    else if (n == "saliency_kernel_us") v = (long long)(h->pre_kernel_ms * 1000.0);
    else if (n == "queue_jobs") v = h->q_jobs;
    else if (n == "stream_retries") { v = h->stream_retries; if (h->pool) for (auto* l : h->pool->lanes) v += l->stream_retries; }
    else if (n == "streams_serialised") { v = h->streams_serialised; if (h->pool) for (auto* l : h->pool->lanes) v |= l->streams_serialised; }
    else if (n == "queue_units_done" || n == "queue_units_skipped" || n == "queue_units_failed" || n == "queue_outstanding" || n == "queue_lanes") {
        v = 0;
        if (h->pool) {
            std::lock_guard<std::mutex> lk(h->pool->m);
            v = n == "queue_units_done" ? h->q_units_done : n == "queue_units_skipped" ? h->q_units_skipped : n == "queue_units_failed" ? h->q_units_failed : n == "queue_outstanding" ? h->pool->outstanding : (long long)h->pool->lanes.size();
        }
    }
    else if (n == "coop_occ16") v = h->coop_occ16;
    else if (n == "coop_occ8") v = h->coop_occ8;
    return v;
}

// the strip rule of k_iter2_rows (pure host arithmetic, no device call): rows per strip and strip count for n active pairs
TF_API void tf_dbg_strip_rule(int n_active, int H, int RY, int slots, int* R, int* S)
{
    int r = 0, sn = 0;
    strip_rule(n_active, H, RY, slots, &r, &sn);
    if (R) *R = r;
    if (S) *S = sn;
}

// per-launch record of the last profiled solve (tvl1_iter launches in issue order): level, warp, first iteration, ms
TF_API int tf_dbg_launch_profile(tf_handle* h, int* level, int* warp, int* it, float* ms, int max_n)
{
    if (!h) return -1;
    int n = 0;
    for (size_t i = 0; i < h->prof_used; ++i) {
        const ProfEv& pe = h->prof_pool[i];
        if (pe.level < 0) continue;                        // warp / median records
        if (n < max_n) {
            if (level) level[n] = pe.level;
            if (warp) warp[n] = pe.warp;
            if (it) it[n] = pe.it;
            if (ms) ms[n] = pe.ms;
        }
        ++n;
    }
    return n;
}

TF_API int tf_set_profile(tf_handle* h, int level)
{
    if (!h) return TF_ERR_INVALID_ARG;
    h->profile = level;
    return TF_OK;
}

TF_API int tf_calc_pair(tf_handle* h, const uint8_t* I0, const uint8_t* I1, int H, int W, float* flow_out, tf_stats* st)
{
    return calc_entry(h, MODE_PAIRS, I0, I1, 1, H, W, 1.0f, flow_out, W_HOST, st);
}

TF_API int tf_calc_pairs(tf_handle* h, const uint8_t* I0s, const uint8_t* I1s, int B, int H, int W, float* flow_out, tf_stats* st)
{
    return calc_entry(h, MODE_PAIRS, I0s, I1s, B, H, W, 1.0f, flow_out, W_HOST, st);
}

// CV_32FC1 frames (values in [0,1]; cv2 multiplies them by 255 when it builds level 0)
TF_API int tf_calc_pairs_f32(tf_handle* h, const float* I0s, const float* I1s, int B, int H, int W, float* flow_out, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    h->src_f32 = 1;
    const int rc = calc_entry(h, MODE_PAIRS, (const uint8_t*)I0s, (const uint8_t*)I1s, B, H, W, 1.0f, flow_out, W_HOST, st);
    h->src_f32 = 0;
    return rc;
}

TF_API int tf_calc_pair_f32(tf_handle* h, const float* I0, const float* I1, int H, int W, float* flow_out, tf_stats* st)
{
    return tf_calc_pairs_f32(h, I0, I1, 1, H, W, flow_out, st);
}

TF_API int tf_calc_seq(tf_handle* h, const uint8_t* frames, int N, int H, int W, float scale, float* flow_out, tf_stats* st)
{
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    return calc_entry(h, MODE_SEQ, frames, nullptr, N - 1, H, W, scale, flow_out, W_HOST, st);
}

TF_API int tf_calc_pairs_device(tf_handle* h, const uint8_t* dI0s, const uint8_t* dI1s, int B, int H, int W, float scale,
                                float* dflow_out, tf_stats* st)
{
    return calc_entry(h, MODE_PAIRS, dI0s, dI1s, B, H, W, scale, dflow_out, W_DEV, st);
}

TF_API int tf_calc_seq_device(tf_handle* h, const uint8_t* dframes, int N, int H, int W, float scale, float* dflow_out, tf_stats* st)
{
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    return calc_entry(h, MODE_SEQ, dframes, nullptr, N - 1, H, W, scale, dflow_out, W_DEV, st);
}

// ---- asynchronous forms: the job is queued on the handle's lanes and the call returns; tf_wait collects it ----------------------
TF_API int tf_submit_pairs_device(tf_handle* h, const uint8_t* dI0s, const uint8_t* dI1s, int B, int H, int W, float scale, float* dflow_out, int* ticket)
{
    return submit_entry(h, MODE_PAIRS, dI0s, dI1s, B, H, W, scale, dflow_out, W_DEV, ticket);
}
TF_API int tf_submit_seq_device(tf_handle* h, const uint8_t* dframes, int N, int H, int W, float scale, float* dflow_out, int* ticket)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    return submit_entry(h, MODE_SEQ, dframes, nullptr, N - 1, H, W, scale, dflow_out, W_DEV, ticket);
}
TF_API int tf_submit_pairs(tf_handle* h, const uint8_t* I0s, const uint8_t* I1s, int B, int H, int W, float* flow_out, int* ticket)
{
    return submit_entry(h, MODE_PAIRS, I0s, I1s, B, H, W, 1.0f, flow_out, W_HOST, ticket);
}
TF_API int tf_submit_seq(tf_handle* h, const uint8_t* frames, int N, int H, int W, float scale, float* flow_out, int* ticket)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    return submit_entry(h, MODE_SEQ, frames, nullptr, N - 1, H, W, scale, flow_out, W_HOST, ticket);
}
namespace { int condition_to_device(tf_handle* h, const uint8_t* rgb, int N, int H, int W, uint8_t** dgray_out, uint8_t* own); }
// tf_calc_seq_rgb without waiting: the frames are conditioned now (on the handle's stream, into a buffer the job owns), the solve is queued
TF_API int tf_submit_seq_rgb(tf_handle* h, const uint8_t* rgb, int N, int H, int W, float scale, float* flow_out, int* ticket)
{
    if (!h || !rgb || !flow_out || !ticket || H < 1 || W < 1) return TF_ERR_INVALID_ARG;
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    HIPC(h, hipSetDevice(h->dev));
    uint8_t* own = nullptr;
    HIPC(h, hipMalloc(&own, (size_t)N * H * W));
    uint8_t* dgray = nullptr;
    int rc = condition_to_device(h, rgb, N, H, W, &dgray, own);
    if (rc) { (void)hipFree(own); return rc; }
    return submit_entry(h, MODE_SEQ, own, nullptr, N - 1, H, W, scale, flow_out, W_IN_DEV, ticket, own);
}

TF_API int tf_wait(tf_handle* h, int ticket, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (ticket < 0) {                                        // every job not yet waited for, oldest first; the first failure is returned
        int rc_all = TF_OK; std::string err_all;
        while (!h->tickets.empty()) {
            const int rc = tf_wait(h, h->tickets.begin()->first, st);
            if (rc && !rc_all) { rc_all = rc; err_all = h->err; }
        }
        if (rc_all) h->err = err_all;
        return rc_all;
    }
    auto f = h->tickets.find(ticket);
    if (f == h->tickets.end()) return fail(h, TF_ERR_INVALID_ARG, "unknown ticket %d (already waited for?)", ticket);
    QJob* j = f->second;
    h->tickets.erase(f);
    const int rc = queue_finish(h, j, st);
    delete j;
    return rc;
}

namespace {
int pre_grow(tf_handle* h, int which, size_t bytes, void** out)
{
    tf_handle::GrowBuf& b = h->pre[which];
    if (b.cap < bytes) {
        if (b.p) { HIPC(h, hipStreamSynchronize(h->stream)); (void)hipFree(b.p); }
        b.p = nullptr; b.cap = 0;
        HIPC(h, hipMalloc(&b.p, bytes));
        b.cap = bytes;
    }
    *out = b.p;
    return TF_OK;
}

// rgb (host) -> conditioned gray frames in the handle's preprocessing buffer (valid until the handle's next preprocessing call)
// (`own`: into that caller-owned buffer instead -- a submitted job's frames must outlive the handle's next preprocessing call)
int condition_to_device(tf_handle* h, const uint8_t* rgb, int N, int H, int W, uint8_t** dgray_out, uint8_t* own = nullptr)
{
    const size_t npx = (size_t)H * W;
    uint8_t* drgb = nullptr; uint8_t* dgray = own; u64* mm = nullptr;
    HIPC(h, hipSetDevice(h->dev));
    int rc;
    if ((rc = pre_grow(h, tf_handle::PRE_SRC, (size_t)N * npx * 3, (void**)&drgb)) || (!own && (rc = pre_grow(h, tf_handle::PRE_OUT, (size_t)N * npx, (void**)&dgray))) ||
        (rc = pre_grow(h, tf_handle::PRE_MX, (size_t)N * 2 * sizeof(u64), (void**)&mm))) return rc;
    std::vector<u64> init((size_t)N * 2);
    for (int f = 0; f < N; ++f) { init[2 * f] = ~0ull; init[2 * f + 1] = 0ull; }
    HIPC(h, hipMemcpyAsync(drgb, rgb, (size_t)N * npx * 3, hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipMemcpyAsync(mm, init.data(), init.size() * sizeof(u64), hipMemcpyHostToDevice, h->stream));
    const int gx = (int)((npx + 255) / 256);
    hipLaunchKernelGGL(k_cond_minmax, dim3(gx < 512 ? gx : 512, N), dim3(256), 0, h->stream, drgb, npx, mm);
    hipLaunchKernelGGL(k_cond_norm, dim3(gx, N), dim3(256), 0, h->stream, drgb, npx, mm, dgray);
    HIPC(h, hipStreamSynchronize(h->stream));            // `init` leaves scope; the solve may run on other streams (lanes)
    *dgray_out = dgray;
    return TF_OK;
}
}  // namespace

TF_API int tf_condition_frames(tf_handle* h, const uint8_t* rgb, int N, int H, int W, uint8_t* gray_out)
{
    if (!h || !rgb || !gray_out || N < 1 || H < 1 || W < 1) return TF_ERR_INVALID_ARG;
    uint8_t* dgray = nullptr;
    int rc = condition_to_device(h, rgb, N, H, W, &dgray);
    if (rc) return rc;
    HIPC(h, hipMemcpy(gray_out, dgray, (size_t)N * H * W, hipMemcpyDeviceToHost));
    return TF_OK;
}

TF_API int tf_calc_seq_rgb(tf_handle* h, const uint8_t* rgb, int N, int H, int W, float scale, float* flow_out, tf_stats* st)
{
    if (!h || !rgb || !flow_out || H < 1 || W < 1) return TF_ERR_INVALID_ARG;
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    uint8_t* dgray = nullptr;
    int rc = condition_to_device(h, rgb, N, H, W, &dgray);
    if (rc) return rc;
    // frames on the device, flows to the caller's host buffer: sub-batch by sub-batch through the pinned, overlapped copy-out path
    return calc_entry(h, MODE_SEQ, dgray, nullptr, N - 1, H, W, scale, flow_out, W_IN_DEV, st);
}

namespace {
// frames (host, uint8 [N][H][W][channels]) -> fine-grained saliency maps [N][H][W] in the handle's preprocessing buffer: uint8, or
// (f32) float = map * (1/255), what computeSaliency() returns in opencv-contrib 4.x.  Frames go through in chunks so that the work
// buffers (17 B per pixel) stay below ~2.3 GB whatever the study's length; the buffers are the handle's and only ever grow.
int saliency_to_device(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, bool f32, void** dout)
{
    const size_t npx = (size_t)H * W, ipx = (size_t)(H + 1) * (W + 1);
    if (H > 65535 || N > 65535) return fail(h, TF_ERR_UNSUPPORTED, "saliency: at most 65535 rows and 65535 frames per call");
    size_t F = ((size_t)1 << 27) / npx;
    F = F < 1 ? 1 : (F > (size_t)N ? (size_t)N : F);
    HIPC(h, hipSetDevice(h->dev));
    uint8_t *src, *g0, *g1, *ion, *ioff, *out; int* P; float* I; uint16_t *mon, *moff; int* mx;
    int rc;
    if ((rc = pre_grow(h, tf_handle::PRE_SRC, F * npx * channels, (void**)&src)) || (rc = pre_grow(h, tf_handle::PRE_G0, F * npx, (void**)&g0)) ||
        (rc = pre_grow(h, tf_handle::PRE_G1, F * npx, (void**)&g1)) || (rc = pre_grow(h, tf_handle::PRE_ION, F * npx, (void**)&ion)) ||
        (rc = pre_grow(h, tf_handle::PRE_IOFF, F * npx, (void**)&ioff)) || (rc = pre_grow(h, tf_handle::PRE_P, F * npx * sizeof(int), (void**)&P)) ||
        (rc = pre_grow(h, tf_handle::PRE_I, F * ipx * sizeof(float), (void**)&I)) || (rc = pre_grow(h, tf_handle::PRE_MON, F * npx * sizeof(uint16_t), (void**)&mon)) ||
        (rc = pre_grow(h, tf_handle::PRE_MOFF, F * npx * sizeof(uint16_t), (void**)&moff)) || (rc = pre_grow(h, tf_handle::PRE_MX, F * SAL_MX * sizeof(int), (void**)&mx)) ||
        (rc = pre_grow(h, tf_handle::PRE_OUT, (size_t)N * npx * (f32 ? sizeof(float) : 1), (void**)&out))) return rc;
    h->pre_kernel_ms = 0;
    for (size_t f0 = 0; f0 < (size_t)N; f0 += F) {
        const int nf = (int)((size_t)N - f0 < F ? (size_t)N - f0 : F);
        const size_t n = (size_t)nf * npx;
        const dim3 g2((W + 255) / 256, H, nf), blk(256);
        HIPC(h, hipMemcpyAsync(src, frames + f0 * npx * channels, n * channels, hipMemcpyHostToDevice, h->stream));
        HIPC(h, hipEventRecord(h->ev[0], h->stream));       // the eight kernels of the chunk, without the upload
        HIPC(h, hipMemsetAsync(mx, 0, (size_t)nf * SAL_MX * sizeof(int), h->stream));
        hipLaunchKernelGGL(sal::k_sal_gray, dim3((unsigned)((n + 255) / 256)), blk, 0, h->stream, src, channels, n, g0);
        hipLaunchKernelGGL(sal::k_sal_blur3, g2, blk, 0, h->stream, g0, g1, H, W);
        hipLaunchKernelGGL(sal::k_sal_blur3, g2, blk, 0, h->stream, g1, g0, H, W);
        hipLaunchKernelGGL(sal::k_sal_rowprefix, dim3(H, nf), dim3(64), 0, h->stream, g0, H, W, P);
        hipLaunchKernelGGL(sal::k_sal_integral, dim3((W + 1 + 255) / 256, nf), blk, 0, h->stream, P, H, W, I);
        const dim3 g8((W + 255) / 256, (H + SAL_ROWS - 1) / SAL_ROWS, nf);
        hipLaunchKernelGGL(sal::k_sal_scales, g8, blk, 0, h->stream, g0, I, H, W, mon, moff, mx);
        hipLaunchKernelGGL(sal::k_sal_mix_scales, g8, blk, 0, h->stream, mon, moff, H, W, ion, ioff, mx);
        hipLaunchKernelGGL(sal::k_sal_mix_onoff, g2, blk, 0, h->stream, ion, ioff, H, W, mx, f32 ? nullptr : out + f0 * npx,
                           f32 ? (float*)out + f0 * npx : nullptr);
        HIPC(h, hipGetLastError());
        HIPC(h, hipEventRecord(h->ev[1], h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));        // (one chunk holds 2^27 pixels: a study is one chunk; the event pair is read per chunk)
        float t = 0;
        HIPC(h, hipEventElapsedTime(&t, h->ev[0], h->ev[1]));
        h->pre_kernel_ms += t;
    }
    *dout = out;
    return TF_OK;
}

int saliency_frames(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, bool f32, void* out)
{
    if (!h || !frames || !out || N < 1 || H < 1 || W < 1) return TF_ERR_INVALID_ARG;
    if (channels != 1 && channels != 3) return fail(h, TF_ERR_INVALID_ARG, "saliency: frames must have 1 or 3 channels, got %d", channels);
    void* d = nullptr;
    int rc = saliency_to_device(h, frames, N, H, W, channels, f32, &d);
    if (rc) return rc;
    HIPC(h, hipMemcpy(out, d, (size_t)N * H * W * (f32 ? sizeof(float) : 1), hipMemcpyDeviceToHost));
    return TF_OK;
}

int calc_seq_saliency(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, bool f32, float scale, float* flow_out, tf_stats* st)
{
    if (!h || !frames || !flow_out || H < 1 || W < 1) return TF_ERR_INVALID_ARG;
    if (channels != 1 && channels != 3) return fail(h, TF_ERR_INVALID_ARG, "saliency: frames must have 1 or 3 channels, got %d", channels);
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames, got %d", N);
    void* dsal = nullptr;
    int rc = saliency_to_device(h, frames, N, H, W, channels, f32, &dsal);
    if (rc) return rc;
    h->src_f32 = f32 ? 1 : 0;                            // float maps reach the solver as CV_32F frames (DualTVL1: x 255; DeepFlow: as they are)
    rc = calc_entry(h, MODE_SEQ, (const uint8_t*)dsal, nullptr, N - 1, H, W, scale, flow_out, W_IN_DEV, st);
    h->src_f32 = 0;
    return rc;
}
}  // namespace

TF_API int tf_saliency_frames(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, uint8_t* saliency_out)
{
    return saliency_frames(h, frames, N, H, W, channels, false, saliency_out);
}
TF_API int tf_saliency_frames_f32(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, float* saliency_out)
{
    return saliency_frames(h, frames, N, H, W, channels, true, saliency_out);
}
TF_API int tf_calc_seq_saliency(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, float scale, float* flow_out, tf_stats* st)
{
    return calc_seq_saliency(h, frames, N, H, W, channels, false, scale, flow_out, st);
}
TF_API int tf_calc_seq_saliency_f32(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, float scale, float* flow_out, tf_stats* st)
{
    return calc_seq_saliency(h, frames, N, H, W, channels, true, scale, flow_out, st);
}

TF_API int tf_radlong_project(tf_handle* h, const float* flow, const double* centroids, int N, int H, int W,
                              double* rad_out, double* long_out, double* minmax, long long* nonzero)
{
    if (!h || !flow || !centroids || !minmax || !nonzero || N < 1 || H < 1 || W < 1) return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    const size_t npx = (size_t)H * W, tot = (size_t)N * npx;
    if (h->an_rad) { (void)hipFree(h->an_rad); h->an_rad = nullptr; }
    if (h->an_lon) { (void)hipFree(h->an_lon); h->an_lon = nullptr; }
    h->anN = 0;
    HIPC(h, hipMalloc(&h->an_rad, tot * sizeof(double)));
    HIPC(h, hipMalloc(&h->an_lon, tot * sizeof(double)));
    float* dflow = nullptr; double* dcent = nullptr; u64* mm = nullptr; unsigned long long* cnt = nullptr;
    hipError_t e = hipMalloc(&dflow, tot * 2 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&dcent, (size_t)N * 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&mm, 4 * sizeof(u64));
    if (e == hipSuccess) e = hipMalloc(&cnt, (size_t)N * 2 * sizeof(unsigned long long));
    const u64 mm0[4] = {~0ull, 0ull, ~0ull, 0ull};
    u64 mmh[4];
    std::vector<unsigned long long> ch((size_t)N * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(dflow, flow, tot * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dcent, centroids, (size_t)N * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mm, mm0, sizeof mm0, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, (size_t)N * 2 * sizeof(unsigned long long), h->stream);
    if (e == hipSuccess) {
        const int gx = (int)((npx + 255) / 256);
        hipLaunchKernelGGL(k_radlong_project, dim3(gx < 256 ? gx : 256, N), dim3(256), 0, h->stream, dflow, dcent, H, W, h->an_rad, h->an_lon, mm, cnt);
        e = hipMemcpyAsync(mmh, mm, sizeof mmh, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ch.data(), cnt, ch.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && rad_out) e = hipMemcpyAsync(rad_out, h->an_rad, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && long_out) e = hipMemcpyAsync(long_out, h->an_lon, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(dflow); (void)hipFree(dcent); (void)hipFree(mm); (void)hipFree(cnt);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "tf_radlong_project: %s", hipGetErrorString(e));
    for (int j = 0; j < 4; ++j) minmax[j] = f64_unkey(mmh[j]);
    for (size_t i = 0; i < ch.size(); ++i) nonzero[i] = (long long)ch[i];
    h->anN = N; h->anH = H; h->anW = W;
    return TF_OK;
}

TF_API int tf_radlong_hist(tf_handle* h, int which, const double* edges, int nbins, long long* freq_out)
{
    if (!h || !edges || !freq_out || nbins < 1 || which < 0 || which > 1) return TF_ERR_INVALID_ARG;
    if (h->anN < 1) return fail(h, TF_ERR_INVALID_ARG, "tf_radlong_hist needs a preceding tf_radlong_project");
    HIPC(h, hipSetDevice(h->dev));
    const size_t npx = (size_t)h->anH * h->anW;
    double* de = nullptr; unsigned long long* df = nullptr;
    HIPC(h, hipMalloc(&de, (size_t)(nbins + 1) * sizeof(double)));
    hipError_t e = hipMalloc(&df, (size_t)h->anN * nbins * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyAsync(de, edges, (size_t)(nbins + 1) * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(df, 0, (size_t)h->anN * nbins * sizeof(unsigned long long), h->stream);
    if (e == hipSuccess) {
        const int gx = (int)((npx + 255) / 256);
        hipLaunchKernelGGL(k_radlong_hist, dim3(gx < 256 ? gx : 256, h->anN), dim3(256), 0, h->stream, which ? h->an_lon : h->an_rad, npx, de, nbins, df);
        e = hipMemcpyAsync(freq_out, df, (size_t)h->anN * nbins * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(de); (void)hipFree(df);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "tf_radlong_hist: %s", hipGetErrorString(e));
    return TF_OK;
}

TF_API int tf_radlong_select(tf_handle* h, int which, const long long* ranks, double* values_out)
{
    if (!h || !ranks || !values_out || which < 0 || which > 1) return TF_ERR_INVALID_ARG;
    if (h->anN < 1) return fail(h, TF_ERR_INVALID_ARG, "tf_radlong_select needs a preceding tf_radlong_project");
    HIPC(h, hipSetDevice(h->dev));
    const int N = h->anN, NS = N * RL_NSLOT;
    const size_t npx = (size_t)h->anH * h->anW;
    u64* pf = nullptr; long long* rk = nullptr; int* ac = nullptr; unsigned* hist = nullptr;
    std::vector<int> act((size_t)NS);
    for (int i = 0; i < NS; ++i) act[i] = ranks[i] >= 0;
    HIPC(h, hipMalloc(&pf, (size_t)NS * sizeof(u64)));
    hipError_t e = hipMalloc(&rk, (size_t)NS * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&ac, (size_t)NS * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&hist, (size_t)NS * 65536 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(pf, 0, (size_t)NS * sizeof(u64), h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rk, ranks, (size_t)NS * sizeof(long long), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ac, act.data(), (size_t)NS * sizeof(int), hipMemcpyHostToDevice, h->stream);
    const double* v = which ? h->an_lon : h->an_rad;
    const int gx = (int)((npx + 255) / 256);
    for (int shift = 48; shift >= 0 && e == hipSuccess; shift -= 16) {
        e = hipMemsetAsync(hist, 0, (size_t)NS * 65536 * sizeof(unsigned), h->stream);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(k_radlong_sel_hist, dim3(gx < 256 ? gx : 256, N), dim3(256), 0, h->stream, v, npx, shift, pf, ac, hist);
        hipLaunchKernelGGL(k_radlong_sel_scan, dim3(NS), dim3(256), 0, h->stream, hist, shift, pf, rk, ac);
    }
    std::vector<u64> keys((size_t)NS);
    if (e == hipSuccess) e = hipMemcpyAsync(keys.data(), pf, (size_t)NS * sizeof(u64), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(pf); (void)hipFree(rk); (void)hipFree(ac); (void)hipFree(hist);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "tf_radlong_select: %s", hipGetErrorString(e));
    for (int i = 0; i < NS; ++i) values_out[i] = act[i] ? f64_unkey(keys[i]) : 0.0;
    return TF_OK;
}

// pinned host memory for results: a destination allocated here makes the host-pointer entry points copy out at PCIe
// speed, overlapped with the solve of the next sub-batch
TF_API void* tf_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
TF_API void tf_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

// ---- WASE background compensation (rows a7 / f2) ---------------------------------------------------------------
namespace {
int wase_grow(tf_handle* h, size_t na, size_t ncnt, size_t nsum, size_t nbg)
{
    if (h->wa_cap < na) { if (h->wa) (void)hipFree(h->wa); h->wa = nullptr; h->wa_cap = 0; HIPC(h, hipMalloc(&h->wa, na * sizeof(float))); h->wa_cap = na; }
    if (h->wcnt_cap < ncnt) {
        if (h->wcnt) (void)hipFree(h->wcnt);
        if (h->woff) (void)hipFree(h->woff);
        h->wcnt = nullptr; h->woff = nullptr; h->wcnt_cap = 0;
        HIPC(h, hipMalloc(&h->wcnt, ncnt * sizeof(unsigned))); HIPC(h, hipMalloc(&h->woff, (ncnt + 1) * sizeof(u64)));
        h->wcnt_cap = ncnt;
    }
    if (h->wsum_cap < nsum) { if (h->wsum) (void)hipFree(h->wsum); h->wsum = nullptr; h->wsum_cap = 0; HIPC(h, hipMalloc(&h->wsum, nsum * sizeof(float))); h->wsum_cap = nsum; }
    if (h->wbg_cap < nbg) { if (h->wbg) (void)hipFree(h->wbg); h->wbg = nullptr; h->wbg_cap = 0; HIPC(h, hipMalloc(&h->wbg, nbg * sizeof(float))); h->wbg_cap = nbg; }
    return TF_OK;
}

// flows, bkgd: device.  Leaves the backgrounds in h->wbg[0..P) and (if apply) the compensated, scaled flows in place.
int wase_device(tf_handle* h, float* flows, const uint8_t* bkgd, int P, int N, int H, int W, float scale, bool apply)
{
    const size_t hw2 = (size_t)H * W * 2;
    const int C = (int)((hw2 + WASE_CHUNK - 1) / WASE_CHUNK);
    const size_t ncnt = (size_t)N * C, na = (size_t)N * hw2, npieces = (na + NP_BUFSIZE - 1) / NP_BUFSIZE;
    int rc = wase_grow(h, na, ncnt, npieces, (size_t)P);
    if (rc) return rc;
    hipStream_t s = h->stream;
    for (int p = 0; p < P; ++p) {
        const float* f = flows + (size_t)p * hw2;
        hipLaunchKernelGGL(k_wase_count, dim3(C, N), dim3(256), 0, s, f, bkgd, hw2, C, h->wcnt);
        hipLaunchKernelGGL(k_wase_scan, dim3(1), dim3(1024), 0, s, h->wcnt, ncnt, h->woff);
        hipLaunchKernelGGL(k_wase_scatter, dim3(C, N), dim3(256), 0, s, f, bkgd, hw2, C, h->woff, h->wa);
        hipLaunchKernelGGL(k_wase_piece_sums, dim3((unsigned)npieces), dim3(512), 0, s, h->wa, h->woff + ncnt, h->wsum);
        hipLaunchKernelGGL(k_wase_finish, dim3(1), dim3(64), 0, s, h->wsum, h->woff + ncnt, h->wbg + p);
    }
    if (apply) {
        const unsigned gx = (unsigned)std::min<size_t>((hw2 + 255) / 256, 1024);
        hipLaunchKernelGGL(k_wase_apply, dim3(gx, P), dim3(256), 0, s, flows, h->wbg, hw2, scale);
    }
    HIPC(h, hipGetLastError());
    return TF_OK;
}
}  // namespace

TF_API int tf_wase_compensate_device(tf_handle* h, float* flows, int n_flows, const uint8_t* bkgd, int n_frames, int H, int W, float scale,
                                     float* background_out)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (!flows || !bkgd || n_flows < 1 || n_frames < 1 || H < 1 || W < 1) return fail(h, TF_ERR_INVALID_ARG, "tf_wase_compensate: bad argument");
    HIPC(h, hipSetDevice(h->dev));
    int rc = wase_device(h, flows, bkgd, n_flows, n_frames, H, W, scale, true);
    if (rc) return rc;
    if (background_out) HIPC(h, hipMemcpyAsync(background_out, h->wbg, (size_t)n_flows * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return TF_OK;
}

TF_API int tf_wase_compensate(tf_handle* h, float* flows, int n_flows, const uint8_t* bkgd, int n_frames, int H, int W, float scale,
                              float* background_out)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (!flows || !bkgd || n_flows < 1 || n_frames < 1 || H < 1 || W < 1) return fail(h, TF_ERR_INVALID_ARG, "tf_wase_compensate: bad argument");
    HIPC(h, hipSetDevice(h->dev));
    const size_t hw2 = (size_t)H * W * 2;
    float* df = nullptr; uint8_t* dm = nullptr;
    HIPC(h, hipMalloc(&dm, (size_t)n_frames * hw2));
    hipError_t e = hipMemcpyAsync(dm, bkgd, (size_t)n_frames * hw2, hipMemcpyHostToDevice, h->stream);
    const int chunk = 64;                                   // flows per round trip
    if (e == hipSuccess) e = hipMalloc(&df, (size_t)std::min(chunk, n_flows) * hw2 * sizeof(float));
    int rc = TF_OK;
    for (int p0 = 0; p0 < n_flows && e == hipSuccess && rc == TF_OK; p0 += chunk) {
        const int np = std::min(chunk, n_flows - p0);
        e = hipMemcpyAsync(df, flows + (size_t)p0 * hw2, (size_t)np * hw2 * sizeof(float), hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) break;
        rc = wase_device(h, df, dm, np, n_frames, H, W, scale, true);
        if (rc) break;
        e = hipMemcpyAsync(flows + (size_t)p0 * hw2, df, (size_t)np * hw2 * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess && background_out)
            e = hipMemcpyAsync(background_out + p0, h->wbg, (size_t)np * sizeof(float), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(df); (void)hipFree(dm);
    if (rc) return rc;
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "tf_wase_compensate: %s", hipGetErrorString(e));
    return TF_OK;
}

TF_API int tf_get_iters(tf_handle* h, int* out, size_t capacity_ints, size_t* written)
{
    if (!h || !out) return TF_ERR_INVALID_ARG;
    const size_t n = h->last_iters.size() < capacity_ints ? h->last_iters.size() : capacity_ints;
    memcpy(out, h->last_iters.data(), n * sizeof(int));
    if (written) *written = n;
    return TF_OK;
}

// ---- multi-GPU: the ONE exchange step of the path -------------------------------------------
// Frame pairs shard over GPUs with no data-path traffic during the solve; at the end every rank contributes its (u,v)
// fields to a single RCCL all-gather over xGMI.  librccl is loaded on first use (dlopen), so single-GPU users never pay
// for it and the library loads where RCCL is absent.
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
Rccl* rccl()
{
    static Rccl R;
    if (R.lib || !R.err.empty()) return &R;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);        // an RCCL the process already holds (torch's) is reused by SONAME
        if (R.lib) break;
    }
    if (!R.lib) { R.err = std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?"); return &R; }
    bool ok = true;
    auto sym = [&](auto& fn, const char* n) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(R.lib, n)); ok = ok && fn; };
    sym(R.GetUniqueId, "ncclGetUniqueId"); sym(R.CommInitRank, "ncclCommInitRank"); sym(R.CommInitAll, "ncclCommInitAll");
    sym(R.AllGather, "ncclAllGather"); sym(R.GroupStart, "ncclGroupStart"); sym(R.GroupEnd, "ncclGroupEnd");
    sym(R.CommDestroy, "ncclCommDestroy"); sym(R.GetErrorString, "ncclGetErrorString");
    if (!ok) { R.err = "librccl lacks an expected ncclXxx symbol"; dlclose(R.lib); R.lib = nullptr; }
    return &R;
}
#define NCCLC(h, call)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r_ = (call);                                                                               \
        if (r_ != ncclSuccess) return fail(h, TF_ERR_HIP, "%s failed: %s", #call, rccl()->GetErrorString(r_));  \
    } while (0)

int comm_streams(tf_handle* h)
{
    if (!h->comm_stream) {
        HIPC(h, hipSetDevice(h->dev));
        HIPC(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        for (auto& e : h->comm_ev) HIPC(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPC(h, hipEventCreateWithFlags(&h->comm_ready, hipEventDisableTiming));
    }
    return TF_OK;
}
}  // namespace

TF_API int tf_comm_unique_id(unsigned char* id)
{
    if (!id) return TF_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R->lib) return fail(nullptr, TF_ERR_UNSUPPORTED, "%s", R->err.c_str());
    ncclUniqueId u;
    NCCLC(nullptr, R->GetUniqueId(&u));
    static_assert(sizeof u == TF_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return TF_OK;
}

TF_API int tf_comm_init_rank(tf_handle* h, int nranks, int rank, const unsigned char* id)
{
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return h ? fail(h, TF_ERR_INVALID_ARG, "tf_comm_init_rank: bad argument") : TF_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R->lib) return fail(h, TF_ERR_UNSUPPORTED, "%s", R->err.c_str());
    if (h->comm) return fail(h, TF_ERR_INVALID_ARG, "this handle already has a communicator");
    int rc = comm_streams(h);
    if (rc) return rc;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    HIPC(h, hipSetDevice(h->dev));
    NCCLC(h, R->CommInitRank(&h->comm, nranks, u, rank));
    h->comm_rank = rank; h->comm_size = nranks;
    return TF_OK;
}

TF_API int tf_comm_init_all(tf_handle** handles, int n)
{
    if (!handles || n < 1 || n > 64) return TF_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R->lib) return fail(handles[0], TF_ERR_UNSUPPORTED, "%s", R->err.c_str());
    std::vector<int> devs((size_t)n);
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) {
        if (!handles[i] || handles[i]->comm) return fail(handles[0], TF_ERR_INVALID_ARG, "handle %d is null or already has a communicator", i);
        devs[i] = handles[i]->dev;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return fail(handles[0], TF_ERR_INVALID_ARG, "handles %d and %d sit on the same device %d", j, i, devs[i]);
        int rc = comm_streams(handles[i]);
        if (rc) return rc;
    }
    NCCLC(handles[0], R->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) { handles[i]->comm = comms[i]; handles[i]->comm_rank = i; handles[i]->comm_size = n; }
    return TF_OK;
}

namespace {
// enqueue this rank's part of the all-gather behind everything the handle's solve stream holds; *ticket names the event
int enqueue_allgather(tf_handle* h, const float* d_send, size_t count, float* d_recv, int* ticket)
{
    Rccl* R = rccl();
    HIPC(h, hipEventRecord(h->comm_ready, h->stream));
    HIPC(h, hipStreamWaitEvent(h->comm_stream, h->comm_ready, 0));
    NCCLC(h, R->AllGather(d_send, d_recv, count, ncclFloat, h->comm, h->comm_stream));
    const unsigned t = h->comm_tickets++;
    HIPC(h, hipEventRecord(h->comm_ev[t % 8], h->comm_stream));
    if (ticket) *ticket = (int)t;
    return TF_OK;
}
}  // namespace

TF_API int tf_allgather_flows(tf_handle* h, const float* d_send, size_t count_floats, float* d_recv, int* ticket)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->comm) return fail(h, TF_ERR_INVALID_ARG, "tf_allgather_flows needs tf_comm_init_rank / tf_comm_init_all first");
    if (!d_send || !d_recv || count_floats == 0) return fail(h, TF_ERR_INVALID_ARG, "tf_allgather_flows: bad argument");
    HIPC(h, hipSetDevice(h->dev));
    return enqueue_allgather(h, d_send, count_floats, d_recv, ticket);
}

TF_API int tf_allgather_flows_all(tf_handle** handles, int n, const float* const* d_send, size_t count_floats, float* const* d_recv)
{
    if (!handles || !d_send || !d_recv || n < 1 || count_floats == 0) return TF_ERR_INVALID_ARG;
    Rccl* R = rccl();
    if (!R->lib) return fail(handles[0], TF_ERR_UNSUPPORTED, "%s", R->err.c_str());
    for (int i = 0; i < n; ++i)
        if (!handles[i] || !handles[i]->comm || handles[i]->comm_size != n) return fail(handles[0], TF_ERR_INVALID_ARG, "handle %d is not part of an %d-rank tf_comm_init_all group", i, n);
    NCCLC(handles[0], R->GroupStart());                      // one process drives every rank: the calls must be grouped
    int rc = TF_OK;
    for (int i = 0; i < n && rc == TF_OK; ++i) {
        if (hipSetDevice(handles[i]->dev) != hipSuccess) { rc = fail(handles[0], TF_ERR_HIP, "hipSetDevice(%d)", handles[i]->dev); break; }
        rc = enqueue_allgather(handles[i], d_send[i], count_floats, d_recv[i], nullptr);
    }
    const ncclResult_t ge = R->GroupEnd();
    if (rc) return rc;
    if (ge != ncclSuccess) return fail(handles[0], TF_ERR_HIP, "ncclGroupEnd failed: %s", R->GetErrorString(ge));
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(handles[i]->dev) != hipSuccess || hipStreamSynchronize(handles[i]->comm_stream) != hipSuccess)
            return fail(handles[0], TF_ERR_HIP, "all-gather on rank %d did not complete", i);
    }
    return TF_OK;
}

TF_API int tf_comm_wait(tf_handle* h, int ticket)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (!h->comm_stream) return TF_OK;
    HIPC(h, hipSetDevice(h->dev));
    if (ticket < 0 || (unsigned)ticket + 8 <= h->comm_tickets) { HIPC(h, hipStreamSynchronize(h->comm_stream)); return TF_OK; }   // all / too old for the ring
    if ((unsigned)ticket >= h->comm_tickets) return fail(h, TF_ERR_INVALID_ARG, "unknown all-gather ticket %d", ticket);
    HIPC(h, hipEventSynchronize(h->comm_ev[(unsigned)ticket % 8]));
    return TF_OK;
}

TF_API int tf_comm_destroy(tf_handle* h)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    if (h->comm && rccl()->lib) (void)rccl()->CommDestroy(h->comm);
    h->comm = nullptr; h->comm_size = 0;
    if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
    for (auto& e : h->comm_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (h->comm_ready) { (void)hipEventDestroy(h->comm_ready); h->comm_ready = nullptr; }
    return TF_OK;
}

// ---- kernel-level hooks --------------------------------------------------------------------------
TF_API int tf_dbg_resize(tf_handle* h, const float* src, int sw, int sh, float* dst, int dw, int dh,
                         double inv_scale_x, double inv_scale_y, float mul)
{
    if (!h || !src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    const Geom gs = make_geom(sw, sh), gd = make_geom(dw, dh);
    DBuf s, d;
    int rc;
    if ((rc = dbg_up(h, s, src, gs)) || (rc = dbg_up(h, d, nullptr, gd))) return rc;
    hipLaunchKernelGGL(k_pyr_down, grid64x4(gd, 1), dim3(256), 0, h->stream, s.p, gs, d.p, gd, 1.0 / inv_scale_x, 1.0 / inv_scale_y);
    HIPC(h, hipStreamSynchronize(h->stream));
    if ((rc = dbg_down(h, dst, d.p, gd))) return rc;
    if (mul != 1.0f) for (size_t i = 0; i < (size_t)dw * dh; ++i) dst[i] *= mul;
    return TF_OK;
}

TF_API int tf_dbg_pyramid(tf_handle* h, const uint8_t* img, int H, int W, int level, float* out, int* ow, int* oh)
{
    if (!h || !img || !ow || !oh || H < 1 || W < 1 || level < 0 || level >= MAXLEV) return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    Geom g = make_geom(W, H);
    uint8_t* d8 = nullptr;
    HIPC(h, hipMalloc(&d8, (size_t)H * W));
    HIPC(h, hipMemcpyAsync(d8, img, (size_t)H * W, hipMemcpyHostToDevice, h->stream));
    DBuf cur;
    int rc = dbg_up(h, cur, nullptr, g);
    if (rc) { (void)hipFree(d8); return rc; }
    hipLaunchKernelGGL(k_u8_to_f32, dim3((g.w + 255) / 256, g.h, 1), dim3(256), 0, h->stream, d8, cur.p, g);
    for (int s = 1; s <= level; ++s) {
        Geom gn = make_geom(cv_round_d(g.w * h->P.scale_step), cv_round_d(g.h * h->P.scale_step));
        if (gn.w < 1 || gn.h < 1) { (void)hipFree(d8); return fail(h, TF_ERR_INVALID_ARG, "pyramid level %d is empty", s); }
        DBuf nxt;
        if ((rc = dbg_up(h, nxt, nullptr, gn))) { (void)hipFree(d8); return rc; }
        const double sc = 1.0 / h->P.scale_step;
        hipLaunchKernelGGL(k_pyr_down, grid64x4(gn, 1), dim3(256), 0, h->stream, cur.p, g, nxt.p, gn, sc, sc);
        HIPC(h, hipStreamSynchronize(h->stream));
        std::swap(cur.p, nxt.p);
        g = gn;
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    (void)hipFree(d8);
    *ow = g.w; *oh = g.h;
    if (out) return dbg_down(h, out, cur.p, g);
    return TF_OK;
}

TF_API int tf_dbg_warp(tf_handle* h, const float* I0, const float* I1, const float* u1, const float* u2, int w, int hgt,
                       float* I1wx, float* I1wy, float* rho_c)
{
    if (!h || !I0 || !I1 || !u1 || !u2 || !I1wx || !I1wy || !rho_c || w < 1 || hgt < 1) return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    const Geom g = make_geom(w, hgt);
    // frames: [I0, I1] in one allocation so that pair 0 = (frame 0, frame 1)
    float* fr = nullptr;
    HIPC(h, hipMalloc(&fr, 2 * (size_t)g.plane * sizeof(float)));
    DBuf keep; keep.p = fr;
    HIPC(h, hipMemsetAsync(fr, 0, 2 * (size_t)g.plane * sizeof(float), h->stream));
    HIPC(h, hipMemcpy2DAsync(fr, (size_t)g.pitch * 4, I0, (size_t)w * 4, (size_t)w * 4, hgt, hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipMemcpy2DAsync(fr + g.plane, (size_t)g.pitch * 4, I1, (size_t)w * 4, (size_t)w * 4, hgt, hipMemcpyHostToDevice, h->stream));
    DBuf du1, du2, dwx, dwy, drho;
    int rc;
    if ((rc = dbg_up(h, du1, u1, g)) || (rc = dbg_up(h, du2, u2, g)) || (rc = dbg_up(h, dwx, nullptr, g)) ||
        (rc = dbg_up(h, dwy, nullptr, g)) || (rc = dbg_up(h, drho, nullptr, g))) return rc;
    PairCtl* ctl = nullptr;
    HIPC(h, hipMalloc(&ctl, sizeof(PairCtl)));
    HIPC(h, hipMemsetAsync(ctl, 0, sizeof(PairCtl), h->stream));
    WarpArgs wa = {};
    wa.pyr = fr; wa.off0 = 0; wa.off1 = 1; wa.sb.u1[0] = du1.p; wa.sb.u2[0] = du2.p; wa.ctl = ctl; wa.tab = h->tab;
    wa.wx = dwx.p; wa.wy = dwy.p; wa.rho = drho.p; wa.g = g;
    DBuf dgx, dgy;
    if (h->P.variant == TF_VARIANT_CUDA) {
        HIPC(h, hipMalloc(&dgx.p, 2 * (size_t)g.plane * sizeof(float))); HIPC(h, hipMalloc(&dgy.p, 2 * (size_t)g.plane * sizeof(float)));
        hipLaunchKernelGGL(k_grad, grid64x4(g, 2), dim3(256), 0, h->stream, fr, dgx.p, dgy.p, g);
    }
    launch_warp(h, wa, 1, h->stream, dgx.p, dgy.p);
    hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(ctl);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "k_warp: %s", hipGetErrorString(e));
    if ((rc = dbg_down(h, I1wx, dwx.p, g)) || (rc = dbg_down(h, I1wy, dwy.p, g)) || (rc = dbg_down(h, rho_c, drho.p, g))) return rc;
    return TF_OK;
}

TF_API int tf_dbg_df_blur(tf_handle* h, const float* src, int w, int hgt, float* dst)
{
    if (!h || !src || !dst || w < 1 || hgt < 1) return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    const Geom g = make_geom(w, hgt);
    DBuf a, b;
    int rc;
    if ((rc = dbg_up(h, a, src, g)) || (rc = dbg_up(h, b, nullptr, g))) return rc;
    float k0, k1;
    df_gauss3(h->DP.sigma > 0 ? h->DP.sigma : 0.6f, &k0, &k1);
    hipLaunchKernelGGL(k_df_blur, grid64x4(g, 1), dim3(256), 0, h->stream, a.p, b.p, g, k0, k1);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "k_df_blur: %s", hipGetErrorString(e));
    return dbg_down(h, dst, b.p, g);
}

TF_API int tf_dbg_df_refine(tf_handle* h, const float* I0, const float* I1, int w, int hgt, float* u, float* v)
{
    if (!h || !I0 || !I1 || !u || !v || w < 1 || hgt < 1) return TF_ERR_INVALID_ARG;
    if (h->P.algo != TF_ALGO_DEEPFLOW) return fail(h, TF_ERR_INVALID_ARG, "tf_dbg_df_refine needs a handle from tf_create_deepflow");
    HIPC(h, hipSetDevice(h->dev));
    const Geom g = make_geom(w, hgt);
    float* fr = nullptr;
    HIPC(h, hipMalloc(&fr, 2 * (size_t)g.plane * sizeof(float)));
    DBuf keep; keep.p = fr;
    HIPC(h, hipMemsetAsync(fr, 0, 2 * (size_t)g.plane * sizeof(float), h->stream));
    HIPC(h, hipMemcpy2DAsync(fr, (size_t)g.pitch * 4, I0, (size_t)w * 4, (size_t)w * 4, hgt, hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipMemcpy2DAsync(fr + g.plane, (size_t)g.pitch * 4, I1, (size_t)w * 4, (size_t)w * 4, hgt, hipMemcpyHostToDevice, h->stream));
    float* planes = nullptr;
    HIPC(h, hipMalloc(&planes, 23 * (size_t)g.plane * sizeof(float)));
    DBuf keep2; keep2.p = planes;
    HIPC(h, hipMemsetAsync(planes, 0, 23 * (size_t)g.plane * sizeof(float), h->stream));
    const DfBufs saved = h->df;
    {
        float* p = planes;
        DfBufs& d = h->df;
        float** slots[] = {&d.avg, &d.Iz, &d.Ix, &d.Iy, &d.Ixx, &d.Ixy, &d.Iyy, &d.Ixz, &d.Iyz, &d.A11, &d.A12, &d.A22, &d.b1, &d.b2, &d.wg,
                           &d.du, &d.dv, &d.du2, &d.dv2, &d.Wu[0], &d.Wu[1], &d.Wv[0], &d.Wv[1]};
        for (auto s_ : slots) { *s_ = p; p += g.plane; }
    }
    hipError_t e = hipMemcpy2DAsync(h->df.Wu[0], (size_t)g.pitch * 4, u, (size_t)w * 4, (size_t)w * 4, hgt, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpy2DAsync(h->df.Wv[0], (size_t)g.pitch * 4, v, (size_t)w * 4, (size_t)w * 4, hgt, hipMemcpyHostToDevice, h->stream);
    CoopClaim claim(h->dev);
    h->coop_share = claim.ok ? h->num_cus : 0;
    int rc = coop_ensure(h);
    if (rc) { h->df = saved; return rc; }
    if (e == hipSuccess) {
        df_refine_level(h, fr, 0, 1, g, 0, 1, h->stream);
        e = hipStreamSynchronize(h->stream);
        bool aborted = false;
        if (e == hipSuccess) rc = coop_aborted(h, &aborted);
        if (!rc && aborted) rc = fail(h, TF_ERR_HIP, "deepflow refine: the co-resident SOR launch gave up waiting for its neighbours");
    }
    if (e != hipSuccess) rc = fail(h, TF_ERR_HIP, "deepflow refine: %s", hipGetErrorString(e));
    if (!rc) rc = dbg_down(h, u, h->df.avg, g);
    if (!rc) rc = dbg_down(h, v, h->df.Iz, g);
    h->df = saved;
    return rc;
}

TF_API int tf_dbg_median(tf_handle* h, const float* src, int w, int hgt, int ksize, float* dst)
{
    if (!h || !src || !dst || w < 1 || hgt < 1 || (ksize != 3 && ksize != 5)) return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    const Geom g = make_geom(w, hgt);
    DBuf a0, a1, b0, b1;
    int rc;
    if ((rc = dbg_up(h, a0, src, g)) || (rc = dbg_up(h, a1, nullptr, g)) || (rc = dbg_up(h, b0, src, g)) || (rc = dbg_up(h, b1, nullptr, g))) return rc;
    PairCtl* ctl = nullptr;
    HIPC(h, hipMalloc(&ctl, sizeof(PairCtl)));
    HIPC(h, hipMemsetAsync(ctl, 0, sizeof(PairCtl), h->stream));
    MedArgs ma = {};
    ma.sb.u1[0] = a0.p; ma.sb.u1[1] = a1.p; ma.sb.u2[0] = b0.p; ma.sb.u2[1] = b1.p;
    ma.ctl = ctl; ma.err = nullptr; ma.errstride = 0; ma.it = 0; ma.thr_q = 0; ma.utog = 0; ma.g = g;
    const dim3 gm((g.w + 63) / 64, (g.h + 15) / 16, 2);
    if (ksize == 5) hipLaunchKernelGGL(k_median<5>, gm, dim3(256), 0, h->stream, ma);
    else hipLaunchKernelGGL(k_median<3>, gm, dim3(256), 0, h->stream, ma);
    hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(ctl);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "k_median: %s", hipGetErrorString(e));
    return dbg_down(h, dst, a1.p, g);
}

TF_API int tf_dbg_iterate(tf_handle* h, const float* I1wx, const float* I1wy, const float* rho_c,
                          float* u1, float* u2, float* p11, float* p12, float* p21, float* p22,
                          int w, int hgt, int nsteps, int p_is_zero, unsigned long long* err_q)
{
    if (!h || !I1wx || !I1wy || !rho_c || !u1 || !u2 || !p11 || !p12 || !p21 || !p22 || w < 1 || hgt < 1 || nsteps < 0)
        return TF_ERR_INVALID_ARG;
    HIPC(h, hipSetDevice(h->dev));
    const Geom g = make_geom(w, hgt);
    DBuf cx, cy, cr, s[12];
    int rc;
    if ((rc = dbg_up(h, cx, I1wx, g)) || (rc = dbg_up(h, cy, I1wy, g)) || (rc = dbg_up(h, cr, rho_c, g))) return rc;
    float* hostp[6] = {u1, u2, p11, p12, p21, p22};
    for (int k = 0; k < 6; ++k) {
        if ((rc = dbg_up(h, s[2 * k], hostp[k], g)) || (rc = dbg_up(h, s[2 * k + 1], nullptr, g))) return rc;
    }
    PairCtl* ctl = nullptr; u64* errs = nullptr;
    HIPC(h, hipMalloc(&ctl, sizeof(PairCtl)));
    HIPC(h, hipMemsetAsync(ctl, 0, sizeof(PairCtl), h->stream));
    HIPC(h, hipMalloc(&errs, (size_t)(nsteps + 1) * sizeof(u64)));
    HIPC(h, hipMemsetAsync(errs, 0, (size_t)(nsteps + 1) * sizeof(u64), h->stream));
    IterArgs ia = {};
    ia.wx = cx.p; ia.wy = cy.p; ia.rho = cr.p;
    for (int k = 0; k < 2; ++k) {
        ia.sb.u1[k] = s[0 + k].p; ia.sb.u2[k] = s[2 + k].p; ia.sb.p11[k] = s[4 + k].p;
        ia.sb.p12[k] = s[6 + k].p; ia.sb.p21[k] = s[8 + k].p; ia.sb.p22[k] = s[10 + k].p;
    }
    ia.ctl = ctl; ia.err = errs; ia.errstride = nsteps + 1; ia.thr_q = -1.0; ia.g = g; ia.host_slot = nullptr; ia.B = 1;
    ia.l_t = (float)(h->P.lambda * h->P.theta); ia.theta = (float)h->P.theta; ia.taut = (float)(h->P.tau / h->P.theta);
    const bool two = h->iter_variant == 2 && (rows_ok(h, g, 1) || h->tile2) && nsteps % 2 == 0;
    int launches = 0;
    if (two) {
        for (int it = 0; it < nsteps; it += 2, ++launches) {
            Iter2Args A2;
            A2.a = ia; A2.a.it = it; A2.a.utog = launches; A2.a.ptog = launches; A2.a.pzero = (p_is_zero && it == 0) ? 1 : 0;
            A2.utog_prev = A2.ptog_prev = A2.pzero_prev = 0; A2.total = nsteps;
            launch_iter2(h, A2, 1, h->stream);
        }
    } else {
        for (int it = 0; it < nsteps; ++it, ++launches) {
            ia.it = it; ia.utog = it; ia.ptog = it; ia.pzero = (p_is_zero && it == 0) ? 1 : 0;
            launch_iter(h, ia, 1, h->stream);
        }
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && err_q && nsteps > 0) {
        e = hipMemcpyAsync(err_q, errs, (size_t)nsteps * sizeof(u64), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(ctl); (void)hipFree(errs);
    if (e != hipSuccess) return fail(h, TF_ERR_HIP, "k_iter: %s", hipGetErrorString(e));
    const int cur = launches & 1;
    for (int k = 0; k < 6; ++k)
        if ((rc = dbg_down(h, hostp[k], s[2 * k + cur].p, g))) return rc;
    return TF_OK;
}
