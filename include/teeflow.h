/*
 * teeflow.h -- C ABI of libteeflow_hip.so, the MI355X (gfx950) dense optical-flow engine.
 *
 * Drop-in boundary.  The reference is pure Python; on its hot path it holds a
 * cv2.DenseOpticalFlow object and calls three things on it.  Each entry point below names the
 * reference interface it replaces (paths relative to /root/reference):
 *
 *   cv2.optflow.createOptFlow_DualTVL1()            optical_flow/calculate_optical_flow.py:577   -> tf_create
 *   cv2.cuda.OpticalFlowDual_TVL1.create()          optical_flow/calculate_optical_flow.py:575   -> tf_create
 *   OF_model.setLambda(config.lambda_value)         optical_flow/calculate_optical_flow.py:578   -> tf_set_param
 *   OF_model.calc(I0, I1, None)                     optical_flow/calculate_optical_flow.py:642   -> tf_calc_pair
 *   GpuMat.upload x2 + calc + download              optical_flow/calculate_optical_flow.py:634-639 -> tf_calc_pair
 *   the per-frame-pair loop of process_video()      optical_flow/calculate_optical_flow.py:584-600 -> tf_calc_seq
 *   (batch of independent pairs, BASELINE config 3)                                              -> tf_calc_pairs
 *
 * Plain pointers and sizes only; no torch / numpy types.  The library owns all device memory and its
 * HIP streams.  A handle is NOT re-entrant (mirrors the cv2 object, which the reference also reuses
 * sequentially from one thread), but handles are independent of each other: several handles on ONE
 * device may be driven from several host threads at once.
 *
 * Sub-batches and lanes.  A handle solves up to max_batch (128) pairs at a time.  A call that holds more
 * -- tf_calc_pairs with 1024 pairs, tf_calc_seq on a 1025-frame study -- is cut into equal sub-batches (as few as fit, a multiple of
 * the lane count of them: 1024 pairs = 9 x 114) that the
 * handle's LANES (engines of their own inside the library: stream, buffers, host thread; three for
 * DualTVL1, one for DeepFlow) take from a queue one at a time: a lane that has finished a sub-batch starts
 * the next at once, so one sub-batch's tail (few pairs still iterating) runs under the others' full
 * launches.  The call returns when all of its sub-batches are done (if one fails, those not yet started
 * are dropped, the ones running finish, and the call returns the failure with nothing left in flight).
 * tf_submit_* queue the same job without waiting (tf_wait collects it), so that consecutive batches --
 * the studies of a folder, the steps of a stream -- keep the lanes busy across calls.  Flows and
 * tf_get_iters are identical whichever lane solved what.
 * All functions return TF_OK (0) or an error
 * code; tf_last_error() gives the message (the Python layer raises OpticalFlowCalculationError,
 * reference optical_flow/exceptions.py:26-28).
 *
 * Flow convention (same as cv2): flow[y][x][0] = x displacement, flow[y][x][1] = y displacement, in
 * pixels per frame, such that I1(x + u, y + v) ~= I0(x, y).
 */
#ifndef TEEFLOW_H
#define TEEFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_ABI_VERSION 2   /* 2: tf_stats grew the per-stage times ms_warp .. ms_sched.  Round 5 ADDED entry points (tf_submit_*, tf_wait,
                              tf_saliency_frames_f32, tf_calc_seq_saliency_f32) and changed no struct and no signature: still 2 */

enum {
    TF_OK = 0,
    TF_ERR_INVALID_ARG = 1,   /* null pointer, non-positive size, ... (cv2 would raise cv2.error) */
    TF_ERR_UNSUPPORTED = 2,   /* parameter combination the engine does not implement */
    TF_ERR_HIP = 3,           /* a HIP runtime call failed; message in tf_last_error */
    TF_ERR_NOMEM = 4,
    TF_ERR_NO_DEVICE = 5      /* no gfx950 device visible: there is NO CPU fallback */
};

enum { TF_ALGO_TVL1 = 0, TF_ALGO_DEEPFLOW = 1 /* handles come from tf_create_deepflow */ };

/* Which OpenCV DualTVL1 the handle reproduces.  TF_VARIANT_CPU (default, the parity target): cv2.optflow's
 * DualTVL1OpticalFlow, what calculate_optical_flow.py:577-578, 642 runs without CUDA.  TF_VARIANT_CUDA (SURVEY.md row
 * a5): the semantics of cv2.cuda.OpticalFlowDual_TVL1 (calculate_optical_flow.py:572-575, 633-639), what the reference
 * runs for OF_algo='TVL1' on a CUDA box -- one loop of inner*outer iterations per warp, no median filtering, the error sum
 * looked at on odd iterations only, weight-normalised Catmull-Rom warp with clamp addressing (details and what is not
 * modelled: oracle/tvl1_oracle.c, variant 1). */
enum { TF_VARIANT_CPU = 0, TF_VARIANT_CUDA = 1 };

/* keys for tf_set_param / tf_get_param: the 12 cv2.DualTVL1OpticalFlow setters */
enum {
    TF_PARAM_TAU = 0, TF_PARAM_LAMBDA = 1, TF_PARAM_THETA = 2, TF_PARAM_NSCALES = 3,
    TF_PARAM_WARPS = 4, TF_PARAM_EPSILON = 5, TF_PARAM_INNER_ITERATIONS = 6,
    TF_PARAM_OUTER_ITERATIONS = 7, TF_PARAM_SCALE_STEP = 8, TF_PARAM_GAMMA = 9,
    TF_PARAM_MEDIAN_FILTERING = 10, TF_PARAM_USE_INITIAL_FLOW = 11,
    TF_PARAM__COUNT = 12
};

typedef struct tf_params {
    double tau, lambda, theta, epsilon, scale_step, gamma;
    int nscales, warps, inner_iterations, outer_iterations, median_filtering, use_initial_flow;
    int algo;        /* TF_ALGO_* */
    int max_batch;   /* pairs resident per sub-batch (0 = default 128) */
    int variant;     /* TF_VARIANT_* (ABI 2) */
} tf_params;

/* Filled by every tf_calc_* call (may be NULL). Times are milliseconds (HIP events on the handle's stream). */
typedef struct tf_stats {
    int n_pairs;               /* pairs solved by the call */
    int nscales_used;          /* pyramid levels actually used (<= nscales) */
    int warps;
    int reserved0;
    double ms_total;           /* whole call, host clock */
    double ms_h2d, ms_device, ms_d2h;
    /* dominant kernel (tvl1_iter): launches, summed duration (only when profiling is on) and bytes */
    unsigned long long iter_launches;
    unsigned long long iter_pair_steps;    /* DualTVL1: sum over launches of pairs that actually iterated; DeepFlow: pixels x pairs summed over
                                              the SOR launches (a launch's compulsory traffic is 40 B per pixel) */
    double iter_ms;                        /* sum of tvl1_iter launch durations (tf_set_profile(h,1)) */
    double iter_bytes;                     /* algorithmic bytes of all executed pair-iterations (60 B/px) */
    double total_bytes;                    /* algorithmic bytes of the whole solve */
    unsigned long long inner_iters_total;  /* sum over pairs/levels/warps of executed inner iterations */
    unsigned long long outer_iters_total;  /* ... of executed outer iterations (= median passes) */
    /* summed launch durations per stage of the solve, filled like iter_ms only under tf_set_profile(h,1), single lane:
     * ms_warp (k_warp_lds) and ms_median (k_median2).  ms_misc and ms_sched are always 0 (they belonged to the free-running
     * scheduler driver removed in round 3; the fields stay so that ABI 2's struct layout does not change). */
    double ms_warp, ms_median, ms_misc, ms_sched;
} tf_stats;

typedef struct tf_handle tf_handle;

/* cv2 defaults: tau .25 lambda .15 theta .3 nscales 5 warps 5 eps .01 inner 30 outer 10 step .8 gamma 0 median 5 */
int tf_default_params(tf_params* p);

/* replaces createOptFlow_DualTVL1() / cuda.OpticalFlowDual_TVL1.create()  (calculate_optical_flow.py:575,577) */
int tf_create(const tf_params* p, int device_id, tf_handle** out);
void tf_destroy(tf_handle* h);

/* cv2.optflow.createOptFlow_DeepFlow() (calculate_optical_flow.py:568) and its OpticalFlowDeepFlow constants; the cv2
 * object exposes no setters in Python, so these are creation-time only.  The handle is used with the same tf_calc_*
 * entry points (OF_model.calc(...), calculate_optical_flow.py:631). */
typedef struct tf_deepflow_params {
    float sigma;                  /* 0.6  Gaussian pre-blur */
    int min_size;                 /* 25   smallest pyramid side */
    float downscale_factor;       /* 0.95 */
    int fixed_point_iterations;   /* 5 */
    int sor_iterations;           /* 25 */
    float alpha, delta, gamma;    /* 1.0, 0.5, 5.0 (VariationalRefinement gets 4*alpha, delta/3, gamma/3) */
    float omega;                  /* 1.6 */
    float zeta, epsilon;          /* 0.1, 0.001 (cv::VariationalRefinement internals) */
    int max_batch;                /* pairs resident per sub-batch (0 = default 128) */
} tf_deepflow_params;
int tf_default_deepflow_params(tf_deepflow_params* p);
int tf_create_deepflow(const tf_deepflow_params* p, int device_id, tf_handle** out);

/* replaces OF_model.setLambda(...) and the 11 sibling setters / getters  (calculate_optical_flow.py:578) */
int tf_set_param(tf_handle* h, int key, double value);
int tf_get_param(tf_handle* h, int key, double* value);

/* use_external != 0: run on the caller's hipStream_t `hip_stream` (NULL = the legacy default stream, which is what
 * torch.cuda.current_stream().cuda_stream reports for torch's default stream); use_external == 0: back to the
 * handle's own non-blocking stream.  Every tf_calc_* still returns only after its work has drained. */
int tf_set_stream(tf_handle* h, void* hip_stream, int use_external);
/* 0 = off; 1 = bracket every tvl1_iter launch with HIP events so tf_stats.iter_ms is filled */
int tf_set_profile(tf_handle* h, int level);

/* replaces OF_model.calc(I0, I1, None)  (calculate_optical_flow.py:631,638,642).
 * I0, I1: host uint8 [H][W] C-contiguous.  flow_out: host float32 [H][W][2]. */
int tf_calc_pair(tf_handle* h, const uint8_t* I0, const uint8_t* I1, int H, int W, float* flow_out, tf_stats* st);

/* replaces the sliding-window loop of process_video() (calculate_optical_flow.py:584-600):
 * frames: host uint8 [N][H][W]; flow_out: host float32 [N-1][H][W][2] = flow(frame i -> frame i+1) * scale.
 * (The caller duplicates the last field, reference :599.) */
int tf_calc_seq(tf_handle* h, const uint8_t* frames, int N, int H, int W, float scale, float* flow_out, tf_stats* st);

/* B independent pairs (BASELINE.json config 3): I0s, I1s host uint8 [B][H][W]; flow_out float32 [B][H][W][2]. */
int tf_calc_pairs(tf_handle* h, const uint8_t* I0s, const uint8_t* I1s, int B, int H, int W, float* flow_out, tf_stats* st);

/* CV_32FC1 frames (host pointers; otherwise identical to tf_calc_pair / tf_calc_pairs).  cv2's DualTVL1 accepts float32 images with
 * values in [0,1] and multiplies them by 255 when it builds level 0; cv2's DeepFlow converts with `convertTo(CV_32F)` and no factor,
 * i.e. takes float frames AS THEY ARE -- a DeepFlow handle does the same (frames in [0,1] stay in [0,1]; zeta and epsilon are not
 * rescaled).  The reference hands over uint8 with no_saliency=True (calculate_optical_flow.py:588) and computeSaliency()'s CV_32F map
 * with no_saliency=False (:586, :631). */
int tf_calc_pair_f32(tf_handle* h, const float* I0, const float* I1, int H, int W, float* flow_out, tf_stats* st);
int tf_calc_pairs_f32(tf_handle* h, const float* I0s, const float* I1s, int B, int H, int W, float* flow_out, tf_stats* st);

/* Same, with every buffer already resident in this device's HBM (pointers are device pointers); work is
 * enqueued on the handle's stream and the call returns after the stream has drained. */
int tf_calc_pairs_device(tf_handle* h, const uint8_t* dI0s, const uint8_t* dI1s, int B, int H, int W,
                         float scale, float* dflow_out, tf_stats* st);
int tf_calc_seq_device(tf_handle* h, const uint8_t* dframes, int N, int H, int W, float scale,
                       float* dflow_out, tf_stats* st);

/* Asynchronous forms of the four calls above (the loop calculate_optical_flow.py:584-597 for SEVERAL studies / batches at a time):
 * the job is queued on the handle's lanes and the call returns with a ticket; every buffer must stay valid and untouched until
 * tf_wait(h, ticket, st) has returned (ticket < 0: all jobs not yet waited for, oldest first; the first failure is returned).
 * Jobs start in submission order; sub-batches of consecutive jobs overlap.  tf_wait leaves the job's iteration counts where
 * tf_get_iters reads them.  While jobs are in flight the synchronous tf_calc_* calls on the same handle queue behind them.
 * uint8 frames only. */
int tf_submit_pairs_device(tf_handle* h, const uint8_t* dI0s, const uint8_t* dI1s, int B, int H, int W, float scale, float* dflow_out, int* ticket);
int tf_submit_seq_device(tf_handle* h, const uint8_t* dframes, int N, int H, int W, float scale, float* dflow_out, int* ticket);
int tf_submit_pairs(tf_handle* h, const uint8_t* I0s, const uint8_t* I1s, int B, int H, int W, float* flow_out, int* ticket);
int tf_submit_seq(tf_handle* h, const uint8_t* frames, int N, int H, int W, float scale, float* flow_out, int* ticket);
/* tf_calc_seq_rgb (below) without waiting: the frames are conditioned at once, into device memory the job owns; `rgb` may be reused as
 * soon as the call returns, `flow_out` belongs to the job until tf_wait.  What process_folder uses to keep the next study's solve on
 * the GPU while the previous one finishes. */
int tf_submit_seq_rgb(tf_handle* h, const uint8_t* rgb, int N, int H, int W, float scale, float* flow_out, int* ticket);
int tf_wait(tf_handle* h, int ticket, tf_stats* st);

/* Frame conditioning of the reference's loop, `img2uint8(rgb2gray(nparr[i]))` (calculate_optical_flow.py:588,
 * optical_flow_utils.py:30-31), on the device: rgb uint8 [N][H][W][3] -> gray uint8 [N][H][W], normalised per frame.
 * tf_calc_seq_rgb = condition + tf_calc_seq without the frames ever returning to the host (flow_out: [N-1][H][W][2]). */
int tf_condition_frames(tf_handle* h, const uint8_t* rgb, int N, int H, int W, uint8_t* gray_out);
int tf_calc_seq_rgb(tf_handle* h, const uint8_t* rgb, int N, int H, int W, float scale, float* flow_out, tf_stats* st);

/* The reference's other preprocessing branch, no_saliency=False (calculate_optical_flow.py:559-560
 * `cv2.saliency.StaticSaliencyFineGrained_create()`, :586 `saliency_obj.computeSaliency(nparr[i])`): the saliency map of every frame
 * takes the place of the conditioned gray frame as the solver's input.  frames: host uint8 [N][H][W][channels], channels 3 (handed to
 * OpenCV's BGR2GRAY in the order given, as the reference does with its RGB frames) or 1.
 *   tf_saliency_frames      -> uint8 [N][H][W]: the algorithm's 8-bit map (what opencv-contrib 3.x returned)
 *   tf_saliency_frames_f32  -> float32 [N][H][W] = that map * (1/255), values in [0,1]: what computeSaliency() returns in
 *                              opencv-contrib 4.x (`dst.convertTo(saliencyMap, CV_32F, 1.0f/255.0f)`), i.e. what the reference's
 *                              OF_model.calc receives under its `opencv-contrib-python>=4.5.0` (requirements.txt:7)
 * tf_calc_seq_saliency / _f32 = maps + tf_calc_seq without the maps returning to the host (flow_out: [N-1][H][W][2]); the _f32 form hands
 * the solver CV_32F frames (DualTVL1 multiplies them by 255 in float, DeepFlow takes them as they are -- see tf_calc_pair_f32) and is
 * the Python layer's default.  Restated from opencv-contrib's saliency module; nothing pins it (oracle/saliency_oracle.c). */
int tf_saliency_frames(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, uint8_t* saliency_out);
int tf_saliency_frames_f32(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, float* saliency_out);
int tf_calc_seq_saliency(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, float scale, float* flow_out,
                         tf_stats* st);
int tf_calc_seq_saliency_f32(tf_handle* h, const uint8_t* frames, int N, int H, int W, int channels, float scale, float* flow_out,
                             tf_stats* st);

/* ---- radial / longitudinal projection + per-frame statistics of the reference's analysis step
 *      (optical_flow/analysis.py:89-212: calculate_comp_magnitude, calc_bidirectional_hist), float64, on the device.
 * tf_radlong_project: flow host float32 [N][H][W][2], centroids host float64 [N][2] = (row, col).  rad_out / long_out
 *   (host float64 [N][H][W]) may be NULL.  minmax[4] = rad min, rad max, long min, long max over the whole arrays
 *   (zeros included, as np.min/np.max); nonzero[2N] = per frame count of non-zero rad / long values.  The projections
 *   stay resident for the two calls below (which: 0 = radial, 1 = longitudinal).
 * tf_radlong_hist: np.histogram(frame[frame != 0], bins=nbins, range=(edges[0], edges[nbins])) per frame, RAW counts.
 * tf_radlong_select: exact order statistics: values[n][j] = sorted(frame n's non-zero values)[ranks[n][j]] for up to 4
 *   ranks per frame (rank < 0 = skip) -- what np.percentile interpolates between. */
int tf_radlong_project(tf_handle* h, const float* flow, const double* centroids, int N, int H, int W,
                       double* rad_out, double* long_out, double* minmax, long long* nonzero);
int tf_radlong_hist(tf_handle* h, int which, const double* edges, int nbins, long long* freq_out);
int tf_radlong_select(tf_handle* h, int which, const long long* ranks, double* values_out);

/* ---- multi-GPU: the single exchange step of the path.  The reference's loop is sequential
 *      (calculate_optical_flow.py:584-597); here pairs shard over the GPUs of a node with no data-path traffic during the
 *      solve, and ONE RCCL all-gather of the (u,v) fields over xGMI assembles the result on every rank.  librccl is loaded
 *      on first use.  Two ways to form the communicator:
 *        one process per GPU : rank 0 calls tf_comm_unique_id, the launcher hands the 128 bytes to every rank (any
 *                              channel: a file, MPI, torch.distributed's store), each rank calls tf_comm_init_rank;
 *        one process, n GPUs : tf_comm_init_all on n handles created on n different devices (ncclCommInitAll).
 *      tf_allgather_flows enqueues this rank's contribution (count floats, device pointers; recv holds nranks*count)
 *      on the handle's communication stream behind the work already queued on its solve stream and returns at once with
 *      a ticket; tf_comm_wait(h, ticket) blocks the host until that all-gather is done (ticket < 0: all of them), which
 *      is what must happen before d_send / d_recv are reused -- so the exchange of step k overlaps the solve of step k+1.
 *      tf_allgather_flows_all is the single-process form: grouped calls for all n ranks, returns when all are done.
 *      ORDER against the solve that produced d_send, stated: tf_allgather_flows orders the collective behind the work queued on
 *      THIS handle's solve stream and nothing else.  Flows written by the handle's lanes (a call larger than one sub-batch, a
 *      tf_submit_* job) or by ANOTHER handle (a communicator handle that never solves, as round 4's bench.py used) are ordered
 *      by the HOST, by design: every tf_calc_* is host-synchronous and so is tf_wait -- when they have returned, every stream
 *      that worked for them has drained, and the all-gather may be issued at once.  There is no earlier point to hook a stream
 *      dependency on: a solve's launches are issued as its stop reports come back (the iteration count is data-dependent), so
 *      until the call returns the producing streams do not yet hold all of its work, and an event recorded on them would order
 *      the collective behind a prefix of the solve only.  Issue the all-gather after the producing call / tf_wait has returned. */
#define TF_COMM_ID_BYTES 128
int tf_comm_unique_id(unsigned char* id /* [TF_COMM_ID_BYTES] */);
int tf_comm_init_rank(tf_handle* h, int nranks, int rank, const unsigned char* id);
int tf_comm_init_all(tf_handle** handles, int n);
int tf_allgather_flows(tf_handle* h, const float* d_send, size_t count_floats, float* d_recv, int* ticket);
int tf_allgather_flows_all(tf_handle** handles, int n, const float* const* d_send, size_t count_floats, float* const* d_recv);
int tf_comm_wait(tf_handle* h, int ticket);
int tf_comm_destroy(tf_handle* h);

/* Executed iteration counts of the last call: int32 [n_pairs][nscales_used][warps][2] = (inner, outer).
 * Returns the number of ints written (<= capacity) through *written. */
int tf_get_iters(tf_handle* h, int* out, size_t capacity_ints, size_t* written);

/* message of the last failure on this handle (or of the last failed tf_create when h == NULL) */
const char* tf_last_error(tf_handle* h);
int tf_abi_version(void);
int tf_device_count(void);

/* ---- kernel-level test hooks (dense host arrays, one image; used by tests/ to compare each kernel
 *      with the oracle bit for bit; not part of the drop-in surface) -------------------------------- */
/* implementation knobs for experiments (results never change).  DualTVL1: "iter_variant" (0 = 64x16 tiles, 1 = full-width row strips,
 * 2 = row strips with two iterations per launch [default]), "min_rows_work" (rows*pairs below which tiles are used), "strip_blocks" (target
 * blocks per tvl1_iter launch), "lag" (launches the host may run ahead of the device's stop reports), "warp_margin" (pixels of flow the
 * LDS-staged warp covers around its tile: 0 = global gathers only, default 8).  DeepFlow: "sor_rt" (0 = one colour per launch, the plain form),
 * "sor_fuse" (sweeps per launch of the tiled register kernel), "sor_rt_shape" (region shape; 3 = chosen per launch), "sor_coop" (1 = all sweeps
 * of a fixed-point iteration in one launch of co-resident regions where a level needs several [default], 2 = always 128x64 regions, 3 = always
 * 128x32 regions for small batches, 0 = never), "sor_coop_small" (0 = small batches keep the tiled form), "sor_coop_s" (sweeps between two exchanges), "sor_coop_min_util" (per cent of its CUs such a launch must fill, else tiled), "df_fuse_ds" (form of the data/smoothness kernel).
 * Both: "lanes" (contiguous parts a call of at most one sub-batch is split into, joined at its end), "queue_lanes" (lanes that take whole
 * sub-batches of larger calls and of tf_submit_* jobs from the queue: -1 = 3 for DualTVL1, 1 for DeepFlow [default]; 0 = no queue, every call is
 * split in contiguous parts as in rounds 1-4), "queue_unit" (pairs per queued sub-batch; 0 = equal sub-batches of at most max_batch pairs, a multiple of the lane count of them). */
int tf_set_tuning(tf_handle* h, const char* name, int value);
/* counters of the handle for tests and tools: "coop_launches" (launches of the co-resident SOR form since the handle was made),
 * "coop_aborts" (calls repeated with the tiled form because such a launch gave up waiting), "coop_disabled"; "queue_jobs", "queue_units_done",
 * "queue_units_failed", "queue_units_skipped" (sub-batches dropped because an earlier one of their call had failed), "queue_outstanding", "queue_lanes";
 * "experimental" (1: built with the experimental tvl1_iter forms); -1 for an unknown name */
long long tf_dbg_counter(tf_handle* h, const char* name);
/* DeepFlow hooks: one cv::VariationalRefinement::calcUV on dense float images (u, v updated in place); 3x3 Gaussian blur */
int tf_dbg_df_refine(tf_handle* h, const float* I0, const float* I1, int w, int hgt, float* u, float* v);
int tf_dbg_df_blur(tf_handle* h, const float* src, int w, int hgt, float* dst);
/* Pinned host memory for flow results.  The reference gets a fresh numpy array from cv2 (`flow = OF_model.calc(...)`,
 * calculate_optical_flow.py:631,642); handing tf_calc_pair/_seq/_pairs a destination from tf_host_alloc (or any pinned
 * host pointer) lets the library copy results out at PCIe speed while the next sub-batch is being solved.  Pageable
 * destinations work too, in order.  tf_host_alloc returns NULL on failure. */
void* tf_host_alloc(size_t bytes);
void  tf_host_free(void* p);
/* "WASE" background compensation (rows a7/f2), replaces the numpy expression of
 * /root/reference/optical_flow/calculate_optical_flow.py:647-652, 659 for ALL flows of a study in one call:
 *   background[p] = np.mean(masked[masked != 0]),  masked = flows[p] * bkgd      (bkgd: bool [n_frames][H][W][2], 0/1 bytes)
 *   flows[p] = (flows[p] - background[p]) * scale                                  (in place)
 * The float32 reduction follows numpy's summation order exactly (8192-element pieces, pairwise sums, float64 divide).
 * tf_wase_compensate takes host pointers, the _device form device pointers; background_out (host, n_flows) may be NULL. */
int tf_wase_compensate(tf_handle* h, float* flows, int n_flows, const uint8_t* bkgd, int n_frames, int H, int W, float scale,
                       float* background_out);
int tf_wase_compensate_device(tf_handle* h, float* flows, int n_flows, const uint8_t* bkgd, int n_frames, int H, int W, float scale,
                              float* background_out);
/* per-launch record of the last solve run with tf_set_profile(h, 1): tvl1_iter launches in issue order (single lane);
 * returns the number of records, fills at most max_n: (level, warp, first iteration) of each launch and its duration. */
int tf_dbg_launch_profile(tf_handle* h, int* level, int* warp, int* it, float* ms, int max_n);
/* strip sizing rule of the tvl1_iter kernel (host arithmetic only): rows per strip R (multiple of RY) and strip count S
 * for n_active pairs still iterating on a device with `slots` resident blocks */
void tf_dbg_strip_rule(int n_active, int H, int RY, int slots, int* R, int* S);
int tf_dbg_pyramid(tf_handle* h, const uint8_t* img, int H, int W, int level, float* out, int* ow, int* oh);
int tf_dbg_resize(tf_handle* h, const float* src, int sw, int sh, float* dst, int dw, int dh,
                  double inv_scale_x, double inv_scale_y, float mul);
int tf_dbg_warp(tf_handle* h, const float* I0, const float* I1, const float* u1, const float* u2, int w, int hgt,
                float* I1wx, float* I1wy, float* rho_c);
int tf_dbg_median(tf_handle* h, const float* src, int w, int hgt, int ksize, float* dst);
int tf_dbg_iterate(tf_handle* h, const float* I1wx, const float* I1wy, const float* rho_c,
                   float* u1, float* u2, float* p11, float* p12, float* p21, float* p22,
                   int w, int hgt, int nsteps, int p_is_zero, unsigned long long* err_q);

#ifdef __cplusplus
}
#endif
#endif /* TEEFLOW_H */
