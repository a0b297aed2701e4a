/*
 * oracle/teeflow_cpu_abi.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * "The same ABI is exported by the CPU oracle library."  This file puts the host-pointer core of
 * include/teeflow.h (tf_create / tf_set_param / tf_calc_pair / tf_calc_seq / tf_calc_pairs / tf_submit_pairs / tf_submit_seq / tf_wait /
 * tf_get_iters / tf_last_error / tf_destroy, and tf_create_deepflow) on top of the CPU restatements in tvl1_oracle.c /
 * deepflow_oracle.c, so that boundary-level tests can drive the checker through the very entry points the product
 * exports (same structs, same error codes, same flow / iteration-count layouts) and compare the two libraries call for
 * call.  It replaces, as a checker, the same reference interface the product does:
 *   cv2.optflow.createOptFlow_DualTVL1() / setLambda / calc   /root/reference/optical_flow/calculate_optical_flow.py:577-578, 642
 *   cv2.cuda.OpticalFlowDual_TVL1.create()                     :575   (tf_params.variant = TF_VARIANT_CUDA)
 *   cv2.optflow.createOptFlow_DeepFlow() / calc                :568, 631
 *   the per-pair loop                                           :584-600
 *
 * The product (tee_optical_flow_amd) never loads this library and has no CPU fallback; only tests/ may.  Entry points of
 * teeflow.h that exist for the GPU engine alone (device pointers, streams, tuning, RCCL, tf_dbg_*) are not exported here.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/teeflow.h"

#define API __attribute__((visibility("default")))

/* the oracles' own parameter blocks and entry points (tvl1_oracle.c, deepflow_oracle.c; linked into this library) */
typedef struct {
    double tau, lambda, theta, epsilon, scale_step, gamma;
    int nscales, warps, inner_iterations, outer_iterations, median_filtering, use_initial_flow;
    int err_mode, variant;
} orc_params;
typedef struct {
    float sigma; int min_size; float downscale_factor; int fixed_point_iterations; int sor_iterations;
    float alpha, delta, gamma, omega, zeta, epsilon;
} dfo_params;
int orc_tvl1_calc(const orc_params* P, const uint8_t* I0, const uint8_t* I1, int H, int W, float* flow, int* iters);
int dfo_deepflow_calc(const dfo_params* P, const uint8_t* I0, const uint8_t* I1, int H, int W, float* flow);

#define MAX_TICKETS 16
struct tf_handle {
    tf_params P;
    tf_deepflow_params DP;
    char err[256];
    int* iters; size_t n_iters;
    int last_pairs, last_nlev;
    /* tf_submit_* jobs: solved at once (one thread of control here), kept until tf_wait */
    struct { int used; tf_stats st; int* iters; size_t n_iters; int pairs, nlev; } job[MAX_TICKETS];
    int next_ticket;
};
static char g_create_err[256];

static int fail(tf_handle* h, int code, const char* msg)
{
    snprintf(h ? h->err : g_create_err, 256, "%s", msg);
    return code;
}

static int validate(tf_handle* h, const tf_params* p)
{
    if (p->algo != TF_ALGO_TVL1 && p->algo != TF_ALGO_DEEPFLOW) return fail(h, TF_ERR_UNSUPPORTED, "unknown algo");
    if (p->nscales < 1 || p->nscales > 64) return fail(h, TF_ERR_INVALID_ARG, "nscales out of range");
    if (p->warps < 1) return fail(h, TF_ERR_INVALID_ARG, "warps must be >= 1");
    if (p->inner_iterations < 1 || p->outer_iterations < 1) return fail(h, TF_ERR_INVALID_ARG, "iterations must be >= 1");
    if (p->median_filtering != 1 && p->median_filtering != 3 && p->median_filtering != 5) return fail(h, TF_ERR_UNSUPPORTED, "medianFiltering must be 1, 3 or 5");
    if (p->gamma != 0.0) return fail(h, TF_ERR_UNSUPPORTED, "gamma != 0 is not implemented");
    if (p->use_initial_flow) return fail(h, TF_ERR_UNSUPPORTED, "useInitialFlow is not implemented");
    if (!(p->scale_step > 0.0 && p->scale_step < 1.0)) return fail(h, TF_ERR_INVALID_ARG, "scaleStep must be in (0,1)");
    if (p->scale_step == 0.5) return fail(h, TF_ERR_UNSUPPORTED, "scaleStep == 0.5: cv::resize's INTER_AREA fast path is not restated");
    if (!(p->theta > 0.0) || !(p->tau > 0.0) || !(p->lambda > 0.0) || !(p->epsilon >= 0.0)) return fail(h, TF_ERR_INVALID_ARG, "tau, lambda, theta must be > 0 and epsilon >= 0");
    if (p->variant != TF_VARIANT_CPU && p->variant != TF_VARIANT_CUDA) return fail(h, TF_ERR_INVALID_ARG, "bad variant");
    if (p->variant == TF_VARIANT_CUDA && (p->inner_iterations * p->outer_iterations) % 2) return fail(h, TF_ERR_UNSUPPORTED, "TF_VARIANT_CUDA needs an even iteration count");
    return TF_OK;
}

API int tf_abi_version(void) { return TF_ABI_VERSION; }
API int tf_device_count(void) { return 0; }

API int tf_default_params(tf_params* p)
{
    if (!p) return TF_ERR_INVALID_ARG;
    p->tau = 0.25; p->lambda = 0.15; p->theta = 0.3; p->epsilon = 0.01; p->scale_step = 0.8; p->gamma = 0.0;
    p->nscales = 5; p->warps = 5; p->inner_iterations = 30; p->outer_iterations = 10; p->median_filtering = 5;
    p->use_initial_flow = 0; p->algo = TF_ALGO_TVL1; p->max_batch = 0; p->variant = TF_VARIANT_CPU;
    return TF_OK;
}

API int tf_default_deepflow_params(tf_deepflow_params* p)
{
    if (!p) return TF_ERR_INVALID_ARG;
    p->sigma = 0.6f; p->min_size = 25; p->downscale_factor = 0.95f; p->fixed_point_iterations = 5; p->sor_iterations = 25;
    p->alpha = 1.0f; p->delta = 0.5f; p->gamma = 5.0f; p->omega = 1.6f; p->zeta = 0.1f; p->epsilon = 0.001f; p->max_batch = 0;
    return TF_OK;
}

API int tf_create(const tf_params* p, int device_id, tf_handle** out)
{
    (void)device_id;                                   /* the checker runs on the host cores */
    if (!out) return fail(NULL, TF_ERR_INVALID_ARG, "out == NULL");
    *out = NULL;
    tf_handle* h = (tf_handle*)calloc(1, sizeof *h);
    if (!h) return TF_ERR_NOMEM;
    if (p) h->P = *p; else tf_default_params(&h->P);
    tf_default_deepflow_params(&h->DP);
    int rc = validate(h, &h->P);
    if (rc) { snprintf(g_create_err, 256, "%s", h->err); free(h); return rc; }
    *out = h;
    return TF_OK;
}

API int tf_create_deepflow(const tf_deepflow_params* p, int device_id, tf_handle** out)
{
    int rc = tf_create(NULL, device_id, out);
    if (rc) return rc;
    (*out)->P.algo = TF_ALGO_DEEPFLOW;
    if (p) (*out)->DP = *p;
    return TF_OK;
}

API void tf_destroy(tf_handle* h) { if (h) { for (int k = 0; k < MAX_TICKETS; ++k) free(h->job[k].iters); free(h->iters); free(h); } }
API const char* tf_last_error(tf_handle* h) { return h ? h->err : g_create_err; }

API int tf_set_param(tf_handle* h, int key, double v)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (h->P.algo == TF_ALGO_DEEPFLOW) return fail(h, TF_ERR_UNSUPPORTED, "DeepFlow handles have creation-time parameters only");
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
        default: return fail(h, TF_ERR_INVALID_ARG, "unknown parameter key");
    }
    int rc = validate(h, &p);
    if (rc) return rc;
    h->P = p;
    return TF_OK;
}

API int tf_get_param(tf_handle* h, int key, double* v)
{
    if (!h || !v) return TF_ERR_INVALID_ARG;
    const tf_params* p = &h->P;
    switch (key) {
        case TF_PARAM_TAU: *v = p->tau; break;
        case TF_PARAM_LAMBDA: *v = p->lambda; break;
        case TF_PARAM_THETA: *v = p->theta; break;
        case TF_PARAM_NSCALES: *v = p->nscales; break;
        case TF_PARAM_WARPS: *v = p->warps; break;
        case TF_PARAM_EPSILON: *v = p->epsilon; break;
        case TF_PARAM_INNER_ITERATIONS: *v = p->inner_iterations; break;
        case TF_PARAM_OUTER_ITERATIONS: *v = p->outer_iterations; break;
        case TF_PARAM_SCALE_STEP: *v = p->scale_step; break;
        case TF_PARAM_GAMMA: *v = p->gamma; break;
        case TF_PARAM_MEDIAN_FILTERING: *v = p->median_filtering; break;
        case TF_PARAM_USE_INITIAL_FLOW: *v = p->use_initial_flow; break;
        default: return fail(h, TF_ERR_INVALID_ARG, "unknown parameter key");
    }
    return TF_OK;
}

/* pairs (a[i], b[i]) -> flow_out[i] * scale; iteration counts compacted to [pair][levels used][warps][2] like the engine */
static int solve(tf_handle* h, const uint8_t* a, const uint8_t* b, size_t stride_a, size_t stride_b, int n, int H, int W,
                 float scale, float* flow_out, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (!a || !b || !flow_out) return fail(h, TF_ERR_INVALID_ARG, "null image/flow pointer");
    if (H < 1 || W < 1 || n < 1) return fail(h, TF_ERR_INVALID_ARG, "bad sizes");
    const size_t fpx = (size_t)H * W;
    const int deep = h->P.algo == TF_ALGO_DEEPFLOW;
    orc_params op;
    op.tau = h->P.tau; op.lambda = h->P.lambda; op.theta = h->P.theta; op.epsilon = h->P.epsilon; op.scale_step = h->P.scale_step;
    op.gamma = h->P.gamma; op.nscales = h->P.nscales; op.warps = h->P.warps; op.inner_iterations = h->P.inner_iterations;
    op.outer_iterations = h->P.outer_iterations; op.median_filtering = h->P.median_filtering; op.use_initial_flow = 0;
    op.err_mode = 0; op.variant = h->P.variant;
    dfo_params dp;
    dp.sigma = h->DP.sigma; dp.min_size = h->DP.min_size; dp.downscale_factor = h->DP.downscale_factor;
    dp.fixed_point_iterations = h->DP.fixed_point_iterations; dp.sor_iterations = h->DP.sor_iterations; dp.alpha = h->DP.alpha;
    dp.delta = h->DP.delta; dp.gamma = h->DP.gamma; dp.omega = h->DP.omega; dp.zeta = h->DP.zeta; dp.epsilon = h->DP.epsilon;
    const size_t per = (size_t)h->P.nscales * h->P.warps * 2;
    int* full = deep ? NULL : (int*)calloc(per * (size_t)n, sizeof(int));
    int nlev = 0;
    unsigned long long n_in = 0, n_out = 0;
    for (int i = 0; i < n; ++i) {
        float* f = flow_out + (size_t)i * fpx * 2;
        const int rc = deep ? dfo_deepflow_calc(&dp, a + i * stride_a, b + i * stride_b, H, W, f)
                            : orc_tvl1_calc(&op, a + i * stride_a, b + i * stride_b, H, W, f, full + per * i);
        if (rc <= 0) { free(full); return fail(h, TF_ERR_INVALID_ARG, "the oracle rejected the call"); }
        nlev = rc;
        if (scale != 1.0f) for (size_t k = 0; k < fpx * 2; ++k) f[k] *= scale;
    }
    free(h->iters); h->iters = NULL; h->n_iters = 0;
    h->last_pairs = n; h->last_nlev = nlev;
    if (!deep) {
        const size_t used = (size_t)nlev * h->P.warps * 2;
        h->iters = (int*)malloc(used * (size_t)n * sizeof(int));
        h->n_iters = used * (size_t)n;
        for (int i = 0; i < n; ++i) {
            memcpy(h->iters + used * i, full + per * i, used * sizeof(int));
            for (size_t k = 0; k < used; k += 2) { n_in += (unsigned)full[per * i + k]; n_out += (unsigned)full[per * i + k + 1]; }
        }
        free(full);
    }
    if (st) {
        memset(st, 0, sizeof *st);
        st->n_pairs = n; st->nscales_used = nlev; st->warps = deep ? 0 : h->P.warps;
        st->inner_iters_total = n_in; st->outer_iters_total = n_out; st->iter_pair_steps = n_in;
    }
    return TF_OK;
}

API int tf_calc_pair(tf_handle* h, const uint8_t* I0, const uint8_t* I1, int H, int W, float* flow_out, tf_stats* st)
{
    return solve(h, I0, I1, 0, 0, 1, H, W, 1.0f, flow_out, st);
}

API int tf_calc_pairs(tf_handle* h, const uint8_t* I0s, const uint8_t* I1s, int B, int H, int W, float* flow_out, tf_stats* st)
{
    const size_t fpx = (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0);
    return solve(h, I0s, I1s, fpx, fpx, B, H, W, 1.0f, flow_out, st);
}

API int tf_calc_seq(tf_handle* h, const uint8_t* frames, int N, int H, int W, float scale, float* flow_out, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    if (N < 2) return fail(h, TF_ERR_INVALID_ARG, "a sequence needs at least 2 frames");
    const size_t fpx = (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0);
    return solve(h, frames, frames ? frames + fpx : NULL, fpx, fpx, N - 1, H, W, scale, flow_out, st);
}

API int tf_get_iters(tf_handle* h, int* out, size_t capacity_ints, size_t* written)
{
    if (!h || !out) return TF_ERR_INVALID_ARG;
    const size_t n = h->n_iters < capacity_ints ? h->n_iters : capacity_ints;
    if (n) memcpy(out, h->iters, n * sizeof(int));
    if (written) *written = n;
    return TF_OK;
}

/* Asynchronous forms (include/teeflow.h): the checker has no lanes, so the job is solved when it is submitted and tf_wait hands its
 * results over -- same observable protocol (tickets, order-free waiting, iteration counts after the wait). */
static int stash(tf_handle* h, int rc, const tf_stats* st, int* ticket)
{
    if (rc) return rc;
    if (!ticket) return fail(h, TF_ERR_INVALID_ARG, "ticket == NULL");
    int k = -1;
    for (int i = 0; i < MAX_TICKETS; ++i) if (!h->job[i].used) { k = i; break; }
    if (k < 0) return fail(h, TF_ERR_NOMEM, "too many jobs not waited for");
    h->job[k].used = ++h->next_ticket; h->job[k].st = *st;
    h->job[k].iters = h->iters; h->job[k].n_iters = h->n_iters; h->job[k].pairs = h->last_pairs; h->job[k].nlev = h->last_nlev;
    h->iters = NULL; h->n_iters = 0;
    *ticket = h->job[k].used;
    return TF_OK;
}
API int tf_submit_pairs(tf_handle* h, const uint8_t* I0s, const uint8_t* I1s, int B, int H, int W, float* flow_out, int* ticket)
{
    tf_stats st;
    if (!h) return TF_ERR_INVALID_ARG;
    return stash(h, tf_calc_pairs(h, I0s, I1s, B, H, W, flow_out, &st), &st, ticket);
}
API int tf_submit_seq(tf_handle* h, const uint8_t* frames, int N, int H, int W, float scale, float* flow_out, int* ticket)
{
    tf_stats st;
    if (!h) return TF_ERR_INVALID_ARG;
    return stash(h, tf_calc_seq(h, frames, N, H, W, scale, flow_out, &st), &st, ticket);
}
API int tf_wait(tf_handle* h, int ticket, tf_stats* st)
{
    if (!h) return TF_ERR_INVALID_ARG;
    for (int pass = 0; pass < (ticket < 0 ? MAX_TICKETS : 1); ++pass) {
        int k = -1;
        for (int i = 0; i < MAX_TICKETS; ++i)
            if (h->job[i].used && (ticket < 0 ? (k < 0 || h->job[i].used < h->job[k].used) : h->job[i].used == ticket)) k = i;
        if (k < 0) { if (ticket < 0) return TF_OK; return fail(h, TF_ERR_INVALID_ARG, "unknown ticket"); }
        free(h->iters);
        h->iters = h->job[k].iters; h->n_iters = h->job[k].n_iters; h->last_pairs = h->job[k].pairs; h->last_nlev = h->job[k].nlev;
        if (st) *st = h->job[k].st;
        h->job[k].iters = NULL; h->job[k].used = 0;
    }
    return TF_OK;
}
