// capi.hip -- host side of libmvslam_hip.so: contexts, resident batches, the C ABI of include/mvslam_hip.h.
// No CPU fallback exists anywhere in this file: every entry point ends in a HIP kernel launch.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "kernels.hpp"

using namespace mvs;

struct mvs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second queue of a large batch: its pairs go down the pipeline as two independent halves, one per stream, so that the
    // latency-bound kernels of one half (list sort, exact solve of the few survivors, selection, ...) run under the
    // throughput-bound kernels of the other (enqueue_pipeline).  Forked from / joined into `stream` with the two events.
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool half_batches = true;
    int cu_count = 256;   // hipDeviceAttributeMultiprocessorCount of `device` (mvs_ctx_create); MI355X: 256
    std::string err;
    mvs_batch *scratch = nullptr;  // batch of one pair backing the single-shot entry points
    double *d_uv1 = nullptr, *d_uv2 = nullptr;
    int uv_cap = 0;
    double *d_small = nullptr;  // 64 doubles of staging (fundamental_kernel)
    unsigned char *d_single = nullptr;   // gathered outputs of a single-shot call (one device-to-host copy)
    size_t single_cap = 0;
    unsigned char *d_single_in = nullptr;   // packed inputs of mvs_image_pair (one host-to-device copy)
    size_t single_in_cap = 0;
    void *d_pnp = nullptr;      // pnp_solve workspace
    size_t pnp_bytes = 0;
    void *d_ref = nullptr;      // sfm_refine / pnp_refine workspace
    size_t ref_bytes = 0;
    void *d_orb = nullptr;      // extraction workspace
    int32_t *h_orb_ovf = nullptr;   // pinned: the extraction's overflow flag travels with the outputs (one stream wait per call)
    size_t orb_bytes = 0;
    bool orb_ready = false;
    // the ~50 launches of one extraction, captured once per (batch shape, parameters, buffers) and replayed
    hipGraphExec_t orb_graph = nullptr;
    OrbDev orb_graph_key{};
    bool orb_graph_valid = false;
    // pinned staging arena of the single-shot entry points: small parameters and results travel through it with
    // hipMemcpyAsync on the ctx stream (no synchronous pageable copies, no sync "so that a stack temporary may die")
    char *h_pin = nullptr;
    size_t h_pin_cap = 0, h_pin_off = 0;
    bool pin_in_flight = false;   // an asynchronous copy may still be reading / writing the arena (set by pin_get, cleared
                                  // by whoever synchronises the stream at the end of the call)
};

struct mvs_seq;

struct mvs_batch {
    mvs_ctx *ctx = nullptr;
    BatchDev d{};
    std::vector<void *> allocs;
    double *uv1 = nullptr, *uv2 = nullptr;   // [n_pairs][max_kp][2] staging of mvs_batch_run_points (first use)
    int hyp_table_cap = 0;  // capacity of the optional per-hypothesis tables
    int32_t *allocs_hc = nullptr;
    double *allocs_hr = nullptr;
    hipEvent_t ev[8]{};
    RefineDev refine{};     // allocated by the first mvs_batch_refine
    bool refine_ready = false, refine_ran = false;
    // pinned staging of the per-pair parameters derived on the host (K^-1, default indices): two slot sets used in turn,
    // each guarded by an event recorded behind its copies, so an asynchronous upload waits for the copies of the upload
    // before last at most -- never for the kernels or downloads queued on the stream in between
    char *h_pin = nullptr;
    hipEvent_t pin_ev[2]{};
    bool pin_ev_live[2] = {false, false};
    int pin_turn = 0;
};

struct mvs_seq {
    mvs_ctx *ctx = nullptr;
    mvs_batch *batch = nullptr;  // n_frames - 1 pairs viewing the frame arrays
    int n_frames = 0, n_tracks = 0, stride = 0, rec_groups = 0;
    SeqJoinDev join{};
    PnpDev pnp{};
    SeqChainDev chain{};
    RefineDev refit{};          // pnp_solve's refit over the inliers of every track (mvs_pnp_params.refit), allocated on demand
    bool refit_ready = false, refit_on = false;
    std::vector<void *> allocs;
};

#define HIP_TRY(ctx_, expr)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            if (ctx_)                                                                          \
                (ctx_)->err = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return MVS_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// every wait for the ctx stream goes through here: once it has drained nothing reads or writes the arena any more
static hipError_t sync_stream(mvs_ctx *ctx)
{
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess)
        ctx->pin_in_flight = false;
    return e;
}
// ---- pinned staging arena -------------------------------------------------------------------------------------------
// pin_begin() opens a call: the arena is made large enough for everything the call will stage (growing it waits for
// the stream first, nothing of an earlier call may still be in flight) and the cursor rewinds.  pin_put() copies a host
// block in and returns its pinned address, pin_get() reserves room for a device -> host copy.
static mvs_status pin_begin(mvs_ctx *ctx, size_t bytes)
{
    bytes += 4096;
    if (ctx->h_pin_cap < bytes) {
        HIP_TRY(ctx, sync_stream(ctx));
        if (ctx->h_pin)
            (void)hipHostFree(ctx->h_pin);
        ctx->h_pin = nullptr;
        ctx->h_pin_cap = 0;
        const size_t cap = std::max<size_t>(bytes * 2, size_t(1) << 20);
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_pin, cap, hipHostMallocDefault));
        ctx->h_pin_cap = cap;
    }
    if (ctx->pin_in_flight) {
        // the previous call left through an error path after it had enqueued copies from / into the arena: wait for
        // them before the cursor rewinds and their source bytes are overwritten
        HIP_TRY(ctx, sync_stream(ctx));
        ctx->pin_in_flight = false;
    }
    ctx->h_pin_off = 0;
    return MVS_OK;
}
static void *pin_get(mvs_ctx *ctx, size_t bytes)
{
    const size_t off = (ctx->h_pin_off + 63) & ~size_t(63);
    if (off + bytes > ctx->h_pin_cap)
        return nullptr;   // pin_begin() was given too small a figure: a bug, reported as MVS_ERR_HIP by the callers
    ctx->h_pin_off = off + bytes;
    ctx->pin_in_flight = true;
    return ctx->h_pin + off;
}
static void *pin_put(mvs_ctx *ctx, const void *src, size_t bytes)
{
    void *p = pin_get(ctx, bytes);
    if (p && bytes)
        std::memcpy(p, src, bytes);
    return p;
}
#define PIN_TRY(ctx_, ptr_)                                   \
    do {                                                      \
        if (!(ptr_)) {                                        \
            (ctx_)->err = "pinned staging arena exhausted";   \
            return MVS_ERR_HIP;                               \
        }                                                     \
    } while (0)
// host block -> pinned arena -> device, asynchronous
static mvs_status up_async(mvs_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    void *p = pin_put(ctx, src, bytes);
    PIN_TRY(ctx, p);
    HIP_TRY(ctx, hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MVS_OK;
}

// pnp_solve's refit (pnp-solve.cpp:53-64) as a refinement problem: points fixed (sigma 1e-9), no prior on the pose
constexpr double kRefitPointSigma = 1e-9, kRefitPoseSigma = 1e6;
static RefineCfg to_cfg(const mvs_refine_params &p, int n_frames);

static RunParams to_run(const mvs_params &p)
{
    RunParams r;
    r.ratio = p.ratio;
    r.max_dist = p.max_dist;
    r.max_error_sq = p.max_error_sq;
    r.num_hypotheses = p.num_hypotheses;
    r.sampler = p.sampler;
    r.seed = p.seed;
    r.min_inliers = p.min_inliers;
    return r;
}

// PinholeCamera caches K.inverse() (vision/camera.cpp:16): fixed-size 3x3 cofactor inverse,
// inv(r, c) = cofactor(c, r) * (1 / det), det expanded along the first column.
static void mat3_inverse(const double *K, double *inv)
{
    auto m = [&](int r, int c) { return K[(r % 3) * 3 + (c % 3)]; };
    auto cof = [&](int i, int j) { return m(i + 1, j + 1) * m(i + 2, j + 2) - m(i + 1, j + 2) * m(i + 2, j + 1); };
    const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const double det = (c00 * K[0] + c10 * K[3]) + c20 * K[6];
    const double invdet = 1.0 / det;
    inv[0] = c00 * invdet;       inv[1] = c10 * invdet;       inv[2] = c20 * invdet;
    inv[3] = cof(0, 1) * invdet; inv[4] = cof(1, 1) * invdet; inv[5] = cof(2, 1) * invdet;
    inv[6] = cof(0, 2) * invdet; inv[7] = cof(1, 2) * invdet; inv[8] = cof(2, 2) * invdet;
}

static bool affine_K(const double *K) { return K[6] == 0.0 && K[7] == 0.0 && K[8] == 1.0 && K[0] != 0.0 && K[4] != 0.0; }

template <typename T>
static mvs_status dev_alloc(mvs_batch *b, T **ptr, size_t count)
{
    void *p = nullptr;
    HIP_TRY(b->ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    b->allocs.push_back(p);
    *ptr = static_cast<T *>(p);
    return MVS_OK;
}

template <typename T>
static mvs_status seq_alloc(mvs_seq *q, T **ptr, size_t count)
{
    void *p = nullptr;
    HIP_TRY(q->ctx, hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    q->allocs.push_back(p);
    *ptr = static_cast<T *>(p);
    return MVS_OK;
}

// release one block of a batch early (superseded by a larger one); the stream must be idle
static void dev_release(mvs_batch *b, void *p)
{
    if (!p)
        return;
    for (size_t i = 0; i < b->allocs.size(); ++i)
        if (b->allocs[i] == p) {
            b->allocs.erase(b->allocs.begin() + i);
            (void)hipFree(p);
            return;
        }
}

// per-hypothesis buffers of the RANSAC stage, sized for the largest hypothesis count seen so far.  All new blocks are
// allocated before any pointer is switched (a failed allocation leaves the batch as it was, its partial blocks owned by
// allocs until destroy); the superseded blocks are freed at once: 77 B x hypotheses x pairs is 2 GB at the bench size.
static mvs_status ensure_groups(mvs_batch *b, int num_hypotheses)
{
    const int G = (num_hypotheses + kHypPerBlock - 1) / kHypPerBlock;
    if (G <= b->d.max_groups)
        return MVS_OK;
    HIP_TRY(b->ctx, sync_stream(b->ctx));
    const size_t P = (size_t)b->d.n_pairs, Hp = (size_t)G * kHypPerBlock;
    if (P * Hp >= (size_t(1) << 32))
        return MVS_ERR_CAPACITY;   // the exact solve's work list holds flat 32-bit record indices
    WgBest *p = nullptr;
    double *hf = nullptr;   // record of every hypothesis (F + counting threshold): 80 B x hypotheses x pairs
    uint8_t *ho = nullptr;
    int32_t *hc = nullptr;
    uint32_t *xl = nullptr;
    float *h32 = nullptr;   // single-precision pre-screen records (mode 1): 48 B x hypotheses x pairs
    mvs_status st;
    if ((st = dev_alloc(b, &p, P * G)) != MVS_OK) return st;
    if ((st = dev_alloc(b, &hf, P * kHypRec * Hp)) != MVS_OK) return st;
    if ((st = dev_alloc(b, &h32, P * kHypRec32 * Hp)) != MVS_OK) return st;
    if ((st = dev_alloc(b, &ho, P * Hp)) != MVS_OK) return st;
    if ((st = dev_alloc(b, &hc, P * Hp)) != MVS_OK) return st;
    if ((st = dev_alloc(b, &xl, P * Hp)) != MVS_OK) return st;
    uint32_t *cl = nullptr;
    if ((st = dev_alloc(b, &cl, 2 * P * Hp)) != MVS_OK) return st;   // the list + its sorted copy
    if (!b->d.bound && (st = dev_alloc(b, &b->d.bound, P)) != MVS_OK) return st;
    if (!b->d.box && (st = dev_alloc(b, &b->d.box, P * 8)) != MVS_OK) return st;
    if (!b->d.mode && (st = dev_alloc(b, &b->d.mode, P)) != MVS_OK) return st;
    if (!b->d.dense_n1 && (st = dev_alloc(b, &b->d.dense_n1, P)) != MVS_OK) return st;
    if (!b->d.ccount && (st = dev_alloc(b, &b->d.ccount, P)) != MVS_OK) return st;
    if (!b->d.pcount && (st = dev_alloc(b, &b->d.pcount, P)) != MVS_OK) return st;
    if (!b->d.cpos && (st = dev_alloc(b, &b->d.cpos, P * kSortBins)) != MVS_OK) return st;
    if (!b->d.m0list && (st = dev_alloc(b, &b->d.m0list, 2 * (P + 1))) != MVS_OK) return st;   // one list per half (batch_view)
    if (!b->d.xcount && (st = dev_alloc(b, &b->d.xcount, 4)) != MVS_OK) return st;
    dev_release(b, b->d.wgbest);
    dev_release(b, b->d.hyp_F);
    dev_release(b, b->d.hyp_r32);
    dev_release(b, b->d.hyp_okf);
    dev_release(b, b->d.hyp_cnt);
    dev_release(b, b->d.xlist);
    dev_release(b, b->d.clist);
    b->d.wgbest = p;
    b->d.hyp_F = hf;
    b->d.hyp_r32 = h32;
    b->d.hyp_okf = ho;
    b->d.hyp_cnt = hc;
    b->d.xlist = xl;
    b->d.clist = cl;
    b->d.clist2 = cl + P * Hp;
    b->d.max_groups = G;
    return MVS_OK;
}

#ifdef MVS_DEBUG_HOOKS
// generic "n inputs -> m outputs" staging for the two probes below
template <typename Launch>
static int probe_io(mvs_ctx *ctx, const void *const *in, const size_t *in_bytes, int n_in, void *const *outp, const size_t *out_bytes,
                    int n_out, Launch launch)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *d[8] = {};
    hipError_t e = hipSuccess;
    for (int k = 0; k < n_in + n_out && e == hipSuccess; ++k)
        e = hipMalloc(&d[k], k < n_in ? in_bytes[k] : out_bytes[k - n_in]);
    for (int k = 0; k < n_in && e == hipSuccess; ++k)
        e = hipMemcpy(d[k], in[k], in_bytes[k], hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch(d);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = sync_stream(ctx);
    for (int k = 0; k < n_out && e == hipSuccess; ++k)
        e = hipMemcpy(outp[k], d[n_in + k], out_bytes[k], hipMemcpyDeviceToHost);
    for (int k = 0; k < n_in + n_out; ++k)
        if (d[k])
            (void)hipFree(d[k]);
    if (e != hipSuccess) {
        ctx->err = std::string("probe: ") + hipGetErrorString(e);
        return MVS_ERR_HIP;
    }
    return MVS_OK;
}

#endif

extern "C" {

int mvs_abi_version(void) { return MVS_ABI_VERSION; }

#ifdef MVS_DEBUG_HOOKS
// Diagnostics, compiled ONLY into libmvslam_hip_dbg.so (make dbg; tests/ and tools/ load that library explicitly).  The
// product library has neither symbol: a process-global kernel-variant switch is not something a caller of the
// reference's interface should be able to reach.
// diagnostics only (not in the public header): bitwise comparison of the unscaled sqrt / div device sequences with
// the compiler's IEEE ones on host-supplied operands.  counts[4] = {sqrt mismatches, div mismatches, sqrt checked,
// div checked}
int mvs_debug_fastmath_check(mvs_ctx *ctx, const double *x, const double *y, int n, unsigned long long counts[4])
{
    if (!ctx || n < 1)
        return MVS_ERR_INVALID_ARG;
    double *dx = nullptr, *dy = nullptr;
    unsigned long long *dc = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc((void **)&dx, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc((void **)&dy, (size_t)n * 8));
    HIP_TRY(ctx, hipMalloc((void **)&dc, 32));
    HIP_TRY(ctx, hipMemcpyAsync(dx, x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dy, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dc, 0, 32, ctx->stream));
    launch_fastmath_check(dx, dy, n, dc, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(counts, dc, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, sync_stream(ctx));
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(dc);
    return MVS_OK;
}

// diagnostics only: one guarded Jacobi pair step per row pair against the IEEE one (pairstep_check_kernel).
// rows: n x 6 doubles.  counts[4] = {steps that differ, steps compared, decisions that differ, bits of the largest
// |1 - gamma * 2 h| seen (accuracy of the reciprocal estimate the seeded divisions start from)}
int mvs_debug_pairstep_check(mvs_ctx *ctx, const double *rows, int n, unsigned long long counts[4])
{
    if (!ctx || n < 1)
        return MVS_ERR_INVALID_ARG;
    double *dr = nullptr;
    unsigned long long *dc = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc((void **)&dr, (size_t)n * 48));
    if (hipMalloc((void **)&dc, 32) != hipSuccess) {
        (void)hipFree(dr);
        return MVS_ERR_HIP;
    }
    hipError_t e = hipMemcpyAsync(dr, rows, (size_t)n * 48, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(dc, 0, 32, ctx->stream);
    if (e == hipSuccess) {
        launch_pairstep_check(dr, n, dc, ctx->stream);
        e = hipMemcpyAsync(counts, dc, 32, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess)
        e = sync_stream(ctx);
    (void)hipFree(dr);
    (void)hipFree(dc);
    return e == hipSuccess ? MVS_OK : MVS_ERR_HIP;
}

// diagnostics only (not in the public header): pick a co-compiled ransac_kernel variant for A/B timing
int mvs_debug_set_ransac_variant(int v)
{
    const int old = get_ransac_variant();
    set_ransac_variant(v);
    return old;
}

// diagnostics only: the per-hypothesis F records the solve launch handed to the scoring launch (pair `pair` of a batch)
int mvs_debug_set_count_dense(int v)
{
    set_count_dense(v);
    return MVS_OK;
}

// one 32 x 32 x 32 bf16 tile through the counting kernels' two MFMAs: a, b = [32][32] bf16 bit patterns on the host,
// out = [32][32] binary32 (point x hypothesis)
int mvs_debug_mfma_probe(mvs_ctx *ctx, const uint16_t *a, const uint16_t *b, float *out)
{
    if (!ctx || !a || !b || !out)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint16_t *da = nullptr, *db = nullptr;
    float *dout = nullptr;
    HIP_TRY(ctx, hipMalloc(&da, 1024 * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMalloc(&db, 1024 * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMalloc(&dout, 1024 * sizeof(float)));
    HIP_TRY(ctx, hipMemcpy(da, a, 1024 * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(db, b, 1024 * sizeof(uint16_t), hipMemcpyHostToDevice));
    launch_mfma_probe(da, db, dout, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipMemcpy(out, dout, 1024 * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return MVS_OK;
}

int mvs_debug_set_split_min_pairs(int v)
{
    set_split_min_pairs(v);
    return MVS_OK;
}

int mvs_debug_set_match_mfma(int v)
{
    set_match_mfma(v);
    return MVS_OK;
}

int mvs_debug_set_prescreen_force(int m)
{
    set_prescreen_force(m);
    return MVS_OK;
}

// records of the RANSAC stage as the last run left them: rec_out n_hyp x 10 (F, counting threshold), state bytes, counts;
// info[0..3] = {mode of the pair, bound of the pair, work-list entries 0, work-list entries 1}
int mvs_debug_read_hyp_rec(mvs_batch *b, int pair, int n_hyp, double *rec_out, unsigned char *state_out, int32_t *cnt_out,
                           int32_t *info)
{
    if (!b || !b->d.hyp_F || pair < 0 || pair >= b->d.n_pairs || n_hyp < 1 || n_hyp > b->d.max_groups * kHypPerBlock)
        return MVS_ERR_INVALID_ARG;
    const size_t Hp = (size_t)b->d.max_groups * kHypPerBlock;
    HIP_TRY(b->ctx, sync_stream(b->ctx));
    std::vector<unsigned char> stv(n_hyp);
    HIP_TRY(b->ctx, hipMemcpy(stv.data(), b->d.hyp_okf + (size_t)pair * Hp, (size_t)n_hyp, hipMemcpyDeviceToHost));
    if (rec_out) {
        HIP_TRY(b->ctx, hipMemcpy(rec_out, b->d.hyp_F + (size_t)pair * Hp * kHypRec, (size_t)n_hyp * kHypRec * sizeof(double),
                                  hipMemcpyDeviceToHost));
        // a pair in mode 1 keeps its APPROXIMATE records in the 48-byte single-precision array: they are returned in the
        // first 48 bytes of the hypothesis' 80-byte slot (the layout the records had when they shared one array)
        int32_t pmode = 0;
        HIP_TRY(b->ctx, hipMemcpy(&pmode, b->d.mode + pair, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (pmode == 1) {
            std::vector<float> r32((size_t)n_hyp * kHypRec32);
            HIP_TRY(b->ctx, hipMemcpy(r32.data(), b->d.hyp_r32 + (size_t)pair * Hp * kHypRec32, r32.size() * sizeof(float),
                                      hipMemcpyDeviceToHost));
            for (int h = 0; h < n_hyp; ++h)
                if (stv[h] == 1)
                    std::memcpy(rec_out + (size_t)h * kHypRec, r32.data() + (size_t)h * kHypRec32, kHypRec32 * sizeof(float));
        }
    }
    if (state_out)
        std::memcpy(state_out, stv.data(), (size_t)n_hyp);
    if (cnt_out)
        HIP_TRY(b->ctx, hipMemcpy(cnt_out, b->d.hyp_cnt + (size_t)pair * Hp, (size_t)n_hyp * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (info) {
        uint32_t xc[2] = {0, 0};
        HIP_TRY(b->ctx, hipMemcpy(&info[0], b->d.mode + pair, sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(b->ctx, hipMemcpy(&info[1], b->d.bound + pair, sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(b->ctx, hipMemcpy(xc, b->d.xcount, sizeof(xc), hipMemcpyDeviceToHost));
        info[2] = (int32_t)xc[0];
        info[3] = (int32_t)xc[1];
    }
    return MVS_OK;
}

// the pre-screen alone over pairs [0, n_active) of a batch that has been run (matches and points resident): pair_prepare +
// ransac_prescreen, every pair forced into pre-screened mode `mode` (1: single-precision records, 2: double precision),
// nothing solved exactly afterwards -- the records then hold F~ and the widened thresholds (state 1), or wait for the exact
// solve (state 2): tests compare them with the oracle's exact F
int mvs_debug_prescreen_only(mvs_batch *b, const mvs_params *params, int n_active, int mode)
{
    if (!b || !params || n_active < 1 || n_active > b->d.n_pairs || (mode != 1 && mode != 2))
        return MVS_ERR_INVALID_ARG;
    mvs_status st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    launch_prescreen_only(b->d, to_run(*params), n_active, mode, b->ctx->stream);
    HIP_TRY(b->ctx, hipGetLastError());
    HIP_TRY(b->ctx, sync_stream(b->ctx));
    return MVS_OK;
}

// full-population audit (kernels.hip: audit_kernel).  phase 0: pair_prepare + ransac_prescreen run here (the probe decides every
// pair's mode), then every record is checked against the exact solve of its sample; phase 1: the caller has just run the
// batch with the same parameters, the stage's decisions are checked.  counters[16] as documented at audit_kernel; maxc /
// bound / mode: [n_active] (largest exact count, the stage's bound, the pair's mode), each may be null.
int mvs_debug_audit(mvs_batch *b, const mvs_params *params, int n_active, int phase, unsigned long long counters[16],
                    int32_t *maxc, int32_t *bound, int32_t *mode)
{
    if (!b || !params || !counters || n_active < 1 || n_active > b->d.n_pairs || (phase != 0 && phase != 1))
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mvs_status st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    const RunParams rp = to_run(*params);
    unsigned long long *dc = nullptr;
    int32_t *dm = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&dc, 16 * sizeof(unsigned long long)));
    if (hipMalloc((void **)&dm, (size_t)n_active * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(dc);
        return MVS_ERR_HIP;
    }
    hipError_t e = hipMemsetAsync(dc, 0, 16 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(dm, 0xff, (size_t)n_active * sizeof(int32_t), ctx->stream);
    if (e == hipSuccess && phase == 0)
        launch_prescreen_only(b->d, rp, n_active, -1, ctx->stream);
    if (e == hipSuccess)
        e = launch_audit(b->d, rp, n_active, phase, dc, dm, ctx->stream);
    if (e == hipSuccess)
        e = sync_stream(ctx);
    if (e == hipSuccess)
        e = hipMemcpy(counters, dc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess && maxc)
        e = hipMemcpy(maxc, dm, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && bound)
        e = hipMemcpy(bound, b->d.bound, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mode)
        e = hipMemcpy(mode, b->d.mode, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipFree(dc);
    (void)hipFree(dm);
    if (e != hipSuccess) {
        ctx->err = std::string("mvs_debug_audit: ") + hipGetErrorString(e);
        return MVS_ERR_HIP;
    }
    return MVS_OK;
}

// The same audit over the device state of ANOTHER library instance's batch (mvs_batch_device_state of libmvslam_hip.so, loaded
// in the same process): the product binary ran the stage, this library only replays every hypothesis exactly and compares.
// phase 1: the stage's decisions; phase 2: the records as the product's pre-screen wrote them and the stage left them (every
// record still in state kPsApprox: (B) on every match, U >= c_J >= L) -- phase 0's checks WITHOUT re-running the pre-screen.
int mvs_debug_audit_state(mvs_ctx *ctx, const void *state, size_t state_bytes, const mvs_params *params, int n_active, int phase,
                          unsigned long long counters[16], int32_t *maxc, int32_t *bound, int32_t *mode)
{
    struct Blob {
        uint64_t bytes, abi;
        BatchDev d;
    };
    if (!ctx || !state || !params || !counters || (phase != 1 && phase != 2))
        return MVS_ERR_INVALID_ARG;
    Blob blob;
    if (state_bytes != sizeof(Blob))
        return MVS_ERR_INVALID_ARG;
    std::memcpy(&blob, state, sizeof(Blob));
    if (blob.bytes != sizeof(Blob) || blob.abi != (uint64_t)MVS_ABI_VERSION)
        return MVS_ERR_INVALID_ARG;   // not the same build
    const BatchDev &d = blob.d;
    if (n_active < 1 || n_active > d.n_pairs || (params->num_hypotheses + kHypPerBlock - 1) / kHypPerBlock > d.max_groups)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const RunParams rp = to_run(*params);
    unsigned long long *dc = nullptr;
    int32_t *dm = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&dc, 16 * sizeof(unsigned long long)));
    if (hipMalloc((void **)&dm, (size_t)n_active * sizeof(int32_t)) != hipSuccess) {
        (void)hipFree(dc);
        return MVS_ERR_HIP;
    }
    hipError_t e = hipMemsetAsync(dc, 0, 16 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(dm, 0xff, (size_t)n_active * sizeof(int32_t), ctx->stream);
    if (e == hipSuccess)
        e = launch_audit(d, rp, n_active, phase == 2 ? 0 : 1, dc, dm, ctx->stream);
    if (e == hipSuccess)
        e = sync_stream(ctx);
    if (e == hipSuccess)
        e = hipMemcpy(counters, dc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess && maxc)
        e = hipMemcpy(maxc, dm, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && bound)
        e = hipMemcpy(bound, d.bound, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mode)
        e = hipMemcpy(mode, d.mode, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipFree(dc);
    (void)hipFree(dm);
    if (e != hipSuccess) {
        ctx->err = std::string("mvs_debug_audit_state: ") + hipGetErrorString(e);
        return MVS_ERR_HIP;
    }
    return MVS_OK;
}

// overwrite the ideal-camera points of one pair: pts4 = m x (x1, y1, x2, y2) doubles (the matcher's output is bypassed)
int mvs_debug_set_points(mvs_batch *b, int pair, int m, const double *pts4)
{
    if (!b || !pts4 || pair < 0 || pair >= b->d.n_pairs || m < 0 || m > b->d.max_kp)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipMemcpy(b->d.pts + (size_t)pair * b->d.max_kp * 4, pts4, (size_t)m * 4 * sizeof(double), hipMemcpyHostToDevice));
    const int32_t mm = m;
    HIP_TRY(ctx, hipMemcpy(b->d.M + pair, &mm, sizeof(mm), hipMemcpyHostToDevice));
    return MVS_OK;
}

// read the ideal-camera points of one pair as the kernels see them: pts4 = capacity x 4 doubles, *m = the pair's match count
int mvs_debug_get_points(mvs_batch *b, int pair, int *m, double *pts4)
{
    if (!b || !pts4 || !m || pair < 0 || pair >= b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_stream(ctx));
    int32_t mm = 0;
    HIP_TRY(ctx, hipMemcpy(&mm, b->d.M + pair, sizeof(mm), hipMemcpyDeviceToHost));
    mm = std::min(mm, b->d.max_kp);
    HIP_TRY(ctx, hipMemcpy(pts4, b->d.pts + (size_t)pair * b->d.max_kp * 4, (size_t)mm * 4 * sizeof(double), hipMemcpyDeviceToHost));
    *m = mm;
    return MVS_OK;
}

// pre-screen (every pair forced into mode pmode: 1 single-, 2 double-precision records) + the counting launches alone, with
// only hypothesis keep[p] of pair p left valid (keep may be null).  dense: 0 = ransac_count32 in one launch, 1 = the product's
// pilot + matrix-core dense phase + finish.  Out, each [n_active] and optional: the kept hypothesis' recorded count (U), the
// pair's bound (best lower bound), the points the dense phase covered.
int mvs_debug_count_only(mvs_batch *b, const mvs_params *params, int n_active, int pmode, int dense, const int32_t *keep,
                         int32_t *cnt_out, int32_t *bound_out, int32_t *n1_out)
{
    if (!b || !params || n_active < 1 || n_active > b->d.n_pairs || (pmode != 1 && pmode != 2) || dense < 0 || dense > 2)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mvs_status st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    const size_t Hp = (size_t)b->d.max_groups * kHypPerBlock;
    int32_t *dk = nullptr;
    if (keep) {
        for (int p = 0; p < n_active; ++p)
            if (keep[p] >= (int32_t)Hp)
                return MVS_ERR_INVALID_ARG;
        HIP_TRY(ctx, hipMalloc((void **)&dk, (size_t)n_active * sizeof(int32_t)));
        HIP_TRY(ctx, hipMemcpy(dk, keep, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipMemsetAsync(b->d.dense_n1, 0, (size_t)n_active * sizeof(int32_t), ctx->stream));
    launch_count_only(b->d, to_run(*params), n_active, pmode, dense, dk, ctx->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = sync_stream(ctx);
    for (int p = 0; p < n_active && e == hipSuccess; ++p) {
        if (cnt_out && keep && keep[p] >= 0)
            e = hipMemcpy(cnt_out + p, b->d.hyp_cnt + (size_t)p * Hp + keep[p], sizeof(int32_t), hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess && bound_out)
        e = hipMemcpy(bound_out, b->d.bound, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess && n1_out)
        e = hipMemcpy(n1_out, b->d.dense_n1, (size_t)n_active * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (dk)
        (void)hipFree(dk);
    if (e != hipSuccess) {
        ctx->err = std::string("mvs_debug_count_only: ") + hipGetErrorString(e);
        return MVS_ERR_HIP;
    }
    return MVS_OK;
}

// the matrix-core counting's compare-free indicator on caller-supplied accumulator values (kernels.hip: indicator_probe_kernel)
int mvs_debug_indicator_probe(mvs_ctx *ctx, const float *a, const float *tu, const float *tl, const float *T, int n, float *ind_u,
                              float *ind_l, float *scale)
{
    if (!ctx || !a || !tu || !tl || !T || !ind_u || !ind_l || !scale || n < 1)
        return MVS_ERR_INVALID_ARG;
    const size_t nb = (size_t)n * sizeof(float);
    const void *in[4] = {a, tu, tl, T};
    const size_t ib[4] = {nb, nb, nb, nb};
    void *out[3] = {ind_u, ind_l, scale};
    const size_t ob[3] = {nb, nb, nb};
    return probe_io(ctx, in, ib, 4, out, ob, 3, [&](void **d) {
        launch_indicator_probe((const float *)d[0], (const float *)d[1], (const float *)d[2], (const float *)d[3], n, (float *)d[4],
                               (float *)d[5], (float *)d[6], ctx->stream);
    });
}

// de-normalisation + fused residual of both paths on caller-supplied (Fn, transforms, point) tuples: in n x 19, out n x 2
int mvs_debug_rounding_probe(mvs_ctx *ctx, const double *in19, int n, double *out2)
{
    if (!ctx || !in19 || !out2 || n < 1)
        return MVS_ERR_INVALID_ARG;
    const void *in[1] = {in19};
    const size_t ib[1] = {(size_t)n * 19 * sizeof(double)};
    void *out[1] = {out2};
    const size_t ob[1] = {(size_t)n * 2 * sizeof(double)};
    return probe_io(ctx, in, ib, 1, out, ob, 1,
                    [&](void **d) { launch_rounding_probe((const double *)d[0], n, (double *)d[1], ctx->stream); });
}
#endif  // MVS_DEBUG_HOOKS

const char *mvs_status_str(mvs_status s)
{
    switch (s) {
    case MVS_OK: return "ok";
    case MVS_NO_MODEL: return "no model (reference: return false)";
    case MVS_ERR_INVALID_ARG: return "invalid argument (reference: assert)";
    case MVS_ERR_NO_DEVICE: return "no HIP device";
    case MVS_ERR_HIP: return "HIP runtime error";
    case MVS_ERR_CAPACITY: return "capacity exceeded";
    case MVS_ERR_BAD_INTRINSICS: return "camera intrinsics must be affine (last row 0 0 1)";
    }
    return "unknown";
}

const char *mvs_last_error(const mvs_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

mvs_status mvs_params_default(mvs_params *p)
{
    if (!p)
        return MVS_ERR_INVALID_ARG;
    p->ratio = 0.7;          // visual-feature.cpp:23
    p->max_dist = 10.0;      // image-pair.cpp:22-23
    p->max_error_sq = 0.0;   // derive from K (sfm-solve.cpp:311)
    p->num_hypotheses = 1;   // sfm-solve.cpp:67
    p->sampler = MVS_SAMPLER_IDENTITY;
    p->seed = 0;
    p->min_inliers = 8;      // sfm-solve.cpp:20-21
    p->reserved = 0;
    return MVS_OK;
}

mvs_status mvs_ctx_create_on_stream(int device_id, void *hip_stream, mvs_ctx **out)
{
    if (!out)
        return MVS_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return MVS_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *c = new mvs_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) {
        delete c;
        return MVS_ERR_NO_DEVICE;
    }
    if (hip_stream) {
        c->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return MVS_ERR_HIP;
        }
        c->own_stream = true;
    }
    {
        // the scoring, refinement and extraction kernels are laid out for the 160 KB of LDS a gfx950 compute unit has
        int lds = 0, cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0)
            c->cu_count = cus;
        const hipError_t e1 = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id);
        const hipError_t e2 = e1 == hipSuccess && lds >= 160 * 1024 ? prepare_kernels() : hipErrorInvalidDevice;
        if (e2 != hipSuccess) {
            std::fprintf(stderr, "mvs_ctx_create: device %d offers %d bytes of LDS per workgroup (%s); this library is built "
                                 "for gfx950 (160 KB)\n", device_id, lds, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            if (c->own_stream)
                (void)hipStreamDestroy(c->stream);
            delete c;
            return MVS_ERR_NO_DEVICE;
        }
    }
    if (hipMalloc((void **)&c->d_small, 64 * sizeof(double)) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        mvs_ctx_destroy(c);
        return MVS_ERR_HIP;
    }
    *out = c;
    return MVS_OK;
}

mvs_status mvs_ctx_create(int device_id, mvs_ctx **out) { return mvs_ctx_create_on_stream(device_id, nullptr, out); }

void mvs_ctx_destroy(mvs_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)sync_stream(ctx);
    if (ctx->scratch)
        mvs_batch_destroy(ctx->scratch);
    if (ctx->d_uv1) (void)hipFree(ctx->d_uv1);
    if (ctx->d_uv2) (void)hipFree(ctx->d_uv2);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->d_pnp) (void)hipFree(ctx->d_pnp);
    if (ctx->d_ref) (void)hipFree(ctx->d_ref);
    if (ctx->d_single) (void)hipFree(ctx->d_single);
    if (ctx->d_single_in) (void)hipFree(ctx->d_single_in);
    if (ctx->orb_graph) (void)hipGraphExecDestroy(ctx->orb_graph);
    if (ctx->d_orb) (void)hipFree(ctx->d_orb);
    if (ctx->h_orb_ovf) (void)hipHostFree(ctx->h_orb_ovf);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *mvs_ctx_stream(mvs_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

mvs_status mvs_ctx_set_half_batches(mvs_ctx *ctx, int enable)
{
    if (!ctx)
        return MVS_ERR_INVALID_ARG;
    ctx->half_batches = enable != 0;
    return MVS_OK;
}

// ---------------------------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------------------------
// n_frames == 0: every pair owns its two images.  n_frames == n_pairs + 1: the images are the frames of a sequence,
// stored once; pair k = (frame k, frame k + 1) is a view into the frame arrays (row f2).
static mvs_status batch_create_impl(mvs_ctx *ctx, int n_pairs, int max_kp, int desc_bytes, int n_frames, mvs_batch **out);

mvs_status mvs_batch_create(mvs_ctx *ctx, int n_pairs, int max_kp, int desc_bytes, mvs_batch **out)
{
    return batch_create_impl(ctx, n_pairs, max_kp, desc_bytes, 0, out);
}

static mvs_status batch_create_impl(mvs_ctx *ctx, int n_pairs, int max_kp, int desc_bytes, int n_frames, mvs_batch **out)
{
    if (!ctx || !out || n_pairs < 1 || max_kp < 1)
        return MVS_ERR_INVALID_ARG;
    if (max_kp > kMaxKp)
        return MVS_ERR_CAPACITY;
    if (!(desc_bytes == 16 || desc_bytes == 32 || desc_bytes == 64))
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mvs_batch *b = new mvs_batch();
    b->ctx = ctx;
    BatchDev &d = b->d;
    d.n_pairs = n_pairs;
    d.max_kp = max_kp;
    d.desc_words = desc_bytes / 4;
    d.max_groups = 0;
    d.cu_count = ctx->cu_count;
    const size_t P = n_pairs, N = max_kp;
    mvs_status st = MVS_OK;
    uint32_t *desc1, *desc2;
    float *kp1, *kp2;
    uint8_t *oct1, *oct2;
    int32_t *n1, *n2;
    double *Kinv, *K;
    int64_t *gidx;
#define ALLOC(ptr, cnt)                                \
    if (st == MVS_OK) st = dev_alloc(b, &(ptr), (cnt));
    const size_t NI = n_frames ? (size_t)n_frames : P;  // images held by desc1 / kp1 / n1
    ALLOC(desc1, NI * N * d.desc_words);
    ALLOC(kp1, NI * N * 2);
    ALLOC(oct1, NI * N);
    ALLOC(n1, NI);
    if (n_frames) {  // pair k's second image is frame k + 1
        desc2 = desc1 + N * d.desc_words;
        kp2 = kp1 + N * 2;
        oct2 = oct1 + N;
        n2 = n1 + 1;
    } else {
        ALLOC(desc2, P * N * d.desc_words);
        ALLOC(kp2, P * N * 2);
        ALLOC(oct2, P * N);
        ALLOC(n2, P);
    }
    ALLOC(Kinv, P * 9);
    ALLOC(K, P * 9);
    ALLOC(gidx, P);
    ALLOC(d.knn_train, P * N);
    ALLOC(d.knn_dist, P * N);
    ALLOC(d.M, P);
    ALLOC(d.matches, P * N);
    ALLOC(d.pts, P * N * 4);
    ALLOC(d.cand_pts, P * 4 * N * 3);
    ALLOC(d.fin, P);
    ALLOC(d.inl, P * N);
    ALLOC(d.okf, P * 4 * N);
    ALLOC(d.results, P);
    ALLOC(d.mask, P * N);
    ALLOC(d.points, P * N * 3);
    ALLOC(d.point_idx, P * N);
    ALLOC(d.stats, 16);
#undef ALLOC
    if (st != MVS_OK) {
        mvs_batch_destroy(b);
        return st;
    }
    d.desc1 = desc1; d.desc2 = desc2; d.kp1 = kp1; d.kp2 = kp2; d.n1 = n1; d.n2 = n2;
    d.oct1 = oct1; d.oct2 = oct2;
    d.Kinv = Kinv; d.K = K; d.gidx = gidx;
    d.wgbest = nullptr;
    d.hyp_F = nullptr;
    d.hyp_r32 = nullptr;
    d.hyp_okf = nullptr;
    d.hyp_cnt = nullptr;
    d.bound = nullptr;
    d.box = nullptr;
    d.mode = nullptr;
    d.dense_n1 = nullptr;
    d.clist = nullptr;
    d.clist2 = nullptr;
    d.cpos = nullptr;
    d.ccount = nullptr;
    d.pcount = nullptr;
    d.m0list = nullptr;
    d.xlist = nullptr;
    d.xcount = nullptr;
    d.hyp_count = nullptr;
    d.hyp_residual = nullptr;
    hipStream_t s = ctx->stream;
    // deterministic contents for rows the caller never uploads
    (void)hipMemsetAsync(desc1, 0, NI * N * d.desc_words * 4, s);
    (void)hipMemsetAsync(kp1, 0, NI * N * 2 * sizeof(float), s);
    (void)hipMemsetAsync(n1, 0, NI * sizeof(int32_t), s);
    (void)hipMemsetAsync(oct1, 0, NI * N, s);   // octave 0 (stddev = sigma_px) unless the extractor / caller says otherwise
    if (!n_frames) {
        (void)hipMemsetAsync(oct2, 0, P * N, s);
        (void)hipMemsetAsync(desc2, 0, P * N * d.desc_words * 4, s);
        (void)hipMemsetAsync(kp2, 0, P * N * 2 * sizeof(float), s);
        (void)hipMemsetAsync(n2, 0, P * sizeof(int32_t), s);
    }
    (void)hipMemsetAsync(gidx, 0, P * sizeof(int64_t), s);
    (void)hipMemsetAsync(d.M, 0, P * sizeof(int32_t), s);
    (void)hipMemsetAsync(d.results, 0, P * sizeof(mvs_pair_result), s);
    (void)hipMemsetAsync(d.mask, 0, P * N, s);
    for (auto &e : b->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            mvs_batch_destroy(b);
            return MVS_ERR_HIP;
        }
    if (sync_stream(ctx) != hipSuccess) {
        ctx->err = "mvs_batch_create: hipStreamSynchronize failed";
        mvs_batch_destroy(b);
        return MVS_ERR_HIP;
    }
    *out = b;
    return MVS_OK;
}

void mvs_batch_destroy(mvs_batch *b)
{
    if (!b)
        return;
    (void)hipSetDevice(b->ctx->device);
    (void)sync_stream(b->ctx);
    for (void *p : b->allocs)
        (void)hipFree(p);
    for (auto &e : b->ev)
        if (e)
            (void)hipEventDestroy(e);
    if (b->h_pin)
        (void)hipHostFree(b->h_pin);
    for (auto &e : b->pin_ev)
        if (e)
            (void)hipEventDestroy(e);
    delete b;
}

static mvs_status batch_upload_impl(mvs_batch *b, int first, int count, const uint8_t *base_desc, const float *base_kp,
                                    const int32_t *n_base, const uint8_t *pair_desc, const float *pair_kp,
                                    const int32_t *n_pair, const double *K, const int64_t *global_index, bool sync)
{
    if (!b || first < 0 || count < 1 || first + count > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const BatchDev &d = b->d;
    hipStream_t s = ctx->stream;
    const size_t N = d.max_kp, DW = d.desc_words;
    if (n_base)
        for (int i = 0; i < count; ++i)
            if (n_base[i] < 0 || n_base[i] > (int)N)
                return MVS_ERR_CAPACITY;
    if (n_pair)
        for (int i = 0; i < count; ++i)
            if (n_pair[i] < 0 || n_pair[i] > (int)N)
                return MVS_ERR_CAPACITY;
    // host-derived per-pair parameters live in pinned memory owned by the batch (slot = pair index), so that the
    // asynchronous form needs no temporaries that outlive the call
    const size_t set_bytes = (size_t)d.n_pairs * (9 * sizeof(double) + sizeof(int64_t));
    if (!b->h_pin) {
        HIP_TRY(ctx, hipHostMalloc((void **)&b->h_pin, 2 * set_bytes, hipHostMallocDefault));
        for (auto &e : b->pin_ev)
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (K)
        for (int i = 0; i < count; ++i)
            if (!affine_K(K + 9 * i))
                return MVS_ERR_BAD_INTRINSICS;
    const bool staged = K || !global_index;
    const int set = b->pin_turn;
    char *base = b->h_pin + (size_t)set * set_bytes;
    double *kinv = reinterpret_cast<double *>(base) + (size_t)first * 9;
    int64_t *gi = reinterpret_cast<int64_t *>(base + (size_t)d.n_pairs * 9 * sizeof(double)) + first;
    if (staged && b->pin_ev_live[set])
        HIP_TRY(ctx, hipEventSynchronize(b->pin_ev[set]));   // only the copies that last read this slot set
    if (K)
        for (int i = 0; i < count; ++i)
            mat3_inverse(K + 9 * i, kinv + 9 * i);
    if (!global_index) {
        for (int i = 0; i < count; ++i)
            gi[i] = first + i;
        global_index = gi;
    }
    const size_t off = first;
#define UP(dst, src, bytes_per_pair)                                                                                   \
    if (src)                                                                                                           \
    HIP_TRY(ctx, hipMemcpyAsync((char *)(dst) + off * (bytes_per_pair), (src), (size_t)count * (bytes_per_pair),       \
                                hipMemcpyHostToDevice, s))
    UP(const_cast<uint32_t *>(d.desc1), base_desc, N * DW * 4);
    UP(const_cast<uint32_t *>(d.desc2), pair_desc, N * DW * 4);
    UP(const_cast<float *>(d.kp1), base_kp, N * 2 * sizeof(float));
    UP(const_cast<float *>(d.kp2), pair_kp, N * 2 * sizeof(float));
    UP(const_cast<int32_t *>(d.n1), n_base, sizeof(int32_t));
    UP(const_cast<int32_t *>(d.n2), n_pair, sizeof(int32_t));
    UP(const_cast<double *>(d.K), K, 9 * sizeof(double));
    const double *kinv_p = K ? kinv : nullptr;
    UP(const_cast<double *>(d.Kinv), kinv_p, 9 * sizeof(double));
    UP(const_cast<int64_t *>(d.gidx), global_index, sizeof(int64_t));
#undef UP
    if (staged) {
        HIP_TRY(ctx, hipEventRecord(b->pin_ev[set], s));
        b->pin_ev_live[set] = true;
        b->pin_turn = set ^ 1;
    }
    if (sync)
        HIP_TRY(ctx, sync_stream(ctx));
    return MVS_OK;
}

mvs_status mvs_batch_upload(mvs_batch *b, int first, int count, const uint8_t *base_desc, const float *base_kp,
                            const int32_t *n_base, const uint8_t *pair_desc, const float *pair_kp,
                            const int32_t *n_pair, const double *K, const int64_t *global_index)
{
    return batch_upload_impl(b, first, count, base_desc, base_kp, n_base, pair_desc, pair_kp, n_pair, K, global_index, true);
}

mvs_status mvs_batch_upload_async(mvs_batch *b, int first, int count, const uint8_t *base_desc, const float *base_kp,
                                  const int32_t *n_base, const uint8_t *pair_desc, const float *pair_kp,
                                  const int32_t *n_pair, const double *K, const int64_t *global_index)
{
    return batch_upload_impl(b, first, count, base_desc, base_kp, n_base, pair_desc, pair_kp, n_pair, K, global_index, false);
}

mvs_status mvs_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0)
        return MVS_ERR_INVALID_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocPortable) == hipSuccess ? MVS_OK : MVS_ERR_HIP;
}

void mvs_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

static mvs_status check_params(const mvs_params *p)
{
    if (!p || p->num_hypotheses < 1 || (p->sampler != MVS_SAMPLER_IDENTITY && p->sampler != MVS_SAMPLER_PHILOX))
        return MVS_ERR_INVALID_ARG;  // estimator-RANSAC.cpp:13 assert(max_iteration > 0)
    return MVS_OK;
}

// Pairs [first, first + count) of a batch as a batch of their own: every per-pair array advanced to its first pair, the two
// per-launch scalars (work-list length, mode-0 pair list) given a slot per half.  The kernels index by the pair's position
// in the launch, so they need no change; flat record indices (xlist) are relative to the view's hyp_F.
static BatchDev batch_view(const BatchDev &b, int first, int count, int half)
{
    BatchDev v = b;
    const size_t f = (size_t)first, N = (size_t)b.max_kp, Hp = (size_t)b.max_groups * kHypPerBlock;
    v.n_pairs = count;
    v.desc1 += f * N * b.desc_words; v.desc2 += f * N * b.desc_words;
    v.kp1 += f * N * 2; v.kp2 += f * N * 2;
    v.oct1 += f * N; v.oct2 += f * N;
    v.n1 += f; v.n2 += f;
    v.Kinv += f * 9; v.K += f * 9; v.gidx += f;
    v.knn_train += f * N; v.knn_dist += f * N;
    v.M += f; v.matches += f * N; v.pts += f * N * 4;
    v.wgbest += f * b.max_groups;
    v.hyp_F += f * Hp * kHypRec; v.hyp_r32 += f * Hp * kHypRec32;
    v.hyp_okf += f * Hp; v.hyp_cnt += f * Hp;
    v.bound += f; v.box += f * 8; v.mode += f;
    v.clist += f * Hp; v.clist2 += f * Hp; v.cpos += f * kSortBins;
    v.ccount += f; v.pcount += f; v.dense_n1 += f;
    v.m0list += (size_t)half * ((size_t)b.n_pairs + 1);
    v.xlist += f * Hp;
    v.xcount += 2 * half;
    v.cand_pts += f * 4 * N * 3; v.fin += f; v.inl += f * N; v.okf += f * 4 * N;
    v.results += f; v.mask += f * N; v.points += f * N * 3; v.point_idx += f * N;
    return v;
}

constexpr int kHalvesMinPairs = 64;   // a batch of at least this many pairs runs as two halves on two streams

static void enqueue_stages(const BatchDev &d, const RunParams &rp, int n_active, bool stats, hipStream_t s, hipEvent_t *ev,
                           LaunchTimer *lt, bool with_match = true)
{
    if (ev) (void)hipEventRecord(ev[0], s);
    if (with_match)
        launch_match_topk(d, rp, n_active, s, lt);
    if (ev) (void)hipEventRecord(ev[1], s);
    if (with_match)
        launch_match_compact(d, rp, n_active, s, lt);
    if (ev) (void)hipEventRecord(ev[2], s);
    launch_ransac(d, rp, n_active, stats, s, lt);
    if (ev) (void)hipEventRecord(ev[3], s);
    launch_finalize(d, rp, n_active, kFinalizeFull, s, lt);
    if (ev) (void)hipEventRecord(ev[4], s);
}

static mvs_status enqueue_pipeline(mvs_batch *b, const RunParams &rp, int n_active, bool stats, hipEvent_t *ev,
                                   LaunchTimer *lt = nullptr, bool with_match = true)
{
    mvs_ctx *ctx = b->ctx;
    hipStream_t s = ctx->stream;
    const bool halves = n_active >= kHalvesMinPairs && !stats && !ev && !lt && !b->d.hyp_count && ctx->half_batches;
    if (!halves) {
        enqueue_stages(b->d, rp, n_active, stats, s, ev, lt, with_match);
    } else {
        // two independent halves: the second on the side stream, after everything already queued on the main stream, and
        // joined back into it (the caller's next operation on the main stream sees both).  Measured on the bench batch:
        // 2 parts +2.2 % pairs/s, +10 % frames/s on the sequence; 3 and 4 parts, a 40:60 split and a lower or higher
        // priority of the side stream are all within noise of or below two equal halves.  The kernel trace shows the heavy
        // kernels of the two halves serialising or sharing the chip at the same total rate and the two latency-bound tails
        // (list sort ... selection ... triangulation, ~0.8 ms per half whatever its size) largely coinciding; holding the
        // second half's counting back with an event until the first half's dense phase has drained puts it under the first
        // half's tail but leaves the second tail exposed: same step time within 1 % (measured, not kept).
        const int na = (n_active + 1) / 2;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));          // nothing is on the side stream yet: a plain return is safe
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        enqueue_stages(batch_view(b->d, 0, na, 0), rp, na, false, s, nullptr, nullptr, with_match);
        const hipError_t e_first = hipGetLastError();            // a launch failure of the first half, attributed to it
        enqueue_stages(batch_view(b->d, na, n_active - na, 1), rp, n_active - na, false, ctx->side, nullptr, nullptr, with_match);
        const hipError_t e_second = hipGetLastError();
        // from here on work may be queued on the side stream: whatever fails, the caller's stream joins it before this call
        // returns (a later download or a release of the groups must never race with the second half) -- ADVICE r4
        hipError_t e_join = hipEventRecord(ctx->ev_join, ctx->side);
        if (e_join == hipSuccess)
            e_join = hipStreamWaitEvent(s, ctx->ev_join, 0);
        if (e_join != hipSuccess)
            (void)hipStreamSynchronize(ctx->side);               // no event edge: wait for the half on the host instead
        HIP_TRY(ctx, e_first);
        HIP_TRY(ctx, e_second);
        HIP_TRY(ctx, e_join);
    }
    if (lt) lt->end();
    HIP_TRY(ctx, hipGetLastError());
    return MVS_OK;
}

mvs_status mvs_batch_run(mvs_batch *b, const mvs_params *params, int n_active)
{
    if (!b || n_active < 1 || n_active > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
    st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    return enqueue_pipeline(b, to_run(*params), n_active, false, nullptr);
}

// A batch of sfm_solve calls (vision/sfm.hpp:30-35, sfm-solve.cpp:285-368) on caller-supplied point pairs: the matcher is
// skipped, everything behind it is mvs_batch_run's (normalise -> RANSAC stage -> decomposition -> triangulation, half batches
// on two streams included).  uv1 / uv2: HOST, [n_active][max_kp][2] doubles (image points of the base / pair frame, row k of
// pair p = match k); m[p]: matches of pair p (0 .. max_kp).  The intrinsics and the sampler's key offsets are the resident
// ones (mvs_batch_upload with null descriptor / keypoint pointers sets just K and global_index).  `matches` rows of the
// batch are cleared (there is no match list: point k IS match k); results / mask / points / point_idx as after mvs_batch_run.
mvs_status mvs_batch_run_points(mvs_batch *b, const mvs_params *params, int n_active, const double *uv1, const double *uv2,
                                const int32_t *m)
{
    if (!b || !uv1 || !uv2 || !m || n_active < 1 || n_active > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    const size_t N = (size_t)b->d.max_kp;
    for (int p = 0; p < n_active; ++p)
        if (m[p] < 0 || m[p] > (int)N)
            return MVS_ERR_CAPACITY;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    if (!b->uv1) {   // staging for the image points, allocated on first use and owned by the batch
        const size_t bytes = (size_t)b->d.n_pairs * N * 2 * sizeof(double);
        void *a = nullptr, *c = nullptr;
        HIP_TRY(ctx, hipMalloc(&a, bytes));
        b->allocs.push_back(a);
        HIP_TRY(ctx, hipMalloc(&c, bytes));
        b->allocs.push_back(c);
        b->uv1 = static_cast<double *>(a);
        b->uv2 = static_cast<double *>(c);
    }
    hipStream_t s = ctx->stream;
    const size_t pb = (size_t)n_active * N * 2 * sizeof(double);
    HIP_TRY(ctx, hipMemcpyAsync(b->uv1, uv1, pb, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(b->uv2, uv2, pb, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(b->d.M, m, (size_t)n_active * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(b->d.matches, 0, (size_t)n_active * N * sizeof(mvs_match), s));
    // the host buffers are the caller's: they may change as soon as this call returns (pageable copies are staged by the
    // runtime before hipMemcpyAsync returns; pinned ones are not) -- wait for the three copies, not for the kernels
    HIP_TRY(ctx, hipStreamSynchronize(s));
    launch_prep_points(b->d, b->uv1, b->uv2, n_active, s);
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    return enqueue_pipeline(b, to_run(*params), n_active, false, nullptr, nullptr, false);
}

mvs_status mvs_batch_sync(mvs_batch *b)
{
    if (!b)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(b->ctx, sync_stream(b->ctx));
    return MVS_OK;
}

mvs_status mvs_batch_time(mvs_batch *b, const mvs_params *params, int n_active, int warmup, int steps,
                          float *ms_total, float *ms_kernel)
{
    if (!b || n_active < 1 || n_active > b->d.n_pairs || steps < 1 || warmup < 0)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    const RunParams rp = to_run(*params);
    hipStream_t s = ctx->stream;
    for (int i = 0; i < warmup; ++i)
        if ((st = enqueue_pipeline(b, rp, n_active, false, nullptr)) != MVS_OK)
            return st;
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipEventRecord(b->ev[5], s));
    for (int i = 0; i < steps; ++i)
        if ((st = enqueue_pipeline(b, rp, n_active, false, nullptr)) != MVS_OK)
            return st;
    HIP_TRY(ctx, hipEventRecord(b->ev[6], s));
    HIP_TRY(ctx, hipEventSynchronize(b->ev[6]));
    if (ms_total)
        HIP_TRY(ctx, hipEventElapsedTime(ms_total, b->ev[5], b->ev[6]));
    if (ms_kernel) {
        for (int k = 0; k < 5; ++k)
            ms_kernel[k] = 0.f;
        for (int i = 0; i < steps; ++i) {
            if ((st = enqueue_pipeline(b, rp, n_active, false, b->ev)) != MVS_OK)
                return st;
            HIP_TRY(ctx, hipEventSynchronize(b->ev[4]));
            for (int k = 0; k < 4; ++k) {
                float ms = 0.f;
                HIP_TRY(ctx, hipEventElapsedTime(&ms, b->ev[k], b->ev[k + 1]));
                ms_kernel[k] += ms;
            }
        }
    }
    return MVS_OK;
}

mvs_status mvs_batch_time_kernels(mvs_batch *b, const mvs_params *params, int n_active, int steps, int cap,
                                  int32_t *kernel_id, float *ms, int *n_launches)
{
    if (!b || !kernel_id || !ms || !n_launches || n_active < 1 || n_active > b->d.n_pairs || steps < 1 || cap < 1)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    const RunParams rp = to_run(*params);
    constexpr int kMaxLaunches = 32;
    cap = std::min(cap, kMaxLaunches);
    hipEvent_t ev[kMaxLaunches + 1] = {};
    int32_t kid[kMaxLaunches];
    for (int k = 0; k <= cap; ++k)
        if (hipEventCreate(&ev[k]) != hipSuccess) {
            for (int j = 0; j < k; ++j)
                (void)hipEventDestroy(ev[j]);
            ctx->err = "mvs_batch_time_kernels: hipEventCreate failed";
            return MVS_ERR_HIP;
        }
    int n = 0;
    st = MVS_OK;
    for (int k = 0; k < cap; ++k)
        ms[k] = 0.f;
    for (int i = 0; i < steps && st == MVS_OK; ++i) {
        LaunchTimer lt{ctx->stream, ev, kid, cap, 0};
        st = enqueue_pipeline(b, rp, n_active, false, nullptr, &lt);
        if (st != MVS_OK)
            break;
        if (hipEventSynchronize(ev[lt.n]) != hipSuccess) {
            ctx->err = "mvs_batch_time_kernels: hipEventSynchronize failed";
            st = MVS_ERR_HIP;
            break;
        }
        n = lt.n;
        for (int k = 0; k < n; ++k) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, ev[k], ev[k + 1]) != hipSuccess) {
                ctx->err = "mvs_batch_time_kernels: hipEventElapsedTime failed";
                st = MVS_ERR_HIP;
                break;
            }
            ms[k] += t;
            kernel_id[k] = kid[k];
        }
    }
    for (int k = 0; k <= cap; ++k)
        (void)hipEventDestroy(ev[k]);
    if (st != MVS_OK)
        return st;
    for (int k = 0; k < n; ++k)
        ms[k] /= (float)steps;
    *n_launches = n;
    return MVS_OK;
}

mvs_status mvs_kernel_info_get(mvs_ctx *ctx, int index, int max_kp, int desc_bytes, mvs_kernel_info *out)
{
    if (!ctx || !out || index < 0 || max_kp < 1 || max_kp > kMaxKp)
        return MVS_ERR_INVALID_ARG;
    KernelDesc kd;
    if (!kernel_desc(index, max_kp, desc_bytes / 4, &kd))
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::memset(out, 0, sizeof(*out));
    std::snprintf(out->name, sizeof(out->name), "%s", kd.name);
    const char *sym = hipKernelNameRefByPtr(kd.fn, ctx->stream);
    std::snprintf(out->symbol, sizeof(out->symbol), "%s", sym ? sym : "");
    hipFuncAttributes fa;
    HIP_TRY(ctx, hipFuncGetAttributes(&fa, kd.fn));
    out->kernel_id = index;
    out->threads_per_block = kd.threads;
    out->num_regs = fa.numRegs;
    out->static_lds_bytes = (int32_t)fa.sharedSizeBytes;
    out->dynamic_lds_bytes = (int32_t)kd.dynamic_lds;
    out->scratch_bytes_per_lane = (int32_t)fa.localSizeBytes;
    out->max_threads_per_block = fa.maxThreadsPerBlock;
    int nb = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kd.fn, kd.threads, kd.dynamic_lds));
    out->blocks_per_cu = nb;
    const int waves = nb * ((kd.threads + 63) / 64);
    out->waves_per_simd = waves ? std::max(1, waves / 4) : 0;
    return MVS_OK;
}

// enqueue the device -> host copies of a batch's outputs on the ctx stream (after whatever has been enqueued: a
// preceding mvs_batch_run needs no synchronisation in between)
static mvs_status batch_download_enqueue(mvs_batch *b, int first, int count, mvs_pair_result *results, mvs_match *matches,
                                         uint8_t *inlier_mask, double *points_xyz, int32_t *point_idx32)
{
    mvs_ctx *ctx = b->ctx;
    hipStream_t s = ctx->stream;
    const BatchDev &d = b->d;
    const size_t N = d.max_kp, off = first, cnt = count;
    if (results)
        HIP_TRY(ctx, hipMemcpyAsync(results, d.results + off, cnt * sizeof(mvs_pair_result), hipMemcpyDeviceToHost, s));
    if (matches)
        HIP_TRY(ctx, hipMemcpyAsync(matches, d.matches + off * N, cnt * N * sizeof(mvs_match), hipMemcpyDeviceToHost, s));
    if (inlier_mask)
        HIP_TRY(ctx, hipMemcpyAsync(inlier_mask, d.mask + off * N, cnt * N, hipMemcpyDeviceToHost, s));
    if (points_xyz)
        HIP_TRY(ctx, hipMemcpyAsync(points_xyz, d.points + off * N * 3, cnt * N * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (point_idx32)
        HIP_TRY(ctx, hipMemcpyAsync(point_idx32, d.point_idx + off * N, cnt * N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return MVS_OK;
}

mvs_status mvs_batch_download(mvs_batch *b, int first, int count, mvs_pair_result *results, mvs_match *matches,
                              uint8_t *inlier_mask, double *points_xyz, int64_t *point_idx)
{
    if (!b || first < 0 || count < 1 || first + count > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t N = b->d.max_kp, cnt = count;
    std::vector<int32_t> tmp(point_idx ? cnt * N : 0);
    mvs_status st = batch_download_enqueue(b, first, count, results, matches, inlier_mask, points_xyz,
                                           point_idx ? tmp.data() : nullptr);
    if (st != MVS_OK)
        return st;
    HIP_TRY(ctx, sync_stream(ctx));
    for (size_t i = 0; i < tmp.size(); ++i)
        point_idx[i] = tmp[i];  // reference type: size_t (sfm.hpp:35)
    return MVS_OK;
}

mvs_status mvs_batch_download_async(mvs_batch *b, int first, int count, mvs_pair_result *results, mvs_match *matches,
                                    uint8_t *inlier_mask, double *points_xyz, int32_t *point_idx32)
{
    if (!b || first < 0 || count < 1 || first + count > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
    return batch_download_enqueue(b, first, count, results, matches, inlier_mask, points_xyz, point_idx32);
}

// The one exchange step of the path: all-gather of the fixed-size result records over RCCL, callable from
// host C++.  librccl is resolved at the first call (dlopen), so the library itself carries no link-time dependency on it
// and single-GPU callers never load it.  `rccl_comm` is the caller's ncclComm_t (one rank per GPU, created by the caller:
// ncclCommInitRank with an id it distributes however it likes); the collective is enqueued on the ctx stream, i.e. after
// the kernels of a preceding mvs_batch_run, and the call returns without waiting.
mvs_status mvs_batch_gather_results(mvs_batch *b, int n_active, void *rccl_comm, void *dst_device)
{
    if (!b || !rccl_comm || !dst_device || n_active < 1 || n_active > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    typedef int (*allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
    typedef const char *(*errstr_fn)(int);
    static allgather_fn fn = nullptr;
    static errstr_fn es = nullptr;
    if (!fn) {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h)
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            ctx->err = std::string("dlopen(librccl.so): ") + dlerror();
            return MVS_ERR_HIP;
        }
        fn = reinterpret_cast<allgather_fn>(dlsym(h, "ncclAllGather"));
        es = reinterpret_cast<errstr_fn>(dlsym(h, "ncclGetErrorString"));
        if (!fn) {
            ctx->err = "librccl.so has no ncclAllGather";
            return MVS_ERR_HIP;
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = fn(b->d.results, dst_device, (size_t)n_active * sizeof(mvs_pair_result), /* ncclUint8 */ 1, rccl_comm,
                      ctx->stream);
    if (rc != 0) {
        ctx->err = std::string("ncclAllGather: ") + (es ? es(rc) : "error");
        return MVS_ERR_HIP;
    }
    return MVS_OK;
}

mvs_status mvs_batch_stats(mvs_batch *b, const mvs_params *params, int n_active, mvs_work_stats *out)
{
    if (!b || !out || n_active < 1 || n_active > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(b->d.stats, 0, 16 * sizeof(unsigned long long), s));
    st = enqueue_pipeline(b, to_run(*params), n_active, true, nullptr);
    if (st != MVS_OK)
        return st;
    HIP_TRY(ctx, sync_stream(ctx));
    unsigned long long h[16];
    HIP_TRY(ctx, hipMemcpy(h, b->d.stats, sizeof(h), hipMemcpyDeviceToHost));
    std::vector<mvs_pair_result> res(n_active);
    HIP_TRY(ctx, hipMemcpy(res.data(), b->d.results, n_active * sizeof(mvs_pair_result), hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof(*out));
    out->rotations9 = (int64_t)h[0];
    out->pairs9 = (int64_t)h[1];
    out->score_evals_executed = (int64_t)(h[2] + h[3] + h[4] + h[5] + h[7] + h[8]);   // double-, single-precision and matrix-core counting
    out->score_evals_executed_mfma_pilot = (int64_t)h[8];
    out->score_evals_executed_mfma_rest = (int64_t)h[7];
    out->score_evals_executed_f32 = (int64_t)h[3];
    out->score_evals_executed_mfma = (int64_t)h[4];
    out->score_evals_executed_mfma_finish = (int64_t)h[5];
    out->max_sweeps9 = (int64_t)h[6];
    if (b->d.mode && b->d.xcount && h[2] + h[3] + h[4] + h[5] + h[8] > 0) {   // the pre-screened stage ran: its bookkeeping
        std::vector<int32_t> mode(n_active);
        uint32_t xc[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpy(mode.data(), b->d.mode, n_active * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(xc, b->d.xcount, sizeof(xc), hipMemcpyDeviceToHost));
        std::vector<int32_t> n1(n_active);
        HIP_TRY(ctx, hipMemcpy(n1.data(), b->d.dense_n1, n_active * sizeof(int32_t), hipMemcpyDeviceToHost));
        int64_t legacy = 0, screened = 0;
        for (int i = 0; i < n_active; ++i) {
            if (res[i].n_matches < 8)
                continue;
            const int m = mode[i] < 0 || mode[i] > 2 ? 0 : mode[i];
            out->pairs_mode[m] += 1;
            (m == 0 ? legacy : screened) += params->num_hypotheses;
            if (m == 1) {
                out->dense_points += n1[i];
                out->matches_mode1 += res[i].n_matches;
            }
        }
        out->exact_solves = legacy + (int64_t)xc[0];
        out->prescreened = screened - (int64_t)xc[0];
    }
    for (const auto &r : res) {
        out->matches += r.n_matches;
        out->inliers += r.n_inliers;
        if (r.n_matches >= 8) {
            out->hypotheses += params->num_hypotheses;
            out->score_evals += (int64_t)params->num_hypotheses * r.n_matches;
        }
    }
    return MVS_OK;
}


// Read-only view of the batch's device-resident state for diagnostics (round 5): the bytes of the internal
// BatchDev (device pointers + capacities, mvslam_amd/csrc/kernels.hpp), so that the audit in libmvslam_hip_dbg.so -- built from
// the same sources -- can check what THIS library's kernels wrote, in place, in the same process.  Nothing is launched,
// copied on the device or changed; the caller synchronises the batch first.  The layout is private: the blob starts with
// its own size and the ABI version, which the consumer compares with its own.
mvs_status mvs_batch_device_state(mvs_batch *b, void *dst, size_t capacity, size_t *size)
{
    struct Blob {
        uint64_t bytes, abi;
        BatchDev d;
    };
    static_assert(std::is_trivially_copyable<BatchDev>::value, "BatchDev is a POD of device pointers and sizes");
    if (!b || !size)
        return MVS_ERR_INVALID_ARG;
    *size = sizeof(Blob);
    if (!dst)
        return MVS_OK;   // size query
    if (capacity < sizeof(Blob))
        return MVS_ERR_CAPACITY;
    Blob blob{sizeof(Blob), (uint64_t)MVS_ABI_VERSION, b->d};
    blob.d.hyp_count = nullptr;      // the optional per-hypothesis tables belong to the table entry points
    blob.d.hyp_residual = nullptr;
    std::memcpy(dst, &blob, sizeof(Blob));
    return MVS_OK;
}

mvs_status mvs_batch_results_device(mvs_batch *b, void **dev_ptr, size_t *record_bytes)
{
    if (!b || !dev_ptr)
        return MVS_ERR_INVALID_ARG;
    *dev_ptr = b->d.results;
    if (record_bytes)
        *record_bytes = sizeof(mvs_pair_result);
    return MVS_OK;
}

mvs_status mvs_batch_copy_results_device(mvs_batch *b, int first, int count, void *dst_device)
{
    if (!b || !dst_device || first < 0 || count < 1 || first + count > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(b->ctx, hipSetDevice(b->ctx->device));
    HIP_TRY(b->ctx, hipMemcpyAsync(dst_device, b->d.results + first, (size_t)count * sizeof(mvs_pair_result),
                                   hipMemcpyDeviceToDevice, b->ctx->stream));
    return MVS_OK;
}

// ---------------------------------------------------------------------------------------------
// single-shot entry points: a resident batch of one pair owned by the context
// ---------------------------------------------------------------------------------------------
static mvs_status ensure_scratch(mvs_ctx *ctx, int max_kp, int desc_bytes)
{
    if (max_kp > kMaxKp)
        return MVS_ERR_CAPACITY;
    if (ctx->scratch && ctx->scratch->d.max_kp >= max_kp && ctx->scratch->d.desc_words * 4 == desc_bytes)
        return MVS_OK;
    if (ctx->scratch) {
        max_kp = std::max(max_kp, ctx->scratch->d.max_kp);
        mvs_batch_destroy(ctx->scratch);
        ctx->scratch = nullptr;
    }
    max_kp = std::max(max_kp, 64);
    return mvs_batch_create(ctx, 1, max_kp, desc_bytes, &ctx->scratch);
}

static mvs_status ensure_uv(mvs_ctx *ctx, int cap)
{
    if (ctx->uv_cap >= cap)
        return MVS_OK;
    if (ctx->d_uv1) (void)hipFree(ctx->d_uv1);
    if (ctx->d_uv2) (void)hipFree(ctx->d_uv2);
    ctx->d_uv1 = ctx->d_uv2 = nullptr;
    ctx->uv_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_uv1, (size_t)cap * 2 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_uv2, (size_t)cap * 2 * sizeof(double)));
    ctx->uv_cap = cap;
    return MVS_OK;
}

// stage image points + camera of a single-shot call; leaves normalised points in batch->pts
static mvs_status stage_points(mvs_ctx *ctx, const double *p1_uv, const double *p2_uv, int m, const double K[9])
{
    if (!affine_K(K))
        return MVS_ERR_BAD_INTRINSICS;
    const int desc_bytes = ctx->scratch ? ctx->scratch->d.desc_words * 4 : 32;
    mvs_status st = ensure_scratch(ctx, std::max(m, 8), desc_bytes);
    if (st != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    st = ensure_uv(ctx, b->d.max_kp);
    if (st != MVS_OK)
        return st;
    hipStream_t s = ctx->stream;
    double kinv[9];
    mat3_inverse(K, kinv);
    const int64_t zero = 0;
    const int32_t M = m;
    const size_t pb = (size_t)m * 2 * sizeof(double);
    // results come back through the same arena (fetch_single): size it for both directions now
    if ((st = pin_begin(ctx, 2 * pb + 1024 + sizeof(mvs_pair_result) + (size_t)b->d.max_kp * (1 + 24 + 4 + 16) + 512)) != MVS_OK)
        return st;
    if ((st = up_async(ctx, ctx->d_uv1, p1_uv, pb)) != MVS_OK) return st;
    if ((st = up_async(ctx, ctx->d_uv2, p2_uv, pb)) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<double *>(b->d.K), K, 9 * sizeof(double))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<double *>(b->d.Kinv), kinv, 9 * sizeof(double))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<int64_t *>(b->d.gidx), &zero, sizeof(zero))) != MVS_OK) return st;
    if ((st = up_async(ctx, b->d.M, &M, sizeof(M))) != MVS_OK) return st;
    launch_prep_points(b->d, ctx->d_uv1, ctx->d_uv2, 1, s);
    return MVS_OK;
}

// results of a single-shot call: asynchronous copies into the pinned arena (opened by the staging half of the call),
// ONE synchronisation, then plain memcpy into the caller's buffers.  The row counts are only known after the copy, so
// the m-row capacity is fetched (m <= 4096: at most 180 KB).
static mvs_status fetch_single(mvs_ctx *ctx, int m, mvs_pair_result *res, double *points_xyz, int64_t *point_idx,
                               uint8_t *mask, mvs_match *matches = nullptr)
{
    mvs_batch *b = ctx->scratch;
    hipStream_t s = ctx->stream;
    const size_t rows = (size_t)std::max(m, 0);
    // pair 0's outputs gathered on the device into one block, ONE copy to the pinned arena
    const int flags = ((mask && rows) ? 1 : 0) | ((points_xyz && rows) ? 2 : 0) | ((point_idx && rows) ? 4 : 0) |
                      ((matches && rows) ? 8 : 0);
    const SingleLayout L = single_layout((int)rows, flags);
    if (ctx->single_cap < L.total) {
        HIP_TRY(ctx, sync_stream(ctx));
        if (ctx->d_single)
            (void)hipFree(ctx->d_single);
        ctx->d_single = nullptr;
        ctx->single_cap = 0;
        const size_t cap = std::max<size_t>(L.total, single_layout(b->d.max_kp, 15).total);
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_single, cap));
        ctx->single_cap = cap;
    }
    unsigned char *h_all = static_cast<unsigned char *>(pin_get(ctx, L.total));
    PIN_TRY(ctx, h_all);
    launch_single_gather(b->d, (int)rows, flags, ctx->d_single, s);
    HIP_TRY(ctx, hipMemcpyAsync(h_all, ctx->d_single, L.total, hipMemcpyDeviceToHost, s));
    const mvs_pair_result *h_res = reinterpret_cast<const mvs_pair_result *>(h_all);
    const uint8_t *h_mask = (flags & 1) ? h_all + L.mask : nullptr;
    const double *h_pts = (flags & 2) ? reinterpret_cast<const double *>(h_all + L.points) : nullptr;
    const int32_t *h_idx = (flags & 4) ? reinterpret_cast<const int32_t *>(h_all + L.idx) : nullptr;
    const mvs_match *h_mt = (flags & 8) ? reinterpret_cast<const mvs_match *>(h_all + L.matches) : nullptr;
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipGetLastError());
    *res = *h_res;
    const size_t M = (size_t)std::min<int>(std::max(res->n_matches, 0), (int)rows);
    if (h_mask)
        std::memcpy(mask, h_mask, matches ? M : rows);
    const int n = res->valid ? res->n_points : 0;
    if (n > 0 && h_pts)
        std::memcpy(points_xyz, h_pts, (size_t)n * 3 * sizeof(double));
    if (n > 0 && h_idx)
        for (int i = 0; i < n; ++i)
            point_idx[i] = h_idx[i];   // reference type: size_t (sfm.hpp:35)
    if (h_mt && M)
        std::memcpy(matches, h_mt, M * sizeof(mvs_match));
    return MVS_OK;
}

mvs_status mvs_match_hamming(mvs_ctx *ctx, const uint8_t *train_desc, int n_train, const uint8_t *query_desc,
                             int n_query, int desc_bytes, double ratio, double max_dist, mvs_match *out, int *n_out)
{
    if (!ctx || !train_desc || !query_desc || !out || !n_out)
        return MVS_ERR_INVALID_ARG;
    *n_out = 0;
    if (n_train < 2 || n_query < 1)  // visual-feature.cpp:56 assert(valid), :67 needs two neighbours
        return MVS_ERR_INVALID_ARG;
    if (!(desc_bytes == 16 || desc_bytes == 32 || desc_bytes == 64))
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mvs_status st = ensure_scratch(ctx, std::max(n_train, n_query), desc_bytes);
    if (st != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    hipStream_t s = ctx->stream;
    const int32_t n1 = n_train, n2 = n_query;
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const size_t cap = (size_t)std::min(n_train, n_query);   // a query matches at most once
    (void)cap;
    const size_t tb = (size_t)n_train * desc_bytes, qb = (size_t)n_query * desc_bytes;
    if ((st = pin_begin(ctx, tb + qb + 1024 + sizeof(int32_t) + ((size_t)n_query) * sizeof(mvs_match))) != MVS_OK)
        return st;
    // descriptors (<= 4096 x 64 B per image) and the small parameters all go through the pinned arena: every device copy
    // is an asynchronous DMA, whatever memory the caller's cv::Mat lives in
    if ((st = up_async(ctx, const_cast<uint32_t *>(b->d.desc1), train_desc, tb)) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<uint32_t *>(b->d.desc2), query_desc, qb)) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<int32_t *>(b->d.n1), &n1, sizeof(n1))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<int32_t *>(b->d.n2), &n2, sizeof(n2))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<double *>(b->d.Kinv), eye, sizeof(eye))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<double *>(b->d.K), eye, sizeof(eye))) != MVS_OK) return st;
    RunParams rp{};
    rp.ratio = ratio;
    rp.max_dist = max_dist;
    launch_match_topk(b->d, rp, 1, s);
    launch_match_compact(b->d, rp, 1, s);
    int32_t *h_M = static_cast<int32_t *>(pin_get(ctx, sizeof(int32_t)));
    mvs_match *h_mt = static_cast<mvs_match *>(pin_get(ctx, (size_t)n_query * sizeof(mvs_match)));
    PIN_TRY(ctx, h_M);
    PIN_TRY(ctx, h_mt);
    HIP_TRY(ctx, hipMemcpyAsync(h_M, b->d.M, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(h_mt, b->d.matches, (size_t)n_query * sizeof(mvs_match), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipGetLastError());
    const int32_t M = *h_M;
    if (M > 0)
        std::memcpy(out, h_mt, (size_t)M * sizeof(mvs_match));
    *n_out = M;
    return MVS_OK;
}

mvs_status mvs_two_view(mvs_ctx *ctx, const double *p1_uv, const double *p2_uv, int m, const double K[9],
                        const mvs_params *params, double R[9], double t[3], double *points_xyz,
                        int64_t *point_idx, int *n_points, uint8_t *inlier_mask, mvs_pair_result *result)
{
    if (!ctx || !p1_uv || !p2_uv || !K || m < 0)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    if (n_points)
        *n_points = 0;
    mvs_pair_result res;
    std::memset(&res, 0, sizeof(res));
    res.best_hyp = -1;
    res.n_matches = m;
    if (m < 8) {  // sfm-solve.cpp:37 asserts; estimator-RANSAC.cpp:25-29 returns false
        if (result)
            *result = res;
        return m < 1 ? MVS_ERR_INVALID_ARG : MVS_NO_MODEL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    st = stage_points(ctx, p1_uv, p2_uv, m, K);
    if (st != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    st = ensure_groups(b, params->num_hypotheses);
    if (st != MVS_OK)
        return st;
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    const RunParams rp = to_run(*params);
    launch_ransac(b->d, rp, 1, false, ctx->stream);
    launch_finalize(b->d, rp, 1, kFinalizeFull, ctx->stream);
    st = fetch_single(ctx, m, &res, points_xyz, point_idx, inlier_mask);
    if (st != MVS_OK)
        return st;
    if (result)
        *result = res;
    if (!res.valid)
        return MVS_NO_MODEL;
    if (R) std::memcpy(R, res.R, sizeof(res.R));
    if (t) std::memcpy(t, res.t, sizeof(res.t));
    if (n_points) *n_points = res.n_points;
    return MVS_OK;
}

// ImagePair::ImagePair + reconstruct (front-end/image-pair.cpp:30-71,116-174) of ONE pair in one device pass: the
// descriptors and keypoints go up once, the four stages run back to back on the stream, everything comes back through the
// pinned arena after a single synchronisation (match -> host gather -> sfm_solve as separate calls costs two round trips).
mvs_status mvs_image_pair(mvs_ctx *ctx, const uint8_t *base_desc, const float *base_kp, int n_base,
                          const uint8_t *pair_desc, const float *pair_kp, int n_pair, int desc_bytes, const double K[9],
                          const mvs_params *params, mvs_pair_result *result, mvs_match *matches, uint8_t *inlier_mask,
                          double *points_xyz, int64_t *point_idx)
{
    if (!ctx || !base_desc || !base_kp || !pair_desc || !pair_kp || !K || !result)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(params);
    if (st != MVS_OK)
        return st;
    if (n_base < 2 || n_pair < 1)   // visual-feature.cpp:56,67
        return MVS_ERR_INVALID_ARG;
    if (!(desc_bytes == 16 || desc_bytes == 32 || desc_bytes == 64))
        return MVS_ERR_INVALID_ARG;
    if (!affine_K(K))
        return MVS_ERR_BAD_INTRINSICS;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((st = ensure_scratch(ctx, std::max(n_base, n_pair), desc_bytes)) != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    if ((st = ensure_groups(b, params->num_hypotheses)) != MVS_OK)
        return st;
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    const size_t db1 = (size_t)n_base * desc_bytes, db2 = (size_t)n_pair * desc_bytes;
    const size_t kb1 = (size_t)n_base * 2 * sizeof(float), kb2 = (size_t)n_pair * 2 * sizeof(float);
    if ((st = pin_begin(ctx, db1 + db2 + kb1 + kb2 + 2048 + single_layout(n_pair, 15).total + 512)) != MVS_OK)
        return st;
    double kinv[9];
    mat3_inverse(K, kinv);
    const int32_t n1 = n_base, n2 = n_pair;
    const int64_t zero = 0;
    // <= 4096 x 64 B per image: through the pinned arena (a memcpy of ~160 KB), so that the device copy is true asynchronous
    // DMA whatever memory the caller's cv::Mat / std::vector lives in -- as ONE block [desc1 | desc2 | kp1 | kp2] and one copy
    // command; the kernel that receives the pair's scalars as arguments also puts the parts in place
    {
        auto up16 = [](size_t x) { return (x + 15) & ~size_t(15); };
        const size_t total = up16(db1) + up16(db2) + up16(kb1) + up16(kb2);
        if (ctx->single_in_cap < total) {
            HIP_TRY(ctx, sync_stream(ctx));
            if (ctx->d_single_in)
                (void)hipFree(ctx->d_single_in);
            ctx->d_single_in = nullptr;
            ctx->single_in_cap = 0;
            const size_t cap = std::max<size_t>(total, (size_t)b->d.max_kp * (2 * 64 + 2 * 8) + 64);
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_single_in, cap));
            ctx->single_in_cap = cap;
        }
        unsigned char *h_in = static_cast<unsigned char *>(pin_get(ctx, total));
        PIN_TRY(ctx, h_in);
        size_t o = 0;
        std::memcpy(h_in + o, base_desc, db1); o += up16(db1);
        std::memcpy(h_in + o, pair_desc, db2); o += up16(db2);
        std::memcpy(h_in + o, base_kp, kb1); o += up16(kb1);
        std::memcpy(h_in + o, pair_kp, kb2);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_single_in, h_in, total, hipMemcpyHostToDevice, ctx->stream));
        SingleParams sp;
        sp.n1 = n1;
        sp.n2 = n2;
        sp.gidx = zero;
        for (int k = 0; k < 9; ++k) {
            sp.K[k] = K[k];
            sp.Kinv[k] = kinv[k];
        }
        sp.part_bytes[0] = (uint32_t)db1; sp.part_bytes[1] = (uint32_t)db2;
        sp.part_bytes[2] = (uint32_t)kb1; sp.part_bytes[3] = (uint32_t)kb2;
        launch_single_params(b->d, sp, ctx->d_single_in, ctx->stream);
    }
    if ((st = enqueue_pipeline(b, to_run(*params), 1, false, nullptr)) != MVS_OK)
        return st;
    if ((st = fetch_single(ctx, n_pair, result, points_xyz, point_idx, inlier_mask, matches)) != MVS_OK)
        return st;
    return result->valid ? MVS_OK : MVS_NO_MODEL;
}

static mvs_status upload_mask(mvs_ctx *ctx, const uint8_t *mask, int m)
{
    mvs_batch *b = ctx->scratch;
    if (mask) {   // the caller's buffer outlives the call: every entry point synchronises before it returns
        HIP_TRY(ctx, hipMemcpyAsync(b->d.mask, mask, (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HIP_TRY(ctx, hipMemsetAsync(b->d.mask, 1, (size_t)m, ctx->stream));
    }
    return MVS_OK;
}

mvs_status mvs_triangulate(mvs_ctx *ctx, const double *p1_uv, const double *p2_uv, int m, const double K[9],
                           const double R1to2[9], const double t1to2[3], double *points_xyz, int64_t *point_idx,
                           int *n_points)
{
    if (!ctx || !p1_uv || !p2_uv || !K || !R1to2 || !t1to2 || !n_points || m < 1)  // sfm-solve.cpp:143-144
        return MVS_ERR_INVALID_ARG;
    *n_points = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mvs_status st = stage_points(ctx, p1_uv, p2_uv, m, K);
    if (st != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    mvs_pair_result res;
    std::memset(&res, 0, sizeof(res));
    std::memcpy(res.R1to2, R1to2, sizeof(res.R1to2));
    std::memcpy(res.t1to2, t1to2, sizeof(res.t1to2));
    if ((st = up_async(ctx, b->d.results, &res, sizeof(res))) != MVS_OK)
        return st;
    if ((st = upload_mask(ctx, nullptr, m)) != MVS_OK)
        return st;
    RunParams rp{};
    rp.num_hypotheses = 1;
    launch_finalize(b->d, rp, 1, kFinalizeTriangulate, ctx->stream);
    st = fetch_single(ctx, m, &res, points_xyz, point_idx, nullptr);
    if (st != MVS_OK)
        return st;
    *n_points = res.valid ? res.n_points : 0;
    return MVS_OK;
}

mvs_status mvs_recover_pose(mvs_ctx *ctx, const double E[9], const double *p1_uv, const double *p2_uv, int m,
                            const double K[9], const uint8_t *inlier_mask, double R[9], double t[3],
                            double *points_xyz, int64_t *point_idx, int *n_points, mvs_pair_result *result)
{
    if (!ctx || !E || !p1_uv || !p2_uv || !K || m < 1)
        return MVS_ERR_INVALID_ARG;
    if (n_points)
        *n_points = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mvs_status st = stage_points(ctx, p1_uv, p2_uv, m, K);
    if (st != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    mvs_pair_result res;
    std::memset(&res, 0, sizeof(res));
    std::memcpy(res.E, E, sizeof(res.E));
    if ((st = up_async(ctx, b->d.results, &res, sizeof(res))) != MVS_OK)
        return st;
    if ((st = upload_mask(ctx, inlier_mask, m)) != MVS_OK)
        return st;
    RunParams rp{};
    rp.num_hypotheses = 1;
    launch_finalize(b->d, rp, 1, kFinalizeFromE, ctx->stream);
    st = fetch_single(ctx, m, &res, points_xyz, point_idx, nullptr);
    if (st != MVS_OK)
        return st;
    if (result)
        *result = res;
    if (!res.valid)
        return MVS_NO_MODEL;
    if (R) std::memcpy(R, res.R, sizeof(res.R));
    if (t) std::memcpy(t, res.t, sizeof(res.t));
    if (n_points) *n_points = res.n_points;
    return MVS_OK;
}

mvs_status mvs_find_fundamental_matrix(mvs_ctx *ctx, const double p1_xy[16], const double p2_xy[16], double F[9])
{
    if (!ctx || !p1_xy || !p2_xy || !F)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    double *d = ctx->d_small;  // [0,16) p1, [16,32) p2, [32,41) F, [48] ok flag (as int)
    HIP_TRY(ctx, hipMemcpyAsync(d, p1_xy, 16 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(d + 16, p2_xy, 16 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, sync_stream(ctx));
    launch_fundamental(d, d + 16, d + 32, reinterpret_cast<int *>(d + 48), s);
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipGetLastError());
    int ok = 0;
    HIP_TRY(ctx, hipMemcpy(F, d + 32, 9 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&ok, d + 48, sizeof(int), hipMemcpyDeviceToHost));
    return ok ? MVS_OK : MVS_NO_MODEL;
}

mvs_status mvs_ransac_fundamental(mvs_ctx *ctx, const double *p1_xy, const double *p2_xy, int m, double max_error_sq,
                                  int num_hypotheses, int sampler, uint64_t seed, double F[9], uint8_t *inlier_mask,
                                  int *best_hyp, int *best_count, double *best_residual, int32_t *count,
                                  double *residual)
{
    if (!ctx || !p1_xy || !p2_xy || m < 0 || num_hypotheses < 1)
        return MVS_ERR_INVALID_ARG;
    if (!(max_error_sq > 2.220446049250313e-16))  // estimator-RANSAC.cpp:12
        return MVS_ERR_INVALID_ARG;
    if (best_hyp) *best_hyp = -1;
    if (best_count) *best_count = 0;
    if (best_residual) *best_residual = 0.0;
    if (m < 8)
        return MVS_NO_MODEL;  // estimator-RANSAC.cpp:25-29
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int desc_bytes = ctx->scratch ? ctx->scratch->d.desc_words * 4 : 32;
    mvs_status st = ensure_scratch(ctx, m, desc_bytes);
    if (st != MVS_OK)
        return st;
    mvs_batch *b = ctx->scratch;
    st = ensure_groups(b, num_hypotheses);
    if (st != MVS_OK)
        return st;
    hipStream_t s = ctx->stream;
    std::vector<double> packed((size_t)m * 4);  // ideal-camera points are already normalised: pure packing
    for (int i = 0; i < m; ++i) {
        packed[4 * i] = p1_xy[2 * i];
        packed[4 * i + 1] = p1_xy[2 * i + 1];
        packed[4 * i + 2] = p2_xy[2 * i];
        packed[4 * i + 3] = p2_xy[2 * i + 1];
    }
    const int32_t M = m;
    const int64_t zero = 0;
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if ((st = pin_begin(ctx, packed.size() * sizeof(double) + 2048 + sizeof(mvs_pair_result) + (size_t)m + 256)) != MVS_OK)
        return st;
    if ((st = up_async(ctx, b->d.pts, packed.data(), packed.size() * sizeof(double))) != MVS_OK) return st;
    if ((st = up_async(ctx, b->d.M, &M, sizeof(M))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<int64_t *>(b->d.gidx), &zero, sizeof(zero))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<double *>(b->d.K), eye, sizeof(eye))) != MVS_OK) return st;
    if ((st = up_async(ctx, const_cast<double *>(b->d.Kinv), eye, sizeof(eye))) != MVS_OK) return st;
    if (count || residual) {
        if (b->hyp_table_cap < num_hypotheses) {
            int32_t *hc;
            double *hr;
            if ((st = dev_alloc(b, &hc, (size_t)num_hypotheses)) != MVS_OK) return st;
            if ((st = dev_alloc(b, &hr, (size_t)num_hypotheses)) != MVS_OK) return st;
            b->d.hyp_count = hc;
            b->d.hyp_residual = hr;
            b->hyp_table_cap = num_hypotheses;
            b->allocs_hc = hc;
            b->allocs_hr = hr;
        } else {
            b->d.hyp_count = b->allocs_hc;
            b->d.hyp_residual = b->allocs_hr;
        }
    } else {
        b->d.hyp_count = nullptr;
        b->d.hyp_residual = nullptr;
    }
    RunParams rp{};
    rp.max_error_sq = max_error_sq;
    rp.num_hypotheses = num_hypotheses;
    rp.sampler = sampler;
    rp.seed = seed;
    rp.min_inliers = 0x7fffffff;  // stop after the mask: no decomposition / triangulation wanted here
    launch_ransac(b->d, rp, 1, false, s);
    launch_finalize(b->d, rp, 1, kFinalizeFull, s);
    mvs_pair_result res;
    st = fetch_single(ctx, m, &res, nullptr, nullptr, inlier_mask);
    if (st == MVS_OK && count)
        HIP_TRY(ctx, hipMemcpy(count, b->d.hyp_count, (size_t)num_hypotheses * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (st == MVS_OK && residual)
        HIP_TRY(ctx, hipMemcpy(residual, b->d.hyp_residual, (size_t)num_hypotheses * sizeof(double),
                               hipMemcpyDeviceToHost));
    b->d.hyp_count = nullptr;
    b->d.hyp_residual = nullptr;
    if (st != MVS_OK)
        return st;
    if (F) std::memcpy(F, res.F, sizeof(res.F));
    if (best_hyp) *best_hyp = res.best_hyp;
    if (best_count) *best_count = res.best_count;
    if (best_residual) *best_residual = res.best_residual;
    return res.best_count > 0 ? MVS_OK : MVS_NO_MODEL;  // estimator-RANSAC.cpp:89
}

// ---------------------------------------------------------------------------------------------
// frame sequences (row f2)
// ---------------------------------------------------------------------------------------------
mvs_status mvs_seq_create(mvs_ctx *ctx, int n_frames, int max_kp, int desc_bytes, mvs_seq **out)
{
    if (!ctx || !out || n_frames < 3)
        return MVS_ERR_INVALID_ARG;
    *out = nullptr;
    mvs_seq *q = new mvs_seq();
    q->ctx = ctx;
    q->n_frames = n_frames;
    q->n_tracks = n_frames - 2;
    mvs_status st = batch_create_impl(ctx, n_frames - 1, max_kp, desc_bytes, n_frames, &q->batch);
    if (st != MVS_OK) {
        delete q;
        return st;
    }
    q->stride = std::min(max_kp, kPnpMaxPoints);
    const size_t T = q->n_tracks, S = q->stride;
    double *X, *uv, *xy, *fb, *K, *Kinv;
    int32_t *n_corr, *inl;
    int64_t *gidx;
    PnpOut *po;
#define SALLOC(ptr, cnt) if (st == MVS_OK) st = seq_alloc(q, &(ptr), (cnt));
    SALLOC(X, T * S * 3);
    SALLOC(uv, T * S * 2);
    SALLOC(xy, T * S * 2);
    SALLOC(fb, T * S * 3);
    SALLOC(K, T * 9);
    SALLOC(Kinv, T * 9);
    SALLOC(n_corr, T);
    SALLOC(inl, T * S);
    SALLOC(gidx, T);
    SALLOC(po, T);
    double *traj_R = nullptr, *traj_t = nullptr, *traj_s = nullptr, *trk_s = nullptr;   // scale-propagation fold outputs
    SALLOC(traj_R, (size_t)n_frames * 9);
    SALLOC(traj_t, (size_t)n_frames * 3);
    SALLOC(traj_s, (size_t)n_frames);
    SALLOC(trk_s, (size_t)n_frames);
#undef SALLOC
    if (st != MVS_OK) {
        mvs_seq_destroy(q);
        return st;
    }
    std::vector<int64_t> g(T);
    for (size_t i = 0; i < T; ++i)
        g[i] = (int64_t)i;
    {   // on the ctx stream (it is non-blocking: NULL-stream copies would not be ordered against the kernels that write
        // these buffers later), and with the partly built object released on failure
        hipError_t e = hipMemcpyAsync(gidx, g.data(), T * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(n_corr, 0, T * sizeof(int32_t), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(po, 0, T * sizeof(PnpOut), ctx->stream);
        if (e == hipSuccess) e = sync_stream(ctx);   // `g` lives on this frame
        if (e != hipSuccess) {
            ctx->err = std::string("mvs_seq_create: ") + hipGetErrorString(e);
            mvs_seq_destroy(q);
            return MVS_ERR_HIP;
        }
    }
    const BatchDev &d = q->batch->d;
    q->join.n_tracks = q->n_tracks;
    q->join.max_kp = d.max_kp;
    q->join.stride = q->stride;
    q->join.results = d.results;
    q->join.matches = d.matches;
    q->join.M = d.M;
    q->join.points = d.points;
    q->join.point_idx = d.point_idx;
    q->join.kp = d.kp1;
    q->join.X = X;
    q->join.uv = uv;
    q->join.n_corr = n_corr;
    q->pnp.n_problems = q->n_tracks;
    q->pnp.stride = q->stride;
    q->pnp.n = n_corr;
    q->pnp.gidx = gidx;
    q->pnp.K = K;
    q->pnp.Kinv = Kinv;
    q->pnp.X = X;
    q->pnp.uv = uv;
    q->pnp.xy = xy;
    q->pnp.fb = fb;
    q->pnp.inliers = inl;
    q->pnp.out = po;
    q->chain.n_frames = n_frames;
    q->chain.results = d.results;
    q->chain.tracks = po;
    q->chain.traj_R = traj_R;
    q->chain.traj_t = traj_t;
    q->chain.traj_sigma = traj_s;
    q->chain.track_scale = trk_s;
    q->pnp.rec = nullptr;
    q->pnp.max_groups = 0;
    *out = q;
    return MVS_OK;
}

void mvs_seq_destroy(mvs_seq *q)
{
    if (!q)
        return;
    (void)hipSetDevice(q->ctx->device);
    (void)sync_stream(q->ctx);
    if (q->batch)
        mvs_batch_destroy(q->batch);
    for (void *p : q->allocs)
        (void)hipFree(p);
    delete q;
}

mvs_status mvs_seq_upload(mvs_seq *q, int first, int count, const uint8_t *desc, const float *kp, const int32_t *n_kp,
                          const double K[9])
{
    if (!q || first < 0 || count < 1 || first + count > q->n_frames)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = q->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const BatchDev &d = q->batch->d;
    hipStream_t s = ctx->stream;
    const size_t N = d.max_kp, DW = d.desc_words, off = first;
    if (n_kp)
        for (int i = 0; i < count; ++i)
            if (n_kp[i] < 0 || n_kp[i] > (int)N)
                return MVS_ERR_CAPACITY;
    if (desc)
        HIP_TRY(ctx, hipMemcpyAsync((char *)d.desc1 + off * N * DW * 4, desc, (size_t)count * N * DW * 4, hipMemcpyHostToDevice, s));
    if (kp)
        HIP_TRY(ctx, hipMemcpyAsync((char *)d.kp1 + off * N * 2 * sizeof(float), kp, (size_t)count * N * 2 * sizeof(float),
                                    hipMemcpyHostToDevice, s));
    if (n_kp)
        HIP_TRY(ctx, hipMemcpyAsync((char *)d.n1 + off * sizeof(int32_t), n_kp, (size_t)count * sizeof(int32_t),
                                    hipMemcpyHostToDevice, s));
    std::vector<double> kk, ki;
    std::vector<int64_t> gi;
    if (K) {
        if (!affine_K(K))
            return MVS_ERR_BAD_INTRINSICS;
        double inv[9];
        mat3_inverse(K, inv);
        const size_t P = d.n_pairs;
        kk.resize(P * 9);
        ki.resize(P * 9);
        gi.resize(P);
        for (size_t i = 0; i < P; ++i) {
            std::memcpy(&kk[9 * i], K, 9 * sizeof(double));
            std::memcpy(&ki[9 * i], inv, 9 * sizeof(double));
            gi[i] = (int64_t)i;
        }
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<double *>(d.K), kk.data(), P * 9 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<double *>(d.Kinv), ki.data(), P * 9 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<int64_t *>(d.gidx), gi.data(), P * sizeof(int64_t), hipMemcpyHostToDevice, s));
        const size_t T = q->n_tracks;
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<double *>(q->pnp.K), kk.data(), T * 9 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<double *>(q->pnp.Kinv), ki.data(), T * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(ctx, sync_stream(ctx));
    return MVS_OK;
}

static mvs_status seq_prepare(mvs_seq *q, const mvs_params *tv, const mvs_pnp_params *pp)
{
    if (!q || !pp || pp->num_hypotheses < 1 || !(pp->reproj_error > 0.0))
        return MVS_ERR_INVALID_ARG;
    if (pp->sampler != MVS_SAMPLER_IDENTITY && pp->sampler != MVS_SAMPLER_PHILOX)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = check_params(tv);
    if (st != MVS_OK)
        return st;
    HIP_TRY(q->ctx, hipSetDevice(q->ctx->device));
    if ((st = ensure_groups(q->batch, tv->num_hypotheses)) != MVS_OK)
        return st;
    const int G = (pp->num_hypotheses + 255) / 256;
    if (G > q->rec_groups) {
        HIP_TRY(q->ctx, sync_stream(q->ctx));
        PnpRec *rec;
        if ((st = seq_alloc(q, &rec, (size_t)q->n_tracks * G)) != MVS_OK)
            return st;
        q->pnp.rec = rec;
        q->pnp.max_groups = G;
        q->rec_groups = G;
    }
    q->batch->d.hyp_count = nullptr;
    q->batch->d.hyp_residual = nullptr;
    q->pnp.num_hypotheses = pp->num_hypotheses;
    q->pnp.sampler = pp->sampler;
    q->pnp.min_inliers = pp->min_inliers;
    q->pnp.seed = pp->seed;
    q->pnp.thr2 = pp->reproj_error * pp->reproj_error;
    q->refit_on = pp->refit != 0;
    if (q->refit_on && !q->refit_ready) {
        const size_t T = q->n_tracks, S = q->stride;
        double *obs0, *oi0, *p0, *pi, *pts, *tmp, *pose;
        int32_t *m;
        mvs_refine_result *out;
        if ((st = seq_alloc(q, &obs0, T * S * 2)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &oi0, T * S * 3)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &p0, T * S * 3)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &pi, T * S * 6)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &pts, T * S * 3)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &tmp, T * S * 3)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &pose, T * 12)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &m, T)) != MVS_OK) return st;
        if ((st = seq_alloc(q, &out, T)) != MVS_OK) return st;
        RefineDev &d = q->refit;
        d = RefineDev{};
        d.n_problems = (int)T;
        d.stride = (int)S;
        d.n_frames = 1;
        d.m = m;
        d.K = q->pnp.K;
        d.pose0 = pose;
        d.obs[0] = obs0;
        d.oinfo[0] = oi0;
        d.pts0 = p0;
        d.pinfo = pi;
        d.pts = pts;
        d.pts_tmp = tmp;
        d.out = out;
        q->refit_ready = true;
    }
    if (q->refit_on) {
        mvs_refine_params rp;
        mvs_refine_params_default(&rp);
        rp.pose_sigma[0] = rp.pose_sigma[1] = kRefitPoseSigma;
        q->refit.cfg = to_cfg(rp, 1);
    }
    return MVS_OK;
}

static mvs_status seq_enqueue(mvs_seq *q, const RunParams &rp)
{
    mvs_status st = enqueue_pipeline(q->batch, rp, q->n_frames - 1, false, nullptr);
    if (st != MVS_OK)
        return st;
    launch_seq_join(q->join, q->ctx->stream);
    launch_pnp(q->pnp, q->ctx->stream);
    if (q->refit_on)
        launch_pnp_refit(q->pnp, q->refit, kRefitPointSigma, q->ctx->stream);
    q->chain.n_corr = q->pnp.n;
    launch_seq_chain(q->chain, q->ctx->stream);
    HIP_TRY(q->ctx, hipGetLastError());
    return MVS_OK;
}

mvs_status mvs_seq_run(mvs_seq *q, const mvs_params *two_view, const mvs_pnp_params *pnp)
{
    mvs_status st = seq_prepare(q, two_view, pnp);
    if (st != MVS_OK)
        return st;
    return seq_enqueue(q, to_run(*two_view));
}

mvs_status mvs_seq_sync(mvs_seq *q)
{
    if (!q)
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(q->ctx, sync_stream(q->ctx));
    return MVS_OK;
}

mvs_status mvs_seq_time(mvs_seq *q, const mvs_params *two_view, const mvs_pnp_params *pnp, int warmup, int steps,
                        float *ms_total)
{
    if (steps < 1 || warmup < 0 || !ms_total)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = seq_prepare(q, two_view, pnp);
    if (st != MVS_OK)
        return st;
    const RunParams rp = to_run(*two_view);
    hipStream_t s = q->ctx->stream;
    for (int i = 0; i < warmup; ++i)
        if ((st = seq_enqueue(q, rp)) != MVS_OK)
            return st;
    HIP_TRY(q->ctx, sync_stream(q->ctx));
    HIP_TRY(q->ctx, hipEventRecord(q->batch->ev[5], s));
    for (int i = 0; i < steps; ++i)
        if ((st = seq_enqueue(q, rp)) != MVS_OK)
            return st;
    HIP_TRY(q->ctx, hipEventRecord(q->batch->ev[6], s));
    HIP_TRY(q->ctx, hipEventSynchronize(q->batch->ev[6]));
    HIP_TRY(q->ctx, hipEventElapsedTime(ms_total, q->batch->ev[5], q->batch->ev[6]));
    return MVS_OK;
}

// per-stage HIP-event timing of the sequence step on the kernels' own stream: ms_stage[4] = summed ms over `steps`
// instrumented passes of {pair pipeline, join, pnp_solve (prep + ransac + finalize), chain}
mvs_status mvs_seq_time_stages(mvs_seq *q, const mvs_params *two_view, const mvs_pnp_params *pnp, int steps, float *ms_stage)
{
    if (steps < 1 || !ms_stage)
        return MVS_ERR_INVALID_ARG;
    mvs_status st = seq_prepare(q, two_view, pnp);
    if (st != MVS_OK)
        return st;
    const RunParams rp = to_run(*two_view);
    hipStream_t s = q->ctx->stream;
    hipEvent_t *ev = q->batch->ev;
    for (int k = 0; k < 4; ++k)
        ms_stage[k] = 0.f;
    for (int i = 0; i < steps; ++i) {
        HIP_TRY(q->ctx, hipEventRecord(ev[0], s));
        if ((st = enqueue_pipeline(q->batch, rp, q->n_frames - 1, false, nullptr)) != MVS_OK)
            return st;
        HIP_TRY(q->ctx, hipEventRecord(ev[1], s));
        launch_seq_join(q->join, s);
        HIP_TRY(q->ctx, hipEventRecord(ev[2], s));
        launch_pnp(q->pnp, s);
        if (q->refit_on)
            launch_pnp_refit(q->pnp, q->refit, kRefitPointSigma, s);
        HIP_TRY(q->ctx, hipEventRecord(ev[3], s));
        q->chain.n_corr = q->pnp.n;
        launch_seq_chain(q->chain, s);
        HIP_TRY(q->ctx, hipEventRecord(ev[4], s));
        HIP_TRY(q->ctx, hipEventSynchronize(ev[4]));
        for (int k = 0; k < 4; ++k) {
            float ms = 0.f;
            HIP_TRY(q->ctx, hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            ms_stage[k] += ms;
        }
    }
    return MVS_OK;
}

mvs_status mvs_seq_download_pairs(mvs_seq *q, int first, int count, mvs_pair_result *results, mvs_match *matches,
                                  uint8_t *inlier_mask, double *points_xyz, int64_t *point_idx)
{
    if (!q)
        return MVS_ERR_INVALID_ARG;
    return mvs_batch_download(q->batch, first, count, results, matches, inlier_mask, points_xyz, point_idx);
}

mvs_status mvs_seq_download_tracks(mvs_seq *q, int first, int count, mvs_track_result *tracks, double *corr_xyz,
                                   double *corr_uv, int64_t *inlier_idx)
{
    if (!q || first < 0 || count < 1 || first + count > q->n_tracks)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = q->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_stream(ctx));
    const size_t S = q->stride, N = q->batch->d.max_kp, off = first, cnt = count;
    if (tracks) {
        std::vector<PnpOut> po(cnt);
        std::vector<int32_t> nc(cnt);
        HIP_TRY(ctx, hipMemcpy(po.data(), q->pnp.out + off, cnt * sizeof(PnpOut), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(nc.data(), q->pnp.n + off, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i) {
            mvs_track_result &t = tracks[i];
            t.ok = nc[i] >= 7 ? po[i].ok : 0;
            t.n_corr = nc[i];
            t.n_inliers = t.ok ? po[i].n_inliers : 0;
            t.best_hyp = nc[i] >= 7 ? po[i].best_hyp : -1;
            std::memcpy(t.R, po[i].R, sizeof(t.R));
            std::memcpy(t.t, po[i].t, sizeof(t.t));
        }
    }
    // device layout has `stride` slots per track, the host layout max_kp
    auto fetch = [&](const void *dev, size_t elem_bytes, size_t per, void *host) -> mvs_status {
        std::vector<char> tmp(cnt * S * per * elem_bytes);
        HIP_TRY(ctx, hipMemcpy(tmp.data(), (const char *)dev + off * S * per * elem_bytes, tmp.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i)
            std::memcpy((char *)host + i * N * per * elem_bytes, tmp.data() + i * S * per * elem_bytes, S * per * elem_bytes);
        return MVS_OK;
    };
    mvs_status st = MVS_OK;
    if (corr_xyz && (st = fetch(q->pnp.X, sizeof(double), 3, corr_xyz)) != MVS_OK)
        return st;
    if (corr_uv && (st = fetch(q->pnp.uv, sizeof(double), 2, corr_uv)) != MVS_OK)
        return st;
    if (inlier_idx) {
        std::vector<int32_t> tmp(cnt * S);
        HIP_TRY(ctx, hipMemcpy(tmp.data(), q->pnp.inliers + off * S, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i)
            for (size_t k = 0; k < S; ++k)
                inlier_idx[i * N + k] = tmp[i * S + k];
    }
    return MVS_OK;
}

mvs_status mvs_pnp_params_default(mvs_pnp_params *p)
{
    if (!p)
        return MVS_ERR_INVALID_ARG;
    p->num_hypotheses = 100;  // pnp-solve.cpp:47
    p->sampler = MVS_SAMPLER_PHILOX;
    p->seed = 0;
    p->reproj_error = 0.05;   // pnp-solve.cpp:48
    p->min_inliers = 4;
    p->refit = 0;
    return MVS_OK;
}

mvs_status mvs_pnp_solve(mvs_ctx *ctx, const double *world_xyz, const double *image_uv, int n, const double K[9],
                         const mvs_pnp_params *params, double R[9], double t[3], int64_t *inlier_idx, int *n_inliers,
                         int *best_hyp)
{
    if (!ctx || !world_xyz || !image_uv || !K || !params || !n_inliers)
        return MVS_ERR_INVALID_ARG;
    *n_inliers = 0;
    if (best_hyp)
        *best_hyp = -1;
    if (n < 7 || params->num_hypotheses < 1 || !(params->reproj_error > 0.0))  // pnp-solve.cpp:13,22-23 (assert)
        return MVS_ERR_INVALID_ARG;
    if (params->sampler != MVS_SAMPLER_IDENTITY && params->sampler != MVS_SAMPLER_PHILOX)
        return MVS_ERR_INVALID_ARG;
    if (n > kPnpMaxPoints)
        return MVS_ERR_CAPACITY;
    if (!affine_K(K))
        return MVS_ERR_BAD_INTRINSICS;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int G = (params->num_hypotheses + 255) / 256;
    // workspace layout: X[3n] uv[2n] xy[2n] fb[3n] | K[9] Kinv[9] | rec[G] | out | inliers[n] n
    const size_t nd = (size_t)n * 10 * sizeof(double);
    const size_t off_k = (nd + 63) & ~size_t(63);
    const size_t off_rec = (off_k + 18 * sizeof(double) + 63) & ~size_t(63);
    const size_t off_out = (off_rec + (size_t)G * sizeof(PnpRec) + 63) & ~size_t(63);
    const size_t off_inl = (off_out + sizeof(PnpOut) + 63) & ~size_t(63);
    const size_t total = off_inl + ((size_t)n + 1) * sizeof(int32_t);
    if (ctx->pnp_bytes < total) {
        if (ctx->d_pnp) (void)hipFree(ctx->d_pnp);
        ctx->d_pnp = nullptr;
        ctx->pnp_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_pnp, total));
        ctx->pnp_bytes = total;
    }
    char *base = static_cast<char *>(ctx->d_pnp);
    double *dX = reinterpret_cast<double *>(base), *duv = dX + 3 * (size_t)n, *dxy = duv + 2 * (size_t)n, *dfb = dxy + 2 * (size_t)n;
    double *dK = reinterpret_cast<double *>(base + off_k);
    int32_t *dinl = reinterpret_cast<int32_t *>(base + off_inl);
    hipStream_t s = ctx->stream;
    double kk[18];
    std::memcpy(kk, K, 9 * sizeof(double));
    mat3_inverse(K, kk + 9);
    const int32_t n32 = n;
    HIP_TRY(ctx, hipMemcpyAsync(dX, world_xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(duv, image_uv, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dK, kk, sizeof(kk), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dinl + n, &n32, sizeof(n32), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, sync_stream(ctx));
    PnpDev p{};
    p.n_problems = 1;
    p.stride = n;
    p.num_hypotheses = params->num_hypotheses;
    p.sampler = params->sampler;
    p.min_inliers = params->min_inliers;
    p.max_groups = G;
    p.seed = params->seed;
    p.thr2 = params->reproj_error * params->reproj_error;
    p.n = dinl + n;
    p.gidx = nullptr;
    p.K = dK;
    p.Kinv = dK + 9;
    p.X = dX;
    p.uv = duv;
    p.xy = dxy;
    p.fb = dfb;
    p.rec = reinterpret_cast<PnpRec *>(base + off_rec);
    p.out = reinterpret_cast<PnpOut *>(base + off_out);
    p.inliers = dinl;
    launch_pnp(p, s);
    HIP_TRY(ctx, sync_stream(ctx));
    HIP_TRY(ctx, hipGetLastError());
    PnpOut out;
    HIP_TRY(ctx, hipMemcpy(&out, p.out, sizeof(out), hipMemcpyDeviceToHost));
    if (best_hyp)
        *best_hyp = out.best_hyp;
    if (!out.ok)
        return MVS_NO_MODEL;
    *n_inliers = out.n_inliers;
    std::vector<int32_t> tmp;
    if ((inlier_idx || params->refit) && out.n_inliers > 0) {
        tmp.resize(out.n_inliers);
        HIP_TRY(ctx, hipMemcpy(tmp.data(), p.inliers, (size_t)out.n_inliers * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (inlier_idx)
            for (int i = 0; i < out.n_inliers; ++i)
                inlier_idx[i] = tmp[i];
    }
    if (params->refit && out.n_inliers >= 4) {
        // cv::solvePnPRansac ends with a refit over the inliers (pnp-solve.cpp:53-64).  Here: the refinement kernel as
        // a motion-only problem -- the inliers' world points held by a prior of sigma 1e-9 (fixed), identity image
        // covariances (unweighted pixels), no prior on the pose.  A failed refit keeps the RANSAC pose.
        const size_t m = out.n_inliers;
        std::vector<double> X(3 * m), XC(9 * m, 0.0), uv(2 * m);
        for (size_t i = 0; i < m; ++i) {
            std::memcpy(&X[3 * i], world_xyz + 3 * (size_t)tmp[i], 3 * sizeof(double));
            std::memcpy(&uv[2 * i], image_uv + 2 * (size_t)tmp[i], 2 * sizeof(double));
            XC[9 * i] = XC[9 * i + 4] = XC[9 * i + 8] = kRefitPointSigma * kRefitPointSigma;
        }
        mvs_refine_params rp;
        mvs_refine_params_default(&rp);
        rp.pose_sigma[0] = rp.pose_sigma[1] = kRefitPoseSigma;
        mvs_refine_result rr;
        if (mvs_pnp_refine(ctx, X.data(), XC.data(), uv.data(), nullptr, (int)m, K, out.R, out.t, &rp, &rr) == MVS_OK && rr.ok) {
            std::memcpy(out.R, rr.R, sizeof(out.R));
            std::memcpy(out.t, rr.t, sizeof(out.t));
        }
    }
    if (R) std::memcpy(R, out.R, sizeof(out.R));
    if (t) std::memcpy(t, out.t, sizeof(out.t));
    return MVS_OK;
}

// ---------------------------------------------------------------------------------------------
// refinement (row f4)
void mvs_refine_params_default(mvs_refine_params *p)
{
    if (!p)
        return;
    p->max_iterations = 100;
    p->reserved = 0;
    p->lambda_initial = 1e-5;
    p->lambda_factor = 10.0;
    p->lambda_upper = 1e5;
    p->rel_tol = 1e-12;
    p->abs_tol = 1e-12;
    p->anchor_sigma[0] = p->anchor_sigma[1] = 1e-5;
    p->pose_sigma[0] = p->pose_sigma[1] = 1e-2;
    p->point_sigma = 1e-2;
}

static bool refine_params_ok(const mvs_refine_params *p)
{
    return p && p->max_iterations >= 0 && p->lambda_initial >= 0.0 && p->lambda_factor > 1.0 &&
           p->lambda_upper > 0.0 && p->anchor_sigma[0] > 0.0 && p->anchor_sigma[1] > 0.0 && p->pose_sigma[0] > 0.0 &&
           p->pose_sigma[1] > 0.0 && p->point_sigma > 0.0;
}

static RefineCfg to_cfg(const mvs_refine_params &p, int n_frames)
{
    RefineCfg c{};
    c.max_iterations = p.max_iterations;
    c.lambda_initial = p.lambda_initial;
    c.lambda_factor = p.lambda_factor;
    c.lambda_upper = p.lambda_upper;
    c.rel_tol = p.rel_tol;
    c.abs_tol = p.abs_tol;
    // the reference fills diagonal entries 0-2 with its "position" and 3-5 with its "orientation" sigma
    // (sfm-refine.cpp:62-66); GTSAM reads the first three as rotation.  Kept as is: entry k gets sigma[k / 3].
    for (int k = 0; k < 6; ++k) {
        const double a = p.anchor_sigma[k / 3], m = p.pose_sigma[k / 3];
        if (n_frames == 2) {
            c.w[0][k] = 1.0 / (a * a);
            c.w[1][k] = 1.0 / (m * m);
        } else {
            c.w[0][k] = 1.0 / (m * m);
            c.w[1][k] = 0.0;
        }
    }
    return c;
}

// extras of the general two-frame problem (mvs_ba_refine); all null for sfm_refine / pnp_refine
struct RefineExtra {
    const double *poses_all = nullptr;    // frames x 12: every frame's guess
    const uint8_t *valid[2] = {nullptr, nullptr};
    const RefineCfg *cfg = nullptr;       // explicit prior weights
    mvs_refine_result *results_all = nullptr;  // frames records out
};

// one problem through the ctx workspace.  frames: 2 = sfm_refine (obs_a = camera 1, obs_b = camera 2), 1 = pnp_refine
static mvs_status refine_single(mvs_ctx *ctx, int frames, const double *obs_a, const double *cov_a, const double *obs_b,
                                const double *cov_b, const double *pts0, const double *cov3, int m, const double K[9],
                                const double R_guess[9], const double t_guess[3], const mvs_refine_params *params,
                                mvs_refine_result *result, double *points_out, double *point_cov_out,
                                const RefineExtra &ex = RefineExtra())
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t M = (size_t)m;
    // doubles: obs0 2 obs1 2 oinfo0 3 oinfo1 3 pts0 3 pinfo 6 pts 3 tmp 3 pcov 9 cov2a 4 cov2b 4 cov3 9 = 51 per point
    const size_t nd = M * 51 + 9 + 12 + 24;
    const size_t off_out = (nd * sizeof(double) + 63) & ~size_t(63);
    const size_t off_all = (off_out + sizeof(mvs_refine_result) + 63) & ~size_t(63);
    const size_t off_m = (off_all + 2 * sizeof(mvs_refine_result) + 63) & ~size_t(63);
    const size_t off_valid = off_m + 64;
    const size_t total = off_valid + 2 * ((M + 63) & ~size_t(63));
    if (ctx->ref_bytes < total) {
        if (ctx->d_ref) (void)hipFree(ctx->d_ref);
        ctx->d_ref = nullptr;
        ctx->ref_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_ref, total));
        ctx->ref_bytes = total;
    }
    char *base = static_cast<char *>(ctx->d_ref);
    double *w = reinterpret_cast<double *>(base);
    double *obs0 = w, *obs1 = obs0 + 2 * M, *oinfo0 = obs1 + 2 * M, *oinfo1 = oinfo0 + 3 * M, *dp0 = oinfo1 + 3 * M;
    double *pinfo = dp0 + 3 * M, *dpts = pinfo + 6 * M, *dtmp = dpts + 3 * M, *dpcov = dtmp + 3 * M;
    double *c2a = dpcov + 9 * M, *c2b = c2a + 4 * M, *c3 = c2b + 4 * M, *dK = c3 + 9 * M, *dpose = dK + 9, *dpose_all = dpose + 12;
    int32_t *dm = reinterpret_cast<int32_t *>(base + off_m);
    uint8_t *dv0 = reinterpret_cast<uint8_t *>(base + off_valid), *dv1 = dv0 + ((M + 63) & ~size_t(63));
    hipStream_t s = ctx->stream;
    const int32_t m32 = m;
    double pose[12];
    std::memcpy(pose, R_guess, 9 * sizeof(double));
    std::memcpy(pose + 9, t_guess, 3 * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(obs0, obs_a, M * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    if (frames == 2)
        HIP_TRY(ctx, hipMemcpyAsync(obs1, obs_b, M * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    if (cov_a)
        HIP_TRY(ctx, hipMemcpyAsync(c2a, cov_a, M * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    if (frames == 2 && cov_b)
        HIP_TRY(ctx, hipMemcpyAsync(c2b, cov_b, M * 4 * sizeof(double), hipMemcpyHostToDevice, s));
    if (cov3)
        HIP_TRY(ctx, hipMemcpyAsync(c3, cov3, M * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    if (ex.poses_all)
        HIP_TRY(ctx, hipMemcpyAsync(dpose_all, ex.poses_all, (size_t)frames * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    if (ex.valid[0])
        HIP_TRY(ctx, hipMemcpyAsync(dv0, ex.valid[0], M, hipMemcpyHostToDevice, s));
    if (frames == 2 && ex.valid[1])
        HIP_TRY(ctx, hipMemcpyAsync(dv1, ex.valid[1], M, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dp0, pts0, M * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dpose, pose, sizeof(pose), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dm, &m32, sizeof(m32), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, sync_stream(ctx));  // the staging buffers above live on this frame
    RefineDev d{};
    d.n_problems = 1;
    d.stride = m;
    d.n_frames = frames;
    d.cfg = ex.cfg ? *ex.cfg : to_cfg(*params, frames);
    d.m = dm;
    d.K = dK;
    d.pose0 = dpose;
    d.pose0_all = ex.poses_all ? dpose_all : nullptr;
    d.obs[0] = obs0;
    d.obs[1] = obs1;
    d.oinfo[0] = oinfo0;
    d.oinfo[1] = oinfo1;
    d.pts0 = dp0;
    d.pinfo = pinfo;
    d.pts = dpts;
    d.pts_tmp = dtmp;
    d.point_cov = point_cov_out ? dpcov : nullptr;
    d.out = reinterpret_cast<mvs_refine_result *>(base + off_out);
    d.out_all = ex.results_all ? reinterpret_cast<mvs_refine_result *>(base + off_all) : nullptr;
    launch_refine_prep(d, cov_a ? c2a : nullptr, (frames == 2 && cov_b) ? c2b : nullptr, cov3 ? c3 : nullptr,
                       1.0 / (params->point_sigma * params->point_sigma), oinfo0, oinfo1, pinfo,
                       ex.valid[0] ? dv0 : nullptr, (frames == 2 && ex.valid[1]) ? dv1 : nullptr, s);
    launch_refine(d, s);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(result, d.out, sizeof(*result), hipMemcpyDeviceToHost, s));
    if (ex.results_all)
        HIP_TRY(ctx, hipMemcpyAsync(ex.results_all, d.out_all, (size_t)frames * sizeof(mvs_refine_result), hipMemcpyDeviceToHost, s));
    if (points_out)
        HIP_TRY(ctx, hipMemcpyAsync(points_out, dpts, M * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (point_cov_out)
        HIP_TRY(ctx, hipMemcpyAsync(point_cov_out, dpcov, M * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, sync_stream(ctx));
    return result->ok ? MVS_OK : MVS_NO_MODEL;
}

mvs_status mvs_sfm_refine(mvs_ctx *ctx, const double *p1, const double *cov1, const double *p2, const double *cov2, int m,
                          const double K[9], const double R_guess[9], const double t_guess[3],
                          const double *points_guess, const mvs_refine_params *params, mvs_refine_result *result,
                          double *points_out, double *point_cov_out)
{
    if (!ctx || !p1 || !p2 || !K || !R_guess || !t_guess || !points_guess || !result || !refine_params_ok(params))
        return MVS_ERR_INVALID_ARG;
    std::memset(result, 0, sizeof(*result));
    if (m < 1)
        return MVS_ERR_INVALID_ARG;
    if (m > kMaxKp)
        return MVS_ERR_CAPACITY;
    if (!affine_K(K))
        return MVS_ERR_BAD_INTRINSICS;
    return refine_single(ctx, 2, p1, cov1, p2, cov2, points_guess, nullptr, m, K, R_guess, t_guess, params, result,
                         points_out, point_cov_out);
}

mvs_status mvs_pnp_refine(mvs_ctx *ctx, const double *world, const double *world_cov, const double *image,
                          const double *image_cov, int m, const double K[9], const double R_guess[9],
                          const double t_guess[3], const mvs_refine_params *params, mvs_refine_result *result)
{
    if (!ctx || !world || !world_cov || !image || !K || !R_guess || !t_guess || !result || !refine_params_ok(params))
        return MVS_ERR_INVALID_ARG;
    std::memset(result, 0, sizeof(*result));
    if (m < 1)
        return MVS_ERR_INVALID_ARG;
    if (m > kMaxKp)
        return MVS_ERR_CAPACITY;
    if (!affine_K(K))
        return MVS_ERR_BAD_INTRINSICS;
    return refine_single(ctx, 1, image, image_cov, nullptr, nullptr, world, world_cov, m, K, R_guess, t_guess, params,
                         result, nullptr, nullptr);
}

mvs_status mvs_ba_refine(mvs_ctx *ctx, const mvs_ba_problem *pb, const mvs_refine_params *params, mvs_refine_result *frames_out,
                         double *points_out, double *point_cov_out)
{
    if (!ctx || !pb || !frames_out || !refine_params_ok(params) || !pb->K || !pb->frame_pose || !pb->frame_prior_var ||
        !pb->points || !pb->obs[0] || (pb->n_frames != 1 && pb->n_frames != 2) || (pb->n_frames == 2 && !pb->obs[1]))
        return MVS_ERR_INVALID_ARG;
    std::memset(frames_out, 0, (size_t)pb->n_frames * sizeof(mvs_refine_result));
    if (pb->n_points < 1)
        return MVS_ERR_INVALID_ARG;
    if (pb->n_points > kMaxKp)
        return MVS_ERR_CAPACITY;
    if (!affine_K(pb->K))
        return MVS_ERR_BAD_INTRINSICS;
    RefineCfg cfg = to_cfg(*params, pb->n_frames);
    for (int f = 0; f < pb->n_frames; ++f)
        for (int k = 0; k < 6; ++k) {
            const double v = pb->frame_prior_var[6 * f + k];
            cfg.w[f][k] = v > 0.0 ? 1.0 / v : 0.0;   // <= 0: no prior on this coordinate
        }
    RefineExtra ex;
    ex.poses_all = pb->frame_pose;
    ex.valid[0] = pb->obs_valid[0];
    ex.valid[1] = pb->obs_valid[1];
    ex.cfg = &cfg;
    ex.results_all = frames_out;
    // a NULL point_prior_cov means "no prior on any point": hand the kernel an all-zero covariance table
    std::vector<double> none;
    const double *pc = pb->point_prior_cov;
    if (!pc) {
        none.assign((size_t)pb->n_points * 9, 0.0);
        pc = none.data();
    }
    mvs_refine_result last;
    const double *pose_last = pb->frame_pose + 12 * (pb->n_frames - 1);
    return refine_single(ctx, pb->n_frames, pb->obs[0], pb->obs_cov[0], pb->obs[1], pb->obs_cov[1], pb->points, pc,
                         pb->n_points, pb->K, pose_last, pose_last + 9, params, &last, points_out, point_cov_out, ex);
}

mvs_status mvs_batch_refine(mvs_batch *b, const mvs_refine_params *params, double sigma_px)
{
    if (!b || !refine_params_ok(params) || !(sigma_px > 0.0))
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const BatchDev &bd = b->d;
    const size_t P = bd.n_pairs, N = bd.max_kp;
    RefineDev &d = b->refine;
    if (!b->refine_ready) {
        auto grab = [&](size_t bytes, void **out) -> hipError_t {
            hipError_t e = hipMalloc(out, bytes);
            if (e == hipSuccess)
                b->allocs.push_back(*out);
            return e;
        };
        double *obs0, *obs1, *oi0, *oi1, *p0, *pi, *pts, *tmp, *pc, *pose;
        int32_t *m;
        mvs_refine_result *out;
        HIP_TRY(ctx, grab(P * N * 2 * sizeof(double), (void **)&obs0));
        HIP_TRY(ctx, grab(P * N * 2 * sizeof(double), (void **)&obs1));
        HIP_TRY(ctx, grab(P * N * 3 * sizeof(double), (void **)&oi0));
        HIP_TRY(ctx, grab(P * N * 3 * sizeof(double), (void **)&oi1));
        HIP_TRY(ctx, grab(P * N * 3 * sizeof(double), (void **)&p0));
        HIP_TRY(ctx, grab(P * N * 6 * sizeof(double), (void **)&pi));
        HIP_TRY(ctx, grab(P * N * 3 * sizeof(double), (void **)&pts));
        HIP_TRY(ctx, grab(P * N * 3 * sizeof(double), (void **)&tmp));
        HIP_TRY(ctx, grab(P * N * 9 * sizeof(double), (void **)&pc));
        HIP_TRY(ctx, grab(P * 12 * sizeof(double), (void **)&pose));
        HIP_TRY(ctx, grab(P * sizeof(int32_t), (void **)&m));
        HIP_TRY(ctx, grab(P * sizeof(mvs_refine_result), (void **)&out));
        d.n_problems = (int)P;
        d.stride = (int)N;
        d.n_frames = 2;
        d.m = m;
        d.K = bd.K;
        d.pose0 = pose;
        d.obs[0] = obs0;
        d.obs[1] = obs1;
        d.oinfo[0] = oi0;
        d.oinfo[1] = oi1;
        d.pts0 = p0;
        d.pinfo = pi;
        d.pts = pts;
        d.pts_tmp = tmp;
        d.point_cov = pc;
        d.out = out;
        b->refine_ready = true;
    }
    d.cfg = to_cfg(*params, 2);
    hipStream_t s = ctx->stream;
    launch_refine_gather(bd, (int)P, sigma_px, params->point_sigma, d.stride, const_cast<int32_t *>(d.m),
                         const_cast<double *>(d.pose0), const_cast<double *>(d.obs[0]), const_cast<double *>(d.obs[1]),
                         const_cast<double *>(d.oinfo[0]), const_cast<double *>(d.oinfo[1]),
                         const_cast<double *>(d.pts0), const_cast<double *>(d.pinfo), s);
    launch_refine(d, s);
    HIP_TRY(ctx, hipGetLastError());
    b->refine_ran = true;
    return MVS_OK;
}

mvs_status mvs_batch_download_refined(mvs_batch *b, mvs_refine_result *refined, double *points_xyz, double *point_cov)
{
    if (!b || !refined || !b->refine_ran)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const RefineDev &d = b->refine;
    const size_t P = d.n_problems, N = d.stride;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(refined, d.out, P * sizeof(mvs_refine_result), hipMemcpyDeviceToHost, s));
    if (points_xyz)
        HIP_TRY(ctx, hipMemcpyAsync(points_xyz, d.pts, P * N * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (point_cov)
        HIP_TRY(ctx, hipMemcpyAsync(point_cov, d.point_cov, P * N * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, sync_stream(ctx));
    return MVS_OK;
}

mvs_status mvs_seq_download_trajectory(mvs_seq *q, double *R, double *t, double *pair_scale, double *track_scale)
{
    if (!q)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = q->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t F = q->n_frames;
    if (R)
        HIP_TRY(ctx, hipMemcpyAsync(R, q->chain.traj_R, F * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (t)
        HIP_TRY(ctx, hipMemcpyAsync(t, q->chain.traj_t, F * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (pair_scale)
        HIP_TRY(ctx, hipMemcpyAsync(pair_scale, q->chain.traj_sigma, (F - 1) * sizeof(double), hipMemcpyDeviceToHost, s));
    if (track_scale)
        HIP_TRY(ctx, hipMemcpyAsync(track_scale, q->chain.track_scale, (F - 2) * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, sync_stream(ctx));
    return MVS_OK;
}

mvs_status mvs_seq_refine_pairs(mvs_seq *q, const mvs_refine_params *params, double sigma_px)
{
    return q ? mvs_batch_refine(q->batch, params, sigma_px) : MVS_ERR_INVALID_ARG;
}

mvs_status mvs_batch_upload_octaves(mvs_batch *b, int first, int count, const uint8_t *base_octave,
                                    const uint8_t *pair_octave)
{
    if (!b || first < 0 || count < 1 || first + count > b->d.n_pairs)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = b->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t N = b->d.max_kp, off = (size_t)first * N, bytes = (size_t)count * N;
    for (const uint8_t *o : {base_octave, pair_octave})
        if (o)
            for (size_t i = 0; i < bytes; ++i)
                if (o[i] > 30)
                    return MVS_ERR_INVALID_ARG;   // stddev = (1 << octave) * 0.5 (visual-feature.cpp:203)
    if (base_octave)
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint8_t *>(b->d.oct1) + off, base_octave, bytes, hipMemcpyHostToDevice,
                                    ctx->stream));
    if (pair_octave)
        HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint8_t *>(b->d.oct2) + off, pair_octave, bytes, hipMemcpyHostToDevice,
                                    ctx->stream));
    HIP_TRY(ctx, sync_stream(ctx));
    return MVS_OK;
}

mvs_status mvs_seq_upload_octaves(mvs_seq *q, int first, int count, const uint8_t *octave)
{
    if (!q || !octave || first < 0 || count < 1 || first + count > q->n_frames)
        return MVS_ERR_INVALID_ARG;
    mvs_ctx *ctx = q->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t N = q->batch->d.max_kp, bytes = (size_t)count * N;
    for (size_t i = 0; i < bytes; ++i)
        if (octave[i] > 30)
            return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint8_t *>(q->batch->d.oct1) + (size_t)first * N, octave, bytes,
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, sync_stream(ctx));
    return MVS_OK;
}

mvs_status mvs_seq_download_refined(mvs_seq *q, mvs_refine_result *refined, double *points_xyz, double *point_cov)
{
    return q ? mvs_batch_download_refined(q->batch, refined, points_xyz, point_cov) : MVS_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------------------------
// extraction (row f3)
void mvs_orb_params_default(mvs_orb_params *p)
{
    if (!p)
        return;
    p->nfeatures = 500;
    p->nlevels = 8;
    p->edge_threshold = 31;
    p->fast_threshold = 20;
}

static void philox_host(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4])
{   // Philox4x32-10 (Random123), counter (c0, c1, 0, 0)
    uint32_t c[4] = {c0, c1, 0, 0};
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1,
                       n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    std::memcpy(out, c, sizeof(c));
}

// the 256 point pairs of the steered-BRIEF tests
static void orb_pattern_host(int8_t P[1024])
{
    for (int i = 0; i < 256; ++i) {
        uint32_t w[8];
        philox_host((uint32_t)i, 0, 0x0B5EED00u, 0x31u, w);
        philox_host((uint32_t)i, 1, 0x0B5EED00u, 0x31u, w + 4);
        int c[4];
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = w[2 * k], b = w[2 * k + 1];
            const int u = (int)((a & 0xffffu) + (a >> 16) + (b & 0xffffu) + (b >> 16)) - 131072;
            c[k] = std::max(-13, std::min(13, (u * 11) / 65536));
        }
        if (c[0] == c[2] && c[1] == c[3])
            c[2] = c[0] + (c[0] < 0 ? 3 : -3);
        for (int k = 0; k < 4; ++k)
            P[4 * i + k] = (int8_t)c[k];
    }
}

// pyramid layout and per-level quotas (cv::ORB: n_l proportional to 1.2^-l, the last level takes the remainder)
static bool orb_layout_host(int w, int h, const mvs_orb_params &p, OrbDev &d, size_t &pixels)
{
    if (p.nlevels < 1 || p.nlevels > kOrbMaxLevels || p.nfeatures < 1)
        return false;
    double s = 1.0;
    pixels = 0;
    for (int l = 0; l < p.nlevels; ++l) {
        OrbLevel &L = d.level[l];
        L.w = (int)std::lrint((double)w / s);
        L.h = (int)std::lrint((double)h / s);
        L.scale = (float)s;
        L.offset = pixels;
        pixels += (size_t)std::max(L.w, 0) * std::max(L.h, 0);
        s = s * 1.2;
    }
    const double factor = 1.0 / 1.2;
    double fn = 1.0;
    for (int l = 0; l < p.nlevels; ++l)
        fn = fn * factor;
    double nd = (double)p.nfeatures * (1.0 - factor) / (1.0 - fn);
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; ++l) {
        // the rounded shares can add up to more than nfeatures when nfeatures is small (cv::ORB then returns more
        // keypoints than asked for); the outputs have room for nfeatures: a level takes at most what is left
        d.level[l].n_keep = std::min((int)std::lrint(nd), p.nfeatures - sum);
        sum += d.level[l].n_keep;
        nd = nd * factor;
    }
    d.level[p.nlevels - 1].n_keep = std::max(p.nfeatures - sum, 0);
    return true;
}

// runs the extraction of n images (host pointer) through the ctx workspace; outputs go to the given DEVICE arrays
// (desc / kp_xy / n_kp may belong to a sequence) and, when kp_rec_out is non-null, records are also left in the workspace
// wait = false: the overflow flag's copy is queued but not waited for -- the caller queues its own downloads behind it, waits
// once and then reads *ctx->h_orb_ovf
static mvs_status orb_run(mvs_ctx *ctx, const uint8_t *images, int n, int w, int h, const mvs_orb_params &prm,
                          uint8_t *d_desc_ext, float *d_kp_xy_ext, int32_t *d_n_ext, OrbDev &d,
                          uint8_t *d_kp_oct_ext = nullptr, bool wait = true)
{
    if (w < 1 || h < 1 || w > 65535 || h > 65535)
        return MVS_ERR_CAPACITY;
    if (prm.fast_threshold < 1 || prm.fast_threshold > 254 || prm.edge_threshold < 19)
        return MVS_ERR_INVALID_ARG;
    d = OrbDev{};
    size_t T = 0;
    if (!orb_layout_host(w, h, prm, d, T))
        return MVS_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->orb_ready) {
        HIP_TRY(ctx, orb_prepare());
        ctx->orb_ready = true;
    }
    const size_t B = (size_t)n, L = (size_t)prm.nlevels, NF = (size_t)prm.nfeatures;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    size_t off = 0;
    const size_t o_pyr = off; off = up(off + T * B);
    const size_t o_blur = off; off = up(off + T * B);
    // resize tables: for every level >= 1, {source index, 11-bit weight} per destination column, then per row.  Their layout
    // is part of the key, their contents are built only when the key is new (below)
    size_t tab_entries = 0;
    for (int l = 1; l < prm.nlevels; ++l) {
        d.level[l].tab_offset = tab_entries;
        tab_entries += (size_t)std::max(d.level[l].w, 0) + (size_t)std::max(d.level[l].h, 0);
    }
    auto build_tab = [&](std::vector<int2> &tab) {
        tab.reserve(tab_entries);
        for (int l = 1; l < prm.nlevels; ++l) {
            const OrbLevel &Lv = d.level[l], &Pv = d.level[l - 1];
            for (int pass = 0; pass < 2; ++pass) {
                const long long sn = pass ? Pv.h : Pv.w, dn = pass ? Lv.h : Lv.w;
                for (long long dd = 0; dd < dn; ++dd) {
                    const long long num = (2 * dd + 1) * sn - dn, den = 2 * dn;   // source coordinate = num / den (>= 0 here)
                    long long si = num >= 0 ? num / den : -1;
                    const long long fr = num - si * den;
                    int wgt = (int)((fr * 4096 + den) / (2 * den));               // round(frac * 2048)
                    if (si < 0) si = 0, wgt = 0;
                    if (si >= sn - 1) si = sn - 1;
                    tab.push_back(make_int2((int)si, wgt));
                }
            }
        }
    };
    const size_t o_tab = off; off = up(off + std::max<size_t>(tab_entries, 1) * sizeof(int2));
    // candidate lists: a level's list holds the non-maximum suppression's own bound -- strict 3x3 maxima of the detection
    // area, at most one per 2x2 block -- so it cannot overflow; only when 2 n_l exceeds what the selection holds in LDS the
    // list is capped there (and a fuller level is reported as MVS_ERR_CAPACITY, as every level was in rounds 2-4)
    int max_keep = 0;
    for (int l = 0; l < prm.nlevels; ++l)
        max_keep = std::max(max_keep, d.level[l].n_keep);
    size_t cand_total = 0;
    for (int l = 0; l < prm.nlevels; ++l) {
        OrbLevel &Lv = d.level[l];
        const long long dw = (long long)Lv.w - 2 * prm.edge_threshold, dh = (long long)Lv.h - 2 * prm.edge_threshold;
        long long cap = dw > 0 && dh > 0 ? ((dw + 1) / 2) * ((dh + 1) / 2) + 64 : 64;
        if (2LL * max_keep > kOrbSelCap)
            cap = std::min<long long>(cap, kOrbSelCap);
        if (cap > 0x7fffffff)
            return MVS_ERR_CAPACITY;
        Lv.cand_cap = (int)cap;
        Lv.cand_off = cand_total;
        cand_total += (size_t)cap;
    }
    d.cand_stride = cand_total;
    const size_t o_keys = off; off = up(off + B * cand_total * 8);
    const size_t o_cc = off; off = up(off + B * L * 4);
    const size_t o_sel = off; off = up(off + B * L * NF * sizeof(OrbSel));
    const size_t o_sc = off; off = up(off + B * L * 4);
    const size_t o_ovf = off; off = up(off + 4);
    const size_t o_pat = off; off = up(off + 1024);
    const size_t o_kp = off; off = up(off + B * NF * sizeof(mvs_keypoint));
    const size_t o_desc = off; off = up(off + B * NF * 32);
    const size_t o_n = off; off = up(off + B * 4);
    if (ctx->orb_bytes < off) {
        if (ctx->d_orb) (void)hipFree(ctx->d_orb);
        ctx->d_orb = nullptr;
        ctx->orb_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_orb, off));
        ctx->orb_bytes = off;
    }
    char *base = static_cast<char *>(ctx->d_orb);
    d.n_images = n;
    d.n_levels = prm.nlevels;
    d.nfeatures = prm.nfeatures;
    d.edge = prm.edge_threshold;
    d.fast_threshold = prm.fast_threshold;
#ifdef MVS_DEBUG_HOOKS
    d.flat_order = std::getenv("MVS_ORB_FLAT_ORDER") != nullptr;   // A/B of the describe kernel's block order (tools/profile_extract.sh)
#endif
    d.pyr = reinterpret_cast<uint8_t *>(base + o_pyr);
    d.blur = reinterpret_cast<uint8_t *>(base + o_blur);
    d.resize_tab = reinterpret_cast<const int2 *>(base + o_tab);
    d.cand_keys = reinterpret_cast<uint64_t *>(base + o_keys);
    d.cand_count = reinterpret_cast<int32_t *>(base + o_cc);
    d.sel = reinterpret_cast<OrbSel *>(base + o_sel);
    d.sel_count = reinterpret_cast<int32_t *>(base + o_sc);
    d.overflow = reinterpret_cast<int32_t *>(base + o_ovf);
    d.pattern = reinterpret_cast<int8_t *>(base + o_pat);
    d.kp = reinterpret_cast<mvs_keypoint *>(base + o_kp);
    d.desc = d_desc_ext ? d_desc_ext : reinterpret_cast<uint8_t *>(base + o_desc);
    d.n_kp = d_n_ext ? d_n_ext : reinterpret_cast<int32_t *>(base + o_n);
    d.kp_xy = d_kp_xy_ext;
    d.kp_oct = d_kp_oct_ext;
    hipStream_t s = ctx->stream;
    // the test pattern and the resize tables depend on (size, parameters, workspace) only: a call with the key of the captured
    // graph finds them in the workspace (round 5: they were uploaded, and the stream synchronised for them, on every call)
    const bool same_key = ctx->orb_graph_valid && std::memcmp(&ctx->orb_graph_key, &d, sizeof(d)) == 0;
    int8_t pat[1024];
    std::vector<int2> tab;
    if (!same_key) {
        orb_pattern_host(pat);
        build_tab(tab);
        HIP_TRY(ctx, hipMemcpyAsync(base + o_pat, pat, sizeof(pat), hipMemcpyHostToDevice, s));
        if (!tab.empty())
            HIP_TRY(ctx, hipMemcpyAsync(base + o_tab, tab.data(), tab.size() * sizeof(int2), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(ctx, hipMemcpyAsync(d.pyr, images, (size_t)w * h * B, hipMemcpyHostToDevice, s));  // level 0 = the input
    if (!same_key)
        HIP_TRY(ctx, sync_stream(ctx));   // `pat` and `tab` live on this frame
#ifdef MVS_DEBUG_HOOKS
    static const bool no_graph = std::getenv("MVS_NO_GRAPH") != nullptr;   // A/B switch for tools/extract_bench.py (diagnostics build only)
#else
    constexpr bool no_graph = false;
#endif
    if (no_graph) {
        launch_orb(d, s);
    } else {
        if (!ctx->orb_graph_valid || std::memcmp(&ctx->orb_graph_key, &d, sizeof(d)) != 0) {
            if (ctx->orb_graph) (void)hipGraphExecDestroy(ctx->orb_graph);
            ctx->orb_graph = nullptr;
            ctx->orb_graph_valid = false;
            hipGraph_t graph = nullptr;
            HIP_TRY(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            launch_orb(d, s);
            HIP_TRY(ctx, hipStreamEndCapture(s, &graph));
            const hipError_t ie = hipGraphInstantiate(&ctx->orb_graph, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HIP_TRY(ctx, ie);
            std::memcpy(&ctx->orb_graph_key, &d, sizeof(d));
            ctx->orb_graph_valid = true;
        }
        HIP_TRY(ctx, hipGraphLaunch(ctx->orb_graph, s));
    }
    HIP_TRY(ctx, hipGetLastError());
    if (!ctx->h_orb_ovf)
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_orb_ovf), 64, hipHostMallocDefault));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_orb_ovf, d.overflow, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (!wait)
        return MVS_OK;
    HIP_TRY(ctx, sync_stream(ctx));
    return *ctx->h_orb_ovf ? MVS_ERR_CAPACITY : MVS_OK;
}

mvs_status mvs_extract_time(mvs_ctx *ctx, int steps, float *ms_total)
{
    if (!ctx || !ms_total || steps < 1)
        return MVS_ERR_INVALID_ARG;
    if (!ctx->orb_graph_valid || !ctx->orb_graph)
        return MVS_ERR_INVALID_ARG;   // nothing has been extracted on this context yet
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        ctx->err = "mvs_extract_time: hipEventCreate failed";
        return MVS_ERR_HIP;
    }
    hipStream_t s = ctx->stream;
    hipError_t e = hipEventRecord(e0, s);
    for (int i = 0; i < steps && e == hipSuccess; ++i)
        e = hipGraphLaunch(ctx->orb_graph, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms_total, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(ctx, e);
    return MVS_OK;
}

mvs_status mvs_extract(mvs_ctx *ctx, const uint8_t *images, int n_images, int width, int height,
                       const mvs_orb_params *params, mvs_keypoint *keypoints, uint8_t *descriptors, int32_t *n_keypoints)
{
    if (!ctx || !images || !params || !keypoints || !descriptors || !n_keypoints || n_images < 1)
        return MVS_ERR_INVALID_ARG;
    OrbDev d;
    // one wait per call: the outputs are queued behind the launches and the overflow flag (round 5: the flag had its own wait)
    const mvs_status st = orb_run(ctx, images, n_images, width, height, *params, nullptr, nullptr, nullptr, d, nullptr, false);
    if (st != MVS_OK)
        return st;
    const size_t B = n_images, NF = params->nfeatures;
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(keypoints, d.kp, B * NF * sizeof(mvs_keypoint), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(descriptors, d.desc, B * NF * 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(n_keypoints, d.n_kp, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, sync_stream(ctx));
    return *ctx->h_orb_ovf ? MVS_ERR_CAPACITY : MVS_OK;
}

mvs_status mvs_seq_upload_images(mvs_seq *q, int first, int count, const uint8_t *images, int width, int height,
                                 const mvs_orb_params *params, const double K[9])
{
    if (!q || !images || !params || first < 0 || count < 1 || first + count > q->n_frames)
        return MVS_ERR_INVALID_ARG;
    const BatchDev &bd = q->batch->d;
    if (bd.desc_words != 8)
        return MVS_ERR_INVALID_ARG;   // the extractor writes 256-bit descriptors
    mvs_orb_params prm = *params;
    prm.nfeatures = bd.max_kp;
    const size_t N = bd.max_kp, off = first;
    OrbDev d;
    mvs_status st = orb_run(q->ctx, images, count, width, height, prm,
                            reinterpret_cast<uint8_t *>(const_cast<uint32_t *>(bd.desc1)) + off * N * 32,
                            const_cast<float *>(bd.kp1) + off * N * 2, const_cast<int32_t *>(bd.n1) + off, d,
                            const_cast<uint8_t *>(bd.oct1) + off * N);
    if (st != MVS_OK)
        return st;
    return K ? mvs_seq_upload(q, first, count, nullptr, nullptr, nullptr, K) : MVS_OK;
}

}  // extern "C"
