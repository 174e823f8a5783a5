/*
 * mvslam_hip.h -- C ABI of libmvslam_hip.so: mvSLAM's front-end two-view geometry
 * hot path (brute-force Hamming match -> 8-point RANSAC -> essential decomposition
 * -> linear triangulation) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (lonelycorn/mvSLAM) has no FFI layer: its boundary for this path is
 * a set of C++ free functions / static methods in namespace mvSLAM.  Every entry
 * point below names the reference interface it stands in for (file:line relative
 * to the reference tree).  The C++ header shim that keeps the reference's own
 * signatures on top of this ABI lives in mvslam_amd/compat/ (see INTEGRATION.md).
 *
 * Conventions
 *   - all pointers are caller-owned HOST memory unless the name says "device";
 *   - matrices are row-major double (reference ScalarType = double, system-config.hpp:6);
 *   - every call returns an mvs_status; nothing here aborts or throws
 *     (the reference asserts on precondition violations, e.g. sfm-solve.cpp:37-41);
 *   - one mvs_ctx per (host thread, GPU); calls on one ctx are serialised by the caller
 *     (the reference path is single-threaded and not re-entrant, visual-feature.cpp:12-25).
 *   - there is NO CPU fallback: without a HIP device mvs_ctx_create fails.
 */
#ifndef MVSLAM_HIP_H
#define MVSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 4): mvs_work_stats grew by FIVE fields (max_sweeps9, dense_points, matches_mode1, score_evals_executed_mfma_rest,
 * score_evals_executed_mfma_pilot; score_evals_executed is the sum of every executed-evaluation counter); every entry point
 * and every other struct is unchanged from version 2 */
/* 4 (round 5): two entry points added -- mvs_batch_device_state (read-only diagnostics view, below) and mvs_batch_run_points
 * (a batch of sfm_solve calls on caller-supplied point pairs); nothing else changed */
#define MVS_ABI_VERSION 4

typedef enum mvs_status {
    MVS_OK = 0,
    MVS_NO_MODEL = 1,          /* the reference's `return false` (sfm-solve.cpp:319-321,330-334,353-356) */
    MVS_ERR_INVALID_ARG = -1,  /* the reference's assert()s */
    MVS_ERR_NO_DEVICE = -2,
    MVS_ERR_HIP = -3,
    MVS_ERR_CAPACITY = -4,
    MVS_ERR_BAD_INTRINSICS = -5 /* K must be affine: K[6..8] == (0, 0, 1) */
} mvs_status;

/* layout-identical to cv::DMatch (reference MatchResultType, base/image.hpp:37-48) */
typedef struct mvs_match {
    int32_t queryIdx; /* index into vf2 / pair frame */
    int32_t trainIdx; /* index into vf1 / base frame */
    int32_t imgIdx;
    float distance;
} mvs_match;

#define MVS_SAMPLER_IDENTITY 0 /* reference behaviour: the first 8 matches (estimator-RANSAC.cpp:41-48) */
#define MVS_SAMPLER_PHILOX 1   /* Philox4x32-10 keyed (seed, hypothesis id) */

/* Knobs of the path */
typedef struct mvs_params {
    double ratio;           /* Lowe ratio, 0.7 as ScalarType = double (visual-feature.cpp:23) */
    double max_dist;        /* max Hamming distance of a match, < 0 disables (image-pair.cpp:22-23: 10) */
    double max_error_sq;    /* <= 0: 5e-2 / K00 / K11 (sfm-solve.cpp:18-19,311) */
    int32_t num_hypotheses; /* reference: 1 (sfm-solve.cpp:67) */
    int32_t sampler;        /* MVS_SAMPLER_* */
    uint64_t seed;          /* hypothesis sampler key; pair p of a batch uses seed + global_index[p] */
    int32_t min_inliers;    /* 8 (sfm-solve.cpp:20-21,330) */
    int32_t reserved;
} mvs_params;

/* Fixed-size per-pair result (what ImagePair keeps after reconstruct(), image-pair.hpp:56-64) */
typedef struct mvs_pair_result {
    int32_t valid;      /* ImagePair::valid */
    int32_t n_matches;  /* M: matches surviving ratio / max_dist */
    int32_t n_inliers;  /* inliers of the winning hypothesis */
    int32_t n_points;   /* triangulated points that passed cheirality = match_inlier_count */
    int32_t best_hyp;   /* winning hypothesis id (-1: none) */
    int32_t best_count;
    double best_residual;
    double F[9];        /* winning fundamental (= essential on ideal cameras) before projection */
    double E[9];        /* after projection to (s, s, 0) (sfm-solve.cpp:74-84) */
    double R1to2[9];    /* winning candidate of decompose_essential_matrix */
    double t1to2[3];
    double R[9];        /* pose2in1 = SE3(SO3(R1to2), t1to2).inverse() (sfm-solve.cpp:364) = T_pair_to_base */
    double t[3];
} mvs_pair_result;

typedef struct mvs_ctx mvs_ctx;
typedef struct mvs_batch mvs_batch;

/* ---- context ------------------------------------------------------------------ */
int mvs_abi_version(void);
const char *mvs_status_str(mvs_status s);
const char *mvs_last_error(const mvs_ctx *ctx); /* text of the last HIP failure on this ctx */
mvs_status mvs_params_default(mvs_params *p);   /* reference defaults + num_hypotheses = 1, identity sampler */
mvs_status mvs_ctx_create(int device_id, mvs_ctx **out);
mvs_status mvs_ctx_create_on_stream(int device_id, void *hip_stream, mvs_ctx **out); /* borrow a stream */
void mvs_ctx_destroy(mvs_ctx *ctx);
void *mvs_ctx_stream(mvs_ctx *ctx); /* hipStream_t the kernels are launched on */
/* A batch of >= 64 pairs goes down the pipeline as two independent halves, the second on a context-owned side stream that is
 * forked from and joined back into the context's stream inside the call (the caller sees one stream; results are identical:
 * pairs are independent, estimator-RANSAC.cpp:76-84 runs per pair).  enable = 0 keeps every launch on the one stream.
 * Default: enabled. */
mvs_status mvs_ctx_set_half_batches(mvs_ctx *ctx, int enable);

/* ---- single-shot entry points (host buffers; the reference's call surface) ------ */

/* VisualFeature::match_visual_features(vf1 = train, vf2 = query, max_dist)
 * (vision/visual-feature.cpp:51-80, decl visual-feature.hpp:23-26).
 * desc: n x desc_bytes row-major CV_8U (desc_bytes multiple of 4, <= 64).
 * out: capacity n_query.  Output order: (distance, queryIdx) ascending.
 * MVS_ERR_INVALID_ARG when n_train < 2 or n_query < 1 (reference: assert / UB). */
mvs_status mvs_match_hamming(mvs_ctx *ctx, const uint8_t *train_desc, int n_train, const uint8_t *query_desc,
                             int n_query, int desc_bytes, double ratio, double max_dist, mvs_match *out,
                             int *n_out);

/* sfm_solve(p1, p2, K, pose2in1, points, point_indexes) (vision/sfm-solve.cpp:285-368, decl sfm.hpp:30-35)
 * with find_essential_matrix's own-RANSAC branch (sfm-solve.cpp:64-90).
 * p1_uv / p2_uv: m x (u, v) image points.  points_xyz: capacity 3*m.  point_idx: capacity m.
 * inlier_mask: capacity m (may be NULL).  result: may be NULL.
 * returns MVS_OK (true) or MVS_NO_MODEL (false). */
mvs_status mvs_two_view(mvs_ctx *ctx, const double *p1_uv, const double *p2_uv, int m, const double K[9],
                        const mvs_params *params, double R[9], double t[3], double *points_xyz,
                        int64_t *point_idx, int *n_points, uint8_t *inlier_mask, mvs_pair_result *result);

/* ImagePair::ImagePair + ImagePair::reconstruct (front-end/image-pair.cpp:30-71,116-174) of ONE pair in a single device
 * pass: match(base = train, pair = query) -> gather + normalise -> sfm_solve, one upload, one synchronisation, one
 * download (the match_visual_features + sfm_solve call pair costs two round trips).  base_kp / pair_kp: n x (x, y)
 * float = cv::KeyPoint::pt.  Outputs (any but `result` may be NULL): matches[<= n_pair] in the canonical order,
 * inlier_mask[n_matches], points_xyz[n_points x 3], point_idx[n_points] (index into matches).  Returns MVS_NO_MODEL when
 * the reference's constructor leaves `valid == false`; result->n_matches etc. are filled either way. */
mvs_status mvs_image_pair(mvs_ctx *ctx, const uint8_t *base_desc, const float *base_kp, int n_base,
                          const uint8_t *pair_desc, const float *pair_kp, int n_pair, int desc_bytes, const double K[9],
                          const mvs_params *params, mvs_pair_result *result, mvs_match *matches, uint8_t *inlier_mask,
                          double *points_xyz, int64_t *point_idx);

/* sfm_triangulate(p1, p2, K, pose1, pose2, points, point_indexes) (sfm-solve.cpp:370-394, decl sfm.hpp:47-53).
 * R1to2 / t1to2 = (pose2^-1 * pose1), composed by the caller-side shim exactly as the reference does. */
mvs_status mvs_triangulate(mvs_ctx *ctx, const double *p1_uv, const double *p2_uv, int m, const double K[9],
                           const double R1to2[9], const double t1to2[3], double *points_xyz, int64_t *point_idx,
                           int *n_points);

/* recover_pose_and_points(E, ...) + pose inverse (sfm-solve.cpp:232-284,364): the tail of sfm_solve for a
 * caller-supplied essential matrix and inlier mask (mask may be NULL = all ones).  Not public in the
 * reference; exported so the cube fixture (test/test-sfm.cpp:17-90) can pin decomposition + triangulation. */
mvs_status mvs_recover_pose(mvs_ctx *ctx, const double E[9], const double *p1_uv, const double *p2_uv, int m,
                            const double K[9], const uint8_t *inlier_mask, double R[9], double t[3],
                            double *points_xyz, int64_t *point_idx, int *n_points, mvs_pair_result *result);

/* find_fundamental_matrix(p1_sample, p2_sample, F21) (vision/fundamental-matrix.cpp:204-267, hpp:16-19).
 * p1 / p2: 8 x (x, y) ideal-camera points (homogeneous 1).  MVS_NO_MODEL for a degenerate sample. */
mvs_status mvs_find_fundamental_matrix(mvs_ctx *ctx, const double p1_xy[16], const double p2_xy[16], double F[9]);

/* FundamentalMatrixEstimatorRANSAC(max_error_sq, max_iteration).compute(p1, p2, F21, inlier_mask)
 * (vision/estimator-RANSAC.cpp:16-90, hpp:20-24).  p1 / p2: m x (x, y) ideal-camera points.
 * count / residual: optional per-hypothesis tables [num_hypotheses] (count -1 = rejected sample). */
mvs_status mvs_ransac_fundamental(mvs_ctx *ctx, const double *p1_xy, const double *p2_xy, int m,
                                  double max_error_sq, int num_hypotheses, int sampler, uint64_t seed, double F[9],
                                  uint8_t *inlier_mask, int *best_hyp, int *best_count, double *best_residual,
                                  int32_t *count, double *residual);

/* pnp_solve(world_points, image_points, K, pose, inlier_point_indexes) (vision/pnp-solve.cpp:16-104, decl pnp.hpp:22-26).
 * The reference forwards to cv::solvePnPRansac(SOLVEPNP_P3P, 100, 0.05, 0.95); this is the build's own P3P-RANSAC
 * (Grunert P3P on 3 points + 1 disambiguation point, reprojection-error inlier count over all points, first
 * hypothesis with the most inliers, no refit).  pose = camera in world, i.e.
 * SE3(R_world_to_camera, t).inverse() (pnp-solve.cpp:99-101).  n >= 7 (PNP_MIN_POINT_COUNT, :13) and n <= 4096 (the keypoint capacity).
 * inlier_idx: capacity n, ascending.  returns MVS_OK (true) / MVS_NO_MODEL (false). */
typedef struct mvs_pnp_params {
    int32_t num_hypotheses; /* reference: iterationsCount = 100 (pnp-solve.cpp:47) */
    int32_t sampler;        /* MVS_SAMPLER_* (identity = points 0..3) */
    uint64_t seed;
    double reproj_error;    /* 0.05 (pnp-solve.cpp:48), pixels of the given image points */
    int32_t min_inliers;    /* 4: the model points of the RANSAC kernel */
    int32_t refit;          /* 0 (default): return the best P3P hypothesis.  1: then minimise the reprojection error over
                               ALL inliers (pose only, points fixed), the refit cv::solvePnPRansac ends with
                               (pnp-solve.cpp:53-64); the inlier set stays the RANSAC one.  Honoured by mvs_pnp_solve and,
                               batched over all tracks on the device, by mvs_seq_run */
} mvs_pnp_params;
mvs_status mvs_pnp_params_default(mvs_pnp_params *p);
mvs_status mvs_pnp_solve(mvs_ctx *ctx, const double *world_xyz, const double *image_uv, int n, const double K[9],
                         const mvs_pnp_params *params, double R[9], double t[3], int64_t *inlier_idx, int *n_inliers,
                         int *best_hyp);

/* ---- batched, device-resident pipeline ("one image pair" = ImagePair ctor + reconstruct,
 *      front-end/image-pair.cpp:30-71,116-174, without refine()) ---------------------- */

/* capacity: n_pairs pairs, max_kp keypoints per image (<= 4096), desc_bytes per descriptor. */
mvs_status mvs_batch_create(mvs_ctx *ctx, int n_pairs, int max_kp, int desc_bytes, mvs_batch **out);
void mvs_batch_destroy(mvs_batch *b);

/* Upload pairs [first, first + count).  base = vf1 = train, pair = vf2 = query.
 * desc: count x max_kp x desc_bytes (rows >= n are ignored); kp: count x max_kp x (x, y) float (cv::KeyPoint::pt);
 * n_base / n_pair: count;  K: count x 9;  global_index: count (NULL = first + i), added to params.seed. */
mvs_status mvs_batch_upload(mvs_batch *b, int first, int count, const uint8_t *base_desc, const float *base_kp,
                            const int32_t *n_base, const uint8_t *pair_desc, const float *pair_kp,
                            const int32_t *n_pair, const double *K, const int64_t *global_index);

/* Asynchronous form: enqueues the copies on the ctx stream and returns.  The host buffers must stay valid and unchanged
 * until mvs_batch_sync; with buffers from mvs_host_alloc (pinned) the copies are true DMA transfers that overlap the
 * kernels of another batch / ctx (double buffering: upload batch k+1 while batch k runs).  K^-1 and the default indices
 * are staged in pinned memory owned by the batch. */
mvs_status mvs_batch_upload_async(mvs_batch *b, int first, int count, const uint8_t *base_desc, const float *base_kp,
                                  const int32_t *n_base, const uint8_t *pair_desc, const float *pair_kp,
                                  const int32_t *n_pair, const double *K, const int64_t *global_index);
/* pinned (page-locked) host memory for the asynchronous transfers; any entry point accepts pageable memory too */
mvs_status mvs_host_alloc(size_t bytes, void **out);
void mvs_host_free(void *p);

/* Enqueue the whole pipeline for pairs [0, n_active) on the ctx stream (asynchronous). */
mvs_status mvs_batch_run(mvs_batch *b, const mvs_params *params, int n_active);
mvs_status mvs_batch_sync(mvs_batch *b);

/* A batch of sfm_solve calls (vision/sfm.hpp:30-35; vision/sfm-solve.cpp:285-368) on caller-supplied matched image points:
 * what mvs_batch_run does behind the matcher (normalise -> 8-point RANSAC -> E -> decomposition -> triangulation), for pairs
 * [0, n_active).  uv1 / uv2: HOST memory, [n_active][max_kp][2] doubles, row k of pair p = match k in the base / pair frame;
 * m[p] = matches of pair p (0 .. max_kp; fewer than 8 -> the pair comes back invalid, estimator-RANSAC.cpp:25-29).  Uses
 * the intrinsics and sampler key offsets resident in the batch (mvs_batch_upload accepts null descriptor / keypoint pointers
 * to set only K and global_index).  The host buffers are free again when the call returns; the kernels are asynchronous on
 * the ctx stream.  Afterwards mvs_batch_download returns results / mask / points / point_idx as usual (point_idx indexes the
 * rows of uv1 / uv2) and cleared match rows. */
mvs_status mvs_batch_run_points(mvs_batch *b, const mvs_params *params, int n_active, const double *uv1, const double *uv2,
                                const int32_t *m);

/* Timed replay: `warmup` untimed + `steps` timed passes over the resident inputs, bracketed by HIP events
 * on the ctx stream.  ms_total: wall ms of the `steps` passes.  ms_kernel[5]: summed ms per kernel over the
 * timed passes, in launch order {match (match_mfma or match_topk), match_compact, ransac, finalize, reserved}; measured with
 * per-kernel events in a SEPARATE instrumented replay of `steps` passes (so ms_total has no event overhead).
 * Either output may be NULL. */
mvs_status mvs_batch_time(mvs_batch *b, const mvs_params *params, int n_active, int warmup, int steps,
                          float *ms_total, float *ms_kernel);

/* Per-launch timing of one pipeline pass: `steps` instrumented passes with a HIP event in front of every kernel launch
 * (on the ctx stream, where the kernels run).  kernel_id[k] / ms[k]: id (index into mvs_kernel_info_get) and mean
 * duration of launch k of a pass, in launch order; *n_launches: launches per pass (<= cap).  Outside any timed region. */
mvs_status mvs_batch_time_kernels(mvs_batch *b, const mvs_params *params, int n_active, int steps, int cap,
                                  int32_t *kernel_id, float *ms, int *n_launches);

/* The kernels of the two-view pipeline as they are launched for a batch of this shape: name (as rocprofv3 prints it) and
 * what the runtime reports for the code object that is actually loaded (hipFuncGetAttributes,
 * hipOccupancyMaxActiveBlocksPerMultiprocessor) -- so a bench line never carries typed-in register counts.
 * index: 0 .. n-1; returns MVS_ERR_INVALID_ARG past the end. */
typedef struct mvs_kernel_info {
    char name[96];
    char symbol[160];              /* mangled name of the code object's kernel (key into lib/kernel_resources.json) */
    int32_t kernel_id;
    int32_t threads_per_block;     /* as launched */
    int32_t num_regs;              /* hipFuncAttributes.numRegs (vector registers per lane, architected + accumulation) */
    int32_t static_lds_bytes;      /* hipFuncAttributes.sharedSizeBytes */
    int32_t dynamic_lds_bytes;     /* as launched for this batch shape */
    int32_t scratch_bytes_per_lane;/* hipFuncAttributes.localSizeBytes */
    int32_t max_threads_per_block;
    int32_t blocks_per_cu;         /* hipOccupancyMaxActiveBlocksPerMultiprocessor at that block size and LDS */
    int32_t waves_per_simd;        /* blocks_per_cu * ceil(threads / 64) / 4 SIMDs, rounded down, at least 1 if resident */
    int32_t reserved;
} mvs_kernel_info;
mvs_status mvs_kernel_info_get(mvs_ctx *ctx, int index, int max_kp, int desc_bytes, mvs_kernel_info *out);

/* Results (host).  Any pointer may be NULL.  results: count;  matches: count x max_kp;  mask: count x max_kp;
 * points: count x max_kp x 3;  point_idx: count x max_kp.  Valid rows of pair p: matches / mask [0, results[p].n_matches),
 * points / point_idx [0, results[p].n_points); every row past them is ZERO (the kernels clear the tails on every run, so
 * the whole-capacity copy is deterministic). */
mvs_status mvs_batch_download(mvs_batch *b, int first, int count, mvs_pair_result *results, mvs_match *matches,
                              uint8_t *inlier_mask, double *points_xyz, int64_t *point_idx);

/* Asynchronous form of mvs_batch_download: enqueues the copies after whatever is already on the ctx stream (no sync is
 * needed between mvs_batch_run and this call) and returns; the data is valid after mvs_batch_sync.  point_idx32 is the
 * device's native int32 index (the synchronous form widens to the reference's size_t on the host).  Row ranges as for
 * mvs_batch_download; rows past them are zero. */
mvs_status mvs_batch_download_async(mvs_batch *b, int first, int count, mvs_pair_result *results, mvs_match *matches,
                                    uint8_t *inlier_mask, double *points_xyz, int32_t *point_idx32);

/* Work statistics of the last run (for the roofline's algorithmic flop count): executed 9x9 Jacobi rotations,
 * visited 9x9 pairs, hypotheses, hypothesis x point evaluations, summed over pairs [0, n_active).
 * Collected by an instrumented replay outside any timed region. */
typedef struct mvs_work_stats {
    int64_t hypotheses;
    int64_t rotations9;
    int64_t pairs9;
    int64_t score_evals;
    int64_t matches; /* sum of M */
    int64_t inliers; /* sum of n_inliers */
    int64_t score_evals_executed; /* (hypothesis, point) evaluations the pruned counting kernels actually executed (incl. the
                                     NaN padding of a pair's last block); 0 when the batch runs on the fused kernel */
    int64_t score_evals_executed_f32; /* the part of them evaluated in single precision (pairs in mode 1) */
    int64_t exact_solves;         /* hypotheses that went through the exact (Jacobi) solve: every hypothesis of a pair in mode
                                     0, + the ones the pre-screen could not certify, + the survivors of the counting */
    int64_t prescreened;          /* hypotheses that only ever got the pre-screen's approximate F (never solved exactly) */
    int64_t pairs_mode[3];        /* pairs per mode: 0 every hypothesis exact, 1 pre-screened + single-precision counting,
                                     2 pre-screened + double-precision counting */
    int64_t score_evals_executed_mfma; /* the part of score_evals_executed done by the dense matrix-core phase (split bf16) */
    int64_t score_evals_executed_mfma_finish; /* ... and by the matrix-core finish (upper and lower bound per evaluation) */
    /* ABI 3 */
    int64_t max_sweeps9;          /* largest number of sweeps any 9x9 Jacobi SVD of the replay took (OpenCV's cap: 30; the
                                     pre-screen's bound assumes the iteration ends by its own test within it) */
    int64_t dense_points;         /* sum over the pairs in mode 1 of n1, the points the dense matrix-core phase covered */
    int64_t matches_mode1;        /* sum of M over the same pairs (dense_points / matches_mode1 = the share of a pair's matches
                                     every hypothesis is counted on before anything can be dropped) */
    int64_t score_evals_executed_mfma_rest; /* evaluations of the finish's second launch (upper counts of the rest of the list);
                                     included in score_evals_executed, not in score_evals_executed_mfma_finish */
    int64_t score_evals_executed_mfma_pilot; /* evaluations of the matrix-core pilot (the first 1024 hypotheses of every pair in
                                     mode 1 on every match, both bounds): included in score_evals_executed */
} mvs_work_stats;
mvs_status mvs_batch_stats(mvs_batch *b, const mvs_params *params, int n_active, mvs_work_stats *out);

/* Diagnostics: a read-only, opaque view of the batch's device-resident state (the library's internal table of device pointers
 * and capacities, prefixed by its size and the ABI version).  Nothing is launched, copied on the device or modified; call
 * mvs_batch_sync first.  Its one consumer is the audit of libmvslam_hip_dbg.so (same sources, same process), which replays
 * every hypothesis exactly and checks the decisions THIS library's kernels left in device memory
 * (tests/audit_gpu_check.py).  dst == NULL: *size receives the number of bytes needed. */
mvs_status mvs_batch_device_state(mvs_batch *b, void *dst, size_t capacity, size_t *size);

/* Device pointer + pitch of the fixed-size result records (mvs_pair_result[n_pairs]) so a caller can hand them
 * to a collective (RCCL all-gather of poses) without a host round trip. */
mvs_status mvs_batch_results_device(mvs_batch *b, void **dev_ptr, size_t *record_bytes);

/* Asynchronous device-to-device copy (on the ctx stream) of the result records of pairs [first, first + count)
 * into caller-owned DEVICE memory, e.g. a torch tensor that is then all-gathered over RCCL. */
mvs_status mvs_batch_copy_results_device(mvs_batch *b, int first, int count, void *dst_device);
/* The one exchange step of the sharded path (pairs are independent: front-end/image-pair.hpp:56-57): one
 * ncclAllGather over RCCL / xGMI of the records of pairs [0, n_active), enqueued on the ctx stream after the batch's
 * kernels.  rccl_comm: the caller's ncclComm_t (one rank per GPU); dst_device: world_size x n_active x
 * sizeof(mvs_pair_result) bytes of device memory, rank-major.  Asynchronous; librccl.so is loaded on first use. */
mvs_status mvs_batch_gather_results(mvs_batch *b, int n_active, void *rccl_comm, void *dst_device);

/* ---- frame sequences, device resident -------------------------------------------------
 * What VisualOdometer::add_frame chains per frame (front-end/visual-odometer.cpp:129-194,384-445,502-615):
 * ImagePair(prev, new) and track_pnp on the 3-D points the previous pair triangulated.  Every frame is uploaded ONCE;
 * pair k = (base = frame k, pair = frame k + 1) is a zero-copy view into the frame arrays (frame k is `pair` of pair
 * k-1 and `base` of pair k).  Track q (q = 0 .. n_frames-3) joins, on the device, the points of pair q (expressed in
 * frame q's camera) to their observations in frame q + 2 through pair q + 1's matches (the vf-index join of
 * visual-odometer.cpp:528-556) and runs pnp_solve on them: pose of frame q + 2 in frame q's camera frame.
 * The VO state machine itself (initialisation gates, scale propagation, BA) stays on the host / out of scope. */
typedef struct mvs_seq mvs_seq;
typedef struct mvs_track_result {
    int32_t ok;        /* pnp_solve returned true */
    int32_t n_corr;    /* 3-D / 2-D correspondences found by the join */
    int32_t n_inliers;
    int32_t best_hyp;
    double R[9];       /* pose of frame q + 2 in frame q's camera frame (pnp-solve.cpp:99-101 convention) */
    double t[3];
} mvs_track_result;

mvs_status mvs_seq_create(mvs_ctx *ctx, int n_frames, int max_kp, int desc_bytes, mvs_seq **out);
void mvs_seq_destroy(mvs_seq *s);
/* frames [first, first + count): desc count x max_kp x desc_bytes, kp count x max_kp x 2 float, n_kp count; one camera K */
mvs_status mvs_seq_upload(mvs_seq *s, int first, int count, const uint8_t *desc, const float *kp, const int32_t *n_kp,
                          const double K[9]);
/* all pairs (batched two-view pipeline) + all tracks (join + batched PnP), asynchronous on the ctx stream */
mvs_status mvs_seq_run(mvs_seq *s, const mvs_params *two_view, const mvs_pnp_params *pnp);
mvs_status mvs_seq_sync(mvs_seq *s);
/* `steps` timed passes after `warmup`; ms_total = wall ms of the timed passes (HIP events on the ctx stream) */
mvs_status mvs_seq_time(mvs_seq *s, const mvs_params *two_view, const mvs_pnp_params *pnp, int warmup, int steps,
                        float *ms_total);
/* per-stage HIP-event timing of the sequence step (the kernels' own stream): ms_stage[4] = summed ms over `steps`
 * instrumented passes of {pair pipeline, join, pnp_solve, scale propagation} */
mvs_status mvs_seq_time_stages(mvs_seq *s, const mvs_params *two_view, const mvs_pnp_params *pnp, int steps, float *ms_stage);
/* pairs [first, first + count) of the n_frames - 1 pairs: same layout as mvs_batch_download */
mvs_status mvs_seq_download_pairs(mvs_seq *s, int first, int count, mvs_pair_result *results, mvs_match *matches,
                                  uint8_t *inlier_mask, double *points_xyz, int64_t *point_idx);
/* tracks [first, first + count) of the n_frames - 2 tracks.  corr_xyz / corr_uv: count x max_kp x 3 / 2 (the joined
 * correspondences, may be NULL); inlier_idx: count x max_kp (indices into the correspondences, may be NULL) */
mvs_status mvs_seq_download_tracks(mvs_seq *s, int first, int count, mvs_track_result *tracks, double *corr_xyz,
                                   double *corr_uv, int64_t *inlier_idx);

/* ---------------------------------------------------------------------------------------------------------------
 * Row f4: refinement.  Replaces sfm_refine (vision/sfm.hpp:56-76, sfm-refine.cpp:20-139) and
 * pnp_refine (vision/pnp.hpp:28-46, pnp-refine.cpp:14-108), i.e. the two callers of ba_frame_pose_and_point
 * (vision/ba.cpp:26-156, GTSAM LevenbergMarquardtOptimizer + Marginals).  The library minimises the same cost
 *   1/2 [ pose priors + point priors + reprojection residuals, each in its Mahalanobis norm ]
 * with its own batched Schur-complement Levenberg-Marquardt kernel and returns the minimiser, the marginal covariances
 * of the linearised problem (ba.cpp:127,141,152) and the final error (ba.cpp:155).  Pose tangent order is GTSAM's:
 * (rotation, translation), right perturbation.  DESIGN.md section 4.7. */
typedef struct mvs_refine_params {
    int32_t max_iterations;  /* 100 = gtsam::LevenbergMarquardtParams default */
    int32_t reserved;
    double lambda_initial;   /* 1e-5 */
    double lambda_factor;    /* 10 */
    double lambda_upper;     /* 1e5 */
    double rel_tol;          /* 1e-12: stop when the error decrease is below rel_tol * error ... */
    double abs_tol;          /* 1e-12: ... or below abs_tol (GTSAM's defaults are 1e-5 / 1e-5) */
    double anchor_sigma[2];  /* sfm-refine.cpp:11-14: prior on camera 1, diagonal entries {0-2, 3-5} = {1e-5, 1e-5} */
    double pose_sigma[2];    /* sfm-refine.cpp:15-18, pnp-refine.cpp:9-12: regulator on the moving camera {1e-2, 1e-2} */
    double point_sigma;      /* sfm-refine.cpp:86-94: regulator on every point, 1e-2 */
} mvs_refine_params;
void mvs_refine_params_default(mvs_refine_params *p);

typedef struct mvs_refine_result {
    int32_t ok;          /* 1: converged or stopped by the iteration / lambda limits with a finite error */
    int32_t iterations;  /* linear solves performed */
    double error;        /* 1/2 sum of squared Mahalanobis residuals at the estimate (optimizer.error()) */
    double R[9];         /* the moving camera in the world frame (sfm: pose2in1) */
    double t[3];
    double pose_cov[36]; /* its marginal covariance, row-major 6 x 6, order (rotation, translation) */
} mvs_refine_result;

/* sfm_refine.  p1 / p2: m x 2 image points (host), cov1 / cov2: m x 4 (row-major 2 x 2 covariances) or NULL = identity,
 * K affine, (R_guess, t_guess) = pose2in1 guess, points_guess m x 3 in camera 1.  points_out: m x 3, point_cov_out:
 * m x 9 (may be NULL).  1 <= m <= 4096.  MVS_NO_MODEL if the problem could not be solved. */
mvs_status mvs_sfm_refine(mvs_ctx *ctx, const double *p1, const double *cov1, const double *p2, const double *cov2, int m,
                          const double K[9], const double R_guess[9], const double t_guess[3],
                          const double *points_guess, const mvs_refine_params *params, mvs_refine_result *result,
                          double *points_out, double *point_cov_out);
/* pnp_refine.  world: m x 3 with covariances world_cov m x 9 (the point priors), image points m x 2 with covariances
 * image_cov m x 4 or NULL = identity; (R_guess, t_guess) = camera in world. */
mvs_status mvs_pnp_refine(mvs_ctx *ctx, const double *world, const double *world_cov, const double *image,
                          const double *image_cov, int m, const double K[9], const double R_guess[9],
                          const double t_guess[3], const mvs_refine_params *params, mvs_refine_result *result);
/* ba_frame_pose_and_point (vision/ba.hpp:25-36, ba.cpp:26-156) for the configurations the reference builds -- one or two
 * frames: besides sfm_refine / pnp_refine that is VisualOdometer::track_refine (front-end/visual-odometer.cpp:618-800:
 * last frame anchored at ITS pose, new frame regularised, tracked points with priors, new points without, each frame
 * observing a subset of the points).  All pointers are host memory. */
typedef struct mvs_ba_problem {
    int32_t n_frames;               /* 1 or 2 */
    int32_t n_points;               /* 1 .. 4096 */
    const double *K;                /* 9, affine */
    const double *frame_pose;       /* n_frames x 12: R (9 row-major), t (3): camera in world = guess = prior mean */
    const double *frame_prior_var;  /* n_frames x 6: DIAGONAL of the prior covariance in tangent order (rotation,
                                       translation); an entry <= 0 = no prior on that coordinate */
    const double *points;           /* n_points x 3 guesses (= prior means) */
    const double *point_prior_cov;  /* n_points x 9, or NULL = no point has a prior; a point whose covariance has a
                                       first entry <= 0 has no prior (track_refine's new points) */
    const double *obs[2];           /* per frame: n_points x 2 image points */
    const double *obs_cov[2];       /* per frame: n_points x 4 covariances, or NULL = identity */
    const uint8_t *obs_valid[2];    /* per frame: n_points flags (0 = the frame does not observe the point), or NULL = all */
} mvs_ba_problem;
/* frames_out[n_frames]: pose estimate and marginal covariance of every frame (error / iterations repeated in each);
 * points_out n_points x 3, point_cov_out n_points x 9 (may be NULL).  Only the LM fields of params are used. */
mvs_status mvs_ba_refine(mvs_ctx *ctx, const mvs_ba_problem *problem, const mvs_refine_params *params,
                         mvs_refine_result *frames_out, double *points_out, double *point_cov_out);
/* Batched ImagePair::refine (front-end/image-pair.cpp:176-238) on a batch that has been run: every valid pair is refined
 * from its own results on the device.  Observations = the matched keypoints with the covariance
 * VisualFeature::get_point_estimates gives them (vision/visual-feature.cpp:192-207): stddev = (1 << kp.octave) * 0.5 px,
 * i.e. (sigma_px * 2^octave)^2 I with sigma_px = 0.5.  The octave of every keypoint is resident next to its
 * coordinates: written by the device extractor (mvs_seq_upload_images), supplied by the caller for host-uploaded
 * keypoints (mvs_batch_upload_octaves / mvs_seq_upload_octaves), 0 otherwise.
 * Asynchronous on the ctx stream; results stay resident until downloaded. */
mvs_status mvs_batch_refine(mvs_batch *b, const mvs_refine_params *params, double sigma_px);
/* cv::KeyPoint::octave of the keypoints uploaded with mvs_batch_upload: count x max_kp bytes per image (NULL = leave);
 * values above 30 are rejected (MVS_ERR_INVALID_ARG).  Only mvs_batch_refine reads them. */
mvs_status mvs_batch_upload_octaves(mvs_batch *b, int first, int count, const uint8_t *base_octave,
                                    const uint8_t *pair_octave);
/* refined[n_pairs]; points_xyz / point_cov: n_pairs x max_kp x 3 / 9 (NULL to skip), rows [0, results[p].n_points) */
mvs_status mvs_batch_download_refined(mvs_batch *b, mvs_refine_result *refined, double *points_xyz, double *point_cov);
/* the same for the n_frames - 1 consecutive pairs of a sequence that has been run (VisualOdometer::initialize refines its
 * queued pairs, front-end/visual-odometer.cpp:282-286) */
mvs_status mvs_seq_refine_pairs(mvs_seq *s, const mvs_refine_params *params, double sigma_px);
/* octaves of the keypoints of frames [first, first + count) uploaded with mvs_seq_upload: count x max_kp bytes */
mvs_status mvs_seq_upload_octaves(mvs_seq *s, int first, int count, const uint8_t *octave);
mvs_status mvs_seq_download_refined(mvs_seq *s, mvs_refine_result *refined, double *points_xyz, double *point_cov);

/* ---------------------------------------------------------------------------------------------------------------
 * Row f3: keypoint + descriptor extraction.  Replaces VisualFeature::extract
 * (vision/visual-feature.cpp:40-49, decl visual-feature.hpp:16-20) = cv::ORB::create(500)->detect + compute
 * (visual-feature.cpp:12-17).  cv::ORB is OpenCV-internal and its learned sampling pattern is not in the reference
 * tree: this is ORB's published pipeline with the reference's parameters and the library's own fully specified
 * resize / blur / ranking / pattern -- keypoints and descriptors are NOT bit-identical to
 * OpenCV's, they are bit-identical to the CPU oracle's. */
typedef struct mvs_orb_params {
    int32_t nfeatures;       /* 500 = MAX_FEATURE_COUNT (visual-feature.cpp:9) */
    int32_t nlevels;         /* 8, scale factor 1.2 (cv::ORB::create defaults) */
    int32_t edge_threshold;  /* 31 */
    int32_t fast_threshold;  /* 20 */
} mvs_orb_params;
typedef struct mvs_keypoint { /* layout of cv::KeyPoint (base/image.hpp:37-48 DetectorResultType element) */
    float x, y, size, angle, response;
    int32_t octave, class_id;
} mvs_keypoint;
void mvs_orb_params_default(mvs_orb_params *p);
/* images: n_images x height x width grayscale (CV_8UC1, continuous), host.  keypoints: n_images x nfeatures,
 * descriptors: n_images x nfeatures x 32, n_keypoints: n_images (rows [0, n_keypoints[i]) are valid; ordered by
 * pyramid level, then response descending).  MVS_ERR_CAPACITY if the image is larger than 65535 in a dimension, or -- only
 * when a level's quota exceeds 8192 features (the selection holds 2 n_l <= 16384 keys in LDS) -- if that level has more than
 * 16384 corners; otherwise a level's candidate list holds every corner the non-maximum suppression can leave (round 5). */
mvs_status mvs_extract(mvs_ctx *ctx, const uint8_t *images, int n_images, int width, int height,
                       const mvs_orb_params *params, mvs_keypoint *keypoints, uint8_t *descriptors,
                       int32_t *n_keypoints);
/* Kernel time of the extraction: replays the launches of the LAST mvs_extract / mvs_seq_upload_images on this context
 * (its captured graph; the workspace still holds that call's images) `steps` times between HIP events on the ctx stream.
 * No transfers are inside the measurement.  MVS_ERR_INVALID_ARG before the first extraction. */
mvs_status mvs_extract_time(mvs_ctx *ctx, int steps, float *ms_total);
/* Extraction straight into a sequence's resident frame arrays (no host round trip of descriptors): frames
 * [first, first + count) of `s` get up to max_kp keypoints each (params->nfeatures is overridden by max_kp). */
mvs_status mvs_seq_upload_images(mvs_seq *s, int first, int count, const uint8_t *images, int width, int height,
                                 const mvs_orb_params *params, const double K[9]);

/* Scale propagation and trajectory of a sequence that has been run (the last stage of mvs_seq_run; row f2,
 * front-end/visual-odometer.cpp:422-445,577-588).  Pair k has a unit baseline; track q is in pair q's scale;
 *   track_scale[q] = |(pair_q^-1 o track_q).t| = baseline(pair q+1) / baseline(pair q)      (n_frames - 2 entries)
 *   pair_scale[k]  = prod_{j<k} track_scale[j] = pair k's baseline in units of pair 0's      (n_frames - 1 entries)
 *   R / t          = pose of frame k in frame 0: G_0 = I, G_1 = pair 0, G_{q+2} = G_q o (R_track_q, pair_scale[q] t_track_q)
 * (a failed track keeps the scale and falls back to the two-view pose of pair q+1).  Any pointer may be NULL. */
mvs_status mvs_seq_download_trajectory(mvs_seq *s, double *R, double *t, double *pair_scale, double *track_scale);

#ifdef __cplusplus
}
#endif
#endif /* MVSLAM_HIP_H */
