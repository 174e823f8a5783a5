/*
 * mvs_oracle.h -- CPU ORACLE for the mvSLAM two-view-geometry hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (mvslam_amd/, include/) never links, imports or executes anything under oracle/.
 *
 * It is a plain-C restatement (no copied source) of the reference algorithm, each
 * function citing the reference file:line it follows (paths relative to the
 * reference tree).  Arithmetic that the reference delegates to OpenCV
 * (cv::SVDecomp, cv::BFMatcher::knnMatch; OpenCV >= 3.0, version unpinned by the
 * reference, README.md:12) is restated from OpenCV's published algorithm
 * (one-sided Hestenes/Jacobi SVD, modules/core/src/lapack.cpp JacobiSVDImpl_;
 * brute-force k-NN insertion rule, modules/core/src/batch_distance.cpp).
 *
 * PARITY PINNING: the oracle is pinned by the reference's own known-answer tests
 * that are reachable without OpenCV/Eigen/GTSAM (test/test-svd.cpp,
 * test/test-sfm.cpp sfm_triangulate_cube + decomposition of the cube's analytic E,
 * test/test-lie-group.cpp, test/test-camera.cpp) -- see tests/test_oracle_kat.py.
 * The 8-point RANSAC estimator itself is PARITY-UNPINNED by the reference (its
 * only fixture is a degenerate configuration and the default build bypasses it,
 * SURVEY.md section 0); it is pinned here by analytic ground truth (L-shape rig)
 * and LAPACK cross-checks.
 *
 * ARITHMETIC CONTRACT (shared, by specification only, with the HIP kernels):
 * IEEE-754 binary64, round-to-nearest-even, no contraction except where fma()
 * is written explicitly, no reassociation.  Build with -ffp-contract=off.
 */
#ifndef MVS_ORACLE_H
#define MVS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout of cv::DMatch (base/image.hpp:37-48 -> MatchResultType) */
typedef struct {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} orc_match;

#define ORC_SAMPLER_IDENTITY 0 /* reference behaviour: index[0..7] (estimator-RANSAC.cpp:41-48) */
#define ORC_SAMPLER_PHILOX 1   /* Philox4x32-10 keyed (seed, hypothesis id) */

typedef struct {
    int64_t rotations9; /* executed 9x9 Jacobi rotations */
    int64_t pairs9;     /* visited 9x9 (i,j) pairs (rotated or skipped) */
    int64_t rotations3;
    int64_t pairs3;
    int64_t rotations4;
    int64_t pairs4;
    int64_t hypotheses;
    int64_t score_evals; /* hypothesis x point residual evaluations */
} orc_counters;

void orc_counters_reset(void);
void orc_counters_get(orc_counters *out);

/* ---- math/lie-group.{hpp,cpp} ---- */
void orc_so3_rectify(double R[9]);                              /* lie-group.hpp:84-96 */
void orc_so3_from_matrix(const double M[9], double R[9]);       /* lie-group.hpp:31-36 */
void orc_so3_from_rpy(double roll, double pitch, double yaw, double R[9]); /* :41-56 */
void orc_so3_ln(const double R[9], double w[3]);                /* lie-group.hpp:138-162 */
void orc_rodrigues(const double v[3], double R[9]);             /* lie-group.cpp:15-32 */
void orc_se3_inverse(const double R[9], const double t[3], double Ro[9], double to[3]); /* hpp:212-216 */
void orc_se3_compose(const double Ra[9], const double ta[3], const double Rb[9], const double tb[3],
                     double Ro[9], double to[3]);               /* hpp:229-234 */
void orc_se3_ln(const double R[9], const double t[3], double se3[6]);  /* hpp:245-269 */
void orc_se3_exp(const double se3[6], double R[9], double t[3]);       /* hpp:275-299 */

/* ---- math/svd.hpp:59-72 == cv::SVDecomp(A, w, u, vt, MODIFY_A | FULL_UV) ---- */
/* A: m x n row-major. w: min(m,n). u: m x m. vt: n x n. */
void orc_svd(const double *A, int m, int n, double *w, double *u, double *vt);

/* ---- vision/camera.cpp ---- */
void orc_mat3_inverse(const double K[9], double Kinv[9]);                 /* camera.cpp:16 */
void orc_normalize_points(const double Kinv[9], const double *uv, int n, double *xy); /* :55-79 */
int orc_project_point(const double K[9], const double Rw2c[9], const double tw2c[3],
                      const double X[3], double uv[2]);                   /* camera.cpp:24-37 */

/* ---- vision/visual-feature.cpp:51-80 ---- */
/* returns number of matches written to out (<= n_query), or -1 on precondition failure */
int orc_match_visual_features(const uint8_t *train_desc, int n_train, const uint8_t *query_desc, int n_query,
                              int desc_bytes, double ratio, double max_dist, orc_match *out);

/* ---- vision/fundamental-matrix.cpp:204-267 ---- */
/* p1, p2: 8 points (x, y) with implicit homogeneous 1.  returns 1 on success. */
int orc_find_fundamental_matrix(const double p1[16], const double p2[16], double F[9]);

/* sampler (the reference has none: estimator-RANSAC.cpp:41-42) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_sample8(uint64_t seed, uint32_t hyp, int M, int sampler, int idx[8]);

/* ---- vision/estimator-RANSAC.cpp:16-129 ---- */
/* p1, p2: M x (x, y).  mask: M bytes.  returns 1 if best_count > 0.
 * per_hyp_count / per_hyp_residual: optional [H] tables (may be NULL). */
int orc_ransac_fundamental(const double *p1, const double *p2, int M, double max_error_sq, int H, int sampler,
                           uint64_t seed, double F[9], uint8_t *mask, int *best_hyp, int *best_count,
                           double *best_residual, int32_t *per_hyp_count, double *per_hyp_residual);
/* study switch, see mvs_oracle.c; 0 = the contract's fused residual (default) */
void orc_set_residual_form(int form);
void orc_set_jacobi_form(int form);
int orc_count_inliers(const double *p1, const double *p2, int M, const double F[9], double max_error_sq,
                      uint8_t *mask, double *residual); /* estimator-RANSAC.cpp:100-129 */

/* ---- vision/sfm-solve.cpp ---- */
void orc_project_essential(const double F[9], double E[9]);             /* sfm-solve.cpp:74-84 */
void orc_decompose_essential(const double E[9], double Ra[9], double Rb[9], double t[3]); /* :97-127 */
int orc_triangulate_points(const double R[9], const double t[3], const double *p1, const double *p2, int M,
                           const uint8_t *mask, double *points, int64_t *idx); /* :134-227 */
int orc_recover_pose_and_points(const double E[9], const double *p1, const double *p2, int M,
                                const uint8_t *mask, double R[9], double t[3], double *points,
                                int64_t *idx, int *n_points);             /* :232-284 */

typedef struct {
    double max_error_sq; /* <= 0: derive 5e-2 / K00 / K11 (sfm-solve.cpp:18-19,311) */
    int32_t num_hypotheses;
    int32_t sampler;
    uint64_t seed;
    int32_t min_inliers; /* 8 (sfm-solve.cpp:20-21) */
} orc_params;

typedef struct {
    int32_t valid;
    int32_t n_matches; /* M */
    int32_t n_inliers;
    int32_t n_points;
    int32_t best_hyp;
    int32_t best_count;
    double best_residual;
    double F[9];
    double E[9];
    double R1to2[9];
    double t1to2[3];
    double R[9]; /* pose2in1 = SE3(SO3(R1to2), t1to2).inverse() */
    double t[3];
} orc_two_view_result;

/* sfm_solve (sfm-solve.cpp:285-368).  uv1/uv2: M x (u, v) image points.  returns 1 (true) / 0. */
int orc_sfm_solve(const double *uv1, const double *uv2, int M, const double K[9], const orc_params *prm,
                  orc_two_view_result *res, uint8_t *mask, double *points, int64_t *idx);

/* sfm_triangulate (sfm-solve.cpp:370-394); poses are camera-in-world. returns n_points */
int orc_sfm_triangulate(const double *uv1, const double *uv2, int M, const double K[9], const double R1[9],
                        const double t1[3], const double R2[9], const double t2[3], double *points,
                        int64_t *idx);

/* ImagePair ctor + reconstruct (front-end/image-pair.cpp:30-71,116-174): match(base=train, pair=query)
 * -> gather keypoints -> sfm_solve.  kp: N x (x, y) float.  matches: capacity n_pair. */
int orc_image_pair(const uint8_t *base_desc, const float *base_kp, int n_base, const uint8_t *pair_desc,
                   const float *pair_kp, int n_pair, int desc_bytes, double ratio, double max_dist,
                   const double K[9], const orc_params *prm, orc_match *matches, orc_two_view_result *res,
                   uint8_t *mask, double *points, int64_t *idx);

/* ---- vision/pnp-solve.cpp:16-104 (row f1 of SURVEY section 8) ----
 * The reference forwards to cv::solvePnPRansac(SOLVEPNP_P3P, 100 iterations, reprojectionError 0.05, confidence 0.95)
 * and inverts the pose (:99-101).  OpenCV's RANSAC kernel / RNG / final EPnP refit are third-party and changed
 * across the 3.x versions the reference admits, so this is the BUILD'S OWN P3P-RANSAC (parity-unpinned by OpenCV,
 * pinned by the reference's pnp_solve_cube test and analytic ground truth):
 *   sample 4 distinct points (3 for Grunert's P3P, the 4th picks among its <= 4 solutions), score by the
 *   division-free reprojection test  fx^2 dx^2 + fy^2 dy^2 <= err^2 zc^2, zc > 0  on all points, keep the first
 *   hypothesis with the most inliers (cv::RANSACPointSetRegistrator's rule), no refit.
 * Only + - * / sqrt are used, so the GPU path can match bit for bit. */
typedef struct {
    int32_t num_hypotheses; /* reference: iterationsCount = 100 (pnp-solve.cpp:47) */
    int32_t sampler;        /* ORC_SAMPLER_* (identity = points 0..3) */
    uint64_t seed;
    double reproj_error;    /* 0.05 (pnp-solve.cpp:48), in pixels of the given image points */
    int32_t min_inliers;    /* 4 = model points of the P3P RANSAC kernel */
} orc_pnp_params;

void orc_sample4(uint64_t seed, uint32_t hyp, int n, int sampler, int idx[4]);
/* f: 3 unit bearings (row-major 3x3), X: 3 world points; R/t: up to 4 solutions of R X + t = s f. returns count */
int orc_p3p(const double f[9], const double X[9], double R[4][9], double t[4][3]);
/* returns 1 on success.  R, t: pose of the camera in the world frame (= SE3(R_w2c, t_w2c).inverse(), :101).
 * inlier_idx: capacity n, ascending.  Rw2c / tw2c / best_hyp may be NULL. */
int orc_pnp_solve(const double *world_xyz, const double *image_uv, int n, const double K[9],
                  const orc_pnp_params *prm, double R[9], double t[3], int64_t *inlier_idx, int *n_inliers,
                  double Rw2c[9], double tw2c[3], int *best_hyp);

/* scale propagation + trajectory of a sequence (row f2; front-end/visual-odometer.cpp:422-445,577-588): see the .c file.
 * pair_*: n_frames - 1 entries, track_*: n_frames - 2; outputs traj_R / traj_t: n_frames, pair_scale: n_frames - 1,
 * track_scale: n_frames - 2 */
void orc_seq_chain(int n_frames, const double *pair_R, const double *pair_t, const int32_t *pair_valid,
                   const double *track_R, const double *track_t, const int32_t *track_ok, double *traj_R, double *traj_t,
                   double *pair_scale, double *track_scale);

/* diagnostics: trace of the 9x9 Jacobi decompositions (one word per sweep: bit = pair rotated; ~0 terminates one SVD) */
void orc_debug_set_jacobi_trace(uint64_t *buf, size_t cap);
size_t orc_debug_jacobi_trace_len(void);

/* ---- vision/sfm-refine.cpp:20-139, vision/pnp-refine.cpp:14-108 -> vision/ba.cpp:26-156 (row f4 of SURVEY section 8)
 * Restated in mvs_refine_oracle.c (see its header: the least-squares problem GTSAM is handed, solved by the build's own
 * Schur-complement Levenberg-Marquardt; parity by tolerance). */
typedef struct {
    int32_t max_iterations;  /* 100 = gtsam::LevenbergMarquardtParams default (ba.cpp:125 uses the defaults) */
    int32_t reserved;
    double lambda_initial;   /* 1e-5, GTSAM default */
    double lambda_factor;    /* 10 */
    double lambda_upper;     /* 1e5 */
    double rel_tol;          /* 1e-12 (GTSAM: 1e-5; tighter so that the result is the minimiser itself) */
    double abs_tol;          /* 1e-12 (GTSAM: 1e-5) */
    double anchor_sigma[2];  /* sfm-refine.cpp:11-14: camera-1 prior, {diag 0-2, diag 3-5} = {1e-5, 1e-5} */
    double pose_sigma[2];    /* sfm-refine.cpp:15-18 / pnp-refine.cpp:9-12: moving-camera regulator {1e-2, 1e-2} */
    double point_sigma;      /* sfm-refine.cpp:15-16,86-94: point regulator 1e-2 */
} orc_refine_params;
void orc_refine_params_default(orc_refine_params *p);
/* p1/p2: m x 2 image points; cov1/cov2: m x 4 (2x2 row-major) or NULL = identity; points_guess: m x 3 in camera 1.
 * outputs: pose2in1 (R, t), its 6x6 marginal covariance (tangent order rotation, translation), refined points m x 3,
 * their 3x3 marginal covariances m x 9, error = optimizer.error().  returns 1 on success. */
int orc_sfm_refine(const double *p1, const double *cov1, const double *p2, const double *cov2, int m, const double K[9],
                   const double R_guess[9], const double t_guess[3], const double *points_guess,
                   const orc_refine_params *prm, double R[9], double t[3], double pose_cov[36], double *points,
                   double *point_cov, double *error, int *iterations);
/* world: m x 3 with covariances m x 9 (the point priors, pnp-refine.cpp:60-64); pose = camera in world. */
int orc_pnp_refine(const double *world, const double *world_cov, const double *img, const double *img_cov, int m,
                   const double K[9], const double R_guess[9], const double t_guess[3], const orc_refine_params *prm,
                   double R[9], double t[3], double pose_cov[36], double *error, int *iterations);

/* the general one / two frame problem (VisualOdometer::track_refine, front-end/visual-odometer.cpp:618-800).  frame_pose:
 * n_frames x 12 (R, t); frame_prior_var: n_frames x 6 (<= 0: none); point_prior_cov: m x 9 or NULL (first entry <= 0: no
 * prior); obs / obs_cov / obs_valid per frame.  outputs per frame: R (9), t (3), pose_cov (36). */
int orc_ba_refine(int n_frames, int m, const double K[9], const double *frame_pose, const double *frame_prior_var,
                  const double *points_guess, const double *point_prior_cov, const double *const obs[2],
                  const double *const obs_cov[2], const uint8_t *const obs_valid[2], const orc_refine_params *prm,
                  double *R_out, double *t_out, double *pose_cov_out, double *points, double *point_cov, double *error,
                  int *iterations);

/* ---- vision/visual-feature.cpp:12-17,40-49: VisualFeature::extract = cv::ORB detect + compute (row f3 of SURVEY
 * section 8).  Restated in mvs_orb_oracle.c -- PARITY UNPINNED by the reference (OpenCV-internal algorithm and learned
 * pattern); see that file's header for what is ORB's published pipeline and what is this build's own choice. */
typedef struct {
    int32_t nfeatures;       /* 500 = MAX_FEATURE_COUNT, visual-feature.cpp:9 */
    int32_t nlevels;         /* 8 */
    int32_t edge_threshold;  /* 31 */
    int32_t fast_threshold;  /* 20 */
} orc_orb_params;
typedef struct { /* layout of cv::KeyPoint */
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;
void orc_orb_params_default(orc_orb_params *p);
void orc_orb_pattern(int8_t P[256 * 4]); /* (x1, y1, x2, y2) per test */
int orc_orb_layout(int w, int h, const orc_orb_params *p, int *lw, int *lh, int *nl, double *scale);
void orc_orb_resize(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);
void orc_orb_fast_scores(const uint8_t *img, int w, int h, int threshold, uint8_t *score);
void orc_orb_blur(const uint8_t *img, int w, int h, uint8_t *out);
float orc_orb_harris(const uint8_t *img, int w, int x0, int y0);
void orc_orb_moments(const uint8_t *img, int w, int x0, int y0, int *m10, int *m01);
float orc_orb_fast_atan2(float y, float x);
/* image: h x w row-major grayscale.  kp / desc: capacity nfeatures (x 32 bytes).  returns 1 on success */
int orc_orb_extract(const uint8_t *image, int w, int h, const orc_orb_params *prm, orc_keypoint *kp, uint8_t *desc,
                    int *n_out);

#ifdef __cplusplus
}
#endif
#endif
