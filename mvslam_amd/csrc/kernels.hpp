// kernels.hpp -- device data layout + kernel launch interface (host side sees only PODs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvslam_hip.h"

namespace mvs {

constexpr int kMaxKp = 4096;          // capacity limit of one image (LDS lists in finalize/compact)
constexpr int kSortBins = kMaxKp + 1;  // counts 0 .. kMaxKp (ransac_list_sort_kernel)
constexpr int kHypPerBlock = 256;     // hypotheses per RANSAC workgroup (one per lane, 4 waves)
constexpr int kMaxDescWords = 16;     // descriptor <= 64 bytes
constexpr int kHypRec = 10;           // doubles per hypothesis record: F[9], counting threshold (thr + band)
constexpr int kHypRec32 = 12;         // floats per single-precision record (mode 1): F~[9], tu, tl, spare

// Best hypothesis of one RANSAC workgroup.  count < 0: no valid hypothesis in the group.
struct WgBest {
    int32_t count;
    uint32_t hyp;
    double residual;
    double F[9];
};
static_assert(sizeof(WgBest) == 88, "WgBest layout");

// What finalize_model hands to triangulate / finalize_select for one pair.
struct FinModel {
    double R[2][9];   // Ra, Rb (raw)
    double Rr[2][9];  // rectified (SO3 ctor) -> P2
    double T[3];
    int32_t n_inl;
    int32_t ncand;
    int32_t proceed;
    int32_t npre;        // inliers [0, npre) are triangulated under EVERY candidate by finalize_model (the prefix)
    int32_t pre_cnt[4];  // points of the prefix in front of both cameras, per candidate
    int32_t best_c;      // candidate with the largest prefix count (lowest index on ties): the one triangulate completes
    int32_t pad[3];
};

// Resident state of a batch (all device pointers).  P pairs, capacity N keypoints per image.
struct BatchDev {
    int n_pairs;
    int max_kp;       // N
    int desc_words;   // descriptor bytes / 4
    int max_groups;   // capacity of wgbest per pair
    int cu_count;     // compute units of the batch's device (hipDeviceAttributeMultiprocessorCount): launch-shape decisions

    // inputs
    const uint32_t *desc1;  // [P][N][desc_words]   base / train (vf1)
    const uint32_t *desc2;  // [P][N][desc_words]   pair / query (vf2)
    const float *kp1;       // [P][N][2]
    const float *kp2;       // [P][N][2]
    const uint8_t *oct1;    // [P][N] pyramid octave of every keypoint (cv::KeyPoint::octave), 0 where unknown
    const uint8_t *oct2;    // [P][N]
    const int32_t *n1;      // [P]
    const int32_t *n2;      // [P]
    const double *Kinv;     // [P][9]  host-computed cofactor inverse (camera.cpp:16)
    const double *K;        // [P][9]
    const int64_t *gidx;    // [P] global pair index (sampler key offset)

    // intermediates
    int32_t *knn_train;  // [P][N] best train index per query, -1 = rejected
    int32_t *knn_dist;   // [P][N]
    int32_t *M;          // [P] number of matches
    mvs_match *matches;  // [P][N]
    double *pts;         // [P][N][4]  (x1, y1, x2, y2) ideal-camera coordinates of match m
    WgBest *wgbest;      // [P][max_groups]
    double *hyp_F;       // [P][max_groups * 256][kHypRec]: F (9) of every hypothesis + its counting threshold thr + band
                         // (solve / pre-screen -> scoring hand-over), may be null
    float *hyp_r32;      // [P][max_groups * 256][kHypRec32]: the SINGLE-PRECISION pre-screen records of the pairs in mode 1 (48
                         // bytes: F~ as 9 floats, upper and lower counting threshold, one spare): written by the pre-screen, read
                         // by the pilot / dense / finish counting kernels.  Exact F (survivors, uncertified) stays in hyp_F
    uint8_t *hyp_okf;    // [P][max_groups * 256] state of the record: 0 rejected sample, 1 approximate F (pre-screen), 2 waits
                         // for the exact solve, 3 exact F
    int32_t *hyp_cnt;    // [P][max_groups * 256] full (upper-bound) inlier count of a hypothesis that can still win, -1 otherwise
    int32_t *bound;      // [P] largest full (lower-bound) count seen so far: the pruning bound of the counting kernel
    double *box;         // [P][8] bounding box of the pair's matches (x1lo, x1hi, y1lo, y1hi, x2lo, x2hi, y2lo, y2hi)
    int32_t *mode;       // [P] 0: every hypothesis is solved exactly; 1: pre-screened, counted in single precision; 2: pre-
                         // screened, counted in double precision
    uint32_t *clist;     // [P][max_groups * 256] hypotheses of the pair the dense counting phase left alive (for the finish)
    uint32_t *clist2;    // [P][max_groups * 256] the same list sorted by partial count (ransac_list_sort_kernel)
    int32_t *cpos;       // [P][kSortBins] cpos[c] = entries of the sorted list with a partial count >= c (the sort's own
                         // offsets): the finish reads how long the list's live prefix is instead of walking the dead rest
    int32_t *ccount;     // [P] their number
    int32_t *pcount;     // [P] entries of the pair's candidate list for the selection (ransac_survivors_kernel -> ransac_select_kernel;
                         // the list itself reuses the first half of clist, which is dead after the list sort)
    int32_t *m0list;     // [1 + P] number of pairs in mode 0, then those pairs (mode0_list_kernel -> ransac_solve_list_kernel)
    int32_t *dense_n1;   // [P] points the dense (matrix-core) counting phase covered for the pair
    uint32_t *xlist;     // [P * max_groups * 256] work list of the list-driven exact solve: flat indices pair * Hp + h
    uint32_t *xcount;    // [2] {entries of the list: flagged by the pre-screen + survivors of the count, unused}
    double *cand_pts;    // [P][4][N][3] triangulation scratch
    FinModel *fin;       // [P]
    uint16_t *inl;       // [P][N] ordered inlier list
    uint8_t *okf;        // [P][4][N] cheirality flags per candidate

    // outputs
    mvs_pair_result *results;  // [P]
    uint8_t *mask;             // [P][N]
    double *points;            // [P][N][3]
    int32_t *point_idx;        // [P][N]

    // optional per-hypothesis tables (single-shot diagnostics), may be null
    int32_t *hyp_count;     // [P][H]
    double *hyp_residual;   // [P][H]

    // work statistics (instrumented replay only), may be null: {rotations9, pairs9}
    unsigned long long *stats;
};

enum FinalizeMode : int {
    kFinalizeFull = 0,      // reduce wgbest -> F -> mask -> E -> decompose -> triangulate -> pose
    kFinalizeFromE = 1,     // results[p].E and mask are given
    kFinalizeTriangulate = 2  // results[p].R1to2 / t1to2 given, mask given; one candidate
};

struct RunParams {
    double ratio;
    double max_dist;
    double max_error_sq;  // <= 0: 5e-2 / K00 / K11 per pair
    int num_hypotheses;
    int sampler;
    uint64_t seed;
    int min_inliers;
};

// ---- pnp_solve (row f1) ----------------------------------------------------------------------------
constexpr int kPnpMaxPoints = kMaxKp;  // = the keypoint capacity (round 5: the point stream is chunked through LDS; 2048 before)

struct PnpRec {  // best hypothesis of one RANSAC workgroup
    int32_t count;
    uint32_t hyp;
    double R[9];
    double t[3];
};

struct PnpOut {
    int32_t ok;
    int32_t n_inliers;
    int32_t best_hyp;
    int32_t pad;
    double R[9];     // camera in world = SE3(SO3(Rw2c), tw2c).inverse()
    double t[3];
    double Rw2c[9];
    double tw2c[3];
};

struct PnpDev {       // a batch of PnP problems; problem q owns slice [q * stride, q * stride + n[q])
    int n_problems;
    int stride;       // capacity (points) of one problem, <= kPnpMaxPoints
    int num_hypotheses;
    int sampler;
    int min_inliers;
    int max_groups;   // records per problem
    uint64_t seed;    // problem q uses seed + gidx[q]
    double thr2;      // reproj_error^2
    const int32_t *n;       // [Q] points of each problem (< 7: no model)
    const int64_t *gidx;    // [Q] sampler key offsets (may be null = 0)
    const double *K;        // [Q][9]
    const double *Kinv;     // [Q][9]
    const double *X;        // [Q][stride][3] world points
    const double *uv;       // [Q][stride][2] image points
    double *xy;             // [Q][stride][2] ideal-camera coordinates
    double *fb;             // [Q][stride][3] unit bearings
    PnpRec *rec;            // [Q][max_groups]
    int32_t *inliers;       // [Q][stride]
    PnpOut *out;            // [Q]
};

void launch_pnp(const PnpDev &p, hipStream_t stream);

// ---- frame sequences (row f2): join of pair q's points with their observations in frame q + 2 ------------------
struct SeqJoinDev {
    int n_tracks;      // n_frames - 2
    int max_kp;
    int stride;        // capacity of one PnP problem = min(max_kp, kPnpMaxPoints)
    const mvs_pair_result *results;  // [n_frames - 1]
    const mvs_match *matches;        // [n_frames - 1][max_kp]
    const int32_t *M;                // [n_frames - 1]
    const double *points;            // [n_frames - 1][max_kp][3]
    const int32_t *point_idx;        // [n_frames - 1][max_kp]
    const float *kp;                 // [n_frames][max_kp][2]
    double *X;                       // [n_tracks][stride][3]
    double *uv;                      // [n_tracks][stride][2]
    int32_t *n_corr;                 // [n_tracks]
};
void launch_seq_join(const SeqJoinDev &j, hipStream_t stream);

// scale propagation + trajectory of a sequence (sequential fold over the frames, pnp.hip)
struct SeqChainDev {
    int n_frames;
    const mvs_pair_result *results;  // [n_frames - 1]
    const PnpOut *tracks;            // [n_frames - 2]
    const int32_t *n_corr;           // [n_frames - 2]
    double *traj_R;                  // [n_frames][9] pose of frame k in frame 0 (pair 0's baseline = 1)
    double *traj_t;                  // [n_frames][3]
    double *traj_sigma;              // [n_frames - 1] sigma_k = pair k's unit baseline in units of pair 0's
    double *track_scale;             // [n_frames - 2]
};
void launch_seq_chain(const SeqChainDev &c, hipStream_t stream);

// ---- sfm_refine / pnp_refine (row f4): batched Schur-complement Levenberg-Marquardt ---------------------------
struct RefineCfg {
    int max_iterations;
    double lambda_initial, lambda_factor, lambda_upper, rel_tol, abs_tol;
    double w[2][6];   // 1 / sigma^2 of the pose priors, frame 0 and 1
};

struct RefineDev {     // G problems; problem g owns slice [g * stride, g * stride + m[g])
    int n_problems;
    int stride;        // point capacity of one problem
    int n_frames;      // 2: frame 0 = camera 1 anchored at the identity, frame 1 = camera 2.  1: the moving camera
    RefineCfg cfg;
    const int32_t *m;        // [G] points (< 1: not solved)
    const double *K;         // [G][9]
    const double *pose0;     // [G][12] guess of the moving camera: R (9), t (3), camera in world
    const double *pose0_all; // optional [G][n_frames][12]: every frame's guess (general two-frame problem), else null
    const double *obs[2];    // [G][stride][2] image points seen by frame f
    const double *oinfo[2];  // [G][stride][3] their information matrices (xx xy yy)
    const double *pts0;      // [G][stride][3] point guesses = prior means
    const double *pinfo;     // [G][stride][6] point prior information (xx xy xz yy yz zz)
    double *pts;             // [G][stride][3] refined points (out)
    double *pts_tmp;         // [G][stride][3] candidate buffer
    double *point_cov;       // [G][stride][9] marginal covariances (out), may be null
    mvs_refine_result *out;  // [G] the moving camera (frame n_frames - 1)
    mvs_refine_result *out_all;  // optional [G][n_frames]: every frame, else null
};
// covariance -> information on the device.  cov2_[f]: [G][stride][4] or null (identity); cov3: [G][stride][9] or null
// (isotropic weight iso3).  Writes d.oinfo / d.pinfo (cast away const by the caller's own buffers).
void launch_refine_prep(const RefineDev &d, const double *cov2_0, const double *cov2_1, const double *cov3, double iso3,
                        double *oinfo0, double *oinfo1, double *pinfo, const uint8_t *valid0, const uint8_t *valid1,
                        hipStream_t stream);
void launch_refine(const RefineDev &d, hipStream_t stream);
// pnp_solve's refit over the inliers for a batch of PnP problems (the tracks of a sequence): gather -> refine_kernel<1>
// -> poses written back into p.out.  d: a one-frame RefineDev over the same problems with its own buffers.
void launch_pnp_refit(const PnpDev &p, const RefineDev &d, double point_sigma, hipStream_t stream);
// batch glue: build the two-view refinement problems of every pair from the batch's own results
void launch_refine_gather(const BatchDev &b, int n_active, double sigma_px, double point_sigma, int stride, int32_t *m,
                          double *pose0, double *obs0, double *obs1, double *oinfo0, double *oinfo1, double *pts0,
                          double *pinfo, hipStream_t stream);

// ---- VisualFeature::extract (row f3): ORB-style extraction for a batch of equally sized images ----------------
constexpr int kOrbMaxLevels = 16;
constexpr int kOrbSelCap = 16384;    // keys of one (image, level) the selection can hold in LDS (128 KB): 2 n_l <= this, or the
                                     // level's candidate list is capped at it (then, and only then, a level can overflow)

struct OrbLevel {
    int w, h;
    int n_keep;        // n_l: keypoints kept at this level
    float scale;       // 1.2^l
    size_t offset;     // pixels of one image's pyramid before this level
    size_t tab_offset; // first entry of this level's resize table (w column entries, then h row entries)
    int cand_cap;      // capacity of the level's candidate list: the non-maximum suppression's own bound (one survivor per 2x2
                       // block of the detection area: the list cannot overflow; round 5 -- 16384 before, MVS_ERR_CAPACITY beyond)
    size_t cand_off;   // first key of the level inside one image's candidate block
};
struct OrbSel {        // a selected keypoint of one level
    int32_t x, y;
    float harris;
};
struct OrbDev {
    int n_images, n_levels, nfeatures, edge, fast_threshold;
    size_t cand_stride;    // keys of one image's candidate block (sum of the levels' cand_cap)
    int flat_order;        // 0: XCD-aware block order of describe_kernel (the product); 1: (level, image, split) as in rounds 2-4 --
                           // only the diagnostics build can set it (MVS_ORB_FLAT_ORDER, tools/profile_extract.sh: the A/B of DESIGN 4.8)
    OrbLevel level[kOrbMaxLevels];
    uint8_t *pyr;          // [level][image][h_l][w_l]; level 0 = the input images
    uint8_t *blur;         // same layout: blurred levels
    const int2 *resize_tab;  // per level >= 1: {source index, 11-bit weight} per destination column and row
    uint64_t *cand_keys;   // [image][cand_stride]: level l at cand_off, cand_cap keys
    int32_t *cand_count;   // [image][level]
    OrbSel *sel;           // [image][level][nfeatures]
    int32_t *sel_count;    // [image][level]
    int32_t *overflow;     // [1] set when a level had more than cand_cap corners (possible only for capped lists, see kOrbSelCap)
    const int8_t *pattern; // [256][4]
    mvs_keypoint *kp;      // [image][nfeatures]
    uint8_t *desc;         // [image][nfeatures][32]
    int32_t *n_kp;         // [image]
    float *kp_xy;          // optional [image][nfeatures][2] (the layout the matcher reads), may be null
    uint8_t *kp_oct;       // optional [image][nfeatures] octave of every keypoint (refinement weights), may be null
};
hipError_t orb_prepare();
void launch_orb(const OrbDev &d, hipStream_t stream);

// ---- single-shot glue: pair 0's scalars as kernel arguments, pair 0's outputs gathered for one device-to-host copy --------
struct SingleParams {
    int32_t n1, n2;
    int64_t gidx;
    double K[9], Kinv[9];
    uint32_t part_bytes[4];   // desc1, desc2, kp1, kp2 of the packed input block (when one is given)
};
struct SingleLayout {
    size_t mask, points, idx, matches, total;
};
// flags: 1 mask, 2 points, 4 point indices, 8 matches
__host__ __device__ inline SingleLayout single_layout(int rows, int flags)
{
    auto up = [](size_t x) { return (x + 15) & ~size_t(15); };
    SingleLayout L;
    size_t o = 512;   // the 368-byte result record
    L.mask = o;
    o += (flags & 1) ? up((size_t)rows) : 0;
    L.points = o;
    o += (flags & 2) ? up((size_t)rows * 24) : 0;
    L.idx = o;
    o += (flags & 4) ? up((size_t)rows * 4) : 0;
    L.matches = o;
    o += (flags & 8) ? up((size_t)rows * 16) : 0;
    L.total = o;
    return L;
}
void launch_single_params(const BatchDev &b, const SingleParams &sp, const void *packed_in, hipStream_t stream);
void launch_single_gather(const BatchDev &b, int rows, int flags, unsigned char *out, hipStream_t stream);

// ---- kernel table (mvs_kernel_info_get) and per-launch timing (mvs_batch_time_kernels) ---------------------------
// One id per kernel of the two-view pipeline.  launch_* record an event in front of every launch when given a
// LaunchTimer, so the per-kernel times of the bench line are measured on the launches' own stream, launch by launch.
enum KernelId : int {
    kKMatchTopk = 0,
    kKMatchCompact,
    kKRansacFused,     // solve + hypothesis-per-lane scoring in one launch (short launches, per-hypothesis tables)
    kKRansacSolve,
    kKRansacSelect,
    kKPairPrepare,     // bounding box of the pair's matches + probe: is this pair pre-screened?
    kKRansacPrescreen, // approximate F + certified band per hypothesis
    kKRansacExactList, // exact solve of the listed hypotheses (flagged by the pre-screen / survivors of the count)
    kKRansacCount2,    // pruned counting with per-hypothesis thresholds (upper / lower bounds of the exact count)
    kKRansacCountPilot,  // ransac_count32_kernel, phase 0: the first kPilotHyp hypotheses in full -> the pair's first bound
    kKRansacCountMfma,   // dense counting of the points that must be seen before anything can be dropped: split bf16 on the
                         // matrix cores, no exit tests
    kKRansacCountFinish, // the first batch of every pair's list (largest partial counts): upper and lower count bounds, matrix cores
    kKRansacCountFinishRest, // the rest of the list: upper counts completed behind the dense phase's points
    kKRansacSurvivors, // hypotheses whose upper bound reaches the pair's best lower bound -> work list
    kKFinModel,
    kKTriangulate,
    kKFinSelect,
    kKMatchTopkVec,    // match_topk_kernel<8>: the vector 2-NN kernel when a 256-bit launch is too small for the matrix-core one
    kKernelCountProduct,   // the product library's table ends here
    // kernels of the experiment ladder: they exist in the diagnostics build (-DMVS_DEBUG_HOOKS) only
    kKRansacScore = kKernelCountProduct,   // hypothesis-per-lane scoring of stored F records
    kKRansacCount,     // round 2's pruned counting (one threshold per pair)
    kKRansacCount32,   // single-precision counting, everything in one launch (A/B: mvs_debug_set_count_dense(0))
    kKernelCount
};
struct LaunchTimer {
    hipStream_t stream;
    hipEvent_t *ev;   // cap + 1 events
    int32_t *kid;     // kernel id of launch k
    int cap;
    int n;
    void mark(int id)
    {
        if (n < cap) {
            (void)hipEventRecord(ev[n], stream);
            kid[n] = id;
            ++n;
        }
    }
    void end() { (void)hipEventRecord(ev[n], stream); }
};
struct KernelDesc {
    const char *name;     // as rocprofv3 prints it
    const void *fn;       // host-side handle of the __global__ function
    int threads;          // threads per workgroup as launched
    size_t dynamic_lds;   // bytes of dynamic LDS as launched for `max_kp`
};
// entry `id` of the table for a batch with `max_kp` keypoints per image; false past the end
bool kernel_desc(int id, int max_kp, int desc_words, KernelDesc *out);

// launch wrappers (all asynchronous on `stream`); lt: optional per-launch timer
void launch_match_topk(const BatchDev &b, const RunParams &rp, int n_active, hipStream_t stream, LaunchTimer *lt = nullptr);
void launch_match_compact(const BatchDev &b, const RunParams &rp, int n_active, hipStream_t stream, LaunchTimer *lt = nullptr);
void launch_prep_points(const BatchDev &b, const double *uv1, const double *uv2, int n_active, hipStream_t stream);
void launch_ransac(const BatchDev &b, const RunParams &rp, int n_active, bool stats, hipStream_t stream, LaunchTimer *lt = nullptr);
void launch_finalize(const BatchDev &b, const RunParams &rp, int n_active, int mode, hipStream_t stream, LaunchTimer *lt = nullptr);
// opt-in to > 64 KB of dynamic LDS for the kernels that need it, once per device; hipSuccess or the first error
hipError_t prepare_kernels();
#ifdef MVS_DEBUG_HOOKS
// diagnostics build only (libmvslam_hip_dbg.so): process-global switches, the experiment ladder, checkers
// pair_prepare + ransac_prescreen only, every pair forced into the pre-screened mode
void launch_prescreen_only(const BatchDev &b, const RunParams &rp, int n_active, int mode, hipStream_t stream);
void launch_mfma_probe(const uint16_t *A, const uint16_t *B, float *out, hipStream_t stream);
void set_count_dense(int v);      // 1 = single-precision counting as pilot + dense MFMA phase + finish
void set_match_mfma(int v);       // 1 (default) 256-bit descriptors on the matrix cores by batch size, 0 the VALU kernel
void set_prescreen_force(int m);   // -1 probe decides (default), 0 every pair exact, 1 every pair pre-screened
void set_split_min_pairs(int v);  // launches with fewer pairs stay on the fused hypothesis-per-lane kernel (default 3)
void set_ransac_variant(int v);  // A/B switch between co-compiled ransac_kernel variants
int get_ransac_variant();
void launch_fastmath_check(const double *x, const double *y, int n, unsigned long long *out, hipStream_t stream);
void launch_pairstep_check(const double *rows, int n, unsigned long long *out, hipStream_t stream);
// full-population audit of the pre-screened stage (kernels.hip: audit_kernel); out: 16 counters, maxc: [n_active]
void launch_count_only(const BatchDev &b, const RunParams &rp, int n_active, int pmode, int dense, const int32_t *keep,
                       hipStream_t stream);
void launch_indicator_probe(const float *a, const float *tu, const float *tl, const float *T, int n, float *ind_u, float *ind_l,
                            float *scale, hipStream_t stream);
void launch_rounding_probe(const double *in, int n, double *out, hipStream_t stream);
hipError_t launch_audit(const BatchDev &b, const RunParams &rp, int n_active, int phase, unsigned long long *out, int32_t *maxc,
                        hipStream_t stream);
#endif
void launch_fundamental(const double *p1, const double *p2, double *F, int *ok, hipStream_t stream);

}  // namespace mvs
