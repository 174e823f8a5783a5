// Replaces source/vision/ba.cpp of the reference (ba_frame_pose_and_point :26-156; decl vision/ba.hpp:25-36) -- the one
// translation unit of the front end that talks to GTSAM.  The same cost (diagonal pose priors, point priors, one
// projection factor per observation with the keypoint's covariance) is minimised by the library's batched
// Schur-complement Levenberg-Marquardt kernel; marginal covariances from the linearised system at the
// estimate, final_error = the optimiser's error, as :130-155.  Supported: what the reference builds -- ONE or TWO frames
// (sfm-refine.cpp:20-139, pnp-refine.cpp:14-108, VisualOdometer::track_refine visual-odometer.cpp:618-800).
// With this file in place sfm-refine.cpp, pnp-refine.cpp and visual-odometer.cpp stay untouched and GTSAM leaves the
// front end's link line.
#include <vision/ba.hpp>

#include <algorithm>
#include <cassert>
#include <cstring>
#include <stdexcept>

#include "mvslam-hip-glue.hpp"

namespace mvSLAM
{
void ba_frame_pose_and_point(const CameraIntrinsics &ci, const std::unordered_set<Id::Type> &frame_id,
                             const std::unordered_set<Id::Type> &point_id,
                             const std::unordered_map<Id::Type, Transformation> &frame_pose_guess,
                             const std::unordered_map<Id::Type, TransformationUncertainty> &frame_pose_prior,
                             const std::unordered_map<Id::Type, Point3> &point_guess,
                             const std::unordered_map<Id::Type, Point3Uncertainty> &point_prior,
                             const std::unordered_map<Id::Type, PointIdToPoint2Estimate> &frame_observation,
                             std::unordered_map<Id::Type, TransformationEstimate> &frame_pose_estimate,
                             std::unordered_map<Id::Type, Point3Estimate> &point_estimate, ScalarType &final_error)
{
    assert(frame_id.size() > 0 && frame_id.size() <= 2);   // ba.cpp:39; more than two frames: the reference builds none
    assert(point_id.size() > 0);
    assert(frame_pose_guess.size() == frame_id.size() && point_guess.size() == point_id.size());
    assert(frame_pose_prior.size() + point_prior.size() >= 2);   // ba.cpp:43
    std::vector<Id::Type> fids(frame_id.begin(), frame_id.end()), pids(point_id.begin(), point_id.end());
    std::sort(fids.begin(), fids.end());
    std::sort(pids.begin(), pids.end());
    const int F = (int)fids.size(), m = (int)pids.size();
    std::unordered_map<Id::Type, int> pidx;
    for (int i = 0; i < m; ++i)
        pidx[pids[i]] = i;
    std::vector<double> pose(12 * (size_t)F), var(6 * (size_t)F, 0.0), pts(3 * (size_t)m), pcov(9 * (size_t)m, 0.0);
    std::vector<double> obs[2], ocov[2];
    std::vector<uint8_t> valid[2];
    for (int f = 0; f < F; ++f) {
        const Transformation &T = frame_pose_guess.at(fids[f]);
        hip::to_row_major(T.rotation().get_matrix(), &pose[12 * f]);
        const Vector3Type t = T.translation();
        std::memcpy(&pose[12 * f + 9], t.data(), 3 * sizeof(double));
        auto pr = frame_pose_prior.find(fids[f]);
        if (pr != frame_pose_prior.end())
            for (int k = 0; k < 6; ++k)          // the reference's priors are diagonal (sfm-refine.cpp:58-78)
                var[6 * f + k] = pr->second(k, k);
        obs[f].assign(2 * (size_t)m, 0.0);
        ocov[f].assign(4 * (size_t)m, 0.0);
        valid[f].assign(m, 0);
        auto ob = frame_observation.find(fids[f]);
        if (ob != frame_observation.end())
            for (const auto &kv : ob->second) {
                const int i = pidx.at(kv.first);
                std::memcpy(&obs[f][2 * i], kv.second.mean().data(), 2 * sizeof(double));
                std::memcpy(&ocov[f][4 * i], kv.second.covar().data(), 4 * sizeof(double));   // symmetric 2x2
                valid[f][i] = 1;
            }
    }
    for (int i = 0; i < m; ++i) {
        std::memcpy(&pts[3 * i], point_guess.at(pids[i]).data(), 3 * sizeof(double));
        auto pr = point_prior.find(pids[i]);
        if (pr != point_prior.end())
            std::memcpy(&pcov[9 * i], pr->second.data(), 9 * sizeof(double));                 // symmetric 3x3
    }
    double Kr[9];
    hip::to_row_major(ci, Kr);
    mvs_ba_problem pb;
    std::memset(&pb, 0, sizeof(pb));
    pb.n_frames = F;
    pb.n_points = m;
    pb.K = Kr;
    pb.frame_pose = pose.data();
    pb.frame_prior_var = var.data();
    pb.points = pts.data();
    pb.point_prior_cov = pcov.data();
    for (int f = 0; f < F; ++f) {
        pb.obs[f] = obs[f].data();
        pb.obs_cov[f] = ocov[f].data();
        pb.obs_valid[f] = valid[f].data();
    }
    mvs_refine_params prm;
    mvs_refine_params_default(&prm);
    std::vector<mvs_refine_result> res(F);
    std::vector<double> po(3 * (size_t)m), pc(9 * (size_t)m);
    if (mvs_ba_refine(hip::context(), &pb, &prm, res.data(), po.data(), pc.data()) != MVS_OK)
        throw std::runtime_error("ba_frame_pose_and_point: indeterminate system");   // GTSAM throws here as well
    frame_pose_estimate.clear();
    for (int f = 0; f < F; ++f) {
        TransformationUncertainty C;
        std::memcpy(C.data(), res[f].pose_cov, 36 * sizeof(double));                   // symmetric 6x6, tangent order
        frame_pose_estimate[fids[f]] = TransformationEstimate(hip::se3_from_arrays(res[f].R, res[f].t), C);   // (rotation, translation) as ba.cpp:141
    }
    point_estimate.clear();
    for (int i = 0; i < m; ++i) {
        Point3Uncertainty C;
        std::memcpy(C.data(), &pc[9 * (size_t)i], 9 * sizeof(double));
        point_estimate[pids[i]] = Point3Estimate(Point3(po[3 * i], po[3 * i + 1], po[3 * i + 2]), C);
    }
    final_error = res[0].error;
}
}  // namespace mvSLAM
