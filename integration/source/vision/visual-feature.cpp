// Replaces source/vision/visual-feature.cpp of the reference (decl vision/visual-feature.hpp:8-94): extraction and
// matching run on the GPU, the container members are the reference's own (restated: they are five-line accessors).
//   extract                           :40-49   -> mvs_extract (ORB's pipeline, NOT OpenCV's learned pattern: DESIGN.md 4.8)
//   match_visual_features             :51-80   -> mvs_match_hamming
//   match_and_filter_visual_features  :93-119  -> mvs_match_hamming + host gather
#include <vision/visual-feature.hpp>

#include <math/utility.hpp>

#include <cassert>
#include <iostream>

#include "mvslam-hip-glue.hpp"
#include <unordered_map>

namespace mvSLAM
{
static_assert(sizeof(cv::DMatch) == sizeof(mvs_match), "cv::DMatch layout");
static_assert(sizeof(cv::KeyPoint) == sizeof(mvs_keypoint), "cv::KeyPoint layout");

VisualFeature
VisualFeature::extract(const ImageGrayscale &image)
{
    assert(image.type() == CV_8UC1 && image.rows > 0 && image.cols > 0);
    const cv::Mat img = image.isContinuous() ? image : image.clone();
    mvs_orb_params prm;
    mvs_orb_params_default(&prm);     // 500 features, scale 1.2, 8 levels, ... (visual-feature.cpp:9,12-17)
    VisualFeature vf;
    vf.m_keypoints.resize(prm.nfeatures);
    vf.m_descriptors.create(prm.nfeatures, 32, CV_8U);
    int32_t n = 0;
    const mvs_status st = mvs_extract(hip::context(), img.data, 1, img.cols, img.rows, &prm,
                                      reinterpret_cast<mvs_keypoint *>(vf.m_keypoints.data()), vf.m_descriptors.data, &n);
    if (st != MVS_OK) {
        // the reference's extract cannot fail; an empty feature set alone would hide a device error behind
        // "valid() == false": say what happened on the reference's own diagnostics channel (std::clog, base/debug.cpp:4-7)
        std::clog << "[VisualFeature::extract] mvs_extract failed: " << mvs_status_str(st) << " ("
                  << mvs_last_error(hip::context()) << ")" << std::endl;
        n = 0;
    }
    vf.m_keypoints.resize(n);
    vf.m_descriptors = vf.m_descriptors.rowRange(0, n).clone();
    vf.m_image_width = image.cols;
    vf.m_image_height = image.rows;
    return vf;
}

VisualFeatureConfig::MatchResultType
VisualFeature::match_visual_features(const VisualFeature &vf1, const VisualFeature &vf2, ScalarType max_dist)
{
    assert(vf1.valid() && vf2.valid());   // :56
    assert(vf1.m_descriptors.isContinuous() && vf2.m_descriptors.isContinuous());
    VisualFeatureConfig::MatchResultType out(vf2.size());
    int n = 0;
    // base = vf1 = train, pair = vf2 = query (:59-62); Lowe ratio 0.7 (:23); canonical order (distance, queryIdx)
    const mvs_status st = mvs_match_hamming(hip::context(), vf1.m_descriptors.data, vf1.m_descriptors.rows,
                                            vf2.m_descriptors.data, vf2.m_descriptors.rows, vf1.m_descriptors.cols, 0.7,
                                            max_dist, reinterpret_cast<mvs_match *>(out.data()), &n);
    out.resize(st == MVS_OK ? n : 0);
    return out;
}

std::pair<VisualFeature, VisualFeature>
VisualFeature::match_and_filter_visual_features(const VisualFeature &vf1, const VisualFeature &vf2, ScalarType max_dist)
{
    const auto matches = match_visual_features(vf1, vf2, max_dist);
    assert(vf1.m_image_height == vf2.m_image_height && vf1.m_image_width == vf2.m_image_width);
    VisualFeature f1, f2;
    f1.m_image_height = f2.m_image_height = vf1.m_image_height;
    f1.m_image_width = f2.m_image_width = vf1.m_image_width;
    for (const auto &m : matches) {
        f1.m_keypoints.push_back(vf1.m_keypoints[m.trainIdx]);
        f1.m_descriptors.push_back(vf1.m_descriptors.row(m.trainIdx));
        f2.m_keypoints.push_back(vf2.m_keypoints[m.queryIdx]);
        f2.m_descriptors.push_back(vf2.m_descriptors.row(m.queryIdx));   // the reference pushes this row into
    }                                                                     // filtered1 (:115): not copied
    return std::make_pair(f1, f2);
}

VisualFeature::VisualFeature() : m_keypoints(), m_descriptors(), m_image_width(-1), m_image_height(-1) {}
VisualFeature::~VisualFeature() {}

bool VisualFeature::equivalent_to(const VisualFeature &other) const
{
    // the reference's semantics (visual-feature.cpp:141-167): both valid, equally many keypoints, and the two keypoint sets
    // agree as sets of cv::KeyPoint::hash() values with every hash of *this met exactly once in `other` -- order is free
    if (!valid() || !other.valid() || size() != other.size())
        return false;
    std::unordered_map<size_t, size_t> seen_in_other;
    for (const auto &kp : m_keypoints)
        seen_in_other.emplace(kp.hash(), 0);
    for (const auto &kp : other.m_keypoints) {
        const auto it = seen_in_other.find(kp.hash());
        if (it == seen_in_other.end())
            return false;
        ++it->second;
    }
    for (const auto &entry : seen_in_other)
        if (entry.second != 1)
            return false;
    return true;
}

size_t VisualFeature::size() const { return m_keypoints.size(); }

const VisualFeatureConfig::DetectorResultType &VisualFeature::get_keypoints() const
{
    assert(valid());
    return m_keypoints;
}

std::vector<ImagePoint> VisualFeature::get_image_points() const
{
    assert(valid());
    std::vector<ImagePoint> result;
    result.reserve(m_keypoints.size());
    for (const auto &kp : m_keypoints)
        result.emplace_back(kp.pt.x, kp.pt.y);
    return result;
}

std::vector<Point2Estimate> VisualFeature::get_point_estimates() const
{
    assert(valid());
    std::vector<Point2Estimate> result;
    result.reserve(m_keypoints.size());
    for (const auto &kp : m_keypoints) {
        const ScalarType stddev = static_cast<ScalarType>(1 << kp.octave) * 0.5;   // :203
        result.emplace_back(Point2(kp.pt.x, kp.pt.y), sqr(stddev) * Point2Uncertainty::Identity());
    }
    return result;
}

bool VisualFeature::valid() const { return (size() > 0) && (m_image_width > 0) && (m_image_height > 0); }

}  // namespace mvSLAM
