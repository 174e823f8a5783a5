// TEST-SIDE STUB (tests/test_integration_syntax.py): the OpenCV declarations the reference's headers and this repo's forwarding
// translation units name.  Declarations only -- see tests/cpp/stubs/Eigen/Core for the purpose and the limits.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5
#define CV_64FC1 6
#define CV_64F 6
#define CV_32F 5

namespace cv
{
typedef unsigned char uchar;
template <typename T>
struct Point_ {
    T x, y;
    Point_();
    Point_(T x_, T y_);
};
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
typedef Point2i Point;
template <typename T>
struct Point3_ {
    T x, y, z;
    Point3_();
    Point3_(T x_, T y_, T z_);
};
typedef Point3_<float> Point3f;
typedef Point3_<double> Point3d;
template <typename T>
struct Size_ {
    T width, height;
    Size_();
    Size_(T w, T h);
};
typedef Size_<int> Size;
struct Scalar {
    Scalar();
    Scalar(double a, double b = 0, double c = 0, double d = 0);
};
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint();
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1);
    size_t hash() const;
};
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch();
    DMatch(int q, int t, float d);
    DMatch(int q, int t, int i, float d);
    bool operator<(const DMatch &m) const;
};
struct Mat {
    int rows, cols, flags, dims;
    uchar *data;
    Mat();
    Mat(int r, int c, int type);
    Mat(int r, int c, int type, void *data, size_t step = 0);
    Mat(int r, int c, int type, const Scalar &s);
    Mat(Size s, int type);
    Mat(const Mat &);
    Mat &operator=(const Mat &);
    ~Mat();
    template <typename T>
    T &at(int i, int j);
    template <typename T>
    const T &at(int i, int j) const;
    template <typename T>
    T &at(int i);
    template <typename T>
    T *ptr(int i = 0);
    template <typename T>
    const T *ptr(int i = 0) const;
    uchar *ptr(int i = 0);
    const uchar *ptr(int i = 0) const;
    Mat row(int i) const;
    Mat rowRange(int a, int b) const;
    Mat colRange(int a, int b) const;
    Mat col(int i) const;
    Mat clone() const;
    Mat t() const;
    void copyTo(Mat &m) const;
    void push_back(const Mat &m);
    void create(int r, int c, int type);
    bool empty() const;
    bool isContinuous() const;
    int type() const;
    int channels() const;
    size_t total() const;
    size_t elemSize() const;
    Size size() const;
    static Mat zeros(int r, int c, int type);
    static Mat eye(int r, int c, int type);
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
template <typename T>
struct Ptr {
    Ptr();
    T *operator->() const;
    T &operator*() const;
    bool empty() const;
    operator bool() const;
};
enum NormTypes { NORM_L1 = 2, NORM_L2 = 4, NORM_HAMMING = 6, NORM_HAMMING2 = 7 };
struct Feature2D {
    virtual ~Feature2D();
    void detect(const Mat &image, std::vector<KeyPoint> &kp, const Mat &mask = Mat());
    void compute(const Mat &image, std::vector<KeyPoint> &kp, Mat &desc);
    void detectAndCompute(const Mat &image, const Mat &mask, std::vector<KeyPoint> &kp, Mat &desc, bool useProvided = false);
};
struct ORB : Feature2D {
    static Ptr<ORB> create(int nfeatures = 500, float scaleFactor = 1.2f, int nlevels = 8, int edgeThreshold = 31,
                           int firstLevel = 0, int WTA_K = 2, int scoreType = 0, int patchSize = 31, int fastThreshold = 20);
};
struct DescriptorMatcher {
    virtual ~DescriptorMatcher();
    void match(const Mat &q, const Mat &t, std::vector<DMatch> &m) const;
    void knnMatch(const Mat &q, const Mat &t, std::vector<std::vector<DMatch> > &m, int k) const;
};
struct BFMatcher : DescriptorMatcher {
    BFMatcher(int normType = NORM_L2, bool crossCheck = false);
};
struct SVD {
    enum { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 };
};
void SVDecomp(const Mat &src, Mat &w, Mat &u, Mat &vt, int flags = 0);
Mat imread(const std::string &f, int flags = 1);
bool imwrite(const std::string &f, const Mat &m);
}  // namespace cv
