// ktlvo_host.hpp -- C++11 host side above the C ABI of ktlvo.h, under the reference's own names.
//
// The reference (scofield1991/slam, src/ktl_vo) reaches the tracking arithmetic through C++ free
// functions over process globals and OpenCV containers.  This header gives the same functions, with
// the same argument order, meaning and side effects (outputs are appended with push_back, inputs are
// untouched), as members of one Frontend object that owns the device context:
//
//   reference                                              here
//   cv::Mat image (featureTracking.cpp:41-46, 351-365)     Image  (a device slot holding the pyramid)
//   cv::calcOpticalFlowPyrLK   (:205-209, 457-479)         Frontend::calcOpticalFlowPyrLK
//   getOpticalFlow             (:181-256)                  Frontend::getOpticalFlow
//   gftt->detect(image(roi))   (:102-103, 408-430)         Frontend::detect / detectSubregions
//   adaptiveNonMaximalSuppresion (:258-311)                Frontend::adaptiveNonMaximalSuppresion
//   stereoImageCallback        (:339-666)                  Frontend::stereoImageCallback
//   UnprojectStereo            (visualOdometry.cpp:153-167) Frontend::UnprojectStereo
//   estimatePoseRANSAC         (visualOdometry.cpp:177-217) Frontend::estimatePoseRANSAC
//   pointCloudCallback         (visualOdometry.cpp:240-356) Frontend::pointCloudCallback
//
// Error behaviour: the reference handles no errors; an OpenCV assertion throws cv::Exception, nobody
// catches it and the node dies.  Here every failed library call throws ktlvo_host::Error (a
// std::runtime_error carrying the ktlvo_status and, for HIP failures, the runtime's text), so an
// unmodified caller dies the same way and a careful one can catch it.
//
// Point2f / Point3f / KeyPoint / Rect below have the member layout of the OpenCV types of the same
// name, so containers of either can be reinterpreted where a node keeps OpenCV for its other work.
// Header only, no HIP or OpenCV includes; link with libktlvo_hip.so.
#ifndef KTLVO_HOST_HPP
#define KTLVO_HOST_HPP

#include "ktlvo.h"

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

namespace ktlvo_host {

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
struct Rect { int x, y, width, height; };
struct KeyPoint { // cv::KeyPoint; GFTTDetector fills pt and size only
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
struct TermCriteria { // cv::TermCriteria(type, maxCount, epsilon)
    enum { COUNT = 1, EPS = 2 };
    int type, maxCount;
    double epsilon;
};
typedef ktlvo_image_point ImagePoint; // pointDefinition.h:18-21
typedef ktlvo_depth_point DepthPoint; // pointDefinition.h:28-34
static_assert(sizeof(Point2f) == 8 && sizeof(Point3f) == 12 && sizeof(Rect) == sizeof(ktlvo_roi), "layout");
static_assert(sizeof(ImagePoint) == 12 && sizeof(DepthPoint) == 24, "layout");

class Error : public std::runtime_error {
  public:
    Error(int status, const std::string &what) : std::runtime_error(what), code(status) {}
    int code; // ktlvo_status
};

// What stands where the reference holds a cv::Mat frame: the frame's pyramid in device memory.
struct Image {
    int slot, cols, rows;
    Image() : slot(-1), cols(0), rows(0) {}
    bool empty() const { return slot < 0; }
};

class Frontend {
  public:
    // cfg == nullptr: the reference's constants (cameraParameters.h:16-27, featureTracking.cpp:68-81,
    // visualOdometry.cpp:122-127).
    explicit Frontend(const ktlvo_config *cfg = nullptr, int device = 0) : ctx_(nullptr) {
        if (cfg) cfg_ = *cfg;
        else ktlvo_default_config(&cfg_);
        ctx_ = ktlvo_create(&cfg_, device);
        if (!ctx_) throw Error(KTLVO_ERR_HIP, "ktlvo_create failed: no usable MI355X device, or the configuration is invalid");
        for (int i = 0; i < 16; i++) mTcw[i] = mTcw_local[i] = (i % 5 == 0) ? 1.f : 0.f; // Mat::eye(4,4,CV_32F), :113-114
    }
    ~Frontend() { if (ctx_) ktlvo_destroy(ctx_); }
    Frontend(const Frontend &) = delete;
    Frontend &operator=(const Frontend &) = delete;

    ktlvo_ctx *context() const { return ctx_; }
    const ktlvo_config &config() const { return cfg_; }

    // `image = imageData->image` (featureTracking.cpp:351-365): the frame goes to device slot `slot` and
    // its pyramid and derivative planes are built once, for every later call that names the Image.
    // step = bytes between rows (cv::Mat::step).
    Image upload(int slot, const uint8_t *data, int cols, int rows, size_t step) {
        check(ktlvo_upload_image(ctx_, slot, data, cols, rows, (int)step), "ktlvo_upload_image");
        check(ktlvo_build_pyramid(ctx_, slot, cfg_.max_level, cfg_.win), "ktlvo_build_pyramid");
        Image im;
        im.slot = slot; im.cols = cols; im.rows = rows;
        return im;
    }

    // cv::calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts, status, err, winSize, maxLevel, criteria,
    // flags, minEigThreshold).  winSize is the side of the square window; flags must be 0 (the reference
    // passes 0: no initial flow, error = mean L1 residual).
    void calcOpticalFlowPyrLK(const Image &prevImg, const Image &nextImg, const std::vector<Point2f> &prevPts,
                              std::vector<Point2f> &nextPts, std::vector<unsigned char> &status, std::vector<float> &err,
                              int winSize = 21, int maxLevel = 4,
                              TermCriteria criteria = {TermCriteria::COUNT | TermCriteria::EPS, 30, 0.01}, int flags = 0,
                              double minEigThreshold = 1e-4) {
        if (flags != 0) throw Error(KTLVO_ERR_UNSUPPORTED, "calcOpticalFlowPyrLK: flags other than 0 are not implemented");
        // cv::TermCriteria semantics inside calcOpticalFlowPyrLK: a missing COUNT means 30, a missing EPS 0.01
        const int count = (criteria.type & TermCriteria::COUNT) ? (criteria.maxCount < 0 ? 0 : criteria.maxCount > 100 ? 100 : criteria.maxCount) : 30;
        const double eps = (criteria.type & TermCriteria::EPS) ? (criteria.epsilon < 0 ? 0. : criteria.epsilon > 10. ? 10. : criteria.epsilon) : 0.01;
        const ktlvo_lk_params p = {winSize, maxLevel, count, eps, minEigThreshold, KTLVO_LK_SUMS_DEFAULT};
        const int n = (int)prevPts.size();
        nextPts.resize(n);
        status.resize(n);
        err.resize(n);
        check(ktlvo_pyrlk(ctx_, prevImg.slot, nextImg.slot, n ? &prevPts[0].x : nullptr, n, n ? &nextPts[0].x : nullptr,
                          status.data(), err.data(), &p), "ktlvo_pyrlk");
    }

    // getOpticalFlow (featureTracking.cpp:181-256): forward track image_1 -> image_2, backward track, the
    // keep predicate (:230-234), and for every kept i in input order
    //   features_1_selected += features_1[i];  features_2_selected += its track;
    //   if (!newKeyFrame)  features_1_indeces_selected += features_1_indeces[i];
    //   if (temporal_flow) features_3_selected += features_depth[i].
    void getOpticalFlow(const Image &image_1, const Image &image_2, const std::vector<Point2f> &features_1,
                        const std::vector<Point2f> &features_depth, std::vector<Point2f> &features_1_selected,
                        std::vector<Point2f> &features_2_selected, std::vector<Point2f> &features_3_selected,
                        std::vector<int> &features_1_indeces, std::vector<int> &features_1_indeces_selected,
                        int y_threshold, bool temporal_flow = false, bool newKeyFrame = false) {
        const int n = (int)features_1.size();
        if (n == 0) return;
        // the reference indexes these unchecked (undefined behaviour when short); here it is an error
        if (!newKeyFrame && (int)features_1_indeces.size() < n) throw Error(KTLVO_ERR_INVALID, "getOpticalFlow: features_1_indeces shorter than features_1");
        if (temporal_flow && (int)features_depth.size() < n) throw Error(KTLVO_ERR_INVALID, "getOpticalFlow: features_depth shorter than features_1");
        std::vector<Point2f> s1(n), s2(n);
        std::vector<int> keep(n);
        int nk = 0;
        check(ktlvo_get_optical_flow(ctx_, image_1.slot, image_2.slot, &features_1[0].x, n, y_threshold, &s1[0].x, &s2[0].x,
                                     keep.data(), &nk), "ktlvo_get_optical_flow");
        for (int k = 0; k < nk; k++) {
            features_1_selected.push_back(s1[k]);
            features_2_selected.push_back(s2[k]);
            if (!newKeyFrame) features_1_indeces_selected.push_back(features_1_indeces[keep[k]]);
            if (temporal_flow) features_3_selected.push_back(features_depth[keep[k]]);
        }
    }

    // The circular stereo-temporal test of featureTracking.cpp:486-501 over the two tracks that end in
    // the current right image; appends the indices of the kept features.
    void circularCheck(const std::vector<Point2f> &featuresPrevCurRight, const std::vector<Point2f> &featuresCurLeftRight,
                       std::vector<int> &kept) {
        const int n = (int)featuresPrevCurRight.size();
        if ((int)featuresCurLeftRight.size() != n) throw Error(KTLVO_ERR_INVALID, "circularCheck: list sizes differ");
        if (n == 0) return;
        std::vector<int> keep(n);
        int nk = 0;
        check(ktlvo_circular_gate(ctx_, &featuresPrevCurRight[0].x, &featuresCurLeftRight[0].x, n, keep.data(), &nk), "ktlvo_circular_gate");
        kept.insert(kept.end(), keep.begin(), keep.begin() + nk);
    }

    // gftt->detect(image(roi), keypoints) with gftt = GFTTDetector::create(maxCorners, quality, minDistance,
    // blockSize, useHarris, k) (featureTracking.cpp:102-103, 419-421): keypoints are ROI-relative, size =
    // blockSize, response 0 (OpenCV 3's detector does not store it).
    void detect(const Image &image, const Rect &roi, std::vector<KeyPoint> &keypoints, const ktlvo_gftt_params *gftt = nullptr) {
        std::vector<std::vector<KeyPoint> > per(1);
        detectSubregions(image, std::vector<Rect>(1, roi), per, gftt);
        keypoints.swap(per[0]);
    }
    // The same for a list of ROIs of one image in one set of launches (the 20 x 8 loop of :408-430).
    void detectSubregions(const Image &image, const std::vector<Rect> &rois, std::vector<std::vector<KeyPoint> > &keypoints,
                          const ktlvo_gftt_params *gftt = nullptr) {
        const ktlvo_gftt_params def = {cfg_.max_per_subregion, 0.01, 10, 3, 1, 0.0};
        const ktlvo_gftt_params &g = gftt ? *gftt : def;
        const int nr = (int)rois.size(), cap = g.max_corners;
        keypoints.assign(nr, std::vector<KeyPoint>());
        if (nr == 0 || cap <= 0) return;
        std::vector<float> xy((size_t)nr * cap * 2), resp((size_t)nr * cap);
        std::vector<int> counts(nr);
        check(ktlvo_detect_gftt(ctx_, image.slot, reinterpret_cast<const ktlvo_roi *>(rois.data()), nr, &g, xy.data(),
                                resp.data(), counts.data(), cap), "ktlvo_detect_gftt");
        for (int r = 0; r < nr; r++)
            for (int k = 0; k < counts[r]; k++) {
                KeyPoint kp = {{xy[((size_t)r * cap + k) * 2], xy[((size_t)r * cap + k) * 2 + 1]}, (float)g.block_size, -1.f, 0.f, 0, -1};
                keypoints[r].push_back(kp);
            }
    }

    // adaptiveNonMaximalSuppresion(keypoints, numToKeep), featureTracking.cpp:258-311; in place.
    void adaptiveNonMaximalSuppresion(std::vector<KeyPoint> &keypoints, const int numToKeep) {
        int n = (int)keypoints.size();
        if (n == 0) return;
        std::vector<float> xy((size_t)n * 2), resp(n);
        for (int i = 0; i < n; i++) { xy[2 * i] = keypoints[i].pt.x; xy[2 * i + 1] = keypoints[i].pt.y; resp[i] = keypoints[i].response; }
        check(ktlvo_anms(ctx_, xy.data(), resp.data(), &n, numToKeep), "ktlvo_anms");
        const KeyPoint proto = keypoints[0];
        keypoints.resize(n);
        for (int i = 0; i < n; i++) {
            keypoints[i] = proto; // size / angle / octave / class_id are the same for every detector output
            keypoints[i].pt.x = xy[2 * i]; keypoints[i].pt.y = xy[2 * i + 1]; keypoints[i].response = resp[i];
        }
    }

    // stereoImageCallback (featureTracking.cpp:339-666) from the two mono8 frames (tightly packed rows) to
    // the contents of the two published clouds; returns newKeyFrame.  Nothing is appended for the first frame.
    bool stereoImageCallback(const uint8_t *left, const uint8_t *right, std::vector<DepthPoint> &imagePointsLast,
                             std::vector<ImagePoint> &imagePointsCur) {
        imagePointsLast.resize(cfg_.max_points); // :368-370: both clouds are cleared every frame
        imagePointsCur.resize(cfg_.max_points);
        int n = 0, key = 0;
        const int rc = ktlvo_track_stereo_frame(ctx_, left, right, imagePointsLast.data(), imagePointsCur.data(), cfg_.max_points, &n, &key);
        imagePointsLast.resize(rc == KTLVO_OK ? n : 0);
        imagePointsCur.resize(rc == KTLVO_OK ? n : 0);
        check(rc, "ktlvo_track_stereo_frame");
        return key != 0;
    }

    // UnprojectStereo (visualOdometry.cpp:153-167) with mRwc = I, mOw = 0 as the node leaves them.  The
    // reference returns an empty Mat for z <= 0 and then reads it; here that is an error.
    Point3f UnprojectStereo(const Point3f &uvz) const {
        if (!(uvz.z > 0)) throw Error(KTLVO_ERR_INVALID, "UnprojectStereo: depth must be positive");
        Point3f p = {(float)((uvz.x - cfg_.cx) * uvz.z / cfg_.fx), (float)((uvz.y - cfg_.cy) * uvz.z / cfg_.fy), uvz.z};
        return p;
    }

    // estimatePoseRANSAC (visualOdometry.cpp:177-217): solvePnPRansac with the P3P kernel; the rotation and
    // translation of the row-major 4x4 mTcw are overwritten, its last row is left alone; `inliers` receives
    // the indices of the inlier correspondences (the N x 1 CV_32S Mat of OpenCV).  When RANSAC fails OpenCV
    // leaves rvec/tvec as they were on entry (Rodrigues of mTcw's rotation, :185-186), so mTcw keeps its value.
    // flags must be SOLVEPNP_P3P (= 2), the only method the reference selects (:122).
    void estimatePoseRANSAC(const std::vector<Point3f> &list_points3d, const std::vector<Point2f> &list_points2d, int flags,
                            std::vector<int> &inliers, int iterationsCount, float reprojectionError, double confidence,
                            float mTcw_io[16]) {
        if (flags != 2) throw Error(KTLVO_ERR_UNSUPPORTED, "estimatePoseRANSAC: only SOLVEPNP_P3P (2) is implemented");
        const int n = (int)list_points3d.size();
        if ((int)list_points2d.size() != n) throw Error(KTLVO_ERR_INVALID, "estimatePoseRANSAC: list sizes differ");
        inliers.clear();
        double rvec[3], tvec[3];
        int ninl = 0, found = 0;
        std::vector<uint8_t> mask(n ? n : 1);
        check(ktlvo_solve_pnp_ransac(ctx_, n ? &list_points3d[0].x : nullptr, n ? &list_points2d[0].x : nullptr, n,
                                     iterationsCount, reprojectionError, confidence, rvec, tvec, mask.data(), &ninl, &found),
              "ktlvo_solve_pnp_ransac");
        if (!found) return;
        for (int i = 0; i < n; i++) if (mask[i]) inliers.push_back(i);
        double R[9];
        rodrigues(rvec, R);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) mTcw_io[4 * i + j] = (float)R[3 * i + j];
            mTcw_io[4 * i + 3] = (float)tvec[i];
        }
    }

    // pointCloudCallback (visualOdometry.cpp:240-356): unproject the last frame's points, estimate the local
    // transform, chain it into the global one.  transform = the float32[16] of ktl_vo/transform_msg.
    float mTcw[16], mTcw_local[16]; // visualOdometry.cpp:113-114
    void pointCloudCallback(const std::vector<DepthPoint> &imagePointsLast, const std::vector<ImagePoint> &imagePointsCur,
                            float transform[16]) {
        const int n = (int)imagePointsLast.size();
        if ((int)imagePointsCur.size() != n) throw Error(KTLVO_ERR_INVALID, "pointCloudCallback: cloud sizes differ");
        std::vector<Point3f> p3d(n);
        std::vector<Point2f> p2d(n);
        for (int i = 0; i < n; i++) { // :266-281
            const Point3f uvz = {imagePointsLast[i].u, imagePointsLast[i].v, imagePointsLast[i].depth};
            p3d[i] = UnprojectStereo(uvz);
            p2d[i].x = imagePointsCur[i].u; p2d[i].y = imagePointsCur[i].v;
        }
        std::vector<int> inliers;
        estimatePoseRANSAC(p3d, p2d, 2, inliers, cfg_.ransac_iters, cfg_.ransac_reproj, cfg_.ransac_conf, mTcw_local); // :286
        float out[16];
        for (int i = 0; i < 4; i++) // mTcw = mTcw * mTcw_local (:326); cv::gemm sums the f32 products in double
            for (int j = 0; j < 4; j++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += (double)mTcw[4 * i + k] * (double)mTcw_local[4 * k + j];
                out[4 * i + j] = (float)s;
            }
        for (int i = 0; i < 16; i++) { mTcw[i] = out[i]; transform[i] = mTcw_local[i]; } // :316-322
    }

  private:
    void check(int rc, const char *what) const {
        if (rc == KTLVO_OK) return;
        std::string msg = std::string(what) + ": " + ktlvo_strerror(rc);
        const char *hip = ktlvo_last_error(ctx_);
        if (rc == KTLVO_ERR_HIP && hip && *hip) msg += std::string(" (") + hip + ")";
        throw Error(rc, msg);
    }
    // cv::Rodrigues, vector -> matrix
    static void rodrigues(const double r[3], double R[9]) {
        const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (theta < 2.220446049250313e-16) {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1. : 0.;
            return;
        }
        const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
        const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
        const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
        const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
        for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1. : 0.) + c1 * rrt[k] + s * rx[k];
    }

    ktlvo_config cfg_;
    ktlvo_ctx *ctx_;
};

} // namespace ktlvo_host
#endif // KTLVO_HOST_HPP
