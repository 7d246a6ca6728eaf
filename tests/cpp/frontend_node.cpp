// A feature-tracking node written the way the reference writes its own (one frame callback that makes
// eight calcOpticalFlowPyrLK calls, two getOpticalFlow calls, a 20 x 8 detector loop and an ANMS call;
// featureTracking.cpp:339-666), but against include/ktlvo_host.hpp instead of OpenCV.  It is run over a
// recorded stereo sequence next to
//   (B) the library's one-call frame path (Frontend::stereoImageCallback / ktlvo_estimate_pose), which
//       must give byte-identical clouds and transforms: the library's batching and caching may not
//       change a single value, and
//   (C) the CPU oracle's restatement of the same callbacks, which must keep the same features with
//       positions inside the LK tolerance.
// It also checks the error behaviour of the mirror (exceptions where OpenCV would assert).
//
// usage: frontend_node frames.bin width height n_frames out.bin [--no-oracle]
//   frames.bin: n_frames x (left, right) mono8 images, tightly packed
//   out.bin:    per frame  int32 n, int32 key, n x DepthPoint, n x ImagePoint, float32[16] mTcw   (flow A)
// exit status 0 = every check passed.
#include "ktlvo_host.hpp"

#include "../../oracle/ktlvo_oracle.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

using namespace ktlvo_host;

namespace {

int g_failures = 0;
#define EXPECT(cond, ...)                                                                                              \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            ++g_failures;                                                                                              \
            std::fprintf(stderr, "FAIL %s:%d  %s  ", __FILE__, __LINE__, #cond);                                       \
            std::fprintf(stderr, __VA_ARGS__);                                                                         \
            std::fprintf(stderr, "\n");                                                                                \
        }                                                                                                              \
    } while (0)

// The reference's node state (featureTracking.cpp:41-46, 98-100, 122-123) and its callback.
struct Node {
    Frontend &fe;
    Image imageCurLeft, imageCurRight, imageLastLeft, imageLastRight;
    int slotCurLeft = 0, slotCurRight = 1, slotLastLeft = 2, slotLastRight = 3;
    std::vector<ImagePoint> imagePointsCur;
    bool systemInited = false;
    explicit Node(Frontend &f) : fe(f) {}

    bool stereoImageCallback(const uint8_t *left, const uint8_t *right, int cols, int rows, std::vector<DepthPoint> &pointsLast,
                             std::vector<ImagePoint> &pointsCur) {
        const ktlvo_config &c = fe.config();
        pointsLast.clear();
        pointsCur.clear();
        if (!systemInited) {
            imageCurLeft = fe.upload(slotCurLeft, left, cols, rows, cols);
            imageCurRight = fe.upload(slotCurRight, right, cols, rows, cols);
            systemInited = true;
            return false;
        }
        // the current pair becomes the last one; the new frames take the freed slots
        std::swap(slotLastLeft, slotCurLeft);
        std::swap(slotLastRight, slotCurRight);
        imageLastLeft = imageCurLeft;
        imageLastRight = imageCurRight;
        imageCurLeft = fe.upload(slotCurLeft, left, cols, rows, cols);
        imageCurRight = fe.upload(slotCurRight, right, cols, rows, cols);

        std::vector<Point2f> featuresLastLeft;
        std::vector<int> featuresLastLeftIndeces;
        for (const ImagePoint &p : imagePointsCur) {
            featuresLastLeft.push_back({p.u, p.v});
            featuresLastLeftIndeces.push_back(p.ind);
        }
        imagePointsCur.clear();

        bool newKeyFrame = false;
        if ((int)featuresLastLeftIndeces.size() < c.keyframe_min_tracked) {
            const double subregionWidth = (double)(cols - 2 * c.x_boundary) / (double)c.x_subregions;
            const double subregionHeight = (double)(rows - 2 * c.y_boundary) / (double)c.y_subregions;
            std::vector<KeyPoint> featuresLeft;
            for (int i = 0; i < c.y_subregions; i++)
                for (int j = 0; j < c.x_subregions; j++) {
                    const Rect roi = {c.x_boundary + (int)(subregionWidth * j), c.y_boundary + (int)(subregionHeight * i),
                                      (int)subregionWidth, (int)subregionHeight};
                    std::vector<KeyPoint> subFeaturesLeft;
                    fe.detect(imageLastLeft, roi, subFeaturesLeft);
                    for (KeyPoint &k : subFeaturesLeft) {
                        k.pt.x += roi.x;
                        k.pt.y += roi.y;
                        featuresLeft.push_back(k);
                    }
                }
            fe.adaptiveNonMaximalSuppresion(featuresLeft, c.anms_keep);
            featuresLastLeft.clear();
            for (const KeyPoint &k : featuresLeft) featuresLastLeft.push_back(k.pt);
            newKeyFrame = true;
        }

        // circular check: last left -> last right -> cur right against last left -> cur left -> cur right
        std::vector<Point2f> featuresPrevRight, featuresCurLeft, featuresPrevCurRight, featuresCurLeftRight;
        std::vector<unsigned char> status;
        std::vector<float> err;
        fe.calcOpticalFlowPyrLK(imageLastLeft, imageLastRight, featuresLastLeft, featuresPrevRight, status, err);
        fe.calcOpticalFlowPyrLK(imageLastLeft, imageCurLeft, featuresLastLeft, featuresCurLeft, status, err);
        fe.calcOpticalFlowPyrLK(imageLastRight, imageCurRight, featuresPrevRight, featuresPrevCurRight, status, err);
        fe.calcOpticalFlowPyrLK(imageCurLeft, imageCurRight, featuresCurLeft, featuresCurLeftRight, status, err);
        std::vector<int> kept;
        fe.circularCheck(featuresPrevCurRight, featuresCurLeftRight, kept);
        std::vector<Point2f> selectedFeatures;
        std::vector<int> selectedIndeces;
        for (int i : kept) {
            selectedFeatures.push_back(featuresLastLeft[i]);
            if (!newKeyFrame) selectedIndeces.push_back(featuresLastLeftIndeces[i]);
        }

        std::vector<Point2f> featuresTemp, featuresLastLeftTemp, featuresLastRightTemp;
        std::vector<Point2f> featuresLastLeftSelect, featuresCurLeftSelect, featuresLastRightSelect;
        std::vector<int> leftRightIndeces, curLeftIndeces;
        fe.getOpticalFlow(imageLastLeft, imageLastRight, selectedFeatures, featuresTemp, featuresLastLeftTemp,
                          featuresLastRightTemp, featuresTemp, selectedIndeces, leftRightIndeces, 1, false, newKeyFrame);
        fe.getOpticalFlow(imageLastLeft, imageCurLeft, featuresLastLeftTemp, featuresLastRightTemp, featuresLastLeftSelect,
                          featuresCurLeftSelect, featuresLastRightSelect, leftRightIndeces, curLeftIndeces, 5, true, newKeyFrame);

        for (size_t i = 0; i < featuresLastRightSelect.size(); i++) {
            const float disparity = featuresLastLeftSelect[i].x - featuresLastRightSelect[i].x;
            const float depth = (float)(c.bf / disparity);
            const float mThDepth = (float)(c.bf * 70.0f / c.fx);
            if (!(depth > 0 && depth < mThDepth)) continue;
            const int ind = newKeyFrame ? (int)i : curLeftIndeces[i];
            const ImagePoint ip = {featuresCurLeftSelect[i].x, featuresCurLeftSelect[i].y, ind};
            const DepthPoint dp = {featuresLastLeftSelect[i].x, featuresLastLeftSelect[i].y, depth, 0, ind, newKeyFrame ? 1 : 0};
            imagePointsCur.push_back(ip);
            pointsLast.push_back(dp);
            pointsCur.push_back(ip);
        }
        return newKeyFrame;
    }
};

template <class F> int thrown_code(F f) {
    try { f(); } catch (const Error &e) { return e.code; }
    return KTLVO_OK;
}

void check_error_behaviour(Frontend &fe, const uint8_t *img, int W, int H) {
    std::vector<Point2f> p(3, Point2f{40.f, 40.f}), q;
    std::vector<unsigned char> st;
    std::vector<float> er;
    Image built = fe.upload(0, img, W, H, W), other = fe.upload(1, img, W, H, W);
    Image never;
    never.slot = 99;
    EXPECT(thrown_code([&] { fe.calcOpticalFlowPyrLK(built, never, p, q, st, er); }) == KTLVO_ERR_INVALID, "bad slot must throw");
    EXPECT(thrown_code([&] { fe.calcOpticalFlowPyrLK(built, other, p, q, st, er, 15); }) == KTLVO_ERR_UNSUPPORTED, "window 15");
    EXPECT(thrown_code([&] { fe.calcOpticalFlowPyrLK(built, other, p, q, st, er, 21, 4, {3, 30, 0.01}, 4); }) == KTLVO_ERR_UNSUPPORTED,
           "flags");
    EXPECT(thrown_code([&] { fe.upload(0, img, W - 1, H, W); }) == KTLVO_ERR_INVALID, "wrong image size");
    std::vector<KeyPoint> kp;
    EXPECT(thrown_code([&] { fe.detect(built, Rect{W - 10, 0, 59, 39}, kp); }) == KTLVO_ERR_INVALID, "ROI outside the frame");
    float T[16] = {0};
    std::vector<int> inl;
    EXPECT(thrown_code([&] { fe.estimatePoseRANSAC({}, {}, 1, inl, 300, 0.1f, 0.99, T); }) == KTLVO_ERR_UNSUPPORTED, "EPnP flag");
    EXPECT(thrown_code([&] { (void)fe.UnprojectStereo(Point3f{1.f, 1.f, 0.f}); }) == KTLVO_ERR_INVALID, "depth 0");
    // legal edge cases: empty lists everywhere, fewer than 4 correspondences leave the pose alone
    p.clear();
    EXPECT(thrown_code([&] { fe.calcOpticalFlowPyrLK(built, other, p, q, st, er); }) == KTLVO_OK && q.empty(), "empty LK");
    std::vector<Point2f> a, b, d;
    std::vector<int> i1, i2;
    EXPECT(thrown_code([&] { fe.getOpticalFlow(built, other, p, p, a, b, d, i1, i2, 5, true, false); }) == KTLVO_OK && a.empty(), "empty flow");
    for (int i = 0; i < 16; i++) T[i] = (float)i;
    std::vector<Point3f> x3(3, Point3f{0.f, 0.f, 5.f});
    std::vector<Point2f> x2(3, Point2f{600.f, 180.f});
    EXPECT(thrown_code([&] { fe.estimatePoseRANSAC(x3, x2, 2, inl, 300, 0.1f, 0.99, T); }) == KTLVO_OK && inl.empty() && T[5] == 5.f,
           "3 correspondences");
    // identical images: every interior point tracks onto itself
    p.assign(1, Point2f{W * 0.5f, H * 0.5f});
    fe.calcOpticalFlowPyrLK(built, other, p, q, st, er);
    EXPECT(q.size() == 1 && st[0] == 1 && std::fabs(q[0].x - p[0].x) < 1e-3f && std::fabs(q[0].y - p[0].y) < 1e-3f && er[0] == 0.f,
           "self track (%g, %g) status %d err %g", q[0].x, q[0].y, (int)st[0], er[0]);
}

} // namespace

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s frames.bin width height n_frames out.bin [--no-oracle]\n", argv[0]);
        return 2;
    }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), NF = std::atoi(argv[4]);
    const bool with_oracle = !(argc > 6 && std::strcmp(argv[6], "--no-oracle") == 0);
    std::vector<uint8_t> frames((size_t)W * H * 2 * NF);
    {
        std::ifstream in(argv[1], std::ios::binary);
        if (!in.read((char *)frames.data(), (std::streamsize)frames.size())) {
            std::fprintf(stderr, "cannot read %zu bytes from %s\n", frames.size(), argv[1]);
            return 2;
        }
    }
    try {
        ktlvo_config cfg;
        ktlvo_default_config(&cfg);
        cfg.width = W;
        cfg.height = H;
        cfg.lk_sums = KTLVO_LK_SUMS_EXACT; // the order-independent reading of LK's window sums: comparable with the oracle bit for bit
        Frontend feA(&cfg), feB(&cfg), feE(&cfg);
        check_error_behaviour(feE, frames.data(), W, H);

        Node node(feA);
        orc_config oc;
        orc_default_config(&oc);
        oc.width = W; oc.height = H; oc.nthreads = 8;
        orc_tracker *otr = with_oracle ? orc_tracker_create(&oc) : nullptr;
        orc_set_lk_sum_mode(3); // exact window sums on both sides: the clouds must then be the same bytes
        std::vector<orc_depth_point> oLast(cfg.max_points);
        std::vector<orc_image_point> oCur(cfg.max_points);

        std::ofstream out(argv[5], std::ios::binary);
        int keyFrames = 0, nKey = 0, nPlain = 0;
        double msA = 0, msB = 0, msKeyA = 0, msKeyB = 0;
        long tracked = 0;
        double worst = 0;
        for (int f = 0; f < NF; f++) {
            const uint8_t *L = &frames[(size_t)W * H * 2 * f], *R = L + (size_t)W * H;
            std::vector<DepthPoint> lastA, lastB;
            std::vector<ImagePoint> curA, curB;
            const auto t0 = std::chrono::steady_clock::now();
            const bool keyA = node.stereoImageCallback(L, R, W, H, lastA, curA);
            const auto t1 = std::chrono::steady_clock::now();
            const bool keyB = feB.stereoImageCallback(L, R, lastB, curB);
            const auto t2 = std::chrono::steady_clock::now();
            if (std::getenv("KTLVO_NODE_VERBOSE"))
                std::fprintf(stderr, "frame %d key %d: call by call %.3f ms, one call %.3f ms\n", f, (int)keyA,
                             std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
            if (f > 2) { // the first calls allocate
                (keyA ? msKeyA : msA) += std::chrono::duration<double, std::milli>(t1 - t0).count();
                (keyA ? msKeyB : msB) += std::chrono::duration<double, std::milli>(t2 - t1).count();
                (keyA ? nKey : nPlain) += 1;
            }
            EXPECT(keyA == keyB, "frame %d", f);
            EXPECT(lastA.size() == lastB.size() && curA.size() == curB.size(), "frame %d: %zu vs %zu", f, lastA.size(), lastB.size());
            if (lastA.size() == lastB.size() && !lastA.empty()) {
                EXPECT(std::memcmp(lastA.data(), lastB.data(), lastA.size() * sizeof(DepthPoint)) == 0, "frame %d last cloud", f);
                EXPECT(std::memcmp(curA.data(), curB.data(), curA.size() * sizeof(ImagePoint)) == 0, "frame %d cur cloud", f);
            }
            keyFrames += keyA;
            tracked += (long)lastA.size();
            if (f > 0) EXPECT(lastA.size() > 100, "frame %d keeps only %zu features", f, lastA.size());

            float transformA[16], TlB[16], TgB[16];
            int ninl = 0;
            feA.pointCloudCallback(lastA, curA, transformA);
            const int rc = ktlvo_estimate_pose(feB.context(), lastB.data(), curB.data(), (int)lastB.size(), TlB, TgB, &ninl);
            EXPECT(rc == KTLVO_OK, "frame %d estimate_pose %d", f, rc);
            EXPECT(std::memcmp(transformA, TlB, sizeof(TlB)) == 0 && std::memcmp(feA.mTcw, TgB, sizeof(TgB)) == 0, "frame %d pose", f);

            if (otr) {
                int okey = 0;
                const int on = orc_track_stereo_frame(otr, L, R, oLast.data(), oCur.data(), cfg.max_points, &okey);
                EXPECT(on == (int)lastA.size() && (okey != 0) == keyA, "frame %d oracle keeps %d, node %zu", f, on, lastA.size());
                if (on == (int)lastA.size()) {
                    for (int i = 0; i < on; i++) {
                        EXPECT(oLast[i].ind == lastA[i].ind && oCur[i].ind == curA[i].ind, "frame %d point %d ids", f, i);
                        const double d = std::fmax(std::fmax(std::fabs(oLast[i].u - lastA[i].u), std::fabs(oLast[i].v - lastA[i].v)),
                                                   std::fmax(std::fabs(oCur[i].u - curA[i].u), std::fabs(oCur[i].v - curA[i].v)));
                        worst = std::fmax(worst, d);
                    }
                    static_assert(sizeof(orc_depth_point) == sizeof(DepthPoint) && sizeof(orc_image_point) == sizeof(ImagePoint), "same PODs");
                    EXPECT(on == 0 || (std::memcmp(oLast.data(), lastA.data(), (size_t)on * sizeof(DepthPoint)) == 0 &&
                                       std::memcmp(oCur.data(), curA.data(), (size_t)on * sizeof(ImagePoint)) == 0),
                           "frame %d: clouds differ from the oracle's", f);
                }
            }
            const int32_t hdr[2] = {(int32_t)lastA.size(), keyA ? 1 : 0};
            out.write((const char *)hdr, sizeof(hdr));
            out.write((const char *)lastA.data(), (std::streamsize)(lastA.size() * sizeof(DepthPoint)));
            out.write((const char *)curA.data(), (std::streamsize)(curA.size() * sizeof(ImagePoint)));
            out.write((const char *)feA.mTcw, sizeof(feA.mTcw));
        }
        if (otr) orc_tracker_destroy(otr);
        EXPECT(keyFrames >= 1, "no key frame in %d frames", NF);
        EXPECT(worst == 0.0, "worst node-vs-oracle deviation %g px", worst);
        if (nPlain) std::printf("per frame, call by call %.3f ms, one call %.3f ms (%d frames)\n", msA / nPlain, msB / nPlain, nPlain);
        if (nKey) std::printf("per key frame, call by call %.3f ms, one call %.3f ms (%d frames)\n", msKeyA / nKey, msKeyB / nKey, nKey);
        std::printf("frontend_node: %d frames, %d key frames, %ld tracked points, worst |node - oracle| = %.5f px, %d failures\n", NF,
                    keyFrames, tracked, worst, g_failures);
    } catch (const Error &e) {
        std::fprintf(stderr, "ktlvo_host::Error %d: %s\n", e.code, e.what());
        return 3;
    }
    return g_failures ? 1 : 0;
}
