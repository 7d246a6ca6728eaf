/*
 * ktlvo_oracle_pose.cpp -- CPU oracle for the pose side of the ktl_vo path and the frame-level
 * restatement of the reference's two callbacks.
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED: see ktlvo_oracle.h.  Build with -ffp-contract=off.
 *
 * Restated here:
 *   cv::solvePnPRansac -> RANSACPointSetRegistrator::run / PnPRansacCallback / projectPoints /
 *   Rodrigues, as reached from visualOdometry.cpp:177-217 (parameters :122-127);
 *   UnprojectStereo :153-167, pose chaining :311-326;
 *   stereoImageCallback featureTracking.cpp:339-666.
 * The minimal solver (orc_p3p, orc_solve_p3p4) is the Gao et al. P3P in ktlvo_oracle_p3p.cpp.
 */
#include "ktlvo_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

extern "C" void orc_rodrigues_v2m(const double r_in[3], double R[9]) {
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
    rx *= it; ry *= it; rz *= it;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int i = 0; i < 9; i++) R[i] = c * ((i % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[i] + s * r_x[i];
}

extern "C" void orc_rodrigues_m2v(const double R[9], double r[3]) {
    /* cvRodrigues2 matrix branch without its SVD re-orthonormalisation (inputs here are
     * orthonormal to rounding). */
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    double s = std::sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { x = y = z = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; x = std::sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5; y = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; z = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (std::fabs(x) < std::fabs(y) && std::fabs(x) < std::fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= std::sqrt(x * x + y * y + z * z);
            x *= theta; y *= theta; z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth; y *= vth; z *= vth;
    }
    r[0] = x; r[1] = y; r[2] = z;
}

extern "C" int orc_project_score(const float *X3, const float *x2, int n, const double Rt[12],
                                 const double K[4], float thr_sq, uint8_t *mask, float *err_out) {
    const double *R = Rt, *t = Rt + 9;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    int good = 0;
    for (int i = 0; i < n; i++) {
        double X = X3[3 * i], Y = X3[3 * i + 1], Z = X3[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        float mx = (float)(x * fx + cx), my = (float)(y * fy + cy);
        float dx = x2[2 * i] - mx, dy = x2[2 * i + 1] - my;
        float e = dx * dx + dy * dy;
        int f = e <= thr_sq;
        if (mask) mask[i] = (uint8_t)f;
        if (err_out) err_out[i] = e;
        good += f;
    }
    return good;
}

namespace {

struct CvRng { /* cv::RNG */
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

}  // namespace

extern "C" int orc_solve_pnp_ransac(const float *X3, const float *x2, int n, const double K[4], int max_iters,
                                    float reproj_err, double confidence, double rvec[3], double tvec[3],
                                    uint8_t *inlier_mask, int *n_inliers, int *iters_run, int *hyp_log) {
    const int modelPoints = 4;
    if (n_inliers) *n_inliers = 0;
    if (iters_run) *iters_run = 0;
    if (n < modelPoints) return 0;
    const double threshold = reproj_err; /* float -> double, then squared, then float */
    const float thr_sq = (float)(threshold * threshold);
    int niters = std::max(max_iters, 1);
    CvRng rng((uint64_t)-1);
    std::vector<uint8_t> mask(n), bestMask(n, 0);
    double bestR[3] = {0, 0, 0}, bestT[3] = {0, 0, 0};
    int maxGood = 0;
    int it = 0;
    for (it = 0; it < niters; it++) {
        int idx[4];
        if (n > modelPoints) {
            for (int i = 0; i < modelPoints;) {
                int j;
                for (;;) {
                    idx[i] = rng.uniform(0, n);
                    for (j = 0; j < i; j++) if (idx[i] == idx[j]) break;
                    if (j == i) break;
                }
                i++;
            }
        } else {
            for (int i = 0; i < 4; i++) idx[i] = i;
        }
        double Xs[12], us[8], R[9], t[3];
        for (int i = 0; i < 4; i++) {
            for (int k = 0; k < 3; k++) Xs[3 * i + k] = X3[3 * idx[i] + k];
            us[2 * i] = x2[2 * idx[i]]; us[2 * i + 1] = x2[2 * idx[i] + 1];
        }
        int ok = orc_solve_p3p4(Xs, us, K, R, t);
        int good = 0;
        double rv[3] = {0, 0, 0};
        if (ok) {
            /* model is stored as (rvec, tvec); computeError re-derives R from rvec */
            orc_rodrigues_m2v(R, rv);
            double Rt[12];
            orc_rodrigues_v2m(rv, Rt);
            Rt[9] = t[0]; Rt[10] = t[1]; Rt[11] = t[2];
            good = orc_project_score(X3, x2, n, Rt, K, thr_sq, mask.data(), nullptr);
        }
        if (hyp_log) {
            int *h = hyp_log + 6 * it;
            h[0] = idx[0]; h[1] = idx[1]; h[2] = idx[2]; h[3] = idx[3]; h[4] = ok; h[5] = good;
        }
        if (!ok) continue;
        if (good > std::max(maxGood, modelPoints - 1)) {
            mask.swap(bestMask);
            memcpy(bestR, rv, sizeof(bestR));
            memcpy(bestT, t, sizeof(bestT));
            maxGood = good;
            niters = ransac_update_num_iters(confidence, (double)(n - good) / n, modelPoints, niters);
        }
    }
    if (iters_run) *iters_run = it;
    if (maxGood <= 0) return 0;
    memcpy(rvec, bestR, sizeof(bestR));
    memcpy(tvec, bestT, sizeof(bestT));
    if (inlier_mask) memcpy(inlier_mask, bestMask.data(), n);
    if (n_inliers) *n_inliers = maxGood;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Frame-level restatement
 * ---------------------------------------------------------------------------------------- */
extern "C" void orc_default_config(orc_config *c) {
    memset(c, 0, sizeof(*c));
    c->width = 1242; c->height = 375;                       /* cameraParameters.h:16-17 */
    c->fx = 707.0912; c->fy = 707.0912; c->cx = 601.8873; c->cy = 183.1104; /* :19-22 */
    c->bf = 386.1448;                                       /* :27 */
    c->x_subregions = 20; c->y_subregions = 8; c->max_per_subregion = 100; /* featureTracking.cpp:68-70 */
    c->x_boundary = 30; c->y_boundary = 30;                 /* :74-75 */
    c->max_track_dis = 100;                                 /* :79 */
    c->win = 21; c->max_level = 4; c->max_count = 30;       /* :80-81, :208-209 */
    c->epsilon = 0.01; c->min_eig_thr = 1e-4;
    c->quality = 0.01; c->min_dist = 10; c->use_harris = 1; c->harris_k = 0.0; /* :103 */
    c->anms_keep = 1000;                                    /* :433 */
    c->keyframe_min_tracked = 300;                          /* :406 */
    c->ransac_iters = 300; c->ransac_reproj = 0.1f;         /* visualOdometry.cpp:125-126 */
    c->ransac_conf = (double)0.99f;                         /* :127, float widened to double */
    c->nthreads = 1;
}

struct orc_tracker {
    orc_config cfg;
    bool inited = false;
    std::vector<uint8_t> curL, curR, lastL, lastR;
    std::vector<orc_image_point> imagePointsCur;
    float mTcw[16], mTcwLocal[16];
};

extern "C" orc_tracker *orc_tracker_create(const orc_config *c) {
    orc_tracker *t = new orc_tracker;
    t->cfg = *c;
    for (int i = 0; i < 16; i++) t->mTcw[i] = t->mTcwLocal[i] = (i % 5 == 0) ? 1.f : 0.f;
    return t;
}
extern "C" void orc_tracker_destroy(orc_tracker *t) { delete t; }

namespace {
struct P2 { float x, y; };

void lk_call(const orc_config &c, const orc_pyramid *a, const orc_pyramid *b, const std::vector<P2> &in,
             std::vector<P2> &out) {
    out.resize(in.size());
    std::vector<uint8_t> st(in.size());
    std::vector<float> er(in.size());
    orc_pyrlk(a, b, (const float *)in.data(), (int)in.size(), (float *)out.data(), st.data(), er.data(), c.win,
              c.max_level, c.max_count, c.epsilon, c.min_eig_thr, c.nthreads);
}

/* getOpticalFlow, featureTracking.cpp:181-256 */
void get_optical_flow(const orc_config &c, const orc_pyramid *im1, const orc_pyramid *im2,
                      const std::vector<P2> &f1, const std::vector<P2> &fdepth, std::vector<P2> &f1sel,
                      std::vector<P2> &f2sel, std::vector<P2> &f3sel, const std::vector<int> &idx,
                      std::vector<int> &idxsel, int y_threshold, bool temporal, bool newKeyFrame) {
    std::vector<P2> f2, f1chk;
    lk_call(c, im1, im2, f1, f2);
    lk_call(c, im2, im1, f2, f1chk);
    std::vector<uint8_t> keep(f1.size());
    orc_fb_gate((const float *)f1.data(), (const float *)f2.data(), (const float *)f1chk.data(), (int)f1.size(),
                c.width, c.height, c.x_boundary, c.y_boundary, c.max_track_dis, y_threshold, keep.data());
    for (size_t i = 0; i < f1.size(); i++)
        if (keep[i]) {
            f1sel.push_back(f1[i]);
            f2sel.push_back(f2[i]);
            if (!newKeyFrame) idxsel.push_back(idx[i]);
            if (temporal) f3sel.push_back(fdepth[i]);
        }
}
}  // namespace

extern "C" int orc_track_stereo_frame(orc_tracker *t, const uint8_t *left, const uint8_t *right,
                                      orc_depth_point *last, orc_image_point *cur, int cap, int *new_key_frame) {
    const orc_config &c = t->cfg;
    const size_t npx = (size_t)c.width * c.height;
    if (new_key_frame) *new_key_frame = 0;
    if (!t->inited) { /* :349-356 */
        t->curL.assign(left, left + npx);
        t->curR.assign(right, right + npx);
        t->inited = true;
        return 0;
    }
    t->lastL.swap(t->curL); t->lastR.swap(t->curR); /* :361-365 */
    t->curL.assign(left, left + npx);
    t->curR.assign(right, right + npx);

    std::vector<P2> featuresLastLeft;
    std::vector<int> featuresLastLeftIndeces;
    for (auto &p : t->imagePointsCur) { /* :385-389 */
        featuresLastLeft.push_back({p.u, p.v});
        featuresLastLeftIndeces.push_back(p.ind);
    }
    t->imagePointsCur.clear();

    bool newKeyFrame = false;
    if ((int)featuresLastLeft.size() < c.keyframe_min_tracked) { /* :406-441 */
        const double subW = (double)(c.width - 2 * c.x_boundary) / (double)c.x_subregions;
        const double subH = (double)(c.height - 2 * c.y_boundary) / (double)c.y_subregions;
        std::vector<float> kxy, kresp;
        std::vector<float> sxy(2 * c.max_per_subregion), sresp(c.max_per_subregion);
        for (int i = 0; i < c.y_subregions; i++)
            for (int j = 0; j < c.x_subregions; j++) {
                int sl = c.x_boundary + (int)(subW * j), st = c.y_boundary + (int)(subH * i);
                int n = orc_good_features(t->lastL.data(), c.width, c.height, c.width, sl, st, (int)subW, (int)subH,
                                          c.max_per_subregion, c.quality, c.min_dist, c.use_harris, c.harris_k,
                                          sxy.data(), sresp.data(), c.max_per_subregion);
                for (int k = 0; k < n; k++) {
                    kxy.push_back(sxy[2 * k] + sl);
                    kxy.push_back(sxy[2 * k + 1] + st);
                    kresp.push_back(0.f); /* OpenCV 3 GFTTDetector: KeyPoint(pt, blockSize), response 0 */
                }
            }
        int n = orc_anms(kxy.data(), kresp.data(), (int)kresp.size(), c.anms_keep, nullptr);
        featuresLastLeft.clear();
        for (int k = 0; k < n; k++) featuresLastLeft.push_back({kxy[2 * k], kxy[2 * k + 1]});
        newKeyFrame = true;
    }
    if (new_key_frame) *new_key_frame = newKeyFrame ? 1 : 0;

    orc_pyramid *pLL = orc_pyramid_create(t->lastL.data(), c.width, c.height, c.width, c.max_level, c.win, 1);
    orc_pyramid *pLR = orc_pyramid_create(t->lastR.data(), c.width, c.height, c.width, c.max_level, c.win, 1);
    orc_pyramid *pCL = orc_pyramid_create(t->curL.data(), c.width, c.height, c.width, c.max_level, c.win, 1);
    orc_pyramid *pCR = orc_pyramid_create(t->curR.data(), c.width, c.height, c.width, c.max_level, c.win, 1);

    /* circular check :457-501 */
    std::vector<P2> featuresPrevRight, featuresCurLeft, featuresCurLeftRight, featuresPrevCurRight;
    lk_call(c, pLL, pLR, featuresLastLeft, featuresPrevRight);
    lk_call(c, pLL, pCL, featuresLastLeft, featuresCurLeft);
    lk_call(c, pLR, pCR, featuresPrevRight, featuresPrevCurRight);
    lk_call(c, pCL, pCR, featuresCurLeft, featuresCurLeftRight);
    std::vector<uint8_t> keep(featuresLastLeft.size());
    orc_circular_gate((const float *)featuresPrevCurRight.data(), (const float *)featuresCurLeftRight.data(),
                      (int)keep.size(), keep.data());
    std::vector<P2> selectedFeatures;
    std::vector<int> selectedIndeces, leftRightIndeces, curLeftIndeces;
    for (size_t i = 0; i < keep.size(); i++)
        if (keep[i]) {
            selectedFeatures.push_back(featuresLastLeft[i]);
            if (!newKeyFrame) selectedIndeces.push_back(featuresLastLeftIndeces[i]);
        }

    std::vector<P2> featuresTemp, featuresLastLeftTemp, featuresLastRightTemp, featuresLastLeftSelect,
        featuresCurLeftSelect, featuresLastRightSelect;
    get_optical_flow(c, pLL, pLR, selectedFeatures, featuresTemp, featuresLastLeftTemp, featuresLastRightTemp,
                     featuresTemp, selectedIndeces, leftRightIndeces, 1, false, newKeyFrame); /* :509 */
    get_optical_flow(c, pLL, pCL, featuresLastLeftTemp, featuresLastRightTemp, featuresLastLeftSelect,
                     featuresCurLeftSelect, featuresLastRightSelect, leftRightIndeces, curLeftIndeces, 5, true,
                     newKeyFrame); /* :513 */
    orc_pyramid_destroy(pLL); orc_pyramid_destroy(pLR); orc_pyramid_destroy(pCL); orc_pyramid_destroy(pCR);

    /* :526-573 */
    int nout = 0;
    const int m = (int)featuresLastRightSelect.size();
    std::vector<uint8_t> dkeep(m);
    std::vector<float> depth(m);
    orc_depth_gate((const float *)featuresLastLeftSelect.data(), (const float *)featuresLastRightSelect.data(), m,
                   c.bf, c.fx, dkeep.data(), depth.data());
    for (int i = 0; i < m; i++) {
        if (!dkeep[i]) continue;
        int ind = i;
        if (!newKeyFrame) ind = curLeftIndeces[i];
        orc_image_point ip{featuresCurLeftSelect[i].x, featuresCurLeftSelect[i].y, ind};
        t->imagePointsCur.push_back(ip);
        if (nout < cap) {
            orc_depth_point dp{featuresLastLeftSelect[i].x, featuresLastLeftSelect[i].y, depth[i], 0, ind,
                               (int)newKeyFrame};
            last[nout] = dp;
            cur[nout] = ip;
        }
        nout++;
    }
    return nout <= cap ? nout : -2;
}

extern "C" int orc_estimate_pose(orc_tracker *t, const orc_depth_point *last, const orc_image_point *cur, int n,
                                 float T_local[16], float T_global[16]) {
    const orc_config &c = t->cfg;
    std::vector<float> X3((size_t)3 * n), x2((size_t)2 * n);
    for (int i = 0; i < n; i++) { /* UnprojectStereo with mRwc = I, mOw = 0, visualOdometry.cpp:153-167 */
        const float z = last[i].depth, u = last[i].u, v = last[i].v;
        const float x = (float)((u - c.cx) * z / c.fx);
        const float y = (float)((v - c.cy) * z / c.fy);
        X3[3 * i] = x; X3[3 * i + 1] = y; X3[3 * i + 2] = z;
        x2[2 * i] = cur[i].u; x2[2 * i + 1] = cur[i].v;
    }
    const double K[4] = {c.fx, c.fy, c.cx, c.cy};
    double rvec[3], tvec[3];
    int ninl = 0, iters = 0;
    int ok = orc_solve_pnp_ransac(X3.data(), x2.data(), n, K, c.ransac_iters, c.ransac_reproj, c.ransac_conf, rvec,
                                  tvec, nullptr, &ninl, &iters, nullptr);
    if (ok) { /* Rodrigues + set_T_matrix (copyTo converts 64F -> 32F), :210-214 */
        double R[9];
        orc_rodrigues_v2m(rvec, R);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) t->mTcwLocal[4 * i + j] = (float)R[3 * i + j];
            t->mTcwLocal[4 * i + 3] = (float)tvec[i];
        }
    } /* on failure the previous mTcw_local stays in place */
    float out[16];
    for (int i = 0; i < 4; i++) /* mTcw = mTcw * mTcw_local, :326; cv::gemm accumulates f32 in double */
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)t->mTcw[4 * i + k] * (double)t->mTcwLocal[4 * k + j];
            out[4 * i + j] = (float)s;
        }
    memcpy(t->mTcw, out, sizeof(out));
    memcpy(T_local, t->mTcwLocal, sizeof(out));
    memcpy(T_global, t->mTcw, sizeof(out));
    return ok ? ninl : -1;
}
