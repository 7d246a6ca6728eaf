/*
 * ktlvo_oracle.h -- C interface of the CPU oracle for the ktl_vo feature-tracking path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under slam_amd/ or include/ may include, link or load
 * this library; only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py do,
 * and only as the checker or as the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of the path lives in OpenCV 3.x (modules video, imgproc,
 * features2d, calib3d), which is not vendored in the reference tree, not installed here and
 * not fetchable.  The reference holds no tests, fixtures or golden vectors.  This oracle is a
 * restatement of the published OpenCV 3.x algorithms at the reference's call sites
 * (src/ktl_vo/src/featureTracking.cpp:205,214,457-479,103,422; visualOdometry.cpp:206) and a
 * literal restatement of the reference-owned code (featureTracking.cpp:223-248, 258-311,
 * 486-501, 526-573; visualOdometry.cpp:153-167, 311-326).  It is pinned only by analytic
 * known-answer tests (tests/test_oracle_kat.py), not by outputs of a real OpenCV.
 */
#ifndef KTLVO_ORACLE_H
#define KTLVO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_pyramid orc_pyramid;

/* cv::pyrDown, 5x5 [1 4 6 4 1]^2/256, reflect-101, dst = ((w+1)/2, (h+1)/2). */
void orc_pyr_down(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);

/* calcSharrDeriv of lkpyramid.cpp: interleaved int16 (dx,dy), reflect-101 inside the image. */
void orc_scharr(const uint8_t *src, int w, int h, int sstride, int16_t *dst /* 2*w*h */);

/* buildOpticalFlowPyramid(img, win, maxLevel, withDerivatives=false, REFLECT_101) plus the
 * per-level Scharr buffers calcOpticalFlowPyrLK derives for its "prev" image. */
orc_pyramid *orc_pyramid_create(const uint8_t *img, int w, int h, int stride, int max_level,
                                int win, int with_deriv);
void orc_pyramid_destroy(orc_pyramid *p);
int orc_pyramid_levels(const orc_pyramid *p); /* highest level index actually built */
int orc_pyramid_level_size(const orc_pyramid *p, int level, int *w, int *h);
/* copy level image (unpadded, tightly packed w*h) */
int orc_pyramid_level_copy(const orc_pyramid *p, int level, uint8_t *dst);
int orc_pyramid_deriv_copy(const orc_pyramid *p, int level, int16_t *dst /* 2*w*h */);

/* Order of the float additions inside orc_pyrlk's window sums (process-wide): 0 = one accumulator in
 * raster order (default; LKTrackerInvoker's scalar path), 1 = four accumulators in turn along a row (a
 * 4-lane SIMD build), 2 = pairwise.  For measuring the sensitivity of LK results to rounding order. */
/* Threads the image-wide stages (pyrDown, border copies, calcSharrDeriv) of the calls below are cut over
 * (process-wide, default 1).  Results do not depend on it: the work is integer and rows are independent. */
void orc_set_threads(int n);
void orc_set_lk_sum_mode(int mode);
int orc_get_lk_sum_mode(void);

/* cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(win,win), maxLevel,
 * TermCriteria(COUNT|EPS, max_count, epsilon), flags=0, minEigThreshold).
 * featureTracking.cpp:205-209.  epsilon is the TermCriteria value (0.01); it is squared
 * internally as OpenCV does.  prev must have been created with with_deriv=1.
 * iters_out (optional, n*(levels)) receives iterations per level for statistics.
 * nthreads: points are range-split per level like cv::parallel_for_. */
int orc_pyrlk(const orc_pyramid *prev, const orc_pyramid *next, const float *prev_xy, int n,
              float *next_xy, uint8_t *status, float *err, int win, int max_level,
              int max_count, double epsilon, double min_eig_thr, int nthreads);

/* Diagnostics: while set, orc_pyrlk records the Gauss-Newton iteration count of every point and
 * level into buf[level * n_points + point] (single-threaded calls only). NULL switches it off. */
void orc_set_iter_log(uint8_t *buf, int n_points);

/* cornerHarris / cornerMinEigenVal (blockSize 3, Sobel 3, 8-bit) on the ROI (rx,ry,rw,rh) of
 * a parent image; Sobel reads parent pixels beyond the ROI (reflect-101 only at the parent's
 * edges), the 3x3 box sum reflects at the ROI's own edges. out is rw*rh floats. */
int orc_corner_response(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw,
                        int rh, int use_harris, double k, float *out);

/* cv::goodFeaturesToTrack on that ROI view (featureTracking.cpp:103, 421-422).  Coordinates
 * are ROI-relative integer-valued floats.  Ties in the response sort are broken by higher
 * raster address first.  Returns the number of corners (<= max_corners, <= cap). */
int orc_good_features(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw,
                      int rh, int max_corners, double quality, double min_dist,
                      int use_harris, double k, float *xy, float *resp, int cap);

/* adaptiveNonMaximalSuppresion, featureTracking.cpp:258-311 (in place; returns new n).
 * perm_out (optional, n ints) receives for each kept output the input index. */
int orc_anms(float *xy, float *resp, int n, int num_to_keep, int *perm_out);

/* getOpticalFlow's keep predicate, featureTracking.cpp:223-248: keep[i] in {0,1}. */
int orc_fb_gate(const float *f1, const float *f2, const float *f1chk, int n, int img_w,
                int img_h, int x_boundary, int y_boundary, double max_track_dis,
                int y_threshold, uint8_t *keep);
/* circular check, featureTracking.cpp:486-501. */
int orc_circular_gate(const float *prev_cur_right, const float *cur_left_right, int n,
                      uint8_t *keep);
/* disparity -> depth gate, featureTracking.cpp:528-573. depth_out[i] is bf/disp. */
int orc_depth_gate(const float *last_left, const float *last_right, int n, double bf,
                   double fx, uint8_t *keep, float *depth_out);

/* cv::Rodrigues both ways (3-vector <-> row-major 3x3), double. */
void orc_rodrigues_v2m(const double r[3], double R[9]);
void orc_rodrigues_m2v(const double R[9], double r[3]);

/* P3P on 3 correspondences (Gao et al., ktlvo_oracle_p3p.cpp): up to 4 (R row-major 9, t 3)
 * solutions. K = fx,fy,cx,cy. */
int orc_p3p(const double X[9], const double uv[6], const double K[4], double R[36],
            double t[12]);
/* solvePnP(..., SOLVEPNP_P3P) on exactly 4 points (4th disambiguates). returns 1/0. */
int orc_solve_p3p4(const double X[12], const double uv[8], const double K[4], double R[9],
                   double t[3]);

/* Test hooks of the P3P: the Gao quartic's coefficients (highest power first) for
 * a = |P1P2|^2/|P0P1|^2, b = |P0P2|^2/|P0P1|^2, p, q, r = twice the cosines between the bearings
 * (1,2), (0,2), (0,1); and the closed-form real roots of a quartic (count returned). */
void orc_gao_quartic(double a, double b, double p, double q, double r, double coef[5]);
int orc_roots4(const double coef[5], double roots[4]);
/* Newton steps applied to each root of the Gao quartic before it is used (process-wide; default 0 =
 * the closed-form roots as they are, as in OpenCV 3.x). */
void orc_p3p_set_polish(int newton_steps);

/* PnPRansacCallback::computeError + findInliers for one model: projectPoints in double,
 * float projections, float squared error, err <= thr_sq.  Rt = R(9 row-major) then t(3).
 * Returns the inlier count; mask (optional) n bytes; err_out (optional) n floats. */
int orc_project_score(const float *X3, const float *x2, int n, const double Rt[12],
                      const double K[4], float thr_sq, uint8_t *mask, float *err_out);

/* cv::solvePnPRansac(P3d, p2d, K, 0, rvec, tvec, false, iters, reproj, conf, inliers, P3P)
 * as the reference calls it (visualOdometry.cpp:206-208): 4-point P3P kernel, RNG((uint64)-1),
 * adaptive iteration count, output = best hypothesis.  Returns 1 on success, 0 otherwise
 * (rvec/tvec untouched on failure). hyp_log (optional) receives per executed iteration:
 * 4 indices, valid flag, inlier count (6 ints each; cap max_iters entries).*/
int orc_solve_pnp_ransac(const float *X3, const float *x2, int n, const double K[4],
                         int max_iters, float reproj_err, double confidence, double rvec[3],
                         double tvec[3], uint8_t *inlier_mask, int *n_inliers,
                         int *iters_run, int *hyp_log);

/* ---- frame-level restatement of the two reference callbacks ---------------------------- */
typedef struct orc_tracker orc_tracker;

typedef struct {
    int width, height;
    double fx, fy, cx, cy, bf;
    int x_subregions, y_subregions, max_per_subregion; /* 20, 8, 100 */
    int x_boundary, y_boundary;                         /* 30, 30 */
    double max_track_dis;                               /* 100 */
    int win, max_level, max_count;                      /* 21, 4, 30 */
    double epsilon, min_eig_thr;                        /* 0.01, 1e-4 */
    double quality, min_dist;                           /* 0.01, 10 */
    int use_harris;                                     /* 1 */
    double harris_k;                                    /* 0.0 */
    int anms_keep;                                      /* 1000 */
    int keyframe_min_tracked;                           /* 300 */
    int ransac_iters;                                   /* 300 */
    float ransac_reproj;                                /* 0.1f */
    double ransac_conf;                                 /* (double)0.99f */
    int nthreads;
} orc_config;

typedef struct { float u, v; int ind; } orc_image_point;                       /* pointDefinition.h:18-21 */
typedef struct { float u, v, depth; int label, ind, newKeyFrame; } orc_depth_point; /* :28-34 */

void orc_default_config(orc_config *c);
orc_tracker *orc_tracker_create(const orc_config *c);
void orc_tracker_destroy(orc_tracker *t);
/* stereoImageCallback, featureTracking.cpp:339-666. Returns the number of point pairs
 * written (0 for the very first frame), or <0 on error. cap = capacity of both arrays. */
int orc_track_stereo_frame(orc_tracker *t, const uint8_t *left, const uint8_t *right,
                           orc_depth_point *last, orc_image_point *cur, int cap,
                           int *new_key_frame);
/* pointCloudCallback, visualOdometry.cpp:240-356: T_local and T_global are row-major 4x4 f32. */
int orc_estimate_pose(orc_tracker *t, const orc_depth_point *last, const orc_image_point *cur,
                      int n, float T_local[16], float T_global[16]);

#ifdef __cplusplus
}
#endif
#endif
