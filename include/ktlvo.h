/*
 * ktlvo.h -- C ABI of libktlvo_hip.so: the MI355X (gfx950) implementation of the ktl_vo
 * feature-tracking hot path (scofield1991/slam, src/ktl_vo).
 *
 * The reference has no FFI or plugin layer for this path; it reaches the arithmetic through
 * C++ free functions and OpenCV calls.  Each entry point below names the reference interface
 * (file:line under src/ktl_vo/src/) it stands behind.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns KTLVO_OK (0) or a negative ktlvo_status; nothing throws;
 *   - the caller owns every host buffer; device memory is owned by the context;
 *   - *_dev variants take DEVICE pointers, enqueue on the context's stream and return
 *     without synchronising; the others take HOST pointers and return after the result
 *     has been copied back;
 *   - a ktlvo_ctx is not thread-safe (the reference's callbacks are serialised by
 *     ros::spin(), featureTracking.cpp:727); use one context per GPU and host thread;
 *   - a call makes its context's device the calling thread's current HIP device (and leaves it
 *     so), so one thread may drive contexts on several GPUs in turn;
 *   - point lists are interleaved float (x,y) like std::vector<cv::Point2f>;
 *   - empty inputs (n == 0) are legal everywhere and return KTLVO_OK.
 */
#ifndef KTLVO_H
#define KTLVO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    KTLVO_OK = 0,
    KTLVO_ERR_INVALID = -1,     /* bad argument (null pointer, size mismatch, bad slot) */
    KTLVO_ERR_UNSUPPORTED = -2, /* parameter value the HIP path does not implement */
    KTLVO_ERR_HIP = -3,         /* a HIP runtime call failed; see ktlvo_last_error() */
    KTLVO_ERR_CAPACITY = -4,    /* more points / ROIs / hypotheses than the context was sized for */
    KTLVO_ERR_STATE = -5,       /* call order (e.g. tracking a slot whose pyramid was not built) */
    KTLVO_ERR_NOMEM = -6
} ktlvo_status;

typedef struct ktlvo_ctx ktlvo_ctx;

/* Run-time form of the reference's compile-time constants. */
typedef struct {
    int width, height;             /* cameraParameters.h:16-17 (1242 x 375) */
    double fx, fy, cx, cy;         /* cameraParameters.h:19-22 */
    double bf;                     /* cameraParameters.h:27 */
    int num_slots;                 /* device image slots (each holds one padded pyramid) */
    int max_level;                 /* featureTracking.cpp:81  lktPyramid = 4 (levels 0..4) */
    int win;                       /* featureTracking.cpp:80  winSize = 21 */
    int max_points;                /* capacity of one point list */
    int max_batch_pairs;           /* capacity of ktlvo_pyrlk_batch_dev (default and upper limit 256) */
    int x_subregions, y_subregions;/* featureTracking.cpp:69-70 (20, 8) */
    int max_per_subregion;         /* featureTracking.cpp:68 (100) */
    int x_boundary, y_boundary;    /* featureTracking.cpp:74-75 (30, 30) */
    double max_track_dis;          /* featureTracking.cpp:79 (100) */
    int anms_keep;                 /* featureTracking.cpp:433 (1000) */
    int keyframe_min_tracked;      /* featureTracking.cpp:406 (300) */
    int ransac_iters;              /* visualOdometry.cpp:125 (300) */
    float ransac_reproj;           /* visualOdometry.cpp:126 (0.1f) */
    double ransac_conf;            /* visualOdometry.cpp:127 ((double)0.99f) */
    int max_hypotheses;            /* capacity of ktlvo_ransac_score */
    int device_p3p;                /* 0: P3P hypotheses solved on the host (bit-parity with a host-side
                                      solvePnPRansac); 1: solved by a HIP kernel, one thread per hypothesis
                                      (same algorithm, device libm: poses agree to ~1e-12) */
    int lk_sums;                   /* how calcOpticalFlowPyrLK's five window sums per level are rounded when a call's
                                      ktlvo_lk_params.sums is KTLVO_LK_SUMS_DEFAULT: KTLVO_LK_SUMS_TREE (the default,
                                      also for 0) or KTLVO_LK_SUMS_EXACT */
    int p3p_polish;                /* Newton steps on the roots of the P3P quartic: 0 = none, the closed-form
                                      roots of OpenCV 3's p3p.cpp as visualOdometry.cpp:206-208 reaches it */
} ktlvo_config;

/* The float accumulators of LKTrackerInvoker (A11, A12, A22, b1, b2: 441 integer-valued terms each) depend on
 * the order of addition, which no parallel implementation shares with a serial one.
 *   KTLVO_LK_SUMS_TREE (the default): per-lane integer partials added as floats in a fixed 64-lane tree;
 *     deterministic, within float rounding of any other order of addition, OpenCV's serial one included.
 *   KTLVO_LK_SUMS_EXACT: every sum is the exact integer total of its terms, rounded to float once (to nearest
 *     even).  Order-independent, so any implementation that sums exactly reproduces positions, status and
 *     error bit for bit (the CPU oracle's matching mode does); four vector instructions more per Gauss-Newton
 *     step, 5 to 6 % of a large launch's time. */
enum { KTLVO_LK_SUMS_DEFAULT = 0, KTLVO_LK_SUMS_TREE = 1, KTLVO_LK_SUMS_EXACT = 2 };

/* cv::calcOpticalFlowPyrLK arguments as passed at featureTracking.cpp:205-209. */
typedef struct {
    int win;            /* 21 (the only window the HIP kernel is specialised for) */
    int max_level;      /* 4 */
    int max_count;      /* TermCriteria count, 30 */
    double epsilon;     /* TermCriteria epsilon, 0.01 (squared internally, as OpenCV does) */
    double min_eig_thr; /* minEigThreshold default 1e-4 */
    int sums;           /* KTLVO_LK_SUMS_*; 0 = the context's ktlvo_config.lk_sums */
} ktlvo_lk_params;

/* cv::GFTTDetector::create(100, 0.01, 10, 3, true, 0.0), featureTracking.cpp:103. */
typedef struct {
    int max_corners;    /* 100 per ROI */
    double quality;     /* 0.01 */
    double min_dist;    /* 10 */
    int block_size;     /* 3 (only value implemented) */
    int use_harris;     /* 1 */
    double harris_k;    /* 0.0 */
} ktlvo_gftt_params;

typedef struct { int x, y, w, h; } ktlvo_roi;                                   /* cv::Rect */
typedef struct { float u, v; int ind; } ktlvo_image_point;                      /* pointDefinition.h:18-21 */
typedef struct { float u, v, depth; int label, ind, newKeyFrame; } ktlvo_depth_point; /* pointDefinition.h:28-34 */

/* Kernel identifiers for ktlvo_profile_get. */
enum { KTLVO_K_PYRDOWN = 0, KTLVO_K_PYRLK = 1, KTLVO_K_CORNER = 2, KTLVO_K_RANSAC = 3, KTLVO_K_SCHARR = 4,
       KTLVO_K_COUNT = 5 };

/* ---- life cycle ----------------------------------------------------------------------- */
/* Fills the reference's constants (cameraParameters.h:16-27, featureTracking.cpp:68-81,
 * visualOdometry.cpp:122-127) and default capacities. */
int ktlvo_default_config(ktlvo_config *cfg);
/* Replaces the reference's process globals (featureTracking.cpp:24-123) with one object. */
ktlvo_ctx *ktlvo_create(const ktlvo_config *cfg, int device_id);
void ktlvo_destroy(ktlvo_ctx *ctx);
/* Use the caller's hipStream_t (NULL = the context's own stream) for all later work. */
int ktlvo_set_stream(ktlvo_ctx *ctx, void *hip_stream);
int ktlvo_sync(ktlvo_ctx *ctx);
const char *ktlvo_strerror(int status);
const char *ktlvo_last_error(const ktlvo_ctx *ctx); /* text of the last HIP failure */

/* ---- images and pyramids ---------------------------------------------------------------- */
/* cv_bridge::toCvCopy(msg, MONO8) -> Mat (featureTracking.cpp:346-365): host image into a slot.
 * Pageable memory is copied before the call returns.  Pinned memory (hipHostMalloc, hipHostRegister)
 * is read by DMA after the call returns: leave it unchanged until a call that waits for the device
 * (ktlvo_sync, or any host-pointer call that returns results, ktlvo_track_stereo_frame included). */
int ktlvo_upload_image(ktlvo_ctx *ctx, int slot, const uint8_t *img, int w, int h, int stride);
/* The same for n frames at once (n <= 256), frame i starting image_stride bytes after frame i - 1 and going
 * into slots[i]: one DMA and one launch for the whole batch instead of one each per frame, which is what
 * a frame-pair stream needs to use the host link at its rate (the reference receives its frames one
 * ROS message at a time, featureTracking.cpp:339-347; a replay harness or a multi-camera rig has them in
 * bulk).  Source memory as for ktlvo_upload_image. */
int ktlvo_upload_images(ktlvo_ctx *ctx, const int *slots, int n, const uint8_t *imgs, int w, int h,
                        int stride, size_t image_stride);
/* Same from a device buffer (frame already resident in HBM). */
int ktlvo_set_image_dev(ktlvo_ctx *ctx, int slot, const uint8_t *d_img, int w, int h, int stride);
/* cv::buildOpticalFlowPyramid as called inside calcOpticalFlowPyrLK (featureTracking.cpp:205):
 * pyrDown per level, reflect-101 border of `win` pixels.  Level count follows OpenCV's rule
 * (stop when the next level is not larger than the window).  The per-level Scharr planes that
 * calcOpticalFlowPyrLK derives for its first image (calcSharrDeriv, zero border) are built in
 * the same call, so any slot can serve as `prev`. */
int ktlvo_build_pyramid(ktlvo_ctx *ctx, int slot, int max_level, int win);
/* The same for several slots in one set of launches. */
int ktlvo_build_pyramids(ktlvo_ctx *ctx, const int *slots, int n_slots, int max_level, int win);
/* Test hooks: copy one level (tightly packed w*h) / its Scharr derivative (calcSharrDeriv,
 * interleaved int16 dx,dy) back to the host. */
int ktlvo_pyramid_levels(ktlvo_ctx *ctx, int slot, int *max_level_built);
int ktlvo_download_level(ktlvo_ctx *ctx, int slot, int level, uint8_t *dst, int *w, int *h);
int ktlvo_download_scharr(ktlvo_ctx *ctx, int slot, int level, int16_t *dst);

/* ---- pyramidal Lucas-Kanade ------------------------------------------------------------- */
/* cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(21,21), 4,
 * TermCriteria(COUNT|EPS,30,0.01), 0): featureTracking.cpp:205-209, 214-218, 457-479.
 * Both slots must hold built pyramids. */
int ktlvo_pyrlk(ktlvo_ctx *ctx, int prev_slot, int next_slot, const float *prev_xy, int n,
                float *next_xy, uint8_t *status, float *err, const ktlvo_lk_params *p);
int ktlvo_pyrlk_dev(ktlvo_ctx *ctx, int prev_slot, int next_slot, const float *d_prev_xy, int n,
                    float *d_next_xy, uint8_t *d_status, float *d_err, const ktlvo_lk_params *p);
/* n_pairs independent calls in one launch; point arrays are [n_pairs][n_per_pair]. */
int ktlvo_pyrlk_batch_dev(ktlvo_ctx *ctx, const int *prev_slots, const int *next_slots,
                          int n_pairs, const float *d_prev_xy, int n_per_pair, float *d_next_xy,
                          uint8_t *d_status, float *d_err, const ktlvo_lk_params *p);

/* ---- corner detection --------------------------------------------------------------------- */
/* gftt->detect(image(roi), keypoints) for a list of ROIs of one slot's level-0 image
 * (featureTracking.cpp:408-430).  Output for ROI r starts at xy[2*r*cap_per_roi] /
 * response[r*cap_per_roi]; coordinates are ROI-relative like cv::KeyPoint::pt; counts[r] is
 * the number of corners of ROI r.  cap_per_roi >= params->max_corners. */
int ktlvo_detect_gftt(ktlvo_ctx *ctx, int slot, const ktlvo_roi *rois, int n_roi,
                      const ktlvo_gftt_params *params, float *xy, float *response, int *counts,
                      int cap_per_roi);
/* cv::goodFeaturesToTrack on the same region of n_frames slots in one call (roi NULL = the whole frame): what
 * featureTracking.cpp:102-103, 421-422 does per image, for a batch of images, as the frame-pair stream and
 * batch workloads need it (2000 to 16384 corners on whole frames).  Every kernel of the whole-frame path takes the
 * frame from its grid, so a batch of KITTI-sized frames costs three launches (candidates, a sort and a greedy pass
 * with a workgroup per frame each); up to 16 such frames, or up to 48 larger ones, take the greedy pass as a parallel
 * relaxation over all candidates instead, and frames with more than 16384 corners above the threshold are sorted by
 * ranges of the response.  Per frame the result is ktlvo_detect_gftt's for that slot and region whatever the route.
 * Outputs [n_frames][cap][2], [n_frames][cap], [n_frames]. */
int ktlvo_detect_gftt_frames(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi *roi,
                             const ktlvo_gftt_params *params, float *xy, float *response, int *counts, int cap);
/* The same with DEVICE output arrays: corners go straight into ktlvo_pyrlk_batch_dev without a trip through
 * host memory.  The arrays are written in the order of the context's stream, so any later call on the context
 * finds them complete.  (The call itself waits for the device once or twice: at its end, to learn whether the
 * route it queued blind stands or a frame has to go through the global sort instead, and for frames with more
 * than 512K pixels before that for the candidate counts that size its lists.) */
int ktlvo_detect_gftt_frames_dev(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi *roi,
                                 const ktlvo_gftt_params *params, float *d_xy, float *d_response, int *d_counts, int cap);
/* Corners of one batch of slots and pyramids of another (usually overlapping) batch in one call: what a frame-pair
 * stream does with every batch of new frames (featureTracking.cpp:408-430 detects on the new key frame;
 * calcOpticalFlowPyrLK, :457-479, builds the pyramids it tracks on).  The two are independent and load the GPU in
 * complementary ways, so the build (ktlvo_build_pyramids of build_slots) runs on a side stream BESIDE the detection
 * (ktlvo_detect_gftt_frames_dev of detect_slots) and the context's stream waits for both: results are those of the two
 * calls made one after the other, a following ktlvo_pyrlk_batch_dev finds corners and pyramids complete. */
int ktlvo_detect_build_frames_dev(ktlvo_ctx *ctx, const int *detect_slots, int n_frames, const ktlvo_roi *roi,
                                  const ktlvo_gftt_params *params, float *d_xy, float *d_response, int *d_counts,
                                  int cap, const int *build_slots, int n_build, int max_level, int win);
/* The frame-pair stream's whole step in one call (featureTracking.cpp:408-430 detect, :457-479 track):
 * ktlvo_detect_build_frames_dev(prev_slots, ..., build_slots) followed by ktlvo_pyrlk_batch_dev(prev_slots, next_slots,
 * d_xy, cap, ...), with the same results.  The LK launch is queued behind the detection before the host waits for the
 * detection's flags, so the device does not idle for that round trip; the call returns once the corners are final, the
 * tracking results are complete in the order of the context's stream.  cap is the corner capacity per frame and the
 * point count per pair (a frame with fewer corners tracks its list's unused tail too; d_counts says how many are real). */
int ktlvo_detect_build_track_dev(ktlvo_ctx *ctx, const int *prev_slots, const int *next_slots, int n_pairs,
                                 const ktlvo_roi *roi, const ktlvo_gftt_params *params, float *d_xy, float *d_response,
                                 int *d_counts, int cap, const int *build_slots, int n_build, float *d_next_xy,
                                 uint8_t *d_status, float *d_err, const ktlvo_lk_params *lk);
/* The same detection for callers that must not allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  It calls no hipMalloc,
 * hipFree or other allocating routine, copies nothing and does not synchronise: the ROI list travels as
 * kernel arguments, the kernels' work memory is a fixed array inside the library (4 MiB + 8 KiB per GPU,
 * held until the process ends) and the outputs are DEVICE arrays of the caller: d_xy [n_roi][cap][2],
 * d_response [n_roi][cap], d_counts [n_roi], valid once the context's stream has been waited for.
 * Limits: n_roi <= 192, ROI areas summing to <= 524288 pixels, every ROI <= 4096 pixels and <= 1024
 * min-distance cells (the reference's 59 x 39 sub-regions are 2301 pixels); beyond them the call returns
 * KTLVO_ERR_CAPACITY and enqueues nothing.  Because the call never waits for the device it cannot report what only the
 * kernels see: d_counts[roi] = -1 marks an ROI whose selection ran out of a kernel limit (more than four corners in one
 * minimum-distance cell; the distance rule itself excludes that, and the sizes are checked before anything is queued,
 * so no accepted call is known to produce it).  All calls on one GPU share the array: calls that overlap on
 * one GPU (two streams, two contexts, two processes' worth of threads) race; calls in stream order do
 * not, every call clears what it accumulates into. */
int ktlvo_detect_gftt_noalloc(ktlvo_ctx *ctx, int slot, const ktlvo_roi *rois, int n_roi,
                              const ktlvo_gftt_params *params, float *d_xy, float *d_response,
                              int *d_counts, int cap_per_roi);
/* Test hook: cornerHarris / cornerMinEigenVal response of one ROI (rw*rh floats). */
int ktlvo_corner_response(ktlvo_ctx *ctx, int slot, const ktlvo_roi *roi, int use_harris,
                          double harris_k, float *out);
/* adaptiveNonMaximalSuppresion(keypoints, numToKeep), featureTracking.cpp:258-311; in place. */
int ktlvo_anms(ktlvo_ctx *ctx, float *xy, float *response, int *n_inout, int num_to_keep);

/* ---- gates ---------------------------------------------------------------------------------- */
/* Keep predicate of getOpticalFlow, featureTracking.cpp:223-248.  keep_idx receives the kept
 * input indices in input order (order-preserving compaction), n_keep their number. */
int ktlvo_fb_gate(ktlvo_ctx *ctx, const float *f1, const float *f2, const float *f1_checked, int n,
                  int y_threshold, int *keep_idx, int *n_keep);
/* Circular stereo-temporal check, featureTracking.cpp:486-501. */
int ktlvo_circular_gate(ktlvo_ctx *ctx, const float *prev_cur_right, const float *cur_left_right,
                        int n, int *keep_idx, int *n_keep);
/* getOpticalFlow, featureTracking.cpp:181-256: forward LK, backward LK and the gate in one
 * call; outputs are the kept (f1, f2) pairs and their input indices, in input order. */
int ktlvo_get_optical_flow(ktlvo_ctx *ctx, int slot_1, int slot_2, const float *features_1, int n,
                           int y_threshold, float *f1_sel, float *f2_sel, int *keep_idx,
                           int *n_keep);

/* ---- RANSAC pose ---------------------------------------------------------------------------- */
/* PnPRansacCallback::computeError + findInliers for H models at once (the inner loop of
 * cv::solvePnPRansac, visualOdometry.cpp:206-208).  Rt holds H x (R row-major 9, t 3) doubles,
 * K = fx,fy,cx,cy; counts[h] = #{ |x - proj(R X + t)|^2 <= thr_sq }.  If best_mask is not
 * NULL it receives the n-byte inlier mask of hypothesis mask_hyp. */
int ktlvo_ransac_score(ktlvo_ctx *ctx, const float *X3, const float *x2, int n, const double *Rt,
                       int H, const double K[4], float thr_sq, int *counts, int mask_hyp,
                       uint8_t *best_mask);
/* The same on device-resident arrays, enqueued on the context's stream without synchronising; no
 * capacity limits apply (the caller owns every buffer: d_counts H ints, d_mask n bytes or NULL). */
int ktlvo_ransac_score_dev(ktlvo_ctx *ctx, const float *d_X3, const float *d_x2, int n,
                           const double *d_Rt, int H, const double K[4], float thr_sq, int *d_counts,
                           int mask_hyp, uint8_t *d_mask);
/* cv::solvePnPRansac(P3d, p2d, K, 0, rvec, tvec, false, iters, reproj, conf, inliers, P3P)
 * as called by estimatePoseRANSAC (visualOdometry.cpp:177-217): hypotheses are drawn with
 * cv::RNG((uint64)-1), solved by P3P, scored in one batch on the GPU, and OpenCV's sequential
 * acceptance rule is replayed over the counts.  Returns KTLVO_OK and *found = 1/0. */
int ktlvo_solve_pnp_ransac(ktlvo_ctx *ctx, const float *X3, const float *x2, int n, int max_iters,
                           float reproj_err, double confidence, double rvec[3], double tvec[3],
                           uint8_t *inlier_mask, int *n_inliers, int *found);

/* ---- frame level ---------------------------------------------------------------------------- */
/* stereoImageCallback(left, right), featureTracking.cpp:339-666: returns in *n the number of
 * (DepthPoint, ImagePoint) pairs written (what the reference publishes on /image_points_last
 * and /image_points_cur); 0 for the first frame. cap = capacity of both arrays. */
int ktlvo_track_stereo_frame(ktlvo_ctx *ctx, const uint8_t *left, const uint8_t *right,
                             ktlvo_depth_point *last, ktlvo_image_point *cur, int cap, int *n,
                             int *new_key_frame);
/* pointCloudCallback(last, cur), visualOdometry.cpp:240-356: T_local is the reference's
 * transform_msg.transform (row-major 4x4 f32, msg/transform_msg.msg), T_global its mTcw. */
int ktlvo_estimate_pose(ktlvo_ctx *ctx, const ktlvo_depth_point *last, const ktlvo_image_point *cur,
                        int n, float T_local[16], float T_global[16], int *n_inliers);

/* ---- wire format ---------------------------------------------------------------------------- */
/* ROS 1 serialisation (little endian, as roscpp puts it on the wire) of the three messages the
 * reference publishes, so that its downstream nodes can consume this library's output unchanged:
 *   /image_points_last  sensor_msgs/PointCloud2 of DepthPoint  featureTracking.cpp:656-659, visualOdometry.cpp:332-335
 *   /image_points_cur   sensor_msgs/PointCloud2 of ImagePoint  featureTracking.cpp:661-664, visualOdometry.cpp:337-340
 *   /local_transform    ktl_vo/transform_msg                   visualOdometry.cpp:316-330, msg/transform_msg.msg:1-2
 * The point clouds are what pcl::toROSMsg makes of a pcl::PointCloud filled by push_back: height 1,
 * width n, one field per registered member (pointDefinition.h:23-26, 36-42; float -> FLOAT32 = 7,
 * int -> INT32 = 5, count 1), little endian, point_step = sizeof(point), row_step = point_step * n,
 * data = the point array, is_dense = 1.  Host-only functions: no context, no GPU work. */
typedef struct {
    uint32_t seq;          /* roscpp numbers messages per publisher; the reference leaves it to roscpp */
    uint32_t stamp_sec;    /* header.stamp, see ktlvo_ros_time_from_sec */
    uint32_t stamp_nsec;
    const char *frame_id;  /* NULL = "" (the reference never sets it) */
} ktlvo_ros_header;
/* ros::Time().fromSec(t) as the reference stamps its messages (featureTracking.cpp:658). t >= 0. */
int ktlvo_ros_time_from_sec(double t, uint32_t *sec, uint32_t *nsec);
/* Serialised sizes in bytes (0 if an argument is invalid). */
size_t ktlvo_wire_size_image_points(int n, const ktlvo_ros_header *h);
size_t ktlvo_wire_size_depth_points(int n, const ktlvo_ros_header *h);
size_t ktlvo_wire_size_transform(const ktlvo_ros_header *h);
/* Write the message body into out[0 .. cap); *written = bytes used.  KTLVO_ERR_CAPACITY if cap is
 * too small (nothing is written then). */
int ktlvo_wire_image_points(const ktlvo_image_point *pts, int n, const ktlvo_ros_header *h, uint8_t *out,
                            size_t cap, size_t *written);
int ktlvo_wire_depth_points(const ktlvo_depth_point *pts, int n, const ktlvo_ros_header *h, uint8_t *out,
                            size_t cap, size_t *written);
/* T = the T_local of ktlvo_estimate_pose, row-major (visualOdometry.cpp:316-322). */
int ktlvo_wire_transform(const float T[16], const ktlvo_ros_header *h, uint8_t *out, size_t cap, size_t *written);

/* ---- profiling ------------------------------------------------------------------------------ */
/* With profiling on, every launch of the kernels listed in the enum above is bracketed by
 * hipEvents on the context's stream; ktlvo_profile_get synchronises and returns the summed
 * device time and launch count since the last ktlvo_profile_reset. */
/* on: 0 = off, 1 = every kernel, otherwise a mask with bit (1 + kernel id) set for each kernel to
 * time (an event pair costs a few microseconds of stream time, which matters to a step of a few
 * launches). */
int ktlvo_profile_enable(ktlvo_ctx *ctx, int on);
int ktlvo_profile_reset(ktlvo_ctx *ctx);
int ktlvo_profile_get(ktlvo_ctx *ctx, int kernel_id, double *total_ms, long long *launches);

#ifdef __cplusplus
}
#endif
#endif /* KTLVO_H */
