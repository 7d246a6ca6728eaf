// Frame-level host logic: the control flow of the reference's two callbacks over the device ops.
//   stereoImageCallback   featureTracking.cpp:339-666
//   pointCloudCallback    visualOdometry.cpp:240-356 (estimatePoseRANSAC :177-217)
//
// Differences from the reference that do not change results: each image's pyramid is built once
// and the cur -> last hand-over is a slot swap (the reference rebuilds 16 pyramids per frame for 4
// distinct images); the 8 LK calls of a frame become 2 launches (see the frame function) with the
// gating and compaction after them done on the device, so the host synchronises once per frame.
#include "ctx.hpp"
#include "internal.hpp"

#include <cstring>

namespace ktlvo {

struct TrackerState {
    bool inited = false;
    int sCurL = 0, sCurR = 1, sLastL = 2, sLastR = 3; // slot roles, swapped every frame
    std::vector<ktlvo_image_point> imagePointsCur;     // featureTracking.cpp:122
    float mTcw[16], mTcwLocal[16];                      // visualOdometry.cpp:113-114
    // device work lists (max_points points each)
    float *d_buf = nullptr;
    // What the host reads back each frame, contiguous so that one copy fetches it: 8 counters, then for
    // the frame's N start points the three gates' kept-index lists (N ints each) and the three final
    // point lists (2N floats each).  h_io is its pinned host image; the start points go up through it too.
    int *d_out = nullptr;
    int *h_io = nullptr;
    enum { A = 0, BC = 1 /* two lists */, DE = 3 /* four lists */, NLIST = 7 };
    static size_t out_ints(size_t n) { return 8 + 9 * n; }
    float *list(int i, int cap) const { return d_buf + (size_t)i * cap * 2; }
};

void tracker_free(ktlvo_ctx *ctx) {
    if (!ctx->tracker) return;
    if (ctx->tracker->d_buf) (void)hipFree(ctx->tracker->d_buf);
    if (ctx->tracker->d_out) (void)hipFree(ctx->tracker->d_out);
    if (ctx->tracker->h_io) (void)hipHostFree(ctx->tracker->h_io);
    delete ctx->tracker;
    ctx->tracker = nullptr;
}

static int tracker_get(ktlvo_ctx *ctx, TrackerState **out) {
    if (!ctx->tracker) {
        if (ctx->cfg.num_slots < 4) return KTLVO_ERR_CAPACITY;
        TrackerState *t = new TrackerState;
        for (int i = 0; i < 16; i++) t->mTcw[i] = t->mTcwLocal[i] = (i % 5 == 0) ? 1.f : 0.f;
        const size_t cap = ctx->cfg.max_points;
        ctx->tracker = t;
        if (hipMalloc((void **)&t->d_buf, (size_t)TrackerState::NLIST * cap * 8) != hipSuccess ||
            hipMalloc((void **)&t->d_out, TrackerState::out_ints(cap) * sizeof(int)) != hipSuccess ||
            hipHostMalloc((void **)&t->h_io, TrackerState::out_ints(cap) * sizeof(int), hipHostMallocDefault) != hipSuccess) {
            tracker_free(ctx);
            return KTLVO_ERR_NOMEM;
        }
    }
    *out = ctx->tracker;
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_track_stereo_frame(ktlvo_ctx *ctx, const uint8_t *left, const uint8_t *right,
                                        ktlvo_depth_point *last, ktlvo_image_point *cur, int cap, int *n_out,
                                        int *new_key_frame) {
    if (!ctx || !left || !right || !n_out || cap < 0 || (cap > 0 && (!last || !cur))) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    *n_out = 0;
    if (new_key_frame) *new_key_frame = 0;
    TrackerState *t;
    int rc = tracker_get(ctx, &t);
    if (rc != KTLVO_OK) return rc;
    const ktlvo_config &c = ctx->cfg;
    const int W = c.width, H = c.height;

    // Once upload_pair has queued the ingest launch, the device may be reading the caller's (pinned) frames, and the
    // pinned block h_io may be in flight: every return after that point comes after the stream has been waited for, the
    // failing ones too, so that the buffers are the caller's again whatever the call returns.
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(ctx->stream);
        return code;
    };
    if (!t->inited) { // :349-356
        if ((rc = upload_pair(ctx, t->sCurL, t->sCurR, left, right, W, H)) != KTLVO_OK) return fail(rc);
        const int s[2] = {t->sCurL, t->sCurR};
        if ((rc = ktlvo_build_pyramids(ctx, s, 2, c.max_level, c.win)) != KTLVO_OK) return fail(rc);
        // the caller may reuse (pinned) frame buffers as soon as this call returns
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        t->inited = true;
        return KTLVO_OK;
    }
    // :361-365: last <- cur, then the new frame becomes cur.  The new roles and the new track list live in
    // locals until the frame's device work has succeeded (commit below): a call that fails on the way leaves
    // the tracker as it was -- the previous frame still "cur", its track list intact -- so the failed frame
    // is dropped instead of silently restarting the track ids.  (The slots the new images go into held the
    // frame before the previous one, which nothing refers to any more.)
    const int sLL = t->sCurL, sLR = t->sCurR, sCL = t->sLastL, sCR = t->sLastR;
    std::vector<ktlvo_image_point> newPointsCur;
    auto commit = [&]() {
        t->sLastL = sLL; t->sLastR = sLR; t->sCurL = sCL; t->sCurR = sCR;
        t->imagePointsCur.swap(newPointsCur);
    };
    if ((rc = upload_pair(ctx, sCL, sCR, left, right, W, H)) != KTLVO_OK) return fail(rc);
    {
        const int s[2] = {sCL, sCR};
        if ((rc = ktlvo_build_pyramids(ctx, s, 2, c.max_level, c.win)) != KTLVO_OK) return fail(rc);
    }

    std::vector<float> featuresLastLeft; // :385-389
    std::vector<int> featuresLastLeftIndeces;
    featuresLastLeft.reserve(t->imagePointsCur.size() * 2);
    for (const ktlvo_image_point &p : t->imagePointsCur) {
        featuresLastLeft.push_back(p.u);
        featuresLastLeft.push_back(p.v);
        featuresLastLeftIndeces.push_back(p.ind);
    }

    bool newKeyFrame = false;
    if ((int)featuresLastLeftIndeces.size() < c.keyframe_min_tracked) { // :406-441
        const double subW = (double)(W - 2 * c.x_boundary) / (double)c.x_subregions;
        const double subH = (double)(H - 2 * c.y_boundary) / (double)c.y_subregions;
        const int n_roi = c.x_subregions * c.y_subregions;
        std::vector<ktlvo_roi> rois(n_roi);
        for (int i = 0; i < c.y_subregions; i++)
            for (int j = 0; j < c.x_subregions; j++)
                rois[i * c.x_subregions + j] = {c.x_boundary + (int)(subW * j), c.y_boundary + (int)(subH * i), (int)subW,
                                                (int)subH};
        const int capr = c.max_per_subregion;
        std::vector<float> xy((size_t)n_roi * capr * 2), resp((size_t)n_roi * capr);
        std::vector<int> counts(n_roi);
        const ktlvo_gftt_params gp = {capr, 0.01, 10, 3, 1, 0.0}; // GFTTDetector::create(100,0.01,10,3,true,0.0), :103
        rc = ktlvo_detect_gftt(ctx, sLL, rois.data(), n_roi, &gp, xy.data(), resp.data(), counts.data(), capr);
        if (rc != KTLVO_OK) return fail(rc);
        std::vector<float> kxy, kresp;
        for (int r = 0; r < n_roi; r++)
            for (int k = 0; k < counts[r]; k++) { // :423-428
                kxy.push_back(xy[((size_t)r * capr + k) * 2] + rois[r].x);
                kxy.push_back(xy[((size_t)r * capr + k) * 2 + 1] + rois[r].y);
                kresp.push_back(0.f); // OpenCV 3 GFTTDetector: KeyPoint(pt, blockSize), response stays 0
            }
        int nk = (int)kresp.size();
        if (nk > c.max_points) return fail(KTLVO_ERR_CAPACITY);
        rc = ktlvo_anms(ctx, kxy.data(), kresp.data(), &nk, c.anms_keep); // :433
        if (rc != KTLVO_OK) return fail(rc);
        featuresLastLeft.assign(kxy.begin(), kxy.begin() + 2 * (size_t)nk); // :434
        newKeyFrame = true;
    }
    if (new_key_frame) *new_key_frame = newKeyFrame ? 1 : 0;

    const int N = (int)(featuresLastLeft.size() / 2);
    if (N > c.max_points) return fail(KTLVO_ERR_CAPACITY);
    if (N == 0) { // nothing to track: the uploads still have to finish before the caller's buffers are free again
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        commit();
        return KTLVO_OK;
    }
    const int CAP = c.max_points;
    typedef TrackerState TS;
    float *dA = t->list(TS::A, CAP), *dBC = t->list(TS::BC, CAP), *dDE = t->list(TS::DE, CAP);
    float *dB = dBC, *dC = dBC + 2 * (size_t)N;
    float *dD = dDE, *dE = dD + 2 * (size_t)N, *dCHK = dE + 2 * (size_t)N, *dCHKB = dCHK + 2 * (size_t)N;
    int *dCnt = t->d_out, *dK1 = dCnt + 8, *dK2 = dK1 + N, *dK3 = dK2 + N;
    float *dU1 = (float *)(dK3 + N), *dU2 = dU1 + 2 * (size_t)N, *dU3 = dU2 + 2 * (size_t)N;
    const ktlvo_lk_params lk = {c.win, c.max_level, 30, 0.01, 1e-4, KTLVO_LK_SUMS_DEFAULT}; // :195-197, :208-209

    // The reference makes 8 LK calls per frame.  A point's track depends on nothing but its two images
    // and its start, so the two forward calls inside getOpticalFlow (:205) repeat, on a subset, what
    // the circular check's first two calls (:457-467) computed, and the two backward calls (:214) start
    // from those same forward tracks: they can be made for every point before the circular check has
    // chosen, together with the check's own second pair of calls.  Two launches (2 and 4 image pairs over
    // all N points) and one gate launch, same values; the tracks of points a gate drops are wasted work
    // on a GPU that a frame's few thousand points leave mostly idle, and the chain of dependent launches
    // that sets the frame's latency is two tracks long instead of three.
    memcpy(t->h_io, featuresLastLeft.data(), (size_t)N * 8); // the previous frame's read of h_io ended with its sync
    {
        const hipError_t e = hipMemcpyAsync(dA, t->h_io, (size_t)N * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(set_hip_error(ctx, e, "hipMemcpyAsync(points)", __FILE__, __LINE__));
    }
    // With the reference's five pyramid levels all six tracks come from one launch (pyrlk_frame_chain: a wave per
    // chain of two tracks, so the frame waits for its slowest chain rather than for the slowest track of each stage
    // in turn); otherwise from a launch per stage.
    float *const chain_out[6] = {dB, dC, dD, dE, dCHK, dCHKB};
    rc = ctx->frame_chain ? launch_frame_chain(ctx, sLL, sLR, sCL, sCR, dA, N, chain_out, &lk) : KTLVO_ERR_UNSUPPORTED;
    if (rc == KTLVO_ERR_UNSUPPORTED) {
        { // circular check :457-479: LL->LR and LL->CL share their input list
            PairList pl;
            pl.prev[0] = (uint16_t)sLL; pl.next[0] = (uint16_t)sLR;
            pl.prev[1] = (uint16_t)sLL; pl.next[1] = (uint16_t)sCL;
            const int po[2] = {0, 0}, no[2] = {0, N};
            rc = launch_pyrlk_ex(ctx, pl, 2, dA, po, dBC, no, ctx->d_status, ctx->d_err, N, nullptr, &lk);
            if (rc != KTLVO_OK) return fail(rc);
        }
        { // LR->CR from featuresPrevRight, CL->CR from featuresCurLeft (:462-467); and the backward tracks of both
          // getOpticalFlow stages (:214 as reached from :509 and :513): LR->LL and CL->LL from the same two lists
            PairList pl;
            pl.prev[0] = (uint16_t)sLR; pl.next[0] = (uint16_t)sCR;
            pl.prev[1] = (uint16_t)sCL; pl.next[1] = (uint16_t)sCR;
            pl.prev[2] = (uint16_t)sLR; pl.next[2] = (uint16_t)sLL;
            pl.prev[3] = (uint16_t)sCL; pl.next[3] = (uint16_t)sLL;
            const int po[4] = {0, N, 0, N}, no[4] = {0, N, 2 * N, 3 * N};
            rc = launch_pyrlk_ex(ctx, pl, 4, dBC, po, dDE, no, ctx->d_status, ctx->d_err, N, nullptr, &lk);
        }
    }
    if (rc != KTLVO_OK) return fail(rc);
    // :486-501 circular check, :509 getOpticalFlow(LastLeft, LastRight, y threshold 1), :513 getOpticalFlow(LastLeft,
    // CurLeft, y threshold 5): kept-index lists of the three and the survivors' featuresLastLeftSelect,
    // featuresCurLeftSelect, featuresLastRightSelect
    rc = launch_frame_gates(ctx, dA, dB, dC, dD, dE, dCHK, dCHKB, N, dK1, dK2, dK3, dCnt, dU1, dU2, dU3);
    if (rc != KTLVO_OK) return fail(rc);
    // One copy and one synchronisation for the whole frame (36 N + 32 bytes into pinned memory).
    {
        const hipError_t e = hipMemcpyAsync(t->h_io, t->d_out, TS::out_ints(N) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) return fail(set_hip_error(ctx, e, "hipMemcpyAsync(results)", __FILE__, __LINE__));
    }
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int *cnt = t->h_io, *K1 = cnt + 8, *K2 = K1 + N, *K3 = K2 + N;
    const float *U1 = (const float *)(K3 + N), *U2 = U1 + 2 * (size_t)N, *U3 = U2 + 2 * (size_t)N;
    const int n3 = cnt[2];

    // :526-573 disparity -> depth and output assembly
    int nout = 0;
    for (int i = 0; i < n3; i++) {
        const float disparity = U1[2 * i] - U3[2 * i];
        const float depth = (float)(c.bf / disparity);
        const float mThDepth = (float)(c.bf * 70.0f / c.fx);
        if (!(depth > 0 && depth < mThDepth)) continue;
        int ind = i;
        if (!newKeyFrame) ind = featuresLastLeftIndeces[K1[K2[K3[i]]]]; // curLeftIndeces[i]
        ktlvo_image_point ip = {U2[2 * i], U2[2 * i + 1], ind};
        newPointsCur.push_back(ip);
        if (nout < cap) {
            ktlvo_depth_point dp = {U1[2 * i], U1[2 * i + 1], depth, 0, ind, newKeyFrame ? 1 : 0};
            last[nout] = dp;
            cur[nout] = ip;
        }
        ++nout;
    }
    commit(); // the frame is done; output arrays that are too small do not undo it
    *n_out = nout < cap ? nout : cap;
    return nout <= cap ? KTLVO_OK : KTLVO_ERR_CAPACITY;
}

extern "C" int ktlvo_estimate_pose(ktlvo_ctx *ctx, const ktlvo_depth_point *last, const ktlvo_image_point *cur, int n,
                                   float T_local[16], float T_global[16], int *n_inliers) {
    if (!ctx || n < 0 || !T_local || !T_global || (n > 0 && (!last || !cur))) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    TrackerState *t;
    int rc = tracker_get(ctx, &t);
    if (rc != KTLVO_OK) return rc;
    const ktlvo_config &c = ctx->cfg;
    std::vector<float> X3((size_t)3 * n), x2((size_t)2 * n);
    for (int i = 0; i < n; i++) { // UnprojectStereo with mRwc = I, mOw = 0 (:153-167, :266-281)
        const float z = last[i].depth, u = last[i].u, v = last[i].v;
        X3[3 * i] = (float)((u - c.cx) * z / c.fx);
        X3[3 * i + 1] = (float)((v - c.cy) * z / c.fy);
        X3[3 * i + 2] = z;
        x2[2 * i] = cur[i].u;
        x2[2 * i + 1] = cur[i].v;
    }
    double rvec[3], tvec[3];
    int ninl = 0, found = 0;
    rc = ktlvo_solve_pnp_ransac(ctx, X3.data(), x2.data(), n, c.ransac_iters, c.ransac_reproj, c.ransac_conf, rvec, tvec,
                                nullptr, &ninl, &found);
    if (rc != KTLVO_OK) return rc;
    if (found) { // Rodrigues + set_T_matrix; Mat::copyTo converts 64F -> 32F (:210-214)
        double R[9];
        rodrigues_v2m(rvec, R);
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) t->mTcwLocal[4 * i + j] = (float)R[3 * i + j];
            t->mTcwLocal[4 * i + 3] = (float)tvec[i];
        }
    } // otherwise mTcw_local keeps its previous value
    float out[16];
    for (int i = 0; i < 4; i++) // mTcw = mTcw * mTcw_local (:326); cv::gemm accumulates f32 products in double
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)t->mTcw[4 * i + k] * (double)t->mTcwLocal[4 * k + j];
            out[4 * i + j] = (float)s;
        }
    memcpy(t->mTcw, out, sizeof(out));
    memcpy(T_local, t->mTcwLocal, sizeof(out));
    memcpy(T_global, t->mTcw, sizeof(out));
    if (n_inliers) *n_inliers = found ? ninl : -1;
    return KTLVO_OK;
}
