// RANSAC PnP for gfx950: cv::solvePnPRansac as called by estimatePoseRANSAC
// (visualOdometry.cpp:177-217, parameters :122-127).
//
// OpenCV's loop is sequential: draw 4 points, P3P, project all N points, count inliers, maybe
// shrink the iteration budget.  Its RNG stream does not depend on the results, so all index
// quadruples are drawn up front, every hypothesis is scored in ONE launch (one workgroup per
// hypothesis), and the sequential acceptance rule is then replayed over the counts; the model it
// selects is the one the sequential loop would have selected.
#include "ctx.hpp"
#include "internal.hpp"
#include "p3p.hpp"

#include <cfloat>
#include <cmath>
#include <cstring>

namespace ktlvo {

void rodrigues_v2m(const double r[3], double R[9]) { p3p::rodrigues_v2m(r, R); }
void rodrigues_m2v(const double R[9], double r[3]) { p3p::rodrigues_m2v(R, r); }

// PnPRansacCallback::computeError (cv::projectPoints in double, float projections, float
// squared error) + RANSACPointSetRegistrator::findInliers for hypothesis blockIdx.x.
__global__ __launch_bounds__(256) void ransac_score(const float *X3, const float *x2, int n, const double *Rt,
                                                    const unsigned char *valid, double fx, double fy, double cx,
                                                    double cy, float thr_sq, int *counts, int mask_hyp,
                                                    unsigned char *mask) {
    const int h = blockIdx.x;
    __shared__ int wave_cnt[4];
    const int tid = threadIdx.x;
    if (valid && !valid[h]) {
        if (tid == 0) counts[h] = 0;
        return;
    }
    const double *R = Rt + 12 * (size_t)h;
    const double r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4], r5 = R[5], r6 = R[6], r7 = R[7], r8 = R[8];
    const double t0 = R[9], t1 = R[10], t2 = R[11];
    const bool want_mask = mask && h == mask_hyp;
    int cnt = 0;
    for (int i = tid; i < n; i += 256) {
        const double X = X3[3 * i], Y = X3[3 * i + 1], Z = X3[3 * i + 2];
        double x = r0 * X + r1 * Y + r2 * Z + t0;
        double y = r3 * X + r4 * Y + r5 * Z + t1;
        double z = r6 * X + r7 * Y + r8 * Z + t2;
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float mx = (float)(x * fx + cx), my = (float)(y * fy + cy);
        const float dx = x2[2 * i] - mx, dy = x2[2 * i + 1] - my;
        const float e = dx * dx + dy * dy;
        const int f = e <= thr_sq;
        if (want_mask) mask[i] = (unsigned char)f;
        cnt += f;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) wave_cnt[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) counts[h] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// Minimal solver on the device: thread h takes index quadruple h, solves P3P in double
// (p3p.hpp, the same code the host path runs; Gao's closed form, `polish` Newton steps on its roots), converts R -> rvec -> R as PnPRansacCallback's model
// round trip does, and leaves R|t, rvec and a validity flag for the scoring launch.
__global__ __launch_bounds__(64) void ransac_p3p(const float *X3, const float *x2, const int *idx, int H, double fx,
                                                 double fy, double cx, double cy, int polish, double *Rt, double *rv,
                                                 unsigned char *valid) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= H) return;
    const double K[4] = {fx, fy, cx, cy};
    double Xs[12], us[8];
    for (int i = 0; i < 4; i++) {
        const int k = idx[4 * h + i];
        Xs[3 * i] = X3[3 * k]; Xs[3 * i + 1] = X3[3 * k + 1]; Xs[3 * i + 2] = X3[3 * k + 2];
        us[2 * i] = x2[2 * k]; us[2 * i + 1] = x2[2 * k + 1];
    }
    p3p::Pose P;
    const bool ok = p3p::solve4(Xs, us, K, polish, P);
    valid[h] = ok ? 1 : 0;
    double r[3] = {0, 0, 0}, R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (ok) {
        p3p::rodrigues_m2v(P.R, r);
        p3p::rodrigues_v2m(r, R);
    }
    for (int i = 0; i < 9; i++) Rt[12 * (size_t)h + i] = R[i];
    for (int i = 0; i < 3; i++) { Rt[12 * (size_t)h + 9 + i] = ok ? P.t[i] : 0.0; rv[3 * (size_t)h + i] = r[i]; }
}

struct CvRng { // cv::RNG (multiply-with-carry)
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

static int update_num_iters(double p, double ep, int modelPoints, int maxIters) { // cv::RANSACUpdateNumIters
    p = p < 0. ? 0. : p > 1. ? 1. : p;
    ep = ep < 0. ? 0. : ep > 1. ? 1. : ep;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

struct ScoreBuffers {
    float *d_X3, *d_x2;
    double *d_Rt, *d_rv;
    unsigned char *d_valid, *d_mask;
    int *d_counts, *d_idx;
    // X3, x2, Rt and valid are adjacent at the head of the work area (offsets o_x, o_Rt, o_valid,
    // up_bytes in all), so a pinned host image of that head goes up in one copy
    size_t o_x, o_Rt, o_valid, up_bytes;
};

static int carve_score(ktlvo_ctx *ctx, int n, int H, ScoreBuffers &b) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_X = 0, o_x = o_X + up((size_t)n * 12), o_Rt = o_x + up((size_t)n * 8);
    const size_t o_valid = o_Rt + up((size_t)H * 96), o_mask = o_valid + up(H), o_cnt = o_mask + up(n);
    const size_t o_rv = o_cnt + up((size_t)H * 4), o_idx = o_rv + up((size_t)H * 24);
    const size_t total = o_idx + up((size_t)H * 16);
    int rc = ensure_misc(ctx, total);
    if (rc != KTLVO_OK) return rc;
    char *p = (char *)ctx->d_misc;
    b.d_X3 = (float *)(p + o_X); b.d_x2 = (float *)(p + o_x); b.d_Rt = (double *)(p + o_Rt);
    b.d_valid = (unsigned char *)(p + o_valid); b.d_mask = (unsigned char *)(p + o_mask); b.d_counts = (int *)(p + o_cnt);
    b.d_rv = (double *)(p + o_rv); b.d_idx = (int *)(p + o_idx);
    b.o_x = o_x; b.o_Rt = o_Rt; b.o_valid = o_valid; b.up_bytes = o_valid + up(H);
    return KTLVO_OK;
}

static int launch_score(ktlvo_ctx *ctx, const ScoreBuffers &b, int n, int H, const unsigned char *d_valid,
                        const double K[4], float thr_sq, int mask_hyp, bool want_mask) {
    ProfScope ps(ctx, KTLVO_K_RANSAC);
    hipLaunchKernelGGL(ransac_score, dim3(H), dim3(256), 0, ctx->stream, b.d_X3, b.d_x2, n, b.d_Rt, d_valid, K[0], K[1],
                       K[2], K[3], thr_sq, b.d_counts, mask_hyp, want_mask ? b.d_mask : nullptr);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_ransac_score(ktlvo_ctx *ctx, const float *X3, const float *x2, int n, const double *Rt, int H,
                                  const double K[4], float thr_sq, int *counts, int mask_hyp, uint8_t *best_mask) {
    if (!ctx || n < 0 || H < 0 || !K) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (H > ctx->cfg.max_hypotheses || n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    if (H == 0) return KTLVO_OK;
    if (!counts || !Rt || (n > 0 && (!X3 || !x2))) return KTLVO_ERR_INVALID;
    if (best_mask && (mask_hyp < 0 || mask_hyp >= H)) return KTLVO_ERR_INVALID;
    if (n == 0) {
        memset(counts, 0, sizeof(int) * H);
        return KTLVO_OK;
    }
    ScoreBuffers b;
    int rc = carve_score(ctx, n, H, b);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(b.d_X3, X3, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(b.d_x2, x2, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(b.d_Rt, Rt, (size_t)H * 96, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_score(ctx, b, n, H, nullptr, K, thr_sq, mask_hyp, best_mask != nullptr);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(counts, b.d_counts, (size_t)H * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (best_mask) KTLVO_HIP(ctx, hipMemcpyAsync(best_mask, b.d_mask, n, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}

extern "C" int ktlvo_ransac_score_dev(ktlvo_ctx *ctx, const float *d_X3, const float *d_x2, int n, const double *d_Rt, int H,
                                      const double K[4], float thr_sq, int *d_counts, int mask_hyp, uint8_t *d_mask) {
    if (!ctx || n < 0 || H < 0 || !K) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (H == 0) return KTLVO_OK;
    if (!d_counts || !d_Rt || (n > 0 && (!d_X3 || !d_x2))) return KTLVO_ERR_INVALID;
    if (d_mask && (mask_hyp < 0 || mask_hyp >= H)) return KTLVO_ERR_INVALID;
    if (n == 0) {
        KTLVO_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int) * (size_t)H, ctx->stream));
        return KTLVO_OK;
    }
    ProfScope ps(ctx, KTLVO_K_RANSAC);
    hipLaunchKernelGGL(ransac_score, dim3(H), dim3(256), 0, ctx->stream, d_X3, d_x2, n, d_Rt, (const unsigned char *)nullptr,
                       K[0], K[1], K[2], K[3], thr_sq, d_counts, mask_hyp, d_mask);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

extern "C" int ktlvo_solve_pnp_ransac(ktlvo_ctx *ctx, const float *X3, const float *x2, int n, int max_iters,
                                      float reproj_err, double confidence, double rvec[3], double tvec[3],
                                      uint8_t *inlier_mask, int *n_inliers, int *found) {
    if (!ctx || n < 0 || !found || !rvec || !tvec) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    *found = 0;
    if (n_inliers) *n_inliers = 0;
    const int modelPoints = 4;
    if (n < modelPoints) return KTLVO_OK; // RANSACPointSetRegistrator::run returns false
    if (!X3 || !x2) return KTLVO_ERR_INVALID;
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    const int H = max_iters > 1 ? max_iters : 1;
    if (H > ctx->cfg.max_hypotheses) return KTLVO_ERR_CAPACITY;
    const double K[4] = {ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy};
    const double threshold = reproj_err;
    const float thr_sq = (float)(threshold * threshold);
    const int polish = ctx->cfg.p3p_polish < 0 ? 0 : ctx->cfg.p3p_polish > 8 ? 8 : ctx->cfg.p3p_polish;

    // 1. all index quadruples, in the order cv::RNG((uint64)-1) hands them to getSubset
    std::vector<int> idx((size_t)4 * H);
    CvRng rng((uint64_t)-1);
    for (int it = 0; it < H; it++) {
        int *q = &idx[4 * (size_t)it];
        if (n == modelPoints) { q[0] = 0; q[1] = 1; q[2] = 2; q[3] = 3; continue; }
        for (int i = 0; i < modelPoints; i++) {
            int j;
            for (;;) {
                q[i] = rng.uniform(0, n);
                for (j = 0; j < i; j++) if (q[i] == q[j]) break;
                if (j == i) break;
            }
        }
    }
    // 2. minimal solver per hypothesis; the model is kept as (rvec, tvec) and R is rebuilt from
    //    rvec for scoring, as PnPRansacCallback does.  3. one launch scores every hypothesis.
    std::vector<double> rv((size_t)3 * H, 0.0), Rt_dev;
    std::vector<unsigned char> valid_dev;
    ScoreBuffers b;
    int rc = carve_score(ctx, n, H, b);
    if (rc != KTLVO_OK) return rc;
    rc = ensure_pin(ctx, b.up_bytes + (size_t)H * 4);
    if (rc != KTLVO_OK) return rc;
    char *hp = (char *)ctx->h_pin;
    const int *counts = (const int *)(hp + b.up_bytes);
    const double *Rt;
    const unsigned char *valid;
    memcpy(hp, X3, (size_t)n * 12);
    memcpy(hp + b.o_x, x2, (size_t)n * 8);
    if (ctx->cfg.device_p3p) {
        KTLVO_HIP(ctx, hipMemcpyAsync(b.d_X3, hp, b.o_x + (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        KTLVO_HIP(ctx, hipMemcpyAsync(b.d_idx, idx.data(), (size_t)H * 16, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(ransac_p3p, dim3((H + 63) / 64), dim3(64), 0, ctx->stream, b.d_X3, b.d_x2, b.d_idx, H, K[0],
                           K[1], K[2], K[3], polish, b.d_Rt, b.d_rv, b.d_valid);
        KTLVO_HIP(ctx, hipGetLastError());
    } else {
        double *Rt_h = (double *)(hp + b.o_Rt);
        unsigned char *valid_h = (unsigned char *)(hp + b.o_valid);
        memset(Rt_h, 0, (size_t)H * 96);
        memset(valid_h, 0, H);
        for (int it = 0; it < H; it++) {
            double Xs[12], us[8];
            for (int i = 0; i < 4; i++) {
                const int k = idx[4 * (size_t)it + i];
                Xs[3 * i] = X3[3 * k]; Xs[3 * i + 1] = X3[3 * k + 1]; Xs[3 * i + 2] = X3[3 * k + 2];
                us[2 * i] = x2[2 * k]; us[2 * i + 1] = x2[2 * k + 1];
            }
            p3p::Pose P;
            if (!p3p::solve4(Xs, us, K, polish, P)) continue;
            valid_h[it] = 1;
            p3p::rodrigues_m2v(P.R, &rv[3 * (size_t)it]);
            p3p::rodrigues_v2m(&rv[3 * (size_t)it], &Rt_h[12 * (size_t)it]);
            memcpy(&Rt_h[12 * (size_t)it + 9], P.t, 3 * sizeof(double));
        }
        // points, poses and flags in one copy from the pinned image of the work area's head
        KTLVO_HIP(ctx, hipMemcpyAsync(b.d_X3, hp, b.o_valid + (size_t)H, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = launch_score(ctx, b, n, H, b.d_valid, K, thr_sq, -1, false);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(hp + b.up_bytes, b.d_counts, (size_t)H * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->cfg.device_p3p) {
        Rt_dev.resize((size_t)12 * H);
        valid_dev.resize(H);
        KTLVO_HIP(ctx, hipMemcpyAsync(valid_dev.data(), b.d_valid, H, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipMemcpyAsync(Rt_dev.data(), b.d_Rt, (size_t)H * 96, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipMemcpyAsync(rv.data(), b.d_rv, (size_t)H * 24, hipMemcpyDeviceToHost, ctx->stream));
        Rt = Rt_dev.data();
        valid = valid_dev.data();
    } else {
        Rt = (const double *)(hp + b.o_Rt);
        valid = (const unsigned char *)(hp + b.o_valid);
    }
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // 4. replay of the sequential acceptance rule
    int niters = H, best = -1, maxGood = 0;
    for (int it = 0; it < niters; it++) {
        if (!valid[it]) continue;
        const int good = counts[it];
        if (good > (maxGood > modelPoints - 1 ? maxGood : modelPoints - 1)) {
            best = it;
            maxGood = good;
            niters = update_num_iters(confidence, (double)(n - good) / n, modelPoints, niters);
        }
    }
    if (best < 0 || maxGood <= 0) return KTLVO_OK;
    if (inlier_mask) {
        KTLVO_HIP(ctx, hipMemcpyAsync(b.d_Rt, &Rt[12 * (size_t)best], 96, hipMemcpyHostToDevice, ctx->stream));
        rc = launch_score(ctx, b, n, 1, nullptr, K, thr_sq, 0, true);
        if (rc != KTLVO_OK) return rc;
        KTLVO_HIP(ctx, hipMemcpyAsync(inlier_mask, b.d_mask, n, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    memcpy(rvec, &rv[3 * (size_t)best], 3 * sizeof(double));
    memcpy(tvec, &Rt[12 * (size_t)best + 9], 3 * sizeof(double));
    if (n_inliers) *n_inliers = maxGood;
    *found = 1;
    return KTLVO_OK;
}
