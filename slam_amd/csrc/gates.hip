// Feature gating with order-preserving compaction, and the radius pass of ANMS.
//   getOpticalFlow keep predicate      featureTracking.cpp:223-248
//   circular stereo-temporal check     featureTracking.cpp:486-501
//   adaptiveNonMaximalSuppresion       featureTracking.cpp:258-311
#include "ctx.hpp"
#include "internal.hpp"

#include <algorithm>
#include <cfloat>
#include <limits>

namespace ktlvo {

struct GateArgs {
    const float *f1, *f2, *f3; // fb: f1, f2, f1_checked; circular: prevCurRight, curLeftRight, unused
    int n;
    int mode; // 0 = forward-backward gate, 1 = circular check
    float x_lo, x_hi, y_lo, y_hi;
    double max_track_dis;
    float y_threshold;
    const int *d_n; // optional: n is read from device memory (capped by the n above)
    int *keep_idx;  // kept input indices, ascending
    int *n_keep;    // their number
    const float *gsrc[MAX_GATHER]; // optional point lists gathered through the kept indices ...
    float *gdst[MAX_GATHER];       // ... into these
};

__device__ __forceinline__ bool gate_keep(const GateArgs &a, int i) {
    if (a.mode == 0) {
        const float x1 = a.f1[2 * i], y1 = a.f1[2 * i + 1], x2 = a.f2[2 * i], y2 = a.f2[2 * i + 1];
        const float cx = a.f3[2 * i], cy = a.f3[2 * i + 1];
        const float s = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2);
        const double trackDis = sqrt((double)s);
        const bool drop = trackDis > a.max_track_dis || x2 < a.x_lo || x2 > a.x_hi || y2 < a.y_lo || y2 > a.y_hi ||
                          x1 - cx > 5.f || y1 - cy > a.y_threshold;
        return !drop;
    }
    // std::abs() wraps the comparison in the reference, so the test is one-sided
    return (a.f1[2 * i] - a.f2[2 * i] < 2.f) && (a.f1[2 * i + 1] - a.f2[2 * i + 1] < 2.f);
}

// One workgroup walks the list in chunks of 1024 and keeps a running output offset, so the
// kept indices come out in input order (the reference appends with push_back).
__global__ __launch_bounds__(1024) void gate_compact(GateArgs a) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    const int n = a.d_n ? min(*a.d_n, a.n) : a.n;
    for (int start = 0; start < n; start += 1024) {
        const int i = start + tid;
        const bool k = i < n && gate_keep(a, i);
        const unsigned long long m = __ballot(k);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wave_cnt[w];
        if (k) {
            const int o = off + before;
            a.keep_idx[o] = i;
#pragma unroll
            for (int g = 0; g < MAX_GATHER; g++)
                if (a.gdst[g]) { a.gdst[g][2 * o] = a.gsrc[g][2 * i]; a.gdst[g][2 * o + 1] = a.gsrc[g][2 * i + 1]; }
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wave_cnt[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *a.n_keep = base;
}

int launch_fb_gate(ktlvo_ctx *ctx, const float *d_f1, const float *d_f2, const float *d_f1chk, int n, const int *d_n,
                   int y_threshold, int *d_keep_idx, int *d_n_keep, const Gather &g) {
    GateArgs a;
    a.f1 = d_f1; a.f2 = d_f2; a.f3 = d_f1chk; a.n = n; a.mode = 0; a.d_n = d_n;
    a.x_lo = (float)ctx->cfg.x_boundary; a.x_hi = (float)(ctx->cfg.width - ctx->cfg.x_boundary);
    a.y_lo = (float)ctx->cfg.y_boundary; a.y_hi = (float)(ctx->cfg.height - ctx->cfg.y_boundary);
    a.max_track_dis = ctx->cfg.max_track_dis;
    a.y_threshold = (float)y_threshold;
    a.keep_idx = d_keep_idx; a.n_keep = d_n_keep;
    for (int i = 0; i < MAX_GATHER; i++) { a.gsrc[i] = g.src[i]; a.gdst[i] = g.dst[i]; }
    hipLaunchKernelGGL(gate_compact, dim3(1), dim3(1024), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

int launch_circular_gate(ktlvo_ctx *ctx, const float *d_pcr, const float *d_clr, int n, int *d_keep_idx, int *d_n_keep,
                         const Gather &g) {
    GateArgs a = {};
    a.f1 = d_pcr; a.f2 = d_clr; a.f3 = nullptr; a.n = n; a.mode = 1; a.d_n = nullptr;
    a.keep_idx = d_keep_idx; a.n_keep = d_n_keep;
    for (int i = 0; i < MAX_GATHER; i++) { a.gsrc[i] = g.src[i]; a.gdst[i] = g.dst[i]; }
    hipLaunchKernelGGL(gate_compact, dim3(1), dim3(1024), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// radius_i = min over j < i with response_j > 1.11 * response_i of |p_i - p_j|; the list is
// sorted by descending response, so the j loop stops at the first failing j (as the
// reference's loop condition does).
__global__ __launch_bounds__(256) void anms_radius(const float *xy, const float *resp, int n, double *radii) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float response = resp[i] * 1.11f;
    const float xi = xy[2 * i], yi = xy[2 * i + 1];
    double radius = DBL_MAX;
    for (int j = 0; j < i && resp[j] > response; ++j) {
        const float dx = xi - xy[2 * j], dy = yi - xy[2 * j + 1];
        const double d = sqrt((double)dx * dx + (double)dy * dy); // cv::norm(Point2f)
        radius = d < radius ? d : radius;
    }
    radii[i] = radius;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_fb_gate(ktlvo_ctx *ctx, const float *f1, const float *f2, const float *f1_checked, int n,
                             int y_threshold, int *keep_idx, int *n_keep) {
    if (!ctx || n < 0 || !n_keep) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    *n_keep = 0;
    if (n == 0) return KTLVO_OK;
    if (!f1 || !f2 || !f1_checked || !keep_idx) return KTLVO_ERR_INVALID;
    const size_t b = (size_t)n * 8;
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], f1, b, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[1], f2, b, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[2], f1_checked, b, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_fb_gate(ctx, ctx->d_pts[0], ctx->d_pts[1], ctx->d_pts[2], n, nullptr, y_threshold, ctx->d_idx,
                            ctx->d_idx + ctx->cfg.max_points, Gather());
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(n_keep, ctx->d_idx + ctx->cfg.max_points, sizeof(int), hipMemcpyDeviceToHost,
                                  ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*n_keep > 0)
        KTLVO_HIP(ctx, hipMemcpy(keep_idx, ctx->d_idx, (size_t)*n_keep * sizeof(int), hipMemcpyDeviceToHost));
    return KTLVO_OK;
}

extern "C" int ktlvo_circular_gate(ktlvo_ctx *ctx, const float *pcr, const float *clr, int n, int *keep_idx,
                                   int *n_keep) {
    if (!ctx || n < 0 || !n_keep) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    *n_keep = 0;
    if (n == 0) return KTLVO_OK;
    if (!pcr || !clr || !keep_idx) return KTLVO_ERR_INVALID;
    const size_t b = (size_t)n * 8;
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], pcr, b, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[1], clr, b, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_circular_gate(ctx, ctx->d_pts[0], ctx->d_pts[1], n, ctx->d_idx, ctx->d_idx + ctx->cfg.max_points,
                                  Gather());
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(n_keep, ctx->d_idx + ctx->cfg.max_points, sizeof(int), hipMemcpyDeviceToHost,
                                  ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*n_keep > 0)
        KTLVO_HIP(ctx, hipMemcpy(keep_idx, ctx->d_idx, (size_t)*n_keep * sizeof(int), hipMemcpyDeviceToHost));
    return KTLVO_OK;
}

namespace ktlvo {
// getOpticalFlow on device-resident features: d_f1 -> forward LK -> backward LK -> gate.
// n_cap bounds the count; if d_n is given the count is read on the device.  d_f2/d_chk are work
// lists of n_cap points.  g gathers further lists through the kept indices.
int device_get_optical_flow(ktlvo_ctx *ctx, int slot_1, int slot_2, const float *d_f1, int n_cap, const int *d_n,
                            int y_threshold, float *d_f2, float *d_chk, int *d_keep_idx, int *d_n_keep,
                            const Gather &g) {
    ktlvo_lk_params p = {ctx->cfg.win, ctx->cfg.max_level, 30, 0.01, 1e-4};
    if (!slot_ok(ctx, slot_1) || !slot_ok(ctx, slot_2)) return KTLVO_ERR_INVALID;
    PairList fw, bw;
    fw.prev[0] = (uint16_t)slot_1; fw.next[0] = (uint16_t)slot_2;
    bw.prev[0] = (uint16_t)slot_2; bw.next[0] = (uint16_t)slot_1;
    const int zero = 0;
    int rc = launch_pyrlk_ex(ctx, fw, 1, d_f1, &zero, d_f2, &zero, ctx->d_status, ctx->d_err, n_cap, d_n, &p); // :205
    if (rc != KTLVO_OK) return rc;
    rc = launch_pyrlk_ex(ctx, bw, 1, d_f2, &zero, d_chk, &zero, ctx->d_status, ctx->d_err, n_cap, d_n, &p);     // :214
    if (rc != KTLVO_OK) return rc;
    return launch_fb_gate(ctx, d_f1, d_f2, d_chk, n_cap, d_n, y_threshold, d_keep_idx, d_n_keep, g);
}
} // namespace ktlvo

extern "C" int ktlvo_get_optical_flow(ktlvo_ctx *ctx, int slot_1, int slot_2, const float *features_1, int n,
                                      int y_threshold, float *f1_sel, float *f2_sel, int *keep_idx, int *n_keep) {
    if (!ctx || n < 0 || !n_keep) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    *n_keep = 0;
    if (n == 0) return KTLVO_OK;
    if (!features_1 || !f1_sel || !f2_sel || !keep_idx) return KTLVO_ERR_INVALID;
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], features_1, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = ensure_misc(ctx, (size_t)n * 8 * 2);
    if (rc != KTLVO_OK) return rc;
    float *d_s1 = (float *)ctx->d_misc, *d_s2 = d_s1 + 2 * (size_t)n;
    int *d_cnt = ctx->d_idx + ctx->cfg.max_points;
    Gather g;
    g.src[0] = ctx->d_pts[0]; g.dst[0] = d_s1;
    g.src[1] = ctx->d_pts[2]; g.dst[1] = d_s2;
    rc = device_get_optical_flow(ctx, slot_1, slot_2, ctx->d_pts[0], n, nullptr, y_threshold, ctx->d_pts[2],
                                 ctx->d_pts[3], ctx->d_idx, d_cnt, g);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(n_keep, d_cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int m = *n_keep;
    if (m > 0) {
        KTLVO_HIP(ctx, hipMemcpy(f1_sel, d_s1, (size_t)m * 8, hipMemcpyDeviceToHost));
        KTLVO_HIP(ctx, hipMemcpy(f2_sel, d_s2, (size_t)m * 8, hipMemcpyDeviceToHost));
        KTLVO_HIP(ctx, hipMemcpy(keep_idx, ctx->d_idx, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
    }
    return KTLVO_OK;
}

extern "C" int ktlvo_anms(ktlvo_ctx *ctx, float *xy, float *response, int *n_inout, int num_to_keep) {
    if (!ctx || !n_inout || *n_inout < 0 || num_to_keep < 0) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    const int n = *n_inout;
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    if (n < num_to_keep || n == 0) return KTLVO_OK; // featureTracking.cpp:261
    if (!xy || !response) return KTLVO_ERR_INVALID;
    struct KP { float x, y, response; };
    std::vector<KP> kp(n);
    for (int i = 0; i < n; i++) kp[i] = {xy[2 * i], xy[2 * i + 1], response[i]};
    // The reference sorts with exactly this comparator (:266-270).  With OpenCV 3 every response is
    // 0 and the call's only effect is libstdc++'s permutation of equal keys, which later code
    // depends on (feature order feeds the index bookkeeping and the RANSAC sampler), so the very
    // same std::sort call is made here.
    std::sort(kp.begin(), kp.end(), [](const KP &l, const KP &r) { return l.response > r.response; });
    std::vector<float> sxy(2 * (size_t)n), sresp(n);
    for (int i = 0; i < n; i++) { sxy[2 * i] = kp[i].x; sxy[2 * i + 1] = kp[i].y; sresp[i] = kp[i].response; }
    std::vector<double> radii(n);
    int rc = ensure_misc(ctx, (size_t)n * sizeof(double));
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], sxy.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_err, sresp.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(anms_radius, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_pts[0], ctx->d_err, n,
                       (double *)ctx->d_misc);
    KTLVO_HIP(ctx, hipGetLastError());
    KTLVO_HIP(ctx, hipMemcpyAsync(radii.data(), ctx->d_misc, (size_t)n * sizeof(double), hipMemcpyDeviceToHost,
                                  ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<double> radiiSorted(radii);
    std::sort(radiiSorted.begin(), radiiSorted.end(), [](const double &l, const double &r) { return l > r; });
    // :299 indexes radiiSorted[numToKeep], one past the end when n == numToKeep; keep everything there
    const double decisionRadius = num_to_keep < n ? radiiSorted[num_to_keep] : -1.0;
    int m = 0;
    for (int i = 0; i < n; i++)
        if (radii[i] >= decisionRadius) {
            xy[2 * m] = sxy[2 * i]; xy[2 * m + 1] = sxy[2 * i + 1]; response[m] = sresp[i];
            ++m;
        }
    *n_inout = m;
    return KTLVO_OK;
}
