// Feature gating with order-preserving compaction, and the radius pass of ANMS.
//   getOpticalFlow keep predicate      featureTracking.cpp:223-248
//   circular stereo-temporal check     featureTracking.cpp:486-501
//   adaptiveNonMaximalSuppresion       featureTracking.cpp:258-311
#include "ctx.hpp"
#include "internal.hpp"

#include <algorithm>
#include <cfloat>
#include <cstring>
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

struct GateBounds {
    float x_lo, x_hi, y_lo, y_hi;
    double max_track_dis;
};
// getOpticalFlow's keep predicate (:223-248) for point i of the lists f1 (start), f2 (forward track), f3 (f2 tracked back)
__device__ __forceinline__ bool fb_keep(const GateBounds &b, const float *f1, const float *f2, const float *f3, float y_threshold,
                                        int i) {
    const float x1 = f1[2 * i], y1 = f1[2 * i + 1], x2 = f2[2 * i], y2 = f2[2 * i + 1];
    const float cx = f3[2 * i], cy = f3[2 * i + 1];
    const float s = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2);
    const double trackDis = sqrt((double)s);
    const bool drop = trackDis > b.max_track_dis || x2 < b.x_lo || x2 > b.x_hi || y2 < b.y_lo || y2 > b.y_hi ||
                      x1 - cx > 5.f || y1 - cy > y_threshold;
    return !drop;
}
// the circular check's predicate (:486-501); std::abs() wraps the comparison in the reference, so the test is one-sided
__device__ __forceinline__ bool circular_keep(const float *pcr, const float *clr, int i) {
    return (pcr[2 * i] - clr[2 * i] < 2.f) && (pcr[2 * i + 1] - clr[2 * i + 1] < 2.f);
}
__device__ __forceinline__ bool gate_keep(const GateArgs &a, int i) {
    if (a.mode == 0) {
        const GateBounds b = {a.x_lo, a.x_hi, a.y_lo, a.y_hi, a.max_track_dis};
        return fb_keep(b, a.f1, a.f2, a.f3, a.y_threshold, i);
    }
    return circular_keep(a.f1, a.f2, i);
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

// The three gates of a frame (circular check, then getOpticalFlow's gate with y threshold 1 on the stereo pair
// and with y threshold 5 on the temporal pair, :486-515) in one pass over lists that are all indexed by the
// frame's start points: a point's tracks depend on nothing but its images and its start, so every track a later
// gate needs can be made before the earlier gate has spoken, and the gates only decide which points go on.
// Output is what three gate_compact launches with gathers in between produce: K1 = start indices kept by
// the first gate, K2 = positions in that list kept by the second, K3 = positions in the second list kept by the
// third, and the survivors' start points, temporal tracks and stereo tracks.
struct FrameGateArgs {
    const float *A, *B, *C;     // start points in the last left image, their tracks into last right and current left
    const float *D, *E;         // B tracked into current right, C tracked into current right
    const float *CHK, *CHKB;    // B and C tracked back into the last left image
    int n;
    GateBounds b;
    int *K1, *K2, *K3, *cnt;    // cnt[0..2] = list lengths
    float *U1, *U2, *U3;        // A, C, B of the points that pass all three
};
__global__ __launch_bounds__(1024) void frame_gates(FrameGateArgs a) {
    __shared__ int wave_cnt[3][16];
    __shared__ int base[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3) base[tid] = 0;
    __syncthreads();
    for (int start = 0; start < a.n; start += 1024) {
        const int i = start + tid;
        bool k[3];
        k[0] = i < a.n && circular_keep(a.D, a.E, i);
        k[1] = k[0] && fb_keep(a.b, a.A, a.B, a.CHK, 1.f, i);
        k[2] = k[1] && fb_keep(a.b, a.A, a.C, a.CHKB, 5.f, i);
        int before[3];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            const unsigned long long m = __ballot(k[g]);
            before[g] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[g][wave] = __popcll(m);
        }
        __syncthreads();
        int o[3];
#pragma unroll
        for (int g = 0; g < 3; g++) {
            int off = base[g];
            for (int w = 0; w < wave; w++) off += wave_cnt[g][w];
            o[g] = off + before[g];
        }
        if (k[0]) a.K1[o[0]] = i;
        if (k[1]) a.K2[o[1]] = o[0];
        if (k[2]) {
            a.K3[o[2]] = o[1];
            a.U1[2 * o[2]] = a.A[2 * i]; a.U1[2 * o[2] + 1] = a.A[2 * i + 1];
            a.U2[2 * o[2]] = a.C[2 * i]; a.U2[2 * o[2] + 1] = a.C[2 * i + 1];
            a.U3[2 * o[2]] = a.B[2 * i]; a.U3[2 * o[2] + 1] = a.B[2 * i + 1];
        }
        __syncthreads();
        if (tid < 3) {
            int t = 0;
            for (int w = 0; w < 16; w++) t += wave_cnt[tid][w];
            base[tid] += t;
        }
        __syncthreads();
    }
    if (tid < 3) a.cnt[tid] = base[tid];
}

int launch_frame_gates(ktlvo_ctx *ctx, const float *dA, const float *dB, const float *dC, const float *dD, const float *dE,
                       const float *dCHK, const float *dCHKB, int n, int *dK1, int *dK2, int *dK3, int *d_cnt, float *dU1,
                       float *dU2, float *dU3) {
    FrameGateArgs a;
    a.A = dA; a.B = dB; a.C = dC; a.D = dD; a.E = dE; a.CHK = dCHK; a.CHKB = dCHKB; a.n = n;
    a.b.x_lo = (float)ctx->cfg.x_boundary; a.b.x_hi = (float)(ctx->cfg.width - ctx->cfg.x_boundary);
    a.b.y_lo = (float)ctx->cfg.y_boundary; a.b.y_hi = (float)(ctx->cfg.height - ctx->cfg.y_boundary);
    a.b.max_track_dis = ctx->cfg.max_track_dis;
    a.K1 = dK1; a.K2 = dK2; a.K3 = dK3; a.cnt = d_cnt; a.U1 = dU1; a.U2 = dU2; a.U3 = dU3;
    hipLaunchKernelGGL(frame_gates, dim3(1), dim3(1024), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
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

// Work area and pinned mirror for a host-pointer gate call: `lists` input lists of n points, then 8
// counters (the first is the kept count) and n kept indices.
static int host_gate_io(ktlvo_ctx *ctx, int n, int lists) {
    const size_t bytes = (size_t)lists * n * 8 + (8 + (size_t)n) * sizeof(int);
    int rc = ensure_misc(ctx, bytes);
    if (rc != KTLVO_OK) return rc;
    return ensure_pin(ctx, bytes);
}

extern "C" int ktlvo_fb_gate(ktlvo_ctx *ctx, const float *f1, const float *f2, const float *f1_checked, int n,
                             int y_threshold, int *keep_idx, int *n_keep) {
    if (!ctx || n < 0 || !n_keep) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    *n_keep = 0;
    if (n == 0) return KTLVO_OK;
    if (!f1 || !f2 || !f1_checked || !keep_idx) return KTLVO_ERR_INVALID;
    // three lists up in one copy, count and kept indices down in one copy, both through pinned memory
    const size_t b = (size_t)n * 8;
    int rc = host_gate_io(ctx, n, 3);
    if (rc != KTLVO_OK) return rc;
    char *hp = (char *)ctx->h_pin;
    memcpy(hp, f1, b); memcpy(hp + b, f2, b); memcpy(hp + 2 * b, f1_checked, b);
    float *d_in = (float *)ctx->d_misc;
    int *d_cnt = (int *)(d_in + 6 * (size_t)n), *d_keep = d_cnt + 8;
    KTLVO_HIP(ctx, hipMemcpyAsync(d_in, hp, 3 * b, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_fb_gate(ctx, d_in, d_in + 2 * (size_t)n, d_in + 4 * (size_t)n, n, nullptr, y_threshold, d_keep, d_cnt, Gather());
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(hp, d_cnt, (8 + (size_t)n) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_keep = ((const int *)hp)[0];
    memcpy(keep_idx, hp + 8 * sizeof(int), (size_t)*n_keep * sizeof(int));
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
    int rc = host_gate_io(ctx, n, 2);
    if (rc != KTLVO_OK) return rc;
    char *hp = (char *)ctx->h_pin;
    memcpy(hp, pcr, b); memcpy(hp + b, clr, b);
    float *d_in = (float *)ctx->d_misc;
    int *d_cnt = (int *)(d_in + 4 * (size_t)n), *d_keep = d_cnt + 8;
    KTLVO_HIP(ctx, hipMemcpyAsync(d_in, hp, 2 * b, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_circular_gate(ctx, d_in, d_in + 2 * (size_t)n, n, d_keep, d_cnt, Gather());
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(hp, d_cnt, (8 + (size_t)n) * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_keep = ((const int *)hp)[0];
    memcpy(keep_idx, hp + 8 * sizeof(int), (size_t)*n_keep * sizeof(int));
    return KTLVO_OK;
}

namespace ktlvo {
// getOpticalFlow on device-resident features: d_f1 -> forward LK -> backward LK -> gate.
// n_cap bounds the count; if d_n is given the count is read on the device.  d_f2/d_chk are work
// lists of n_cap points.  g gathers further lists through the kept indices.
int device_get_optical_flow(ktlvo_ctx *ctx, int slot_1, int slot_2, const float *d_f1, int n_cap, const int *d_n,
                            int y_threshold, float *d_f2, float *d_chk, int *d_keep_idx, int *d_n_keep,
                            const Gather &g) {
    ktlvo_lk_params p = {ctx->cfg.win, ctx->cfg.max_level, 30, 0.01, 1e-4, KTLVO_LK_SUMS_DEFAULT};
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
    // outputs back to back in the work area: 8 counters, n kept indices, the two gathered lists;
    // one copy up and one copy down through pinned memory
    const size_t out_bytes = (8 + (size_t)n) * sizeof(int) + (size_t)n * 16;
    int rc = ensure_misc(ctx, out_bytes);
    if (rc != KTLVO_OK) return rc;
    rc = ensure_pin(ctx, out_bytes);
    if (rc != KTLVO_OK) return rc;
    char *hp = (char *)ctx->h_pin;
    int *d_cnt = (int *)ctx->d_misc, *d_keep = d_cnt + 8;
    float *d_s1 = (float *)(d_keep + n), *d_s2 = d_s1 + 2 * (size_t)n;
    memcpy(hp, features_1, (size_t)n * 8);
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], hp, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    Gather g;
    g.src[0] = ctx->d_pts[0]; g.dst[0] = d_s1;
    g.src[1] = ctx->d_pts[2]; g.dst[1] = d_s2;
    rc = device_get_optical_flow(ctx, slot_1, slot_2, ctx->d_pts[0], n, nullptr, y_threshold, ctx->d_pts[2],
                                 ctx->d_pts[3], d_keep, d_cnt, g);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(hp, ctx->d_misc, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int m = *n_keep = ((const int *)hp)[0];
    const char *h_keep = hp + 8 * sizeof(int), *h_s1 = h_keep + (size_t)n * sizeof(int), *h_s2 = h_s1 + (size_t)n * 8;
    memcpy(keep_idx, h_keep, (size_t)m * sizeof(int));
    memcpy(f1_sel, h_s1, (size_t)m * 8);
    memcpy(f2_sel, h_s2, (size_t)m * 8);
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
