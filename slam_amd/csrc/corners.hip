// Corner detection for gfx950: cv::GFTTDetector::detect -> cv::goodFeaturesToTrack as the reference
// runs it per sub-region (featureTracking.cpp:102-103, 408-430), and on whole frames.
//
//   corner_response_tile  cornerEigenValsVecs (Sobel 3x3, 3x3 box) + calcHarris / calcMinEigenVal;
//                         a 34x34 covariance tile is staged in LDS, the response map and the
//                         ROI maximum (minMaxLoc) go to global memory
//   corner_nms_compact    threshold (THRESH_TOZERO at quality*max), 3x3 dilate-equality test,
//                         compaction of (value, raster index) keys
//   corner_select         per ROI: bitonic sort of the keys in LDS (std::sort(greaterThanPtr)),
//                         then the greedy minimum-distance pass over a cell grid in LDS
// ROIs larger than 4096 pixels keep the first two kernels and do the (inherently sequential)
// sort + greedy pass on the host.
#include "ctx.hpp"
#include "internal.hpp"

#include <cstring>

#include <algorithm>
#include <cmath>

namespace ktlvo {

struct RoiDev {
    int x, y, w, h;
    int off; // offset of this ROI's pixels in the response / candidate arrays
};

constexpr int CT = 32;       // response tile
constexpr int CC = CT + 2;   // covariance tile
constexpr int SEL_MAX = 4096;  // largest ROI area / candidate count handled by corner_select
constexpr int CELL_MAX = 1024; // largest min-distance cell grid handled by corner_select

__device__ __forceinline__ int reflect101d(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// monotone float -> uint map so the ROI maximum can be taken with atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __host__ __forceinline__ float ord2f(unsigned u) {
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

__global__ __launch_bounds__(256) void corner_response_tile(const SlotDesc *slots, int slot, const RoiDev *rois,
                                                            float s1, float s2, int harris, double k, float *resp,
                                                            unsigned *roi_max) {
    const RoiDev R = rois[blockIdx.z];
    const int tx0 = blockIdx.x * CT, ty0 = blockIdx.y * CT;
    if (tx0 >= R.w || ty0 >= R.h) return;
    const LevelDesc L = slots[slot].lv[0];
    __shared__ float cxx[CC][CC + 1], cxy[CC][CC + 1], cyy[CC][CC + 1];
    const int tid = threadIdx.x;

    for (int i = tid; i < CC * CC; i += 256) {
        const int cy = i / CC, cx = i - cy * CC;
        // the box filter reflects at the ROI's own edges ...
        const int x = reflect101d(tx0 - 1 + cx, R.w), y = reflect101d(ty0 - 1 + cy, R.h);
        // ... while Sobel reads the parent image (its reflect-101 frame is materialised)
        const uint8_t *p = L.origin + (ptrdiff_t)(R.y + y) * L.pitch + (R.x + x);
        const int P = L.pitch;
        const float a00 = p[-P - 1], a01 = p[-P], a02 = p[-P + 1];
        const float a10 = p[-1], a12 = p[1];
        const float a20 = p[P - 1], a21 = p[P], a22 = p[P + 1];
        const float r0 = a02 - a00, r1 = a12 - a10, r2 = a22 - a20;
        const float dx = (r0 + r2) * s1 + r1 * s2;
        float q0 = s1 * a00; q0 += s2 * a01; q0 += s1 * a02;
        float q2 = s1 * a20; q2 += s2 * a21; q2 += s1 * a22;
        const float dy = q2 - q0;
        cxx[cy][cx] = dx * dx; cxy[cy][cx] = dx * dy; cyy[cy][cx] = dy * dy;
    }
    __syncthreads();

    float vmax = -INFINITY;
    for (int i = tid; i < CT * CT; i += 256) {
        const int py = i / CT, px = i - py * CT;
        const int x = tx0 + px, y = ty0 + py;
        if (x >= R.w || y >= R.h) continue;
        double a = 0, b = 0, c = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double ra = 0, rb = 0, rc = 0;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                ra += (double)cxx[py + j][px + q];
                rb += (double)cxy[py + j][px + q];
                rc += (double)cyy[py + j][px + q];
            }
            a += ra; b += rb; c += rc;
        }
        const float fa = (float)a, fb = (float)b, fc = (float)c;
        float r;
        if (harris) {
            const float t = fa * fc - fb * fb;
            r = (float)((double)t - k * (double)(fa + fc) * (double)(fa + fc));
        } else {
            const float ha = fa * 0.5f, hc = fc * 0.5f;
            r = (ha + hc) - sqrtf((ha - hc) * (ha - hc) + fb * fb);
        }
        resp[R.off + y * R.w + x] = r;
        vmax = fmaxf(vmax, r);
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((tid & 63) == 0 && vmax > -INFINITY) atomicMax(&roi_max[blockIdx.z], f2ord(vmax));
}

// A workgroup covers 64 x 16 pixels of one ROI.  Candidates are counted per wave (ballot), the four
// waves' counts meet in LDS, and the workgroup takes its places in the ROI's list with ONE global
// atomic: returning atomics on a single address serialise in L2 (about 12 ns each), which made a
// textured whole frame (thousands of local maxima, one counter) cost 60 us when every wave bumped it.
// The order of candidates in the list does not matter: the keys are sorted later and are unique.
constexpr int NMS_ROWS = 16;
__global__ __launch_bounds__(256) void corner_nms_compact(const RoiDev *rois, const float *resp,
                                                          const unsigned *roi_max, double quality,
                                                          unsigned long long *cand, int *cand_cnt) {
    const RoiDev R = rois[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const double maxVal = (double)ord2f(roi_max[blockIdx.z]);
    const float thr = (float)(maxVal * quality);
    __shared__ int wave_cnt[4][NMS_ROWS / 4];
    __shared__ int block_base;
    unsigned long long votes[NMS_ROWS / 4];
    float vals[NMS_ROWS / 4];
#pragma unroll
    for (int k = 0; k < NMS_ROWS / 4; k++) { // wave w takes rows w, w + 4, ...
        const int y = blockIdx.y * NMS_ROWS + wave + 4 * k;
        bool keep = x >= 1 && y >= 1 && x < R.w - 1 && y < R.h - 1;
        float v = 0.f;
        if (keep) {
            const float *e = resp + R.off + y * R.w + x;
            v = e[0] > thr ? e[0] : 0.f;
            keep = v != 0.f;
            if (keep) {
                float m = v;
#pragma unroll
                for (int j = -1; j <= 1; j++)
#pragma unroll
                    for (int i = -1; i <= 1; i++) {
                        const float t = e[j * R.w + i];
                        m = fmaxf(m, t > thr ? t : 0.f);
                    }
                keep = v == m;
            }
        }
        vals[k] = v;
        votes[k] = __ballot(keep);
        if (lane == 0) wave_cnt[wave][k] = __popcll(votes[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int k = 0; k < NMS_ROWS / 4; k++) { const int c = wave_cnt[w][k]; wave_cnt[w][k] = total; total += c; }
        block_base = total ? atomicAdd(&cand_cnt[blockIdx.z], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NMS_ROWS / 4; k++) {
        if (!((votes[k] >> lane) & 1ull)) continue;
        const int y = blockIdx.y * NMS_ROWS + wave + 4 * k;
        const int place = block_base + wave_cnt[wave][k] + __popcll(votes[k] & ((1ull << lane) - 1ull));
        cand[R.off + place] = ((unsigned long long)__float_as_uint(vals[k]) << 32) | (unsigned)(y * R.w + x);
    }
}

struct SelArgs {
    const RoiDev *rois;
    const unsigned long long *cand;
    const int *cand_cnt;
    int max_corners, cap;
    double min_dist;
    float *xy, *resp;
    int *counts;
    int *overflow;
};

// Keys sort descending by value, ties by higher raster index first.
__global__ __launch_bounds__(256) void corner_select(SelArgs a) {
    const int roi = blockIdx.x;
    const RoiDev R = a.rois[roi];
    const int n = a.cand_cnt[roi];
    const int tid = threadIdx.x;
    __shared__ unsigned long long keys[SEL_MAX];
    __shared__ unsigned short cell_cnt[CELL_MAX];
    __shared__ unsigned cell_pts[CELL_MAX][4]; // x | y << 16
    if (n > SEL_MAX || R.w * R.h > SEL_MAX) {
        if (tid == 0) { *a.overflow = 1; a.counts[roi] = 0; }
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) keys[i] = i < n ? a.cand[R.off + i] : 0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long x = keys[i], y = keys[p];
                    const bool desc = (i & k) == 0;
                    if (desc ? x < y : x > y) { keys[i] = y; keys[p] = x; }
                }
            }
            __syncthreads();
        }

    float *oxy = a.xy + (size_t)roi * a.cap * 2;
    float *oresp = a.resp + (size_t)roi * a.cap;
    const int limit = a.max_corners > 0 ? (a.max_corners < a.cap ? a.max_corners : a.cap) : a.cap;
    if (a.min_dist < 1) { // no distance test: first `limit` candidates in sorted order
        const int m = n < limit ? n : limit;
        for (int i = tid; i < m; i += 256) {
            const unsigned idx = (unsigned)keys[i];
            oxy[2 * i] = (float)(idx % R.w); oxy[2 * i + 1] = (float)(idx / R.w);
            oresp[i] = __uint_as_float((unsigned)(keys[i] >> 32));
        }
        if (tid == 0) a.counts[roi] = m;
        return;
    }
    const int cell = __double2int_rn(a.min_dist);
    const int gw = (R.w + cell - 1) / cell, gh = (R.h + cell - 1) / cell;
    if (gw * gh > CELL_MAX) { // the host checks this before launching
        if (tid == 0) { *a.overflow = 1; a.counts[roi] = 0; }
        return;
    }
    for (int i = tid; i < gw * gh; i += 256) cell_cnt[i] = 0;
    __syncthreads();
    if (tid >= 64) return; // the greedy pass is sequential over candidates; one wave runs it
    const double md2 = a.min_dist * a.min_dist;
    volatile unsigned short *vcnt = cell_cnt;
    volatile unsigned(*vpts)[4] = cell_pts;
    const int nb = tid >> 2, slot = tid & 3; // lanes 0..35: 9 neighbour cells x 4 slots
    int ncorners = 0;
    for (int i = 0; i < n; i++) {
        const unsigned idx = (unsigned)keys[i];
        const int y = idx / R.w, x = idx - y * R.w;
        const int xc = x / cell, yc = y / cell;
        bool conflict = false;
        if (tid < 36) {
            const int cx = xc + nb % 3 - 1, cy = yc + nb / 3 - 1;
            if (cx >= 0 && cy >= 0 && cx < gw && cy < gh) {
                const int ci = cy * gw + cx;
                if (slot < vcnt[ci]) {
                    const unsigned p = vpts[ci][slot];
                    const float dx = (float)x - (float)(p & 0xffff), dy = (float)y - (float)(p >> 16);
                    conflict = (double)(dx * dx + dy * dy) < md2;
                }
            }
        }
        if (__ballot(conflict) != 0ull) continue;
        const int ci = yc * gw + xc;
        const int cnt = vcnt[ci];
        if (cnt >= 4) { // cannot happen for cell = round(min_dist); reported instead of dropped
            if (tid == 0) *a.overflow = 2;
            break;
        }
        if (tid == 0) {
            vpts[ci][cnt] = (unsigned)x | ((unsigned)y << 16);
            vcnt[ci] = (unsigned short)(cnt + 1);
            oxy[2 * ncorners] = (float)x; oxy[2 * ncorners + 1] = (float)y;
            oresp[ncorners] = __uint_as_float((unsigned)(keys[i] >> 32));
        }
        __builtin_amdgcn_wave_barrier();
        ++ncorners;
        if (ncorners == limit) break;
    }
    if (tid == 0) a.counts[roi] = ncorners;
}

// Host form of the sort + greedy pass for large ROIs (same rule as corner_select).
static int host_select(const RoiDev &R, std::vector<unsigned long long> &keys, const ktlvo_gftt_params &p, int cap,
                       float *xy, float *resp) {
    std::sort(keys.begin(), keys.end(), [](unsigned long long a, unsigned long long b) { return a > b; });
    const int limit = p.max_corners > 0 ? std::min(p.max_corners, cap) : cap;
    int ncorners = 0;
    auto emit = [&](unsigned long long key, int x, int y) {
        xy[2 * ncorners] = (float)x; xy[2 * ncorners + 1] = (float)y;
        unsigned b = (unsigned)(key >> 32);
        float f;
        __builtin_memcpy(&f, &b, 4);
        resp[ncorners] = f;
        ++ncorners;
    };
    if (p.min_dist < 1) {
        for (unsigned long long key : keys) {
            if (ncorners == limit) break;
            const unsigned idx = (unsigned)key;
            emit(key, idx % R.w, idx / R.w);
        }
        return ncorners;
    }
    const int cell = (int)lrint(p.min_dist);
    const int gw = (R.w + cell - 1) / cell, gh = (R.h + cell - 1) / cell;
    std::vector<std::vector<unsigned>> grid((size_t)gw * gh);
    const double md2 = p.min_dist * p.min_dist;
    for (unsigned long long key : keys) {
        const unsigned idx = (unsigned)key;
        const int y = idx / R.w, x = idx % R.w;
        const int xc = x / cell, yc = y / cell;
        bool good = true;
        for (int yy = std::max(0, yc - 1); yy <= std::min(gh - 1, yc + 1) && good; yy++)
            for (int xx = std::max(0, xc - 1); xx <= std::min(gw - 1, xc + 1) && good; xx++)
                for (unsigned q : grid[(size_t)yy * gw + xx]) {
                    const float dx = (float)x - (float)(q & 0xffff), dy = (float)y - (float)(q >> 16);
                    if ((double)(dx * dx + dy * dy) < md2) { good = false; break; }
                }
        if (!good) continue;
        grid[(size_t)yc * gw + xc].push_back((unsigned)x | ((unsigned)y << 16));
        emit(key, x, y);
        if (ncorners == limit) break;
    }
    return ncorners;
}

struct CornerWork {
    RoiDev *d_rois;
    float *d_resp;
    unsigned *d_max;
    int *d_cnt;
    int *d_overflow;
    unsigned long long *d_cand;
    float *d_xy, *d_oresp;
    int *d_counts;
    // the results (xy, responses, counts, overflow flag) are adjacent: out_bytes from d_xy on
    size_t off_oresp, off_counts, off_overflow, out_bytes;
};

static int carve(ktlvo_ctx *ctx, int n_roi, size_t total_px, int cap, CornerWork &w) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o_rois = 0;
    size_t o_resp = o_rois + up(sizeof(RoiDev) * n_roi);
    size_t o_max = o_resp + up(total_px * 4);
    size_t o_cnt = o_max + up(4 * (size_t)n_roi);
    size_t o_cand = o_cnt + up(4 * (size_t)n_roi);
    size_t o_xy = o_cand + up(total_px * 8);
    size_t o_or = o_xy + up((size_t)n_roi * cap * 8);
    size_t o_counts = o_or + up((size_t)n_roi * cap * 4);
    size_t o_ovf = o_counts + up(4 * (size_t)n_roi);
    size_t total = o_ovf + 256;
    int rc = ensure_misc(ctx, total);
    if (rc != KTLVO_OK) return rc;
    char *b = (char *)ctx->d_misc;
    w.d_rois = (RoiDev *)(b + o_rois); w.d_resp = (float *)(b + o_resp); w.d_max = (unsigned *)(b + o_max);
    w.d_cnt = (int *)(b + o_cnt); w.d_overflow = (int *)(b + o_ovf); w.d_cand = (unsigned long long *)(b + o_cand);
    w.d_xy = (float *)(b + o_xy); w.d_oresp = (float *)(b + o_or); w.d_counts = (int *)(b + o_counts);
    w.off_oresp = o_or - o_xy; w.off_counts = o_counts - o_xy; w.off_overflow = o_ovf - o_xy;
    w.out_bytes = o_ovf + 4 - o_xy;
    return KTLVO_OK;
}

static int check_rois(const ktlvo_ctx *ctx, const ktlvo_roi *rois, int n_roi, std::vector<RoiDev> &out, size_t &total,
                      int &maxw, int &maxh) {
    total = 0; maxw = 0; maxh = 0;
    out.resize(n_roi);
    for (int i = 0; i < n_roi; i++) {
        const ktlvo_roi &r = rois[i];
        if (r.w <= 0 || r.h <= 0 || r.x < 0 || r.y < 0 || r.x + r.w > ctx->cfg.width || r.y + r.h > ctx->cfg.height)
            return KTLVO_ERR_INVALID;
        if (r.w > 65535 || r.h > 65535) return KTLVO_ERR_UNSUPPORTED;
        out[i] = {r.x, r.y, r.w, r.h, (int)total};
        total += (size_t)r.w * r.h;
        maxw = std::max(maxw, r.w); maxh = std::max(maxh, r.h);
    }
    if (total > 0x7fffffff) return KTLVO_ERR_CAPACITY;
    return KTLVO_OK;
}

static int run_response(ktlvo_ctx *ctx, int slot, const std::vector<RoiDev> &rois, int maxw, int maxh, int use_harris,
                        double k, const CornerWork &w) {
    const int n_roi = (int)rois.size();
    KTLVO_HIP(ctx, hipMemcpyAsync(w.d_rois, rois.data(), sizeof(RoiDev) * n_roi, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemsetAsync(w.d_max, 0, 4 * (size_t)n_roi, ctx->stream));
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0); // cornerEigenValsVecs: 2^(ksize-1) * blockSize * 255
    const float s1 = (float)(1.0 * scale), s2 = (float)(2.0 * scale);
    dim3 grid((maxw + CT - 1) / CT, (maxh + CT - 1) / CT, n_roi);
    hipLaunchKernelGGL(corner_response_tile, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot, w.d_rois, s1, s2,
                       use_harris, k, w.d_resp, w.d_max);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_corner_response(ktlvo_ctx *ctx, int slot, const ktlvo_roi *roi, int use_harris, double harris_k,
                                     float *out) {
    if (!ctx || !roi || !out || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (ctx->built[slot] < 0) return KTLVO_ERR_STATE;
    std::vector<RoiDev> rois;
    size_t total;
    int maxw, maxh;
    int rc = check_rois(ctx, roi, 1, rois, total, maxw, maxh);
    if (rc != KTLVO_OK) return rc;
    CornerWork w;
    rc = carve(ctx, 1, total, 1, w);
    if (rc != KTLVO_OK) return rc;
    rc = run_response(ctx, slot, rois, maxw, maxh, use_harris, harris_k, w);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(out, w.d_resp, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}

extern "C" int ktlvo_detect_gftt(ktlvo_ctx *ctx, int slot, const ktlvo_roi *rois_in, int n_roi,
                                 const ktlvo_gftt_params *p, float *xy, float *response, int *counts, int cap) {
    if (!ctx || !p || n_roi < 0 || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_roi == 0) return KTLVO_OK;
    if (!rois_in || !xy || !response || !counts || cap <= 0) return KTLVO_ERR_INVALID;
    if (p->block_size != 3) return KTLVO_ERR_UNSUPPORTED;
    if (p->max_corners > cap) return KTLVO_ERR_INVALID;
    if (!(p->quality > 0) || !(p->min_dist >= 0)) return KTLVO_ERR_INVALID; // goodFeaturesToTrack's own assertion
    if (ctx->built[slot] < 0) return KTLVO_ERR_STATE;
    std::vector<RoiDev> rois;
    size_t total;
    int maxw, maxh;
    int rc = check_rois(ctx, rois_in, n_roi, rois, total, maxw, maxh);
    if (rc != KTLVO_OK) return rc;
    CornerWork w;
    rc = carve(ctx, n_roi, total, cap, w);
    if (rc != KTLVO_OK) return rc;
    ProfScope ps(ctx, KTLVO_K_CORNER);
    rc = run_response(ctx, slot, rois, maxw, maxh, p->use_harris, p->harris_k, w);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemsetAsync(w.d_cnt, 0, 4 * (size_t)n_roi, ctx->stream));
    KTLVO_HIP(ctx, hipMemsetAsync(w.d_overflow, 0, 4, ctx->stream));
    dim3 grid((maxw + 63) / 64, (maxh + NMS_ROWS - 1) / NMS_ROWS, n_roi);
    hipLaunchKernelGGL(corner_nms_compact, grid, dim3(256), 0, ctx->stream, w.d_rois, w.d_resp, w.d_max, p->quality,
                       w.d_cand, w.d_cnt);
    KTLVO_HIP(ctx, hipGetLastError());

    bool small = true;
    for (const RoiDev &r : rois) {
        small = small && (r.w * r.h <= SEL_MAX);
        if (p->min_dist >= 1) {
            const int cell = (int)lrint(p->min_dist);
            small = small && (((r.w + cell - 1) / cell) * ((r.h + cell - 1) / cell) <= CELL_MAX);
        }
    }
    if (small) {
        SelArgs a;
        a.rois = w.d_rois; a.cand = w.d_cand; a.cand_cnt = w.d_cnt; a.max_corners = p->max_corners; a.cap = cap;
        a.min_dist = p->min_dist; a.xy = w.d_xy; a.resp = w.d_oresp; a.counts = w.d_counts; a.overflow = w.d_overflow;
        hipLaunchKernelGGL(corner_select, dim3(n_roi), dim3(256), 0, ctx->stream, a);
        KTLVO_HIP(ctx, hipGetLastError());
        // one copy home through pinned memory, then into the caller's three arrays
        rc = ensure_pin(ctx, w.out_bytes);
        if (rc != KTLVO_OK) return rc;
        const char *hp = (const char *)ctx->h_pin;
        KTLVO_HIP(ctx, hipMemcpyAsync(ctx->h_pin, w.d_xy, w.out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(xy, hp, (size_t)n_roi * cap * 8);
        memcpy(response, hp + w.off_oresp, (size_t)n_roi * cap * 4);
        memcpy(counts, hp + w.off_counts, 4 * (size_t)n_roi);
        const int ovf = *(const int *)(hp + w.off_overflow);
        if (ovf) { ctx->last_error = "corner_select: cell capacity exceeded"; return KTLVO_ERR_UNSUPPORTED; }
        return KTLVO_OK;
    }
    // large ROIs: candidates come back, the sequential pass runs on the host
    std::vector<int> cnt(n_roi);
    KTLVO_HIP(ctx, hipMemcpyAsync(cnt.data(), w.d_cnt, 4 * (size_t)n_roi, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < n_roi; r++) {
        std::vector<unsigned long long> keys(cnt[r]);
        if (cnt[r] > 0)
            KTLVO_HIP(ctx, hipMemcpy(keys.data(), w.d_cand + rois[r].off, 8 * (size_t)cnt[r], hipMemcpyDeviceToHost));
        counts[r] = host_select(rois[r], keys, *p, cap, xy + (size_t)r * cap * 2, response + (size_t)r * cap);
    }
    return KTLVO_OK;
}
