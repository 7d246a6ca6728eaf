// Pyramidal Lucas-Kanade for gfx950: the LKTrackerInvoker of cv::calcOpticalFlowPyrLK as the
// reference calls it (featureTracking.cpp:205-209: Size(21,21), maxLevel 4,
// TermCriteria(COUNT|EPS, 30, 0.01), flags 0, minEigThreshold 1e-4).
//
// One 64-lane wavefront tracks one feature through all pyramid levels.  The 21x21 window is cut
// into 63 horizontal runs of 7 pixels (21 rows x 3 runs); lane l owns run (row l % 21,
// columns 7*(l / 21) ..+6), lane 63 idles.  A lane keeps its seven template values I, Ix, Iy in
// registers for the whole level, so the template never touches LDS; per Gauss-Newton step a
// lane fetches two 8-byte runs of the next image (rows r and r+1) and forms its seven bilinear
// samples from them with v_perm_b32 + v_dot2 (two taps per instruction).  The 2x2 tensor and the
// mismatch vector are summed across the wave with DPP row operations.  Derivatives come from the
// Scharr planes built with the pyramid (int16 dx|dy per pixel, zero outside the level).
//
// The kernel is VALU-bound (rocprof: VALU busy > 95 %), so the design minimises instructions per
// pixel and per iteration rather than bytes: images and planes stay L2-resident.
//
// Arithmetic follows OpenCV's fixed-point scheme exactly (14-bit bilinear weights, CV_DESCALE,
// int16 patch values, 2^-20 scale); only the accumulation order of the float sums differs.
#include "ctx.hpp"

namespace ktlvo {

struct LkArgs {
    const SlotDesc *slots;
    const float *prev;
    float *next;
    uint8_t *status;
    float *err;
    int n_per_pair;
    int max_level;
    int max_count;
    double eps_sq;
    double min_eig_thr;
    const int *d_n; // optional: actual point count lives in device memory (<= n_per_pair)
    PairList pairs;
    int prev_off[MAX_BATCH]; // first point of pair i in prev[]
    int next_off[MAX_BATCH]; // first point of pair i in next[], status[], err[]
};

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) uint32_t *gptr32; // planes live in global memory

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // v + (v moved by the DPP pattern); lanes outside ROW_MASK add 0
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes, returned wave-uniform.  Fixed tree, so the result is deterministic.
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v); // row_half_mirror
    v = dpp_add<0x140, 0xF>(v); // row_mirror: every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xA>(v); // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v); // row_bcast31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// a * b + c on 24-bit signed factors (every factor here is below 2^15 in magnitude); written as
// asm because clang has no builtin for it and falls back to the quarter-rate 32-bit multiply.
__device__ __forceinline__ int mad24(int a, int b, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

// Bilinear weights as OpenCV forms them: cvRound of float products scaled by 2^14, the fourth
// as the remainder.  w01 = top pair (w00 | w01 << 16), w23 = bottom pair (w10 | w11 << 16).
struct Weights {
    uint32_t w01, w23;
};
__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    const int w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    const int w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    const int w10 = __float2int_rn((1.f - a) * b * 16384.f);
    const int w11 = 16384 - w00 - w01 - w10;
    Weights w;
    w.w01 = (uint32_t)w00 | ((uint32_t)w01 << 16);
    w.w23 = (uint32_t)w10 | ((uint32_t)w11 << 16); // w10 >= 0, so no sign bits leak into the high half
    return w;
}

// Eight consecutive bytes starting at an arbitrary byte address, as two dwords.
struct Run8 {
    uint32_t lo, hi;
};
__device__ __forceinline__ Run8 load_run8(const uint8_t *p) {
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    gptr32 q = (gptr32)(p - sh);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    Run8 r;
    r.lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return r;
}
// bytes i and i+1 of a run, zero-extended into a u16 pair
template <int I>
__device__ __forceinline__ uint32_t pix_pair(const Run8 &r) {
    constexpr uint32_t sel = 0x0c000c00u | (uint32_t)I | ((uint32_t)(I + 1) << 16);
    return __builtin_amdgcn_perm(r.hi, r.lo, sel);
}
// the seven fixed-point bilinear samples (scaled by 32) of a lane's run
__device__ __forceinline__ void sample_run(const Run8 &t, const Run8 &b, const Weights &w, int (&v)[7]) {
    // signed dot: the remainder weight w11 can be -1 when the three rounded weights sum to 16385
#define KTLVO_SAMPLE(I) v[I] = sdot2(pix_pair<I>(b), w.w23, sdot2(pix_pair<I>(t), w.w01, 256)) >> 9
    KTLVO_SAMPLE(0); KTLVO_SAMPLE(1); KTLVO_SAMPLE(2); KTLVO_SAMPLE(3); KTLVO_SAMPLE(4); KTLVO_SAMPLE(5); KTLVO_SAMPLE(6);
#undef KTLVO_SAMPLE
}

constexpr int WIN = 21;
constexpr int RUN = 7;

__global__ __launch_bounds__(256) void pyrlk_track_w21(LkArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int f = blockIdx.x * 4 + wave;
    const int n = a.d_n ? min(*a.d_n, a.n_per_pair) : a.n_per_pair;
    if (f >= n) return;
    const int pair = blockIdx.y;
    const SlotDesc &SP = a.slots[a.pairs.prev[pair]];
    const SlotDesc &SN = a.slots[a.pairs.next[pair]];
    const size_t gp = (size_t)a.prev_off[pair] + f;
    const size_t gi = (size_t)a.next_off[pair] + f;
    const float ppx = a.prev[2 * gp], ppy = a.prev[2 * gp + 1];

    const bool active = lane < 63;
    const int c = active ? lane / 21 : 0;
    const int r = active ? lane - 21 * c : 0;
    const int cx = c * RUN;
    const float halfWin = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    float ox = 0.f, oy = 0.f; // nextPts[i]
    int status = 1;
    float errv = 0.f;

    for (int level = a.max_level; level >= 0; --level) {
        const LevelDesc I = SP.lv[level];
        const LevelDesc J = SN.lv[level];
        const float sc = __int_as_float((127 - level) << 23); // (float)(1./(1 << level))
        float px = ppx * sc, py = ppy * sc;
        if (level == a.max_level) { ox = px; oy = py; }
        else { ox = ox * 2.f; oy = oy * 2.f; }

        px -= halfWin; py -= halfWin;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
            if (level == 0) { status = 0; errv = 0.f; }
            continue;
        }
        Weights w = bilinear_weights(px - ipx, py - ipy);

        // ---- template: image rows ipy+r, +1 and Scharr rows ipy+r, +1, columns ipx+cx .. +7 -----
        int Iv[RUN], ixv[RUN], iyv[RUN];
        {
            const ptrdiff_t e0 = (ptrdiff_t)(ipy + r) * I.pitch + (ipx + cx);
            const Run8 t = load_run8(I.origin + e0), b = load_run8(I.origin + e0 + I.pitch);
            gptr32 dt = (gptr32)(I.dorigin + e0), db = (gptr32)(I.dorigin + e0 + I.pitch);
            uint32_t d0[RUN + 1], d1[RUN + 1];
#pragma unroll
            for (int k = 0; k <= RUN; k++) { d0[k] = dt[k]; d1[k] = db[k]; }
            sample_run(t, b, w, Iv);
#pragma unroll
            for (int i = 0; i < RUN; i++) {
                // (dx_i, dx_i+1) and (dy_i, dy_i+1) as int16 pairs
                const uint32_t xt = __builtin_amdgcn_perm(d0[i + 1], d0[i], 0x05040100u);
                const uint32_t yt = __builtin_amdgcn_perm(d0[i + 1], d0[i], 0x07060302u);
                const uint32_t xb = __builtin_amdgcn_perm(d1[i + 1], d1[i], 0x05040100u);
                const uint32_t yb = __builtin_amdgcn_perm(d1[i + 1], d1[i], 0x07060302u);
                const int ix = sdot2(xb, w.w23, sdot2(xt, w.w01, 1 << 13)) >> 14;
                const int iy = sdot2(yb, w.w23, sdot2(yt, w.w01, 1 << 13)) >> 14;
                ixv[i] = active ? ix : 0;
                iyv[i] = active ? iy : 0;
            }
        }
        int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
        for (int i = 0; i < RUN; i++) {
            s11 = mad24(ixv[i], ixv[i], s11);
            s12 = mad24(ixv[i], iyv[i], s12);
            s22 = mad24(iyv[i], iyv[i], s22);
        }
        const float A11 = wave_sum((float)s11) * FLT_SCALE;
        const float A12 = wave_sum((float)s12) * FLT_SCALE;
        const float A22 = wave_sum((float)s22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig =
            (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * WIN * WIN);
        if ((double)minEig < a.min_eig_thr || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;

        // ---- Gauss-Newton iterations -----------------------------------------------------------
        const uint8_t *jrow = J.origin + (ptrdiff_t)r * J.pitch + cx; // lane's run at window origin (0,0)
        float nx = ox - halfWin, ny = oy - halfWin;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            w = bilinear_weights(nx - inx, ny - iny);
            const uint8_t *jb = jrow + ((ptrdiff_t)iny * J.pitch + inx);
            const Run8 t = load_run8(jb), b = load_run8(jb + J.pitch);
            int v[RUN];
            sample_run(t, b, w, v);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int i = 0; i < RUN; i++) {
                const int diff = v[i] - Iv[i];
                b1 = mad24(diff, ixv[i], b1);
                b2 = mad24(diff, iyv[i], b2);
            }
            const float fb1 = wave_sum((float)b1) * FLT_SCALE;
            const float fb2 = wave_sum((float)b2) * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D;
            const float dy = (A12 * fb1 - A11 * fb2) * D;
            nx += dx; ny += dy;
            ox = nx + halfWin; oy = ny + halfWin;
            if ((double)dx * dx + (double)dy * dy <= a.eps_sq) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                ox -= dx * 0.5f; oy -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }

        // ---- final error (level 0 only) ------------------------------------------------------
        if (level == 0 && status) {
            const float fx = ox - halfWin, fy = oy - halfWin;
            const int inx = (int)floorf(fx), iny = (int)floorf(fy);
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                status = 0;
            } else {
                w = bilinear_weights(fx - inx, fy - iny);
                const uint8_t *jb = jrow + ((ptrdiff_t)iny * J.pitch + inx);
                const Run8 t = load_run8(jb), b = load_run8(jb + J.pitch);
                int v[RUN];
                sample_run(t, b, w, v);
                int e = 0;
#pragma unroll
                for (int i = 0; i < RUN; i++) {
                    const int diff = v[i] - Iv[i];
                    e += diff < 0 ? -diff : diff;
                }
                if (!active) e = 0;
                errv = wave_sum((float)e) * 1.f / (float)(32 * WIN * WIN);
            }
        }
    }
    if (lane == 0) {
        a.next[2 * gi] = ox;
        a.next[2 * gi + 1] = oy;
        a.status[gi] = (uint8_t)status;
        a.err[gi] = errv;
    }
}

int launch_pyrlk(ktlvo_ctx *ctx, const PairList &pairs, int n_pairs, const float *d_prev, int n_per_pair,
                 float *d_next, uint8_t *d_status, float *d_err, const ktlvo_lk_params *p) {
    int po[MAX_BATCH], no[MAX_BATCH];
    for (int i = 0; i < n_pairs && i < MAX_BATCH; i++) po[i] = no[i] = i * n_per_pair;
    return launch_pyrlk_ex(ctx, pairs, n_pairs, d_prev, po, d_next, no, d_status, d_err, n_per_pair, nullptr, p);
}

int launch_pyrlk_ex(ktlvo_ctx *ctx, const PairList &pairs, int n_pairs, const float *d_prev, const int *prev_off,
                    float *d_next, const int *next_off, uint8_t *d_status, float *d_err, int n_per_pair,
                    const int *d_n, const ktlvo_lk_params *p) {
    if (!p) return KTLVO_ERR_INVALID;
    if (p->win != WIN) return KTLVO_ERR_UNSUPPORTED;
    if (n_pairs <= 0 || n_pairs > MAX_BATCH || n_per_pair < 0) return KTLVO_ERR_INVALID;
    if (n_per_pair == 0) return KTLVO_OK;
    if (!d_prev || !d_next || !d_status || !d_err) return KTLVO_ERR_INVALID;
    int max_level = p->max_level;
    for (int i = 0; i < n_pairs; i++) {
        if (!slot_ok(ctx, pairs.prev[i]) || !slot_ok(ctx, pairs.next[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[pairs.prev[i]] < 0 || ctx->built[pairs.next[i]] < 0) return KTLVO_ERR_STATE;
        if (!ctx->deriv_ok[pairs.prev[i]]) return KTLVO_ERR_STATE; // ktlvo_build_pyramid makes the Scharr planes
        max_level = effective_max_level(ctx, pairs.prev[i], pairs.next[i], max_level);
    }
    for (int i = 0; i < n_pairs; i++) // every pair of one launch runs the same level count
        if (effective_max_level(ctx, pairs.prev[i], pairs.next[i], p->max_level) != max_level) return KTLVO_ERR_STATE;
    LkArgs a;
    a.slots = ctx->d_slots;
    a.prev = d_prev; a.next = d_next; a.status = d_status; a.err = d_err;
    a.n_per_pair = n_per_pair;
    a.max_level = max_level;
    int mc = p->max_count; // OpenCV clamps TermCriteria the same way
    a.max_count = mc < 0 ? 0 : mc > 100 ? 100 : mc;
    double eps = p->epsilon < 0 ? 0 : p->epsilon > 10 ? 10 : p->epsilon;
    a.eps_sq = eps * eps;
    a.min_eig_thr = p->min_eig_thr;
    a.d_n = d_n;
    a.pairs = pairs;
    for (int i = 0; i < n_pairs; i++) { a.prev_off[i] = prev_off[i]; a.next_off[i] = next_off[i]; }
    dim3 grid((n_per_pair + 3) / 4, n_pairs);
    ProfScope ps(ctx, KTLVO_K_PYRLK);
    hipLaunchKernelGGL(pyrlk_track_w21, grid, dim3(256), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_pyrlk_batch_dev(ktlvo_ctx *ctx, const int *prev_slots, const int *next_slots, int n_pairs,
                                     const float *d_prev_xy, int n_per_pair, float *d_next_xy, uint8_t *d_status,
                                     float *d_err, const ktlvo_lk_params *p) {
    if (!ctx || !prev_slots || !next_slots || n_pairs < 0) return KTLVO_ERR_INVALID;
    if (n_pairs > ctx->cfg.max_batch_pairs) return KTLVO_ERR_CAPACITY;
    if (n_pairs == 0) return KTLVO_OK;
    PairList pl;
    for (int i = 0; i < n_pairs; i++) { pl.prev[i] = prev_slots[i]; pl.next[i] = next_slots[i]; }
    return launch_pyrlk(ctx, pl, n_pairs, d_prev_xy, n_per_pair, d_next_xy, d_status, d_err, p);
}

extern "C" int ktlvo_pyrlk_dev(ktlvo_ctx *ctx, int prev_slot, int next_slot, const float *d_prev_xy, int n,
                               float *d_next_xy, uint8_t *d_status, float *d_err, const ktlvo_lk_params *p) {
    if (!ctx) return KTLVO_ERR_INVALID;
    PairList pl;
    pl.prev[0] = prev_slot; pl.next[0] = next_slot;
    return launch_pyrlk(ctx, pl, 1, d_prev_xy, n, d_next_xy, d_status, d_err, p);
}

extern "C" int ktlvo_pyrlk(ktlvo_ctx *ctx, int prev_slot, int next_slot, const float *prev_xy, int n, float *next_xy,
                           uint8_t *status, float *err, const ktlvo_lk_params *p) {
    if (!ctx || n < 0) return KTLVO_ERR_INVALID;
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    if (n == 0) {
        if (!p) return KTLVO_ERR_INVALID;
        return p->win == WIN ? KTLVO_OK : KTLVO_ERR_UNSUPPORTED;
    }
    if (!prev_xy || !next_xy || !status || !err) return KTLVO_ERR_INVALID;
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], prev_xy, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = ktlvo_pyrlk_dev(ctx, prev_slot, next_slot, ctx->d_pts[0], n, ctx->d_pts[1], ctx->d_status, ctx->d_err, p);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(next_xy, ctx->d_pts[1], (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(status, ctx->d_status, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(err, ctx->d_err, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}
