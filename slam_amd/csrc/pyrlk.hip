// Pyramidal Lucas-Kanade for gfx950: the LKTrackerInvoker of cv::calcOpticalFlowPyrLK as the
// reference calls it (featureTracking.cpp:205-209: Size(21,21), maxLevel 4,
// TermCriteria(COUNT|EPS, 30, 0.01), flags 0, minEigThreshold 1e-4).
//
// One 64-lane wavefront tracks one feature through all pyramid levels.  The 21x21 window is cut
// into 63 horizontal runs of 7 pixels (21 rows x 3 runs); lane l owns run (row l % 21,
// columns 7*(l / 21) ..+6), lane 63 idles.  A lane keeps its seven template values I, Ix, Iy in
// registers for the whole level, so the template never touches LDS; per Gauss-Newton step a
// lane fetches two 8-byte runs of the next image (rows r and r+1) and forms its seven bilinear
// samples from them.  The 2x2 tensor and the mismatch vector are summed across the wave with
// DPP row operations.  Scharr derivatives (calcSharrDeriv) are computed on the fly from a
// 4x10 pixel neighbourhood of the padded previous image, so no derivative plane is stored.
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

__device__ __forceinline__ float dpp_add(float v, const int ctrl, const int row_mask) {
    // v + (v moved by the DPP pattern); lanes outside row_mask add 0
    int moved;
    switch (ctrl) { // the control word must be an immediate
    case 0xB1: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false); break;
    case 0x4E: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false); break;
    case 0x141: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false); break;
    case 0x140: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false); break;
    case 0x142: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false); break;
    default: moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false); break;
    }
    (void)row_mask;
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes, returned wave-uniform.  Fixed tree, so the result is deterministic.
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add(v, 0xB1, 0xF);  // quad_perm [1,0,3,2]
    v = dpp_add(v, 0x4E, 0xF);  // quad_perm [2,3,0,1]
    v = dpp_add(v, 0x141, 0xF); // row_half_mirror
    v = dpp_add(v, 0x140, 0xF); // row_mirror: every lane holds its 16-lane row sum
    v = dpp_add(v, 0x142, 0xA); // row_bcast15 into rows 1 and 3
    v = dpp_add(v, 0x143, 0xC); // row_bcast31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// N+1 aligned dwords starting at the dword holding p, shifted so that out[] begins at byte p.
template <int N>
__device__ __forceinline__ void load_bytes(const uint8_t *p, uint32_t (&out)[N]) {
    // image planes live in global memory; say so, or the loads become flat_load
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
    gptr q = (gptr)(p - sh);
    uint32_t d[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) d[i] = q[i];
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
}

template <int N>
__device__ __forceinline__ int byte_at(const uint32_t (&d)[N], int k) {
    return (int)((d[k >> 2] >> (8 * (k & 3))) & 0xffu);
}

__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
// 24-bit multiply: every product below has factors under 2^15 in magnitude
__device__ __forceinline__ int m24(int a, int b) { return __mul24(a, b); }

struct Weights {
    int w00, w01, w10, w11;
};
__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    Weights w;
    w.w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    w.w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    w.w10 = __float2int_rn((1.f - a) * b * 16384.f);
    w.w11 = 16384 - w.w00 - w.w01 - w.w10;
    return w;
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

        // ---- template: rows ipy+r-1 .. ipy+r+2, columns ipx+cx-1 .. ipx+cx+8 ------------------
        int Iv[RUN], ixv[RUN], iyv[RUN];
        {
            uint32_t rowd[4][3];
            const uint8_t *base = I.origin + (ptrdiff_t)(ipy + r - 1) * I.pitch + (ipx + cx - 1);
#pragma unroll
            for (int j = 0; j < 4; j++) load_bytes<3>(base + (ptrdiff_t)j * I.pitch, rowd[j]);
            // derivative buffers are zero outside the level (copyMakeBorder BORDER_CONSTANT)
            const int gx0 = ipx + cx, gy0 = ipy + r;
            const bool rowok[2] = {gy0 >= 0 && gy0 < I.h, gy0 + 1 >= 0 && gy0 + 1 < I.h};
            int dxv[2][RUN + 1], dyv[2][RUN + 1];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int k = 0; k <= RUN; k++) {
                    // position (row j+1, byte k+1) of the 4x10 neighbourhood
                    const int a00 = byte_at(rowd[j], k), a01 = byte_at(rowd[j], k + 1), a02 = byte_at(rowd[j], k + 2);
                    const int a10 = byte_at(rowd[j + 1], k), a12 = byte_at(rowd[j + 1], k + 2);
                    const int a20 = byte_at(rowd[j + 2], k), a21 = byte_at(rowd[j + 2], k + 1),
                              a22 = byte_at(rowd[j + 2], k + 2);
                    const bool ok = rowok[j] && (gx0 + k >= 0) && (gx0 + k < I.w);
                    const int dx = m24(3, (a02 - a00) + (a22 - a20)) + m24(10, a12 - a10);
                    const int dy = m24(3, (a20 - a00) + (a22 - a02)) + m24(10, a21 - a01);
                    dxv[j][k] = ok ? dx : 0;
                    dyv[j][k] = ok ? dy : 0;
                }
#pragma unroll
            for (int i = 0; i < RUN; i++) {
                const int p00 = byte_at(rowd[1], i + 1), p01 = byte_at(rowd[1], i + 2);
                const int p10 = byte_at(rowd[2], i + 1), p11 = byte_at(rowd[2], i + 2);
                Iv[i] = descale(m24(p00, w.w00) + m24(p01, w.w01) + m24(p10, w.w10) + m24(p11, w.w11), 14 - 5);
                const int ix = descale(m24(dxv[0][i], w.w00) + m24(dxv[0][i + 1], w.w01) + m24(dxv[1][i], w.w10) +
                                       m24(dxv[1][i + 1], w.w11), 14);
                const int iy = descale(m24(dyv[0][i], w.w00) + m24(dyv[0][i + 1], w.w01) + m24(dyv[1][i], w.w10) +
                                       m24(dyv[1][i + 1], w.w11), 14);
                ixv[i] = active ? ix : 0;
                iyv[i] = active ? iy : 0;
            }
        }
        int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
        for (int i = 0; i < RUN; i++) {
            s11 += m24(ixv[i], ixv[i]);
            s12 += m24(ixv[i], iyv[i]);
            s22 += m24(iyv[i], iyv[i]);
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
        float nx = ox - halfWin, ny = oy - halfWin;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                if (level == 0) status = 0;
                break;
            }
            w = bilinear_weights(nx - inx, ny - iny);
            uint32_t t0[2], t1[2];
            const uint8_t *jb = J.origin + (ptrdiff_t)(iny + r) * J.pitch + (inx + cx);
            load_bytes<2>(jb, t0);
            load_bytes<2>(jb + J.pitch, t1);
            int b1 = 0, b2 = 0;
#pragma unroll
            for (int i = 0; i < RUN; i++) {
                const int v = m24(byte_at(t0, i), w.w00) + m24(byte_at(t0, i + 1), w.w01) +
                              m24(byte_at(t1, i), w.w10) + m24(byte_at(t1, i + 1), w.w11);
                const int diff = descale(v, 14 - 5) - Iv[i];
                b1 += m24(diff, ixv[i]);
                b2 += m24(diff, iyv[i]);
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
                uint32_t t0[2], t1[2];
                const uint8_t *jb = J.origin + (ptrdiff_t)(iny + r) * J.pitch + (inx + cx);
                load_bytes<2>(jb, t0);
                load_bytes<2>(jb + J.pitch, t1);
                int e = 0;
#pragma unroll
                for (int i = 0; i < RUN; i++) {
                    const int v = m24(byte_at(t0, i), w.w00) + m24(byte_at(t0, i + 1), w.w01) +
                                  m24(byte_at(t1, i), w.w10) + m24(byte_at(t1, i + 1), w.w11);
                    const int diff = descale(v, 14 - 5) - Iv[i];
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
