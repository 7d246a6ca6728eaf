// Image ingest and Gaussian pyramid (cv::buildOpticalFlowPyramid / cv::pyrDown as reached from
// featureTracking.cpp:205) for gfx950.
//
// Layout: each level lives in a padded plane (ctx.hpp: PAD); pyr_down_u8 writes the level's
// interior AND the reflect-101 copies of every pixel that fall into the frame, so the next
// level's loader and the LK kernel read plain rows with no border arithmetic.
#include "ctx.hpp"

namespace ktlvo {

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// Fills the frame of one level from its interior: thread per frame pixel.
__global__ __launch_bounds__(256) void pad_border_u8(const SlotDesc *slots, SlotList list, int level) {
    const LevelDesc L = slots[list.s[blockIdx.z]].lv[level];
    const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)) - PAD;
    const int y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6)) - PAD;
    if (x >= L.w + PAD || y >= L.h + PAD) return;
    if (x >= 0 && x < L.w && y >= 0 && y < L.h) return;
    L.origin[(ptrdiff_t)y * L.pitch + x] = L.origin[(ptrdiff_t)reflect101(y, L.h) * L.pitch + reflect101(x, L.w)];
}

constexpr int TW = 64; // output tile
constexpr int TH = 16;
constexpr int SR = 2 * TH + 4;  // source rows per tile (36)
constexpr int SD = (2 * TW + 8) / 4; // source dwords per tile row (34): bytes [2*tx0-4, 2*tx0+132)

// One level of cv::pyrDown: dst(x,y) = (sum_{i,j} k_i k_j src(2x+i, 2y+j) + 128) >> 8, k = [1 4 6 4 1].
// The source frame already holds the reflect-101 continuation, which is what pyrDown's own
// borderInterpolate yields.
__global__ __launch_bounds__(256) void pyr_down_u8(const SlotDesc *slots, SlotList list, int src_level) {
    const SlotDesc &S = slots[list.s[blockIdx.z]];
    const LevelDesc src = S.lv[src_level];
    const LevelDesc dst = S.lv[src_level + 1];
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    const int tid = threadIdx.x;

    __shared__ uint32_t tile[SR][SD + 1];
    __shared__ uint16_t hbuf[SR][TW];

    // source rows 2*ty0-2 .. +35, bytes 2*tx0-4 .. +135 (dword aligned: origin and pitch are)
    for (int i = tid; i < SR * SD; i += 256) {
        const int r = i / SD, c = i - r * SD;
        const int sy = 2 * ty0 - 2 + r, sx = 2 * tx0 - 4 + 4 * c;
        uint32_t v = 0;
        if (sy < src.h + PAD && sx + 3 < src.w + PAD) // upper-left is always inside the frame (PAD >= 4)
            v = *reinterpret_cast<const uint32_t *>(src.origin + (ptrdiff_t)sy * src.pitch + sx);
        tile[r][c] = v;
    }
    __syncthreads();

    // horizontal 1-4-6-4-1 at even source columns: two outputs per item from three dwords
    for (int i = tid; i < SR * (TW / 2); i += 256) {
        const int r = i / (TW / 2), j = i - r * (TW / 2);
        const uint32_t d0 = tile[r][j], d1 = tile[r][j + 1], d2 = tile[r][j + 2];
        // output x = 2j is centred on tile byte 4j+4, x = 2j+1 on byte 4j+6
        const uint32_t b2 = (d0 >> 16) & 0xff, b3 = d0 >> 24;
        const uint32_t c0 = d1 & 0xff, c1 = (d1 >> 8) & 0xff, c2 = (d1 >> 16) & 0xff, c3 = d1 >> 24;
        const uint32_t e0 = d2 & 0xff;
        hbuf[r][2 * j] = (uint16_t)(b2 + 4 * b3 + 6 * c0 + 4 * c1 + c2);
        hbuf[r][2 * j + 1] = (uint16_t)(c0 + 4 * c1 + 6 * c2 + 4 * c3 + e0);
    }
    __syncthreads();

    // vertical pass: thread -> row y, four consecutive x
    const int y = tid >> 4, xq = (tid & 15) * 4;
    const int oy = ty0 + y, ox = tx0 + xq;
    if (oy >= dst.h || ox >= dst.w) return;
    uint32_t out[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = xq + k;
        const uint32_t v = hbuf[2 * y][x] + 4 * hbuf[2 * y + 1][x] + 6 * hbuf[2 * y + 2][x] +
                           4 * hbuf[2 * y + 3][x] + hbuf[2 * y + 4][x];
        out[k] = (v + 128) >> 8;
    }
    uint8_t *row = dst.origin + (ptrdiff_t)oy * dst.pitch;
    if (ox + 3 < dst.w) {
        *reinterpret_cast<uint32_t *>(row + ox) = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    } else {
        for (int k = 0; k < 4 && ox + k < dst.w; k++) row[ox + k] = (uint8_t)out[k];
    }

    // reflect-101 copies into the frame (edge tiles only)
    const bool ynear = oy <= PAD || oy >= dst.h - 1 - PAD;
    const bool xnear = ox <= PAD || ox + 3 >= dst.w - 1 - PAD;
    if (!(ynear || xnear)) return;
    int ys[3], ny = 0;
    ys[ny++] = oy;
    if (oy >= 1 && oy <= PAD) ys[ny++] = -oy;
    if (oy <= dst.h - 2 && 2 * (dst.h - 1) - oy < dst.h + PAD) ys[ny++] = 2 * (dst.h - 1) - oy;
    for (int k = 0; k < 4 && ox + k < dst.w; k++) {
        const int x = ox + k;
        int xs[3], nx = 0;
        xs[nx++] = x;
        if (x >= 1 && x <= PAD) xs[nx++] = -x;
        if (x <= dst.w - 2 && 2 * (dst.w - 1) - x < dst.w + PAD) xs[nx++] = 2 * (dst.w - 1) - x;
        for (int a = 0; a < ny; a++)
            for (int b = 0; b < nx; b++)
                if (a | b) dst.origin[(ptrdiff_t)ys[a] * dst.pitch + xs[b]] = (uint8_t)out[k];
    }
}

// Scharr derivative of one level (calcSharrDeriv), used by the test hook only; the LK kernel
// computes the same expression on the fly from the padded image.
__global__ __launch_bounds__(256) void scharr_level(const SlotDesc *slots, int slot, int level, short2 *out) {
    const LevelDesc L = slots[slot].lv[level];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= L.w || y >= L.h) return;
    const uint8_t *p = L.origin + (ptrdiff_t)y * L.pitch + x;
    const int P = L.pitch;
    const int a00 = p[-P - 1], a01 = p[-P], a02 = p[-P + 1];
    const int a10 = p[-1], a12 = p[1];
    const int a20 = p[P - 1], a21 = p[P], a22 = p[P + 1];
    const int dx = 3 * (a02 - a00) + 10 * (a12 - a10) + 3 * (a22 - a20);
    const int dy = 3 * (a20 - a00) + 10 * (a21 - a01) + 3 * (a22 - a02);
    out[(size_t)y * L.w + x] = make_short2((short)dx, (short)dy);
}

static int ingest(ktlvo_ctx *ctx, int slot, const uint8_t *img, int w, int h, int stride, hipMemcpyKind kind) {
    if (!ctx || !img || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    if (w != ctx->cfg.width || h != ctx->cfg.height || stride < w) return KTLVO_ERR_INVALID;
    const LevelDesc &L = ctx->h_slots[slot].lv[0];
    KTLVO_HIP(ctx, hipMemcpy2DAsync(L.origin, L.pitch, img, stride, w, h, kind, ctx->stream));
    SlotList list;
    list.n = 1;
    list.s[0] = slot;
    dim3 grid((w + 2 * PAD + 63) / 64, (h + 2 * PAD + 3) / 4, 1);
    hipLaunchKernelGGL(pad_border_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, 0);
    KTLVO_HIP(ctx, hipGetLastError());
    ctx->built[slot] = 0;
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_upload_image(ktlvo_ctx *ctx, int slot, const uint8_t *img, int w, int h, int stride) {
    int rc = ingest(ctx, slot, img, w, h, stride, hipMemcpyHostToDevice);
    if (rc != KTLVO_OK) return rc;
    // the source is pageable host memory the caller may reuse as soon as we return
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}

extern "C" int ktlvo_set_image_dev(ktlvo_ctx *ctx, int slot, const uint8_t *d_img, int w, int h, int stride) {
    return ingest(ctx, slot, d_img, w, h, stride, hipMemcpyDeviceToDevice);
}

extern "C" int ktlvo_build_pyramids(ktlvo_ctx *ctx, const int *slots, int n_slots, int max_level, int win) {
    if (!ctx || n_slots < 0 || (n_slots > 0 && !slots) || max_level < 0) return KTLVO_ERR_INVALID;
    if (win != ctx->cfg.win) return KTLVO_ERR_UNSUPPORTED; // level count and frame were sized for cfg.win
    const int top = max_level < ctx->n_levels - 1 ? max_level : ctx->n_levels - 1;
    for (int i = 0; i < n_slots; i++) {
        if (!slot_ok(ctx, slots[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[slots[i]] < 0) return KTLVO_ERR_STATE;
    }
    for (int base = 0; base < n_slots; base += MAX_BATCH) {
        SlotList list;
        list.n = n_slots - base < MAX_BATCH ? n_slots - base : MAX_BATCH;
        for (int i = 0; i < list.n; i++) list.s[i] = slots[base + i];
        ProfScope ps(ctx, KTLVO_K_PYRDOWN);
        for (int l = 0; l < top; l++) {
            dim3 grid((ctx->lw[l + 1] + TW - 1) / TW, (ctx->lh[l + 1] + TH - 1) / TH, list.n);
            hipLaunchKernelGGL(pyr_down_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, l);
        }
        KTLVO_HIP(ctx, hipGetLastError());
    }
    for (int i = 0; i < n_slots; i++) ctx->built[slots[i]] = top;
    return KTLVO_OK;
}

extern "C" int ktlvo_build_pyramid(ktlvo_ctx *ctx, int slot, int max_level, int win) {
    return ktlvo_build_pyramids(ctx, &slot, 1, max_level, win);
}

extern "C" int ktlvo_pyramid_levels(ktlvo_ctx *ctx, int slot, int *max_level_built) {
    if (!ctx || !slot_ok(ctx, slot) || !max_level_built) return KTLVO_ERR_INVALID;
    *max_level_built = ctx->built[slot];
    return KTLVO_OK;
}

extern "C" int ktlvo_download_level(ktlvo_ctx *ctx, int slot, int level, uint8_t *dst, int *w, int *h) {
    if (!ctx || !slot_ok(ctx, slot) || !dst || level < 0) return KTLVO_ERR_INVALID;
    if (level > ctx->built[slot]) return KTLVO_ERR_STATE;
    const LevelDesc &L = ctx->h_slots[slot].lv[level];
    KTLVO_HIP(ctx, hipMemcpy2DAsync(dst, L.w, L.origin, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (w) *w = L.w;
    if (h) *h = L.h;
    return KTLVO_OK;
}

extern "C" int ktlvo_download_scharr(ktlvo_ctx *ctx, int slot, int level, int16_t *dst) {
    if (!ctx || !slot_ok(ctx, slot) || !dst || level < 0) return KTLVO_ERR_INVALID;
    if (level > ctx->built[slot]) return KTLVO_ERR_STATE;
    const LevelDesc &L = ctx->h_slots[slot].lv[level];
    const size_t bytes = (size_t)L.w * L.h * 4;
    int rc = ensure_misc(ctx, bytes);
    if (rc != KTLVO_OK) return rc;
    dim3 grid((L.w + 63) / 64, (L.h + 3) / 4);
    hipLaunchKernelGGL(scharr_level, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot, level, (short2 *)ctx->d_misc);
    KTLVO_HIP(ctx, hipGetLastError());
    KTLVO_HIP(ctx, hipMemcpyAsync(dst, ctx->d_misc, bytes, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}
