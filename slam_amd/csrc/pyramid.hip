// Image ingest and Gaussian pyramid (cv::buildOpticalFlowPyramid / cv::pyrDown as reached from
// featureTracking.cpp:205) for gfx950.
//
// Layout: each level lives in a padded plane (ctx.hpp: PAD); pyr_down_u8 writes the level's
// interior AND the reflect-101 copies of every pixel that fall into the frame, so the next
// level's loader and the LK kernel read plain rows with no border arithmetic.
#include "ctx.hpp"

#include <cstring>

namespace ktlvo {

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// Ingest: copies a w x h image (row stride `stride`) into level 0 of a slot and fills the
// reflect-101 frame in the same pass: one thread per byte of the padded plane.
__global__ __launch_bounds__(256) void ingest_pad_u8(const SlotDesc *slots, int slot, const uint8_t *src, int stride) {
    const LevelDesc L = slots[slot].lv[0];
    const int x = (int)(blockIdx.x * 64 + (threadIdx.x & 63)) - PAD;
    const int y = (int)(blockIdx.y * 4 + (threadIdx.x >> 6)) - PAD;
    if (x >= L.w + PAD || y >= L.h + PAD) return;
    L.origin[(ptrdiff_t)y * L.pitch + x] = src[(size_t)reflect101(y, L.h) * stride + reflect101(x, L.w)];
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

typedef short i16x2 __attribute__((ext_vector_type(2)));

// calcSharrDeriv for every level of the listed slots in one launch: grid (row segments of level 0,
// level, slot).  A thread produces eight consecutive pixels (two 16-byte stores of dx|dy pairs) from
// three 16-byte row pieces of the padded image, whose frame supplies calcSharrDeriv's reflect-101
// neighbours.  The arithmetic runs on packed int16 pairs (|values| <= 16 * 255):
//   T = 3 (r0 + r2) + 10 r1,  dx[x] = T[x+1] - T[x-1];   D = r2 - r0,  dy[x] = 3 (D[x-1] + D[x+1]) + 10 D[x].
// Only the interior of the plane is written; its frame stays zero (copyMakeBorder BORDER_CONSTANT).
__global__ __launch_bounds__(256) void scharr_planes(const SlotDesc *slots, SlotList list, int n_levels) {
    const int level = blockIdx.y;
    if (level >= n_levels) return;
    const LevelDesc L = slots[list.s[blockIdx.z]].lv[level];
    const int ow = (L.w + 7) >> 3;                 // octets per row
    const int q = blockIdx.x * 256 + threadIdx.x;  // octet index in raster order
    if (q >= ow * L.h) return;
    const int y = q / ow, x0 = (q - y * ow) * 8;
    // bytes x0-4 .. x0+11 of rows y-1, y, y+1 (origin and pitch are multiples of 4); byte i is pixel x0-4+i
    uint4 r[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
        r[j] = *reinterpret_cast<const uint4 *>(L.origin + (ptrdiff_t)(y - 1 + j) * L.pitch + x0 - 4);
    auto pr = [](uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(hi, lo, sel)); };
    // pairs (b[3+2j], b[4+2j]), j = 0..4, and (b[4+2j], b[5+2j]), j = 0..3, zero-extended
    auto odd = [&](const uint4 &d, i16x2 (&o)[5]) {
        o[0] = pr(d.y, d.x, 0x0c040c03u); o[1] = pr(d.y, d.y, 0x0c020c01u); o[2] = pr(d.z, d.y, 0x0c040c03u);
        o[3] = pr(d.z, d.z, 0x0c020c01u); o[4] = pr(d.w, d.z, 0x0c040c03u);
    };
    auto even = [&](const uint4 &d, i16x2 (&e)[4]) {
        e[0] = pr(d.y, d.y, 0x0c010c00u); e[1] = pr(d.y, d.y, 0x0c030c02u);
        e[2] = pr(d.z, d.z, 0x0c010c00u); e[3] = pr(d.z, d.z, 0x0c030c02u);
    };
    i16x2 o0[5], o1[5], o2[5], e0[4], e2[4];
    odd(r[0], o0); odd(r[1], o1); odd(r[2], o2);
    even(r[0], e0); even(r[2], e2);
    const i16x2 three = {3, 3}, ten = {10, 10};
    i16x2 T[5], Do[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        T[j] = three * (o0[j] + o2[j]) + ten * o1[j];
        Do[j] = o2[j] - o0[j];
    }
    uint32_t out[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {   // pixels x0 + 2j, x0 + 2j + 1
        const i16x2 dx = T[j + 1] - T[j];
        const i16x2 dy = three * (Do[j] + Do[j + 1]) + ten * (e2[j] - e0[j]);
        const uint32_t ux = __builtin_bit_cast(uint32_t, dx), uy = __builtin_bit_cast(uint32_t, dy);
        out[2 * j] = __builtin_amdgcn_perm(uy, ux, 0x05040100u);
        out[2 * j + 1] = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    }
    uint32_t *o = L.dorigin + (ptrdiff_t)y * L.pitch + x0;
    if (x0 + 7 < L.w) {
        reinterpret_cast<uint4 *>(o)[0] = make_uint4(out[0], out[1], out[2], out[3]);
        reinterpret_cast<uint4 *>(o)[1] = make_uint4(out[4], out[5], out[6], out[7]);
    } else {
        for (int k = 0; k < 8 && x0 + k < L.w; k++) o[k] = out[k];
    }
}

static int launch_ingest(ktlvo_ctx *ctx, int slot, const uint8_t *d_src, int stride) {
    dim3 grid((ctx->cfg.width + 2 * PAD + 63) / 64, (ctx->cfg.height + 2 * PAD + 3) / 4, 1);
    hipLaunchKernelGGL(ingest_pad_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot, d_src, stride);
    KTLVO_HIP(ctx, hipGetLastError());
    ctx->built[slot] = 0;
    ctx->deriv_ok[slot] = 0;
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

// Host image -> pinned staging ring -> device staging -> padded plane.  A pageable 2-D copy with
// differing pitches degenerates into one small transfer per row (measured 3.4 ms per KITTI frame);
// staging through pinned memory makes it one contiguous DMA and lets the call return as soon as
// the caller's buffer has been read.
extern "C" int ktlvo_upload_image(ktlvo_ctx *ctx, int slot, const uint8_t *img, int w, int h, int stride) {
    if (!ctx || !img || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    if (w != ctx->cfg.width || h != ctx->cfg.height || stride < w) return KTLVO_ERR_INVALID;
    const size_t bytes = (size_t)w * h;
    StageRing &r = ctx->stage;
    if (!r.h_buf[0]) {
        for (int i = 0; i < StageRing::N; i++) {
            KTLVO_HIP(ctx, hipHostMalloc((void **)&r.h_buf[i], bytes, hipHostMallocDefault));
            KTLVO_HIP(ctx, hipMalloc((void **)&r.d_buf[i], bytes));
            KTLVO_HIP(ctx, hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming));
        }
    }
    const int k = r.next;
    r.next = (r.next + 1) % StageRing::N;
    if (r.used[k]) KTLVO_HIP(ctx, hipEventSynchronize(r.done[k])); // that buffer's previous DMA and ingest are over
    if (stride == w) {
        memcpy(r.h_buf[k], img, bytes);
    } else {
        for (int y = 0; y < h; y++) memcpy(r.h_buf[k] + (size_t)y * w, img + (size_t)y * stride, w);
    }
    KTLVO_HIP(ctx, hipMemcpyAsync(r.d_buf[k], r.h_buf[k], bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_ingest(ctx, slot, r.d_buf[k], w);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipEventRecord(r.done[k], ctx->stream));
    r.used[k] = true;
    return KTLVO_OK;
}

extern "C" int ktlvo_set_image_dev(ktlvo_ctx *ctx, int slot, const uint8_t *d_img, int w, int h, int stride) {
    if (!ctx || !d_img || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    if (w != ctx->cfg.width || h != ctx->cfg.height || stride < w) return KTLVO_ERR_INVALID;
    return launch_ingest(ctx, slot, d_img, stride);
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
        {
            ProfScope ps(ctx, KTLVO_K_PYRDOWN);
            for (int l = 0; l < top; l++) {
                dim3 grid((ctx->lw[l + 1] + TW - 1) / TW, (ctx->lh[l + 1] + TH - 1) / TH, list.n);
                hipLaunchKernelGGL(pyr_down_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, l);
            }
        }
        KTLVO_HIP(ctx, hipGetLastError());
        {
            ProfScope ps2(ctx, KTLVO_K_SCHARR);
            const int octets0 = ((ctx->lw[0] + 7) / 8) * ctx->lh[0];
            dim3 grid((octets0 + 255) / 256, top + 1, list.n);
            hipLaunchKernelGGL(scharr_planes, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, top + 1);
            KTLVO_HIP(ctx, hipGetLastError());
        }
    }
    for (int i = 0; i < n_slots; i++) { ctx->built[slots[i]] = top; ctx->deriv_ok[slots[i]] = 1; }
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
    if (level > ctx->built[slot] || !ctx->deriv_ok[slot]) return KTLVO_ERR_STATE;
    const LevelDesc &L = ctx->h_slots[slot].lv[level];
    KTLVO_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)L.w * 4, L.dorigin, (size_t)L.pitch * 4, (size_t)L.w * 4, L.h,
                                    hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}
