// Corner response, ROI maximum and local-maximum test of cv::goodFeaturesToTrack in one pass over the image
// (featureTracking.cpp:102-103, 421-422), and the set-up of the detector's work area.  See corners.hpp.
#include "corners.hpp"
#include "sqrt_rn.hpp"

#include <cstring>

#include <algorithm>
#include <cmath>

namespace ktlvo {

constexpr int CT = 32;     // response tile owned by a workgroup
constexpr int CI = CT + 6; // image tile (3-pixel frame)
constexpr int CC = CT + 4; // covariance tile (2-pixel frame)
constexpr int CR = CT + 2; // response tile with its 1-pixel frame for the local-maximum test

__device__ __forceinline__ int reflect101d(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}


// Writes the ROI table into device memory and clears the per-ROI maxima and counters: one launch
// instead of a copy from pageable memory and two memsets (and no allocation on any path).
__global__ void corner_setup(RoiTable t, int n_roi, RoiDev *rois, unsigned *roi_max, int *cand_cnt, int *overflow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_roi) {
        RoiDev r;
        r.x = t.r[i].x; r.y = t.r[i].y; r.w = t.r[i].w; r.h = t.r[i].h; r.off = t.r[i].off;
        rois[i] = r;
        roi_max[i] = 0u;
        cand_cnt[i] = 0;
    }
    if (i == 0) { overflow[0] = 0; overflow[1] = 0; } // [1]: a blind route asks for the call to be done again the other way
}

// The same for a batch of frames that share one region: frame f's table entry is the region at candidate offset
// f * area, and its slot number goes into the list the candidates kernel reads.  One launch instead of two copies from
// host memory and three memsets (some 25 us of stream time in front of the batch's first kernel).
struct FrameSlots {
    int s[MAX_BATCH];
};
__global__ void corner_setup_frames(FrameSlots t, int n_frames, RoiDev region, RoiDev *rois, int *roi_slots, unsigned *roi_max,
                                    int *cand_cnt, int *overflow) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_frames) {
        RoiDev r = region;
        r.off = region.off * i; // region.off holds the area
        rois[i] = r;
        roi_slots[i] = t.s[i];
        roi_max[i] = 0u;
        cand_cnt[i] = 0;
    }
    if (i == 0) { overflow[0] = 0; overflow[1] = 0; }
}

__global__ __launch_bounds__(256) void corner_candidates(const SlotDesc *slots, int slot, const RoiDev *rois, float s1,
                                                         float s2, int harris, double k, float *resp_out,
                                                         unsigned *roi_max, unsigned long long *cand, int *cand_cnt,
                                                         int3 tiles, const int *roi_slots) {
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Consecutive ids therefore take
    // tiles that lie an eighth of the launch apart, and the workgroups of one XCD walk a contiguous run of
    // tiles in raster order: neighbouring tiles, which share image lines (32-pixel tiles, 128-byte lines) and
    // halo rows, then meet in one L2 instead of fetching those lines from memory once per XCD.
    const int tiles_x = tiles.x, tiles_per_roi = tiles.x * tiles.y, n_tiles = tiles_per_roi * tiles.z;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int roi_id = tile / tiles_per_roi, tr = tile - roi_id * tiles_per_roi;
    const RoiDev R = rois[roi_id];
    const int tx0 = (tr % tiles_x) * CT, ty0 = (tr / tiles_x) * CT;
    if (tx0 >= R.w || ty0 >= R.h) return;
    const LevelDesc L = slots[roi_slots ? roi_slots[roi_id] : slot].lv[0]; // a batch of frames: every ROI names its slot
    __shared__ __attribute__((aligned(16))) uint8_t img[CI][CI + 2]; // rows of 40 bytes: filled a dword at a time
    __shared__ float cxx[CC][CC + 1], cxy[CC][CC + 1], cyy[CC][CC + 1];
    __shared__ float rsp[CR][CR + 1];
    __shared__ int wave_cnt[4][4];
    __shared__ int block_base;
    const int tid = threadIdx.x;

    // image tile: ROI-relative (tx0 - 3 + ix, ty0 - 3 + iy).  Sobel reads the PARENT image beyond the ROI
    // (cv::Sobel on a view is not BORDER_ISOLATED; the level's reflect-101 frame is materialised), so the
    // tile is fetched unreflected, four bytes per load at whatever byte address the tile's rows start.  Only pixels
    // within one of the ROI can reach a result; rows are kept inside the plane's frame, and a row's last bytes may
    // come from beyond its own frame (the next row's), which is memory of the same plane.
    {
        typedef const __attribute__((address_space(1))) uint8_t *gsrc;
        typedef uint32_t u32_a1 __attribute__((aligned(1)));
        const gsrc base = (gsrc)L.origin + (R.x + tx0 - 3);
        const int last_row = L.h + PAD - 1;
        for (int i = tid; i < CI * (CI + 2) / 4; i += 256) {
            const int iy = i / ((CI + 2) / 4), k = i - iy * ((CI + 2) / 4);
            const int row = min(R.y + ty0 - 3 + iy, last_row);
            *(uint32_t *)&img[iy][4 * k] = *(const __attribute__((address_space(1))) u32_a1 *)(base + (ptrdiff_t)row * L.pitch + 4 * k);
        }
    }
    __syncthreads();
    // covariance tile: ROI-relative (tx0 - 2 + cx, ty0 - 2 + cy); the box filter runs on the ROI-sized
    // covariance image and reflects at the ROI's own edges.  The Sobel pair separates by rows: with
    // d(row) = a[x+1] - a[x-1] and q(row) = s1 a[x-1] + s2 a[x] + s1 a[x+1], dx = (d(y-1) + d(y+1)) s1 + d(y) s2 and
    // dy = q(y+1) - q(y-1).  A thread owns a column cx and a strip of S2 rows: it walks down the strip's S2 + 2 image
    // rows, forms d and q of a row once and keeps the last two rows' in registers.  A column beyond the ROI's edge
    // takes its mirror column's pixels right away; the two ROWS beyond the ROI's edges that a response can reach
    // (-1 and R.h) are copied from their mirror rows afterwards.
    constexpr int S2 = 6;
    static_assert(CC % S2 == 0, "strips tile the covariance rows");
    if (tid < CC * (CC / S2)) {
        const int st = tid / CC, cx = tid - st * CC, cy0 = st * S2;
        const int x = reflect101d(tx0 - 2 + cx, R.w) - (tx0 - 3);
        float d0 = 0.f, d1 = 0.f, q0 = 0.f, q1 = 0.f; // rows iy - 2 and iy - 1
#pragma unroll
        for (int j = 0; j < S2 + 2; j++) {
            const int iy = cy0 + j; // image-tile row; covariance row cy reads image rows cy, cy + 1, cy + 2
            const float am = img[iy][x - 1], a0 = img[iy][x], ap = img[iy][x + 1];
            const float d2 = ap - am;
            float q2 = s1 * am; q2 += s2 * a0; q2 += s1 * ap;
            if (j >= 2) {
                const int cy = iy - 2;
                const float dx = (d0 + d2) * s1 + d1 * s2;
                const float dy = q2 - q0;
                cxx[cy][cx] = dx * dx; cxy[cy][cx] = dx * dy; cyy[cy][cx] = dy * dy;
            }
            d0 = d1; d1 = d2; q0 = q1; q1 = q2;
        }
    }
    __syncthreads();
    if (tid < 2 * CC) {
        const int which = tid / CC, cx = tid - which * CC;
        const int yy = which ? R.h : -1;                     // ROI-relative row beyond the edge
        const int cy = yy - (ty0 - 2);                       // its place in the tile, if the tile reaches it ...
        const int my = reflect101d(yy, R.h) - (ty0 - 2);     // ... and its mirror row's
        if (cy >= 0 && cy < CC && my >= 0 && my < CC) {
            cxx[cy][cx] = cxx[my][cx]; cxy[cy][cx] = cxy[my][cx]; cyy[cy][cx] = cyy[my][cx];
        }
    }
    __syncthreads();
    // response tile: ROI-relative (tx0 - 1 + px, ty0 - 1 + py), inside the ROI only.  cv::boxFilter sums float input
    // in double, rows first ((a0 + a1) + a2 per row, then (r0 + r1) + r2 down the column).  A thread owns a column
    // px and a strip of STRIP rows: it walks down the strip's STRIP + 2 covariance rows, forms each row's three sums
    // once and keeps the last two in registers, so an output costs 1.4 row sums instead of three.
    float vmax = -INFINITY;
    constexpr int STRIP = 5, NSTRIP = (CR + STRIP - 1) / STRIP;
    if (tid < CR * NSTRIP) {
        const int st = tid / CR, px = tid - st * CR;
        const int py0 = st * STRIP, rows = min(STRIP, CR - py0);
        const int x = tx0 - 1 + px;
        double r0a = 0, r0b = 0, r0c = 0, r1a = 0, r1b = 0, r1c = 0;
        for (int j = 0; j < rows + 2; j++) {
            const int cy = py0 + j;
            double sa = (double)cxx[cy][px], sb = (double)cxy[cy][px], sc = (double)cyy[cy][px];
            sa += (double)cxx[cy][px + 1]; sb += (double)cxy[cy][px + 1]; sc += (double)cyy[cy][px + 1];
            sa += (double)cxx[cy][px + 2]; sb += (double)cxy[cy][px + 2]; sc += (double)cyy[cy][px + 2];
            if (j >= 2) {
                const int py = cy - 2, y = ty0 - 1 + py;
                float r = -INFINITY;
                if (x >= 0 && y >= 0 && x < R.w && y < R.h) {
                    const double a = (r0a + r1a) + sa, b = (r0b + r1b) + sb, c = (r0c + r1c) + sc;
                    const float fa = (float)a, fb = (float)b, fc = (float)c;
                    if (harris) {
                        const float t = fa * fc - fb * fb;
                        r = (float)((double)t - k * (double)(fa + fc) * (double)(fa + fc));
                    } else {
                        const float ha = fa * 0.5f, hc = fc * 0.5f;
                        r = (ha + hc) - sqrt_rn_radicand((ha - hc) * (ha - hc) + fb * fb);
                    }
                    if (px >= 1 && py >= 1 && px <= CT && py <= CT) { // this workgroup's own pixels
                        vmax = fmaxf(vmax, r);
                        if (resp_out) resp_out[R.off + y * R.w + x] = r;
                    }
                }
                rsp[py][px] = r;
            }
            r0a = r1a; r0b = r1b; r0c = r1c;
            r1a = sa; r1b = sb; r1c = sc;
        }
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((tid & 63) == 0 && vmax > -INFINITY) atomicMax(&roi_max[roi_id], f2ord(vmax));
    if (!cand) return;
    __syncthreads();

    // positive local maxima of the interior (goodFeaturesToTrack scans 1 <= x < w - 1, 1 <= y < h - 1).
    // Candidates are counted per wave (ballot), the counts meet in LDS and the workgroup takes its places
    // in the ROI's list with ONE global atomic: returning atomics on a single address serialise in L2.
    // The order of candidates in the list does not matter: the keys are sorted later and are unique.
    // A thread owns column ox and four rows: it forms the maximum of each of its six response rows' three entries
    // once and combines three of them per pixel.
    const int lane = tid & 63, wave = tid >> 6;
    const int ox = tid % CT, oy0 = (tid / CT) * 4;
    unsigned long long votes[4];
    float vals[4];
    float rm[6];
#pragma unroll
    for (int j = 0; j < 6; j++) rm[j] = fmaxf(fmaxf(rsp[oy0 + j][ox], rsp[oy0 + j][ox + 1]), rsp[oy0 + j][ox + 2]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int oy = oy0 + q;
        const int x = tx0 + ox, y = ty0 + oy;
        const float v = rsp[oy + 1][ox + 1];
        const float m = fmaxf(fmaxf(rm[q], rm[q + 1]), rm[q + 2]);
        const bool keep = x >= 1 && y >= 1 && x < R.w - 1 && y < R.h - 1 && v > 0.f && v == m;
        vals[q] = v;
        votes[q] = __ballot(keep);
        if (lane == 0) wave_cnt[wave][q] = __popcll(votes[q]);
    }
    __syncthreads();
    if (tid == 0) {
        int total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int q = 0; q < 4; q++) { const int c = wave_cnt[w][q]; wave_cnt[w][q] = total; total += c; }
        block_base = total ? atomicAdd(&cand_cnt[roi_id], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!((votes[q] >> lane) & 1ull)) continue;
        const int oy = oy0 + q;
        const int place = block_base + wave_cnt[wave][q] + __popcll(votes[q] & ((1ull << lane) - 1ull));
        cand[R.off + place] = ((unsigned long long)f2ord(vals[q]) << 32) | (unsigned)(((ty0 + oy) << 16) | (tx0 + ox));
    }
}

// ---- the same pass for large ROIs (whole frames), on registers ------------------------------------------------------
// corner_candidates pays for its LDS tiles: every stage writes a tile, waits for the workgroup and reads each value
// back three times, and two thirds of its instructions are that traffic and its addressing.  Here a WAVE owns a strip
// of 58 columns and a band of rows and walks down it one image row at a time: lane l holds column x0 - 3 + l, every
// stage keeps the two or three rows it still needs in registers, and the horizontal neighbours come from the
// neighbouring lanes by DPP wave shifts (no LDS, no barrier).  The three columns at each end of the wave are halo:
// lanes 1..62 have valid covariance terms, 2..61 valid responses, 3..60 are the strip's own pixels.  Per pixel the
// float and double operations, and their order, are corner_candidates' own, so responses, ROI maximum and candidate
// keys are identical (the candidates' order in the list differs, which the sort undoes: keys are unique).
// Candidates are gathered in a per-wave LDS buffer and go out with one returning atomic per flush.
// For ROIs of at least 8 x 8 pixels (mirror rows and columns 1 and n - 2 exist and are distinct from the edges).
constexpr int CW_COLS = 58;   // own columns of a wave
constexpr int CW_BUF = 448;   // candidate keys a wave holds before it flushes
// (the lane at the end of the wave that has no such neighbour reads zero: it is a halo lane)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// v with lane `to` replaced by lane `from`'s value (both wave-uniform)
__device__ __forceinline__ float copy_lane(float v, int from, int to, int lane) {
    const float src = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), from));
    return lane == to ? src : v;
}
constexpr int DPP_FROM_RIGHT = 0x130; // wave_shl:1, lane l reads lane l + 1
constexpr int DPP_FROM_LEFT = 0x138;  // wave_shr:1, lane l reads lane l - 1
struct CwGrid {
    int strips, bands, band_rows; // per ROI: strips x bands wave tasks, band_rows output rows per band
};
template <bool HARRIS>
__global__ __launch_bounds__(256) void corner_candidates_wave(const SlotDesc *slots, int slot, const RoiDev *rois, float s1, float s2,
                                                              double k, float *resp_out, unsigned *roi_max,
                                                              unsigned long long *cand, int *cand_cnt, CwGrid g, int n_roi,
                                                              const int *roi_slots) {
    __shared__ unsigned long long buf_all[4][CW_BUF + 64];
    __shared__ int wg_n[4], wg_roi[4], wg_base[4]; // the waves' last flush: keys held, ROI, places reserved by the ROI's first wave
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long *const buf = buf_all[wave];
    // wave tasks are dealt like corner_candidates' tiles: the workgroups of one XCD walk a contiguous run of tasks
    const int per_roi = g.strips * g.bands, n_tasks = per_roi * n_roi;
    const int n_groups = (n_tasks + 3) >> 2, per_xcd = (n_groups + 7) >> 3;
    const int group = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int task = group * 4 + wave;
    if (group >= n_groups) return; // (the whole workgroup)
    // a wave without work -- beyond the last task, or a strip or band beyond a smaller ROI of the list -- stays until the
    // workgroup's last flush, in which it holds no keys
    const bool has_task = task < n_tasks;
    const int roi_id = has_task ? task / per_roi : 0, tr = has_task ? task - roi_id * per_roi : 0;
    const int band = tr / g.strips, strip = tr - band * g.strips;
    const RoiDev R = rois[roi_id];
    const int W = R.w, H = R.h;
    const int x0 = strip * CW_COLS, y0 = band * g.band_rows, y1 = min(y0 + g.band_rows, H);
    const bool live = has_task && x0 < W && y0 < H;
    const LevelDesc L = slots[roi_slots ? roi_slots[roi_id] : slot].lv[0];
    typedef const __attribute__((address_space(1))) uint8_t *gsrc;
    const int x = x0 - 3 + lane; // ROI-relative column of this lane
    const gsrc col = (gsrc)L.origin + (ptrdiff_t)R.y * L.pitch + (R.x + x);
    const bool in_cols = x >= 0 && x < W;                       // a response exists here
    const bool own = lane >= 3 && lane <= 60 && x < W;          // this wave's own pixels
    const bool cand_col = own && x >= 1 && x < W - 1;           // goodFeaturesToTrack scans the interior
    // the box filter reflects at the ROI's edges: covariance column -1 is column 1, column W is column W - 2
    const int lane_m1 = 2 - x0, lane_w = W - x0 + 3;            // lanes of columns -1 and W, if this strip holds them
    const bool fix_left = lane_m1 >= 1, fix_right = lane_w <= 62;
    // A band needs the covariance rows y0 - 2 ... y1 + 1.  Row -1 is row 1 (only the top band reaches it: bands are at
    // least 8 rows), row H is row H - 2 -- which the last band reaches, and also the band before it when the last one
    // is a single row (several ROIs of different heights share one band height).  The step that would form row H has
    // no image row to read: "image row H + 1" stands for it.
    const bool top = y0 == 0;
    const int iy_first = top ? -1 : y0 - 3, iy_last = min(y1 + 2, H + 1);
    const int n_iter = live ? iy_last - iy_first + 1 : 0;
    const int yr_first = top ? 0 : y0 - 1;

    float vmax = -INFINITY;
    int n_buf = 0; // wave-uniform
    auto flush = [&]() {
        if (n_buf == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(&cand_cnt[roi_id], n_buf);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int i = lane; i < n_buf; i += 64) cand[R.off + base + i] = buf[i];
        n_buf = 0;
    };
    // One image row.  (dA, dB, dC) are the Sobel row terms d of rows iy - 2, iy - 1, iy (dC written here), (qA, ., qC)
    // the terms q; (sA, sB, sC) the box filter's row sums of covariance rows yc - 2, yc - 1, yc (sC written here);
    // (mA, mB, mC) the three-wide maxima of response rows yr - 2, yr - 1, yr, and vB the responses of row yr - 1.
    // The caller rotates the roles from step to step, so nothing is copied.
    struct Sums { double a, b, c; }; // of cxx, cxy, cyy
    auto step = [&](const int i, float &dA, float &dB, float &dC, float &qA, float &qC, Sums &sA, Sums &sB,
                    Sums &sC, float &mA, float &mB, float &mC, float &vB, float &vC) __attribute__((always_inline)) {
        const int iy = iy_first + i;
        const bool virt = iy == H + 1;
        const int yc = iy - 1;
        if (!virt) {
            const float a = (float)col[(ptrdiff_t)iy * L.pitch];
            const float am = dpp_f<DPP_FROM_LEFT>(a), ap = dpp_f<DPP_FROM_RIGHT>(a);
            dC = ap - am;
            float q = s1 * am; q += s2 * a; q += s1 * ap;
            qC = q;
            if (i < 2) return;
            const float dx = (dA + dC) * s1 + dB * s2;
            const float dy = qC - qA;
            float cxx = dx * dx, cxy = dx * dy, cyy = dy * dy;
            if (fix_left) {
                cxx = copy_lane(cxx, lane_m1 + 2, lane_m1, lane);
                cxy = copy_lane(cxy, lane_m1 + 2, lane_m1, lane);
                cyy = copy_lane(cyy, lane_m1 + 2, lane_m1, lane);
            }
            if (fix_right) {
                cxx = copy_lane(cxx, lane_w - 2, lane_w, lane);
                cxy = copy_lane(cxy, lane_w - 2, lane_w, lane);
                cyy = copy_lane(cyy, lane_w - 2, lane_w, lane);
            }
            // cv::boxFilter sums float input in double, a row's three entries left to right
            const double ca = (double)cxx, cb = (double)cxy, cc = (double)cyy;
            sC.a = (dpp_d<DPP_FROM_LEFT>(ca) + ca) + dpp_d<DPP_FROM_RIGHT>(ca);
            sC.b = (dpp_d<DPP_FROM_LEFT>(cb) + cb) + dpp_d<DPP_FROM_RIGHT>(cb);
            sC.c = (dpp_d<DPP_FROM_LEFT>(cc) + cc) + dpp_d<DPP_FROM_RIGHT>(cc);
            if (top && yc == 1) sA = sC; // covariance row -1 is row 1
        } else {
            sC = sA;                     // covariance row H is row H - 2
        }
        const int yr = yc - 1;
        if (yr < yr_first) return;
        // then down the column: (row yr - 1 + row yr) + row yr + 1
        const float fa = (float)((sA.a + sB.a) + sC.a), fb = (float)((sA.b + sB.b) + sC.b), fc = (float)((sA.c + sB.c) + sC.c);
        float r;
        if (HARRIS) {
            const float t = fa * fc - fb * fb;
            r = (float)((double)t - k * (double)(fa + fc) * (double)(fa + fc));
        } else {
            const float ha = fa * 0.5f, hc = fc * 0.5f;
            r = (ha + hc) - sqrt_rn_radicand((ha - hc) * (ha - hc) + fb * fb);
        }
        if (lane >= 2 && lane <= 61 && in_cols) vmax = fmaxf(vmax, r);
        if (resp_out && own && yr >= y0 && yr < y1) resp_out[R.off + yr * W + x] = r;
        r = in_cols ? r : -INFINITY;
        vC = r;
        mC = fmaxf(fmaxf(dpp_f<DPP_FROM_LEFT>(r), r), dpp_f<DPP_FROM_RIGHT>(r));
        const int yo = yr - 1;
        if (!cand || yo < y0 || yo < 1 || yo >= H - 1) return;
        const float m = fmaxf(fmaxf(mA, mB), mC);
        const bool keep = cand_col && vB > 0.f && vB == m;
        const unsigned long long vote = __builtin_amdgcn_ballot_w64(keep);
        if (vote) {
            if (keep) buf[n_buf + __popcll(vote & ((1ull << lane) - 1ull))] = ((unsigned long long)f2ord(vB) << 32) | (unsigned)((yo << 16) | x);
            n_buf += __popcll(vote);
            if (n_buf > CW_BUF) flush();
        }
    };
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
    Sums s0 = {0, 0, 0}, sb1 = {0, 0, 0}, sb2 = {0, 0, 0};
    float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, v0 = -INFINITY, v1 = -INFINITY, v2 = -INFINITY;
    // roles rotate with period three: step i writes slot i mod 3 of every stage
    for (int i = 0; i < n_iter; i += 3) {
        step(i, d1, d2, d0, q1, q0, sb1, sb2, s0, m1, m2, m0, v2, v0);
        if (i + 1 >= n_iter) break;
        step(i + 1, d2, d0, d1, q2, q1, sb2, s0, sb1, m2, m0, m1, v0, v1);
        if (i + 2 >= n_iter) break;
        step(i + 2, d0, d1, d2, q0, q2, s0, sb1, sb2, m0, m1, m2, v1, v2);
    }
    // The last flush -- for most waves the only one -- reserves the places of all waves of the workgroup that work on the
    // same ROI with ONE atomic: returning atomics on a single address queue up at one L2 channel at about 10 ns each, and
    // a 3840 x 2160 frame on its own has 3000 waves.
    if (lane == 0) { wg_n[wave] = n_buf; wg_roi[wave] = live ? roi_id : -1; }
    __syncthreads();
    {
        int before = 0, total = 0, first = wave;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (!live || wg_roi[w] != roi_id) continue;
            const int c = wg_n[w];
            if (w < wave) { before += c; first = min(first, w); }
            total += c;
        }
        if (wave == first && lane == 0) wg_base[wave] = total ? atomicAdd(&cand_cnt[roi_id], total) : 0;
        __syncthreads();
        const int base = wg_base[first] + before;
        for (int i = lane; i < n_buf; i += 64) cand[R.off + base + i] = buf[i];
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    // The maximum only grows, so a wave whose own is not above what is already there has nothing to add (atomics that
    // return nothing do not queue up the way the reservations above did: leaving these out changed little).
    if (lane == 0 && vmax > -INFINITY) {
        const unsigned o = f2ord(vmax);
        if (o > __hip_atomic_load(&roi_max[roi_id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&roi_max[roi_id], o);
    }
}

size_t corner_scratch_bytes(int n_roi, size_t total_px) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(sizeof(RoiDev) * (size_t)n_roi) + 2 * up(4 * (size_t)n_roi) + 256 + up(total_px * 8);
}

void corner_scratch_carve(void *base, int n_roi, size_t total_px, CornerScratch &s) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char *b = (char *)base;
    s.d_rois = (RoiDev *)b; b += up(sizeof(RoiDev) * (size_t)n_roi);
    s.d_max = (unsigned *)b; b += up(4 * (size_t)n_roi);
    s.d_cnt = (int *)b; b += up(4 * (size_t)n_roi);
    s.d_overflow = (int *)b; b += 256;
    s.d_cand = (unsigned long long *)b;
    (void)total_px;
}

// ROI table + cleared counters into the scratch area.  Up to RoiTable::MAX ROIs travel as kernel
// arguments; longer lists are copied.
bool rois_fit_table(const RoiDev *rois, int n_roi) {
    bool fits = n_roi <= RoiTable::MAX;
    for (int i = 0; i < n_roi && fits; i++) fits = rois[i].x <= 65535 && rois[i].y <= 65535;
    return fits;
}

int corner_scratch_setup(ktlvo_ctx *ctx, const RoiDev *rois, int n_roi, const CornerScratch &s) {
    if (rois_fit_table(rois, n_roi)) {
        RoiTable t;
        for (int i = 0; i < n_roi; i++) {
            t.r[i].x = (uint16_t)rois[i].x; t.r[i].y = (uint16_t)rois[i].y;
            t.r[i].w = (uint16_t)rois[i].w; t.r[i].h = (uint16_t)rois[i].h; t.r[i].off = rois[i].off;
        }
        hipLaunchKernelGGL(corner_setup, dim3((n_roi + 255) / 256), dim3(256), 0, ctx->stream, t, n_roi, s.d_rois, s.d_max,
                           s.d_cnt, s.d_overflow);
        KTLVO_HIP(ctx, hipGetLastError());
        return KTLVO_OK;
    }
    KTLVO_HIP(ctx, hipMemcpyAsync(s.d_rois, rois, sizeof(RoiDev) * n_roi, hipMemcpyHostToDevice, ctx->stream));
    KTLVO_HIP(ctx, hipMemsetAsync(s.d_max, 0, 4 * (size_t)n_roi, ctx->stream));
    KTLVO_HIP(ctx, hipMemsetAsync(s.d_cnt, 0, 4 * (size_t)n_roi, ctx->stream));
    KTLVO_HIP(ctx, hipMemsetAsync(s.d_overflow, 0, 8, ctx->stream));
    return KTLVO_OK;
}

int frames_scratch_setup(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi &roi, size_t area, const CornerScratch &s,
                         int *d_roi_slots) {
    FrameSlots t;
    for (int f = 0; f < n_frames; f++) t.s[f] = slots[f];
    hipLaunchKernelGGL(corner_setup_frames, dim3((n_frames + 255) / 256), dim3(256), 0, ctx->stream, t, n_frames,
                       RoiDev{roi.x, roi.y, roi.w, roi.h, (int)area}, s.d_rois, d_roi_slots, s.d_max, s.d_cnt, s.d_overflow);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// minw, minh: the smallest ROI of the call; ROIs of at least 8 x 8 pixels that do not go through corner_select's LDS
// (whole frames) take the register kernel, corner_candidates_wave, unless the context says otherwise.
int run_candidates(ktlvo_ctx *ctx, int slot, int n_roi, int maxw, int maxh, int use_harris, double k,
                          const CornerScratch &s, float *d_resp_out, bool want_candidates, const int *d_roi_slots,
                          bool large, int minw, int minh) {
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0); // cornerEigenValsVecs: 2^(ksize-1) * blockSize * 255
    const float s1 = (float)(1.0 * scale), s2 = (float)(2.0 * scale);
    if (large && ctx->candidates_wave && minw >= 8 && minh >= 8) {
        CwGrid g;
        g.strips = (maxw + CW_COLS - 1) / CW_COLS;
        // Bands of about 48 rows keep the rows a band computes beyond its own (five) at a tenth.  A launch that way too
        // small to give every SIMD a few waves (one frame, a few) takes lower bands, down to 12 rows: a wave's time is
        // its rows one after the other, and more, shorter waves finish sooner where the chip is idle anyway.
        g.bands = std::max(1, (maxh + 24) / 48);
        const long per_band = (long)g.strips * n_roi;
        if (per_band * g.bands < 2048)
            g.bands = std::max(g.bands, (int)std::min<long>(std::max(1, maxh / 12), (2048 + per_band - 1) / per_band));
        g.band_rows = (maxh + g.bands - 1) / g.bands;
        const int groups = (g.strips * g.bands * n_roi + 3) / 4;
        const dim3 grid(8 * ((groups + 7) / 8));
        unsigned long long *d_cand = want_candidates ? s.d_cand : nullptr;
        if (use_harris)
            hipLaunchKernelGGL(corner_candidates_wave<true>, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot, s.d_rois, s1, s2, k,
                               d_resp_out, s.d_max, d_cand, s.d_cnt, g, n_roi, d_roi_slots);
        else
            hipLaunchKernelGGL(corner_candidates_wave<false>, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot, s.d_rois, s1, s2, k,
                               d_resp_out, s.d_max, d_cand, s.d_cnt, g, n_roi, d_roi_slots);
        KTLVO_HIP(ctx, hipGetLastError());
        return KTLVO_OK;
    }
    const int3 tiles = {(maxw + CT - 1) / CT, (maxh + CT - 1) / CT, n_roi};
    const int n_tiles = tiles.x * tiles.y * tiles.z;
    hipLaunchKernelGGL(corner_candidates, dim3(8 * ((n_tiles + 7) / 8)), dim3(256), 0, ctx->stream, ctx->d_slots, slot, s.d_rois,
                       s1, s2, use_harris, k, d_resp_out, s.d_max, want_candidates ? s.d_cand : nullptr, s.d_cnt, tiles, d_roi_slots);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

} // namespace ktlvo
