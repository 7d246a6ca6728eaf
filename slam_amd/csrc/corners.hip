// Corner detection for gfx950: cv::GFTTDetector::detect -> cv::goodFeaturesToTrack as the reference
// runs it per sub-region (featureTracking.cpp:102-103, 408-430), and on whole frames.
//
//   corner_candidates     cornerEigenValsVecs (Sobel 3x3, 3x3 box) + calcHarris / calcMinEigenVal, the
//                         ROI maximum (minMaxLoc) and the 3x3 local-maximum test in ONE pass over the
//                         image: image, covariance and response tiles live in LDS, only candidate keys
//                         (value, raster index) and the ROI maximum go to global memory.  The response
//                         map is never written (ktlvo_corner_response, a test hook, asks for it).
//   corner_candidates_wave  the same pass for larger ROIs (whole frames) on registers: a wave walks down a strip of
//                         58 columns, every stage's last rows in registers, neighbours by DPP wave shifts; no LDS
//                         tiles, no barriers; same operations in the same order, half the instructions
//   corner_select         per ROI of at most 4096 pixels: bitonic sort of the keys in LDS
//                         (std::sort(greaterThanPtr)), threshold at quality * max, then the greedy
//                         minimum-distance pass over a cell grid in LDS
//   cand_filter, bitonic_*, corner_select_large
//                         the same for larger ROIs (whole frames): keys above the threshold are
//                         compacted, sorted in global memory, and one workgroup runs the greedy pass
//                         over chunks of 512 candidates against a cell grid, which lives in LDS when it
//                         fits and in global memory otherwise
//   corner_sort_frames    a batch of frames: one workgroup per frame compacts and sorts the frame's keys
//                         wholly in LDS (up to 16384 keys)
//   corner_sort_hist, _plan, _scatter, corner_sort_frames<true>
//                         frames with more keys: the keys are split into ranges of the response of at most 16384
//                         keys each, and every range is sorted in LDS by a workgroup of its own (the global sort
//                         takes over for a frame with thousands of equal responses)
//   lf_count ... lf_write the greedy pass as a relaxation over all candidates of a few frames at once (accepted once
//                         every stronger candidate within reach is rejected, rejected once one is accepted) for
//                         calls of few frames, where the walk's one workgroup per frame leaves the chip idle
//
// goodFeaturesToTrack thresholds the response map at quality * max (THRESH_TOZERO), dilates it 3x3 and
// keeps pixels that equal the dilated value.  For a pixel above a non-negative threshold that is the
// same as "not smaller than any of its eight raw neighbours": a neighbour at or below the threshold
// counts as zero, which is smaller anyway.  So the local-maximum test does not need the threshold, and
// the one pass emits every positive local maximum; the threshold, known once the whole ROI has been
// seen, is applied to the sorted keys.  An ROI whose maximum is not positive has no corners (for
// quality <= 1 nothing exceeds quality * max then), and none are emitted.
#include "ctx.hpp"

#include <functional>
#include "internal.hpp"

#include <cstring>

#include <algorithm>
#include <cmath>

namespace ktlvo {

constexpr int CT = 32;     // response tile owned by a workgroup
constexpr int CI = CT + 6; // image tile (3-pixel frame)
constexpr int CC = CT + 4; // covariance tile (2-pixel frame)
constexpr int CR = CT + 2; // response tile with its 1-pixel frame for the local-maximum test
constexpr int SEL_MAX = 4096;  // largest ROI area / candidate count handled by corner_select
constexpr int CELL_MAX = 1024; // largest min-distance cell grid handled by corner_select

__device__ __forceinline__ int reflect101d(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// monotone float -> uint map: the ROI maximum is taken with atomicMax on it, and candidate keys carry
// it in their upper half so that one unsigned compare orders them by value for either sign
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
                        r = (ha + hc) - sqrtf((ha - hc) * (ha - hc) + fb * fb);
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
            r = (ha + hc) - sqrtf((ha - hc) * (ha - hc) + fb * fb);
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
        const unsigned long long vote = __ballot(keep);
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

struct SelArgs {
    const RoiDev *rois;
    const unsigned long long *cand;
    const int *cand_cnt;
    const unsigned *roi_max;
    double quality;
    int max_corners, cap;
    double min_dist;
    float *xy, *resp;
    int *counts;
    int *overflow;
    int relax; // ROIs with up to 256 keys above the threshold take the greedy pass as a relaxation
};

// goodFeaturesToTrack's threshold for an ROI whose maximum response is maxVal: eig > maxVal * quality
// (cv::threshold(eig, eig, maxVal * qualityLevel, 0, THRESH_TOZERO) on a float image)
__device__ __forceinline__ float roi_threshold(unsigned max_ord, double quality) {
    return (float)((double)ord2f(max_ord) * quality);
}

// Keys sort descending by value, ties by higher raster index first.
__global__ __launch_bounds__(256) void corner_select(SelArgs a) {
    const int roi = blockIdx.x;
    const RoiDev R = a.rois[roi];
    int n = a.cand_cnt[roi];
    const int tid = threadIdx.x;
    __shared__ unsigned long long keys[SEL_MAX];
    __shared__ unsigned short cell_cnt[CELL_MAX];
    __shared__ unsigned cell_pts[CELL_MAX][4]; // x | y << 16
    __shared__ int n_above;
    if (n > SEL_MAX || R.w * R.h > SEL_MAX) {
        if (tid == 0) { *a.overflow = 1; a.counts[roi] = -1; }
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    if (tid == 0) n_above = 0;
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
    { // the keys above the threshold are a prefix of the sorted list
        const float thr = roi_threshold(a.roi_max[roi], a.quality);
        int c = 0;
        for (int i = tid; i < n; i += 256) c += ord2f((unsigned)(keys[i] >> 32)) > thr;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((tid & 63) == 0 && c) atomicAdd(&n_above, c);
        __syncthreads();
        n = n_above;
    }

    float *oxy = a.xy + (size_t)roi * a.cap * 2;
    float *oresp = a.resp + (size_t)roi * a.cap;
    const int limit = a.max_corners > 0 ? (a.max_corners < a.cap ? a.max_corners : a.cap) : a.cap;
    if (a.min_dist < 1) { // no distance test: first `limit` candidates in sorted order
        const int m = n < limit ? n : limit;
        for (int i = tid; i < m; i += 256) {
            const unsigned idx = (unsigned)keys[i];
            oxy[2 * i] = (float)(idx & 0xffffu); oxy[2 * i + 1] = (float)(idx >> 16);
            oresp[i] = ord2f((unsigned)(keys[i] >> 32));
        }
        if (tid == 0) a.counts[roi] = m;
        return;
    }
    const int cell = __double2int_rn(a.min_dist);
    const int gw = (R.w + cell - 1) / cell, gh = (R.h + cell - 1) / cell;
    if (gw * gh > CELL_MAX) { // the host checks this before launching
        if (tid == 0) { *a.overflow = 1; a.counts[roi] = -1; }
        return;
    }
    const double md2 = a.min_dist * a.min_dist;
    if (n <= 256 && a.relax) {
        // Up to 256 keys above the threshold (the rule for the reference's sub-regions): the walk's result without
        // the walk.  A candidate is accepted once every stronger candidate that the walk would test it against is
        // rejected, and rejected once one of them is accepted.  Thread i owns candidate i: it finds the stronger ones
        // within reach (in the nine cells around its own, as the walk's grid look-up) as a 256-bit mask in registers,
        // and rounds over two 256-bit masks in LDS (accepted, rejected; a wave writes its 64 bits by ballot) settle
        // everyone, one barrier per round; the accepted leave in rank order, the first `limit` of them.
        __shared__ uint2 info[256]; // x | y << 16, cell x | cell y << 16
        __shared__ unsigned long long s_acc[4], s_rej[4];
        const int lane = tid & 63, wv = tid >> 6;
        const bool have = tid < n;
        int x = 0, y = 0, xc = 0, yc = 0;
        if (have) {
            const unsigned idx = (unsigned)keys[tid];
            x = idx & 0xffffu; y = idx >> 16; xc = x / cell; yc = y / cell;
            info[tid] = uint2{idx, (unsigned)xc | ((unsigned)yc << 16)};
        }
        if (tid < 4) { s_acc[tid] = 0ull; s_rej[tid] = 0ull; }
        __syncthreads();
        // Squared distances as integers: a region has at most 4096 pixels, so dx^2 + dy^2 stays below 2^24, where the
        // walk's float arithmetic is exact as well, and an integer is below md^2 exactly when it is below its ceiling.
        const int md2i = md2 >= 1073741824.0 ? 0x7fffffff : (int)ceil(md2);
        // Key j's data is the same in every lane (one LDS address: a broadcast), so it goes through readfirstlane and
        // is unpacked by the scalar unit; the mask grows 32 keys a word, the bit a scalar too.
        unsigned pw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < 8; w++) {
            if (w * 32 > wv * 64 + 63) break; // only stronger candidates count: none of these 32 ranks before this wave's
            unsigned m = 0u;
            for (int j0 = w * 32; j0 < (w + 1) * 32 && j0 < n; j0 += 8) { // eight keys' data fetched together, then used
                uint2 q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) q[u] = info[j0 + u]; // (beyond n: whatever LDS holds, masked out below)
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const unsigned q0 = __builtin_amdgcn_readfirstlane(q[u].x), q1 = __builtin_amdgcn_readfirstlane(q[u].y);
                    const int dx = x - (int)(q0 & 0xffffu), dy = y - (int)(q0 >> 16);
                    const unsigned ax = (unsigned)(xc + 1 - (int)(q1 & 0xffffu)), ay = (unsigned)(yc + 1 - (int)(q1 >> 16));
                    // in the nine cells around, and within reach
                    const bool c = j0 + u < n && ax <= 2u && ay <= 2u && dx * dx + dy * dy < md2i;
                    m |= c ? 1u << ((j0 + u) & 31) : 0u;
                }
            }
            // of these 32 only the ranks below this thread's
            const int below = tid - w * 32;
            pw[w] = m & (below >= 32 ? 0xffffffffu : below <= 0 ? 0u : (1u << below) - 1u);
        }
        unsigned long long pred[4];
#pragma unroll
        for (int w = 0; w < 4; w++) pred[w] = (unsigned long long)pw[2 * w] | ((unsigned long long)pw[2 * w + 1] << 32);
        int st = have ? 0 : 2; // 0 undecided, 1 accepted, 2 rejected (or no candidate)
        for (int round = 0; round < 257; round++) { // the strongest undecided candidate decides in every round
            if (st == 0) {
                bool hit = false, clear = true;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    hit = hit || (pred[w] & s_acc[w]) != 0ull;
                    clear = clear && (pred[w] & ~s_rej[w]) == 0ull;
                }
                st = hit ? 2 : clear ? 1 : 0;
            }
            // (a wave may write its words while another still reads the old ones: the masks only grow, and a
            // decision made on an older mask is one made a round earlier)
            const unsigned long long am = __ballot(st == 1), rm = __ballot(st == 2 && have);
            if (lane == 0) { s_acc[wv] = am; s_rej[wv] = rm; }
            if (!__syncthreads_or(st == 0)) break;
        }
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = __popcll(s_acc[w]);
            before += w < wv ? c : 0;
            total += c;
        }
        if (st == 1) {
            const int pos = before + __popcll(s_acc[wv] & ((1ull << lane) - 1ull));
            if (pos < limit) {
                oxy[2 * pos] = (float)x; oxy[2 * pos + 1] = (float)y;
                oresp[pos] = ord2f((unsigned)(keys[tid] >> 32));
            }
        }
        if (tid == 0) a.counts[roi] = total < limit ? total : limit;
        return;
    }
    for (int i = tid; i < gw * gh; i += 256) cell_cnt[i] = 0;
    __syncthreads();
    if (tid >= 64) return; // the greedy pass is sequential over candidates; one wave runs it
    volatile unsigned short *vcnt = cell_cnt;
    volatile unsigned(*vpts)[4] = cell_pts;
    const int nb = tid >> 2, slot = tid & 3; // lanes 0..35: 9 neighbour cells x 4 slots
    int ncorners = 0;
    for (int i = 0; i < n; i++) {
        const unsigned idx = (unsigned)keys[i];
        const int y = idx >> 16, x = idx & 0xffffu;
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
            ncorners = -1;
            break;
        }
        if (tid == 0) {
            vpts[ci][cnt] = (unsigned)x | ((unsigned)y << 16);
            vcnt[ci] = (unsigned short)(cnt + 1);
            oxy[2 * ncorners] = (float)x; oxy[2 * ncorners + 1] = (float)y;
            oresp[ncorners] = ord2f((unsigned)(keys[i] >> 32));
        }
        __builtin_amdgcn_wave_barrier();
        ++ncorners;
        if (ncorners == limit) break;
    }
    if (tid == 0) a.counts[roi] = ncorners;
}

// ---- large ROIs (whole frames) ---------------------------------------------------------------------------
// Keys of one ROI whose value exceeds the ROI's threshold, compacted (in any order) into out[]; the
// remainder of out[] up to out_cap is cleared so that a power-of-two sort can run over it.
// blockIdx.y selects one of several ROIs handled by the launch (a batch of frames): ROI f's candidates start
// cand_stride keys after ROI f-1's, its output out_stride keys, its counters and maximum one element.
__global__ __launch_bounds__(256) void cand_filter(const unsigned long long *cand, const int *cand_cnt,
                                                   const unsigned *roi_max, double quality, unsigned long long *out,
                                                   int *out_cnt, size_t cand_stride, size_t out_stride) {
    cand += blockIdx.y * cand_stride; out += blockIdx.y * out_stride;
    cand_cnt += blockIdx.y; roi_max += blockIdx.y; out_cnt += blockIdx.y;
    const int n = *cand_cnt;
    const float thr = roi_threshold(*roi_max, quality);
    __shared__ int wave_off[4];
    __shared__ int block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) { // uniform trip count per workgroup
        const int i = base + threadIdx.x;
        unsigned long long key = 0;
        bool keep = false;
        if (i < n) {
            key = cand[i];
            keep = ord2f((unsigned)(key >> 32)) > thr;
        }
        const unsigned long long vote = __ballot(keep);
        if (lane == 0) wave_off[wave] = __popcll(vote);
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int w = 0; w < 4; w++) { const int c = wave_off[w]; wave_off[w] = total; total += c; }
            block_base = total ? atomicAdd(out_cnt, total) : 0;
        }
        __syncthreads();
        if (keep) out[block_base + wave_off[wave] + __popcll(vote & ((1ull << lane) - 1ull))] = key;
        __syncthreads();
    }
}

// Bitonic sort, descending, of np2 (a power of two) keys in global memory.  A workgroup owns BL
// consecutive keys in LDS for all compare distances below BL; larger distances run as global steps.
constexpr int BL = 4096;
__device__ __forceinline__ void bitonic_lds_steps(unsigned long long *s, int n_local, size_t gbase, size_t k, int j_from) {
    for (int j = j_from; j > 0; j >>= 1) {
        // thread t owns the pair (i, i | j), i = t with a zero bit inserted at position log2(j): every thread
        // has work in every step, and its pairs' loads go out together
#pragma unroll 4
        for (int t = threadIdx.x; t < (n_local >> 1); t += blockDim.x) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
            const unsigned long long x = s[i], y = s[p];
            const bool desc = ((gbase + i) & k) == 0;
            if (desc ? x < y : x > y) { s[i] = y; s[p] = x; }
        }
        __syncthreads();
    }
}
// (blockIdx.y selects one of several lists sorted by the same launch, list_stride keys apart.  A list's keys sit at
// its front and the rest is zero, the smallest key; n_used[list] says how many there are.  The steps that build
// sorted runs of length k move keys only inside aligned runs of k, so before and during them every key beyond n_used
// rounded up to a multiple of k is still zero; a compare-exchange between two zeros changes nothing, and blocks and
// pairs out there are skipped: the network is the same, only its idle part is not executed.)
__global__ __launch_bounds__(1024) void bitonic_local_sort(unsigned long long *keys, int np2, size_t list_stride,
                                                           const int *n_used) {
    keys += blockIdx.y * list_stride;
    if ((size_t)blockIdx.x * BL >= (size_t)n_used[blockIdx.y]) return;
    __shared__ unsigned long long s[BL];
    const int n_local = np2 < BL ? np2 : BL;
    const size_t gbase = (size_t)blockIdx.x * BL;
    for (int i = threadIdx.x; i < n_local; i += blockDim.x) s[i] = keys[gbase + i];
    __syncthreads();
    for (int k = 2; k <= n_local; k <<= 1) bitonic_lds_steps(s, n_local, gbase, (size_t)k, k >> 1);
    for (int i = threadIdx.x; i < n_local; i += blockDim.x) keys[gbase + i] = s[i];
}
__global__ __launch_bounds__(1024) void bitonic_local_merge(unsigned long long *keys, unsigned k, size_t list_stride,
                                                            const int *n_used) {
    keys += blockIdx.y * list_stride;
    if ((size_t)blockIdx.x * BL >= (((size_t)n_used[blockIdx.y] + k - 1) & ~((size_t)k - 1))) return;
    __shared__ unsigned long long s[BL];
    const size_t gbase = (size_t)blockIdx.x * BL;
    for (int i = threadIdx.x; i < BL; i += blockDim.x) s[i] = keys[gbase + i];
    __syncthreads();
    bitonic_lds_steps(s, BL, gbase, (size_t)k, BL >> 1);
    for (int i = threadIdx.x; i < BL; i += blockDim.x) keys[gbase + i] = s[i];
}
__global__ __launch_bounds__(256) void bitonic_global_step(unsigned long long *keys, unsigned j, unsigned k, size_t list_stride,
                                                           const int *n_used) {
    keys += blockIdx.y * list_stride;
    // thread t owns the pair (i, i | j) with i = t with a zero bit inserted at position log2(j)
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t i = ((t & ~((size_t)j - 1)) << 1) | (t & ((size_t)j - 1));
    if (i >= (((size_t)n_used[blockIdx.y] + k - 1) & ~((size_t)k - 1))) return; // i and i | j both lie in the zeros
    const size_t p = i | j;
    const unsigned long long x = keys[i], y = keys[p];
    const bool desc = (i & k) == 0;
    if (desc ? x < y : x > y) { keys[i] = y; keys[p] = x; }
}

// Filter and sort of one frame's candidates by ONE workgroup, wholly in LDS and registers (blockIdx.x = frame of a
// batch): the keys above the frame's threshold are compacted into LDS as they are read, padded with zeros to a power
// of two and sorted descending by the same bitonic network; the first out_cnt keys go to out[].  For frames of up to
// SORT_CAP keys above the threshold this replaces cand_filter, the clearing of the key buffer and the dozen launches
// of the global sort, and nothing has to be sized on the host, so the host need not wait for the candidate counts.
// A frame with more keys is left alone: its count is set to zero and *too_many is raised, and the caller runs the
// global path instead.
//
// The network runs on registers: each of the 1024 threads holds 16 keys, namely those whose 14-bit index agrees with
// the thread's in all but a window of four bits [LO, LO + 3] (the register number).  A compare-exchange at distance
// 2^b with b inside the window is then between two registers of one thread and needs neither LDS nor a barrier; for
// bits outside it the keys are re-dealt through LDS into another window.  Stage S of the network (runs of 2^S keys,
// distances 2^(S-1) ... 1) takes the windows [10,13], [8,11], [4,7], [0,3] as far as it needs them, highest first, so
// the whole sort of 16384 keys costs 28 passes through LDS instead of the 105 of a step-by-step network.  A key's
// place in LDS is its index plus index / 16: with the gap, the sixteen keys a thread owns in window 0 (consecutive
// ones) do not fall on the banks of its neighbours' keys.
constexpr int SORT_CAP = 16384;
constexpr int SORT_LDS_KEYS = SORT_CAP + SORT_CAP / 16;
typedef unsigned long long u64;
__device__ __forceinline__ int sort_place(int i) { return i + (i >> 4); }
template <int LO>
__device__ __forceinline__ int sort_index(int t, int r) { return ((t >> LO) << (LO + 4)) | (r << LO) | (t & ((1 << LO) - 1)); }
template <int LO>
__device__ __forceinline__ void sort_load(const u64 *sk, int t, int np2, u64 (&k)[16]) {
    if (sort_index<LO>(t, 0) >= np2) return; // this thread's sixteen keys all lie in the zero padding beyond np2
#pragma unroll
    for (int r = 0; r < 16; r++) k[r] = sk[sort_place(sort_index<LO>(t, r))];
}
template <int LO>
__device__ __forceinline__ void sort_store(u64 *sk, int t, int np2, const u64 (&k)[16]) {
    if (sort_index<LO>(t, 0) >= np2) return;
#pragma unroll
    for (int r = 0; r < 16; r++) sk[sort_place(sort_index<LO>(t, r))] = k[r];
}
// A compare-exchange on two keys as IEEE doubles.  Every key of a kept candidate has its top bit set (the order-
// preserving map of a positive response) and a finite exponent field (a finite response), and the padding is zero: read
// as doubles the keys are negative numbers, ordered the other way round than as unsigned integers, and the padding is
// +0.0, the largest -- "descending as unsigned" is "ascending as double", zeros last either way.  So the larger key is
// the smaller double, and a compare-exchange is one v_min_f64 and one v_max_f64 instead of a 64-bit compare and four
// selects (asm: the builtins would first canonicalise their inputs against signalling NaNs, which these never are).
__device__ __forceinline__ void sort_ce(u64 &first, u64 &second) { // first gets the larger key
    const double dx = __builtin_bit_cast(double, first), dy = __builtin_bit_cast(double, second);
    double mn, mx;
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(mn), "=&v"(mx) : "v"(dx), "v"(dy));
    first = __builtin_bit_cast(u64, mn);
    second = __builtin_bit_cast(u64, mx);
}
// the compare-exchanges of stage S for the bits BHI ... LO (all inside the window)
template <int LO, int S, int BHI>
__device__ __forceinline__ void sort_steps(int t, int np2, u64 (&k)[16]) {
    if (sort_index<LO>(t, 0) >= np2) return;
    // a run of 2^S keys is sorted descending if bit S of its keys' index is clear: inside the window that is a
    // property of the register number, above it of the thread.  A thread whose runs ascend flips its keys' sign bits
    // for the duration (which reverses their order as doubles) and exchanges like the others.
    bool flip = false;
    if constexpr (S > LO + 3) flip = (sort_index<LO>(t, 0) >> S) & 1;
    const u64 sign = flip ? 0x8000000000000000ull : 0ull;
    if constexpr (S > LO + 3) {
#pragma unroll
        for (int r = 0; r < 16; r++) k[r] ^= sign;
    }
#pragma unroll
    for (int b = BHI; b >= LO; --b) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r & (1 << (b - LO))) continue;
            bool asc = false;
            if constexpr (S <= LO + 3) asc = ((r >> (S - LO)) & 1) != 0;
            if (asc) sort_ce(k[r | (1 << (b - LO))], k[r]);
            else sort_ce(k[r], k[r | (1 << (b - LO))]);
        }
    }
    if constexpr (S > LO + 3) {
#pragma unroll
        for (int r = 0; r < 16; r++) k[r] ^= sign;
    }
}
template <int FROM, int TO>
__device__ __forceinline__ void sort_redeal(u64 *sk, int t, int np2, u64 (&k)[16]) {
    __syncthreads(); // every thread has read what the previous re-deal stored
    sort_store<FROM>(sk, t, np2, k);
    __syncthreads();
    sort_load<TO>(sk, t, np2, k);
}
constexpr int cmin(int a, int b) { return a < b ? a : b; }
// stage S, entered and left with the keys dealt for window 0
template <int S>
__device__ __forceinline__ void sort_stage(u64 *sk, int t, int np2, u64 (&k)[16]) {
    if constexpr (S >= 13) {
        sort_redeal<0, 10>(sk, t, np2, k);
        sort_steps<10, S, S - 1>(t, np2, k);
        sort_redeal<10, 8>(sk, t, np2, k);
        sort_steps<8, S, 9>(t, np2, k);
        sort_redeal<8, 4>(sk, t, np2, k);
        sort_steps<4, S, 7>(t, np2, k);
        sort_redeal<4, 0>(sk, t, np2, k);
    } else if constexpr (S >= 9) {
        sort_redeal<0, 8>(sk, t, np2, k);
        sort_steps<8, S, cmin(S - 1, 11)>(t, np2, k);
        sort_redeal<8, 4>(sk, t, np2, k);
        sort_steps<4, S, 7>(t, np2, k);
        sort_redeal<4, 0>(sk, t, np2, k);
    } else if constexpr (S >= 5) {
        sort_redeal<0, 4>(sk, t, np2, k);
        sort_steps<4, S, cmin(S - 1, 7)>(t, np2, k);
        sort_redeal<4, 0>(sk, t, np2, k);
    }
    sort_steps<0, S, cmin(S - 1, 3)>(t, np2, k);
}
// Frames with more keys than one workgroup sorts: the keys above the threshold are split by response into stretches
// ("buckets") of at most SORT_CAP keys, each sorted by a workgroup of its own behind the stronger ones.  Four launches:
// corner_sort_hist counts the keys in PLAN_BINS bins of equal width in the responses' ordered integer form, strongest
// first (a workgroup per slice of 16384 or 4096 candidates: histogram in LDS, then added to the frame's in global memory);
// corner_sort_plan (a workgroup per frame) cuts the row of bins wherever the running count passes a multiple of T, the
// frame's bucket size -- a bucket then holds fewer than T keys plus one bin's, which is why a bin with more than
// SORT_CAP - T keys (thousands of equal responses) sends the call to the global sort.  T is the frame's key count over
// the number of buckets wanted (PLAN_BUCKETS for a frame on its own: short sorts side by side where most of the chip
// is idle anyway; a compute unit's worth per frame for a batch), within PLAN_T_MIN ... PLAN_T_MAX, and never more than
// SORT_MAXB buckets;
// corner_sort_scatter moves every key into its bucket's stretch of the output (a workgroup per slice of candidates
// claims a run of places per bucket); corner_sort_frames<true> sorts each stretch where it lies.
constexpr int PLAN_BINS = 8192, SORT_MAXB = 63;
constexpr int PLAN_SLICE_BIG = 16384, PLAN_SLICE_SMALL = 4096; // candidates per workgroup of the histogram and the scatter; the small one where few workgroups would result
constexpr int PLAN_BUCKETS = 48, PLAN_T_MIN = 1792, PLAN_T_MAX = 7680;
struct SortPlan {
    unsigned max_ord;
    int shift, nb, total;
    int first_bin[SORT_MAXB + 1]; // bucket b holds bins first_bin[b] ... first_bin[b + 1] - 1
    int off[SORT_MAXB + 1];       // and starts at this place of the sorted list
    int cursor[SORT_MAXB + 1];    // places of the bucket's stretch already claimed
};
__device__ __forceinline__ int wg_scan_1024(int v, int *s_wave, int *total);
__device__ __forceinline__ int plan_shift(unsigned omax, float thr) {
    const unsigned othr = f2ord(thr), span = omax > othr ? omax - othr : 0u;
    int shift = 0;
    while ((span >> shift) >= (unsigned)PLAN_BINS) shift++;
    return shift;
}
__device__ __forceinline__ unsigned plan_bin(unsigned omax, int shift, u64 key) {
    return min((omax - (unsigned)(key >> 32)) >> shift, (unsigned)PLAN_BINS - 1);
}
template <int PLAN_SLICE>
__global__ __launch_bounds__(1024) void corner_sort_hist(const unsigned long long *cand, const int *cand_cnt,
                                                         const unsigned *roi_max, double quality, size_t cand_stride, int *g_hist) {
    __shared__ int hist[PLAN_BINS];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int n = cand_cnt[f], base0 = blockIdx.x * PLAN_SLICE;
    if (base0 >= n) return;
    cand += f * cand_stride;
    const unsigned omax = roi_max[f];
    const float thr = roi_threshold(omax, quality);
    const int shift = plan_shift(omax, thr);
    for (int i = tid; i < PLAN_BINS; i += 1024) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PLAN_SLICE / 4096; q++) {
        u64 key4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base0 + q * 4096 + u * 1024 + tid;
            key4[u] = i < n ? cand[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (key4[u] != 0ull && ord2f((unsigned)(key4[u] >> 32)) > thr) atomicAdd(&hist[plan_bin(omax, shift, key4[u])], 1);
    }
    __syncthreads();
    for (int i = tid; i < PLAN_BINS; i += 1024) {
        const int c = hist[i];
        if (c) atomicAdd(&g_hist[(size_t)f * PLAN_BINS + i], c);
    }
}
// (max_total: the most keys a list may have for what is queued behind; more raises *too_many as well)
__global__ __launch_bounds__(1024) void corner_sort_plan(const int *g_hist, const unsigned *roi_max, double quality, SortPlan *plans,
                                                         int *out_cnt, int *too_many, int max_total, int buckets_wanted) {
    __shared__ int pre[PLAN_BINS];
    __shared__ int s_wave[16], s_first[SORT_MAXB + 2], s_maxbin;
    const int f = blockIdx.x, tid = threadIdx.x;
    const unsigned omax = roi_max[f];
    if (tid < SORT_MAXB + 2) s_first[tid] = PLAN_BINS;
    if (tid == 0) s_maxbin = 0;
    __syncthreads();
    int c[8], sum = 0, mx = 0;
    const int4 lo = *(const int4 *)(g_hist + (size_t)f * PLAN_BINS + tid * 8), hi = *(const int4 *)(g_hist + (size_t)f * PLAN_BINS + tid * 8 + 4);
    c[0] = lo.x; c[1] = lo.y; c[2] = lo.z; c[3] = lo.w; c[4] = hi.x; c[5] = hi.y; c[6] = hi.z; c[7] = hi.w;
#pragma unroll
    for (int j = 0; j < 8; j++) { sum += c[j]; mx = max(mx, c[j]); }
    int total;
    int e = wg_scan_1024(sum, s_wave, &total);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0 && mx > 0) atomicMax(&s_maxbin, mx);
    int T = (total + buckets_wanted - 1) / buckets_wanted;
    T = T < PLAN_T_MIN ? PLAN_T_MIN : T > PLAN_T_MAX ? PLAN_T_MAX : T;
    // A bin's bucket is the running count before it over T; the first bin of a bucket is the lowest that has it.  Bucket
    // numbers do not fall from bin to bin, so a bin whose bucket is that of a non-empty bin before it -- one of this
    // thread's, or of an earlier lane of the wave (a running maximum over the lanes' last buckets) -- cannot be the
    // first, and is spared its atomic (thousands of LDS atomics on a handful of addresses: 1.7 of this kernel's 11 us;
    // the largest bin, one atomic per thread on ONE address before it was reduced over the wave first: 2.5 us).
    int bk[8], last = -1;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        pre[tid * 8 + j] = e;
        bk[j] = c[j] > 0 ? e / T : -1;
        last = c[j] > 0 ? bk[j] : last;
        e += c[j];
    }
    int seen = last;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(seen, d, 64);
        if ((tid & 63) >= d) seen = max(seen, o);
    }
    seen = __shfl_up(seen, 1, 64);
    if ((tid & 63) == 0) seen = -1; // (a wave's first lane knows of no bin before its own)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (bk[j] < 0) continue;
        if (bk[j] != seen && bk[j] <= SORT_MAXB) atomicMin(&s_first[bk[j]], tid * 8 + j);
        seen = bk[j];
    }
    __syncthreads();
    const int nb = total > 0 ? (total - 1) / T + 1 : 0;
    const bool bad = nb > SORT_MAXB || s_maxbin > SORT_CAP - T || total > max_total;
    SortPlan &P = plans[f];
    if (tid <= SORT_MAXB) {
        int fb = PLAN_BINS; // buckets no bin starts in are empty: they begin where the next one does
#pragma unroll
        for (int b = 0; b <= SORT_MAXB; b++) { // (every lane reads every entry: broadcasts, all in flight together)
            const int v = s_first[b];
            fb = b >= tid ? min(fb, v) : fb;
        }
        P.first_bin[tid] = fb;
        P.off[tid] = fb == PLAN_BINS ? total : pre[fb];
        P.cursor[tid] = 0;
    }
    if (tid == 0) {
        P.max_ord = omax; P.shift = plan_shift(omax, roi_threshold(omax, quality)); P.nb = bad ? 0 : nb; P.total = total;
        out_cnt[f] = bad ? 0 : total;
        if (bad) *too_many = 1;
    }
}
template <int PLAN_SLICE>
__global__ __launch_bounds__(1024) void corner_sort_scatter(const unsigned long long *cand, const int *cand_cnt,
                                                            const unsigned *roi_max, double quality, size_t cand_stride,
                                                            SortPlan *plans, unsigned long long *out, size_t out_stride) {
    __shared__ int s_first[SORT_MAXB + 1], s_cnt[SORT_MAXB + 1], s_base[SORT_MAXB + 1];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int n = cand_cnt[f], base0 = blockIdx.x * PLAN_SLICE;
    SortPlan &P = plans[f];
    if (base0 >= n || P.nb == 0) return;
    cand += f * cand_stride; out += f * out_stride;
    const unsigned omax = P.max_ord;
    const int shift = P.shift;
    const float thr = roi_threshold(roi_max[f], quality);
    if (tid <= SORT_MAXB) { s_first[tid] = P.first_bin[tid]; s_cnt[tid] = 0; }
    __syncthreads();
    u64 key[PLAN_SLICE / 1024];
    int where[PLAN_SLICE / 1024]; // bucket << 16 | place among the workgroup's keys of that bucket; -1: not kept
#pragma unroll
    for (int q = 0; q < PLAN_SLICE / 1024; q++) {
        const int i = base0 + q * 1024 + tid;
        key[q] = i < n ? cand[i] : 0ull;
    }
#pragma unroll
    for (int q = 0; q < PLAN_SLICE / 1024; q++) {
        where[q] = -1;
        if (key[q] != 0ull && ord2f((unsigned)(key[q] >> 32)) > thr) {
            const int bin = (int)plan_bin(omax, shift, key[q]);
            int b = 0; // the last bucket whose first bin is not beyond this one
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (b + step <= SORT_MAXB && s_first[b + step] <= bin) b += step;
            where[q] = (b << 16) | atomicAdd(&s_cnt[b], 1);
        }
    }
    __syncthreads();
    if (tid <= SORT_MAXB) s_base[tid] = s_cnt[tid] ? P.off[tid] + atomicAdd(&P.cursor[tid], s_cnt[tid]) : 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PLAN_SLICE / 1024; q++)
        if (where[q] >= 0) out[s_base[where[q] >> 16] + (where[q] & 0xffff)] = key[q];
}

// BUCKETS: blockIdx.y = the bucket of frame blockIdx.x the workgroup sorts, after plans[]; the frame's count is the plan's.
// (cells: where the relaxation follows, the keys are counted per minimum-distance cell as they are written out, which
// spares the relaxation a pass over them; cell == 0: not wanted)
struct CellCount {
    int cell, gw;
    int *cell_cnt;
    size_t cells_stride;
};
template <bool BUCKETS>
__global__ __launch_bounds__(1024) void corner_sort_frames(const unsigned long long *cand, const int *cand_cnt,
                                                           const unsigned *roi_max, double quality, unsigned long long *out,
                                                           int *out_cnt, size_t cand_stride, size_t out_stride, int *too_many,
                                                           const SortPlan *plans, CellCount cells) {
    extern __shared__ u64 sk[]; // SORT_LDS_KEYS keys
    __shared__ int s_n;
    cand += blockIdx.x * cand_stride; out += blockIdx.x * out_stride;
    const int n = cand_cnt[blockIdx.x];
    const float thr = roi_threshold(roi_max[blockIdx.x], quality);
    const int tid = threadIdx.x, lane = tid & 63;
    if constexpr (BUCKETS) { // the stretch is already in out[]: bring it into LDS
        const SortPlan &P = plans[blockIdx.x];
        const int b = blockIdx.y;
        if (b >= P.nb) return;
        const int lo = P.off[b], cnt = (b == SORT_MAXB ? P.total : P.off[b + 1]) - lo;
        if (cnt <= 0) return;
        out += lo;
        if (cnt <= SORT_CAP)
            for (int i = tid; i < cnt; i += 1024) sk[sort_place(i)] = out[i];
        if (tid == 0) s_n = cnt;
    } else {
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (int base = 0; base < n; base += 4096) { // uniform trip count; four loads in flight per thread
            u64 key4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = base + u * 1024 + tid;
                key4[u] = i < n ? cand[i] : 0ull; // a zero key is below every threshold's keys (responses kept are positive)
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const u64 key = key4[u];
                const bool keep = key != 0ull && ord2f((unsigned)(key >> 32)) > thr;
                const u64 vote = __ballot(keep);
                int woff = 0;
                if (lane == 0 && vote) woff = atomicAdd(&s_n, __popcll(vote));
                woff = __builtin_amdgcn_readfirstlane(woff);
                if (keep) {
                    const int p = woff + __popcll(vote & ((1ull << lane) - 1ull));
                    if (p < SORT_CAP) sk[sort_place(p)] = key;
                }
            }
        }
    }
    __syncthreads();
    const int m = s_n;
    if (m > SORT_CAP) { // (with buckets: cannot happen after a plan that was not turned down)
        if (tid == 0) { out_cnt[blockIdx.x] = 0; *too_many = 1; }
        return;
    }
    int np2 = 16, lg = 4; // at least one thread's worth
    while (np2 < m) { np2 <<= 1; lg++; }
    for (int i = m + tid; i < np2; i += 1024) sk[sort_place(i)] = 0ull;
    __syncthreads();
    u64 k[16];
#pragma unroll
    for (int r = 0; r < 16; r++) k[r] = 0ull;
    sort_load<0>(sk, tid, np2, k);
    sort_stage<1>(sk, tid, np2, k);
    sort_stage<2>(sk, tid, np2, k);
    sort_stage<3>(sk, tid, np2, k);
    sort_stage<4>(sk, tid, np2, k);
    if (lg >= 5) sort_stage<5>(sk, tid, np2, k);
    if (lg >= 6) sort_stage<6>(sk, tid, np2, k);
    if (lg >= 7) sort_stage<7>(sk, tid, np2, k);
    if (lg >= 8) sort_stage<8>(sk, tid, np2, k);
    if (lg >= 9) sort_stage<9>(sk, tid, np2, k);
    if (lg >= 10) sort_stage<10>(sk, tid, np2, k);
    if (lg >= 11) sort_stage<11>(sk, tid, np2, k);
    if (lg >= 12) sort_stage<12>(sk, tid, np2, k);
    if (lg >= 13) sort_stage<13>(sk, tid, np2, k);
    if (lg >= 14) sort_stage<14>(sk, tid, np2, k);
    // window 0: thread t holds keys 16 t ... 16 t + 15 of the sorted list
#pragma unroll
    for (int r = 0; r < 16; r++)
        if (16 * tid + r < m) out[16 * tid + r] = k[r];
    if (!BUCKETS && tid == 0) out_cnt[blockIdx.x] = m;
    if (BUCKETS && cells.cell) {
        int *cnt = cells.cell_cnt + blockIdx.x * cells.cells_stride;
#pragma unroll
        for (int r = 0; r < 16; r++)
            if (16 * tid + r < m) {
                const unsigned idx = (unsigned)k[r];
                atomicAdd(&cnt[(int)((idx >> 16) / (unsigned)cells.cell) * cells.gw + (int)((idx & 0xffffu) / (unsigned)cells.cell)], 1);
            }
    }
}

struct SelLargeArgs {
    const unsigned long long *keys; // sorted descending
    const int *n_ptr;               // number of keys above the threshold
    int w, h;
    double min_dist;
    int cell, gw, gh;
    int *cell_cnt;   // gw * gh, cleared                                         (grid in global memory only)
    uint4 *cell_pts; // gw * gh: up to four accepted corners as x | y << 16, empty places 0xFFFFFFFF   (likewise)
    int limit;
    float *xy, *resp;
    int *count;
    int *overflow;
    // a batch of ROIs of one size (blockIdx.x = ROI f): f's keys, cells and outputs lie these many elements after f-1's
    size_t keys_stride, cells_stride, out_stride;
};
constexpr int CH = 512;  // candidates per chunk of the greedy pass
constexpr int HT = 2048; // chunk-local hash of the survivors' cells (load factor <= 1/4)
constexpr int PAIRS_MAX = 192; // survivors of a chunk up to which all pairs are tested; beyond, the cell hash finds the neighbours
// Grid in LDS: a count (4 bytes) and four places of 2 bytes per cell, a corner being kept as its offset inside its
// cell (x | y << 8, empty 0xFFFF), which a cell of up to 256 pixels allows.
constexpr size_t SEL_LDS_PER_CELL = 12;
constexpr size_t SEL_LDS_GRID_MAX = (size_t)100 << 10; // what the kernel's static arrays leave of a compute unit's 160 KiB
static bool grid_fits_lds(int cell, int gw, int gh) {
    return cell <= 256 && (size_t)gw * gh * SEL_LDS_PER_CELL <= SEL_LDS_GRID_MAX;
}

// goodFeaturesToTrack's greedy pass: walk the candidates in sorted order and accept one unless an accepted
// corner lies closer than min_dist.  A candidate's fate depends only on the accepted corners ranked before
// it, so a chunk of CH consecutive candidates is handled in parallel: (1) every candidate is tested
// against the corners accepted from earlier chunks (the cell grid, owned by this single workgroup; the nine
// cells around a candidate are nine independent loads); (2) the survivors are
// compacted in rank order, and each is tested against the survivors ranked before it (all pairs, spread
// over the workgroup's threads, or through a hash of the survivors' cells when there are many), which
// leaves a bit mask of those too close per survivor; (3) rounds in
// which a survivor with an accepted predecessor in its mask is rejected, and one whose predecessors are
// all rejected is accepted, until all are decided -- the lowest undecided survivor always decides, and
// chains are short because a chunk's candidates are scattered over the frame; (4) the accepted ones take
// their output places in rank order and enter the grid.
// LG: the grid lives in LDS (dynamic: gw * gh * SEL_LDS_PER_CELL bytes), where a look-up costs a fraction of the L2
// round trip that bounds a chunk otherwise; frames whose grid does not fit keep it in global memory.
// Workgroup barrier of the greedy pass.  With the grid in LDS everything the waves tell each other goes through LDS,
// so the barrier only has to wait for LDS traffic: the next chunk's keys (fetched a chunk ahead) and the corners
// written out stay in flight across it, where __syncthreads would wait for both at every barrier.
template <bool LDS_ONLY>
__device__ __forceinline__ void sync_wg() {
    if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}
template <bool LG>
__global__ __launch_bounds__(CH) void corner_select_large(SelLargeArgs a) {
    a.keys += blockIdx.x * a.keys_stride; a.n_ptr += blockIdx.x; a.count += blockIdx.x;
    if (!LG) { a.cell_cnt += blockIdx.x * a.cells_stride; a.cell_pts += blockIdx.x * a.cells_stride; }
    a.xy += blockIdx.x * a.out_stride * 2; a.resp += blockIdx.x * a.out_stride;
    extern __shared__ unsigned char dyn_lds[];
    int *const g_cnt = (int *)dyn_lds;                                            // LG: gw * gh counts ...
    unsigned short *const g_pts = (unsigned short *)(dyn_lds + 4 * (size_t)a.gw * a.gh); // ... then gw * gh * 4 places
    __shared__ unsigned long long s_key[CH];
    __shared__ __attribute__((aligned(16))) unsigned s_pt[CH]; // survivors in rank order: x | y << 16
    __shared__ unsigned conf[CH][CH / 32 + 1];    // survivor i's predecessors within reach, a bit each (rows padded by a word: no two rows on the same banks)
    __shared__ int s_hit[CH];                     // survivor i has such a predecessor
    __shared__ unsigned ht_key[HT];               // many survivors: hash of their cells (cell index + 1, 0 = empty) ...
    __shared__ unsigned short ht_val[HT];         // ... and the survivor in that place
    __shared__ unsigned accm[CH / 32], rejm[CH / 32];
    __shared__ int wave_cnt[2][CH / 64];          // per-wave counts of a vote, in turn for the two votes of a chunk
    __shared__ int s_any[CH / 64];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = *a.n_ptr;
    const float md = (float)a.min_dist;
    const double md2 = a.min_dist * a.min_dist;
    auto too_close_xy = [&](int x, int y, int px, int py) {
        const float dx = (float)x - (float)px, dy = (float)y - (float)py;
        return (double)(dx * dx + dy * dy) < md2;
    };
    auto too_close = [&](int x, int y, unsigned p) { return too_close_xy(x, y, (int)(p & 0xffff), (int)(p >> 16)); };
    // n / cell for 0 <= n < 65536 without a division: with M = ceil(2^24 / cell), floor(n M / 2^24) = floor(n / cell)
    // (the excess n (M cell - 2^24) / (cell 2^24) stays below 1 / cell for n < 2^24 / cell, which covers 16 bits for
    // cells of up to 256 pixels; larger cells, which only the grid in global memory has, divide)
    const unsigned cell_m = (unsigned)(((1u << 24) + (unsigned)a.cell - 1u) / (unsigned)a.cell);
    auto cell_of = [&](int n) { return a.cell <= 256 ? (int)(((unsigned long long)(unsigned)n * cell_m) >> 24) : n / a.cell; };
    // Distances between points at most two cells apart, in the LDS variant at most 512 pixels: dx, dy and
    // dx^2 + dy^2 are integers that float holds exactly, so OpenCV's (double)(dx * dx + dy * dy) < minDistance^2 is the
    // integer comparison with the threshold rounded up.
    const int md2_up = (int)ceil(md2);
    auto near = [&](int dx, int dy) { return dx * dx + dy * dy < md2_up; };
    if (tid == 0) s_total = 0;
    if (LG) {
        const int cells = a.gw * a.gh;
        for (int q = tid; q < cells; q += CH) g_cnt[q] = 0;
        for (int q = tid; q < 2 * cells; q += CH) ((unsigned *)g_pts)[q] = 0xFFFFFFFFu;
    }
    sync_wg<LG>();
    unsigned long long next_key = tid < n ? a.keys[tid] : 0ull; // a chunk's keys are fetched one chunk ahead
    for (int base = 0; base < n; base += CH) {
        const int total0 = s_total;
        if (total0 >= a.limit) break;
        const unsigned long long my_key = next_key;
        if (base + CH + tid < n) next_key = a.keys[base + CH + tid];
        // What the previous chunk stored into a grid in global memory is read here by other waves of the same
        // workgroup: they run on one CU and share its L1, so workgroup scope orders it (a wider scope would write back
        // and invalidate the XCD's whole L2 twice per chunk, under every other frame of a batch).
        if (!LG) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // (1) against the grid
        const int i = base + tid;
        unsigned long long key = 0;
        int x = 0, y = 0, xc = 0, yc = 0;
        bool surv = false;
        if (i < n) {
            key = my_key;
            const unsigned idx = (unsigned)key;
            y = (int)(idx >> 16); x = (int)(idx & 0xffffu);
            xc = cell_of(x); yc = cell_of(y);
            bool hit = false;
            if (LG) {
                uint2 c[9];
                int ox[9], oy[9];
#pragma unroll
                for (int q = 0; q < 9; q++) { // cells off the grid are read as their clamped neighbour: a harmless repeat
                    const int cx = min(max(xc + q % 3 - 1, 0), a.gw - 1), cy = min(max(yc + q / 3 - 1, 0), a.gh - 1);
                    c[q] = *(const uint2 *)(g_pts + 4 * (cy * a.gw + cx));
                    ox[q] = cx * a.cell; oy[q] = cy * a.cell;
                }
                // a cell's places fill in order, and a second corner in a cell of min_dist pixels is rare: the places
                // beyond the first are only looked at when some lane of the wave has one
                bool more = false;
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    const unsigned v0 = c[q].x & 0xffffu;
                    hit = hit | ((v0 != 0xffffu) & near(x - ox[q] - (int)(v0 & 0xff), y - oy[q] - (int)(v0 >> 8)));
                    more = more | ((c[q].x >> 16) != 0xffffu);
                }
                if (__ballot(more)) {
#pragma unroll
                    for (int q = 0; q < 9; q++) {
                        const unsigned v[3] = {c[q].x >> 16, c[q].y & 0xffffu, c[q].y >> 16};
                        const int bx = x - ox[q], by = y - oy[q];
#pragma unroll
                        for (int k = 0; k < 3; k++)
                            hit = hit | ((v[k] != 0xffffu) & near(bx - (int)(v[k] & 0xff), by - (int)(v[k] >> 8)));
                    }
                }
            } else {
                uint4 c[9];
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    const int cx = min(max(xc + q % 3 - 1, 0), a.gw - 1), cy = min(max(yc + q / 3 - 1, 0), a.gh - 1);
                    c[q] = a.cell_pts[cy * a.gw + cx];
                }
#pragma unroll
                for (int q = 0; q < 9; q++)
                    hit = hit || too_close(x, y, c[q].x) || too_close(x, y, c[q].y) || too_close(x, y, c[q].z) || too_close(x, y, c[q].w);
            }
            surv = !hit;
        }
        // (2) survivors in rank order.  Every wave adds up the counts of the waves before it itself: one barrier, no
        // serial pass by one thread.
        const unsigned long long vote = __ballot(surv);
        if (lane == 0) wave_cnt[0][wave] = __popcll(vote);
        sync_wg<LG>();
        int S = 0, before = 0;
#pragma unroll
        for (int w = 0; w < CH / 64; w++) { const int c = wave_cnt[0][w]; S += c; before += w < wave ? c : 0; }
        if (surv) {
            const int p = before + __popcll(vote & ((1ull << lane) - 1ull));
            s_key[p] = key; s_pt[p] = (unsigned)x | ((unsigned)y << 16);
        }
        for (int q = tid; q < S * (CH / 32 + 1); q += CH) (&conf[0][0])[q] = 0u;
        if (tid < S) s_hit[tid] = 0;
        const bool hashed = S > PAIRS_MAX;
        if (hashed) {
            for (int q = tid; q < HT; q += CH) ht_key[q] = 0u;
            sync_wg<LG>();
            if (surv) { // several survivors of one cell take several places: the probe for a cell walks to the first empty place
                const int p = before + __popcll(vote & ((1ull << lane) - 1ull));
                const unsigned ck = (unsigned)(yc * a.gw + xc) + 1u;
                unsigned hsh = (ck * 2654435761u) >> 21; // 11 bits
                while (atomicCAS(&ht_key[hsh], 0u, ck) != 0u) hsh = (hsh + 1) & (HT - 1);
                ht_val[hsh] = (unsigned short)p;
            }
        }
        sync_wg<LG>();
        // Survivor i against the survivors ranked before it: all pairs.  A chunk's survivors are few (a few dozen once
        // the strongest corners are placed), and a pair costs a quarter of an LDS read that a whole wave shares and a
        // handful of instructions, none waiting for another -- a cell hash would find the few neighbours with far fewer
        // tests, but through chains of dependent LDS reads that took half of a chunk's time.  The threads form CH / S
        // groups, and group g takes the g-th block of predecessors, four to a read.
        // A chunk with many survivors (the first chunks of a frame, every chunk when the minimum distance is small) would
        // pay for S^2 pairs; there every survivor looks its neighbours up in a hash of the survivors' cells instead,
        // the nine cells' first probes going out together.
        auto close_to = [&](int dxi, int dyi) {
            if (LG) return near(dxi, dyi);
            const float dx = (float)dxi, dy = (float)dyi;
            return fabsf(dx) < md && fabsf(dy) < md && (double)(dx * dx + dy * dy) < md2;
        };
        if (hashed) {
            if (tid < S) {
                const int mx = (int)(s_pt[tid] & 0xffffu), my = (int)(s_pt[tid] >> 16);
                const int mxc = cell_of(mx), myc = cell_of(my);
                unsigned ck9[9], h9[9], k9[9];
#pragma unroll
                for (int c9 = 0; c9 < 9; c9++) {
                    const int cx = mxc + c9 % 3 - 1, cy = myc + c9 / 3 - 1;
                    const bool in = cx >= 0 && cy >= 0 && cx < a.gw && cy < a.gh;
                    ck9[c9] = in ? (unsigned)(cy * a.gw + cx) + 1u : 0u;
                    h9[c9] = (ck9[c9] * 2654435761u) >> 21;
                    k9[c9] = in ? ht_key[h9[c9]] : 0u;
                }
#pragma unroll
                for (int c9 = 0; c9 < 9; c9++) {
                    const unsigned ck = ck9[c9];
                    unsigned kk = k9[c9];
                    for (unsigned hsh = h9[c9]; kk != 0u; hsh = (hsh + 1) & (HT - 1), kk = ht_key[hsh]) {
                        if (kk != ck) continue;
                        const int q = ht_val[hsh];
                        if (q >= tid) continue;
                        if (close_to(mx - (int)(s_pt[q] & 0xffffu), my - (int)(s_pt[q] >> 16))) {
                            conf[tid][q >> 5] |= 1u << (q & 31);
                            s_hit[tid] = 1;
                        }
                    }
                }
            }
        } else if (S > 0) {
            const int ng = max(1, CH / S), g = tid / S, i = tid - g * S;
            const int blk = (((S + ng - 1) / ng) + 3) & ~3; // predecessors per group, a multiple of four
            if (g < ng) {
                const unsigned pi = s_pt[i];
                const int xi = (int)(pi & 0xffffu), yi = (int)(pi >> 16);
                const int j_end = min(i, (g + 1) * blk);
                for (int j0 = g * blk; j0 < j_end; j0 += 4) {
                    const uint4 q4 = *(const uint4 *)&s_pt[j0]; // entries beyond the survivors are stale and not looked at
                    const unsigned pj4[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int j = j0 + u;
                        const int dxi = xi - (int)(pj4[u] & 0xffffu), dyi = yi - (int)(pj4[u] >> 16);
                        if (close_to(dxi, dyi) && j < i) {
                            atomicOr(&conf[i][j >> 5], 1u << (j & 31));
                            s_hit[i] = 1;
                        }
                    }
                }
            }
        }
        sync_wg<LG>();
        int state = 3; // 0 undecided, 1 accepted, 2 rejected, 3 not a survivor
        if (tid < S) state = s_hit[tid] ? 0 : 1;
        // (3) rounds (none when no survivor of the chunk has a predecessor within reach, the usual case)
        for (;;) {
            // any survivor still undecided?  (a wave's vote meets the others' in LDS; the barrier further down, which
            // every wave passes before it votes again, keeps a vote from being overwritten before it has been read)
            const unsigned long long vu = __ballot(state == 0);
            if (lane == 0) s_any[wave] = vu != 0ull;
            sync_wg<LG>();
            int undecided = 0;
#pragma unroll
            for (int w = 0; w < CH / 64; w++) undecided |= s_any[w];
            if (!undecided) break;
            const unsigned long long va = __ballot(state == 1), vr = __ballot(state == 2);
            if (lane == 0) {
                accm[2 * wave] = (unsigned)va; accm[2 * wave + 1] = (unsigned)(va >> 32);
                rejm[2 * wave] = (unsigned)vr; rejm[2 * wave + 1] = (unsigned)(vr >> 32);
            }
            sync_wg<LG>();
            if (state == 0) {
                bool hit = false, pending = false;
                for (int wq = 0; wq <= (tid >> 5); wq++) {
                    const unsigned c = conf[tid][wq];
                    hit = hit || (c & accm[wq]);
                    pending = pending || (c & ~(accm[wq] | rejm[wq]));
                }
                if (hit) state = 2;
                else if (!pending) state = 1;
            }
        }
        // (4) accepted survivors, in rank order
        const unsigned long long va = __ballot(state == 1);
        if (lane == 0) wave_cnt[1][wave] = __popcll(va);
        sync_wg<LG>();
        int n_acc = 0, acc_before = 0;
#pragma unroll
        for (int w = 0; w < CH / 64; w++) { const int c = wave_cnt[1][w]; n_acc += c; acc_before += w < wave ? c : 0; }
        if (tid == 0) s_total = min(total0 + n_acc, a.limit); // read at the next chunk's start, beyond the barrier below
        if (state == 1) {
            const int pos = total0 + acc_before + __popcll(va & ((1ull << lane) - 1ull));
            if (pos < a.limit) {
                const int mx = (int)(s_pt[tid] & 0xffffu), my = (int)(s_pt[tid] >> 16);
                a.xy[2 * pos] = (float)mx; a.xy[2 * pos + 1] = (float)my;
                a.resp[pos] = ord2f((unsigned)(s_key[tid] >> 32));
                const int cxi = cell_of(mx), cyi = cell_of(my);
                const int ci = cyi * a.gw + cxi;
                if (LG) {
                    const int slot = atomicAdd(&g_cnt[ci], 1);
                    if (slot < 4) g_pts[4 * ci + slot] = (unsigned short)((mx - cxi * a.cell) | ((my - cyi * a.cell) << 8));
                    else *a.overflow = 2;
                } else {
                    const int slot = atomicAdd(&a.cell_cnt[ci], 1);
                    if (slot < 4) ((unsigned *)&a.cell_pts[ci])[slot] = (unsigned)mx | ((unsigned)my << 16);
                    else *a.overflow = 2; // cannot happen for cell = round(min_dist); reported instead of dropped
                }
            }
        }
        if (!LG) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        sync_wg<LG>();
    }
    if (tid == 0) *a.count = s_total;
}

// ---- the greedy pass in rounds (frames whose cell grid does not fit LDS) ------------------------------------------
// corner_select_large walks a frame's candidates in rank order, one workgroup per frame: for a 1920 x 1080 frame with
// 26,000 candidates that is fifty chunks against a grid in global memory, a millisecond during which a batch of a few
// dozen frames leaves most of the GPU idle.  The same result comes out of a relaxation that all candidates of all
// frames take part in at once.  A candidate's fate depends only on the candidates ranked before it that lie within
// min_dist of it (its predecessors): it is rejected as soon as one of them is accepted, and accepted as soon as all of
// them are rejected (a candidate without predecessors at once).  Decisions are final and only taken when they are
// certain, so rounds in any order and with any interleaving reach the sequential pass's result; on textured frames the
// undecided shrink by a factor of three to five per round and eight to twelve rounds settle everything.  The first
// `limit` accepted candidates in rank order are goodFeaturesToTrack's corners: cutting the sequential pass short at
// `limit` corners changes nothing about the decisions taken before.
// Candidates are binned by cell first (count, scan, fill: lf_count, lf_scan_cells, lf_scan_blocks, lf_fill; an item
// holds position and rank) so that a candidate looks at the nine cells around it; the output (lf_block_counts,
// lf_write) is a rank-ordered compaction.
// All arrays of frame f lie `*_stride` elements after those of frame f - 1; blockIdx.y = frame.
struct LfArgs {
    const unsigned long long *keys; // sorted descending, n_ptr[f] of them
    const int *n_ptr;
    unsigned char *state;           // 0 undecided, 1 accepted, 2 rejected
    int *cell_cnt, *cell_fill;      // candidates per cell; places of the cell already handed out
    int *start;                     // first place of a cell among its scan block's (LF_SCAN cells)
    int *blk_off;                   // first place of a scan block
    uint2 *items;                   // candidates in cell order: (x | y << 16, rank)
    int *blk_cnt;                   // accepted candidates per block of 1024 ranks
    int *flags;                     // [round * n_frames + f]: the launch left a candidate of the frame undecided
    int *unsettled; int last_round; // launch number last_round raises *unsettled as well (launches queued without looking at the flags)
    int tries;                      // times a candidate looks at its undecided predecessors again within a launch
    int cell, gw, gh, scan_blocks;
    double min_dist;
    int limit;
    float *xy, *resp;
    int *count;
    size_t keys_stride, cells_stride, out_stride, blk_stride;
    int n_frames;
};
constexpr int LF_SCAN = 4096;        // cells per workgroup of the scan
constexpr int LF_SCAN_BLOCKS = 1024; // at most: grids of up to 4M cells
constexpr int LF_PEND = 24;          // undecided predecessors a candidate keeps track of within a launch
__global__ __launch_bounds__(256) void lf_count(LfArgs a) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_ptr[f]) return;
    a.state[f * a.keys_stride + i] = 0;
    const unsigned idx = (unsigned)a.keys[f * a.keys_stride + i];
    const int c = (int)((idx >> 16) / (unsigned)a.cell) * a.gw + (int)((idx & 0xffffu) / (unsigned)a.cell);
    atomicAdd(&a.cell_cnt[f * a.cells_stride + c], 1);
}
// Exclusive sums of 1024 values, one per thread; *total = their sum (the same in every thread).
__device__ __forceinline__ int wg_scan_1024(int v, int *s_wave, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int t = s_wave[w];
        before += w < wave ? t : 0;
        all += t;
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}
// The cells' first places in two steps: a workgroup per LF_SCAN cells (four per thread) writes the places within its
// stretch and the stretch's total, then a workgroup per frame turns the totals into the stretches' first places.
// A cell's first place is the sum of the two; the entry behind the last cell closes the last cell's range.
__global__ __launch_bounds__(1024) void lf_scan_cells(LfArgs a) {
    __shared__ int s_wave[16];
    const int f = blockIdx.y;
    const size_t c = f * a.cells_stride + (size_t)blockIdx.x * LF_SCAN + threadIdx.x * 4;
    const bool in = (size_t)blockIdx.x * LF_SCAN + threadIdx.x * 4 < a.cells_stride; // the stride is a multiple of four
    const int4 v = in ? *(const int4 *)(a.cell_cnt + c) : int4{0, 0, 0, 0};
    int total;
    const int e = wg_scan_1024(v.x + v.y + v.z + v.w, s_wave, &total);
    if (in) *(int4 *)(a.start + c) = int4{e, e + v.x, e + v.x + v.y, e + v.x + v.y + v.z};
    if (threadIdx.x == 0) a.blk_off[f * LF_SCAN_BLOCKS + blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void lf_scan_blocks(LfArgs a) {
    __shared__ int s_wave[16];
    int *b = a.blk_off + blockIdx.x * LF_SCAN_BLOCKS;
    const int v = (int)threadIdx.x < a.scan_blocks ? b[threadIdx.x] : 0;
    int total;
    const int e = wg_scan_1024(v, s_wave, &total);
    if ((int)threadIdx.x < a.scan_blocks) b[threadIdx.x] = e;
}
// The same in one launch for grids of a few stretches: a workgroup per frame takes them one after the other.
__global__ __launch_bounds__(1024) void lf_scan_few(LfArgs a) {
    __shared__ int s_wave[16];
    const int f = blockIdx.x;
    int run = 0;
    for (int b = 0; b < a.scan_blocks; b++) {
        const size_t off = (size_t)b * LF_SCAN + threadIdx.x * 4;
        const bool in = off < a.cells_stride;
        const size_t c = f * a.cells_stride + off;
        const int4 v = in ? *(const int4 *)(a.cell_cnt + c) : int4{0, 0, 0, 0};
        int total;
        const int e = wg_scan_1024(v.x + v.y + v.z + v.w, s_wave, &total);
        if (in) *(int4 *)(a.start + c) = int4{e, e + v.x, e + v.x + v.y, e + v.x + v.y + v.z};
        if (threadIdx.x == 0) a.blk_off[f * LF_SCAN_BLOCKS + b] = run;
        run += total;
    }
}
__device__ __forceinline__ int lf_first(const int *start, const int *blk_off, int c) { return start[c] + blk_off[c / LF_SCAN]; }
__global__ __launch_bounds__(256) void lf_fill(LfArgs a) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_ptr[f]) return;
    const unsigned idx = (unsigned)a.keys[f * a.keys_stride + i];
    const int c = (int)((idx >> 16) / (unsigned)a.cell) * a.gw + (int)((idx & 0xffffu) / (unsigned)a.cell);
    const int k = lf_first(a.start + f * a.cells_stride, a.blk_off + f * LF_SCAN_BLOCKS, c) +
                  atomicAdd(&a.cell_fill[f * a.cells_stride + c], 1);
    a.items[f * a.keys_stride + k] = uint2{idx, (unsigned)i};
}
// One launch of the relaxation.  A candidate collects its predecessors within reach from the three rows of cells
// around it (the three cells of a row are one stretch of the item list; the first four items of every stretch are
// fetched at once, the rest one by one), reads their states and decides if it can; otherwise it keeps the undecided
// ones in LDS and looks at them again, up to a.tries times (64 unless the context says otherwise).  Ranks run with the thread index and workgroups start in
// index order, so a candidate's predecessors belong to workgroups that started no later than its own: most chains
// settle inside one launch, and what is left (a candidate that ran out of tries, or has more undecided predecessors
// than it can keep) raises the launch's flag for its frame.
__global__ __launch_bounds__(256) void lf_round(LfArgs a, int round) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = a.n_ptr[f];
    if (i >= n) return;
    unsigned char *state = a.state + f * a.keys_stride;
    if (__hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    __shared__ int s_pend[256][LF_PEND + 1];
    int *pend = s_pend[threadIdx.x];
    const int *start = a.start + f * a.cells_stride, *blk_off = a.blk_off + f * LF_SCAN_BLOCKS;
    const uint2 *items = a.items + f * a.keys_stride;
    const unsigned idx = (unsigned)a.keys[f * a.keys_stride + i];
    const int x = (int)(idx & 0xffffu), y = (int)(idx >> 16), xc = x / a.cell, yc = y / a.cell;
    const int cx0 = max(xc - 1, 0), cx1 = min(xc + 1, a.gw - 1);
    const float md = (float)a.min_dist;
    const double md2 = a.min_dist * a.min_dist;
    int k0[3], k1[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const int cy = yc + r - 1;
        const bool in = cy >= 0 && cy < a.gh;
        const int row = (in ? cy : yc) * a.gw;
        k0[r] = lf_first(start, blk_off, row + cx0);
        k1[r] = in ? lf_first(start, blk_off, row + cx1 + 1) : k0[r];
    }
    uint2 head[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) head[r][j] = items[min(k0[r] + j, n - 1)];
    bool rejected = false, lost = false;
    int npend = 0;
    // the stronger candidates within reach are collected first and their states read afterwards, several at a time
    // (a read followed by a branch on its value would make every one of them wait for the one before)
    auto look = [&](uint2 it) {
        const int q = (int)it.y;
        if (q >= i) return;
        const float dx = (float)(x - (int)(it.x & 0xffffu)), dy = (float)(y - (int)(it.x >> 16));
        if (!(fabsf(dx) < md && fabsf(dy) < md && (double)(dx * dx + dy * dy) < md2)) return;
        if (npend < LF_PEND) { pend[npend++] = q; return; }
        const unsigned char sq = __hip_atomic_load(&state[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // no room: looked at once
        if (sq == 1) rejected = true;
        else if (sq == 0) lost = true;
    };
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (k0[r] + j < k1[r]) look(head[r][j]);
        for (int k = k0[r] + 4; k < k1[r]; k++) look(items[k]);
    }
    // A decision is published the moment it is made, inside the loop: lanes of one wave that wait for each other
    // would otherwise only see it once the whole wave has left the loop.
    bool done = false;
    for (int t = 0; t <= a.tries && !done; t++) {
        if (t) __builtin_amdgcn_s_sleep(2);
        int keep = 0;
        for (int k = 0; k < npend; k += 4) {
            unsigned char sq[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                sq[j] = k + j < npend ? __hip_atomic_load(&state[pend[k + j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned char)2;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (sq[j] == 1) rejected = true;
                else if (sq[j] == 0) pend[keep++] = pend[k + j]; // keep <= k + j: nothing not yet read is overwritten
            }
        }
        npend = keep;
        if (rejected) __hip_atomic_store(&state[i], (unsigned char)2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (npend == 0 && !lost) __hip_atomic_store(&state[i], (unsigned char)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        done = rejected || (npend == 0 && !lost);
        if (lost && npend == 0) break; // predecessors it could not keep: the next launch
    }
    if (!done) {
        a.flags[round * a.n_frames + f] = 1;
        if (round == a.last_round) *a.unsettled = 1;
    }
}
__global__ __launch_bounds__(1024) void lf_block_counts(LfArgs a) {
    const int f = blockIdx.y, i = blockIdx.x * 1024 + threadIdx.x;
    const bool acc = i < a.n_ptr[f] && a.state[f * a.keys_stride + i] == 1;
    const int c = __syncthreads_count(acc);
    if (threadIdx.x == 0) a.blk_cnt[f * a.blk_stride + blockIdx.x] = c;
}
__global__ __launch_bounds__(1024) void lf_write(LfArgs a) {
    const int f = blockIdx.y, i = blockIdx.x * 1024 + threadIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n_ptr[f], nblk = (n + 1023) / 1024;
    __shared__ int s_red[16], s_wave[16];
    // accepted candidates of the blocks before this one
    int before = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 1024) before += a.blk_cnt[f * a.blk_stride + b];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if (lane == 0) s_red[wave] = before;
    const bool acc = i < n && a.state[f * a.keys_stride + i] == 1;
    const unsigned long long vote = __ballot(acc);
    if (lane == 0) s_wave[wave] = __popcll(vote);
    __syncthreads();
    int base = 0, in_block = 0, own = 0;
    for (int w = 0; w < 16; w++) { base += s_red[w]; own += w < wave ? s_wave[w] : 0; in_block += s_wave[w]; }
    if (acc) {
        const int pos = base + own + __popcll(vote & ((1ull << lane) - 1ull));
        if (pos < a.limit) {
            const unsigned long long key = a.keys[f * a.keys_stride + i];
            float *xy = a.xy + f * a.out_stride * 2;
            xy[2 * pos] = (float)((unsigned)key & 0xffffu); xy[2 * pos + 1] = (float)((unsigned)key >> 16);
            a.resp[f * a.out_stride + pos] = ord2f((unsigned)(key >> 32));
        }
    }
    if ((int)blockIdx.x == max(nblk - 1, 0) && tid == 0) a.count[f] = min(base + in_block, a.limit);
}

// first `limit` keys of a sorted list, no distance test (min_dist < 1)
__global__ void corner_take_first(const unsigned long long *keys, const int *n_ptr, int w, int limit, float *xy,
                                  float *resp, int *count, size_t keys_stride, size_t out_stride) {
    keys += blockIdx.y * keys_stride; n_ptr += blockIdx.y; count += blockIdx.y; // blockIdx.y = ROI of a batch
    xy += blockIdx.y * out_stride * 2; resp += blockIdx.y * out_stride;
    const int n = min(*n_ptr, limit);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned idx = (unsigned)keys[i];
        xy[2 * i] = (float)(idx & 0xffffu); xy[2 * i + 1] = (float)(idx >> 16);
        resp[i] = ord2f((unsigned)(keys[i] >> 32));
    }
    if (i == 0) *count = n;
}

static int check_rois(const ktlvo_ctx *ctx, const ktlvo_roi *rois, int n_roi, RoiDev *out, size_t &total, int &maxw,
                      int &maxh) {
    total = 0; maxw = 0; maxh = 0;
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

static bool rois_fit_lds(const RoiDev *rois, int n_roi, double min_dist) {
    for (int i = 0; i < n_roi; i++) {
        const RoiDev &r = rois[i];
        if (r.w * r.h > SEL_MAX) return false;
        if (min_dist >= 1) {
            const int cell = (int)lrint(min_dist);
            if (((r.w + cell - 1) / cell) * ((r.h + cell - 1) / cell) > CELL_MAX) return false;
        }
    }
    return true;
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
static bool rois_fit_table(const RoiDev *rois, int n_roi) {
    bool fits = n_roi <= RoiTable::MAX;
    for (int i = 0; i < n_roi && fits; i++) fits = rois[i].x <= 65535 && rois[i].y <= 65535;
    return fits;
}

static int corner_scratch_setup(ktlvo_ctx *ctx, const RoiDev *rois, int n_roi, const CornerScratch &s) {
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

// minw, minh: the smallest ROI of the call; ROIs of at least 8 x 8 pixels that do not go through corner_select's LDS
// (whole frames) take the register kernel, corner_candidates_wave, unless the context says otherwise.
static int run_candidates(ktlvo_ctx *ctx, int slot, int n_roi, int maxw, int maxh, int use_harris, double k,
                          const CornerScratch &s, float *d_resp_out, bool want_candidates, const int *d_roi_slots = nullptr,
                          bool large = false, int minw = 0, int minh = 0) {
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
        if (per_band * g.bands < 2048 && ctx->candidates_low_bands)
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

// Detection for ROIs that fit corner_select's LDS (the reference's sub-regions): everything is enqueued on the
// context's stream, nothing is allocated or synchronised; outputs are device arrays [n_roi][cap].
static int corner_detect_small_dev(ktlvo_ctx *ctx, int slot, const RoiDev *rois, int n_roi, int maxw, int maxh,
                                   const ktlvo_gftt_params *p, const CornerScratch &s, float *d_xy, float *d_resp,
                                   int *d_counts, int cap) {
    int rc = corner_scratch_setup(ctx, rois, n_roi, s);
    if (rc != KTLVO_OK) return rc;
    rc = run_candidates(ctx, slot, n_roi, maxw, maxh, p->use_harris, p->harris_k, s, nullptr, true);
    if (rc != KTLVO_OK) return rc;
    SelArgs a;
    a.rois = s.d_rois; a.cand = s.d_cand; a.cand_cnt = s.d_cnt; a.roi_max = s.d_max; a.quality = p->quality;
    a.max_corners = p->max_corners; a.cap = cap; a.min_dist = p->min_dist;
    a.xy = d_xy; a.resp = d_resp; a.counts = d_counts; a.overflow = s.d_overflow; a.relax = ctx->greedy_rounds;
    hipLaunchKernelGGL(corner_select, dim3(n_roi), dim3(256), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

static int ensure_aux(ktlvo_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->aux_bytes) return KTLVO_OK;
    if (ctx->d_aux) {
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        KTLVO_HIP(ctx, hipFree(ctx->d_aux));
        ctx->d_aux = nullptr;
        ctx->aux_bytes = 0;
    }
    bytes = (bytes + 4095) & ~(size_t)4095;
    KTLVO_HIP(ctx, hipMalloc(&ctx->d_aux, bytes));
    ctx->aux_bytes = bytes;
    return KTLVO_OK;
}

// n_lists lists of np2 keys each (np2 a power of two), back to back, each sorted descending.
// d_n_used[list] = number of keys at the front of the list, the rest being zero.
static int sort_keys(ktlvo_ctx *ctx, unsigned long long *d_keys, size_t np2, int n_lists, const int *d_n_used) {
    const unsigned L = (unsigned)n_lists;
    hipLaunchKernelGGL(bitonic_local_sort, dim3((unsigned)((np2 + BL - 1) / BL), L), dim3(1024), 0, ctx->stream, d_keys, (int)np2, np2,
                       d_n_used);
    for (size_t k = 2 * (size_t)BL; k <= np2; k <<= 1) {
        for (size_t j = k >> 1; j >= (size_t)BL; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_step, dim3((unsigned)(np2 / 512), L), dim3(256), 0, ctx->stream, d_keys, (unsigned)j,
                               (unsigned)k, np2, d_n_used);
        hipLaunchKernelGGL(bitonic_local_merge, dim3((unsigned)(np2 / BL), L), dim3(1024), 0, ctx->stream, d_keys, (unsigned)k, np2,
                           d_n_used);
    }
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// Kernels that ask for more than 64 KiB of dynamic LDS have to be told so once per device.
// (bytes: the most the kernel will be launched with; together with its static arrays it must fit the 160 KiB of a compute unit)
template <class K>
static int allow_big_lds(ktlvo_ctx *ctx, K kernel, size_t bytes, bool &done) {
    if (done) return KTLVO_OK;
    KTLVO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done = true;
    return KTLVO_OK;
}
// Filter and sort of n_lists candidate lists (cand_stride keys apart; counts and maxima one element apart) with more
// keys than one workgroup sorts, by response ranges (corner_sort_hist, corner_sort_plan, corner_sort_scatter,
// corner_sort_frames<true>).  n_sup = the largest candidate count, d_work = n_lists plans, a flag and n_lists
// histograms (bucket_work_bytes), h_flag = one pinned int.  *sorted says whether out[] (lists out_stride keys apart) and out_cnt[]
// are now what cand_filter and sort_keys would have left at the lists' fronts; if not (a list the plan turned down),
// the caller takes those two.  With d_redo the call does not wait: a plan turned down raises *d_redo (and leaves its
// list empty, so whatever is queued behind does nothing with it), and the caller finds out when it next waits; n_sup
// may then be a bound (the ROI's area) and max_total says how many keys above the threshold out[] has room for.
static size_t bucket_plans_bytes(int n_lists) { return (sizeof(SortPlan) * (size_t)n_lists + 255) & ~(size_t)255; }
static size_t bucket_zero_bytes(int n_lists) { return 256 + (size_t)n_lists * PLAN_BINS * 4; }
static size_t bucket_work_bytes(int n_lists) { return bucket_plans_bytes(n_lists) + bucket_zero_bytes(n_lists); }
static int bucket_sort(ktlvo_ctx *ctx, const unsigned long long *d_cand, const int *d_cnt, const unsigned *d_max, double quality,
                       unsigned long long *d_out, int *d_out_cnt, size_t cand_stride, size_t out_stride, int n_lists, int n_sup,
                       char *d_work, int *h_flag, bool *sorted, int *d_redo = nullptr, int max_total = 0x7fffffff,
                       char *d_zero = nullptr, CellCount cells = CellCount{0, 0, nullptr, 0}) {
    SortPlan *d_plans = (SortPlan *)d_work;
    // (d_zero: memory of bucket_zero_bytes() the caller has cleared for the flag and the histograms)
    int *d_own = d_zero ? (int *)d_zero : (int *)(d_work + bucket_plans_bytes(n_lists)), *d_hist = d_own + 64;
    int *d_flag = d_redo ? d_redo : d_own;
    int rc = allow_big_lds(ctx, corner_sort_frames<true>, (size_t)SORT_LDS_KEYS * 8, ctx->big_lds_sort_buckets);
    if (rc != KTLVO_OK) return rc;
    if (!d_zero) KTLVO_HIP(ctx, hipMemsetAsync(d_own, 0, bucket_zero_bytes(n_lists), ctx->stream)); // flag and histograms
    // (small slices where big ones would give fewer workgroups than the chip has compute units)
    const bool small = ((size_t)n_sup + PLAN_SLICE_BIG - 1) / PLAN_SLICE_BIG * (size_t)n_lists < 256;
    const int slice = small ? PLAN_SLICE_SMALL : PLAN_SLICE_BIG;
    const dim3 slices((unsigned)((n_sup + slice - 1) / slice), (unsigned)n_lists);
    if (small)
        hipLaunchKernelGGL(corner_sort_hist<PLAN_SLICE_SMALL>, slices, dim3(1024), 0, ctx->stream, d_cand, d_cnt, d_max, quality,
                           cand_stride, d_hist);
    else
        hipLaunchKernelGGL(corner_sort_hist<PLAN_SLICE_BIG>, slices, dim3(1024), 0, ctx->stream, d_cand, d_cnt, d_max, quality,
                           cand_stride, d_hist);
    hipLaunchKernelGGL(corner_sort_plan, dim3((unsigned)n_lists), dim3(1024), 0, ctx->stream, (const int *)d_hist, d_max, quality,
                       d_plans, d_out_cnt, d_flag, max_total, std::max(4, std::min(PLAN_BUCKETS, 256 / n_lists)));
    if (small)
        hipLaunchKernelGGL(corner_sort_scatter<PLAN_SLICE_SMALL>, slices, dim3(1024), 0, ctx->stream, d_cand, d_cnt, d_max, quality,
                           cand_stride, d_plans, d_out, out_stride);
    else
        hipLaunchKernelGGL(corner_sort_scatter<PLAN_SLICE_BIG>, slices, dim3(1024), 0, ctx->stream, d_cand, d_cnt, d_max, quality,
                           cand_stride, d_plans, d_out, out_stride);
    const unsigned nb = (unsigned)std::min(SORT_MAXB, std::min(n_sup, max_total) / PLAN_T_MIN + 1);
    hipLaunchKernelGGL(corner_sort_frames<true>, dim3((unsigned)n_lists, nb), dim3(1024), (size_t)SORT_LDS_KEYS * 8, ctx->stream,
                       d_cand, d_cnt, d_max, quality, d_out, d_out_cnt, cand_stride, out_stride, d_flag, (const SortPlan *)d_plans, cells);
    KTLVO_HIP(ctx, hipGetLastError());
    if (d_redo) { // the caller looks at the flag when it next waits for the stream
        *sorted = true;
        return KTLVO_OK;
    }
    KTLVO_HIP(ctx, hipMemcpyAsync(h_flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *sorted = *h_flag == 0;
    return KTLVO_OK;
}

// The greedy pass over n_rois sorted key lists, the grid in LDS when it fits (the cells in global memory are then not
// used and need not be cleared).
static bool select_uses_lds_grid(const SelLargeArgs &a) { return grid_fits_lds(a.cell, a.gw, a.gh); }
static int launch_select_large(ktlvo_ctx *ctx, const SelLargeArgs &a, int n_rois) {
    if (select_uses_lds_grid(a)) {
        int rc = allow_big_lds(ctx, corner_select_large<true>, SEL_LDS_GRID_MAX, ctx->big_lds_select);
        if (rc != KTLVO_OK) return rc;
        hipLaunchKernelGGL(corner_select_large<true>, dim3((unsigned)n_rois), dim3(CH), (size_t)a.gw * a.gh * SEL_LDS_PER_CELL,
                           ctx->stream, a);
    } else {
        hipLaunchKernelGGL(corner_select_large<false>, dim3((unsigned)n_rois), dim3(CH), 0, ctx->stream, a);
    }
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// The greedy pass in rounds over n_frames sorted key lists (np2 keys apart, n_sup bounds every list's length): work
// memory `work` of rounds_work_bytes(), h_flags = n_frames pinned ints.  After every launch of the relaxation the host
// looks whether any frame still has an undecided candidate; one launch is the rule.  Returns KTLVO_ERR_STATE for the
// caller to take the sequential pass if 32 launches have not settled a frame.  With queued > 0 that many launches are
// queued and the output written without a look at the flags (n_sup and np2 may then be bounds instead of counts the
// host has fetched); the last of them raises *d_unsettled if it leaves a candidate undecided, which the caller reads
// when it next waits for the stream, and then does the frames again another way.
constexpr int LF_MAX_ROUNDS = 32;
constexpr int LF_SCAN_FEW = 4; // stretches of LF_SCAN cells up to which one workgroup per frame scans them all (one launch instead of two)
// The sequential pass takes a workgroup per frame and as long for 200 frames as for one; the relaxation uses the whole
// chip and takes as long as its frames add up to.  Measured on 1920 x 1080 and 3840 x 2160 frames the two meet at
// about 50 frames per call.
constexpr int LF_FRAMES_MAX = 48;
constexpr int LF_FRAMES_QUEUED_MAX = 16; // the same for frames whose grid fits LDS (a faster sequential pass): measured on 1241 x 376 frames
constexpr int LF_QUEUED = 2;             // launches of the relaxation queued without looking at the flags
constexpr int FEW_KEYS_MAX = 2 * SORT_CAP; // keys above the threshold per frame that the routes queued on bounds make room for
constexpr int FEW_FRAMES_SORT_MAX = 8;     // frames per call up to which the sort by ranges beats a workgroup per frame (1241 x 376: 4 frames 0.27 against 0.31 ms, 16 frames 0.52 against 0.48)
static size_t lf_cells_stride(size_t cells) { return (cells + 1 + 3) & ~(size_t)3; }
static bool rounds_possible(size_t cells) { return (lf_cells_stride(cells) + LF_SCAN - 1) / LF_SCAN <= (size_t)LF_SCAN_BLOCKS; }
static size_t rounds_work_bytes(size_t np2, size_t cells, int n_frames) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t F = (size_t)n_frames, cs = lf_cells_stride(cells);
    return up(np2 * F) + up(np2 * 8 * F) + 2 * up(cs * 4 * F) + up(cs * 4 * F) + up((size_t)LF_SCAN_BLOCKS * 4 * F) +
           up((np2 / 1024 + 1) * 4 * F) + up((size_t)LF_MAX_ROUNDS * F * 4);
}
// (d_cells_zero: memory of rounds_zero_bytes() the caller has cleared -- together with other things, in one go -- for
// the cells' counts and handed-out places; without it they lie in `work` and are cleared here)
// (the cleared memory also holds the candidates' states; with `counted` the cells' counts are already in it -- the sort
// by ranges wrote them, rounds_cell_count() -- and the counting pass is left out)
static size_t rounds_zero_bytes(size_t cells, int n_frames, size_t np2) {
    return 2 * ((lf_cells_stride(cells) * 4 * (size_t)n_frames + 255) & ~(size_t)255) + ((np2 * (size_t)n_frames + 255) & ~(size_t)255);
}
static CellCount rounds_cell_count(char *d_cells_zero, int cell, int gw, int gh) {
    return CellCount{cell, gw, (int *)d_cells_zero, lf_cells_stride((size_t)gw * gh)};
}
static int run_rounds(ktlvo_ctx *ctx, LfArgs a, int n_sup, size_t np2, char *work, int *h_flags, int queued = 0,
                      int *d_unsettled = nullptr, char *d_cells_zero = nullptr, bool counted = false) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t F = (size_t)a.n_frames, cs = lf_cells_stride((size_t)a.gw * a.gh);
    a.keys_stride = np2; a.cells_stride = cs; a.blk_stride = np2 / 1024 + 1;
    a.scan_blocks = (int)((cs + LF_SCAN - 1) / LF_SCAN);
    char *w = work;
    a.state = (unsigned char *)w; w += up(np2 * F);
    a.items = (uint2 *)w; w += up(np2 * 8 * F);
    a.cell_cnt = (int *)w; w += up(cs * 4 * F);
    a.cell_fill = (int *)w; w += up(cs * 4 * F);
    a.start = (int *)w; w += up(cs * 4 * F);
    a.blk_off = (int *)w; w += up((size_t)LF_SCAN_BLOCKS * 4 * F);
    a.blk_cnt = (int *)w; w += up(a.blk_stride * 4 * F);
    a.flags = (int *)w;
    if (d_cells_zero) {
        a.cell_cnt = (int *)d_cells_zero; a.cell_fill = (int *)(d_cells_zero + up(cs * 4 * F));
        a.state = (unsigned char *)(d_cells_zero + 2 * up(cs * 4 * F));
    } else {
        KTLVO_HIP(ctx, hipMemsetAsync(a.cell_cnt, 0, 2 * up(cs * 4 * F), ctx->stream)); // counts and handed-out places
    }
    a.last_round = queued - 1; a.unsettled = queued ? d_unsettled : a.flags;
    a.tries = ctx->relax_tries;
    if (!queued) KTLVO_HIP(ctx, hipMemsetAsync(a.flags, 0, (size_t)LF_MAX_ROUNDS * F * 4, ctx->stream));
    const dim3 g256((unsigned)((n_sup + 255) / 256), (unsigned)F), g1024((unsigned)((n_sup + 1023) / 1024), (unsigned)F);
    if (!counted) hipLaunchKernelGGL(lf_count, g256, dim3(256), 0, ctx->stream, a);
    if (a.scan_blocks <= LF_SCAN_FEW) {
        hipLaunchKernelGGL(lf_scan_few, dim3((unsigned)F), dim3(1024), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(lf_scan_cells, dim3((unsigned)a.scan_blocks, (unsigned)F), dim3(1024), 0, ctx->stream, a);
        hipLaunchKernelGGL(lf_scan_blocks, dim3((unsigned)F), dim3(1024), 0, ctx->stream, a);
    }
    hipLaunchKernelGGL(lf_fill, g256, dim3(256), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    bool settled = queued > 0;
    for (int r = 0; r < queued; r++) hipLaunchKernelGGL(lf_round, g256, dim3(256), 0, ctx->stream, a, r);
    for (int r = 0; r < LF_MAX_ROUNDS && !settled; r++) {
        hipLaunchKernelGGL(lf_round, g256, dim3(256), 0, ctx->stream, a, r);
        KTLVO_HIP(ctx, hipGetLastError());
        KTLVO_HIP(ctx, hipMemcpyAsync(h_flags, a.flags + (size_t)r * F, 4 * F, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        settled = true;
        for (size_t f = 0; f < F; f++) settled = settled && h_flags[f] == 0;
    }
    if (!settled) return KTLVO_ERR_STATE;
    hipLaunchKernelGGL(lf_block_counts, g1024, dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(lf_write, g1024, dim3(1024), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// Sort + greedy pass of one large ROI on the device.  n_sup = the ROI's candidate count (already on the host).
// blind: nothing waits for the device -- the sort by response ranges and two launches of the relaxation are queued, and
// a plan turned down or a candidate left undecided raises the scratch's second flag (s.d_overflow[1]); the caller looks
// at it when the results are home and then calls again with blind = false.
static int select_large(ktlvo_ctx *ctx, const RoiDev &R, int roi, int n_sup, const ktlvo_gftt_params *p,
                        const CornerScratch &s, float *d_xy, float *d_resp, int *d_count, int cap, int *h_flags, bool blind) {
    if (n_sup == 0) {
        KTLVO_HIP(ctx, hipMemsetAsync(d_count, 0, 4, ctx->stream));
        return KTLVO_OK;
    }
    size_t np2 = 1;
    while (np2 < (size_t)n_sup) np2 <<= 1;
    const bool dist = p->min_dist >= 1;
    const int cell = dist ? (int)lrint(p->min_dist) : 1;
    const int gw = dist ? (R.w + cell - 1) / cell : 0, gh = dist ? (R.h + cell - 1) / cell : 0;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_keys = 0, o_n = o_keys + up(np2 * 8), o_cc = o_n + 256, o_cp = o_cc + up((size_t)gw * gh * 4);
    const size_t o_rounds = o_cp + up((size_t)gw * gh * 16);
    const bool rounds = dist && ctx->greedy_rounds && !grid_fits_lds(cell, gw, gh) && rounds_possible((size_t)gw * gh);
    const size_t o_bucket = o_rounds + (rounds ? rounds_work_bytes(np2, (size_t)gw * gh, 1) : 0);
    const size_t o_zero = o_bucket + bucket_work_bytes(1); // for the blind route, cleared in one go: sort flag, histogram, cell counters
    const size_t total = o_zero + bucket_zero_bytes(1) + (rounds ? rounds_zero_bytes((size_t)gw * gh, 1, np2) : 0);
    int rc = ensure_aux(ctx, total);
    if (rc != KTLVO_OK) return rc;
    char *b = (char *)ctx->d_aux;
    unsigned long long *d_keys = (unsigned long long *)(b + o_keys);
    int *d_n = (int *)(b + o_n);
    const bool global_grid = dist && !grid_fits_lds(cell, gw, gh);
    bool sorted = false;
    if (ctx->frames_lds_sort) { // by response ranges in LDS
        if (blind) KTLVO_HIP(ctx, hipMemsetAsync(b + o_zero, 0, total - o_zero, ctx->stream));
        rc = bucket_sort(ctx, s.d_cand + R.off, s.d_cnt + roi, s.d_max + roi, p->quality, d_keys, d_n, 0, 0, 1, n_sup, b + o_bucket,
                         h_flags, &sorted, blind ? s.d_overflow + 1 : nullptr, 0x7fffffff, blind ? b + o_zero : nullptr,
                         blind && rounds ? rounds_cell_count(b + o_zero + bucket_zero_bytes(1), cell, gw, gh) : CellCount{0, 0, nullptr, 0});
        if (rc != KTLVO_OK) return rc;
    }
    if (!sorted) {
        // keys and counter cleared in one go (the sort runs over all np2 keys; zeros sort last); a grid in global memory
        // is cleared when the sequential pass is taken (its empty places hold a far-away point)
        KTLVO_HIP(ctx, hipMemsetAsync(b, 0, o_cc, ctx->stream));
        const int fblocks = (int)std::min<size_t>(((size_t)n_sup + 255) / 256, 1024);
        hipLaunchKernelGGL(cand_filter, dim3(fblocks), dim3(256), 0, ctx->stream, s.d_cand + R.off, s.d_cnt + roi, s.d_max + roi,
                           p->quality, d_keys, d_n, (size_t)0, (size_t)0);
        KTLVO_HIP(ctx, hipGetLastError());
        // How many passed is only known on the device; the sort is sized for the candidate count, which bounds it.
        rc = sort_keys(ctx, d_keys, np2, 1, d_n);
        if (rc != KTLVO_OK) return rc;
    }
    const int limit = p->max_corners > 0 ? std::min(p->max_corners, cap) : cap;
    if (!dist) {
        hipLaunchKernelGGL(corner_take_first, dim3((limit + 255) / 256), dim3(256), 0, ctx->stream, d_keys, d_n, R.w, limit, d_xy,
                           d_resp, d_count, (size_t)0, (size_t)0);
    } else {
        if (rounds) { // a frame whose grid does not fit LDS: the relaxation over all candidates at once
            LfArgs l;
            l.keys = d_keys; l.n_ptr = d_n; l.cell = cell; l.gw = gw; l.gh = gh; l.min_dist = p->min_dist; l.limit = limit;
            l.xy = d_xy; l.resp = d_resp; l.count = d_count; l.out_stride = 0; l.n_frames = 1;
            if (blind && sorted)
                return run_rounds(ctx, l, n_sup, np2, b + o_rounds, nullptr, LF_QUEUED, s.d_overflow + 1, b + o_zero + bucket_zero_bytes(1), true);
            rc = run_rounds(ctx, l, n_sup, np2, b + o_rounds, h_flags);
            if (rc != KTLVO_ERR_STATE) return rc;
        }
        if (global_grid) {
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_cc, 0, (size_t)gw * gh * 4, ctx->stream));
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_cp, 0xFF, (size_t)gw * gh * 16, ctx->stream));
        }
        SelLargeArgs a;
        a.keys = d_keys; a.n_ptr = d_n; a.w = R.w; a.h = R.h; a.min_dist = p->min_dist; a.cell = cell; a.gw = gw; a.gh = gh;
        a.cell_cnt = (int *)(b + o_cc); a.cell_pts = (uint4 *)(b + o_cp); a.limit = limit;
        a.xy = d_xy; a.resp = d_resp; a.count = d_count; a.overflow = s.d_overflow;
        a.keys_stride = a.cells_stride = a.out_stride = 0;
        rc = launch_select_large(ctx, a, 1);
        if (rc != KTLVO_OK) return rc;
    }
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// Whether the LDS sort is worth trying on ROIs of this area.  It is tried without knowing the key counts, and a ROI with
// more keys above its threshold than it holds sends the whole call through the global sort afterwards, so it is only
// tried where that is unlikely: textured frames have a local maximum per 25 to 30 pixels, of which the threshold keeps
// two thirds (a 1241 x 376 frame: 12,400 keys; a 1920 x 1080 frame: four times as many).
static bool lds_sort_likely(size_t area) { return area <= ((size_t)512 << 10); }

// The same for a list of large ROIs without a word from the device in between: every ROI's keys are filtered and sorted
// in LDS by response ranges (bucket_sort with bounds for counts), then its greedy pass runs; the ROIs take turns on one
// work area, in stream order.  *d_many_out is a device flag that is raised when some ROI had more than FEW_KEYS_MAX keys
// above its threshold or a plan turned down, or candidates the queued launches of the relaxation left undecided; the caller looks at it once
// everything has run and then takes select_large for the whole list.
static int select_large_lds(ktlvo_ctx *ctx, const RoiDev *rois, int n_roi, const ktlvo_gftt_params *p, const CornerScratch &s,
                            float *d_xy, float *d_resp, int *d_counts, int cap, int **d_many_out) {
    const bool dist = p->min_dist >= 1;
    const int cell = dist ? (int)lrint(p->min_dist) : 1;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t cells_max = 0; // of the ROIs whose grid stays in global memory
    for (int r = 0; r < n_roi && dist; r++) {
        const int gw = (rois[r].w + cell - 1) / cell, gh = (rois[r].h + cell - 1) / cell;
        if (!grid_fits_lds(cell, gw, gh)) cells_max = std::max(cells_max, (size_t)gw * gh);
    }
    const size_t o_n = up((size_t)FEW_KEYS_MAX * 8), o_flag = o_n + 256, o_cc = o_flag + 256, o_cp = o_cc + up(cells_max * 4);
    const size_t o_rounds = o_cp + up(cells_max * 16);
    size_t rounds_bytes = 0; // the greedy pass as a relaxation, two launches queued blind, where the grid allows it
    for (int r = 0; r < n_roi && dist && ctx->greedy_rounds; r++) {
        const size_t cells = (size_t)((rois[r].w + cell - 1) / cell) * ((rois[r].h + cell - 1) / cell);
        if (rounds_possible(cells)) rounds_bytes = std::max(rounds_bytes, rounds_work_bytes(FEW_KEYS_MAX, cells, 1));
    }
    const size_t o_bucket = o_rounds + rounds_bytes;
    // what has to start at zero for a ROI (the sort's flag and histogram, the relaxation's cell counters) lies together
    // and is cleared in one go
    size_t zero_max = 0;
    for (int r = 0; r < n_roi && dist && ctx->greedy_rounds; r++)
        zero_max = std::max(zero_max, rounds_zero_bytes((size_t)((rois[r].w + cell - 1) / cell) * ((rois[r].h + cell - 1) / cell), 1, FEW_KEYS_MAX));
    const size_t o_zero = o_bucket + bucket_work_bytes(1);
    int rc = ensure_aux(ctx, o_zero + bucket_zero_bytes(1) + zero_max);
    if (rc != KTLVO_OK) return rc;
    char *b = (char *)ctx->d_aux;
    unsigned long long *d_keys = (unsigned long long *)b;
    int *d_n = (int *)(b + o_n), *d_many = (int *)(b + o_flag);
    *d_many_out = d_many;
    KTLVO_HIP(ctx, hipMemsetAsync(d_many, 0, 4, ctx->stream));
    const int limit = p->max_corners > 0 ? std::min(p->max_corners, cap) : cap;
    for (int r = 0; r < n_roi; r++) {
        const RoiDev &R = rois[r];
        // one ROI at a time leaves the chip idle: its keys are sorted by response ranges, short sorts side by side
        bool sorted;
        const size_t cells_r = dist ? (size_t)((R.w + cell - 1) / cell) * ((R.h + cell - 1) / cell) : 0;
        const bool relax_r = dist && ctx->greedy_rounds && rounds_possible(cells_r);
        KTLVO_HIP(ctx, hipMemsetAsync(b + o_zero, 0, bucket_zero_bytes(1) + (relax_r ? rounds_zero_bytes(cells_r, 1, FEW_KEYS_MAX) : 0), ctx->stream));
        rc = bucket_sort(ctx, s.d_cand + R.off, s.d_cnt + r, s.d_max + r, p->quality, d_keys, d_n, 0, 0, 1, R.w * R.h, b + o_bucket,
                         nullptr, &sorted, d_many, FEW_KEYS_MAX, b + o_zero,
                         relax_r ? rounds_cell_count(b + o_zero + bucket_zero_bytes(1), cell, (R.w + cell - 1) / cell, (R.h + cell - 1) / cell)
                                 : CellCount{0, 0, nullptr, 0});
        if (rc != KTLVO_OK) return rc;
        float *xy = d_xy + (size_t)r * cap * 2, *resp = d_resp + (size_t)r * cap;
        if (!dist) {
            hipLaunchKernelGGL(corner_take_first, dim3((limit + 255) / 256), dim3(256), 0, ctx->stream, d_keys, d_n, R.w, limit, xy,
                               resp, d_counts + r, (size_t)0, (size_t)0);
            KTLVO_HIP(ctx, hipGetLastError());
            continue;
        }
        SelLargeArgs a;
        a.keys = d_keys; a.n_ptr = d_n; a.w = R.w; a.h = R.h; a.min_dist = p->min_dist; a.cell = cell;
        a.gw = (R.w + cell - 1) / cell; a.gh = (R.h + cell - 1) / cell;
        a.cell_cnt = (int *)(b + o_cc); a.cell_pts = (uint4 *)(b + o_cp); a.limit = limit;
        a.xy = xy; a.resp = resp; a.count = d_counts + r; a.overflow = s.d_overflow;
        a.keys_stride = a.cells_stride = a.out_stride = 0;
        if (relax_r) {
            LfArgs l;
            l.keys = d_keys; l.n_ptr = d_n; l.cell = cell; l.gw = a.gw; l.gh = a.gh; l.min_dist = p->min_dist; l.limit = limit;
            l.xy = xy; l.resp = resp; l.count = d_counts + r; l.out_stride = 0; l.n_frames = 1;
            rc = run_rounds(ctx, l, FEW_KEYS_MAX, FEW_KEYS_MAX, b + o_rounds, nullptr, LF_QUEUED, d_many, b + o_zero + bucket_zero_bytes(1), true);
            if (rc != KTLVO_OK) return rc;
            continue;
        }
        if (!select_uses_lds_grid(a)) {
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_cc, 0, (size_t)a.gw * a.gh * 4, ctx->stream));
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_cp, 0xFF, (size_t)a.gw * a.gh * 16, ctx->stream));
        }
        rc = launch_select_large(ctx, a, 1);
        if (rc != KTLVO_OK) return rc;
    }
    return KTLVO_OK;
}

// ---- a batch of frames ------------------------------------------------------------------------------------
// goodFeaturesToTrack on the same ROI of n_frames slots with the launches of ONE detection: every kernel of the
// whole-frame path takes a frame index from its grid.  A single frame leaves the GPU mostly idle (its greedy pass is
// one workgroup, its sort a few dozen); a batch fills it, and the cost per frame drops by two orders of magnitude.
// Per frame the result is ktlvo_detect_gftt's for that slot and ROI (same candidates, same unique keys in the same
// order, same greedy pass).  Outputs are device arrays [n_frames][cap]; the call returns once they are complete.
// before_candidates, if given, is called right before the candidates launch is queued on the context's stream, once the
// work area is set up (the detect-and-build entry point queues the pyramid build on the side stream there).
static int detect_frames_dev(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi &roi, const ktlvo_gftt_params *p,
                             float *d_xy, float *d_resp, int *d_counts, int cap, const std::function<int()> &before_candidates = nullptr) {
    const size_t area = (size_t)roi.w * roi.h;
    if (area * (size_t)n_frames > 0x7fffffff) return KTLVO_ERR_CAPACITY; // candidate offsets are 32-bit
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // work area: the detector's scratch for n_frames ROIs, then the frames' slot numbers, then their candidate counts' pinned mirror
    const size_t o_slots = up(corner_scratch_bytes(n_frames, area * n_frames));
    int rc = ensure_misc(ctx, o_slots + up(4 * (size_t)n_frames));
    if (rc != KTLVO_OK) return rc;
    rc = ensure_pin(ctx, 12 * (size_t)n_frames + 48);
    if (rc != KTLVO_OK) return rc;
    CornerScratch s;
    corner_scratch_carve(ctx->d_misc, n_frames, area * n_frames, s);
    int *d_roi_slots = (int *)((char *)ctx->d_misc + o_slots);
    int *h_cnt = (int *)ctx->h_pin + n_frames, *h_ovf = h_cnt + n_frames, *h_many = h_ovf + 1;
    {
        FrameSlots t;
        for (int f = 0; f < n_frames; f++) t.s[f] = slots[f];
        hipLaunchKernelGGL(corner_setup_frames, dim3((n_frames + 255) / 256), dim3(256), 0, ctx->stream, t, n_frames,
                           RoiDev{roi.x, roi.y, roi.w, roi.h, (int)area}, s.d_rois, d_roi_slots, s.d_max, s.d_cnt, s.d_overflow);
        KTLVO_HIP(ctx, hipGetLastError());
    }
    if (before_candidates) {
        rc = before_candidates();
        if (rc != KTLVO_OK) return rc;
    }
    rc = run_candidates(ctx, -1, n_frames, roi.w, roi.h, p->use_harris, p->harris_k, s, nullptr, true, d_roi_slots, true, roi.w, roi.h);
    if (rc != KTLVO_OK) return rc;
    const bool dist = p->min_dist >= 1;
    const int cell = dist ? (int)lrint(p->min_dist) : 1;
    const int gw = dist ? (roi.w + cell - 1) / cell : 0, gh = dist ? (roi.h + cell - 1) / cell : 0;
    const size_t cells = (size_t)gw * gh;
    const bool global_grid = dist && !grid_fits_lds(cell, gw, gh);
    const int limit = p->max_corners > 0 ? std::min(p->max_corners, cap) : cap;
    SelLargeArgs a;
    a.w = roi.w; a.h = roi.h; a.min_dist = p->min_dist; a.cell = cell; a.gw = gw; a.gh = gh; a.limit = limit;
    a.xy = d_xy; a.resp = d_resp; a.count = d_counts; a.overflow = s.d_overflow;
    a.cells_stride = cells; a.out_stride = (size_t)cap;

    if (ctx->frames_lds_sort && lds_sort_likely(area)) {
        // Every frame filtered and sorted by one workgroup in LDS, then the greedy pass: nothing is sized on the host, so
        // the whole detection is queued in one go and the host waits once, at the end, to learn whether a frame had more
        // keys than the LDS sort holds; then, and only then, the global path below does the batch again.
        // few frames leave the chip idle: their keys are sorted by response ranges (short sorts side by side) and the
        // greedy pass is the relaxation; many frames get a workgroup each for the sort and for the walk
        const bool few = n_frames <= FEW_FRAMES_SORT_MAX;
        const size_t kstride = few ? (size_t)FEW_KEYS_MAX : (size_t)SORT_CAP;
        const size_t o_n = up(kstride * 8 * n_frames), o_flag = o_n + up(4 * (size_t)n_frames), o_cc = o_flag + 256;
        const size_t o_cp = o_cc + (global_grid ? up(cells * 4 * n_frames) : 0);
        const size_t o_rounds = o_cp + (global_grid ? up(cells * 16 * n_frames) : 0);
        // few frames: the greedy pass as a relaxation, two launches queued blind (a frame they do not settle raises
        // the same flag as one with too many keys)
        const bool rounds = dist && ctx->greedy_rounds && rounds_possible(cells) && n_frames <= LF_FRAMES_QUEUED_MAX;
        const size_t o_bucket = o_rounds + (rounds ? rounds_work_bytes(kstride, cells, n_frames) : 0);
        const size_t o_zero = o_bucket + (few ? bucket_work_bytes(n_frames) : 0); // cleared in one go: sort flag, histograms, cell counters
        const size_t total = o_zero + (few ? bucket_zero_bytes(n_frames) + (rounds ? rounds_zero_bytes(cells, n_frames, kstride) : 0) : 0);
        rc = ensure_aux(ctx, total);
        if (rc != KTLVO_OK) return rc;
        rc = allow_big_lds(ctx, corner_sort_frames<false>, (size_t)SORT_LDS_KEYS * 8, ctx->big_lds_sort);
        if (rc != KTLVO_OK) return rc;
        char *b = (char *)ctx->d_aux;
        unsigned long long *d_keys = (unsigned long long *)b;
        int *d_n = (int *)(b + o_n), *d_many = (int *)(b + o_flag);
        KTLVO_HIP(ctx, hipMemsetAsync(b + o_flag, 0, o_cp - o_flag, ctx->stream)); // the flag and, if used, the cells' counts
        if (global_grid) KTLVO_HIP(ctx, hipMemsetAsync(b + o_cp, 0xFF, cells * 16 * n_frames, ctx->stream));
        if (few) {
            bool sorted;
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_zero, 0, total - o_zero, ctx->stream));
            rc = bucket_sort(ctx, s.d_cand, s.d_cnt, s.d_max, p->quality, d_keys, d_n, area, kstride, n_frames, (int)area, b + o_bucket,
                             nullptr, &sorted, d_many, FEW_KEYS_MAX, b + o_zero,
                             rounds ? rounds_cell_count(b + o_zero + bucket_zero_bytes(n_frames), cell, gw, gh) : CellCount{0, 0, nullptr, 0});
            if (rc != KTLVO_OK) return rc;
        } else {
            hipLaunchKernelGGL(corner_sort_frames<false>, dim3((unsigned)n_frames), dim3(1024), (size_t)SORT_LDS_KEYS * 8, ctx->stream,
                               s.d_cand, s.d_cnt, s.d_max, p->quality, d_keys, d_n, area, kstride, d_many, (const SortPlan *)nullptr, CellCount{0, 0, nullptr, 0});
            KTLVO_HIP(ctx, hipGetLastError());
        }
        if (!dist) {
            hipLaunchKernelGGL(corner_take_first, dim3((limit + 255) / 256, (unsigned)n_frames), dim3(256), 0, ctx->stream, d_keys,
                               d_n, roi.w, limit, d_xy, d_resp, d_counts, kstride, (size_t)cap);
            KTLVO_HIP(ctx, hipGetLastError());
        } else {
            a.keys = d_keys; a.n_ptr = d_n; a.keys_stride = kstride;
            a.cell_cnt = (int *)(b + o_cc); a.cell_pts = (uint4 *)(b + o_cp);
            if (rounds) {
                LfArgs l;
                l.keys = d_keys; l.n_ptr = d_n; l.cell = cell; l.gw = gw; l.gh = gh; l.min_dist = p->min_dist; l.limit = limit;
                l.xy = d_xy; l.resp = d_resp; l.count = d_counts; l.out_stride = (size_t)cap; l.n_frames = n_frames;
                rc = run_rounds(ctx, l, (int)kstride, kstride, b + o_rounds, nullptr, LF_QUEUED, d_many,
                                few ? b + o_zero + bucket_zero_bytes(n_frames) : nullptr, few);
            } else {
                rc = launch_select_large(ctx, a, n_frames);
            }
            if (rc != KTLVO_OK) return rc;
        }
        KTLVO_HIP(ctx, hipMemcpyAsync(h_ovf, s.d_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipMemcpyAsync(h_many, d_many, 4, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!*h_many) return KTLVO_OK;
    }

    // Global path: the sort runs in global memory over the batch's largest candidate count, which the host fetches first.
    KTLVO_HIP(ctx, hipMemcpyAsync(h_cnt, s.d_cnt, 4 * (size_t)n_frames, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n_sup = 0;
    for (int f = 0; f < n_frames; f++) n_sup = std::max(n_sup, h_cnt[f]);
    if (n_sup == 0) {
        KTLVO_HIP(ctx, hipMemsetAsync(d_counts, 0, 4 * (size_t)n_frames, ctx->stream));
        *h_ovf = 0;
        return KTLVO_OK;
    }
    size_t np2 = 1;
    while (np2 < (size_t)n_sup) np2 <<= 1;
    const size_t o_keys = 0, o_n = o_keys + up(np2 * 8 * n_frames), o_cc = o_n + up(4 * (size_t)n_frames);
    const size_t o_cp = o_cc + up(cells * 4 * n_frames), o_rounds = o_cp + up(cells * 16 * n_frames);
    const bool rounds = global_grid && ctx->greedy_rounds && rounds_possible(cells) && n_frames <= LF_FRAMES_MAX;
    const size_t o_bucket = o_rounds + (rounds ? rounds_work_bytes(np2, cells, n_frames) : 0);
    const size_t o_zero = o_bucket + bucket_work_bytes(n_frames); // for the blind route, cleared in one go
    const size_t total = o_zero + bucket_zero_bytes(n_frames) + (rounds ? rounds_zero_bytes(cells, n_frames, np2) : 0);
    rc = ensure_aux(ctx, total);
    if (rc != KTLVO_OK) return rc;
    char *b = (char *)ctx->d_aux;
    unsigned long long *d_keys = (unsigned long long *)(b + o_keys);
    int *d_n = (int *)(b + o_n);
    for (int attempt = 0; attempt < 2; attempt++) {
        // first without a word from the device (sort by response ranges, two launches of the relaxation; the scratch's
        // second flag says at the end whether that stands), then, if it does not, step by step
        const bool blind = attempt == 0 && ctx->frames_lds_sort && rounds;
        bool sorted = false;
        if (ctx->frames_lds_sort) { // by response ranges in LDS
            if (blind) KTLVO_HIP(ctx, hipMemsetAsync(b + o_zero, 0, total - o_zero, ctx->stream));
            rc = bucket_sort(ctx, s.d_cand, s.d_cnt, s.d_max, p->quality, d_keys, d_n, area, np2, n_frames, n_sup, b + o_bucket, h_many,
                             &sorted, blind ? s.d_overflow + 1 : nullptr, 0x7fffffff, blind ? b + o_zero : nullptr,
                             blind ? rounds_cell_count(b + o_zero + bucket_zero_bytes(n_frames), cell, gw, gh) : CellCount{0, 0, nullptr, 0});
            if (rc != KTLVO_OK) return rc;
        }
        if (!sorted) {
            KTLVO_HIP(ctx, hipMemsetAsync(b, 0, o_cc, ctx->stream)); // keys, counters
            const unsigned fblocks = (unsigned)std::min<size_t>(((size_t)n_sup + 255) / 256, 1024);
            hipLaunchKernelGGL(cand_filter, dim3(fblocks, (unsigned)n_frames), dim3(256), 0, ctx->stream, s.d_cand, s.d_cnt, s.d_max,
                               p->quality, d_keys, d_n, area, np2);
            KTLVO_HIP(ctx, hipGetLastError());
            rc = sort_keys(ctx, d_keys, np2, n_frames, d_n);
            if (rc != KTLVO_OK) return rc;
        }
        if (!dist) {
            hipLaunchKernelGGL(corner_take_first, dim3((limit + 255) / 256, (unsigned)n_frames), dim3(256), 0, ctx->stream, d_keys, d_n,
                               roi.w, limit, d_xy, d_resp, d_counts, np2, (size_t)cap);
            KTLVO_HIP(ctx, hipGetLastError());
        } else {
            rc = KTLVO_ERR_STATE;
            if (rounds) { // frames whose grid does not fit LDS: the relaxation over all candidates of all frames at once
                LfArgs l;
                l.keys = d_keys; l.n_ptr = d_n; l.cell = cell; l.gw = gw; l.gh = gh; l.min_dist = p->min_dist; l.limit = limit;
                l.xy = d_xy; l.resp = d_resp; l.count = d_counts; l.out_stride = (size_t)cap; l.n_frames = n_frames;
                rc = blind ? run_rounds(ctx, l, n_sup, np2, b + o_rounds, nullptr, LF_QUEUED, s.d_overflow + 1,
                                        b + o_zero + bucket_zero_bytes(n_frames), true)
                           : run_rounds(ctx, l, n_sup, np2, b + o_rounds, h_many + 1);
                if (rc != KTLVO_OK && rc != KTLVO_ERR_STATE) return rc;
            }
            if (rc == KTLVO_ERR_STATE) { // the sequential pass, a workgroup per frame
                if (global_grid) {
                    KTLVO_HIP(ctx, hipMemsetAsync(b + o_cc, 0, cells * 4 * n_frames, ctx->stream));
                    KTLVO_HIP(ctx, hipMemsetAsync(b + o_cp, 0xFF, cells * 16 * n_frames, ctx->stream));
                }
                a.keys = d_keys; a.n_ptr = d_n; a.keys_stride = np2;
                a.cell_cnt = (int *)(b + o_cc); a.cell_pts = (uint4 *)(b + o_cp);
                rc = launch_select_large(ctx, a, n_frames);
                if (rc != KTLVO_OK) return rc;
            }
        }
        if (!blind) break;
        KTLVO_HIP(ctx, hipMemcpyAsync(h_ovf, s.d_overflow, 8, hipMemcpyDeviceToHost, ctx->stream)); // (the second lands on h_many)
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!*h_many) return KTLVO_OK;
    }
    // the overflow flag lands in pinned memory behind the counts' mirror; every return of this function finds it there
    KTLVO_HIP(ctx, hipMemcpyAsync(h_ovf, s.d_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}

// After detect_frames_dev: a cell of the minimum-distance grid that ran out of places (more than four accepted corners
// in one cell, which the distance rule excludes for cells of the rule's own size) would have dropped a corner.
static int frames_overflow_check(ktlvo_ctx *ctx, int n_frames) {
    const int *h_ovf = (const int *)ctx->h_pin + 2 * n_frames;
    if (*h_ovf) { ctx->last_error = "corner selection: cell capacity exceeded"; return KTLVO_ERR_UNSUPPORTED; }
    return KTLVO_OK;
}

static int frames_args_ok(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi *roi_in, const ktlvo_gftt_params *p,
                          int cap, ktlvo_roi &roi) {
    if (p->block_size != 3) return KTLVO_ERR_UNSUPPORTED;
    if (cap <= 0 || p->max_corners > cap) return KTLVO_ERR_INVALID;
    if (!(p->quality > 0) || !(p->min_dist >= 0)) return KTLVO_ERR_INVALID;
    if (n_frames > MAX_BATCH) return KTLVO_ERR_CAPACITY;
    for (int f = 0; f < n_frames; f++) {
        if (!slot_ok(ctx, slots[f])) return KTLVO_ERR_INVALID;
        if (ctx->built[slots[f]] < 0) return KTLVO_ERR_STATE;
    }
    roi = roi_in ? *roi_in : ktlvo_roi{0, 0, ctx->cfg.width, ctx->cfg.height};
    if (roi.w <= 0 || roi.h <= 0 || roi.x < 0 || roi.y < 0 || roi.x + roi.w > ctx->cfg.width || roi.y + roi.h > ctx->cfg.height)
        return KTLVO_ERR_INVALID;
    if (roi.w > 65535 || roi.h > 65535) return KTLVO_ERR_UNSUPPORTED;
    return KTLVO_OK;
}

// ---- scratch of the allocation-free entry point -----------------------------------------------------------
// ktlvo_detect_gftt_noalloc runs the same kernels as ktlvo_detect_gftt on a fixed device array instead of
// the context's grown-on-demand work area.  Every GPU the library runs on holds its own copy of the array
// (4 MiB + 8 KiB) until the process ends.  In host code the array's name is only a handle: its device
// address is looked up once per context, with the context's device current (corner_noalloc_init, called by
// ktlvo_create), and that address is what the kernels are given.
constexpr int NOALLOC_MAX_ROIS = RoiTable::MAX;          // 192
constexpr size_t NOALLOC_MAX_PIXELS = (size_t)512 << 10; // sum of the ROI areas of one call
constexpr size_t NOALLOC_SCRATCH_BYTES = ((size_t)4 << 20) + ((size_t)8 << 10);
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

int corner_noalloc_init(ktlvo_ctx *ctx) {
    void *ptr = nullptr;
    KTLVO_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    ctx->noalloc_scratch = ptr;
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_detect_gftt_noalloc(ktlvo_ctx *ctx, int slot, const ktlvo_roi *rois_in, int n_roi,
                                         const ktlvo_gftt_params *p, float *d_xy, float *d_response, int *d_counts,
                                         int cap) {
    if (!ctx || !p || n_roi < 0 || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_roi == 0) return KTLVO_OK;
    if (!rois_in || !d_xy || !d_response || !d_counts || cap <= 0) return KTLVO_ERR_INVALID;
    if (p->block_size != 3) return KTLVO_ERR_UNSUPPORTED;
    if (p->max_corners > cap) return KTLVO_ERR_INVALID;
    if (!(p->quality > 0) || !(p->min_dist >= 0)) return KTLVO_ERR_INVALID;
    if (ctx->built[slot] < 0) return KTLVO_ERR_STATE;
    if (!ctx->noalloc_scratch) return KTLVO_ERR_STATE;
    if (n_roi > NOALLOC_MAX_ROIS) {
        ctx->last_error = "ktlvo_detect_gftt_noalloc: more than 192 ROIs";
        return KTLVO_ERR_CAPACITY;
    }
    RoiDev rois[NOALLOC_MAX_ROIS];
    size_t total;
    int maxw, maxh;
    int rc = check_rois(ctx, rois_in, n_roi, rois, total, maxw, maxh);
    if (rc != KTLVO_OK) return rc;
    if (total > NOALLOC_MAX_PIXELS || corner_scratch_bytes(n_roi, total) > NOALLOC_SCRATCH_BYTES) {
        ctx->last_error = "ktlvo_detect_gftt_noalloc: ROI areas sum to more than 524288 pixels";
        return KTLVO_ERR_CAPACITY;
    }
    if (!rois_fit_lds(rois, n_roi, p->min_dist) || !rois_fit_table(rois, n_roi)) {
        ctx->last_error = "ktlvo_detect_gftt_noalloc: an ROI exceeds 4096 pixels or 1024 min-distance cells";
        return KTLVO_ERR_CAPACITY;
    }
    CornerScratch s;
    corner_scratch_carve(ctx->noalloc_scratch, n_roi, total, s);
    // corner_setup clears the counters the kernels accumulate into; the candidate keys of an earlier call
    // are dead (each ROI reads only the first cand_cnt keys, which this call writes)
    return corner_detect_small_dev(ctx, slot, rois, n_roi, maxw, maxh, p, s, d_xy, d_response, d_counts, cap);
}

extern "C" int ktlvo_detect_gftt_frames_dev(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi *roi_in,
                                            const ktlvo_gftt_params *p, float *d_xy, float *d_response, int *d_counts,
                                            int cap) {
    if (!ctx || !p || n_frames < 0 || (n_frames > 0 && !slots)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_frames == 0) return KTLVO_OK;
    if (!d_xy || !d_response || !d_counts) return KTLVO_ERR_INVALID;
    ktlvo_roi roi;
    int rc = frames_args_ok(ctx, slots, n_frames, roi_in, p, cap, roi);
    if (rc != KTLVO_OK) return rc;
    ProfScope ps(ctx, KTLVO_K_CORNER);
    rc = detect_frames_dev(ctx, slots, n_frames, roi, p, d_xy, d_response, d_counts, cap);
    return rc != KTLVO_OK ? rc : frames_overflow_check(ctx, n_frames);
}

// Detection and pyramid build of a batch side by side.  The two share nothing but their input (the detector reads
// the level-0 image planes, the build reads them and writes the other levels and the derivative planes), and they load
// the chip in complementary ways: the detector's first kernel (candidates) is bound by instruction issue, its sort and
// greedy pass are a workgroup per frame and leave most of the chip idle, the pyramid kernels are bound by memory
// traffic.  The build is queued on the device's side stream behind a fence on the context's stream, the detection on
// the context's stream, and the context's stream then waits for the build, so whatever the caller queues next
// (ktlvo_pyrlk_batch_dev) finds corners and pyramids complete.  Measured on the benchmark's 254 + 256 frames, same box:
// 0.87 ms for the two calls in turn, 0.69 ms side by side (the build queued beside the candidates launch; queued
// behind it, beside the sort and the greedy pass only, 0.73 ms: the pyramid kernels then end after the greedy pass).
extern "C" int ktlvo_detect_build_frames_dev(ktlvo_ctx *ctx, const int *detect_slots, int n_frames, const ktlvo_roi *roi_in,
                                             const ktlvo_gftt_params *p, float *d_xy, float *d_response, int *d_counts,
                                             int cap, const int *build_slots, int n_build, int max_level, int win) {
    if (!ctx || !p || n_frames < 0 || (n_frames > 0 && !detect_slots) || n_build < 0 || (n_build > 0 && !build_slots))
        return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_build == 0) return ktlvo_detect_gftt_frames_dev(ctx, detect_slots, n_frames, roi_in, p, d_xy, d_response, d_counts, cap);
    if (n_frames == 0) return ktlvo_build_pyramids(ctx, build_slots, n_build, max_level, win);
    if (!d_xy || !d_response || !d_counts) return KTLVO_ERR_INVALID;
    ktlvo_roi roi; // refuse bad arguments of either half before anything is queued
    int rc = frames_args_ok(ctx, detect_slots, n_frames, roi_in, p, cap, roi);
    if (rc != KTLVO_OK) return rc;
    if (max_level < 0) return KTLVO_ERR_INVALID;
    if (win != ctx->cfg.win) return KTLVO_ERR_UNSUPPORTED;
    for (int i = 0; i < n_build; i++) {
        if (!slot_ok(ctx, build_slots[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[build_slots[i]] < 0) return KTLVO_ERR_STATE;
    }
    rc = device_streams(ctx);
    if (rc != KTLVO_OK) return rc;
    hipStream_t const main_stream = ctx->stream;
    bool forked = false;
    auto build_on_side_stream = [&]() -> int {
        KTLVO_HIP(ctx, hipEventRecord(ctx->ev_fork, main_stream));
        KTLVO_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        ctx->stream = ctx->side_stream; // the build's launches (and its profiling events) go to the side stream
        const int rb = ktlvo_build_pyramids(ctx, build_slots, n_build, max_level, win);
        ctx->stream = main_stream;
        if (rb != KTLVO_OK) return rb;
        KTLVO_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
        forked = true;
        return KTLVO_OK;
    };
    {
        ProfScope ps(ctx, KTLVO_K_CORNER);
        rc = detect_frames_dev(ctx, detect_slots, n_frames, roi, p, d_xy, d_response, d_counts, cap, build_on_side_stream);
    }
    // (also after a failed detection: the planes are being written)
    if (forked) KTLVO_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
    return rc != KTLVO_OK ? rc : frames_overflow_check(ctx, n_frames);
}

extern "C" int ktlvo_detect_gftt_frames(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi *roi_in,
                                        const ktlvo_gftt_params *p, float *xy, float *response, int *counts, int cap) {
    if (!ctx || !p || n_frames < 0 || (n_frames > 0 && !slots)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_frames == 0) return KTLVO_OK;
    if (!xy || !response || !counts) return KTLVO_ERR_INVALID;
    ktlvo_roi roi;
    int rc = frames_args_ok(ctx, slots, n_frames, roi_in, p, cap, roi);
    if (rc != KTLVO_OK) return rc;
    // the results (n_frames x cap points) get a device buffer of their own, kept with the context and grown on demand
    const size_t b_xy = (size_t)n_frames * cap * 8, b_r = (size_t)n_frames * cap * 4, b_c = 4 * (size_t)n_frames;
    if (b_xy + b_r + b_c > ctx->frames_out_bytes) {
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_frames_out) KTLVO_HIP(ctx, hipFree(ctx->d_frames_out));
        ctx->d_frames_out = nullptr; ctx->frames_out_bytes = 0;
        KTLVO_HIP(ctx, hipMalloc(&ctx->d_frames_out, b_xy + b_r + b_c));
        ctx->frames_out_bytes = b_xy + b_r + b_c;
    }
    float *d_xy = (float *)ctx->d_frames_out, *d_resp = (float *)((char *)ctx->d_frames_out + b_xy);
    int *d_counts = (int *)((char *)ctx->d_frames_out + b_xy + b_r);
    ProfScope ps(ctx, KTLVO_K_CORNER);
    rc = detect_frames_dev(ctx, slots, n_frames, roi, p, d_xy, d_resp, d_counts, cap);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(xy, d_xy, b_xy, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(response, d_resp, b_r, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipMemcpyAsync(counts, d_counts, b_c, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return frames_overflow_check(ctx, n_frames);
}

extern "C" int ktlvo_corner_response(ktlvo_ctx *ctx, int slot, const ktlvo_roi *roi, int use_harris, double harris_k,
                                     float *out) {
    if (!ctx || !roi || !out || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (ctx->built[slot] < 0) return KTLVO_ERR_STATE;
    RoiDev rois[1];
    size_t total;
    int maxw, maxh;
    int rc = check_rois(ctx, roi, 1, rois, total, maxw, maxh);
    if (rc != KTLVO_OK) return rc;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_resp = up(corner_scratch_bytes(1, 0));
    rc = ensure_misc(ctx, o_resp + total * 4);
    if (rc != KTLVO_OK) return rc;
    CornerScratch s;
    corner_scratch_carve(ctx->d_misc, 1, 0, s);
    float *d_resp = (float *)((char *)ctx->d_misc + o_resp);
    rc = corner_scratch_setup(ctx, rois, 1, s);
    if (rc != KTLVO_OK) return rc;
    rc = run_candidates(ctx, slot, 1, maxw, maxh, use_harris, harris_k, s, d_resp, false, nullptr,
                        (size_t)maxw * maxh > (size_t)SEL_MAX, maxw, maxh);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(out, d_resp, total * 4, hipMemcpyDeviceToHost, ctx->stream));
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
    std::vector<RoiDev> rois_v(n_roi);
    RoiDev *rois = rois_v.data();
    size_t total;
    int maxw, maxh;
    int rc = check_rois(ctx, rois_in, n_roi, rois, total, maxw, maxh);
    if (rc != KTLVO_OK) return rc;

    // work area: scratch (ROI table, maxima, counters, candidate keys), then the results, adjacent so
    // that they come home in one copy: xy, responses, counts, overflow flag
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_xy = up(corner_scratch_bytes(n_roi, total));
    const size_t o_or = o_xy + up((size_t)n_roi * cap * 8), o_counts = o_or + up((size_t)n_roi * cap * 4);
    const size_t out_bytes = o_counts + 4 * (size_t)n_roi - o_xy;
    rc = ensure_misc(ctx, o_counts + up(4 * (size_t)n_roi));
    if (rc != KTLVO_OK) return rc;
    rc = ensure_pin(ctx, out_bytes + 4 * (size_t)n_roi + 48);
    if (rc != KTLVO_OK) return rc;
    CornerScratch s;
    corner_scratch_carve(ctx->d_misc, n_roi, total, s);
    char *base = (char *)ctx->d_misc;
    float *d_xy = (float *)(base + o_xy), *d_oresp = (float *)(base + o_or);
    int *d_counts = (int *)(base + o_counts);
    char *hp = (char *)ctx->h_pin;

    ProfScope ps(ctx, KTLVO_K_CORNER);
    if (rois_fit_lds(rois, n_roi, p->min_dist)) {
        rc = corner_detect_small_dev(ctx, slot, rois, n_roi, maxw, maxh, p, s, d_xy, d_oresp, d_counts, cap);
        if (rc != KTLVO_OK) return rc;
    } else {
        // larger ROIs (whole frames)
        rc = corner_scratch_setup(ctx, rois, n_roi, s);
        if (rc != KTLVO_OK) return rc;
        int minw = maxw, minh = maxh;
        for (int r = 0; r < n_roi; r++) { minw = std::min(minw, rois[r].w); minh = std::min(minh, rois[r].h); }
        rc = run_candidates(ctx, slot, n_roi, maxw, maxh, p->use_harris, p->harris_k, s, nullptr, true, nullptr, true, minw, minh);
        if (rc != KTLVO_OK) return rc;
        int *h_cnt = (int *)(hp + out_bytes);
        bool try_lds = ctx->frames_lds_sort;
        for (int r = 0; r < n_roi; r++) try_lds = try_lds && lds_sort_likely((size_t)rois[r].w * rois[r].h);
        if (try_lds) {
            // every ROI sorted in LDS and selected without a word from the device in between; the results come home
            // with a flag that says whether some ROI had too many keys for that, in which case the route below does
            // the list again
            int *d_many = nullptr;
            rc = select_large_lds(ctx, rois, n_roi, p, s, d_xy, d_oresp, d_counts, cap, &d_many);
            if (rc != KTLVO_OK) return rc;
            int *h_ovf = (int *)(hp + out_bytes + 4 * (size_t)n_roi), *h_many = h_ovf + 1;
            KTLVO_HIP(ctx, hipMemcpyAsync(hp, d_xy, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            KTLVO_HIP(ctx, hipMemcpyAsync(h_ovf, s.d_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
            KTLVO_HIP(ctx, hipMemcpyAsync(h_many, d_many, 4, hipMemcpyDeviceToHost, ctx->stream));
            KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!*h_many) {
                memcpy(xy, hp, (size_t)n_roi * cap * 8);
                memcpy(response, hp + (o_or - o_xy), (size_t)n_roi * cap * 4);
                memcpy(counts, hp + (o_counts - o_xy), 4 * (size_t)n_roi);
                if (*h_ovf) { ctx->last_error = "corner selection: cell capacity exceeded"; return KTLVO_ERR_UNSUPPORTED; }
                return KTLVO_OK;
            }
        }
        // the global sort is sized by the candidate counts, so they come home first
        KTLVO_HIP(ctx, hipMemcpyAsync(h_cnt, s.d_cnt, 4 * (size_t)n_roi, hipMemcpyDeviceToHost, ctx->stream));
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int *h_two = (int *)(hp + out_bytes + 4 * (size_t)n_roi);
        for (int attempt = 0; attempt < 2; attempt++) {
            const bool blind = attempt == 0 && ctx->frames_lds_sort && ctx->greedy_rounds;
            for (int r = 0; r < n_roi; r++) {
                rc = select_large(ctx, rois[r], r, h_cnt[r], p, s, d_xy + (size_t)r * cap * 2, d_oresp + (size_t)r * cap,
                                  d_counts + r, cap, h_two + 4, blind);
                if (rc != KTLVO_OK) return rc;
            }
            if (!blind) break;
            // blind: the results come home with the flag that says whether they stand
            KTLVO_HIP(ctx, hipMemcpyAsync(hp, d_xy, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            KTLVO_HIP(ctx, hipMemcpyAsync(h_two, s.d_overflow, 8, hipMemcpyDeviceToHost, ctx->stream));
            KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!h_two[1]) {
                memcpy(xy, hp, (size_t)n_roi * cap * 8);
                memcpy(response, hp + (o_or - o_xy), (size_t)n_roi * cap * 4);
                memcpy(counts, hp + (o_counts - o_xy), 4 * (size_t)n_roi);
                if (h_two[0]) { ctx->last_error = "corner selection: cell capacity exceeded"; return KTLVO_ERR_UNSUPPORTED; }
                return KTLVO_OK;
            }
        }
    }
    // one copy home through pinned memory, then into the caller's three arrays
    KTLVO_HIP(ctx, hipMemcpyAsync(hp, d_xy, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    int *h_ovf = (int *)(hp + out_bytes + 4 * (size_t)n_roi);
    KTLVO_HIP(ctx, hipMemcpyAsync(h_ovf, s.d_overflow, 4, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(xy, hp, (size_t)n_roi * cap * 8);
    memcpy(response, hp + (o_or - o_xy), (size_t)n_roi * cap * 4);
    memcpy(counts, hp + (o_counts - o_xy), 4 * (size_t)n_roi);
    if (*h_ovf) { ctx->last_error = "corner selection: cell capacity exceeded"; return KTLVO_ERR_UNSUPPORTED; }
    return KTLVO_OK;
}
