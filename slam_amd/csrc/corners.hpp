// Declarations shared by the translation units of the corner detector (cv::GFTTDetector::detect ->
// cv::goodFeaturesToTrack, featureTracking.cpp:102-103, 408-430):
//   corners_candidates.hip  response, ROI maximum and local-maximum test in one pass (sub-regions: LDS tiles; whole
//                           frames: registers), and the work area's set-up
//   corners_sort.hip        the candidates above the threshold in descending order: a workgroup per frame in LDS,
//                           ranges of the response for frames with more keys, a bitonic network in global memory as
//                           the last resort
//   corners_select.hip      the greedy minimum-distance pass: sub-regions in one workgroup each, whole frames as a
//                           walk over chunks of candidates or as a relaxation over all candidates at once
//   corners.hip             which of them a call takes, and the entry points
#pragma once
#include "ctx.hpp"
#include "internal.hpp"

#include <cstddef>

namespace ktlvo {

constexpr int SEL_MAX = 4096;  // largest ROI area / candidate count handled by corner_select
constexpr int CELL_MAX = 1024; // largest min-distance cell grid handled by corner_select
constexpr int SORT_CAP = 16384; // keys one workgroup sorts in LDS

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


// (cells: where the relaxation follows, the keys are counted per minimum-distance cell as they are written out, which
// spares the relaxation a pass over them; cell == 0: not wanted)
struct CellCount {
    int cell, gw;
    int *cell_cnt;
    size_t cells_stride;
};

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
// Grid in LDS: a count (4 bytes) and four places of 2 bytes per cell, a corner being kept as its offset inside its
// cell (x | y << 8, empty 0xFFFF), which a cell of up to 256 pixels allows.
constexpr size_t SEL_LDS_PER_CELL = 12;
constexpr size_t SEL_LDS_GRID_MAX = (size_t)100 << 10; // what the kernel's static arrays leave of a compute unit's 160 KiB
inline bool grid_fits_lds(int cell, int gw, int gh) {
    return cell <= 256 && (size_t)gw * gh * SEL_LDS_PER_CELL <= SEL_LDS_GRID_MAX;
}

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

// Kernels that ask for more than 64 KiB of dynamic LDS have to be told so once per device.
// (bytes: the most the kernel will be launched with; together with its static arrays it must fit the 160 KiB of a compute unit)
template <class K>
static int allow_big_lds(ktlvo_ctx *ctx, K kernel, size_t bytes, bool &done) {
    if (done) return KTLVO_OK;
    KTLVO_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done = true;
    return KTLVO_OK;
}

// ---- corners_candidates.hip ----
bool rois_fit_table(const RoiDev *rois, int n_roi);
int corner_scratch_setup(ktlvo_ctx *ctx, const RoiDev *rois, int n_roi, const CornerScratch &s);
// the same for n_frames frames that share one region (frame f's candidates at offset f * area, its slot in d_roi_slots[f])
int frames_scratch_setup(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi &roi, size_t area, const CornerScratch &s,
                         int *d_roi_slots);
int run_candidates(ktlvo_ctx *ctx, int slot, int n_roi, int maxw, int maxh, int use_harris, double k, const CornerScratch &s,
                   float *d_resp_out, bool want_candidates, const int *d_roi_slots = nullptr, bool large = false, int minw = 0,
                   int minh = 0);

// ---- corners_sort.hip ----
int launch_cand_filter(ktlvo_ctx *ctx, unsigned blocks, int n_lists, const unsigned long long *d_cand, const int *d_cnt,
                       const unsigned *d_max, double quality, unsigned long long *d_out, int *d_out_cnt, size_t cand_stride,
                       size_t out_stride);
int sort_keys(ktlvo_ctx *ctx, unsigned long long *d_keys, size_t np2, int n_lists, const int *d_n_used);
// a workgroup per list: the list's keys above its threshold, compacted and sorted in LDS (at most SORT_CAP of them; more raise *d_many)
int launch_sort_frames(ktlvo_ctx *ctx, int n_lists, const unsigned long long *d_cand, const int *d_cnt, const unsigned *d_max,
                       double quality, unsigned long long *d_out, int *d_out_cnt, size_t cand_stride, size_t out_stride, int *d_many);
size_t bucket_zero_bytes(int n_lists);
size_t bucket_work_bytes(int n_lists);
int bucket_sort(ktlvo_ctx *ctx, const unsigned long long *d_cand, const int *d_cnt, const unsigned *d_max, double quality,
                unsigned long long *d_out, int *d_out_cnt, size_t cand_stride, size_t out_stride, int n_lists, int n_sup,
                char *d_work, int *h_flag, bool *sorted, int *d_redo = nullptr, int max_total = 0x7fffffff,
                char *d_zero = nullptr, CellCount cells = CellCount{0, 0, nullptr, 0});

// ---- corners_select.hip ----
int launch_select_small(ktlvo_ctx *ctx, const SelArgs &a, int n_roi);
int launch_take_first(ktlvo_ctx *ctx, int n_lists, const unsigned long long *d_keys, const int *d_n, int w, int limit, float *d_xy,
                      float *d_resp, int *d_counts, size_t keys_stride, size_t out_stride);
bool select_uses_lds_grid(const SelLargeArgs &a);
int launch_select_large(ktlvo_ctx *ctx, const SelLargeArgs &a, int n_rois);
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
bool rounds_possible(size_t cells);
size_t rounds_work_bytes(size_t np2, size_t cells, int n_frames);
size_t rounds_zero_bytes(size_t cells, int n_frames, size_t np2);
CellCount rounds_cell_count(char *d_cells_zero, int cell, int gw, int gh);
int run_rounds(ktlvo_ctx *ctx, LfArgs a, int n_sup, size_t np2, char *work, int *h_flags, int queued = 0,
               int *d_unsettled = nullptr, char *d_cells_zero = nullptr, bool counted = false);

} // namespace ktlvo
