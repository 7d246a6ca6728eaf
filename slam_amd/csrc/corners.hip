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
#include "corners.hpp"

#include <functional>

#include <cstring>

#include <algorithm>
#include <cmath>

namespace ktlvo {

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
    return launch_select_small(ctx, a, n_roi);
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
        rc = launch_cand_filter(ctx, (unsigned)fblocks, 1, s.d_cand + R.off, s.d_cnt + roi, s.d_max + roi, p->quality, d_keys, d_n, 0, 0);
        if (rc != KTLVO_OK) return rc;
        // How many passed is only known on the device; the sort is sized for the candidate count, which bounds it.
        rc = sort_keys(ctx, d_keys, np2, 1, d_n);
        if (rc != KTLVO_OK) return rc;
    }
    const int limit = p->max_corners > 0 ? std::min(p->max_corners, cap) : cap;
    if (!dist) {
        rc = launch_take_first(ctx, 1, d_keys, d_n, R.w, limit, d_xy, d_resp, d_count, 0, 0);
        if (rc != KTLVO_OK) return rc;
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
            rc = launch_take_first(ctx, 1, d_keys, d_n, R.w, limit, xy, resp, d_counts + r, 0, 0);
            if (rc != KTLVO_OK) return rc;
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
// behind_detection, if given, is called once the route that needs no word from the device has queued everything, before
// the host waits to learn whether that route stands: what it queues on the context's stream (the LK launch that reads
// the corners) starts the moment the detection ends instead of a host round trip later.  *done_again tells the caller
// that the route did not stand and the outputs were written a second time, behind whatever the hook queued.
static int detect_frames_dev(ktlvo_ctx *ctx, const int *slots, int n_frames, const ktlvo_roi &roi, const ktlvo_gftt_params *p,
                             float *d_xy, float *d_resp, int *d_counts, int cap, const std::function<int()> &before_candidates = nullptr,
                             const std::function<int()> &behind_detection = nullptr, bool *done_again = nullptr) {
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
    rc = frames_scratch_setup(ctx, slots, n_frames, roi, area, s, d_roi_slots);
    if (rc != KTLVO_OK) return rc;
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
        char *b = (char *)ctx->d_aux;
        unsigned long long *d_keys = (unsigned long long *)b;
        // the "do it again" flag: the work area's second flag, which the set-up kernel has cleared (no memset, and it
        // comes home with the overflow flag in one copy); only a grid in global memory needs clearing here
        int *d_n = (int *)(b + o_n), *d_many = s.d_overflow + 1;
        if (global_grid) {
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_cc, 0, o_cp - o_cc, ctx->stream));
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_cp, 0xFF, cells * 16 * n_frames, ctx->stream));
        }
        if (few) {
            bool sorted;
            KTLVO_HIP(ctx, hipMemsetAsync(b + o_zero, 0, total - o_zero, ctx->stream));
            rc = bucket_sort(ctx, s.d_cand, s.d_cnt, s.d_max, p->quality, d_keys, d_n, area, kstride, n_frames, (int)area, b + o_bucket,
                             nullptr, &sorted, d_many, FEW_KEYS_MAX, b + o_zero,
                             rounds ? rounds_cell_count(b + o_zero + bucket_zero_bytes(n_frames), cell, gw, gh) : CellCount{0, 0, nullptr, 0});
            if (rc != KTLVO_OK) return rc;
        } else {
            rc = launch_sort_frames(ctx, n_frames, s.d_cand, s.d_cnt, s.d_max, p->quality, d_keys, d_n, area, kstride, d_many);
            if (rc != KTLVO_OK) return rc;
        }
        if (!dist) {
            rc = launch_take_first(ctx, n_frames, d_keys, d_n, roi.w, limit, d_xy, d_resp, d_counts, kstride, (size_t)cap);
            if (rc != KTLVO_OK) return rc;
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
        KTLVO_HIP(ctx, hipMemcpyAsync(h_ovf, s.d_overflow, 8, hipMemcpyDeviceToHost, ctx->stream)); // (the second lands on h_many)
        if (behind_detection) {
            // wait for the two flags only, not for what the hook queues behind them
            if (!ctx->ev_flags) KTLVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_flags, hipEventDisableTiming));
            KTLVO_HIP(ctx, hipEventRecord(ctx->ev_flags, ctx->stream));
            rc = behind_detection();
            if (rc != KTLVO_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
            KTLVO_HIP(ctx, hipEventSynchronize(ctx->ev_flags));
        } else {
            KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (!*h_many) return KTLVO_OK;
        if (done_again) *done_again = true;
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
            rc = launch_cand_filter(ctx, fblocks, n_frames, s.d_cand, s.d_cnt, s.d_max, p->quality, d_keys, d_n, area, np2);
            if (rc != KTLVO_OK) return rc;
            rc = sort_keys(ctx, d_keys, np2, n_frames, d_n);
            if (rc != KTLVO_OK) return rc;
        }
        if (!dist) {
            rc = launch_take_first(ctx, n_frames, d_keys, d_n, roi.w, limit, d_xy, d_resp, d_counts, np2, (size_t)cap);
            if (rc != KTLVO_OK) return rc;
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

// The frame-pair stream's whole step in one call: corners of every pair's first frame, the pyramids, and the LK launch
// that tracks the corners into the pairs' second frames.  Made as two calls (ktlvo_detect_build_frames_dev, then
// ktlvo_pyrlk_batch_dev) the LK launch is queued only after the host has waited for the detection's flags, and the
// device idles for that round trip (about 50 us of a 3.3 ms step); here it is queued behind the detection before the
// host waits.  Results are those of the two calls; if the detection has to be done again by its other route (a frame
// with more corners above the threshold than the sort in LDS holds), the LK launch is done again as well.
extern "C" int ktlvo_detect_build_track_dev(ktlvo_ctx *ctx, const int *prev_slots, const int *next_slots, int n_pairs,
                                            const ktlvo_roi *roi_in, const ktlvo_gftt_params *p, float *d_xy, float *d_response,
                                            int *d_counts, int cap, const int *build_slots, int n_build, float *d_next_xy,
                                            uint8_t *d_status, float *d_err, const ktlvo_lk_params *lk) {
    if (!ctx || !p || !lk || n_pairs < 0 || (n_pairs > 0 && (!prev_slots || !next_slots)) || n_build < 0 ||
        (n_build > 0 && !build_slots))
        return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_pairs == 0) return n_build ? ktlvo_build_pyramids(ctx, build_slots, n_build, lk->max_level, lk->win) : KTLVO_OK;
    if (!d_xy || !d_response || !d_counts || !d_next_xy || !d_status || !d_err) return KTLVO_ERR_INVALID;
    if (n_pairs > ctx->cfg.max_batch_pairs || cap > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    ktlvo_roi roi;
    int rc = frames_args_ok(ctx, prev_slots, n_pairs, roi_in, p, cap, roi);
    if (rc != KTLVO_OK) return rc;
    if (lk->max_level < 0) return KTLVO_ERR_INVALID;
    if (lk->win != ctx->cfg.win) return KTLVO_ERR_UNSUPPORTED;
    for (int i = 0; i < n_pairs; i++) {
        if (!slot_ok(ctx, next_slots[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[next_slots[i]] < 0) return KTLVO_ERR_STATE;
    }
    for (int i = 0; i < n_build; i++) {
        if (!slot_ok(ctx, build_slots[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[build_slots[i]] < 0) return KTLVO_ERR_STATE;
    }
    rc = device_streams(ctx);
    if (rc != KTLVO_OK) return rc;
    hipStream_t const main_stream = ctx->stream;
    bool forked = false, joined = false, tracked = false, again = false;
    auto build_on_side_stream = [&]() -> int {
        if (n_build == 0) return KTLVO_OK;
        KTLVO_HIP(ctx, hipEventRecord(ctx->ev_fork, main_stream));
        KTLVO_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        ctx->stream = ctx->side_stream;
        const int rb = ktlvo_build_pyramids(ctx, build_slots, n_build, lk->max_level, lk->win);
        ctx->stream = main_stream;
        if (rb != KTLVO_OK) return rb;
        KTLVO_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
        forked = true;
        return KTLVO_OK;
    };
    auto track = [&]() -> int {
        if (forked && !joined) {
            KTLVO_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
            joined = true;
        }
        tracked = true;
        return ktlvo_pyrlk_batch_dev(ctx, prev_slots, next_slots, n_pairs, d_xy, cap, d_next_xy, d_status, d_err, lk);
    };
    rc = detect_frames_dev(ctx, prev_slots, n_pairs, roi, p, d_xy, d_response, d_counts, cap, build_on_side_stream, track, &again);
    if (forked && !joined) { // (also after a failed detection: the planes are being written)
        KTLVO_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        joined = true;
    }
    if (rc != KTLVO_OK) return rc;
    rc = frames_overflow_check(ctx, n_pairs);
    if (rc != KTLVO_OK) return rc;
    return tracked && !again ? KTLVO_OK : track();
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
