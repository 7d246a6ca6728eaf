// Device-pointer entry points shared between translation units (all asynchronous on ctx->stream).
#pragma once
#include "ctx.hpp"

namespace ktlvo {

// Up to four point lists copied through the kept indices by a gate launch.
constexpr int MAX_GATHER = 4;
struct Gather {
    const float *src[MAX_GATHER] = {nullptr, nullptr, nullptr, nullptr};
    float *dst[MAX_GATHER] = {nullptr, nullptr, nullptr, nullptr};
};

int launch_fb_gate(ktlvo_ctx *ctx, const float *d_f1, const float *d_f2, const float *d_f1chk, int n, const int *d_n,
                   int y_threshold, int *d_keep_idx, int *d_n_keep, const Gather &g);
int launch_circular_gate(ktlvo_ctx *ctx, const float *d_pcr, const float *d_clr, int n, int *d_keep_idx,
                         int *d_n_keep, const Gather &g);
int launch_frame_gates(ktlvo_ctx *ctx, const float *dA, const float *dB, const float *dC, const float *dD, const float *dE,
                       const float *dCHK, const float *dCHKB, int n, int *dK1, int *dK2, int *dK3, int *d_cnt, float *dU1,
                       float *dU2, float *dU3);
int device_get_optical_flow(ktlvo_ctx *ctx, int slot_1, int slot_2, const float *d_f1, int n_cap, const int *d_n,
                            int y_threshold, float *d_f2, float *d_chk, int *d_keep_idx, int *d_n_keep,
                            const Gather &g);

// ---- corner detection (corners.hip) --------------------------------------------------------------------
struct RoiDev {
    int x, y, w, h;
    int off; // offset of this ROI's pixels in the candidate array
};
// ROI list passed to a kernel by value (kernel arguments are limited to 4 KiB)
struct RoiTable {
    static constexpr int MAX = 192;
    struct {
        uint16_t x, y, w, h;
        int off;
    } r[MAX];
};
// Device memory the corner kernels need beyond the image and their outputs: the ROI table, per-ROI maximum
// response (as an order-preserving uint) and candidate count, an overflow flag, and the candidate keys
// (8 bytes per ROI pixel at most: every pixel of a plateau is a local maximum).
struct CornerScratch {
    RoiDev *d_rois;
    unsigned *d_max;
    int *d_cnt;
    int *d_overflow;
    unsigned long long *d_cand;
};
size_t corner_scratch_bytes(int n_roi, size_t total_px);
void corner_scratch_carve(void *base, int n_roi, size_t total_px, CornerScratch &s);

// Host-side pose helpers (ransac.hip)
void rodrigues_v2m(const double r[3], double R[9]);
void rodrigues_m2v(const double R[9], double r[3]);

} // namespace ktlvo
