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
int device_get_optical_flow(ktlvo_ctx *ctx, int slot_1, int slot_2, const float *d_f1, int n_cap, const int *d_n,
                            int y_threshold, float *d_f2, float *d_chk, int *d_keep_idx, int *d_n_keep,
                            const Gather &g);

// Host-side pose helpers (ransac.hip)
void rodrigues_v2m(const double r[3], double R[9]);
void rodrigues_m2v(const double R[9], double r[3]);

} // namespace ktlvo
