// Internal declarations shared by the translation units of libktlvo_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "ktlvo.h"

namespace ktlvo {

constexpr int MAX_LEVELS = 8;
// Every pyramid level is stored with a PAD-pixel frame on all sides.  The frame holds the
// reflect-101 continuation of the level (what cv::buildOpticalFlowPyramid's copyMakeBorder
// produces for `win` pixels), so the LK kernel never needs border index arithmetic.
constexpr int PAD = 32;
// The Scharr planes store SCHARR_GAIN times calcSharrDeriv's values (see scharr_planes and the LK template).
constexpr int SCHARR_GAIN = 4;
constexpr int MAX_BATCH = 256; // slots / pairs passed by value to one launch (kernel arguments are limited to 4 KiB)

struct LevelDesc {
    uint8_t *origin;   // address of pixel (0,0)
    uint32_t *dorigin; // Scharr plane, element (0,0): int16 4 dx | int16 4 dy << 16; zero outside the level
    int w, h, pitch;   // pitch in elements (both planes); (x,y) addressable for x in [-PAD, w+PAD), y likewise
};
struct SlotDesc {
    LevelDesc lv[MAX_LEVELS];
};
struct SlotList {
    int n;
    int s[MAX_BATCH];
};
struct PairList { // slot ids; ktlvo_create limits num_slots to 65536
    uint16_t prev[MAX_BATCH];
    uint16_t next[MAX_BATCH];
};

struct Profiler {
    unsigned mask = 0;                            // bit k: launches of kernel id k are bracketed by events
    std::vector<hipEvent_t> pool;                 // recycled events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[KTLVO_K_COUNT];
    double total_ms[KTLVO_K_COUNT] = {0, 0, 0, 0, 0};
    long long launches[KTLVO_K_COUNT] = {0, 0, 0, 0, 0};
};

// Pinned host + device staging buffers for ktlvo_upload_image, used round-robin.
struct StageRing {
    static constexpr int N = 4;
    uint8_t *h_buf[N] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t *d_buf[N] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[N] = {nullptr, nullptr, nullptr, nullptr};
    bool used[N] = {false, false, false, false};
    int next = 0;
    // Whole batches (ktlvo_upload_images, and the stereo pair of the per-frame path): two device staging buffers
    // (and, for pageable sources, two pinned ones) used in turn, each made on its first use.  ktlvo_upload_images
    // issues its DMA on the context's copy stream, so the transfer of one batch runs while the kernels queued
    // before the call still work; the ingest launch that unpacks it waits for it on the context's stream.
    uint8_t *d_batch[2] = {nullptr, nullptr}, *h_batch[2] = {nullptr, nullptr};
    size_t batch_bytes = 0;                          // size of every buffer that exists
    hipEvent_t batch_done[2] = {nullptr, nullptr};   // the ingest launch reading d_batch[k] has run
    hipEvent_t copy_done[2] = {nullptr, nullptr};    // the DMA into d_batch[k] has landed
    bool batch_used[2] = {false, false};
    int batch_next = 0;
};

struct TrackerState; // tracker.hip

} // namespace ktlvo

struct ktlvo_ctx {
    ktlvo_config cfg;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr; // stream in use (own_stream or the caller's)
    // The process keeps ONE copy stream and ONE side stream per device, shared by all its contexts on that device
    // (ctx.hip, device_streams): a process has few hardware queues (4 by default), and streams beyond them share
    // queues, where a wait packet can sit in front of the very work it waits for.  Both are taken on first use.
    hipStream_t copy_stream = nullptr; // host-to-device DMA of ktlvo_upload_images
    hipStream_t side_stream = nullptr; // the pyramid build of ktlvo_detect_build_frames_dev, beside the detection
    hipEvent_t ev_fork = nullptr, ev_join = nullptr; // this context's fences between its stream and the side stream
    hipEvent_t ev_flags = nullptr;                   // behind the batched detection's two flags (the host waits for these, not for the stream)
    bool holds_device_streams = false;
    std::string last_error;

    // image slots
    int n_levels = 0; // levels allocated per slot (cfg.max_level limited by OpenCV's size rule)
    int lw[ktlvo::MAX_LEVELS], lh[ktlvo::MAX_LEVELS], lpitch[ktlvo::MAX_LEVELS];
    size_t loff[ktlvo::MAX_LEVELS];  // image plane offset inside a slot
    size_t doff[ktlvo::MAX_LEVELS];  // Scharr plane offset inside a slot
    size_t slot_bytes = 0;
    uint8_t *d_slab = nullptr;
    ktlvo::SlotDesc *d_slots = nullptr;
    std::vector<ktlvo::SlotDesc> h_slots;
    std::vector<int> built;    // highest level holding valid data per slot, -1 = no image
    std::vector<int> deriv_ok; // Scharr planes of the built levels are valid

    // device scratch for the host-pointer entry points
    float *d_pts[4] = {nullptr, nullptr, nullptr, nullptr}; // max_points x 2 floats each
    uint8_t *d_status = nullptr;
    float *d_err = nullptr;
    int *d_idx = nullptr;   // max_points + 1 ints
    void *d_misc = nullptr; // corner / ransac work area
    size_t misc_bytes = 0;
    bool frame_chain = true;    // the frame path's six tracks from one launch (KTLVO_FRAME_PER_STAGE=1 at creation: a launch per stage)
    bool pyr_per_level = false; // one pyrDown launch per level instead of the fused small-level kernel (A/B, tests)
    bool big_lds_select = false, big_lds_sort = false, big_lds_sort_buckets = false; // corner kernels that use more than 64 KiB of LDS have been enabled
    bool candidates_wave = true; // large ROIs take corner_candidates_wave (KTLVO_CANDIDATES_TILES=1 at creation: the LDS-tile kernel; A/B, tests)
    int relax_tries = 64; // times a candidate of the relaxation looks at its undecided predecessors again within a launch before it leaves the rest to the next one (KTLVO_RELAX_TRIES at creation: tests set 0 to see launches that do not settle)
    bool greedy_rounds = true;   // frames whose cell grid does not fit LDS take the greedy pass in rounds (KTLVO_GREEDY_SEQUENTIAL=1 at creation: a workgroup per frame walks the candidates; A/B, tests)
    bool frames_lds_sort = true; // whole frames are sorted in LDS, one workgroup per frame or per response range of a frame (KTLVO_FRAMES_GLOBAL_SORT=1 at creation: the bitonic network in global memory; A/B, tests)
    void *noalloc_scratch = nullptr; // this device's address of the fixed scratch array of ktlvo_detect_gftt_noalloc
    void *d_frames_out = nullptr; // results of ktlvo_detect_gftt_frames before they go home
    size_t frames_out_bytes = 0;
    void *d_aux = nullptr;  // sort buffer and cell grid of whole-frame corner selection
    size_t aux_bytes = 0;
    void *h_pin = nullptr;  // pinned host mirror of (part of) the work area: one DMA instead of one per array
    size_t pin_bytes = 0;

    ktlvo::StageRing stage;
    ktlvo::Profiler prof;
    ktlvo::TrackerState *tracker = nullptr;
};

namespace ktlvo {

int set_hip_error(ktlvo_ctx *ctx, hipError_t e, const char *what, const char *file, int line);

#define KTLVO_HIP(ctx, call)                                                                      \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return ktlvo::set_hip_error((ctx), e_, #call, __FILE__, __LINE__); \
    } while (0)

// First statement of every entry point after its argument check: the context's device becomes the
// calling thread's current device, so that lazy allocations and launches of a process that holds
// contexts on several GPUs land on the right one.
#define KTLVO_BIND(ctx) KTLVO_HIP((ctx), hipSetDevice((ctx)->device))

// RAII bracket: records start/stop events around a launch when profiling is on.
struct ProfScope {
    ktlvo_ctx *ctx;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(ktlvo_ctx *c, int k);
    ~ProfScope();
};

int device_streams(ktlvo_ctx *ctx); // ctx.hip: fills ctx->copy_stream and ctx->side_stream (shared per device)
int corner_noalloc_init(ktlvo_ctx *ctx); // corners.hip
int upload_pair(ktlvo_ctx *ctx, int slot_a, int slot_b, const uint8_t *a, const uint8_t *b, int w, int h); // pyramid.hip
int ensure_misc(ktlvo_ctx *ctx, size_t bytes);
int ensure_pin(ktlvo_ctx *ctx, size_t bytes);
inline bool slot_ok(const ktlvo_ctx *ctx, int s) { return s >= 0 && s < ctx->cfg.num_slots; }
int effective_max_level(const ktlvo_ctx *ctx, int a, int b, int requested);

// internal launchers (device pointers, asynchronous)
int launch_pyrlk(ktlvo_ctx *ctx, const PairList &pairs, int n_pairs, const float *d_prev, int n_per_pair,
                 float *d_next, uint8_t *d_status, float *d_err, const ktlvo_lk_params *p);

// General form: pair i reads prev[prev_off[i] ..] and writes next/status/err[next_off[i] ..];
// if d_n is given the point count is read from device memory (capped by n_per_pair), so a chain
// of gated stages can be enqueued without a host round trip.
int launch_pyrlk_ex(ktlvo_ctx *ctx, const PairList &pairs, int n_pairs, const float *d_prev, const int *prev_off,
                    float *d_next, const int *next_off, uint8_t *d_status, float *d_err, int n_per_pair,
                    const int *d_n, const ktlvo_lk_params *p);

// The six tracks of a stereo frame from one launch (pyrlk.hip); d_out = B, C, D, E, CHK, CHKB.
int launch_frame_chain(ktlvo_ctx *ctx, int sLL, int sLR, int sCL, int sCR, const float *dA, int n, float *const d_out[6],
                       const ktlvo_lk_params *p);

} // namespace ktlvo
