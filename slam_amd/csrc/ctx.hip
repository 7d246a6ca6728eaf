// Context life cycle, stream handling, profiling events and error strings of libktlvo_hip.so.
#include "ctx.hpp"

#include <mutex>

#include <cstdlib>
#include <cstring>

namespace ktlvo {

int set_hip_error(ktlvo_ctx *ctx, hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx) ctx->last_error = buf;
    return KTLVO_ERR_HIP;
}

static hipEvent_t take_event(ktlvo_ctx *ctx) {
    Profiler &p = ctx->prof;
    if (!p.pool.empty()) {
        hipEvent_t e = p.pool.back();
        p.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(ktlvo_ctx *c, int k) : ctx(c), kid(k) {
    if (!(ctx->prof.mask & (1u << kid))) return;
    a = take_event(ctx);
    b = take_event(ctx);
    if (a && b) (void)hipEventRecord(a, ctx->stream);
}
ProfScope::~ProfScope() {
    if (!a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof.pending[kid].push_back({a, b});
}

int ensure_misc(ktlvo_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->misc_bytes) return KTLVO_OK;
    if (ctx->d_misc) {
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        KTLVO_HIP(ctx, hipFree(ctx->d_misc));
        ctx->d_misc = nullptr;
        ctx->misc_bytes = 0;
    }
    bytes = (bytes + 4095) & ~(size_t)4095;
    KTLVO_HIP(ctx, hipMalloc(&ctx->d_misc, bytes));
    ctx->misc_bytes = bytes;
    return KTLVO_OK;
}

int ensure_pin(ktlvo_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pin_bytes) return KTLVO_OK;
    const size_t old_bytes = ctx->pin_bytes;
    if (ctx->h_pin) {
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        KTLVO_HIP(ctx, hipHostFree(ctx->h_pin));
        ctx->h_pin = nullptr;
        ctx->pin_bytes = 0;
    }
    // at least 256 KiB and at least twice the old size: a sequence of slowly growing requests must
    // not turn into a sequence of reallocations
    if (bytes < 2 * old_bytes) bytes = 2 * old_bytes;
    if (bytes < ((size_t)256 << 10)) bytes = (size_t)256 << 10;
    bytes = (bytes + 4095) & ~(size_t)4095;
    KTLVO_HIP(ctx, hipHostMalloc(&ctx->h_pin, bytes, hipHostMallocDefault));
    ctx->pin_bytes = bytes;
    return KTLVO_OK;
}

int effective_max_level(const ktlvo_ctx *ctx, int a, int b, int requested) {
    int m = requested;
    if (ctx->built[a] < m) m = ctx->built[a];
    if (ctx->built[b] < m) m = ctx->built[b];
    return m;
}

} // namespace ktlvo

using namespace ktlvo;

namespace ktlvo {
// One copy stream and one side stream per device for the whole process, counted by the contexts that hold them and
// destroyed with the last one.  A context per host thread is this library's threading model, and a stream-feeding
// harness runs several; with three streams of its own per context, a handful of contexts outgrows the process's
// hardware queues (GPU_MAX_HW_QUEUES, 4 by default).  Streams that share a hardware queue execute in the order they
// were submitted to it, so a hipStreamWaitEvent packet of one stream can sit in that queue AHEAD of the kernel of
// another stream that is to signal the event, and neither moves.  Sharing keeps a process at (contexts + 2) streams
// per device instead of 3 x contexts; the streams themselves are thread-safe.
struct DeviceStreams {
    hipStream_t copy = nullptr, side = nullptr;
    int users = 0;
};
static std::mutex g_streams_mutex;
static DeviceStreams g_streams[64];

int device_streams(ktlvo_ctx *ctx) {
    if (ctx->holds_device_streams) return KTLVO_OK;
    if (ctx->device < 0 || ctx->device >= 64) return KTLVO_ERR_INVALID;
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    DeviceStreams &d = g_streams[ctx->device];
    if (!d.copy) KTLVO_HIP(ctx, hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking));
    if (!d.side) KTLVO_HIP(ctx, hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    if (!ctx->ev_fork) KTLVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->ev_join) KTLVO_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    d.users++;
    ctx->copy_stream = d.copy;
    ctx->side_stream = d.side;
    ctx->holds_device_streams = true;
    return KTLVO_OK;
}

static void release_device_streams(ktlvo_ctx *ctx) {
    if (!ctx->holds_device_streams) return;
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    DeviceStreams &d = g_streams[ctx->device];
    (void)hipStreamSynchronize(d.copy);
    (void)hipStreamSynchronize(d.side);
    if (--d.users == 0) {
        (void)hipStreamDestroy(d.copy);
        (void)hipStreamDestroy(d.side);
        d.copy = d.side = nullptr;
    }
    ctx->copy_stream = ctx->side_stream = nullptr;
    ctx->holds_device_streams = false;
}
} // namespace ktlvo

extern "C" int ktlvo_default_config(ktlvo_config *c) {
    if (!c) return KTLVO_ERR_INVALID;
    memset(c, 0, sizeof(*c));
    c->width = 1242; c->height = 375;
    c->fx = 707.0912; c->fy = 707.0912; c->cx = 601.8873; c->cy = 183.1104;
    c->bf = 386.1448;
    c->num_slots = 4;
    c->max_level = 4;
    c->win = 21;
    c->max_points = 20000;
    c->max_batch_pairs = MAX_BATCH;
    c->x_subregions = 20; c->y_subregions = 8; c->max_per_subregion = 100;
    c->x_boundary = 30; c->y_boundary = 30;
    c->max_track_dis = 100;
    c->anms_keep = 1000;
    c->keyframe_min_tracked = 300;
    c->ransac_iters = 300;
    c->ransac_reproj = 0.1f;
    c->ransac_conf = (double)0.99f;
    c->max_hypotheses = 1024;
    c->lk_sums = KTLVO_LK_SUMS_TREE;
    c->p3p_polish = 0;
    return KTLVO_OK;
}

extern "C" ktlvo_ctx *ktlvo_create(const ktlvo_config *cfg, int device_id) {
    if (!cfg || cfg->width <= 0 || cfg->height <= 0 || cfg->num_slots <= 0 || cfg->num_slots > 65536 || cfg->max_level < 0 ||
        cfg->max_level >= MAX_LEVELS || cfg->win < 3 || cfg->win + 2 > PAD || cfg->max_points <= 0 ||
        cfg->max_batch_pairs <= 0 || cfg->max_batch_pairs > MAX_BATCH)
        return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) {
        fprintf(stderr, "ktlvo_create: no HIP device %d (found %d)\n", device_id, ndev);
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    ktlvo_ctx *ctx = new ktlvo_ctx;
    ctx->cfg = *cfg;
    ctx->device = device_id;
    {
        const char *e = getenv("KTLVO_PYR_PER_LEVEL");
        ctx->pyr_per_level = e && e[0] == '1';
        e = getenv("KTLVO_FRAME_PER_STAGE");
        ctx->frame_chain = !(e && e[0] == '1');
        e = getenv("KTLVO_CANDIDATES_TILES");
        ctx->candidates_wave = !(e && e[0] == '1');
        e = getenv("KTLVO_RELAX_TRIES");
        if (e && e[0] >= '0' && e[0] <= '9') ctx->relax_tries = std::min(atoi(e), 1024);
        e = getenv("KTLVO_GREEDY_SEQUENTIAL");
        ctx->greedy_rounds = !(e && e[0] == '1');
        e = getenv("KTLVO_FRAMES_GLOBAL_SORT");
        ctx->frames_lds_sort = !(e && e[0] == '1');
    }

    // level geometry: OpenCV's buildOpticalFlowPyramid rule
    int w = cfg->width, h = cfg->height;
    size_t off = 0;
    ctx->n_levels = 0;
    for (int l = 0; l <= cfg->max_level; l++) {
        ctx->lw[l] = w; ctx->lh[l] = h;
        ctx->lpitch[l] = (w + 2 * PAD + 63) & ~63;
        ctx->loff[l] = off;
        off += (size_t)ctx->lpitch[l] * (h + 2 * PAD);
        off = (off + 255) & ~(size_t)255;
        ctx->doff[l] = off;
        off += (size_t)ctx->lpitch[l] * (h + 2 * PAD) * 4;
        off = (off + 255) & ~(size_t)255;
        ctx->n_levels = l + 1;
        int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= cfg->win || nh <= cfg->win) break;
        w = nw; h = nh;
    }
    ctx->slot_bytes = off + 256; // tail slack: aligned 16-byte loads may run past the last row

    bool ok = true;
    ok = ok && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) == hipSuccess;
    ctx->stream = ctx->own_stream;
    ok = ok && hipMalloc((void **)&ctx->d_slab, ctx->slot_bytes * cfg->num_slots) == hipSuccess;
    ok = ok && hipMemset(ctx->d_slab, 0, ctx->slot_bytes * cfg->num_slots) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_slots, sizeof(SlotDesc) * cfg->num_slots) == hipSuccess;
    if (ok) {
        ctx->h_slots.resize(cfg->num_slots);
        ctx->built.assign(cfg->num_slots, -1);
        ctx->deriv_ok.assign(cfg->num_slots, 0);
        for (int s = 0; s < cfg->num_slots; s++) {
            memset(&ctx->h_slots[s], 0, sizeof(SlotDesc));
            for (int l = 0; l < ctx->n_levels; l++) {
                LevelDesc &d = ctx->h_slots[s].lv[l];
                d.w = ctx->lw[l]; d.h = ctx->lh[l]; d.pitch = ctx->lpitch[l];
                uint8_t *slot = ctx->d_slab + ctx->slot_bytes * s;
                d.origin = slot + ctx->loff[l] + (size_t)PAD * d.pitch + PAD;
                d.dorigin = reinterpret_cast<uint32_t *>(slot + ctx->doff[l]) + (size_t)PAD * d.pitch + PAD;
            }
        }
        ok = hipMemcpy(ctx->d_slots, ctx->h_slots.data(), sizeof(SlotDesc) * cfg->num_slots,
                       hipMemcpyHostToDevice) == hipSuccess;
    }
    const size_t np = (size_t)cfg->max_points;
    for (int i = 0; i < 4 && ok; i++) ok = hipMalloc((void **)&ctx->d_pts[i], np * 2 * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_status, 4 * np) == hipSuccess; // four lists: the frame path's second LK launch
    ok = ok && hipMalloc((void **)&ctx->d_err, 4 * np * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_idx, (np + 1) * sizeof(int)) == hipSuccess;
    ok = ok && corner_noalloc_init(ctx) == KTLVO_OK;
    if (!ok) {
        fprintf(stderr, "ktlvo_create: device allocation failed: %s\n", hipGetErrorString(hipGetLastError()));
        ktlvo_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

namespace ktlvo { void tracker_free(ktlvo_ctx *ctx); }

extern "C" void ktlvo_destroy(ktlvo_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    tracker_free(ctx);
    for (int k = 0; k < KTLVO_K_COUNT; k++)
        for (auto &pr : ctx->prof.pending[k]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : ctx->prof.pool) (void)hipEventDestroy(e);
    for (int i = 0; i < StageRing::N; i++) {
        if (ctx->stage.h_buf[i]) (void)hipHostFree(ctx->stage.h_buf[i]);
        if (ctx->stage.d_buf[i]) (void)hipFree(ctx->stage.d_buf[i]);
        if (ctx->stage.done[i]) (void)hipEventDestroy(ctx->stage.done[i]);
    }
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (int k = 0; k < 2; k++) {
        if (ctx->stage.d_batch[k]) (void)hipFree(ctx->stage.d_batch[k]);
        if (ctx->stage.h_batch[k]) (void)hipHostFree(ctx->stage.h_batch[k]);
        if (ctx->stage.batch_done[k]) (void)hipEventDestroy(ctx->stage.batch_done[k]);
        if (ctx->stage.copy_done[k]) (void)hipEventDestroy(ctx->stage.copy_done[k]);
    }
    release_device_streams(ctx);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_flags) (void)hipEventDestroy(ctx->ev_flags);
    for (int i = 0; i < 4; i++) if (ctx->d_pts[i]) (void)hipFree(ctx->d_pts[i]);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    if (ctx->d_idx) (void)hipFree(ctx->d_idx);
    if (ctx->d_misc) (void)hipFree(ctx->d_misc);
    if (ctx->d_aux) (void)hipFree(ctx->d_aux);
    if (ctx->d_frames_out) (void)hipFree(ctx->d_frames_out);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->d_slab) (void)hipFree(ctx->d_slab);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int ktlvo_set_stream(ktlvo_ctx *ctx, void *hip_stream) {
    if (!ctx) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return KTLVO_OK;
}

extern "C" int ktlvo_sync(ktlvo_ctx *ctx) {
    if (!ctx) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KTLVO_OK;
}

extern "C" const char *ktlvo_strerror(int s) {
    switch (s) {
    case KTLVO_OK: return "ok";
    case KTLVO_ERR_INVALID: return "invalid argument";
    case KTLVO_ERR_UNSUPPORTED: return "parameter value not supported by the HIP path";
    case KTLVO_ERR_HIP: return "HIP runtime error";
    case KTLVO_ERR_CAPACITY: return "capacity of the context exceeded";
    case KTLVO_ERR_STATE: return "call order error (image or pyramid missing)";
    case KTLVO_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
    }
}

extern "C" const char *ktlvo_last_error(const ktlvo_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int ktlvo_profile_enable(ktlvo_ctx *ctx, int on) {
    if (!ctx) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    // 0: off; 1: every kernel; otherwise bit (1 + kernel id) selects a kernel
    ctx->prof.mask = on == 0 ? 0u : on == 1 ? ~0u : ((unsigned)on >> 1);
    return KTLVO_OK;
}

static int profile_drain(ktlvo_ctx *ctx) {
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    Profiler &p = ctx->prof;
    for (int k = 0; k < KTLVO_K_COUNT; k++) {
        for (auto &pr : p.pending[k]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                p.total_ms[k] += ms;
                p.launches[k] += 1;
            }
            p.pool.push_back(pr.first);
            p.pool.push_back(pr.second);
        }
        p.pending[k].clear();
    }
    return KTLVO_OK;
}

extern "C" int ktlvo_profile_reset(ktlvo_ctx *ctx) {
    if (!ctx) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    int rc = profile_drain(ctx);
    for (int k = 0; k < KTLVO_K_COUNT; k++) { ctx->prof.total_ms[k] = 0; ctx->prof.launches[k] = 0; }
    return rc;
}

extern "C" int ktlvo_profile_get(ktlvo_ctx *ctx, int kid, double *total_ms, long long *launches) {
    if (!ctx || kid < 0 || kid >= KTLVO_K_COUNT) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    int rc = profile_drain(ctx);
    if (total_ms) *total_ms = ctx->prof.total_ms[kid];
    if (launches) *launches = ctx->prof.launches[kid];
    return rc;
}
