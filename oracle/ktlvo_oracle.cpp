/*
 * ktlvo_oracle.cpp -- CPU oracle for the ktl_vo feature-tracking path (image side).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED: see ktlvo_oracle.h.
 * Build with -ffp-contract=off: every float/double expression below is meant to round after
 * each operation, as an SSE2 build of OpenCV 3.x does.
 *
 * Restated here:
 *   pyrDown / buildOpticalFlowPyramid / calcSharrDeriv / LKTrackerInvoker
 *       -> reached from featureTracking.cpp:205-209, 214-218, 457-479
 *   cornerEigenValsVecs / calcHarris / calcMinEigenVal / goodFeaturesToTrack / GFTTDetector
 *       -> featureTracking.cpp:102-103, 421-422
 *   adaptiveNonMaximalSuppresion -> featureTracking.cpp:258-311
 *   gates -> featureTracking.cpp:223-248, 486-501, 526-573
 */
#include "ktlvo_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

inline int reflect101(int p, int n) {
    /* cv::borderInterpolate(p, n, BORDER_REFLECT_101) */
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

inline int cv_round(double v) { return (int)lrint(v); }   /* round-half-even */
inline int cv_floor(float v) { return (int)std::floor(v); }

/* Blocks of a few sizes come and go with every pyramid (a frame stream builds two per call, always of the same
 * sizes); fresh blocks of that size come from the kernel page by page, which costs a single thread more than filling
 * them costs sixteen.  Freed blocks are kept by size and handed out again, as an image library's buffer pool does. */
class BlockCache {
public:
    static BlockCache &get() {
        static BlockCache c;
        return c;
    }
    void *take(size_t bytes) {
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].second == bytes) {
                    void *p = free_[i].first;
                    free_[i] = free_.back();
                    free_.pop_back();
                    return p;
                }
        }
        return ::operator new(bytes);
    }
    void give(void *p, size_t bytes) {
        {
            std::lock_guard<std::mutex> g(m_);
            if (free_.size() < 64 && bytes >= 4096) {
                free_.emplace_back(p, bytes);
                return;
            }
        }
        ::operator delete(p);
    }
    ~BlockCache() {
        for (auto &f : free_) ::operator delete(f.first);
    }

private:
    std::mutex m_;
    std::vector<std::pair<void *, size_t>> free_;
};

/* storage that is not cleared when it is made (every element is written by the code that fills it) */
template <class T>
struct Raw {
    T *p = nullptr;
    size_t n = 0;
    Raw() = default;
    Raw(const Raw &) = delete;
    Raw &operator=(const Raw &) = delete;
    Raw(Raw &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    Raw &operator=(Raw &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~Raw() { release(); }
    void release() {
        if (p) BlockCache::get().give(p, n * sizeof(T));
        p = nullptr; n = 0;
    }
    void alloc(size_t count) {
        release();
        p = (T *)BlockCache::get().take(count * sizeof(T));
        n = count;
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *data() { return p; }
    const T *data() const { return p; }
};

struct Level {
    int w = 0, h = 0, pad = 0;
    int pitch = 0;                 /* padded row length in pixels */
    Raw<uint8_t> img;              /* (h+2pad) x pitch, reflect-101 border */
    Raw<int16_t> deriv;            /* (h+2pad) x pitch x 2, zero border */
    const uint8_t *ipx(int x, int y) const { return &img[(size_t)(y + pad) * pitch + (x + pad)]; }
    const int16_t *dpx(int x, int y) const { return &deriv[((size_t)(y + pad) * pitch + (x + pad)) * 2]; }
};

/* A fixed set of worker threads that run ranges of a loop, the way cv::parallel_for_ hands stripes of a Range to
 * its pool (featureTracking.cpp:205 reaches it through calcOpticalFlowPyrLK).  Workers are made once and kept:
 * spawning threads per pyramid level costs more than a level's work at the upper levels. */
class Pool {
public:
    static Pool &get() {
        static Pool p;
        return p;
    }
    /* fn(i0, i1) over [0, n) cut into `parts` ranges, taken one after the other by `threads` threads (the caller's
     * among them) until none is left */
    void run(int n, int parts, int threads, const std::function<void(int, int)> &fn) {
        if (n <= 0) return;
        parts = std::max(1, std::min(parts, n));
        threads = std::max(1, std::min(threads, parts));
        if (threads == 1) {
            fn(0, n);
            return;
        }
        std::lock_guard<std::mutex> call(call_mutex_); /* one loop at a time */
        while ((int)th_.size() < threads - 1) th_.emplace_back([this] { work(); });
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn; n_ = n; parts_ = parts; next_ = 0; finished_ = 0; helpers_ = 0; max_helpers_ = threads - 1;
            gen_++;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return finished_ == parts_; });
        fn_ = nullptr;
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }

private:
    void drain() {
        for (;;) {
            int part, n, parts;
            const std::function<void(int, int)> *fn;
            {
                std::lock_guard<std::mutex> g(m_);
                if (next_ >= parts_) return;
                part = next_++; n = n_; parts = parts_; fn = fn_;
            }
            (*fn)((int)((long long)n * part / parts), (int)((long long)n * (part + 1) / parts));
            std::lock_guard<std::mutex> g(m_);
            if (++finished_ == parts_) done_.notify_one();
        }
    }
    void work() {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (helpers_ >= max_helpers_) continue; /* this loop asked for fewer threads than the pool holds */
                helpers_++;
            }
            drain();
        }
    }
    std::mutex call_mutex_, m_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> th_;
    const std::function<void(int, int)> *fn_ = nullptr;
    int n_ = 0, parts_ = 0, next_ = 0, finished_ = 0, helpers_ = 0, max_helpers_ = 0;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

int g_threads = 1; /* threads the image-wide stages (pyramid levels, borders, Scharr) are cut over; orc_pyrlk has its own argument */

void pyr_down_impl(const uint8_t *src, int w, int h, int ss, uint8_t *dst, int ds) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    /* straightforward 2-D form; integer arithmetic, so any evaluation order is exact.  Output rows are independent:
     * each thread takes a range of them with its own five row buffers. */
    Pool::get().run(dh, g_threads, g_threads, [&](int y_from, int y_to) {
        std::vector<int> rowbuf((size_t)5 * dw);
        for (int y = y_from; y < y_to; y++) {
            for (int j = 0; j < 5; j++) {
                const uint8_t *s = src + (size_t)reflect101(2 * y + j - 2, h) * ss;
                int *r = &rowbuf[(size_t)j * dw];
                const int x_in0 = std::min(dw, 1), x_in1 = std::max(x_in0, (w - 3) / 2 + 1 > dw ? dw : (w - 3) / 2 + 1);
                for (int x = 0; x < dw; x++) {
                    if (x >= x_in0 && x < x_in1) { /* 2x - 2 >= 0 and 2x + 2 <= w - 1: no reflection */
                        const uint8_t *c = s + 2 * x;
                        r[x] = c[-2] + 4 * c[-1] + 6 * c[0] + 4 * c[1] + c[2];
                        continue;
                    }
                    int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x;
                    int x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                    r[x] = s[x0] + 4 * s[x1] + 6 * s[x2] + 4 * s[x3] + s[x4];
                }
            }
            for (int x = 0; x < dw; x++) {
                int v = rowbuf[x] + 4 * rowbuf[dw + x] + 6 * rowbuf[2 * dw + x] + 4 * rowbuf[3 * dw + x] +
                        rowbuf[4 * dw + x];
                dst[(size_t)y * ds + x] = (uint8_t)((v + 128) >> 8);
            }
        }
    });
}

void scharr_impl(const uint8_t *src, int w, int h, int ss, int16_t *dst, int dpitch /* in pairs */) {
    Pool::get().run(h, g_threads, g_threads, [&](int y_from, int y_to) {
        std::vector<int> t0(w + 2), t1(w + 2);
        for (int y = y_from; y < y_to; y++) {
            const uint8_t *r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * ss;
            const uint8_t *r1 = src + (size_t)y * ss;
            const uint8_t *r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * ss;
            int *a = &t0[1], *b = &t1[1];
            for (int x = 0; x < w; x++) {
                a[x] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
                b[x] = r2[x] - r0[x];
            }
            int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
            a[-1] = a[x0]; a[w] = a[x1];
            b[-1] = b[x0]; b[w] = b[x1];
            int16_t *d = dst + (size_t)y * dpitch * 2;
            for (int x = 0; x < w; x++) {
                d[2 * x] = (int16_t)(a[x + 1] - a[x - 1]);
                d[2 * x + 1] = (int16_t)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
            }
        }
    });
}

}  // namespace

struct orc_pyramid {
    std::vector<Level> levels;
    int win = 21;
    bool with_deriv = false;
};

extern "C" void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n > 256 ? 256 : n; }

extern "C" void orc_pyr_down(const uint8_t *src, int w, int h, int ss, uint8_t *dst, int ds) {
    pyr_down_impl(src, w, h, ss, dst, ds);
}

extern "C" void orc_scharr(const uint8_t *src, int w, int h, int ss, int16_t *dst) {
    scharr_impl(src, w, h, ss, dst, w);
}

extern "C" orc_pyramid *orc_pyramid_create(const uint8_t *img, int w, int h, int stride, int max_level,
                                           int win, int with_deriv) {
    if (!img || w <= 0 || h <= 0 || win < 3 || max_level < 0) return nullptr;
    orc_pyramid *p = new orc_pyramid;
    p->win = win;
    p->with_deriv = with_deriv != 0;
    int lw = w, lh = h;
    Raw<uint8_t> cur, nxt;
    cur.alloc((size_t)w * h);
    for (int y = 0; y < h; y++) memcpy(&cur[(size_t)y * w], img + (size_t)y * stride, w);
    for (int level = 0; level <= max_level; level++) {
        Level L;
        L.w = lw; L.h = lh; L.pad = win; L.pitch = lw + 2 * win;
        L.img.alloc((size_t)(lh + 2 * win) * L.pitch);
        /* copyMakeBorder(..., BORDER_REFLECT_101 | BORDER_ISOLATED) */
        Pool::get().run(lh + 2 * win, g_threads, g_threads, [&](int r_from, int r_to) {
            for (int y = r_from - win; y < r_to - win; y++) {
                const uint8_t *s = &cur[(size_t)reflect101(y, lh) * lw];
                uint8_t *d = &L.img[(size_t)(y + win) * L.pitch];
                for (int x = -win; x < 0; x++) d[x + win] = s[reflect101(x, lw)];
                memcpy(d + win, s, lw);
                for (int x = lw; x < lw + win; x++) d[x + win] = s[reflect101(x, lw)];
            }
        });
        if (with_deriv) {
            /* calcSharrDeriv into the interior, copyMakeBorder(BORDER_CONSTANT) -> zero frame */
            L.deriv.alloc((size_t)(lh + 2 * win) * L.pitch * 2);
            Pool::get().run(lh + 2 * win, g_threads, g_threads, [&](int r_from, int r_to) {
                memset(&L.deriv[(size_t)r_from * L.pitch * 2], 0, (size_t)(r_to - r_from) * L.pitch * 2 * sizeof(int16_t));
            });
            scharr_impl(cur.data(), lw, lh, lw, &L.deriv[((size_t)win * L.pitch + win) * 2], L.pitch);
        }
        p->levels.push_back(std::move(L));
        if (level == max_level) break;
        int nw = (lw + 1) / 2, nh = (lh + 1) / 2;
        if (nw <= win || nh <= win) break; /* buildOpticalFlowPyramid stops here */
        nxt.alloc((size_t)nw * nh);
        pyr_down_impl(cur.data(), lw, lh, lw, nxt.data(), nw);
        std::swap(cur, nxt);
        lw = nw; lh = nh;
    }
    return p;
}

extern "C" void orc_pyramid_destroy(orc_pyramid *p) { delete p; }
extern "C" int orc_pyramid_levels(const orc_pyramid *p) { return p ? (int)p->levels.size() - 1 : -1; }
extern "C" int orc_pyramid_level_size(const orc_pyramid *p, int level, int *w, int *h) {
    if (!p || level < 0 || level >= (int)p->levels.size()) return -1;
    *w = p->levels[level].w; *h = p->levels[level].h;
    return 0;
}
extern "C" int orc_pyramid_level_copy(const orc_pyramid *p, int level, uint8_t *dst) {
    if (!p || level < 0 || level >= (int)p->levels.size()) return -1;
    const Level &L = p->levels[level];
    for (int y = 0; y < L.h; y++) memcpy(dst + (size_t)y * L.w, L.ipx(0, y), L.w);
    return 0;
}
extern "C" int orc_pyramid_deriv_copy(const orc_pyramid *p, int level, int16_t *dst) {
    if (!p || !p->with_deriv || level < 0 || level >= (int)p->levels.size()) return -1;
    const Level &L = p->levels[level];
    for (int y = 0; y < L.h; y++) memcpy(dst + (size_t)y * L.w * 2, L.dpx(0, y), (size_t)L.w * 4);
    return 0;
}

static uint8_t *g_iter_log = nullptr; /* optional diagnostics: iterations per (level, point) */
static int g_iter_log_n = 0;
extern "C" void orc_set_iter_log(uint8_t *buf, int n_points) { g_iter_log = buf; g_iter_log_n = n_points; }

/* Order in which the 441 terms of each window sum are added.  0: one float accumulator, raster order
 * (LKTrackerInvoker's scalar path; the oracle's reference behaviour).  1: four float accumulators taken in
 * turn along a row and added at the end, the way a 4-lane SIMD build accumulates.  2: pairwise (tree) float
 * sum.  3: the integer terms added exactly (int64) and the total converted to float once, round to nearest
 * even: the one order-independent reading of the sum, which a parallel implementation can reproduce bit for
 * bit whatever its own order of addition.  Modes 1 to 3 also measure how far the RESULT moves when only the
 * rounding of these sums changes; every integer term is the same in all modes. */
static int g_lk_sum_mode = 0;
extern "C" void orc_set_lk_sum_mode(int mode) { g_lk_sum_mode = mode < 0 || mode > 3 ? 0 : mode; }
extern "C" int orc_get_lk_sum_mode(void) { return g_lk_sum_mode; }

namespace {

/* accumulator for one window sum under sum mode MODE */
template <int MODE>
struct WinSum {
    float lane[4] = {0, 0, 0, 0};
    float terms[MODE == 2 ? 32 * 32 : 1];
    long long exact = 0;
    int n = 0;
    /* the term is an integer (a product of two int16 values, or |diff|); OpenCV converts it to float and
     * adds it to a float accumulator */
    inline void add(int t, int x) {
        if (MODE == 3) { exact += t; return; }
        const float v = (float)t;
        if (MODE == 0) lane[0] += v;
        else if (MODE == 1) lane[x & 3] += v;
        else terms[n++] = v;
    }
    static float tree(const float *t, int n) {
        if (n == 1) return t[0];
        if (n == 2) return t[0] + t[1];
        const int h = n / 2;
        return tree(t, h) + tree(t + h, n - h);
    }
    inline float total() const {
        if (MODE == 0) return lane[0];
        if (MODE == 1) return (lane[0] + lane[2]) + (lane[1] + lane[3]);
        if (MODE == 3) return (float)exact; /* one rounding, to nearest even */
        return n ? tree(terms, n) : 0.f;
    }
};

/* LKTrackerInvoker::operator() (scalar path: float accumulators) for points [i0,i1) of one level. */
template <int MODE>
void lk_level_range(const Level &I, const Level &J, const float *prevPts, float *nextPts, uint8_t *status,
                    float *err, int level, int maxLevel, int win, int maxCount, double epsSq,
                    double minEigThr, int i0, int i1) {
    const float halfWin = (win - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    std::vector<int16_t> Ibuf((size_t)win * win), dIbuf((size_t)win * win * 2);
    const int stepI = I.pitch, stepJ = J.pitch, dstep = I.pitch * 2;
    auto descale = [](int v, int n) { return (v + (1 << (n - 1))) >> n; };

    for (int pt = i0; pt < i1; pt++) {
        float px = prevPts[2 * pt] * (float)(1. / (1 << level));
        float py = prevPts[2 * pt + 1] * (float)(1. / (1 << level));
        float nx, ny;
        if (level == maxLevel) { nx = px; ny = py; }
        else { nx = nextPts[2 * pt] * 2.f; ny = nextPts[2 * pt + 1] * 2.f; }
        nextPts[2 * pt] = nx; nextPts[2 * pt + 1] = ny;

        px -= halfWin; py -= halfWin;
        int ipx = cv_floor(px), ipy = cv_floor(py);
        if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
            if (level == 0) { status[pt] = 0; err[pt] = 0; }
            continue;
        }
        float a = px - ipx, b = py - ipy;
        int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

        WinSum<MODE> sA11, sA12, sA22;
        for (int y = 0; y < win; y++) {
            const uint8_t *src = I.ipx(ipx, ipy + y);
            const int16_t *dsrc = I.dpx(ipx, ipy + y);
            int16_t *Ip = &Ibuf[(size_t)y * win];
            int16_t *dIp = &dIbuf[(size_t)y * win * 2];
            for (int x = 0; x < win; x++, dsrc += 2, dIp += 2) {
                int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 +
                                   src[x + stepI + 1] * iw11, W_BITS - 5);
                int ixval = descale(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 +
                                    dsrc[dstep + 2] * iw11, W_BITS);
                int iyval = descale(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 +
                                    dsrc[dstep + 3] * iw11, W_BITS);
                Ip[x] = (int16_t)ival;
                dIp[0] = (int16_t)ixval;
                dIp[1] = (int16_t)iyval;
                sA11.add(ixval * ixval, x);
                sA12.add(ixval * iyval, x);
                sA22.add(iyval * iyval, x);
            }
        }
        const float iA11 = sA11.total(), iA12 = sA12.total(), iA22 = sA22.total();
        float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < minEigThr || D < FLT_EPSILON) {
            if (level == 0) status[pt] = 0;
            continue;
        }
        D = 1.f / D;

        nx -= halfWin; ny -= halfWin;
        float pdx = 0, pdy = 0;
        int jcount = 0;
        for (int j = 0; j < maxCount; j++) {
            jcount = j + 1;
            int inx = cv_floor(nx), iny = cv_floor(ny);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (level == 0) status[pt] = 0;
                break;
            }
            a = nx - inx; b = ny - iny;
            iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            WinSum<MODE> sb1, sb2;
            for (int y = 0; y < win; y++) {
                const uint8_t *Jp = J.ipx(inx, iny + y);
                const int16_t *Ip = &Ibuf[(size_t)y * win];
                const int16_t *dIp = &dIbuf[(size_t)y * win * 2];
                for (int x = 0; x < win; x++, dIp += 2) {
                    int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 +
                                       Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                    sb1.add(diff * dIp[0], x);
                    sb2.add(diff * dIp[1], x);
                }
            }
            const float ib1 = sb1.total(), ib2 = sb2.total();
            float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
            float dx = (float)((A12 * b2 - A22 * b1) * D);
            float dy = (float)((A12 * b1 - A11 * b2) * D);
            nx += dx; ny += dy;
            nextPts[2 * pt] = nx + halfWin; nextPts[2 * pt + 1] = ny + halfWin;
            if ((double)dx * dx + (double)dy * dy <= epsSq) break;
            if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
                nextPts[2 * pt] -= dx * 0.5f;
                nextPts[2 * pt + 1] -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (g_iter_log && pt < g_iter_log_n) g_iter_log[(size_t)level * g_iter_log_n + pt] = (uint8_t)jcount;

        if (status[pt] && level == 0) {
            float fx = nextPts[2 * pt] - halfWin, fy = nextPts[2 * pt + 1] - halfWin;
            int inx = cv_floor(fx), iny = cv_floor(fy);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                status[pt] = 0;
                continue;
            }
            float aa = fx - inx, bb = fy - iny;
            iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            WinSum<MODE> serr;
            for (int y = 0; y < win; y++) {
                const uint8_t *Jp = J.ipx(inx, iny + y);
                const int16_t *Ip = &Ibuf[(size_t)y * win];
                for (int x = 0; x < win; x++) {
                    int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 +
                                       Jp[x + stepJ + 1] * iw11, W_BITS - 5) - Ip[x];
                    serr.add(std::abs(diff), x); /* |(float)diff|: diff is exact in float */
                }
            }
            const float errval = serr.total();
            err[pt] = errval * 1.f / (32 * win * win);
        }
    }
}

}  // namespace

extern "C" int orc_pyrlk(const orc_pyramid *prev, const orc_pyramid *next, const float *prev_xy, int n,
                         float *next_xy, uint8_t *status, float *err, int win, int max_level, int max_count,
                         double epsilon, double min_eig_thr, int nthreads) {
    if (!prev || !next || n < 0) return -1;
    if (!prev->with_deriv || prev->win != win || next->win != win) return -2;
    if (n == 0) return 0;
    int maxLevel = std::min(max_level, std::min(orc_pyramid_levels(prev), orc_pyramid_levels(next)));
    max_count = std::min(std::max(max_count, 0), 100);
    epsilon = std::min(std::max(epsilon, 0.), 10.);
    const double epsSq = epsilon * epsilon;
    for (int i = 0; i < n; i++) { status[i] = 1; err[i] = 0; }
    nthreads = std::max(1, std::min(nthreads, n));
    if (win > 32) return -2; /* WinSum's term buffer */
    auto range_fn = g_lk_sum_mode == 1 ? lk_level_range<1> : g_lk_sum_mode == 2 ? lk_level_range<2>
                    : g_lk_sum_mode == 3 ? lk_level_range<3> : lk_level_range<0>;
    for (int level = maxLevel; level >= 0; level--) {
        const Level &I = prev->levels[level];
        const Level &J = next->levels[level];
        if (I.w != J.w || I.h != J.h) return -3;
        /* stripes of the point range per level, as LKTrackerInvoker's parallel_for_ has them; more stripes than
         * threads, because iteration counts differ widely between points */
        Pool::get().run(n, nthreads * 4, nthreads, [&](int i0, int i1) {
            range_fn(I, J, prev_xy, next_xy, status, err, level, maxLevel, win, max_count, epsSq, min_eig_thr, i0, i1);
        });
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Corner response and goodFeaturesToTrack
 * ---------------------------------------------------------------------------------------- */
namespace {

/* cornerEigenValsVecs for blockSize=3, ksize=3, CV_8U input; writes cov-derived response. */
void corner_response_impl(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw, int rh,
                          bool harris, double k, float *out) {
    const double scale_d = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float s1 = (float)(1.0 * scale_d), s2 = (float)(2.0 * scale_d);
    auto P = [&](int x, int y) -> float {
        /* parent image pixel with reflect-101 at the parent's edges (filter is not isolated) */
        return (float)img[(size_t)reflect101(y, h) * stride + reflect101(x, w)];
    };
    std::vector<float> cxx((size_t)rw * rh), cxy((size_t)rw * rh), cyy((size_t)rw * rh);
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            int gx = rx + x, gy = ry + y;
            /* Sobel dx: row filter [-1 0 1] (exact), column filter (S0+S2)*f1 + S1*f0 */
            float r0 = P(gx + 1, gy - 1) - P(gx - 1, gy - 1);
            float r1 = P(gx + 1, gy) - P(gx - 1, gy);
            float r2 = P(gx + 1, gy + 1) - P(gx - 1, gy + 1);
            float dx = (r0 + r2) * s1 + r1 * s2;
            /* Sobel dy: row filter [s1 s2 s1] sequential, column filter S2 - S0 */
            float q0 = s1 * P(gx - 1, gy - 1); q0 += s2 * P(gx, gy - 1); q0 += s1 * P(gx + 1, gy - 1);
            float q2 = s1 * P(gx - 1, gy + 1); q2 += s2 * P(gx, gy + 1); q2 += s1 * P(gx + 1, gy + 1);
            float dy = q2 - q0;
            size_t i = (size_t)y * rw + x;
            cxx[i] = dx * dx; cxy[i] = dx * dy; cyy[i] = dy * dy;
        }
    /* boxFilter(3x3, normalize=false) on the ROI-sized cov, reflect-101 at its own edges,
     * double accumulation (sumType CV_64F for CV_32F input), result rounded to float. */
    for (int y = 0; y < rh; y++)
        for (int x = 0; x < rw; x++) {
            double a = 0, b = 0, c = 0;
            for (int j = -1; j <= 1; j++) {
                int yy = reflect101(y + j, rh);
                double ra = 0, rb = 0, rc = 0;
                for (int i = -1; i <= 1; i++) {
                    size_t q = (size_t)yy * rw + reflect101(x + i, rw);
                    ra += (double)cxx[q]; rb += (double)cxy[q]; rc += (double)cyy[q];
                }
                a += ra; b += rb; c += rc;
            }
            float fa = (float)a, fb = (float)b, fc = (float)c;
            float r;
            if (harris) {
                float t = fa * fc - fb * fb;
                r = (float)((double)t - k * (double)(fa + fc) * (double)(fa + fc));
            } else {
                float ha = fa * 0.5f, hc = fc * 0.5f;
                r = (float)((ha + hc) - std::sqrt((ha - hc) * (ha - hc) + fb * fb));
            }
            out[(size_t)y * rw + x] = r;
        }
}

}  // namespace

extern "C" int orc_corner_response(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw,
                                   int rh, int use_harris, double k, float *out) {
    if (!img || rw <= 0 || rh <= 0 || rx < 0 || ry < 0 || rx + rw > w || ry + rh > h) return -1;
    corner_response_impl(img, w, h, stride, rx, ry, rw, rh, use_harris != 0, k, out);
    return 0;
}

extern "C" int orc_good_features(const uint8_t *img, int w, int h, int stride, int rx, int ry, int rw,
                                 int rh, int max_corners, double quality, double min_dist, int use_harris,
                                 double k, float *xy, float *resp, int cap) {
    if (!img || rw <= 0 || rh <= 0 || rx < 0 || ry < 0 || rx + rw > w || ry + rh > h) return -1;
    std::vector<float> eig((size_t)rw * rh);
    corner_response_impl(img, w, h, stride, rx, ry, rw, rh, use_harris != 0, k, eig.data());
    double maxVal = 0; /* minMaxLoc */
    {
        float m = eig[0];
        for (float v : eig) m = std::max(m, v);
        maxVal = m;
    }
    const float thr = (float)(maxVal * quality);
    for (float &v : eig) v = v > thr ? v : 0.f; /* THRESH_TOZERO */
    std::vector<int> cand;
    for (int y = 1; y < rh - 1; y++)
        for (int x = 1; x < rw - 1; x++) {
            float v = eig[(size_t)y * rw + x];
            if (v == 0) continue;
            float m = v; /* dilate 3x3 */
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) m = std::max(m, eig[(size_t)(y + j) * rw + x + i]);
            if (v == m) cand.push_back(y * rw + x);
        }
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
        float va = eig[a], vb = eig[b];
        return va > vb ? true : va < vb ? false : a > b;
    });
    int ncorners = 0;
    if (min_dist >= 1) {
        const int cell = cv_round(min_dist);
        const int gw = (rw + cell - 1) / cell, gh = (rh + cell - 1) / cell;
        std::vector<std::vector<std::pair<float, float>>> grid((size_t)gw * gh);
        const double md2 = min_dist * min_dist;
        for (int c : cand) {
            int y = c / rw, x = c % rw;
            int xc = x / cell, yc = y / cell;
            int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
            int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (auto &m : grid[(size_t)yy * gw + xx]) {
                        float dx = x - m.first, dy = y - m.second;
                        if (dx * dx + dy * dy < md2) { good = false; break; }
                    }
            if (good) {
                grid[(size_t)yc * gw + xc].push_back({(float)x, (float)y});
                if (ncorners < cap) {
                    xy[2 * ncorners] = (float)x; xy[2 * ncorners + 1] = (float)y;
                    if (resp) resp[ncorners] = eig[c];
                }
                ++ncorners;
                if ((max_corners > 0 && ncorners == max_corners) || ncorners == cap) break;
            }
        }
    } else {
        for (int c : cand) {
            if (ncorners < cap) {
                xy[2 * ncorners] = (float)(c % rw); xy[2 * ncorners + 1] = (float)(c / rw);
                if (resp) resp[ncorners] = eig[c];
            }
            ++ncorners;
            if ((max_corners > 0 && ncorners == max_corners) || ncorners == cap) break;
        }
    }
    return ncorners;
}

/* adaptiveNonMaximalSuppresion, featureTracking.cpp:258-311.  The std::sort call and its
 * comparator are kept literally: with the all-zero responses OpenCV 3's GFTTDetector yields,
 * what remains of this function is exactly that call's permutation. */
extern "C" int orc_anms(float *xy, float *resp, int n, int num_to_keep, int *perm_out) {
    struct KP { float x, y, response; int src; };
    if (n < num_to_keep) { /* :261 */
        if (perm_out) for (int i = 0; i < n; i++) perm_out[i] = i;
        return n;
    }
    std::vector<KP> kp(n);
    for (int i = 0; i < n; i++) kp[i] = {xy[2 * i], xy[2 * i + 1], resp[i], i};
    std::sort(kp.begin(), kp.end(), [](const KP &l, const KP &r) { return l.response > r.response; });
    std::vector<double> radii(n), radiiSorted(n);
    const float robustCoeff = 1.11f;
    for (int i = 0; i < n; ++i) {
        const float response = kp[i].response * robustCoeff;
        double radius = std::numeric_limits<double>::max();
        for (int j = 0; j < i && kp[j].response > response; ++j) {
            float dx = kp[i].x - kp[j].x, dy = kp[i].y - kp[j].y;
            /* cv::norm(Point2f) = sqrt((double)x*x + (double)y*y) */
            radius = std::min(radius, std::sqrt((double)dx * dx + (double)dy * dy));
        }
        radii[i] = radius; radiiSorted[i] = radius;
    }
    std::sort(radiiSorted.begin(), radiiSorted.end(), [](const double &l, const double &r) { return l > r; });
    /* :299 reads radiiSorted[numToKeep], out of bounds when n == numToKeep; keep-all there. */
    const double decisionRadius = num_to_keep < n ? radiiSorted[num_to_keep] : -1.0;
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (radii[i] >= decisionRadius) {
            xy[2 * m] = kp[i].x; xy[2 * m + 1] = kp[i].y; resp[m] = kp[i].response;
            if (perm_out) perm_out[m] = kp[i].src;
            ++m;
        }
    return m;
}

extern "C" int orc_fb_gate(const float *f1, const float *f2, const float *f1chk, int n, int img_w, int img_h,
                           int xb, int yb, double max_track_dis, int y_threshold, uint8_t *keep) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        float x1 = f1[2 * i], y1 = f1[2 * i + 1], x2 = f2[2 * i], y2 = f2[2 * i + 1];
        float cx = f1chk[2 * i], cy = f1chk[2 * i + 1];
        /* float products and sum, sqrt in double (featureTracking.cpp:224-227) */
        double trackDis = std::sqrt((double)((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2)));
        bool drop = trackDis > max_track_dis || x2 < xb || x2 > img_w - xb || y2 < yb || y2 > img_h - yb ||
                    x1 - cx > 5 || y1 - cy > y_threshold;
        keep[i] = drop ? 0 : 1;
        m += keep[i];
    }
    return m;
}

extern "C" int orc_circular_gate(const float *pcr, const float *clr, int n, uint8_t *keep) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        /* std::abs is applied to the bool: the test is one-sided (featureTracking.cpp:489-490) */
        bool k = (pcr[2 * i] - clr[2 * i] < 2) && (pcr[2 * i + 1] - clr[2 * i + 1] < 2);
        keep[i] = k ? 1 : 0;
        m += keep[i];
    }
    return m;
}

extern "C" int orc_depth_gate(const float *ll, const float *lr, int n, double bf, double fx, uint8_t *keep,
                              float *depth_out) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        float disparity = ll[2 * i] - lr[2 * i];
        float depth = (float)(bf / disparity);
        float mThDepth = (float)(bf * 70.0f / fx);
        keep[i] = (depth > 0 && depth < mThDepth) ? 1 : 0;
        depth_out[i] = depth;
        m += keep[i];
    }
    return m;
}
