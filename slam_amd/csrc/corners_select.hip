// The greedy minimum-distance pass of cv::goodFeaturesToTrack (featureTracking.cpp:421-422 reaches it through
// GFTTDetector::detect): walk the candidates by rank, accept one unless an accepted corner lies within the distance.
// See corners.hpp.
#include "corners.hpp"

#include <cstring>

#include <algorithm>
#include <cmath>

namespace ktlvo {

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

constexpr int CH = 512;  // candidates per chunk of the greedy pass
constexpr int HT = 2048; // chunk-local hash of the survivors' cells (load factor <= 1/4)
constexpr int PAIRS_MAX = 192; // survivors of a chunk up to which all pairs are tested; beyond, the cell hash finds the neighbours
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
__global__ __launch_bounds__(256) void lf_count(LfArgs a) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_ptr[f]) return;
    a.state[f * a.keys_stride + i] = 0;
    const unsigned idx = (unsigned)a.keys[f * a.keys_stride + i];
    const int c = (int)((idx >> 16) / (unsigned)a.cell) * a.gw + (int)((idx & 0xffffu) / (unsigned)a.cell);
    atomicAdd(&a.cell_cnt[f * a.cells_stride + c], 1);
}
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

int launch_select_small(ktlvo_ctx *ctx, const SelArgs &a, int n_roi) {
    hipLaunchKernelGGL(corner_select, dim3(n_roi), dim3(256), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

int launch_take_first(ktlvo_ctx *ctx, int n_lists, const unsigned long long *d_keys, const int *d_n, int w, int limit, float *d_xy,
                      float *d_resp, int *d_counts, size_t keys_stride, size_t out_stride) {
    hipLaunchKernelGGL(corner_take_first, dim3((limit + 255) / 256, (unsigned)n_lists), dim3(256), 0, ctx->stream, d_keys, d_n, w, limit,
                       d_xy, d_resp, d_counts, keys_stride, out_stride);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// The greedy pass over n_rois sorted key lists, the grid in LDS when it fits (the cells in global memory are then not
// used and need not be cleared).
bool select_uses_lds_grid(const SelLargeArgs &a) { return grid_fits_lds(a.cell, a.gw, a.gh); }
int launch_select_large(ktlvo_ctx *ctx, const SelLargeArgs &a, int n_rois) {
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
static size_t lf_cells_stride(size_t cells) { return (cells + 1 + 3) & ~(size_t)3; }
bool rounds_possible(size_t cells) { return (lf_cells_stride(cells) + LF_SCAN - 1) / LF_SCAN <= (size_t)LF_SCAN_BLOCKS; }
size_t rounds_work_bytes(size_t np2, size_t cells, int n_frames) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t F = (size_t)n_frames, cs = lf_cells_stride(cells);
    return up(np2 * F) + up(np2 * 8 * F) + 2 * up(cs * 4 * F) + up(cs * 4 * F) + up((size_t)LF_SCAN_BLOCKS * 4 * F) +
           up((np2 / 1024 + 1) * 4 * F) + up((size_t)LF_MAX_ROUNDS * F * 4);
}
// (d_cells_zero: memory of rounds_zero_bytes() the caller has cleared -- together with other things, in one go -- for
// the cells' counts and handed-out places; without it they lie in `work` and are cleared here)
// (the cleared memory also holds the candidates' states; with `counted` the cells' counts are already in it -- the sort
// by ranges wrote them, rounds_cell_count() -- and the counting pass is left out)
size_t rounds_zero_bytes(size_t cells, int n_frames, size_t np2) {
    return 2 * ((lf_cells_stride(cells) * 4 * (size_t)n_frames + 255) & ~(size_t)255) + ((np2 * (size_t)n_frames + 255) & ~(size_t)255);
}
CellCount rounds_cell_count(char *d_cells_zero, int cell, int gw, int gh) {
    return CellCount{cell, gw, (int *)d_cells_zero, lf_cells_stride((size_t)gw * gh)};
}
int run_rounds(ktlvo_ctx *ctx, LfArgs a, int n_sup, size_t np2, char *work, int *h_flags, int queued,
                      int *d_unsettled, char *d_cells_zero, bool counted) {
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

} // namespace ktlvo
