// The candidates of cv::goodFeaturesToTrack above the threshold in descending order (std::sort(greaterThanPtr) in
// OpenCV; featureTracking.cpp:421-422 reaches it through GFTTDetector::detect).  See corners.hpp.
#include "corners.hpp"

#include <cstring>

#include <algorithm>
#include <cmath>

namespace ktlvo {

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


int launch_cand_filter(ktlvo_ctx *ctx, unsigned blocks, int n_lists, const unsigned long long *d_cand, const int *d_cnt,
                       const unsigned *d_max, double quality, unsigned long long *d_out, int *d_out_cnt, size_t cand_stride,
                       size_t out_stride) {
    hipLaunchKernelGGL(cand_filter, dim3(blocks, (unsigned)n_lists), dim3(256), 0, ctx->stream, d_cand, d_cnt, d_max, quality, d_out,
                       d_out_cnt, cand_stride, out_stride);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

int launch_sort_frames(ktlvo_ctx *ctx, int n_lists, const unsigned long long *d_cand, const int *d_cnt, const unsigned *d_max,
                       double quality, unsigned long long *d_out, int *d_out_cnt, size_t cand_stride, size_t out_stride, int *d_many) {
    const int rc = allow_big_lds(ctx, corner_sort_frames<false>, (size_t)SORT_LDS_KEYS * 8, ctx->big_lds_sort);
    if (rc != KTLVO_OK) return rc;
    hipLaunchKernelGGL(corner_sort_frames<false>, dim3((unsigned)n_lists), dim3(1024), (size_t)SORT_LDS_KEYS * 8, ctx->stream, d_cand,
                       d_cnt, d_max, quality, d_out, d_out_cnt, cand_stride, out_stride, d_many, (const SortPlan *)nullptr,
                       CellCount{0, 0, nullptr, 0});
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

// n_lists lists of np2 keys each (np2 a power of two), back to back, each sorted descending.
// d_n_used[list] = number of keys at the front of the list, the rest being zero.
int sort_keys(ktlvo_ctx *ctx, unsigned long long *d_keys, size_t np2, int n_lists, const int *d_n_used) {
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

// Filter and sort of n_lists candidate lists (cand_stride keys apart; counts and maxima one element apart) with more
// keys than one workgroup sorts, by response ranges (corner_sort_hist, corner_sort_plan, corner_sort_scatter,
// corner_sort_frames<true>).  n_sup = the largest candidate count, d_work = n_lists plans, a flag and n_lists
// histograms (bucket_work_bytes), h_flag = one pinned int.  *sorted says whether out[] (lists out_stride keys apart) and out_cnt[]
// are now what cand_filter and sort_keys would have left at the lists' fronts; if not (a list the plan turned down),
// the caller takes those two.  With d_redo the call does not wait: a plan turned down raises *d_redo (and leaves its
// list empty, so whatever is queued behind does nothing with it), and the caller finds out when it next waits; n_sup
// may then be a bound (the ROI's area) and max_total says how many keys above the threshold out[] has room for.
static size_t bucket_plans_bytes(int n_lists) { return (sizeof(SortPlan) * (size_t)n_lists + 255) & ~(size_t)255; }
size_t bucket_zero_bytes(int n_lists) { return 256 + (size_t)n_lists * PLAN_BINS * 4; }
size_t bucket_work_bytes(int n_lists) { return bucket_plans_bytes(n_lists) + bucket_zero_bytes(n_lists); }
int bucket_sort(ktlvo_ctx *ctx, const unsigned long long *d_cand, const int *d_cnt, const unsigned *d_max, double quality,
                       unsigned long long *d_out, int *d_out_cnt, size_t cand_stride, size_t out_stride, int n_lists, int n_sup,
                       char *d_work, int *h_flag, bool *sorted, int *d_redo, int max_total, char *d_zero, CellCount cells) {
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

} // namespace ktlvo
