// Pyramidal Lucas-Kanade for gfx950: the LKTrackerInvoker of cv::calcOpticalFlowPyrLK as the
// reference calls it (featureTracking.cpp:205-209: Size(21,21), maxLevel 4,
// TermCriteria(COUNT|EPS, 30, 0.01), flags 0, minEigThreshold 1e-4).
//
// One 64-lane wavefront tracks one feature through all pyramid levels.  The 21x21 window is cut
// into 63 horizontal runs of 7 pixels (21 rows x 3 runs); lane l owns run (row l / 3,
// columns 7*(l % 3) ..+6), lane 63 idles.  A lane keeps its seven template values I, Ix, Iy in
// registers for the whole level, so the template never touches LDS; per Gauss-Newton step a
// lane fetches two 8-byte runs of the next image (rows r and r+1) and forms its seven bilinear
// samples from them with v_perm_b32 + v_dot2 (two taps per instruction).  The 2x2 tensor and the
// mismatch vector are summed across the wave with DPP row operations.  Derivatives come from the
// Scharr planes built with the pyramid (int16 4 dx | 4 dy per pixel, zero outside the level).
//
// The kernel is VALU-bound (rocprof: VALU busy > 95 %), so the design minimises instructions per
// pixel and per iteration rather than bytes: images and planes stay L2-resident.
//
// Arithmetic follows OpenCV's fixed-point scheme exactly (14-bit bilinear weights, CV_DESCALE,
// int16 patch values, 2^-20 scale); only the accumulation order of the float sums differs.
#include "ctx.hpp"

#include <cstring>

namespace ktlvo {

// Termination and acceptance criteria of a track, as the launcher prepares them; every kernel argument struct
// carries these fields under these names (lk_track reads them from whichever struct it is given).
#define KTLVO_LK_CRITERIA                                                                                       \
    int max_level;                                                                                              \
    int max_count;                                                                                              \
    double eps_sq;                                                                                              \
    float eps_lo, eps_hi; /* float step^2 at or below eps_lo passes, at or above eps_hi fails, without doubles */ \
    double min_eig_thr;                                                                                         \
    float min_eig_lo, min_eig_hi; /* float minEig estimates outside [lo, hi] decide without the exact value */
struct LkArgs {
    const SlotDesc *slots;
    const float *prev;
    float *next;
    uint8_t *status;
    float *err;
    int n_per_pair;
    int n_pairs;
    int blocks_per_pair; // ceil(n_per_pair / LK_WPB)
    KTLVO_LK_CRITERIA
    const int *d_n; // optional: actual point count lives in device memory (<= n_per_pair)
    PairList pairs;
    int prev_off[MAX_BATCH]; // first point of pair i in prev[]
    int next_off[MAX_BATCH]; // first point of pair i in next[], status[], err[]
};

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) uint32_t *gptr32; // planes live in global memory

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // v + (v moved by the DPP pattern); lanes outside ROW_MASK add 0
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(moved);
}

// Sum over the 64 lanes, returned wave-uniform.  Fixed tree, so the result is deterministic.
// The two cross-row steps are written as asm: v_add_f32_dpp with a row mask leaves the other rows'
// registers untouched, which is the fused form of "add the broadcast lane, or add 0".  The s_nops
// are the VALU-write -> DPP-read / readlane wait states, which the compiler does not insert for asm.
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v); // row_half_mirror
    v = dpp_add<0x140, 0xF>(v); // row_mirror: every lane holds its 16-lane row sum
    int total;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa\n\t" // rows 1, 3 += lane 15 of rows 0, 2
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc\n\t" // rows 2, 3 += lane 31
        "s_nop 1\n\t"
        "v_readlane_b32 %0, %1, 63\n\t"
        "s_nop 1" // before a VALU read of the SGPR
        : "=s"(total), "+v"(v));
    return __int_as_float(total);
}

// Two sums at once: v_permlane32_swap puts a's two halves side by side in one register pair and b's
// in the other, so one add folds both to 32 lanes and the remaining tree runs on a single register
// (lanes 0-31 carry a, lanes 32-63 carry b).
struct Sum2 {
    float a, b;
};
// The operands are exact integer partial sums of at most 2^30 in magnitude, so the first fold (two
// lanes) is done in integers and only its result is converted: one conversion instead of two, and
// one rounding fewer.
__device__ __forceinline__ Sum2 wave_sum2(int a, int b) {
    const auto sw = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
    float v = (float)(int)(sw[0] + sw[1]);
    v = dpp_add<0xB1, 0xF>(v);
    v = dpp_add<0x4E, 0xF>(v);
    v = dpp_add<0x141, 0xF>(v);
    v = dpp_add<0x140, 0xF>(v);
    int ta, tb;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa\n\t" // rows 1, 3 += lane 15 of rows 0, 2
        "s_nop 1\n\t"
        "v_readlane_b32 %0, %2, 31\n\t"
        "v_readlane_b32 %1, %2, 63\n\t"
        "s_nop 1" // before a VALU read of the SGPRs
        : "=&s"(ta), "=&s"(tb), "+v"(v));
    Sum2 r;
    r.a = __int_as_float(ta);
    r.b = __int_as_float(tb);
    return r;
}

// The same two sums EXACTLY: the integer total of each, rounded to float once (to nearest even), whatever the
// order of addition -- what a serial int64 accumulator followed by one conversion yields.  Each lane's operand
// is below 2^29 in magnitude (7 products of two values of at most 8161), the total of 64 of them below 2^35,
// which no 32-bit register holds; but split into its low 16 bits L and the rest H = p - L (a multiple of
// 2^16), either part of the total is exact in a float: sum L < 2^22, sum H a multiple of 2^16 below 2^35.
// After the integer fold of lane pairs (as above) v_permlane16_swap puts the L and H parts of a's 32 lanes
// into rows 0 and 1 and those of b's into rows 2 and 3, folding once more on the way (integers still: two H
// parts stay below 2^31), one conversion and the row tree of wave_sum2 add up each row exactly, and the
// row_bcast:15 add that joined two row sums there joins sum L and sum H here: the single rounding.
// Four instructions more than wave_sum2.
__device__ __forceinline__ Sum2 wave_sum2_exact(int a, int b) {
    const auto sw = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
    const uint32_t p = sw[0] + sw[1];
    uint32_t lo, hi;
    asm("v_and_b32 %0, %2, %3\n\t"
        "v_and_b32 %1, %2, %4"
        : "=&v"(lo), "=&v"(hi)
        : "v"(p), "s"(0xFFFFu), "s"(0xFFFF0000u));
    // rows of lo: [l0 l1 l2 l3], of hi: [h0 h1 h2 h3] -> [l0 h0 l2 h2], [l1 h1 l3 h3]
    const auto sx = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
    float v = (float)(int)(sx[0] + sx[1]);
    v = dpp_add<0xB1, 0xF>(v);
    v = dpp_add<0x4E, 0xF>(v);
    v = dpp_add<0x141, 0xF>(v);
    v = dpp_add<0x140, 0xF>(v);
    int ta, tb;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa\n\t" // rows 1, 3 (sum H) += lane 15 of rows 0, 2 (sum L)
        "s_nop 1\n\t"
        "v_readlane_b32 %0, %2, 31\n\t"
        "v_readlane_b32 %1, %2, 63\n\t"
        "s_nop 1" // before a VALU read of the SGPRs
        : "=&s"(ta), "=&s"(tb), "+v"(v));
    Sum2 r;
    r.a = __int_as_float(ta);
    r.b = __int_as_float(tb);
    return r;
}

// The int16-pair dot products (v_dot2_i32_i16: a.lo * b.lo + a.hi * b.hi + c) are written as asm
// blocks, because the builtin is selected as v_dot2c (dst == accumulator) and then costs a v_mov per
// use whenever the accumulator input has to survive, as the template does here.  gfx950 needs three
// wait states between a dot instruction's write and any other kind of VALU instruction touching that
// register (a following dot may take it as its accumulator at once), and the compiler does not see
// into asm: so every block keeps its dots together and ends with s_nop 2, after which its results
// are ordinary registers.

// Bilinear weights as OpenCV forms them: cvRound of float products scaled by 2^14, the fourth
// as the remainder.  They are packed per COLUMN: wl = (w00 | w10 << 16) weighs the left pixel of
// the top and bottom row, wr = (w01 | w11 << 16) the right one.  A sample is then
// dot2(col[i], wl) + dot2(col[i+1], wr) with col[k] = (top[k], bottom[k]), so one vertical pixel
// pair serves two neighbouring samples (8 v_perm per run instead of 14).
struct Weights {
    uint32_t wl, wr;
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool NEGATED = false>
__device__ __forceinline__ Weights bilinear_weights(f32x2 frac) {
    // frac = (a, b) = (x - floor(x), y - floor(y)) as OpenCV forms them;
    // iw00 = cvRound((1-a)*(1-b)*2^14), iw01 = cvRound(a*(1-b)*2^14), iw10 = cvRound((1-a)*b*2^14).
    // Scaling by 2^14 is exact and commutes with the roundings of 1 - b
    // and of the products, so it is applied to b first; adding 2^23 rounds a float in [0, 2^23) to the
    // nearest-even integer and leaves it in the low mantissa bits.  Written on float pairs so that the
    // compiler emits packed instructions: X = (a, b 2^14), Y = (1 - a, (1 - b) 2^14), and one
    // v_pk_mul_f32 with crossed halves yields (a * Y.y, X.y * (1 - a)).
    // NEGATED yields the four weights with their signs flipped at the same cost: the product is subtracted
    // from 1.5 * 2^23 (floats in [2^23, 2^24) are the integers, and ties-to-even is symmetric), which leaves
    // -iw in the low 16 mantissa bits.
    const f32x2 scale = {1.f, 16384.f};
    const float M = NEGATED ? 12582912.f : 8388608.f;
    const uint32_t MBITS = NEGATED ? 0x4B400000u : 0x4B000000u;
    const f32x2 magic = {M, M};
    const f32x2 X = frac * scale;
    const f32x2 Y = scale - X;
    const f32x2 prod = __builtin_shufflevector(X, X, 0, 1) * __builtin_shufflevector(Y, Y, 1, 0);
    const f32x2 cross = NEGATED ? magic - prod : prod + magic;
    const float p00 = Y.x * Y.y;
    const uint32_t m00 = __float_as_uint(NEGATED ? M - p00 : p00 + M);
    const uint32_t m01 = __float_as_uint(cross.x), m10 = __float_as_uint(cross.y);
    // +-(16384 - iw00 - iw01 - iw10) in the low 16 bits (the exponent fields cancel mod 2^32)
    uint32_t m3; // m00 + m01 + m10 in one instruction; the s_nop covers the packed result read by asm
    asm("s_nop 0\n\tv_add3_u32 %0, %1, %2, %3" : "=v"(m3) : "v"(m00), "v"(m01), "v"(m10));
    const uint32_t m11 = (NEGATED ? 3u * MBITS - 16384u : 16384u + 3u * MBITS) - m3;
    Weights w;
    w.wl = __builtin_amdgcn_perm(m10, m00, 0x05040100u);
    w.wr = __builtin_amdgcn_perm(m11, m01, 0x05040100u);
    return w;
}

template <bool NEGATED = false>
__device__ __forceinline__ Weights bilinear_weights(float x, float y, float floor_x, float floor_y) {
    return bilinear_weights<NEGATED>(f32x2{x, y} - f32x2{floor_x, floor_y});
}

// Eight consecutive bytes starting at byte offset `off` of a plane whose base is dword aligned.
struct Run8 {
    uint32_t lo, hi;
};
__device__ __forceinline__ Run8 load_run8(const uint8_t *base, uint32_t off) {
    typedef const __attribute__((address_space(1))) uint8_t *gptr8;
    gptr32 q = (gptr32)((gptr8)base + (off & ~3u));
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    Run8 r;
    r.lo = __builtin_amdgcn_alignbyte(d1, d0, off & 3u);
    r.hi = __builtin_amdgcn_alignbyte(d2, d1, off & 3u);
    return r;
}
// The same in two steps, so that the fetch can be issued long before its data is needed.
struct Raw12 {
    uint32_t d0, d1, d2;
};
__device__ __forceinline__ Raw12 fetch_run8(const uint8_t *base, uint32_t off) {
    typedef const __attribute__((address_space(1))) uint8_t *gptr8;
    gptr32 q = (gptr32)((gptr8)base + (off & ~3u));
    Raw12 r;
    r.d0 = q[0]; r.d1 = q[1]; r.d2 = q[2];
    return r;
}
__device__ __forceinline__ Run8 align_run8(const Raw12 &w, uint32_t off) {
    Run8 r;
    r.lo = __builtin_amdgcn_alignbyte(w.d1, w.d0, off & 3u);
    r.hi = __builtin_amdgcn_alignbyte(w.d2, w.d1, off & 3u);
    return r;
}
// byte K of the top run and byte K of the bottom run, zero-extended into an int16 pair
template <int K>
__device__ __forceinline__ uint32_t col_pair(const Run8 &t, const Run8 &b) {
    constexpr uint32_t sel = 0x0c000c00u | (uint32_t)(K & 3) | ((uint32_t)((K & 3) + 4) << 16);
    return K < 4 ? __builtin_amdgcn_perm(b.lo, t.lo, sel) : __builtin_amdgcn_perm(b.hi, t.hi, sel);
}
// The eight vertical pixel pairs of a lane's two fetched rows.
struct Cols {
    uint32_t c[8];
};
__device__ __forceinline__ Cols col_pairs(const Run8 &t, const Run8 &b) {
    Cols k;
    k.c[0] = col_pair<0>(t, b); k.c[1] = col_pair<1>(t, b); k.c[2] = col_pair<2>(t, b); k.c[3] = col_pair<3>(t, b);
    k.c[4] = col_pair<4>(t, b); k.c[5] = col_pair<5>(t, b); k.c[6] = col_pair<6>(t, b); k.c[7] = col_pair<7>(t, b);
    return k;
}
// The seven fixed-point bilinear samples (scaled by 32) of a lane's run: v[i] = CV_DESCALE(sum, 9)
// + (acc0[i] - 256) / 512.  With acc0 = 256 this is the sample itself; with acc0 = 256 - 512 * I it is
// sample - I at no extra cost (the subtrahend is a multiple of 2^9, so it commutes with the shift).
// Signed dots: the remainder weight w11 can be -1 when the three rounded weights sum to 16385.
__device__ __forceinline__ void sample_sums(const Cols &k, const Weights &w, const int (&acc0)[7], int (&d)[7]) {
    int d0, d1, d2, d3, d4, d5, d6;
    asm("v_dot2_i32_i16 %0, %7, %15, %17\n\t"
        "v_dot2_i32_i16 %1, %8, %15, %18\n\t"
        "v_dot2_i32_i16 %2, %9, %15, %19\n\t"
        "v_dot2_i32_i16 %3, %10, %15, %20\n\t"
        "v_dot2_i32_i16 %4, %11, %15, %21\n\t"
        "v_dot2_i32_i16 %5, %12, %15, %22\n\t"
        "v_dot2_i32_i16 %6, %13, %15, %23\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %16, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %16, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %16, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %16, %5\n\t"
        "v_dot2_i32_i16 %6, %14, %16, %6\n\t"
        "s_nop 2"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6)
        : "v"(k.c[0]), "v"(k.c[1]), "v"(k.c[2]), "v"(k.c[3]), "v"(k.c[4]), "v"(k.c[5]), "v"(k.c[6]), "v"(k.c[7]),
          "v"(w.wl), "v"(w.wr), "v"(acc0[0]), "v"(acc0[1]), "v"(acc0[2]), "v"(acc0[3]), "v"(acc0[4]), "v"(acc0[5]),
          "v"(acc0[6]));
    d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = d4; d[5] = d5; d[6] = d6;
}
__device__ __forceinline__ void sample_run(const Cols &k, const Weights &w, const int (&acc0)[7], int (&v)[7]) {
    int d[7];
    sample_sums(k, w, acc0, d);
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = d[i] >> 9;
}
// The same with one accumulator start for all seven samples (a rounding constant) and the descale
// shift as a parameter.
__device__ __forceinline__ void sample_sums(const Cols &k, const Weights &w, int acc0, int (&d)[7]) {
    int d0, d1, d2, d3, d4, d5, d6;
    asm("v_dot2_i32_i16 %0, %7, %15, %17\n\t"
        "v_dot2_i32_i16 %1, %8, %15, %17\n\t"
        "v_dot2_i32_i16 %2, %9, %15, %17\n\t"
        "v_dot2_i32_i16 %3, %10, %15, %17\n\t"
        "v_dot2_i32_i16 %4, %11, %15, %17\n\t"
        "v_dot2_i32_i16 %5, %12, %15, %17\n\t"
        "v_dot2_i32_i16 %6, %13, %15, %17\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %16, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %16, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %16, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %16, %5\n\t"
        "v_dot2_i32_i16 %6, %14, %16, %6\n\t"
        "s_nop 2"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6)
        : "v"(k.c[0]), "v"(k.c[1]), "v"(k.c[2]), "v"(k.c[3]), "v"(k.c[4]), "v"(k.c[5]), "v"(k.c[6]), "v"(k.c[7]),
          "v"(w.wl), "v"(w.wr), "s"(acc0)); // the shared start value rides in a scalar register
    d[0] = d0; d[1] = d1; d[2] = d2; d[3] = d3; d[4] = d4; d[5] = d5; d[6] = d6;
}
template <int SHIFT>
__device__ __forceinline__ void sample_run(const Cols &k, const Weights &w, int acc0, int (&v)[7]) {
    int d[7];
    sample_sums(k, w, acc0, d);
#pragma unroll
    for (int i = 0; i < 7; i++) v[i] = d[i] >> SHIFT;
}
// int16 pairs (v0, v1), (v2, v3), (v4, v5), (v6, 0) of seven values that fit 16 bits
struct Packed7 {
    uint32_t p[4];
};
__device__ __forceinline__ Packed7 pack7(const int (&v)[7]) {
    Packed7 r;
    r.p[0] = __builtin_amdgcn_perm((uint32_t)v[1], (uint32_t)v[0], 0x05040100u);
    r.p[1] = __builtin_amdgcn_perm((uint32_t)v[3], (uint32_t)v[2], 0x05040100u);
    r.p[2] = __builtin_amdgcn_perm((uint32_t)v[5], (uint32_t)v[4], 0x05040100u);
    r.p[3] = (uint32_t)v[6] & 0xFFFFu;
    return r;
}
// pack7 of (d[i] >> 9) for sums |d[i]| < 2^23: bytes 1-2 of a sum are d >> 8 as an exact int16, so one
// v_perm_b32 packs two of them and one packed arithmetic shift by 1 finishes both (two instructions
// per pair instead of three).
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Packed7 pack7_descale9(const int (&d)[7]) {
    Packed7 r;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const i16x2 h = __builtin_bit_cast(i16x2, __builtin_amdgcn_perm((uint32_t)d[2 * j + 1], (uint32_t)d[2 * j], 0x06050201u));
        r.p[j] = __builtin_bit_cast(uint32_t, h >> (short)1);
    }
    r.p[3] = ((uint32_t)d[6] >> 9) & 0xFFFFu;
    return r;
}
// pack7 of the upper halves (d[i] >> 16) of seven sums
__device__ __forceinline__ Packed7 pack7_high(const int (&d)[7]) {
    Packed7 r;
#pragma unroll
    for (int j = 0; j < 3; j++) r.p[j] = __builtin_amdgcn_perm((uint32_t)d[2 * j + 1], (uint32_t)d[2 * j], 0x07060302u);
    r.p[3] = (uint32_t)d[6] >> 16;
    return r;
}
// sample_sums followed by pack7_descale9 as one block.  A dot's result may be read by another kind of
// instruction three wait states later; here the packing instructions are ordered so that every one of
// them comes at least that long after the dot it depends on, and the block needs no idle wait at all
// (the last-but-one sum is complete at the 13th dot and first read by the third permute).
__device__ __forceinline__ Packed7 sample_pack(const Cols &k, const Weights &w, const int (&acc0)[7]) {
    uint32_t t0, t1, t2, t3, t4, t5, t6;
    asm("v_dot2_i32_i16 %0, %7, %15, %17\n\t"
        "v_dot2_i32_i16 %1, %8, %15, %18\n\t"
        "v_dot2_i32_i16 %2, %9, %15, %19\n\t"
        "v_dot2_i32_i16 %3, %10, %15, %20\n\t"
        "v_dot2_i32_i16 %4, %11, %15, %21\n\t"
        "v_dot2_i32_i16 %5, %12, %15, %22\n\t"
        "v_dot2_i32_i16 %6, %13, %15, %23\n\t"
        "v_dot2_i32_i16 %0, %8, %16, %0\n\t"
        "v_dot2_i32_i16 %1, %9, %16, %1\n\t"
        "v_dot2_i32_i16 %2, %10, %16, %2\n\t"
        "v_dot2_i32_i16 %3, %11, %16, %3\n\t"
        "v_dot2_i32_i16 %4, %12, %16, %4\n\t"
        "v_dot2_i32_i16 %5, %13, %16, %5\n\t"
        "v_dot2_i32_i16 %6, %14, %16, %6\n\t"
        "v_perm_b32 %0, %1, %0, %24\n\t"
        "v_perm_b32 %2, %3, %2, %24\n\t"
        "v_perm_b32 %4, %5, %4, %24\n\t"
        "v_pk_ashrrev_i16 %0, 1, %0 op_sel_hi:[0,1]\n\t"
        "v_pk_ashrrev_i16 %2, 1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_ashrrev_i16 %4, 1, %4 op_sel_hi:[0,1]\n\t"
        "v_bfe_u32 %6, %6, 9, 16"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6)
        : "v"(k.c[0]), "v"(k.c[1]), "v"(k.c[2]), "v"(k.c[3]), "v"(k.c[4]), "v"(k.c[5]), "v"(k.c[6]), "v"(k.c[7]),
          "v"(w.wl), "v"(w.wr), "v"(acc0[0]), "v"(acc0[1]), "v"(acc0[2]), "v"(acc0[3]), "v"(acc0[4]), "v"(acc0[5]),
          "v"(acc0[6]), "s"(0x06050201u));
    Packed7 r;
    r.p[0] = t0; r.p[1] = t2; r.p[2] = t4; r.p[3] = t6;
    return r;
}
// (sum of x[i] * y[i], sum of x[i] * z[i]) over the seven values, exact in int32
struct Dot2 {
    int xy, xz;
};
__device__ __forceinline__ Dot2 dot7x2(const Packed7 &x, const Packed7 &y, const Packed7 &z) {
    Dot2 r;
    asm("v_dot2_i32_i16 %0, %2, %6, 0\n\t"
        "v_dot2_i32_i16 %1, %2, %10, 0\n\t"
        "v_dot2_i32_i16 %0, %3, %7, %0\n\t"
        "v_dot2_i32_i16 %1, %3, %11, %1\n\t"
        "v_dot2_i32_i16 %0, %4, %8, %0\n\t"
        "v_dot2_i32_i16 %1, %4, %12, %1\n\t"
        "v_dot2_i32_i16 %0, %5, %9, %0\n\t"
        "v_dot2_i32_i16 %1, %5, %13, %1\n\t"
        "s_nop 2"
        : "=&v"(r.xy), "=&v"(r.xz)
        : "v"(x.p[0]), "v"(x.p[1]), "v"(x.p[2]), "v"(x.p[3]), "v"(y.p[0]), "v"(y.p[1]), "v"(y.p[2]), "v"(y.p[3]),
          "v"(z.p[0]), "v"(z.p[1]), "v"(z.p[2]), "v"(z.p[3]));
    return r;
}
__device__ __forceinline__ int dot7(const Packed7 &x, const Packed7 &y) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %5, 0\n\t"
        "v_dot2_i32_i16 %0, %2, %6, %0\n\t"
        "v_dot2_i32_i16 %0, %3, %7, %0\n\t"
        "v_dot2_i32_i16 %0, %4, %8, %0\n\t"
        "s_nop 2"
        : "=&v"(r)
        : "v"(x.p[0]), "v"(x.p[1]), "v"(x.p[2]), "v"(x.p[3]), "v"(y.p[0]), "v"(y.p[1]), "v"(y.p[2]), "v"(y.p[3]));
    return r;
}

constexpr int WIN = 21;
constexpr int RUN = 7;

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Lane 0's (x, y) as scalar registers.  Unlike the builtin this is opaque to the compiler, so the
// values and everything derived from them (bounds tests, row offset) stay on the scalar unit instead
// of being recomputed per lane.  gfx950 wait states the compiler cannot insert around asm: one after
// the VALU write of a VGPR that v_readfirstlane reads, two before a VALU read of the SGPR it wrote.
__device__ __forceinline__ void scalar_xy(int vx, int vy, int &sx, int &sy) {
    asm("s_nop 0\n\t"
        "v_readfirstlane_b32 %0, %2\n\t"
        "v_readfirstlane_b32 %1, %3\n\t"
        "s_nop 1"
        : "=&s"(sx), "=&s"(sy)
        : "v"(vx), "v"(vy));
}

// Waves (= features) per workgroup.  One: a finished feature frees its wave slot at once instead of
// waiting for the slowest of its workgroup (iteration counts differ widely between features), which
// measured 1.8 % faster than four on the same box; nothing is shared between the waves anyway.
#ifndef LK_WPB
#define LK_WPB 1
#endif
// One feature's track through all levels, by one wavefront: prev point (ppx, ppy) in the image of slot SP, result
// (ox, oy), status and error as calcOpticalFlowPyrLK returns them for that point.  All arguments and results are
// wave-uniform.
// MAXL >= 0 fixes the top level at compile time, and the level loop is then unrolled (level-dependent scalars
// become constants; 0.5 % faster at five times the code); MAXL < 0 takes it from the criteria.
struct LkResult {
    float x, y, err;
    int status;
};
// EXACT: the five window sums of a level (A11, A12, A22, b1, b2) are the exact integer totals rounded once
// (wave_sum2_exact) instead of float sums in the order of the DPP tree; the error sum is exact either way
// (its total stays below 2^24).
template <int MAXL, bool EXACT, class Args>
__device__ __forceinline__ LkResult lk_track(const Args &cr, const int max_level, const SlotDesc &SP, const SlotDesc &SN,
                                             const float ppx, const float ppy, const int lane) {
    const bool active = lane < 63;
    // Row-major: the three lanes of a window row are neighbours, so the four lanes the memory pipeline
    // takes together touch two cache lines instead of four (1.8 % faster than run-major on the same box).
    const int r = active ? lane / 3 : 0;
    const int c = active ? lane - 3 * r : 0;
    const int cx = c * RUN;
    const float halfWin = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);

    uint32_t k255;
    asm("v_mov_b32 %0, 0xff" : "=v"(k255));
    float ox = 0.f, oy = 0.f; // nextPts[i]
    int status = 1;
    float errv = 0.f;

#pragma unroll
    for (int level = max_level; level >= 0; --level) {
        const LevelDesc I = SP.lv[level];
        const LevelDesc J = SN.lv[level];
        const int pitch = I.pitch; // both slots share the level geometry
        // plane bases (element (-PAD,-PAD)) and this lane's run offset from there: offsets stay >= 0
        const int org = PAD * pitch + PAD;
        const uint8_t *Ibase = I.origin - org, *Jbase = J.origin - org;
        const uint32_t *Dbase = I.dorigin - org;
        const uint32_t lane_off = (uint32_t)(org + r * pitch + cx);

        const float sc = __int_as_float((127 - level) << 23); // (float)(1./(1 << level))
        float px = ppx * sc, py = ppy * sc;
        if (level == max_level) { ox = px; oy = py; }
        else { ox = ox * 2.f; oy = oy * 2.f; }

        px -= halfWin; py -= halfWin;
        const float fpx = floorf(px), fpy = floorf(py);
        int ipx, ipy;
        scalar_xy((int)fpx, (int)fpy, ipx, ipy);
        if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
            if (level == 0) { status = 0; errv = 0.f; }
            continue;
        }
        Weights w = bilinear_weights(px, py, fpx, fpy);

        // The first step's search rows do not depend on the template, only on the position handed
        // down from the level above: their fetch goes out together with the template's, which saves
        // one dependent memory round trip per level.
        float nx = ox - halfWin, ny = oy - halfWin;
        float flx = floorf(nx), fly = floorf(ny);
        int inx, iny;
        scalar_xy((int)flx, (int)fly, inx, iny);
        const bool first_inside = !(inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h);
        // (unconditional, from a corner clamped into the accepted range when the real one is outside:
        // a conditional fetch would need its registers merged with defaults, which waits for the data)
        const int cjx = min(max(inx, -WIN), J.w - 1), cjy = min(max(iny, -WIN), J.h - 1);
        uint32_t je = lane_off + (uint32_t)(cjy * pitch + cjx);
        Raw12 jt = fetch_run8(Jbase, je), jb = fetch_run8(Jbase, je + pitch);

        // ---- template: image rows ipy+r, +1 and Scharr rows ipy+r, +1, columns ipx+cx .. +7 -----
        // 512 * I + 255: accumulator start of the mismatch samples, which are summed with NEGATED weights:
        // (512 I + 255 - sum) >> 9 = I - CV_DESCALE(sum, 9), the mismatch with its sign flipped, and the
        // start value is one v_and_or_b32 away from the template's own sum
        int Ineg[RUN];
        Packed7 ixp, iyp;  // the template's derivatives as int16 pairs
        {
            const uint32_t e0 = lane_off + (uint32_t)(ipy * pitch + ipx);
            const Run8 t = load_run8(Ibase, e0), b = load_run8(Ibase, e0 + pitch);
            // the idle lane reads the plane's zero frame, so its derivatives (and all its sums) are 0
            const uint32_t ed = active ? e0 : 0u;
            gptr32 dt = (gptr32)Dbase + ed, db = (gptr32)Dbase + (ed + pitch);
            uint32_t d0[RUN + 1], d1[RUN + 1];
#pragma unroll
            for (int k = 0; k <= RUN; k++) { d0[k] = dt[k]; d1[k] = db[k]; }
            {
                int Is[RUN]; // 512 * I + remainder
                sample_sums(col_pairs(t, b), w, 256, Is);
#pragma unroll
                for (int i = 0; i < RUN; i++) // (Is[i] & ~511) | 255 (no literals in this encoding: mask in a scalar, 255 in a vector register)
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(Ineg[i]) : "v"(Is[i]), "s"(0xFFFFFE00u), "v"(k255));
            }
            // column pairs (top, bottom) of dx and of dy; each serves two neighbouring samples
            Cols cxp, cyp;
#pragma unroll
            for (int k = 0; k <= RUN; k++) {
                cxp.c[k] = __builtin_amdgcn_perm(d1[k], d0[k], 0x05040100u);
                cyp.c[k] = __builtin_amdgcn_perm(d1[k], d0[k], 0x07060302u);
            }
            // The planes hold 4 dx, 4 dy, so a sum started at 2^15 has CV_DESCALE(sum / 4, 14) in its
            // upper half, and one permute packs two of them.
            static_assert(SCHARR_GAIN == 4, "the descale below assumes planes scaled by four");
            int ixs[RUN], iys[RUN];
            sample_sums(cxp, w, 1 << 15, ixs);
            sample_sums(cyp, w, 1 << 15, iys);
            ixp = pack7_high(ixs);
            iyp = pack7_high(iys);
        }
        const Dot2 sxx = dot7x2(ixp, ixp, iyp);
        const int s11 = sxx.xy, s12 = sxx.xz, s22 = dot7(iyp, iyp);
        const Sum2 sa = EXACT ? wave_sum2_exact(s11, s12) : wave_sum2(s11, s12);
        const float A11 = sa.a * FLT_SCALE, A12 = sa.b * FLT_SCALE;
        const float A22 = (EXACT ? wave_sum2_exact(s22, 0).a : wave_sum((float)s22)) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        // minEig = (A22 + A11 - sqrt((A11 - A22)^2 + 4 A12^2)) / (2 * 21 * 21) against the threshold.
        // The correctly rounded square root and division cost some thirty instructions, so they are
        // only spent on values next to the threshold: the hardware square root (1 ulp) and a
        // reciprocal multiply land within tr * 2^-18 / 882 + 2^-19 |minEig| of the exact float result.
        const float tr = A22 + A11;
        const float disc = (A11 - A22) * (A11 - A22) + 4.f * A12 * A12;
        const float quick = (tr - __builtin_amdgcn_sqrtf(disc)) * (1.f / (float)(2 * WIN * WIN));
        const float slack = tr * (1.f / (262144.f * (float)(2 * WIN * WIN)));
        bool weak;
        if (quick > cr.min_eig_hi + slack) weak = false;
        else if (quick < cr.min_eig_lo - slack) weak = true;
        else weak = (double)((tr - sqrtf(disc)) / (float)(2 * WIN * WIN)) < cr.min_eig_thr;
        if (weak || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        // FLT_SCALE is a power of two, so scaling b1, b2 by it (as OpenCV does before forming the step)
        // commutes with every rounding below and can be applied to D once per level instead
        // (negative: the mismatch sums come with their sign flipped)
        const float Ds = -(D * FLT_SCALE);

        // ---- Gauss-Newton iterations -----------------------------------------------------------
        // The rows fetched for one step serve the following steps as long as the window's integer
        // corner stays put, which is the rule once the first step or two of a level are done: only
        // the weights change then, and the step needs no memory access at all.  Hence two loops: the
        // outer one fetches for a new corner, the inner one steps with the column pairs held fixed.
        int have_x = INT_MIN, have_y = INT_MIN;
        Cols cols;
#pragma unroll
        // defined without a fill, read only once fetched; the index operand keeps the eight statements
        // distinct (identical ones are merged and then copied apart again), and they are not volatile, which
        // would count as a store and turn the scalar loads of the level descriptors into vector loads
        for (int k = 0; k < 8; k++) asm("; cols %1" : "=v"(cols.c[k]) : "i"(k));
        int j = 0;
        bool fetched = first_inside; // the rows for (inx, iny) are in jt, jb already
        bool stepped = false;        // nextPts[i] is only rewritten by a step
        float backx = 0.f, backy = 0.f; // half of the last step, taken back when the steps alternate
        // One step: reads the previous step (qx, qy), leaves its own in (dx, dy).  Returns 0 to go on
        // with the same rows, 1 when the level is finished, 2 when the window corner has moved.
        // frac = (nx - flx, ny - fly) travels from step to step: it is the next step's weight input, and it
        // doubles as the "has the window corner moved?" test.  n - floor_old lies in [0, 1) exactly when
        // floor(n) is still floor_old, except that a difference just below 1 may round up to 1; that rare
        // case takes the slow path, finds the corner unchanged and goes on with the rounded value, which
        // is what OpenCV's own subtraction yields.  For floats, 0 <= f < 1 is one unsigned compare of the
        // bit pattern against that of 1.0f (negative values have the top bit set).
        f32x2 frac = f32x2{nx, ny} - f32x2{flx, fly};
        auto gn_step = [&](const float qx, const float qy, float &dx, float &dy) __attribute__((always_inline)) -> int {
            w = bilinear_weights<true>(frac);
            const Packed7 dp = sample_pack(cols, w, Ineg); // |sum| <= 255 * 32 * 512 + 511, |diff| <= 255 * 32
            const Dot2 bb = dot7x2(dp, ixp, iyp);
            const Sum2 bs = EXACT ? wave_sum2_exact(bb.xy, bb.xz) : wave_sum2(bb.xy, bb.xz);
            dx = (A12 * bs.b - A22 * bs.a) * Ds;
            dy = (A12 * bs.a - A11 * bs.b) * Ds;
            nx += dx; ny += dy;
            // dx*dx + dy*dy <= eps^2 is decided in double, as OpenCV's arithmetic promotes it; the
            // float sum is within 2^-22 of it, so only values next to the threshold need the doubles
            const float step_sq = dx * dx + dy * dy;
            if (step_sq < cr.eps_hi) {
                if (step_sq <= cr.eps_lo || (double)dx * dx + (double)dy * dy <= cr.eps_sq) return 1;
            }
            // OpenCV compares |.| with the double 0.01; float(0.01) is the largest float below it,
            // so "<= 0.01f" is the same predicate without leaving single precision
            // (before a level's first step the "previous step" is a huge value, so the test fails there
            // without asking for j > 0; the steps are finite, so the maximum of the two magnitudes
            // decides both comparisons at once)
            const f32x2 back = f32x2{dx, dy} + f32x2{qx, qy};
            if (fmaxf(fabsf(back.x), fabsf(back.y)) <= 0.01f) {
                backx = dx * 0.5f; backy = dy * 0.5f;
                return 1;
            }
            if (++j >= cr.max_count) return 1;
            frac = f32x2{nx, ny} - f32x2{flx, fly};
            if (__float_as_uint(frac.x) < 0x3F800000u && __float_as_uint(frac.y) < 0x3F800000u) return 0;
            flx = floorf(nx); fly = floorf(ny);
            frac = f32x2{nx, ny} - f32x2{flx, fly};
            int kx, ky;
            scalar_xy((int)flx, (int)fly, kx, ky);
            if (kx != inx || ky != iny) { // new corner: bounds test and fetch in the outer loop
                inx = kx; iny = ky;
                return 2;
            }
            return 0;
        };
        // The step before the current one lives in (p1x, p1y) whenever the inner loop is entered; the
        // loop is unrolled twice with the two register pairs changing roles, so that no step copies
        // its result into "previous".
        float p0x, p0y, p1x = 1e30f, p1y = 1e30f;
        if (cr.max_count > 0) {
            for (;;) {
                if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                    if (level == 0) status = 0;
                    break;
                }
                if (!fetched) {
                    je = lane_off + (uint32_t)(iny * pitch + inx);
                    jt = fetch_run8(Jbase, je);
                    jb = fetch_run8(Jbase, je + pitch);
                }
                fetched = false;
                cols = col_pairs(align_run8(jt, je), align_run8(jb, je)); // pitch is a multiple of 4: both rows shift alike
                have_x = inx; have_y = iny;
                stepped = true;
                for (;;) {
                    int rc = gn_step(p1x, p1y, p0x, p0y);
                    if (rc == 1) goto level_done;
                    if (rc == 2) {
                        p1x = p0x; p1y = p0y;
                        break;
                    }
                    rc = gn_step(p0x, p0y, p1x, p1y);
                    if (rc == 1) goto level_done;
                    if (rc == 2) break;
                }
            }
        }
    level_done:
        if (stepped) { // nextPts[i] = nextPt + halfWin, then nextPts[i] -= delta * 0.5f where the steps alternated
            ox = (nx + halfWin) - backx;
            oy = (ny + halfWin) - backy;
        }

        // ---- final error (level 0 only) ------------------------------------------------------
        if (level == 0 && status) {
            const float fx = ox - halfWin, fy = oy - halfWin;
            const int inx = uniform((int)floorf(fx)), iny = uniform((int)floorf(fy));
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                status = 0;
            } else {
                w = bilinear_weights<true>(fx, fy, (float)inx, (float)iny);
                if (inx != have_x || iny != have_y) {
                    const uint32_t e0 = lane_off + (uint32_t)(iny * pitch + inx);
                    cols = col_pairs(load_run8(Jbase, e0), load_run8(Jbase, e0 + pitch));
                }
                int diff[RUN];
                sample_run(cols, w, Ineg, diff);
                int e = 0;
#pragma unroll
                for (int i = 0; i < RUN; i++) e += diff[i] < 0 ? -diff[i] : diff[i];
                if (!active) e = 0;
                errv = wave_sum((float)e) * 1.f / (float)(32 * WIN * WIN);
            }
        }
    }
    LkResult res;
    res.x = ox; res.y = oy; res.err = errv; res.status = status;
    return res;
}

template <int MAXL, bool EXACT>
__global__ __launch_bounds__(64 * LK_WPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void pyrlk_track_w21(LkArgs a) {
    const int max_level = MAXL >= 0 ? MAXL : a.max_level;
    const int lane = threadIdx.x & 63;
    const int wave = uniform((int)(threadIdx.x >> 6));
    // Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share one), each with its
    // own 4 MiB L2.  Pairs are taken in groups of 8: all workgroups of a group's pair i get block ids
    // of residue class i mod 8, so a pair's two pyramids are fetched into one L2 instead of eight.
    // The pairs left over after the last full group are spread over all XCDs, so none runs dry
    // early.  This is a locality hint only; any placement gives the same results.
    const int grouped = (a.n_pairs >> 3) * 8 * a.blocks_per_pair; // blocks of the full groups
    int pair, blk;
    if ((int)blockIdx.x < grouped) {
        const int q = blockIdx.x >> 3;
        pair = (blockIdx.x & 7) + 8 * (q / a.blocks_per_pair);
        blk = q % a.blocks_per_pair;
    } else {
        const int q = blockIdx.x - grouped;
        pair = (a.n_pairs & ~7) + q / a.blocks_per_pair;
        blk = q % a.blocks_per_pair;
    }
    const int f = blk * LK_WPB + wave;
    const int n = a.d_n ? min(*a.d_n, a.n_per_pair) : a.n_per_pair;
    if (f >= n) return;
    const SlotDesc &SP = a.slots[a.pairs.prev[pair]];
    const SlotDesc &SN = a.slots[a.pairs.next[pair]];
    const size_t gp = (size_t)a.prev_off[pair] + f;
    const size_t gi = (size_t)a.next_off[pair] + f;
    const float ppx = a.prev[2 * gp], ppy = a.prev[2 * gp + 1];

    const LkResult res = lk_track<MAXL, EXACT>(a, max_level, SP, SN, ppx, ppy, lane);
    if (lane == 0) {
        a.next[2 * gi] = res.x;
        a.next[2 * gi + 1] = res.y;
        a.status[gi] = (uint8_t)res.status;
        a.err[gi] = res.err;
    }
}

// The six tracks of a stereo frame (featureTracking.cpp:457-467 and the four calls inside the two getOpticalFlow
// stages, :205-214 as reached from :509 and :513) from ONE launch.  A point's tracks form two chains that
// share nothing: last left -> last right -> {current right, back to last left}, and last left -> current left ->
// {current right, back to last left}.  A workgroup of four waves takes one start point; wave `role` makes the
// first track of its chain and then its own second track, so no wave ever waits for another (the first track
// of each chain is made twice, which costs nothing on a GPU that a frame's points leave mostly idle):
//   role 0: LL->LR (writes B), LR->CR (writes D)      role 1: LL->CL (writes C), CL->CR (writes E)
//   role 2: LL->LR,            LR->LL (writes CHK)    role 3: LL->CL,            CL->LL (writes CHKB)
// Compared with a launch per stage, the frame waits for its slowest chain instead of the slowest track of each
// stage in turn.  Values are those of pyrlk_track_w21 (same lk_track); status and error are not needed by the
// frame's gates and are not stored.
struct ChainArgs {
    const SlotDesc *slots;
    const float *A;
    float *out[6]; // B, C, D, E, CHK, CHKB
    int n;
    int sLL, sLR, sCL, sCR;
    KTLVO_LK_CRITERIA
};
template <int MAXL, bool EXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void pyrlk_frame_chain(ChainArgs a) {
    const int lane = threadIdx.x & 63;
    const int role = uniform((int)(threadIdx.x >> 6));
    const int f = blockIdx.x;
    if (f >= a.n) return;
    const bool stereo = !(role & 1);
    const int mid = stereo ? a.sLR : a.sCL;
    const int last = (role & 2) ? a.sLL : a.sCR;
    float x = a.A[2 * f], y = a.A[2 * f + 1];
    for (int t = 0; t < 2; t++) {
        const SlotDesc &SP = a.slots[t ? mid : a.sLL];
        const SlotDesc &SN = a.slots[t ? last : mid];
        const LkResult res = lk_track<MAXL, EXACT>(a, MAXL, SP, SN, x, y, lane);
        x = res.x; y = res.y;
        float *dst = t ? a.out[2 + role] : (role < 2 ? a.out[role] : nullptr);
        if (lane == 0 && dst) { dst[2 * f] = x; dst[2 * f + 1] = y; }
    }
}

// Fills the criteria fields of a kernel argument struct from the caller's parameters, as OpenCV clamps them.
template <class Args>
static void set_criteria(Args &a, const ktlvo_lk_params *p, int max_level) {
    a.max_level = max_level;
    int mc = p->max_count; // OpenCV clamps TermCriteria the same way
    a.max_count = mc < 0 ? 0 : mc > 100 ? 100 : mc;
    double eps = p->epsilon < 0 ? 0 : p->epsilon > 10 ? 10 : p->epsilon;
    a.eps_sq = eps * eps;
    if (a.eps_sq >= 1e-30) {
        a.eps_lo = (float)(a.eps_sq * (1.0 - 1e-6));
        a.eps_hi = (float)(a.eps_sq * (1.0 + 1e-6));
    } else { // float squares may underflow here: always decide in double
        a.eps_lo = -1.f;
        a.eps_hi = __builtin_inff();
    }
    a.min_eig_thr = p->min_eig_thr;
    if (p->min_eig_thr >= 1e-12 && p->min_eig_thr <= 1e12) {
        a.min_eig_lo = (float)(p->min_eig_thr * (1.0 - 1e-5));
        a.min_eig_hi = (float)(p->min_eig_thr * (1.0 + 1e-5));
    } else { // unusual thresholds (zero, negative, huge, NaN): always take the exact value
        a.min_eig_lo = -__builtin_inff();
        a.min_eig_hi = __builtin_inff();
    }
}

// Which reading of the window sums a call asks for: its own setting, else the context's.
static bool exact_sums(const ktlvo_ctx *ctx, const ktlvo_lk_params *p) {
    const int m = p->sums != KTLVO_LK_SUMS_DEFAULT ? p->sums : ctx->cfg.lk_sums;
    return m == KTLVO_LK_SUMS_EXACT;
}

// d_out: B, C, D, E, CHK, CHKB as in pyrlk_frame_chain, n points each.  Returns KTLVO_ERR_UNSUPPORTED when the four
// slots do not all hold the reference's five levels (the caller then takes the launch-per-stage route).
int launch_frame_chain(ktlvo_ctx *ctx, int sLL, int sLR, int sCL, int sCR, const float *dA, int n, float *const d_out[6],
                       const ktlvo_lk_params *p) {
    if (!p || p->win != WIN) return KTLVO_ERR_UNSUPPORTED;
    const int s[4] = {sLL, sLR, sCL, sCR};
    for (int i = 0; i < 4; i++) {
        if (!slot_ok(ctx, s[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[s[i]] < 0) return KTLVO_ERR_STATE;
        if (ctx->built[s[i]] < p->max_level || p->max_level != 4) return KTLVO_ERR_UNSUPPORTED;
    }
    if (!ctx->deriv_ok[sLL] || !ctx->deriv_ok[sLR] || !ctx->deriv_ok[sCL]) return KTLVO_ERR_STATE; // the template images
    if (n <= 0) return n == 0 ? KTLVO_OK : KTLVO_ERR_INVALID;
    ChainArgs a;
    a.slots = ctx->d_slots;
    a.A = dA;
    for (int i = 0; i < 6; i++) a.out[i] = d_out[i];
    a.n = n;
    a.sLL = sLL; a.sLR = sLR; a.sCL = sCL; a.sCR = sCR;
    set_criteria(a, p, 4);
    ProfScope ps(ctx, KTLVO_K_PYRLK);
    if (exact_sums(ctx, p)) hipLaunchKernelGGL((pyrlk_frame_chain<4, true>), dim3((unsigned)n), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((pyrlk_frame_chain<4, false>), dim3((unsigned)n), dim3(256), 0, ctx->stream, a);
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

int launch_pyrlk(ktlvo_ctx *ctx, const PairList &pairs, int n_pairs, const float *d_prev, int n_per_pair,
                 float *d_next, uint8_t *d_status, float *d_err, const ktlvo_lk_params *p) {
    int po[MAX_BATCH], no[MAX_BATCH];
    for (int i = 0; i < n_pairs && i < MAX_BATCH; i++) po[i] = no[i] = i * n_per_pair;
    return launch_pyrlk_ex(ctx, pairs, n_pairs, d_prev, po, d_next, no, d_status, d_err, n_per_pair, nullptr, p);
}

int launch_pyrlk_ex(ktlvo_ctx *ctx, const PairList &pairs, int n_pairs, const float *d_prev, const int *prev_off,
                    float *d_next, const int *next_off, uint8_t *d_status, float *d_err, int n_per_pair,
                    const int *d_n, const ktlvo_lk_params *p) {
    if (!p) return KTLVO_ERR_INVALID;
    if (p->win != WIN) return KTLVO_ERR_UNSUPPORTED;
    if (n_pairs <= 0 || n_pairs > MAX_BATCH || n_per_pair < 0 || p->max_level < 0) return KTLVO_ERR_INVALID;
    if ((long long)n_pairs * n_per_pair > 0x7fffffffLL) return KTLVO_ERR_CAPACITY; // point indices are 32-bit
    if (n_per_pair == 0) return KTLVO_OK;
    if (!d_prev || !d_next || !d_status || !d_err) return KTLVO_ERR_INVALID;
    int max_level = p->max_level;
    for (int i = 0; i < n_pairs; i++) {
        if (!slot_ok(ctx, pairs.prev[i]) || !slot_ok(ctx, pairs.next[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[pairs.prev[i]] < 0 || ctx->built[pairs.next[i]] < 0) return KTLVO_ERR_STATE;
        if (!ctx->deriv_ok[pairs.prev[i]]) return KTLVO_ERR_STATE; // ktlvo_build_pyramid makes the Scharr planes
        max_level = effective_max_level(ctx, pairs.prev[i], pairs.next[i], max_level);
    }
    for (int i = 0; i < n_pairs; i++) // every pair of one launch runs the same level count
        if (effective_max_level(ctx, pairs.prev[i], pairs.next[i], p->max_level) != max_level) return KTLVO_ERR_STATE;
    LkArgs a;
    a.slots = ctx->d_slots;
    a.prev = d_prev; a.next = d_next; a.status = d_status; a.err = d_err;
    a.n_per_pair = n_per_pair;
    set_criteria(a, p, max_level);
    a.d_n = d_n;
    a.pairs = pairs;
    for (int i = 0; i < n_pairs; i++) { a.prev_off[i] = prev_off[i]; a.next_off[i] = next_off[i]; }
    a.n_pairs = n_pairs;
    a.blocks_per_pair = (n_per_pair + LK_WPB - 1) / LK_WPB;
    dim3 grid((unsigned)(n_pairs * a.blocks_per_pair));
    ProfScope ps(ctx, KTLVO_K_PYRLK);
    // the reference's maxLevel (featureTracking.cpp:208) gets the unrolled kernel
    // (level counts up to the reference's get a kernel of their own, with the level loop unrolled; deeper pyramids
    // share the looping one, which spills a few registers at eight waves per SIMD)
    const dim3 block(64 * LK_WPB);
#define KTLVO_LK_LAUNCH(L)                                                                                  \
    do {                                                                                                    \
        if (exact) hipLaunchKernelGGL((pyrlk_track_w21<L, true>), grid, block, 0, ctx->stream, a);           \
        else hipLaunchKernelGGL((pyrlk_track_w21<L, false>), grid, block, 0, ctx->stream, a);                \
    } while (0)
    const bool exact = exact_sums(ctx, p);
    switch (max_level) {
    case 4: KTLVO_LK_LAUNCH(4); break;
    case 3: KTLVO_LK_LAUNCH(3); break;
    case 2: KTLVO_LK_LAUNCH(2); break;
    case 1: KTLVO_LK_LAUNCH(1); break;
    case 0: KTLVO_LK_LAUNCH(0); break;
    default: KTLVO_LK_LAUNCH(-1); break;
    }
#undef KTLVO_LK_LAUNCH
    KTLVO_HIP(ctx, hipGetLastError());
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

extern "C" int ktlvo_pyrlk_batch_dev(ktlvo_ctx *ctx, const int *prev_slots, const int *next_slots, int n_pairs,
                                     const float *d_prev_xy, int n_per_pair, float *d_next_xy, uint8_t *d_status,
                                     float *d_err, const ktlvo_lk_params *p) {
    if (!ctx || !prev_slots || !next_slots || n_pairs < 0) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n_pairs > ctx->cfg.max_batch_pairs) return KTLVO_ERR_CAPACITY;
    if (n_pairs == 0) return KTLVO_OK;
    PairList pl;
    for (int i = 0; i < n_pairs; i++) {
        if (!slot_ok(ctx, prev_slots[i]) || !slot_ok(ctx, next_slots[i])) return KTLVO_ERR_INVALID; // before narrowing
        pl.prev[i] = (uint16_t)prev_slots[i]; pl.next[i] = (uint16_t)next_slots[i];
    }
    return launch_pyrlk(ctx, pl, n_pairs, d_prev_xy, n_per_pair, d_next_xy, d_status, d_err, p);
}

extern "C" int ktlvo_pyrlk_dev(ktlvo_ctx *ctx, int prev_slot, int next_slot, const float *d_prev_xy, int n,
                               float *d_next_xy, uint8_t *d_status, float *d_err, const ktlvo_lk_params *p) {
    if (!ctx) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (!slot_ok(ctx, prev_slot) || !slot_ok(ctx, next_slot)) return KTLVO_ERR_INVALID;
    PairList pl;
    pl.prev[0] = (uint16_t)prev_slot; pl.next[0] = (uint16_t)next_slot;
    return launch_pyrlk(ctx, pl, 1, d_prev_xy, n, d_next_xy, d_status, d_err, p);
}

extern "C" int ktlvo_pyrlk(ktlvo_ctx *ctx, int prev_slot, int next_slot, const float *prev_xy, int n, float *next_xy,
                           uint8_t *status, float *err, const ktlvo_lk_params *p) {
    if (!ctx || n < 0) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n > ctx->cfg.max_points) return KTLVO_ERR_CAPACITY;
    if (n == 0) {
        if (!p) return KTLVO_ERR_INVALID;
        return p->win == WIN ? KTLVO_OK : KTLVO_ERR_UNSUPPORTED;
    }
    if (!prev_xy || !next_xy || !status || !err) return KTLVO_ERR_INVALID;
    // Results land back to back (positions, errors, status bytes) in the work area, so that they come
    // home in one copy through pinned memory; a copy per array into pageable memory costs more than
    // the kernel for a frame's worth of points.
    const size_t res_bytes = (size_t)n * 13;
    int rc = ensure_misc(ctx, res_bytes);
    if (rc != KTLVO_OK) return rc;
    rc = ensure_pin(ctx, res_bytes);
    if (rc != KTLVO_OK) return rc;
    char *hp = (char *)ctx->h_pin;
    float *d_next = (float *)ctx->d_misc, *d_err = d_next + 2 * (size_t)n;
    uint8_t *d_status = (uint8_t *)(d_err + n);
    memcpy(hp, prev_xy, (size_t)n * 8);
    KTLVO_HIP(ctx, hipMemcpyAsync(ctx->d_pts[0], hp, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = ktlvo_pyrlk_dev(ctx, prev_slot, next_slot, ctx->d_pts[0], n, d_next, d_status, d_err, p);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipMemcpyAsync(hp, ctx->d_misc, res_bytes, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(next_xy, hp, (size_t)n * 8);
    memcpy(err, hp + (size_t)n * 8, (size_t)n * 4);
    memcpy(status, hp + (size_t)n * 12, (size_t)n);
    return KTLVO_OK;
}
