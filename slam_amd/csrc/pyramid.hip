// Image ingest and Gaussian pyramid (cv::buildOpticalFlowPyramid / cv::pyrDown as reached from
// featureTracking.cpp:205) for gfx950.
//
// Layout: each level lives in a padded plane (ctx.hpp: PAD); pyr_down_u8 writes the level's
// interior AND the reflect-101 copies of every pixel that fall into the frame, so the next
// level's loader and the LK kernel read plain rows with no border arithmetic.
#include "ctx.hpp"

#include <cstring>

namespace ktlvo {

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// Ingest: copies a w x h image (row stride `stride`) into level 0 of a slot and fills the reflect-101 frame
// in the same pass.  A thread makes sixteen consecutive bytes of the padded plane (one aligned 16-byte store;
// the plane's origin, pitch and PAD are multiples of sixteen); away from the left and right edges its sixteen
// source bytes are consecutive too and come in one load, at whatever byte address the row starts (global
// loads need no alignment on this target).  Threads are numbered along the rows, `groups` of them per row.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
__device__ __forceinline__ void ingest_16(const LevelDesc &L, const uint8_t *src, int stride, int groups, uint32_t i) {
    typedef const __attribute__((address_space(1))) uint8_t *gsrc;
    typedef __attribute__((address_space(1))) uint8_t *gdst;
    const int yy = (int)(i / (uint32_t)groups);
    const int x = (int)(i - (uint32_t)yy * (uint32_t)groups) * 16 - PAD, y = yy - PAD;
    if (y >= L.h + PAD) return;
    gsrc row = (gsrc)src + (size_t)reflect101(y, L.h) * stride;
    gdst out = (gdst)L.origin + (ptrdiff_t)y * L.pitch + x;
    if (x >= 0 && x + 15 < L.w) {
        *(__attribute__((address_space(1))) u32x4 *)out = *(const __attribute__((address_space(1))) u32x4_a1 *)(row + x);
        return;
    }
    uint32_t v[4] = {0, 0, 0, 0};
    if (L.w > PAD) { // one reflection reaches every column of the frame: no loop per byte
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int c = x + k, m = c < 0 ? -c : c >= L.w ? 2 * L.w - 2 - c : c;
            v[k >> 2] |= (uint32_t)row[m] << (8 * (k & 3));
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k >> 2] |= (uint32_t)row[reflect101(x + k, L.w)] << (8 * (k & 3));
    }
    if (x + 15 < L.w + PAD) {
        *(__attribute__((address_space(1))) u32x4 *)out = u32x4{v[0], v[1], v[2], v[3]};
    } else { // the plane's rows end at w + PAD: what lies beyond stays as it is
        for (int k = 0; k < 16 && x + k < L.w + PAD; k++) out[k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
    }
}
__host__ __device__ inline int ingest_groups(int w) { return (w + 2 * PAD + 15) / 16; }
__global__ __launch_bounds__(256) void ingest_pad_u8(const SlotDesc *slots, int slot, const uint8_t *src, int stride) {
    const LevelDesc L = slots[slot].lv[0];
    ingest_16(L, src, stride, ingest_groups(L.w), blockIdx.x * 256 + threadIdx.x);
}
// The same for a batch: image z of a contiguous device buffer (image_stride bytes apart) goes into slot list.s[z].
__global__ __launch_bounds__(256) void ingest_pad_batch_u8(const SlotDesc *slots, SlotList list, const uint8_t *src,
                                                           size_t image_stride, int stride) {
    const LevelDesc L = slots[list.s[blockIdx.z]].lv[0];
    ingest_16(L, src + blockIdx.z * image_stride, stride, ingest_groups(L.w), blockIdx.x * 256 + threadIdx.x);
}
// A stereo pair from two separate buffers.  The buffers may lie in pinned host memory: the loads then cross the
// host link themselves, which for two frames is quicker than two DMA transfers followed by a launch.
__global__ __launch_bounds__(256) void ingest_pad_pair_u8(const SlotDesc *slots, int slot_a, int slot_b, const uint8_t *src_a,
                                                          const uint8_t *src_b, int stride) {
    const LevelDesc L = slots[blockIdx.z ? slot_b : slot_a].lv[0];
    ingest_16(L, blockIdx.z ? src_b : src_a, stride, ingest_groups(L.w), blockIdx.x * 256 + threadIdx.x);
}
static inline unsigned ingest_blocks(int w, int h) { return (unsigned)(((size_t)ingest_groups(w) * (h + 2 * PAD) + 255) / 256); }

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// Plane pointers come out of a descriptor in memory, so the compiler would treat them as generic
// (flat) addresses; they are global.
typedef __attribute__((address_space(1))) uint8_t *gbyte;
typedef const __attribute__((address_space(1))) uint8_t *gcbyte;
typedef __attribute__((address_space(1))) uint32_t *gword;
// the same sixteen bytes at an address that is only dword aligned (rows start at x - 4 for any x that is a
// multiple of 4): global_load_dwordx4 takes such addresses, and the type says so to the compiler
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

constexpr int PD_ROWS = 4;         // output rows per thread
constexpr int PD_TW = 256;         // output columns per workgroup (64 lanes x 4 pixels)
constexpr int PD_TH = 4 * PD_ROWS; // output rows per workgroup (4 waves)

// Horizontal 1-4-6-4-1 at four even source columns, as two packed uint16 pairs.  d holds the 16
// source bytes starting four bytes left of the first output's centre: output k is centred on byte
// 4 + 2k.  Sums stay below 16 * 255, so uint16 lanes are exact.
__device__ __forceinline__ void pyr_h4(const u32x4 &d, u16x2 &h01, u16x2 &h23) {
    auto pr = [](uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, sel)); };
    // v_perm_b32 selectors: bytes 0-3 of the second operand, 4-7 of the first, 0x0c = zero
    const u16x2 e0 = pr(d.y, d.x, 0x0c040c02u); // (b2, b4)
    const u16x2 e1 = pr(d.y, d.y, 0x0c020c00u); // (b4, b6)
    const u16x2 e2 = pr(d.z, d.y, 0x0c040c02u); // (b6, b8)
    const u16x2 e3 = pr(d.z, d.z, 0x0c020c00u); // (b8, b10)
    const u16x2 e4 = pr(d.w, d.z, 0x0c040c02u); // (b10, b12)
    const u16x2 o0 = pr(d.y, d.x, 0x0c050c03u); // (b3, b5)
    const u16x2 o1 = pr(d.y, d.y, 0x0c030c01u); // (b5, b7)
    const u16x2 o2 = pr(d.z, d.y, 0x0c050c03u); // (b7, b9)
    const u16x2 o3 = pr(d.z, d.z, 0x0c030c01u); // (b9, b11)
    const u16x2 six = {6, 6}, four = {4, 4};
    h01 = e0 + e2 + six * e1 + four * (o0 + o1);
    h23 = e2 + e4 + six * e3 + four * (o2 + o3);
}

// Stores the four pixels v (byte k = pixel x + k) of one row and their reflect-101 copies that fall
// into the left / right frame: pixel x goes to -x (1 <= x <= PAD) and to 2 (w - 1) - x (if that lies
// in [w, w + PAD)).  Whole mirrored quads go out as one byte-reversed word at an address that is
// in general not word aligned (memcpy lets the compiler pick the store), ragged ones bytewise.
__device__ __forceinline__ void pyr_store_row(gbyte row, int x, int w, uint32_t v) {
    const int nvalid = w - x < 4 ? w - x : 4;
    if (nvalid == 4) *(gword)(row + x) = v;
    else for (int k = 0; k < nvalid; k++) row[x + k] = (uint8_t)(v >> (8 * k));
    const uint32_t rev = __builtin_amdgcn_perm(0u, v, 0x00010203u);
    if (x <= PAD) { // left frame: pixels x .. x+3 -> -x-3 .. -x (pixel 0 lands on itself)
        if (nvalid == 4 && x + 3 <= PAD) __builtin_memcpy(row - x - 3, &rev, 4);
        else for (int k = 0; k < nvalid; k++) if (x + k >= 1 && x + k <= PAD) row[-(x + k)] = (uint8_t)(v >> (8 * k));
    }
    const int m = 2 * (w - 1); // right frame: pixel p -> m - p, valid for w - 1 - PAD <= p <= w - 2
    if (x + 3 >= w - 1 - PAD) {
        if (nvalid == 4 && x >= w - 1 - PAD && x + 3 <= w - 1) __builtin_memcpy(row + m - x - 3, &rev, 4); // pixel w-1 lands on itself
        else for (int k = 0; k < nvalid; k++) {
            const int p = x + k;
            if (p <= w - 2 && m - p < w + PAD) row[m - p] = (uint8_t)(v >> (8 * k));
        }
    }
}

// One level of cv::pyrDown: dst(x,y) = (sum_{i,j} k_i k_j src(2x+i, 2y+j) + 128) >> 8, k = [1 4 6 4 1].
// The source frame already holds the reflect-101 continuation, which is what pyrDown's own
// borderInterpolate yields.  A thread makes a 4 x PD_ROWS block of outputs: it walks down
// 2 * PD_ROWS + 3 source rows (one 16-byte load each), keeps their horizontal sums in registers as
// packed uint16 pairs and combines five of them per output row, so nothing goes through LDS.  Each
// output row is also written to the rows of the top / bottom frame it is mirrored to.
__device__ __forceinline__ void pyr_down_block(const LevelDesc &src, const LevelDesc &dst, int ox, int oy0) {
    if (ox >= dst.w || oy0 >= dst.h) return;

    constexpr int NS = 2 * PD_ROWS + 3;
    u16x2 h01[NS], h23[NS];
    const int last_row = src.h + PAD - 1; // source rows past the level's last output row are not used: keep them inside the plane
#pragma unroll
    for (int r = 0; r < NS; r++) {
        int sy = 2 * oy0 - 2 + r;
        sy = sy < last_row ? sy : last_row;
        const u32x4 d = *(const __attribute__((address_space(1))) u32x4_a4 *)((gcbyte)src.origin + (ptrdiff_t)sy * src.pitch + 2 * ox - 4);
        pyr_h4(d, h01[r], h23[r]);
    }
    const u16x2 six = {6, 6}, four = {4, 4}, half = {128, 128};
#pragma unroll
    for (int j = 0; j < PD_ROWS; j++) {
        const int oy = oy0 + j;
        if (oy >= dst.h) break;
        // vertical sums stay below 256 * 255
        const u16x2 v01 = (h01[2 * j] + h01[2 * j + 4] + six * h01[2 * j + 2] + four * (h01[2 * j + 1] + h01[2 * j + 3]) + half) >> 8;
        const u16x2 v23 = (h23[2 * j] + h23[2 * j + 4] + six * h23[2 * j + 2] + four * (h23[2 * j + 1] + h23[2 * j + 3]) + half) >> 8;
        const uint32_t v = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, v23), __builtin_bit_cast(uint32_t, v01), 0x06040200u);
        const gbyte org = (gbyte)dst.origin;
        pyr_store_row(org + (ptrdiff_t)oy * dst.pitch, ox, dst.w, v);
        if (oy >= 1 && oy <= PAD) pyr_store_row(org - (ptrdiff_t)oy * dst.pitch, ox, dst.w, v);
        const int my = 2 * (dst.h - 1) - oy;
        if (oy <= dst.h - 2 && my < dst.h + PAD) pyr_store_row(org + (ptrdiff_t)my * dst.pitch, ox, dst.w, v);
    }
}
__global__ __launch_bounds__(256) void pyr_down_u8(const SlotDesc *slots, SlotList list, int src_level) {
    const SlotDesc &S = slots[list.s[blockIdx.z]];
    pyr_down_block(S.lv[src_level], S.lv[src_level + 1], blockIdx.x * PD_TW + (threadIdx.x & 63) * 4,
                   blockIdx.y * PD_TH + (threadIdx.x >> 6) * PD_ROWS);
}

// The last two levels of a large batch from ONE launch: a workgroup per image forms level first + 1 from level
// `first` with pyr_down_u8's own blocks (a 4 x 4 block of outputs per thread, as many rounds as the level needs),
// waits for its waves, and forms level first + 2 from what it has just written.  At these sizes (a few thousand
// pixels per image) a launch per level is all latency: descriptor loads, one round of row loads, the stores and the
// launch itself, some 11 us each for a microsecond of arithmetic.  The second level reads what other waves of the
// same workgroup stored: they share the compute unit's L1, which held none of these lines before, and the barrier
// waits for the stores, so workgroup scope orders it.
constexpr int PT2_THREADS = 512;
__global__ __launch_bounds__(PT2_THREADS) void pyr_down_tail2(const SlotDesc *slots, SlotList list, int first) {
    const SlotDesc &S = slots[list.s[blockIdx.x]];
    for (int l = first; l < first + 2; l++) {
        const LevelDesc src = S.lv[l], dst = S.lv[l + 1];
        const int bw = (dst.w + 3) / 4, bh = (dst.h + PD_ROWS - 1) / PD_ROWS;
        for (int b = threadIdx.x; b < bw * bh; b += PT2_THREADS) {
            const int by = b / bw, bx = b - by * bw;
            pyr_down_block(src, dst, bx * 4, by * PD_ROWS);
        }
        __syncthreads();
    }
}

// ---- three levels in one launch ---------------------------------------------------------------------------
// Levels first+1, first+2, first+3 from level `first`.  The small levels of a pyramid are a few tens of
// kilobytes each, and a launch per level costs more in launch latency than in work; here a workgroup owns an
// TT x TT tile of the LAST level and forms everything that tile depends on in LDS: for TT = 2 a 37 x 37 region
// of the source level, 17 x 17 of the first new level, 7 x 7 of the second (the halos are recomputed by the
// neighbouring workgroups; the kernel serves small batches, where more and smaller workgroups finish sooner
// and the recomputation costs nothing that matters).  It writes its exclusive part of each level, 4 TT, 2 TT
// and TT pixels square, each with the pixels' reflect-101 copies in the level's frame, as pyr_down_u8 does.
// pyrDown reflects at the edges of EACH source level, so an intermediate level's entries beyond the level are
// not computed from mirrored sources (that differs when the level below has an even size) but copied from
// their mirror images inside the level, which the same region holds.
constexpr int TT = 2;                 // tile of the last level
constexpr int R2 = 2 * TT + 3;        // second new level, coordinates 2 x3 - 2 ...
constexpr int R1 = 2 * R2 + 3;        // first new level, coordinates 4 x3 - 6 ...
constexpr int R0 = 2 * R1 + 3;        // source level, coordinates 8 x3 - 14 ...

// pixel (x, y) of level L and its single reflections into the frame (pyr_store_row's rule, per pixel)
__device__ __forceinline__ void store_mirrored(const LevelDesc &L, int x, int y, uint8_t v) {
    int xs[3], ys[3], nx = 0, ny = 0;
    xs[nx++] = x;
    if (x >= 1 && x <= PAD) xs[nx++] = -x;
    if (x <= L.w - 2 && 2 * (L.w - 1) - x < L.w + PAD) xs[nx++] = 2 * (L.w - 1) - x;
    ys[ny++] = y;
    if (y >= 1 && y <= PAD) ys[ny++] = -y;
    if (y <= L.h - 2 && 2 * (L.h - 1) - y < L.h + PAD) ys[ny++] = 2 * (L.h - 1) - y;
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++) L.origin[(ptrdiff_t)ys[j] * L.pitch + xs[i]] = v;
}

// One pyrDown step inside LDS.  src: SR x SR bytes whose entry (0, 0) is source coordinate (2 ox - 2, 2 oy - 2);
// dst: DR x DR bytes whose entry (0, 0) is destination coordinate (ox, oy).  Entries whose coordinate lies in the
// destination level are computed (separably, through `tmp`), then the two rows and columns beyond each edge are
// copied from their mirror images.
template <int SR, int DR>
__device__ __forceinline__ void lds_pyr_step(const uint8_t (*src)[SR + 3], uint8_t (*dst)[DR + 3], unsigned short *tmp,
                                             int ox, int oy, int dw, int dh) {
    const int tid = threadIdx.x;
    // horizontal: tmp[r][c] for source row r, destination column c
    for (int i = tid; i < SR * DR; i += 256) {
        const int r = i / DR, c = i - r * DR;
        const uint8_t *p = &src[r][2 * c];
        tmp[i] = (unsigned short)(p[0] + p[4] + 4 * (p[1] + p[3]) + 6 * p[2]);
    }
    __syncthreads();
    for (int i = tid; i < DR * DR; i += 256) {
        const int r = i / DR, c = i - r * DR;
        const int x = ox + c, y = oy + r;
        if (x < 0 || y < 0 || x >= dw || y >= dh) continue;
        const unsigned short *q = tmp + (2 * r) * DR + c;
        const int v = q[0] + q[4 * DR] + 4 * (q[DR] + q[3 * DR]) + 6 * q[2 * DR];
        dst[r][c] = (uint8_t)((v + 128) >> 8);
    }
    __syncthreads();
    for (int i = tid; i < DR * DR; i += 256) {
        const int r = i / DR, c = i - r * DR;
        const int x = ox + c, y = oy + r;
        if (x >= 0 && y >= 0 && x < dw && y < dh) continue;
        if (x < -2 || y < -2 || x > dw + 1 || y > dh + 1) continue; // never read
        const int mx = reflect101(x, dw) - ox, my = reflect101(y, dh) - oy;
        if (mx >= 0 && my >= 0 && mx < DR && my < DR) dst[r][c] = dst[my][mx];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void pyr_down_tail3(const SlotDesc *slots, SlotList list, int first) {
    const SlotDesc &S = slots[list.s[blockIdx.z]];
    const LevelDesc L0 = S.lv[first], L1 = S.lv[first + 1], L2 = S.lv[first + 2], L3 = S.lv[first + 3];
    const int x3 = blockIdx.x * TT, y3 = blockIdx.y * TT;
    if (x3 >= L3.w || y3 >= L3.h) return;
    __shared__ uint8_t s0[R0][R0 + 3], s1[R1][R1 + 3], s2[R2][R2 + 3], s3[TT][TT + 3];
    __shared__ unsigned short tmp[R0 * R1];
    const int tid = threadIdx.x;
    const int x0 = 8 * x3 - 14, y0 = 8 * y3 - 14;
    for (int i = tid; i < R0 * R0; i += 256) {
        const int r = i / R0, c = i - r * R0;
        s0[r][c] = L0.origin[(ptrdiff_t)reflect101(y0 + r, L0.h) * L0.pitch + reflect101(x0 + c, L0.w)];
    }
    __syncthreads();
    lds_pyr_step<R0, R1>(s0, s1, tmp, 4 * x3 - 6, 4 * y3 - 6, L1.w, L1.h);
    lds_pyr_step<R1, R2>(s1, s2, tmp, 2 * x3 - 2, 2 * y3 - 2, L2.w, L2.h);
    lds_pyr_step<R2, TT>(s2, s3, tmp, x3, y3, L3.w, L3.h);
    // exclusive parts: 4 TT x 4 TT of L1 at (4 x3, 4 y3), entry (6, 6) of s1; 2 TT x 2 TT of L2 at (2 x3, 2 y3), entry (2, 2) of s2
    for (int i = tid; i < 16 * TT * TT; i += 256) {
        const int r = i / (4 * TT), c = i - r * (4 * TT), x = 4 * x3 + c, y = 4 * y3 + r;
        if (x < L1.w && y < L1.h) store_mirrored(L1, x, y, s1[r + 6][c + 6]);
    }
    for (int i = tid; i < 4 * TT * TT; i += 256) {
        const int r = i / (2 * TT), c = i - r * (2 * TT), x = 2 * x3 + c, y = 2 * y3 + r;
        if (x < L2.w && y < L2.h) store_mirrored(L2, x, y, s2[r + 2][c + 2]);
    }
    for (int i = tid; i < TT * TT; i += 256) {
        const int r = i / TT, c = i - r * TT, x = x3 + c, y = y3 + r;
        if (x < L3.w && y < L3.h) store_mirrored(L3, x, y, s3[r][c]);
    }
}

typedef short i16x2 __attribute__((ext_vector_type(2)));

// calcSharrDeriv for every level of the listed slots in one launch: grid (row segments of level 0,
// level, slot).  A thread produces eight consecutive pixels (two 16-byte stores of dx|dy pairs) from
// three 16-byte row pieces of the padded image, whose frame supplies calcSharrDeriv's reflect-101
// neighbours.  The arithmetic runs on packed int16 pairs (|values| <= 16 * 255):
//   T = 3 (r0 + r2) + 10 r1,  dx[x] = T[x+1] - T[x-1];   D = r2 - r0,  dy[x] = 3 (D[x-1] + D[x+1]) + 10 D[x].
// The planes hold FOUR TIMES these values (coefficients 12 and 40; |4 d| <= 64 * 255 still fits int16):
// the tracker's bilinear sums then carry CV_DESCALE(sum, 14) in their upper 16 bits, where one byte
// permute packs two of them.  ktlvo_download_scharr divides by four again.
// Only the interior of the plane is written; its frame stays zero (copyMakeBorder BORDER_CONSTANT).
// STREAM: the planes are written with non-temporal stores (see the launch site for when).
template <bool STREAM>
__global__ __launch_bounds__(256) void scharr_planes(const SlotDesc *slots, SlotList list, int n_levels) {
    const int level = blockIdx.y;
    if (level >= n_levels) return;
    const LevelDesc L = slots[list.s[blockIdx.z]].lv[level];
    const int ow = (L.w + 7) >> 3;                 // octets per row
    const int q = blockIdx.x * 256 + threadIdx.x;  // octet index in raster order
    if (q >= ow * L.h) return;
    const int y = q / ow, x0 = (q - y * ow) * 8;
    // bytes x0-4 .. x0+11 of rows y-1, y, y+1 (origin and pitch are multiples of 4); byte i is pixel x0-4+i
    u32x4 r[3];
#pragma unroll
    for (int j = 0; j < 3; j++)
        r[j] = *(const __attribute__((address_space(1))) u32x4_a4 *)((gcbyte)L.origin + (ptrdiff_t)(y - 1 + j) * L.pitch + x0 - 4);
    auto pr = [](uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_bit_cast(i16x2, __builtin_amdgcn_perm(hi, lo, sel)); };
    // pairs (b[3+2j], b[4+2j]), j = 0..4, and (b[4+2j], b[5+2j]), j = 0..3, zero-extended
    auto odd = [&](const u32x4 &d, i16x2 (&o)[5]) {
        o[0] = pr(d.y, d.x, 0x0c040c03u); o[1] = pr(d.y, d.y, 0x0c020c01u); o[2] = pr(d.z, d.y, 0x0c040c03u);
        o[3] = pr(d.z, d.z, 0x0c020c01u); o[4] = pr(d.w, d.z, 0x0c040c03u);
    };
    auto even = [&](const u32x4 &d, i16x2 (&e)[4]) {
        e[0] = pr(d.y, d.y, 0x0c010c00u); e[1] = pr(d.y, d.y, 0x0c030c02u);
        e[2] = pr(d.z, d.z, 0x0c010c00u); e[3] = pr(d.z, d.z, 0x0c030c02u);
    };
    i16x2 o0[5], o1[5], o2[5], e0[4], e2[4];
    odd(r[0], o0); odd(r[1], o1); odd(r[2], o2);
    even(r[0], e0); even(r[2], e2);
    const i16x2 three = {3 * SCHARR_GAIN, 3 * SCHARR_GAIN}, ten = {10 * SCHARR_GAIN, 10 * SCHARR_GAIN};
    i16x2 T[5], Do[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        T[j] = three * (o0[j] + o2[j]) + ten * o1[j];
        Do[j] = o2[j] - o0[j];
    }
    uint32_t out[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {   // pixels x0 + 2j, x0 + 2j + 1
        const i16x2 dx = T[j + 1] - T[j];
        const i16x2 dy = three * (Do[j] + Do[j + 1]) + ten * (e2[j] - e0[j]);
        const uint32_t ux = __builtin_bit_cast(uint32_t, dx), uy = __builtin_bit_cast(uint32_t, dy);
        out[2 * j] = __builtin_amdgcn_perm(uy, ux, 0x05040100u);
        out[2 * j + 1] = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
    }
    const gword o = (gword)L.dorigin + (ptrdiff_t)y * L.pitch + x0;
    if (x0 + 7 < L.w) {
        typedef __attribute__((address_space(1))) u32x4 *gquad;
        if (STREAM) {
            __builtin_nontemporal_store(u32x4{out[0], out[1], out[2], out[3]}, &((gquad)o)[0]);
            __builtin_nontemporal_store(u32x4{out[4], out[5], out[6], out[7]}, &((gquad)o)[1]);
        } else {
            ((gquad)o)[0] = u32x4{out[0], out[1], out[2], out[3]};
            ((gquad)o)[1] = u32x4{out[4], out[5], out[6], out[7]};
        }
    } else {
        for (int k = 0; k < 8 && x0 + k < L.w; k++) o[k] = out[k];
    }
}

static int launch_ingest(ktlvo_ctx *ctx, int slot, const uint8_t *d_src, int stride) {
    dim3 grid(ingest_blocks(ctx->cfg.width, ctx->cfg.height), 1, 1);
    hipLaunchKernelGGL(ingest_pad_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot, d_src, stride);
    KTLVO_HIP(ctx, hipGetLastError());
    ctx->built[slot] = 0;
    ctx->deriv_ok[slot] = 0;
    return KTLVO_OK;
}

} // namespace ktlvo

using namespace ktlvo;

// Host image -> pinned staging ring -> device staging -> padded plane.  A pageable 2-D copy with
// differing pitches degenerates into one small transfer per row (measured 3.4 ms per KITTI frame);
// staging through pinned memory makes it one contiguous DMA and lets the call return as soon as
// the caller's buffer has been read.
extern "C" int ktlvo_upload_image(ktlvo_ctx *ctx, int slot, const uint8_t *img, int w, int h, int stride) {
    if (!ctx || !img || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (w != ctx->cfg.width || h != ctx->cfg.height || stride < w) return KTLVO_ERR_INVALID;
    const size_t bytes = (size_t)w * h;
    StageRing &r = ctx->stage;
    // A frame that already sits in pinned host memory (hipHostMalloc / hipHostRegister) is read by the
    // DMA engine where it lies; the staging copy below is for pageable memory only.
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, img) == hipSuccess && attr.type == hipMemoryTypeHost) {
        const int k = r.next;
        r.next = (r.next + 1) % StageRing::N;
        if (!r.d_buf[k]) KTLVO_HIP(ctx, hipMalloc((void **)&r.d_buf[k], bytes));
        if (!r.done[k]) KTLVO_HIP(ctx, hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming));
        if (r.used[k]) KTLVO_HIP(ctx, hipEventSynchronize(r.done[k]));
        if (stride == w)
            KTLVO_HIP(ctx, hipMemcpyAsync(r.d_buf[k], img, bytes, hipMemcpyHostToDevice, ctx->stream));
        else
            KTLVO_HIP(ctx, hipMemcpy2DAsync(r.d_buf[k], (size_t)w, img, (size_t)stride, (size_t)w, (size_t)h,
                                            hipMemcpyHostToDevice, ctx->stream));
        int rc = launch_ingest(ctx, slot, r.d_buf[k], w);
        if (rc != KTLVO_OK) return rc;
        KTLVO_HIP(ctx, hipEventRecord(r.done[k], ctx->stream));
        r.used[k] = true;
        return KTLVO_OK;
    }
    (void)hipGetLastError(); // an unregistered pointer is reported as an error by some runtimes
    const int k = r.next;
    // each ring entry is made on its first use; a failure leaves it partly made and the next call retries
    if (!r.h_buf[k]) KTLVO_HIP(ctx, hipHostMalloc((void **)&r.h_buf[k], bytes, hipHostMallocDefault));
    if (!r.d_buf[k]) KTLVO_HIP(ctx, hipMalloc((void **)&r.d_buf[k], bytes));
    if (!r.done[k]) KTLVO_HIP(ctx, hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming));
    r.next = (r.next + 1) % StageRing::N;
    if (r.used[k]) KTLVO_HIP(ctx, hipEventSynchronize(r.done[k])); // that buffer's previous DMA and ingest are over
    if (stride == w) {
        memcpy(r.h_buf[k], img, bytes);
    } else {
        for (int y = 0; y < h; y++) memcpy(r.h_buf[k] + (size_t)y * w, img + (size_t)y * stride, w);
    }
    KTLVO_HIP(ctx, hipMemcpyAsync(r.d_buf[k], r.h_buf[k], bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_ingest(ctx, slot, r.d_buf[k], w);
    if (rc != KTLVO_OK) return rc;
    KTLVO_HIP(ctx, hipEventRecord(r.done[k], ctx->stream));
    r.used[k] = true;
    return KTLVO_OK;
}

namespace ktlvo {
// Staging buffer k of the batch uploads, at least `bytes` large and no longer read by an earlier ingest launch.
static int stage_take(ktlvo_ctx *ctx, size_t bytes, int k, bool want_pinned, bool want_device = true) {
    StageRing &r = ctx->stage;
    if (bytes > r.batch_bytes) { // grow: every buffer goes, and each comes back on its next use
        KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->copy_stream) KTLVO_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        for (int i = 0; i < 2; i++) {
            if (r.d_batch[i]) KTLVO_HIP(ctx, hipFree(r.d_batch[i]));
            r.d_batch[i] = nullptr;
            if (r.h_batch[i]) KTLVO_HIP(ctx, hipHostFree(r.h_batch[i]));
            r.h_batch[i] = nullptr;
            r.batch_used[i] = false;
        }
        r.batch_bytes = bytes;
    }
    if (want_device && !r.d_batch[k]) KTLVO_HIP(ctx, hipMalloc((void **)&r.d_batch[k], r.batch_bytes));
    if (want_pinned && !r.h_batch[k]) KTLVO_HIP(ctx, hipHostMalloc((void **)&r.h_batch[k], r.batch_bytes, hipHostMallocDefault));
    if (!r.batch_done[k]) KTLVO_HIP(ctx, hipEventCreateWithFlags(&r.batch_done[k], hipEventDisableTiming));
    if (!r.copy_done[k]) KTLVO_HIP(ctx, hipEventCreateWithFlags(&r.copy_done[k], hipEventDisableTiming));
    if (r.batch_used[k]) KTLVO_HIP(ctx, hipEventSynchronize(r.batch_done[k]));
    return KTLVO_OK;
}
static bool is_pinned(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost) return true;
    (void)hipGetLastError(); // an unregistered pointer is reported as an error by some runtimes
    return false;
}
} // namespace ktlvo

// A whole batch of frames in one DMA and one ingest launch.  A per-frame upload moves 466 KB per DMA for a
// KITTI-sized frame, which the host link serves at well under half its rate; one transfer of the whole batch
// runs at the link's rate.  The DMA goes to the context's copy stream and only the ingest launch, on the
// context's stream, waits for it: a caller that uploads the next batch right after queueing the work on the
// current one has the transfer run under that work, with one context and one set of slots (the ingest launch
// overwrites the slots in stream order, after the kernels that still read them).
extern "C" int ktlvo_upload_images(ktlvo_ctx *ctx, const int *slots, int n, const uint8_t *imgs, int w, int h, int stride,
                                   size_t image_stride) {
    if (!ctx || n < 0 || (n > 0 && (!slots || !imgs))) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (n == 0) return KTLVO_OK;
    if (w != ctx->cfg.width || h != ctx->cfg.height || stride < w || image_stride < (size_t)stride * (h - 1) + w)
        return KTLVO_ERR_INVALID;
    if (n > MAX_BATCH) return KTLVO_ERR_CAPACITY;
    for (int i = 0; i < n; i++)
        if (!slot_ok(ctx, slots[i])) return KTLVO_ERR_INVALID;
    const size_t bytes = image_stride * (size_t)(n - 1) + (size_t)stride * (h - 1) + w;
    StageRing &r = ctx->stage;
    {
        const int rc_s = device_streams(ctx);
        if (rc_s != KTLVO_OK) return rc_s;
    }
    const bool pinned = is_pinned(imgs);
    const int k = r.batch_next;
    int rc = stage_take(ctx, bytes, k, !pinned);
    if (rc != KTLVO_OK) return rc;
    r.batch_next = k ^ 1;
    const uint8_t *src = imgs;
    if (!pinned) { // pageable source: staged, so the call returns with the caller's buffer free
        memcpy(r.h_batch[k], imgs, bytes);
        src = r.h_batch[k];
    }
    KTLVO_HIP(ctx, hipMemcpyAsync(r.d_batch[k], src, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    KTLVO_HIP(ctx, hipEventRecord(r.copy_done[k], ctx->copy_stream));
    KTLVO_HIP(ctx, hipStreamWaitEvent(ctx->stream, r.copy_done[k], 0));
    SlotList list;
    list.n = n;
    for (int i = 0; i < n; i++) list.s[i] = slots[i];
    dim3 grid(ingest_blocks(w, h), 1, n);
    hipLaunchKernelGGL(ingest_pad_batch_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, r.d_batch[k], image_stride, stride);
    KTLVO_HIP(ctx, hipGetLastError());
    KTLVO_HIP(ctx, hipEventRecord(r.batch_done[k], ctx->stream));
    r.batch_used[k] = true;
    for (int i = 0; i < n; i++) { ctx->built[slots[i]] = 0; ctx->deriv_ok[slots[i]] = 0; }
    return KTLVO_OK;
}

namespace ktlvo {
// Two frames (a stereo pair) into two slots with one launch and no DMA: the ingest kernel reads the frames
// from pinned host memory across the link (the caller's buffers if they are pinned, else a pinned copy made
// here).  For the 0.9 MB of a KITTI-sized pair the link time is the same either way, and the two DMA
// set-ups (some 25 us on the frame call's critical path) go away.  Buffers the runtime cannot map for the
// device take the staged DMA route.
int upload_pair(ktlvo_ctx *ctx, int slot_a, int slot_b, const uint8_t *a, const uint8_t *b, int w, int h) {
    const size_t bytes = (size_t)w * h, img_stride = (bytes + 255) & ~(size_t)255;
    StageRing &r = ctx->stage;
    const bool pinned = is_pinned(a) && is_pinned(b);
    int rc = stage_take(ctx, 2 * img_stride, 0, !pinned, false); // (a device staging buffer only if the DMA route is taken)
    if (rc != KTLVO_OK) return rc;
    const uint8_t *ha = a, *hb = b;
    if (!pinned) {
        memcpy(r.h_batch[0], a, bytes);
        memcpy(r.h_batch[0] + img_stride, b, bytes);
        ha = r.h_batch[0]; hb = r.h_batch[0] + img_stride;
    }
    void *da = nullptr, *db = nullptr;
    dim3 grid(ingest_blocks(w, h), 1, 2);
    if (hipHostGetDevicePointer(&da, (void *)ha, 0) == hipSuccess && hipHostGetDevicePointer(&db, (void *)hb, 0) == hipSuccess) {
        hipLaunchKernelGGL(ingest_pad_pair_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot_a, slot_b, (const uint8_t *)da,
                           (const uint8_t *)db, w);
    } else {
        (void)hipGetLastError();
        rc = stage_take(ctx, 2 * img_stride, 0, !pinned, true);
        if (rc != KTLVO_OK) return rc;
        uint8_t *d = r.d_batch[0];
        KTLVO_HIP(ctx, hipMemcpyAsync(d, ha, bytes, hipMemcpyHostToDevice, ctx->stream));
        KTLVO_HIP(ctx, hipMemcpyAsync(d + img_stride, hb, bytes, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(ingest_pad_pair_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, slot_a, slot_b, d, d + img_stride, w);
    }
    KTLVO_HIP(ctx, hipGetLastError());
    KTLVO_HIP(ctx, hipEventRecord(r.batch_done[0], ctx->stream));
    r.batch_used[0] = true;
    ctx->built[slot_a] = ctx->built[slot_b] = 0;
    ctx->deriv_ok[slot_a] = ctx->deriv_ok[slot_b] = 0;
    return KTLVO_OK;
}
} // namespace ktlvo

extern "C" int ktlvo_set_image_dev(ktlvo_ctx *ctx, int slot, const uint8_t *d_img, int w, int h, int stride) {
    if (!ctx || !d_img || !slot_ok(ctx, slot)) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (w != ctx->cfg.width || h != ctx->cfg.height || stride < w) return KTLVO_ERR_INVALID;
    return launch_ingest(ctx, slot, d_img, stride);
}

extern "C" int ktlvo_build_pyramids(ktlvo_ctx *ctx, const int *slots, int n_slots, int max_level, int win) {
    if (!ctx || n_slots < 0 || (n_slots > 0 && !slots) || max_level < 0) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (win != ctx->cfg.win) return KTLVO_ERR_UNSUPPORTED; // level count and frame were sized for cfg.win
    const int top = max_level < ctx->n_levels - 1 ? max_level : ctx->n_levels - 1;
    for (int i = 0; i < n_slots; i++) {
        if (!slot_ok(ctx, slots[i])) return KTLVO_ERR_INVALID;
        if (ctx->built[slots[i]] < 0) return KTLVO_ERR_STATE;
    }
    for (int base = 0; base < n_slots; base += MAX_BATCH) {
        SlotList list;
        list.n = n_slots - base < MAX_BATCH ? n_slots - base : MAX_BATCH;
        for (int i = 0; i < list.n; i++) list.s[i] = slots[base + i];
        {
            ProfScope ps(ctx, KTLVO_K_PYRDOWN);
            // level 0 -> 1 streams; with the reference's five levels and a handful of images (the per-frame path
            // builds two), levels 2, 3 and 4 then come from ONE launch: three launches of a few workgroups each
            // cost 25 us in launch latency there, the fused kernel 9.  For large batches the per-level launches are
            // faster (the fused kernel recomputes halos through LDS, several times over with its small tiles).
            // (KTLVO_PYR_PER_LEVEL=1 in the environment at context creation keeps a launch per level: A/B and tests)
            const bool tail = top == 4 && list.n <= 8 && !ctx->pyr_per_level;
            // a large batch: a launch per level down to level 2, then levels 3 and 4 from one launch, a workgroup per image
            const bool tail2 = top == 4 && !tail && !ctx->pyr_per_level;
            for (int l = 0; l < (tail ? 1 : tail2 ? 2 : top); l++) {
                dim3 grid((ctx->lw[l + 1] + PD_TW - 1) / PD_TW, (ctx->lh[l + 1] + PD_TH - 1) / PD_TH, list.n);
                hipLaunchKernelGGL(pyr_down_u8, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, l);
            }
            if (tail) {
                dim3 grid((ctx->lw[4] + TT - 1) / TT, (ctx->lh[4] + TT - 1) / TT, list.n);
                hipLaunchKernelGGL(pyr_down_tail3, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, 1);
            }
            if (tail2) hipLaunchKernelGGL(pyr_down_tail2, dim3(list.n), dim3(PT2_THREADS), 0, ctx->stream, ctx->d_slots, list, 2);
        }
        KTLVO_HIP(ctx, hipGetLastError());
        {
            ProfScope ps2(ctx, KTLVO_K_SCHARR);
            const int octets0 = ((ctx->lw[0] + 7) / 8) * ctx->lh[0];
            dim3 grid((octets0 + 255) / 256, top + 1, list.n);
            // Planes of a batch far larger than the 256 MiB of last-level cache are not read again before
            // they have been evicted, so their stores should not allocate there (a quarter off the kernel's
            // time for the 256 KITTI-sized frames of a benchmark step); a smaller batch, down to the two
            // frames of the per-frame path, is read by the LK launch that follows while still cached.
            size_t plane_bytes = 0;
            for (int l = 0; l <= top; l++) plane_bytes += (size_t)ctx->lw[l] * ctx->lh[l] * 4;
            if (plane_bytes * list.n >= ((size_t)512 << 20))
                hipLaunchKernelGGL(scharr_planes<true>, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, top + 1);
            else
                hipLaunchKernelGGL(scharr_planes<false>, grid, dim3(256), 0, ctx->stream, ctx->d_slots, list, top + 1);
            KTLVO_HIP(ctx, hipGetLastError());
        }
    }
    for (int i = 0; i < n_slots; i++) { ctx->built[slots[i]] = top; ctx->deriv_ok[slots[i]] = 1; }
    return KTLVO_OK;
}

extern "C" int ktlvo_build_pyramid(ktlvo_ctx *ctx, int slot, int max_level, int win) {
    return ktlvo_build_pyramids(ctx, &slot, 1, max_level, win);
}

extern "C" int ktlvo_pyramid_levels(ktlvo_ctx *ctx, int slot, int *max_level_built) {
    if (!ctx || !slot_ok(ctx, slot) || !max_level_built) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    *max_level_built = ctx->built[slot];
    return KTLVO_OK;
}

extern "C" int ktlvo_download_level(ktlvo_ctx *ctx, int slot, int level, uint8_t *dst, int *w, int *h) {
    if (!ctx || !slot_ok(ctx, slot) || !dst || level < 0) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (level > ctx->built[slot]) return KTLVO_ERR_STATE;
    const LevelDesc &L = ctx->h_slots[slot].lv[level];
    KTLVO_HIP(ctx, hipMemcpy2DAsync(dst, L.w, L.origin, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (w) *w = L.w;
    if (h) *h = L.h;
    return KTLVO_OK;
}

extern "C" int ktlvo_download_scharr(ktlvo_ctx *ctx, int slot, int level, int16_t *dst) {
    if (!ctx || !slot_ok(ctx, slot) || !dst || level < 0) return KTLVO_ERR_INVALID;
    KTLVO_BIND(ctx);
    if (level > ctx->built[slot] || !ctx->deriv_ok[slot]) return KTLVO_ERR_STATE;
    const LevelDesc &L = ctx->h_slots[slot].lv[level];
    KTLVO_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)L.w * 4, L.dorigin, (size_t)L.pitch * 4, (size_t)L.w * 4, L.h,
                                    hipMemcpyDeviceToHost, ctx->stream));
    KTLVO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0, n = (size_t)L.w * L.h * 2; i < n; i++) dst[i] = (int16_t)(dst[i] / SCHARR_GAIN); // exact: stored as multiples
    return KTLVO_OK;
}
