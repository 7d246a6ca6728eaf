// Vector-instruction issue rate of a gfx950 SIMD for the instruction kinds the LK kernel is made of.
//
// Every wave runs ITERS x 16 independent instructions of one kind (8 accumulators, so no
// instruction waits for the one before it); workgroups are 256 threads (one wave per SIMD of a CU)
// and the grid is 256 CUs x W workgroups.  Where the waves really ran is read back per wave (HW_ID,
// XCC_ID) together with its start and end on the chip-wide 100 MHz counter.  Reported per kind and W:
//   cyc_per_inst_wave = shader cycles one wave needs per instruction (s_memtime, median over waves),
//   waves_per_simd, concurrent_waves = waves placed on a SIMD, and how many of them ran at a time,
//   cyc_per_inst_simd = (last end - first start on a SIMD, in shader cycles) / (instructions of all
//                       waves placed on it): the SIMD's issue cost of one wave-instruction when that
//                       many waves compete -- the number a roofline needs (median over SIMDs),
//   launch_ms         = first start to last end over the whole grid.
// The "mix" kind is the instruction mix of one Gauss-Newton step of pyrlk_track_w21 (22 dots,
// 7 permutes/packed shifts, 4 packed f32, 10 DPP adds, 20 plain VALU; 63 in all) without its
// dependencies: the issue ceiling of that mix.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                            \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

enum Kind { ADD_F32, FMA_F32, DOT2_I16, PERM_B32, ADD_DPP, PK_MUL_F32, PK_ASHR_I16, AND_OR, ADD3, CVT_I2F, MIX, DOT_NOP, N_KINDS };
static const char *kind_name[N_KINDS] = {"v_add_f32",     "v_fma_f32",        "v_dot2_i32_i16", "v_perm_b32",
                                         "v_add_f32_dpp", "v_pk_mul_f32",     "v_pk_ashrrev_i16", "v_and_or_b32",
                                         "v_add3_u32",    "v_cvt_f32_i32",    "lk_step_mix",    "7dot+s_nop2"};
// instructions per asm block
static const int kind_per_block[N_KINDS] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 63, 14};

#define R8(OP)                                                                                  \
    OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7") OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") \
        OP("%5") OP("%6") OP("%7")

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *cyc, unsigned long long *real, float *sink, int iters) {
    // real[] receives, per wave: start and end on the chip-wide 100 MHz counter, then the wave's placement
    // (HW_ID: SIMD, CU, shader engine; XCC_ID), from which the host works out how many waves really
    // shared each SIMD and for how long
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x = 1.0001f, y = 0.5f;
    float p0 = a0, p1 = a1; // second halves of two register pairs for the packed forms
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == ADD_F32) {
#define OP(D) "v_add_f32 " D ", " D ", %8\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == FMA_F32) {
#define OP(D) "v_fma_f32 " D ", " D ", %8, %9\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == DOT2_I16) {
#define OP(D) "v_dot2_i32_i16 " D ", %8, %9, " D "\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == PERM_B32) {
#define OP(D) "v_perm_b32 " D ", " D ", %8, %9\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == ADD_DPP) {
#define OP(D) "v_add_f32_dpp " D ", %8, " D " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == PK_MUL_F32) {
            // four independent 64-bit pairs
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 q0 = {a0, a1}, q1 = {a2, a3}, q2 = {a4, a5}, q3 = {a6, a7}, s = {x, y};
#define OP4 "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
            asm volatile(OP4 OP4 OP4 OP4 : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(s));
#undef OP4
            a0 = q0.x; a1 = q0.y; a2 = q1.x; a3 = q1.y; a4 = q2.x; a5 = q2.y; a6 = q3.x; a7 = q3.y;
        } else if (KIND == PK_ASHR_I16) {
#define OP(D) "v_pk_ashrrev_i16 " D ", 1, " D " op_sel_hi:[0,1]\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == AND_OR) {
#define OP(D) "v_and_or_b32 " D ", " D ", %8, %9\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == ADD3) {
#define OP(D) "v_add3_u32 " D ", " D ", %8, %9\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == CVT_I2F) {
#define OP(D) "v_cvt_f32_i32 " D ", " D "\n\t"
            asm volatile(R8(OP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
#undef OP
        } else if (KIND == DOT_NOP) {
            // the shape of the kernel's dot blocks: 14 dots, then the wait states before another kind reads them
            asm volatile("v_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %1, %8, %9, %1\n\tv_dot2_i32_i16 %2, %8, %9, %2\n\t"
                         "v_dot2_i32_i16 %3, %8, %9, %3\n\tv_dot2_i32_i16 %4, %8, %9, %4\n\tv_dot2_i32_i16 %5, %8, %9, %5\n\t"
                         "v_dot2_i32_i16 %6, %8, %9, %6\n\tv_dot2_i32_i16 %0, %8, %9, %0\n\tv_dot2_i32_i16 %1, %8, %9, %1\n\t"
                         "v_dot2_i32_i16 %2, %8, %9, %2\n\tv_dot2_i32_i16 %3, %8, %9, %3\n\tv_dot2_i32_i16 %4, %8, %9, %4\n\t"
                         "v_dot2_i32_i16 %5, %8, %9, %5\n\tv_dot2_i32_i16 %6, %8, %9, %6\n\ts_nop 2\n\t"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x), "v"(y));
        } else if (KIND == MIX) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 q0 = {p0, p1}, s = {x, y};
            asm volatile(
                // weights: 4 packed f32, 6 plain
                "v_pk_mul_f32 %8, %8, %11\n\tv_pk_add_f32 %8, %8, %11\n\tv_pk_mul_f32 %8, %8, %11\n\tv_pk_add_f32 %8, %8, %11\n\t"
                "v_mul_f32 %0, %0, %9\n\tv_add_f32 %1, %1, %9\n\tv_add3_u32 %2, %2, %9, %10\n\tv_sub_u32 %3, %3, %9\n\t"
                "v_perm_b32 %4, %4, %9, %10\n\tv_perm_b32 %5, %5, %9, %10\n\t"
                // 14 sample dots + 7 pack
                "v_dot2_i32_i16 %0, %9, %10, %0\n\tv_dot2_i32_i16 %1, %9, %10, %1\n\tv_dot2_i32_i16 %2, %9, %10, %2\n\t"
                "v_dot2_i32_i16 %3, %9, %10, %3\n\tv_dot2_i32_i16 %4, %9, %10, %4\n\tv_dot2_i32_i16 %5, %9, %10, %5\n\t"
                "v_dot2_i32_i16 %6, %9, %10, %6\n\tv_dot2_i32_i16 %0, %9, %10, %0\n\tv_dot2_i32_i16 %1, %9, %10, %1\n\t"
                "v_dot2_i32_i16 %2, %9, %10, %2\n\tv_dot2_i32_i16 %3, %9, %10, %3\n\tv_dot2_i32_i16 %4, %9, %10, %4\n\t"
                "v_dot2_i32_i16 %5, %9, %10, %5\n\tv_dot2_i32_i16 %6, %9, %10, %6\n\t"
                "v_perm_b32 %0, %1, %0, %10\n\tv_perm_b32 %2, %3, %2, %10\n\tv_perm_b32 %4, %5, %4, %10\n\t"
                "v_pk_ashrrev_i16 %0, 1, %0 op_sel_hi:[0,1]\n\tv_pk_ashrrev_i16 %2, 1, %2 op_sel_hi:[0,1]\n\t"
                "v_pk_ashrrev_i16 %4, 1, %4 op_sel_hi:[0,1]\n\tv_bfe_u32 %6, %6, 9, 16\n\t"
                // 8 mismatch dots
                "v_dot2_i32_i16 %1, %0, %9, 0\n\tv_dot2_i32_i16 %3, %0, %10, 0\n\tv_dot2_i32_i16 %1, %2, %9, %1\n\t"
                "v_dot2_i32_i16 %3, %2, %10, %3\n\tv_dot2_i32_i16 %1, %4, %9, %1\n\tv_dot2_i32_i16 %3, %4, %10, %3\n\t"
                "v_dot2_i32_i16 %1, %6, %9, %1\n\tv_dot2_i32_i16 %3, %6, %10, %3\n\ts_nop 2\n\t"
                // cross-lane sum: swap, add, convert, 4 DPP adds, bcast add, 2 readlane-like (counted as v_mov here)
                "v_permlane32_swap_b32 %1, %3\n\tv_add_u32 %1, %1, %3\n\tv_cvt_f32_i32 %1, %1\n\t"
                "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %5, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %7, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %1, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %5, %1, %1 row_bcast:15 row_mask:0xa\n\t"
                "v_mov_b32 %7, %5\n\tv_mov_b32 %1, %5\n\t"
                // step arithmetic and tests: 14 plain float
                "v_mul_f32 %0, %0, %9\n\tv_mul_f32 %2, %2, %9\n\tv_sub_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %10\n\t"
                "v_mul_f32 %4, %4, %9\n\tv_mul_f32 %6, %6, %9\n\tv_sub_f32 %4, %4, %6\n\tv_mul_f32 %4, %4, %10\n\t"
                "v_add_f32 %5, %5, %0\n\tv_add_f32 %7, %7, %4\n\tv_mul_f32 %0, %0, %0\n\tv_fma_f32 %0, %4, %4, %0\n\t"
                "v_max_f32 %2, %0, %4\n\tv_sub_f32 %6, %5, %7\n\t"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(q0)
                : "v"(x), "v"(y), "v"(s));
            p0 = q0.x; p1 = q0.y;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        cyc[wave] = t1 - t0;
        real[3 * wave] = r0;
        real[3 * wave + 1] = r1;
        real[3 * wave + 2] = ((unsigned long long)xcc << 32) | hw;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0 + p1;
}

typedef void (*kern_t)(unsigned long long *, unsigned long long *, float *, int);
static kern_t kernels[N_KINDS] = {probe<ADD_F32>,  probe<FMA_F32>,     probe<DOT2_I16>, probe<PERM_B32>, probe<ADD_DPP>, probe<PK_MUL_F32>,
                                  probe<PK_ASHR_I16>, probe<AND_OR>,   probe<ADD3>,     probe<CVT_I2F>,  probe<MIX>,     probe<DOT_NOP>};

#include <map>
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int max_waves = cus * 4 * 8;
    unsigned long long *d_cyc, *d_real;
    float *d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * max_waves));
    CHECK(hipMalloc(&d_real, sizeof(unsigned long long) * max_waves * 3));
    CHECK(hipMalloc(&d_sink, sizeof(float) * max_waves * 64));
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"rows\": [\n", prop.gcnArchName, cus, iters);
    bool first = true;
    for (int k = 0; k < N_KINDS; k++) {
        for (int W : {1, 2, 4, 8}) {
            const int blocks = cus * W;
            const int waves = blocks * 4;
            // warm-up launch (code load, clock run-up), then the measured one
            hipLaunchKernelGGL(kernels[k], dim3(blocks), dim3(256), 0, 0, d_cyc, d_real, d_sink, iters);
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(kernels[k], dim3(blocks), dim3(256), 0, 0, d_cyc, d_real, d_sink, iters);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> cyc(waves), real(3 * (size_t)waves);
            CHECK(hipMemcpy(cyc.data(), d_cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(real.data(), d_real, sizeof(unsigned long long) * waves * 3, hipMemcpyDeviceToHost));
            const double n_inst = (double)iters * kind_per_block[k];
            // per SIMD: waves placed there, first start, last end, summed cycles and 100 MHz ticks
            struct Simd { int n = 0; unsigned long long s = ~0ull, e = 0, cyc = 0, ticks = 0; };
            std::map<unsigned long long, Simd> simds;
            unsigned long long g0 = ~0ull, g1 = 0;
            for (int i = 0; i < waves; i++) {
                const unsigned long long id = real[3 * i + 2];
                const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
                // SIMD_ID 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13 of HW_ID
                const unsigned long long key = ((unsigned long long)xcc << 16) | (hw & 0xff30);
                Simd &sd = simds[key];
                sd.n++;
                sd.s = std::min(sd.s, real[3 * i]);
                sd.e = std::max(sd.e, real[3 * i + 1]);
                sd.cyc += cyc[i];
                sd.ticks += real[3 * i + 1] - real[3 * i];
                g0 = std::min(g0, real[3 * i]);
                g1 = std::max(g1, real[3 * i + 1]);
            }
            std::vector<double> per_simd, nw, conc, ghz;
            for (auto &kv : simds) {
                const Simd &sd = kv.second;
                const double clock = (double)sd.cyc / (double)sd.ticks * 0.1; // GHz
                const double span_cyc = (double)(sd.e - sd.s) * 10.0 * clock;
                per_simd.push_back(span_cyc / (sd.n * n_inst));
                nw.push_back(sd.n);
                conc.push_back((double)sd.ticks / (double)(sd.e - sd.s));
                ghz.push_back(clock);
            }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            std::sort(cyc.begin(), cyc.end());
            printf("%s {\"kind\": \"%s\", \"blocks_per_cu\": %d, \"simds_used\": %zu, \"waves_per_simd\": %.1f, \"concurrent_waves\": %.2f, "
                   "\"cyc_per_inst_wave\": %.3f, \"cyc_per_inst_simd\": %.3f, \"clock_ghz\": %.3f, \"launch_ms\": %.3f}",
                   first ? " " : ",\n ", kind_name[k], W, simds.size(), med(nw), med(conc), (double)cyc[waves / 2] / n_inst,
                   med(per_simd), med(ghz), (double)(g1 - g0) * 1e-5);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
