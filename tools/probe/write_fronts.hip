// Write-pattern probe for gfx950: the bandwidth of storing the Scharr planes of a batch of images (4 bytes per pixel,
// non-temporal 16-byte stores) when the stores are issued (a) in raster order over each image by independent threads,
// as scharr_planes does, and (b) by waves that each walk down a strip of 256 pixels row by row, as a kernel that
// keeps its rows in registers would -- thousands of write fronts instead of one per image.
//   hipcc -O3 --offload-arch=gfx950 -o write_fronts write_fronts.hip && ./write_fronts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int W = 1241, H = 376, PITCH = 1344, PLANE_ROWS = H + 64, N = 256;
__global__ __launch_bounds__(256) void raster(unsigned *planes, int nt) {
    unsigned *p = planes + (size_t)blockIdx.y * PITCH * PLANE_ROWS;
    const int ow = (W + 7) / 8, q = blockIdx.x * 256 + threadIdx.x;
    if (q >= ow * H) return;
    const int y = q / ow, x0 = (q - y * ow) * 8;
    u32x4 *o = (u32x4 *)(p + (size_t)(y + 32) * PITCH + 32 + x0);
    const u32x4 v = {(unsigned)q, 1u, 2u, 3u};
    if (nt) { __builtin_nontemporal_store(v, o); __builtin_nontemporal_store(v, o + 1); } else { o[0] = v; o[1] = v; }
}
__global__ __launch_bounds__(256) void strips(unsigned *planes, int band_rows, int strips_x, int bands, int nt, const unsigned char *img) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_img = strips_x * bands;
    const int im = wave / per_img, r = wave - im * per_img;
    if (im >= N) return;
    const int band = r / strips_x, strip = r - band * strips_x;
    unsigned *p = planes + (size_t)im * PITCH * PLANE_ROWS;
    const unsigned char *src = img + (size_t)im * PITCH * PLANE_ROWS;
    const int x = strip * 248 + 4 * lane - 4;
    const int y0 = band * band_rows, y1 = min(y0 + band_rows, H);
    unsigned acc = 0;
    for (int y = y0; y < y1; y++) {
        if (img) acc += *(const unsigned *)(src + (size_t)(y + 32) * PITCH + 32 + x); // the image row this wave would read
        if (lane >= 1 && lane <= 62 && x + 3 < W) {
            u32x4 *o = (u32x4 *)(p + (size_t)(y + 32) * PITCH + 32 + x);
            const u32x4 v = {(unsigned)y + acc, 1u, 2u, 3u};
            if (nt) __builtin_nontemporal_store(v, o); else *o = v;
        }
    }
}
int main() {
    unsigned *d; unsigned char *img;
    const size_t bytes = (size_t)N * PITCH * PLANE_ROWS * 4;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&img, (size_t)N * PITCH * PLANE_ROWS));
    CK(hipMemset(d, 0, bytes)); CK(hipMemset(img, 1, (size_t)N * PITCH * PLANE_ROWS));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double mb = (double)N * W * H * 4 / 1e6;
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        CK(hipEventRecord(a)); for (int i = 0; i < 20; i++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        printf("%-50s %.4f ms  %.2f TB/s of plane bytes (%.0f MB)\n", name, ms, mb / ms / 1e3, mb);
    };
    for (int nt = 1; nt >= 0; nt--) {
        time(nt ? "raster, non-temporal" : "raster, ordinary stores", [&] { hipLaunchKernelGGL(raster, dim3(((W + 7) / 8 * H + 255) / 256, N), dim3(256), 0, 0, d, nt); });
        for (int rows : {24, 47, 94, 188, 376}) {
            const int bands = (H + rows - 1) / rows, sx = (W + 247) / 248;
            char name[96];
            snprintf(name, sizeof name, "strips of 248 px, bands of %d rows%s", rows, nt ? ", non-temporal" : "");
            time(name, [&] { hipLaunchKernelGGL(strips, dim3((N * sx * bands + 3) / 4), dim3(256), 0, 0, d, rows, sx, bands, nt, (const unsigned char *)nullptr); });
            snprintf(name, sizeof name, "  the same, reading the image rows too");
            time(name, [&] { hipLaunchKernelGGL(strips, dim3((N * sx * bands + 3) / 4), dim3(256), 0, 0, d, rows, sx, bands, nt, (const unsigned char *)img); });
        }
    }
    return 0;
}
