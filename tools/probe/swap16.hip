// Prints what v_permlane16_swap / v_permlane32_swap do to two registers holding lane ids (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    o[l] = r[0]; o[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    o[128 + l] = q[0]; o[192 + l] = q[1];
}
int main() {
    unsigned *d, h[256];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char *names[4] = {"swap16 first ", "swap16 second", "swap32 first ", "swap32 second"};
    for (int r = 0; r < 4; r++) {
        printf("%s:", names[r]);
        for (int i = 0; i < 64; i += 8) printf(" [%u..%u]", h[64 * r + i], h[64 * r + i + 7]);
        printf("\n");
    }
    return 0;
}
