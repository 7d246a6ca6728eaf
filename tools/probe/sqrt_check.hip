// Checks ktlvo::sqrt_rn_radicand against the correctly rounded root over its whole domain and beyond: zero and every
// float in (0, 1), subnormals included (the kernel's radicands are zero or lie in [2^-104, 1)).  The reference is (float)sqrt((double)x): the double root of a float, rounded to float, is the correctly
// rounded float root (53 >= 2 * 24 + 2 bits).  Prints the number of inputs and of mismatches.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../slam_amd/csrc sqrt_check.hip -o sqrt_check
#include <cstdint>
#include <cstdio>

#include "sqrt_rn.hpp"

__global__ void check(int e_first, int e_last, unsigned long long *bad, unsigned *first_bad) {
    // one exponent per blockIdx.y, 2^23 mantissas spread over the x grid
    const int e = e_first + (int)blockIdx.y; // biased exponent field; 0 = subnormals
    if (e > e_last) return;
    unsigned long long n_bad = 0;
    for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
        const unsigned bits = ((unsigned)e << 23) | m;
        if (bits == 0u) continue;
        const float x = __uint_as_float(bits);
        const float got = ktlvo::sqrt_rn_radicand(x), want = (float)sqrt((double)x);
        if (__float_as_uint(got) != __float_as_uint(want)) {
            n_bad++;
            atomicMin(first_bad, bits);
        }
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

__global__ void check_zero(unsigned long long *bad) {
    if (__float_as_uint(ktlvo::sqrt_rn_radicand(0.f)) != 0u) atomicAdd(bad, 1ull);
}

int main() {
    unsigned long long *d_bad, h_bad = 0;
    unsigned *d_first, h_first = 0xffffffffu;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4);
    hipMemcpy(d_bad, &h_bad, 8, hipMemcpyHostToDevice); hipMemcpy(d_first, &h_first, 4, hipMemcpyHostToDevice);
    const int e_first = 0, e_last = 126;
    check_zero<<<1, 1>>>(d_bad);
    check<<<dim3(256, e_last - e_first + 1), 256>>>(e_first, e_last, d_bad, d_first);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&h_first, d_first, 4, hipMemcpyDeviceToHost);
    printf("inputs: 0 and the %llu floats in (0, 1); mismatches: %llu", ((unsigned long long)(e_last - e_first + 1) << 23) - 1, h_bad);
    if (h_bad) printf(" (smallest input bits 0x%08x)", h_first);
    printf("\n");
    return h_bad ? 1 : 0;
}
