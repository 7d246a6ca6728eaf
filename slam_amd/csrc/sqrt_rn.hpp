// Correctly rounded square root of the corner response's radicand (see the function's comment); a header of its own so
// that tools/probe/sqrt_check.hip can test it against the double-precision root over its whole domain.
#pragma once
#include <hip/hip_runtime.h>

namespace ktlvo {

// sqrtf (correctly rounded) for the minimum-eigenvalue response's radicand (ha - hc)^2 + fb^2, which is zero or lies in
// [2^-104, 1): the sums behind it are nine products of at least 2^-28 each or zero, so a difference of two of them is a
// multiple of 2^-52 and a non-zero radicand is at least 2^-104.  The hardware root is within one ulp; the two residuals
// x - s (s -/+ 1 ulp), one fused operation each, tell which neighbour is the rounded root (the compiler's own expansion
// of sqrtf).  That expansion also tests for tiny radicands, whose residuals would be subnormal, and scales those;
// scaling every radicand by 2^64 (the root by 2^-32, both exact here) needs no test and no special case for zero, which
// keeps its value through the not-a-number and subnormal neighbours the bit arithmetic gives it: 11 instructions
// instead of 16.  tools/probe/sqrt_check.hip compares it with the double-precision root rounded to float for zero and
// every float in (0, 1), subnormals included: no mismatch.
__device__ __forceinline__ float sqrt_rn_radicand(float x) {
    const float xs = x * 0x1p+64f;
    float s = __builtin_amdgcn_sqrtf(xs);
    const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float rd = __builtin_fmaf(-dn, s, xs), ru = __builtin_fmaf(-up, s, xs);
    s = rd <= 0.f ? dn : s;
    s = ru > 0.f ? up : s;
    return s * 0x1p-32f;
}

} // namespace ktlvo
