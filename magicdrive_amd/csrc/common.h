// common.h — device helpers shared by the gfx950 kernels of libmdx.
// CDNA4 only: wave64, MFMA bf16 (v_mfma_f32_32x32x16_bf16), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 storage (upper half of an fp32)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MDX_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even: hipcc lowers the __bf16 casts to v_cvt_pk_bf16_f32 on gfx950
// (one instruction per PAIR of values — the hand-written integer rounding was ~7 VALU ops per value).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<uint32_t*>(&v);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 v = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&v);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf GELU (attention.py:259-280 uses F.gelu's default, exact form).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the bf16 output rounding of 4e-3 relative): 1 rcp + 1 exp + 6 fma instead of the
// ~50-instruction libm erff — the GEGLU epilogue evaluates 8192 of these per 128x128 tile.
__device__ __forceinline__ float erf_as_f(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float y = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }

union Frag8 {
    uint4 u;
    bf16x8_t v;
    bf16_t h[8];
    uint2 d2[2];
};

// MFMA 32x32x16 bf16 C/D layout (cdna guide §3): lane l holds column (l & 31);
// register r holds row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
