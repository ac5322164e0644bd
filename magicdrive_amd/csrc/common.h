// common.h — device helpers shared by the gfx950 kernels of libmdx.
// CDNA4 only: wave64, MFMA (v_mfma_f32_32x32x16 / 16x16x32, bf16 or f16 operands), 160 KiB LDS per CU.
//
// The 16-bit storage / operand type is a BUILD parameter: every kernel source is compiled twice (csrc/Makefile), with MDX_F16 = 0
// (bf16: the default, namespace mdx, entry points mdx_*_bf16 / mdx_*) and with MDX_F16 = 1 (IEEE fp16 — what the reference samples in,
// magicdrive/misc/test_utils.py:95 — namespace mdx_f16 through -Dmdx=mdx_f16, entry points mdx_*_f16, launch.h).  Accumulation is fp32
// in both; the MFMA rate is the same.  `bf16_t` and the helpers bf2f / f2bf / pack2bf / add2bf keep their round-1 names: they mean
// "the 16-bit type of this build".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef MDX_F16
#define MDX_F16 0
#endif

typedef unsigned short bf16_t;  // raw 16-bit storage: bf16 (upper half of an fp32) or, in the MDX_F16 build, IEEE fp16
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MDX_WAVE 64

#if MDX_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;  // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;
#define MDX_ONE16 0x3C00u                                        // 1.0 in the 16-bit type
#define MDX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define MDX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float bf2f(bf16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
// fp32 -> fp16, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950 for the pair)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    bf16x2_t v = {(_Float16)lo, (_Float16)hi};
    return *reinterpret_cast<uint32_t*>(&v);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
    _Float16 v = (_Float16)f;
    return *reinterpret_cast<bf16_t*>(&v);
}
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define MDX_ONE16 0x3F80u
#define MDX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MDX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even: hipcc lowers the __bf16 casts to v_cvt_pk_bf16_f32 on gfx950
// (one instruction per PAIR of values — the hand-written integer rounding was ~7 VALU ops per value).
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<uint32_t*>(&v);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 v = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&v);
}
#endif

// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup: ~12 instructions)
__device__ __forceinline__ float silu_f(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }
// erf GELU (attention.py:259-280 uses F.gelu's default, exact form): gelu(x) = 0.5 x (1 + erf(x / sqrt 2)).
// erf(z) = z P(z^2) on |z| <= 3 (clamped beyond: erf(3) = 1 - 2.2e-5), P = the degree-8 minimax polynomial of erf(z) / z in z^2 (linear
// programme on 3000 points, tools/fit_erf.py): |error| <= 2.7e-5 in fp32 Horner arithmetic — 150x below the bf16 rounding of the
// product it feeds (2^-9 relative).  14 full-rate VALU operations and NO transcendental: the GEGLU epilogue of the level-0 feed-forward
// evaluates 1.4 G of these per launch and was VALU-bound on the round-2 form (Abramowitz-Stegun 7.1.26: 13 operations + v_rcp_f32 +
// v_exp_f32 at quarter rate = 21 issue slots; profiles/README.md round 3).  -DMDX_GELU_AS selects the old form (A/B side builds).
#ifdef MDX_GELU_AS
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
#else
__device__ __forceinline__ float erf_poly_f(float z) {
    const float zc = __builtin_amdgcn_fmed3f(z, -3.0f, 3.0f);
    const float u = zc * zc;
    float p = 4.074397617e-08f;
    p = __builtin_fmaf(p, u, -1.944883433e-06f);
    p = __builtin_fmaf(p, u, 4.106127751e-05f);
    p = __builtin_fmaf(p, u, -5.110412727e-04f);
    p = __builtin_fmaf(p, u, 4.235439367e-03f);
    p = __builtin_fmaf(p, u, -2.510287440e-02f);
    p = __builtin_fmaf(p, u, 1.110793533e-01f);
    p = __builtin_fmaf(p, u, -3.753149504e-01f);
    p = __builtin_fmaf(p, u, 1.128268531e+00f);
    return zc * p;
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float hx = 0.5f * x;
    return __builtin_fmaf(hx, erf_poly_f(x * 0.70710678118654752440f), hx);
}
#endif

// Two GELUs at once on packed fp32 (v_pk_mul_f32 / v_pk_fma_f32: one instruction, two lanes' worth of elements each): 15 instructions per PAIR (two
// clamps + 13 packed) instead of 2 x 14 — the GEGLU epilogues evaluate 1.4 G of these per level-0 launch and are VALU-bound (profiles/README.md round 3).
// Same polynomial, same operation order per element as gelu_erf_f: bit-identical results.  -DMDX_GELU_PK=0 keeps the scalar form (A/B side builds).
#ifndef MDX_GELU_PK
#define MDX_GELU_PK 1
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#ifndef MDX_GELU_AS
__device__ __forceinline__ f32x2_t gelu_erf_f2(f32x2_t x) {
#if MDX_GELU_PK
    const f32x2_t z = x * 0.70710678118654752440f;
    f32x2_t zc;
    zc.x = __builtin_amdgcn_fmed3f(z.x, -3.0f, 3.0f); zc.y = __builtin_amdgcn_fmed3f(z.y, -3.0f, 3.0f);
    const f32x2_t u = zc * zc;
    auto c2 = [](float c) { return f32x2_t{c, c}; };
    f32x2_t p = c2(4.074397617e-08f);
    p = __builtin_elementwise_fma(p, u, c2(-1.944883433e-06f));
    p = __builtin_elementwise_fma(p, u, c2(4.106127751e-05f));
    p = __builtin_elementwise_fma(p, u, c2(-5.110412727e-04f));
    p = __builtin_elementwise_fma(p, u, c2(4.235439367e-03f));
    p = __builtin_elementwise_fma(p, u, c2(-2.510287440e-02f));
    p = __builtin_elementwise_fma(p, u, c2(1.110793533e-01f));
    p = __builtin_elementwise_fma(p, u, c2(-3.753149504e-01f));
    p = __builtin_elementwise_fma(p, u, c2(1.128268531e+00f));
    const f32x2_t e = zc * p;
    const f32x2_t hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
#else
    return f32x2_t{gelu_erf_f(x.x), gelu_erf_f(x.y)};
#endif
}
#else
__device__ __forceinline__ f32x2_t gelu_erf_f2(f32x2_t x) { return f32x2_t{gelu_erf_f(x.x), gelu_erf_f(x.y)}; }
#endif

// Two 16-bit products + fp32 accumulate in one instruction (v_dot2_f32_bf16 / v_dot2_f32_f16), no unpacking: a and b hold two values of the
// build's 16-bit type each.  The products are exact in fp32.
__device__ __forceinline__ float dot2_16(unsigned a, unsigned b, float c) {
#if MDX_F16
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), c, false);
#else
    typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, a), __builtin_bit_cast(b2_t, b), c, false);
#endif
}

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
