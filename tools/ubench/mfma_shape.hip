// mfma_shape.hip — which MFMA shape is cheaper at the power limit?  gemm_xl.hip runs v_mfma_f32_16x16x32_bf16 (its 80-column wave tiles are
// not a multiple of 32), gemm_ws / gemm_conv and the ubench loops run v_mfma_f32_32x32x16_bf16.  On this chip a pure MFMA stream on random data is
// bound by the power budget (profiles/r03_ubench_gemm4w.log: 2.41 PFLOP/s on zeros, 1.66-1.81 on random operands), so the shape that moves
// fewer register-file bytes per flop should clock higher:
//     32x32x16: 2 KB of A / B operands + 2 x 4 KB of accumulators per 32768 flop;   16x16x32: 2 KB of operands + 2 x 1 KB of accumulators per 16384 flop.
// Both variants cover the same 128 x 64 wave tile (128 fp32 accumulators in AGPRs) with 12 operand fragments of a 32-deep k-step held in
// registers (random bf16, loaded once); no LDS, no memory traffic in the loop.  Block 0 stamps s_memtime / s_memrealtime: effective clock.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
union Frag8 { uint4 u; bf16x8_t v; };

#define MFMA_32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define MFMA_16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

// SHAPE 32: per 32-deep step 2 k-halves x (4 A x 2 B) = 16 MFMAs of 32768 flop.  SHAPE 16: 8 A x 4 B = 32 MFMAs of 16384 flop.  Same flops, same tile.
template <int SHAPE, int WPS>
__global__ __launch_bounds__(WPS * 256) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
void mfma_shape_kernel(const uint4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamp, int iters) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    Frag8 fr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) fr[i].u = src[(blockIdx.x % 64) * 12 * 512 + i * 512 + (threadIdx.x & 511)];
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)                                 // fragments 0-3 / 6-9: A of the two 16-deep halves, 4-5 / 10-11: B
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n) MFMA_32(acc[i][n], fr[kh * 6 + 4 + n].v, fr[kh * 6 + i].v);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n) s += acc[i][n][0] + acc[i][n][15];
    } else {
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][n][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)                                    // fragments 0-7: A (16 rows x 32 k each), 8-11: B
#pragma unroll
                for (int n = 0; n < 4; ++n) MFMA_16(acc[i][n], fr[8 + n].v, fr[i].v);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int n = 0; n < 4; ++n) s += acc[i][n][0] + acc[i][n][3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stamp[0] = __builtin_amdgcn_s_memtime() - c0;
        stamp[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int SHAPE, int WPS>
static void run(const char* what, const uint4* src, float* out, unsigned long long* stamp, int iters, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_shape_kernel<SHAPE, WPS>), dim3(blocks), dim3(WPS * 256), 0, 0, src, out, stamp, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_shape_kernel<SHAPE, WPS>), dim3(blocks), dim3(WPS * 256), 0, 0, src, out, stamp, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
    const double flops = (double)blocks * WPS * 4 * iters * 16.0 * 32768.0;   // per wave and iteration: 16 x 32768 = 32 x 16384
    const double tf = flops / (ms / reps * 1e-3) / 1e12;
    printf("  %-46s %7.1f TFLOP/s @ %.2f GHz  (MFMA busy %.2f)\n", what, tf, ghz, tf * 1e12 / (256.0 * 4096.0 * ghz * 1e9));
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

int main() {
    const size_t n16 = (size_t)64 * 12 * 512;                              // uint4 fragments
    std::vector<unsigned short> h(n16 * 8);
    uint4* src; float* out; unsigned long long* stamp;
    (void)hipMalloc(&src, n16 * 16); (void)hipMalloc(&out, 4096); (void)hipMalloc(&stamp, 64);
    for (int pass = 0; pass < 3; ++pass) {
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 9) & 0xffff) / 65536.0f - 0.5f; };
        // pass 0: random values of O(1) (what the ubench GEMMs use); pass 1: activation-like (half of them exactly zero, post-SiLU-ish small magnitudes); pass 2: zeros
        for (auto& v : h) {
            const float x = rnd();
            v = pass == 0 ? f2bf(x) : pass == 1 ? f2bf(x < 0.f ? 0.f : x * 0.25f) : (unsigned short)0;
        }
        (void)hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice);
        printf("%s operands, MFMA stream only (128 accumulators per wave, 12 resident operand fragments):\n",
               pass == 0 ? "random" : pass == 1 ? "half-zero small" : "all-zero");
        const int iters = 4000;
        run<32, 2>("32x32x16, 2 waves / SIMD", src, out, stamp, iters, 1024);
        run<16, 2>("16x16x32, 2 waves / SIMD", src, out, stamp, iters, 1024);
        run<32, 1>("32x32x16, 1 wave / SIMD", src, out, stamp, iters, 2048);
        run<16, 1>("16x16x32, 1 wave / SIMD", src, out, stamp, iters, 2048);
    }
    return 0;
}
