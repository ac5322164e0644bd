// Which VALU instructions hide in the shadow of an MFMA on gfx950?  Per wave: loop over 4 x { one 32x32x16 bf16 MFMA ; K independent
// ops of one kind } — the stream is interleaved by hand (asm volatile keeps the order).  Printed: SIMD cycles per {MFMA + K ops} unit
// at 1, 2 and 4 waves per SIMD, next to the MFMA-only and ops-only times.  hidden  <=>  t(both) ~ max(t(mfma), t(ops)).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_fill mfma_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
template <int OP>
__device__ __forceinline__ void op(float& a, float& b, unsigned& u, unsigned& w) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
    if (OP == 2) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));
    if (OP == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
    if (OP == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(w));
    if (OP == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a) : "v"(*(double*)&b));   // placeholder, not used
    if (OP == 9) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
    if (OP == 10) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
}
template <int OP, int K, int M>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed;
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 2 + i); }
    float v[8], c[8]; unsigned u[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed + threadIdx.x * 1e-3f + i; c[i] = 1.0001f + i * 1e-4f; u[i] = threadIdx.x + i; w[i] = threadIdx.x * 3 + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (M) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < K; ++j) op<OP>(v[j & 7], c[j & 7], u[j & 7], w[(j + 3) & 7]);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i] + (float)w[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP, int K, int M>
static double run(float* d, int wps) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, K, M>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 0.1f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, K, M>), dim3(256 * wps), dim3(256), 0, 0, d, iters, 0.1f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 * 2.1e9 / ((double)iters * 4 * wps);      // SIMD cycles (at 2.1 GHz) per unit
}
template <int OP>
static void row(float* d, const char* name) {
    for (int wps : {1, 2, 4}) {
        printf("%-20s waves/SIMD %d | mfma only %5.1f | 4 ops only %5.1f, +mfma %5.1f | 8 ops only %5.1f, +mfma %5.1f | 12 ops only %5.1f, +mfma %5.1f\n", name, wps,
               run<OP, 0, 1>(d, wps), run<OP, 4, 0>(d, wps), run<OP, 4, 1>(d, wps), run<OP, 8, 0>(d, wps), run<OP, 8, 1>(d, wps), run<OP, 12, 0>(d, wps), run<OP, 12, 1>(d, wps));
    }
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 24);
    row<0>(d, "v_fma_f32"); row<5>(d, "v_mul_f32"); row<7>(d, "v_add_f32"); row<11>(d, "v_sub_f32"); row<10>(d, "v_max_f32"); row<2>(d, "v_max3_f32"); row<1>(d, "v_exp_f32");
    row<3>(d, "v_cvt_pk_bf16_f32"); row<4>(d, "v_permlane32_swap"); row<9>(d, "v_permlane16_swap"); row<6>(d, "v_mov_b32");
    return 0;
}
