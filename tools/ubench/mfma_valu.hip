// Do MFMA and ordinary VALU instructions of DIFFERENT waves (and of one wave) overlap on a gfx950 SIMD?
// Each wave loops over { NM independent 32x32x16 bf16 MFMAs ; NV independent VALU ops }.  Printed: SIMD cycles per loop iteration
// (wall time x 2.1 GHz / iterations / waves per SIMD).  overlap  <=>  t(NM, NV) ~ max(t(NM, 0), t(0, NV)).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;
template <int NM, int NV, int VOP, int M16>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f16v acc[4]; f4v acc4[4];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = seed; for (int r = 0; r < 4; ++r) acc4[i][r] = seed; }
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x); b[i] = (__bf16)(seed * 2 + i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (M16) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i & 3], 0, 0, 0);
            else acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 3], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (VOP == 0) v[i & 15] = __builtin_fmaf(v[i & 15], 1.0001f, 0.5f);
            if (VOP == 1) v[i & 15] = __builtin_amdgcn_exp2f(v[i & 15]);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV, int VOP, int M16>
static void run(float* d, int wps) {
    const int iters = 4000;
    dim3 g(256 * wps), b(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, VOP, M16>), g, b, 0, 0, d, iters, 0.1f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, VOP, M16>), g, b, 0, 0, d, iters, 0.1f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD %d  mfma %s x%-2d  %s x%-2d : %7.1f SIMD-cycles per iteration (per wave-iteration)\n", wps, M16 ? "16x16x32" : "32x32x16", NM, VOP ? "v_exp" : "v_fma", NV,
           ms * 1e-3 * 2.1e9 / ((double)iters * wps));
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    for (int wps : {1, 2, 4}) {
        if (wps == 1) { run<4, 0, 0, 0>(d, 1); run<0, 32, 0, 0>(d, 1); run<4, 32, 0, 0>(d, 1); run<0, 16, 1, 0>(d, 1); run<4, 16, 1, 0>(d, 1); run<8, 0, 0, 1>(d, 1); run<8, 32, 0, 1>(d, 1); }
        if (wps == 2) { run<4, 0, 0, 0>(d, 2); run<0, 32, 0, 0>(d, 2); run<4, 32, 0, 0>(d, 2); run<0, 16, 1, 0>(d, 2); run<4, 16, 1, 0>(d, 2); run<8, 0, 0, 1>(d, 2); run<8, 32, 0, 1>(d, 2); }
        if (wps == 4) { run<4, 0, 0, 0>(d, 4); run<0, 32, 0, 0>(d, 4); run<4, 32, 0, 0>(d, 4); run<0, 16, 1, 0>(d, 4); run<4, 16, 1, 0>(d, 4); run<8, 0, 0, 1>(d, 4); run<8, 32, 0, 1>(d, 4); run<4, 64, 0, 0>(d, 4); run<0, 64, 0, 0>(d, 4); }
    }
    return 0;
}
