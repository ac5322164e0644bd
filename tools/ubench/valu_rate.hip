// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction of the ops the attention softmax is made of, at 1 and 4
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(2))) float f2;
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
                if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
                if (OP == 2) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), a[(i + 2) & 7]);
                if (OP == 3) { __attribute__((ext_vector_type(2))) __bf16 v = {(__bf16)a[i], (__bf16)a[(i + 1) & 7]}; u[i] ^= *(unsigned*)&v; }
                if (OP == 4) { auto rr = __builtin_amdgcn_permlane32_swap(u[i], u[(i + 1) & 7], false, false); u[i] = rr[0]; u[(i + 1) & 7] = rr[1]; }
                if (OP == 5) { auto rr = __builtin_amdgcn_permlane16_swap(u[i], u[(i + 1) & 7], false, false); u[i] = rr[0]; u[(i + 1) & 7] = rr[1]; }
                if (OP == 6) a[i] = a[i] * 1.0001f;
                if (OP == 7) { f2 v = {a[i], a[(i + 1) & 7]}; v = v * (f2){1.0001f, 0.9999f}; a[i] = v.x; a[(i + 1) & 7] = v.y; }
                if (OP == 8) a[i] = __shfl_xor(a[i], 32, 64);
                if (OP == 9) a[i] = (a[i] > 1.f) ? a[(i + 3) & 7] : a[i];
            }
    }
    long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long*)out)[100000] = t1 - t0;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_max3_f32", "v_cvt_pk_bf16_f32(+xor)", "v_permlane32_swap", "v_permlane16_swap", "v_mul_f32", "v_pk_mul_f32 (2 values)", "ds_bpermute (shfl_xor 32)", "v_cmp+v_cndmask"};
    const int iters = 2000;
    for (int blocksPerCU : {1, 4}) {
        printf("== %d workgroup(s) of 256 threads per CU (%d wave(s) per SIMD)\n", blocksPerCU, blocksPerCU);
        for (int op = 0; op < 10; ++op) {
            dim3 g(256 * blocksPerCU), b(256);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define L(OPN) case OPN: hipLaunchKernelGGL(k<OPN>, g, b, 0, 0, d, iters, 0.1f); hipEventRecord(e0); hipLaunchKernelGGL(k<OPN>, g, b, 0, 0, d, iters, 0.1f); hipEventRecord(e1); break;
            switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) }
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long cyc; hipMemcpy(&cyc, ((long*)d) + 100000, 8, hipMemcpyDeviceToHost);
            double n = (double)iters * 32;                     // instructions per wave (loop overhead ignored)
            printf("%-28s %7.2f shader cycles per wave-instruction (one wave's clock), wall %.3f ms -> %.2f SIMD-cycles per instr at %d waves/SIMD\n", names[op], cyc / n, ms,
                   ms * 1e-3 * 2.1e9 / (n * blocksPerCU), blocksPerCU);
        }
    }
    return 0;
}
