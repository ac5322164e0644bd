// L2 -> CU streaming ceiling on gfx950: LDS-DMA (buffer_load_dwordx4 ... lds) vs plain buffer loads into registers.
// Every workgroup (4 waves) streams a REGION-byte region `rounds` times; SHARE consecutive workgroups share one region (so it is
// L2 resident, like the K / V^T of one attention head shared by its query blocks).  Each wave keeps P 1-KiB pieces in flight.
// Prints aggregate TB/s and bytes/clk/CU (at 2.0 GHz).   Build: hipcc --offload-arch=gfx950 -O3 -o l2_stream l2_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned __attribute__((ext_vector_type(4))) rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    unsigned long long a = (unsigned long long)p; rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a); r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu); r.z = 0x80000000u; r.w = 0x00020000u; return r;
}
__device__ __forceinline__ void glds(const rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
template <int MODE, int P>
__global__ __launch_bounds__(256) void k(const char* src, int region, int share, int rounds, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * P * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware: workgroups sharing a region sit on one XCD
    const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
    const int reg = (idx / share) * 8 + xcd;
    const rsrc_t rs = make_rsrc(src + (size_t)reg * region);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)reg * region), 0, 0x7fffffff, 0x00020000);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem + wave * P * 1024;
    const int chunk = 4 * P * 1024;                      // bytes per workgroup iteration
    const int nit = region / chunk;
    unsigned acc = 0;
    uint4 v[P];
    for (int r = 0; r < rounds; ++r) {
        for (int it = 0; it < nit; ++it) {
            const int base = it * chunk + wave * P * 1024;
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < P; ++j) glds(rs, lds0 + j * 1024, lane * 16, base + j * 1024);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");      // the previous iteration's pieces have landed
            } else {
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    if (it > 0 || r > 0) acc ^= v[j].x ^ v[j].w;
                    { auto t_ = __builtin_amdgcn_raw_buffer_load_b128(rsb, lane * 16, base + j * 1024, 0); v[j] = *(uint4*)&t_; }
                }
                if (MODE == 2) {                                                 // + the ds_write a register-staged pipeline would do
#pragma unroll
                    for (int j = 0; j < P; ++j) *(uint4*)(smem + wave * P * 1024 + j * 1024 + lane * 16) = v[j];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) acc = *(unsigned*)(smem + threadIdx.x * 4);
    else for (int j = 0; j < P; ++j) acc ^= v[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE, int P>
static void run(const char* src, unsigned* out, int region, int share, int wgs_per_cu) {
    const int rounds = 8;
    const int nwg = 256 * wgs_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, P>), dim3(nwg), dim3(256), 0, 0, src, region, share, rounds, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, P>), dim3(nwg), dim3(256), 0, 0, src, region, share, rounds, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nwg * rounds * (region / (4 * P * 1024)) * (4 * P * 1024);
    printf("%-26s P=%d  %d WG/CU  region %4d KB shared by %2d WGs: %6.2f TB/s = %5.1f B/clk/CU\n", MODE == 0 ? "LDS-DMA dwordx4" : MODE == 1 ? "buffer_load_b128 -> VGPR" : "buffer_load_b128 + ds_write", P,
           wgs_per_cu, region >> 10, share, bytes / ms * 1e-9, bytes / (ms * 1e-3) / 256 / 2.0e9);
}
int main() {
    char* src; unsigned* out;
    (void)hipMalloc(&src, (size_t)1 << 30); (void)hipMemset(src, 1, (size_t)1 << 30); (void)hipMalloc(&out, 1 << 24);
    const int region = 192 << 10;
    for (int wpc : {1, 2, 4, 8}) {
        for (int share : {11, 1}) {
            run<0, 2>(src, out, region, share, wpc); run<0, 4>(src, out, region, share, wpc); run<0, 8>(src, out, region, share, wpc);
            run<1, 2>(src, out, region, share, wpc); run<1, 4>(src, out, region, share, wpc); run<1, 8>(src, out, region, share, wpc);
            run<2, 4>(src, out, region, share, wpc);
        }
    }
    return 0;
}
