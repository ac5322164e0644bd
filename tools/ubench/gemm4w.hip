// gemm4w.hip — experiment for the round-4 XL main loop: a 256 x 256 tile computed by FOUR waves (2 x 2), each owning a 128 x 128
// quadrant whose 256 fp32 accumulators live in AGPRs (one wave per SIMD: 512 registers per lane on gfx950's unified file).
//
// Why try it: gemm_xl.hip runs 8 waves with 128-160 accumulators each in VGPRs; per 64-deep slab every wave reads (128 + 64) x 128 B of
// fragments = 197 KB per CU, and two waves per SIMD interleave MFMAs with each other's load phases.  With 128 x 128 per wave the
// fragment traffic is 131 KB per slab, the barrier couples 4 waves instead of 8, and each SIMD's MFMA stream comes from one wave.
// Unknown (hence a microbenchmark): whether ONE wave per SIMD keeps the matrix pipe fed between its own ds_reads / DMA issues.
//
// C[m][n] = sum_k A[m][k] W[n][k]  (bf16 in, fp32 accumulate, bf16 out), M, N multiples of 256, K a multiple of BK.
// Operands global -> LDS by LDS-DMA (1-KiB lane-linear pieces, XOR swizzle on the SOURCE address and on the fragment reads:
// the scheme of gemm_ws.hip for 128-byte rows; for 64-byte rows the chunk is XORed with (row >> 2) & 3 — each ds_read_b128
// lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} then covers all 16 sixteen-byte slots of the 256-byte bank row).
// A ring of NSTG units of BK columns, ONE barrier per unit, counted vmcnt (loads of the next NSTG-1 units stay in flight):
//   variant A: BK = 64, NSTG = 2 (2 x 64 KB): the next slab has one slab of MFMAs (~1 us) to land
//   variant B: BK = 32, NSTG = 4 (4 x 32 KB): three half-slabs of lead (~1.5 us), but 64-byte global segments per row
// Fragment reads run one k-step ahead of their MFMAs (two register sets).
//
// Build:  hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip        (NO -mllvm -amdgpu-mfma-vgpr-form: the accumulators belong in AGPRs)
// Run:    ./gemm4w            prints TFLOP/s per shape and variant, with and without the store epilogue, and a sampled check vs the host
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;

union Frag8 { uint4 u; bf16x8_t v; };

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = 0xfffffff0u;                                       // num_records: offsets >= this return 0 (the past-the-end units of the ring)
    r.w = 0x00020000u;
    return r;
}
// one 1-KiB piece: lane l's 16 bytes land at lds_addr + 16 l; source = base + voff (per lane, fixed for the launch) + soff (the unit's k offset).
// m0 is clobbered, not saved: nothing else in this kernel uses it (three scalar-side instructions per piece, no VALU).
__device__ __forceinline__ void glds(const rsrc_t rs, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
        : "memory", "m0");
}
// The MFMAs are volatile asm statements so that they and the DMA issues keep EXACTLY the source order (a pure builtin floats past the
// DMA statements at instruction selection: all 16 pieces of a unit ended up in one block in front of the unit's 64 MFMAs — with one
// wave per SIMD that is ~400 cycles of idle matrix pipe per unit).  The compiler still places the fragment reads and their waits.
#define MFMA_32x32x16(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<unsigned*>(&v);
}


// NW = 4: 2 x 2 waves of 128 x 128 (one wave per SIMD).  NW = 8: 2 x 4 waves of 128 x 64 (two waves per SIMD, 128 accumulators each) —
// the "plain" 8-wave form of the same loop, for comparison with gemm_xl.hip's staggered quadrant phases.
// ABL (ablations; results are wrong, timing only): 1 = no DMA inside the loop (the stages keep the first units), 2 = no barrier,
// 4 = no fragment reads inside the loop (MFMA stream only).
template <int NW, int BK, int NSTG, bool STORE, int ABL>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C, int M, int N, int K, int nt) {
    constexpr int ROWB = BK * 2;                 // bytes per LDS row (128 / 64)
    constexpr int CPR = BK / 8;                  // 16-byte chunks per row (8 / 4)
    constexpr int RPP = 1024 / ROWB;             // rows per DMA piece (8 / 16)
    constexpr int REG = 256 * ROWB;              // bytes of the A (or W) half of a unit
    constexpr int UNIT = 2 * REG;                // 64 KB / 32 KB
    constexpr int PPW = UNIT / 1024 / NW;        // pieces per wave per unit (4 waves: 16 / 8)
    constexpr int KS = BK / 16;                  // MFMA k-steps per unit (4 / 2)
    constexpr int WNC = NW / 2;                  // waves along N
    constexpr int NN = 4 / (NW / 4);             // 32-column accumulator tiles per wave along N (4 / 2)
    constexpr int SLOTS = KS * 4;                // {NN MFMAs} groups per unit; one DMA piece behind every SLOTS / PPW-th group
    static_assert(SLOTS % PPW == 0, "pieces spread evenly over the MFMA groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WNC, wn = wave % WNC;
    const int frow = lane & 31, half = lane >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8; an XCD walks the N-tiles of consecutive M-tiles
    const int nblk = gridDim.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = blockIdx.x & 7;          // XCD x owns a contiguous run of xq (+1 for x < xr) tiles
    const int bid = xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3);
    const int tm = bid / nt, tn = bid - tm * nt;
    const int m0 = tm * 256, n0 = tn * 256;

    const bool loadsA = wave < NW / 2;                                     // the first half of the waves loads the A rows, the second the W rows
    rsrc_t rs = make_rsrc(loadsA ? (const void*)A : (const void*)W);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    // swizzle of the 16-byte chunk index by the row: 128-byte rows (row >> 1) & 7, 64-byte rows (row >> 2) & 3
    auto swz = [](int row) { return CPR == 8 ? (row >> 1) & 7 : (row >> 2) & 3; };

    // ---- DMA: this wave's PPW pieces of a unit.  Pieces 0 .. UNIT/2048-1 are A rows, the rest W rows; waves 0,1 load A, waves 2,3 load W.
    unsigned src_off[PPW];                       // byte offset of this lane's source chunk at k = 0
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int piece = (wave % (NW / 2)) * PPW + j;           // piece within the A (or W) region
        const int row = piece * RPP + lane / CPR;                // 0..255
        const int cp = lane % CPR;
        const int c = cp ^ swz(row);
        const long grow = (loadsA ? m0 : n0) + row;
        src_off[j] = (unsigned)(grow * (long)K * 2 + c * 16);    // operands < 4 GiB (host checks)
    }
    const unsigned dst_reg = (loadsA ? 0 : REG) + (wave % (NW / 2)) * PPW * 1024;
    const int T = K / BK;
    // piece j of unit t (past-the-end units: num_records = 0 -> the DMA writes zeros; the vmcnt bookkeeping stays uniform)
    auto issue_piece = [&](int t, int j) {
        rsrc_t r = rs;
        r.z = t < T ? 0xfffffff0u : 0u;
        glds(r, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(t % NSTG) * UNIT + dst_reg + j * 1024), src_off[j], (unsigned)t * ROWB);
    };

    f32x16_t acc[4][NN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

#pragma unroll
    for (int t = 0; t < NSTG - 1 + (ABL & 1); ++t)
#pragma unroll
        for (int j = 0; j < PPW; ++j) issue_piece(t, j);
    if (ABL & 1) { wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }

    const int x0 = half ^ swz(frow);                                       // physical chunk of k-step 0; k-step j: x0 ^ (j << 1)
    const unsigned a_rd = (unsigned)((wm * 128 + frow) * ROWB);            // + i * 32 * ROWB
    const unsigned w_rd = (unsigned)(REG + (wn * 32 * NN + frow) * ROWB);
    Frag8 af[2][4], wf[2][NN];
    if (ABL & 4) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[b][i].u = *(const uint4*)(smem + a_rd + i * 32 * ROWB + (x0 << 4));
#pragma unroll
            for (int n = 0; n < NN; ++n) wf[b][n].u = *(const uint4*)(smem + w_rd + n * 32 * ROWB + (x0 << 4));
        }
    }

    for (int t = 0; t < T; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 1)) wait_vmcnt<(NSTG - 2) * PPW>();                    // unit t's pieces of this wave have landed
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();                      // ... everyone's; and everyone is done reading unit t - 1
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ub = smem + (t % NSTG) * UNIT;
        if (!(ABL & 4)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][i].u = *(const uint4*)(ub + a_rd + i * 32 * ROWB + (x0 << 4));
#pragma unroll
            for (int n = 0; n < NN; ++n) wf[0][n].u = *(const uint4*)(ub + w_rd + n * 32 * ROWB + (x0 << 4));
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS && !(ABL & 4)) {                               // fragment reads one k-step ahead (second register set)
                const int co = (x0 ^ ((ks + 1) << 1)) << 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) af[(ks + 1) & 1][i].u = *(const uint4*)(ub + a_rd + i * 32 * ROWB + co);
#pragma unroll
                for (int n = 0; n < NN; ++n) wf[(ks + 1) & 1][n].u = *(const uint4*)(ub + w_rd + n * 32 * ROWB + co);
            }
            asm volatile("" ::: "memory");                                 // the reads above stay above this k-step's MFMAs
            // D[n'][m'] orientation (W fragment as the A operand): lane = token row m', registers = 4-column groups of n' -> 8-byte row stores.
            // The DMA pieces of unit t + NSTG - 1 (it refills the stage unit t - 1 occupied) are spread over the unit's MFMA groups.
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int n = 0; n < NN; ++n) MFMA_32x32x16(acc[i][n], wf[ks & 1][n].v, af[ks & 1][i].v);
                const int slot = ks * 4 + i;
                if (!(ABL & 1) && slot % (SLOTS / PPW) == 0) issue_piece(t + NSTG - 1, slot / (SLOTS / PPW));
            }
        }
    }
    wait_vmcnt<0>();                                                       // zero-fill tail
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                     // the last MFMAs' results vs the compiler's v_accvgpr_read (it cannot see into the asm)
    if (STORE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + wm * 128 + i * 32 + frow;
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                bf16_t* crow = C + m * (long)N + n0 + wn * 32 * NN + n * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 o;
                    o.x = pack2bf(acc[i][n][4 * g], acc[i][n][4 * g + 1]);
                    o.y = pack2bf(acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]);
                    *(uint2*)(crow + 8 * g) = o;
                }
            }
        }
    } else {
        // keep the accumulators alive: one value per lane
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < NN; ++n) s += acc[i][n][0] + acc[i][n][15];
        if (s == 12345.678f) C[threadIdx.x] = 1;
    }
}

static float bf2f(bf16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

template <int NW, int BK, int NSTG, bool STORE, int ABL = 0>
static double run(const bf16_t* dA, const bf16_t* dW, bf16_t* dC, int M, int N, int K, int reps) {
    const int mt = M / 256, nt = N / 256, nblk = mt * nt;
    const size_t smem = (size_t)NSTG * 2 * 256 * BK * 2;
    auto kern = gemm4w_kernel<NW, BK, NSTG, STORE, ABL>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(NW * 64), smem, 0, dA, dW, dC, M, N, K, nt);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(nblk), dim3(NW * 64), smem, 0, dA, dW, dC, M, N, K, nt);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(e)); return 0; }
    return 2.0 * M * N * (double)K / (ms / reps * 1e-3) / 1e12;
}

int main() {
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {
        {268800, 1280, 640, "qk L1 (768 views)"}, {69888, 2560, 1280, "qk L2"}, {69888, 1280, 5120, "ff.out L2"}, {8192, 8192, 8192, "square 8k"},
    };
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (auto& s : shapes) {
        maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N);
        if ((size_t)s.M * s.K * 2 >= 0xfffffff0ull || (s.M % 256) || (s.N % 256) || (s.K % 64)) { printf("bad shape %s\n", s.what); return 1; }
    }
    std::vector<bf16_t> hA(maxA), hW(maxW);
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 9) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hW) v = f2bf(rnd() * 0.25f);
    bf16_t *dA, *dW, *dC;
    (void)hipMalloc(&dA, maxA * 2); (void)hipMalloc(&dW, maxW * 2); (void)hipMalloc(&dC, maxC * 2);
    (void)hipMemcpy(dA, hA.data(), maxA * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dW, hW.data(), maxW * 2, hipMemcpyHostToDevice);
    std::vector<bf16_t> hC(4096);
    for (auto& s : shapes) {
        const int reps = 4;
        // correctness of the four complete variants on sampled outputs (the operands are the leading M x K / N x K elements of the buffers)
        for (int variant = 0; variant < 4; ++variant) {
            (void)hipMemset(dC, 0xff, (size_t)s.M * s.N * 2);
            if (variant == 0) run<4, 64, 2, true>(dA, dW, dC, s.M, s.N, s.K, 1);
            else if (variant == 1) run<4, 32, 4, true>(dA, dW, dC, s.M, s.N, s.K, 1);
            else if (variant == 2) run<8, 64, 2, true>(dA, dW, dC, s.M, s.N, s.K, 1);
            else run<8, 32, 4, true>(dA, dW, dC, s.M, s.N, s.K, 1);
            double worst = 0;
            for (int q = 0; q < 48; ++q) {
                const long m = ((long)q * 7919 + (q % 3 == 0 ? s.M - 1 - q : 0)) % s.M, n = ((long)q * 104729 + (q % 5 == 0 ? s.N - 1 : 0)) % s.N;
                bf16_t got; (void)hipMemcpy(&got, dC + m * (long)s.N + n, 2, hipMemcpyDeviceToHost);
                double ref = 0;
                for (int k = 0; k < s.K; ++k) ref += (double)bf2f(hA[m * (long)s.K + k]) * bf2f(hW[n * (long)s.K + k]);
                const double err = fabs(bf2f(got) - ref) / (fabs(ref) + 0.05 * sqrt((double)s.K) * 0.07);
                worst = std::max(worst, err);
            }
            printf("%-20s %d waves, %s: worst sampled relative error %.4f %s\n", s.what, variant < 2 ? 4 : 8, variant & 1 ? "BK 32 x 4" : "BK 64 x 2", worst, worst < 2e-2 ? "ok" : "MISMATCH");
        }
        printf("%-20s M=%6d N=%5d K=%5d  TFLOP/s: full | no stores | + no DMA in loop | + no barrier | MFMA stream only\n", s.what, s.M, s.N, s.K);
#define ROW(NW_, BK_, NS_)                                                                                                               \
        printf("    %d waves, BK %2d x %d :  %7.1f | %7.1f | %7.1f | %7.1f | %7.1f\n", NW_, BK_, NS_,                                      \
               run<NW_, BK_, NS_, true, 0>(dA, dW, dC, s.M, s.N, s.K, reps), run<NW_, BK_, NS_, false, 0>(dA, dW, dC, s.M, s.N, s.K, reps), \
               run<NW_, BK_, NS_, false, 1>(dA, dW, dC, s.M, s.N, s.K, reps), run<NW_, BK_, NS_, false, 3>(dA, dW, dC, s.M, s.N, s.K, reps), \
               run<NW_, BK_, NS_, false, 7>(dA, dW, dC, s.M, s.N, s.K, reps));
        ROW(4, 64, 2) ROW(4, 32, 4) ROW(8, 64, 2) ROW(8, 32, 4)
        if (&s == &shapes[sizeof(shapes) / sizeof(shapes[0]) - 1]) {
            // the same loops on ALL-ZERO operands: the matrix pipe's power draw depends on the data, and the clock on the power
            (void)hipMemset(dA, 0, maxA * 2); (void)hipMemset(dW, 0, maxW * 2);
            printf("%-20s ... on all-zero operands (power, not the schedule, sets the random-data rates above if these are higher)\n", s.what);
            ROW(4, 64, 2) ROW(8, 64, 2)
        }
#undef ROW
    }
    return 0;
}
