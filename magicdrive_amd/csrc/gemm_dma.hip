// gemm_dma.hip — the pipelined main loop for GEMM / implicit-GEMM conv: operands go global -> LDS by
// LDS-DMA (`buffer_load_dwordx4 ... lds`), a STAGES-deep ring keeps STAGES-1 K-slabs in flight, waits are
// counted `s_waitcnt vmcnt(N)` (never a drain in steady state) and there is ONE raw `s_barrier` per slab.
//
// Why (cdna_hip_programming.md §5): the register-staged kernel in gemm_conv.hip spends its time waiting —
// one slab of prefetch, VGPR round trip, ds_write pass, a full drain at every barrier.  LDS-DMA removes the
// staging VGPRs and the ds_write pass and lets loads span barriers.
//
// LDS image: a slab is [rows][BK] bf16, UNPADDED (the DMA writes 64 lanes x 16 B = 1 KiB contiguous per wave
// instruction, so the LDS side must be linear).  Bank conflicts of the ds_read_b128 fragment reads are removed
// by an XOR swizzle applied on the SOURCE side: lane i of a piece fetches the 16-byte chunk that belongs at its
// linear LDS position, i.e. logical chunk c = c' ^ f(row) (rule 21: linear dest + swizzled source + the same
// swizzle on the read).  f(row) = (row>>2)&3 for BK=32 (4 chunks/row), (row>>1)&7 for BK=64: the 16 lanes of a
// ds_read_b128 group then cover 16 distinct 16-byte slots of the 256-byte bank row.
//
// Out-of-range rows / K tail / conv zero padding: the lane's voffset is set beyond the buffer descriptor's
// num_records, the hardware bounds check returns 0 and the DMA writes zeros — no branches, no stale LDS.
#include "common.h"
#include "launch.h"
#include "gemm_params.h"

namespace mdx {

constexpr unsigned OOB_OFF = 0xFFFFFF00u;      // >= num_records -> buffer load returns 0
constexpr unsigned NUM_RECORDS = 0x80000000u;  // 2 GiB window per operand (every tensor here is far smaller)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WM x WN waves; each wave owns a (BM/WM) x (BN/WN) sub-tile made of 32x32 MFMA tiles.
// Why big tiles: a CU's vector-memory path delivers ~64 B/clk, its four matrix pipes consume a 32x32x16 MFMA
// every 8 clk.  Per 16-deep k-step a BM x BN tile loads (BM+BN)*32 B and issues BM*BN/1024 MFMAs, so
// load-cycles / MFMA-cycles = 64 (BM+BN) / (BM BN): 1.0 for 128x128 (memory path saturated at <50% MFMA use),
// 0.75 for 256x128, 0.5 for 256x256.
template <int BM, int BN, int WM, int WN, int BK, int ST, bool CONV>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_kernel(GCParams p) {
    constexpr int NWV = WM * WN;
    constexpr int CPR = BK / 8;            // 16-byte chunks per row
    constexpr int RPP = 64 / CPR;          // rows per 1-KiB DMA piece
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NPA = BM / RPP / NWV, NPB = BN / RPP / NWV;   // pieces per wave per slab
    static_assert(NPA >= 1 && NPB >= 1, "tile too small for the wave count");
    constexpr int LPS = NPA + NPB;         // DMA instructions per wave per slab
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    int tile_m, tile_n;
    if (!tile_coords(p, tile_m, tile_n)) return;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    long zb = 0;
    int kz = 0;
    if (p.batch > 1) zb = blockIdx.z; else kz = blockIdx.z;
    const int kbeg = kz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg + BK - 1) / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + zb * p.sA), 0, NUM_RECORDS, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + zb * p.sW), 0, NUM_RECORDS, 0x00020000);

    // ---- per-piece source bookkeeping (fixed over the K loop except the conv tap cursor) ----
    const int crow = lane / CPR;           // row within the piece
    const int cphys = lane % CPR;          // physical chunk slot
    unsigned a_base[NPA];                  // GEMM: byte offset of (row, logical chunk) at k = 0; CONV: byte offset of batch image
    int a_c8[NPA];                         // logical chunk * 8 (element offset inside the slab)
    int a_iy0[NPA], a_ix0[NPA];
    bool a_ok[NPA];
    int a_ky[NPA], a_kx[NPA], a_ci[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int row = (wave * NPA + j) * RPP + crow;
        const int sw = (CPR == 4) ? ((row >> 2) & 3) : ((row >> 1) & 7);
        const int c = cphys ^ sw;
        a_c8[j] = c * 8;
        const int m = m0 + row;
        a_ok[j] = m < p.M;
        a_iy0[j] = a_ix0[j] = a_ky[j] = a_kx[j] = a_ci[j] = 0;
        if (CONV) {
            int mm = a_ok[j] ? m : 0;
            int hw = p.Ho * p.Wo;
            int b = mm / hw;
            int rem = mm - b * hw;
            int oy = rem / p.Wo;
            int ox = rem - oy * p.Wo;
            a_iy0[j] = oy * p.sh - p.ph;
            a_ix0[j] = ox * p.sw - p.pw;
            a_base[j] = (unsigned)((long)b * p.Hi * p.Wi * p.lda * 2);
            int kk = kbeg + c * 8;
            int tap = kk / p.Cin;
            a_ci[j] = kk - tap * p.Cin;
            a_ky[j] = tap / p.kw;
            a_kx[j] = tap - a_ky[j] * p.kw;
        } else {
            a_base[j] = (unsigned)(((long)(a_ok[j] ? m : 0) * p.lda + c * 8) * 2);
        }
    }
    unsigned b_base[NPB];
    int b_c8[NPB];
    bool b_ok[NPB];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const int row = (wave * NPB + j) * RPP + crow;
        const int sw = (CPR == 4) ? ((row >> 2) & 3) : ((row >> 1) & 7);
        const int c = cphys ^ sw;
        b_c8[j] = c * 8;
        const int n = n0 + row;
        b_ok[j] = n < p.N;
        b_base[j] = (unsigned)(((long)(b_ok[j] ? n : 0) * p.ldw + c * 8) * 2);
    }

    auto issue = [&](int t) {   // DMA slab t into ring slot t % ST
        unsigned char* sbase = smem + (t % ST) * STAGE_BYTES;
        const int k0 = kbeg + t * BK;
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            unsigned off = OOB_OFF;
            if (CONV) {
                const int iy = a_iy0[j] + a_ky[j], ix = a_ix0[j] + a_kx[j];
                if (a_ok[j] && (k0 + a_c8[j] < kend) && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                    off = a_base[j] + (unsigned)((((long)iy * p.Wi + ix) * p.lda + a_ci[j]) * 2);
                a_ci[j] += BK;
                while (a_ci[j] >= p.Cin) {
                    a_ci[j] -= p.Cin;
                    if (++a_kx[j] == p.kw) { a_kx[j] = 0; ++a_ky[j]; }
                }
            } else {
                if (a_ok[j] && (k0 + a_c8[j] < kend)) off = a_base[j] + (unsigned)(k0 * 2);
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(sbase + (wave * NPA + j) * 1024), 16, off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            unsigned off = OOB_OFF;
            if (b_ok[j] && (k0 + b_c8[j] < kend)) off = b_base[j] + (unsigned)(k0 * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(sbase + A_BYTES + (wave * NPB + j) * 1024), 16, off, 0, 0, 0);
        }
    };

    EpiRegs<BM, BN, TN, NWV * 64> er;
    const bool coalesced_out = p.splitk <= 1 && !p.c_f32;   // block-uniform
    if (coalesced_out) epi_prefetch<BM, BN, TN, NWV * 64>(p, zb, m0, n0, wn * TN * 32, lane, tid, er);

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: ST-1 slabs in flight ----
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
        if (s < nt) issue(s);

    const int frow = lane & 31;
    const int fhalf = lane >> 5;
    // fragment byte offsets inside a slab (swizzled), per M/N tile and k-step
    int a_foff[TM][BK / 16], b_foff[TN][BK / 16];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * TM * 32 + i * 32 + frow;
        const int sw = (CPR == 4) ? ((r >> 2) & 3) : ((r >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) a_foff[i][ks] = (r * CPR + ((ks * 2 + fhalf) ^ sw)) * 16;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * TN * 32 + j * 32 + frow;
        const int sw = (CPR == 4) ? ((r >> 2) & 3) : ((r >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) b_foff[j][ks] = A_BYTES + (r * CPR + ((ks * 2 + fhalf) ^ sw)) * 16;
    }

    for (int t = 0; t < nt; ++t) {
        // slab t must have landed: at most the (ST-2) younger slabs may still be in flight
        if (t + ST - 2 < nt) wait_vmcnt<(ST - 2) * LPS>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + ST - 1 < nt) issue(t + ST - 1);     // refill the slot every wave finished reading last iteration
        const unsigned char* sb = smem + (t % ST) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            Frag8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i].u = *(const uint4*)(sb + a_foff[i][ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j].u = *(const uint4*)(sb + b_foff[j][ks]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j].v, af[i].v, acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue (same contract as gemm_conv.hip) ----
    if (coalesced_out) {   // block-uniform: bf16 output goes through the LDS transpose (ring is dead now)
        epilogue_coalesced<BM, BN, TM, TN, NWV * 64>(p, zb, m0, n0, wm * TM * 32, wn * TN * 32, lane, tid, acc, smem, er);
        return;
    }
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 32 + i * 32 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wn * TN * 32 + j * 32 + 8 * g + 4 * half;
                if (nb >= p.N) continue;
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (p.splitk > 1) {
                    float* w = p.ws + ((long)kz * p.M + m) * p.N + nb;
                    *(float4*)w = make_float4(v[0], v[1], v[2], v[3]);
                } else if (p.epi == 1) {
                    if (TN == 2 && j == 0) {   // GEGLU needs a 64-wide wave sub-tile: [32 value | 32 gate]
                        float gte[4] = {acc[i][TN - 1][4 * g], acc[i][TN - 1][4 * g + 1], acc[i][TN - 1][4 * g + 2], acc[i][TN - 1][4 * g + 3]};
                        epilogue_store(p, zb, m, nb, v, gte);
                    }
                } else {
                    epilogue_store(p, zb, m, nb, v, nullptr);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, int ST, bool CONV>
static int launch_dma_one(const GCParams& p, hipStream_t st) {
    constexpr size_t ring = (size_t)ST * (BM + BN) * BK * 2, ctile = (size_t)BM * (BN + 8) * 2;   // the C tile reuses the ring
    constexpr size_t smem = ring > ctile ? ring : ctile;
    auto kern = gemm_dma_kernel<BM, BN, WM, WN, BK, ST, CONV>;
    if (int rc = ensure_dyn_smem((const void*)kern, smem, "dma")) return rc;
    GCParams q = p;
    q.mt = (p.M + BM - 1) / BM; q.nt = (p.N + BN - 1) / BN; q.swz = 0;
    dim3 grid((unsigned)(q.mt * q.nt), 1, p.batch > 1 ? p.batch : p.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, q);
    return check_launch("gemm_dma_kernel");
}

// tile ids: 0 = 128x128 (4 waves, ring 4), 1 = 256x128 (8 waves, ring 4), 2 = 256x256 (8 waves, ring 4),
//           3 = 64x128, 4 = 128x64, 5 = 64x64 (4 waves)
int launch_gemm_dma(const GCParams& p, bool conv, int tile, hipStream_t st) {
#define MDX_DMA(BM_, BN_, WM_, WN_, BK_, ST_) \
    (conv ? launch_dma_one<BM_, BN_, WM_, WN_, BK_, ST_, true>(p, st) : launch_dma_one<BM_, BN_, WM_, WN_, BK_, ST_, false>(p, st))
    switch (tile) {
        case 1: return MDX_DMA(256, 128, 4, 2, 32, 4);
        case 2: return MDX_DMA(256, 256, 2, 4, 32, 4);
        case 3: return MDX_DMA(64, 128, 2, 2, 32, 4);
        case 4: return MDX_DMA(128, 64, 2, 2, 32, 4);
        case 5: return MDX_DMA(64, 64, 2, 2, 32, 4);
        default: return MDX_DMA(128, 128, 2, 2, 32, 4);
    }
#undef MDX_DMA
}

void dma_tile_dims(int tile, int* bm, int* bn) {
    static const int d[6][2] = {{128, 128}, {256, 128}, {256, 256}, {64, 128}, {128, 64}, {64, 64}};
    *bm = d[tile][0]; *bn = d[tile][1];
}

}  // namespace mdx
