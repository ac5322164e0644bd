// gemm_pp.hip — large-tile "ping-pong" bf16 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
// Why a second main loop: with 128x128 tiles (gemm_conv.hip) the LDS, not the matrix pipe, is the bound.  Per 64-deep slab
// and per pair of resident workgroups the CU spends 830 cycles on ds_write_b128 (13 cycles per wave-instruction,
// MI355X_MICROARCH.md §LDS) + 512 on ds_read_b128 against 1024 cycles of MFMA issue; measured 2.1-2.4 k cycles per slab.
// Operand traffic through LDS per FLOP falls with (BM*BN)/(BM+BN), so this kernel uses the largest tile the register
// file allows (256 x 256 or 256 x 320: 128 / 160 fp32 accumulators per lane at 2 waves per SIMD) and hides what LDS
// traffic remains structurally:
//
//   * 8 waves = two groups of 4; waves w and w+4 share a SIMD.  Every slab has two phases separated by s_barrier:
//       phase 2s   : group 0 multiplies slab s        | group 1 writes its half of slab s+1 to the other LDS buffer
//       phase 2s+1 : group 0 writes its half of s+1   | group 1 multiplies slab s
//     so each SIMD's matrix pipe always has exactly one wave issuing MFMAs back to back while its partner wave does the
//     global->register->LDS traffic ("matrix beside memory", MI355X_MICROARCH.md "Two waves per SIMD").
//   * LDS rows are 128 B (one 64-deep k slab) with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7: conflict-free
//     for the ds_read_b128 lane groups {0-3,12-15,20-27 | 4-11,16-19,28-31} and for 8-lane row writes, no padding, which is
//     what lets two 256+320-row buffers fit into 160 KiB.
//   * Epilogue: bias (+ the per-(step, batch) temb row) is staged once per tile into LDS as fp32 "addend" rows, applied to
//     the accumulators in registers with activation / GEGLU, the bf16 tile is transposed through LDS in two 128-row halves
//     and written as full row segments with the residual added (same arithmetic per element as gemm_conv.hip).
//
// Used for the big shapes only (cost model in gemm_conv.hip: N % 256 == 0, K >= 1024, enough tiles); everything else stays on
// gemm_conv.hip / conv3x3.hip / gemm_ws.hip.  Same GCParams, same reduction order per output element (k ascending in 16-wide
// MFMA steps, fp32 accumulation).
#include "common.h"
#include "launch.h"
#include "gemm_params.h"
#include <cstdlib>

namespace mdx {

constexpr int PP_SLOTS = 8;     // distinct temb rows (batch entries) one 256-row tile may span

template <int WM, int WN, int TM, int TN, bool CONV, bool EARLY>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GCParams p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTH = 512;
    constexpr int A_CH = BM * 8 / NTH, B_CH = BN * 8 / NTH;     // 16-byte chunks per thread per slab
    static_assert(WM * WN == 8 && (BM * 8) % NTH == 0 && (BN * 8) % NTH == 0, "tile / wave layout");
    static_assert(BM == 256, "the two-half epilogue assumes 256 tile rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* As = (bf16_t*)smem;                     // [2][BM][64], chunk-swizzled
    bf16_t* Bs = As + 2 * BM * 64;                  // [2][BN][64]
    float* addend = (float*)(Bs + 2 * BN * 64);     // [PP_SLOTS][BN] fp32

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int grp = wave >> 2;                      // ping-pong group; waves w and w+4 sit on the same SIMD
    const int wm = wave % WM, wn = wave / WM;
    int tile_m, tile_n;
    if (!tile_coords(p, tile_m, tile_n)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nt = (p.K + 63) / 64;

    // ---- global -> register staging: thread owns chunk kc of rows rbase + 64 i ----
    const int kc = tid & 7, rbase = tid >> 3;
    const int wchunk = (kc ^ ((rbase >> 1) & 7)) << 3;          // swizzled element offset inside a 64-element LDS row
    // Loads are UNCONDITIONAL (clamped addresses): rows past M / N only feed accumulator rows / columns that are never
    // stored, so they may hold anything; K is a multiple of 64 here (pp_supported); only conv padding taps must read as
    // zero and are masked after the load.  (Predicated loads put every global_load into its own basic block; measured
    // 150 cycles per load in the load phase.)
    long a_off[A_CH];            // GEMM: element offset of the (clamped) row; CONV: element offset of the image (b * Hi * Wi * lda)
    int a_yx[A_CH];              // CONV: ((iy0 + 64) << 16) | (ix0 + 64)
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = min(m0 + rbase + 64 * i, p.M - 1);
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;
            a_off[i] = (long)b * p.Hi * p.Wi * p.lda;
            a_yx[i] = (int)(((unsigned)(iy0 + 64) << 16) | (unsigned)(ix0 + 64));
        } else {
            a_off[i] = (long)m * p.lda;
            a_yx[i] = 0;
        }
    }
    int ky = 0, kx = 0, ci = 0;
    const bool cim = CONV && p.cimajor;             // channel-block-major slab order (gemm_params.h)
    if (CONV) {
        const int kk = kc * 8;
        const int tap = cim ? 0 : kk / p.Cin;
        ci = kk - tap * p.Cin;
        ky = tap / p.kw;
        kx = tap - ky * p.kw;
    }
    long w_off[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) w_off[i] = (long)min(n0 + rbase + 64 * i, p.N - 1) * p.ldw;

    uint4 a_reg[A_CH], b_reg[B_CH];
    // (macros, not lambdas taking the chunk index: the staging arrays must stay in registers)
#define PP_LOAD_A(i)                                                                                          \
    if (CONV) {                                                                                               \
        const int iy = (a_yx[i] >> 16) - 64 + ky, ix = (a_yx[i] & 0xffff) - 64 + kx;                          \
        const bool inb = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;                      \
        const int cy = min(max(iy, 0), p.Hi - 1), cx = min(max(ix, 0), p.Wi - 1);                             \
        const uint4 v = *(const uint4*)(p.A + a_off[i] + ((long)cy * p.Wi + cx) * p.lda + ci);               \
        a_reg[i] = inb ? v : make_uint4(0, 0, 0, 0);                                                          \
    } else {                                                                                                  \
        const uint4 v = *(const uint4*)(p.A + a_off[i] + kk_cur);                                             \
        a_reg[i] = make_uint4(v.x, v.y, v.z, v.w);                                                            \
    }
#define PP_LOAD_B(i) { const uint4 v = *(const uint4*)(p.W + w_off[i] + kk_cur); b_reg[i] = make_uint4(v.x, v.y, v.z, v.w); }
    auto slab_end = [&]() {
        if (CONV) {
            if (cim) {
                if (++kx == p.kw) { kx = 0; if (++ky == p.kh) { ky = 0; ci += 64; } }
            } else {
                ci += 64;
                while (ci >= p.Cin) {
                    ci -= p.Cin;
                    if (++kx == p.kw) { kx = 0; ++ky; }
                }
            }
        }
    };
    auto load_tile = [&](int t) {
        if (p.dbg & 2) return;
        const int kk_cur = cim ? (ky * p.kw + kx) * p.Cin + ci : t * 64 + kc * 8;   // k offset of this thread's chunk
#pragma unroll
        for (int i = 0; i < A_CH; ++i) { PP_LOAD_A(i) }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) { PP_LOAD_B(i) }
        slab_end();
    };
    bf16_t* const as_w = As + rbase * 64 + wchunk;
    bf16_t* const bs_w = Bs + rbase * 64 + wchunk;
    auto store_tile = [&](int buf) {
        if (p.dbg & 1) return;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) *(uint4*)(as_w + buf * BM * 64 + i * 64 * 64) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) *(uint4*)(bs_w + buf * BN * 64 + i * 64 * 64) = b_reg[i];
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, half = lane >> 5;
    const int x0 = half ^ ((frow >> 1) & 7);                    // swizzled chunk of k-step 0; k-step ks flips bits 1-2
    const int a_rd = (wm * TM * 32 + frow) * 64, b_rd = (wn * TN * 32 + frow) * 64;
    // One compute phase = 4 k-steps of TM*TN MFMAs.  The partner wave on this SIMD is in its load phase, so nothing else
    // fills the matrix pipe while this wave waits on LDS: the fragment reads of k-step ks+1 are issued (into the other
    // register set) BEFORE the MFMAs of k-step ks and pinned there with sched_group_barrier.
    constexpr bool DB = TM * TN <= 8;               // the 256 x 320 tile has no registers left for a second fragment set
    auto compute = [&](int buf) {
        if (p.dbg & 4) return;
        const bf16_t* as = As + buf * BM * 64 + a_rd;
        const bf16_t* bs = Bs + buf * BN * 64 + b_rd;
        if constexpr (!DB) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int co = (x0 ^ (ks << 1)) << 3;
                Frag8 a1[TM], b1[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a1[i].u = *(const uint4*)(as + i * 32 * 64 + co);
#pragma unroll
                for (int j = 0; j < TN; ++j) b1[j].u = *(const uint4*)(bs + j * 32 * 64 + co);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1[j].v, a1[i].v, acc[i][j], 0, 0, 0);
            }
            return;
        } else {
            Frag8 af[2][TM], bfr[2][TN];
#define PP_READ_FRAGS(set, ks_)                                                                   \
            {                                                                                     \
                const int co = (x0 ^ ((ks_) << 1)) << 3;                                          \
                af[set][0].u = *(const uint4*)(as + co);                                          \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bfr[set][j].u = *(const uint4*)(bs + j * 32 * 64 + co); \
                _Pragma("unroll") for (int i = 1; i < TM; ++i) af[set][i].u = *(const uint4*)(as + i * 32 * 64 + co);  \
            }
            PP_READ_FRAGS(0, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks < 3) PP_READ_FRAGS(nxt, ks + 1)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j].v, af[cur][i].v, acc[i][j], 0, 0, 0);
                if (ks < 3) {
                    // inside this k-step: {1 MFMA, 1 ds_read} x (TM+TN), then the remaining MFMAs; nothing crosses k-steps
#pragma unroll
                    for (int n = 0; n < TM + TN; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef PP_READ_FRAGS
        }
    };
    // A group's load phase for slab t: its staged registers -> LDS buffer t & 1, then fetch slab t + 1.
    // EARLY: the fetch is issued here and stays in flight across the group's next compute phase (+A_CH+B_CH uint4 live);
    // otherwise it is issued at the start of the next load phase and waited for there.
    auto load_phase = [&](int t) {
        if (t >= nt) return;
        if (EARLY) {
            store_tile(t & 1);
            if (t + 1 < nt) load_tile(t + 1);
        } else {
            if (t >= 2) load_tile(t);
            store_tile(t & 1);
        }
    };

    // ---- prologue: slab 0 in LDS; slab 1 in registers ----
    load_tile(0);
    store_tile(0);
    if (nt > 1) load_tile(1);
    __syncthreads();

    // debug (MDX_GEMM_TIMING=1): per block and group, s_memtime sums of {compute, barrier after compute, load phase, barrier after it}
    unsigned long long tc = 0, tcb = 0, tl = 0, tlb = 0, t0 = 0, t1 = 0;
    const bool timing = p.timing != nullptr;
#define PP_STAMP(acc_) if (timing) { t1 = __builtin_amdgcn_s_memtime(); acc_ += t1 - t0; t0 = t1; }
    if (timing) t0 = __builtin_amdgcn_s_memtime();
    if (grp == 0) {
        for (int s = 0; s < nt; ++s) {
            compute(s & 1);
            PP_STAMP(tc)
            __syncthreads();
            PP_STAMP(tcb)
            load_phase(s + 1);
            PP_STAMP(tl)
            __syncthreads();
            PP_STAMP(tlb)
        }
    } else {
        for (int s = 0; s < nt; ++s) {
            load_phase(s + 1);
            PP_STAMP(tl)
            __syncthreads();
            PP_STAMP(tlb)
            compute(s & 1);
            PP_STAMP(tc)
            __syncthreads();
            PP_STAMP(tcb)
        }
    }
#undef PP_STAMP
    if (timing && (tid & 255) == 0) {
        unsigned long long* t = p.timing + ((long)blockIdx.x * 2 + grp) * 4;
        t[0] = tc; t[1] = tcb; t[2] = tl; t[3] = tlb;
    }

#undef PP_LOAD_A
#undef PP_LOAD_B
    // ---- epilogue ----
    const bool geglu = p.epi == 1;
    const bool has_t = p.temb != nullptr && !geglu;
    const int b0 = has_t ? m0 / p.rows_per_b : 0;
    {   // fp32 addend rows: bias[n] (+ temb[sel][b0 + slot][n]); columns past N are zero
        const int nslots = has_t ? PP_SLOTS : 1;
        const int sel = (has_t && p.sel) ? *p.sel : 0;
        const int bmax = has_t ? (p.M - 1) / p.rows_per_b : 0;
        for (int idx = tid; idx < nslots * BN; idx += NTH) {
            const int slot = idx / BN, n = idx - slot * BN, col = n0 + n;
            float v = 0.f;
            if (col < p.N) {
                if (p.bias) v = p.bias[col];
                if (has_t && b0 + slot <= bmax) v += p.temb[(long)sel * p.temb_sel_stride + (long)(b0 + slot) * p.temb_b_stride + col];
            }
            addend[idx] = v;
        }
    }
    constexpr int BNO_MAX = BN;
    bf16_t* Cs = (bf16_t*)smem;                    // 128 x (BNo + 8) bf16 staging, aliases the (dead) operand ring
    const int BNo = geglu ? BN / 2 : BN;
    const int CSTR = BNo + 8;
    const int n0o = geglu ? n0 / 2 : n0, Nout = geglu ? p.N / 2 : p.N;
    const bf16_t* Rg = p.R ? (const bf16_t*)p.R : nullptr;
    bf16_t* Cg = (bf16_t*)p.C;
    constexpr int RPW = TM * 32;                   // tile rows per wave row-block
    (void)BNO_MAX;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        __syncthreads();                           // operand ring / previous half's staging is dead; addend visible
        if ((wm * RPW) / 128 == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mt = wm * RPW + i * 32 + frow;          // row inside the tile
                const int ml = mt - h * 128;                      // row inside the half
                int slot = 0;
                if (has_t) {
                    const int m = min(m0 + mt, p.M - 1);
                    slot = min(m / p.rows_per_b - b0, PP_SLOTS - 1);
                }
                const float* ad = addend + slot * BN;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (geglu && (j & 1)) continue;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wn * TN * 32 + j * 32 + 8 * g + 4 * half;
                        const float4 a4 = *(const float4*)(ad + nl);
                        const float bb[4] = {a4.x, a4.y, a4.z, a4.w};
                        float o[4];
                        if (geglu) {
                            const float4 g4 = *(const float4*)(ad + nl + 32);
                            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x = acc[i][j][4 * g + e] + bb[e];
                                const float gt = acc[i][(TN > 1) ? (j | 1) : j][4 * g + e] + gg[e];
                                o[e] = x * gelu_erf_f(gt);
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = acc[i][j][4 * g + e] + bb[e];
                                if (p.epi == 2) x = silu_f(x);
                                o[e] = x;
                            }
                        }
                        const int cl = geglu ? ((nl >> 6) * 32 + (nl & 63)) : nl;
                        uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
                        *(uint2*)(Cs + ml * CSTR + cl) = ov;
                    }
                }
            }
        }
        __syncthreads();
        // row-major walk of the half: consecutive lanes -> consecutive 16-byte (wide) / 8-byte pieces of one output row
        const int mh = m0 + h * 128;
        if (p.wide) {
            const int cpr = BNo >> 3;
            const int total = 128 * cpr;
#pragma unroll 1
            for (int i0 = 0; i0 < total; i0 += 4 * NTH) {
                uint4 rv[4];
                int row[4], c8[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * NTH + tid;
                    row[u] = idx / cpr;
                    c8[u] = (idx - row[u] * cpr) * 8;
                    ok[u] = idx < total && mh + row[u] < p.M && n0o + c8[u] < Nout;
                    rv[u] = make_uint4(0, 0, 0, 0);
                    if (Rg && ok[u]) rv[u] = *(const uint4*)(Rg + (long)(mh + row[u]) * p.ldr + n0o + c8[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    uint4 v = *(const uint4*)(Cs + row[u] * CSTR + c8[u]);
                    if (Rg) { v.x = add2bf(v.x, rv[u].x); v.y = add2bf(v.y, rv[u].y); v.z = add2bf(v.z, rv[u].z); v.w = add2bf(v.w, rv[u].w); }
                    *(uint4*)(Cg + (long)(mh + row[u]) * p.ldc + n0o + c8[u]) = v;
                }
            }
        } else {
            const int cpr = BNo >> 2;
            const int total = 128 * cpr;
#pragma unroll 1
            for (int i0 = 0; i0 < total; i0 += 4 * NTH) {
                uint2 rv[4];
                int row[4], c4[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * NTH + tid;
                    row[u] = idx / cpr;
                    c4[u] = (idx - row[u] * cpr) * 4;
                    ok[u] = idx < total && mh + row[u] < p.M && n0o + c4[u] < Nout;
                    rv[u] = make_uint2(0, 0);
                    if (Rg && ok[u]) rv[u] = *(const uint2*)(Rg + (long)(mh + row[u]) * p.ldr + n0o + c4[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
                    uint2 v = *(const uint2*)(Cs + row[u] * CSTR + c4[u]);
                    if (Rg) { v.x = add2bf(v.x, rv[u].x); v.y = add2bf(v.y, rv[u].y); }
                    *(uint2*)(Cg + (long)(mh + row[u]) * p.ldc + n0o + c4[u]) = v;
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool CONV, bool EARLY>
static int launch_pp(const GCParams& p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t smem = (size_t)2 * (BM + BN) * 64 * sizeof(bf16_t) + (size_t)PP_SLOTS * BN * sizeof(float);
    static_assert(smem <= 163840, "LDS budget");
    static_assert((size_t)128 * (BN + 8) * 2 <= (size_t)2 * (BM + BN) * 64 * 2, "C staging must fit the operand ring");
    auto kern = gemm_pp_kernel<WM, WN, TM, TN, CONV, EARLY>;
    if (int rc = ensure_dyn_smem((const void*)kern, smem, "pp")) return rc;
    GCParams q = p;
    q.mt = (p.M + BM - 1) / BM; q.nt = (p.N + BN - 1) / BN;
    static const int swz = [] { const char* e = getenv("MDX_GEMM_SWZ"); return e ? atoi(e) : 1; }();
    static const int dbg = [] { const char* e = getenv("MDX_PP_DBG"); return e ? atoi(e) : 0; }();
    q.dbg = dbg;
    q.swz = swz && q.nt > 1 && q.mt >= 64;
    const unsigned nblk = q.swz ? (unsigned)((q.mt + 7) / 8 * 8 * q.nt) : (unsigned)(q.mt * q.nt);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), smem, st, q);
    char tag[96];
    snprintf(tag, sizeof tag, "gemm_pp_kernel<%dx%d,%s>", BM, BN, CONV ? "conv" : "gemm");
    return check_launch(tag);
}

// cfg 0: 256 x 256 tile (waves 2 x 4, 128 x 64 each) — GEGLU and N % 256 == 0;  cfg 1: 256 x 320 (waves 4 x 2, 64 x 160 each).
// Returns tile dims for the caller's cost model.
void pp_tile_dims(int cfg, int* bm, int* bn) { *bm = 256; *bn = cfg == 1 ? 320 : 256; }

// Whether the ping-pong kernel can run this problem at all (the caller decides whether it should).
bool pp_supported(const GCParams& p, int cfg) {
    if (p.batch > 1 || p.splitk > 1 || p.c_f32 || (p.N % 4) || (p.K % 64)) return false;   // K tail: gemm_conv.hip
    if (p.epi == 1 && (cfg != 0 || (p.N % 64))) return false;
    if (p.temb && p.epi != 1) {
        int bn_rows = 256 / (p.rows_per_b > 0 ? p.rows_per_b : 1) + 2;       // batch entries a 256-row tile can touch
        if (bn_rows > PP_SLOTS) return false;
    }
    return true;
}

int launch_gemm_pp(const GCParams& p, bool conv, int cfg, hipStream_t st) {
    static const int early = [] { const char* e = getenv("MDX_PP_EARLY"); return e ? atoi(e) : -1; }();
    const bool e0 = early < 0 ? true : (early & 1), e1 = early < 0 ? false : (early & 2);
    if (cfg == 0) {
        if (conv) return e0 ? launch_pp<2, 4, 4, 2, true, true>(p, st) : launch_pp<2, 4, 4, 2, true, false>(p, st);
        return e0 ? launch_pp<2, 4, 4, 2, false, true>(p, st) : launch_pp<2, 4, 4, 2, false, false>(p, st);
    }
    if (conv) return e1 ? launch_pp<4, 2, 2, 5, true, true>(p, st) : launch_pp<4, 2, 2, 5, true, false>(p, st);
    return e1 ? launch_pp<4, 2, 2, 5, false, true>(p, st) : launch_pp<4, 2, 2, 5, false, false>(p, st);
}

}  // namespace mdx
