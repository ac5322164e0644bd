// gemm_params.h — parameter block and fused epilogue shared by the two GEMM/conv main loops
// (gemm_conv.hip: register-staged, gemm_dma.hip: LDS-DMA ring).
#pragma once
#include "common.h"

namespace mdx {

struct GCParams {
    const bf16_t* A; const bf16_t* W; void* C; const void* R;
    const float* bias; const float* temb; const int* sel; float* ws;
    int M, N, K;
    long lda, ldw, ldc, ldr;
    long sA, sW, sC, sR;
    long temb_sel_stride, temb_b_stride;
    int rows_per_b;
    int epi, splitk, kchunk, c_f32, batch;
    long ws_bytes;
    int mt, nt, swz;              // tile counts along M / N; swz: XCD-aware tile order (1-D grid); gemm_xl.hip: 2 = XCD-blocked panels
    int gm, gn;                   // swz == 2: an XCD walks panels of gm M-tiles x gn N-tiles (xl_tile_coords in gemm_xl.hip)
    int nblk;                     // blocks of the tile order incl. the padding blocks of a ragged grid (persistent kernels walk it)
    unsigned long long* timing;   // debug: per-block s_memtime stamps (MDX_GEMM_TIMING=1), else null
    // conv geometry (CONV only); lda doubles as the pixel stride of X
    int Hi, Wi, Cin, Ho, Wo, kh, kw, sh, sw, ph, pw;
    // conv K order: 0 = (ky, kx, ci) as the weights are stored; 1 = channel-block major: slab T covers tap T % (kh kw) of the
    // 64-channel block T / (kh kw), so the kh*kw consecutive slabs of a block re-read the SAME input lines (shifted by a pixel / a row)
    // while they are still in L2 — in storage order the reuse distance is Cin/64 slabs x every resident tile, which evicts them
    // (measured: 28x50 convs ran at half the per-tile rate of 14x25 ones once the activations outgrew the Infinity Cache).
    // Only a permutation of the reduction order; needs Cin % 64 == 0.
    int cimajor;
    int wide;                     // epilogue may use 16-byte global accesses for C / R (alignment + N % 8 checked by the launcher)
    // fused q/k/v projection (MdxGemmDesc.Vt): raw columns >= vt_from are stored transposed to Vt[m / vt_T][n - vt_from][m % vt_T]
    bf16_t* Vt; int vt_from, vt_T; long vt_ld, vt_stride;
    // batch-flattened GEMM (gemm_xl.hip only): N counts the columns of ALL batches (W is one [batch * col_split][K] matrix, A is shared);
    // output column n lands in batch n / col_split at column n % col_split: C + (n / col_split) * sC + m * ldc + n % col_split.
    // 0 = off.  Used for the level-1/2 V^T projections: V^T[b][c][t] = sum_k Wv[c][k] X[b][t][k] as ONE GEMM over all views.
    int col_split;
    // LayerNorm of the A rows fused into the GEMM (MdxGemmDesc.ln_eps; K is the whole row): ln_eps > 0 -> A holds the RAW rows, W / bias carry
    // the affine part, ln_csum[n] = sum_k W[n][k].  gemm_ws.hip computes the row statistics from the slabs it streams and applies
    // C = rstd_m (acc - mean_m csum_n) + bias_n; every other route normalises into ln_scratch first (launch_gemm_conv).
    const float* ln_csum; float ln_eps; bf16_t* ln_scratch;
    // Row statistics (MdxGemmDesc.rowstat_out / ln_stats, ABI 9): rowstat [rowstat_parts][M][2] = (sum, sum of squares) of the stored C rows per column
    // part, emitted by the store phase of gemm_ws.hip's plain kernel (one part per 128-column tile) or by rowstat_kernel behind any other route;
    // ln_stats [ln_stats_parts][M][2]: the same sums of the A rows, from which the fused LayerNorm takes mean / rstd instead of recomputing them.
    float* rowstat; int rowstat_parts;
    const float* ln_stats; int ln_stats_parts;
    // W in MFMA-fragment order for the W-direct kernel (MdxGemmDesc.Wq, ABI 10; layout in gemm_xd.hip / include/mdx.h); null: not given
    const bf16_t* Wq;
    int dbg;                      // debug knobs of gemm_pp.hip (MDX_PP_DBG): 1 skip LDS stores, 2 skip global loads, 4 skip MFMAs
};

// ---- tile order ------------------------------------------------------------------------
// 1-D grid -> (M-tile, N-tile).  Workgroup b lands on XCD b % 8 (observed dispatch rule, used for speed only).
// With swz, XCD x owns M-tiles x, x+8, x+16, ... and walks all N-tiles of one M-tile back to back, so the
// A rows of that M-tile (and, for conv, the overlapping rows of its 9 taps) are fetched into ONE XCD's L2 once
// instead of N-tiles times into different L2s at different times (activations at b>=4 exceed the 4 MiB L2s;
// without this the re-reads come from the Infinity Cache).  Returns false for padding blocks of a ragged grid.
__device__ __forceinline__ bool tile_coords_at(const GCParams& p, int bid, int& tm, int& tn) {
    if (!p.swz) { tm = bid % p.mt; tn = bid / p.mt; return true; }
    const int xcd = bid & 7, local = bid >> 3;
    tn = local % p.nt;
    tm = (local / p.nt) * 8 + xcd;
    return tm < p.mt;
}
__device__ __forceinline__ bool tile_coords(const GCParams& p, int& tm, int& tn) { return tile_coords_at(p, (int)blockIdx.x, tm, tn); }

// ---- shared epilogue ---------------------------------------------------------------
// v[4] are raw accumulators for output row m, raw columns nb..nb+3 (nb % 4 == 0).
// For GEGLU, v = value columns and gte = gate columns (raw column nb+32+j).
__device__ __forceinline__ void epilogue_store(const GCParams& p, long zb, int m, int nb,
                                               const float* v, const float* gte) {
    float o[4];
    int ncol;  // output column of o[0]
    const float* tb = nullptr;
    if (p.temb) {
        int sel = p.sel ? *p.sel : 0;
        tb = p.temb + (long)sel * p.temb_sel_stride + (long)(m / p.rows_per_b) * p.temb_b_stride;
    }
    if (p.epi == 1) {  // GEGLU
        ncol = (nb >> 6) * 32 + (nb & 63);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float h = v[j], g = gte[j];
            if (p.bias) { h += p.bias[nb + j]; g += p.bias[nb + 32 + j]; }
            o[j] = h * gelu_erf_f(g);
        }
    } else {
        ncol = nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (p.bias) x += p.bias[nb + j];
            if (tb) x += tb[nb + j];
            if (p.epi == 2) x = silu_f(x);
            o[j] = x;
        }
    }
    if (p.c_f32) {
        float* c = (float*)p.C + zb * p.sC + (long)m * p.ldc + ncol;
        if (p.R) {
            const float* r = (const float*)p.R + zb * p.sR + (long)m * p.ldr + ncol;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        *(float4*)c = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        bf16_t* c = (bf16_t*)p.C + zb * p.sC + (long)m * p.ldc + ncol;
        if (p.R) {
            const bf16_t* r = (const bf16_t*)p.R + zb * p.sR + (long)m * p.ldr + ncol;
            uint2 rv = *(const uint2*)r;
            o[0] += bf2f((bf16_t)(rv.x & 0xffff)); o[1] += bf2f((bf16_t)(rv.x >> 16));
            o[2] += bf2f((bf16_t)(rv.y & 0xffff)); o[3] += bf2f((bf16_t)(rv.y >> 16));
        }
        uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
        *(uint2*)c = ov;
    }
}


// Phase 2 of the coalesced epilogue: walk the LDS tile row-major and write 256 contiguous bytes per row.
// ALL residual loads of a thread are issued before the first is consumed (they are independent; issuing them
// one per loop iteration serialised 16 memory round trips = 33k cycles per tile, measured).  R may alias C
// (in-place accumulate): a thread reads exactly the addresses it later writes, so load-all-then-store is safe.
// Registers of the epilogue that do not depend on the accumulators: they are loaded BEFORE the main loop so
// their memory latency (2-4k cycles each, three dependent round trips per tile when issued in the epilogue)
// hides under the MFMA work.  bias: per column.  rv: the residual tile in the row-major order phase 2 walks
// (R may alias C for in-place accumulation: a thread reads exactly the addresses it later writes, nobody else
// writes them, so reading early is safe).
template <int BM, int BN, int TN, int NTHR>
struct EpiRegs {
    static constexpr int IT = (BM * (BN / 8) + NTHR - 1) / NTHR;      // 16-byte chunks of the tile per thread
    float4 bv[TN][4];
    uint4 rv[IT];
};

template <int BM, int BN, int TN, int NTHR>
__device__ __forceinline__ void epi_prefetch(const GCParams& p, long zb, int m0, int n0, int wcol0, int lane, int tid,
                                             EpiRegs<BM, BN, TN, NTHR>& er) {
    const int half = lane >> 5;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nb = n0 + wcol0 + j * 32 + 8 * g + 4 * half;
            er.bv[j][g] = (p.bias && nb < p.N) ? *(const float4*)(p.bias + nb) : z4;
        }
    const bool geglu = p.epi == 1;
    const int cpr = (geglu ? BN / 2 : BN) / 8;
    const int n0o = geglu ? n0 / 2 : n0, Nout = geglu ? p.N / 2 : p.N;
    const bf16_t* Rg = (p.R && p.wide) ? (const bf16_t*)p.R + zb * p.sR : nullptr;    // narrow path loads R in the store loop
#pragma unroll
    for (int i = 0; i < EpiRegs<BM, BN, TN, NTHR>::IT; ++i) {
        const int idx = tid + i * NTHR;
        const int row = idx / cpr;
        const int c8 = (idx - row * cpr) * 8;
        const int m = m0 + row, n = n0o + c8;
        er.rv[i] = make_uint4(0, 0, 0, 0);
        if (Rg && idx < BM * cpr && m < p.M && n < Nout) er.rv[i] = *(const uint4*)(Rg + (long)m * p.ldr + n);
    }
}

__device__ __forceinline__ unsigned add2bf(unsigned a, unsigned b) {
    return pack2bf(bf2f((bf16_t)(a & 0xffff)) + bf2f((bf16_t)(b & 0xffff)), bf2f((bf16_t)(a >> 16)) + bf2f((bf16_t)(b >> 16)));
}

// Phase 2 of the coalesced epilogue: walk the LDS tile row-major.  Wide path: 16 bytes per lane (a wave instruction covers
// 1 KiB = full 256-byte row segments; the 8-byte form was store-ISSUE bound: twice the instructions for the same bytes), the
// residual comes from the registers prefetched before the main loop.  Narrow path (N % 8 != 0 or unaligned pitch): 8 bytes
// per lane, residual loaded here in batches of four.
template <int BM, int BNO, int NTHR, int ITMAX>
__device__ __forceinline__ void store_tile_rows(const GCParams& p, long zb, int m0, int n0o, int Nout, int tid, const bf16_t* Cs,
                                                const uint4 (&rv)[ITMAX]) {
    constexpr int CSTR = BNO + 8;
    bf16_t* Cg = (bf16_t*)p.C + zb * p.sC;
    const bool has_r = p.R != nullptr;
    if (p.wide) {
        constexpr int CPR = BNO / 8;                       // 16-byte chunks per tile row
        constexpr int IT = (BM * CPR + NTHR - 1) / NTHR;
        static_assert(IT <= ITMAX, "residual register tile too small");
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / CPR;
            const int c8 = (idx - row * CPR) * 8;
            const int m = m0 + row, n = n0o + c8;
            if (!((idx < BM * CPR) && m < p.M && n < Nout)) continue;
            uint4 v = *(const uint4*)(Cs + row * CSTR + c8);
            if (has_r) { v.x = add2bf(v.x, rv[i].x); v.y = add2bf(v.y, rv[i].y); v.z = add2bf(v.z, rv[i].z); v.w = add2bf(v.w, rv[i].w); }
            *(uint4*)(Cg + (long)m * p.ldc + n) = v;
        }
        return;
    }
    constexpr int CPR = BNO / 4;                           // 8-byte chunks per tile row
    constexpr int IT = (BM * CPR + NTHR - 1) / NTHR;
    const bf16_t* Rg = has_r ? (const bf16_t*)p.R + zb * p.sR : nullptr;
#pragma unroll 1
    for (int i0 = 0; i0 < IT; i0 += 4) {
        uint2 r2[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + (i0 + u) * NTHR;
            const int row = idx / CPR, c4 = (idx - row * CPR) * 4;
            ok[u] = (i0 + u) < IT && idx < BM * CPR && m0 + row < p.M && n0o + c4 < Nout;
            r2[u] = make_uint2(0, 0);
            if (Rg && ok[u]) r2[u] = *(const uint2*)(Rg + (long)(m0 + row) * p.ldr + n0o + c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const int idx = tid + (i0 + u) * NTHR;
            const int row = idx / CPR, c4 = (idx - row * CPR) * 4;
            uint2 v = *(const uint2*)(Cs + row * CSTR + c4);
            if (has_r) { v.x = add2bf(v.x, r2[u].x); v.y = add2bf(v.y, r2[u].y); }
            *(uint2*)(Cg + (long)(m0 + row) * p.ldc + n0o + c4) = v;
        }
    }
}

// ---- coalesced epilogue ----------------------------------------------------------------
// The MFMA accumulator layout gives a lane 4 consecutive n of ONE output row, so storing straight from
// registers writes 8-byte pieces at a row stride: every wave store touches 32 different 128-byte lines and
// fills 1/8 of each (measured: small-K GEMMs were bound by exactly this, not by their main loop).
// Instead the bf16 tile is transposed through LDS (the A/B ring is dead after the main loop): phase 1 applies
// bias / temb / activation in registers and writes the tile to LDS (row stride BNo+8 elements: conflict-free
// ds_write_b64), phase 2 walks the tile row-major, 16 lanes x 16 B = 256 contiguous bytes per output row, adds
// the residual (read with the same coalesced pattern) and stores.
// Uniform control flow: every thread of the block must call it (it contains __syncthreads()).
template <int BM, int BN, int TM, int TN, int NTHR>
__device__ __forceinline__ void epilogue_coalesced(const GCParams& p, long zb, int m0, int n0, int wrow0, int wcol0, int lane,
                                                   int tid, f32x16_t (&acc)[TM][TN], unsigned char* smem,
                                                   const EpiRegs<BM, BN, TN, NTHR>& er) {
    const bool geglu = p.epi == 1;
    const int BNo = geglu ? BN / 2 : BN;              // output columns of this tile
    const int CSTR = BNo + 8;                          // LDS row stride (elements): 16-byte aligned rows, conflict-free ds_write_b64
    bf16_t* Cs = (bf16_t*)smem;
    const int frow = lane & 31, half = lane >> 5;
    // Per-column addends first, ALL loads issued before any use (one memory round trip, not one per element —
    // element-wise `bias[n]` loads behind per-group branches cost ~30k cycles per tile, measured):
    // bias is a function of the column only; temb of (batch row, column).
    float4 tv[TM][TN][4];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_t = p.temb != nullptr && !geglu;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wrow0 + i * 32 + frow;
        const float* tb = nullptr;
        if (has_t) {
            const int sel = p.sel ? *p.sel : 0;
            tb = p.temb + (long)sel * p.temb_sel_stride + (long)((m < p.M ? m : 0) / p.rows_per_b) * p.temb_b_stride;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wcol0 + j * 32 + 8 * g + 4 * half;
                tv[i][j][g] = (has_t && nb < p.N) ? *(const float4*)(tb + nb) : z4;
            }
    }
    __syncthreads();                                   // all waves finished reading the operand slabs
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wrow0 + i * 32 + frow;          // row inside the tile
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (geglu && (j & 1)) continue;            // gate tiles are consumed with their value tile
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wcol0 + j * 32 + 8 * g + 4 * half;      // raw column inside the tile
                const float bb[4] = {er.bv[j][g].x, er.bv[j][g].y, er.bv[j][g].z, er.bv[j][g].w};
                const float tt[4] = {tv[i][j][g].x, tv[i][j][g].y, tv[i][j][g].z, tv[i][j][g].w};
                const float4 bg4 = er.bv[(TN > 1) ? (j | 1) : j][g];
                const float bgt[4] = {bg4.x, bg4.y, bg4.z, bg4.w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * g + e] + bb[e];
                    if (geglu) {
                        float gt = acc[i][(TN > 1) ? (j | 1) : j][4 * g + e] + bgt[e];
                        x = x * gelu_erf_f(gt);
                    } else {
                        x += tt[e];
                        if (p.epi == 2) x = silu_f(x);
                    }
                    o[e] = x;
                }
                const int cl = geglu ? ((nl >> 6) * 32 + (nl & 63)) : nl;   // output column inside the tile
                uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
                *(uint2*)(Cs + ml * CSTR + cl) = ov;
            }
        }
    }
    __syncthreads();
    const int n0o = geglu ? n0 / 2 : n0;
    const int Nout = geglu ? p.N / 2 : p.N;
    if (geglu) store_tile_rows<BM, BN / 2, NTHR>(p, zb, m0, n0o, Nout, tid, Cs, er.rv);
    else store_tile_rows<BM, BN, NTHR>(p, zb, m0, n0o, Nout, tid, Cs, er.rv);
}

}  // namespace mdx
