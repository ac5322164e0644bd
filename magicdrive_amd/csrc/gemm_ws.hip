// gemm_ws.hip — weight-stationary bf16 MFMA GEMM for K = 320 (the level-0 transformer projections and GEGLU) on gfx950.
//
// Why: the 128x128 GEMM of gemm_conv.hip is bound by the L2 -> LDS load path (measured ~32 B/clk per CU with two resident
// workgroups: 160 KB of operand slabs per 128x128x320 tile = 5 k cycles against 2.5 k cycles of MFMA issue) and, at K = 320,
// spends two thirds of a tile in its prologue and epilogue.  Here
//   * each wave keeps ITS 32 weight rows x 320 k in registers (20 MFMA B-fragments = 80 VGPRs) for the whole launch; only the
//     activations stream through LDS (80 KB per tile: half the load-path bytes, half the LDS stores, no W fragment reads);
//   * a workgroup is persistent: it owns one 128-column N-tile and walks M-tiles, so the slab stream never drains — the next
//     tile's first slabs are already in flight / in LDS while this tile's epilogue runs (no per-tile prologue);
//   * workgroups that share an M range (one per N-tile) sit on the same XCD and walk it at the same pace: the activation tile
//     is fetched into that L2 once.
// Tile: 128 x 128, 4 waves side by side along N (each 128 rows x 32 columns: 4 accumulator tiles, A fragments shared), K slabs of
// 64 through a 3-stage LDS ring filled by LDS-DMA, one barrier per slab.  GEGLU: a wave's 32 weight rows are 16 value rows + their
// 16 gate rows of the packed [32 value | 32 gate] layout (packing.py: pack_geglu), so value and gate of one output meet in one lane
// (accumulator registers r and r + 8).
// Epilogue through LDS (separate staging, so the ring keeps streaming): bias / GEGLU in registers, 16-byte row-major stores,
// residual added from 16-byte loads.  Same arithmetic per output element as gemm_conv.hip (k ascending, fp32 accumulation).
#include "common.h"
#include "launch.h"
#include "options.h"
#include "gemm_params.h"

namespace mdx {

constexpr unsigned WS_OOB = 0xFFFFFF00u;       // >= num_records: the buffer load returns 0 and the DMA writes zeros
constexpr unsigned WS_RECORDS = 0x80000000u;   // 2 GiB window over A

template <int N>
__device__ __forceinline__ void ws_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// One LDS-DMA piece by inline asm (same statement as xl_glds in gemm_xl.hip).  The builtin (`__builtin_amdgcn_raw_ptr_buffer_load_lds`)
// is modelled by hipcc as an LDS store of unknown extent: it put `s_waitcnt vmcnt(0)` in front of every slab's fragment reads, i.e.
// every slab waited for the refills issued just before it — the three-stage ring never ran ahead (seen in the .s; round 2).
typedef __attribute__((ext_vector_type(4))) unsigned ws_rsrc_t;
__device__ __forceinline__ void ws_glds(const ws_rsrc_t rs, unsigned lds_addr, unsigned voff) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(rs)
        : "memory");
}
// Workgroup barrier for the epilogue's LDS staging: waits for this wave's LDS traffic only.  `__syncthreads()` also drains the VM
// counter whenever the compiler has loads / stores of its own outstanding (the residual fetch, the C stores) — and with them the
// activation slabs the DMA ring has in flight for the NEXT tile.
__device__ __forceinline__ void ws_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ST-stage LDS ring filled by LDS-DMA (buffer_load_dwordx4 ... lds): ST-1 slabs of the activation stream are in flight while
// one is multiplied — a K = 320 slab is only 16 MFMAs per wave, far shorter than a memory round trip, and with register staging
// (one slab ahead) every slab waited on its loads (measured: 25 k cycles per tile for 2.6 k cycles of MFMA work).
// LDS rows are 128 B, unpadded (the DMA writes 1 KiB linearly: 8 rows per wave instruction); the 16-byte chunk index is
// XOR-swizzled by (row >> 1) & 7 on the SOURCE side and again on the fragment reads: conflict-free ds_read_b128.
// VT: every tile of the launch is stored transposed to p.Vt (the V part of a fused q/k/v projection is its own launch over the
// V rows of the weight: one kernel carrying both operand orders spilled 89 VGPRs).
// LN: LayerNorm of the A rows fused in (GCParams.ln_eps): K = 320 IS the whole row and every workgroup streams all of it through
// LDS, so wave w sums x and x^2 of rows 32 w .. 32 w + 31 from the slabs as they land (one extra 16-byte LDS read per k-step, 24 VALU
// operations beside 4 MFMAs), publishes (rstd, -mean rstd) per row in LDS at the end of the tile, and the epilogue turns the raw
// product into rstd_m acc + (-mean_m rstd_m) csum_n + bias_n.  gamma / beta are in W / bias (packed by the host).  The tokens are
// read ONCE (by this GEMM) instead of read + written by layernorm_kernel and read again here.  One-pass variance in fp32
// (E[x^2] - mean^2 over 320 values of a 16-bit tensor).  Every N-tile's workgroup repeats the row sums, and they are NOT free: measured
// (tools/lnone.py, 768 views, side builds -DMDX_WS_LN_ABLATE) the fused q/k/v launch pair is 150 us slower than the plain one (1200 us) —
// 100 us the 96 VALU operations per slab (each costs its 4-cycle issue slot whether it sits in front of the slab's MFMAs or between
// them), 15 us the extra fragment read, 35 us the epilogue — against the 240-400 us LayerNorm pass it replaces; to_q: +105 us.  For
// the GEGLU (20 N-tiles, a VALU-heavy epilogue already) the same scheme cost +485 us per launch, more than the pass: not built
// (the engine keeps layernorm_kernel in front of ff.net.0; profiles/README.md round 3).
// LN = 2 (round 6, MdxGemmDesc.ln_stats): mean / rstd come from the (sum, sum of squares) that the PRODUCER of A wrote from its store phase
// (RS below), summed over its column parts — no sums in the slab loop, and with that the GEGLU epilogue can carry norm3 as well.  The partial sums
// of a tile's 128 rows are fetched by hand-counted loads right behind the barrier of the tile's first slab (older than that slab's refill pieces, so
// the next slab's `vmcnt` wait covers them) and published in LDS before the barrier of the third slab.
// RS (MdxGemmDesc.rowstat_out): the wide store phase of the plain kernel also writes, per stored row segment of its 128-column tile, the sum and the
// sum of squares of the values it stores (16 lanes share a row: a DPP row rotation tree), part = the N-tile — what the LayerNorm behind it needs.
template <bool GEGLU, int ST, bool VT, int LN, bool RS>
__global__ __launch_bounds__(256, 2) void gemm_ws_kernel(GCParams p) {
    static_assert(!(GEGLU && LN == 1), "in-kernel LayerNorm statistics: plain and V^T epilogues only");
    static_assert(!(RS && (GEGLU || VT)), "row statistics: plain store phase only");

    constexpr int BM = 128, BN = 128, BK = 64, KS = 20, NSLAB = 5;
    constexpr int BNO = GEGLU ? BN / 2 : BN;
    constexpr int CSTR = BNO + 8;
    constexpr int ROWS_PASS = GEGLU ? 128 : 64;                  // epilogue staging holds this many rows (keeps LDS <= 80 KB: 2 WG/CU)
    constexpr int CSTR_T = ROWS_PASS + 8;                        // transposed staging (V tiles): [128 channels][ROWS_PASS tokens]
    constexpr int STAGE = BM * BK * 2;                           // 16 KiB
    constexpr int PPW = 4;                                       // 1-KiB DMA pieces per wave per slab
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Cs = (bf16_t*)(smem + ST * STAGE);                   // [ROWS_PASS][CSTR]
    float2* Ls = (float2*)(smem + ST * STAGE + 128 * (64 + 8) * 2);   // LN: (rstd, -mean rstd) of the tile's 128 rows (after the staging)
    float* Lc = (float*)(Ls + 128);                                   // LN: column sums of W, this tile's 128 columns
    float* Lb = Lc + 128;                                             // LN: bias of the same columns (LDS instead of 2 x 16 VGPRs)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, half = lane >> 5;
    // block -> (N-tile, M-walker): blocks b, b+8, ... land on one XCD; all N-tiles of a walker share it
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int tn = kq % p.nt;
    const int walker = (kq / p.nt) * 8 + xcd;
    const int nwalk = p.swz;                                     // walkers per N-tile (multiple of 8), set by the launcher
    const int n0 = tn * BN;
    const int mt = p.mt;
    const int ntiles = walker < mt ? (mt - walker + nwalk - 1) / nwalk : 0;
    if (ntiles == 0) return;

    // ---- this wave's weight rows -> registers ----
    int wrow, ocol;                                              // W row of this lane's n position; first output column of the wave
    if (GEGLU) {
        const int grp = wave >> 1, hv = wave & 1;
        wrow = n0 + grp * 64 + (frow < 16 ? 16 * hv + frow : 32 + 16 * hv + (frow - 16));
        ocol = grp * 32 + 16 * hv;                               // within the tile's 64 output columns
    } else {
        wrow = n0 + 32 * wave + frow;
        ocol = 32 * wave;
    }
    Frag8 wf[KS];
    {
        // all 20 loads issued back to back from a clamped row, masked afterwards: written as `ok ? load : 0` each load became its own
        // branch + `s_waitcnt vmcnt(0)` (24 serial memory round trips at the start of every workgroup, seen in the .s; round 3)
        const bf16_t* wp = p.W + (long)min(wrow, p.N - 1) * p.ldw + half * 8;
        const unsigned keep = wrow < p.N ? 0xffffffffu : 0u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks].u = *(const uint4*)(wp + ks * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { wf[ks].u.x &= keep; wf[ks].u.y &= keep; wf[ks].u.z &= keep; wf[ks].u.w &= keep; }
    }
    // bias of this lane's output positions (fixed for the whole launch)
    float4 bA[4];                                                // GEGLU: {value g0, value g1, gate g0, gate g1}; else g = 0..3
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (LN) { bA[g] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }          // LN: bias sits in LDS next to the column sums (Lb)
        int col;
        if (GEGLU) col = n0 + (wave >> 1) * 64 + 16 * (wave & 1) + 8 * (g & 1) + 4 * half + 32 * (g >> 1);
        else col = n0 + ocol + 8 * g + 4 * half;
        const float4 bv = p.bias ? *(const float4*)(p.bias + min(col, p.N - 4)) : make_float4(0.f, 0.f, 0.f, 0.f);   // unconditional, clamped (p.bias is wave-uniform)
        bA[g] = col < p.N ? bv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // V tiles: lane = channel, so bias (and the LN column sum) are one value per lane
    float bch = 0.f, csch = 0.f;
    if (VT) {
        bch = p.bias ? p.bias[min(wrow, p.N - 1)] : 0.f;
        if (LN) csch = p.ln_csum[min(wrow, p.N - 1)];
        if (wrow >= p.N) { bch = 0.f; csch = 0.f; }
    }
    if (LN && !VT) {                                             // column sums of W for the tile's 128 (packed) columns: LDS (read per tile; 16 VGPRs otherwise)
        if (tid < 128) {
            Lc[tid] = n0 + tid < p.N ? p.ln_csum[n0 + tid] : 0.f;
            Lb[tid] = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
        }
    }

    // ---- activation stream: DMA cursor (runs ST-1 slabs ahead of the multiply) ----
    ws_rsrc_t rsA;
    {
        const unsigned long long a_ = (unsigned long long)p.A;
        rsA.x = __builtin_amdgcn_readfirstlane((unsigned)a_);
        rsA.y = __builtin_amdgcn_readfirstlane((unsigned)(a_ >> 32) & 0xffffu);
        rsA.z = WS_RECORDS; rsA.w = 0x00020000u;
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    // The compiler cannot see the hand-counted DMA traffic on the VM counter: settle its bookkeeping for the registers loaded above
    // (weights, bias) HERE, or it guards their first use inside the tile loop with `s_waitcnt vmcnt(0)` on every iteration.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ks].u.x), "+v"(wf[ks].u.y), "+v"(wf[ks].u.z), "+v"(wf[ks].u.w));
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bA[g].x), "+v"(bA[g].y), "+v"(bA[g].z), "+v"(bA[g].w));
    if (VT) asm volatile("" : "+v"(bch), "+v"(csch));
    const int crow = lane >> 3, cphys = lane & 7;
    unsigned r_row[PPW], r_coff[PPW];                            // row inside the tile; byte offset of the (swizzled) source chunk
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = (wave * PPW + j) * 8 + crow;
        r_row[j] = row;
        r_coff[j] = (unsigned)(((cphys ^ ((row >> 1) & 7)) * 8) * 2);
    }
    const unsigned ldab = (unsigned)(p.lda * 2);
    int f_tile = walker, f_slab = 0, f_cnt = 0;
    const int total = ntiles * NSLAB;
    // One 1-KiB piece of the cursor's slab -> ring stage f_cnt % ST.  (All four pieces of a slab are issued right after the
    // barrier; spreading them between the k-steps' MFMAs measured 30-50 % slower.)
#define WS_ISSUE_PIECE(j)                                                                                                       \
    {                                                                                                                           \
        const unsigned m = (unsigned)(f_tile * BM) + r_row[j];                                                                  \
        const unsigned off = (f_cnt < total && m < (unsigned)p.M) ? m * ldab + (unsigned)(f_slab * BK * 2) + r_coff[j] : WS_OOB; \
        ws_glds(rsA, lds0 + (unsigned)((f_cnt % ST) * STAGE + (wave * PPW + (j)) * 1024), off);                                 \
    }
#define WS_ADVANCE()                                                                                                            \
    {                                                                                                                           \
        ++f_cnt;                                                                                                                \
        if (++f_slab == NSLAB) { f_slab = 0; f_tile += nwalk; }                                                                 \
    }

    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#pragma unroll
    for (int s = 0; s < ST - 1; ++s) {                           // past-the-end slabs are zero fills: the wait counts stay uniform
#pragma unroll
        for (int j = 0; j < PPW; ++j) WS_ISSUE_PIECE(j)
        WS_ADVANCE()
    }

    const int x0 = half ^ ((frow >> 1) & 7);                     // swizzled chunk of k-step 0; k-step ks flips bits 1-2
    const unsigned char* const a_rd = smem + frow * 128;
    const bf16_t* Rg = p.R ? (const bf16_t*)p.R : nullptr;
    bf16_t* Cg = (bf16_t*)p.C;
    constexpr bool vtile = VT;
    // Ablations (WS_DBG: 1 GEGLU without GELU, 2 main loop only, 4 no MFMAs) exist only in -DMDX_WS_ABLATE builds: as runtime tests they
    // sat INSIDE the slab loop — one scalar branch per k-step, which cut the loop into 20 basic blocks of {4 fragment reads, wait, 4 MFMAs}
    // per tile and kept the compiler from reading a slab's fragments ahead of its MFMAs (seen in the .s; round 3).
#ifdef MDX_WS_ABLATE
    const bool do_mma = !(p.dbg & 4), no_gelu = p.dbg & 1, no_epi = p.dbg & 2;
#else
    constexpr bool do_mma = true, no_gelu = false, no_epi = false;
#endif
    const int n0o = GEGLU ? n0 / 2 : n0, Nout = GEGLU ? p.N / 2 : p.N;
    int q = 0;                                                   // slab counter (ring stage = q % ST)
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;            // LN: sums of x / x^2 of row 32 wave + frow over this lane's half of the k chunks
    const float ln_eps = p.ln_eps;
    constexpr int LNS_MAXP = 4;                                  // LN == 2: column parts of the producer's row statistics (3 for N = 320 out of this kernel)
    unsigned long long lns[LNS_MAXP] = {0, 0, 0, 0};             // raw (sum, sum of squares) pairs of row m0 + (tid & 127), in flight over one slab
    for (int t = 0; t < ntiles; ++t) {
        const int m0 = (walker + t * nwalk) * BM;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s, ++q) {
            // slab q has landed once at most the ST-2 younger slabs are outstanding (an epilogue's younger loads / stores only
            // make this wait conservative); the barrier also frees stage (q-1) % ST for the refill below
            // Nothing of the previous slab may sink below this barrier: its last fragment reads must have RETURNED (their MFMAs carry the
            // lgkmcnt waits) before any wave refills that stage.
            __builtin_amdgcn_sched_barrier(0);
            ws_wait_vmcnt<(ST - 2) * PPW>();
            if constexpr (LN == 2) {
                if (s == 1) {                                    // the partial sums fetched behind slab 0's barrier have landed (in-order return)
#pragma unroll
                    for (int k = 0; k < LNS_MAXP; ++k) asm volatile("" : "+v"(lns[k]));
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int k = 0; k < LNS_MAXP; ++k) {
                        const float keep = k < p.ln_stats_parts ? 1.f : 0.f;
                        t1 = __builtin_fmaf(__uint_as_float((unsigned)(lns[k] & 0xffffffffull)), keep, t1);
                        t2 = __builtin_fmaf(__uint_as_float((unsigned)(lns[k] >> 32)), keep, t2);
                    }
                    const float mean = t1 * (1.0f / 320.0f);
                    const float var = fmaxf(__builtin_fmaf(-mean, mean, t2 * (1.0f / 320.0f)), 0.f);
                    const float rs = rsqrtf(var + ln_eps);
                    if (tid < 128) Ls[tid] = make_float2(rs, -mean * rs);      // read after >= 3 more barriers (the epilogue)
                }
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LN == 2) {
                if (s == 0) {                                    // every wave is past the previous tile's epilogue (it read that tile's Ls)
                    const long row = min(m0 + (tid & 127), p.M - 1);
#pragma unroll
                    for (int k = 0; k < LNS_MAXP; ++k) {
                        const float* sp = p.ln_stats + ((long)min(k, p.ln_stats_parts - 1) * p.M + row) * 2;
                        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(lns[k]) : "v"(sp) : "memory");
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < PPW; ++j) WS_ISSUE_PIECE(j)      // refill of the stage freed by this iteration's barrier
            WS_ADVANCE()
            const unsigned char* as = a_rd + (q % ST) * STAGE;
            Frag8 af[2][4];
            // LN: this wave also sums x and x^2 of rows 32 wave .. 32 wave + 31 (the chunks its m-tile `wave` fragments hold — a second
            // read: selecting af[.][wave] by a wave-uniform index would need a branch per k-step).  The 24 VALU operations of a k-step sit
            // between its 4 MFMAs, 6 behind each.
            Frag8 sf[2];
            const unsigned char* sr = as + wave * (32 * 128);
#if defined(MDX_WS_LN_ABLATE) && (MDX_WS_LN_ABLATE & 2)              // ... and the read too (one per tile keeps the code shape)
            if constexpr (LN == 1) { if (s == 0) sf[0].u = *(const uint4*)(sr + (x0 << 4)); }
#else
            if constexpr (LN == 1) sf[0].u = *(const uint4*)(sr + (x0 << 4));
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) af[0][i].u = *(const uint4*)(as + i * 32 * 128 + (x0 << 4));
            if constexpr (LN == 1) __builtin_amdgcn_sched_barrier(0); else __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
                    const int co = (x0 ^ ((ks + 1) << 1)) << 4;
#if defined(MDX_WS_LN_ABLATE) && (MDX_WS_LN_ABLATE & 2)
                    if constexpr (LN == 1) sf[(ks + 1) & 1].u = sf[ks & 1].u;
#else
                    if constexpr (LN == 1) sf[(ks + 1) & 1].u = *(const uint4*)(sr + co);
#endif
#pragma unroll
                    for (int i = 0; i < 4; ++i) af[(ks + 1) & 1][i].u = *(const uint4*)(as + i * 32 * 128 + co);
                }
                if constexpr (LN == 1) {
                    // {1 MFMA, the sums of 2 of the fragment's 8 values (6 VALU)} x 4, each group fenced: the VALU work issues in the shadow of
                    // the MFMA in front of it (sched_group_barrier pipelines of this shape were only partly honoured: half of the VALU ended up
                    // in one block behind the slab's last MFMA)
                    if (ks + 1 < 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (do_mma) {
                            if (vtile) acc[i] = MDX_MFMA_32x32x16(af[ks & 1][i].v, wf[s * 4 + ks].v, acc[i]);
                            else acc[i] = MDX_MFMA_32x32x16(wf[s * 4 + ks].v, af[ks & 1][i].v, acc[i]);
                        }
#if defined(MDX_WS_LN_ABLATE) && (MDX_WS_LN_ABLATE & 1)      // side builds (tools/lnone.py): drop the sums, keep the extra fragment read
                        if (i == 0) s1a += bf2f(sf[ks & 1].h[0]);
#else
                        const float xa = bf2f(sf[ks & 1].h[2 * i]), xb = bf2f(sf[ks & 1].h[2 * i + 1]);
                        s1a += xa; s1b += xb;
                        s2a = __builtin_fmaf(xa, xa, s2a); s2b = __builtin_fmaf(xb, xb, s2b);
#endif
                        // (the empty asm ties the sums to this point: instruction selection orders only side-effecting nodes against the fence)
                        asm volatile("" : "+v"(s1a), "+v"(s1b), "+v"(s2a), "+v"(s2b));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    continue;
                }
                if (do_mma) {
                    if (vtile) {          // operands swapped: the accumulator comes out transposed (lane = channel, registers = tokens)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i] = MDX_MFMA_32x32x16(af[ks & 1][i].v, wf[s * 4 + ks].v, acc[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i] = MDX_MFMA_32x32x16(wf[s * 4 + ks].v, af[ks & 1][i].v, acc[i]);
                    }
                }
                // pin the order {fragment reads of k-step ks + 1} {4 MFMAs of k-step ks}: the machine scheduler otherwise sinks the
                // reads back next to their use to save registers
                if (ks + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            // LN: the sums must be formed in THIS slab: left alone, LLVM sinks the whole add chain to the end of the tile and keeps all 20
            // fragments of the tile alive until then (80 VGPRs -> 70 spilled registers, reloaded behind s_waitcnt vmcnt(0) inside the ring)
            if constexpr (LN == 1) asm volatile("" : "+v"(s1a), "+v"(s1b), "+v"(s2a), "+v"(s2b));
        }
        // ---- epilogue of tile t: staging is separate from the ring, which keeps streaming ----
        if (no_epi) continue;                                    // ablation: main loop only
        if constexpr (LN == 1) {
            float s1 = s1a + s1b, s2 = s2a + s2b;
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);          // the other half of the k chunks
            const float mean = s1 * (1.0f / 320.0f);
            const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * (1.0f / 320.0f)), 0.f);
            const float rs = rsqrtf(var + ln_eps);
            if (half == 0) Ls[wave * 32 + frow] = make_float2(rs, -mean * rs);
            s1a = s1b = s2a = s2b = 0.f;
            ws_lds_barrier();                                    // (the next tile's statistics are written >= NSLAB barriers from here)
        }
#pragma unroll
        for (int pass = 0; pass < BM / ROWS_PASS; ++pass) {
            if (pass > 0) ws_lds_barrier();                       // previous pass's row walk is done with Cs
            if constexpr (VT) {
                // V tile: lane = channel 32 wave + frow, accumulator register r = token (r&3) + 8 (r>>2) + 4 half of m-tile i
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i * 32 / ROWS_PASS != pass) continue;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int tl = i * 32 + 8 * g + 4 * half - pass * ROWS_PASS;       // token inside the pass
                        float o[4];
                        if constexpr (LN != 0) {
                            const float4 sa = *(const float4*)(Ls + i * 32 + 8 * g + 4 * half), sb = *(const float4*)(Ls + i * 32 + 8 * g + 4 * half + 2);
                            o[0] = __builtin_fmaf(sa.x, acc[i][4 * g], __builtin_fmaf(sa.y, csch, bch));
                            o[1] = __builtin_fmaf(sa.z, acc[i][4 * g + 1], __builtin_fmaf(sa.w, csch, bch));
                            o[2] = __builtin_fmaf(sb.x, acc[i][4 * g + 2], __builtin_fmaf(sb.y, csch, bch));
                            o[3] = __builtin_fmaf(sb.z, acc[i][4 * g + 3], __builtin_fmaf(sb.w, csch, bch));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[i][4 * g + e] + bch;
                        }
                        uint2 ov;
                        ov.x = pack2bf(o[0], o[1]);
                        ov.y = pack2bf(o[2], o[3]);
                        *(uint2*)(Cs + (32 * wave + frow) * CSTR_T + tl) = ov;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                }
                ws_lds_barrier();
                const int mp = m0 + pass * ROWS_PASS;
                constexpr int CPRT = ROWS_PASS / 8;                                    // 16-byte token chunks per channel row
#pragma unroll
                for (int u = 0; u < 128 * CPRT / 256; ++u) {
                    const int idx = tid + u * 256;
                    const int ch = idx / CPRT, c8 = (idx - ch * CPRT) * 8;
                    const int m = mp + c8, cv = n0 + ch;
                    if (m >= p.M || cv >= p.N) continue;
                    const int b = m / p.vt_T, t = m - b * p.vt_T;
                    *(uint4*)(p.Vt + (long)b * p.vt_stride + (long)cv * p.vt_ld + t) = *(const uint4*)(Cs + ch * CSTR_T + c8);
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i * 32 / ROWS_PASS != pass) continue;
                const int ml = i * 32 + frow - pass * ROWS_PASS;
                if (GEGLU) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        float bvv[4] = {bA[g].x, bA[g].y, bA[g].z, bA[g].w};
                        float bgg[4] = {bA[2 + g].x, bA[2 + g].y, bA[2 + g].z, bA[2 + g].w};
                        float o[4];
                        float cvv[4] = {0.f, 0.f, 0.f, 0.f}, cgg[4] = {0.f, 0.f, 0.f, 0.f};
                        float2 st = make_float2(1.f, 0.f);
                        if constexpr (LN != 0) {                     // norm3 folded in: rstd_m acc + (-mean_m rstd_m) csum_n + bias_n on value AND gate columns
                            st = Ls[i * 32 + frow];
                            const int cb = (wave >> 1) * 64 + 16 * (wave & 1) + 8 * g + 4 * half;      // raw (packed) column of the value; its gate sits 32 further
                            const float4 c_v = *(const float4*)(Lc + cb), c_g = *(const float4*)(Lc + cb + 32);
                            const float4 b_v = *(const float4*)(Lb + cb), b_g = *(const float4*)(Lb + cb + 32);
                            cvv[0] = c_v.x; cvv[1] = c_v.y; cvv[2] = c_v.z; cvv[3] = c_v.w;
                            cgg[0] = c_g.x; cgg[1] = c_g.y; cgg[2] = c_g.z; cgg[3] = c_g.w;
                            bvv[0] = b_v.x; bvv[1] = b_v.y; bvv[2] = b_v.z; bvv[3] = b_v.w;
                            bgg[0] = b_g.x; bgg[1] = b_g.y; bgg[2] = b_g.z; bgg[3] = b_g.w;
                        }
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {             // pairs: the GELU runs on packed fp32 (common.h: gelu_erf_f2)
                            float x0, x1;
                            f32x2_t gt;
                            if constexpr (LN != 0) {
                                x0 = __builtin_fmaf(st.x, acc[i][4 * g + e], __builtin_fmaf(st.y, cvv[e], bvv[e]));
                                x1 = __builtin_fmaf(st.x, acc[i][4 * g + e + 1], __builtin_fmaf(st.y, cvv[e + 1], bvv[e + 1]));
                                gt.x = __builtin_fmaf(st.x, acc[i][8 + 4 * g + e], __builtin_fmaf(st.y, cgg[e], bgg[e]));
                                gt.y = __builtin_fmaf(st.x, acc[i][8 + 4 * g + e + 1], __builtin_fmaf(st.y, cgg[e + 1], bgg[e + 1]));
                            } else {
                                x0 = acc[i][4 * g + e] + bvv[e]; x1 = acc[i][4 * g + e + 1] + bvv[e + 1];
                                gt.x = acc[i][8 + 4 * g + e] + bgg[e]; gt.y = acc[i][8 + 4 * g + e + 1] + bgg[e + 1];
                            }
                            const f32x2_t ge = no_gelu ? gt : gelu_erf_f2(gt);
                            o[e] = x0 * ge.x; o[e + 1] = x1 * ge.y;
                        }
                        uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
                        *(uint2*)(Cs + ml * CSTR + ocol + 8 * g + 4 * half) = ov;
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 ov;
                        if constexpr (LN != 0) {
                            const float2 st = Ls[i * 32 + frow];                                     // (rstd, -mean rstd) of this lane's row
                            const float4 cs = *(const float4*)(Lc + ocol + 8 * g + 4 * half), bb = *(const float4*)(Lb + ocol + 8 * g + 4 * half);
                            ov.x = pack2bf(__builtin_fmaf(st.x, acc[i][4 * g], __builtin_fmaf(st.y, cs.x, bb.x)),
                                           __builtin_fmaf(st.x, acc[i][4 * g + 1], __builtin_fmaf(st.y, cs.y, bb.y)));
                            ov.y = pack2bf(__builtin_fmaf(st.x, acc[i][4 * g + 2], __builtin_fmaf(st.y, cs.z, bb.z)),
                                           __builtin_fmaf(st.x, acc[i][4 * g + 3], __builtin_fmaf(st.y, cs.w, bb.w)));
                        } else {
                        ov.x = pack2bf(acc[i][4 * g] + bA[g].x, acc[i][4 * g + 1] + bA[g].y);
                        ov.y = pack2bf(acc[i][4 * g + 2] + bA[g].z, acc[i][4 * g + 3] + bA[g].w);
                        }
                        *(uint2*)(Cs + ml * CSTR + ocol + 8 * g + 4 * half) = ov;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            }
            ws_lds_barrier();
            const int mp = m0 + pass * ROWS_PASS;
            if (p.wide) {
                constexpr int CPR = BNO / 8, IT = ROWS_PASS * CPR / 256;     // 16-byte chunks per row; per thread: 4
                static_assert(IT == 4, "row walk batch");
                uint4 rv[IT];
                bool ok[IT];
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    const int idx = tid + u * 256;
                    const int row = idx / CPR, c8 = (idx - row * CPR) * 8;
                    ok[u] = mp + row < p.M && n0o + c8 < Nout;
                }
                if (Rg) {                                        // one batch of unconditional loads from clamped addresses (a guarded load
#pragma unroll                                                   // becomes its own branch + full wait: four serial round trips per pass)
                    for (int u = 0; u < IT; ++u) {
                        const int idx = tid + u * 256;
                        const int row = idx / CPR, c8 = (idx - row * CPR) * 8;
                        const int rr = min(mp + row, p.M - 1), cc = min(n0o + c8, Nout - 8);
                        rv[u] = *(const uint4*)(Rg + (long)rr * p.ldr + cc);
                    }
                }
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    if (!RS && !ok[u]) continue;
                    const int idx = tid + u * 256;
                    const int row = idx / CPR, c8 = (idx - row * CPR) * 8;
                    uint4 v = *(const uint4*)(Cs + row * CSTR + c8);
                    if (Rg) { v.x = add2bf(v.x, rv[u].x); v.y = add2bf(v.y, rv[u].y); v.z = add2bf(v.z, rv[u].z); v.w = add2bf(v.w, rv[u].w); }
                    if (ok[u]) *(uint4*)(Cg + (long)(mp + row) * p.ldc + n0o + c8) = v;
                    if constexpr (RS) {
                        // (sum, sum of squares) of the 8 values this lane stores; the 16 lanes of a row segment (CPR = 16: lanes 16 k .. 16 k + 15 of the
                        // wave) are combined by a DPP row-rotation tree — every lane takes part (no early exit above), fixed order: deterministic
                        static_assert(CPR == 16, "row statistics: one 16-lane DPP row per stored row segment");
                        // packed dot products (dot2_16: two 16-bit products + fp32 accumulate per instruction): sum = x . (1, 1), squares = x . x —
                        // 8 instructions per 8 values (unpack + add + fma: 32; measured +48 us on a 350 us launch at 576 views, r6b A/B log)
                        const unsigned km = ok[u] ? 0xffffffffu : 0u;
                        const unsigned one2 = MDX_ONE16 | (MDX_ONE16 << 16);
                        const unsigned w0 = v.x & km, w1 = v.y & km, w2 = v.z & km, w3 = v.w & km;
                        float q1 = dot2_16(w0, one2, 0.f), q2 = dot2_16(w0, w0, 0.f);
                        q1 = dot2_16(w1, one2, q1); q2 = dot2_16(w1, w1, q2);
                        q1 = dot2_16(w2, one2, q1); q2 = dot2_16(w2, w2, q2);
                        q1 = dot2_16(w3, one2, q1); q2 = dot2_16(w3, w3, q2);
                        // row_ror:n (dpp_ctrl 0x120 + n): lane l of a 16-lane row reads lane (l + n) % 16 of its row
#define WS_ROR_ADD(x, n) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, true))
                        WS_ROR_ADD(q1, 8); WS_ROR_ADD(q2, 8); WS_ROR_ADD(q1, 4); WS_ROR_ADD(q2, 4);
                        WS_ROR_ADD(q1, 2); WS_ROR_ADD(q2, 2); WS_ROR_ADD(q1, 1); WS_ROR_ADD(q2, 1);
#undef WS_ROR_ADD
                        if ((tid & 15) == 0 && mp + row < p.M) *(float2*)(p.rowstat + ((long)tn * p.M + mp + row) * 2) = make_float2(q1, q2);
                    }
                }
            } else {
                constexpr int CPR = BNO / 4, IT = ROWS_PASS * CPR / 256;     // 8
#pragma unroll 1
                for (int i0 = 0; i0 < IT; i0 += 4) {
                    uint2 rv[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = tid + (i0 + u) * 256;
                        const int row = idx / CPR, c4 = (idx - row * CPR) * 4;
                        ok[u] = mp + row < p.M && n0o + c4 < Nout;
                        rv[u] = make_uint2(0, 0);
                        if (Rg && ok[u]) rv[u] = *(const uint2*)(Rg + (long)(mp + row) * p.ldr + n0o + c4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!ok[u]) continue;
                        const int idx = tid + (i0 + u) * 256;
                        const int row = idx / CPR, c4 = (idx - row * CPR) * 4;
                        uint2 v = *(const uint2*)(Cs + row * CSTR + c4);
                        if (Rg) { v.x = add2bf(v.x, rv[u].x); v.y = add2bf(v.y, rv[u].y); }
                        *(uint2*)(Cg + (long)(mp + row) * p.ldc + n0o + c4) = v;
                    }
                }
            }
        }
        // the next tile's first Cs write is >= NSLAB barriers away: no extra barrier needed here
    }
#undef WS_ISSUE_PIECE
#undef WS_ADVANCE
    ws_wait_vmcnt<0>();                                          // drain the zero-fill tail before the LDS is released
}

// Whether the weight-stationary kernel can run this problem (the caller decides whether it should).
bool ws_supported(const GCParams& p) {
    return p.K == 320 && p.batch <= 1 && p.splitk <= 1 && !p.c_f32 && !p.temb && (p.epi == 0 || p.epi == 1) && (p.N % 4) == 0 &&
           (p.epi != 1 || (p.N % 64) == 0);
}

template <bool GEGLU, bool VT, int LN, bool RS>
static int launch_ws_one(const GCParams& p, hipStream_t st) {
    constexpr int ST = 3;
    const size_t smem = (size_t)ST * 128 * 64 * 2 + (size_t)128 * (64 + 8) * 2 +   // ring + staging (also holds 64 x 136 and the transposed 128 x 72)
                        (LN ? 128 * sizeof(float2) + 256 * sizeof(float) : 0);      // + the tile's row statistics, W's column sums, bias
    auto kern = gemm_ws_kernel<GEGLU, ST, VT, LN, RS>;
    if (int rc = ensure_dyn_smem((const void*)kern, smem, "ws")) return rc;
    GCParams q = p;
    q.mt = (p.M + 127) / 128; q.nt = (p.N + 127) / 128;
    // walkers per N-tile: fill the 512 workgroup slots (2 per CU), multiple of 8 (one XCD per walker), at most one per M-tile
    const int slots = (int)opt(OPT_WS_SLOTS);
    int nwalk = slots / q.nt / 8 * 8;
    if (nwalk < 8) nwalk = 8;
    const int mt8 = (q.mt + 7) / 8 * 8;
    if (nwalk > mt8) nwalk = mt8;
    q.swz = nwalk;
    const int dbg = (int)opt(OPT_WS_DBG);
    q.dbg = dbg;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nwalk * q.nt)), dim3(256), smem, st, q);
    return check_launch(GEGLU ? (LN ? "gemm_ws_kernel<geglu,lns>" : "gemm_ws_kernel<geglu>")
                              : LN == 2 ? (VT ? "gemm_ws_kernel<vT,lns>" : "gemm_ws_kernel<plain,lns>")
                              : LN == 1 ? (VT ? "gemm_ws_kernel<vT,ln>" : "gemm_ws_kernel<plain,ln>")
                                        : (VT ? "gemm_ws_kernel<vT>" : RS ? "gemm_ws_kernel<plain,rs>" : "gemm_ws_kernel<plain>"));
}

// Whether launch_gemm_ws normalises the A rows itself when p.ln_eps > 0 (else the caller must have done it: launch_gemm_conv): plain / V^T epilogues
// with statistics from the streamed rows or from the producer; GEGLU only with the producer's statistics (MdxGemmDesc.ln_stats).
bool ws_fuses_layernorm(const GCParams& p) {
    if (!p.ln_csum || (p.ln_stats && p.ln_stats_parts > 4)) return false;
    return p.epi == 0 || (p.epi == 1 && p.ln_stats != nullptr);
}
// Whether the plain kernel's store phase can emit the row statistics of C (MdxGemmDesc.rowstat_out): 16-byte row walk, one part per 128-column tile.
bool ws_emits_rowstat(const GCParams& p) {
    return p.rowstat && p.epi == 0 && !p.Vt && p.wide && !p.c_f32 && p.ln_eps <= 0.f && (p.N + 127) / 128 <= p.rowstat_parts;
}

int launch_gemm_ws(const GCParams& p, hipStream_t st) {
    if ((long)p.M * p.lda * 2 >= 0x7FFF0000L) return set_error(MDX_EINVAL, "gemm_ws: A exceeds the 2 GiB buffer window");
    const bool ln = p.ln_eps > 0.f;
    if (ln && !ws_fuses_layernorm(p)) return set_error(MDX_EINVAL, "gemm_ws: fused LayerNorm needs ln_csum and a plain epilogue (GEGLU: also ln_stats)");
    const bool lns = ln && p.ln_stats != nullptr;
    if (p.rowstat) {
        if (!ws_emits_rowstat(p)) return set_error(MDX_EINVAL, "gemm_ws: rowstat_out needs the plain wide store phase and one part per 128-column tile");
        // parts beyond the N-tiles stay zero: clear them here (a tiny memset node; the common N = 320 case has exactly 3 parts and skips it)
        const int ntile = (p.N + 127) / 128;
        if (p.rowstat_parts > ntile)
            if (hipMemsetAsync(p.rowstat + (long)ntile * p.M * 2, 0, (size_t)(p.rowstat_parts - ntile) * p.M * 2 * sizeof(float), st) != hipSuccess)
                return set_error(MDX_ELAUNCH, "gemm_ws: rowstat memset");
        return launch_ws_one<false, false, 0, true>(p, st);
    }
    if (p.epi == 1) return lns ? launch_ws_one<true, false, 2, false>(p, st) : launch_ws_one<true, false, 0, false>(p, st);
    auto plain = [&](const GCParams& c) {
        return lns ? launch_ws_one<false, false, 2, false>(c, st) : ln ? launch_ws_one<false, false, 1, false>(c, st) : launch_ws_one<false, false, 0, false>(c, st);
    };
    if (!p.Vt) return plain(p);
    // fused q/k/v: the C columns [0, vt_from) and the transposed V columns [vt_from, N) are two launches over two row ranges of W
    GCParams c = p;
    c.N = p.vt_from; c.Vt = nullptr;
    int rc = plain(c);
    if (rc != MDX_OK) return rc;
    GCParams v = p;
    v.W = p.W + (long)p.vt_from * p.ldw; v.N = p.N - p.vt_from; v.bias = p.bias ? p.bias + p.vt_from : nullptr; v.R = nullptr; v.C = nullptr;
    v.ln_csum = p.ln_csum ? p.ln_csum + p.vt_from : nullptr;
    return lns ? launch_ws_one<false, true, 2, false>(v, st) : ln ? launch_ws_one<false, true, 1, false>(v, st) : launch_ws_one<false, true, 0, false>(v, st);
}

}  // namespace mdx
