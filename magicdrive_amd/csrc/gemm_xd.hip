// gemm_xd.hip — the "W-direct" persistent GEMM for the short-K transformer projections (round 6): 256 x 256 tiles, ONE wave per SIMD.
//
// Why (profiles/r03_xl_epilogue_ablation.log, profiles/README.md "one operand past the LDS"): at K = 640 a tile of gemm_xlp_kernel is 16 us of
// main loop + ~7 us of epilogue that overlaps with nothing (accumulators -> 16-bit, residual round trip, the store burst of all CUs at once), and
// its main loop moves both operands through the LDS (64 KB written + 192 KB read per 64-deep unit against 2048 cycles of MFMA).  Here
//   * the WEIGHTS never touch the LDS: the host packs them once in MFMA-fragment order (MdxGemmDesc.Wq, layout below) and every wave loads the
//     fragments of its own 64 columns global -> VGPR with fully coalesced buffer_load_dwordx4 (1 KiB per instruction), one unit (64 k) ahead,
//     in a second register set;
//   * the LDS ring carries only the activations: three 32 KB stages filled by LDS-DMA two units ahead, ONE s_barrier per unit;
//   * the wave grid is 1 x 4 (wave w owns all 256 rows x columns 64 w .. 64 w + 63: 16 x 4 tiles of v_mfma_f32_16x16x32, 256 accumulators in
//     AGPRs), so no weight fragment is fetched twice and the A fragments are read 4x instead of 8x; with one wave per SIMD the arch VGPRs
//     (256) are free for
//   * the finished tile, kept as 32 packed 16-byte row segments per lane (16 in registers, 16 in lane-private LDS) and DRAINED UNDER THE NEXT
//     TILE'S MAIN LOOP: one residual load + one 16-byte store per quarter unit, at fixed slots between the MFMAs, every one counted in the
//     hand-kept vmcnt arithmetic (buffer stores with a bounded descriptor: row / column tails are dropped by the hardware, so the
//     instruction count never depends on the data).
// The unit counter runs across tiles: the next tile's first activations / weights are fetched by the last units of this one.
//
// VM bookkeeping (gfx9: one in-order counter for loads, LDS-DMA and stores).  A unit is eight quarters (k-step q >> 2, row blocks
// 4 (q & 3) .. + 3: 16 MFMAs); quarter q issues, in this order: [R load] after MFMA 1, the W fragment v = q of the NEXT unit after MFMA 3,
// the A piece e = q of the unit TWO ahead after MFMA 9, [store | bias load] after MFMA 13.  Waits:
//   top of the unit (q = 0): the weights of k-step 0 (issued in quarters 0..3 of the previous unit) -> everything younger than W(3) may fly;
//   q = 4: the weights of k-step 1 (quarters 4..7 of the previous unit);
//   q = 7: everything older than this unit (this wave's A pieces of the next unit, issued one unit ago), then lgkmcnt(0) + the barrier:
//          behind it the next unit's stage is complete for every wave and the stage two ahead is free.
// XdSched computes the three counts from the table of optional operations per (unit of the tile, quarter).
//
// Rules this file follows about the registers its asm loads fill (weight fragments, residual segments, bias) — to hipcc they are ordinary values,
// defined at the asm statement; the data arrives later (tests/test_xd_isa.py checks the compiled ISA, DESIGN.md section 6 has the story):
//   * no control-flow join between an asm load and the wait that covers it, except the two loop back-edges: ONE straight-line tile body (the first
//     tile drains a zero-record pending tile), and before the tile loop's back-edge everything is waited for and every such value re-defined
//     behind the wait — a phi there is resolved by v_mov copies, and a copy of a register whose load is in flight copies stale bits;
//   * asm-loaded registers are consumed through an in-place re-definition behind their wait, never through a by-value copy;
//   * the accumulators are pinned to their AGPRs until the row block that converts them (otherwise the allocator copies all 256 to VGPRs behind the
//     last MFMA, spills long-lived offsets, and their scratch reloads wait vmcnt(0) inside the counted pipeline);
//   * wave-uniform per-unit offsets are made opaque where they are used, so that they are not hoisted out of the tile loop into ~100 SGPRs.
// Measured (profiles/r06_xd_ab.log): bit-identical to gemm_xlp_kernel, 0.89-1.04x its speed — the stores are issued for free between the MFMAs, but
// their write traffic slows the loop's loads by what the separate epilogue used to cost.  Option XD, off by default.
// Side builds: -DXD_TIMING (per-workgroup s_memtime sums into the op workspace: tools/xdone.py --timing), -DXD_ABL=bits (timing-only ablations),
// -DXD_STQ=1 (store quarters {0,3,4,7}), -DXD_STPOL=1/2/3 (nt / sc1 / sc0 sc1 stores).
//
// Packed weights: Wq[n / 16][k / 32][lane][8] with lane = ((k % 32) / 8) * 16 + n % 16 — the 64 lanes' A operands of one 16 x 32 MFMA block are one
// contiguous KiB; columns padded with zero blocks to a multiple of 256 (packing.pack_wq).  GEGLU: rows in the [32 value | 32 gate] order of
// packing.pack_geglu, so tiles j = 0, 1 of a wave are values and j = 2, 3 their gates.
//
// Arithmetic per element: k ascending in 32-wide MFMA steps, fp32; + bias; (GEGLU) value * gelu(gate); round to the 16-bit type; + residual in
// fp32, round — the same as gemm_xlp_kernel (the two agree bit for bit).
// Takes: plain / GEGLU epilogue, optional residual (plain only), K % 128 == 0, K >= 640, 16-byte C / R rows.  Replaces the persistent XL tile
// for attention_processor.py:141-157 (to_q / to_k / to_out), attention.py:200-280 (ff.net) at levels 1-2 and Transformer2DModel's proj_in / proj_out.
#include "common.h"
#include "launch.h"
#include "options.h"
#include "gemm_params.h"
#include "xl_layout.h"
#include "xl_dma.h"
#include <type_traits>

namespace mdx {

using namespace mdx_xl;

#ifndef XD_ABL
#define XD_ABL 0     // timing-only ablations (wrong results): 1 no stores, 2 no conversion, 4 no barrier, 8 no weight loads, 16 no activation DMA, 32 no residual loads, 128 every store into ONE 1-KiB window (no write traffic past the L2)
#endif
#if MDX_F16
#define XD_MFMA_NAME "v_mfma_f32_16x16x32_f16"
#else
#define XD_MFMA_NAME "v_mfma_f32_16x16x32_bf16"
#endif
#define XD_MFMA(acc_, w_, a_) asm volatile(XD_MFMA_NAME " %0, %1, %2, %0" : "+a"(acc_) : "v"(w_), "v"(a_))
#define XD_MFMA0(acc_, w_, a_) asm volatile(XD_MFMA_NAME " %0, %1, %2, 0" : "=a"(acc_) : "v"(w_), "v"(a_))

typedef __attribute__((ext_vector_type(4))) unsigned xd_u4_t;
union XdFrag { uint4 u; bf16x8_t v; xd_u4_t r; };             // r: written by the asm loads in place (no copy may sit between the load and its wait)

__device__ __forceinline__ void xd_gload(xd_u4_t& dst, const xl_rsrc_t rs, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int IMM>
__device__ __forceinline__ void xd_gload_imm(xd_u4_t& dst, const xl_rsrc_t rs, unsigned voff) {       // + IMM bytes (instruction offset: inside the range check)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(rs), "n"(IMM) : "memory");
}
__device__ __forceinline__ void xd_gstore(const xd_u4_t src, const xl_rsrc_t rs, unsigned voff, int soff) {
#if defined(XD_STPOL) && XD_STPOL == 1
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" : : "v"(src), "v"(voff), "s"(rs), "s"(soff) : "memory");
#elif defined(XD_STPOL) && XD_STPOL == 2
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1" : : "v"(src), "v"(voff), "s"(rs), "s"(soff) : "memory");
#elif defined(XD_STPOL) && XD_STPOL == 3
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc0 sc1" : : "v"(src), "v"(voff), "s"(rs), "s"(soff) : "memory");
#else
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(src), "v"(voff), "s"(rs), "s"(soff) : "memory");
#endif
}

// Optional VM operations per (unit u of the tile, quarter q).  u = 0..9 explicit, anything else (the generic unit): none.
// pre: residual load (after MFMA 1); post: store or bias load (after MFMA 13).
template <bool GEGLU, bool HAS_R>
struct XdSched {
    static constexpr int GEN = 10;
#ifndef XD_STQ
#define XD_STQ 0
#endif
    // store quarters.  XD_STQ = 1 (no-residual kernels): {0, 3, 4, 7} / GEGLU {3, 7} — a store behind the weight load of quarter 3 / 7 is younger than
    // every load the next unit's first waits cover, so it has 8-9 quarters to be acknowledged instead of 6-8
    static constexpr int store_slot(int q) {                     // index of the unit's store issued in quarter q, -1: none
        if (XD_STQ && !HAS_R) return GEGLU ? (q == 3 ? 0 : q == 7 ? 1 : -1) : (q == 0 ? 0 : q == 3 ? 1 : q == 4 ? 2 : q == 7 ? 3 : -1);
        return GEGLU ? (q == 0 ? 0 : q == 4 ? 1 : -1) : ((q & 1) ? -1 : q >> 1);
    }
    static constexpr bool has_store(bool dr, int u, int q) { return dr && u >= 1 && u <= 8 && store_slot(q) >= 0; }
    static constexpr bool has_bias(int u, int q) { return u == 0 && q >= 4; }
    static constexpr bool has_rload(bool dr, int u, int q) { return dr && HAS_R && u >= 0 && u <= 7 && (q & 1); }
    static constexpr int pre(bool dr, int u, int q) { return has_rload(dr, u, q) ? 1 : 0; }
    static constexpr int post(bool dr, int u, int q) { return (has_store(dr, u, q) ? 1 : 0) + (has_bias(u, q) ? 1 : 0); }
    // the unit in front of unit u of a tile: u - 1, or (u = 0 / generic) a unit without optional operations
    static constexpr int prev(int u) { return (u >= 1 && u <= 9) ? u - 1 : GEN; }
    static constexpr int top(bool dr, int u) {                 // younger than W(3) of the previous unit
        const int pu = prev(u);
        int n = 1 + post(dr, pu, 3);
        for (int q = 4; q < 8; ++q) n += 2 + pre(dr, pu, q) + post(dr, pu, q);
        return n;
    }
    static constexpr int mid(bool dr, int u) {                 // younger than W(7) of the previous unit, at the top of quarter 4
        int n = 1 + post(dr, prev(u), 7);
        for (int q = 0; q < 4; ++q) n += 2 + pre(dr, u, q) + post(dr, u, q);
        return n;
    }
    static constexpr int n7(bool dr, int u) {                  // issued by this unit in front of quarter 7 (+ what the previous unit issued behind its last A piece)
        int n = post(dr, prev(u), 7);
        for (int q = 0; q < 7; ++q) n += 2 + pre(dr, u, q) + post(dr, u, q);
        return n;
    }
};

template <bool GEGLU, bool HAS_R>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_xd_kernel(GCParams p) {
    using S = XdSched<GEGLU, HAS_R>;
    constexpr int UNIT = 256 * 128, NSTG = 3, RING = NSTG * UNIT;
    constexpr int NCH = GEGLU ? 16 : 32;                          // 16-byte row segments of the finished tile per lane
    constexpr int NHR = GEGLU ? 0 : 16;                           // ... of which in registers; the rest (16) in lane-private LDS behind the ring
    constexpr int GEN = S::GEN;
    static_assert(!(GEGLU && HAS_R), "GEGLU takes no residual");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int T = p.K >> 6;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_t*)smem;
    unsigned char* hold = smem + RING + tid * 16;                 // chunk c >= NHR at hold + (c - NHR) * 4096

    // ---- tile-invariant per-lane offsets ----
    // piece e of this wave: rows (wave * 8 + e) * 8 + lane / 8, logical chunk (lane & 7) ^ swz(row) = c0 ^ 4 (e & 1): two per-lane bases, the
    // piece's 8 e rows added as a scalar (into the VECTOR offset: the range check that zero-fills the rows past M sees only that)
    const int a_row0 = wave * 64 + (lane >> 3);
    const unsigned a_vb0 = (unsigned)((long)a_row0 * p.lda * 2 + (((lane & 7) ^ swz(a_row0)) << 4));
    const unsigned a_vb1 = (unsigned)((long)a_row0 * p.lda * 2 + (((lane & 7) ^ swz(a_row0) ^ 4) << 4));
    const int a_e8 = __builtin_amdgcn_readfirstlane((int)(8 * p.lda * 2));
    const unsigned b_voff = (unsigned)((wave * 64 + 4 * fq) * 4);
    const int sw = (fr >> 1) & 7;                                 // swz(16 i + fr) for every i
    const unsigned a_rd0 = (unsigned)(fr * 128 + ((fq ^ sw) << 4)), a_rd1 = (unsigned)(fr * 128 + (((4 + fq) ^ sw) << 4));
    const unsigned w_voff = (unsigned)lane * 16u;
    const int nkb = p.K >> 5;                                     // KiB per 16-column block of Wq
    const int No = GEGLU ? p.N / 2 : p.N;
    // store layout of gemm_xlp_kernel: after the transposes lane (fr, fq) carries, per 16-row block, segment A = row fr & 7, segment B = row 8 + (fr & 7)
    const bool lo8 = fr < 8;
    const int lcolA = GEGLU ? wave * 32 + 8 * fq : wave * 64 + 8 * fq + (lo8 ? 0 : 32);
    const int lcolB = wave * 64 + 8 * fq + (lo8 ? 32 : 0);
    const int lrowA = GEGLU ? fr : (fr & 7), lrowB = 8 + (fr & 7);
    const int c_blk = __builtin_amdgcn_readfirstlane((int)(16 * p.ldc * 2)), r_blk = __builtin_amdgcn_readfirstlane((int)(16 * p.ldr * 2));

    // ---- tile walk ----
    int tm = 0, tn = 0;
    auto advance = [&](int& b) -> bool {
        for (; b < p.nblk; b += (int)gridDim.x) {
            const bool ok = p.swz != 2 ? tile_coords_at(p, b, tm, tn) : raster_tile(b, p.mt, p.nt, p.gm, p.gn, &tm, &tn);
            if (ok) return true;
        }
        return false;
    };
    int bid = blockIdx.x;
    if (!advance(bid)) return;

    xl_rsrc_t rsA, rsW, rsAn, rsWn, rsB, rsC, rsR;                // this tile's A / Wq / bias, the next tile's A / Wq, the PENDING tile's C / R
    unsigned cvA = XL_OOB, cvB = XL_OOB, rvA = XL_OOB, rvB = XL_OOB;                  // pending tile: per-lane offsets incl. the column tail
    auto dead = [&]() { xl_rsrc_t r = xl_make_rsrc(p.A); r.z = 0u; return r; };
    auto mk_a = [&](int m) { return xl_make_rsrc_bounded(p.A + (long)m * p.lda, (long)min(p.M - m, 256) * p.lda * 2); };
    auto mk_w = [&](int n) { return xl_make_rsrc((const unsigned char*)p.Wq + (long)(n >> 4) * nkb * 1024); };
    auto mk_b = [&](int n) {
        if (!p.bias) return dead();
        return xl_make_rsrc_bounded(p.bias + n, (long)(p.N - n) * 4);
    };
    int m0 = tm * 256, n0 = tn * 256;
    rsA = mk_a(m0); rsW = mk_w(n0); rsB = mk_b(n0);
    rsAn = dead(); rsWn = dead(); rsC = dead(); rsR = dead();

    f32x4_t acc[16][4];
    XdFrag wf[2][2][4], af[2][4];
    xd_u4_t bq[4], held[NHR ? NHR : 1], rr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = xd_u4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 4; ++e) rr[e] = xd_u4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < (NHR ? NHR : 1); ++c) held[c] = xd_u4_t{0u, 0u, 0u, 0u};

    int stg = 0;                                                  // ring stage of the current unit
    auto stage_of = [&](int ahead) { int s = stg + ahead; return s >= NSTG ? s - NSTG : s; };
    // A piece e of local unit tt (of this tile, or past its end: of the next tile) into the stage `ahead` units in front
    // (opaque(): a wave-uniform value the optimiser may not fold or hoist — otherwise every per-unit scalar offset of the unrolled units is
    // precomputed outside the tile loop and the ~100 SGPRs spill into VGPR lanes, then VGPRs into scratch, whose reloads drain vmcnt inside the loop)
    auto opaque = [](int x) { asm volatile("" : "+s"(x)); return x; };
    auto issue_a = [&](int e, int tt, int ahead) {
        const bool nx = tt >= T;
        xl_rsrc_t r;
        r.x = nx ? rsAn.x : rsA.x; r.y = nx ? rsAn.y : rsA.y; r.z = nx ? rsAn.z : rsA.z; r.w = rsA.w;
        xl_glds(r, lds0 + (unsigned)(stage_of(ahead) * UNIT + (wave * 8 + e) * 1024), ((e & 1) ? a_vb1 : a_vb0) + (unsigned)(e * opaque(a_e8)), (nx ? tt - T : tt) * 128);
    };
    // W fragment v (k-step v >> 2, column block v & 3) of local unit tt into register set b
    auto issue_w = [&](auto B_, int v, int tt) {
        constexpr int b = decltype(B_)::value;
        const bool nx = tt >= T;
        xl_rsrc_t r;
        r.x = nx ? rsWn.x : rsW.x; r.y = nx ? rsWn.y : rsW.y; r.z = nx ? rsWn.z : rsW.z; r.w = rsW.w;
        const int u2 = (nx ? tt - T : tt) * 2 + (v >> 2);
        xd_gload(wf[b][v >> 2][v & 3].r, r, w_voff, ((wave * 4 + (v & 3)) * nkb + u2) << 10);
    };
    auto read_af = [&](int buf, int stage, unsigned rd, int g) {
        const unsigned char* s_ = smem + stage * UNIT + rd + g * (4 * 2048);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) af[buf][ii].u = *(const uint4*)(s_ + ii * 2048);
    };
    auto take = [&](auto C_) -> xd_u4_t {                         // packed segment c of the pending tile
        constexpr int c = decltype(C_)::value;
        if constexpr (c < NHR) return held[c];
        else { const uint4 v = *(const uint4*)(hold + (c - NHR) * 4096); return xd_u4_t{v.x, v.y, v.z, v.w}; }
    };
    // (the residual registers are written behind the compiler's back: re-define them IN PLACE where they are consumed — behind the wait that
    // covers their load — so that neither the sum nor a register copy of them can be scheduled in front of that wait)
    auto add_r = [&](xd_u4_t v, xd_u4_t& r) {
        asm volatile("" : "+v"(r));
        return xd_u4_t{add2bf(v.x, r.x), add2bf(v.y, r.y), add2bf(v.z, r.z), add2bf(v.w, r.w)};
    };
    // store / residual addressing of pending segment c: plain c = 2 i + h (h = 1: segment B), GEGLU c = i
    // (the 16-row block offset rides in the VECTOR offset: the descriptor's range check — which drops the rows past M — does not see soffset)
    auto seg_voff = [&](int c, bool res) {
        const int i = GEGLU ? c : (c >> 1);
        const bool b = !GEGLU && (c & 1);
        return (res ? (b ? rvB : rvA) : (b ? cvB : cvA)) + (unsigned)(i * opaque(res ? r_blk : c_blk));
    };

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<bool, true> BT;
    typedef std::integral_constant<bool, false> BF_;

    // ---- one unit (64 k) of the main loop: eight quarters ----
    auto quarter = [&](auto SET_, auto U_, auto DR_, auto Q_, int t) {
        constexpr int SET = decltype(SET_)::value, U = decltype(U_)::value, q = decltype(Q_)::value;
        constexpr bool DR = decltype(DR_)::value;
        constexpr int ks = q >> 2, g = q & 3;
        constexpr bool ST = S::has_store(DR, U, q), RL = S::has_rload(DR, U, q), BL = S::has_bias(U, q);
        constexpr int cst = (GEGLU ? 2 : 4) * (U - 1) + (S::store_slot(q) >= 0 ? S::store_slot(q) : 0);     // segment stored in this quarter
        constexpr int crl = 4 * U + (q >> 1);                                             // segment whose residual is fetched in this quarter
        typedef std::integral_constant<int, SET ^ 1> OSET;
        if constexpr (q == 0) xl_wait_vmcnt<S::top(DR, U)>();
        if constexpr (q == 4) xl_wait_vmcnt<S::mid(DR, U)>();
        if constexpr (q == 7) {
            xl_wait_vmcnt<S::n7(DR, U)>();
            xl_wait_lgkm0();
            if (!(XD_ABL & 4)) __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        if constexpr (q < 7) read_af((q + 1) & 1, stg, ((q + 1) >> 2) ? a_rd1 : a_rd0, (q + 1) & 3);
        else read_af(0, stage_of(1), a_rd0, 0);
        xd_u4_t sv = xd_u4_t{0u, 0u, 0u, 0u};
        if constexpr (ST) sv = take(std::integral_constant<int, ST ? cst : 0>{});
        asm volatile("" ::: "memory");
#pragma unroll
        for (int idx = 0; idx < 16; ++idx) {
            const int ii = idx >> 2, j = idx & 3;
            if constexpr (U == 0 && ks == 0) XD_MFMA0(acc[4 * g + ii][j], wf[SET][ks][j].v, af[q & 1][ii].v);
            else XD_MFMA(acc[4 * g + ii][j], wf[SET][ks][j].v, af[q & 1][ii].v);
            if (idx == 1) {
                if constexpr (RL) { if (!(XD_ABL & 32)) xd_gload(rr[q >> 1], rsR, seg_voff(crl, true), 0); }   // used by the store at (U + 1, q - 1)
            }
            if (idx == 3 && !(XD_ABL & 8)) issue_w(OSET{}, q, t + 1);
            if (idx == 9 && !(XD_ABL & 16)) issue_a(q, t + 2, 2);
            if (idx == 13) {
                if constexpr (BL) xd_gload_imm<(BL ? q - 4 : 0) * 64>(bq[BL ? q - 4 : 0], rsB, b_voff);
                if constexpr (ST) {
                    if constexpr (HAS_R) sv = add_r(sv, rr[q >> 1]);
                    if (!(XD_ABL & 1)) xd_gstore(sv, rsC, (XD_ABL & 128) ? (unsigned)(lane * 16) : seg_voff(cst, false), 0);
                }
            }
        }
    };
    auto unit = [&](auto SET_, auto U_, auto DR_, int t) {
        t = opaque(t);
        quarter(SET_, U_, DR_, std::integral_constant<int, 0>{}, t); quarter(SET_, U_, DR_, std::integral_constant<int, 1>{}, t);
        quarter(SET_, U_, DR_, std::integral_constant<int, 2>{}, t); quarter(SET_, U_, DR_, std::integral_constant<int, 3>{}, t);
        quarter(SET_, U_, DR_, std::integral_constant<int, 4>{}, t); quarter(SET_, U_, DR_, std::integral_constant<int, 5>{}, t);
        quarter(SET_, U_, DR_, std::integral_constant<int, 6>{}, t); quarter(SET_, U_, DR_, std::integral_constant<int, 7>{}, t);
        stg = stage_of(1);
    };

    // ---- finished accumulators -> packed row segments (registers / lane-private LDS); gemm_xlp_kernel's transposes ----
    auto put = [&](auto C_, uint4 v) {
        constexpr int c = decltype(C_)::value;
        if constexpr (c < NHR) held[c] = xd_u4_t{v.x, v.y, v.z, v.w};
        else *(uint4*)(hold + (c - NHR) * 4096) = v;
    };
    auto convert = [&]() {
        if (XD_ABL & 2) return;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the last MFMAs have written their accumulators
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bq[j]));
        auto conv_i = [&](auto I_) {
            constexpr int i = decltype(I_)::value;
            // one row block at a time: left alone, the allocator copies all 256 accumulators to VGPRs right behind the last MFMA (and spills
            // long-lived offsets to make room: their reloads then sit in the main loop behind a vmcnt(0)).  Pinning the block's accumulators to
            // AGPRs HERE keeps them there until now.
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[i][j]));
            unsigned tx[4], ty[4];
#pragma unroll
            for (int j = 0; j < (GEGLU ? 2 : 4); ++j) {
                const float bb[4] = {__uint_as_float(bq[j].x), __uint_as_float(bq[j].y), __uint_as_float(bq[j].z), __uint_as_float(bq[j].w)};
                float o[4];
                if (GEGLU) {
                    const float gg[4] = {__uint_as_float(bq[j + 2].x), __uint_as_float(bq[j + 2].y), __uint_as_float(bq[j + 2].z), __uint_as_float(bq[j + 2].w)};
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2_t ge = gelu_erf_f2(f32x2_t{acc[i][j + 2][e] + gg[e], acc[i][j + 2][e + 1] + gg[e + 1]});
                        o[e] = (acc[i][j][e] + bb[e]) * ge.x; o[e + 1] = (acc[i][j][e + 1] + bb[e + 1]) * ge.y;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e] + bb[e];
                }
                tx[j] = pack2bf(o[0], o[1]); ty[j] = pack2bf(o[2], o[3]);
            }
            uint4 ch[2];
#pragma unroll
            for (int c = 0; c < (GEGLU ? 1 : 2); ++c) {
                unsigned ax = tx[2 * c], ay = ty[2 * c], bx = tx[2 * c + 1], by = ty[2 * c + 1];
                { auto r_ = __builtin_amdgcn_permlane32_swap(ax, bx, false, false); ax = r_[0]; bx = r_[1]; }
                { auto r_ = __builtin_amdgcn_permlane32_swap(ay, by, false, false); ay = r_[0]; by = r_[1]; }
                { auto r_ = __builtin_amdgcn_permlane16_swap(ax, bx, false, false); ax = r_[0]; bx = r_[1]; }
                { auto r_ = __builtin_amdgcn_permlane16_swap(ay, by, false, false); ay = r_[0]; by = r_[1]; }
                ch[c] = make_uint4(ax, ay, bx, by);
            }
            if constexpr (GEGLU) {
                put(std::integral_constant<int, i>{}, ch[0]);
            } else {
                uint4 rot;
                rot.x = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[1].x, 0x128, 0xf, 0xf, false);
                rot.y = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[1].y, 0x128, 0xf, 0xf, false);
                rot.z = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[1].z, 0x128, 0xf, 0xf, false);
                rot.w = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[1].w, 0x128, 0xf, 0xf, false);
                uint4 vA, vB;
                vA.x = lo8 ? ch[0].x : rot.x; vA.y = lo8 ? ch[0].y : rot.y; vA.z = lo8 ? ch[0].z : rot.z; vA.w = lo8 ? ch[0].w : rot.w;
                vB.x = lo8 ? rot.x : ch[0].x; vB.y = lo8 ? rot.y : ch[0].y; vB.z = lo8 ? rot.z : ch[0].z; vB.w = lo8 ? rot.w : ch[0].w;
                put(std::integral_constant<int, 2 * i>{}, vA);
                put(std::integral_constant<int, 2 * i + 1>{}, vB);
            }
        };
        conv_i(std::integral_constant<int, 0>{}); conv_i(std::integral_constant<int, 1>{}); conv_i(std::integral_constant<int, 2>{}); conv_i(std::integral_constant<int, 3>{});
        conv_i(std::integral_constant<int, 4>{}); conv_i(std::integral_constant<int, 5>{}); conv_i(std::integral_constant<int, 6>{}); conv_i(std::integral_constant<int, 7>{});
        conv_i(std::integral_constant<int, 8>{}); conv_i(std::integral_constant<int, 9>{}); conv_i(std::integral_constant<int, 10>{}); conv_i(std::integral_constant<int, 11>{});
        conv_i(std::integral_constant<int, 12>{}); conv_i(std::integral_constant<int, 13>{}); conv_i(std::integral_constant<int, 14>{}); conv_i(std::integral_constant<int, 15>{});
    };
    // the pending tile's C / R descriptors and per-lane offsets (column tail: out-of-range offset, the hardware drops the lane)
    auto set_pending = [&](int cm, int cn) {
        const int co = GEGLU ? cn / 2 : cn;                       // first output column of the tile
        const long rows = min(p.M - cm, 256);
        rsC = xl_make_rsrc_bounded((bf16_t*)p.C + (long)cm * p.ldc + co, rows * p.ldc * 2 - (long)co * 2);
        const bool okA = co + lcolA + 8 <= No, okB = co + lcolB + 8 <= No;
        cvA = okA ? (unsigned)(((long)lrowA * p.ldc + lcolA) * 2) : XL_OOB; cvB = okB ? (unsigned)(((long)lrowB * p.ldc + lcolB) * 2) : XL_OOB;
        if (HAS_R) {
            rsR = xl_make_rsrc_bounded((const bf16_t*)p.R + (long)cm * p.ldr + co, rows * p.ldr * 2 - (long)co * 2);
            rvA = okA ? (unsigned)(((long)lrowA * p.ldr + lcolA) * 2) : XL_OOB; rvB = okB ? (unsigned)(((long)lrowB * p.ldr + lcolB) * 2) : XL_OOB;
        }
    };

    // ---- prologue: activations of units 0 and 1, weights of unit 0 ----
#pragma unroll
    for (int e = 0; e < 8; ++e) issue_a(e, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) issue_a(e, 1, 1);
#pragma unroll
    for (int v = 0; v < 8; ++v) issue_w(I0{}, v, 0);
#pragma unroll
    for (int v = 0; v < 8; ++v) wf[1][v >> 2][v & 3].u = make_uint4(0, 0, 0, 0);
    xl_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    read_af(0, 0, a_rd0, 0);

    // ONE straight-line tile body: the first tile drains a pending tile too — one whose descriptors have zero records, so its stores are dropped
    // and its residual loads return zeros (they still count, so the wait arithmetic is the same for every tile).  No control-flow join sits
    // between an asynchronous load and the wait that covers it except the two loop back-edges, see below.
#ifdef XD_TIMING
    unsigned long long tm_loop = 0, tm_conv = 0, tm_tiles = 0, tm_u04 = 0, tm_u58 = 0;
    const unsigned long long tm_begin = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
#ifdef XD_TIMING
        const unsigned long long tm_a = __builtin_amdgcn_s_memtime();
#endif
        int nb = bid + (int)gridDim.x;
        const int cm0 = m0, cn0 = n0;
        const bool has_next = advance(nb);
        if (has_next) { m0 = tm * 256; n0 = tn * 256; rsAn = mk_a(m0); rsWn = mk_w(n0); }
        else { rsAn = dead(); rsWn = dead(); }
        unit(I0{}, std::integral_constant<int, 0>{}, BT{}, 0);
        unit(I1{}, std::integral_constant<int, 1>{}, BT{}, 1);
        unit(I0{}, std::integral_constant<int, 2>{}, BT{}, 2);
        unit(I1{}, std::integral_constant<int, 3>{}, BT{}, 3);
        unit(I0{}, std::integral_constant<int, 4>{}, BT{}, 4);
#ifdef XD_TIMING
        const unsigned long long tm_m1 = __builtin_amdgcn_s_memtime();
#endif
        unit(I1{}, std::integral_constant<int, 5>{}, BT{}, 5);
        unit(I0{}, std::integral_constant<int, 6>{}, BT{}, 6);
        unit(I1{}, std::integral_constant<int, 7>{}, BT{}, 7);
        unit(I0{}, std::integral_constant<int, 8>{}, BT{}, 8);
#ifdef XD_TIMING
        const unsigned long long tm_m2 = __builtin_amdgcn_s_memtime();
        tm_u04 += tm_m1 - tm_a; tm_u58 += tm_m2 - tm_m1;
#endif
        unit(I1{}, std::integral_constant<int, 9>{}, BT{}, 9);
        for (int t = 10; t < T; t += 2) {
            unit(I0{}, std::integral_constant<int, GEN>{}, BF_{}, t);
            unit(I1{}, std::integral_constant<int, GEN>{}, BF_{}, t + 1);
        }
#ifdef XD_TIMING
        const unsigned long long tm_b = __builtin_amdgcn_s_memtime();
#endif
        convert();
#ifdef XD_TIMING
        tm_loop += tm_b - tm_a; tm_conv += __builtin_amdgcn_s_memtime() - tm_b; ++tm_tiles;
#endif
        set_pending(cm0, cn0);
        // The registers the asm loads fill are ordinary values to the register allocator: where two paths meet (this back-edge) it may MOVE
        // them — and a v_mov of a register whose load is still in flight copies stale bits (seen: phi copies of the weight fragments here,
        // ~10 % of the launches wrong in the fragments issued last).  Everything issued so far landed during convert(); wait for it formally
        // and re-define the values behind the wait, so that any copy the allocator makes for the next iteration is a copy of landed data.
        xl_wait_vmcnt<0>();
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 8; ++v) asm volatile("" : "+v"(wf[b][v >> 2][v & 3].r));
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(rr[e]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bq[j]));
        if (!has_next) break;
        bid = nb;
        rsA = rsAn; rsW = rsWn; rsB = mk_b(n0);
    }
#ifdef XD_TIMING
    if (p.timing && tid == 0) {                                   // cycles (s_memtime) per workgroup: main loops, conversions, tiles, all
        p.timing[blockIdx.x * 8 + 0] = tm_loop; p.timing[blockIdx.x * 8 + 1] = tm_conv; p.timing[blockIdx.x * 8 + 2] = tm_tiles;
        p.timing[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime() - tm_begin;
        p.timing[blockIdx.x * 8 + 4] = tm_u04; p.timing[blockIdx.x * 8 + 5] = tm_u58;
    }
#endif
    // ---- the last tile's segments: nothing left to hide them under ----
    xl_wait_vmcnt<0>();
    auto flush = [&](auto C_) {
        constexpr int c = decltype(C_)::value;
        xd_u4_t v = take(C_);
        if constexpr (HAS_R) {
            xd_u4_t r;
            xd_gload(r, rsR, seg_voff(c, true), 0);
            xl_wait_vmcnt<0>();
            v = add_r(v, r);
        }
        xd_gstore(v, rsC, seg_voff(c, false), 0);
    };
    flush(std::integral_constant<int, 0>{}); flush(std::integral_constant<int, 1>{}); flush(std::integral_constant<int, 2>{}); flush(std::integral_constant<int, 3>{});
    flush(std::integral_constant<int, 4>{}); flush(std::integral_constant<int, 5>{}); flush(std::integral_constant<int, 6>{}); flush(std::integral_constant<int, 7>{});
    flush(std::integral_constant<int, 8>{}); flush(std::integral_constant<int, 9>{}); flush(std::integral_constant<int, 10>{}); flush(std::integral_constant<int, 11>{});
    flush(std::integral_constant<int, 12>{}); flush(std::integral_constant<int, 13>{}); flush(std::integral_constant<int, 14>{}); flush(std::integral_constant<int, 15>{});
    if constexpr (NCH > 16) {
        flush(std::integral_constant<int, 16>{}); flush(std::integral_constant<int, 17>{}); flush(std::integral_constant<int, 18>{}); flush(std::integral_constant<int, 19>{});
        flush(std::integral_constant<int, 20>{}); flush(std::integral_constant<int, 21>{}); flush(std::integral_constant<int, 22>{}); flush(std::integral_constant<int, 23>{});
        flush(std::integral_constant<int, 24>{}); flush(std::integral_constant<int, 25>{}); flush(std::integral_constant<int, 26>{}); flush(std::integral_constant<int, 27>{});
        flush(std::integral_constant<int, 28>{}); flush(std::integral_constant<int, 29>{}); flush(std::integral_constant<int, 30>{}); flush(std::integral_constant<int, 31>{});
    }
}

// Does the W-direct kernel take this problem?  (q: the GCParams launch_xl prepared — tile order, wide, nblk.)
bool xd_supported(const GCParams& q) {
    if (!q.Wq || q.batch > 1 || q.splitk > 1 || q.c_f32 || q.Vt || q.col_split || q.temb || q.rowstat || q.ln_eps > 0.f) return false;
    if (q.epi != 0 && q.epi != 1) return false;
    if (q.epi == 1 && (q.R || (q.N % 64))) return false;
    if ((q.K % 128) || q.K < 640 || (q.N % 16) || !q.wide) return false;
    if (q.R && ((q.ldr % 8) || ((uintptr_t)q.R & 15))) return false;
    if (((uintptr_t)q.Wq & 15) || (q.bias && ((uintptr_t)q.bias & 15))) return false;
    if ((long)256 * q.lda * 2 >= 0x40000000L || (long)256 * q.ldc * 2 >= 0x40000000L || (long)256 * q.ldr * 2 >= 0x40000000L) return false;
    if ((long)16 * (q.K >> 5) * 1024 >= 0x40000000L) return false;
    return true;
}

int launch_gemm_xd(const GCParams& q0, int cus, hipStream_t st) {
    GCParams q = q0;
#ifdef XD_TIMING
    q.timing = (q0.ws && q0.ws_bytes >= (long)cus * 64) ? (unsigned long long*)q0.ws : nullptr;
#endif
    const bool geglu = q.epi == 1, has_r = q.R != nullptr;
    constexpr size_t ring = 3 * 256 * 128;
    auto go = [&](auto kern, size_t smem, const char* tag) -> int {
        if (int rc = ensure_dyn_smem((const void*)kern, smem, "xd")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(256), smem, st, q);
        return check_launch(tag);
    };
    if (geglu) return go(gemm_xd_kernel<true, false>, ring + 16 * 4096, "gemm_xd_kernel<256x256,geglu>");
    if (has_r) return go(gemm_xd_kernel<false, true>, ring + 16 * 4096, "gemm_xd_kernel<256x256,gemm+res>");
    return go(gemm_xd_kernel<false, false>, ring + 16 * 4096, "gemm_xd_kernel<256x256,gemm>");
}

}  // namespace mdx
