// gemm_xl.hip — the large-shape bf16 MFMA GEMM / implicit-GEMM convolution main loop for gfx950: 256 x {256,160} x 64 tiles,
// operands global -> LDS by LDS-DMA, quadrant phases, counted vmcnt, two wave groups staggered by one barrier.
//
// Why a third main loop (after gemm_conv.hip's 128x128 register-staged tile and gemm_pp.hip's register-staged ping-pong):
// round 1 measured both as bound by the global -> register -> LDS staging path (profiles/README.md: ~32 B/clk per CU, a load phase
// of 1.2-1.7 k cycles beside 1.0 k cycles of MFMA issue).  Here nothing is staged through registers:
//   * every operand byte goes global -> LDS with `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction, full 128-byte lines per
//     row), the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE address
//     (xl_layout.h); conv zero padding / row tails are lanes whose voffset is out of the descriptor's range (the DMA writes zeros);
//   * a K slab (64 deep) is four load units — A0, B0, B1, A1: the rows the four quadrant phases of a wave tile read first.  A unit's
//     LDS slot is refilled one phase after its last fragment read, for the slab TWO ahead, so three units are always in flight
//     and the only wait in the loop is ONE counted `s_waitcnt vmcnt(3 units)` per slab (never a drain);
//   * a phase = {fragment reads of one quadrant + the unit refill} | s_barrier | {MFMAs of the quadrant} | s_barrier.  Waves 0-3
//     and 4-7 (w and w + 4 share a SIMD) run the same sequence shifted by one barrier: one group's MFMA segment is the other
//     group's LDS / DMA segment ("matrix beside memory", MI355X_MICROARCH.md "Two waves per SIMD"); s_setprio 1 around the MFMAs;
//   * fragments are read ONCE per slab (A half 32 + both B parts in registers beside the accumulators).
// Reduction order per output element: k ascending in 32-wide MFMA steps (conv: (channel block, ky, kx) slabs), fp32 accumulation —
// a permutation of gemm_conv.hip's order for conv, identical for GEMM.
//
// Epilogue: bias (+ the per-(step, image) temb row) staged once per tile as fp32 addend rows in LDS; applied in registers with
// SiLU / GEGLU; the bf16 tile is transposed through LDS (whole 256-row tile at once: the operand ring is dead) and written as
// 16-byte row segments with the residual added — same arithmetic per element as gemm_conv.hip / gemm_pp.hip.
//
// Replaces (through mdx_gemm_bf16 / mdx_conv2d_bf16, include/mdx.h): ATen addmm / conv2d of ResnetBlock2D (resnet.py:590-640),
// Down/Upsample2D convs (resnet.py:165-170, 198-222), the transformer projections and feed-forward (attention.py:200-280,
// attention_processor.py:141-157) at shapes with >= ~200 tiles.
#include "common.h"
#include "launch.h"
#include "options.h"
#include "gemm_params.h"
#include "xl_layout.h"
#include "xl_dma.h"
#include <type_traits>

namespace mdx {

using namespace mdx_xl;

constexpr int XL_SLOTS = 12;                   // distinct temb rows (images) one 256-row tile may span (4x7 images: 11)

// Tile order.  swz == 2 ("XCD-blocked panels", round 3; index math in xl_layout.h: raster_tile): the 32 workgroups an XCD runs side
// by side are ONE panel of gm M-tiles x gn N-tiles, so the operand bytes that XCD's L2 takes in per round of tiles are gm A-panels +
// gn W-panels instead of ~32 / nt A-panels + nt W-panels: at N = 5120 (20 N-tiles) the round-2 order streamed the whole 6.5 MB weight
// matrix through every 4 MiB L2 once per 1.6 M-tiles (13x read over-fetch, 60 % L2 hits: profiles/r02_pmc_summary.json); contiguous
// M-tiles also share their 3x3 halo rows inside one L2 (conv).
__device__ __forceinline__ bool xl_tile_coords_at(const GCParams& p, int bid, int& tm, int& tn) {
    if (p.swz != 2) return tile_coords_at(p, bid, tm, tn);
    return raster_tile(bid, p.mt, p.nt, p.gm, p.gn, &tm, &tn);
}
__device__ __forceinline__ bool xl_tile_coords(const GCParams& p, int& tm, int& tn) { return xl_tile_coords_at(p, (int)blockIdx.x, tm, tn); }

template <int BN, bool CONV, int SCHED>
__global__ __launch_bounds__(512, 2) void gemm_xl_kernel(GCParams p) {
    using G = Geo<BN>;
    constexpr int BM = 256, NTH = 512;
    constexpr int TI = G::TI, TJ = G::TJ, TJ0 = G::TJ0, TJ1 = TJ - TJ0, TIH = TI / 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = G::BNP * 128, BUF = A_BYTES + B_BYTES;
    constexpr int PA = G::PA, PB0 = G::PB0, PB1 = G::PB1;
    constexpr int UNIT_MAX = PA > PB0 ? PA : PB0;
    // SCHED 4 (round 6, 320-wide 3x3 / stride 1 / pad 1 convs): the three HORIZONTAL taps of a (channel block, ky) share ONE A slab.  In the
    // flattened pixel order tap kx of output pixel m reads input pixel m + (ky - 1) W + (kx - 1): the slab of tap kx = 0 shifted by kx rows.  So the
    // A region holds 258 rows — LDS row r = input pixel m0 - 1 + r (+ the ky row shift) — loaded once per group of three K slabs (two 272-row A
    // buffers alternating by group, the 40 KB weight slabs still alternate by slab), and the fragment reads of tap kx start kx rows further
    // down.  The left / right image border, where the flattened neighbour is a pixel of the next image row, is a per-lane select of the fragment
    // address: border lanes read a zero row.  Operand bytes into the CU per channel block: 3 x 34 + 9 x 40 KB instead of 9 x 72 (-29 %) — the lever
    // the round-4/6 ablations point at (zero-filling the A pieces of kx = 1, 2 without touching the instruction stream: +6...8 %,
    // profiles/r06_xl_a_bytes_ablation.log).  The row swizzle of this A region is row & 7 (conflict-free ds_read_b128 for row shifts 0, 1, 2;
    // the (row >> 1) & 7 rule of the other regions pairs even / odd rows and collides 2-way under an odd shift: tests/xl_layout_check.cpp).
    constexpr bool KXS = SCHED == 4;
    static_assert(!KXS || (BN == 320 && CONV), "the shared-tap schedule is the 320-wide conv's");
    constexpr int AS_ROWS = 272, AS_BYTES = AS_ROWS * 128;       // 256 + 2 halo rows (+ 6 of their piece) + 8 zero rows (the dummy pieces' target, the border select's source)
    constexpr int ZROW_OFF = 264 * 128;
    constexpr int B_STRIDE = KXS ? (G::BNP * 128) : (BM * 128 + G::BNP * 128);      // SCHED 4: [A buf 0 | A buf 1 | B buf 0 | B buf 1]; else [A | B] x 2
    constexpr int B_BASE = KXS ? 2 * AS_BYTES - BM * 128 : 0;                        // (b_lds / b_rd carry the A_BYTES of the interleaved layout)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                  // waves w and w + 4 share a SIMD
    const int wm = wave % G::WM, wn = wave / G::WM;
    int tile_m, tile_n;
    if (!xl_tile_coords(p, tile_m, tile_n)) return;
    // MDX_XL_TIMING=1 (tools/xl_timing.py): s_memtime stamps of wave 0 at 0 entry, 1 bookkeeping done (first DMA issue), 2 first slab
    // landed, 3 main loop done, 4 C tile staged (last half), 5 stores issued — (6 after the epilogue's first barrier, 7 accumulators staged, before the second barrier) — 8 x 8 bytes per workgroup into the caller's workspace
    unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define XL_STAMP(k) if (p.timing) ts_[k] = __builtin_amdgcn_s_memtime();
    XL_STAMP(0)
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nt = p.K / 64;                                    // K % 64 == 0 (xl_supported)

    // ---- DMA source bookkeeping -------------------------------------------------------------------------------------------
    // A: GEMM: descriptor base = row m0, voffset = row * lda + chunk, soffset = k.  CONV: base = the tile's first receptive-field
    // pixel, voffset = (input pixel of the row's tap (0,0)) - that, soffset = tap shift + channel block; invalid taps -> XL_OOB.
    long a_base_el;                                              // element offset of the descriptor base inside p.A
    int hw = 1, Wo = 1;
    if (CONV) {
        hw = p.Ho * p.Wo; Wo = p.Wo;
        const int b = m0 / hw, rem = m0 - b * hw;
        const int oy = rem / Wo, ox = rem - oy * Wo;
        a_base_el = (((long)b * p.Hi + (oy * p.sh - p.ph)) * p.Wi + (ox * p.sw - p.pw)) * p.lda;   // may point before p.A: never dereferenced there
    } else {
        a_base_el = (long)m0 * p.lda;
    }
    const xl_rsrc_t rsA = xl_make_rsrc(p.A + a_base_el);
    const xl_rsrc_t rsB = xl_make_rsrc(p.W + (long)n0 * p.ldw);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_t*)smem;      // LDS byte address of the dynamic region

    // SCHED 4 keeps its per-lane state in TWO registers (the 320-wide kernel has none to spare: 160 accumulators + 56 fragment registers):
    //   xs_bits  bits 3 (2 h + e) + ky: LDS row of piece (h, e) exists for row shift ky - 1;  12 + ky: the halo piece's;  16 + i / 24 + i: MFMA row
    //            tile i of this lane sits on the left / right image border
    //   xs_vb    byte offset of this lane's 16 bytes inside ANY A piece: (lane >> 3) rows + the chunk ((lane & 7) ^ (row & 7)) — piece rows start at
    //            multiples of 8, so the swizzle does not depend on the piece; the piece's first row is added as a scalar at issue time
    unsigned xs_bits = 0;
    const unsigned xs_vb = (unsigned)((long)(lane >> 3) * p.lda * 2 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    const unsigned xb_vb = (unsigned)((long)(lane >> 3) * p.ldw * 2 + (((lane & 7) ^ ((lane >> 4) & 3)) << 4));     // (row >> 1) & 7 of rows 0..7
    unsigned a_voff[2][PA];                                      // [unit A0 / A1][piece]
    unsigned a_taps[2][PA];                                      // CONV: bit t set = tap t reads inside the image (else zero)
    int a_lds[2][PA];                                            // LDS byte offset of the piece inside a buffer (wave-uniform)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < PA; ++e) {
            const int row0 = a_piece_row0<BN>(h, wave, e);
            a_lds[h][e] = row0 * 128;
            const int R = piece_lane_row(row0, lane);
            const int cl = piece_lane_chunk(row0, lane);
            const int m = m0 + R;
            a_taps[h][e] = 0;
            if constexpr (KXS) {
                // LDS row R = input pixel q = m0 - 1 + R of the centre image row (the ky shift rides in the scalar offset): bit ky = q exists and
                // its row y + ky - 1 is inside the image (the x position is the pixel's own: always inside)
                const long q = (long)m0 + R - 1;
                unsigned bits = 0;
                if (q >= 0 && q < p.M) {
                    const int b_ = (int)(q / hw), y_ = (int)(q - (long)b_ * hw) / Wo;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) if ((unsigned)(y_ + ky - 1) < (unsigned)p.Hi) bits |= 1u << ky;
                }
                xs_bits |= bits << (3 * (2 * h + e));             // (a_voff / a_taps stay unused: ONE base offset + the piece's wave-uniform row offset, XS_PIECE_A)
            } else
            if (CONV) {
                const int mm = min(m, p.M - 1);
                const int b = mm / hw, rem = mm - b * hw;
                const int oy = rem / Wo, ox = rem - oy * Wo;
                const int iy0 = oy * p.sh - p.ph, ix0 = ox * p.sw - p.pw;
                const long pix = ((long)b * p.Hi + iy0) * p.Wi + ix0;
                a_voff[h][e] = (unsigned)((pix * p.lda - a_base_el) * 2 + cl * 16);
                unsigned bits = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                    if (m < p.M && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) bits |= 1u << t;
                }
                a_taps[h][e] = bits;
            } else {
                a_voff[h][e] = m < p.M ? (unsigned)(((long)R * p.lda) * 2 + cl * 16) : XL_OOB;
            }
        }
    unsigned b_voff[2][UNIT_MAX];
    int b_lds[2][UNIT_MAX];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int e = 0; e < (part ? PB1 : PB0); ++e) {
            const int row0 = b_piece_row0<BN>(part, wave, e);
            if (row0 < 0) {                                      // dummy piece (BN = 160): zeros into scratch rows
                b_lds[part][e] = A_BYTES + b_dummy_row0<BN>(wave) * 128;
                b_voff[part][e] = XL_OOB;
            } else {
                b_lds[part][e] = A_BYTES + row0 * 128;
                const int R = piece_lane_row(row0, lane);
                const int cl = piece_lane_chunk(row0, lane);
                b_voff[part][e] = (n0 + R < p.N) ? (unsigned)(((long)R * p.ldw) * 2 + cl * 16) : XL_OOB;
            }
        }

    // SCHED 4: the slab's rows 256, 257 (+ 6 zero-filled rows of their piece) come from wave 7; the other waves' extra piece zero-fills rows 264-271
    if constexpr (KXS) {
        const int R = 256 + (lane >> 3);
        const long q = (long)m0 + R - 1;
        if (wave == 7 && R < 258 && q < p.M) {
            const int b_ = (int)(q / hw), y_ = (int)(q - (long)b_ * hw) / Wo;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) if ((unsigned)(y_ + ky - 1) < (unsigned)p.Hi) xs_bits |= 1u << (12 + ky);
        }
    }
    // slab T -> scalar byte offsets of its k position in A and W
    const int row_bytes = (int)(p.lda * 2);
    auto a_soff = [&](int T, int& tap) -> int {
        if (CONV) {
            const int cb = T / 9;
            tap = T - cb * 9;
            const int ky = tap / 3, kx = tap - ky * 3;
            return (ky * p.Wi + kx) * row_bytes + cb * 128;
        }
        tap = 0;
        return T * 128;
    };
    auto b_soff = [&](int T) -> int {
        if (CONV) {
            const int cb = T / 9, tap = T - cb * 9;
            return (tap * p.Cin + cb * 64) * 2;
        }
        return T * 128;
    };
    // Debug ablations (XL_DBG: 1 skip the MFMAs, 2 skip the DMA, 4 return after the main loop, 8 stage C but store nothing) exist only in builds with -DMDX_XL_ABLATE: a runtime test inside
    // the MFMA clusters splits their scheduling regions.
#ifdef MDX_XL_ABLATE
    const bool do_mma = !(p.dbg & 1), do_dma = !(p.dbg & 2);
    // round 6, the price of the 3x3 conv's nine A fetches per channel block (what a shared input halo would save): 16 = the A pieces of taps 1..8 are
    // issued with out-of-range offsets (zero fill: same instruction stream, none of their bytes enter the CU); 32 = they are not issued at all
    // (the hand-counted waits then cover less than they should: timing only, an optimistic bound)
    const bool a_oob = CONV && (p.dbg & 16), a_skip = CONV && (p.dbg & 32);
    const bool a_oob_kx = CONV && (p.dbg & 64);                   // 64 = zero fill only for kx != 0 (what sharing a row's three horizontal taps would save)
#else
    constexpr bool do_mma = true, do_dma = true, a_oob = false, a_skip = false, a_oob_kx = false;
#endif
    // ONE 1-KiB piece (e) of a load unit of slab T into buffer T & 1; nothing past the last slab (the waits account for it)
#define XL_PIECE_A(h, e, T)                                                                                                       \
    if constexpr ((e) < PA) {                                                                                                     \
        if ((T) < nt && do_dma && !(a_skip && ((T) % 9) != 0)) {                                                                  \
            int tap_;                                                                                                             \
            const int so_ = a_soff((T), tap_);                                                                                    \
            unsigned vo_ = CONV ? (((a_taps[h][(e) < PA ? (e) : 0] >> tap_) & 1u) ? a_voff[h][(e) < PA ? (e) : 0] : XL_OOB)       \
                                : a_voff[h][(e) < PA ? (e) : 0];                                                                  \
            if ((a_oob && tap_ != 0) || (a_oob_kx && (tap_ % 3) != 0)) vo_ = XL_OOB;                                              \
            xl_glds(rsA, lds0 + ((T) & 1) * BUF + a_lds[h][(e) < PA ? (e) : 0], vo_, so_);                                        \
        }                                                                                                                         \
    }
#define XL_PIECE_B(part, e, T)                                                                                                    \
    if constexpr ((e) < ((part) ? PB1 : PB0)) {                                                                                   \
        if ((T) < nt && do_dma)                                                                                                   \
            xl_glds_b(rsB, lds0 + ((T) & 1) * B_STRIDE + B_BASE + b_lds[part][(e) < UNIT_MAX ? (e) : 0],                            \
                      b_voff[part][(e) < UNIT_MAX ? (e) : 0], b_soff((T)));                                                       \
    }
#define XL_ISSUE_A(h, T) { XL_PIECE_A(h, 0, T) XL_PIECE_A(h, 1, T) }
    // SCHED 4: unit h of the shared A slab of GROUP G = (channel block, ky) into A buffer G & 1; xs_ky / xs_so = ky and scalar offset of that group
#define XS_PIECE_A(h, e, G)                                                                                                       \
    if ((G) < ng && do_dma) {                                                                                                     \
        unsigned va_ = xs_vb;                                                                                                     \
        asm volatile("" : "+v"(va_));                                                                                             \
        const unsigned vo_ = va_ + (unsigned)((a_lds[h][e] >> 7) * row_bytes);                                                    \
        xl_glds(rsA, lds0 + ((G) & 1) * AS_BYTES + a_lds[h][e], ((xs_bits >> (3 * (2 * (h) + (e)) + xs_ky)) & 1u) ? vo_ : XL_OOB, xs_so); \
    }
    // B pieces of SCHED 4 from ONE per-lane register too (N % 320 == 0: no column tail): piece rows start at multiples of 8, so only the parity of
    // row0 / 8 enters the (row >> 1) & 7 swizzle — bit 2 of the chunk = bit 6 of the byte offset (ldw * 2 is a multiple of 128)
#define XS_PIECE_B(part, e, T)                                                                                                    \
    if constexpr ((e) < ((part) ? PB1 : PB0)) {                                                                                   \
        if ((T) < nt && do_dma) {                                                                                                 \
            const int r0_ = (b_lds[part][(e) < UNIT_MAX ? (e) : 0] - A_BYTES) >> 7;                                               \
            unsigned vb_ = xb_vb;                                                                                                 \
            asm volatile("" : "+v"(vb_));   /* formed HERE: as loop invariants the five offsets are hoisted and then spilled (scratch reloads + vmcnt(0) in the loop) */ \
            const unsigned vo_ = (vb_ ^ (unsigned)(((r0_ >> 3) & 1) << 6)) + (unsigned)(r0_ * (int)(p.ldw * 2));                  \
            xl_glds_b(rsB, lds0 + ((T) & 1) * B_STRIDE + B_BASE + b_lds[part][(e) < UNIT_MAX ? (e) : 0], vo_, b_soff((T)));        \
        }                                                                                                                         \
    }
#define XS_ISSUE_B(part, T) { XS_PIECE_B(part, 0, T) XS_PIECE_B(part, 1, T) XS_PIECE_B(part, 2, T) }
#define XS_ISSUE_A0(G) { XS_PIECE_A(0, 0, G) XS_PIECE_A(0, 1, G) }
#define XS_ISSUE_A1X(G)                                                                                                           \
    {                                                                                                                             \
        XS_PIECE_A(1, 0, G) XS_PIECE_A(1, 1, G)                                                                                   \
        unsigned va2_ = xs_vb;                                                                                                    \
        asm volatile("" : "+v"(va2_));                                                                                            \
        if ((G) < ng && do_dma)                                                                                                   \
            xl_glds(rsA, lds0 + ((G) & 1) * AS_BYTES + (wave == 7 ? 256 * 128 : ZROW_OFF),                                        \
                    ((xs_bits >> (12 + xs_ky)) & 1u) ? va2_ + (unsigned)(256 * row_bytes) : XL_OOB, xs_so);                       \
    }
#define XL_ISSUE_B(part, T) { XL_PIECE_B(part, 0, T) XL_PIECE_B(part, 1, T) XL_PIECE_B(part, 2, T) }

    // ---- fragment read offsets (bytes inside a buffer) ----
    const int fo0 = frag_off(0, lane, 0), fo1 = frag_off(0, lane, 1);     // row part = (lane & 15) * 128: tile row blocks add multiples of 2048
    const int a_rd = a_tile_row0<BN>(wm, 0) * 128;
    const int b_rd = A_BYTES + b_tile_row0<BN>(wn, 0) * 128;
    if constexpr (KXS) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int m_ = m0 + a_tile_row0<BN>(wm, i) + (lane & 15);
            const int ox_ = m_ % Wo;                              // (rows past M: their outputs are never stored)
            if (ox_ == 0) xs_bits |= 1u << (16 + i);
            if (ox_ == Wo - 1) xs_bits |= 1u << (24 + i);
        }
    }

    f32x4_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    constexpr int NBF = (BN == 320) ? TJ0 : TJ;                   // BN = 320: B0 (3 tiles) and B1 (2 tiles) share one register set
    Frag8 af[TIH][2], bfr[NBF][2];                                // one A half + the B tiles of the wave, both k32 steps
#define XL_READ_A(h, buf_)                                                                                                        \
    {                                                                                                                             \
        const unsigned char* s_ = smem + (buf_) * BUF + a_rd + (h) * TIH * 2048;                                                  \
        _Pragma("unroll") for (int i = 0; i < TIH; ++i) {                                                                         \
            af[i][0].u = *(const uint4*)(s_ + i * 2048 + fo0);                                                                    \
            af[i][1].u = *(const uint4*)(s_ + i * 2048 + fo1);                                                                    \
        }                                                                                                                         \
    }
    // SCHED 4: A half h of the shared slab in A buffer ab_, tap kx_ (rows shifted by kx_); lanes whose output pixel sits on the left (kx 0) / right
    // (kx 2) image border read the zero row instead (xs_lb / xs_rb: bit i = MFMA row tile i of this lane is on that border)
#define XS_READ_A(h, ab_)                                                                                                         \
    {                                                                                                                             \
        const unsigned char* s_ = smem + (ab_) * AS_BYTES;                                                                        \
        unsigned bb_ = xs_bsel;                                                                                                   \
        asm volatile("" : "+v"(bb_));   /* the selects are formed HERE: hoisted out of the loop they are 48 live values (seen as scratch reloads inside the slab loop) */ \
        const int f0_ = xs_f0, f1_ = xs_f0 ^ 64;                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TIH; ++i) {                                                                         \
            const int it_ = (h) * TIH + i;                                                                                        \
            const bool z_ = (bb_ >> it_) & 1u;                                                                                    \
            const int o0_ = z_ ? ZROW_OFF - it_ * 2048 : f0_, o1_ = z_ ? ZROW_OFF - it_ * 2048 : f1_;                             \
            af[i][0].u = *(const uint4*)(s_ + o0_ + it_ * 2048);                                                                  \
            af[i][1].u = *(const uint4*)(s_ + o1_ + it_ * 2048);                                                                  \
        }                                                                                                                         \
    }
#define XL_READ_B(J0, NJ, buf_)                                                                                                   \
    {                                                                                                                             \
        const unsigned char* s_ = smem + (buf_) * B_STRIDE + B_BASE + b_rd;                                                       \
        _Pragma("unroll") for (int j = (J0); j < (J0) + (NJ); ++j) {                                                              \
            bfr[j][0].u = *(const uint4*)(s_ + j * 2048 + fo0);                                                                   \
            bfr[j][1].u = *(const uint4*)(s_ + j * 2048 + fo1);                                                                   \
        }                                                                                                                         \
    }
#define XL_READ_B_TO(D0, J0, NJ, buf_)                                                                                            \
    {                                                                                                                             \
        const unsigned char* s_ = smem + (buf_) * B_STRIDE + B_BASE + b_rd;                                                       \
        _Pragma("unroll") for (int j = 0; j < (NJ); ++j) {                                                                        \
            bfr[(D0) + j][0].u = *(const uint4*)(s_ + ((J0) + j) * 2048 + fo0);                                                   \
            bfr[(D0) + j][1].u = *(const uint4*)(s_ + ((J0) + j) * 2048 + fo1);                                                   \
        }                                                                                                                         \
    }
#define XL_MMA_TO(h, D0, J0, NJ)                                                                                                  \
    if (do_mma) {                                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                          \
            _Pragma("unroll") for (int i = 0; i < TIH; ++i)                                                                       \
                _Pragma("unroll") for (int j = 0; j < (NJ); ++j)                                                                  \
                    acc[(h) * TIH + i][(J0) + j] =                                                                                \
                        MDX_MFMA_16x16x32(bfr[(D0) + j][kk].v, af[i][kk].v, acc[(h) * TIH + i][(J0) + j]); \
        __builtin_amdgcn_s_setprio(0);                                                                                            \
    }
    // D = Wfrag x Afrag: the accumulator holds 4 consecutive n (rows of D) of one m (column of D) per lane
#define XL_MMA(h, J0, NJ)                                                                                                         \
    if (do_mma) {                                                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                          \
            _Pragma("unroll") for (int i = 0; i < TIH; ++i)                                                                       \
                _Pragma("unroll") for (int j = (J0); j < (J0) + (NJ); ++j)                                                        \
                    acc[(h) * TIH + i][j] =                                                                                       \
                        MDX_MFMA_16x16x32(bfr[j][kk].v, af[i][kk].v, acc[(h) * TIH + i][j]);       \
        __builtin_amdgcn_s_setprio(0);                                                                                            \
    }
    // MFMA segment of the two-phase schedule: one A half x every B tile, with up to four DMA pieces interleaved between the MFMAs
    // (pinned by sched_barrier: an LDS-DMA issued among MFMAs costs ~60 cycles of this wave's issue slot, hidden under the matrix
    // pipe's 16 cycles per MFMA; in the load segment the same instruction costs 100-185 and sits on the critical path).
#define XL_MSEG(h, I0, I1, I2, I3)                                                                                                \
    {                                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                            \
        constexpr int P_ = 2 * TIH;                                                                                               \
        _Pragma("unroll") for (int pr = 0; pr < P_; ++pr) {                                                                       \
            const int kk = pr / TIH, i = pr % TIH;                                                                                \
            if (pr == 1) { I0 }                                                                                                   \
            if (pr == 1 + (P_ - 1) / 4) { I1 }                                                                                    \
            if (pr == 1 + 2 * (P_ - 1) / 4) { I2 }                                                                                \
            if (pr == 1 + 3 * (P_ - 1) / 4) { I3 }                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
            if (do_mma) {                                                                                                         \
                _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                                    \
                    acc[(h) * TIH + i][j] =                                                                                       \
                        MDX_MFMA_16x16x32(bfr[j][kk].v, af[i][kk].v, acc[(h) * TIH + i][j]);       \
            }                                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
        }                                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                                            \
    }
    // end of a load segment: this wave's fragment reads have RETURNED before the barrier (so a unit may be refilled by anyone one
    // phase later), then the barrier that hands the matrix pipe over
#define XL_SEG_END()                                                                                                              \
    {                                                                                                                             \
        xl_wait_lgkm0();                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    }
#define XL_MMA_END()                                                                                                              \
    {                                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
    }

    // Per-column addends (bias + the temb rows of the images this tile touches): fetched NOW into registers, written to their LDS rows
    // (outside the operand ring) in the epilogue — fetched there, their memory round trip sat between the main loop and the staging of
    // the accumulators.  Not for the 320-wide tile (no registers to spare).
    constexpr bool ADD_EARLY = BN != 320;
    constexpr int ADD_N = ADD_EARLY ? (BN + NTH - 1) / NTH : 1;      // one addend row (bias) over the workgroup's threads
    float addb[ADD_N];                                           // raw loads: NOT touched before the epilogue (a use here would wait for them)
    const bool geglu_e = p.epi == 1;
    const bool has_t_e = p.temb != nullptr && !geglu_e;
    // (only the bias: the temb rows need the step selector first, a dependent round trip that would delay the first slab — tiles with
    // temb rows fetch their addends in the epilogue as before.)  Issued BEFORE the prologue's DMA pieces so that the hand-counted
    // vmcnt waits of the main loop, which count the pieces younger than the one waited for, are unaffected.
    const bool add_early = ADD_EARLY && !has_t_e;
    if (add_early) {
#pragma unroll
        for (int u = 0; u < ADD_N; ++u) addb[u] = 0.f;
        if (p.bias) {
#pragma unroll
            for (int u = 0; u < ADD_N; ++u) {
                const int idx = u * NTH + tid;
                addb[u] = p.bias[min(n0 + (idx < BN ? idx : 0), p.N - 1)];
            }
        }
    }
    XL_STAMP(1)
    if constexpr (KXS) {
        // ===== schedule 4: the 256 x 320 quadrant order with ONE A slab per (channel block, ky) shared by its three kx slabs ==================
        //   group g = (cb, ky) = slabs T = 3 g + kx; A buffer g & 1, B buffer T & 1; per slab the quadrants of the 320-wide schedule:
        //   reads:   q0 A0,B0   q1 A1   q2 B1   q3 A0
        //   issues:  q0 A0(g+1) [kx 0]   q1 B0(T+2)   q2 A1(g+1) + the halo piece [kx 1]   q3 B1(T+2)
        //   A buffer (g+1) & 1 was last read by group g - 1; B0 / B1 of buffer T & 1 were read in q0 / q2 of this slab (as in the per-slab schedule).
        //   wait in q3(T) for everything up to B1(T+1) (issued in q3(T-1)): the pieces issued during slab T follow it:
        //     kx 0: PA + PB0 + PB1   kx 1: PB0 + PA + 1 + PB1   kx 2: PB0 + PB1 (the next group's A, issued in the two slabs before, is older)
        //   last group: no A is issued, and its last two slabs issue no B either (as the per-slab schedule's tail)
        const int ng = nt / 3;                                       // nt = 9 Cin / 64
        int xs_ky = 0, xs_so = 0;                                    // ky / scalar offset of the group being ISSUED
        auto xs_group = [&](int g_) { const int cb_ = g_ / 3; xs_ky = g_ - cb_ * 3; xs_so = xs_ky * p.Wi * row_bytes + cb_ * 128; };
        xs_group(0);
        XS_ISSUE_A0(0)
        XS_ISSUE_A1X(0)
        XS_ISSUE_B(0, 0)
        XS_ISSUE_B(1, 0)
        XS_ISSUE_B(0, 1)
        XS_ISSUE_B(1, 1)
        xl_wait_vmcnt<PB0 + PB1>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        XL_STAMP(2)
        if (grp == 1) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ONE loop over the slabs (kx, the group and its A buffer are scalar state): as two nested loops the accumulators were loop-carried through
        // both headers and the allocator copied all 160 of them every slab (78 v_mov_b64 per iteration in the .s)
        int kx = 0, g = 0, ab = 0;
        bool more = ng > 1;
        xs_group(1);
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            // fragment offset (k32 step 0; step 1 = ^ 64) of row (lane & 15) + kx — tile row blocks are multiples of 16, they do not change
            // row & 7 — and the border bits of this tap: left border for kx 0, right border for kx 2, none for the centre tap
            const int r_ = (lane & 15) + kx;
            const int xs_f0 = a_rd + r_ * 128 + (((lane >> 4) ^ (r_ & 7)) << 4);
            const unsigned xs_bsel = kx == 0 ? (xs_bits >> 16) & 0xffu : kx == 2 ? xs_bits >> 24 : 0u;
            XS_READ_A(0, ab)
            XL_READ_B_TO(0, 0, TJ0, buf)
            if (kx == 0) XS_ISSUE_A0(g + 1)
            XL_SEG_END()
            XL_MMA_TO(0, 0, 0, TJ0)
            XL_MMA_END()
            XS_READ_A(1, ab)
            XS_ISSUE_B(0, t + 2)
            XL_SEG_END()
            XL_MMA_TO(1, 0, 0, TJ0)
            XL_MMA_END()
            XL_READ_B_TO(0, TJ0, TJ1, buf)
            if (kx == 1) XS_ISSUE_A1X(g + 1)
            XL_SEG_END()
            XL_MMA_TO(1, 0, TJ0, TJ1)
            XL_MMA_END()
            XS_READ_A(0, ab)
            XS_ISSUE_B(1, t + 2)
            if (kx == 0) { if (more) xl_wait_vmcnt<PA + PB0 + PB1>(); else xl_wait_vmcnt<PB0 + PB1>(); }
            else if (kx == 1) { if (more) xl_wait_vmcnt<PB0 + PA + 1 + PB1>(); else xl_wait_vmcnt<0>(); }
            else { if (more) xl_wait_vmcnt<PB0 + PB1>(); else xl_wait_vmcnt<0>(); }
            XL_SEG_END()
            XL_MMA_TO(0, 0, TJ0, TJ1)
            XL_MMA_END()
            if (++kx == 3) { kx = 0; ++g; ab ^= 1; more = g + 1 < ng; xs_group(g + 1); }
        }
    } else
    if constexpr (BN == 320) {
        // ===== 256 x 320: quadrant order (A0,B0) (A1,B0) (A1,B1) (A0,B1) so that B0 and B1 are never live together (160 accumulators
        // leave room for one A half + 3 B tiles); A0 is read twice per slab (34 instead of 26 fragment reads: LDS port time is not the
        // bound, the global -> LDS path is — this tile moves 0.0070 operand bytes per MAC against 0.0078 / 0.0102 for 256 / 160 wide).
        //   reads:   q0 A0,B0   q1 A1   q2 B1   q3 A0            last read of a unit: B0 q0, A1 q1, B1 q2, A0 q3
        //   refills: q1 B0(t+2) q2 A1(t+2) q3 B1(t+2) q0' A0(t+2)  (each one phase after the last read, lgkmcnt(0) before the barrier)
        //   wait in q3(t) for A0(t+1) (issued in q0(t)): followed by B0(t+2), A1(t+2), B1(t+2) -> vmcnt(PB0 + PA + PB1)
        constexpr int INFLIGHT = PB0 + PA + PB1;
        XL_ISSUE_A(0, 0)
        XL_ISSUE_B(0, 0)
        XL_ISSUE_A(1, 0)
        XL_ISSUE_B(1, 0)
        XL_ISSUE_B(0, 1)
        XL_ISSUE_A(1, 1)
        XL_ISSUE_B(1, 1)
        if (nt > 1) xl_wait_vmcnt<INFLIGHT>(); else xl_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        XL_STAMP(2)
        if (grp == 1) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            XL_READ_A(0, buf)
            XL_READ_B_TO(0, 0, TJ0, buf)
            XL_ISSUE_A(0, t + 1)
            XL_SEG_END()
            XL_MMA_TO(0, 0, 0, TJ0)
            XL_MMA_END()
            XL_READ_A(1, buf)
            XL_ISSUE_B(0, t + 2)
            XL_SEG_END()
            XL_MMA_TO(1, 0, 0, TJ0)
            XL_MMA_END()
            XL_READ_B_TO(0, TJ0, TJ1, buf)
            XL_ISSUE_A(1, t + 2)
            XL_SEG_END()
            XL_MMA_TO(1, 0, TJ0, TJ1)
            XL_MMA_END()
            XL_READ_A(0, buf)
            XL_ISSUE_B(1, t + 2)
            if (t + 2 < nt) xl_wait_vmcnt<INFLIGHT>(); else xl_wait_vmcnt<0>();
            XL_SEG_END()
            XL_MMA_TO(0, 0, TJ0, TJ1)
            XL_MMA_END()
        }
    } else if constexpr (SCHED == 2 || SCHED == 3) {
        // ===== schedule 2: schedule 0's quadrants with every refill >= 2 phases after the slot's last read, so the fragment reads'
        // lgkmcnt(0) can sit AFTER the barrier (the load segment ends when the reads are ISSUED; their latency overlaps the barrier).
        // schedule 3: same refill order, lgkmcnt(0) before the barrier, DMA issued after it (no contention with the wave's own reads).
        //   issue order per wave: A1(t+1)@q0  B0(t+1)@q1  A0(t+2)@q2  B1(t+2)@q3 ; reads: q0 A0,B0  q1 B1  q2 A1
        //   wait in q3(t) for B0(t+1) (and everything older): followed by A0(t+2), B1(t+2) -> vmcnt(PA + PB1)
        XL_ISSUE_A(0, 0)
        XL_ISSUE_B(1, 0)
        XL_ISSUE_A(1, 0)
        XL_ISSUE_B(0, 0)
        XL_ISSUE_A(0, 1)
        XL_ISSUE_B(1, 1)
        if (nt > 1) xl_wait_vmcnt<PA + PB1>(); else xl_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#define XL_SEG2(ISSUE)                                                                                                            \
        if constexpr (SCHED == 2) {                                                                                               \
            ISSUE                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
            __builtin_amdgcn_s_barrier();                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
        } else {                                                                                                                  \
            xl_wait_lgkm0();                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
            ISSUE                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
            __builtin_amdgcn_s_barrier();                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
        }
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            XL_READ_A(0, buf)
            XL_READ_B(0, TJ0, buf)
            XL_SEG2(XL_ISSUE_A(1, t + 1))
            XL_MMA(0, 0, TJ0)
            XL_MMA_END()
            XL_READ_B(TJ0, TJ1, buf)
            XL_SEG2(XL_ISSUE_B(0, t + 1))
            XL_MMA(0, TJ0, TJ1)
            XL_MMA_END()
            XL_READ_A(1, buf)
            XL_SEG2(XL_ISSUE_A(0, t + 2))
            XL_MMA(1, TJ0, TJ1)
            XL_MMA_END()
            XL_SEG2(XL_ISSUE_B(1, t + 2) if (t + 2 < nt) xl_wait_vmcnt<PA + PB1>(); else xl_wait_vmcnt<0>();)
            XL_MMA(1, 0, TJ0)
            XL_MMA_END()
        }
#undef XL_SEG2
    } else     if constexpr (SCHED == 0) {
        // ===== schedule 0: four quadrant phases per slab, one unit refilled per load segment ==========================================
        // units A0, B0, B1, A1 (the rows the quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0) read first); a unit's slot is refilled one
        // phase after its last fragment read, for the slab two ahead; ONE counted wait per slab.
        constexpr int INFLIGHT = PA + PB0 + PB1;                     // A0, B0, B1 of the next slab follow a slab's last unit (A1)
        XL_ISSUE_A(0, 0)
        XL_ISSUE_B(0, 0)
        XL_ISSUE_B(1, 0)
        XL_ISSUE_A(1, 0)
        XL_ISSUE_A(0, 1)
        XL_ISSUE_B(0, 1)
        XL_ISSUE_B(1, 1)
        if (nt > 1) xl_wait_vmcnt<INFLIGHT>(); else xl_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        XL_STAMP(2)
        if (grp == 1) {                                              // the stagger: group 1 runs one barrier behind group 0
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            // q0: quadrant (A0, B0).  Reads A0, B0; refills A1 of slab t + 1 (last read in q2 of slab t - 1).
            XL_READ_A(0, buf)
            XL_READ_B(0, TJ0, buf)
            XL_ISSUE_A(1, t + 1)
            XL_SEG_END()
            XL_MMA(0, 0, TJ0)
            XL_MMA_END()
            // q1: (A0, B1).  Reads B1; refills A0 of slab t + 2 (last read in q0).
            XL_READ_B(TJ0, TJ1, buf)
            XL_ISSUE_A(0, t + 2)
            XL_SEG_END()
            XL_MMA(0, TJ0, TJ1)
            XL_MMA_END()
            // q2: (A1, B1).  Reads A1; refills B0 of slab t + 2 (last read in q0).
            XL_READ_A(1, buf)
            XL_ISSUE_B(0, t + 2)
            XL_SEG_END()
            XL_MMA(1, TJ0, TJ1)
            XL_MMA_END()
            // q3: (A1, B0), B0 still in registers.  Refills B1 of slab t + 2 (last read in q1); slab t + 1 must have landed: its last
            // unit (A1, issued in q0) is followed by exactly the three units A0, B0, B1 of slab t + 2 when that slab exists.
            XL_ISSUE_B(1, t + 2)
            if (t + 2 < nt) xl_wait_vmcnt<INFLIGHT>(); else xl_wait_vmcnt<0>();
            XL_SEG_END()
            XL_MMA(1, 0, TJ0)
            XL_MMA_END()
        }
    } else {
        // ===== schedule 1: two phases per slab (A half x all B tiles), DMA issued from INSIDE the MFMA segments ======================
        // Round-2 ablation of schedule 0 (profiles/README.md): per slab and wave group the serial path is L + M with L = 24 fragment
        // reads (~380 cycles) + 8 DMA issues (~100 each in a load segment) + 4 exposed LDS latencies vs M = 1024 cycles of MFMA
        // issue, i.e. ~2.6 k + 8 barriers against the matrix pipe's 2.05 k per slab.  Here L carries only the fragment reads and two
        // latencies, the DMA instructions ride between the MFMAs (whose wave has 12 idle issue cycles per MFMA), and there are
        // 4 barriers per slab.
        //   P0(t): L: read A0, B(all) of slab t; wait for A1(t)          M: A0 x B, issuing B1(t+1), A1(t+1) into the other buffer
        //   P1(t): L: read A1 of slab t; wait for A0, B0, B1 of t+1      M: A1 x B, issuing A0(t+2), B0(t+2) into this buffer
        // Issue order per wave: ... A1(t) | A0(t+1) B0(t+1) | B1(t+1) A1(t+1) | A0(t+2) B0(t+2) ...
        //   wait in P0(t): A1(t) is followed by A0(t+1), B0(t+1)          -> vmcnt(PA + PB0)   (0 past the end)
        //   wait in P1(t): B1(t+1) is followed by A1(t+1)                 -> vmcnt(PA)
        // Slot reuse: B1/A1 of the other buffer were last read in P0(t-1) / P1(t-1); A0/B0 of this buffer in P0(t) — every refill
        // is issued at least one full barrier interval after both groups' reads of the slot returned (lgkmcnt(0) before the barrier).
        XL_ISSUE_A(0, 0)
        XL_ISSUE_B(0, 0)
        XL_ISSUE_B(1, 0)
        XL_ISSUE_A(1, 0)
        XL_ISSUE_A(0, 1)
        XL_ISSUE_B(0, 1)
        if (nt > 1) xl_wait_vmcnt<PA + PB0>(); else xl_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        XL_STAMP(2)
        if (grp == 1) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            XL_READ_A(0, buf)
            XL_READ_B(0, TJ, buf)
            if (t + 1 < nt) xl_wait_vmcnt<PA + PB0>(); else xl_wait_vmcnt<0>();
            XL_SEG_END()
            XL_MSEG(0, XL_PIECE_B(1, 0, t + 1), XL_PIECE_B(1, 1, t + 1), XL_PIECE_A(1, 0, t + 1), XL_PIECE_A(1, 1, t + 1))
            XL_MMA_END()
            XL_READ_A(1, buf)
            if (t + 1 < nt) xl_wait_vmcnt<PA>();
            XL_SEG_END()
            XL_MSEG(1, XL_PIECE_A(0, 0, t + 2), XL_PIECE_A(0, 1, t + 2), XL_PIECE_B(0, 0, t + 2), XL_PIECE_B(0, 1, t + 2))
            XL_MMA_END()
        }
    }
    if (grp == 0) {                                              // balance the stagger
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
#undef XL_PIECE_A
#undef XS_PIECE_A
#undef XS_PIECE_B
#undef XS_ISSUE_B
#undef XS_ISSUE_A0
#undef XS_ISSUE_A1X
#undef XS_READ_A
#undef XL_PIECE_B
#undef XL_ISSUE_A
#undef XL_ISSUE_B
#undef XL_READ_A
#undef XL_READ_B
#undef XL_READ_B_TO
#undef XL_MMA_TO
#undef XL_MMA
#undef XL_MSEG
#undef XL_SEG_END
#undef XL_MMA_END

    XL_STAMP(3)
#ifdef MDX_XL_ABLATE
    if (p.dbg & 4) return;                                       // ablation: prologue + main loop only (no epilogue at all)
#endif
    // ---- epilogue ----
    constexpr int NH = (BN == 320) ? 2 : 1;                       // the 320-wide bf16 tile goes through LDS in two 128-row halves
    constexpr int HROWS = BM / NH;
    const bool geglu = p.epi == 1;
    const bool has_t = p.temb != nullptr && !geglu;
    const int BNo = geglu ? BN / 2 : BN;
    const int CSTR = BNo + 8;                                    // LDS row stride of the bf16 tile (elements); rows stay 16-byte aligned
    bf16_t* Cs = (bf16_t*)smem;                                  // [HROWS][CSTR], aliases the (dead) operand ring
    float* addend = (float*)(smem + (size_t)HROWS * (BN + 8) * 2);   // [XL_SLOTS][BN] fp32, behind the largest C staging tile
    const int b0 = has_t ? m0 / p.rows_per_b : 0;
    if (add_early) {
#pragma unroll
        for (int u = 0; u < ADD_N; ++u) {
            const int idx = u * NTH + tid;
            if (idx < BN) addend[idx] = (n0 + idx < p.N) ? addb[u] : 0.f;
        }
    } else {
        const int nslots = has_t ? XL_SLOTS : 1;
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
    const int n0o = geglu ? n0 / 2 : n0, Nout = geglu ? p.N / 2 : p.N;
    const bf16_t* Rg = p.R ? (const bf16_t*)p.R : nullptr;
    bf16_t* Cg = (bf16_t*)p.C;
    // The residual of a (half) tile is fetched in ONE batch of unconditional loads from clamped addresses (a guarded load becomes its
    // own branch + wait; in batches of four behind the staging it cost four to six dependent HBM round trips per tile).  256- / 160-wide:
    // before the accumulators are staged (the round trip overlaps the staging and the two barriers).  320-wide (160 accumulators live,
    // two 128-row halves): right behind each half's staging barrier.
    // Row walk of the residual / store phase: thread t owns the 16-byte column chunk t % CPR of rows t / CPR + u * RPP — one base address
    // and a wave-uniform row stride per pass (the linear index / CPR of round 2 was an integer division and a 64-bit address per chunk:
    // at CPR = 40 that alone pushed the 320-wide kernel into scratch).  Threads >= RPP * CPR idle (32 of 512 at 320 columns).
    constexpr int CPR = BN >> 3;
    constexpr int RPP = NTH / CPR;
    constexpr int RIT = (HROWS + RPP - 1) / RPP;                  // passes: 16 (256-wide), 11 (320-wide half, 160-wide)
    const int rrow0 = tid / CPR, rc8 = (tid - rrow0 * CPR) * 8;
    const bool ractive = tid < RPP * CPR && n0 + rc8 < p.N;
#ifdef MDX_XL_R2_RESIDUAL                                       // A/B side build: the 320-wide tile keeps round 2's batches of four
    const bool rpref = NH == 1 && Rg != nullptr && p.wide && !geglu;
#else
    const bool rpref = Rg != nullptr && p.wide && !geglu;
#endif
    uint4 rpre[RIT];
    auto fetch_residual = [&](int hh) {
        const int mh_ = m0 + hh * HROWS;
        const int c8 = (n0 + rc8 + 8 > p.N) ? (p.N - 8 - n0 < 0 ? 0 : p.N - 8 - n0) : rc8;
        const bf16_t* rb = Rg + n0 + c8;
        // rows past M read row M - 1: a 320-wide edge tile whose SECOND half starts at or past M has rmax < 0, and mh_ + rmax is still
        // M - 1 (round 5: a clamp of `row` to 0 here read up to 128 rows past the end of R — a memory fault when R closes its allocation)
        const int rmax = min(HROWS - 1, p.M - 1 - mh_);
#pragma unroll
        for (int u = 0; u < RIT; ++u) {
            const int row = min(rrow0 + u * RPP, rmax);
            rpre[u] = *(const uint4*)(rb + (long)(mh_ + row) * p.ldr);
        }
    };
#pragma unroll 1
    for (int hh = 0; hh < NH; ++hh) {
        if (rpref && NH == 1) fetch_residual(hh);
        // raw barrier + LDS wait only: __syncthreads() also drains the VM counter, i.e. it would sit out the residual fetch just issued
        xl_wait_lgkm0();
        __builtin_amdgcn_s_barrier();                            // ring dead / previous half stored (its LDS reads returned); addend visible
        asm volatile("" ::: "memory");
        XL_STAMP(6)
        {
            // The epilogue kind (plain / GEGLU / SiLU) and "addend rows per image" are wave-uniform RUNTIME facts: tested per element
            // (round 2) the compiler kept them as scalar branches inside the unrolled loops — ~300 branches and a full IEEE division per
            // SiLU in a 128-accumulator staging pass that took 4.7 us of a 25 us K = 640 tile (s_memtime stamps, profiles/README.md
            // round 3).  The body is instantiated per (kind, addend mode) and the choice is made ONCE per tile.
            const int fr = lane & 15, fq = lane >> 4;
            auto stage = [&](auto epi_c, auto hast_c) {
                constexpr int EPI = decltype(epi_c)::value;
                constexpr bool HAS_T = decltype(hast_c)::value;
                constexpr bool HOIST = BN != 320 && !HAS_T;       // column-only addends: read the wave's vectors once (the 320-wide kernel has no registers to spare)
                float4 a4h[HOIST ? TJ : 1], g4h[HOIST ? TJ : 1];
                if constexpr (HOIST) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const int nl = wn * TJ * 16 + j * 16 + 4 * fq;
                        a4h[j] = *(const float4*)(addend + nl);
                        if constexpr (EPI == 1) g4h[j] = *(const float4*)(addend + ((nl + 32) < BN ? nl + 32 : nl));
                    }
                }
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int mt_ = wm * TI * 16 + i * 16;        // first row of this MFMA tile inside the block tile
                    if (NH > 1 && mt_ / HROWS != hh) continue;    // wave-uniform
                    const int ml = mt_ + fr;
                    int slot = 0;
                    if constexpr (HAS_T) slot = min(min(m0 + ml, p.M - 1) / p.rows_per_b - b0, XL_SLOTS - 1);
                    const float* ad = addend + slot * BN;
                    bf16_t* crow = Cs + (ml - hh * HROWS) * CSTR;
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        if (EPI == 1 && (j & 2)) continue;        // gate tiles (columns 32..63 of a 64 group) are consumed with their value tile
                        const int nl = wn * TJ * 16 + j * 16 + 4 * fq;   // raw column inside the tile
                        float4 a4;
                        if constexpr (HOIST) a4 = a4h[j]; else a4 = *(const float4*)(ad + nl);
                        const float bb[4] = {a4.x, a4.y, a4.z, a4.w};
                        float o[4];
                        if constexpr (EPI == 1) {
                            float4 g4;
                            if constexpr (HOIST) g4 = g4h[j]; else g4 = *(const float4*)(ad + nl + 32);
                            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {         // pairs: the GELU runs on packed fp32 (common.h: gelu_erf_f2)
                                const float x0 = acc[i][j][e] + bb[e], x1 = acc[i][j][e + 1] + bb[e + 1];
                                const f32x2_t gt = {acc[i][(TJ == 4) ? (j | 2) : j][e] + gg[e], acc[i][(TJ == 4) ? (j | 2) : j][e + 1] + gg[e + 1]};
                                const f32x2_t ge = gelu_erf_f2(gt);
                                o[e] = x0 * ge.x; o[e + 1] = x1 * ge.y;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = acc[i][j][e] + bb[e];
                                if constexpr (EPI == 2) x = silu_f(x);
                                o[e] = x;
                            }
                        }
                        const int cl = EPI == 1 ? ((nl >> 6) * 32 + (nl & 31)) : nl;
                        uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
                        *(uint2*)(crow + cl) = ov;
                    }
                }
            };
            using std::integral_constant;
            if (geglu) {
                if constexpr (BN == 256) stage(integral_constant<int, 1>{}, integral_constant<bool, false>{});   // (xl_supported: GEGLU only on the 256-wide tile)
            } else {                                              // (SiLU epilogues never reach this kernel: xl_supported)
                if (has_t) stage(integral_constant<int, 0>{}, integral_constant<bool, true>{});
                else stage(integral_constant<int, 0>{}, integral_constant<bool, false>{});
            }
        }
        XL_STAMP(7)
        xl_wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        XL_STAMP(4)
#ifdef MDX_XL_ABLATE
        if (p.dbg & 8) continue;                                 // ablation: accumulators staged, nothing stored
#endif
        const int mh = m0 + hh * HROWS;
        if (p.col_split) {
            // batch-flattened output: a tile's columns are tokens of consecutive images; column n -> image n / cs, token n % cs.
            // 4-byte stores of token pairs when cs is even (pairs never straddle an image, addresses stay 4-byte aligned), single
            // elements otherwise; a wave instruction still covers contiguous row segments.  No bias / residual in this mode.
            const int cs = p.col_split;
            const bool pairs = (cs & 1) == 0;
            const int cpr = BNo >> 1;
            const int total = HROWS * cpr;
            const int bfirst = n0 / cs;
            const int tfirst = n0 - bfirst * cs;
            bool fast_done = false;
            // compile-time part first: instantiations whose width does not divide the workgroup never carry the fast body; the
            // run-time !geglu keeps the branch honest (GEGLU halves the output width: cpr below is BNo >> 1, RPP2 assumes BN >> 1 — xl_supported
            // keeps GEGLU out of col_split anyway)
            if constexpr (NTH % (BN >> 1) == 0) if (!geglu) {
                fast_done = true;
                // (round 4) a thread owns ONE column pair for all its rows: image / token / validity of the pair are resolved once (the
                // generic walk below divides and searches per element pair: ~3 k scalar-ish instructions per thread and tile beside a 16 us
                // main loop at K = 640 — the level-1/2 V^T projections ran at 520-690 TFLOP/s, their q/k siblings at 970-1130), then
                // batches of eight LDS reads followed by eight stores.
                constexpr int RPP2 = NTH / (BN >> 1);                    // rows per pass (4 at 256 columns)
                constexpr int UB = 8;
                static_assert((HROWS / RPP2) % UB == 0, "col_split fast epilogue: the row walk has no remainder handling");
                const int c2 = (tid % cpr) * 2, r0 = tid / cpr;
                int b = bfirst, t = tfirst + c2;
                while (t >= cs) { t -= cs; ++b; }
                const bool ok0 = n0 + c2 < p.N, ok1 = n0 + c2 + 1 < p.N;
                const bool split = !pairs && t + 1 >= cs;                // odd cs: the pair's second token is token 0 of the next image
                bf16_t* d0 = Cg + (long)b * p.sC + (long)(mh + r0) * p.ldc + t;
                bf16_t* d1 = split ? Cg + (long)(b + 1) * p.sC + (long)(mh + r0) * p.ldc : d0 + 1;
                const bf16_t* src = Cs + r0 * CSTR + c2;
                const int rmax = min(HROWS, p.M - mh);                   // rows of this (half) tile that exist
#pragma unroll 1
                for (int u0 = 0; u0 < HROWS / RPP2; u0 += UB) {
                    unsigned v[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) v[u] = *(const unsigned*)(src + (long)(u0 + u) * RPP2 * CSTR);
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int row = r0 + (u0 + u) * RPP2;
                        if (row >= rmax || !ok0) continue;
                        const long ro = (long)(u0 + u) * RPP2 * p.ldc;
                        if (pairs && ok1) {
                            *(unsigned*)(d0 + ro) = v[u];
                        } else {
                            d0[ro] = (bf16_t)(v[u] & 0xffffu);
                            if (ok1) d1[ro] = (bf16_t)(v[u] >> 16);
                        }
                    }
                }
            }
            if (!fast_done) {
#pragma unroll 1
            for (int idx = tid; idx < total; idx += NTH) {
                const int row = idx / cpr, c2 = (idx - row * cpr) * 2;
                if (mh + row >= p.M || n0 + c2 >= p.N) continue;
                const unsigned v = *(const unsigned*)(Cs + row * CSTR + c2);
                int b = bfirst, t = tfirst + c2;
                while (t >= cs) { t -= cs; ++b; }
                bf16_t* dst = Cg + (long)b * p.sC + (long)(mh + row) * p.ldc + t;
                if (pairs) {
                    *(unsigned*)dst = v;
                } else {
                    *dst = (bf16_t)(v & 0xffffu);
                    if (n0 + c2 + 1 < p.N) {
                        if (t + 1 < cs) dst[1] = (bf16_t)(v >> 16);
                        else Cg[(long)(b + 1) * p.sC + (long)(mh + row) * p.ldc] = (bf16_t)(v >> 16);
                    }
                }
            }
            }
        } else if (rpref) {
            if (NH == 2) fetch_residual(hh);                      // 320-wide: one batch right behind the staging barrier
            const bf16_t* cs_ = Cs + rrow0 * CSTR + rc8;
            bf16_t* cg_ = Cg + (long)(mh + rrow0) * p.ldc + n0 + rc8;
#pragma unroll
            for (int u = 0; u < RIT; ++u) {
                const int row = rrow0 + u * RPP;
                if (!ractive || row >= HROWS || mh + row >= p.M) continue;
                uint4 v = *(const uint4*)(cs_ + u * RPP * CSTR);
                v.x = add2bf(v.x, rpre[u].x); v.y = add2bf(v.y, rpre[u].y); v.z = add2bf(v.z, rpre[u].z); v.w = add2bf(v.w, rpre[u].w);
                *(uint4*)(cg_ + (long)u * RPP * p.ldc) = v;
            }
        } else if (BN != 320 && p.wide && !Rg) {
            // no residual: batches of eight unconditional LDS reads (clamped index), then the guarded stores
            const int cpr = BNo >> 3;
            const int total = HROWS * cpr;
#pragma unroll 1
            for (int i0 = 0; i0 < total; i0 += 8 * NTH) {
                uint4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = min(i0 + u * NTH + tid, total - 1);
                    const int row = idx / cpr;
                    v[u] = *(const uint4*)(Cs + row * CSTR + (idx - row * cpr) * 8);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * NTH + tid;
                    const int row = idx / cpr, c8 = (idx - row * cpr) * 8;
                    if (idx < total && mh + row < p.M && n0o + c8 < Nout) *(uint4*)(Cg + (long)(mh + row) * p.ldc + n0o + c8) = v[u];
                }
            }
        } else if (p.wide) {
            const int cpr = BNo >> 3;
            const int total = HROWS * cpr;
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
            const int total = HROWS * cpr;
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
    XL_STAMP(5)
    if (p.timing && tid == 0) {
        unsigned long long* t = p.timing + 8 * (long)blockIdx.x;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = ts_[k];
    }
#undef XL_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_xlp_kernel — the PERSISTENT form of the 256 x 256 GEMM tile (round 3).
//
// Why: at K = 640 a tile of the kernel above is 16.2 us of main loop + 7.2 us that overlap with nothing (first-slab latency 2.7,
// accumulator staging through LDS 1.2, store issue 2.2, setup + barriers 1.1: s_memtime stamps, profiles/r03_xl_timing_stamps.log), and
// with the whole epilogue compiled out the K = 640 GEMMs run 23-37 % faster (profiles/r03_xl_epilogue_ablation.log).  One workgroup
// per CU owns the CU for the whole launch and walks the tile order; per tile:
//   main loop (the same quadrant-phase schedule, same LDS layout, same reduction order)
//   -> the ring is dead: the NEXT tile's slabs 0 and 1 are issued right away (they land while this tile is stored)
//   -> epilogue WITHOUT LDS: bias / GEGLU in registers, the 16 x 64 strip of a wave is transposed between the four 16-lane rows with
//      4 v_permlane32_swap + 4 v_permlane16_swap so that every lane holds 16-byte row segments (64 contiguous bytes per row per
//      instruction), residual added from 16-byte loads, 16-byte stores — no staging pass, no epilogue barriers.
// The per-lane DMA offsets do not depend on the tile: row / column tails are expressed through the buffer descriptors' num_records
// (xl_make_rsrc_bounded), so moving to the next tile is two descriptors in SGPRs.
// Hand-counted VM counter (gfx9: ONE in-order counter for loads, stores and LDS-DMA): at the top of a tile its slab 0 must have
// landed; younger than it are slab 1's three units and whatever the previous epilogue issued afterwards (its stores, the next bias
// vectors).  Those are counted exactly when every store instruction of the previous tile was executed by every wave (interior tile:
// unconditional stores); after an edge tile the wait is conservative (it sits the stores out).
// Takes: plain / GEGLU epilogue, optional residual, wide (16-byte) C rows; everything else stays on gemm_xl_kernel.
template <bool GEGLU, bool HAS_R>
__global__ __launch_bounds__(512, 2) void gemm_xlp_kernel(GCParams p) {
    using G = Geo<256>;
    constexpr int BM = 256, BN = 256;
    constexpr int TI = G::TI, TJ = G::TJ, TJ0 = G::TJ0, TJ1 = TJ - TJ0, TIH = TI / 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = G::BNP * 128, BUF = A_BYTES + B_BYTES;
    constexpr int PA = G::PA, PB0 = G::PB0, PB1 = G::PB1;
    static_assert(TJ == 4 && TI == 8 && PB0 == PB1, "the register epilogue is written for the 128 x 64 wave tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave % G::WM, wn = wave / G::WM;
    const int nt = p.K / 64;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_t*)smem;

    // ---- tile-invariant DMA bookkeeping ----
    unsigned a_voff[2][PA]; int a_lds[2][PA];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < PA; ++e) {
            const int row0 = a_piece_row0<BN>(h, wave, e);
            a_lds[h][e] = row0 * 128;
            a_voff[h][e] = (unsigned)(((long)piece_lane_row(row0, lane) * p.lda) * 2 + piece_lane_chunk(row0, lane) * 16);
        }
    unsigned b_voff[2][PB0]; int b_lds[2][PB0];
#pragma unroll
    for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int e = 0; e < PB0; ++e) {
            const int row0 = b_piece_row0<BN>(part, wave, e);
            b_lds[part][e] = A_BYTES + row0 * 128;
            b_voff[part][e] = (unsigned)(((long)piece_lane_row(row0, lane) * p.ldw) * 2 + piece_lane_chunk(row0, lane) * 16);
        }
    const int fo0 = frag_off(0, lane, 0), fo1 = frag_off(0, lane, 1);
    const int a_rd = a_tile_row0<BN>(wm, 0) * 128;
    const int b_rd = A_BYTES + b_tile_row0<BN>(wn, 0) * 128;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- tile walk ----
    int bid = blockIdx.x, tm = 0, tn = 0;
    auto advance = [&](int& b) -> bool {
        for (; b < p.nblk; b += (int)gridDim.x)
            if (xl_tile_coords_at(p, b, tm, tn)) return true;
        return false;
    };
    if (!advance(bid)) return;
    int m0 = tm * BM, n0 = tn * BN;
    xl_rsrc_t rsA, rsB;
    auto set_tile = [&](int m, int n) {
        rsA = xl_make_rsrc_bounded(p.A + (long)m * p.lda, (long)(p.M - m) * p.lda * 2);
        rsB = xl_make_rsrc_bounded(p.W + (long)n * p.ldw, (long)(p.N - n) * p.ldw * 2);
    };
    set_tile(m0, n0);

#define XLP_PIECE_A(h, e, T) { if ((T) < nt) xl_glds(rsA, lds0 + ((T) & 1) * BUF + a_lds[h][e], a_voff[h][e], (T) * 128); }
#define XLP_PIECE_B(part, e, T) { if ((T) < nt) xl_glds_b(rsB, lds0 + ((T) & 1) * BUF + b_lds[part][e], b_voff[part][e], (T) * 128); }
#define XLP_ISSUE_A(h, T) { XLP_PIECE_A(h, 0, T) XLP_PIECE_A(h, 1, T) }
#define XLP_ISSUE_B(part, T) { XLP_PIECE_B(part, 0, T) XLP_PIECE_B(part, 1, T) }
#define XLP_PROLOGUE() { XLP_ISSUE_A(0, 0) XLP_ISSUE_B(0, 0) XLP_ISSUE_B(1, 0) XLP_ISSUE_A(1, 0) XLP_ISSUE_A(0, 1) XLP_ISSUE_B(0, 1) XLP_ISSUE_B(1, 1) }

    // bias of this lane's 16 columns (4 per MFMA tile j; GEGLU: j = 0, 1 values, j = 2, 3 their gates), fetched per tile
    float4 bq[TJ];
    const bool has_bias = p.bias != nullptr;
    auto load_bias = [&](int n) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int col = n + wn * 64 + j * 16 + 4 * fq;
                bq[j] = *(const float4*)(p.bias + min(col, p.N - 4));      // clamped: columns past N are never stored
            }
        }
    };
    load_bias(n0);
    XLP_PROLOGUE()

    constexpr int INFLIGHT = PA + PB0 + PB1;                      // slab 1's first three units follow slab 0's last
    constexpr int NST = GEGLU ? TI : 2 * TI;                      // 16-byte stores per lane per tile
    bool first = true, counted = false;
    f32x4_t acc[TI][TJ];
    Frag8 af[TIH][2], bfr[TJ][2];
#define XLP_READ_A(h, buf_)                                                                                                       \
    {                                                                                                                             \
        const unsigned char* s_ = smem + (buf_) * BUF + a_rd + (h) * TIH * 2048;                                                  \
        _Pragma("unroll") for (int i = 0; i < TIH; ++i) {                                                                         \
            af[i][0].u = *(const uint4*)(s_ + i * 2048 + fo0);                                                                    \
            af[i][1].u = *(const uint4*)(s_ + i * 2048 + fo1);                                                                    \
        }                                                                                                                         \
    }
#define XLP_READ_B(J0, NJ, buf_)                                                                                                  \
    {                                                                                                                             \
        const unsigned char* s_ = smem + (buf_) * BUF + b_rd;                                                                     \
        _Pragma("unroll") for (int j = (J0); j < (J0) + (NJ); ++j) {                                                              \
            bfr[j][0].u = *(const uint4*)(s_ + j * 2048 + fo0);                                                                   \
            bfr[j][1].u = *(const uint4*)(s_ + j * 2048 + fo1);                                                                   \
        }                                                                                                                         \
    }
#define XLP_MMA(h, J0, NJ)                                                                                                        \
    {                                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                          \
            _Pragma("unroll") for (int i = 0; i < TIH; ++i)                                                                       \
                _Pragma("unroll") for (int j = (J0); j < (J0) + (NJ); ++j)                                                        \
                    acc[(h) * TIH + i][j] = MDX_MFMA_16x16x32(bfr[j][kk].v, af[i][kk].v, acc[(h) * TIH + i][j]);                  \
        __builtin_amdgcn_s_setprio(0);                                                                                            \
    }
#define XLP_SEG_END() { xl_wait_lgkm0(); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define XLP_MMA_END() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    for (;;) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        // slab 0 of this tile has landed once only the younger VM operations are outstanding
        if (nt <= 1) xl_wait_vmcnt<0>();
        else if (first) xl_wait_vmcnt<INFLIGHT>();
        else if (counted) { if (has_bias) xl_wait_vmcnt<INFLIGHT + NST + TJ>(); else xl_wait_vmcnt<INFLIGHT + NST>(); }
        else xl_wait_vmcnt<INFLIGHT>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 1) {                                              // the stagger: group 1 runs one barrier behind group 0
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int t = 0; t < nt; ++t) {                               // schedule 0 of gemm_xl_kernel
            const int buf = t & 1;
            XLP_READ_A(0, buf)
            XLP_READ_B(0, TJ0, buf)
            XLP_ISSUE_A(1, t + 1)
            XLP_SEG_END()
            XLP_MMA(0, 0, TJ0)
            XLP_MMA_END()
            XLP_READ_B(TJ0, TJ1, buf)
            XLP_ISSUE_A(0, t + 2)
            XLP_SEG_END()
            XLP_MMA(0, TJ0, TJ1)
            XLP_MMA_END()
            XLP_READ_A(1, buf)
            XLP_ISSUE_B(0, t + 2)
            XLP_SEG_END()
            XLP_MMA(1, TJ0, TJ1)
            XLP_MMA_END()
            XLP_ISSUE_B(1, t + 2)
            if (t + 2 < nt) xl_wait_vmcnt<INFLIGHT>(); else xl_wait_vmcnt<0>();
            XLP_SEG_END()
            XLP_MMA(1, 0, TJ0)
            XLP_MMA_END()
        }
        if (grp == 0) {                                              // balance the stagger: after this barrier the ring is dead
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // close the compiler's bookkeeping of the bias loads BEFORE new DMA traffic is issued (it would otherwise guard their first
        // use in the epilogue with a wait that also sits out the prefetch)
#pragma unroll
        for (int j = 0; j < TJ; ++j) asm volatile("" : "+v"(bq[j].x), "+v"(bq[j].y), "+v"(bq[j].z), "+v"(bq[j].w));

        // ---- next tile: descriptors + its first two slabs, before this tile is stored ----
        int nb = bid + (int)gridDim.x;
        const int cm0 = m0, cn0 = n0;
        const bool has_next = advance(nb);
        if (has_next) {
            m0 = tm * BM; n0 = tn * BN;
            set_tile(m0, n0);
            XLP_PROLOGUE()
        }

        // ---- epilogue of tile (cm0, cn0): registers -> global ----
        {
            const int No = GEGLU ? p.N / 2 : p.N;
            const int ocol0 = (GEGLU ? cn0 / 2 + wn * 32 : cn0 + wn * 64) + 8 * fq;     // first of this lane's 8 (+ 8 at +32) output columns
            const int rowb = cm0 + wm * (TI * 16) + fr;                                  // row of MFMA tile i = rowb + 16 i
            const bool interior = cm0 + BM <= p.M && cn0 + BN <= p.N;                   // wave-uniform
            bf16_t* cg = (bf16_t*)p.C + (long)rowb * p.ldc + ocol0;
            constexpr int NCH = GEGLU ? 1 : 2;                                            // 16-byte chunks per lane per row tile
            // Store layout.  After the transpose lane (fr, fq) holds, for row tile i, chunk 0 = columns 8 fq .. +7 and (plain epilogue)
            // chunk 1 = the same columns + 32 of row fr: storing them as they are writes 16 rows x 64 bytes per instruction — half cache
            // lines, twice the requests of the LDS-staged epilogue's 512-byte row segments (measured: the prefetch's gain was eaten by the
            // stores).  Instead chunk 1 is rotated by 8 lanes inside each 16-lane row (DPP row_ror:8) and the two instructions of a row
            // tile cover rows 0-7 and rows 8-15 in FULL 128-byte lines: lanes fr < 8 carry chunk 0 of their row in A and the rotated
            // chunk 1 (row fr + 8) in B; lanes fr >= 8 carry the rotated chunk 1 (row fr - 8) in A and chunk 0 of their row in B.
            // GEGLU has one chunk (64 bytes per row and wave): stored as it is.
            const bool lo8 = fr < 8;
            const int rA = (fr & 7), rB = 8 + (fr & 7);                                   // row inside the 16-row tile of store A / B
            const int cA = lo8 ? 0 : 32, cB = lo8 ? 32 : 0;                               // column offset of this lane's 16 bytes in store A / B
            const int rowt = cm0 + wm * (TI * 16);                                        // first row of the wave's strip
            bf16_t* cgA = (bf16_t*)p.C + (long)(rowt + rA) * p.ldc + ocol0 + cA;
            bf16_t* cgB = (bf16_t*)p.C + (long)(rowt + rB) * p.ldc + ocol0 + cB;
            uint4 rres[HAS_R ? TI : 1][NCH];
            if (HAS_R) {                                                                   // one batch, clamped rows (edge tiles), unconditional
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    if (GEGLU) {
                        rres[HAS_R ? i : 0][0] = *(const uint4*)((const bf16_t*)p.R + (long)min(rowb + 16 * i, p.M - 1) * p.ldr + min(ocol0, No - 8));
                    } else {
                        const int ra = min(rowt + 16 * i + rA, p.M - 1), rb = min(rowt + 16 * i + rB, p.M - 1);
                        rres[HAS_R ? i : 0][0] = *(const uint4*)((const bf16_t*)p.R + (long)ra * p.ldr + min(ocol0 + cA, No - 8));
                        rres[HAS_R ? i : 0][NCH - 1] = *(const uint4*)((const bf16_t*)p.R + (long)rb * p.ldr + min(ocol0 + cB, No - 8));
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                unsigned tx[TJ], ty[TJ];                                                  // packed pairs of MFMA tile j: columns 4 fq + {0,1} / {2,3}
#pragma unroll
                for (int j = 0; j < (GEGLU ? 2 : TJ); ++j) {
                    const float bb[4] = {bq[j].x, bq[j].y, bq[j].z, bq[j].w};
                    float o[4];
                    if (GEGLU) {
                        const float gg[4] = {bq[j + 2].x, bq[j + 2].y, bq[j + 2].z, bq[j + 2].w};
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {             // pairs: the GELU runs on packed fp32 (common.h: gelu_erf_f2)
                            const f32x2_t ge = gelu_erf_f2(f32x2_t{acc[i][j + 2][e] + gg[e], acc[i][j + 2][e + 1] + gg[e + 1]});
                            o[e] = (acc[i][j][e] + bb[e]) * ge.x; o[e + 1] = (acc[i][j][e + 1] + bb[e + 1]) * ge.y;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[i][j][e] + bb[e];
                    }
                    tx[j] = pack2bf(o[0], o[1]); ty[j] = pack2bf(o[2], o[3]);
                }
                // 4 x 4 transpose of 8-byte items between the four 16-lane rows (fq) of the wave: lane fq ends up with columns
                // 8 fq .. 8 fq + 7 of tiles (0, 1) [chunk 0] and of tiles (2, 3) [chunk 1, +32 columns]
                uint4 ch[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    unsigned ax = tx[2 * c], ay = ty[2 * c], bx = tx[2 * c + 1], by = ty[2 * c + 1];
                    { auto r_ = __builtin_amdgcn_permlane32_swap(ax, bx, false, false); ax = r_[0]; bx = r_[1]; }
                    { auto r_ = __builtin_amdgcn_permlane32_swap(ay, by, false, false); ay = r_[0]; by = r_[1]; }
                    { auto r_ = __builtin_amdgcn_permlane16_swap(ax, bx, false, false); ax = r_[0]; bx = r_[1]; }
                    { auto r_ = __builtin_amdgcn_permlane16_swap(ay, by, false, false); ay = r_[0]; by = r_[1]; }
                    ch[c] = make_uint4(ax, ay, bx, by);
                }
                if (GEGLU) {
                    if (HAS_R) {
                        const uint4 r4 = rres[HAS_R ? i : 0][0];
                        ch[0].x = add2bf(ch[0].x, r4.x); ch[0].y = add2bf(ch[0].y, r4.y); ch[0].z = add2bf(ch[0].z, r4.z); ch[0].w = add2bf(ch[0].w, r4.w);
                    }
                    bf16_t* crow = cg + (long)(16 * i) * p.ldc;
                    if (interior) *(uint4*)crow = ch[0];
                    else if (rowb + 16 * i < p.M && ocol0 + 8 <= No) *(uint4*)crow = ch[0];
                } else {
                    // rotate chunk 1 by 8 lanes inside each row of 16 (DPP row_ror:8 = 0x128), then pick the store layout
                    uint4 rot;
                    rot.x = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[NCH - 1].x, 0x128, 0xf, 0xf, false);
                    rot.y = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[NCH - 1].y, 0x128, 0xf, 0xf, false);
                    rot.z = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[NCH - 1].z, 0x128, 0xf, 0xf, false);
                    rot.w = (unsigned)__builtin_amdgcn_mov_dpp((int)ch[NCH - 1].w, 0x128, 0xf, 0xf, false);
                    uint4 vA, vB;
                    vA.x = lo8 ? ch[0].x : rot.x; vA.y = lo8 ? ch[0].y : rot.y; vA.z = lo8 ? ch[0].z : rot.z; vA.w = lo8 ? ch[0].w : rot.w;
                    vB.x = lo8 ? rot.x : ch[0].x; vB.y = lo8 ? rot.y : ch[0].y; vB.z = lo8 ? rot.z : ch[0].z; vB.w = lo8 ? rot.w : ch[0].w;
                    if (HAS_R) {
                        const uint4 r0 = rres[HAS_R ? i : 0][0], r1 = rres[HAS_R ? i : 0][NCH - 1];
                        vA.x = add2bf(vA.x, r0.x); vA.y = add2bf(vA.y, r0.y); vA.z = add2bf(vA.z, r0.z); vA.w = add2bf(vA.w, r0.w);
                        vB.x = add2bf(vB.x, r1.x); vB.y = add2bf(vB.y, r1.y); vB.z = add2bf(vB.z, r1.z); vB.w = add2bf(vB.w, r1.w);
                    }
                    bf16_t* pa = cgA + (long)(16 * i) * p.ldc;
                    bf16_t* pb = cgB + (long)(16 * i) * p.ldc;
                    if (interior) {                                                       // every lane stores: the instruction count is exact
                        *(uint4*)pa = vA; *(uint4*)pb = vB;
                    } else {
                        if (rowt + 16 * i + rA < p.M && ocol0 + cA + 8 <= No) *(uint4*)pa = vA;
                        if (rowt + 16 * i + rB < p.M && ocol0 + cB + 8 <= No) *(uint4*)pb = vB;
                    }
                }
            }
            counted = interior;
        }
        if (!has_next) break;
        bid = nb;
        first = false;
        load_bias(n0);
    }
#undef XLP_PIECE_A
#undef XLP_PIECE_B
#undef XLP_ISSUE_A
#undef XLP_ISSUE_B
#undef XLP_PROLOGUE
#undef XLP_READ_A
#undef XLP_READ_B
#undef XLP_MMA
#undef XLP_SEG_END
#undef XLP_MMA_END
}

// LDS: operand ring (2 slabs) or the bf16 C tile, whichever is larger, + the addend rows
template <int BN>
constexpr size_t xl_smem_bytes() {
    constexpr size_t ring = (size_t)2 * (256 + Geo<BN>::BNP) * 128;
    constexpr size_t epi = (size_t)(BN == 320 ? 128 : 256) * (BN + 8) * 2 + (size_t)XL_SLOTS * BN * sizeof(float);   // C staging + addend rows
    return ring > epi ? ring : epi;
}

bool xd_supported(const GCParams& q);                                          // gemm_xd.hip
int launch_gemm_xd(const GCParams& q, int cus, hipStream_t st);

template <int BN, bool CONV, int SCHED>
static int launch_xl(const GCParams& p, hipStream_t st) {
    constexpr size_t smem_kxs = (size_t)2 * 272 * 128 + (size_t)2 * Geo<BN>::BNP * 128;      // SCHED 4: two 272-row A buffers + two weight slabs
    constexpr size_t smem = (SCHED == 4 && smem_kxs > xl_smem_bytes<BN>()) ? smem_kxs : xl_smem_bytes<BN>();
    static_assert(smem <= 163840, "LDS budget");
    auto kern = gemm_xl_kernel<BN, CONV, SCHED>;
    if (int rc = ensure_dyn_smem((const void*)kern, smem, "xl")) return rc;
    GCParams q = p;
    q.mt = (p.M + 255) / 256; q.nt = (p.N + BN - 1) / BN;
    const int swz = (int)opt(OPT_GEMM_SWZ);
    const int dbg = (int)opt(OPT_XL_DBG);
    q.dbg = dbg;
    q.swz = swz && q.nt > 1 && q.mt >= 64;
    unsigned nblk_raster = 0;
    const int raster = (int)opt(OPT_XL_RASTER);
    if (raster == 0) q.swz = 0;
    if (raster == 2 && q.mt >= 64) {
        // only when every XCD gets >= 8 M-groups: M-groups are dealt out whole, and with few of them the XCD that draws one more runs
        // an extra round of tiles (measured: the 4x7-level convs, 84 M-tiles x 5 = 17 groups, went from 0.49 to 0.72 ms)
        raster_shape(q.mt, q.nt, &q.gm, &q.gn);
        if (const int fgm = (int)opt(OPT_XL_GM)) q.gm = fgm < q.mt ? fgm : q.mt;        // sweeps (tools/xlone.py --raster)
        if (const int fgn = (int)opt(OPT_XL_GN)) q.gn = fgn < q.nt ? fgn : q.nt;
        const long nb = raster_blocks(q.mt, q.nt, q.gm, q.gn);
        if ((q.mt + q.gm - 1) / q.gm >= 64 && nb < 0x7fffffffL) { q.swz = 2; nblk_raster = (unsigned)nb; }
    }
    const int timing = (int)opt(OPT_XL_TIMING);
    const unsigned nblk_t = q.swz == 2 ? nblk_raster : q.swz ? (unsigned)((q.mt + 7) / 8 * 8 * q.nt) : (unsigned)(q.mt * q.nt);
    q.timing = (timing && p.ws && (long)nblk_t * 64 <= p.ws_bytes) ? (unsigned long long*)p.ws : nullptr;
    const unsigned nblk = nblk_t;
    q.nblk = (int)nblk;
    if constexpr (BN == 256 && !CONV && SCHED == 0) {
        // persistent form (gemm_xlp_kernel): one workgroup per CU walks the tile order, the next tile's first slabs land while this one
        // is stored.  Plain / GEGLU epilogue, optional residual, 16-byte C rows, enough tiles for the walk to matter.
        static const int cus = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
        if (opt(OPT_XL_PERSIST) && q.wide && !q.col_split && !q.temb && (q.epi == 0 || q.epi == 1) && !q.timing && nblk >= 2u * (unsigned)cus &&
            (long)q.M * q.lda * 2 > 0 && q.bias != (const float*)q.C) {
            const bool geglu = q.epi == 1, has_r = q.R != nullptr;
            if (geglu && has_r) goto not_persistent;
            // W-direct form (gemm_xd.hip): the caller supplied the weights in fragment order and K is a whole number of unit pairs
            if (opt(OPT_XD) && xd_supported(q)) return launch_gemm_xd(q, cus, st);
            {
                auto launch_p = [&](auto kp) -> int {
                    if (int rc = ensure_dyn_smem((const void*)kp, smem, "xlp")) return rc;
                    hipLaunchKernelGGL(kp, dim3((unsigned)cus), dim3(512), smem, st, q);
                    char tag[96];
                    snprintf(tag, sizeof tag, "gemm_xlp_kernel<256x256,%s%s>", geglu ? "geglu" : "gemm", has_r ? "+res" : "");
                    return check_launch(tag);
                };
                if (geglu) return launch_p(gemm_xlp_kernel<true, false>);
                if (has_r) return launch_p(gemm_xlp_kernel<false, true>);
                return launch_p(gemm_xlp_kernel<false, false>);
            }
        }
    not_persistent:;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), smem, st, q);
    char tag[96];
    snprintf(tag, sizeof tag, "gemm_xl_kernel<256x%d,%s>", BN, CONV ? (SCHED == 4 ? "conv,kxs" : "conv") : "gemm");   // (schedules 0-3 are tuning knobs, not part of the name)
    return check_launch(tag);
}

// Can the XL main loop run this problem at all?  (The caller's cost model decides whether it should.)  bn = 256 or 160.
bool xl_supported(const GCParams& p, bool conv, int bn) {
    if (p.batch > 1 || p.splitk > 1 || p.c_f32 || (p.N % 4) || (p.K % 64) || p.Vt) return false;
    if (p.col_split && (conv || p.bias || p.temb || p.R || p.epi || (p.sC & 1) || (p.ldc & 1) || p.col_split < 16)) return false;
    if (bn != 256 && bn != 160 && bn != 320) return false;
    if (p.epi == 1 && (bn != 256 || (p.N % 64))) return false;
    // SiLU epilogue: only the prologue's map-encoder convs and the time MLP use it (never >= 160 tiles); instantiating it here cost the
    // 256-wide kernels 17 spilled VGPRs (the residual prefetch went through scratch behind a full vmcnt wait)
    if (p.epi == 2) return false;
    if (conv) {
        if (p.kh != 3 || p.kw != 3 || p.ph != 1 || p.pw != 1 || (p.Cin % 64) || !p.cimajor) return false;   // pad 1: input pixel index monotonic in m
        // voffsets are relative to the tile's first receptive-field pixel: 256 output pixels span < 2^31 bytes for every real shape,
        // but keep the arithmetic honest
        const long span = ((long)(256 / p.Wo + 3) * p.sh * p.Wi + 256L * p.sw + 3 * p.Wi) * p.lda * 2;
        if (span >= 0x40000000L) return false;
    } else if ((long)256 * p.lda * 2 >= 0x40000000L) return false;
    if ((long)bn * p.ldw * 2 >= 0x40000000L) return false;
    if (p.temb && p.epi != 1) {
        const int rows = p.rows_per_b > 0 ? p.rows_per_b : 1;
        if (256 / rows + 2 > XL_SLOTS) return false;
    }
    return true;
}

// MDX_XL_SCHED: 0 (default) = four quadrant phases per slab, refill in the load segments; 1 = two phases per slab with the DMA
// issued between the MFMAs (measured 10 % slower: the DMA issue lengthens the MFMA segments, which are the serial resource);
// 2 / 3 = quadrant variants (see the kernel).
int launch_gemm_xl(const GCParams& p, bool conv, int bn, hipStream_t st) {
    const int sched = (int)opt(OPT_XL_SCHED);
#define XL_GO(BN_, S_) (conv ? launch_xl<BN_, true, S_>(p, st) : launch_xl<BN_, false, S_>(p, st))
    if (bn == 320) {
        // 3x3 / stride 1 convs: the three horizontal taps of a (channel block, ky) share one A slab (schedule 4; XL_KXSHARE = 0: per-tap slabs, A/B)
        // Built, correct (tests/test_routes_gpu.py::test_xl_conv in a -DMDX_XL_KXS build) and 18-21 % SLOWER than one slab per tap: the border
        // selects + per-piece offset arithmetic put ~100 more VALU instructions per slab into the load segments, which are the serial resource
        // (profiles/r06_xl_kxshare_ab.log) — more than the -29 % operand bytes buy (+6...8 %, profiles/r06_xl_a_bytes_ablation.log).  Not in the
        // product build; `make side SIDE=kxs FLAGS=-DMDX_XL_KXS` builds it for A/B runs.
#ifdef MDX_XL_KXS
        if (conv && opt(OPT_XL_KXSHARE) && p.sh == 1 && p.sw == 1 && p.Hi == p.Ho && p.Wi == p.Wo && p.Wo >= 2 && (p.K / 64) % 9 == 0 && p.N % 320 == 0)
            return launch_xl<320, true, 4>(p, st);
#endif
        return XL_GO(320, 0);
    }
    if (bn == 256) return sched == 1 ? XL_GO(256, 1) : sched == 2 ? XL_GO(256, 2) : sched == 3 ? XL_GO(256, 3) : XL_GO(256, 0);
    return sched == 1 ? XL_GO(160, 1) : sched == 2 ? XL_GO(160, 2) : sched == 3 ? XL_GO(160, 3) : XL_GO(160, 0);
#undef XL_GO
}

}  // namespace mdx
