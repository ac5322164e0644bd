// xl_layout.h — index arithmetic of the 256-row "XL" GEMM / conv main loop (gemm_xl.hip), kept free of HIP types so the SAME
// functions are compiled into the kernel and into a host-side model (tests/test_xl_layout.py builds tests/xl_layout_check.cpp with
// g++ and replays the LDS-DMA placement + the fragment reads of every wave and lane against the operand matrices).
//
// Geometry: tile 256 x BN x 64, 8 waves.  16x16x32 bf16 MFMA tiles: a wave owns TI x TJ of them.
//   BN = 256: waves 2 (M) x 4 (N), wave tile 128 x 64  (TI = 8, TJ = 4)
//   BN = 160: waves 4 (M) x 2 (N), wave tile  64 x 80  (TI = 4, TJ = 5)
//   BN = 320: waves 2 (M) x 4 (N), wave tile 128 x 80  (TI = 8, TJ = 5)
// LDS buffer (one K slab of 64): A rows [0, 256) then B rows [0, BNP), 128 B per row, UNPADDED (the LDS-DMA writes 1 KiB = 8 rows per
// wave instruction, lane l -> row l >> 3, 16-byte slot l & 7); the slot index is XOR-swizzled by (row >> 1) & 7 on the SOURCE side
// (lane l fetches logical chunk (l & 7) ^ swz(row)) and again on the fragment reads (rule: linear destination, same involution on
// source and read).  With that, the 16-lane groups of a ds_read_b128 {0-3,12-15,20-27 | 4-11,16-19,28-31 | +32} hit 16 distinct
// 16-byte slots of the 256-byte bank row for the 16x16x32 operand layout (lane l: row l & 15, k chunk l >> 4).
//
// Load units ("half tiles"): what one phase of the main loop refills = the rows ONE quadrant phase reads first.
//   A0 / A1: the first / second half of every wave's rows; B0 / B1: the first / second part of every wave's columns.
// Each unit is a whole number of 1-KiB pieces per wave (BN = 160: the B1 unit is padded with dummy pieces that land in scratch rows).
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define XL_HD __host__ __device__ __forceinline__
#else
#define XL_HD inline
#endif

namespace mdx_xl {

template <int BN> struct Geo;
template <> struct Geo<256> {
    static constexpr int WM = 2, WN = 4, TI = 8, TJ = 4;      // wave grid, MFMA tiles per wave (M, N)
    static constexpr int TJ0 = 2;                              // tiles in the wave's first column part (B0); the rest is B1
    static constexpr int BNP = 256;                            // B rows held in LDS (incl. scratch)
    static constexpr int PA = 2, PB0 = 2, PB1 = 2;             // 1-KiB pieces per wave per unit (A0 = A1 = PA)
};
template <> struct Geo<320> {                                      // waves 2 x 4, wave tile 128 x 80: the widest tile the registers allow
    static constexpr int WM = 2, WN = 4, TI = 8, TJ = 5;          // (160 accumulators; B0 = 3 column tiles, B1 = 2 share one register set)
    static constexpr int TJ0 = 3;
    static constexpr int BNP = 320;
    static constexpr int PA = 2, PB0 = 3, PB1 = 2;
};
template <> struct Geo<160> {
    static constexpr int WM = 4, WN = 2, TI = 4, TJ = 5;
    static constexpr int TJ0 = 3;
    static constexpr int BNP = 192;                            // 160 real rows + 32 scratch rows for the dummy pieces
    static constexpr int PA = 2, PB0 = 2, PB1 = 1;
};

// swizzle of a row's 16-byte slots
XL_HD int swz(int row) { return (row >> 1) & 7; }

// ---- rows of a wave's MFMA tiles (LDS row index inside the A / B region) ----
template <int BN> XL_HD int a_tile_row0(int wm, int i) { return wm * (Geo<BN>::TI * 16) + i * 16; }
template <int BN> XL_HD int b_tile_row0(int wn, int j) { return wn * (Geo<BN>::TJ * 16) + j * 16; }

// ---- LDS-DMA pieces: first LDS row of piece e (0..P-1) that wave w issues for a unit -----------------------------------------
// A unit h (0 = A0, 1 = A1): the rows { wm * TI*16 + h * TI*8 + [0, TI*8) : wm } in wave-major order, 8 rows per piece.
template <int BN> XL_HD int a_piece_row0(int h, int w, int e) {
    constexpr int HALF = Geo<BN>::TI * 8;                      // rows of one wave's half
    constexpr int PPW = HALF / 8;                              // pieces per wave-row-block
    const int pc = w * Geo<BN>::PA + e;                        // piece index inside the unit
    const int wm = pc / PPW, within = pc - wm * PPW;
    return wm * (2 * HALF) + h * HALF + within * 8;
}
// B unit part (0 = B0: tiles [0, TJ0), 1 = B1: tiles [TJ0, TJ)).  Returns -1 for a dummy piece (BN = 160, part 1, waves 4..7).
template <int BN> XL_HD int b_piece_row0(int part, int w, int e) {
    constexpr int R0 = Geo<BN>::TJ0 * 16, R1 = (Geo<BN>::TJ - Geo<BN>::TJ0) * 16, WROWS = Geo<BN>::TJ * 16;
    const int rows = part ? R1 : R0;                           // rows of one wave-column-block in this part
    const int ppw = rows / 8;
    const int pc = w * (part ? Geo<BN>::PB1 : Geo<BN>::PB0) + e;
    const int wn = pc / ppw, within = pc - wn * ppw;
    if (wn >= Geo<BN>::WN) return -1;
    return wn * WROWS + (part ? R0 : 0) + within * 8;
}
// scratch row block for dummy piece of wave w (never read)
template <int BN> XL_HD int b_dummy_row0(int w) { return Geo<BN>::WN * Geo<BN>::TJ * 16 + (w & 3) * 8; }

// lane l of a piece whose first LDS row is row0: LDS row and the LOGICAL 16-byte chunk (0..7 of the 64-wide k slab) it must fetch
XL_HD int piece_lane_row(int row0, int lane) { return row0 + (lane >> 3); }
XL_HD int piece_lane_chunk(int row0, int lane) { return (lane & 7) ^ swz(row0 + (lane >> 3)); }

// ---- fragment reads: byte offset inside the A (or B) region of lane `lane`'s 16 bytes for tile row block row0, k32-step kk ----
XL_HD int frag_off(int row0, int lane, int kk) {
    const int row = row0 + (lane & 15);
    const int chunk = (kk << 2) | (lane >> 4);                 // logical chunk: k = 8 * chunk .. +7
    return row * 128 + ((chunk ^ swz(row)) << 4);
}

// ---- tile order: XCD-blocked panels (GCParams.swz == 2; gemm_xl.hip: xl_tile_coords) ----
// Workgroup b runs on XCD b % 8 (dispatch rule, used for speed only); the ~32 workgroups an XCD runs side by side are ONE panel of
// gm consecutive M-tiles x gn consecutive N-tiles.  An XCD walks the N-groups of one M-group back to back (its A panels stay in that
// L2), then takes its next M-group (M-group g belongs to XCD g % 8).  Blocks whose tile falls outside the ragged edge exit.
XL_HD void raster_shape(int mt, int nt, int* gm, int* gn) {
    // gm x gn ~ 32 (the workgroups an XCD runs side by side).  Per tile that XCD's L2 takes in (gm + gn) / (gm gn) operand panels, so
    // the squarest shape wins — weighed against the padding blocks of a ragged grid (they are cheap for the per-tile kernel but
    // unbalance the persistent one: a workgroup that draws padding idles while its XCD neighbours multiply) and against having
    // fewer than 64 M-groups (8 per XCD: groups are dealt out whole; with few of them the XCD that draws one more runs an extra round).
    // Narrow problems (nt <= 5) take all their N-tiles in one group.
    int bm = 1, bn = nt <= 5 ? nt : 1;
    double best = 1e30;
    for (int n = (nt <= 5 ? nt : 2); n <= (nt <= 5 ? nt : (nt < 32 ? nt : 32)); ++n) {
        int m = 32 / n;
        if (m > mt / 16) m = mt / 16;                          // short problems: at least two M-groups per XCD
        if (m < 1) m = 1;
        const int gmn = (mt + m - 1) / m, gnn = (nt + n - 1) / n;
        const double pad = (double)gmn * m * gnn * n / ((double)mt * nt);
        double cost = (double)(m + n) / (double)(m * n) * pad * pad;
        if (gmn < 64) cost *= 1.5;
        if (cost < best) { best = cost; bm = m; bn = n; }
    }
    *gm = bm; *gn = bn;
}
XL_HD long raster_blocks(int mt, int nt, int gm, int gn) {
    const int ngm = (mt + gm - 1) / gm, ngn = (nt + gn - 1) / gn;
    return (long)((ngm + 7) / 8) * 8 * ngn * gm * gn;
}
XL_HD bool raster_tile(int bid, int mt, int nt, int gm, int gn, int* tm, int* tn) {
    const int xcd = bid & 7, local = bid >> 3;
    const int per = gm * gn;
    const int panel = local / per, within = local - panel * per;
    const int ngn = (nt + gn - 1) / gn;
    const int mg_local = panel / ngn, ng = panel - mg_local * ngn;
    const int tn_in = within / gm, tm_in = within - tn_in * gm;
    *tm = (mg_local * 8 + xcd) * gm + tm_in;
    *tn = ng * gn + tn_in;
    return *tm < mt && *tn < nt;
}

}  // namespace mdx_xl
