// Host-side model of gemm_xl.hip's LDS traffic, compiled with g++ by tests/test_xl_layout.py (no GPU needed).
// It replays, with the kernel's own index functions (magicdrive_amd/csrc/xl_layout.h):
//   1. the LDS-DMA placement of every load unit (A0, A1, B0, B1) of one K slab: wave w, piece e, lane l writes 16 bytes at
//      piece_row0 * 128 + l * 16 and fetches logical (row, chunk);
//   2. the fragment reads of every wave / MFMA tile / lane / k32 step;
// and checks that (a) every operand slot is written exactly once and only by the unit that owns its row, (b) a fragment read returns
// the (row, k-chunk) the 16x16x32 MFMA operand layout expects, (c) each 16-lane group of a ds_read_b128 touches 16 distinct 16-byte
// slots of the 256-byte bank row (conflict-free), (d) the rows a quadrant phase reads belong to the unit the schedule says it reads.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../magicdrive_amd/csrc/xl_layout.h"

using namespace mdx_xl;

struct Tag { int region, unit, row, chunk; };   // region 0 = A, 1 = B

template <int BN>
static int check() {
    using G = Geo<BN>;
    int errors = 0;
    const int a_rows = 256, b_rows = G::BNP;
    std::vector<Tag> A(a_rows * 8, Tag{-1, -1, -1, -1}), B(b_rows * 8, Tag{-1, -1, -1, -1});
    std::vector<int> Acnt(a_rows * 8, 0), Bcnt(b_rows * 8, 0);
    // ---- DMA placement ----
    for (int w = 0; w < 8; ++w) {
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < G::PA; ++e) {
                const int row0 = a_piece_row0<BN>(h, w, e);
                if (row0 < 0 || row0 % 8 || row0 + 8 > a_rows) { printf("BN=%d: bad A piece row0 %d\n", BN, row0); ++errors; continue; }
                for (int l = 0; l < 64; ++l) {
                    const int slot = row0 * 8 + l;                  // 16-byte slot index inside the A region (linear destination)
                    A[slot] = Tag{0, h, piece_lane_row(row0, l), piece_lane_chunk(row0, l)};
                    ++Acnt[slot];
                }
            }
        for (int part = 0; part < 2; ++part)
            for (int e = 0; e < (part ? G::PB1 : G::PB0); ++e) {
                int row0 = b_piece_row0<BN>(part, w, e);
                const bool dummy = row0 < 0;
                if (dummy) row0 = b_dummy_row0<BN>(w);
                if (row0 % 8 || row0 + 8 > b_rows || (dummy && row0 < BN)) { printf("BN=%d: bad B piece row0 %d\n", BN, row0); ++errors; continue; }
                if (dummy) continue;                                // zeros into scratch rows
                for (int l = 0; l < 64; ++l) {
                    const int slot = row0 * 8 + l;
                    B[slot] = Tag{1, part, piece_lane_row(row0, l), piece_lane_chunk(row0, l)};
                    ++Bcnt[slot];
                }
            }
    }
    for (int s = 0; s < a_rows * 8; ++s)
        if (Acnt[s] != 1) { if (errors < 20) printf("BN=%d: A slot %d written %d times\n", BN, s, Acnt[s]); ++errors; }
    for (int s = 0; s < BN * 8; ++s)
        if (Bcnt[s] != 1) { if (errors < 20) printf("BN=%d: B slot %d written %d times\n", BN, s, Bcnt[s]); ++errors; }
    // a slot's tag row must be the LDS row it sits in (the DMA never moves data across rows), chunks of a row a permutation
    for (int s = 0; s < a_rows * 8; ++s) if (A[s].row != s / 8) { ++errors; if (errors < 20) printf("BN=%d: A slot %d holds row %d\n", BN, s, A[s].row); }
    for (int s = 0; s < BN * 8; ++s) if (B[s].row != s / 8) { ++errors; if (errors < 20) printf("BN=%d: B slot %d holds row %d\n", BN, s, B[s].row); }

    // ---- fragment reads ----
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    for (int w = 0; w < 8; ++w) {
        const int wm = w % G::WM, wn = w / G::WM;
        for (int region = 0; region < 2; ++region) {
            const int ntile = region ? G::TJ : G::TI;
            for (int t = 0; t < ntile; ++t)
                for (int kk = 0; kk < 2; ++kk) {
                    const int row0 = region ? b_tile_row0<BN>(wn, t) : a_tile_row0<BN>(wm, t);
                    int off[64];
                    for (int l = 0; l < 64; ++l) {
                        off[l] = frag_off(row0, l, kk);
                        const Tag& tg = (region ? B : A)[off[l] / 16];
                        const int want_row = row0 + (l & 15), want_chunk = kk * 4 + (l >> 4);
                        if (tg.row != want_row || tg.chunk != want_chunk) {
                            ++errors;
                            if (errors < 20) printf("BN=%d w%d %c tile %d kk %d lane %d: got (row %d, chunk %d) want (%d, %d)\n", BN, w, region ? 'B' : 'A', t, kk, l,
                                                    tg.row, tg.chunk, want_row, want_chunk);
                        }
                        // the unit that owns the row must be the one the phase schedule reads first: A half = t / (TI/2), B part = t >= TJ0
                        const int want_unit = region ? (t >= G::TJ0) : (t / (G::TI / 2));
                        if (tg.unit != want_unit) { ++errors; if (errors < 20) printf("BN=%d: %c tile %d row %d sits in unit %d, schedule reads it with unit %d\n", BN, region ? 'B' : 'A', t, want_row, tg.unit, want_unit); }
                    }
                    for (int g = 0; g < 4; ++g) {                   // bank check: 16 lanes -> 16 distinct 16-byte slots mod 256 B
                        unsigned seen = 0;
                        for (int k = 0; k < 16; ++k) {
                            const int slot = (off[groups[g][k]] % 256) / 16;
                            if (seen & (1u << slot)) { ++errors; if (errors < 20) printf("BN=%d: bank conflict w%d %c tile %d kk %d group %d\n", BN, w, region ? 'B' : 'A', t, kk, g); }
                            seen |= 1u << slot;
                        }
                    }
                }
        }
    }
    // frag_off(row0) == frag_off(0) + row0 * 128 for the tile row blocks the kernel uses (it hoists frag_off(0, lane, kk))
    for (int l = 0; l < 64; ++l)
        for (int kk = 0; kk < 2; ++kk)
            for (int row0 = 0; row0 < 256; row0 += 16)
                if (frag_off(row0, l, kk) != frag_off(0, l, kk) + row0 * 128) { ++errors; if (errors < 20) printf("frag_off not affine in row0=%d\n", row0); }
    printf("BN=%d: %s (%d errors)\n", BN, errors ? "FAIL" : "ok", errors);
    return errors;
}

// XCD-blocked tile order: every tile exactly once; the tiles an XCD holds side by side (32 consecutive local indices) span at most
// gm M-tiles x gn N-tiles of ONE panel or the tail of one and the head of the next; M-tiles of a panel are contiguous.
static int check_raster() {
    int errors = 0;
    const int shapes[][2] = {{64, 1}, {64, 2}, {65, 3}, {84, 5}, {273, 5}, {1050, 20}, {1050, 3}, {4200, 2}, {333, 7}, {100, 40}, {71, 4}, {4200, 1}, {273, 10}, {273, 40}, {1050, 5}};
    for (auto& sh : shapes) {
        const int mt = sh[0], nt = sh[1];
        int gm, gn;
        raster_shape(mt, nt, &gm, &gn);
        const long nb = raster_blocks(mt, nt, gm, gn);
        std::vector<int> seen((size_t)mt * nt, 0);
        for (long b = 0; b < nb; ++b) {
            int tm, tn;
            if (!raster_tile((int)b, mt, nt, gm, gn, &tm, &tn)) continue;
            if (tm < 0 || tn < 0 || tm >= mt || tn >= nt) { ++errors; continue; }
            ++seen[(size_t)tm * nt + tn];
        }
        for (int v : seen) if (v != 1) ++errors;
        // one panel per gm * gn consecutive locals of an XCD: distinct M-tiles <= gm, distinct N-tiles <= gn, M-tiles contiguous
        for (int xcd = 0; xcd < 8; ++xcd)
            for (long l0 = 0; l0 * 8 + xcd < nb; l0 += gm * gn) {
                int mlo = 1 << 30, mhi = -1, nlo = 1 << 30, nhi = -1;
                for (int k = 0; k < gm * gn; ++k) {
                    int tm, tn;
                    const long b = (l0 + k) * 8 + xcd;
                    if (b >= nb || !raster_tile((int)b, mt, nt, gm, gn, &tm, &tn)) continue;
                    mlo = tm < mlo ? tm : mlo; mhi = tm > mhi ? tm : mhi; nlo = tn < nlo ? tn : nlo; nhi = tn > nhi ? tn : nhi;
                }
                if (mhi >= 0 && (mhi - mlo + 1 > gm || nhi - nlo + 1 > gn)) ++errors;
            }
        std::printf("raster mt=%d nt=%d: panels %d x %d, %ld blocks for %d tiles: %s\n", mt, nt, gm, gn, nb, mt * nt, errors ? "FAIL" : "ok");
    }
    return errors;
}

int main() {
    int e = check<256>() + check<160>() + check<320>() + check_raster();
    return e ? 1 : 0;
}
