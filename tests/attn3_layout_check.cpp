// Host-side model of attention3.hip's data flow for ONE wave tile (32 queries x 64 kv, head dim 40), compiled with g++ by
// tests/test_attn3_layout.py (no GPU needed).  With the kernel's own index functions (magicdrive_amd/csrc/attn3_layout.h, xl_layout.h) it
//   1. builds the LDS image of a K tile (lane-linear 80-byte rows) and of a V^T tile (40 rows + the ones row, 16-byte slots XOR-swizzled on
//      the DMA source side exactly as the kernel's pieces fetch them),
//   2. forms every lane's K / Q / V^T fragments from the byte offsets the kernel reads (incl. the one-cell of the FOLD pad slot),
//   3. runs them through an emulated v_mfma_f32_32x32x16 (operand / result layout as documented in attn3_layout.h),
//   4. checks (a) score register r of lane l is s(query l & 31, kv s_kv(r, l)) - m, (b) registers 8 t .. 8 t + 7 of a lane, used UNPERMUTED
//      as its B operand, give O^T = V^T P^T exactly, with the row sums in row 40, (c) each 16-lane group of the ds_read_b128 fragment
//      reads touches 16 distinct 16-byte slots of a 256-byte bank row (conflict-free), K and V^T.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <set>
#include <vector>
#include "../magicdrive_amd/csrc/xl_layout.h"
#include "../magicdrive_amd/csrc/attn3_layout.h"

using namespace mdx_a3;

static const int D = 40, KROW = 80, SUBB = 32 * KROW;
static const int LANE_GROUPS[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                       {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

// D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies a[e] = A[l & 31][8 (l >> 5) + e], b[e] = B[8 (l >> 5) + e][l & 31]; receives acc[r] = D[acc_row(r, l)][l & 31]
static void mfma32(const double a[64][8], const double b[64][8], double acc[64][16]) {
    double A[32][16], B[16][32];
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) { A[l & 31][8 * (l >> 5) + e] = a[l][e]; B[8 * (l >> 5) + e][l & 31] = b[l][e]; }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int i = acc_row(r, l), j = l & 31;
            double s = 0;
            for (int k = 0; k < 16; ++k) s += A[i][k] * B[k][j];
            acc[l][r] += s;
        }
}

static int conflict_free(const int off[64], const char* what) {
    int bad = 0;
    for (int g = 0; g < 4; ++g) {
        std::set<int> slots;
        for (int i = 0; i < 16; ++i) slots.insert((off[LANE_GROUPS[g][i]] % 256) / 16);
        if ((int)slots.size() != 16) { printf("%s: lane group %d touches %d distinct slots\n", what, g, (int)slots.size()); ++bad; }
    }
    return bad;
}

int main() {
    int errors = 0;
    srand(7);
    std::vector<double> Q(32 * D), K(64 * D), V(64 * D);
    for (auto& x : Q) x = rand() % 7 - 3;
    for (auto& x : K) x = rand() % 5 - 2;
    for (auto& x : V) x = rand() % 9 - 4;
    const double m[32] = {3, -2, 0, 5, 1, 1, -7, 2, 0, 0, 4, -1, 6, 2, -3, 8, 3, -2, 0, 5, 1, 1, -7, 2, 0, 0, 4, -1, 6, 2, -3, 8};   // per-query subtracted maximum
    // ---- LDS images (as doubles per bf16 element) ----
    std::vector<double> Kimg(64 * D + 64, 777.0);                 // lane-linear: element (row, d) at row * 40 + d; what follows the last row is garbage here
    for (int r = 0; r < 64; ++r) for (int d = 0; d < D; ++d) Kimg[r * D + d] = K[r * D + d];
    std::vector<double> Vimg(64 * 64, 0.0);                       // [row 0..63][64 elements = 8 slots of 8]; slot index = what the byte offset selects
    for (int pc = 0; pc < 5; ++pc) {                              // the kernel's five V^T pieces: 8 rows each, lane -> (row, logical chunk), written lane-linear
        const int row0 = pc * 8;
        for (int l = 0; l < 64; ++l) {
            const int row = mdx_xl::piece_lane_row(row0, l), ch = mdx_xl::piece_lane_chunk(row0, l);
            const int dst = row0 * 64 + l * 8;                   // element index of the lane's 16 bytes
            for (int e = 0; e < 8; ++e) Vimg[dst + e] = V[(ch * 8 + e) * D + row];
        }
    }
    for (int e = 0; e < 64; ++e) Vimg[40 * 64 + e] = 1.0;          // the ones row
    // ---- QK ----
    double S[2][64][16] = {};
    for (int s = 0; s < 2; ++s)
        for (int ks = 0; ks < 3; ++ks) {
            double a[64][8], b[64][8];
            int off[64];
            for (int l = 0; l < 64; ++l) {
                const int half = l >> 5, col = l & 31;
                const int k0 = k_row(l) * KROW + half * 16;      // the kernel's k0
                const bool cell = ks == 2 && half;               // upper half lanes of the pad k-step read the one-cell
                const int byte = s * SUBB + (ks == 2 ? k0 + 64 - (half ? 16 : 0) : k0 + ks * 32);
                off[l] = ks == 2 ? s * SUBB + k_row(l) * KROW + 64 : byte;
                for (int e = 0; e < 8; ++e) {
                    a[l][e] = cell ? (e == 0 ? 1.0 : 0.0) : Kimg[(ks == 2 ? s * SUBB + k_row(l) * KROW + 64 : byte) / 2 + e];
                    const int dd = ks * 16 + half * 8 + e;
                    b[l][e] = dd < D ? Q[col * D + dd] : (dd == D ? -m[col] : 0.0);
                }
            }
            if (ks < 2) errors += conflict_free(off, "K fragment read");
            mfma32(a, b, S[s]);
        }
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int q = l & 31, kv = 32 * s + s_kv(r, l);
                double ref = -m[q];
                for (int d = 0; d < D; ++d) ref += Q[q * D + d] * K[kv * D + d];
                if (S[s][l][r] != ref) { if (errors < 10) printf("score mismatch s=%d lane=%d r=%d: %g vs %g\n", s, l, r, S[s][l][r], ref); ++errors; }
            }
    // ---- PV with P = the scores themselves (any per-element function of them would do) ----
    double O[2][64][16] = {};
    for (int s = 0; s < 2; ++s)
        for (int t = 0; t < 2; ++t)
            for (int rt = 0; rt < 2; ++rt) {
                double a[64][8], b[64][8];
                int off[64];
                for (int l = 0; l < 64; ++l) {
                    off[l] = vt_off(rt, s, t, l);
                    for (int e = 0; e < 8; ++e) {
                        a[l][e] = Vimg[off[l] / 2 + e];
                        b[l][e] = S[s][l][8 * t + e];
                    }
                }
                errors += conflict_free(off, "V^T fragment read");
                mfma32(a, b, O[rt]);
            }
    for (int rt = 0; rt < 2; ++rt)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int q = l & 31, d = 32 * rt + acc_row(r, l);
                double ref = 0;
                for (int kv = 0; kv < 64; ++kv) {
                    double p = -m[q];
                    for (int dd = 0; dd < D; ++dd) p += Q[q * D + dd] * K[kv * D + dd];
                    ref += p * (d < D ? V[kv * D + d] : (d == D ? 1.0 : 0.0));
                }
                if (O[rt][l][r] != ref) { if (errors < 20) printf("O mismatch rt=%d lane=%d r=%d (d=%d): %g vs %g\n", rt, l, r, d, O[rt][l][r], ref); ++errors; }
            }
    // the store map: lane (q, half) register 4 u + e of row tile 0 is d = 8 u + 4 half + e; registers 0..3 of row tile 1 are d = 32 + 4 half + e; row 40 = register 4 of the lower half
    for (int l = 0; l < 64; ++l) {
        for (int u = 0; u < 4; ++u) for (int e = 0; e < 4; ++e) if (acc_row(4 * u + e, l) != 8 * u + 4 * (l >> 5) + e) ++errors;
        for (int e = 0; e < 4; ++e) if (32 + acc_row(e, l) != 32 + 4 * (l >> 5) + e) ++errors;
    }
    if (32 + acc_row(4, 5) != 40) ++errors;
    printf(errors ? "attn3 layout: %d errors\n" : "attn3 layout: ok\n", errors);
    return errors ? 1 : 0;
}
