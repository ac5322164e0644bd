// attn3_layout.h — index arithmetic of the pipelined head-dim-40 attention kernel (attention3.hip), free of HIP types so the SAME
// functions are compiled into the kernel and into a host-side model (tests/test_attn3_layout.py builds tests/attn3_layout_check.cpp with
// g++: it replays every lane's fragment of one (32 queries x 64 kv) wave tile through an emulated v_mfma_f32_32x32x16 and compares with
// softmax-free P V computed directly).
//
// The point of the layout: S^T = K Q^T and O^T = V^T P^T both run on 32x32x16 MFMAs, and the probabilities never leave their lane.
//   v_mfma_f32_32x32x16 (A 32 x 16, B 16 x 32, C/D 32 x 32):  lane l supplies A row (l & 31), k = 8 (l >> 5) .. + 7, B column (l & 31), the
//   same k; it receives D column (l & 31), rows (r & 3) + 8 (r >> 2) + 4 (l >> 5) for r = 0..15.
// QK: A = K fragment (rows = kv of a 32-kv sub-tile), B = Q fragment (column = query).  Lane (q, h) ends up with the scores of query q for
//   the MFMA rows rho(r, h) = (r & 3) + 8 (r >> 2) + 4 h.
// PV: A = V^T fragment (rows = head dims), B = P^T (column = query, k = kv): lane (q, hh) must supply 8 consecutive k slots 8 hh .. 8 hh + 7 of
//   query q — the SAME lane that holds that query's scores.  Registers 8 t .. 8 t + 7 of the lane become its B operand of k-step t if MFMA row
//   rho(8 t + e, h) carries the kv with index 16 t + 8 h + e: that is rho with bits 2 and 3 swapped.  So the K fragment of lane (i, h) reads K row
//   swap23(i) of the sub-tile (a per-lane constant address) and nothing else changes: no v_permlane, no LDS round trip for P.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define A3_HD __host__ __device__ __forceinline__
#else
#define A3_HD inline
#endif

namespace mdx_a3 {

constexpr int KV = 64;                       // kv per tile (two 32-kv sub-tiles)

A3_HD int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
// K row (inside a 32-kv sub-tile) whose 16 bytes lane `lane` loads as its A fragment
A3_HD int k_row(int lane) { return swap23(lane & 31); }
// kv index (inside the sub-tile) of score register r of lane `lane`
A3_HD int s_kv(int r, int lane) { return 16 * (r >> 3) + 8 * (lane >> 5) + (r & 7); }
// MFMA C/D row of register r
A3_HD int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
// V^T tile (rows = head dims incl. pad rows, 128 bytes = 64 kv per row, 16-byte slots XOR-swizzled by (row >> 1) & 7): byte offset of the
// fragment lane `lane` reads for row tile rt (32 rows), sub-tile s, k-step t: row 32 rt + (lane & 31), kv chunk 4 s + 2 t + (lane >> 5)
A3_HD int vt_swz(int row) { return (row >> 1) & 7; }
A3_HD int vt_off(int rt, int s, int t, int lane) {
    const int row = 32 * rt + (lane & 31);
    const int chunk = 4 * s + 2 * t + (lane >> 5);
    return row * 128 + ((chunk ^ vt_swz(row)) << 4);
}

}  // namespace mdx_a3
