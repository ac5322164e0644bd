// attention2.hip — the lean fused attention forward for the UNet's big token counts (head dim 40 / 80, Tq >= 256) on gfx950.
//
// Why a second kernel: the round-2 counters of attention.hip (profiles/README.md) show it issue-bound, not matrix- or memory-bound:
// ~500 instructions per (wave, 64-kv tile) of which ~140 are the algorithm (14 MFMAs, 33 exp, 32 fma, 16 max, 16 cvt, 14 LDS
// reads); the rest is staging predication, per-tile address arithmetic, kv masks and register glue around the 8-byte V^T reads —
// and with all of its MFMAs / exps removed the kernel runs at the same speed.  This kernel keeps the algorithm (S^T = K Q^T on
// 32x32x16 MFMAs so a lane owns one query: lane-local online softmax, fp32 statistics, exp2 with folded scale; two-source
// "cross-view" mode with separate softmax per neighbour, summed) and removes the overhead:
//   * K and V^T tiles go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, inline asm as in gemm_xl.hip): no staging registers,
//     no ds_write pass, no per-chunk predicates — rows / kv chunks past the end are lanes whose voffset is out of range (zero fill);
//     three LDS buffers, tiles issued two ahead, ONE counted vmcnt and ONE barrier per tile;
//   * every per-lane address (DMA source offsets, fragment read offsets) is computed once per kernel;
//   * O^T += V^T P^T runs on 16x16x32 MFMAs: the head dim pads to a multiple of 16 instead of 32 (d = 40: 48 rows instead of 64,
//     -25 % PV work) and V^T fragments are single conflict-free ds_read_b128 (XOR-swizzled 128-byte rows, as in xl_layout.h);
//     P^T moves from the 32x32 accumulator layout to the 16x16 B-operand layout with 4 v_permlane32_swap + 4 v_permlane16_swap
//     per 32 kv (see p_to_b_operand);
//   * the row sums l = sum_kv p come out of the PV MFMAs through a row of ones in the padded V^T tile (d % 16 != 0);
//   * the kv mask (-inf for kv >= Tk) and the V^T pad-column scrub exist only in a peeled copy of the loop body for the last tile.
// Numerics: identical to attention.hip except that l is accumulated from the bf16-rounded probabilities (the same values the
// numerator uses) instead of the fp32 ones.
//
// Replaces xformers' CUTLASS fMHA as called by XFormersAttnProcessor (diffusers/models/attention_processor.py:1165-1171), incl.
// MagicDrive's cross-view attention (magicdrive/networks/blocks.py:106-222); same C entry point (mdx_attention_bf16, include/mdx.h).
#include "common.h"
#include "launch.h"
#include "options.h"
#include "xl_layout.h"
#include "attn2.h"
#include "attn3_layout.h"

namespace mdx {


typedef __attribute__((ext_vector_type(4))) unsigned a2_rsrc_t;
typedef __attribute__((address_space(3))) void a2_lds_t;
constexpr unsigned A2_OOB = 0x80000000u, A2_RECORDS = 0x80000000u;
#ifndef A2_RING
#define A2_RING 3          // LDS ring depth: tiles g .. g + A2_RING - 2 are in flight / in use while tile g is multiplied
#endif
constexpr int A2_KV = 64, A2_NW = 4, A2_NT = 256, A2_NBUF = A2_RING;

__device__ __forceinline__ a2_rsrc_t a2_make_rsrc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    a2_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = A2_RECORDS;
    r.w = 0x00020000u;
    return r;
}
// one 1-KiB LDS-DMA piece (see xl_glds in gemm_xl.hip for why this is inline asm and how completion is counted)
__device__ __forceinline__ void a2_glds(const a2_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    unsigned keep;
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}
template <int N>
__device__ __forceinline__ void a2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// P^T from the 32x32x16 accumulator layout to two 16x16x32 B operands, in place.
// In: the lane's 16 probabilities of one 32-kv sub-tile packed in pairs pk[u], u = 0..7: lane (q = l & 31, h = l >> 5) holds
//     kv 8 (u >> 1) + 2 (u & 1) + {0, 1} + 4 h of the sub-tile, i.e. {0-3, 8-11, 16-19, 24-27} + 4 h — stored as
//     x = (pk0, pk1, pk4, pk5), y = (pk2, pk3, pk6, pk7), so that every swap below exchanges like components of x and y (or two
//     components of the same vector) and the results ARE the operands: no register moves.
// Out: x / y = B operands (k = kv 32, n = q 16) of query tiles q = 0..15 / 16..31: lane (n = l & 15, c = l >> 4) holds the 8
//     consecutive kv 8c .. 8c+7 of query 16 t + n.
// Step 1 (v_permlane32_swap, lanes l <-> l + 32: the two halves of one query): pk[u].upper <-> pk[u + 4].lower, u = 0..3: the lower
//     lane collects kv 0..15, the upper 16..31.  Now chunk A = (pk0, pk1, pk4, pk5) = kv 0..7 (lower lanes) / 16..23 (upper),
//     chunk B = (pk2, pk3, pk6, pk7) = kv 8..15 / 24..31.
// Step 2 (v_permlane16_swap, lanes l <-> l ^ 16): a lane keeps the chunk of its own 16-lane row for one query tile and trades the
//     other chunk for the one its row needs of the other query tile.
__device__ __forceinline__ void p_to_b_operand(Frag8& x, Frag8& y) {
    { auto r = __builtin_amdgcn_permlane32_swap(x.u.x, x.u.z, false, false); x.u.x = r[0]; x.u.z = r[1]; }   // pk0 <-> pk4
    { auto r = __builtin_amdgcn_permlane32_swap(x.u.y, x.u.w, false, false); x.u.y = r[0]; x.u.w = r[1]; }   // pk1 <-> pk5
    { auto r = __builtin_amdgcn_permlane32_swap(y.u.x, y.u.z, false, false); y.u.x = r[0]; y.u.z = r[1]; }   // pk2 <-> pk6
    { auto r = __builtin_amdgcn_permlane32_swap(y.u.y, y.u.w, false, false); y.u.y = r[0]; y.u.w = r[1]; }   // pk3 <-> pk7
    { auto r = __builtin_amdgcn_permlane16_swap(x.u.x, y.u.x, false, false); x.u.x = r[0]; y.u.x = r[1]; }   // pk0 <-> pk2
    { auto r = __builtin_amdgcn_permlane16_swap(x.u.y, y.u.y, false, false); x.u.y = r[0]; y.u.y = r[1]; }   // pk1 <-> pk3
    { auto r = __builtin_amdgcn_permlane16_swap(x.u.z, y.u.z, false, false); x.u.z = r[0]; y.u.z = r[1]; }   // pk4 <-> pk6
    { auto r = __builtin_amdgcn_permlane16_swap(x.u.w, y.u.w, false, false); x.u.w = r[0]; y.u.w = r[1]; }   // pk5 <-> pk7
}

// A2_ABL: compile-time ablation bits for timing experiments only (tools/attn_ablate.sh builds side libraries with -DA2_ABL=..;
// results are WRONG with any bit set): 1 no exp, 2 no operand permute, 4 no rescale, 8 no QK MFMA, 16 no PV MFMA, 64 no steady-state DMA
#ifndef A2_ABL
#define A2_ABL 0
#endif
constexpr float A2_DEFER = 4.0f;   // log2 units: the running max is only raised (and O rescaled) when a tile exceeds it by more than this

// D8 = d / 8 (5 or 10); QT = 32-query tiles per wave; one workgroup = 4 waves = 128 QT queries of one (batch, head).
// QT = 2 halves, per query, everything that is per (workgroup, kv tile): the DMA pieces, the K / V^T fragment reads, the barrier —
// and the per-workgroup prologue / epilogue; it runs at 2 waves per SIMD (<= 256 VGPRs).
// FOLD (round 3; needs a spare k slot, d % 16 != 0, and Q pre-scaled by scale * log2(e) — MdxAttnDesc.q_prescaled): the running maximum
// is subtracted INSIDE the QK MFMA.  The padded k slot d of the Q fragment carries -m (bf16) and the same slot of the K fragment is
// forced to 1, so the accumulator comes out as s - m in base-2 units and feeds v_exp_f32 directly: the 32 v_fma_f32 per (32 queries x
// 64 kv) of `exp2(s * scale - m)` — a fifth of the kernel's VALU time, and the kernel is VALU-bound — disappear.  m is kept on the bf16
// grid (any value works as the subtracted maximum as long as numerator and row sum use the same one); a tile that raises it (rare:
// deferred maximum) re-bases its own scores with an explicit subtraction.
// RES (round 4): "resident" K / V^T for short kv sequences — the text-context attention of level 0 (S = 1 + 77 (+ boxes) tokens, Tq = 1400).
// With one workgroup per (view, head, 128-query block) that launch spent more than half of its time on per-workgroup fixed cost (zeroing
// the LDS image, the DMA round trip of two tiles, launch / drain): 168 TFLOP/s.  Here ONE workgroup per (view, head) stages all
// ntile <= A2_RING tiles once and then walks the query blocks: no DMA, no barrier and no vmcnt wait inside the walk.
// PF (round 5): "permute-free" P — O^T += V^T P^T on 32x32x16 MFMAs like S^T = K Q^T, with the K fragment of lane (i, h) reading K row swap23(i)
// of its 32-kv sub-tile (attn3_layout.h; replayed on the host by tests/attn3_layout_check.cpp): the 16 scores a lane receives for its query are
// then, in register order, the two 8-wide B operands the PV MFMAs want from that very lane.  Gone: the 16 v_permlane swaps per tile (a fifth
// of the VALU issue time of a kernel that is VALU-bound) and the cross-lane fetch of alpha in the rescale (O lives in the lane of its query);
// the price: the head dim pads to two 32-row tiles (rows 48..63 alias a zero row of the 48-row V^T tile), 8 MFMAs of 32 cycles per tile instead
// of 12 of 16.  The same idea as attention3.hip, without its software pipeline (which did not pay: DESIGN.md section 6).
template <int D8, bool TWO, int QT, bool FOLD, bool RES = false, bool PF = false>
#ifndef A2_LB4_TWO
#define A2_LB4_TWO 1       // 4 waves per SIMD also for the 32-query cross-view form (128 VGPRs, 9 spilled outside the tile loop): 4147-4215 us vs 4238 at
                           // 3 waves and 4253 with 64-query waves (576 views, profiles/r04_attn_xview_lb4_ab.log); -DA2_LB4_TWO=0 restores 3 waves
#endif
__global__ __launch_bounds__(A2_NT, (QT == 2 || D8 > 5) ? 2 : ((RES || (A2_LB4_TWO && TWO && !PF)) ? 4 : 3)) void attn2_kernel(Attn2Params p) {
    static_assert(!RES || (!TWO && QT == 1), "resident K / V^T: one kv source, 32-query waves");
    static_assert(!PF || (QT == 1 && D8 == 5 && !RES), "permute-free P: head dim 40, 32-query waves, streaming form");
    constexpr int D = D8 * 8;
    static_assert(!FOLD || (D % 16) == 8, "FOLD needs the 8 spare k slots of a head dim that is 8 mod 16");
    constexpr int D16 = (D + 15) / 16;             // QK k-steps of 16 and PV row tiles of 16
    constexpr int KROW = D * 2;                    // K tile row bytes (contiguous rows: the DMA image is lane-linear)
    constexpr int K_BYTES = A2_KV * KROW;          // d = 40: 5120
    constexpr int VROWS = D16 * 16;                // V^T tile rows incl. the pad rows
    constexpr int V_BYTES = VROWS * 128;           // 128 bytes = 64 kv per row, 16-byte slots XOR-swizzled by (row >> 1) & 7
    constexpr int PAD_TAIL = 64;                   // the last K row's QK reads run past it when d % 16 != 0: keep them inside the buffer
    constexpr int BUF = K_BYTES + PAD_TAIL + V_BYTES;
    constexpr bool ONES = (D % 16) != 0;           // a spare V^T row of ones carries the row sums
    constexpr int KP = D8, VP = D8;                // 1-KiB pieces per tile: K 64 * D8 chunks, V^T D rows * 8 chunks
    constexpr int PPW = (KP + VP + A2_NW - 1) / A2_NW;     // most pieces a wave issues per tile (some waves issue one fewer)
    constexpr int SCRATCH = A2_NBUF * BUF;         // (+ 1 KiB of slack behind the ring)
    constexpr int QW = 32 * QT;                    // queries per wave
    static_assert(SCRATCH + 1024 <= 65536 || A2_RING > 3, "LDS budget: d = 40 -> 35 KB (4 workgroups per CU), d = 80 -> 61 KB (2)");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SCRATCH + 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    int qb, h, b;
    {   // XCD-aware order (workgroup L runs on XCD L % 8: speed only).  viewmap 0 (rounds 2-3): the query blocks of one (batch, head) on one
        // XCD, back to back — its K / V^T stay in that L2, but the eight heads of a view land on eight XCDs, and a head's Q / K / O are
        // 80-byte slices of rows that all heads share: every L2 fetches (and partially writes back) the same 128-byte lines.  viewmap 1
        // (round 4): EVERY head and query block of a view on one XCD, head-major, so a line is fetched once per L2 and the heads' O
        // slices merge there before they are written back.
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
        const int nq = RES ? 1 : p.qblocks;
        if (p.viewmap) {
            const int per_view = p.H * nq;
            const int vl = idx / per_view, rem = idx - vl * per_view;
            b = vl * 8 + xcd;
            if (b >= p.B) return;
            h = rem / nq; qb = rem - h * nq;
        } else {
            const int bh = (idx / nq) * 8 + xcd;
            if (bh >= p.B * p.H) return;
            qb = idx % nq; b = bh / p.H; h = bh - b * p.H;
        }
    }
    int q0w = qb * (A2_NW * QW) + wave * QW;                 // this wave's first query
    // 32-query tiles of this wave that hold a real query (wave-uniform): the waves past Tq still stage and synchronise
    int nact = (p.Tq - q0w + 31) >> 5;
    nact = __builtin_amdgcn_readfirstlane(nact < 0 ? 0 : (nact > QT ? QT : nact));
    unsigned lds0 = (unsigned)(unsigned long long)(a2_lds_t*)smem;
    if (A2_ABL) asm volatile("" : "+s"(lds0));

    // ---- zero the LDS image once (pad rows / pad columns are never restaged), then the ones row ----
    for (int c = tid; c < (SCRATCH + 1024) / 16; c += A2_NT) *(uint4*)(smem + c * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (ONES) {
        for (int c = tid; c < A2_NBUF * 8; c += A2_NT) {       // row D of every buffer's V^T tile: 8 slots of 16 bytes (the swizzle permutes them within the row)
            const int bufi = c >> 3, slot = c & 7;
            *(uint4*)(smem + bufi * BUF + K_BYTES + PAD_TAIL + D * 128 + slot * 16) = make_uint4(MDX_ONE16 * 0x10001u, MDX_ONE16 * 0x10001u, MDX_ONE16 * 0x10001u, MDX_ONE16 * 0x10001u);
        }
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane -> query column, 8 consecutive dims ----
    Frag8 qf[QT][D16];
    auto load_q = [&]() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int q = q0w + qt * 32 + col;
            const bf16_t* qp = p.Q + (long)b * p.sQ + (long)(q < p.Tq ? q : 0) * p.ldq + (long)h * D;
#pragma unroll
            for (int ks = 0; ks < D16; ++ks) {
                const int dd = ks * 16 + half * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (q < p.Tq && dd < D) v = *(const uint4*)(qp + dd);
                qf[qt][ks].u = v;
            }
        }
        // The compiler cannot see the hand-counted LDS-DMA traffic on the VM counter: left alone it guards the first in-loop use of the Q
        // registers with `s_waitcnt vmcnt(0)` on EVERY iteration (the loads are outside the loop), which drains the tile just issued and
        // puts its whole latency on the critical path.  Consuming the registers here settles its bookkeeping before the loop.
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int ks = 0; ks < D16; ++ks) asm volatile("" : "+v"(qf[qt][ks].u.x), "+v"(qf[qt][ks].u.y), "+v"(qf[qt][ks].u.z), "+v"(qf[qt][ks].u.w));
    };
    if (!RES) load_q();

    // ---- DMA bookkeeping (round 4): piece pc = wave + j * A2_NW of the tile's NP = KP + VP pieces (K pieces first, then V^T); a wave issues
    // PPW or PPW - 1 of them (d = 40: 10 pieces over 4 waves = 3, 3, 2, 2 — round 3 padded every wave to 3 with dummy pieces into a scratch
    // row).  Everything a piece needs is resolved HERE, once: round 3's per-tile "kind" selects were wave-uniform values the compiler kept
    // in SGPRs and tested with scalar branches — ~180 scalar instructions, 16 branches and 34 v_readlane (spilled SGPRs) per tile per wave
    // in the ISA of the self-attention kernel, around three DMA instructions.  Now a tile's issue is straight-line: per piece one s_mul
    // (tile offset), one s_add (LDS address), one v_cndmask (the last tile's row / kv bound, pre-computed per lane) and the DMA statement.
    constexpr int NP = KP + VP;
    unsigned pv_off[PPW];          // per-lane byte offset inside the (b, h) K or V^T matrix at tile 0
    unsigned pv_last[PPW];         // the same for the LAST tile of a source: A2_OOB where the lane's row / kv chunk lies past Tk (zero fill)
    int p_lds[PPW];                // wave-uniform LDS byte offset inside a buffer
    int p_step[PPW];               // wave-uniform source bytes per tile: K 64 rows, V^T 64 kv
    bool p_isk[PPW];               // wave-uniform: piece of K (else of V^T)
    const int ntile = (p.Tk + A2_KV - 1) / A2_KV;
    const int tk_last = p.Tk - (ntile - 1) * A2_KV;          // valid kv of a source's last tile (1..64)
    const bool full_wave = wave + (PPW - 1) * A2_NW < NP;    // this wave issues PPW pieces per tile (else PPW - 1)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = min(wave + j * A2_NW, NP - 1);          // (clamped: a wave without piece j never issues it)
        if (pc < KP) {
            const int n = pc * 64 + lane;                      // linear 16-byte chunk of the K tile
            const int row = n / D8, ch = n - row * D8;
            p_isk[j] = true; p_lds[j] = pc * 1024; p_step[j] = A2_KV * (int)p.ldk * 2;
            pv_off[j] = (unsigned)((row * p.ldk + ch * 8) * 2);
            pv_last[j] = row < tk_last ? pv_off[j] : A2_OOB;
        } else {
            const int row0 = (pc - KP) * 8;
            const int row = mdx_xl::piece_lane_row(row0, lane), ch = mdx_xl::piece_lane_chunk(row0, lane);
            p_isk[j] = false; p_lds[j] = K_BYTES + PAD_TAIL + row0 * 128; p_step[j] = A2_KV * 2;
            pv_off[j] = (unsigned)((row * p.ldv + ch * 8) * 2);
            pv_last[j] = ch * 8 < tk_last ? pv_off[j] : A2_OOB;
        }
        p_lds[j] = __builtin_amdgcn_readfirstlane(p_lds[j]);
        p_step[j] = __builtin_amdgcn_readfirstlane(p_step[j]);
    }
    a2_rsrc_t rs_p[PPW];           // the descriptor each piece loads through (K or V^T of the current source)
    auto set_source = [&](int sidx) {
        const int bkv = p.kvmap ? p.kvmap[b * p.nsrc + sidx] : b;
        const a2_rsrc_t rsK = a2_make_rsrc(p.K + (long)bkv * p.sK + (long)h * D);
        const a2_rsrc_t rsV = a2_make_rsrc(p.Vt + (long)bkv * p.sV + (long)h * D * p.ldv);
#pragma unroll
        for (int j = 0; j < PPW; ++j) rs_p[j] = p_isk[j] ? rsK : rsV;
    };
    const int total = ntile * p.nsrc;
    // issue tile `t` of the current source into ring slot `slot`
    auto issue = [&](int t, int slot) {
        const bool last = t == ntile - 1;                      // wave-uniform
        const unsigned dst = lds0 + slot * BUF;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (j == PPW - 1 && !full_wave) break;             // wave-uniform: the one scalar branch of the issue path
            const unsigned vo = last ? pv_last[j] : pv_off[j];
            a2_glds(rs_p[j], dst + p_lds[j], vo, t * p_step[j]);
        }
    };

    // ---- fragment read offsets ----
    int k_rd[2];                                               // S^T sub-tile s: K rows s * 32 + col, this half's 16 bytes of k-step 0
#pragma unroll
    for (int s = 0; s < 2; ++s) k_rd[s] = (s * 32 + (PF ? mdx_a3::k_row(lane) : col)) * KROW + half * 16;
    // PF: V^T fragments of the 32x32x16 PV MFMAs: row tile 0 = rows col, (s, t) = st >> 1, st & 1; row tile 1 = rows 32 + col — only 32..47 exist
    // in the 48-row tile (40 = the ones row, 41..47 zero): the lanes of rows 48..63 all read zero row 47
    int vpf[PF ? 2 : 1][PF ? 4 : 1];
    if (PF) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            vpf[0][PF ? st : 0] = K_BYTES + PAD_TAIL + mdx_a3::vt_off(0, st >> 1, st & 1, lane);
            vpf[PF ? 1 : 0][PF ? st : 0] = col < 16 ? K_BYTES + PAD_TAIL + mdx_a3::vt_off(1, st >> 1, st & 1, lane) : K_BYTES + PAD_TAIL + 47 * 128;
        }
    }
    f32x16_t oaccp[PF ? 2 : 1];                                  // PF: O^T rows 32 rt + acc_row(r, lane) of query `col`
#pragma unroll
    for (int rt = 0; rt < (PF ? 2 : 1); ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oaccp[rt][r] = 0.f;
    const int v_rd0 = K_BYTES + PAD_TAIL + mdx_xl::frag_off(0, lane, 0), v_rd1 = K_BYTES + PAD_TAIL + mdx_xl::frag_off(0, lane, 1);

    f32x4_t oacc[QT][D16][2];
    // cross-view: the first neighbour's normalised output waits here, rounded to bf16 (the reference adds two fp16 attention outputs,
    // blocks.py:213-217) — half the registers of an fp32 copy, which is what keeps the 64-query kernel spill-free
    unsigned osum[TWO ? QT : 1][TWO ? D16 : 1][2][2];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = FOLD ? 0.f : -INFINITY; l_run[qt] = 0.f;      // FOLD: the subtracted maximum starts at 0 (the Q pad slot is zero); the first tile of a source always re-bases
#pragma unroll
        for (int i = 0; i < D16; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    oacc[qt][i][t][e] = 0.f;
                }
    }
    const unsigned one_slot = half ? MDX_ONE16 : 0u;          // FOLD: K fragment of the pad k-step for the upper half lanes = (1, 0, ..., 0)

    // One kv tile for the first NQ_ query tiles of the wave: scores, online softmax, PV.  The K / V^T fragments are read once and used
    // by every query tile.  LAST adds the kv >= Tk mask (the V^T pad columns were scrubbed by the caller).
    // Online softmax with a deferred maximum: m_run is raised only when a tile exceeds it by more than A2_DEFER (wave-uniform
    // decision), so most tiles skip the O rescale and its two cross-lane fetches.  Until then probabilities are relative to the older
    // maximum (at most 2^A2_DEFER instead of 1) — the numerator and the row sum carry the same factor and it cancels in O = (P V) / l.
#define A2_TILE(LAST, NQ_, slot_, j0_, FIRST_, NS_)                                                                        \
    {                                                                                                                  \
        const unsigned char* sb_ = smem + (slot_) * BUF;                                                               \
        Frag8 kf_[2][D16], vf_[2][D16];                                                                                \
        _Pragma("unroll") for (int s = 0; s < (NS_); ++s)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < D16; ++ks) kf_[s][ks].u = *(const uint4*)(sb_ + k_rd[s] + ks * 32); \
        _Pragma("unroll") for (int s = 0; s < (NS_); ++s)                                                                  \
            _Pragma("unroll") for (int i = 0; i < D16; ++i) vf_[s][i].u = *(const uint4*)(sb_ + (s ? v_rd1 : v_rd0) + i * 2048); \
        if (FOLD) {   /* upper-half lanes of the last k-step hold only pad dims (bytes of the NEXT K row in LDS): make them (1, 0 x 7) */ \
            _Pragma("unroll") for (int s = 0; s < (NS_); ++s) {                                                            \
                kf_[s][D16 - 1].u.x = half ? one_slot : kf_[s][D16 - 1].u.x; kf_[s][D16 - 1].u.y = half ? 0u : kf_[s][D16 - 1].u.y; \
                kf_[s][D16 - 1].u.z = half ? 0u : kf_[s][D16 - 1].u.z; kf_[s][D16 - 1].u.w = half ? 0u : kf_[s][D16 - 1].u.w; \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int qt = 0; qt < NQ_; ++qt) {                                                           \
            f32x16_t sacc[2];                                                                                          \
            _Pragma("unroll") for (int s = 0; s < (NS_); ++s) {                                                            \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) sacc[s][r] = 0.f;                                       \
                _Pragma("unroll") for (int ks = 0; ks < D16; ++ks) {                                                   \
                    if (A2_ABL & 8) sacc[s][ks] += __uint_as_float(kf_[s][ks].u.x ^ qf[qt][ks].u.x);                   \
                    else sacc[s] = MDX_MFMA_32x32x16(kf_[s][ks].v, qf[qt][ks].v, sacc[s]); \
                }                                                                                                      \
            }                                                                                                          \
            if (LAST) {                                                                                                \
                _Pragma("unroll") for (int s = 0; s < (NS_); ++s)                                                          \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                     \
                        if ((j0_) + s * 32 + mfma32_row(r, lane) >= p.Tk) sacc[s][r] = -INFINITY;                      \
            }                                                                                                          \
            float mx = -INFINITY;                                                                                      \
            _Pragma("unroll") for (int s = 0; s < (NS_); ++s)                                                              \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[s][r]);                             \
            {   /* max with the other half of this query (lane ^ 32): one v_permlane32_swap instead of a ds_bpermute */  \
                const unsigned mu_ = __float_as_uint(mx);                                                              \
                auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                                   \
                mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                          \
                if (!FOLD) mx *= p.scale_log2;                                                                         \
            }                                                                                                          \
            float alpha = 1.0f;                                                                                        \
            if (FOLD) {                                                                                                \
                /* scores are already relative to m_run; raise it when a tile exceeds it by A2_DEFER — and always on a source's first tile */ \
                if (((FIRST_) || __builtin_amdgcn_ballot_w64(mx > A2_DEFER) != 0) && !(A2_ABL & 4)) {                   \
                    const float inc_ = (FIRST_) ? mx : fmaxf(mx, 0.f);                                                 \
                    const float m_new = bf2f((bf16_t)(pack2bf(m_run[qt] + inc_, 0.f) & 0xffffu));   /* on the 16-bit grid */ \
                    const float delta_ = m_new - m_run[qt];                                                            \
                    alpha = (FIRST_) ? 0.f : __builtin_amdgcn_exp2f(-delta_);      /* first tile: O is zero; exp2 of a large -delta would be inf */ \
                    m_run[qt] = m_new;                                                                                 \
                    if (half) qf[qt][D16 - 1].u.x = pack2bf(-m_new, 0.f);          /* -m into the pad k slot of this query */ \
                    _Pragma("unroll") for (int s = 0; s < (NS_); ++s)                                                      \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) sacc[s][r] -= delta_;      /* this tile was multiplied with the old m */ \
                    const float al0 = __shfl(alpha, lane & 15, 64), al1 = __shfl(alpha, 16 + (lane & 15), 64);          \
                    _Pragma("unroll") for (int i = 0; i < D16; ++i)                                                    \
                        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
                            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oacc[qt][i][0][e]) : "v"(al0));                 \
                            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oacc[qt][i][1][e]) : "v"(al1));                 \
                        }                                                                                              \
                }                                                                                                      \
            } else                                                                                                     \
            if (__builtin_amdgcn_ballot_w64(mx > m_run[qt] + A2_DEFER) != 0 && !(A2_ABL & 4)) {                        \
                const float m_new = fmaxf(m_run[qt], mx);                                                              \
                alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);           /* first tile: exp2(-inf) = 0 on zeros */ \
                m_run[qt] = m_new;                                                                                     \
                /* the accumulators are in the 16x16 layout (lane -> query 16 t + (lane & 15)): fetch that query's alpha */ \
                const float al0 = __shfl(alpha, lane & 15, 64), al1 = __shfl(alpha, 16 + (lane & 15), 64);              \
                /* in place (asm): as plain C the rare arm gets its own result registers and the common arm pays 24 copies */ \
                _Pragma("unroll") for (int i = 0; i < D16; ++i)                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oacc[qt][i][0][e]) : "v"(al0));                     \
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oacc[qt][i][1][e]) : "v"(al1));                     \
                    }                                                                                                  \
            }                                                                                                          \
            const float mneg_ = -m_run[qt];                                                                            \
            float psum = 0.f;                                                                                          \
            _Pragma("unroll") for (int s = 0; s < (NS_); ++s) {                                                            \
                Frag8 b0, b1;                                                                                          \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
                    float p0 = FOLD ? sacc[s][2 * u] : __builtin_fmaf(sacc[s][2 * u], p.scale_log2, mneg_);            \
                    float p1 = FOLD ? sacc[s][2 * u + 1] : __builtin_fmaf(sacc[s][2 * u + 1], p.scale_log2, mneg_);    \
                    if (!(A2_ABL & 1)) { p0 = __builtin_amdgcn_exp2f(p0); p1 = __builtin_amdgcn_exp2f(p1); }           \
                    if (!ONES) psum += p0 + p1;                                                                        \
                    const unsigned pk_ = pack2bf(p0, p1);                                                              \
                    /* x = (pk0, pk1, pk4, pk5), y = (pk2, pk3, pk6, pk7): see p_to_b_operand */                        \
                    if (u == 0) b0.u.x = pk_; if (u == 1) b0.u.y = pk_; if (u == 4) b0.u.z = pk_; if (u == 5) b0.u.w = pk_; \
                    if (u == 2) b1.u.x = pk_; if (u == 3) b1.u.y = pk_; if (u == 6) b1.u.z = pk_; if (u == 7) b1.u.w = pk_; \
                }                                                                                                      \
                if (!(A2_ABL & 2)) p_to_b_operand(b0, b1);                                                             \
                _Pragma("unroll") for (int i = 0; i < D16; ++i) {                                                      \
                    if (A2_ABL & 16) { oacc[qt][i][0][0] += __uint_as_float(vf_[s][i].u.x ^ b0.u.x ^ b0.u.y ^ b0.u.z ^ b0.u.w); oacc[qt][i][1][0] += __uint_as_float(vf_[s][i].u.y ^ b1.u.x ^ b1.u.y ^ b1.u.z ^ b1.u.w); } \
                    else {                                                                                             \
                        oacc[qt][i][0] = MDX_MFMA_16x16x32(vf_[s][i].v, b0.v, oacc[qt][i][0]); \
                        oacc[qt][i][1] = MDX_MFMA_16x16x32(vf_[s][i].v, b1.v, oacc[qt][i][1]); \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
            if (!ONES) l_run[qt] = l_run[qt] * alpha + psum;                                                           \
        }                                                                                                              \
    }
    // The permute-free tile (PF; one 32-query tile per wave): scores as in A2_TILE (K rows permuted), online softmax with the rescale lane-local,
    // then per 32-kv sub-tile and k-step the packed probabilities ARE the B operand of two 32x32x16 PV MFMAs.
#define A2_TILE_PF(LAST, slot_, j0_, FIRST_)                                                                           \
    {                                                                                                                  \
        const unsigned char* sb_ = smem + (slot_) * BUF;                                                               \
        Frag8 kf_[2][D16];                                                                                             \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                  \
            _Pragma("unroll") for (int ks = 0; ks < D16; ++ks) kf_[s][ks].u = *(const uint4*)(sb_ + k_rd[s] + ks * 32); \
        if (FOLD) {                                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                            \
                kf_[s][D16 - 1].u.x = half ? one_slot : kf_[s][D16 - 1].u.x; kf_[s][D16 - 1].u.y = half ? 0u : kf_[s][D16 - 1].u.y; \
                kf_[s][D16 - 1].u.z = half ? 0u : kf_[s][D16 - 1].u.z; kf_[s][D16 - 1].u.w = half ? 0u : kf_[s][D16 - 1].u.w; \
            }                                                                                                          \
        }                                                                                                              \
        f32x16_t sacc[2];                                                                                              \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) sacc[s][r] = 0.f;                                           \
            _Pragma("unroll") for (int ks = 0; ks < D16; ++ks) sacc[s] = MDX_MFMA_32x32x16(kf_[s][ks].v, qf[0][ks].v, sacc[s]); \
        }                                                                                                              \
        if (LAST) {                                                                                                    \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                              \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                         \
                    if ((j0_) + s * 32 + mdx_a3::s_kv(r, lane) >= p.Tk) sacc[s][r] = -INFINITY;                        \
        }                                                                                                              \
        float mx = -INFINITY;                                                                                          \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[s][r]);                                 \
        {                                                                                                              \
            const unsigned mu_ = __float_as_uint(mx);                                                                  \
            auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                                       \
            mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                              \
            if (!FOLD) mx *= p.scale_log2;                                                                             \
        }                                                                                                              \
        if (FOLD) {                                                                                                    \
            if ((FIRST_) || __builtin_amdgcn_ballot_w64(mx > A2_DEFER) != 0) {                                         \
                const float inc_ = (FIRST_) ? mx : fmaxf(mx, 0.f);                                                     \
                const float m_new = bf2f((bf16_t)(pack2bf(m_run[0] + inc_, 0.f) & 0xffffu));                           \
                const float delta_ = m_new - m_run[0];                                                                 \
                const float alpha_ = (FIRST_) ? 0.f : __builtin_amdgcn_exp2f(-delta_);                                 \
                m_run[0] = m_new;                                                                                      \
                if (half) qf[0][D16 - 1].u.x = pack2bf(-m_new, 0.f);                                                   \
                _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                          \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) sacc[s][r] -= delta_;                               \
                _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                       \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oaccp[PF ? rt : 0][r]) : "v"(alpha_)); \
            }                                                                                                          \
        } else if (__builtin_amdgcn_ballot_w64(mx > m_run[0] + A2_DEFER) != 0) {                                       \
            const float m_new = fmaxf(m_run[0], mx);                                                                   \
            const float alpha_ = __builtin_amdgcn_exp2f(m_run[0] - m_new);                                             \
            m_run[0] = m_new;                                                                                          \
            _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                           \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oaccp[PF ? rt : 0][r]) : "v"(alpha_)); \
        }                                                                                                              \
        const float mneg_ = -m_run[0];                                                                                 \
        _Pragma("unroll") for (int st = 0; st < 4; ++st) {                                                             \
            Frag8 b_, v0_, v1_;                                                                                        \
            v0_.u = *(const uint4*)(sb_ + vpf[0][PF ? st : 0]); v1_.u = *(const uint4*)(sb_ + vpf[PF ? 1 : 0][PF ? st : 0]); \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
                float p0 = sacc[st >> 1][8 * (st & 1) + 2 * u], p1 = sacc[st >> 1][8 * (st & 1) + 2 * u + 1];          \
                if (!FOLD) { p0 = __builtin_fmaf(p0, p.scale_log2, mneg_); p1 = __builtin_fmaf(p1, p.scale_log2, mneg_); } \
                p0 = __builtin_amdgcn_exp2f(p0); p1 = __builtin_amdgcn_exp2f(p1);                                      \
                const unsigned pk_ = pack2bf(p0, p1);                                                                  \
                if (u == 0) b_.u.x = pk_; if (u == 1) b_.u.y = pk_; if (u == 2) b_.u.z = pk_; if (u == 3) b_.u.w = pk_; \
            }                                                                                                          \
            oaccp[0] = MDX_MFMA_32x32x16(v0_.v, b_.v, oaccp[0]);                                                       \
            oaccp[PF ? 1 : 0] = MDX_MFMA_32x32x16(v1_.v, b_.v, oaccp[PF ? 1 : 0]);                                      \
        }                                                                                                              \
    }
    if constexpr (RES) {
        // ---- resident K / V^T: stage every tile of the (one) source once, then walk the query blocks ----
        set_source(0);
        for (int t = 0; t < ntile; ++t) issue(t, t);              // ntile <= A2_NBUF (launch_attn2)
        a2_wait_vmcnt<0>();
        __syncthreads();
        const int nfull_r = p.Tk / A2_KV;
        if (nfull_r < ntile) {
            // scrub the V^T pad columns of the partial tile (kv >= Tk inside the last partially valid 16-byte chunk may hold anything)
            const int kv_lo = p.Tk - nfull_r * A2_KV;
            if ((kv_lo & 7) != 0) {
                const int chunk = kv_lo >> 3, e0 = kv_lo & 7;
                for (int r = tid; r < D; r += A2_NT) {
                    bf16_t* rowp = (bf16_t*)(smem + nfull_r * BUF + K_BYTES + PAD_TAIL + r * 128 + ((chunk ^ mdx_xl::swz(r)) << 4));
                    for (int e = e0; e < 8; ++e) rowp[e] = 0;
                }
                __syncthreads();
            }
        }
        const bool short_last = nfull_r < ntile && p.Tk - nfull_r * A2_KV <= 32;      // the partial tile fits its first 32-kv sub-tile
        const int nqb = (p.Tq + A2_NW * QW - 1) / (A2_NW * QW);
        for (int qbi = 0; qbi < nqb; ++qbi) {
            q0w = qbi * (A2_NW * QW) + wave * QW;
            if (q0w >= p.Tq) break;                                // wave-uniform; nothing synchronises inside the walk
            load_q();
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                m_run[qt] = FOLD ? 0.f : -INFINITY; l_run[qt] = 0.f;
#pragma unroll
                for (int i = 0; i < D16; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) oacc[qt][i][t][e] = 0.f;
            }
            int fresh_r = 1;
            for (int t = 0; t < nfull_r; ++t) {
                A2_TILE(false, 1, t, t * A2_KV, fresh_r, 2)
                fresh_r = 0;
            }
            if (nfull_r < ntile) {
                if (short_last) { A2_TILE(true, 1, nfull_r, nfull_r * A2_KV, fresh_r, 1) }
                else { A2_TILE(true, 1, nfull_r, nfull_r * A2_KV, fresh_r, 2) }
            }
            // normalise and store this block's O
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float inv0, inv1;
                if (ONES) {
                    constexpr int LT = ONES ? D / 16 : 0;
                    inv0 = 1.0f / __shfl(oacc[qt][LT][0][0], 32 + (lane & 15), 64);
                    inv1 = 1.0f / __shfl(oacc[qt][LT][1][0], 32 + (lane & 15), 64);
                } else {
                    const float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32, 64);
                    const float inv = 1.0f / l_tot;
                    inv0 = __shfl(inv, lane & 15, 64); inv1 = __shfl(inv, 16 + (lane & 15), 64);
                }
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) {
                    const int qq = q0w + qt * 32 + tq * 16 + (lane & 15);
                    if (qq >= p.Tq) continue;
                    bf16_t* op = p.O + (long)b * p.sO + (long)qq * p.ldo + (long)h * D;
                    const float inv = tq ? inv1 : inv0;
#pragma unroll
                    for (int i = 0; i < D16; ++i) {
                        const int dd = i * 16 + 4 * (lane >> 4);
                        if (dd < D) {
                            const f32x4_t& a = oacc[qt][i][tq];
                            uint2 ov;
                            ov.x = pack2bf(a[0] * inv, a[1] * inv);
                            ov.y = pack2bf(a[2] * inv, a[3] * inv);
                            *(uint2*)(op + dd) = ov;
                        }
                    }
                }
                if (FOLD && half) qf[qt][D16 - 1].u.x = 0u;
            }
        }
        return;
    }

    // ---- prologue: tiles 0 and 1 of the stream in flight ----
    int ds = 0;                          // source the descriptors currently describe
    set_source(0);
    int it = 0, is_ = 0;                 // issue cursor: tile inside its source, source index
    auto issue_next = [&](int slot_) {
        if (is_ != ds) { set_source(is_); ds = is_; }
        issue(it, slot_);
        if (++it == ntile) { it = 0; ++is_; }
    };
    issue_next(0);
#pragma unroll
    for (int k = 1; k < A2_NBUF - 1; ++k)
        if (total > k) issue_next(k);
    int g = 0;                           // tile of the whole stream (both sources) being multiplied
    int slot = 0;
    // start of tile g: its pieces have landed for every wave, every wave is done with tile g - 1, tile g + A2_NBUF - 1 goes out
    auto tile_sync = [&]() {
        // at most the A2_NBUF - 2 younger tiles' pieces outstanding (fewer at the end of the stream); a wave's tile is PPW or PPW - 1 pieces
        const int younger = total - 1 - g;
        if (younger >= A2_NBUF - 2) { if (full_wave) a2_wait_vmcnt<PPW * (A2_NBUF - 2)>(); else a2_wait_vmcnt<(PPW - 1) * (A2_NBUF - 2)>(); }
        else if (A2_NBUF > 3 && younger == 1) { if (full_wave) a2_wait_vmcnt<PPW>(); else a2_wait_vmcnt<PPW - 1>(); }
        else a2_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g + A2_NBUF - 1 < total && !(A2_ABL & 64)) {
            int s2 = slot + A2_NBUF - 1; if (s2 >= A2_NBUF) s2 -= A2_NBUF;
            issue_next(s2);
        }
    };
    int fresh = 1;                       // FOLD: the next tile is the first of a softmax (a source; all sources when joint): it always re-bases m
    auto tile_done = [&]() { ++g; if (++slot == A2_NBUF) slot = 0; fresh = 0; };
    const int nfull = p.Tk / A2_KV;      // full tiles per source; a partial one follows when Tk % 64 != 0

    // The loop nest of one source, for the first NQ_ query tiles of the wave.  The full tiles run in a loop with ONE body (so the
    // accumulators stay in place: with the masked body as a second arm of the same loop the compiler copies all of them every
    // iteration); the partial last tile is peeled behind it.
#define A2_SOURCE(NQ_)                                                                                                 \
    {                                                                                                                  \
        for (int t = 0; t < nfull; ++t) {                                                                              \
            tile_sync();                                                                                               \
            if constexpr (PF) { if (NQ_) A2_TILE_PF(false, slot, t * A2_KV, fresh) } else A2_TILE(false, NQ_, slot, t * A2_KV, fresh, 2)   \
            tile_done();                                                                                               \
        }                                                                                                              \
        if (nfull < ntile) {                                                                                           \
            tile_sync();                                                                                               \
            /* scrub the V^T pad columns (kv >= Tk inside the last partially valid 16-byte chunk may hold anything, and 0 * NaN is NaN) */ \
            const int kv_lo = p.Tk - nfull * A2_KV;             /* first invalid kv inside the tile (1..63) */          \
            if ((kv_lo & 7) != 0) {                                                                                    \
                const int chunk = kv_lo >> 3, e0 = kv_lo & 7;                                                          \
                for (int r = tid; r < D; r += A2_NT) {                                                                 \
                    bf16_t* rowp = (bf16_t*)(smem + slot * BUF + K_BYTES + PAD_TAIL + r * 128 + ((chunk ^ mdx_xl::swz(r)) << 4)); \
                    for (int e = e0; e < 8; ++e) rowp[e] = 0;                                                          \
                }                                                                                                      \
                __syncthreads();                                                                                       \
            }                                                                                                          \
            if constexpr (PF) { if (NQ_) A2_TILE_PF(true, slot, nfull * A2_KV, fresh) } else A2_TILE(true, NQ_, slot, nfull * A2_KV, fresh, 2) \
            tile_done();                                                                                               \
        }                                                                                                              \
    }

    for (int src = 0; src < p.nsrc; ++src) {
        if (QT == 2 && nact == 2) A2_SOURCE(QT)
        else if (nact >= 1) A2_SOURCE(1)
        else A2_SOURCE(0)
        // ---- end of a source: normalise, accumulate (cross-view), reset.  joint: the sources are one kv sequence — only after the last ----
        if (p.joint && src + 1 < p.nsrc) continue;
        if constexpr (PF) {
            // row 40 of O^T = sum of the (16-bit) probabilities: row tile 1, local row 8 -> register 4 of the lower half lanes
            const float inv = 1.0f / __shfl(oaccp[PF ? 1 : 0][4], col, 64);
            if (TWO && src == 0) {                                // first neighbour done: park it (packed), restart the accumulators
#pragma unroll
                for (int u = 0; u < 8; ++u) osum[0][u >> 2][(u >> 1) & 1][u & 1] = pack2bf(oaccp[0][2 * u] * inv, oaccp[0][2 * u + 1] * inv);
#pragma unroll
                for (int u = 0; u < 2; ++u) osum[0][TWO ? 2 : 0][0][u] = pack2bf(oaccp[PF ? 1 : 0][2 * u] * inv, oaccp[PF ? 1 : 0][2 * u + 1] * inv);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oaccp[PF ? rt : 0][r] = 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) oaccp[0][r] *= inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) oaccp[PF ? 1 : 0][r] *= inv;
                if (TWO) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned w = osum[0][u >> 2][(u >> 1) & 1][u & 1];
                        oaccp[0][2 * u] += bf2f((bf16_t)(w & 0xffffu)); oaccp[0][2 * u + 1] += bf2f((bf16_t)(w >> 16));
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned w = osum[0][TWO ? 2 : 0][0][u];
                        oaccp[PF ? 1 : 0][2 * u] += bf2f((bf16_t)(w & 0xffffu)); oaccp[PF ? 1 : 0][2 * u + 1] += bf2f((bf16_t)(w >> 16));
                    }
                }
            }
            m_run[0] = FOLD ? 0.f : -INFINITY;
            if (FOLD && half) qf[0][D16 - 1].u.x = 0u;
            fresh = 1;
            continue;
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float inv0, inv1;
            if (ONES) {
                // row D of O^T = sum of the (bf16) probabilities: tile D / 16, local row D % 16 = 8 -> lanes 32..47, register 0
                constexpr int LT = ONES ? D / 16 : 0;
                inv0 = 1.0f / __shfl(oacc[qt][LT][0][0], 32 + (lane & 15), 64);
                inv1 = 1.0f / __shfl(oacc[qt][LT][1][0], 32 + (lane & 15), 64);
            } else {
                const float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32, 64);          // per S-layout lane: query lane & 31
                const float inv = 1.0f / l_tot;
                inv0 = __shfl(inv, lane & 15, 64); inv1 = __shfl(inv, 16 + (lane & 15), 64);
            }
#pragma unroll
            for (int i = 0; i < D16; ++i)
#pragma unroll
                for (int tq = 0; tq < 2; ++tq) {
                    const float inv = tq ? inv1 : inv0;
                    f32x4_t& a = oacc[qt][i][tq];
                    if (TWO && src == 0) {                     // first neighbour done: park it, restart the accumulators
                        osum[TWO ? qt : 0][TWO ? i : 0][tq][0] = pack2bf(a[0] * inv, a[1] * inv);
                        osum[TWO ? qt : 0][TWO ? i : 0][tq][1] = pack2bf(a[2] * inv, a[3] * inv);
                        a[0] = 0.f; a[1] = 0.f; a[2] = 0.f; a[3] = 0.f;
                    } else {
                        a[0] *= inv; a[1] *= inv; a[2] *= inv; a[3] *= inv;
                        if (TWO) {
                            const unsigned u0 = osum[TWO ? qt : 0][TWO ? i : 0][tq][0], u1 = osum[TWO ? qt : 0][TWO ? i : 0][tq][1];
                            a[0] += bf2f((bf16_t)(u0 & 0xffffu)); a[1] += bf2f((bf16_t)(u0 >> 16));
                            a[2] += bf2f((bf16_t)(u1 & 0xffffu)); a[3] += bf2f((bf16_t)(u1 >> 16));
                        }
                    }
                }
            m_run[qt] = FOLD ? 0.f : -INFINITY; l_run[qt] = 0.f;
            if (FOLD && half) qf[qt][D16 - 1].u.x = 0u;
        }
        fresh = 1;
    }
#undef A2_SOURCE
#undef A2_TILE
    a2_wait_vmcnt<0>();

    if constexpr (PF) {
        // ---- store O[q][h * 40 + d]: lane (q, half) holds d = 8 u + 4 half + {0..3} of row tile 0 (u = 0..3) and d = 32 + 4 half + {0..3} ----
        const int qq = q0w + col;
        if (qq < p.Tq) {
            bf16_t* op = p.O + (long)b * p.sO + (long)qq * p.ldo + (long)h * D + 4 * half;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint2 ov;
                ov.x = pack2bf(oaccp[0][4 * u], oaccp[0][4 * u + 1]);
                ov.y = pack2bf(oaccp[0][4 * u + 2], oaccp[0][4 * u + 3]);
                *(uint2*)(op + 8 * u) = ov;
            }
            uint2 ov;
            ov.x = pack2bf(oaccp[PF ? 1 : 0][0], oaccp[PF ? 1 : 0][1]);
            ov.y = pack2bf(oaccp[PF ? 1 : 0][2], oaccp[PF ? 1 : 0][3]);
            *(uint2*)(op + 32) = ov;
        }
        return;
    }
    // ---- store O[q][h * d + dd]: 16x16 layout: lane -> query 16 t + (lane & 15), rows dd = 16 i + 4 (lane >> 4) + e ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int qq = q0w + qt * 32 + tq * 16 + (lane & 15);
            if (qq >= p.Tq) continue;
            bf16_t* op = p.O + (long)b * p.sO + (long)qq * p.ldo + (long)h * D;
#pragma unroll
            for (int i = 0; i < D16; ++i) {
                const int dd = i * 16 + 4 * (lane >> 4);
                if (dd < D) {
                    const f32x4_t& src = oacc[qt][i][tq];
                    uint2 ov;
                    ov.x = pack2bf(src[0], src[1]);
                    ov.y = pack2bf(src[2], src[3]);
                    *(uint2*)(op + dd) = ov;
                }
            }
        }
}

// resident K / V^T (attn2_kernel<.., RES>): one workgroup per (batch, head)
template <int D8, bool FOLD>
static int launch_attn2_res(const Attn2Params& p, hipStream_t st) {
    Attn2Params q = p;
    q.qblocks = 1;
    q.viewmap = opt(OPT_ATTN2_VIEWMAP) != 0;
    const dim3 grid(q.viewmap ? (unsigned)(((long)p.B + 7) / 8 * 8 * p.H) : (unsigned)(((long)p.B * p.H + 7) / 8 * 8), 1, 1);
    hipLaunchKernelGGL((attn2_kernel<D8, false, 1, FOLD, true>), grid, dim3(A2_NT), 0, st, q);
    char tag[64];
    snprintf(tag, sizeof tag, "attn2_kernel<%d,resident,q32%s>", D8 * 8, FOLD ? ",fold" : "");
    return check_launch(tag);
}

template <int D8, int QT, bool FOLD, bool PF = false>
static int launch_attn2_d(const Attn2Params& p, hipStream_t st) {
    Attn2Params q = p;
    q.qblocks = (p.Tq + A2_NW * 32 * QT - 1) / (A2_NW * 32 * QT);
    q.viewmap = opt(OPT_ATTN2_VIEWMAP) != 0;
    const dim3 grid(q.viewmap ? (unsigned)(((long)p.B + 7) / 8 * 8 * p.H * q.qblocks) : (unsigned)(((long)p.B * p.H + 7) / 8 * 8 * q.qblocks), 1, 1);
    const bool two = p.nsrc == 2 && !p.joint;                   // TWO = the summed two-neighbour form; everything else is one softmax over nsrc sources
    if (two) hipLaunchKernelGGL((attn2_kernel<D8, true, QT, FOLD, false, PF>), grid, dim3(A2_NT), 0, st, q);
    else hipLaunchKernelGGL((attn2_kernel<D8, false, QT, FOLD, false, PF>), grid, dim3(A2_NT), 0, st, q);
    char tag[64];
    snprintf(tag, sizeof tag, "attn2_kernel<%d,%s,q%d%s%s>", D8 * 8, two ? "xview" : (p.nsrc > 1 ? "joint" : "self"), 32 * QT, FOLD ? ",fold" : "", PF ? ",pf" : "");
    return check_launch(tag);
}

// Shapes the lean kernel takes (everything else stays on attention.hip): head dim 40 or 80, enough query blocks to fill the chip,
// 16-byte aligned K rows / V^T rows, matrices within the 2 GiB DMA window.
bool attn2_supported(const Attn2Params& p) {
    const int on = (int)opt(OPT_ATTN2);
    // d = 80 (level 1, T = 350): 0 never, 1 always, 2 (default) only the two-source cross-view form — measured at 768 views with the
    // round-4 issue path (profiles/r04_attn_d80.log): cross-view 923 vs 1000-1012 us on attention.hip, self 545 vs 517-529 us
    const int d80 = (int)opt(OPT_ATTN2_D80);
    const bool xview2 = p.nsrc == 2 && !p.joint;
    if (!on || (p.d != 40 && !(p.d == 80 && (d80 == 1 || (d80 == 2 && xview2))))) return false;
    if (p.Tq < 256 || (long)((p.Tq + 127) / 128) * p.H * p.B < 128) return false;
    if ((p.ldk % 8) || (p.ldv % 8) || (p.sK % 8) || (p.sV % 8) || (p.ldv < ((p.Tk + 7) / 8) * 8)) return false;
    if ((long)p.Tk * p.ldk * 2 >= 0x40000000L || (long)p.d * p.H * p.ldv * 2 >= 0x40000000L) return false;
    return true;
}

int launch_attn2(const Attn2Params& p, hipStream_t st) {
    // 64 queries per wave (see attn2_kernel) when there are enough queries per head; MDX_ATTN2_QT=1 forces 32
    const int qt_opt = (int)opt(OPT_ATTN2_QT);          // 0 (default): 64-query waves except where the 32-query FOLD form wins; 1 / 2: force
    const int qt = qt_opt == 0 ? 2 : qt_opt;
    const bool qt_forced64 = qt_opt == 2;
    const bool two = qt == 2 && p.Tq >= 512 && p.d == 40;
    // pre-scaled Q (scale * log2 e folded into to_q at pack time): scores are base-2 exponents as they come out of the MFMA
    const bool fold = p.q_prescaled && opt(OPT_ATTN2_FOLD) != 0;
    // short kv sequences (the text context): every tile resident in the ring, one workgroup per (batch, head) walks the query blocks
    // (ATTN2_RES: 1 = when the launch has enough (batch, head) pairs to fill the chip on its own and several query blocks to walk; 2 = whenever supported)
    const int res_opt = (int)opt(OPT_ATTN2_RES);
    if (p.d == 40 && fold && p.nsrc == 1 && (p.Tk + A2_KV - 1) / A2_KV <= A2_NBUF &&
        (res_opt == 2 || (res_opt == 1 && p.Tq >= 512 && (long)p.B * p.H >= 1024)))
        return launch_attn2_res<5, true>(p, st);
#ifdef MDX_ATTN3                                                  // tools/attn3/attention3.hip, only in `make ATTN3=1` builds (round 6: out of the product library)
    if (attn3_supported(p)) return launch_attn3(p, st);          // round 5: the software-pipelined, permute-free form
#endif
    if (p.d == 40) {
        // FOLD frees the registers / VALU slots of the scale-and-subtract: with it the 32-query form (<= 142 VGPRs: three waves per SIMD)
        // is the faster one for one kv source (768 views, T = 1400: self 3157 vs 3332 us, text context 640 vs 795 us; the two-source
        // cross-view form is equal, 5888 vs 5882 us: profiles/r03_attn_qt_fold_ab.log); ATTN2_QT = 2 keeps 64-query waves everywhere.
        // round 4: with the straight-line issue path and four waves per SIMD the 32-query form also wins (by 1-2.5 %) for the two-source launches
        if (fold && !(two && qt_forced64) && opt(OPT_ATTN2_PF)) return launch_attn2_d<5, 1, true, true>(p, st);      // round 5: permute-free P
        if (fold) return (two && qt_forced64) ? launch_attn2_d<5, 2, true>(p, st) : launch_attn2_d<5, 1, true>(p, st);
        return two ? launch_attn2_d<5, 2, false>(p, st) : launch_attn2_d<5, 1, false>(p, st);
    }
    return launch_attn2_d<10, 1, false>(p, st);
}

}  // namespace mdx
