// attention3.hip — the level-0 attention (head dim 40, 1400 tokens: self, cross-view, joint) with the softmax's VALU stream running in the
// shadow of the MFMAs, nothing waiting for the LDS, and one persistent workgroup per CU slot (round 5).
//
// Why a third kernel.  attention2.hip is VALU-bound and — the round-4 counters say — not overlapped at all: per (32 queries x 64 kv) wave
// tile 384 cycles of MFMA and ~560 cycles of VALU issue (32 v_exp_f32 + 17 lane permutes at 7.2 cycles, 16 conversions, 16 v_max3) add up
// to the measured 944-960 (`mfma_util` 0.40 = 384 / 944); and a fifth of its launch is per-workgroup fixed cost (see below).
//   * PERMUTE-FREE P.  O^T += V^T P^T runs on 32x32x16 MFMAs like S^T = K Q^T, and the K fragment of lane (i, h) reads K row swap23(i) of
//     its 32-kv sub-tile (attn3_layout.h): with that row order the 16 scores a lane receives for its query ARE, in register order, the two
//     8-wide B operands the PV MFMAs want from that lane.  v_exp -> v_cvt_pk -> MFMA, nothing in between.  The head dim pads to 64 rows
//     (two 32-row tiles) instead of 48: +64 MFMA cycles per tile for -17 quarter-rate lane permutes (-122 issue cycles) — the right trade
//     once the two streams overlap and the VALU is the longer one.  The running sums still come out of the PV MFMAs (a row of ones at
//     row 40), and the O accumulators live in the lane of their query: the (rare) rescale needs no cross-lane fetch.
//   * SOFTWARE PIPELINE OVER 32-KV SUB-TILES.  Two score sets of 16 registers: while sub-tile u's 16 exponentials and 8 conversions issue, the
//     3 QK MFMAs of sub-tile u + 1 and then the 4 PV MFMAs of u run beside them — 7 MFMAs x 32 cycles = 224 per step beside ~45 VALU
//     instructions, one MFMA per group of <= 6, the groups fenced with sched_barrier so the order survives the machine scheduler.  (The first
//     version pipelined whole 64-kv tiles: two 32-register score sets + all fragments of a tile put the kernel at 256 VGPRs with spills —
//     and a scratch reload is a memory round trip behind the DMA ring.)
//   * NOTHING WAITS FOR THE LDS.  Measured on the first version (profiles/r05_attn3_v1_ablation.log): with every MFMA and every exponential
//     compiled out the kernel lost only a third of its time — at 2 waves per SIMD the per-tile chain barrier -> ds_read -> wait -> use is
//     exposed, 420 cycles per wave tile.  So the fragments are prefetched a step ahead, in place: a K fragment register is refilled (for
//     sub-tile u + 2) two MFMAs after the QK MFMA that consumed it, a V^T fragment (u + 1) right after its PV MFMA; the barrier at the top
//     of tile x publishes tile x + 1 (whose K is read during tile x).  4-deep LDS ring: tile x + 2 in flight, x + 3 issued between the MFMAs.
//   * PERSISTENT WORKGROUPS.  The same measurement put the per-workgroup fixed cost (dispatch, LDS image, Q fetch, first DMA round trip, the
//     first unoverlapped QK, the O stores) at 430 us of a 2.1 ms launch of 50 688 workgroups — for both kernels.  Here 2 workgroups per CU
//     walk the (view, head, 128-query block) items of their XCD, view-major (the 64 workgroups of an XCD are on ~6 neighbouring (view, head)
//     pairs at any time: their K / V^T stay in that L2 — a workgroup that walked all blocks of ONE pair re-streamed K / V^T 11 times from
//     HBM: profiles/r05_attn3_v2_sweep.log).  The K / V^T stream simply continues across an item seam: the next item's first tiles are in
//     flight, its Q is fetched during the last tile, and the last step of an item already multiplies the next item's first scores.
// Requires the FOLD form (Q pre-scaled by scale * log2 e and the maximum subtracted inside the QK MFMA through the spare k slot 40:
// attention2.hip), which is what the level-0 projections produce (MdxAttnDesc.q_prescaled); everything else stays on attention2.hip.
// Numerics: identical to attention2.hip's FOLD form (bf16 / fp16 probabilities, fp32 accumulation, row sums from the rounded
// probabilities, deferred maximum on the 16-bit grid) except that the maximum is tracked per 32 kv instead of per 64.
//
// Replaces xformers' CUTLASS fMHA as called by XFormersAttnProcessor (diffusers/models/attention_processor.py:1165-1171) incl. MagicDrive's
// cross-view attention (magicdrive/networks/blocks.py:106-222); same C entry point (mdx_attention_bf16, include/mdx.h).
#include "common.h"
#include "launch.h"
#include "options.h"
#include "xl_layout.h"
#include "xl_dma.h"
#include "attn2.h"
#include "attn3_layout.h"

namespace mdx {

constexpr int A3_NW = 4, A3_NT = 256, A3_NBUF = 4;
constexpr int A3_BUF = 64 * 80 + 128 + 64 * 128;                    // one ring slot: K tile (64 rows x 80 B), pad, V^T tile (64 rows x 128 B)
constexpr int A3_QST = 3 * 1024;                                    // Q staging per wave and parity: 32 queries x 80 B in three 1-KiB pieces
constexpr int A3_SMEM = A3_NBUF * A3_BUF + 32 * 80 + 128 + A3_NW * A3_QST;   // ring + one-cells + Q staging: 68 736 B, two workgroups per CU
constexpr float A3_DEFER = 4.0f;   // log2 units (see attention2.hip: A2_DEFER)

// A3_ABL: compile-time ablation bits for timing experiments (side builds; results are WRONG): 1 no exp, 8 no QK MFMA, 16 no PV MFMA, 64 no steady-state DMA
#ifndef A3_ABL
#define A3_ABL 0
#endif

// One LDS-DMA piece (xl_dma.h: xl_glds) WITHOUT saving / restoring M0 around it: nothing else in this kernel uses M0 (gfx9 LDS instructions do not,
// the kernel has no s_movrel / s_sendmsg / GWS), and the two s_mov per piece were 15 of the ~330 instructions a wave issued per kv tile in a loop
// that turned out to be bound by instruction issue (profiles/r05_attn3_v3_ablation.log: without exp, without MFMA, without DMA: -12 / -12 / -23 %).
__device__ __forceinline__ void a3_glds(xl_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    // every scalar operand provably wave-uniform (an "s" constraint on a value the compiler believes divergent is silently given a VGPR)
    rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y);
    rs.z = __builtin_amdgcn_readfirstlane(rs.z); rs.w = __builtin_amdgcn_readfirstlane(rs.w);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <bool TWO>
__global__ __launch_bounds__(A3_NT, 2) void attn3_kernel(Attn2Params p) {
    constexpr int D = 40, D8 = 5;
    constexpr int KV = mdx_a3::KV;
    constexpr int KROW = D * 2;                    // K tile row bytes (contiguous rows: the DMA image is lane-linear)
    constexpr int K_BYTES = KV * KROW;             // 5120
    constexpr int K_PAD = 128;                     // keeps the V^T tile (and BUF) a multiple of 128 bytes
    constexpr int VROWS = 64;                      // V^T rows incl. the ones row (40) and the zero rows 41..63 of the second 32-row tile
    constexpr int VOFF = K_BYTES + K_PAD;
    constexpr int BUF = A3_BUF;                    // 13440
    constexpr int KP = D8, VP = D8, NP = KP + VP;  // 1-KiB pieces per tile: K 64 rows x 5 chunks, V^T 40 rows x 8 chunks
    constexpr int RING = A3_NBUF * BUF;
    constexpr int SUB = 32 * KROW;                 // byte distance of the two 32-kv sub-tiles of a K tile (2560)
    constexpr int ONE0 = RING;                     // two 16-byte cells (1, 0 x 7), SUB apart: the K fragment of the pad k-step for the upper half lanes
    constexpr int QST0 = RING + SUB + 128;             // Q staging: [wave][3 KiB] — private to a wave: the next item's Q lands here while this item's is in registers
    constexpr int NSTORE = 5;                      // O stores per wave and item (asm statements: the count is part of the wait arithmetic)
    static_assert(BUF == VOFF + VROWS * 128 && BUF % 128 == 0 && QST0 + A3_NW * A3_QST == A3_SMEM && 2 * A3_SMEM <= 163840, "LDS layout");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    // ---- work items: (view, head, 128-query block), the views b = 8 vl + xcd of this workgroup's XCD in view-major order; workgroup wl of the
    // G on this XCD takes items wl, wl + G, ... (workgroup L runs on XCD L % 8: a dispatch rule used for speed only) ----
    const int xcd = blockIdx.x & 7, wl = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int nqb = p.qblocks;
    const int per_view = p.H * nqb;
    const int nitem = ((p.B - xcd + 7) >> 3) * per_view;
    if (wl >= nitem) return;
    auto decode = [&](int i, int& b_, int& h_, int& qb_) {
        const int vl = i / per_view, rem = i - vl * per_view;
        b_ = vl * 8 + xcd; h_ = rem / nqb; qb_ = rem - h_ * nqb;
    };
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void_t*)smem;

    // ---- LDS image: the bytes the DMA never writes (K pad, V^T rows 41..63), the ones row, the two one-cells ----
    {
        constexpr int ZPB = K_PAD / 16 + (VROWS - D - 1) * 8;   // 16-byte cells to zero per buffer: 8 + 184
        for (int c = tid; c < A3_NBUF * ZPB; c += A3_NT) {
            const int bufi = c / ZPB, i = c - bufi * ZPB;
            const int off = i < K_PAD / 16 ? K_BYTES + i * 16 : VOFF + (D + 1) * 128 + (i - K_PAD / 16) * 16;
            *(uint4*)(smem + bufi * BUF + off) = make_uint4(0, 0, 0, 0);
        }
        const unsigned o2 = MDX_ONE16 * 0x10001u;
        for (int c = tid; c < A3_NBUF * 8; c += A3_NT)          // row D of every buffer's V^T tile (the swizzle permutes its slots within the row)
            *(uint4*)(smem + (c >> 3) * BUF + VOFF + D * 128 + (c & 7) * 16) = make_uint4(o2, o2, o2, o2);
        for (int c = tid; c < (QST0 - RING) / 16; c += A3_NT)
            *(uint4*)(smem + RING + c * 16) = make_uint4((c * 16 == 0 || c * 16 == SUB) ? MDX_ONE16 : 0u, 0, 0, 0);
    }
    __syncthreads();

    // ---- DMA bookkeeping: piece pc = wave + 4 j of the tile's 10 pieces (K first, then V^T): waves 0 / 1 issue three per tile, waves 2 / 3 two.
    // K pieces: the rows past Tk of a source's last tile lie beyond the descriptor's num_records (Tk rows): zero fill, no per-lane state.  V^T
    // pieces: the kv chunks past Tk of the last tile are per-lane out-of-range voffsets (pv_last), selected once per tile (vo[]). ----
    constexpr int PPW = 3;
    unsigned pv_off[PPW], vo[PPW];
    unsigned long long oob_last[PPW];                          // lanes whose V^T chunk lies past Tk in a source's last tile (wave-wide masks, SGPRs)
    int p_lds[PPW], p_step[PPW];
    bool p_isk[PPW];
    const int ntile = (p.Tk + KV - 1) / KV;
    const int tk_last = p.Tk - (ntile - 1) * KV;               // valid kv of a source's last tile (1..64)
    const bool partial = tk_last < KV;
    const bool full_wave = wave + 2 * A3_NW < NP;               // this wave issues three pieces per tile (else two): wave-uniform
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pc = min(wave + j * A3_NW, NP - 1);          // (clamped: a wave without piece j never issues it)
        if (pc < KP) {
            const int n = pc * 64 + lane;                      // linear 16-byte chunk of the K tile
            const int row = n / D8, ch = n - row * D8;
            p_isk[j] = true; p_lds[j] = pc * 1024; p_step[j] = KV * (int)p.ldk * 2;
            pv_off[j] = (unsigned)((row * p.ldk + ch * 8) * 2);
            oob_last[j] = 0ull;
        } else {
            const int row0 = (pc - KP) * 8;
            const int row = mdx_xl::piece_lane_row(row0, lane), ch = mdx_xl::piece_lane_chunk(row0, lane);
            p_isk[j] = false; p_lds[j] = VOFF + row0 * 128; p_step[j] = KV * 2;
            pv_off[j] = (unsigned)((row * p.ldv + ch * 8) * 2);
            oob_last[j] = __builtin_amdgcn_ballot_w64(ch * 8 >= tk_last);
        }
        p_lds[j] = __builtin_amdgcn_readfirstlane(p_lds[j] + (int)lds0);
        p_step[j] = __builtin_amdgcn_readfirstlane(p_step[j]);
        vo[j] = (ntile == 1 && ((oob_last[j] >> lane) & 1ull)) ? XL_OOB : pv_off[j];
    }
    xl_rsrc_t rs_p[PPW];
    auto set_source = [&](int b_, int h_, int sidx) {
        const int bkv = p.kvmap ? p.kvmap[b_ * p.nsrc + sidx] : b_;
        const xl_rsrc_t rsK = xl_make_rsrc_bounded(p.K + (long)bkv * p.sK + (long)h_ * D, (long)(p.Tk - 1) * p.ldk * 2 + KROW);
        const xl_rsrc_t rsV = xl_make_rsrc(p.Vt + (long)bkv * p.sV + (long)h_ * D * p.ldv);
#pragma unroll
        for (int j = 0; j < PPW; ++j) rs_p[j] = p_isk[j] ? rsK : rsV;
    };
    // issue cursor: the tile being issued is tile `it` of source `is_` of item `ii` (view ib, head ih), into the ring slot at byte offset `ioff`.
    // Past the last item the descriptors are empty (num_records 0: zeros into a free slot), so that every wait finds exactly one younger tile in flight.
    int ii = wl, ib, ih, iqb, it = 0, is_ = 0, ioff = 0;
    decode(ii, ib, ih, iqb);
    set_source(ib, ih, 0);
    bool need_src = false;
#define A3_PIECE(j) a3_glds(rs_p[j], p_lds[j] + ioff, vo[j], it * p_step[j]);
#define A3_ISSUE_BEGIN() { if (need_src) { set_source(ib, ih, is_); need_src = false; } }
#define A3_ISSUE_END()                                                                                                 \
    {                                                                                                                  \
        ioff += BUF; if (ioff == RING) ioff = 0;                                                                       \
        if (++it == ntile) {                                                                                           \
            it = 0;                                                                                                    \
            if (++is_ == p.nsrc) { is_ = 0; ii += G; if (ii < nitem) decode(ii, ib, ih, iqb); }                        \
            need_src = ii < nitem;                                                                                     \
            if (ii >= nitem) { _Pragma("unroll") for (int j = 0; j < PPW; ++j) rs_p[j].z = 0u; }                       \
        }                                                                                                              \
        if (partial) {                       /* the next tile's V^T voffsets (the last tile of a source: chunks past Tk out of range) */ \
            const bool l_ = it == ntile - 1;                                                                           \
            _Pragma("unroll") for (int j = 0; j < PPW; ++j) vo[j] = (l_ && ((oob_last[j] >> lane) & 1ull)) ? XL_OOB : pv_off[j]; \
        }                                                                                                              \
    }
    auto issue_tile = [&]() {                                    // a whole tile at once (prologue, idle waves)
        A3_ISSUE_BEGIN()
        if (!(A3_ABL & 64)) { A3_PIECE(0) A3_PIECE(1) if (full_wave) A3_PIECE(2) }
        A3_ISSUE_END()
    };
    // the Q rows of this wave for item (b_, h_, qb_) -> the wave's staging area
    auto issue_q = [&](int b_, int h_, int qb_) {
        const int q0 = qb_ * (A3_NW * 32) + wave * 32;
        const xl_rsrc_t rsQ = xl_make_rsrc(p.Q + (long)b_ * p.sQ + (long)h_ * D);
        const int rows = p.Tq - q0;                              // rows of the wave that exist (<= 0: none)
        // the wave's 32 query rows x 80 bytes as a lane-linear image (chunk n -> row n / 5, 16-byte chunk n % 5): three pieces, the last one half
        // empty; rows past Tq are out of range (zeros).  Computed here, once per item: six registers less in the tile loop.
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = j * 64 + lane, row = n / D8;
            const unsigned off = (n < 32 * D8 && row < rows) ? (unsigned)((row * p.ldq + (n - row * D8) * 8) * 2) : XL_OOB;
            a3_glds(rsQ, lds0 + QST0 + wave * A3_QST + j * 1024, off, q0 * (int)p.ldq * 2);
        }
    };

    // ---- fragment read offsets ----
    const int k0 = mdx_a3::k_row(lane) * KROW + half * 16;        // K row swap23(col) of a sub-tile, this half's 16 bytes of k-step 0
    int v0[4];                                                   // V^T fragment of row tile 0 for (s, t) = st >> 1, st & 1; row tile 1 is 4096 bytes further
#pragma unroll
    for (int st = 0; st < 4; ++st) v0[st] = VOFF + mdx_a3::vt_off(0, st >> 1, st & 1, lane);
    const int qrd = QST0 + wave * A3_QST + col * KROW + half * 16;   // Q fragment of k-step 0 in the staging area

    // W: wait until the next tile of the stream has landed for every wave.  Exactly one younger tile is in flight behind it (+ EXTRA_ younger
    // operations of this wave: the O stores of an item seam); the tile after that goes out next, into the slot every wave has just left.
#define A3_W(EXTRA_)                                                                                                   \
    {                                                                                                                  \
        if (full_wave) xl_wait_vmcnt<3 + (EXTRA_)>(); else xl_wait_vmcnt<2 + (EXTRA_)>();                              \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
    }
    // the partial last tile of a source (slot sl): scrub the V^T pad columns — kv >= Tk inside the last partially valid 16-byte chunk may
    // hold anything, and 0 * NaN is NaN.  Every wave of the workgroup calls this at the same point of the barrier sequence: right behind the W
    // that published the tile, before anyone reads its V^T fragments.
    const bool scrub = partial && (tk_last & 7) != 0;
    auto scrub_tile = [&](int sl) {
        if (tid < D) {
            const int chunk = tk_last >> 3, e0 = tk_last & 7;
            uint4* cell = (uint4*)(smem + sl + VOFF + tid * 128 + ((chunk ^ mdx_a3::vt_swz(tid)) << 4));
            uint4 v = *cell;                                     // keep elements 0 .. e0 - 1 of the 8
            const unsigned keep = (e0 & 1) ? 0xffffu : 0u;       // odd e0: the low half of word e0 / 2 stays
            const int w = e0 >> 1;                               // first word that is (partly) cleared
            v.x = w > 0 ? v.x : (v.x & keep); v.y = w > 1 ? v.y : (w == 1 ? (v.y & keep) : 0u);
            v.z = w > 2 ? v.z : (w == 2 ? (v.z & keep) : 0u); v.w = w == 3 ? (v.w & keep) : 0u;
            *cell = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    f32x16_t SA, SB, oacc[2];
    unsigned osum[TWO ? 10 : 1];                                 // cross-view: the first neighbour's normalised output, packed (blocks.py:213-217 adds two 16-bit outputs)
    float m_run = 0.f, mx = 0.f;
    Frag8 qf[3], kf[3], vf[2][2], pk[2];
    // Q fragments (B operand of S^T = K Q^T) from the staging area: lane -> query column, 8 consecutive dims; dims >= 40 are zero (slot 40 carries -m)
#define A3_READ_Q(dst_)                                                                                                \
    {                                                                                                                  \
        const unsigned char* qa_ = smem + qrd;                                                                         \
        dst_[0].u = *(const uint4*)(qa_); dst_[1].u = *(const uint4*)(qa_ + 32);                                       \
        const uint4 q2_ = *(const uint4*)(qa_ + 64 - 16 * half);     /* lower half: dims 32..39; upper half: nothing (the same bytes, discarded) */ \
        dst_[2].u = half ? make_uint4(0, 0, 0, 0) : q2_;                                                               \
    }
    // K fragment of k-step ks_ for sub-tile sub_ of the tile in the ring slot at byte offset sl_; V^T fragments (both row tiles) of k-step t_ of that sub-tile
#define A3_READ_K1(sl_, sub_, ks_)                                                                                     \
    {                                                                                                                  \
        const unsigned kb_ = (sl_) + (sub_) * SUB;                                                                     \
        if ((ks_) < 2) kf[ks_].u = *(const uint4*)(smem + kb_ + k0 + (ks_) * 32);                                      \
        else kf[2].u = *(const uint4*)(smem + (half ? (unsigned)ONE0 : kb_ + k0 + 64));                                \
    }
#define A3_READ_V1(sl_, sub_, t_)                                                                                      \
    {                                                                                                                  \
        const unsigned char* va_ = smem + (sl_) + ((sub_) ? ((t_) ? v0[3] : v0[2]) : ((t_) ? v0[1] : v0[0]));          \
        vf[0][t_].u = *(const uint4*)(va_); vf[1][t_].u = *(const uint4*)(va_ + 4096);                                 \
    }
    // The MFMAs are asm statements: as builtins they are pure nodes to instruction selection, which let the PV MFMAs of a tile sink below the
    // fences to the end of the block (both accumulator chains back to back, nothing beside them) however the source was ordered; a volatile
    // asm keeps its place between the fences and the other asm statements.  The price: the compiler no longer sees an MFMA, so the wait
    // states between an MFMA's result and a VALU instruction that reads or overwrites it (gfx950: up to 18 for a 16-pass MFMA) are
    // OURS to keep — see A3_MFMA_SETTLE and the distance notes at its uses.  MFMA -> MFMA on the same accumulator tuple needs none.
#if MDX_F16
#define A3_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define A3_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif
#define A3_MFMA_SETTLE() asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory")     /* 24 wait states: any MFMA issued before this has written its result */
#define A3_ZERO1(V_) { _Pragma("unroll") for (int r = 0; r < 16; ++r) V_[r] = 0.f; }
#define A3_QK0(S_, Q_)                                                                                                 \
    {                                                                                                                  \
        if (A3_ABL & 8) { A3_ZERO1(S_) S_[0] += __uint_as_float(kf[0].u.x ^ Q_[0].u.x); }                              \
        else asm volatile(A3_MFMA_OP " %0, %1, %2, 0" : "=&v"(S_) : "v"(kf[0].v), "v"(Q_[0].v));                       \
    }
#define A3_QK(S_, ks_, Q_)                                                                                             \
    {                                                                                                                  \
        if (A3_ABL & 8) S_[ks_] += __uint_as_float(kf[ks_].u.x ^ Q_[ks_].u.x);                                         \
        else asm volatile(A3_MFMA_OP " %0, %1, %2, %0" : "+v"(S_) : "v"(kf[ks_].v), "v"(Q_[ks_].v));                   \
    }
#define A3_PV(t_, rt_)                                                                                                 \
    {                                                                                                                  \
        if (A3_ABL & 16) oacc[rt_][t_] += __uint_as_float(vf[rt_][t_].u.x ^ pk[t_].u.x ^ pk[t_].u.y ^ pk[t_].u.z ^ pk[t_].u.w); \
        else asm volatile(A3_MFMA_OP " %0, %1, %2, %0" : "+v"(oacc[rt_]) : "v"(vf[rt_][t_].v), "v"(pk[t_].v));         \
    }
    // half (4 of 8) of the probabilities of k-step t: score registers 8 t + 4 hf .. + 3 -> two packed words of pk[t]
#define A3_EXP(S_, t_, hf_)                                                                                            \
    {                                                                                                                  \
        float e0_ = S_[8 * (t_) + 4 * (hf_) + 0], e1_ = S_[8 * (t_) + 4 * (hf_) + 1];                                  \
        float e2_ = S_[8 * (t_) + 4 * (hf_) + 2], e3_ = S_[8 * (t_) + 4 * (hf_) + 3];                                  \
        if (!(A3_ABL & 1)) { e0_ = __builtin_amdgcn_exp2f(e0_); e1_ = __builtin_amdgcn_exp2f(e1_); e2_ = __builtin_amdgcn_exp2f(e2_); e3_ = __builtin_amdgcn_exp2f(e3_); } \
        const unsigned w0_ = pack2bf(e0_, e1_), w1_ = pack2bf(e2_, e3_);                                               \
        if (hf_) { pk[t_].u.z = w0_; pk[t_].u.w = w1_; asm volatile("" : "+v"(pk[t_].u.z), "+v"(pk[t_].u.w)); }         \
        else { pk[t_].u.x = w0_; pk[t_].u.y = w1_; asm volatile("" : "+v"(pk[t_].u.x), "+v"(pk[t_].u.y)); }             \
    }
#define A3_FENCE() __builtin_amdgcn_sched_barrier(0)
    // maximum of a score set (this lane's 16 kv), then with the other half of the query (lane ^ 32)
    // (8 v_max3_f32 statements: fmaxf() also canonicalises every input — `v_max_f32 x, x, x` — 13 extra instructions per sub-tile)
#define A3_MAX_LOCAL(S_, dst_)                                                                                         \
    {                                                                                                                  \
        float mm_;                                                                                                     \
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(mm_) : "v"(S_[0]), "v"(S_[1]), "v"(S_[2]));                    \
        _Pragma("unroll") for (int r = 3; r < 15; r += 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mm_) : "v"(S_[r]), "v"(S_[r + 1])); \
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(mm_) : "v"(S_[15]));                                                \
        dst_ = mm_;                                                                                                    \
    }
#define A3_MAX_HALVES(v_)                                                                                              \
    {                                                                                                                  \
        const unsigned mu_ = __float_as_uint(v_);                                                                      \
        auto sw_ = __builtin_amdgcn_permlane32_swap(mu_, mu_, false, false);                                           \
        float ma_ = __uint_as_float(sw_[0]);                                                                           \
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(ma_) : "v"(__uint_as_float(sw_[1])));                               \
        v_ = ma_;                                                                                                      \
    }
    // the deferred maximum (attention2.hip, FOLD): the scores are already relative to m_run (the value folded into the Q fragment when they
    // were multiplied); raise it when a sub-tile exceeds it by A3_DEFER — and always on the first sub-tile of a softmax (there m_run is
    // whatever the previous softmax left: any reference works, the O accumulators are zero).  Everything in place (asm): as plain C the
    // rare arm gets its own result registers and the common arm pays the copies.
#define A3_RESCALE(S_, FIRST_)                                                                                         \
    if ((FIRST_) || __builtin_amdgcn_ballot_w64(mx > A3_DEFER) != 0) {                                                 \
        A3_MFMA_SETTLE();                    /* the previous step's last PV MFMAs may still be writing oacc */               \
        const float inc_ = (FIRST_) ? mx : fmaxf(mx, 0.f);                                                             \
        const float m_new = bf2f((bf16_t)(pack2bf(m_run + inc_, 0.f) & 0xffffu));      /* on the 16-bit grid */          \
        const float delta_ = m_new - m_run;                                                                            \
        const float alpha_ = (FIRST_) ? 0.f : __builtin_amdgcn_exp2f(-delta_);  /* first sub-tile: O is zero; exp2 of a large -delta would be inf */ \
        m_run = m_new;                                                                                                 \
        if (half) qf[2].u.x = pack2bf(-m_new, 0.f);                              /* -m into the pad k slot of this query */ \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(S_[r]) : "v"(delta_)); \
        _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(oacc[rt][r]) : "v"(alpha_)); \
    }
    // kv >= Tk of sub-tile sub_ of the partial last tile -> -inf, then the maximum again.  This lane's register r holds kv 32 sub + 16 (r >> 3) +
    // 8 half + (r & 7) of the tile.  In place (asm), like the rescale: as plain C every tile paid ~90 register copies where this arm rejoined.
#define A3_MASK1(S_, r_, K_) asm volatile("v_cmp_ge_i32 vcc, " #K_ ", %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(S_[r_]) : "v"(lim_), "v"(ninf_) : "vcc");
#define A3_MASK(S_, sub_)                                                                                              \
    {                                                                                                                  \
        const int lim_ = tk_last - 32 * (sub_) - 8 * half;                                                             \
        const float ninf_ = -INFINITY;                                                                                 \
        A3_MASK1(S_, 0, 0) A3_MASK1(S_, 1, 1) A3_MASK1(S_, 2, 2) A3_MASK1(S_, 3, 3)                                    \
        A3_MASK1(S_, 4, 4) A3_MASK1(S_, 5, 5) A3_MASK1(S_, 6, 6) A3_MASK1(S_, 7, 7)                                    \
        A3_MASK1(S_, 8, 16) A3_MASK1(S_, 9, 17) A3_MASK1(S_, 10, 18) A3_MASK1(S_, 11, 19)                              \
        A3_MASK1(S_, 12, 20) A3_MASK1(S_, 13, 21) A3_MASK1(S_, 14, 22) A3_MASK1(S_, 15, 23)                            \
        A3_MAX_LOCAL(S_, mx)                                                                                           \
        A3_MAX_HALVES(mx)                                                                                              \
    }
    // One pipeline step: consumes the scores SC_ of a sub-tile (its V^T fragments in vf), produces the scores SN_ of the next one (its K fragments
    // in kf, Q operand Q_), refills kf for the sub-tile after that (slot KSL_, sub-tile KSUB_) and vf for the next (VSL_, VSUB_) as the MFMAs
    // release them, and leaves the next sub-tile's maximum in mx.  P0_ / P1_: DMA pieces issued beside the PV MFMAs.
    //   G0-2: 3 QK MFMAs (next)   beside  exp / pack of k-step 0 and half of k-step 1
    //   G3-6: 4 PV MFMAs (this)   beside  the rest, the K refills, two DMA pieces, the maximum of the next scores, the V^T refills
    // Distances kept by hand (asm MFMAs): the maximum reads SN_ three MFMAs (>= 96 cycles) after its last QK MFMA; a fragment register is
    // refilled >= 2 MFMAs after the MFMA that read it.
#define A3_STEP(SC_, SN_, Q_, KSL_, KSUB_, VSL_, VSUB_, FIRST_, P0_, P1_)                                              \
    {                                                                                                                  \
        A3_RESCALE(SC_, FIRST_)                                                                                        \
        A3_FENCE();                                                                                                    \
        A3_QK0(SN_, Q_) A3_EXP(SC_, 0, 0) A3_FENCE();                                                                  \
        A3_QK(SN_, 1, Q_) A3_EXP(SC_, 0, 1) A3_FENCE();                                                                \
        A3_QK(SN_, 2, Q_) A3_EXP(SC_, 1, 0) A3_READ_K1(KSL_, KSUB_, 0) A3_FENCE();                                     \
        A3_PV(0, 0) A3_EXP(SC_, 1, 1) A3_READ_K1(KSL_, KSUB_, 1) A3_FENCE();                                           \
        A3_PV(0, 1) P0_ A3_READ_K1(KSL_, KSUB_, 2) A3_FENCE();                                                         \
        A3_PV(1, 0) A3_MAX_LOCAL(SN_, mx) A3_READ_V1(VSL_, VSUB_, 0) A3_FENCE();                                       \
        A3_PV(1, 1) P1_ A3_FENCE();                                                                                    \
        A3_MAX_HALVES(mx)                                                                                              \
        A3_READ_V1(VSL_, VSUB_, 1)                                                                                     \
    }
    // scores of sub-tile 0 of the tile in slot xs on their own (stream start; the item after one this wave sat out): everything exposed
#define A3_PRODUCE_ALONE()                                                                                             \
    {                                                                                                                  \
        A3_READ_K1(xs, 0, 0) A3_READ_K1(xs, 0, 1) A3_READ_K1(xs, 0, 2)                                                 \
        A3_QK0(SA, qf) A3_QK(SA, 1, qf) A3_QK(SA, 2, qf)                                                               \
        A3_READ_V1(xs, 0, 0) A3_READ_V1(xs, 0, 1)                                                                      \
        A3_MFMA_SETTLE();                                                                                              \
        A3_READ_K1(xs, 1, 0) A3_READ_K1(xs, 1, 1) A3_READ_K1(xs, 1, 2)                                                 \
        A3_MAX_LOCAL(SA, mx)                                                                                           \
        A3_MAX_HALVES(mx)                                                                                              \
    }
#define A3_NOP_ {}
#define A3_P0_ { if (!(A3_ABL & 64)) A3_PIECE(0) }
#define A3_P1_ { if (!(A3_ABL & 64)) A3_PIECE(1) }
#define A3_P2_ { if (!(A3_ABL & 64) && full_wave) A3_PIECE(2) }
    // One kv tile (ring slot xs; xs1 = the next tile's) behind its W: two steps.  NEWQ_ (an item's last tile): between the steps — the first step's QK
    // MFMAs were the last readers of this item's Q — qf is re-read from the staging area: the NEXT item's Q, with the current reference maximum in
    // its pad slot (m_run is never reset: it is only the value the scores were multiplied relative to), so the second step multiplies the next
    // item's first scores.
#define A3_TILE(NEWQ_, LT_)                                                                                            \
    {                                                                                                                  \
        A3_ISSUE_BEGIN()                                                                                               \
        if (scrub && next_last) scrub_tile(xs1);                                                                       \
        if ((LT_) && partial) A3_MASK(SA, 0)                                                                           \
        A3_STEP(SA, SB, qf, xs1, 0, xs, 1, first, A3_P0_, A3_P1_)                                                      \
        if ((LT_) && partial) A3_MASK(SB, 1)                                                                           \
        NEWQ_                                                                                                          \
        A3_STEP(SB, SA, qf, xs1, 1, xs1, 0, false, A3_P2_, A3_NOP_)                                                    \
        A3_ISSUE_END()                                                                                                 \
    }

    // ---- prologue: the first item's Q, tiles 0 and 1 out, W_0, tile 2 out ----
    int cb, ch, cqb;                                             // the item being consumed
    decode(wl, cb, ch, cqb);
    issue_q(cb, ch, cqb);
    issue_tile(); issue_tile();
    A3_W(0)
    issue_tile();
    if (scrub && ntile == 1) scrub_tile(0);
    A3_READ_Q(qf)
    int xs = 0, xs1 = BUF;                                       // ring slots (byte offsets) of the tile being consumed and of the next
    bool have_s = false;                                         // SA / kf / vf hold the first sub-tile of the coming item (produced by the previous item's last step)
    bool seam = false;                                           // the previous item's O stores are the youngest operations of this wave
    const int total = ntile * p.nsrc;                            // tiles per item

    for (int i = wl; i < nitem; i += G) {
        if (i != wl) decode(i, cb, ch, cqb);
        const int q0w = cqb * (A3_NW * 32) + wave * 32;          // this wave's first query of the item
        const bool active = __builtin_amdgcn_readfirstlane(q0w < p.Tq ? 1 : 0) != 0;
        const bool more = i + G < nitem;                         // another item follows: its Q goes to the other staging parity behind the first W
        int nb = cb, nh = ch, nqb_ = cqb;
        if (more) decode(i + G, nb, nh, nqb_);
        int it_c = 0;                                            // tile inside its source
        if (!active) {
            // a wave without a real query in this item (only a (view, head)'s last block): stage and synchronise — the barrier sequence of the active waves
            for (int x = 0; x < total; ++x) {
                if (seam) { A3_W(NSTORE) seam = false; } else { A3_W(0) }
                if (x == 0 && more) issue_q(nb, nh, nqb_);
                issue_tile();
                const bool next_last = (it_c + 1 == ntile ? 0 : it_c + 1) == ntile - 1;
                if (scrub && next_last) scrub_tile(xs1);
                if (++it_c == ntile) it_c = 0;
                xs = xs1; xs1 += BUF; if (xs1 == RING) xs1 = 0;
            }
            if (total == 1 && more) xl_wait_vmcnt<0>();
            if (more) { A3_READ_Q(qf) if (half) qf[2].u.x = pack2bf(-m_run, 0.f); }
            have_s = false;
            continue;
        }
        if (!have_s) A3_PRODUCE_ALONE()
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[rt][r] = 0.f;
        bool first = true;
        for (int src = 0; src < p.nsrc; ++src) {
            const bool last_src = src + 1 == p.nsrc;
            // the tiles before a source's last one: the hot loop (no mask, never an item's last tile)
            for (int t = 0; t < ntile - 1; ++t) {
                const bool next_last = t + 1 == ntile - 1;
                // W; behind the item's first W the NEXT item's Q goes out (older than every tile a later W leaves in flight: it has landed by the
                // W of the item's last tile)
                if (seam) { A3_W(NSTORE) } else { A3_W(0) }
                if (src == 0 && t == 0 && more) issue_q(nb, nh, nqb_);
                A3_TILE(A3_NOP_, false)
                seam = false;
                first = false;
                xs = xs1; xs1 += BUF; if (xs1 == RING) xs1 = 0;
            }
            {   // the source's last tile: the kv mask; on the item's last tile the second step multiplies the NEXT item's first scores
                const bool next_last = ntile == 1;
                const bool x0 = src == 0 && ntile == 1;          // ... which is also its first: the next item's Q goes out here and is waited for in full
                if (seam) { A3_W(NSTORE) } else { A3_W(0) }
                if (x0 && more) issue_q(nb, nh, nqb_);
                if (last_src) {
                    if (x0 && more) xl_wait_vmcnt<0>();
                    A3_TILE({ if (more) { A3_READ_Q(qf) if (half) qf[2].u.x = pack2bf(-m_run, 0.f); } }, true)
                } else {
                    A3_TILE(A3_NOP_, true)
                }
                seam = false;
                first = false;
                xs = xs1; xs1 += BUF; if (xs1 == RING) xs1 = 0;
            }
            // ---- end of a source: normalise, accumulate (cross-view), restart.  joint: the sources are one kv sequence — only after the last ----
            if (p.joint && !last_src) continue;
            A3_MFMA_SETTLE();                                    // the last PV MFMAs have written oacc
            {
                // row 40 of O^T = sum of the (16-bit) probabilities: row tile 1, local row 8 -> register 4 of the lower half lanes
                const float inv = 1.0f / __shfl(oacc[1][4], col, 64);
                if (TWO && src == 0) {                            // first neighbour done: park it, restart the accumulators
#pragma unroll
                    for (int u = 0; u < 8; ++u) osum[TWO ? u : 0] = pack2bf(oacc[0][2 * u] * inv, oacc[0][2 * u + 1] * inv);
#pragma unroll
                    for (int u = 0; u < 2; ++u) osum[TWO ? 8 + u : 0] = pack2bf(oacc[1][2 * u] * inv, oacc[1][2 * u + 1] * inv);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[rt][r] = 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[0][r] *= inv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[1][r] *= inv;
                    if (TWO) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const unsigned w = osum[TWO ? u : 0];
                            oacc[0][2 * u] += bf2f((bf16_t)(w & 0xffffu)); oacc[0][2 * u + 1] += bf2f((bf16_t)(w >> 16));
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const unsigned w = osum[TWO ? 8 + u : 0];
                            oacc[1][2 * u] += bf2f((bf16_t)(w & 0xffffu)); oacc[1][2 * u + 1] += bf2f((bf16_t)(w >> 16));
                        }
                    }
                }
                first = true;                                    // the next source (if any) starts a softmax; m_run carries over as its reference
            }
        }
        // ---- store O[q][h * 40 + d]: lane (q, half) holds d = 8 u + 4 half + {0..3} of row tile 0 (u = 0..3) and d = 32 + 4 half + {0..3}.  asm: NSTORE
        // statements per wave, counted by the next W (a wave with a real query executes all of them: lanes past Tq are masked, not skipped) ----
        {
            const int qq = q0w + col;
            bf16_t* op = p.O + (long)cb * p.sO + (long)min(qq, p.Tq - 1) * p.ldo + (long)ch * D + 4 * half;
            uint2 ov[5];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ov[u].x = pack2bf(oacc[0][4 * u], oacc[0][4 * u + 1]);
                ov[u].y = pack2bf(oacc[0][4 * u + 2], oacc[0][4 * u + 3]);
            }
            ov[4].x = pack2bf(oacc[1][0], oacc[1][1]);
            ov[4].y = pack2bf(oacc[1][2], oacc[1][3]);
            const bool ok = qq < p.Tq;
            const unsigned long long exec_all = __builtin_amdgcn_ballot_w64(ok);
            unsigned long long keep_exec;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(keep_exec) : "s"(exec_all) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(op), "v"(ov[0]) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off offset:16" :: "v"(op), "v"(ov[1]) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off offset:32" :: "v"(op), "v"(ov[2]) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off offset:48" :: "v"(op), "v"(ov[3]) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off offset:64" :: "v"(op), "v"(ov[4]) : "memory");
            asm volatile("s_mov_b64 exec, %0" :: "s"(keep_exec) : "memory");
            seam = true;
        }
        // ---- the next item: its Q is in qf, its first scores in SA (relative to m_run, which its pad slot carries) ----
        have_s = more;
    }
#undef A3_TILE
#undef A3_STEP
#undef A3_PRODUCE_ALONE
    xl_wait_vmcnt<0>();                                          // the dummy pieces past the stream must have landed before the LDS is handed back
}

// Shapes the pipelined kernel takes: what attention2.hip's d = 40 FOLD instances take (self, the two-source cross-view sum, joint sources).
bool attn3_supported(const Attn2Params& p) {
    if (!opt(OPT_ATTN3) || p.d != 40 || !p.q_prescaled || !opt(OPT_ATTN2_FOLD)) return false;
    if ((long)p.Tq * p.ldq * 2 >= 0x40000000L || (p.ldq % 8) || (p.sQ % 8)) return false;      // Q goes through LDS-DMA too
    return p.nsrc >= 1;
}

int launch_attn3(const Attn2Params& p, hipStream_t st) {
    Attn2Params q = p;
    q.qblocks = (p.Tq + A3_NW * 32 - 1) / (A3_NW * 32);          // 128-query blocks per (view, head)
    // persistent: two workgroups per CU (ATTN3_WGS per XCD: 0 = automatic) walk the items of their XCD; fewer when there are fewer items
    static const int cus = [] { int d = 0, n = 256; if (hipGetDevice(&d) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
    long per_xcd = (long)opt(OPT_ATTN3_WGS);
    if (per_xcd <= 0) per_xcd = 2L * cus / 8;
    const long items0 = (long)((p.B + 7) / 8) * p.H * q.qblocks;  // items of XCD 0 (the longest list)
    if (per_xcd > items0) per_xcd = items0;
    const dim3 grid((unsigned)(8 * per_xcd), 1, 1);
    const bool two = p.nsrc == 2 && !p.joint;
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_dyn_smem((const void*)kern, A3_SMEM, "attn3")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(A3_NT), A3_SMEM, st, q);
        char tag[64];
        snprintf(tag, sizeof tag, "attn3_kernel<40,%s>", two ? "xview" : (p.nsrc > 1 ? "joint" : "self"));
        return check_launch(tag);
    };
    return two ? go(attn3_kernel<true>) : go(attn3_kernel<false>);
}

}  // namespace mdx
