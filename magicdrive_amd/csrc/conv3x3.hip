// conv3x3.hip — 3x3 / stride 1 / pad 1 channels-last convolution (every ResnetBlock2D conv of the UNet and the ControlNet)
// as an implicit GEMM whose A slab is shared by the three horizontal taps.
//
// Why: the generic kernel (gemm_conv.hip) is bound by the L2 -> LDS load path (~32 B/clk per CU, measured), and it pushes the
// SAME input pixels through that path nine times (once per tap).  For stride 1 the three taps of one kernel row read the same
// pixels shifted by one: output pixel m (linear over b, oy, ox) at tap (ky, kx) reads input pixel m + (ky-1) W + (kx-1).
// So per (64-channel block, ky) ONE "super slab" of 128 + 2 consecutive pixels is staged in LDS and the fragment reads of tap
// kx start kx rows further down; only the weight slab changes per tap.  Load-path bytes per three taps: 16.6 KB (A) + 48 KB (W)
// instead of 96 KB.  The slab is addressed linearly (pixel g -> A + g * ldx): no per-row (iy, ix) arithmetic, unconditional
// loads.  What linear addressing gets wrong — horizontal / vertical padding and rows that wrap into the next image row / image
// — is exactly the set of (output pixel, tap) pairs that must read zero, and is fixed where the fragments are read: a lane's
// A fragment belongs to ONE output pixel, so a per-lane (oy, ox) mask zeroes it (4 v_cndmask per fragment).
//
// Everything else is gemm_conv.hip's 128x128x64 tile: 4 waves (2 x 2), register-staged double-buffered LDS ring, XCD-aware tile
// order, the shared fused epilogue (bias, temb row, SiLU, residual, LDS-transposed 16-byte stores).  K order is
// (channel block, ky, kx): a permutation of the reduction, same fp32 accumulation.
#include "common.h"
#include "launch.h"
#include "options.h"
#include "gemm_params.h"

namespace mdx {

__global__ __launch_bounds__(256, 2) void conv3x3_kernel(GCParams p) {
    constexpr int BM = 128, BN = 128, BK = 64, LSTR = BK + 8, AR = BM + 2, TM = 2, TN = 2, NTH = 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* As = (bf16_t*)smem;                  // [2][AR][LSTR]
    bf16_t* Bs = As + 2 * AR * LSTR;             // [2][BN][LSTR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int frow = lane & 31, half = lane >> 5;
    int tile_m, tile_n;
    if (!tile_coords(p, tile_m, tile_n)) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int W_ = p.Wi, H_ = p.Hi;
    const int ncb = p.Cin / BK;
    const int nunits = 9 * ncb;

    // ---- staging: thread owns 16-byte chunk kc of rows rbase + 32 i (+ rows 128, 129 of the super slab for tid < 16) ----
    const int kc = tid & 7, rbase = tid >> 3;
    const bool extra = tid < 16;                                     // rows 128 + (tid >> 3)
    const bf16_t* const a_col = p.A + kc * 8;
    long w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w_off[i] = (long)min(n0 + rbase + 32 * i, p.N - 1) * p.ldw + kc * 8;
    uint4 a_reg[5], b_reg[4];
    const int P = p.M;                                               // pixels in the tensor
#define C3_LOAD_A(cb_, ky_)                                                                            \
    {                                                                                                  \
        const int gb = m0 - 1 + ((ky_) - 1) * W_ + rbase;                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int g = min(max(gb + 32 * i, 0), P - 1);                                             \
            const uint4 v = *(const uint4*)(a_col + (long)g * p.lda + (cb_) * BK);                     \
            a_reg[i] = make_uint4(v.x, v.y, v.z, v.w);                                                 \
        }                                                                                              \
        {                                                                                              \
            const int g = min(max(gb + 128, 0), P - 1);                                                \
            const uint4 v = *(const uint4*)(a_col + (long)g * p.lda + (cb_) * BK);                     \
            a_reg[4] = make_uint4(v.x, v.y, v.z, v.w);                                                 \
        }                                                                                              \
    }
#define C3_LOAD_B(cb_, tap_)                                                                           \
    {                                                                                                  \
        const long ko = (long)(tap_) * p.Cin + (cb_) * BK;                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const uint4 v = *(const uint4*)(p.W + w_off[i] + ko);                                      \
            b_reg[i] = make_uint4(v.x, v.y, v.z, v.w);                                                 \
        }                                                                                              \
    }
#define C3_STORE_A(buf)                                                                                \
    {                                                                                                  \
        bf16_t* d = As + (buf) * AR * LSTR + rbase * LSTR + kc * 8;                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *(uint4*)(d + i * 32 * LSTR) = a_reg[i];         \
        if (extra) *(uint4*)(d + 128 * LSTR) = a_reg[4];                                               \
    }
#define C3_STORE_B(buf)                                                                                \
    {                                                                                                  \
        bf16_t* d = Bs + (buf) * BN * LSTR + rbase * LSTR + kc * 8;                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) *(uint4*)(d + i * 32 * LSTR) = b_reg[i];         \
    }

    // ---- this lane's two output pixels (A-fragment rows) ----
    int oy[TM], ox[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = min(m0 + wm * 64 + i * 32 + frow, p.M - 1);
        const int rem = m % (H_ * W_);
        oy[i] = rem / W_;
        ox[i] = rem - oy[i] * W_;
    }

    EpiRegs<BM, BN, TN, NTH> er;
    epi_prefetch<BM, BN, TN, NTH>(p, 0, m0, n0, wn * TN * 32, lane, tid, er);

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    C3_LOAD_A(0, 0)
    C3_LOAD_B(0, 0)
    C3_STORE_A(0)
    C3_STORE_B(0)
    __syncthreads();

    int cb = 0, ky = 0, kx = 0, apar = 0;                            // current unit; A buffer holding its group
    for (int u = 0; u < nunits; ++u) {
        // next unit
        int ncb_ = cb, nky = ky, nkx = kx + 1;
        if (nkx == 3) { nkx = 0; if (++nky == 3) { nky = 0; ++ncb_; } }
        const bool more = u + 1 < nunits;
        const bool newgrp = more && nkx == 0;
        if (!(p.dbg & 4)) {
            if (more) C3_LOAD_B(ncb_, nky * 3 + nkx)
            if (newgrp) C3_LOAD_A(ncb_, nky)
        }

        // ---- multiply unit u ----
        bool z[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) z[i] = !((unsigned)(oy[i] + ky - 1) < (unsigned)H_ && (unsigned)(ox[i] + kx - 1) < (unsigned)W_);
        const bf16_t* as = As + apar * AR * LSTR + (wm * 64 + frow + kx) * LSTR + half * 8;
        const bf16_t* bs = Bs + (u & 1) * BN * LSTR + (wn * 64 + frow) * LSTR + half * 8;
        // Fragment reads run one k-step ahead of the MFMAs (second register set) and are pinned between them: left to the
        // scheduler every ds_read sits right before its first use and each k-step exposes an LDS round trip (measured: the
        // reads + MFMAs of this loop alone ran at half the matrix-pipe rate).
        if (!(p.dbg & 8)) {
            Frag8 af[2][TM], bfr[2][TN];
#define C3_READ(set, ks_)                                                                                   \
            {                                                                                               \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                            \
                    const uint4 v = *(const uint4*)(as + i * 32 * LSTR + (ks_) * 16);                       \
                    af[set][i].u = z[i] ? make_uint4(0, 0, 0, 0) : v;                                       \
                }                                                                                           \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bfr[set][j].u = *(const uint4*)(bs + j * 32 * LSTR + (ks_) * 16); \
            }
            C3_READ(0, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < BK / 16) C3_READ(nxt, ks + 1)
                if (p.dbg & 1) {                                      // debug: fragment reads without the MFMAs
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][0][0] += __uint_as_float(af[cur][i].u.x ^ bfr[cur][i].u.y);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = MDX_MFMA_32x32x16(bfr[cur][j].v, af[cur][i].v, acc[i][j]);
                }
                if (ks + 1 < BK / 16) {
#pragma unroll
                    for (int n = 0; n < TM + TN; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef C3_READ
        }
        if (!(p.dbg & 2)) {
            if (more) C3_STORE_B((u + 1) & 1)
            if (newgrp) { C3_STORE_A(apar ^ 1) }
        }
        __syncthreads();
        if (newgrp) apar ^= 1;
        cb = ncb_; ky = nky; kx = nkx;
    }
#undef C3_LOAD_A
#undef C3_LOAD_B
#undef C3_STORE_A
#undef C3_STORE_B
    epilogue_coalesced<BM, BN, TM, TN, NTH>(p, 0, m0, n0, wm * TM * 32, wn * TN * 32, lane, tid, acc, smem, er);
}

bool conv3x3_supported(const GCParams& p) {
    return p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.ph == 1 && p.pw == 1 && p.Hi == p.Ho && p.Wi == p.Wo &&
           (p.Cin % 64) == 0 && p.splitk <= 1 && !p.c_f32 && p.epi != 1 && p.batch <= 1;
}

int launch_conv3x3(const GCParams& p, hipStream_t st) {
    constexpr int BM = 128, BN = 128;
    constexpr size_t ring = (size_t)2 * (130 + 128) * 72 * 2, ctile = (size_t)BM * (BN + 8) * 2;
    constexpr size_t smem = ring > ctile ? ring : ctile;
    if (int rc = ensure_dyn_smem((const void*)conv3x3_kernel, smem, "conv3x3")) return rc;
    GCParams q = p;
    q.mt = (p.M + BM - 1) / BM; q.nt = (p.N + BN - 1) / BN;
    const int swz = (int)opt(OPT_GEMM_SWZ);
    q.swz = swz && q.nt > 1 && q.mt >= 128;
    const int dbg = (int)opt(OPT_C3_DBG);
    q.dbg = dbg;
    const unsigned nblk = q.swz ? (unsigned)((q.mt + 7) / 8 * 8 * q.nt) : (unsigned)(q.mt * q.nt);
    hipLaunchKernelGGL(conv3x3_kernel, dim3(nblk), dim3(256), smem, st, q);
    return check_launch("conv3x3_kernel");
}

}  // namespace mdx
