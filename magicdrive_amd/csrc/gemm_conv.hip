// gemm_conv.hip — bf16 MFMA GEMM and channels-last implicit-GEMM convolution for gfx950.
//
// One kernel template serves both: the only difference is how a row of the "A" operand is
// addressed (a token row of a [M][K] matrix, or the (ky,kx,ci) gather of an output pixel's
// receptive field in a [B][Hi][Wi][Cin] feature map — no im2col buffer ever exists in HBM).
//
//   C[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] + temb[row(m)][n] ) + R[m][n]
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile BM x BN x 64, each wave (BM/2) x (BN/2) out
// of 32x32x16 bf16 MFMAs with fp32 accumulation.  Operands are staged global -> registers -> LDS
// (double-buffered; the next tile's global loads are in flight while the current tile is
// multiplied) with 16-byte vector accesses; LDS rows are padded by 16 B so the ds_read_b128
// fragment reads of a 16-lane group hit 16 distinct 16-byte slots (conflict-free).
// The MFMA is issued as D = Wfrag x Afrag so a lane ends up holding 4 CONSECUTIVE n for one m:
// the epilogue does 8-byte bf16x4 stores and float4 bias loads instead of 2-byte scatters.
//
// Small-M / huge-K shapes (the 7x13 and 4x7 UNet levels: M = 546, 168 rows, K up to 23040)
// are split along K across blockIdx.z into fp32 slabs and reduced by splitk_reduce_kernel,
// which applies the same epilogue.  Slabs (not atomics) keep results run-to-run deterministic.
//
// Reference semantics replaced: ATen conv2d/addmm call sites listed in include/mdx.h.
#include "common.h"
#include "launch.h"
#include "options.h"
#include "gemm_params.h"

namespace mdx {


int launch_gemm_ws(const GCParams& p, hipStream_t st);                          // gemm_ws.hip: weight-stationary K = 320 GEMM
bool ws_supported(const GCParams& p);
bool ws_fuses_layernorm(const GCParams& p);
int launch_layernorm_plain(const bf16_t* X, bf16_t* Y, int M, int C, long ldx, long ldy, float eps, hipStream_t st);   // norm.hip
int launch_rowstat(const bf16_t* X, int M, int C, long ldx, float* stat, int parts, hipStream_t st);                      // norm.hip
bool ws_emits_rowstat(const GCParams& p);                                                                                  // gemm_ws.hip
int launch_gemm_xl(const GCParams& p, bool conv, int bn, hipStream_t st);       // gemm_xl.hip: 256 x {256,160} LDS-DMA quadrant-phase tiles
bool xl_supported(const GCParams& p, bool conv, int bn);

// WM x WN waves (NTH = 64 WM WN threads); each wave owns a (BM/WM) x (BN/WN) sub-tile of 32x32 MFMA tiles.
template <int BM, int BN, int BK, int WM, int WN, bool CONV, bool PIPE>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 8) ? 2 : 1) void gemm_conv_kernel(GCParams p) {
    constexpr int NTH = WM * WN * 64;
    constexpr int LSTR = BK + 8;  // LDS row stride in elements (+16 B pad: conflict-free ds_read_b128)
    constexpr int KCH = BK / 8;   // 16-byte chunks per row of a slab
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_CH = BM * BK / 8 / NTH;
    constexpr int B_CH = BN * BK / 8 / NTH;
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for the thread count");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* As = (bf16_t*)smem;              // [2][BM][LSTR]
    bf16_t* Bs = As + 2 * BM * LSTR;         // [2][BN][LSTR]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (p.timing) ts0 = __builtin_amdgcn_s_memtime();
    int tile_m, tile_n;
    if (!tile_coords(p, tile_m, tile_n)) return;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    long zb = 0;       // batch index
    int kz = 0;        // split-K slice
    if (p.batch > 1) zb = blockIdx.z; else kz = blockIdx.z;
    const int kbeg = kz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nt = (kend - kbeg + BK - 1) / BK;

    constexpr int RSTEP = NTH / KCH;   // rows covered by one pass of the workgroup's threads
    const int kc = tid % KCH;     // 16-byte chunk within the BK-wide k slab
    const int rbase = tid / KCH;

    // ---- per-thread row bookkeeping (fixed over the K loop) ----
    const bf16_t* a_ptr[A_CH];
    bool a_ok[A_CH];
    int a_iy0[A_CH], a_ix0[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int m = m0 + rbase + RSTEP * i;
        a_ok[i] = m < p.M;
        if (CONV) {
            int mm = a_ok[i] ? m : 0;
            int hw = p.Ho * p.Wo;
            int b = mm / hw;
            int rem = mm - b * hw;
            int oy = rem / p.Wo;
            int ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.sh - p.ph;
            a_ix0[i] = ox * p.sw - p.pw;
            a_ptr[i] = p.A + (long)b * p.Hi * p.Wi * p.lda;
        } else {
            a_iy0[i] = 0; a_ix0[i] = 0;
            a_ptr[i] = p.A + zb * p.sA + (long)(a_ok[i] ? m : 0) * p.lda;
        }
    }
    const bf16_t* b_ptr[B_CH];
    bool b_ok[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        int n = n0 + rbase + RSTEP * i;
        b_ok[i] = n < p.N;
        b_ptr[i] = p.W + zb * p.sW + (long)(b_ok[i] ? n : 0) * p.ldw;
    }
    // conv tap tracking for this thread's k chunk
    int ky = 0, kx = 0, ci = 0;
    const bool cim = CONV && p.cimajor;
    if (CONV) {
        if (cim) {          // slab T = kbeg / BK + t  ->  tap T % ntaps, channel block (T / ntaps) * BK
            const int T0 = kbeg / BK, ntaps = p.kh * p.kw;
            const int tap = T0 % ntaps;
            ci = (T0 / ntaps) * BK + kc * 8;
            ky = tap / p.kw;
            kx = tap - ky * p.kw;
        } else {
            int kk = kbeg + kc * 8;
            int tap = kk / p.Cin;
            ci = kk - tap * p.Cin;
            ky = tap / p.kw;
            kx = tap - ky * p.kw;
        }
    }

    uint4 a_reg[A_CH], b_reg[B_CH];
    auto load_tile = [&](int t) {
        const int kk = cim ? (ky * p.kw + kx) * p.Cin + ci : kbeg + t * BK + kc * 8;
        const bool kok = cim ? (kbeg + t * BK < kend) : kk < kend;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_ok[i] && kok) {
                if (CONV) {
                    int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
                    if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                        v = *(const uint4*)(a_ptr[i] + ((long)iy * p.Wi + ix) * p.lda + ci);
                } else {
                    v = *(const uint4*)(a_ptr[i] + kk);
                }
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b_ok[i] && kok) v = *(const uint4*)(b_ptr[i] + kk);
            b_reg[i] = v;
        }
        if (CONV) {  // advance the tap cursor by one K slab
            if (cim) {
                if (++kx == p.kw) { kx = 0; if (++ky == p.kh) { ky = 0; ci += BK; } }
            } else {
                ci += BK;
                while (ci >= p.Cin) {
                    ci -= p.Cin;
                    if (++kx == p.kw) { kx = 0; ++ky; }
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        bf16_t* as = As + buf * BM * LSTR;
        bf16_t* bs = Bs + buf * BN * LSTR;
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            *(uint4*)(as + (rbase + RSTEP * i) * LSTR + kc * 8) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            *(uint4*)(bs + (rbase + RSTEP * i) * LSTR + kc * 8) = b_reg[i];
    };

    EpiRegs<BM, BN, TN, NTH> er;
    const bool coalesced_out = p.splitk <= 1 && !p.c_f32;   // block-uniform
    if (coalesced_out) epi_prefetch<BM, BN, TN, NTH>(p, zb, m0, n0, wn * TN * 32, lane, tid, er);

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    if (p.timing) ts1 = __builtin_amdgcn_s_memtime();

    const int frow = lane & 31;
    const int fk = (lane >> 5) * 8;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const bf16_t* as = As + buf * BM * LSTR + (wm * TM * 32 + frow) * LSTR + fk;
        const bf16_t* bs = Bs + buf * BN * LSTR + (wn * TN * 32 + frow) * LSTR + fk;
        if constexpr (PIPE && TM == 2 && TN == 2) {
            // Fragment reads run one k-step ahead of the MFMAs (second register set), pinned between them with
            // sched_group_barrier: left alone every ds_read sits right before its first use and each k-step exposes an LDS
            // round trip (round 2 measured +7 % from this alone).
            Frag8 af[2][TM], bfr[2][TN];
#define GC_READ(set, ks_)                                                                                             \
            {                                                                                                         \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) af[set][i].u = *(const uint4*)(as + i * 32 * LSTR + (ks_) * 16);  \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bfr[set][j].u = *(const uint4*)(bs + j * 32 * LSTR + (ks_) * 16); \
            }
            GC_READ(0, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < BK / 16) GC_READ(nxt, ks + 1)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = MDX_MFMA_32x32x16(bfr[cur][j].v, af[cur][i].v, acc[i][j]);
                if (ks + 1 < BK / 16) {                              // {1 MFMA, 1 ds_read} x 4 (PIPE is only instantiated for TM = TN = 2)
                    static_assert(TM == 2 && TN == 2, "interleave pattern");
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef GC_READ
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                Frag8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i].u = *(const uint4*)(as + i * 32 * LSTR + ks * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j].u = *(const uint4*)(bs + j * 32 * LSTR + ks * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = MDX_MFMA_32x32x16(bfr[j].v, af[i].v, acc[i][j]);
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    if (p.timing) ts2 = __builtin_amdgcn_s_memtime();
    if (coalesced_out) {   // block-uniform: bf16 output goes through the LDS transpose
        epilogue_coalesced<BM, BN, TM, TN, NTH>(p, zb, m0, n0, wm * TM * 32, wn * TN * 32, lane, tid, acc, smem, er);
        if (p.timing && tid == 0) {
            unsigned long long* t = p.timing + 5 * ((long)tile_n * p.mt + tile_m);
            t[0] = ts0; t[1] = ts1; t[2] = ts2; t[3] = __builtin_amdgcn_s_memtime();
            t[4] = 0;
        }
        return;
    }
    // fp32 output / split-K slabs: lane holds m = column (lane&31), n rows 8g + 4*(lane>>5) + (0..3)
    const int half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 32 + i * 32 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = n0 + wn * TN * 32 + j * 32 + 8 * g + 4 * half;
                if (nb >= p.N) continue;
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (p.splitk > 1) {
                    float* w = p.ws + ((long)kz * p.M + m) * p.N + nb;
                    *(float4*)w = make_float4(v[0], v[1], v[2], v[3]);
                } else if (p.epi == 1) {
                    if (TN == 2 && j == 0) {
                        float gte[4] = {acc[i][TN - 1][4 * g], acc[i][TN - 1][4 * g + 1],
                                        acc[i][TN - 1][4 * g + 2], acc[i][TN - 1][4 * g + 3]};
                        epilogue_store(p, zb, m, nb, v, gte);
                    }
                } else {
                    epilogue_store(p, zb, m, nb, v, nullptr);
                }
            }
        }
    }
}

// Sum the split-K slabs and apply the epilogue.  One thread per (m, 4 raw columns).
// Round 6: every load of a thread is issued before the first is consumed — the epilogue operands (bias, temb row, residual) at the top, the slabs four at a
// time through addresses clamped to the last slab.  As first written (one float4 per loop iteration, bias / temb / residual loaded where they are used, each
// behind its own branch) the kernel was a chain of splitk + ~6 serial memory round trips: 8-12 us for a launch that moves a few hundred KB, and a 1-scene step
// runs ~100 of them.  The slabs are still added in ascending z (one select per slab for the clamped ones): bit-identical sums.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GCParams p) {
    const int n4 = p.N / 4;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * n4) return;
    int m = (int)(idx / n4);
    int nb = (int)(idx - (long)m * n4) * 4;
    const bool geglu = p.epi == 1;
    if (geglu && (nb & 63) >= 32) return;  // gate columns are consumed by their value thread
    // ---- epilogue operands: all requested here ----
    const int sel = (p.temb && p.sel) ? *p.sel : 0;
    float bv[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0}, tv[4] = {0, 0, 0, 0};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { bv[j] = p.bias[nb + j]; if (geglu) bg[j] = p.bias[nb + 32 + j]; }
    }
    if (p.temb && !geglu) {
        const float* tb = p.temb + (long)sel * p.temb_sel_stride + (long)(m / p.rows_per_b) * p.temb_b_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) tv[j] = tb[nb + j];
    }
    const int ncol = geglu ? (nb >> 6) * 32 + (nb & 63) : nb;
    float rf[4] = {0, 0, 0, 0};
    if (p.R) {
        if (p.c_f32) {
            const float4 r4 = *(const float4*)((const float*)p.R + (long)m * p.ldr + ncol);
            rf[0] = r4.x; rf[1] = r4.y; rf[2] = r4.z; rf[3] = r4.w;
        } else {
            const uint2 rv = *(const uint2*)((const bf16_t*)p.R + (long)m * p.ldr + ncol);
            rf[0] = bf2f((bf16_t)(rv.x & 0xffff)); rf[1] = bf2f((bf16_t)(rv.x >> 16));
            rf[2] = bf2f((bf16_t)(rv.y & 0xffff)); rf[3] = bf2f((bf16_t)(rv.y >> 16));
        }
    }
    // ---- the slabs, four requests in flight ----
    float v[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
    const float* w0 = p.ws + (long)m * p.N + nb;
    const long zs = (long)p.M * p.N;
    const int last = p.splitk - 1;
    for (int z0 = 0; z0 < p.splitk; z0 += 4) {
        float4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* w = w0 + (long)min(z0 + i, last) * zs;
            a[i] = *(const float4*)w;
            if (geglu) b[i] = *(const float4*)(w + 32);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = z0 + i <= last;
            v[0] = ok ? v[0] + a[i].x : v[0]; v[1] = ok ? v[1] + a[i].y : v[1]; v[2] = ok ? v[2] + a[i].z : v[2]; v[3] = ok ? v[3] + a[i].w : v[3];
            if (geglu) { g[0] = ok ? g[0] + b[i].x : g[0]; g[1] = ok ? g[1] + b[i].y : g[1]; g[2] = ok ? g[2] + b[i].z : g[2]; g[3] = ok ? g[3] + b[i].w : g[3]; }
        }
    }
    // ---- epilogue: the arithmetic of epilogue_store (gemm_params.h), operand for operand ----
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (geglu) {
            float h = v[j], gt = g[j];
            if (p.bias) { h += bv[j]; gt += bg[j]; }
            o[j] = h * gelu_erf_f(gt);
        } else {
            float x = v[j];
            if (p.bias) x += bv[j];
            if (p.temb) x += tv[j];
            if (p.epi == 2) x = silu_f(x);
            o[j] = x;
        }
        if (p.R) o[j] += rf[j];
    }
    if (p.c_f32) {
        *(float4*)((float*)p.C + (long)m * p.ldc + ncol) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
        *(uint2*)((bf16_t*)p.C + (long)m * p.ldc + ncol) = ov;
    }
}

template <int BM, int BN, int BK, int WM, int WN, bool CONV, bool PIPE>
static int launch_one_(const GCParams& p, hipStream_t st) {
    constexpr size_t ring = (size_t)2 * (BM + BN) * (BK + 8) * sizeof(bf16_t), ctile = (size_t)BM * (BN + 8) * 2;
    constexpr size_t smem = ring > ctile ? ring : ctile;
    auto kern = gemm_conv_kernel<BM, BN, BK, WM, WN, CONV, PIPE>;
    if (int rc = ensure_dyn_smem((const void*)kern, smem, "gemm_conv")) return rc;
    GCParams q = p;
    q.mt = (p.M + BM - 1) / BM; q.nt = (p.N + BN - 1) / BN;
    const int swz = (int)opt(OPT_GEMM_SWZ);
    q.swz = swz && q.nt > 1 && q.mt >= 128;   // pays when A (activations) dwarfs W; mid-size M prefers W-tile reuse (measured)
    const unsigned nblk = q.swz ? (unsigned)((q.mt + 7) / 8 * 8 * q.nt) : (unsigned)(q.mt * q.nt);
    dim3 grid(nblk, 1, p.batch > 1 ? p.batch : p.splitk);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, st, q);
    char tag[96];
    snprintf(tag, sizeof tag, "gemm_conv_kernel<%d,%d,%d,%d,%d,%s>", BM, BN, BK, WM, WN, CONV ? "conv" : "gemm");
    return check_launch(tag);
}

// MDX_GEMM_PIPE (default 1): software-pipelined fragment reads for the 128x128x64 GEMM tile (the one with registers to spare).
template <int BM, int BN, int BK, int WM, int WN, bool CONV>
static int launch_one(const GCParams& p, hipStream_t st) {
    const int pipe = (int)opt(OPT_GEMM_PIPE);
    if constexpr (BM == 128 && BN == 128 && BK == 64 && WM == 2 && WN == 2) {
        if (pipe) return launch_one_<BM, BN, BK, WM, WN, CONV, true>(p, st);
    }
    return launch_one_<BM, BN, BK, WM, WN, CONV, false>(p, st);
}

// Routing (four main loops; options.h lists the switches, all settable in-process through mdx_set_option):
//   gemm_ws.hip   K = 320 projections / GEGLU with M >= 8192 (level 0 of the UNet): weights in registers, activations streamed
//   gemm_xl.hip   every conv with Cin % 64 == 0 and every GEMM with K % 64 == 0 that yields >= xl_min_tiles 256-row tiles
//                 (>= ~8 scenes per GPU at level 0, >= ~32 at the 7x13 level): 256 x {160, 256, 320} LDS-DMA tiles
//   this file     everything else: 128 / 64-row register-staged tiles (64 x 64 for small grids), split-K for the 7x13 / 4x7 levels at small batches
// (round 3 removed gemm_dma.hip — an LDS-DMA ring that never beat register staging — and gemm_pp.hip, whose 256 x 256 ping-pong tile
// was superseded by gemm_xl.hip at every batch where it used to be chosen; round 6 removed conv3x3.hip — one A slab shared by the three
// horizontal taps on a 128 x 128 tile — which by then only 1- and 2-scene calls reached and which ran 8-12 % behind the generic tile
// there: 56 vs 50 us for the level-0 320 -> 320 conv at one scene, profiles/r06_lat1_small_grids.log.)
int launch_gemm_conv(GCParams p, bool conv, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return MDX_OK;
    const bool geglu = p.epi == 1;
    {   // 16-byte epilogue accesses need 16-byte aligned rows of C and R and whole 8-column chunks
        const long nout = geglu ? p.N / 2 : p.N;
        const int wide_on = (int)opt(OPT_EPI_WIDE);
        p.wide = wide_on && !p.c_f32 && (nout % 8) == 0 && (p.ldc % 8) == 0 && (p.sC % 8) == 0 && (((uintptr_t)p.C) & 15) == 0 &&
                 (!p.R || ((p.ldr % 8) == 0 && (p.sR % 8) == 0 && (((uintptr_t)p.R) & 15) == 0));
    }
    if (!opt(OPT_LN_STATS)) { p.rowstat = nullptr; p.rowstat_parts = 0; p.ln_stats = nullptr; p.ln_stats_parts = 0; }   // A/B: the round-5 data flow
    if (p.rowstat) {
        // Row statistics of C for the LayerNorm that reads it next (MdxGemmDesc.rowstat_out): the K = 320 weight-stationary kernel emits them
        // from its store phase; every other route gets them from a small kernel over the finished C (part 0 = whole rows, the rest zeros).
        if (conv || p.batch > 1 || p.epi != 0 || p.c_f32 || p.Vt || p.rowstat_parts < 1) return set_error(MDX_EINVAL, "rowstat_out: plain 2-D GEMM with 16-bit C only");
        const int ws_mode_ = (int)opt(OPT_GEMM_WS);
        const bool ws_route = ws_mode_ > 0 && p.splitk <= 1 && ws_supported(p) && (ws_mode_ >= 2 || p.M >= 8192) &&
                              !((int)opt(OPT_XL_K320) && (int)opt(OPT_GEMM_XL) > 0) && (int)opt(OPT_GEMM_XL) < 2;
        if (!(ws_route && ws_emits_rowstat(p) && opt(OPT_LN_FUSE))) {
            GCParams q = p;
            q.rowstat = nullptr; q.rowstat_parts = 0;
            if (int rc = launch_gemm_conv(q, conv, st)) return rc;
            return launch_rowstat((const bf16_t*)p.C, p.M, p.N, p.ldc, p.rowstat, p.rowstat_parts, st);
        }
    }
    if (geglu && (p.N % 64) != 0) return set_error(MDX_EINVAL, "GEGLU needs packed N %% 64 == 0 (N=%d)", p.N);
    constexpr int impl = 0;
    // K = 320 projections with many rows: weights in registers, activations streamed (gemm_ws.hip).  MDX_GEMM_WS: 0 off, 1 when
    // M >= 8192 (default), 2 whenever supported.
    const int ws_mode = (int)opt(OPT_GEMM_WS);
    // Large shapes: the LDS-DMA quadrant-phase kernel (gemm_xl.hip).  MDX_GEMM_XL: 0 off, 1 cost model (default), 2 whenever supported.
    // Tile width: 256 columns, or 160 when that wastes fewer padded columns (N = 320 / 640 / 960 / 1920); a launch must give most of
    // the 256 CUs a tile (one workgroup per CU).  MDX_XL_K320 = 1 lets it take the K = 320 projections from gemm_ws.hip as well.
    const int xl_mode = (int)opt(OPT_GEMM_XL);
    const int xl_k320 = (int)opt(OPT_XL_K320);
    const int xl_min_tiles = (int)opt(OPT_XL_MIN_TILES);
    // Width choice: time model fitted on MI355X at 384 views (profiles/README.md, round 2): a tile costs a(bn) + b(bn) * K/64
    // microseconds — b falls with the tile width (operand bytes per MAC through the global -> LDS path), a (prologue + the
    // HBM-bound epilogue burst; the 320-wide tile stages C in two halves) rises — times the rounds of tiles over the 256 CUs.
    auto try_xl = [&](int& bn_out) -> bool {
        if (impl != 0 || xl_mode <= 0 || p.splitk > 1) return false;
        const int xl_bn = (int)opt(OPT_XL_BN);   // benchmarking: force a width
        double best = 1e300;
        bn_out = 0;
        const double nslab = p.K / 64.0;
        for (int bn : {320, 256, 160}) {
            if (xl_bn && bn != xl_bn) continue;
            if (!xl_supported(p, conv, bn)) continue;
            const long t = (long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn);
            if (xl_mode < 2 && t < xl_min_tiles) continue;
            const double a = bn == 320 ? 23.7 : bn == 256 ? 13.4 : 16.6, b = bn == 320 ? 1.896 : bn == 256 ? 1.565 : 1.116;
            const double c = (double)((t + 255) / 256) * (a + b * nslab);
            if (c < best) { best = c; bn_out = bn; }
        }
        return bn_out != 0;
    };
    // K = 320 GEGLU with many rows: gemm_ws.hip (384 views: 1642 us) vs the 256 x 256 XL tile (1694-1757 us).  Before the ring of
    // gemm_ws.hip really ran ahead (its DMA builtin drained the VM counter every slab: 1994 us) the XL tile was the faster one;
    // MDX_XL_GEGLU320=1 selects it again.
    const int xl_geglu320 = (int)opt(OPT_XL_GEGLU320);
    const bool geglu_xl = xl_geglu320 && xl_mode == 1 && impl == 0 && !conv && geglu && p.K == 320 && p.splitk <= 1 && ws_mode < 2 &&
                          xl_supported(p, false, 256) && (long)((p.M + 255) / 256) * ((p.N + 255) / 256) >= 1024;
    const bool ws_first = !conv && ws_mode > 0 && p.splitk <= 1 && ws_supported(p) && (ws_mode >= 2 || p.M >= 8192);
    const bool ws_taken = impl == 0 && !geglu_xl && ws_first && !(xl_k320 && xl_mode > 0);
    if (p.ln_eps > 0.f) {
        // LayerNorm fused into this GEMM (MdxGemmDesc.ln_eps): the weight-stationary kernel normalises in-kernel; every other route gets
        // the rows normalised (no affine part: it is in W / bias) into the caller's scratch first.
        if (conv || p.batch > 1) return set_error(MDX_EINVAL, "fused LayerNorm: plain 2-D GEMM only");
        if (!(ws_taken && ws_fuses_layernorm(p) && opt(OPT_LN_FUSE))) {
            if (!p.ln_scratch) return set_error(MDX_EINVAL, "fused LayerNorm: this shape is not normalised in-kernel and no ln_scratch was given");
            if (int rc = launch_layernorm_plain(p.A, p.ln_scratch, p.M, p.K, p.lda, p.lda, p.ln_eps, st)) return rc;
            p.A = p.ln_scratch; p.ln_eps = 0.f; p.ln_csum = nullptr; p.ln_stats = nullptr; p.ln_stats_parts = 0;
        }
    }
    if (xl_mode >= 2 && p.splitk <= 1 && p.batch <= 1 && !(p.K == 320 && !conv && !xl_k320)) {
        // "whenever supported" (tests / benchmarking): ahead of the automatic split-K below, which would otherwise claim small grids
        GCParams q = p;
        q.splitk = 1;
        int bn_f;
        const int keep = p.splitk;
        p.splitk = 1;
        const bool ok = try_xl(bn_f);
        p.splitk = keep;
        if (ok) return launch_gemm_xl(q, conv, bn_f, st);
    }
    if (geglu_xl) return launch_gemm_xl(p, false, 256, st);
    if (ws_taken) return launch_gemm_ws(p, st);
    int BM, BN;
    {
        BN = 128;
        if (!geglu && (p.N <= 64 || (p.N % 128 != 0 && p.N <= 192))) BN = 64;
        BM = p.M >= 2048 ? 128 : 64;
        // Small grids (round 6; the reference's own operating point is 1-4 scenes per call, where every launch of the step program lands here): with fewer
        // workgroups than the chip has room for, the SMALLER tile is the faster one — four 64 x 64 workgroups share a CU (37 KB of LDS each) where two
        // 128 x 128 ones fit, and a 4-wave workgroup alone on its CU has nothing to run beside its load phase.  Measured over every GEMM / conv shape
        // of the 1-, 2- and 4-scene step programs under forced tiles (tools/tile_sweep.py, profiles/r06_tile_sweep.log): plain GEMMs are fastest on
        // 64 x 64 up to ~1400 such tiles (9.5 vs 14.7 us at M 2100, N = K = 640; 18.7 vs 28.4 us at M 2184, N = K = 1280); implicit-GEMM convs up to
        // ~40 k (tile, slab) units of work, beyond that on 128 x 128 — never on the 64 x 128 tile rounds 1-5 gave every M < 2048; GEGLU (needs 128
        // columns) on 128 rows from M = 512.  Same k order in every tile: results do not depend on the choice (split-K aside).
        if (opt(OPT_GEMM_SMALL_TILES) && p.batch <= 1) {
            const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
            if (geglu) {
                BM = p.M >= 512 ? 128 : 64;
            } else if (!conv) {
                if (t64 <= 1408) BM = BN = 64;
            } else if (t64 * (long)((p.K + 63) / 64) <= 40000) {
                BM = BN = 64;
            } else if (BN == 128) {
                BM = 128;
            }
        }
        const int big = (int)opt(OPT_GEMM_BM256);
        if (big && BN == 128 && p.M >= big) BM = 256;
        const int fbm = (int)opt(OPT_GEMM_BM), fbn = (int)opt(OPT_GEMM_BN);
        if (fbm == 64 || fbm == 128) BM = fbm;
        if ((fbn == 64 && !geglu) || fbn == 128) BN = fbn;
    }
    long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.batch > 1 ? p.batch : 1);
    int splitk = 1;
    if (p.splitk > 0) {
        splitk = p.splitk;  // caller forced
    } else if (p.batch <= 1 && p.ws && tiles < 384 && p.K >= 1024 && (p.N % 4) == 0) {
        long want = (768 + tiles - 1) / tiles;
        long maxs = p.K / 512;  // keep >= 8 K-slabs per slice
        splitk = (int)min(min(want, maxs), 32L);
        if (splitk < 1) splitk = 1;
    }
    if (p.batch > 1) splitk = 1;
    if (splitk > 1) {  // fit the fp32 slabs into the caller's workspace
        long per = (long)p.M * p.N * (long)sizeof(float);
        long fit = per > 0 ? p.ws_bytes / per : 0;
        if (fit < splitk) splitk = fit < 1 ? 1 : (int)fit;
    }
    constexpr int BK = 64;   // split-K slices are multiples of the largest slab
    int kchunk = ((p.K + splitk - 1) / splitk + BK - 1) / BK * BK;
    splitk = (p.K + kchunk - 1) / kchunk;
    if (splitk > 1 && !p.ws) return set_error(MDX_EINVAL, "split-K needs a workspace");
    p.splitk = splitk;
    p.kchunk = kchunk;
    const int timing = (int)opt(OPT_GEMM_TIMING);
    p.timing = (timing && p.ws && splitk == 1) ? (unsigned long long*)p.ws : nullptr;
    if (splitk == 1) {
        int bn_xl;
        if (try_xl(bn_xl)) return launch_gemm_xl(p, conv, bn_xl, st);
    }
    if (impl == 0 && ws_first) return launch_gemm_ws(p, st);       // MDX_XL_K320 was set but the XL kernel declined the shape
    int rc;
    {
        const int bk = (int)opt(OPT_GEMM_BK);
#define MDX_GC2(BM_, BN_, BK_, WM_, WN_) (conv ? launch_one<BM_, BN_, BK_, WM_, WN_, true>(p, st) : launch_one<BM_, BN_, BK_, WM_, WN_, false>(p, st))
#define MDX_GC(BM_, BN_) (bk == 32 ? MDX_GC2(BM_, BN_, 32, 2, 2) : MDX_GC2(BM_, BN_, 64, 2, 2))
        if (BM == 256 && BN == 128) rc = (bk == 32) ? MDX_GC2(256, 128, 32, 4, 2) : MDX_GC2(256, 128, 64, 4, 2);
        else if (BM == 128 && BN == 128) rc = MDX_GC(128, 128);
        else if (BM == 128 && BN == 64) rc = MDX_GC(128, 64);
        else if (BM == 64 && BN == 128) rc = MDX_GC(64, 128);
        else rc = MDX_GC(64, 64);
#undef MDX_GC2
#undef MDX_GC
    }
    if (rc != MDX_OK) return rc;
    if (splitk > 1) {
        long n = (long)p.M * (p.N / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
        return check_launch("splitk_reduce_kernel", false);
    }
    return MDX_OK;
}

}  // namespace mdx

using namespace mdx;

static int check_common(int64_t K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr, const void* A, const void* W,
                        const void* C, int64_t N) {
    if (K % 8) return set_error(MDX_EINVAL, "K=%ld must be a multiple of 8", (long)K);
    if ((lda % 8) || (ldw % 8)) return set_error(MDX_EINVAL, "lda/ldw must be multiples of 8");
    if ((ldc % 4) || (ldr % 4)) return set_error(MDX_EINVAL, "ldc/ldr must be multiples of 4");
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7))
        return set_error(MDX_EINVAL, "operand pointers must be 16-byte (A, W) / 8-byte (C) aligned");
    (void)N;
    return MDX_OK;
}

// One XL launch for a batch-flattened GEMM (see mdx_gemm_bf16); MDX_EUNSUPPORTED when the XL kernel does not take the shape.
static int launch_gemm_flat(const mdx::GCParams& q, hipStream_t st) {
    using namespace mdx;
    const int xl_mode = (int)opt(OPT_GEMM_XL);
    if (xl_mode <= 0) return MDX_EUNSUPPORTED;
    double best = 1e300; int bn_best = 0;
    for (int bn : {320, 256, 160}) {
        if (!xl_supported(q, false, bn)) continue;
        const long t = (long)((q.M + 255) / 256) * ((q.N + bn - 1) / bn);
        if (t < 128) continue;
        const double a = bn == 320 ? 23.7 : bn == 256 ? 13.4 : 16.6, b = bn == 320 ? 1.896 : bn == 256 ? 1.565 : 1.116;
        const double c = (double)((t + 255) / 256) * (a + b * q.K / 64.0);
        if (c < best) { best = c; bn_best = bn; }
    }
    if (!bn_best) return MDX_EUNSUPPORTED;
    return launch_gemm_xl(q, false, bn_best, st);
}

extern "C" int mdx_gemm_bf16(const MdxGemmDesc* d, void* stream) {
    if (!d || !d->A || !d->W || !d->C) return set_error(MDX_EINVAL, "mdx_gemm_bf16: null operand");
    int rc = check_common(d->K, d->lda, d->ldw, d->ldc, d->ldr, d->A, d->W, d->C, d->N);
    if (rc) return rc;
    if (d->N % 4) {
        // ragged N (V^T with Tk % 4 != 0): columns N..roundup4(N)-1 are written as exact zeros
        // (their W rows are zero-filled), so they must exist in the row pitch and carry no epilogue.
        if (d->bias || d->temb || d->R || d->epilogue || d->splitk > 1 || d->ldc < (d->N + 3) / 4 * 4)
            return set_error(MDX_EINVAL, "N=%ld not a multiple of 4 needs a plain epilogue and ldc >= roundup4(N)", (long)d->N);
    }
    GCParams p = {};
    p.A = (const bf16_t*)d->A; p.W = (const bf16_t*)d->W; p.C = d->C; p.R = d->R;
    p.bias = d->bias; p.temb = d->temb; p.sel = d->sel_ptr; p.ws = d->ws;
    p.M = (int)d->M; p.N = (int)d->N; p.K = (int)d->K;
    p.lda = d->lda; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr;
    p.batch = d->batch > 1 ? (int)d->batch : 1;
    p.sA = d->sA; p.sW = d->sW; p.sC = d->sC; p.sR = d->sR;
    p.temb_sel_stride = d->temb_sel_stride; p.temb_b_stride = d->temb_b_stride;
    p.rows_per_b = d->rows_per_b > 0 ? (int)d->rows_per_b : 1;
    p.epi = (int)d->epilogue; p.splitk = (int)d->splitk; p.c_f32 = (int)d->c_is_f32; p.ws_bytes = d->ws_bytes;
    if (d->ln_eps > 0.0) {   // LayerNorm of the A rows fused in: W / bias carry gamma / beta, ln_csum the row sums of W (see include/mdx.h)
        if (!d->ln_csum || p.batch > 1 || d->splitk > 1 || d->c_is_f32 || d->temb || ((uintptr_t)d->ln_scratch & 15))
            return set_error(MDX_EINVAL, "mdx_gemm_bf16: fused LayerNorm needs ln_csum, one batch, no split-K / fp32 C / temb, 16-byte aligned ln_scratch");
        p.ln_eps = (float)d->ln_eps; p.ln_csum = d->ln_csum; p.ln_scratch = (bf16_t*)d->ln_scratch;
        if (d->ln_stats) {
            if (d->ln_stats_parts < 1 || ((uintptr_t)d->ln_stats & 7)) return set_error(MDX_EINVAL, "mdx_gemm_bf16: ln_stats needs ln_stats_parts >= 1 and 8-byte alignment");
            p.ln_stats = d->ln_stats; p.ln_stats_parts = (int)d->ln_stats_parts;
        }
    } else if (d->ln_stats) {
        return set_error(MDX_EINVAL, "mdx_gemm_bf16: ln_stats without ln_eps");
    }
    if (d->Wq) {
        if ((uintptr_t)d->Wq & 15) return set_error(MDX_EINVAL, "mdx_gemm_bf16: Wq must be 16-byte aligned");
        p.Wq = (const bf16_t*)d->Wq;
    }
    if (d->rowstat_out) {
        if (d->rowstat_parts < 1 || ((uintptr_t)d->rowstat_out & 7) || d->Vt || d->epilogue || d->c_is_f32 || p.batch > 1)
            return set_error(MDX_EINVAL, "mdx_gemm_bf16: rowstat_out needs rowstat_parts >= 1, 8-byte alignment, a plain epilogue, 16-bit C, one batch, no Vt");
        p.rowstat = d->rowstat_out; p.rowstat_parts = (int)d->rowstat_parts;
    }
    if (d->Vt) {   // fused q/k/v projection with a transposed V output: weight-stationary kernel only
        p.Vt = (bf16_t*)d->Vt; p.vt_from = (int)d->vt_from; p.vt_T = (int)d->vt_T; p.vt_ld = d->vt_ld; p.vt_stride = d->vt_stride;
        if (!ws_supported(p) || d->epilogue || d->R || (d->vt_from % 128) || d->vt_from <= 0 || d->vt_from >= d->N || d->vt_T <= 0 ||
            (d->vt_T % 8) || (d->M % d->vt_T) || (d->vt_ld % 8) || (d->vt_stride % 8) || ((uintptr_t)d->Vt & 15))
            return set_error(MDX_EINVAL, "mdx_gemm_bf16: transposed V output needs K=320, plain epilogue, no residual, vt_from %% 128 == 0, vt_T %% 8 == 0, aligned Vt");
        if (!opt(OPT_LN_STATS)) { p.ln_stats = nullptr; p.ln_stats_parts = 0; }
        if (p.ln_eps > 0.f && !opt(OPT_LN_FUSE)) {            // A/B switch: normalised copy first, then the plain fused q/k/v launch pair
            if (!p.ln_scratch) return set_error(MDX_EINVAL, "fused LayerNorm: LN_FUSE=0 needs ln_scratch");
            if (int rc2 = launch_layernorm_plain(p.A, p.ln_scratch, p.M, p.K, p.lda, p.lda, p.ln_eps, (hipStream_t)stream)) return rc2;
            p.A = p.ln_scratch; p.ln_eps = 0.f; p.ln_csum = nullptr; p.ln_stats = nullptr; p.ln_stats_parts = 0;
        }
        { GCParams q = p; const long nout = d->vt_from;      // wide-path check of the C part
          q.wide = (nout % 8) == 0 && (p.ldc % 8) == 0 && (((uintptr_t)p.C) & 15) == 0;
          return launch_gemm_ws(q, (hipStream_t)stream); }
    }
    // Batched GEMM with a shared A and W batches that are rows of ONE matrix (the per-view V^T projections at levels 1 and 2: 384
    // products of 640 x 350 x 640): run it as a single GEMM over all batches' columns on the XL main loop, the epilogue scattering
    // each column to its batch (GCParams.col_split).  MDX_GEMM_FLATTEN=0 keeps the per-batch launches.
    const int flatten = (int)opt(OPT_GEMM_FLATTEN);
    if (flatten && p.batch > 1 && p.sA == 0 && p.sW == (long)p.N * p.ldw && !p.bias && !p.temb && !p.R && !p.epi && !p.c_f32 && p.splitk <= 1 &&
        (long)p.batch * p.N < 0x7fffff00L && ((long)p.batch * p.N) % 4 == 0) {
        GCParams q = p;
        q.col_split = p.N; q.N = p.batch * p.N; q.batch = 1; q.sW = 0; q.wide = 0;
        int rcq = launch_gemm_flat(q, (hipStream_t)stream);
        if (rcq != MDX_EUNSUPPORTED) return rcq;
    }
    return launch_gemm_conv(p, false, (hipStream_t)stream);
}

extern "C" int mdx_conv2d_bf16(const MdxConvDesc* d, void* stream) {
    if (!d || !d->X || !d->Wt || !d->Y) return set_error(MDX_EINVAL, "mdx_conv2d_bf16: null operand");
    if (d->Cin % 8) return set_error(MDX_EINVAL, "mdx_conv2d_bf16: Cin=%ld must be a multiple of 8", (long)d->Cin);
    int64_t K = d->kh * d->kw * d->Cin;
    int rc = check_common(K, d->ldx, K, d->ldy, d->ldr, d->X, d->Wt, d->Y, d->Cout);
    if (rc) return rc;
    if (d->epilogue == MDX_EPI_GEGLU) return set_error(MDX_EINVAL, "conv has no GEGLU epilogue");
    if (d->Cout % 4) return set_error(MDX_EINVAL, "mdx_conv2d_bf16: Cout=%ld must be a multiple of 4", (long)d->Cout);
    GCParams p = {};
    p.A = (const bf16_t*)d->X; p.W = (const bf16_t*)d->Wt; p.C = d->Y; p.R = d->R;
    p.bias = d->bias; p.temb = d->temb; p.sel = d->sel_ptr; p.ws = d->ws;
    p.M = (int)(d->B * d->Ho * d->Wo); p.N = (int)d->Cout; p.K = (int)K;
    p.lda = d->ldx; p.ldw = K; p.ldc = d->ldy; p.ldr = d->ldr;
    p.batch = 1;
    p.temb_sel_stride = d->temb_sel_stride; p.temb_b_stride = d->temb_b_stride;
    p.rows_per_b = (int)(d->Ho * d->Wo);
    p.epi = (int)d->epilogue; p.splitk = (int)d->splitk; p.c_f32 = 0; p.ws_bytes = d->ws_bytes;
    p.Hi = (int)d->Hi; p.Wi = (int)d->Wi; p.Cin = (int)d->Cin; p.Ho = (int)d->Ho; p.Wo = (int)d->Wo;
    p.kh = (int)d->kh; p.kw = (int)d->kw; p.sh = (int)d->sh; p.sw = (int)d->sw; p.ph = (int)d->ph; p.pw = (int)d->pw;
    const int cim = (int)opt(OPT_CONV_CIMAJOR);
    p.cimajor = (cim && d->kh * d->kw > 1 && d->Cin % 64 == 0) ? 1 : 0;
    return launch_gemm_conv(p, true, (hipStream_t)stream);
}
