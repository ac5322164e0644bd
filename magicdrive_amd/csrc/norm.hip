// norm.hip — GroupNorm(+SiLU) and LayerNorm on channels-last bf16 tensors, fp32 statistics.
//
// GroupNorm: ATen native_group_norm + silu as used by ResnetBlock2D (diffusers/models/resnet.py:
//   596-598, 626-630), Transformer2DModel.norm (transformer_2d.py:278, eps 1e-6) and conv_norm_out
//   (unet_2d_condition_multiview.py:519-521).  One 1024-thread workgroup per (batch, group); the group's
//   HW x (C/G) slab (<= 1400 x 80 elements, L2 resident) is read twice: pivot-shifted sum / sum of
//   squares, then normalise+affine(+SiLU)+store.
// LayerNorm: nn.LayerNorm over C (attention.py:85,104,120; blocks.py:67-71), one wave per token
//   row, 16-byte loads, shuffle reductions.
#include "common.h"
#include "launch.h"
#include "options.h"

namespace mdx {

struct GNParams {
    const bf16_t* X; bf16_t* Y; const float* gamma; const float* beta;
    int B, HW, C, G; long ldx, ldy; float eps; int silu;
};

constexpr int GN_THREADS = 1024;
constexpr int GN_MAX_CPG = 2560;     // channels per group the one-launch kernel keeps gamma / beta for in LDS (mdx_groupnorm_bf16 rejects more)

// block-wide sum of two values at once (16 waves)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[wave] = a; red[16 + wave] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int i = 0; i < GN_THREADS / 64; ++i) { ta += red[i]; tb += red[16 + i]; }
    a = ta; b = tb;
}

// One workgroup (16 waves) per (batch, group).  Pass 1: sum and sum of squares of (x - pivot), pivot = the
// group's first element (shifting by a sample of the data keeps E[d^2] - E[d]^2 well conditioned in fp32);
// pass 2: normalise + affine (+SiLU) + store.  32-bit index math, VEC channels per thread-iteration.
template <int VEC>
__global__ __launch_bounds__(GN_THREADS) void groupnorm_kernel(GNParams p) {
    __shared__ float red[32];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.G;
    const int vpp = cpg / VEC;                 // vectors per pixel in this group
    const int nvec = p.HW * vpp;
    const bf16_t* xb = p.X + (long)b * p.HW * p.ldx + (long)g * cpg;
    bf16_t* yb = p.Y + (long)b * p.HW * p.ldy + (long)g * cpg;
    const int ldx = (int)p.ldx, ldy = (int)p.ldy;

    auto load = [&](int px, int cv, float* v) {
        const bf16_t* s = xb + px * ldx + cv;
        if constexpr (VEC == 8) {
            Frag8 f; f.u = *(const uint4*)s;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f.h[e]);
        } else if constexpr (VEC == 4) {
            uint2 u = *(const uint2*)s;
            v[0] = bf2f((bf16_t)(u.x & 0xffff)); v[1] = bf2f((bf16_t)(u.x >> 16));
            v[2] = bf2f((bf16_t)(u.y & 0xffff)); v[3] = bf2f((bf16_t)(u.y >> 16));
        } else if constexpr (VEC == 2) {
            uint32_t u = *(const uint32_t*)s;
            v[0] = bf2f((bf16_t)(u & 0xffff)); v[1] = bf2f((bf16_t)(u >> 16));
        } else {
            v[0] = bf2f(*s);
        }
    };

    // Loads are issued GN_U at a time before any is consumed (the loop is latency-, not bandwidth-bound:
    // a group's slab sits in L2), and a group that fits in GN_U vectors per thread is read only ONCE.
    // Round 6 (this kernel is what 1-4 scene calls run, 88 launches per step): (pixel, channel vector) of a thread's vectors advance incrementally —
    // vector i + 1024 is q pixels and r vectors further, (q, r) = divmod(1024, vpp) — instead of one 32-bit division by a run-time divisor per
    // vector and pass (~30 VALU operations each, 14 per thread at level 0: ~3 us of a 12 us kernel with 4 waves per SIMD), and the group's
    // gamma / beta wait in LDS (fetched beside the data, behind the reduction's barriers) instead of being a second global round trip in pass 2.
    constexpr int GN_U = (VEC == 8) ? 4 : 8;          // <= 32 live values per thread (1024-thread WG: 128 VGPR budget)
    __shared__ float gb[2][GN_MAX_CPG];               // (a global or a "flat" load of gamma / beta in pass 2 waits on vmcnt(0), i.e. on the PREVIOUS vector's store: the
    for (int c = threadIdx.x; c < cpg; c += GN_THREADS) { gb[0][c] = p.gamma[g * cpg + c]; gb[1][c] = p.beta[g * cpg + c]; }   // stores ran one round trip at a time)
    const float pivot = bf2f(xb[0]);
    float s1 = 0.f, s2 = 0.f;
    float v[GN_U][VEC];
    const bool single = nvec <= GN_U * GN_THREADS;
    const int stepq = GN_THREADS / vpp, stepr = GN_THREADS - stepq * vpp;        // vector i + 1024 = pixel + stepq, vector-in-pixel + stepr (mod vpp)
    const int bigq = (GN_U * GN_THREADS) / vpp, bigr = GN_U * GN_THREADS - bigq * vpp;
    const int px0 = (int)threadIdx.x / vpp, r0 = (int)threadIdx.x - px0 * vpp;    // ONE division per thread
    {
        int pxb = px0, rb = r0;
        for (int base = threadIdx.x; base < nvec; base += GN_U * GN_THREADS) {
            int px = pxb, r = rb;
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                // unconditional loads (a vector past the end re-reads vector 0 and is replaced by the pivot): behind a per-lane branch every load got
                // its own `s_waitcnt vmcnt(0)` inside the branch — the batch of GN_U loads was GN_U serial round trips (18 vs 10 us at level 0)
                const int i = base + u * GN_THREADS;
                const bool ok = i < nvec;
                load(ok ? px : 0, ok ? r * VEC : 0, v[u]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[u][e] = ok ? v[u][e] : pivot;
                px += stepq; r += stepr;
                if (r >= vpp) { r -= vpp; ++px; }
            }
#pragma unroll
            for (int u = 0; u < GN_U; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { float dlt = v[u][e] - pivot; s1 += dlt; s2 += dlt * dlt; }
            pxb += bigq; rb += bigr;
            if (rb >= vpp) { rb -= vpp; ++pxb; }
        }
    }
    block_sum2(s1, s2, red);                          // (its barriers also publish gb)
    const float n = (float)p.HW * (float)cpg;
    const float md = s1 / n;                          // mean of (x - pivot)
    const float mean = pivot + md;
    const float var = fmaxf(s2 / n - md * md, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    int pxb = px0, rb = r0;
    for (int base = threadIdx.x; base < nvec; base += GN_U * GN_THREADS) {
        if (!single) {
            int px = pxb, r = rb;
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const int i = base + u * GN_THREADS;
                const bool ok = i < nvec;
                load(ok ? px : 0, ok ? r * VEC : 0, v[u]);
                px += stepq; r += stepr;
                if (r >= vpp) { r -= vpp; ++px; }
            }
        }
        int px = pxb, r = rb;
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int i = base + u * GN_THREADS;
            const int cpx = px, cv = r * VEC;
            px += stepq; r += stepr;
            if (r >= vpp) { r -= vpp; ++px; }
            if (i >= nvec) continue;
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float ga = gb[0][cv + e], be = gb[1][cv + e];
                float y = (v[u][e] - mean) * rstd * ga + be;
                if (p.silu) y = silu_f(y);
                o[e] = y;
            }
            bf16_t* dptr = yb + cpx * ldy + cv;
            if constexpr (VEC == 8) {
                uint4 u4; u4.x = pack2bf(o[0], o[1]); u4.y = pack2bf(o[2], o[3]); u4.z = pack2bf(o[4], o[5]); u4.w = pack2bf(o[6], o[7]);
                *(uint4*)dptr = u4;
            } else if constexpr (VEC == 4) {
                uint2 u2; u2.x = pack2bf(o[0], o[1]); u2.y = pack2bf(o[2], o[3]);
                *(uint2*)dptr = u2;
            } else if constexpr (VEC == 2) {
                *(uint32_t*)dptr = pack2bf(o[0], o[1]);
            } else {
                *dptr = f2bf(o[0]);
            }
        }
        pxb += bigq; rb += bigr;
        if (rb >= vpp) { rb -= vpp; ++pxb; }
    }
}

// ------------------------------------------------------------------------------------------------
// Two-stage GroupNorm for the big (HBM/Infinity-Cache-bound) feature maps: pure streaming, every global access a full 16-byte-per-lane
// coalesced row segment (the one-workgroup-per-group kernel above reads 20-byte pieces at a 640-byte stride when C/G = 10), several
// loads in flight per lane, no LDS staging.  Deterministic: no atomics, partials combined in a fixed order (Chan).
// Thread layout of both stages: the workgroup is ROWS x (C/8) threads (ROWS = 256 / (C/8), at least 1: 160 - 320 threads; measured:
// 320-thread workgroups with more rows are slower), thread
// (r, cc) owns the SAME 8 channels cc*8.. of pixel rows r, r + ROWS, ... of its chunk — so the per-channel state (sums in stage 1,
// scale / shift in stage 2) lives in registers.  C/8 > 320 (C > 2560): 320 threads, a thread owns NCV = 2 channel vectors.
//   stage 1  gn_stats_kernel : grid (chunks, B); per-channel sums of (x - pivot_g) and (x - pivot_g)^2 in registers (pivot_g = the
//            group's first element of the chunk: keeps E[d^2] - E[d]^2 well conditioned in fp32), reduced over rows and over the
//            group's channels through LDS in a fixed order -> (n, mean, M2) per (b, chunk, g)
//   stage 2  gn_apply_kernel : grid (chunks, B); combines the chunk partials of its b, then y = [silu](x * scale[c] + shift[c]).
struct GN2Params {
    const bf16_t* X; bf16_t* Y; const float* gamma; const float* beta; float* part;   // part: [B][nchunk][G][3]
    int B, HW, C, G, PCH, nchunk; long ldx, ldy; float eps; int silu;
    int rev;      // statistics pass walks images and chunks in DEscending order (see mdx_groupnorm_bf16)
    const float* stat;   // [B][G][2] (mean, rstd) from gn_finalize_kernel, or nullptr: the apply pass combines the chunk partials itself
};

constexpr int GN2_NT = 320;       // most threads per workgroup
constexpr int GN2_MAXC = 5120;    // channels (stage 1 LDS: 2 x rows x C floats, rows x C <= 2560 NCV)
constexpr int GN2_U = 4;          // pixel rows in flight per thread

__host__ __device__ __forceinline__ int gn2_rows(int C8) { return C8 >= 256 ? 1 : 256 / C8; }

template <int NCV>
__global__ __launch_bounds__(GN2_NT) void gn_stats_kernel(GN2Params p) {
    __shared__ float ls1[GN2_NT * 8 * NCV], ls2[GN2_NT * 8 * NCV];       // [row][C] sums (rows * C <= 2560 * NCV)
    __shared__ float lpiv[256];                                    // per group pivot (G <= 256)
    const int chunk = p.rev ? p.nchunk - 1 - (int)blockIdx.x : (int)blockIdx.x, b = p.rev ? p.B - 1 - (int)blockIdx.y : (int)blockIdx.y, tid = threadIdx.x;
    const int px0 = chunk * p.PCH;
    const int npx = min(p.PCH, p.HW - px0);
    const int C8 = p.C / 8, cpg = p.C / p.G;
    const int rows = gn2_rows(C8);
    const bf16_t* xb = p.X + ((long)b * p.HW + px0) * p.ldx;
    if (tid < p.G) lpiv[tid] = bf2f(xb[tid * cpg]);
    __syncthreads();
    const int r0 = NCV == 1 ? tid / C8 : 0;
    const bool active = NCV == 1 ? r0 < rows : true;
    float s1[NCV][8], s2[NCV][8], pv[NCV][8];
    int cc[NCV];
#pragma unroll
    for (int k = 0; k < NCV; ++k) {
        cc[k] = NCV == 1 ? tid - r0 * C8 : tid + GN2_NT * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[k][e] = 0.f; s2[k][e] = 0.f; pv[k][e] = (cc[k] < C8) ? lpiv[(cc[k] * 8 + e) / cpg] : 0.f; }
    }
    if (active) {
        const int ldx = (int)p.ldx;
        int ccl[NCV];
#pragma unroll
        for (int k = 0; k < NCV; ++k) ccl[k] = cc[k] < C8 ? cc[k] : C8 - 1;
        for (int px = r0; px < npx; px += rows * GN2_U) {
            // unconditional loads from clamped addresses, masked on use (a guarded load becomes its own branch with a full wait inside)
            Frag8 f[GN2_U][NCV];
#pragma unroll
            for (int u = 0; u < GN2_U; ++u) {
                const int pxu = px + u * rows < npx ? px + u * rows : npx - 1;
#pragma unroll
                for (int k = 0; k < NCV; ++k) f[u][k].u = *(const uint4*)(xb + (long)pxu * ldx + ccl[k] * 8);
            }
#pragma unroll
            for (int u = 0; u < GN2_U; ++u)
#pragma unroll
                for (int k = 0; k < NCV; ++k) {
                    const float m = (px + u * rows < npx && cc[k] < C8) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = (bf2f(f[u][k].h[e]) - pv[k][e]) * m; s1[k][e] += d; s2[k][e] = __builtin_fmaf(d, d, s2[k][e]); }
                }
        }
#pragma unroll
        for (int k = 0; k < NCV; ++k)
            if (cc[k] < C8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { ls1[r0 * p.C + cc[k] * 8 + e] = s1[k][e]; ls2[r0 * p.C + cc[k] * 8 + e] = s2[k][e]; }
            }
    }
    __syncthreads();
    if (tid < p.G) {                            // fixed order: channels of the group, rows inside a channel
        float a1 = 0.f, a2 = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c)
            for (int r = 0; r < rows; ++r) { a1 += ls1[r * p.C + c]; a2 += ls2[r * p.C + c]; }
        const float n = (float)npx * (float)cpg;
        float* o = p.part + (((long)b * p.nchunk + chunk) * p.G + tid) * 3;
        o[0] = n; o[1] = lpiv[tid] + a1 / n; o[2] = fmaxf(a2 - a1 * a1 / n, 0.f);
    }
}

// Many chunks per image (few, large images: the VAE decoder's 6 x 224 x 400 maps split into ~620 chunks each): every apply workgroup combining all
// of its image's partials itself is 620 dependent L2 round trips per workgroup — the VAE's GroupNorms ran at 0.27-0.46 TB/s, 10 of the
// decoder's 17.6 ms (round 5, tools/vaeone.py).  Here ONE wave per (image, group) combines them once: lane l takes chunks l, l + 64, ... in
// ascending order, then the 64 lane results are combined in a fixed tree (Chan) — deterministic — and (mean, rstd) go to a table.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* part, float* stat, int nchunk, int G, float eps) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float na = 0.f, ma = 0.f, qa = 0.f;
    const float* pp = part + ((long)b * nchunk * G + g) * 3;
    for (int k = lane; k < nchunk; k += 64) {
        const float nb = pp[(long)k * G * 3], mb = pp[(long)k * G * 3 + 1], qb = pp[(long)k * G * 3 + 2];
        if (nb > 0.f) {
            const float nn = na + nb, dl = mb - ma;
            ma += dl * nb / nn; qa += qb + dl * dl * na * nb / nn; na = nn;
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                        // lane l absorbs lane l + o (the lower index keeps the result: a fixed order)
        const float nb = __shfl_down(na, o, 64), mb = __shfl_down(ma, o, 64), qb = __shfl_down(qa, o, 64);
        if (nb > 0.f) {
            const float nn = na + nb, dl = mb - ma;
            ma += dl * nb / nn; qa += qb + dl * dl * na * nb / nn; na = nn;
        }
    }
    if (lane == 0) { stat[((long)b * G + g) * 2] = ma; stat[((long)b * G + g) * 2 + 1] = rsqrtf(qa / na + eps); }
}

template <int NCV>
__global__ __launch_bounds__(GN2_NT) void gn_apply_kernel(GN2Params p) {
    __shared__ float gmean[256], grstd[256];
    const int chunk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < p.G && p.stat) {
        gmean[tid] = p.stat[((long)b * p.G + tid) * 2];
        grstd[tid] = p.stat[((long)b * p.G + tid) * 2 + 1];
    } else if (tid < p.G) {
        float na = 0.f, ma = 0.f, qa = 0.f;
        const float* pp = p.part + ((long)b * p.nchunk * p.G + tid) * 3;
        for (int k = 0; k < p.nchunk; ++k) {
            const float nb = pp[(long)k * p.G * 3], mb = pp[(long)k * p.G * 3 + 1], qb = pp[(long)k * p.G * 3 + 2];
            if (nb > 0.f) {
                const float nn = na + nb, dl = mb - ma;
                ma += dl * nb / nn; qa += qb + dl * dl * na * nb / nn; na = nn;
            }
        }
        gmean[tid] = ma;
        grstd[tid] = rsqrtf(qa / na + p.eps);
    }
    __syncthreads();
    const int C8 = p.C / 8, cpg = p.C / p.G;
    const int rows = gn2_rows(C8);
    const int r0 = NCV == 1 ? tid / C8 : 0;
    if (NCV == 1 && r0 >= rows) return;
    float sc[NCV][8], sh[NCV][8];
    int cc[NCV], ccl[NCV];
#pragma unroll
    for (int k = 0; k < NCV; ++k) {
        cc[k] = NCV == 1 ? tid - r0 * C8 : tid + GN2_NT * k;
        ccl[k] = cc[k] < C8 ? cc[k] : C8 - 1;
        const float4 g0 = *(const float4*)(p.gamma + ccl[k] * 8), g1 = *(const float4*)(p.gamma + ccl[k] * 8 + 4);
        const float4 b0 = *(const float4*)(p.beta + ccl[k] * 8), b1 = *(const float4*)(p.beta + ccl[k] * 8 + 4);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (ccl[k] * 8 + e) / cpg;
            const float s = grstd[g] * gv[e];
            sc[k][e] = s; sh[k][e] = bv[e] - gmean[g] * s;
        }
    }
    const int px0 = chunk * p.PCH;
    const int npx = min(p.PCH, p.HW - px0);
    const bf16_t* xb = p.X + ((long)b * p.HW + px0) * p.ldx;
    bf16_t* yb = p.Y + ((long)b * p.HW + px0) * p.ldy;
    const int ldx = (int)p.ldx, ldy = (int)p.ldy;
    for (int px = r0; px < npx; px += rows * GN2_U) {
        Frag8 f[GN2_U][NCV];
#pragma unroll
        for (int u = 0; u < GN2_U; ++u) {
            const int pxu = px + u * rows < npx ? px + u * rows : npx - 1;
#pragma unroll
            for (int k = 0; k < NCV; ++k) f[u][k].u = *(const uint4*)(xb + (long)pxu * ldx + ccl[k] * 8);
        }
#pragma unroll
        for (int u = 0; u < GN2_U; ++u)
#pragma unroll
            for (int k = 0; k < NCV; ++k) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = __builtin_fmaf(bf2f(f[u][k].h[e]), sc[k][e], sh[k][e]);
                    if (p.silu) y = silu_f(y);
                    o[e] = y;
                }
                uint4 w; w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]); w.z = pack2bf(o[4], o[5]); w.w = pack2bf(o[6], o[7]);
                if (px + u * rows < npx && cc[k] < C8) *(uint4*)(yb + (long)(px + u * rows) * ldy + cc[k] * 8) = w;
            }
    }
}

struct LNParams {
    const bf16_t* X; bf16_t* Y; const float* gamma; const float* beta;
    int M, C; long ldx, ldy; float eps;
};

// RPW rows per wave, processed together so that RPW * MAXV 16-byte loads per lane are in flight (one row per wave leaves the kernel
// latency-bound at ~60 % of the streaming rate); C % 8 == 0, C <= 8 * 64 * MAXV.  Two-pass statistics in registers, shuffle reductions.
// AFFINE = false: plain normalisation, gamma / beta not read (launch_layernorm_plain).  A compile-time switch: as a run-time test on
// p.gamma every gamma / beta load became its own branch with a full wait and the kernel ran 2-3 x slower at C = 640 / 1280 (round 3).
template <int MAXV, int RPW, bool AFFINE>
__global__ __launch_bounds__(256) void layernorm_kernel(LNParams p) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= p.M) return;
    const int nv = p.C / 8;
    // every load is unconditional (clamped address, value masked afterwards): as `if (valid) load` the compiler wraps each load in its own
    // exec-masked branch with a full wait inside, i.e. one memory latency after the other
    float v[RPW][MAXV][8];
    float s1[RPW];
    Frag8 f[RPW][MAXV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < p.M ? row0 + r : p.M - 1;
        const bf16_t* x = p.X + (long)row * p.ldx;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            f[r][i].u = *(const uint4*)(x + (c < nv ? c : 0) * 8);
        }
    }
    float g[MAXV][8], bt[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i, cl = c < nv ? c : 0;
        float4 g0, g1, b0, b1;
        if constexpr (AFFINE) {
            g0 = *(const float4*)(p.gamma + cl * 8); g1 = *(const float4*)(p.gamma + cl * 8 + 4);
            b0 = *(const float4*)(p.beta + cl * 8); b1 = *(const float4*)(p.beta + cl * 8 + 4);
        } else {
            g0 = g1 = make_float4(1.f, 1.f, 1.f, 1.f); b0 = b1 = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
        bt[i][0] = b0.x; bt[i][1] = b0.y; bt[i][2] = b0.z; bt[i][3] = b0.w; bt[i][4] = b1.x; bt[i][5] = b1.y; bt[i][6] = b1.z; bt[i][7] = b1.w;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        s1[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const bool ok = lane + 64 * i < nv;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[r][i][e] = ok ? bf2f(f[r][i].h[e]) : 0.f;
        }
    }
    const float invC = 1.0f / (float)p.C;
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[r] += v[r][i][e];           // lanes / vectors past C hold zeros
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) mean[r] = wave_sum(s1[r]) * invC;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dlt = v[r][i][e] - mean[r]; s2 = __builtin_fmaf(dlt, dlt, s2); }
            }
        }
        rstd[r] = s2;
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) rstd[r] = rsqrtf(wave_sum(rstd[r]) * invC + p.eps);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        if (row0 + r >= p.M) break;
        bf16_t* y = p.Y + (long)(row0 + r) * p.ldy;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                Frag8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f.h[e] = f2bf((v[r][i][e] - mean[r]) * rstd[r] * g[i][e] + bt[i][e]);
                *(uint4*)(y + c * 8) = f.u;
            }
        }
    }
}

}  // namespace mdx

using namespace mdx;

extern "C" int mdx_groupnorm_bf16(const MdxGroupNormDesc* d, void* stream) {
    if (!d || !d->X || !d->Y || !d->gamma || !d->beta) return set_error(MDX_EINVAL, "mdx_groupnorm_bf16: null operand");
    if (d->G <= 0 || d->C % d->G) return set_error(MDX_EINVAL, "groupnorm: C=%ld not divisible by G=%ld", (long)d->C, (long)d->G);
    if (d->B <= 0 || d->HW <= 0) return MDX_OK;
    if (d->C / d->G > GN_MAX_CPG) return set_error(MDX_EUNSUPPORTED, "groupnorm: %ld channels per group (at most %d)", (long)(d->C / d->G), GN_MAX_CPG);
    GNParams p;
    p.X = (const bf16_t*)d->X; p.Y = (bf16_t*)d->Y; p.gamma = d->gamma; p.beta = d->beta;
    p.B = (int)d->B; p.HW = (int)d->HW; p.C = (int)d->C; p.G = (int)d->G; p.ldx = d->ldx; p.ldy = d->ldy;
    p.eps = (float)d->eps; p.silu = (int)d->silu;
    const int cpg = p.C / p.G;
    hipStream_t st = (hipStream_t)stream;
    // big maps: two-stage streaming path (needs 16-byte rows, a partials workspace, and one thread per group: both kernels index
    // pivots / partials / mean-rstd with `tid < G`, so G must not exceed the workgroup — rows * C8 threads, e.g. 160 at C = 1280)
    const int two_stage = (int)opt(OPT_GN_TWO_STAGE);
    const long elems = (long)p.HW * p.C;
    const int nt_ws = (p.C % 8 == 0) ? ((p.C / 8) > GN2_NT ? GN2_NT : gn2_rows(p.C / 8) * (p.C / 8)) : 0;
    // Small tensors (round 6: 1-4 scenes per call) stay on the one-workgroup-per-(image, group) kernel below: ONE launch of ~8-20 us instead of two of
    // ~10 us each, whatever the size; from ~4 M elements on the two streaming passes win (tools/lat1.py rows with GN_TWO_STAGE = 0 / 1 at 1, 2 and 4
    // scenes, profiles/r06_lat1_small_grids.log: 6 x 350 x 640: 12.5 vs 18.8 us; 12 x 1400 x 320: 29.4 vs 24.1 us).
    const bool small = (long)p.B * elems <= opt(OPT_GN_ONE_KERNEL_ELEMS);
    if (two_stage && !small && d->ws && elems >= 32768 && p.C % 8 == 0 && p.C <= GN2_MAXC && p.G <= nt_ws && p.ldx % 8 == 0 && p.ldy % 8 == 0 &&
        ((uintptr_t)p.X & 15) == 0 && ((uintptr_t)p.Y & 15) == 0) {
        GN2Params q;
        q.X = p.X; q.Y = p.Y; q.gamma = p.gamma; q.beta = p.beta; q.part = (float*)d->ws;
        q.B = p.B; q.HW = p.HW; q.C = p.C; q.G = p.G; q.ldx = p.ldx; q.ldy = p.ldy; q.eps = p.eps; q.silu = p.silu;
        // Read order vs the 256 MiB Infinity Cache: the producer wrote the tensor in ascending order, so its TAIL is what is still
        // cached — the statistics pass walks the tensor from the end, and ends at the head, which is where the apply pass (ascending)
        // starts: both passes begin on cached lines instead of on the lines evicted longest ago.  GN_REVERSE = 0 restores ascending.
        q.rev = (int)opt(OPT_GN_REVERSE);
        q.stat = nullptr;
        // chunks: about 4096 workgroups in the grid, at least GN2_U passes of the workgroup's rows each
        const int C8 = p.C / 8, rows = gn2_rows(C8);
        const int nt = C8 > GN2_NT ? GN2_NT : rows * C8;
        int target = (int)(4096 / p.B); if (target < 1) target = 1;
        int pch = (p.HW + target - 1) / target;
        if (pch < rows * GN2_U) pch = rows * GN2_U;
        pch = (pch + rows - 1) / rows * rows;
        if (pch > p.HW) pch = p.HW;
        q.PCH = pch;
        q.nchunk = (p.HW + q.PCH - 1) / q.PCH;
        const long part_bytes = (long)p.B * q.nchunk * p.G * 3 * (long)sizeof(float);
        if (part_bytes <= d->ws_bytes) {
            dim3 grid2(q.nchunk, p.B);
            if (C8 > GN2_NT) hipLaunchKernelGGL(gn_stats_kernel<2>, grid2, dim3(nt), 0, st, q);
            else hipLaunchKernelGGL(gn_stats_kernel<1>, grid2, dim3(nt), 0, st, q);
            int rc = check_launch("gn_stats_kernel", false);
            if (rc) return rc;
            // more than GN_FINALIZE_CHUNKS chunks per image: combine the partials ONCE (gn_finalize_kernel) instead of in every apply workgroup
            q.stat = nullptr;
            const long stat_bytes = (long)p.B * p.G * 2 * (long)sizeof(float);
            const long fin = opt(OPT_GN_FINALIZE_CHUNKS);
            if (fin > 0 && q.nchunk > fin && part_bytes + stat_bytes <= d->ws_bytes) {
                float* stat = (float*)((char*)d->ws + part_bytes);
                hipLaunchKernelGGL(gn_finalize_kernel, dim3(p.G, p.B), dim3(64), 0, st, q.part, stat, q.nchunk, p.G, p.eps);
                rc = check_launch("gn_finalize_kernel", false);
                if (rc) return rc;
                q.stat = stat;
            }
            if (C8 > GN2_NT) hipLaunchKernelGGL(gn_apply_kernel<2>, grid2, dim3(nt), 0, st, q);
            else hipLaunchKernelGGL(gn_apply_kernel<1>, grid2, dim3(nt), 0, st, q);
            return check_launch("gn_stats_kernel+gn_apply_kernel");
        }
    }
    // vector width limited by cpg and by the alignment of every group start / row stride
    int vec = 1;
    for (int v = 8; v > 1; v >>= 1) {
        if (cpg % v == 0 && p.ldx % v == 0 && p.ldy % v == 0 && ((uintptr_t)p.X % (2 * v)) == 0 && ((uintptr_t)p.Y % (2 * v)) == 0) { vec = v; break; }
    }
    dim3 grid(p.G, p.B);
    switch (vec) {
        case 8: hipLaunchKernelGGL(groupnorm_kernel<8>, grid, dim3(GN_THREADS), 0, st, p); break;
        case 4: hipLaunchKernelGGL(groupnorm_kernel<4>, grid, dim3(GN_THREADS), 0, st, p); break;
        case 2: hipLaunchKernelGGL(groupnorm_kernel<2>, grid, dim3(GN_THREADS), 0, st, p); break;
        default: hipLaunchKernelGGL(groupnorm_kernel<1>, grid, dim3(GN_THREADS), 0, st, p); break;
    }
    return check_launch("groupnorm_kernel");
}

namespace mdx {
struct SMParams { const float* X; bf16_t* Y; long rows; int T; long ldx, ldy; float scale; };

// One wave per row (4 rows per 256-thread block): lanes stride the row (coalesced fp32 reads), three passes over a row that
// stays in L1/L2 (a 1400-column row is 5.6 KB): max, sum of exp, normalised write.  exp2 with the scale folded in.
__global__ __launch_bounds__(256) void softmax_rows_kernel(SMParams p) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* x = p.X + r * p.ldx;
    bf16_t* y = p.Y + r * p.ldy;
    const float k = p.scale * 1.4426950408889634f;      // log2(e)
    float m = -INFINITY;
    for (int c = lane; c < p.T; c += 64) m = fmaxf(m, x[c] * k);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < p.T; c += 64) sum += exp2f(x[c] * k - m);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int c = lane; c < p.ldy; c += 64) y[c] = c < p.T ? f2bf(exp2f(x[c] * k - m) * inv) : (bf16_t)0;
}
}  // namespace mdx

extern "C" int mdx_softmax_rows(const MdxSoftmaxDesc* d, void* stream) {
    if (!d || !d->X || !d->Y) return set_error(MDX_EINVAL, "mdx_softmax_rows: null operand");
    if (d->T <= 0 || d->ldx < d->T || d->ldy < d->T) return set_error(MDX_EINVAL, "softmax: need 0 < T <= ldx, ldy");
    if (d->rows <= 0) return MDX_OK;
    mdx::SMParams p{d->X, (bf16_t*)d->Y, d->rows, (int)d->T, d->ldx, d->ldy, (float)d->scale};
    hipLaunchKernelGGL(mdx::softmax_rows_kernel, dim3((unsigned)((d->rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return mdx::check_launch("softmax_rows_kernel");
}

namespace mdx {
template <bool AFFINE>
static int launch_layernorm(const LNParams& p, hipStream_t st) {
    const int rpw = p.C <= 512 ? 4 : (p.C <= 1024 ? 2 : 1);
    dim3 grid((p.M + 4 * rpw - 1) / (4 * rpw));
    if (p.C <= 512) hipLaunchKernelGGL((layernorm_kernel<1, 4, AFFINE>), grid, dim3(256), 0, st, p);
    else if (p.C <= 1024) hipLaunchKernelGGL((layernorm_kernel<2, 2, AFFINE>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((layernorm_kernel<4, 1, AFFINE>), grid, dim3(256), 0, st, p);
    return check_launch("layernorm_kernel");
}
// (x - mean) * rstd without the affine part, for GEMM routes that cannot normalise their A rows in-kernel (MdxGemmDesc.ln_eps:
// gamma / beta are folded into that GEMM's weights and bias).  Same checks as the entry point below.
int launch_layernorm_plain(const bf16_t* X, bf16_t* Y, int M, int C, long ldx, long ldy, float eps, hipStream_t st) {
    if (C % 8 || ldx % 8 || ldy % 8 || C > 8 * 64 * 4 || ((uintptr_t)X & 15) || ((uintptr_t)Y & 15))
        return set_error(MDX_EINVAL, "fused-LayerNorm fallback: C=%d (multiple of 8, <= 2048), 16-byte aligned rows required", C);
    if (M <= 0) return MDX_OK;
    LNParams p;
    p.X = X; p.Y = Y; p.gamma = nullptr; p.beta = nullptr; p.M = M; p.C = C; p.ldx = ldx; p.ldy = ldy; p.eps = eps;
    return launch_layernorm<false>(p, st);
}

// Row statistics of a finished C for GEMM routes that do not emit them from their store phase (MdxGemmDesc.rowstat_out, gemm_conv.hip:
// launch_gemm_conv): one wave per row, 16-byte loads from clamped addresses, part 0 = (sum, sum of squares) of the whole row in a fixed
// order (lane-strided chunks, then the wave tree), parts 1.. = zeros.  A narrow row (C % 8 != 0) takes the element loop.
__global__ __launch_bounds__(256) void rowstat_kernel(const bf16_t* X, int M, int C, long ldx, float* stat, int parts) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = X + (long)row * ldx;
    float s1 = 0.f, s2 = 0.f;
    if ((C & 7) == 0 && (ldx & 7) == 0 && (((uintptr_t)X) & 15) == 0) {
        for (int c8 = lane; c8 < C / 8; c8 += 64) {
            Frag8 f; f.u = *(const uint4*)(xr + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = bf2f(f.h[e]); s1 += x; s2 = __builtin_fmaf(x, x, s2); }
        }
    } else {
        for (int c = lane; c < C; c += 64) { const float x = bf2f(xr[c]); s1 += x; s2 = __builtin_fmaf(x, x, s2); }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
        *(float2*)(stat + (long)row * 2) = make_float2(s1, s2);
        for (int k = 1; k < parts; ++k) *(float2*)(stat + ((long)k * M + row) * 2) = make_float2(0.f, 0.f);
    }
}
int launch_rowstat(const bf16_t* X, int M, int C, long ldx, float* stat, int parts, hipStream_t st) {
    if (M <= 0) return MDX_OK;
    hipLaunchKernelGGL(rowstat_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, X, M, C, ldx, stat, parts);
    return check_launch("rowstat_kernel");
}
}  // namespace mdx

extern "C" int mdx_layernorm_bf16(const MdxLayerNormDesc* d, void* stream) {
    if (!d || !d->X || !d->Y || !d->gamma || !d->beta) return set_error(MDX_EINVAL, "mdx_layernorm_bf16: null operand");
    if (d->C % 8 || d->ldx % 8 || d->ldy % 8) return set_error(MDX_EINVAL, "layernorm: C, ldx, ldy must be multiples of 8");
    if (((uintptr_t)d->X & 15) || ((uintptr_t)d->Y & 15)) return set_error(MDX_EINVAL, "layernorm: 16-byte alignment required");
    if (d->C > 8 * 64 * 4) return set_error(MDX_EUNSUPPORTED, "layernorm: C=%ld > 2048", (long)d->C);
    if (d->M <= 0) return MDX_OK;
    LNParams p;
    p.X = (const bf16_t*)d->X; p.Y = (bf16_t*)d->Y; p.gamma = d->gamma; p.beta = d->beta;
    p.M = (int)d->M; p.C = (int)d->C; p.ldx = d->ldx; p.ldy = d->ldy; p.eps = (float)d->eps;
    return launch_layernorm<true>(p, (hipStream_t)stream);
}
