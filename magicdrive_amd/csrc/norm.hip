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

namespace mdx {

struct GNParams {
    const bf16_t* X; bf16_t* Y; const float* gamma; const float* beta;
    int B, HW, C, G; long ldx, ldy; float eps; int silu;
};

constexpr int GN_THREADS = 1024;

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
    constexpr int GN_U = (VEC == 8) ? 4 : 8;          // <= 32 live values per thread (1024-thread WG: 128 VGPR budget)
    const float pivot = bf2f(xb[0]);
    float s1 = 0.f, s2 = 0.f;
    float v[GN_U][VEC];
    const bool single = nvec <= GN_U * GN_THREADS;
    for (int base = threadIdx.x; base < nvec; base += GN_U * GN_THREADS) {
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int i = base + u * GN_THREADS;
            if (i < nvec) {
                int px = i / vpp;
                load(px, (i - px * vpp) * VEC, v[u]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[u][e] = pivot;
            }
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { float dlt = v[u][e] - pivot; s1 += dlt; s2 += dlt * dlt; }
    }
    block_sum2(s1, s2, red);
    const float n = (float)p.HW * (float)cpg;
    const float md = s1 / n;                          // mean of (x - pivot)
    const float mean = pivot + md;
    const float var = fmaxf(s2 / n - md * md, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    for (int base = threadIdx.x; base < nvec; base += GN_U * GN_THREADS) {
        if (!single) {
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const int i = base + u * GN_THREADS;
                if (i < nvec) {
                    int px = i / vpp;
                    load(px, (i - px * vpp) * VEC, v[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            const int i = base + u * GN_THREADS;
            if (i >= nvec) continue;
            const int px = i / vpp;
            const int cv = (i - px * vpp) * VEC;
            const int c0 = g * cpg + cv;
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float y = (v[u][e] - mean) * rstd * p.gamma[c0 + e] + p.beta[c0 + e];
                if (p.silu) y = silu_f(y);
                o[e] = y;
            }
            bf16_t* dptr = yb + px * ldy + cv;
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
    }
}

struct LNParams {
    const bf16_t* X; bf16_t* Y; const float* gamma; const float* beta;
    int M, C; long ldx, ldy; float eps;
};

// one wave per row; C % 8 == 0, C <= 8 * 64 * MAXV
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(LNParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const bf16_t* x = p.X + (long)row * p.ldx;
    const int nv = p.C / 8;
    float v[MAXV][8];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
            Frag8 f; f.u = *(const uint4*)(x + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(f.h[e]); s1 += v[i][e]; }
        }
    }
    const float mean = wave_sum(s1) / (float)p.C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float dlt = v[i][e] - mean; s2 += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(wave_sum(s2) / (float)p.C + p.eps);
    bf16_t* y = p.Y + (long)row * p.ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
            Frag8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = (v[i][e] - mean) * rstd * p.gamma[c * 8 + e] + p.beta[c * 8 + e];
                f.h[e] = f2bf(o);
            }
            *(uint4*)(y + c * 8) = f.u;
        }
    }
}

}  // namespace mdx

using namespace mdx;

extern "C" int mdx_groupnorm_bf16(const MdxGroupNormDesc* d, void* stream) {
    if (!d || !d->X || !d->Y || !d->gamma || !d->beta) return set_error(MDX_EINVAL, "mdx_groupnorm_bf16: null operand");
    if (d->G <= 0 || d->C % d->G) return set_error(MDX_EINVAL, "groupnorm: C=%ld not divisible by G=%ld", (long)d->C, (long)d->G);
    if (d->B <= 0 || d->HW <= 0) return MDX_OK;
    GNParams p;
    p.X = (const bf16_t*)d->X; p.Y = (bf16_t*)d->Y; p.gamma = d->gamma; p.beta = d->beta;
    p.B = (int)d->B; p.HW = (int)d->HW; p.C = (int)d->C; p.G = (int)d->G; p.ldx = d->ldx; p.ldy = d->ldy;
    p.eps = (float)d->eps; p.silu = (int)d->silu;
    const int cpg = p.C / p.G;
    // vector width limited by cpg and by the alignment of every group start / row stride
    int vec = 1;
    for (int v = 8; v > 1; v >>= 1) {
        if (cpg % v == 0 && p.ldx % v == 0 && p.ldy % v == 0 && ((uintptr_t)p.X % (2 * v)) == 0 && ((uintptr_t)p.Y % (2 * v)) == 0) { vec = v; break; }
    }
    dim3 grid(p.G, p.B);
    hipStream_t st = (hipStream_t)stream;
    switch (vec) {
        case 8: hipLaunchKernelGGL(groupnorm_kernel<8>, grid, dim3(GN_THREADS), 0, st, p); break;
        case 4: hipLaunchKernelGGL(groupnorm_kernel<4>, grid, dim3(GN_THREADS), 0, st, p); break;
        case 2: hipLaunchKernelGGL(groupnorm_kernel<2>, grid, dim3(GN_THREADS), 0, st, p); break;
        default: hipLaunchKernelGGL(groupnorm_kernel<1>, grid, dim3(GN_THREADS), 0, st, p); break;
    }
    return check_launch("groupnorm_kernel");
}

extern "C" int mdx_layernorm_bf16(const MdxLayerNormDesc* d, void* stream) {
    if (!d || !d->X || !d->Y || !d->gamma || !d->beta) return set_error(MDX_EINVAL, "mdx_layernorm_bf16: null operand");
    if (d->C % 8 || d->ldx % 8 || d->ldy % 8) return set_error(MDX_EINVAL, "layernorm: C, ldx, ldy must be multiples of 8");
    if (((uintptr_t)d->X & 15) || ((uintptr_t)d->Y & 15)) return set_error(MDX_EINVAL, "layernorm: 16-byte alignment required");
    if (d->C > 8 * 64 * 4) return set_error(MDX_EUNSUPPORTED, "layernorm: C=%ld > 2048", (long)d->C);
    if (d->M <= 0) return MDX_OK;
    LNParams p;
    p.X = (const bf16_t*)d->X; p.Y = (bf16_t*)d->Y; p.gamma = d->gamma; p.beta = d->beta;
    p.M = (int)d->M; p.C = (int)d->C; p.ldx = d->ldx; p.ldy = d->ldy; p.eps = (float)d->eps;
    dim3 grid((p.M + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (p.C <= 512) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, st, p);
    else if (p.C <= 1024) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, st, p);
    return check_launch("layernorm_kernel");
}
