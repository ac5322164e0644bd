// norm.hip — GroupNorm(+SiLU) and LayerNorm on channels-last bf16 tensors, fp32 statistics.
//
// GroupNorm: ATen native_group_norm + silu as used by ResnetBlock2D (diffusers/models/resnet.py:
//   596-598, 626-630), Transformer2DModel.norm (transformer_2d.py:278, eps 1e-6) and conv_norm_out
//   (unet_2d_condition_multiview.py:519-521).  One workgroup per (batch, group); the group's
//   HW x (C/G) slab (<= 8400 x 80 elements, L2 resident) is read three times: mean, centred
//   variance (two-pass: no E[x^2]-E[x]^2 cancellation), normalise+affine(+SiLU)+store.
// LayerNorm: nn.LayerNorm over C (attention.py:85,104,120; blocks.py:67-71), one wave per token
//   row, 16-byte loads, shuffle reductions.
#include "common.h"
#include "launch.h"

namespace mdx {

struct GNParams {
    const bf16_t* X; bf16_t* Y; const float* gamma; const float* beta;
    int B, HW, C, G; long ldx, ldy; float eps; int silu;
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0] + red[1] + red[2] + red[3];
    return t;
}

// VEC = channels handled per thread-iteration (cpg % VEC == 0), VEC in {1,2,4,8}
template <int VEC>
__global__ __launch_bounds__(256) void groupnorm_kernel(GNParams p) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.G;
    const int vpp = cpg / VEC;                 // vectors per pixel in this group
    const long nvec = (long)p.HW * vpp;
    const bf16_t* xb = p.X + (long)b * p.HW * p.ldx + (long)g * cpg;
    bf16_t* yb = p.Y + (long)b * p.HW * p.ldy + (long)g * cpg;

    auto load = [&](long i, float* v) {
        long px = i / vpp;
        int cv = (int)(i - px * vpp) * VEC;
        const bf16_t* s = xb + px * p.ldx + cv;
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

    float s1 = 0.f;
    for (long i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC]; load(i, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s1 += v[e];
    }
    const float n = (float)((long)p.HW * cpg);
    const float mean = block_sum_256(s1, red) / n;
    float s2 = 0.f;
    for (long i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC]; load(i, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { float dlt = v[e] - mean; s2 += dlt * dlt; }
    }
    const float var = block_sum_256(s2, red) / n;
    const float rstd = rsqrtf(var + p.eps);
    for (long i = threadIdx.x; i < nvec; i += 256) {
        float v[VEC]; load(i, v);
        long px = i / vpp;
        int cv = (int)(i - px * vpp) * VEC;
        const int c0 = g * cpg + cv;
        bf16_t o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float y = (v[e] - mean) * rstd * p.gamma[c0 + e] + p.beta[c0 + e];
            if (p.silu) y = silu_f(y);
            o[e] = f2bf(y);
        }
        bf16_t* dptr = yb + px * p.ldy + cv;
        if constexpr (VEC == 8) {
            Frag8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f.h[e] = o[e];
            *(uint4*)dptr = f.u;
        } else if constexpr (VEC == 4) {
            uint2 u; u.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); u.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            *(uint2*)dptr = u;
        } else if constexpr (VEC == 2) {
            *(uint32_t*)dptr = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        } else {
            *dptr = o[0];
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
        case 8: hipLaunchKernelGGL(groupnorm_kernel<8>, grid, dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL(groupnorm_kernel<4>, grid, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(groupnorm_kernel<2>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(groupnorm_kernel<1>, grid, dim3(256), 0, st, p); break;
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
