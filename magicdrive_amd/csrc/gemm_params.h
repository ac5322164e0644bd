// gemm_params.h — parameter block and fused epilogue shared by the two GEMM/conv main loops
// (gemm_conv.hip: register-staged, gemm_dma.hip: LDS-DMA ring).
#pragma once
#include "common.h"

namespace mdx {

struct GCParams {
    const bf16_t* A; const bf16_t* W; void* C; const void* R;
    const float* bias; const float* temb; const int* sel; float* ws;
    int M, N, K;
    long lda, ldw, ldc, ldr;
    long sA, sW, sC, sR;
    long temb_sel_stride, temb_b_stride;
    int rows_per_b;
    int epi, splitk, kchunk, c_f32, batch;
    long ws_bytes;
    // conv geometry (CONV only); lda doubles as the pixel stride of X
    int Hi, Wi, Cin, Ho, Wo, kh, kw, sh, sw, ph, pw;
};

// ---- shared epilogue ---------------------------------------------------------------
// v[4] are raw accumulators for output row m, raw columns nb..nb+3 (nb % 4 == 0).
// For GEGLU, v = value columns and gte = gate columns (raw column nb+32+j).
__device__ __forceinline__ void epilogue_store(const GCParams& p, long zb, int m, int nb,
                                               const float* v, const float* gte) {
    float o[4];
    int ncol;  // output column of o[0]
    const float* tb = nullptr;
    if (p.temb) {
        int sel = p.sel ? *p.sel : 0;
        tb = p.temb + (long)sel * p.temb_sel_stride + (long)(m / p.rows_per_b) * p.temb_b_stride;
    }
    if (p.epi == 1) {  // GEGLU
        ncol = (nb >> 6) * 32 + (nb & 63);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float h = v[j], g = gte[j];
            if (p.bias) { h += p.bias[nb + j]; g += p.bias[nb + 32 + j]; }
            o[j] = h * gelu_erf_f(g);
        }
    } else {
        ncol = nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (p.bias) x += p.bias[nb + j];
            if (tb) x += tb[nb + j];
            if (p.epi == 2) x = silu_f(x);
            o[j] = x;
        }
    }
    if (p.c_f32) {
        float* c = (float*)p.C + zb * p.sC + (long)m * p.ldc + ncol;
        if (p.R) {
            const float* r = (const float*)p.R + zb * p.sR + (long)m * p.ldr + ncol;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += r[j];
        }
        *(float4*)c = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        bf16_t* c = (bf16_t*)p.C + zb * p.sC + (long)m * p.ldc + ncol;
        if (p.R) {
            const bf16_t* r = (const bf16_t*)p.R + zb * p.sR + (long)m * p.ldr + ncol;
            uint2 rv = *(const uint2*)r;
            o[0] += bf2f((bf16_t)(rv.x & 0xffff)); o[1] += bf2f((bf16_t)(rv.x >> 16));
            o[2] += bf2f((bf16_t)(rv.y & 0xffff)); o[3] += bf2f((bf16_t)(rv.y >> 16));
        }
        uint2 ov; ov.x = pack2bf(o[0], o[1]); ov.y = pack2bf(o[2], o[3]);
        *(uint2*)c = ov;
    }
}

}  // namespace mdx
