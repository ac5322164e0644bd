// elementwise.hip — the small kernels around the MFMA ops: direct (vector-ALU) convolution /
// linear for tiny channel counts, residual adds / copies / nearest upsample / layout changes,
// Fourier (NeRF) embedding, class-token gather, sinusoidal timestep features and the fused
// classifier-free-guidance + DDIM update.  All are HBM/L2-bound byte movers: coalesced,
// 16-byte vectorised where the layout allows, fp32 math.  Reference call sites: include/mdx.h.
#include "common.h"
#include "launch.h"
#include "options.h"

namespace mdx {

// ------------------------------------------------------------------------------------------
// direct convolution / linear
// ------------------------------------------------------------------------------------------
struct CDParams {
    const void* X; const bf16_t* W; void* Y; const void* R;
    const float* bias; const float* temb; const int* sel;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw;
    long ldx, ldy, ldr, temb_sel_stride, temb_b_stride;
    int epi, x_f32, y_f32;
};

__device__ __forceinline__ float cd_finish(const CDParams& p, float acc, long m, int n, int b) {
    if (p.bias) acc += p.bias[n];
    if (p.temb) {
        int sel = p.sel ? *p.sel : 0;
        acc += p.temb[(long)sel * p.temb_sel_stride + (long)b * p.temb_b_stride + n];
    }
    if (p.epi == 2) acc = silu_f(acc);
    if (p.R) acc += p.y_f32 ? ((const float*)p.R)[m * p.ldr + n] : bf2f(((const bf16_t*)p.R)[m * p.ldr + n]);
    return acc;
}

// one thread per output element (n fastest): tiny K (conv_in K=36, cam2token K=189, bbox_proj K=216,
// time-embedding MLP rows).
template <bool XF32>
__global__ __launch_bounds__(256) void conv_direct_simple_kernel(CDParams p) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)p.B * p.Ho * p.Wo * p.Cout;
    if (idx >= total) return;
    int n = (int)(idx % p.Cout);
    long m = idx / p.Cout;
    int hw = p.Ho * p.Wo;
    int b = (int)(m / hw);
    int rem = (int)(m - (long)b * hw);
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const bf16_t* w = p.W + (long)n * p.kh * p.kw * p.Cin;
    float acc = 0.f;
    for (int ky = 0; ky < p.kh; ++ky) {
        int iy = oy * p.sh - p.ph + ky;
        if ((unsigned)iy >= (unsigned)p.Hi) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
            int ix = ox * p.sw - p.pw + kx;
            if ((unsigned)ix >= (unsigned)p.Wi) continue;
            long xo = (((long)b * p.Hi + iy) * p.Wi + ix) * p.ldx;
            const bf16_t* wk = w + (ky * p.kw + kx) * p.Cin;
            if (XF32) {
                const float* x = (const float*)p.X + xo;
                for (int c = 0; c < p.Cin; ++c) acc += x[c] * bf2f(wk[c]);
            } else {
                const bf16_t* x = (const bf16_t*)p.X + xo;
                for (int c = 0; c < p.Cin; ++c) acc += bf2f(x[c]) * bf2f(wk[c]);
            }
        }
    }
    acc = cd_finish(p, acc, m, n, b);
    if (p.y_f32) ((float*)p.Y)[m * p.ldy + n] = acc; else ((bf16_t*)p.Y)[m * p.ldy + n] = f2bf(acc);
}

// K-parallel variant for few output channels and long K (conv_out: Cout=4, K=2880):
// one wave per output pixel, lanes stride over 16-byte chunks of (tap, ci), shuffle reduction.
template <int NOUT>
__global__ __launch_bounds__(256) void conv_direct_kpar_kernel(CDParams p) {
    const int lane = threadIdx.x & 63;
    long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long M = (long)p.B * p.Ho * p.Wo;
    if (m >= M) return;
    int hw = p.Ho * p.Wo;
    int b = (int)(m / hw);
    int rem = (int)(m - (long)b * hw);
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int cpc = p.Cin / 8;                 // chunks per tap
    const int nch = p.kh * p.kw * cpc;
    const long K = (long)p.kh * p.kw * p.Cin;
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int c = lane; c < nch; c += 64) {
        int tap = c / cpc;
        int ci = (c - tap * cpc) * 8;
        int ky = tap / p.kw, kx = tap - ky * p.kw;
        int iy = oy * p.sh - p.ph + ky, ix = ox * p.sw - p.pw + kx;
        if ((unsigned)iy >= (unsigned)p.Hi || (unsigned)ix >= (unsigned)p.Wi) continue;
        Frag8 xv;
        xv.u = *(const uint4*)((const bf16_t*)p.X + (((long)b * p.Hi + iy) * p.Wi + ix) * p.ldx + ci);
        float xf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[e] = bf2f(xv.h[e]);
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            if (n < p.Cout) {
                Frag8 wv;
                wv.u = *(const uint4*)(p.W + (long)n * K + (long)tap * p.Cin + ci);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[n] += xf[e] * bf2f(wv.h[e]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = wave_sum(acc[n]);
    if (lane < p.Cout && lane < NOUT) {
        float a = 0.f;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) if (n == lane) a = acc[n];
        a = cd_finish(p, a, m, lane, b);
        if (p.y_f32) ((float*)p.Y)[m * p.ldy + lane] = a; else ((bf16_t*)p.Y)[m * p.ldy + lane] = f2bf(a);
    }
}

// Weight-stationary form of the K-parallel kernel (round 4; conv_out of the UNet: Cout = 4, K = 9 x 320 = 2880, one launch per step over
// every latent pixel).  The kernel above re-fetches its 4 x 16 bytes of weights per chunk and pixel — 23 KB of L1 traffic per pixel for
// 5.8 KB of activations: 1.97 ms per step at 768 views, 13 TFLOP/s.  Here a wave keeps ITS share of the weights in registers
// (lane l owns the 16-byte chunks l, l + 64, ... of the (tap, channel) axis for all Cout rows: NCHK x 4 x 4 packed words) and walks
// PIX consecutive output pixels: per pixel NCHK activation loads (zero for taps outside the image), NCHK x 16 packed dot products
// (v_dot2c_f32_bf16 / _f16: two 16-bit products + fp32 accumulate per instruction, no unpacking) and one wave reduction per output channel.
// Summation order per output element: lane-local over the lane's chunks, then across lanes — a permutation of the kernel above's;
// fp32 throughout (the 16-bit products are exact in fp32).
template <int NCHK, int PIX>
__global__ __launch_bounds__(256) void conv_direct_kpar_ws_kernel(CDParams p) {
    constexpr int NOUT = 4;
    const int lane = threadIdx.x & 63;
    const long M = (long)p.B * p.Ho * p.Wo;
    const long m_first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PIX;
    if (m_first >= M) return;
    const int cpc = p.Cin / 8;                 // chunks per tap
    const int nch = p.kh * p.kw * cpc;
    const long K = (long)p.kh * p.kw * p.Cin;
    // this lane's chunks: (tap, channel offset) and the weights of the Cout rows
    uint4 w[NCHK][NOUT];
    int c_dy[NCHK], c_dx[NCHK], c_ci[NCHK];
    bool c_on[NCHK];
#pragma unroll
    for (int i = 0; i < NCHK; ++i) {
        const int c = lane + 64 * i;
        c_on[i] = c < nch;
        const int cc = c_on[i] ? c : 0;
        const int tap = cc / cpc;
        c_ci[i] = (cc - tap * cpc) * 8;
        const int ky = tap / p.kw;
        c_dy[i] = ky - p.ph; c_dx[i] = tap - ky * p.kw - p.pw;
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c_on[i] && n < p.Cout) v = *(const uint4*)(p.W + (long)n * K + (long)tap * p.Cin + c_ci[i]);
            w[i][n] = v;
        }
    }
    const int hw = p.Ho * p.Wo;
    for (int q = 0; q < PIX; ++q) {
        const long m = m_first + q;
        if (m >= M) break;                     // wave-uniform
        const int b = (int)(m / hw);
        const int rem = (int)(m - (long)b * hw);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        uint4 x[NCHK];
#pragma unroll
        for (int i = 0; i < NCHK; ++i) {       // unconditional loads from clamped addresses, masked afterwards (no per-load branch + wait)
            const int iy = oy * p.sh + c_dy[i], ix = ox * p.sw + c_dx[i];
            const bool ok = c_on[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const int iyc = min(max(iy, 0), p.Hi - 1), ixc = min(max(ix, 0), p.Wi - 1);
            const uint4 v = *(const uint4*)((const bf16_t*)p.X + (((long)b * p.Hi + iyc) * p.Wi + ixc) * p.ldx + c_ci[i]);
            x[i].x = ok ? v.x : 0u; x[i].y = ok ? v.y : 0u; x[i].z = ok ? v.z : 0u; x[i].w = ok ? v.w : 0u;
        }
        float acc[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < NCHK; ++i) {
                a = dot2_16(x[i].x, w[i][n].x, a); a = dot2_16(x[i].y, w[i][n].y, a);
                a = dot2_16(x[i].z, w[i][n].z, a); a = dot2_16(x[i].w, w[i][n].w, a);
            }
            acc[n] = a;
        }
        // Reduce the four per-lane sums over the wave TOGETHER (10 VALU operations instead of 4 x 6 ds_bpermute round trips): two
        // v_permlane32_swap fold the wave's halves (lanes 0-31 keep outputs 0 / 1, lanes 32-63 outputs 2 / 3), one v_permlane16_swap
        // folds the 16-lane rows (row r keeps output r), four DPP row rotations finish inside each row: every lane of row r ends up
        // with the total of output channel r.
        {
            unsigned a0 = __float_as_uint(acc[0]), a1 = __float_as_uint(acc[1]), a2 = __float_as_uint(acc[2]), a3 = __float_as_uint(acc[3]);
            { auto r_ = __builtin_amdgcn_permlane32_swap(a0, a2, false, false); a0 = r_[0]; a2 = r_[1]; }
            { auto r_ = __builtin_amdgcn_permlane32_swap(a1, a3, false, false); a1 = r_[0]; a3 = r_[1]; }
            unsigned b0 = __float_as_uint(__uint_as_float(a0) + __uint_as_float(a2));     // lanes 0-31: output 0, lanes 32-63: output 2
            unsigned b1 = __float_as_uint(__uint_as_float(a1) + __uint_as_float(a3));     // lanes 0-31: output 1, lanes 32-63: output 3
            { auto r_ = __builtin_amdgcn_permlane16_swap(b0, b1, false, false); b0 = r_[0]; b1 = r_[1]; }
            float c = __uint_as_float(b0) + __uint_as_float(b1);                           // row r (lanes 16 r .. 16 r + 15): output r
            c += __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(c), 0x128, 0xf, 0xf, false));   // row_ror:8
            c += __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(c), 0x124, 0xf, 0xf, false));   // row_ror:4
            c += __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(c), 0x122, 0xf, 0xf, false));   // row_ror:2
            c += __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(c), 0x121, 0xf, 0xf, false));   // row_ror:1
            const int n = lane >> 4;
            if ((lane & 15) == 0 && n < p.Cout) {
                const float a = cd_finish(p, c, m, n, b);
                if (p.y_f32) ((float*)p.Y)[m * p.ldy + n] = a; else ((bf16_t*)p.Y)[m * p.ldy + n] = f2bf(a);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// element-wise family
// ------------------------------------------------------------------------------------------
struct EWParams {
    const void* X; void* Y; const int* ymap; const int* xmap;
    int kind; long M; int C; long ldx, ldy;
    int B, Hi, Wi, Ho, Wo, x_f32, y_f32; float alpha;
};

__device__ __forceinline__ float ew_load(const void* p, long i, int f32) {
    return f32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void ew_store(void* p, long i, int f32, float v) {
    if (f32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}

// generic scalar path: one thread per element
__global__ __launch_bounds__(256) void ew_scalar_kernel(EWParams p) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (p.kind == MDX_EW_UPSAMPLE) {
        long total = (long)p.B * p.Ho * p.Wo * p.C;
        if (idx >= total) return;
        int c = (int)(idx % p.C);
        long pix = idx / p.C;
        int ox = (int)(pix % p.Wo);
        long t = pix / p.Wo;
        int oy = (int)(t % p.Ho);
        int b = (int)(t / p.Ho);
        int iy = p.ymap[oy], ix = p.xmap[ox];
        float v = ew_load(p.X, (((long)b * p.Hi + iy) * p.Wi + ix) * p.ldx + c, p.x_f32);
        ew_store(p.Y, pix * p.ldy + c, p.y_f32, v);
        return;
    }
    if (p.kind == MDX_EW_NCHW_TO_NHWC || p.kind == MDX_EW_NHWC_TO_NCHW) {
        // X/Y indexed by (b, c, y, x); B,C,Hi,Wi describe the tensor
        long total = (long)p.B * p.C * p.Hi * p.Wi;
        if (idx >= total) return;
        int x = (int)(idx % p.Wi);
        long t = idx / p.Wi;
        int y = (int)(t % p.Hi);
        t /= p.Hi;
        int c = (int)(t % p.C);
        int b = (int)(t / p.C);
        long nchw = idx;
        long nhwc_pix = ((long)b * p.Hi + y) * p.Wi + x;
        if (p.kind == MDX_EW_NCHW_TO_NHWC)
            ew_store(p.Y, nhwc_pix * p.ldy + c, p.y_f32, ew_load(p.X, nchw, p.x_f32));
        else
            ew_store(p.Y, nchw, p.y_f32, ew_load(p.X, nhwc_pix * p.ldx + c, p.x_f32));
        return;
    }
    long total = p.M * p.C;
    if (idx >= total) return;
    long m = idx / p.C;
    int c = (int)(idx - m * p.C);
    float x = ew_load(p.X, m * p.ldx + c, p.x_f32);
    float r;
    switch (p.kind) {
        case MDX_EW_ADD: r = ew_load(p.Y, m * p.ldy + c, p.y_f32) + x; break;
        case MDX_EW_COPY: r = x; break;
        case MDX_EW_SILU: r = silu_f(x); break;
        case MDX_EW_SCALE: r = x * p.alpha; break;
        default: r = x; break;
    }
    ew_store(p.Y, m * p.ldy + c, p.y_f32, r);
}

// bf16 -> bf16 fast path, 8 channels per thread (ADD / COPY / SILU / SCALE)
__global__ __launch_bounds__(256) void ew_vec8_kernel(EWParams p) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = p.C / 8;
    if (idx >= p.M * c8) return;
    long m = idx / c8;
    int c = (int)(idx - m * c8) * 8;
    Frag8 x, y;
    x.u = *(const uint4*)((const bf16_t*)p.X + m * p.ldx + c);
    bf16_t* yp = (bf16_t*)p.Y + m * p.ldy + c;
    if (p.kind == MDX_EW_ADD) y.u = *(const uint4*)yp;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float xv = bf2f(x.h[e]);
        float r;
        switch (p.kind) {
            case MDX_EW_ADD: r = bf2f(y.h[e]) + xv; break;
            case MDX_EW_SILU: r = silu_f(xv); break;
            case MDX_EW_SCALE: r = xv * p.alpha; break;
            default: r = xv; break;
        }
        y.h[e] = f2bf(r);
    }
    *(uint4*)yp = y.u;
}

// bf16 nearest resize, 8 channels (16 bytes) per thread: each output pixel row segment is one coalesced copy of its source pixel's
// (the scalar path moved 2 bytes per lane: 1.04 ms for the 14x25 -> 28x50 x 640-channel upsample of 384 views, 7x its HBM time)
__global__ __launch_bounds__(256) void ew_upsample_vec8_kernel(EWParams p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = p.C / 8;
    const long total = (long)p.B * p.Ho * p.Wo * c8;
    if (idx >= total) return;
    const long pix = idx / c8;
    const int c = (int)(idx - pix * c8) * 8;
    const int ox = (int)(pix % p.Wo);
    const long t = pix / p.Wo;
    const int oy = (int)(t % p.Ho);
    const int b = (int)(t / p.Ho);
    const int iy = p.ymap[oy], ix = p.xmap[ox];
    const uint4 v = *(const uint4*)((const bf16_t*)p.X + (((long)b * p.Hi + iy) * p.Wi + ix) * p.ldx + c);
    *(uint4*)((bf16_t*)p.Y + pix * p.ldy + c) = v;
}

// ------------------------------------------------------------------------------------------
// Fourier embedding / gather / timestep features / CFG + DDIM
// ------------------------------------------------------------------------------------------
struct FourierParams { const float* X; bf16_t* Y; const uint8_t* mask; const float* null_feat; long n; int P, F; long ldy; };

__global__ __launch_bounds__(256) void fourier_kernel(FourierParams p) {
    const int per_pt = 3 + 6 * p.F;
    const int width = p.P * per_pt;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.n * width) return;
    long i = idx / width;
    int j = (int)(idx - i * width);
    float out;
    if (p.mask && !p.mask[i]) {
        out = p.null_feat ? p.null_feat[j] : 0.f;
    } else {
        int pt = j / per_pt;
        int k = j - pt * per_pt;     // 0..per_pt-1 : [x(3) | sin f0 (3) | cos f0 (3) | sin f1 ...]
        int comp = k % 3;
        int fn = k / 3;              // 0 = identity, 1 = sin f0, 2 = cos f0, 3 = sin f1, ...
        float x = p.X[(i * p.P + pt) * 3 + comp];
        if (fn == 0) out = x;
        else {
            int fi = (fn - 1) >> 1;
            float a = x * (float)(1 << fi);      // freq bands 2^0 .. 2^(F-1) (embedder.py:26-29)
            out = ((fn - 1) & 1) ? cosf(a) : sinf(a);
        }
    }
    p.Y[i * p.ldy + j] = f2bf(out);
}

struct GatherParams { const bf16_t* T; bf16_t* Y; const int64_t* idx; const uint8_t* mask; const bf16_t* null_row; long n; int C; long ldt, ldy; int n_rows; };

__global__ __launch_bounds__(256) void gather_kernel(GatherParams p) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= p.n * p.C) return;
    long i = id / p.C;
    int c = (int)(id - i * p.C);
    bool m = p.mask ? p.mask[i] != 0 : true;
    bf16_t v;
    if (m) {
        long r = p.idx[i];
        if (r < 0) r += p.n_rows;                 // python-style negative index (bbox_embedder.py:179)
        if (r < 0 || r >= p.n_rows) r = 0;
        v = p.T[r * p.ldt + c];
    } else {
        v = p.null_row ? p.null_row[c] : (bf16_t)0;
    }
    p.Y[i * p.ldy + c] = v;
}

struct TimeEmbParams { const float* t; void* Y; long n; int dim, flip; long ldy; float shift, max_period; int y_f32; };

__global__ __launch_bounds__(256) void timeemb_kernel(TimeEmbParams p) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= p.n * p.dim) return;
    long i = id / p.dim;
    int j = (int)(id - i * p.dim);
    const int half = p.dim / 2;
    float out = 0.f;
    if (j < 2 * half) {
        int first = j < half;          // first half of the output
        int k = first ? j : j - half;
        // embeddings.py:47-61: exponent = -ln(max_period) * k / (half - shift); emb = t * exp(exponent)
        float e = -logf(p.max_period) * (float)k / ((float)half - p.shift);
        float a = p.t[i] * expf(e);
        // unflipped layout is [sin | cos]; flip_sin_to_cos -> [cos | sin]
        bool use_cos = p.flip ? first : !first;
        out = use_cos ? cosf(a) : sinf(a);
    }
    if (p.y_f32) ((float*)p.Y)[i * p.ldy + j] = out; else ((bf16_t*)p.Y)[i * p.ldy + j] = f2bf(out);
}

struct DdimParams { float* x; const float* eps; const float* coef; int* step; void* x_in; long n; int cfg; float g; int xin_c, xin_ld;
                    const float* gv_cond; const float* gv_noise; const unsigned char* gv_mask; int gv_mode; long gv_view; int gv_last; };

__global__ __launch_bounds__(256) void ddim_kernel(DdimParams p) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int step = *p.step;
    if (i < p.n && step < 0) {          // a row index of -1 = "timestep not in the scheduler's list" (DDIMScheduler.step with a device timestep):
        p.x[i] = __builtin_nanf("");     // poison the result instead of silently using row 0
        return;
    }
    if (i < p.n) {
        const float* c = p.coef + 4 * step;
        float e;
        if (p.cfg) {
            float eu = p.eps[i], ec = p.eps[p.n + i];
            e = eu + p.g * (ec - eu);
        } else {
            e = p.eps[i];
        }
        // given views (pipeline_bev_controlnet_given_view.py:263-291, 380-390; include/mdx.h: MdxDdimDesc.gv_*)
        const bool given = p.gv_mode != 0 && p.gv_mask[i / p.gv_view] != 0;
        if (given && p.gv_mode == 2) e = p.gv_noise[i];
        float x = p.x[i];
        float x0 = (x - c[1] * e) / c[0];
        float xn = c[2] * x0 + c[3] * e;
        if (given && p.gv_mode == 1 && step < p.gv_last) xn = c[2] * p.gv_cond[i] + c[3] * p.gv_noise[i];
        p.x[i] = xn;
        if (p.x_in) {
            if (p.xin_ld > 0) {          // bf16 channels-last copy with padded pixel stride
                long px = i / p.xin_c;
                int c = (int)(i - px * p.xin_c);
                bf16_t* xi = (bf16_t*)p.x_in;
                bf16_t v = f2bf(xn);
                xi[px * p.xin_ld + c] = v;
                if (p.cfg) xi[(p.n / p.xin_c + px) * p.xin_ld + c] = v;
            } else {
                float* xi = (float*)p.x_in;
                xi[i] = xn;
                if (p.cfg) xi[p.n + i] = xn;
            }
        }
    }
}
struct UniPCParams { float* x; const float* eps; const float* coef; int* step; void* x_in; float* x_last; float* m1; float* m2; long n; int cfg; float g; int xin_c, xin_ld;
                     const float* gv_cond; const float* gv_noise; const unsigned char* gv_mask; int gv_mode; long gv_view; int gv_last; };

__global__ __launch_bounds__(256) void unipc_kernel(UniPCParams p) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int step = *p.step;
    if (i >= p.n) return;
    const float* c = p.coef + 12 * step;
    float e;
    if (p.cfg) {
        float eu = p.eps[i], ec = p.eps[p.n + i];
        e = eu + p.g * (ec - eu);
    } else {
        e = p.eps[i];
    }
    // given views (pipeline_bev_controlnet_given_view.py:263-291, 380-390 under the scheduler build_pipe installs; MdxUniPCDesc.gv_*)
    const bool given = p.gv_mode != 0 && p.gv_mask[i / p.gv_view] != 0;
    if (given && p.gv_mode == 2) e = p.gv_noise[i];
    const float x = p.x[i];
    const float m1 = p.m1[i], m2 = p.m2[i];
    const float mt = c[0] * x + c[1] * e;
    float xc = x;
    if (c[2] != 0.f) xc = c[3] * p.x_last[i] + c[4] * m1 + c[5] * m2 + c[6] * mt;
    float xn = c[7] * xc + c[8] * mt + c[9] * m1;
    if (given && p.gv_mode == 1 && step < p.gv_last) xn = c[10] * p.gv_cond[i] + c[11] * p.gv_noise[i];
    p.x[i] = xn; p.x_last[i] = xc; p.m2[i] = m1; p.m1[i] = mt;
    if (p.x_in) {
        if (p.xin_ld > 0) {
            long px = i / p.xin_c;
            int ch = (int)(i - px * p.xin_c);
            bf16_t* xi = (bf16_t*)p.x_in;
            bf16_t v = f2bf(xn);
            xi[px * p.xin_ld + ch] = v;
            if (p.cfg) xi[(p.n / p.xin_c + px) * p.xin_ld + ch] = v;
        } else {
            float* xi = (float*)p.x_in;
            xi[i] = xn;
            if (p.cfg) xi[p.n + i] = xn;
        }
    }
}

// Separate 1-thread launch so every block of ddim_kernel has read *step before it changes.
__global__ void step_inc_kernel(int* step) { *step += 1; }

}  // namespace mdx

using namespace mdx;

extern "C" int mdx_conv2d_direct(const MdxConvDirectDesc* d, void* stream) {
    if (!d || !d->X || !d->Wt || !d->Y) return set_error(MDX_EINVAL, "mdx_conv2d_direct: null operand");
    if (d->epilogue == MDX_EPI_GEGLU) return set_error(MDX_EINVAL, "direct conv has no GEGLU epilogue");
    CDParams p;
    p.X = d->X; p.W = (const bf16_t*)d->Wt; p.Y = d->Y; p.R = d->R; p.bias = d->bias; p.temb = d->temb; p.sel = d->sel_ptr;
    p.B = (int)d->B; p.Hi = (int)d->Hi; p.Wi = (int)d->Wi; p.Cin = (int)d->Cin; p.Ho = (int)d->Ho; p.Wo = (int)d->Wo; p.Cout = (int)d->Cout;
    p.kh = (int)d->kh; p.kw = (int)d->kw; p.sh = (int)d->sh; p.sw = (int)d->sw; p.ph = (int)d->ph; p.pw = (int)d->pw;
    p.ldx = d->ldx; p.ldy = d->ldy; p.ldr = d->ldr; p.temb_sel_stride = d->temb_sel_stride; p.temb_b_stride = d->temb_b_stride;
    p.epi = (int)d->epilogue; p.x_f32 = (int)d->x_is_f32; p.y_f32 = (int)d->y_is_f32;
    long M = (long)p.B * p.Ho * p.Wo;
    if (M <= 0 || p.Cout <= 0) return MDX_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool kpar = !p.x_f32 && p.Cout <= 8 && (p.Cin % 8) == 0 && (p.ldx % 8) == 0 && (long)p.kh * p.kw * p.Cin >= 512 &&
                      ((uintptr_t)p.X & 15) == 0 && ((uintptr_t)p.W & 15) == 0;
    // weight-stationary form: Cout <= 4, at most 6 x 64 chunks of (tap, channel) — conv_out at every width of the SD family up to Cin = 336
    // ... Cin = 320: 360 chunks.  Enough pixels that a wave's weight prologue (6 x 4 x 16 bytes per lane) is amortised over PIX of them.
    if (kpar && p.Cout <= 4 && (long)p.kh * p.kw * (p.Cin / 8) <= 384 && M >= 65536 && opt(OPT_CONV_OUT_WS) != 0) {
        constexpr int PIX = 16;
        dim3 grid((unsigned)((M + 4 * PIX - 1) / (4 * PIX)));
        hipLaunchKernelGGL((conv_direct_kpar_ws_kernel<6, PIX>), grid, dim3(256), 0, st, p);
        return check_launch("conv_direct_kpar_ws_kernel");
    }
    if (kpar) {
        dim3 grid((unsigned)((M + 3) / 4));
        if (p.Cout <= 4) hipLaunchKernelGGL(conv_direct_kpar_kernel<4>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(conv_direct_kpar_kernel<8>, grid, dim3(256), 0, st, p);
        return check_launch("conv_direct_kpar_kernel");
    }
    long total = M * p.Cout;
    dim3 grid((unsigned)((total + 255) / 256));
    if (p.x_f32) hipLaunchKernelGGL(conv_direct_simple_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_direct_simple_kernel<false>, grid, dim3(256), 0, st, p);
    return check_launch("conv_direct_simple_kernel");
}

extern "C" int mdx_elementwise(const MdxEwDesc* d, void* stream) {
    if (!d || !d->X || !d->Y) return set_error(MDX_EINVAL, "mdx_elementwise: null operand");
    EWParams p;
    p.X = d->X; p.Y = d->Y; p.ymap = d->ymap; p.xmap = d->xmap; p.kind = (int)d->kind; p.M = d->M; p.C = (int)d->C;
    p.ldx = d->ldx; p.ldy = d->ldy; p.B = (int)d->B; p.Hi = (int)d->Hi; p.Wi = (int)d->Wi; p.Ho = (int)d->Ho; p.Wo = (int)d->Wo;
    p.x_f32 = (int)d->x_is_f32; p.y_f32 = (int)d->y_is_f32; p.alpha = (float)d->alpha;
    hipStream_t st = (hipStream_t)stream;
    long total;
    switch (p.kind) {
        case MDX_EW_UPSAMPLE:
            if (!p.ymap || !p.xmap) return set_error(MDX_EINVAL, "upsample needs ymap/xmap");
            total = (long)p.B * p.Ho * p.Wo * p.C; break;
        case MDX_EW_NCHW_TO_NHWC: case MDX_EW_NHWC_TO_NCHW:
            total = (long)p.B * p.C * p.Hi * p.Wi; break;
        case MDX_EW_ADD: case MDX_EW_COPY: case MDX_EW_SILU: case MDX_EW_SCALE:
            total = p.M * p.C; break;
        default: return set_error(MDX_EINVAL, "mdx_elementwise: unknown kind %d", p.kind);
    }
    if (total <= 0) return MDX_OK;
    const bool flat = p.kind == MDX_EW_ADD || p.kind == MDX_EW_COPY || p.kind == MDX_EW_SILU || p.kind == MDX_EW_SCALE;
    if (flat && !p.x_f32 && !p.y_f32 && p.C % 8 == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0 &&
        ((uintptr_t)p.X & 15) == 0 && ((uintptr_t)p.Y & 15) == 0) {
        long nv = p.M * (p.C / 8);
        hipLaunchKernelGGL(ew_vec8_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, st, p);
        return check_launch("ew_vec8_kernel");
    }
    if (p.kind == MDX_EW_UPSAMPLE && !p.x_f32 && !p.y_f32 && p.C % 8 == 0 && p.ldx % 8 == 0 && p.ldy % 8 == 0 &&
        ((uintptr_t)p.X & 15) == 0 && ((uintptr_t)p.Y & 15) == 0) {
        hipLaunchKernelGGL(ew_upsample_vec8_kernel, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, st, p);
        return check_launch("ew_upsample_vec8_kernel");
    }
    hipLaunchKernelGGL(ew_scalar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return check_launch("ew_scalar_kernel");
}

extern "C" int mdx_fourier_embed(const MdxFourierDesc* d, void* stream) {
    if (!d || !d->X || !d->Y) return set_error(MDX_EINVAL, "mdx_fourier_embed: null operand");
    if (d->F < 0 || d->F > 16) return set_error(MDX_EINVAL, "fourier: F out of range");
    FourierParams p{d->X, (bf16_t*)d->Y, d->mask, d->null_feat, d->n, (int)d->P, (int)d->F, d->ldy};
    long total = p.n * p.P * (3 + 6 * p.F);
    if (total <= 0) return MDX_OK;
    hipLaunchKernelGGL(fourier_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("fourier_kernel");
}

extern "C" int mdx_gather_rows(const MdxGatherDesc* d, void* stream) {
    if (!d || !d->T || !d->Y || !d->idx) return set_error(MDX_EINVAL, "mdx_gather_rows: null operand");
    GatherParams p{(const bf16_t*)d->T, (bf16_t*)d->Y, d->idx, d->mask, (const bf16_t*)d->null_row, d->n, (int)d->C, d->ldt, d->ldy, (int)d->n_rows};
    long total = p.n * p.C;
    if (total <= 0) return MDX_OK;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("gather_kernel");
}

extern "C" int mdx_timestep_embedding(const MdxTimeEmbDesc* d, void* stream) {
    if (!d || !d->t || !d->Y) return set_error(MDX_EINVAL, "mdx_timestep_embedding: null operand");
    TimeEmbParams p{d->t, d->Y, d->n, (int)d->dim, (int)d->flip_sin_to_cos, d->ldy, (float)d->freq_shift, (float)d->max_period, 1};
    long total = p.n * p.dim;
    if (total <= 0) return MDX_OK;
    hipLaunchKernelGGL(timeemb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("timeemb_kernel");
}

extern "C" int mdx_cfg_ddim_step(const MdxDdimDesc* d, void* stream) {
    if (!d || !d->x || !d->eps || !d->coef || !d->step_ptr) return set_error(MDX_EINVAL, "mdx_cfg_ddim_step: null operand");
    DdimParams p{d->x, d->eps, d->coef, d->step_ptr, d->x_in, d->n, (int)d->cfg, (float)d->guidance, (int)d->xin_c, (int)d->xin_ld,
                 d->gv_cond, d->gv_noise, d->gv_mask, d->gv_mask ? (int)d->gv_mode : 0, d->gv_view_elems, (int)d->gv_last_step};
    if (p.xin_ld > 0 && (p.xin_c <= 0 || p.xin_ld < p.xin_c || p.n % p.xin_c)) return set_error(MDX_EINVAL, "ddim: bad x_in channel layout");
    if (p.gv_mode != 0 && (p.gv_mode < 0 || p.gv_mode > 2 || !p.gv_noise || (p.gv_mode == 1 && !p.gv_cond) || p.gv_view <= 0 || p.n % p.gv_view))
        return set_error(MDX_EINVAL, "ddim: given-view mode %d needs gv_noise (+ gv_cond for mode 1) and gv_view_elems dividing n", p.gv_mode);
    if (p.n <= 0) return MDX_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ddim_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
    int rc = check_launch("ddim_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, p.step);
    return check_launch("step_inc_kernel", false);
}

extern "C" int mdx_cfg_unipc_step(const MdxUniPCDesc* d, void* stream) {
    if (!d || !d->x || !d->eps || !d->coef || !d->step_ptr || !d->x_last || !d->m1 || !d->m2)
        return set_error(MDX_EINVAL, "mdx_cfg_unipc_step: null operand");
    UniPCParams p{d->x, d->eps, d->coef, d->step_ptr, d->x_in, d->x_last, d->m1, d->m2, d->n, (int)d->cfg, (float)d->guidance, (int)d->xin_c, (int)d->xin_ld,
                  d->gv_cond, d->gv_noise, d->gv_mask, d->gv_mask ? (int)d->gv_mode : 0, d->gv_view_elems, (int)d->gv_last_step};
    if (p.xin_ld > 0 && (p.xin_c <= 0 || p.xin_ld < p.xin_c || p.n % p.xin_c)) return set_error(MDX_EINVAL, "unipc: bad x_in channel layout");
    if (p.gv_mode != 0 && (p.gv_mode < 0 || p.gv_mode > 2 || !p.gv_noise || (p.gv_mode == 1 && !p.gv_cond) || p.gv_view <= 0 || p.n % p.gv_view))
        return set_error(MDX_EINVAL, "unipc: given-view mode %d needs gv_noise (+ gv_cond for mode 1) and gv_view_elems dividing n", p.gv_mode);
    if (p.n <= 0) return MDX_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(unipc_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, st, p);
    int rc = check_launch("unipc_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, p.step);
    return check_launch("step_inc_kernel", false);
}
