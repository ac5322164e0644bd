// attention.hip — fused flash-style attention forward for gfx950 (wave64, MFMA 32x32x16 bf16).
//
// Replaces xformers' CUTLASS fMHA (third_party/xformers/xformers/csrc/attention/cuda/fmha/
// kernel_forward.h) as called by XFormersAttnProcessor (diffusers/models/attention_processor.py:
// 1165-1171) for attn1 / attn2 / attn4 of every transformer block, incl. the MagicDrive
// cross-view attention (magicdrive/networks/blocks.py:106-222).
//
// Shape regime here: 8 heads, head dim d in {40, 80, 160}, Tq in {1400, 350, 91, 28}, Tk = Tq
// (self / cross-view) or 78+L (context).  Nothing is a multiple of 64, so everything is masked.
//
// Design (one workgroup = NW waves = NW*32 query rows of one (batch, head)):
//   * Q fragments live in registers for the whole kernel (d/16 x 4 VGPRs).
//   * K tile [64 kv][d] and V^T tile [d][64 kv] are staged into LDS with 16-byte loads.
//     V arrives already transposed from the projection GEMM (it is emitted as W_v · X^T), so the
//     PV product needs no transposing LDS reads: a lane fetches its 8 kv values with two
//     ds_read_b64.  LDS strides (d+8 resp. 64+4 elements) make both fragment reads conflict-free.
//   * S^T = K · Q^T (operands swapped) so each lane owns ONE query column: the online-softmax
//     max / sum / rescale are lane-local (one cross-half shuffle per tile), and the P^T
//     registers are directly the B operand of O^T += V^T · P^T — no LDS round trip for P.
//     The kv order inside a 16-wide MFMA k-step is permuted (kv = 4h + j, 8 + 4h + j) to match
//     the accumulator layout; V^T is read with the same permutation, so the sum is unchanged.
//   * softmax in fp32 with exp2 and a folded log2(e)*scale; running max initialised to -inf.
//   * nsrc == 2 (cross-view): the kv loop runs once per neighbour with its own softmax state and
//     the two normalised outputs are summed in registers (blocks.py:213-217).
#include "common.h"
#include "launch.h"
#include "options.h"
#include "attn2.h"

namespace mdx {

struct AttnParams {
    const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
    const int* kvmap;
    int B, H, Tq, Tk, d, nsrc;
    int joint;         // 1: one softmax over the concatenated sources; 0: per-source softmax, outputs summed
    long ldq, sQ, ldk, sK, ldv, sV, ldo, sO;
    float scale_log2;  // scale * log2(e)
    int qblocks;       // query blocks per (batch, head); >0 selects the XCD-aware 1-D grid
    int viewmap;       // with qblocks > 0: 1 = every head and query block of a view on one XCD (ATTN_SWZ = 2), 0 = per (batch, head)
};

constexpr int KVT = 64;          // kv tile
constexpr int VSTR = KVT + 4;    // V^T LDS row stride (136 B): ds_read_b64 conflict-free

template <int D16, int NW, bool TWO>
__global__ __launch_bounds__(NW * 64) void attn_kernel(AttnParams p) {
    constexpr int DT = (D16 + 1) / 2;   // 32-row d tiles of O^T
    constexpr int DP = D16 * 16;        // padded head dim for QK^T
    constexpr int KSTR = DP + 8;        // K LDS row stride (elements)
    constexpr int NT = NW * 64;
    constexpr int KTOT = KVT * (DP / 8), VTOT = DT * 32 * (KVT / 8);       // 16-byte chunks per K / V^T tile
    constexpr int KCH = (KTOT + NT - 1) / NT, VCH = (VTOT + NT - 1) / NT;  // chunks per thread
    // Staging: issue ALL of a thread's 16-byte global loads for the tile back-to-back (one latency per tile
    // instead of one per chunk), then write them to LDS.  The registers are live only across the load, not
    // across the MFMA/softmax phase, so occupancy is unaffected.  Very wide heads fall back to a streaming loop.
    constexpr bool UNR = (KCH + VCH) <= 24;
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KVT * KSTR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[DT * 32 * VSTR];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int col = lane & 31;
    // XCD-aware order (1-D grid): workgroup L runs on XCD L % 8 (observed dispatch rule; speed only).  All query
    // blocks of one (batch, head) are given to ONE XCD, back to back, so that pair's K / V^T (224 KB at T=1400)
    // is fetched into one L2 once instead of into all eight (measured: 6x over-fetch, 63 % L2 hit rate before).
    int qb, h, b;
    if (p.qblocks > 0) {
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3;
        if (p.viewmap) {          // round 4 (see attention2.hip): the heads' Q / K / O are slices of rows whose 128-byte lines all heads share
            const int per_view = p.H * p.qblocks;
            const int vl = idx / per_view, rem = idx - vl * per_view;
            b = vl * 8 + xcd;
            if (b >= p.B) return;
            h = rem / p.qblocks; qb = rem - h * p.qblocks;
        } else {
            const int bh = (idx / p.qblocks) * 8 + xcd;
            if (bh >= p.B * p.H) return;
            qb = idx % p.qblocks; b = bh / p.H; h = bh - b * p.H;
        }
    } else {
        qb = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    }
    const int d = p.d;
    const int q = qb * (NW * 32) + wave * 32 + col;

    // ---- Q fragments (B operand of S^T = K Q^T): lane -> query column, 8 consecutive dims ----
    Frag8 qf[D16];
    {
        const bf16_t* qp = p.Q + (long)b * p.sQ + (long)(q < p.Tq ? q : 0) * p.ldq + (long)h * d;
#pragma unroll
        for (int ks = 0; ks < D16; ++ks) {
            int dd = ks * 16 + half * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < p.Tq && dd < d) v = *(const uint4*)(qp + dd);
            qf[ks].u = v;
        }
    }

    f32x16_t oacc[DT];
    f32x16_t osum[TWO ? DT : 1];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    if (TWO) {
#pragma unroll
        for (int i = 0; i < (TWO ? DT : 1); ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) osum[i][r] = 0.f;
    }

    float m_run = -INFINITY;
    float l_run = 0.f;
    for (int s = 0; s < p.nsrc; ++s) {
        const int bkv = p.kvmap ? p.kvmap[b * p.nsrc + s] : b;
        const bf16_t* kbase = p.K + (long)bkv * p.sK + (long)h * d;
        const bf16_t* vbase = p.Vt + (long)bkv * p.sV + (long)h * d * p.ldv;
        if (!p.joint || s == 0) { m_run = -INFINITY; l_run = 0.f; }      // joint: the sources are ONE kv sequence (one softmax)

        for (int j0 = 0; j0 < p.Tk; j0 += KVT) {
            if constexpr (UNR) {
                uint4 kreg[KCH];
                Frag8 vreg[VCH];
#pragma unroll
                for (int i = 0; i < KCH; ++i) {
                    const int c = tid + i * NT;
                    const int row = c / (DP / 8);
                    const int cc = c - row * (DP / 8);
                    // unconditional loads from clamped addresses, masked afterwards: a guarded load (`if (ok) v = load`) compiles to its own
                    // branch with a full `s_waitcnt vmcnt(0)` inside — the tile's staging loads then ran one memory round trip after the other
                    const bool ok = c < KTOT && j0 + row < p.Tk && cc * 8 < d;
                    const int rr = min(j0 + row, p.Tk - 1), cq = min(cc * 8, d - 8);
                    const uint4 v = *(const uint4*)(kbase + (long)rr * p.ldk + cq);
                    kreg[i] = ok ? v : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < VCH; ++i) {
                    const int c = tid + i * NT;
                    const int row = c >> 3;
                    const int kv0 = j0 + (c & 7) * 8;
                    const bool ok = c < VTOT && row < d && kv0 < p.Tk;
                    Frag8 v;
                    v.u = *(const uint4*)(vbase + (long)min(row, d - 1) * p.ldv + (kv0 < p.Tk ? kv0 : 0));
                    if (!ok) v.u = make_uint4(0, 0, 0, 0);
                    vreg[i] = v;
                }
#pragma unroll
                for (int i = 0; i < KCH; ++i) {
                    const int c = tid + i * NT;
                    const int row = c / (DP / 8);
                    const int cc = c - row * (DP / 8);
                    if (c < KTOT) *(uint4*)(Ks + row * KSTR + cc * 8) = kreg[i];
                }
#pragma unroll
                for (int i = 0; i < VCH; ++i) {
                    const int c = tid + i * NT;
                    if (c < VTOT) {
                        Frag8 v = vreg[i];
                        const int kv0 = j0 + (c & 7) * 8;
                        if (kv0 + 8 > p.Tk) {          // V^T pad columns may hold anything: zero kv >= Tk
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (kv0 + e >= p.Tk) v.h[e] = 0;
                        }
                        uint2* dst = (uint2*)(Vs + (c >> 3) * VSTR + (c & 7) * 8);
                        dst[0] = v.d2[0];
                        dst[1] = v.d2[1];
                    }
                }
            } else {
                // ---- stage K tile: KVT rows x DP cols, 16-byte chunks ----
                for (int c = tid; c < KVT * (DP / 8); c += NT) {
                    int row = c / (DP / 8);
                    int cc = c - row * (DP / 8);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (j0 + row < p.Tk && cc * 8 < d) v = *(const uint4*)(kbase + (long)(j0 + row) * p.ldk + cc * 8);
                    *(uint4*)(Ks + row * KSTR + cc * 8) = v;
                }
                // ---- stage V^T tile: DT*32 rows (head dims) x KVT kv, 16-byte chunks along kv ----
                for (int c = tid; c < DT * 32 * (KVT / 8); c += NT) {
                    int row = c >> 3;
                    int cc = c & 7;
                    Frag8 v;
                    v.u = make_uint4(0, 0, 0, 0);
                    int kv0 = j0 + cc * 8;
                    if (row < d && kv0 < p.Tk) {
                        v.u = *(const uint4*)(vbase + (long)row * p.ldv + kv0);
                        if (kv0 + 8 > p.Tk) {
    #pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (kv0 + e >= p.Tk) v.h[e] = 0;
                        }
                    }
                    uint2* dst = (uint2*)(Vs + row * VSTR + cc * 8);
                    dst[0] = v.d2[0];
                    dst[1] = v.d2[1];
                }
            }
            __syncthreads();

            // ---- S^T[kv][q] for two 32-kv sub-tiles ----
            f32x16_t sacc[2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
                const bf16_t* kr = Ks + (sub * 32 + col) * KSTR + half * 8;
#pragma unroll
                for (int ks = 0; ks < D16; ++ks) {
                    Frag8 kf;
                    kf.u = *(const uint4*)(kr + ks * 16);
                    sacc[sub] = MDX_MFMA_32x32x16(kf.v, qf[ks].v, sacc[sub]);
                }
            }
            // ---- online softmax (this lane: one query, 32 of the 64 kv) ----
            // raw scores stay unscaled: max on raw values (scale > 0), then one fma + one v_exp per score.
            if (j0 + KVT > p.Tk) {                 // only the last tile has kv >= Tk to mask (wave-uniform branch)
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int kv = j0 + sub * 32 + mfma32_row(r, lane);
                        if (kv >= p.Tk) sacc[sub][r] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[sub][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;
            const float m_new = fmaxf(m_run, mx);       // finite: every tile has >= 1 valid kv
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[sub][r], p.scale_log2, -m_new));
                    sacc[sub][r] = pv;
                    psum += pv;
                }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

            // ---- O^T[dd][q] += V^T[dd][kv] * P^T[kv][q] ----
#pragma unroll
            for (int kstep = 0; kstep < 4; ++kstep) {
                const int sub = kstep >> 1, kk = kstep & 1;
                Frag8 pf;
                pf.u.x = pack2bf(sacc[sub][kk * 8 + 0], sacc[sub][kk * 8 + 1]);
                pf.u.y = pack2bf(sacc[sub][kk * 8 + 2], sacc[sub][kk * 8 + 3]);
                pf.u.z = pack2bf(sacc[sub][kk * 8 + 4], sacc[sub][kk * 8 + 5]);
                pf.u.w = pack2bf(sacc[sub][kk * 8 + 6], sacc[sub][kk * 8 + 7]);
                const bf16_t* vr = Vs + col * VSTR + kstep * 16 + 4 * half;
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    Frag8 vf;
                    vf.d2[0] = *(const uint2*)(vr + i * 32 * VSTR);
                    vf.d2[1] = *(const uint2*)(vr + i * 32 * VSTR + 8);
                    oacc[i] = MDX_MFMA_32x32x16(vf.v, pf.v, oacc[i]);
                }
            }
            __syncthreads();
        }
        // ---- finish this source (joint: only after the last one) ----
        if (p.joint && s + 1 < p.nsrc) continue;
        float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        float inv = 1.0f / l_tot;
        if (TWO) {
#pragma unroll
            for (int i = 0; i < (TWO ? DT : 1); ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    osum[i][r] += oacc[i][r] * inv;
                    oacc[i][r] = 0.f;
                }
        } else {
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= inv;
        }
    }

    // ---- store O[q][h*d + dd]: lane has 4 consecutive dd per register group ----
    if (q < p.Tq) {
        bf16_t* op = p.O + (long)b * p.sO + (long)q * p.ldo + (long)h * d;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int dd = i * 32 + 8 * g + 4 * half;
                if (dd < d) {
                    const f32x16_t& src = TWO ? osum[TWO ? i : 0] : oacc[i];
                    uint2 ov;
                    ov.x = pack2bf(src[4 * g], src[4 * g + 1]);
                    ov.y = pack2bf(src[4 * g + 2], src[4 * g + 3]);
                    *(uint2*)(op + dd) = ov;
                }
            }
    }
}

template <int D16, int NW>
static int launch_attn(const AttnParams& p, hipStream_t st) {
    AttnParams q = p;
    const int qblocks = (p.Tq + NW * 32 - 1) / (NW * 32);
    const int swz = (int)opt(OPT_ATTN_SWZ);
    dim3 grid(qblocks, p.H, p.B);
    q.qblocks = 0;
    q.viewmap = 0;
    if (swz && qblocks > 1) {
        q.qblocks = qblocks;
        q.viewmap = swz >= 2;
        grid = q.viewmap ? dim3((unsigned)(((long)p.B + 7) / 8 * 8 * p.H * qblocks), 1, 1)
                         : dim3((unsigned)(((long)p.B * p.H + 7) / 8 * 8 * qblocks), 1, 1);
    }
    const bool two = p.nsrc == 2 && !p.joint;
    if (two)
        hipLaunchKernelGGL((attn_kernel<D16, NW, true>), grid, dim3(NW * 64), 0, st, q);
    else
        hipLaunchKernelGGL((attn_kernel<D16, NW, false>), grid, dim3(NW * 64), 0, st, q);
    char tag[64];
    snprintf(tag, sizeof tag, "attn_kernel<%d,%d,%s>", D16, NW, two ? "xview" : (p.nsrc > 1 ? "joint" : "self"));
    return check_launch(tag);
}

template <int D16>
static int launch_attn_nw(const AttnParams& p, hipStream_t st) {
    // 4 waves (128 queries) per workgroup whenever (batch x heads) alone fills the chip — also for 91- and 28-token sequences: every
    // thread of the workgroup stages K / V^T, so the idle query rows cost less than narrower workgroups lose on staging (384 views,
    // T = 91, d = 160: 4 waves 103 us, 2 waves 152 us, 1 wave 448 us; T = 28: 39 / 44 / 71 us; cross-view at T = 28: 2 waves 73 vs 108 us).
    // With few (batch, head) pairs the sequence is split finer so that short sequences still spread over the chip.
    long blocks4 = (long)((p.Tq + 127) / 128) * p.H * p.B;
    const long thr4 = (long)opt(OPT_ATTN_NW4_BLOCKS);
    const long thr8 = (long)opt(OPT_ATTN_NW8_BLOCKS);
    long blocks8 = (long)((p.Tq + 255) / 256) * p.H * p.B;
    const int force = (int)opt(OPT_ATTN_NW);     // experiments: force 1 / 2 / 4 / 8 waves
    if (force == 8) return launch_attn<D16, 8>(p, st);
    if (force == 4) return launch_attn<D16, 4>(p, st);
    if (force == 2) return launch_attn<D16, 2>(p, st);
    if (force == 1) return launch_attn<D16, 1>(p, st);
    if (p.Tq >= 512 && blocks8 >= thr8) return launch_attn<D16, 8>(p, st);
    if (p.Tq >= 256 && blocks4 >= thr4) return launch_attn<D16, 4>(p, st);
    if ((long)p.H * p.B >= 2 * thr4) return (p.Tq <= 32 && p.nsrc == 2) ? launch_attn<D16, 2>(p, st) : launch_attn<D16, 4>(p, st);
    if (p.Tq >= 64) return launch_attn<D16, 2>(p, st);
    return launch_attn<D16, 1>(p, st);
}

}  // namespace mdx

using namespace mdx;

extern "C" int mdx_attention_bf16(const MdxAttnDesc* a, void* stream) {
    if (!a || !a->Q || !a->K || !a->Vt || !a->O) return set_error(MDX_EINVAL, "mdx_attention_bf16: null operand");
    if (a->d % 8 || a->d <= 0 || a->d > 160) return set_error(MDX_EINVAL, "head dim %ld unsupported (d %% 8 == 0, d <= 160)", (long)a->d);
    if (a->joint != 0 && a->joint != 1) return set_error(MDX_EINVAL, "joint must be 0 or 1");
    if (a->nsrc < 1 || a->nsrc > (a->joint ? 8 : 2)) return set_error(MDX_EINVAL, "nsrc must be 1 or 2 (joint: 1..8)");
    if (a->nsrc > 1 && !a->kvmap) return set_error(MDX_EINVAL, "nsrc > 1 needs a kvmap");
    if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->sQ % 8) || (a->sK % 8) || (a->sV % 8) || (a->ldo % 4) || (a->sO % 4))
        return set_error(MDX_EINVAL, "attention strides must be multiples of 8 (Q,K,Vt) / 4 (O)");
    if (((uintptr_t)a->Q & 15) || ((uintptr_t)a->K & 15) || ((uintptr_t)a->Vt & 15) || ((uintptr_t)a->O & 7))
        return set_error(MDX_EINVAL, "attention operands must be 16-byte aligned");
    if (a->Tq <= 0 || a->Tk <= 0 || a->B <= 0 || a->H <= 0) return MDX_OK;
    if (a->ldv < a->Tk) return set_error(MDX_EINVAL, "ldv < Tk");
    AttnParams p;
    p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.Vt = (const bf16_t*)a->Vt; p.O = (bf16_t*)a->O;
    p.kvmap = a->kvmap;
    p.B = (int)a->B; p.H = (int)a->H; p.Tq = (int)a->Tq; p.Tk = (int)a->Tk; p.d = (int)a->d; p.nsrc = (int)a->nsrc; p.joint = (int)a->joint;
    p.ldq = a->ldq; p.sQ = a->sQ; p.ldk = a->ldk; p.sK = a->sK; p.ldv = a->ldv; p.sV = a->sV; p.ldo = a->ldo; p.sO = a->sO;
    // q_prescaled: the caller folded scale * log2(e) into the query projection: scores are base-2 exponents already
    if (a->q_prescaled != 0 && a->q_prescaled != 1) return set_error(MDX_EINVAL, "q_prescaled must be 0 or 1");
    p.scale_log2 = a->q_prescaled ? 1.0f : (float)(a->scale * 1.4426950408889634);
    hipStream_t st = (hipStream_t)stream;
    {
        Attn2Params p2;
        p2.Q = p.Q; p2.K = p.K; p2.Vt = p.Vt; p2.O = p.O; p2.kvmap = p.kvmap;
        p2.B = p.B; p2.H = p.H; p2.Tq = p.Tq; p2.Tk = p.Tk; p2.d = p.d; p2.nsrc = p.nsrc; p2.joint = p.joint;
        p2.ldq = p.ldq; p2.sQ = p.sQ; p2.ldk = p.ldk; p2.sK = p.sK; p2.ldv = p.ldv; p2.sV = p.sV; p2.ldo = p.ldo; p2.sO = p.sO;
        p2.scale_log2 = p.scale_log2; p2.qblocks = 0; p2.q_prescaled = (int)a->q_prescaled;
        if (attn2_supported(p2)) return launch_attn2(p2, st);
    }
    int d16 = (int)(a->d + 15) / 16;
    switch (d16) {
        case 1: return launch_attn_nw<1>(p, st);
        case 2: return launch_attn_nw<2>(p, st);
        case 3: return launch_attn_nw<3>(p, st);
        case 4: return launch_attn_nw<4>(p, st);
        case 5: return launch_attn_nw<5>(p, st);
        case 6: return launch_attn_nw<6>(p, st);
        case 8: return launch_attn_nw<8>(p, st);
        case 10: return launch_attn_nw<10>(p, st);
        default: return set_error(MDX_EUNSUPPORTED, "head dim %ld: no kernel instance (d/16 = %d)", (long)a->d, d16);
    }
}
