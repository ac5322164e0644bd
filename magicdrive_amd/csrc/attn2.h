// attn2.h — parameters of the lean attention kernel (attention2.hip), shared with the dispatcher in attention.hip
#pragma once
#include "common.h"
namespace mdx {
struct Attn2Params {
    const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
    const int* kvmap;
    int B, H, Tq, Tk, d, nsrc;
    int joint;         // 1: one softmax over the concatenation of the nsrc sources; 0: per-source softmax, outputs summed (nsrc <= 2)
    long ldq, sQ, ldk, sK, ldv, sV, ldo, sO;
    float scale_log2;  // scale * log2(e); 1 when q_prescaled
    int q_prescaled;   // Q already carries scale * log2(e) (MdxAttnDesc.q_prescaled)
    int qblocks;       // query blocks per (batch, head), filled in by launch_attn2
    int viewmap;       // block order (launch_attn2, option ATTN2_VIEWMAP): 1 = every head and query block of a view on ONE XCD

};
bool attn2_supported(const Attn2Params& p);
int launch_attn2(const Attn2Params& p, hipStream_t st);
// attention3.hip: the pipelined head-dim-40 FOLD form (called from launch_attn2)
bool attn3_supported(const Attn2Params& p);
int launch_attn3(const Attn2Params& p, hipStream_t st);
}  // namespace mdx
