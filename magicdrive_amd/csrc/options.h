// options.h — the tuning / routing switches of libmdx, in ONE table.
//
// Every switch is a named 64-bit integer with a default.  Product code reads them through opt(OPT_x) on every launch (an array
// load); mdx_set_option(key, value) / mdx_get_option(key, &value) (include/mdx.h) change and query them in-process — tests force a
// route without spawning an interpreter, an embedding application never has to touch the environment.  For the measurement scripts
// under tools/ each switch can still be PRE-SET through the environment variable MDX_<KEY>, read once when the library is loaded.
#pragma once
#include <cstdint>

namespace mdx_rt {
int64_t opt(int id);                 // current value of a switch (api.hip; one table for the bf16 and the fp16 kernels)
}

namespace mdx {

#define MDX_OPTIONS(X) \
    X(ATTN_SWZ, 2, "XCD-aware block order of attention.hip: 0 off, 1 per (view, head), 2 per view (all heads of a view on one XCD)") \
    X(ATTN_NW4_BLOCKS, 256L, "attention.hip: 4-wave workgroups from this many blocks") \
    X(ATTN_NW8_BLOCKS, (1L << 40), "attention.hip: 8-wave workgroups from this many blocks") \
    X(ATTN_NW, 0, "force the waves per workgroup of attention.hip (0 = heuristic)") \
    X(ATTN2, 1, "attention2.hip for head dim 40") \
    X(ATTN3, 0, "only in `make ATTN3=1` builds: tools/attn3/attention3.hip (permute-free P, QK of the next kv tile beside this tile's softmax) for the head-dim-40 FOLD launches: self, cross-view, joint") \
    X(ATTN3_WGS, 0, "attention3.hip: persistent workgroups per XCD (0 = automatic: two per CU)") \
    X(ATTN2_PF, 1, "attention2.hip, head dim 40 FOLD 32-query form: permute-free P (PV on 32x32x16 MFMAs, K rows permuted so that a lane's scores are its PV operand)") \
    X(ATTN2_D80, 2, "attention2.hip for head dim 80: 0 never, 1 always, 2 only the two-source cross-view form") \
    X(ATTN2_FOLD, 1, "attention2.hip: subtract the running maximum inside the QK MFMA when Q is pre-scaled (head dim 40)") \
    X(ATTN2_RES, 1, "attention2.hip: kv sequences of <= 3 tiles (text context) resident in LDS, one workgroup per (view, head) walks the query blocks: 0 off, 1 for launches of >= 1024 (view, head) pairs, 2 whenever supported") \
    X(ATTN2_VIEWMAP, 1, "attention2.hip block order: 1 = all heads and query blocks of a view on one XCD (a row's 128-byte lines are shared by the heads), 0 = per (view, head)") \
    X(ATTN2_QT, 0, "32-query tiles per wave in attention2.hip: 0 = automatic (2; 1 for one-source FOLD launches), 1 / 2 = force") \
    X(CONV_OUT_WS, 1, "direct conv with Cout <= 4 and a long K (conv_out): weight-stationary K-parallel kernel (weights in registers, 16 pixels per wave)") \
    X(GEMM_SWZ, 1, "XCD-aware tile order of the generic kernel") \
    X(GEMM_SMALL_TILES, 1, "generic tile choice for small grids (1-4 scenes per call): 64 x 64 for plain GEMMs up to 1408 such tiles and for convs up to 40 k (tile, slab) units, 128 x 128 for larger convs with M < 2048, 128-row GEGLU from M = 512; 0 = the rounds 1-5 rule (128 rows from M = 2048, else 64 x 128)") \
    X(GEMM_BM, 0, "force the generic tile's rows (64 / 128; 0 = heuristic)") \
    X(GEMM_BN, 0, "force the generic tile's columns (64 / 128; 0 = heuristic)") \
    X(GEMM_PIPE, 1, "software-pipelined fragment reads in the generic 128x128x64 tile") \
    X(EPI_WIDE, 1, "16-byte epilogue accesses when alignment allows") \
    X(GEMM_WS, 1, "gemm_ws.hip: 0 off, 1 when M >= 8192, 2 whenever supported") \
    X(GEMM_XL, 1, "gemm_xl.hip: 0 off, 1 cost model, 2 whenever supported") \
    X(XL_K320, 0, "let XL take the K = 320 projections from gemm_ws.hip") \
    X(XL_MIN_TILES, 64, "fewest 256-row tiles an XL launch must have (round 6: 64; 160 until then — 2...8 scenes per call are 2-6 % faster with the lower bound, 12+ scenes and 1 scene do not care: profiles/r06_lat1_small_grids.log)") \
    X(XL_BN, 0, "force an XL tile width (160 / 256 / 320; 0 = cost model)") \
    X(XL_GEGLU320, 0, "K = 320 GEGLU on the 256-wide XL tile instead of gemm_ws.hip") \
    X(GEMM_BM256, 0, "generic 256x128 8-wave tile from M >= value (0 = never)") \
    X(GEMM_TIMING, 0, "s_memtime stamps of the generic kernel into the op workspace") \
    X(GEMM_BK, 64, "generic tile slab depth (64 or 32)") \
    X(GEMM_FLATTEN, 1, "batched shared-A GEMM as ONE col_split XL launch") \
    X(CONV_CIMAJOR, 1, "channel-block-major K order of implicit-GEMM convs") \
    X(WS_SLOTS, 512, "workgroup slots the gemm_ws M walkers are sized for") \
    X(WS_DBG, 0, "gemm_ws ablation bits (wrong results)") \
    X(XL_DBG, 0, "ablation bits, only in -DMDX_XL_ABLATE builds") \
    X(XL_TIMING, 0, "per-workgroup s_memtime stamps into the op workspace") \
    X(XL_SCHED, 0, "XL main-loop schedule variant 0..3 (0 = four quadrant phases)") \
    X(GN_REVERSE, 1, "two-stage GroupNorm: statistics pass reads the tensor back to front (Infinity-Cache reuse between producer / passes)") \
    X(GN_FINALIZE_CHUNKS, 16, "two-stage GroupNorm: with more chunks per image than this the chunk partials are combined once by gn_finalize_kernel (0 = always in the apply pass)") \
    X(GN_ONE_KERNEL_ELEMS, (4L << 20), "GroupNorm tensors of at most this many elements (all images) take the one-launch kernel (one workgroup per (image, group)) instead of the two streaming passes") \
    X(GN_TWO_STAGE, 1, "streaming two-stage GroupNorm for maps >= 32768 elements") \
    X(XL_PERSIST, 1, "256x256 XL GEMMs (plain / GEGLU, optional residual) on the persistent kernel gemm_xlp_kernel") \
    X(XD, 0, "W-direct persistent GEMM (gemm_xd.hip: weights global -> registers, finished tile stored under the next tile's main loop) for the 256x256 XL GEMMs whose descriptor carries Wq (K % 128 == 0, K >= 640); bit-identical to gemm_xlp_kernel, measured 0.89-1.04x of it (profiles/r06_xd_ab.log): off by default") \
    X(XL_RASTER, 2, "XL tile order: 0 row-major, 1 XCD-strided M-tiles, 2 XCD-blocked (M-group x N-group panels per XCD)") \
    X(XL_KXSHARE, 1, "only in -DMDX_XL_KXS side builds: 320-wide XL 3x3 / stride 1 convs share one A slab in LDS between the three horizontal taps of a (channel block, ky) (schedule 4; measured slower, gemm_xl.hip: launch_gemm_xl)") \
    X(XL_GM, 0, "XL_RASTER 2: force the M-tiles per panel (0 = cost model, xl_layout.h: raster_shape)") \
    X(XL_GN, 0, "XL_RASTER 2: force the N-tiles per panel (0 = cost model)") \
    X(STREAMS, 2, "host side (pipeline): HIP streams a pipe() call spreads its scene chunks over (chunks of >= 16 scenes, one plan + hipGraph each)") \
    X(PLAN_CACHE, 6, "host side (pipeline): sampler plans (captured graphs + buffers) kept per pipeline (LRU); a multi-stream call pins one plan per chunk") \
    X(CHECK_TIMESTEPS, 0, "host side (schedulers): 1 = a device-resident timestep that is not in the scheduler's list raises (one sync per step) instead of poisoning the sample with NaN") \
    X(LN_FUSE, 1, "MdxGemmDesc.ln_eps: 1 = gemm_ws.hip normalises the rows in-kernel, 0 = always normalise into ln_scratch first (A/B)") \
    X(LN_STATS, 1, "MdxGemmDesc.rowstat_out / ln_stats: 1 = row statistics travel from the producer's store phase to the fused LayerNorm, 0 = round-5 behaviour (nothing emitted, in-kernel sums, GEGLU through ln_scratch) (A/B)")

enum Opt : int {
#define MDX_OPT_ENUM(key, dflt, doc) OPT_##key,
    MDX_OPTIONS(MDX_OPT_ENUM)
#undef MDX_OPT_ENUM
    OPT_COUNT
};

using mdx_rt::opt;

}  // namespace mdx
