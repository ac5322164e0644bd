// xl_dma.h — LDS-DMA plumbing shared by the XL main loops (gemm_xl.hip, gemm_xlp.hip): raw buffer descriptors in SGPRs, the
// `buffer_load_dwordx4 ... lds` statement, hand-counted waits.
#pragma once
#include "common.h"

namespace mdx {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr unsigned XL_OOB = 0x80000000u;       // voffset >= num_records (also with any soffset < 2^31 added): the load returns 0
constexpr unsigned XL_RECORDS = 0x80000000u;

template <int N>
__device__ __forceinline__ void xl_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void xl_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef __attribute__((ext_vector_type(4))) unsigned xl_rsrc_t;   // buffer descriptor words, held in SGPRs

// Raw buffer descriptor (stride 0, 2 GiB window) over `base`.  Every word is made provably wave-uniform so the inline-asm "s"
// operands below get SGPRs.
__device__ __forceinline__ xl_rsrc_t xl_make_rsrc(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    xl_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = XL_RECORDS;
    r.w = 0x00020000u;
    return r;
}

// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor base + soff + voff[lane]) to LDS bytes [lds_addr, lds_addr + 1024).
// Inline asm on purpose: hipcc models the builtin as an LDS store of unknown extent and drains `vmcnt(0)` in front of the next
// ds_read — every phase — which serialises the whole pipeline (seen in the .s of the builtin version).  An asm statement is absent
// from the compiler's wait bookkeeping: completion is counted by hand (xl_wait_vmcnt + s_barrier before any read of the slot, see
// the schedule in the kernel).  M0 (the DMA's LDS base) is compiler-reserved: saved and restored inside the statement; the s_nop
// covers the SALU-write-M0 -> LDS-DMA hazard.
__device__ __forceinline__ void xl_glds(const xl_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}

// The weight (B operand) pieces may carry a cache policy (side builds: -DMDX_XL_BPOL=1 sc1, 2 nt, 3 sc0 sc1): `sc1` loads are served by
// the L2 WITHOUT allocating in the CU's 32 KiB vector L1 (MI355X_MICROARCH.md, visibility table).  Why that could pay for the 3x3
// convolutions: the nine taps of a channel block re-read (almost) the same 256 activation lines — a 256-row A slab is exactly the L1's
// capacity — but each tap's 40 KB weight slab streams through the same L1 in between and evicts them; with the weights bypassing L1 the
// taps kx = 1, 2 (and most of ky + 1) of the A operand can hit.  Measured: profiles/r04_xl_bpol_ab.log.
#ifndef MDX_XL_BPOL
#define MDX_XL_BPOL 0
#endif
__device__ __forceinline__ void xl_glds_b(const xl_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
#if MDX_XL_BPOL == 0
    xl_glds(rs, lds_addr, voff, soff);
#else
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
#if MDX_XL_BPOL == 1
        "buffer_load_dwordx4 %2, %3, %4 offen sc1 lds\n\t"
#elif MDX_XL_BPOL == 2
        "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
#else
        "buffer_load_dwordx4 %2, %3, %4 offen sc0 sc1 lds\n\t"
#endif
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
#endif
}

// The same descriptor with a byte bound: lanes whose (voffset + soffset) reaches `records` read zeros — row / column tails of a tile
// expressed through the descriptor instead of through per-lane offsets (which then do not depend on the tile: gemm_xlp.hip).
__device__ __forceinline__ xl_rsrc_t xl_make_rsrc_bounded(const void* base, long records) {
    xl_rsrc_t r = xl_make_rsrc(base);
    r.z = __builtin_amdgcn_readfirstlane((unsigned)(records < (long)XL_RECORDS ? (records < 0 ? 0 : records) : (long)XL_RECORDS));
    return r;
}

}  // namespace mdx
