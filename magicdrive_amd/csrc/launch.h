// launch.h — host-side helpers: error reporting and launch checks for libmdx.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/mdx.h"

// The fp16 build of a kernel source (-DMDX_F16=1 -Dmdx=mdx_f16, csrc/Makefile) defines the same op entry points under *_f16 names;
// api.hip (compiled once) dispatches an op to either set by MdxOp.dtype.
#if defined(MDX_F16) && MDX_F16
#define mdx_gemm_bf16 mdx_gemm_f16
#define mdx_conv2d_bf16 mdx_conv2d_f16
#define mdx_conv2d_direct mdx_conv2d_direct_f16
#define mdx_attention_bf16 mdx_attention_f16
#define mdx_groupnorm_bf16 mdx_groupnorm_f16
#define mdx_layernorm_bf16 mdx_layernorm_f16
#define mdx_elementwise mdx_elementwise_f16
#define mdx_fourier_embed mdx_fourier_embed_f16
#define mdx_gather_rows mdx_gather_rows_f16
#define mdx_timestep_embedding mdx_timestep_embedding_f16
#define mdx_cfg_ddim_step mdx_cfg_ddim_step_f16
#define mdx_cfg_unipc_step mdx_cfg_unipc_step_f16
#define mdx_softmax_rows mdx_softmax_rows_f16
#endif

// process-wide runtime state (api.hip): ONE instance shared by the bf16 and the fp16 build of the kernels
namespace mdx_rt {
char* error_buffer();  // thread-local, 512 bytes
char* kernel_tag_buffer();  // thread-local, 128 bytes: name of the last primary kernel launched (mdx_last_kernel)
int ensure_dyn_smem(const void* kernel, size_t bytes, const char* what);
}  // namespace mdx_rt

namespace mdx {

using mdx_rt::error_buffer;
using mdx_rt::kernel_tag_buffer;

inline int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// Launch errors (bad configuration) surface through hipGetLastError without a sync.
// `primary` = the kernel that does the op's work (reduce / finalise helpers pass false and keep the previous tag).
inline int check_launch(const char* what, bool primary = true) {
    if (primary) snprintf(kernel_tag_buffer(), 128, "%s", what);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(MDX_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MDX_OK;
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device).  Thread-safe (api.hip keeps the set behind a mutex); returns MDX_OK
// or MDX_ELAUNCH with the error text set.
using mdx_rt::ensure_dyn_smem;

}  // namespace mdx
