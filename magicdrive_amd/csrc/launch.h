// launch.h — host-side helpers: error reporting and launch checks for libmdx.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/mdx.h"

namespace mdx {

char* error_buffer();  // thread-local, 512 bytes (api.hip)
char* kernel_tag_buffer();  // thread-local, 128 bytes (api.hip): name of the last primary kernel launched (mdx_last_kernel)

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
int ensure_dyn_smem(const void* kernel, size_t bytes, const char* what);

}  // namespace mdx
