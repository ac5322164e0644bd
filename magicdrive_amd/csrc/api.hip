// api.hip — C-ABI plumbing of libmdx: error buffer, op-program executor, hipGraph capture.
//
// A "program" is the flattened per-step (or prologue) op list the Python host builds once per
// (model, batch shape): ~700 descriptors with every device pointer already resolved.  Running it
// is one FFI call; capturing it into a hipGraph removes the per-kernel launch cost of the
// ~1.2k-launch reference step — per-replay variation (the DDIM step index) lives in
// device memory (sel_ptr / step_ptr), never in kernel arguments.
#include <cstring>
#include <mutex>
#include <set>
#include <utility>
#include "launch.h"
#include "options.h"
#include <atomic>
#include <cstdlib>

namespace mdx_rt {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
char* kernel_tag_buffer() {
    static thread_local char buf[128] = {0};
    return buf;
}
int ensure_dyn_smem(const void* kernel, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return mdx::set_error(MDX_ELAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(kernel, dev);
    if (done.count(key)) return MDX_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return mdx::set_error(MDX_ELAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    done.insert(key);
    return MDX_OK;
}
}  // namespace mdx_rt

namespace mdx {
// ---- options (options.h) ----------------------------------------------------------------------------------------------
namespace {
struct OptRow { const char* key; int64_t dflt; const char* doc; };
const OptRow kOptRows[OPT_COUNT] = {
#define MDX_OPT_ROW(key, dflt, doc) {#key, (int64_t)(dflt), doc},
    MDX_OPTIONS(MDX_OPT_ROW)
#undef MDX_OPT_ROW
};
std::atomic<int64_t> g_opt[OPT_COUNT];
// defaults, then the MDX_<KEY> environment presets — once, when the library is loaded (before any launch can read a value)
struct OptInit {
    OptInit() {
        for (int i = 0; i < OPT_COUNT; ++i) {
            int64_t v = kOptRows[i].dflt;
            char name[64];
            snprintf(name, sizeof name, "MDX_%s", kOptRows[i].key);
            if (const char* e = getenv(name)) v = strtoll(e, nullptr, 0);
            g_opt[i].store(v, std::memory_order_relaxed);
        }
    }
} g_opt_init;
int find_opt(const char* key) {
    if (!key) return -1;
    if (!strncmp(key, "MDX_", 4)) key += 4;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(key, kOptRows[i].key)) return i;
    return -1;
}
}  // namespace
}  // namespace mdx
namespace mdx_rt {
int64_t opt(int id) { return mdx::g_opt[id].load(std::memory_order_relaxed); }
}

using namespace mdx;

extern "C" int mdx_set_option(const char* key, int64_t value) {
    const int i = find_opt(key);
    if (i < 0) return set_error(MDX_EINVAL, "mdx_set_option: unknown key '%s' (see magicdrive_amd/csrc/options.h)", key ? key : "(null)");
    g_opt[i].store(value, std::memory_order_relaxed);
    return MDX_OK;
}
extern "C" int mdx_get_option(const char* key, int64_t* value_out) {
    const int i = find_opt(key);
    if (i < 0 || !value_out) return set_error(MDX_EINVAL, "mdx_get_option: unknown key '%s'", key ? key : "(null)");
    *value_out = opt(i);
    return MDX_OK;
}
extern "C" const char* mdx_option_name(int64_t index) { return (index >= 0 && index < OPT_COUNT) ? kOptRows[index].key : nullptr; }

static int run_op(const MdxOp* op, hipStream_t st) {
    const void* d = op->desc;
    if (op->dtype == MDX_DTYPE_F16) {
        switch (op->opcode) {
            case MDX_OP_GEMM: return mdx_gemm_f16((const MdxGemmDesc*)d, st);
            case MDX_OP_CONV: return mdx_conv2d_f16((const MdxConvDesc*)d, st);
            case MDX_OP_CONV_DIRECT: return mdx_conv2d_direct_f16((const MdxConvDirectDesc*)d, st);
            case MDX_OP_ATTN: return mdx_attention_f16((const MdxAttnDesc*)d, st);
            case MDX_OP_GROUPNORM: return mdx_groupnorm_f16((const MdxGroupNormDesc*)d, st);
            case MDX_OP_LAYERNORM: return mdx_layernorm_f16((const MdxLayerNormDesc*)d, st);
            case MDX_OP_EW: return mdx_elementwise_f16((const MdxEwDesc*)d, st);
            case MDX_OP_FOURIER: return mdx_fourier_embed_f16((const MdxFourierDesc*)d, st);
            case MDX_OP_GATHER: return mdx_gather_rows_f16((const MdxGatherDesc*)d, st);
            case MDX_OP_TIMEEMB: return mdx_timestep_embedding_f16((const MdxTimeEmbDesc*)d, st);
            case MDX_OP_DDIM: return mdx_cfg_ddim_step_f16((const MdxDdimDesc*)d, st);
            case MDX_OP_UNIPC: return mdx_cfg_unipc_step_f16((const MdxUniPCDesc*)d, st);
            case MDX_OP_SOFTMAX: return mdx_softmax_rows_f16((const MdxSoftmaxDesc*)d, st);
            default: return set_error(MDX_EINVAL, "unknown opcode %ld", (long)op->opcode);
        }
    }
    if (op->dtype != MDX_DTYPE_BF16) return set_error(MDX_EINVAL, "MdxOp.dtype %ld: 0 (bf16) or 1 (fp16)", (long)op->dtype);
    switch (op->opcode) {
        case MDX_OP_GEMM: return mdx_gemm_bf16((const MdxGemmDesc*)d, st);
        case MDX_OP_CONV: return mdx_conv2d_bf16((const MdxConvDesc*)d, st);
        case MDX_OP_CONV_DIRECT: return mdx_conv2d_direct((const MdxConvDirectDesc*)d, st);
        case MDX_OP_ATTN: return mdx_attention_bf16((const MdxAttnDesc*)d, st);
        case MDX_OP_GROUPNORM: return mdx_groupnorm_bf16((const MdxGroupNormDesc*)d, st);
        case MDX_OP_LAYERNORM: return mdx_layernorm_bf16((const MdxLayerNormDesc*)d, st);
        case MDX_OP_EW: return mdx_elementwise((const MdxEwDesc*)d, st);
        case MDX_OP_FOURIER: return mdx_fourier_embed((const MdxFourierDesc*)d, st);
        case MDX_OP_GATHER: return mdx_gather_rows((const MdxGatherDesc*)d, st);
        case MDX_OP_TIMEEMB: return mdx_timestep_embedding((const MdxTimeEmbDesc*)d, st);
        case MDX_OP_DDIM: return mdx_cfg_ddim_step((const MdxDdimDesc*)d, st);
        case MDX_OP_UNIPC: return mdx_cfg_unipc_step((const MdxUniPCDesc*)d, st);
        case MDX_OP_SOFTMAX: return mdx_softmax_rows((const MdxSoftmaxDesc*)d, st);
        default: return set_error(MDX_EINVAL, "unknown opcode %ld", (long)op->opcode);
    }
}

extern "C" int mdx_program_run(const MdxOp* ops, int64_t n, void* stream) {
    if (!ops && n > 0) return set_error(MDX_EINVAL, "mdx_program_run: null program");
    for (int64_t i = 0; i < n; ++i) {
        int rc = run_op(&ops[i], (hipStream_t)stream);
        if (rc != MDX_OK) {
            char tmp[400];
            strncpy(tmp, error_buffer(), sizeof(tmp) - 1);
            tmp[sizeof(tmp) - 1] = 0;
            return set_error(rc, "op %ld (opcode %ld): %s", (long)i, (long)ops[i].opcode, tmp);
        }
    }
    return MDX_OK;
}

struct GraphHandle {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int mdx_graph_create(const MdxOp* ops, int64_t n, void** graph_out) {
    if (!graph_out) return set_error(MDX_EINVAL, "mdx_graph_create: null out");
    *graph_out = nullptr;
    hipStream_t cs;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) return set_error(MDX_ELAUNCH, "hipStreamCreate: %s", hipGetErrorString(e));
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { (void)hipStreamDestroy(cs); return set_error(MDX_ELAUNCH, "BeginCapture: %s", hipGetErrorString(e)); }
    int rc = mdx_program_run(ops, n, cs);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cs, &g);
    (void)hipStreamDestroy(cs);
    if (rc != MDX_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) return set_error(MDX_ELAUNCH, "EndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return set_error(MDX_ELAUNCH, "GraphInstantiate: %s", hipGetErrorString(e)); }
    GraphHandle* h = new GraphHandle{g, ex};
    *graph_out = h;
    return MDX_OK;
}

extern "C" int mdx_graph_launch(void* graph, void* stream) {
    if (!graph) return set_error(MDX_EINVAL, "mdx_graph_launch: null graph");
    hipError_t e = hipGraphLaunch(((GraphHandle*)graph)->exec, (hipStream_t)stream);
    if (e != hipSuccess) return set_error(MDX_ELAUNCH, "hipGraphLaunch: %s", hipGetErrorString(e));
    return MDX_OK;
}

extern "C" int mdx_graph_destroy(void* graph) {
    if (!graph) return MDX_OK;
    GraphHandle* h = (GraphHandle*)graph;
    (void)hipGraphExecDestroy(h->exec);
    (void)hipGraphDestroy(h->graph);
    delete h;
    return MDX_OK;
}

extern "C" int mdx_abi_version(void) { return MDX_ABI_VERSION; }
#ifndef MDX_BUILD_ID
#define MDX_BUILD_ID "unknown"
#endif
extern "C" const char* mdx_build_id(void) { return MDX_BUILD_ID; }
extern "C" const char* mdx_last_error(void) { return error_buffer(); }

extern "C" const char* mdx_last_kernel(void) { return kernel_tag_buffer(); }

extern "C" int mdx_device_info(int64_t* out3) {
    if (!out3) return set_error(MDX_EINVAL, "mdx_device_info: null out");
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(MDX_ELAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t pr;
    e = hipGetDeviceProperties(&pr, dev);
    if (e != hipSuccess) return set_error(MDX_ELAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    out3[0] = pr.multiProcessorCount;
    out3[1] = pr.clockRate;
    out3[2] = (int64_t)pr.totalGlobalMem;
    return MDX_OK;
}
