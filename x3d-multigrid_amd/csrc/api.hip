// Error reporting and version of the C ABI (include/x3dhip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void x3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* x3d_last_error(void) { return g_err; }

// name of the kernel template the last launching entry point of this thread chose (measurement aid: bench.py groups its
// HIP-event timings by it, so that they line up with the kernel names of a rocprofv3 summary)
static thread_local const char* g_last_kernel = "";
void x3d_note_kernel(const char* name) { g_last_kernel = name; }
extern "C" const char* x3d_last_kernel(void) { return g_last_kernel; }
extern "C" int x3d_abi_version(void) { return X3D_ABI_VERSION; }

// ---------------------------------------------------------------------------------------
// Tuning / A-B options (include/x3dhip.h: x3d_set_option).  ONE table, read at call time by every entry point through
// x3d_opt(): a test can flip an option in-process (e.g. a small persistent grid to drive the multi-chunk loops of the fused
// kernels with small tensors) and the tile-count queries see the same value as the launch that follows.  Each option
// takes its initial value from an environment variable once, at first use (the round 1-2 switches keep working from the
// shell); nothing else in the library calls getenv.
// ---------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <stdlib.h>

namespace {
struct OptDef { const char* name; const char* env; int def; int env_flag; };   // env_flag: the variable's presence means 1
const OptDef kOpts[X3D_OPT_COUNT] = {
    /* X3D_OPT_FB_GRID        */ {"fb_grid", "X3D_FB_GRID", 512, 0},
    /* X3D_OPT_PW_PGRID       */ {"pw_pgrid", "X3D_PW_PGRID", 512, 0},
    /* X3D_OPT_PW_NT4_MIN     */ {"pw_nt4_min", "X3D_PW_NT4_MIN", 256, 0},
    /* X3D_OPT_PW_NO_PERSIST  */ {"pw_no_persist", "X3D_PW_NO_PERSIST", 0, 1},
    /* X3D_OPT_DW_TH          */ {"dw_th", "X3D_DW_TH", 16, 0},
    /* X3D_OPT_DW_BALANCE     */ {"dw_balance", "X3D_DW_BALANCE", 1, 0},
    /* X3D_OPT_DW_NO_V2       */ {"dw_no_v2", "X3D_DW_NO_V2", 0, 1},
    /* X3D_OPT_NO_PW6         */ {"no_pw6", "X3D_NO_PW6", 0, 1},
    /* X3D_OPT_NO_PW7         */ {"no_pw7", "X3D_NO_PW7", 0, 1},
    /* X3D_OPT_NO_PWFS        */ {"no_pwfs", "X3D_NO_PWFS", 0, 1},
    /* X3D_OPT_DGRAD_F32      */ {"dgrad_f32", "X3D_DGRAD_F32", 0, 1},
    /* X3D_OPT_WGRAD_F32      */ {"wgrad_f32", "X3D_WGRAD_F32", 0, 1},
    /* X3D_OPT_BWD_TERMS      */ {"bwd_terms", "X3D_BWD_TERMS", 3, 0},
    /* X3D_OPT_NO_WGRAD4      */ {"no_wgrad4", "X3D_NO_WGRAD4", 0, 1},
    /* X3D_OPT_WG_CPW         */ {"wg_cpw", "X3D_WG_CPW", 8, 0},
    /* X3D_OPT_WG_CAP         */ {"wg_cap", "X3D_WG_CAP", 256, 0},
    /* X3D_OPT_STEM_WG_CAP    */ {"stem_wg_cap", "X3D_STEM_WG_CAP", 512, 0},
    /* X3D_OPT_DW_TSPLIT_WGS  */ {"dw_tsplit_wgs", "X3D_DW_TSPLIT_WGS", 256, 0},
    /* X3D_OPT_DW_CPB_MAX     */ {"dw_cpb_max", "X3D_DW_CPB_MAX", 16, 0},
    /* X3D_OPT_PW6_MIN_M      */ {"pw6_min_m", "X3D_PW6_MIN_M", 96, 0},
    /* X3D_OPT_PW_TWO_TILES_K */ {"pw_two_tiles_k", "X3D_PW_TWO_TILES_K", 320, 0},
    /* X3D_OPT_DW_TSPLIT_WGS_FWD */ {"dw_tsplit_wgs_fwd", "X3D_DW_TSPLIT_WGS_FWD", 512, 0},
    /* X3D_OPT_NO_PW8         */ {"no_pw8", "X3D_NO_PW8", 0, 1},
    /* X3D_OPT_PW8_GRID       */ {"pw8_grid", "X3D_PW8_GRID", 0, 0},
    /* X3D_OPT_PW8_MAX_K      */ {"pw8_max_k", "X3D_PW8_MAX_K", 128, 0},
    /* X3D_OPT_NO_SE_BWD_MERGE */ {"no_se_bwd_merge", "X3D_NO_SE_BWD_MERGE", 0, 1},
    /* X3D_OPT_PW_WAVES16     */ {"pw_waves16", "X3D_PW_WAVES16", 2, 0},
    /* X3D_OPT_DW_TQUAD_WGS   */ {"dw_tquad_wgs", "X3D_DW_TQUAD_WGS", 0, 0},
    /* X3D_OPT_DW_TQUAD_WGS_FWD */ {"dw_tquad_wgs_fwd", "X3D_DW_TQUAD_WGS_FWD", 0, 0},
};
std::atomic<int> g_opt[X3D_OPT_COUNT];
std::once_flag g_opt_once;

bool opt_valid(int id, int v);
// initial value of an option: its default, or the environment variable of the same meaning when that holds a value
// x3d_set_option would accept (an out-of-range value -- X3D_DW_TH=0 would divide by zero in the tile geometry -- is
// ignored with a warning on stderr)
int opt_initial(int id) {
    const OptDef& d = kOpts[id];
    const char* e = getenv(d.env);
    if (e == nullptr) return d.def;
    const int v = d.env_flag ? 1 : atoi(e);
    if (!opt_valid(id, v)) {
        fprintf(stderr, "libx3dhip: %s=%s is out of range for option '%s': using the default %d\n", d.env, e, d.name, d.def);
        return d.def;
    }
    return v;
}
void opt_init() {
    for (int i = 0; i < X3D_OPT_COUNT; ++i) g_opt[i].store(opt_initial(i), std::memory_order_relaxed);
}
int opt_find(const char* name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < X3D_OPT_COUNT; ++i) if (strcmp(name, kOpts[i].name) == 0) return i;
    return -1;
}
bool opt_valid(int id, int v) {
    switch (id) {
        case X3D_OPT_FB_GRID: case X3D_OPT_PW_PGRID: case X3D_OPT_WG_CAP: case X3D_OPT_STEM_WG_CAP: return v >= 1 && v <= 65535;
        case X3D_OPT_PW_NT4_MIN: case X3D_OPT_DW_TSPLIT_WGS: case X3D_OPT_DW_TSPLIT_WGS_FWD: return v >= 0;
        case X3D_OPT_DW_TQUAD_WGS: case X3D_OPT_DW_TQUAD_WGS_FWD: return v >= 0;
        case X3D_OPT_PW8_GRID: return v >= 0 && v <= 65535;
        case X3D_OPT_PW8_MAX_K: return v >= 0 && v <= 224;
        case X3D_OPT_PW_WAVES16: return v >= 0 && v <= 3;
        case X3D_OPT_DW_TH: case X3D_OPT_DW_CPB_MAX: return v >= 1 && v <= 16;
        case X3D_OPT_WG_CPW: return v >= 1 && v <= 4096;
        case X3D_OPT_PW6_MIN_M: case X3D_OPT_PW_TWO_TILES_K: return v >= 16 && v <= 4096;
        case X3D_OPT_BWD_TERMS: return v == 2 || v == 3;
        default: return v == 0 || v == 1;
    }
}
}  // namespace

int x3d_opt(int id) {
    std::call_once(g_opt_once, opt_init);
    return g_opt[id].load(std::memory_order_relaxed);
}

// CUs of the current device (cached per device ordinal: the library may serve several devices from several threads)
int x3d_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

extern "C" int x3d_set_option(const char* name, int value) {
    std::call_once(g_opt_once, opt_init);
    const int id = opt_find(name);
    if (id < 0) { x3d_set_error("x3d_set_option: unknown option '%s'", name ? name : "(null)"); return X3D_EINVAL; }
    if (!opt_valid(id, value)) { x3d_set_error("x3d_set_option: value %d is out of range for '%s'", value, name); return X3D_EINVAL; }
    g_opt[id].store(value, std::memory_order_relaxed);
    return X3D_OK;
}

extern "C" int x3d_get_option(const char* name, int* value) {
    std::call_once(g_opt_once, opt_init);
    const int id = opt_find(name);
    if (id < 0 || value == nullptr) { x3d_set_error("x3d_get_option: unknown option '%s'", name ? name : "(null)"); return X3D_EINVAL; }
    *value = g_opt[id].load(std::memory_order_relaxed);
    return X3D_OK;
}

// back to the start-up values (defaults, or what the environment gave)
extern "C" int x3d_reset_options(void) {
    std::call_once(g_opt_once, opt_init);
    opt_init();
    return X3D_OK;
}

extern "C" int x3d_option_count(void) { return X3D_OPT_COUNT; }
extern "C" const char* x3d_option_name(int index) { return (index >= 0 && index < X3D_OPT_COUNT) ? kOpts[index].name : nullptr; }


// ---------------------------------------------------------------------------------------
// Debug aid: fill the LDS of (very likely) every CU with a NaN pattern.  LDS is not cleared between kernels: a kernel that
// consumes an LDS word it did not write sees whatever the previous workgroup on that CU left there.  Interleaved between the
// launches of a step (tests/test_model_gpu.py), this turns such a read into a NaN in a named tensor instead of a
// plausible-looking wrong value.  0x7FC07FC0 is a NaN as fp32, as a bf16 pair and (doubled) as fp64.
// ---------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(1024) void poison_lds_kernel(int words, unsigned* sink) {
    extern __shared__ unsigned lds_poison[];
    for (int i = threadIdx.x; i < words; i += 1024) lds_poison[i] = 0x7FC07FC0u;
    __syncthreads();
    // keep the stores alive and the workgroup resident for a moment so that the grid spreads over all CUs
    unsigned acc = 0;
    for (int rep = 0; rep < 8; ++rep)
        for (int i = threadIdx.x; i < words; i += 1024) acc += lds_poison[(i + rep) % words];
    if (acc == 0x12345u) sink[0] = acc;        // never true for the pattern above
}
}  // namespace

extern "C" int x3d_debug_poison_lds(void* sink, void* stream) {
    X3D_CHECK_ARG(sink != nullptr);
    const int bytes = 160 * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_done = true;
    }
    // one workgroup fills a CU's whole LDS; 4 waves of workgroups per CU: every CU gets at least one with near certainty
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(1024), bytes, (hipStream_t)stream, bytes / 4, (unsigned*)sink);
    X3D_LAUNCH_CHECK();
    return X3D_OK;
}
