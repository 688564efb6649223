// Version + error plumbing of the C ABI (include/afhip.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void afhip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int afhip_version(void) { return 100; }
extern "C" const char* afhip_last_error(void) { return g_err; }

int afhip_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// ---- the switch table (common.h: afhip_opt_id) -----------------------------------------------------------------------------
namespace {
struct OptDef { const char* name; int def; };
// -1 = "the kernel's own default" where two kernels read one switch with different defaults (SKINNY_ALDS)
const OptDef kOpts[AFHIP_OPT_COUNT] = {
    {"ATTN_NBUF", 2},               // LDS stages of attn_kernel (1 or 2)
    {"ATTN_LDS_PAD", 0},            // extra dynamic LDS bytes of attn_kernel (occupancy experiment)
    {"ATTN_LAG", 1},                // lagged-maximum softmax of the generic bf16 hd-64 prescaled form
    {"ATTN_ENC64", 1},              // the one-wave-per-SIMD encoder attention (attention_enc.hip)
    {"ENC64_ONE_BLOCK_PER_WG", 0},  // encoder attention: one query block per workgroup instead of the persistent walk
    {"DECODE_IMAGED", 1},           // decode step GEMMs: 1 = persistent imaged phases (decode_phases.hip), 0 = the round-3 launches over row-major activations
    {"DECODE_MERGE", 0},            // split-context decode attention: merge by the last-arriving workgroup inside the launch
    {"DECODE_KEY_SPLIT", 0},        // keys per workgroup of the split-context decode attention (0 = the built-in 128)
    {"DECODE_LEAN", 1},             // decode attention at head_dim 128 / bf16: the lean kernel (attn_decode128_kernel); 0 = the generic kernel
    {"FP8_MASK", 6},                // e4m3 encoder mode: which projections take e4m3 operands (bit 0 qkv, 1 out, 2 fc1, 3 fc2)
    {"FP8_FC2", 0},                 // ... fc2 as well
    {"GEMM_SMALL_TILE", 0},
    {"GEMM_GROUP_M", 4},            // tile rasterisation of the 256 x 256 GEMMs
    {"GEMM_MFMA16", 0},
    {"GEMM_PP", 1},                 // the persistent ping-pong GEMM (0 = the one-barrier 256 x 256 kernel)
    {"SKINNY_ALDS", -1},            // activations staged in LDS by skinny_kernel (bf16 default off, e4m3 default on)
    {"SKINNY_STREAM", 1},           // the persistent form of the bf16 decode GEMMs (gemm_stream.hip)
    {"SKINNY_PERSIST", 1},          // the persistent SwiGLU pair form of skinny_kernel
    {"LOGMEL_DFT", 0},              // the folded-DFT MFMA form of the log-mel kernel
};
int g_opt[AFHIP_OPT_COUNT];
struct OptInit {
    OptInit() {
        for (int i = 0; i < AFHIP_OPT_COUNT; ++i) {
            char key[64];
            snprintf(key, sizeof(key), "AFHIP_%s", kOpts[i].name);
            const char* e = getenv(key);
            g_opt[i] = (e && e[0]) ? atoi(e) : kOpts[i].def;
        }
    }
} g_opt_init;       // runs when the shared library is loaded
}  // namespace

int afhip_opt(int id) { return g_opt[id]; }

extern "C" int afhip_set_option(const char* name, int value) {
    AFHIP_CHECK(name != nullptr, "afhip_set_option: null name");
    for (int i = 0; i < AFHIP_OPT_COUNT; ++i)
        if (strcmp(name, kOpts[i].name) == 0) { g_opt[i] = value; return 0; }
    afhip_set_error("afhip_set_option: unknown option '%s'", name);
    return AFHIP_ERR_INVALID;
}
