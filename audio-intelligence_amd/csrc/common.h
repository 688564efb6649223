// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the AF3/UALM forward pass.
// Wave = 64 lanes everywhere; MFMA 32x32 tiles; f32 accumulation for both storage types.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/afhip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define AFHIP_WAVE 64

// ---- error plumbing: every entry point returns 0 or a negative code; message kept per thread ----
void afhip_set_error(const char* fmt, ...);
#define AFHIP_CHECK(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            afhip_set_error(__VA_ARGS__);                        \
            return AFHIP_ERR_INVALID;                            \
        }                                                        \
    } while (0)
#define AFHIP_LAUNCH_CHECK()                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            afhip_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return AFHIP_ERR_LAUNCH;                                                \
        }                                                                           \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) is PER DEVICE: true the first time the calling site runs on the
// current device of this thread, so a process that drives several GPUs sets the attribute on each of them.
static inline bool afhip_first_use_on_device(unsigned long long* done_mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    return (__atomic_fetch_or(done_mask, bit, __ATOMIC_RELAXED) & bit) == 0;
}
static inline size_t dtype_size(int dt) { return dt == AFHIP_BF16 ? 2 : 4; }

// compute units of the current device (cached per process; 256 on MI355X)
int afhip_cu_count();
// ---- tuning / A-B switches.  Every switch is an entry of ONE table (api.hip): its default, overridden by the environment variable
// AFHIP_<NAME> read ONCE when the library is loaded, or by afhip_set_option() (tests flip a switch inside one process with it).  Launch
// paths read an int from the table: no getenv, no string compare.
enum afhip_opt_id {
    AFHIP_OPT_ATTN_NBUF, AFHIP_OPT_ATTN_LDS_PAD, AFHIP_OPT_ATTN_LAG, AFHIP_OPT_ATTN_ENC64, AFHIP_OPT_ENC64_ONE_BLOCK_PER_WG,
    AFHIP_OPT_DECODE_IMAGED, AFHIP_OPT_DECODE_MERGE, AFHIP_OPT_DECODE_KEY_SPLIT, AFHIP_OPT_DECODE_LEAN, AFHIP_OPT_FP8_MASK, AFHIP_OPT_FP8_FC2,
    AFHIP_OPT_GEMM_SMALL_TILE, AFHIP_OPT_GEMM_GROUP_M, AFHIP_OPT_GEMM_MFMA16, AFHIP_OPT_GEMM_PP,
    AFHIP_OPT_SKINNY_ALDS, AFHIP_OPT_SKINNY_STREAM, AFHIP_OPT_SKINNY_PERSIST, AFHIP_OPT_LOGMEL_DFT, AFHIP_OPT_COUNT
};
int afhip_opt(int id);
// measurement hook of bench.py (afhip_prof_enable / afhip_prof_collect, gemm.hip): HIP events around one launch on its stream
int afhip_prof_begin(hipStream_t s);
void afhip_prof_end(int slot, hipStream_t s, double flops, int tag);
// packed (ragged) encoder batches, norm.hip (C++ linkage: internal to the library)
int afhip_ragged_row_offsets(const int32_t* len, int32_t* row_off, int B, hipStream_t s);
int afhip_ragged_pack_rows(const void* src, void* dst, const int32_t* row_off, const int32_t* len, int B, int T, int row_bytes, hipStream_t s);
int afhip_ragged_avgpool_ln(const void* x, const int32_t* row_off, const int32_t* len, const void* w, const void* b, void* y, int B, int Tout,
                            int D, float eps, int dtype, hipStream_t s);

// ---- element access in f32 regardless of storage type ----
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---- wave-level reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- MFMA "K=16 step" on 32x32 tiles, uniform over storage types ----
// Lane l = (r = l & 31, h = l >> 5) holds 8 consecutive K elements [16*step + 8*h, +8) of row r of the
// A tile (M x K, K contiguous) and of row r of the B^T tile (N x K, K contiguous).
//   bf16: one v_mfma_f32_32x32x16_bf16 (hardware lane map is exactly this one).
//   f32 : eight v_mfma_f32_32x32x2_f32; instruction j consumes element j of both lanes' fragments, so the
//         k index (h, j) is paired consistently on both operands and the 16 products are all summed.
template <typename T> struct Frag8;
template <> struct Frag8<bf16> { typedef bf16x8 type; };
template <> struct Frag8<float> { typedef f32x8 type; };

__device__ __forceinline__ f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma16(f32x8 a, f32x8 b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
    return c;
}
// C/D map of every 32x32 MFMA on gfx950: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// 16-byte vector load/store helpers
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7): one rcp, one exp, five FMAs -- the GEMM epilogue of the
// throughput (bf16) path, where libm's erff would cost as many VALU cycles as the tile's MFMAs.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn expands to a full IEEE division
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(e, x);
}
template <typename T> __device__ __forceinline__ float gelu_act(float x);
template <> __device__ __forceinline__ float gelu_act<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_act<bf16>(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// GELU for the e4m3-operand GEMM epilogues ONLY (no reference counterpart; their contract is the token-level one of tests/test_gpu_config5.py):
// the tanh form x * sigmoid(2 sqrt(2/pi) (x + 0.044715 x^3)), 5 VALU + exp2 + rcp instead of 11 + 2.  |error| <= 5e-4 against the erf form --
// two orders of magnitude below an e4m3 step of the value it feeds, but above a bf16 ulp near zero, so the bf16 path never uses it.
__device__ __forceinline__ float gelu_tanh_fast(float x) {
    const float u = x * x;
    const float t = x * fmaf(u, -0.044715f * 1.5957691216f * 1.4426950409f, -1.5957691216f * 1.4426950409f);   // -2 sqrt(2/pi) (x + c x^3) log2(e)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// a*ca + b*sb with THREE roundings (two products, one sum), never contracted into an FMA: the rotate-half RoPE of
// modeling_qwen2.py:133-134 (q * cos + rotate_half(q) * sin as separate tensor ops).  __fmul_rn / __fadd_rn are plain
// operators in HIP and -ffp-contract=fast fuses them (v_fma_f32 in the f32 kernels), which is 1 ulp away from the reference
// and differs between code sites; every RoPE in the library goes through this helper so they agree bit for bit.
__device__ __forceinline__ float rope_mad(float a, float ca, float b, float sb) {
#pragma clang fp contract(off)
    const float p0 = a * ca;
    const float p1 = b * sb;
    return p0 + p1;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
