// Dynamic per-row e4m3 quantisation of activations for the fp8-operand GEMM (gemm_pp.hip F8 form), with the LayerNorm / RMSNorm
// that precedes the projection fused in: one workgroup per row, the row stays in registers between the statistics, the amax and
// the conversion, so the pass costs one bf16 read and one byte write per element.  No reference counterpart (BASELINE config 5).
#include "common.h"

namespace {

constexpr int QR_THREADS = 256;
constexpr int QR_MAXCH = 10;          // 16-byte chunks (8 bf16) per thread: D <= 256 * 10 * 8 = 20480

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// MODE 0 plain, 1 LayerNorm, 2 RMSNorm
template <int MODE>
__global__ __launch_bounds__(QR_THREADS) void quant_rows_kernel(const bf16* __restrict__ x, long long ld_x, const bf16* __restrict__ w,
                                                                const bf16* __restrict__ b, float eps, unsigned char* __restrict__ q,
                                                                float* __restrict__ scale, int D) {
    __shared__ float sh[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const bf16* row = x + (long long)r * ld_x;
    const int nch = D >> 3;
    float v[QR_MAXCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < QR_MAXCH; ++c) {
        const int ch = tid + c * QR_THREADS;
        if (ch < nch) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(row + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s1 += v[c][e]; s2 = fmaf(v[c][e], v[c][e], s2); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    if constexpr (MODE == 1) {
        const float mean = block_sum(s1, sh) / (float)D;
        float d2 = 0.f;                                   // two-pass variance on the registers: no E[x^2] - mean^2 cancellation
#pragma unroll
        for (int c = 0; c < QR_MAXCH; ++c) {
            const int ch = tid + c * QR_THREADS;
            if (ch < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; d2 = fmaf(d, d, d2); }
            }
        }
        const float rstd = rsqrtf(block_sum(d2, sh) / (float)D + eps);
#pragma unroll
        for (int c = 0; c < QR_MAXCH; ++c) {
            const int ch = tid + c * QR_THREADS;
            if (ch < nch) {
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(w + ch * 8), bb = *reinterpret_cast<const bf16x8*>(b + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = fmaf((v[c][e] - mean) * rstd, (float)g[e], (float)bb[e]);
            }
        }
    } else if constexpr (MODE == 2) {
        const float rstd = rsqrtf(block_sum(s2, sh) / (float)D + eps);
#pragma unroll
        for (int c = 0; c < QR_MAXCH; ++c) {
            const int ch = tid + c * QR_THREADS;
            if (ch < nch) {
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(w + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = (float)g[e] * (float)(bf16)(v[c][e] * rstd);     // modeling_qwen2.py:250-252
            }
        }
    }
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < QR_MAXCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[c][e]));
    am = block_max(am, sh);
    const float sc = am > 0.f ? am / 448.0f : 1.0f;
    const float inv = 1.0f / sc;
    if (tid == 0) scale[r] = sc;
    unsigned char* qrow = q + (long long)r * D;
#pragma unroll
    for (int c = 0; c < QR_MAXCH; ++c) {
        const int ch = tid + c * QR_THREADS;
        if (ch < nch) {
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
            *reinterpret_cast<int2*>(qrow + ch * 8) = make_int2(lo, hi);
        }
    }
}

// Rows of up to 64 * 8 * QW_MAXCH elements (the encoder's 1280, the LLM's 3584): ONE WAVE per row, four rows per workgroup -- no
// LDS, no barriers, every reduction a wave reduction; the workgroup-per-row form above leaves 96 of its 256 lanes idle at D = 1280
// and pays six barriers per row (measured 1.25 TB/s; this form streams).
constexpr int QW_MAXCH = 8;
template <int MODE, int NCH>
__global__ __launch_bounds__(QR_THREADS) void quant_rows_wave_kernel(const bf16* __restrict__ x, long long ld_x, const bf16* __restrict__ w,
                                                                     const bf16* __restrict__ b, float eps, unsigned char* __restrict__ q,
                                                                     float* __restrict__ scale, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (QR_THREADS / 64) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16* row = x + (long long)r * ld_x;
    const int nch = D >> 3;
    float v[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(row + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[c][e] = (float)t[e]; s1 += v[c][e]; s2 = fmaf(v[c][e], v[c][e], s2); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    if constexpr (MODE == 1) {
        const float mean = wave_sum(s1) / (float)D;
        float d2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (lane + c * 64 < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; d2 = fmaf(d, d, d2); }
            }
        }
        const float rstd = rsqrtf(wave_sum(d2) / (float)D + eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(w + ch * 8), bb = *reinterpret_cast<const bf16x8*>(b + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = fmaf((v[c][e] - mean) * rstd, (float)g[e], (float)bb[e]);
            }
        }
    } else if constexpr (MODE == 2) {
        const float rstd = rsqrtf(wave_sum(s2) / (float)D + eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = lane + c * 64;
            if (ch < nch) {
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(w + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = (float)g[e] * (float)(bf16)(v[c][e] * rstd);     // modeling_qwen2.py:250-252
            }
        }
    }
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[c][e]));
    am = wave_max(am);
    const float sc = am > 0.f ? am / 448.0f : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[r] = sc;
    unsigned char* qrow = q + (long long)r * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nch) {
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
            *reinterpret_cast<int2*>(qrow + ch * 8) = make_int2(lo, hi);
        }
    }
}

template <int MODE>
void launch_wave_form(const bf16* x, long long ld_x, const bf16* w, const bf16* b, float eps, unsigned char* q, float* scale, int rows, int D,
                      hipStream_t s) {
    const dim3 grid((rows + QR_THREADS / 64 - 1) / (QR_THREADS / 64)), block(QR_THREADS);
    const int nch = ((D >> 3) + 63) / 64;
    if (nch <= 2) hipLaunchKernelGGL((quant_rows_wave_kernel<MODE, 2>), grid, block, 0, s, x, ld_x, w, b, eps, q, scale, rows, D);
    else if (nch <= 3) hipLaunchKernelGGL((quant_rows_wave_kernel<MODE, 3>), grid, block, 0, s, x, ld_x, w, b, eps, q, scale, rows, D);
    else if (nch <= 5) hipLaunchKernelGGL((quant_rows_wave_kernel<MODE, 5>), grid, block, 0, s, x, ld_x, w, b, eps, q, scale, rows, D);
    else hipLaunchKernelGGL((quant_rows_wave_kernel<MODE, QW_MAXCH>), grid, block, 0, s, x, ld_x, w, b, eps, q, scale, rows, D);
}

}  // namespace

// max |x| of a bf16 buffer (calibration of static activation scales): 16 B per lane per step, wave max, one atomic per wave.  The maximum is
// taken over the BIT PATTERNS of |x| (sign cleared): non-negative floats order like their patterns, and an infinity or a NaN has a larger
// pattern than every finite value, so a non-finite activation comes out as a non-finite maximum instead of being dropped (fmaxf ignores NaN).
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const u32x4* __restrict__ x, long long n16, unsigned* __restrict__ out) {
    unsigned m = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
        const u32x4 v = x[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned lo = (v[e] << 16) & 0x7fffffffu, hi = v[e] & 0x7fff0000u;
            m = lo > m ? lo : m;
            m = hi > m ? hi : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

extern "C" int afhip_absmax_bf16(const void* x, long long n, float* out, void* stream) {
    AFHIP_CHECK(x && out && n > 0 && (n % 8) == 0 && ((uintptr_t)x % 16) == 0, "afhip_absmax_bf16: needs n %% 8 == 0 and a 16-byte aligned buffer");
    const long long n16 = n / 8;
    const int blocks = (int)(n16 / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, n16, (unsigned*)out);
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_quant_rows(const void* x, int ld_x, const void* w, const void* b, float eps, int mode, void* q, float* scale,
                                int rows, int D, void* stream) {
    AFHIP_CHECK(x && q && scale && rows > 0 && D > 0, "afhip_quant_rows: bad args");
    AFHIP_CHECK(mode >= 0 && mode <= 2, "afhip_quant_rows: mode %d", mode);
    AFHIP_CHECK((mode == 0) || (w != nullptr && (mode == 2 || b != nullptr)), "afhip_quant_rows: the norm needs its gain (and LayerNorm its bias)");
    AFHIP_CHECK(D % 8 == 0 && D <= QR_THREADS * QR_MAXCH * 8 && ld_x >= D && ld_x % 8 == 0, "afhip_quant_rows: D=%d must be a multiple of 8, <= %d, rows 16-byte aligned", D, QR_THREADS * QR_MAXCH * 8);
    AFHIP_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 8) == 0, "afhip_quant_rows: x must be 16-byte, q 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (D <= 64 * 8 * QW_MAXCH) {
        if (mode == 1) launch_wave_form<1>((const bf16*)x, ld_x, (const bf16*)w, (const bf16*)b, eps, (unsigned char*)q, scale, rows, D, s);
        else if (mode == 2) launch_wave_form<2>((const bf16*)x, ld_x, (const bf16*)w, (const bf16*)b, eps, (unsigned char*)q, scale, rows, D, s);
        else launch_wave_form<0>((const bf16*)x, ld_x, (const bf16*)w, (const bf16*)b, eps, (unsigned char*)q, scale, rows, D, s);
        AFHIP_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid(rows), block(QR_THREADS);
    if (mode == 1) hipLaunchKernelGGL(quant_rows_kernel<1>, grid, block, 0, s, (const bf16*)x, (long long)ld_x, (const bf16*)w, (const bf16*)b, eps, (unsigned char*)q, scale, D);
    else if (mode == 2) hipLaunchKernelGGL(quant_rows_kernel<2>, grid, block, 0, s, (const bf16*)x, (long long)ld_x, (const bf16*)w, (const bf16*)b, eps, (unsigned char*)q, scale, D);
    else hipLaunchKernelGGL(quant_rows_kernel<0>, grid, block, 0, s, (const bf16*)x, (long long)ld_x, (const bf16*)w, (const bf16*)b, eps, (unsigned char*)q, scale, D);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
