// Row-wise and element-wise kernels (HBM-bound): LayerNorm, AvgPool+LayerNorm, RMSNorm, embedding sum,
// RoPE + KV-cache append, layout transpose.  One wave per row, 16-byte (bf16) / 32-byte (f32) per-lane
// vectors, f32 statistics via wave shuffles; no LDS except the transpose tile.
#include "common.h"

namespace {

constexpr int MAXC = 8;  // 8-element chunks per lane: rows up to 64*8*8 = 4096 elements

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = x;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// MODE 0: LayerNorm(x[row]); MODE 1: LayerNorm(round_T((x[2row] + x[2row+1]) / 2)) within a clip of Tout rows.
// MODE 2: MODE 1 on PACKED clips (encoder.hip ragged forward): output row (b, t) = row / Tout, row % Tout pools x rows
// row_off[b] + 2t, + 2t + 1 when t < (len[b] - 2) / 2 + 1 (the rows the reference keeps, audio.py:1163-1187), zeros otherwise.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ b, T* __restrict__ y, int rows, int D,
                                                        float eps, const int* __restrict__ row_off = nullptr,
                                                        const int* __restrict__ len = nullptr, int Tout = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    long long src = 2 * (long long)row;
    if (MODE == 2) {
        const int bi = row / Tout, t = row - bi * Tout, L = len[bi];
        const int n_out = L >= 2 ? (L - 2) / 2 + 1 : 0;
        if (t >= n_out) {                                   // wave-uniform: a wave owns one row
            float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int c = lane; c < nch; c += 64) store8<T>(y + (long long)row * D + c * 8, z);
            return;
        }
        src = (long long)row_off[bi] + 2 * t;
    }
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            if (MODE == 0) {
                load8<T>(x + (long long)row * D + c * 8, v[i]);
            } else {
                float a[8], bb[8];
                load8<T>(x + src * D + c * 8, a);
                load8<T>(x + (src + 1) * D + c * 8, bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = to_f32<T>(from_f32<T>((a[e] + bb[e]) * 0.5f));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float ww[8], bv[8], o[8];
            load8<T>(w + c * 8, ww);
            load8<T>(b + c * 8, bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * ww[e] + bv[e];
            store8<T>(y + (long long)row * D + c * 8, o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y,
                                                      int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            load8<T>(x + (long long)row * D + c * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e] * v[i][e];
        }
    }
    const float r = rsqrtf(wave_sum(s) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float ww[8], o[8];
            load8<T>(w + c * 8, ww);
            // modeling_qwen2.py:247-252: weight * (x * rsqrt(var + eps)).to(input_dtype)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ww[e] * to_f32<T>(from_f32<T>(v[i][e] * r));
            store8<T>(y + (long long)row * D + c * 8, o);
        }
    }
}

// out[tok] = sum_s table[ids[tok, s]] (f32 accumulate); one workgroup per token
template <typename T>
__global__ __launch_bounds__(256) void embed_sum_kernel(const int64_t* __restrict__ ids, const T* __restrict__ table,
                                                        T* __restrict__ out, int S, int H, int vocab) {
    const int tok = blockIdx.x;
    for (int c = threadIdx.x; c < (H >> 3); c += blockDim.x) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < S; ++s) {
            long long id = ids[(long long)tok * S + s];
            id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
            float v[8];
            load8<T>(table + id * H + c * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
        store8<T>(out + (long long)tok * H + c * 8, acc);
    }
}

// RoPE (rotate-half) on q and k of the fused qkv rows, in place for q; k (rotated) and v go to the cache.
// grid: (B*T, n_q + 2*n_kv); block: hd/2 threads (>= 32)
template <typename T>
__global__ void rope_kv_kernel(T* __restrict__ qkv, int ld, const float* __restrict__ cosT, const float* __restrict__ sinT,
                               T* __restrict__ kc, T* __restrict__ vc, int T_, int n_q, int n_kv, int hd, int cap, int pos0) {
    const int row = blockIdx.x, head = blockIdx.y;
    const int b = row / T_, t = row - b * T_;
    const int pos = pos0 + t;
    const int half = hd >> 1;
    const int i = threadIdx.x;
    if (i >= half) return;
    T* src = qkv + (long long)row * ld + (long long)head * hd;
    if (head < n_q + n_kv) {
        const float c = cosT[(long long)pos * half + i], s = sinT[(long long)pos * half + i];
        const float x1 = to_f32<T>(src[i]), x2 = to_f32<T>(src[i + half]);
        // q*cos + rotate_half(q)*sin with separate roundings (modeling_qwen2.py:133-134)
        const float o1 = rope_mad(x1, c, -x2, s);
        const float o2 = rope_mad(x2, c, x1, s);
        if (head < n_q) {
            src[i] = from_f32<T>(o1);
            src[i + half] = from_f32<T>(o2);
        } else {
            T* dst = kc + (((long long)b * n_kv + (head - n_q)) * cap + pos) * hd;
            dst[i] = from_f32<T>(o1);
            dst[i + half] = from_f32<T>(o2);
        }
    } else {
        T* dst = vc + (((long long)b * n_kv + (head - n_q - n_kv)) * cap + pos) * hd;
        dst[i] = src[i];
        dst[i + half] = src[i + half];
    }
}

// y[b, c, r] = cast(x[b, r, c]); 32x32 tiles through LDS
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, int R, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < R && c < C) tile[j][tx] = to_f32<TI>(x[((long long)b * R + r) * C + c]);
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < R && c < C) y[((long long)b * C + c) * R + r] = from_f32<TO>(tile[tx][j]);
    }
}

}  // namespace

#define DISPATCH_T(dt, EXPR_BF16, EXPR_F32) \
    do { if ((dt) == AFHIP_BF16) { EXPR_BF16; } else { EXPR_F32; } } while (0)

static int check_row(const char* who, int rows, int D, int dtype) {
    AFHIP_CHECK(dtype == AFHIP_F32 || dtype == AFHIP_BF16, "%s: bad dtype %d", who, dtype);
    AFHIP_CHECK(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * MAXC, "%s: rows=%d D=%d unsupported (D%%8==0, D<=%d)", who, rows, D, 64 * 8 * MAXC);
    return 0;
}

extern "C" int afhip_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps, int dtype, void* stream) {
    if (int e = check_row("afhip_layernorm", rows, D, dtype)) return e;
    AFHIP_CHECK(x && w && b && y, "afhip_layernorm: null pointer");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((layernorm_kernel<bf16, 0>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, rows, D, eps),
               hipLaunchKernelGGL((layernorm_kernel<float, 0>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float*)x, (const float*)w, (const float*)b, (float*)y, rows, D, eps));
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_avgpool_ln(const void* x, const void* w, const void* b, void* y, int B, int Tout, int D, float eps, int dtype, void* stream) {
    AFHIP_CHECK(B > 0 && Tout > 0, "afhip_avgpool_ln: bad B=%d Tout=%d", B, Tout);
    const int rows = B * Tout;  // row r of y pools x rows 2r, 2r+1 (clips are 2*Tout rows, so pairs never straddle clips)
    if (int e = check_row("afhip_avgpool_ln", rows, D, dtype)) return e;
    AFHIP_CHECK(x && w && b && y, "afhip_avgpool_ln: null pointer");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((layernorm_kernel<bf16, 1>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, rows, D, eps),
               hipLaunchKernelGGL((layernorm_kernel<float, 1>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float*)x, (const float*)w, (const float*)b, (float*)y, rows, D, eps));
    AFHIP_LAUNCH_CHECK();
    return 0;
}

// ---- packed (ragged) encoder batches: helpers of encoder.hip's afhip_encoder_forward_ragged (C++ linkage, not part of the ABI) ----
namespace {
// row_off[b] = sum of len[0..b): B is small (one wave scans it)
__global__ void row_offsets_kernel(const int* __restrict__ len, int* __restrict__ row_off, int B) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < B; ++i) { row_off[i] = acc; acc += len[i]; }
        row_off[B] = acc;
    }
}
// dst[row_off[b] + t] = src[b, t] for t < len[b]: one wave per source row, 16 B per lane
__global__ __launch_bounds__(256) void pack_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, const int* __restrict__ row_off,
                                                        const int* __restrict__ len, int B, int T, int row_bytes) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (long long)B * T) return;
    const int b = (int)(r / T), t = (int)(r - (long long)b * T);
    if (t >= len[b]) return;
    const char* s = src + r * row_bytes;
    char* d = dst + ((long long)row_off[b] + t) * row_bytes;
    for (int c = lane * 16; c < row_bytes; c += 64 * 16) *reinterpret_cast<u32x4*>(d + c) = *reinterpret_cast<const u32x4*>(s + c);
}
}  // namespace

int afhip_ragged_row_offsets(const int32_t* len, int32_t* row_off, int B, hipStream_t s) {
    hipLaunchKernelGGL(row_offsets_kernel, dim3(1), dim3(64), 0, s, len, row_off, B);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
int afhip_ragged_pack_rows(const void* src, void* dst, const int32_t* row_off, const int32_t* len, int B, int T, int row_bytes, hipStream_t s) {
    AFHIP_CHECK(row_bytes % 16 == 0, "ragged pack: rows must be multiples of 16 bytes");
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)cdiv(B * T, 4)), dim3(256), 0, s, (const char*)src, (char*)dst, row_off, len, B, T, row_bytes);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
int afhip_ragged_avgpool_ln(const void* x, const int32_t* row_off, const int32_t* len, const void* w, const void* b, void* y, int B, int Tout,
                            int D, float eps, int dtype, hipStream_t s) {
    const int rows = B * Tout;
    if (int e = check_row("afhip_ragged_avgpool_ln", rows, D, dtype)) return e;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((layernorm_kernel<bf16, 2>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, rows, D, eps, row_off, len, Tout),
               hipLaunchKernelGGL((layernorm_kernel<float, 2>), dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float*)x, (const float*)w, (const float*)b, (float*)y, rows, D, eps, row_off, len, Tout));
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_rmsnorm(const void* x, const void* w, void* y, int rows, int D, float eps, int dtype, void* stream) {
    if (int e = check_row("afhip_rmsnorm", rows, D, dtype)) return e;
    AFHIP_CHECK(x && w && y, "afhip_rmsnorm: null pointer");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(rmsnorm_kernel<bf16>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const bf16*)x, (const bf16*)w, (bf16*)y, rows, D, eps),
               hipLaunchKernelGGL(rmsnorm_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float*)x, (const float*)w, (float*)y, rows, D, eps));
    AFHIP_LAUNCH_CHECK();
    return 0;
}


// ---- LayerNorm statistics for the GEMM-folded form (afhip.h: afhip_gemm_args.ln_stats) ----
namespace {
__global__ __launch_bounds__(64) void ln_stats_finalize_kernel(const float* __restrict__ part, int P, int rows, float inv_d, float eps,
                                                               float* __restrict__ stats) {
    // one wave per 64 rows (750 workgroups for the encoder's 48000 rows: every CU gets some), partial loads unrolled by 4 so
    // several are in flight per lane
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= rows) return;
    float s = 0.f, ss = 0.f;
    int p = 0;
    for (; p + 4 <= P; p += 4) {
        f32x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x2*>(part + 2 * ((long long)(p + u) * rows + r));
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += v[u][0]; ss += v[u][1]; }
    }
    for (; p < P; ++p) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(part + 2 * ((long long)p * rows + r));
        s += v[0]; ss += v[1];
    }
    const float mean = s * inv_d;
    const float var = fmaxf(ss * inv_d - mean * mean, 0.f);
    *reinterpret_cast<f32x2*>(stats + 2 * (long long)r) = f32x2{mean, rsqrtf(var + eps)};
}

// one wave per row, two-pass (mean, then centred sum of squares) like layernorm_kernel, no output row
template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, int rows, int D, float eps, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = D >> 3;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            load8<T>(x + (long long)row * D + c * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) *reinterpret_cast<f32x2*>(stats + 2 * (long long)row) = f32x2{mean, rstd};
}
}  // namespace

extern "C" int afhip_ln_stats_finalize(const float* partials, int P, int rows, int D, float eps, float* stats, void* stream) {
    AFHIP_CHECK(partials && stats && P > 0 && rows > 0 && D > 0, "afhip_ln_stats_finalize: bad args P=%d rows=%d D=%d", P, rows, D);
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream, partials, P, rows, 1.0f / (float)D, eps, stats);
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_row_stats(const void* x, int rows, int D, float eps, int dtype, float* stats, void* stream) {
    if (int e = check_row("afhip_row_stats", rows, D, dtype)) return e;
    AFHIP_CHECK(x && stats, "afhip_row_stats: null pointer");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(row_stats_kernel<bf16>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const bf16*)x, rows, D, eps, stats),
               hipLaunchKernelGGL(row_stats_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, s, (const float*)x, rows, D, eps, stats));
    AFHIP_LAUNCH_CHECK();
    return 0;
}


// Row gather of the AF3 / Qwen2-Audio placeholder merge (modeling_whisper.py:1056-1104): out row r is text row plan[r]
// (plan >= 0), audio row -(plan[r] + 2) (plan <= -2) or zeros (plan == -1, padding).  One wave per row, 16 B per lane.
namespace {
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ text, const char* __restrict__ audio,
                                                          const int32_t* __restrict__ plan, char* __restrict__ out,
                                                          int n_rows, int row_bytes) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const int pl = plan[row];
    const char* src = pl >= 0 ? text + (long long)pl * row_bytes : (pl <= -2 ? audio + (long long)(-(pl + 2)) * row_bytes : nullptr);
    char* dst = out + (long long)row * row_bytes;
    for (int off = lane * 16; off < row_bytes; off += 64 * 16) {
        const u32x4 v = src ? ld16(src + off) : u32x4{0u, 0u, 0u, 0u};
        st16(dst + off, v);
    }
}
}  // namespace

extern "C" int afhip_gather_rows(const void* text_rows, const void* audio_rows, const int32_t* plan, void* out, int n_rows,
                                 int n_text_rows, int n_audio_rows, int row_bytes, void* stream) {
    AFHIP_CHECK(plan && out && n_rows > 0, "afhip_gather_rows: bad args n_rows=%d", n_rows);
    AFHIP_CHECK(row_bytes > 0 && row_bytes % 16 == 0, "afhip_gather_rows: row_bytes=%d must be a positive multiple of 16", row_bytes);
    AFHIP_CHECK((n_text_rows == 0 || text_rows) && (n_audio_rows == 0 || audio_rows), "afhip_gather_rows: null source with rows to read");
    AFHIP_CHECK(((uintptr_t)out % 16) == 0 && ((uintptr_t)text_rows % 16) == 0 && ((uintptr_t)audio_rows % 16) == 0, "afhip_gather_rows: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, (const char*)text_rows,
                       (const char*)audio_rows, plan, (char*)out, n_rows, row_bytes);
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_embed_sum(const int64_t* ids, const void* table, void* out, int n_tok, int S, int H, int vocab, int dtype, void* stream) {
    AFHIP_CHECK(dtype == AFHIP_F32 || dtype == AFHIP_BF16, "afhip_embed_sum: bad dtype %d", dtype);
    AFHIP_CHECK(ids && table && out && n_tok > 0 && S > 0 && H > 0 && H % 8 == 0 && vocab > 0, "afhip_embed_sum: bad args n_tok=%d S=%d H=%d vocab=%d", n_tok, S, H, vocab);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(embed_sum_kernel<bf16>, dim3(n_tok), dim3(256), 0, s, ids, (const bf16*)table, (bf16*)out, S, H, vocab),
               hipLaunchKernelGGL(embed_sum_kernel<float>, dim3(n_tok), dim3(256), 0, s, ids, (const float*)table, (float*)out, S, H, vocab));
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_rope_kv(void* qkv, int ld_qkv, const float* cos_table, const float* sin_table, int pos0,
                             void* k_cache, void* v_cache, int B, int T, int n_q, int n_kv, int hd, int cache_cap,
                             int rope_max_pos, int dtype, void* stream) {
    AFHIP_CHECK(dtype == AFHIP_F32 || dtype == AFHIP_BF16, "afhip_rope_kv: bad dtype %d", dtype);
    AFHIP_CHECK(qkv && cos_table && sin_table && k_cache && v_cache, "afhip_rope_kv: null pointer");
    AFHIP_CHECK(B > 0 && T > 0 && n_q > 0 && n_kv > 0 && hd >= 2 && hd % 2 == 0 && hd / 2 <= 1024, "afhip_rope_kv: bad shape");
    AFHIP_CHECK(ld_qkv >= (n_q + 2 * n_kv) * hd, "afhip_rope_kv: ld_qkv=%d too small", ld_qkv);
    AFHIP_CHECK(pos0 >= 0 && pos0 + T <= cache_cap, "afhip_rope_kv: positions [%d,%d) exceed cache capacity %d", pos0, pos0 + T, cache_cap);
    AFHIP_CHECK(pos0 + T <= rope_max_pos, "afhip_rope_kv: positions [%d,%d) exceed rope table %d", pos0, pos0 + T, rope_max_pos);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(B * T, n_q + 2 * n_kv), block(hd / 2 < 64 ? 64 : hd / 2);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(rope_kv_kernel<bf16>, grid, block, 0, s, (bf16*)qkv, ld_qkv, cos_table, sin_table, (bf16*)k_cache, (bf16*)v_cache, T, n_q, n_kv, hd, cache_cap, pos0),
               hipLaunchKernelGGL(rope_kv_kernel<float>, grid, block, 0, s, (float*)qkv, ld_qkv, cos_table, sin_table, (float*)k_cache, (float*)v_cache, T, n_q, n_kv, hd, cache_cap, pos0));
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_transpose_cast(const void* x, void* y, int B, int R, int C, int in_dtype, int out_dtype, void* stream) {
    AFHIP_CHECK(x && y && B > 0 && R > 0 && C > 0, "afhip_transpose_cast: bad args");
    AFHIP_CHECK((in_dtype == AFHIP_F32 || in_dtype == AFHIP_BF16) && (out_dtype == AFHIP_F32 || out_dtype == AFHIP_BF16), "afhip_transpose_cast: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(cdiv(C, 32), cdiv(R, 32), B), block(256);
    if (in_dtype == AFHIP_F32 && out_dtype == AFHIP_F32)
        hipLaunchKernelGGL((transpose_cast_kernel<float, float>), grid, block, 0, s, (const float*)x, (float*)y, R, C);
    else if (in_dtype == AFHIP_F32)
        hipLaunchKernelGGL((transpose_cast_kernel<float, bf16>), grid, block, 0, s, (const float*)x, (bf16*)y, R, C);
    else if (out_dtype == AFHIP_F32)
        hipLaunchKernelGGL((transpose_cast_kernel<bf16, float>), grid, block, 0, s, (const bf16*)x, (float*)y, R, C);
    else
        hipLaunchKernelGGL((transpose_cast_kernel<bf16, bf16>), grid, block, 0, s, (const bf16*)x, (bf16*)y, R, C);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
