// Decode GEMM with FP8 (OCP e4m3) weights and bf16 activations (W8A16): C[M,N] = A'[M,K] . (s_n * Wq[N,K])^T.
// The decode step is bound by the bytes of W it must stream (gemm_skinny.hip); storing W as e4m3 with one f32 scale per
// output row halves them.  Same structure as the bf16 kernel -- weights straight from HBM into registers, 8 waves
// interleaved over K steps, LDS combine -- with a K step of 128 so a lane still reads 32 contiguous bytes of its weight
// row (4 lanes = one 128-byte line); the 32 weights are widened to bf16 in registers (v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32)
// and fed to four v_mfma_f32_16x16x32_bf16 against the lane's 32 activations.  The row scale is applied to the f32
// accumulator in the epilogue, next to the fused RMSNorm row scale / SwiGLU epilogue of the bf16 kernel.
// Reference counterpart: none (BASELINE config 5 "fp8 MFMA GEMMs"); checked against the dequantised weights in f32.
#include "common.h"
#include <stdlib.h>

#ifndef SKINNY_DEPTH
#define SKINNY_DEPTH 2      /* K steps in flight per wave, NT >= 2 (3 costs the second workgroup per CU: measured slower) */
#endif
#ifndef SKINNY_DEPTH1
#define SKINNY_DEPTH1 2     /* the same for the narrow NT = 1 tiles */
#endif

namespace {

enum { A_PLAIN = 0, A_RMSNORM = 1 };

struct SkinnyF8P {
    const char* A;
    const char* W;        // e4m3 bytes [N, ldw]
    const float* wscale;  // [N]
    const char* bias;
    const char* res;
    const char* norm_w;
    char* C;
    int M, N, K;
    long long lda, ldw, ldc, ldres;
    int out_f32;
    float norm_eps;
    int swiglu_out;
    int a_rows;      // ALDS: rows of the activation image kept in LDS (8 or 16, >= M)
    int tile_rows;   // weight rows per 16-wide MFMA tile that carry work (<= 16): narrow outputs are cut into ceil(N / CUs)-row shares so every CU streams the same bytes
};

constexpr int KS8 = 128;

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 4 fp8 bytes -> 4 bf16 packed in two dwords: v_cvt_scalef32_pk_bf16_fp8 widens two e4m3 values per instruction (exact)
__device__ __forceinline__ void fp8x4_to_bf16x4(uint32_t v, uint32_t& lo, uint32_t& hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)v, 1.0f, false);
    const bf16x2_t p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)v, 1.0f, true);
    lo = __builtin_bit_cast(uint32_t, p0);
    hi = __builtin_bit_cast(uint32_t, p1);
}

template <int NT, int MT> struct StepRegs8 { u32x4 w[NT][2]; u32x4 a[MT][4]; u32x4 n[4]; };

// ALDS: activations staged once per workgroup into LDS in MFMA A-operand order, RMSNorm gain applied and squares summed on the
// way (gemm_skinny.hip): with e4m3 weights the activation + gain fragments were 8 of the 8 + 2 NT vector loads of a K step.
template <int NT, int MT, int AMODE, bool ALDS = false>
__global__ __launch_bounds__(512) void skinny_fp8_kernel(SkinnyF8P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);                 // [8 waves][NT][MT][64 lanes][4]
    float* red_ss = red + 8 * NT * MT * 256;                      // [8 waves][MT][16]
    char* aimg = reinterpret_cast<char*>(red_ss + 8 * MT * 16);   // ALDS: [K / 128][4][4][a_rows] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    // SwiGLU epilogue (NT == 2): workgroup b owns gate rows [64 j + 16 t, +16) and the matching up rows 32 further
    // (j = b >> 1, t = b & 1) of the 32-row interleaved gate/up weight: 16-row granularity keeps the per-CU byte share even
    const bool pair = (NT == 2) && p.swiglu_out;
    const int TR = p.tile_rows;                        // pair mode: gate rows (= up rows) per workgroup, <= 16
    const int g0 = (int)blockIdx.x * TR;               // pair mode: first gate index of this workgroup, gate g = W row 64 (g >> 5) + (g & 31)
    const int n_base = pair ? 0 : (int)blockIdx.x * (NT * TR);
    const int cr = c16 < TR ? c16 : TR - 1;           // lanes past the share re-read its last row (same line: no extra traffic), their results are dropped

    const char* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int n;
        if (pair) {
            int g = g0 + cr;
            g = g < (p.N >> 1) ? g : (p.N >> 1) - 1;
            n = ((g >> 5) << 6) + (g & 31) + t * 32;
        } else {
            n = n_base + t * TR + cr;
            n = n < p.N ? n : p.N - 1;
        }
        wrow[t] = p.W + (long long)n * p.ldw;
    }
    const char* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c16;
        m = m < p.M ? m : p.M - 1;
        arow[t] = p.A + (long long)m * p.lda * 2;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss[MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) ss[b] = 0.f;

    const int nsteps = p.K / KS8;
    const int RM = p.a_rows;
    auto stage_a = [&]() {
        if constexpr (ALDS) {
            static_assert(MT == 1, "ALDS: one row tile");
            // 16-B chunk cc of a row covers k = 8 cc ..: step cc >> 4, q = (cc >> 2) & 3, chunk-in-lane cc & 3
            for (int m = wave; m < RM; m += 8) {
                const bool real = m < p.M;
                const char* src = p.A + (long long)m * p.lda * 2;
                float sq = 0.f;
                for (int cc = lane; cc < p.K / 8; cc += 64) {
                    u32x4 v = real ? ld16(src + cc * 16) : u32x4{0u, 0u, 0u, 0u};
                    if constexpr (AMODE == A_RMSNORM) {
                        const u32x4 g = ld16(p.norm_w + cc * 16);
                        u32x4 o;
    #pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const float x0 = bf16_lo(v[d]), x1 = bf16_hi(v[d]);
                            sq += x0 * x0 + x1 * x1;
                            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                            bf16x2_t pr;
                            pr[0] = (bf16)(x0 * bf16_lo(g[d]));
                            pr[1] = (bf16)(x1 * bf16_hi(g[d]));
                            o[d] = __builtin_bit_cast(uint32_t, pr);
                        }
                        v = o;
                    }
                    st16(aimg + (cc >> 4) * (RM * 256) + (cc & 3) * (RM * 64) + ((((cc >> 2) & 3) * RM + m) << 4), v);
                }
                if constexpr (AMODE == A_RMSNORM) {
                    sq = wave_sum(sq);
                    if (lane == 0) red_ss[m] = sq;
                }
            }
            __syncthreads();
        }
    };
    const int a_lane = (q * RM + (c16 & (RM - 1))) << 4;
    auto issue = [&](int s, StepRegs8<NT, MT>& r) {
        const long long k0 = (long long)s * KS8 + q * 32;          // first of this lane's 32 K elements
#pragma unroll
        for (int t = 0; t < NT; ++t) { r.w[t][0] = ld16(wrow[t] + k0); r.w[t][1] = ld16(wrow[t] + k0 + 16); }
        if constexpr (ALDS) {
            // read from LDS at consume time
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) r.a[t][c] = ld16(arow[t] + k0 * 2 + c * 16);
            if constexpr (AMODE == A_RMSNORM) {
#pragma unroll
                for (int c = 0; c < 4; ++c) r.n[c] = ld16(p.norm_w + k0 * 2 + c * 16);
            }
        }
    };
    auto consume = [&](StepRegs8<NT, MT>& r, int s) {
        if constexpr (ALDS) {
#pragma unroll
            for (int c = 0; c < 4; ++c) r.a[0][c] = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 256) + c * (RM * 64) + a_lane);
        }
        if constexpr (AMODE == A_RMSNORM && !ALDS) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float x0 = bf16_lo(r.a[mt][c][d]), x1 = bf16_hi(r.a[mt][c][d]);
                        ss[mt] += x0 * x0 + x1 * x1;
                        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                        bf16x2_t pr;
                        pr[0] = (bf16)(x0 * bf16_lo(r.n[c][d]));
                        pr[1] = (bf16)(x1 * bf16_hi(r.n[c][d]));
                        o[d] = __builtin_bit_cast(uint32_t, pr);
                    }
                    r.a[mt][c] = o;
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // widen this tile's 32 fp8 weights: dword d of half h holds k = 16 h + 4 d .. +4 -> bf16 chunk c = 2 h + (d >> 1)
            u32x4 wb[4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    uint32_t lo, hi;
                    fp8x4_to_bf16x4(r.w[nt][h][d], lo, hi);
                    wb[2 * h + (d >> 1)][2 * (d & 1)] = lo;
                    wb[2 * h + (d >> 1)][2 * (d & 1) + 1] = hi;
                }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r.a[mt][c]), __builtin_bit_cast(bf16x8, wb[c]), acc[nt][mt], 0, 0, 0);
        }
    };

    // rolling window of two steps per wave (see gemm_skinny.hip): a register set is re-issued as soon as it is consumed
    {
        constexpr int DEPTH = (NT * MT >= 8) ? 1 : (NT == 1 ? SKINNY_DEPTH1 : SKINNY_DEPTH);   // NT * MT = 8 tiles: one step set in flight (two spill)   // register sets = K steps in flight per wave
        StepRegs8<NT, MT> r[DEPTH];
        int sx[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            sx[d] = wave + 8 * d;
            if (sx[d] < nsteps) issue(sx[d], r[d]);
        }
        stage_a();      // ALDS: the activation image is built while the first weight fragments are already in flight
        bool more = sx[0] < nsteps;
        while (more) {
            more = false;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (sx[d] < nsteps) {
                    consume(r[d], sx[d]);
                    sx[d] += 8 * DEPTH;
                    if (sx[d] < nsteps) { issue(sx[d], r[d]); more = true; }
                }
            }
        }
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(red + ((((wave * NT + nt) * MT + mt) * 64 + lane) << 2)) = acc[nt][mt];
    if constexpr (AMODE == A_RMSNORM && !ALDS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v = ss[mt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (q == 0) red_ss[(wave * MT + mt) * 16 + c16] = v;
        }
    }
    __syncthreads();
    auto row_scale = [&](int mt, int mrow) {
        if constexpr (AMODE == A_RMSNORM) {
            float sq = 0.f;
            if constexpr (ALDS) sq = red_ss[mrow];
            else {
#pragma unroll
                for (int w = 0; w < 8; ++w) sq += red_ss[(w * MT + mt) * 16 + mrow];
            }
            return rsqrtf(sq / (float)p.K + p.norm_eps);
        } else {
            return 1.0f;
        }
    };
    if constexpr (NT == 4 || NT == 2) {
        if (p.swiglu_out) {
            constexpr int NG = NT / 2;
            for (int o = tid; o < NG * MT * 256; o += 512) {
                const int reg = o & 3, ln = (o >> 2) & 63, tile = o >> 8;
                const int mt = tile % MT, nt = tile / MT;
                float g = 0.f, u = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    g += red[((((w * NT + nt) * MT + mt) * 64 + ln) << 2) + reg];
                    u += red[((((w * NT + nt + NG) * MT + mt) * 64 + ln) << 2) + reg];
                }
                const int mrow = 4 * (ln >> 4) + reg, m = mt * 16 + mrow;
                const int gi = g0 + (ln & 15);                                   // pair form: gate index
                const int ng = pair ? ((gi >> 5) << 6) + (gi & 31) : n_base + nt * 16 + (ln & 15);   // gate row index inside W
                const bool live = pair ? ((ln & 15) < TR && gi < (p.N >> 1)) : (ng + 32 < p.N + 1);
                if (live && m < p.M) {
                    const float r = row_scale(mt, mrow);
                    g *= r * p.wscale[ng];
                    u *= r * p.wscale[ng + 32];
                    reinterpret_cast<bf16*>(p.C)[(long long)m * p.ldc + ((ng >> 6) << 5) + (ng & 31)] = (bf16)(silu(g) * u);
                }
            }
            return;
        }
    }
    for (int o = tid; o < NT * MT * 256; o += 512) {
        const int reg = o & 3, ln = (o >> 2) & 63, tile = o >> 8;
        const int mt = tile % MT, nt = tile / MT;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[((((w * NT + nt) * MT + mt) * 64 + ln) << 2) + reg];
        const int n = n_base + nt * TR + (ln & 15);
        const int mrow = 4 * (ln >> 4) + reg, m = mt * 16 + mrow;
        if ((ln & 15) < TR && n < p.N && m < p.M) {
            v *= row_scale(mt, mrow) * p.wscale[n];
            if (p.bias) v += (float)reinterpret_cast<const bf16*>(p.bias)[n];
            if (p.res) v += (float)reinterpret_cast<const bf16*>(p.res)[(long long)m * p.ldres + n];
            if (p.out_f32) reinterpret_cast<float*>(p.C)[(long long)m * p.ldc + n] = v;
            else reinterpret_cast<bf16*>(p.C)[(long long)m * p.ldc + n] = (bf16)v;
        }
    }
}

// ---- persistent form of the SwiGLU pair GEMM with e4m3 weights (decode gate/up, M <= 16): see skinny_pair_persist_kernel in
// gemm_skinny.hip.  A K step is 128 deep here, so K = 3584 gives the eight waves 4 or 3 steps per unit: every wave walks its own
// step count per unit and all of them meet at the unit's combine barrier.  Same K order per row, same K-slice sum order: same bits.
template <int AMODE, int DEPTH>
__global__ __launch_bounds__(512) void skinny_fp8_pair_persist_kernel(SkinnyF8P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);                 // [2 unit parities][8 waves][2 tiles][64 lanes][4]
    float* red_ss = red + 2 * 8 * 2 * 256;                        // [16]
    char* aimg = reinterpret_cast<char*>(red_ss + 16);            // [K / 128][4][4][a_rows] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int TR = p.tile_rows, gates = p.N >> 1, RM = p.a_rows;
    const int n_units = (gates + TR - 1) / TR;
    const int nsteps = p.K / KS8;
    const int spw = (nsteps - wave + 7) / 8;                      // this wave's K steps per unit (>= 1: the host checks K >= 1024)
    const int my_units = (n_units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * spw;
    const int cr = c16 < TR ? c16 : TR - 1;
    const char* wrow0 = nullptr;
    const char* wrow1 = nullptr;
    auto set_rows = [&](int ui) {
        int g = ((int)blockIdx.x + ui * (int)gridDim.x) * TR + cr;
        g = g < gates ? g : gates - 1;
        const long long n = ((long long)(g >> 5) << 6) + (g & 31);
        wrow0 = p.W + n * p.ldw;
        wrow1 = wrow0 + 32 * p.ldw;
    };
    struct Regs { u32x4 g[2], u[2]; };
    Regs r[DEPTH];
    int ig = 0, iu = 0, ij = 0;
    auto issue = [&](Regs& x) {
        const long long k0 = (long long)(wave + 8 * ij) * KS8 + q * 32;
        x.g[0] = ld16(wrow0 + k0); x.g[1] = ld16(wrow0 + k0 + 16);
        x.u[0] = ld16(wrow1 + k0); x.u[1] = ld16(wrow1 + k0 + 16);
        ++ig;
        if (++ij == spw) { ij = 0; ++iu; if (iu < my_units) set_rows(iu); }
    };
    set_rows(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (ig < total) issue(r[d]);
    // activations -> LDS once (image and arithmetic of skinny_fp8_kernel's ALDS form)
    for (int m = wave; m < RM; m += 8) {
        const bool real = m < p.M;
        const char* src = p.A + (long long)m * p.lda * 2;
        float sq = 0.f;
        for (int cc = lane; cc < p.K / 8; cc += 64) {
            u32x4 v = real ? ld16(src + cc * 16) : u32x4{0u, 0u, 0u, 0u};
            if constexpr (AMODE == A_RMSNORM) {
                const u32x4 g = ld16(p.norm_w + cc * 16);
                u32x4 o;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float x0 = bf16_lo(v[d]), x1 = bf16_hi(v[d]);
                    sq += x0 * x0 + x1 * x1;
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    bf16x2_t pr;
                    pr[0] = (bf16)(x0 * bf16_lo(g[d]));
                    pr[1] = (bf16)(x1 * bf16_hi(g[d]));
                    o[d] = __builtin_bit_cast(uint32_t, pr);
                }
                v = o;
            }
            st16(aimg + (cc >> 4) * (RM * 256) + (cc & 3) * (RM * 64) + ((((cc >> 2) & 3) * RM + m) << 4), v);
        }
        if constexpr (AMODE == A_RMSNORM) {
            sq = wave_sum(sq);
            if (lane == 0) red_ss[m] = sq;
        }
    }
    __syncthreads();
    const int a_lane = (q * RM + (c16 & (RM - 1))) << 4;
    f32x4 accg = f32x4{0.f, 0.f, 0.f, 0.f}, accu = f32x4{0.f, 0.f, 0.f, 0.f};
    int cu = 0, cj = 0;
    auto widen = [&](const u32x4 (&w)[2], u32x4 (&wb)[4]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t lo, hi;
                fp8x4_to_bf16x4(w[h][d], lo, hi);
                wb[2 * h + (d >> 1)][2 * (d & 1)] = lo;
                wb[2 * h + (d >> 1)][2 * (d & 1) + 1] = hi;
            }
    };
    auto finish_unit = [&]() {
        float* rp = red + (cu & 1) * (8 * 2 * 256);
        *reinterpret_cast<f32x4*>(rp + (((wave * 2 + 0) * 64 + lane) << 2)) = accg;
        *reinterpret_cast<f32x4*>(rp + (((wave * 2 + 1) * 64 + lane) << 2)) = accu;
        accg = f32x4{0.f, 0.f, 0.f, 0.f}; accu = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (tid < 256) {
            const int reg = tid & 3, ln = tid >> 2;
            float g = 0.f, u = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                g += rp[(((w * 2 + 0) * 64 + ln) << 2) + reg];
                u += rp[(((w * 2 + 1) * 64 + ln) << 2) + reg];
            }
            const int mrow = 4 * (ln >> 4) + reg;
            const int gi = ((int)blockIdx.x + cu * (int)gridDim.x) * TR + (ln & 15);
            if ((ln & 15) < TR && gi < gates && mrow < p.M) {
                const int ng = ((gi >> 5) << 6) + (gi & 31);
                float rs = 1.0f;
                if constexpr (AMODE == A_RMSNORM) rs = rsqrtf(red_ss[mrow] / (float)p.K + p.norm_eps);
                g *= rs * p.wscale[ng];
                u *= rs * p.wscale[ng + 32];
                reinterpret_cast<bf16*>(p.C)[(long long)mrow * p.ldc + gi] = (bf16)(silu(g) * u);
            }
        }
    };
    for (int g0 = 0; g0 < total; g0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g0 + d < total) {                                  // wave-uniform; every wave reaches each unit's barrier exactly once
                const int s = wave + 8 * cj;
                u32x4 a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 256) + c * (RM * 64) + a_lane);
                u32x4 wb[4];
                widen(r[d].g, wb);
#pragma unroll
                for (int c = 0; c < 4; ++c) accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[c]), __builtin_bit_cast(bf16x8, wb[c]), accg, 0, 0, 0);
                widen(r[d].u, wb);
#pragma unroll
                for (int c = 0; c < 4; ++c) accu = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[c]), __builtin_bit_cast(bf16x8, wb[c]), accu, 0, 0, 0);
                if (ig < total) issue(r[d]);
                if (++cj == spw) { finish_unit(); cj = 0; ++cu; }
            }
        }
    }
}

template <int NT, int MT>
void launch_mode8(const SkinnyF8P& p, int amode, hipStream_t s) {
    const dim3 grid((NT == 2 && p.swiglu_out) ? (unsigned)cdiv(p.N / 2, p.tile_rows) : (unsigned)cdiv(p.N, NT * p.tile_rows)), block(512);
    const size_t lds = ((size_t)8 * NT * MT * 256 + 8 * MT * 16) * sizeof(float);
    if constexpr (MT == 1) {
        const size_t img = (size_t)p.a_rows * p.K * 2;
        const int use = afhip_opt(AFHIP_OPT_SKINNY_ALDS) != 0;     // A/B switch (default on for e4m3 weights)
        if (use && lds + img <= 120 * 1024) {
            static unsigned long long attr_done = 0;
            if (afhip_first_use_on_device(&attr_done)) {
                (void)hipFuncSetAttribute((const void*)skinny_fp8_kernel<NT, MT, A_RMSNORM, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
                (void)hipFuncSetAttribute((const void*)skinny_fp8_kernel<NT, MT, A_PLAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
            }
            if (amode == A_RMSNORM) hipLaunchKernelGGL((skinny_fp8_kernel<NT, MT, A_RMSNORM, true>), grid, block, lds + img, s, p);
            else hipLaunchKernelGGL((skinny_fp8_kernel<NT, MT, A_PLAIN, true>), grid, block, lds + img, s, p);
            return;
        }
    }
    if (amode == A_RMSNORM) hipLaunchKernelGGL((skinny_fp8_kernel<NT, MT, A_RMSNORM>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((skinny_fp8_kernel<NT, MT, A_PLAIN>), grid, block, lds, s, p);
}

template <int NT>
void launch_mt8(const SkinnyF8P& p, int mt, int amode, hipStream_t s) {
    if (mt == 1) launch_mode8<NT, 1>(p, amode, s);
    else launch_mode8<NT, 2>(p, amode, s);
}

}  // namespace

int afhip_gemm_skinny_fp8_impl(const afhip_gemm_args* a, void* stream) {
    AFHIP_CHECK(a->dtype == AFHIP_BF16, "afhip_gemm_skinny: fp8 weights need bf16 activations");
    AFHIP_CHECK(a->M > 0 && a->M <= 32, "afhip_gemm_skinny(fp8): M=%d must be in [1,32]", a->M);
    AFHIP_CHECK(a->K % KS8 == 0, "afhip_gemm_skinny(fp8): K=%d must be a multiple of %d", a->K, KS8);
    AFHIP_CHECK(a->w_scale != nullptr && !a->a_swiglu, "afhip_gemm_skinny(fp8): w_scale missing / a_swiglu unsupported");
    AFHIP_CHECK(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0 && (a->lda * 2) % 16 == 0 && a->ldw % 16 == 0,
                "afhip_gemm_skinny(fp8): A/W rows must be 16-byte aligned");
    const bool sw_out = a->act == AFHIP_ACT_SWIGLU;
    AFHIP_CHECK(a->lda >= a->K && a->ldw >= a->K && a->ldc >= (sw_out ? a->N / 2 : a->N), "afhip_gemm_skinny(fp8): leading dimension too small");
    SkinnyF8P p;
    p.A = (const char*)a->A; p.W = (const char*)a->W; p.wscale = a->w_scale; p.bias = (const char*)a->bias;
    p.res = (const char*)a->residual; p.norm_w = (const char*)a->a_norm_w; p.norm_eps = a->a_norm_eps;
    p.C = (char*)a->C;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldres = a->ldres;
    p.out_f32 = a->out_f32;
    const int mt = cdiv(a->M, 16);
    const bool wide = a->N >= 8192;
    if (sw_out) AFHIP_CHECK(wide && a->N % 64 == 0 && !a->bias && !a->residual && !a->out_f32, "afhip_gemm_skinny(fp8): SWIGLU epilogue needs N >= 8192, N %% 64 == 0, no bias/residual");
    p.swiglu_out = sw_out ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int amode = a->a_norm_w ? A_RMSNORM : A_PLAIN;
    p.a_rows = a->M <= 8 ? 8 : 16;
    p.tile_rows = 16;
    if (sw_out) {
        // SwiGLU pairs: a workgroup owns TR gate rows + their up rows.  TR = 16 cuts 18 944 gate rows into 1184 workgroups = 4.6 per CU
        // (the last round 62 % full); pick the TR <= 16 whose ceil(units / CUs) * TR is smallest (15 -> 1263 units = 4.93 per CU)
        const int cus = afhip_cu_count(), gates = a->N / 2;
        int best = 16, best_cost = cdiv(cdiv(gates, 16), cus) * 16;
        for (int tr = 15; tr >= 12; --tr) {
            const int cost = cdiv(cdiv(gates, tr), cus) * tr;
            if (cost < best_cost) { best_cost = cost; best = tr; }
        }
        p.tile_rows = best;
    }
    int nt_narrow = 1;
    if (!wide && !sw_out) {                 // one equal share of weight rows per CU (gemm_skinny.hip)
        const int rpw = cdiv(a->N, afhip_cu_count());
        nt_narrow = rpw <= 16 ? 1 : 2;
        p.tile_rows = rpw <= 16 ? rpw : (rpw <= 32 ? cdiv(rpw, 2) : 16);
    }
    if (sw_out && mt == 1 && a->K >= 1024) {
        // decode gate/up: persistent pair form (gemm_skinny.hip); AFHIP_SKINNY_PERSIST=0 keeps the plain form
        const int persist = afhip_opt(AFHIP_OPT_SKINNY_PERSIST) != 0;      // A/B switch (afhip_set_option flips it inside one process)
        const size_t lds = (size_t)(2 * 8 * 2 * 256 + 16) * sizeof(float) + (size_t)p.a_rows * a->K * 2;
        if (persist && lds <= 150 * 1024) {
            constexpr int PD = 4;
            static unsigned long long attr_done = 0;
            if (afhip_first_use_on_device(&attr_done)) {
                (void)hipFuncSetAttribute((const void*)skinny_fp8_pair_persist_kernel<A_RMSNORM, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                (void)hipFuncSetAttribute((const void*)skinny_fp8_pair_persist_kernel<A_PLAIN, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            }
            const int units = cdiv(a->N / 2, p.tile_rows), cus = afhip_cu_count();
            const dim3 grid((unsigned)(units < cus ? units : cus)), block(512);
            if (amode == A_RMSNORM) hipLaunchKernelGGL((skinny_fp8_pair_persist_kernel<A_RMSNORM, PD>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((skinny_fp8_pair_persist_kernel<A_PLAIN, PD>), grid, block, lds, s, p);
            AFHIP_LAUNCH_CHECK();
            return 0;
        }
    }
    if (sw_out) launch_mt8<2>(p, mt, amode, s);
    else if (wide) launch_mt8<4>(p, mt, amode, s);
    else if (nt_narrow == 2) launch_mt8<2>(p, mt, amode, s);
    else launch_mt8<1>(p, mt, amode, s);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
