// Skinny GEMM for decode (M <= 64 rows): C[M,N] = A'[M,K] . W[N,K]^T (+bias, +residual).  HBM-bound weight stream:
// W goes straight from global memory into MFMA B-operand registers (no LDS: every weight byte is used once),
// 32 contiguous bytes per lane per step so that the 4 lanes that share a weight row read one full 128-byte line;
// the K index inside a step is permuted identically on both operands, which MFMA does not care about.
//   bf16: step = 64 K = 2 x v_mfma_f32_16x16x32_bf16      f32: step = 32 K = 8 x v_mfma_f32_16x16x4_f32
// One workgroup = 8 waves = NT 16-row weight tiles; the waves interleave over K steps (wave w takes steps
// w, w+8, ...; two steps of loads in flight) and their partial sums are combined through LDS.  Activations (tiny,
// L2-resident) are re-read per step; NT = 4 amortises them for wide outputs (gate/up, lm_head), NT = 1 keeps the grid
// >= 224 workgroups for the 3584-wide projections.
// Two producer ops of the decoder layer are folded into the A-operand load so they cost no launch and no pass:
//   A_RMSNORM: A' = x (.) w_ln, the row scale rsqrt(mean(x^2)+eps) is applied to the accumulator at the end
//              (Qwen2RMSNorm, modeling_qwen2.py:238-252, in front of q/k/v and gate/up);
//   A_SWIGLU : A' = silu(g) * u read from the 32-row interleaved gate/up buffer (kept for completeness: every workgroup
//              re-evaluates the activation for its whole K range, so it only pays for very narrow outputs);
// and SwiGLU can instead be the EPILOGUE of the gate/up GEMM (`swiglu_out`): a NT = 4 workgroup owns exactly one 64-row
// gate/up block pair, so the K-slice combine emits silu(gate) * up straight into the [M, N/2] activation (modeling_qwen2.py:46-48).
#include "skinny.h"
#include <stdlib.h>

#ifndef SKINNY_DEPTH
#define SKINNY_DEPTH 2      /* K steps in flight per wave, NT >= 2 (3 costs the second workgroup per CU: measured slower) */
#endif
#ifndef SKINNY_NW1
#define SKINNY_NW1 8        /* waves per workgroup for the narrow bf16 tiles; 16 measured the same (3.79 vs 3.78 ms per 7B step): those launches are at the launch floor, not chain-bound */
#endif
#ifndef SKINNY_DEPTH1
#define SKINNY_DEPTH1 2     /* the same for the narrow NT = 1 tiles */
#endif

namespace {

enum { A_PLAIN = SKINNY_A_PLAIN, A_RMSNORM = SKINNY_A_RMSNORM, A_SWIGLU = SKINNY_A_SWIGLU };

template <typename T> struct Step;
template <> struct Step<bf16> { static constexpr int K = 64; };
template <> struct Step<float> { static constexpr int K = 32; };

// 16 bytes = 8 bf16 / 4 f32, viewed as f32 values
template <typename T> struct Half;
template <> struct Half<bf16> { static constexpr int N = 8; };
template <> struct Half<float> { static constexpr int N = 4; };

template <typename T> __device__ __forceinline__ float elem(const u32x4& v, int e);
template <> __device__ __forceinline__ float elem<bf16>(const u32x4& v, int e) {
    const uint32_t w = v[e >> 1];
    return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
}
template <> __device__ __forceinline__ float elem<float>(const u32x4& v, int e) { return __uint_as_float(v[e]); }

template <typename T> __device__ __forceinline__ u32x4 pack(const float* f);
template <> __device__ __forceinline__ u32x4 pack<bf16>(const float* f) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)f[i];
    return __builtin_bit_cast(u32x4, o);
}
template <> __device__ __forceinline__ u32x4 pack<float>(const float* f) {
    return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
}

// x0 / x1 (the up half of an interleaved gate/up A operand) exist only in A_SWIGLU mode: with MT = 4 they were 32 dead registers per
// step set and pushed skinny_kernel<*, 1, 4, *> into scratch
template <int NT, int MT, int XM> struct StepRegs { u32x4 w0[NT], w1[NT], a0[MT], a1[MT], x0[XM], x1[XM], n0, n1; };

// ALDS (bf16, MT == 1, K small enough): the activations are staged ONCE per workgroup into LDS -- RMSNorm gain applied, row
// sums of squares taken on the way -- in the order the MFMA A operand wants them ([K step][16-B half][q][row]: a wave reads
// 16 B per lane at consecutive addresses, conflict-free), so the K loop issues weight loads only.  Without it every wave re-reads
// its activation (and gain) fragments from L2 next to each weight fragment: as many vector-memory instructions again as the
// weight stream itself for NT = 1, which is what held the 3584-wide projections at 4 TB/s while NT = 4 reached 6.
template <typename T, int NT, int MT, int AMODE, int NW = 8, bool ALDS = false>
__global__ __launch_bounds__(NW * 64) void skinny_kernel(SkinnyP p) {
    constexpr int SZ = sizeof(T);
    constexpr int KS = Step<T>::K;
    constexpr int HN = Half<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);                 // [8 waves][NT][MT][64 lanes][4]
    float* red_ss = red + NW * NT * MT * 256;                      // [8 waves][MT][16] row sums of squares (A_RMSNORM)
    char* aimg = reinterpret_cast<char*>(red_ss + NW * MT * 16);   // ALDS: [K / 64][2][4][a_rows] x 16 B
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
        wrow[t] = p.W + (long long)n * p.ldw * SZ;
    }
    const char* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c16;
        m = m < p.M ? m : p.M - 1;
        arow[t] = p.A + (long long)m * p.lda * SZ;
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss[MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) ss[b] = 0.f;

    const int nsteps = p.K / KS;
    const int RM = p.a_rows;
    auto stage_a = [&]() {
        if constexpr (ALDS) {
            static_assert(sizeof(T) == 2 && MT == 1 && AMODE != A_SWIGLU, "ALDS: bf16, one row tile, plain / RMSNorm A");
            // wave w stages rows w, w + NW, ...: 16-B chunk cc of a row covers k = 8 cc .. 8 cc + 7 = step cc >> 3, q = (cc >> 1) & 3, half cc & 1
            for (int m = wave; m < RM; m += NW) {
                const bool real = m < p.M;
                const char* src = p.A + (long long)m * p.lda * SZ;
                float sq = 0.f;
                for (int cc = lane; cc < p.K / 8; cc += 64) {
                    u32x4 v = real ? ld16(src + cc * 16) : u32x4{0u, 0u, 0u, 0u};
                    if constexpr (AMODE == A_RMSNORM) {
                        const u32x4 g = ld16(p.norm_w + cc * 16);
                        float f[8];
    #pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float x = elem<T>(v, e);
                            sq += x * x;
                            f[e] = x * elem<T>(g, e);
                        }
                        v = pack<T>(f);
                    }
                    st16(aimg + (cc >> 3) * (RM * 128) + (cc & 1) * (RM * 64) + ((((cc >> 1) & 3) * RM + m) << 4), v);
                }
                if constexpr (AMODE == A_RMSNORM) {
                    sq = wave_sum(sq);
                    if (lane == 0) red_ss[m] = sq;
                }
            }
            __syncthreads();
        }
    };
    const int a_lane = (q * RM + (c16 & (RM - 1))) << 4;          // ALDS: this lane's 16 B inside a [q][row] plane (rows >= a_rows alias: dropped later)

    constexpr int XM = AMODE == A_SWIGLU ? MT : 1;
    typedef StepRegs<NT, MT, XM> Regs;
    auto issue = [&](int s, Regs& r) {
        const long long koff = (long long)s * KS * SZ + q * 32;          // byte offset of this lane's 32 B inside a K row
#pragma unroll
        for (int t = 0; t < NT; ++t) { r.w0[t] = ld16(wrow[t] + koff); r.w1[t] = ld16(wrow[t] + koff + 16); }
        if constexpr (AMODE == A_SWIGLU) {
            // logical k0 = s*KS + q*(KS/4): 32-block = k0 >> 5, offset inside = k0 & 31; gate at 64*block + off, up 32 further
            const int k0 = s * KS + q * (KS / 4);
            const long long goff = ((long long)(k0 >> 5) * 64 + (k0 & 31)) * SZ;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                r.a0[t] = ld16(arow[t] + goff); r.a1[t] = ld16(arow[t] + goff + 16);
                r.x0[t] = ld16(arow[t] + goff + 32 * SZ); r.x1[t] = ld16(arow[t] + goff + 32 * SZ + 16);
            }
        } else if constexpr (ALDS) {
            // the activation fragment is read from LDS at consume time (the image may not exist yet when the first steps are issued)
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t) { r.a0[t] = ld16(arow[t] + koff); r.a1[t] = ld16(arow[t] + koff + 16); }
            if constexpr (AMODE == A_RMSNORM) { r.n0 = ld16(p.norm_w + koff); r.n1 = ld16(p.norm_w + koff + 16); }
        }
    };
    auto consume = [&](Regs& r, int s) {
        if constexpr (ALDS) {
            r.a0[0] = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 128) + a_lane);
            r.a1[0] = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 128) + RM * 64 + a_lane);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (ALDS) {
                // staged: gain already applied, squares already summed
            } else if constexpr (AMODE == A_RMSNORM) {
                float f0[HN], f1[HN];
#pragma unroll
                for (int e = 0; e < HN; ++e) {
                    const float x0 = elem<T>(r.a0[mt], e), x1 = elem<T>(r.a1[mt], e);
                    ss[mt] += x0 * x0 + x1 * x1;
                    f0[e] = x0 * elem<T>(r.n0, e);
                    f1[e] = x1 * elem<T>(r.n1, e);
                }
                r.a0[mt] = pack<T>(f0); r.a1[mt] = pack<T>(f1);
            } else if constexpr (AMODE == A_SWIGLU) {
                float f0[HN], f1[HN];
#pragma unroll
                for (int e = 0; e < HN; ++e) {
                    f0[e] = silu(elem<T>(r.a0[mt], e)) * elem<T>(r.x0[mt], e);
                    f1[e] = silu(elem<T>(r.a1[mt], e)) * elem<T>(r.x1[mt], e);
                }
                r.a0[mt] = pack<T>(f0); r.a1[mt] = pack<T>(f1);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (SZ == 2) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r.a0[mt]), __builtin_bit_cast(bf16x8, r.w0[nt]), acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r.a1[mt]), __builtin_bit_cast(bf16x8, r.w1[nt]), acc[nt][mt], 0, 0, 0);
                } else {
                    const f32x4 x0 = __builtin_bit_cast(f32x4, r.a0[mt]), x1 = __builtin_bit_cast(f32x4, r.a1[mt]);
                    const f32x4 y0 = __builtin_bit_cast(f32x4, r.w0[nt]), y1 = __builtin_bit_cast(f32x4, r.w1[nt]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], y0[j], acc[nt][mt], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], y1[j], acc[nt][mt], 0, 0, 0);
                }
            }
    };

    // rolling window of two steps per wave: a register set is re-issued right after it is consumed, so the wave never
    // drains to zero loads in flight (issue-two / consume-two exposed one full HBM round trip per pair of steps; with
    // K = 3584 a wave only has 7 steps).  Steps are still consumed in ascending order: sums are bit-identical.
    {
        constexpr int DEPTH = NT == 1 ? SKINNY_DEPTH1 : SKINNY_DEPTH;   // register sets = K steps in flight per wave
        Regs r[DEPTH];
        int sx[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            sx[d] = wave + NW * d;
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
                    sx[d] += NW * DEPTH;
                    if (sx[d] < nsteps) { issue(sx[d], r[d]); more = true; }
                }
            }
        }
    }

    // ---- combine the 8 K-slices through LDS ----
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
    // C/D map of the 16x16 MFMA: col (n) = lane & 15, row (m) = 4 * (lane >> 4) + reg
    if constexpr (NT == 4 || NT == 2) {
        if (p.swiglu_out) {
            // first NT/2 tiles = gate rows, last NT/2 tiles = the matching up rows (32 further in W); C is [M, N/2]
            constexpr int NG = NT / 2;
            for (int o = tid; o < NG * MT * 256; o += NW * 64) {
                const int reg = o & 3, ln = (o >> 2) & 63, tile = o >> 8;
                const int mt = tile % MT, nt = tile / MT;        // gate tile
                float g = 0.f, u = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    g += red[((((w * NT + nt) * MT + mt) * 64 + ln) << 2) + reg];
                    u += red[((((w * NT + nt + NG) * MT + mt) * 64 + ln) << 2) + reg];
                }
                const int mrow = 4 * (ln >> 4) + reg;
                const int m = mt * 16 + mrow;
                const int gi = g0 + (ln & 15);                                   // pair form: gate index
                const int ng = pair ? ((gi >> 5) << 6) + (gi & 31) : n_base + nt * 16 + (ln & 15);   // gate row index inside W
                const bool live = pair ? ((ln & 15) < TR && gi < (p.N >> 1)) : (ng + 32 < p.N + 1);
                if (live && m < p.M) {
                    if constexpr (AMODE == A_RMSNORM) {
                        float sq = 0.f;
                        if constexpr (ALDS) sq = red_ss[mrow];
                        else {
#pragma unroll
                            for (int w = 0; w < NW; ++w) sq += red_ss[(w * MT + mt) * 16 + mrow];
                        }
                        const float r = rsqrtf(sq / (float)p.K + p.norm_eps);
                        g *= r; u *= r;
                    }
                    const int col = ((ng >> 6) << 5) + (ng & 31);
                    reinterpret_cast<T*>(p.C)[(long long)m * p.ldc + col] = from_f32<T>(silu(g) * u);
                }
            }
            return;
        }
    }
    for (int o = tid; o < NT * MT * 256; o += NW * 64) {
        const int reg = o & 3, ln = (o >> 2) & 63, tile = o >> 8;
        const int mt = tile % MT, nt = tile / MT;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[((((w * NT + nt) * MT + mt) * 64 + ln) << 2) + reg];
        const int n = n_base + nt * TR + (ln & 15);
        const int mrow = 4 * (ln >> 4) + reg;
        const int m = mt * 16 + mrow;
        if ((ln & 15) < TR && n < p.N && m < p.M) {
            if constexpr (AMODE == A_RMSNORM) {
                float sq = 0.f;
                if constexpr (ALDS) sq = red_ss[mrow];
                else {
#pragma unroll
                    for (int w = 0; w < NW; ++w) sq += red_ss[(w * MT + mt) * 16 + mrow];
                }
                v *= rsqrtf(sq / (float)p.K + p.norm_eps);
            }
            if (p.bias) v += to_f32<T>(reinterpret_cast<const T*>(p.bias)[n]);
            if (p.res) v += to_f32<T>(reinterpret_cast<const T*>(p.res)[(long long)m * p.ldres + n]);
            if (p.out_f32) reinterpret_cast<float*>(p.C)[(long long)m * p.ldc + n] = v;
            else reinterpret_cast<T*>(p.C)[(long long)m * p.ldc + n] = from_f32<T>(v);
        }
    }
}

// ---- persistent form of the SwiGLU pair GEMM (decode gate/up: bf16, M <= 16) --------------------------------------------------
// One workgroup per CU walks its pair units (TR gate rows + their up rows each) in ONE continuous weight stream: the per-wave
// window of DEPTH K steps rolls across unit boundaries, so the first loads of the next unit are in flight while the K slices of
// the finished unit are combined; the activations (RMSNorm gain applied, row sums of squares taken) are staged in LDS once per
// workgroup, not once per unit, so the stream issues weight loads only.  The plain form above starts 1263 workgroups of 7 K
// steps per wave: every one of them pays an HBM round trip before its first MFMA and re-reads its activation fragments from L2.
// A row's K order and the order of the K-slice sum are those of skinny_kernel: same bits.
template <int AMODE, int DEPTH>
__global__ __launch_bounds__(512) void skinny_pair_persist_kernel(SkinnyP p) {
    constexpr int NW = 8, KS = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);                 // [2 unit parities][8 waves][2 tiles][64 lanes][4]
    float* red_ss = red + 2 * NW * 2 * 256;                       // [16] row sums of squares (A_RMSNORM)
    char* aimg = reinterpret_cast<char*>(red_ss + 16);            // [K / 64][2 halves][4 q][a_rows] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int TR = p.tile_rows, gates = p.N >> 1, RM = p.a_rows;
    const int n_units = (gates + TR - 1) / TR;
    const int spw = p.K / (KS * NW);                              // K steps per wave per unit (the host checks K % 512 == 0)
    const int my_units = (n_units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * spw;                             // this wave's steps over all its units, in stream order
    const int cr = c16 < TR ? c16 : TR - 1;

    // ---- issue side: which unit / step the stream is at ----
    const char* wrow0 = nullptr;
    const char* wrow1 = nullptr;
    auto set_rows = [&](int ui) {
        int g = ((int)blockIdx.x + ui * (int)gridDim.x) * TR + cr;
        g = g < gates ? g : gates - 1;
        const long long n = ((long long)(g >> 5) << 6) + (g & 31);      // gate g = W row 64 (g >> 5) + (g & 31), its up row 32 further
        wrow0 = p.W + n * p.ldw * 2;
        wrow1 = wrow0 + 32 * p.ldw * 2;
    };
    struct Regs { u32x4 g0, g1, u0, u1; };
    Regs r[DEPTH];
    int ig = 0, iu = 0, ij = 0;                                    // next step to issue: stream index, unit, step inside the unit
    auto issue = [&](Regs& x) {
        const long long koff = (long long)(wave + NW * ij) * (KS * 2) + q * 32;
        x.g0 = ld16(wrow0 + koff); x.g1 = ld16(wrow0 + koff + 16);
        x.u0 = ld16(wrow1 + koff); x.u1 = ld16(wrow1 + koff + 16);
        ++ig;
        if (++ij == spw) { ij = 0; ++iu; if (iu < my_units) set_rows(iu); }
    };
    set_rows(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (ig < total) issue(r[d]);

    // ---- activations -> LDS, once (same image and arithmetic as skinny_kernel's ALDS form) ----
    for (int m = wave; m < RM; m += NW) {
        const bool real = m < p.M;
        const char* src = p.A + (long long)m * p.lda * 2;
        float sq = 0.f;
        for (int cc = lane; cc < p.K / 8; cc += 64) {
            u32x4 v = real ? ld16(src + cc * 16) : u32x4{0u, 0u, 0u, 0u};
            if constexpr (AMODE == A_RMSNORM) {
                const u32x4 g = ld16(p.norm_w + cc * 16);
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = elem<bf16>(v, e);
                    sq += x * x;
                    f[e] = x * elem<bf16>(g, e);
                }
                v = pack<bf16>(f);
            }
            st16(aimg + (cc >> 3) * (RM * 128) + (cc & 1) * (RM * 64) + ((((cc >> 1) & 3) * RM + m) << 4), v);
        }
        if constexpr (AMODE == A_RMSNORM) {
            sq = wave_sum(sq);
            if (lane == 0) red_ss[m] = sq;
        }
    }
    __syncthreads();
    const int a_lane = (q * RM + (c16 & (RM - 1))) << 4;

    f32x4 accg = f32x4{0.f, 0.f, 0.f, 0.f}, accu = f32x4{0.f, 0.f, 0.f, 0.f};
    int cu = 0, cj = 0;                                            // consume side: unit, step inside the unit
    auto finish_unit = [&]() {
        float* rp = red + (cu & 1) * (NW * 2 * 256);              // two buffers: the next unit's barrier orders the reuse
        *reinterpret_cast<f32x4*>(rp + (((wave * 2 + 0) * 64 + lane) << 2)) = accg;
        *reinterpret_cast<f32x4*>(rp + (((wave * 2 + 1) * 64 + lane) << 2)) = accu;
        accg = f32x4{0.f, 0.f, 0.f, 0.f}; accu = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (tid < 256) {
            const int reg = tid & 3, ln = tid >> 2;
            float g = 0.f, u = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                g += rp[(((w * 2 + 0) * 64 + ln) << 2) + reg];
                u += rp[(((w * 2 + 1) * 64 + ln) << 2) + reg];
            }
            const int mrow = 4 * (ln >> 4) + reg;
            const int gi = ((int)blockIdx.x + cu * (int)gridDim.x) * TR + (ln & 15);
            if ((ln & 15) < TR && gi < gates && mrow < p.M) {
                if constexpr (AMODE == A_RMSNORM) {
                    const float rs = rsqrtf(red_ss[mrow] / (float)p.K + p.norm_eps);
                    g *= rs; u *= rs;
                }
                reinterpret_cast<bf16*>(p.C)[(long long)mrow * p.ldc + gi] = (bf16)(silu(g) * u);
            }
        }
    };
    for (int g0 = 0; g0 < total; g0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (g0 + d < total) {                                  // workgroup-uniform: every wave has the same step count
                const int s = wave + NW * cj;
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 128) + a_lane);
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 128) + RM * 64 + a_lane);
                accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, r[d].g0), accg, 0, 0, 0);
                accg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, r[d].g1), accg, 0, 0, 0);
                accu = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, r[d].u0), accu, 0, 0, 0);
                accu = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, r[d].u1), accu, 0, 0, 0);
                if (ig < total) issue(r[d]);
                if (++cj == spw) { finish_unit(); cj = 0; ++cu; }
            }
        }
    }
}


// NW = waves that split K inside a workgroup.  The narrow bf16 tiles (NT = 1: q/k/v and o of the decoder, 224-288 workgroups of
// 7 K steps per wave) can split K over 16 waves (SKINNY_NW1); f32 always keeps 8 (its summation order is part of
// the bit-exact parity contract).
constexpr size_t SKINNY_ALDS_MAX = 120 * 1024;   // activation image + K-slice combine must leave the workgroup resident (160 KiB LDS)

template <typename T, int NT, int MT>
void launch_mode(const SkinnyP& p, int amode, hipStream_t s) {
    constexpr int NW = (NT == 1 && sizeof(T) == 2 && MT <= 2) ? SKINNY_NW1 : 8;
    const dim3 grid((NT == 2 && p.swiglu_out) ? (unsigned)cdiv(p.N / 2, p.tile_rows) : (unsigned)cdiv(p.N, NT * p.tile_rows)), block(NW * 64);
    const size_t lds = ((size_t)NW * NT * MT * 256 + NW * MT * 16) * sizeof(float);
    if constexpr (sizeof(T) == 2 && MT == 1) {
        const size_t img = (size_t)p.a_rows * p.K * 2;
        // A/B switch, default OFF for bf16 weights: measured 3.81-3.93 vs 3.63 ms per 7B decode step (the 57-KiB image halves the
        // workgroups per CU of gate/up, and co-resident workgroups hiding each other's ramp matter more than the saved loads);
        // the e4m3 kernel, whose K step carries 8 activation / gain loads per 2 NT weight loads, gains 3.5 % and keeps it on
        const int use = afhip_opt(AFHIP_OPT_SKINNY_ALDS) == 1;
        if (use && amode != A_SWIGLU && p.K % 64 == 0 && lds + img <= SKINNY_ALDS_MAX) {
            static unsigned long long attr_done = 0;
            if (afhip_first_use_on_device(&attr_done)) {
                (void)hipFuncSetAttribute((const void*)skinny_kernel<T, NT, MT, A_RMSNORM, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SKINNY_ALDS_MAX);
                (void)hipFuncSetAttribute((const void*)skinny_kernel<T, NT, MT, A_PLAIN, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SKINNY_ALDS_MAX);
            }
            if (amode == A_RMSNORM) hipLaunchKernelGGL((skinny_kernel<T, NT, MT, A_RMSNORM, NW, true>), grid, block, lds + img, s, p);
            else hipLaunchKernelGGL((skinny_kernel<T, NT, MT, A_PLAIN, NW, true>), grid, block, lds + img, s, p);
            return;
        }
    }
    switch (amode) {
        case A_RMSNORM: hipLaunchKernelGGL((skinny_kernel<T, NT, MT, A_RMSNORM, NW>), grid, block, lds, s, p); break;
        case A_SWIGLU: hipLaunchKernelGGL((skinny_kernel<T, NT, MT, A_SWIGLU, NW>), grid, block, lds, s, p); break;
        default: hipLaunchKernelGGL((skinny_kernel<T, NT, MT, A_PLAIN, NW>), grid, block, lds, s, p); break;
    }
}

template <typename T, int NT>
bool launch_mt(const SkinnyP& p, int mt, int amode, hipStream_t s) {
    switch (mt) {
        case 1: launch_mode<T, NT, 1>(p, amode, s); break;
        case 2: launch_mode<T, NT, 2>(p, amode, s); break;
        default:
            // NT = 4 is only ever chosen with mt <= 2 (the `wide` rule: 8 KiB of combine LDS per tile); not instantiating <*, 4, 3..4, *> keeps
            // their 60-154 spilled registers out of the binary altogether (every skinny kernel that CAN run is spill-free: tools/check_spills.py)
            if constexpr (NT < 4) {
                if (mt == 3) launch_mode<T, NT, 3>(p, amode, s);
                else launch_mode<T, NT, 4>(p, amode, s);
            } else {
                return false;
            }
            break;
    }
    return true;
}

}  // namespace

int afhip_gemm_skinny_fp8_impl(const afhip_gemm_args* a, void* stream);   // gemm_skinny_fp8.hip

extern "C" int afhip_gemm_skinny(const afhip_gemm_args* a, void* stream) {
    AFHIP_CHECK(a != nullptr, "afhip_gemm_skinny: null args");
    if (a->w_scale != nullptr) return afhip_gemm_skinny_fp8_impl(a, stream);
    AFHIP_CHECK(a->dtype == AFHIP_F32 || a->dtype == AFHIP_BF16, "afhip_gemm_skinny: bad dtype %d", a->dtype);
    AFHIP_CHECK(a->M > 0 && a->M <= 64, "afhip_gemm_skinny: M=%d must be in [1,64] (use afhip_gemm)", a->M);
    AFHIP_CHECK(a->N > 0 && a->K > 0, "afhip_gemm_skinny: bad shape N=%d K=%d", a->N, a->K);
    const int ks = a->dtype == AFHIP_BF16 ? 64 : 32;
    AFHIP_CHECK(a->K % ks == 0, "afhip_gemm_skinny: K=%d must be a multiple of %d", a->K, ks);
    AFHIP_CHECK(a->A && a->W && a->C, "afhip_gemm_skinny: null operand");
    AFHIP_CHECK((a->act == AFHIP_ACT_NONE || a->act == AFHIP_ACT_SWIGLU) && a->conv_C == 0 && a->res_row_mod == 0, "afhip_gemm_skinny: act/conv/row-mod unsupported");
    AFHIP_CHECK(!(a->a_norm_w && a->a_swiglu), "afhip_gemm_skinny: a_norm_w and a_swiglu are exclusive");
    const size_t sz = dtype_size(a->dtype);
    AFHIP_CHECK(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0 && (a->lda * sz) % 16 == 0 && (a->ldw * sz) % 16 == 0,
                "afhip_gemm_skinny: A/W rows must be 16-byte aligned");
    const bool sw_out = a->act == AFHIP_ACT_SWIGLU;
    AFHIP_CHECK(a->lda >= (a->a_swiglu ? 2 * a->K : a->K) && a->ldw >= a->K && a->ldc >= (sw_out ? a->N / 2 : a->N), "afhip_gemm_skinny: leading dimension too small");
    if (a->residual) AFHIP_CHECK(a->ldres >= a->N, "afhip_gemm_skinny: ldres < N");
    if (a->a_norm_w) AFHIP_CHECK(((uintptr_t)a->a_norm_w % 16) == 0, "afhip_gemm_skinny: a_norm_w must be 16-byte aligned");
    SkinnyP p;
    p.A = (const char*)a->A; p.W = (const char*)a->W; p.bias = (const char*)a->bias; p.res = (const char*)a->residual;
    p.norm_w = (const char*)a->a_norm_w; p.norm_eps = a->a_norm_eps;
    p.C = (char*)a->C;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldres = a->ldres;
    p.out_f32 = a->out_f32;
    const int amode = a->a_norm_w ? A_RMSNORM : (a->a_swiglu ? A_SWIGLU : A_PLAIN);
    const int mt = cdiv(a->M, 16);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = a->N >= 8192 && mt <= 2;   // NT=4 needs 8*NT*MT KiB of LDS for the K-slice combine
    if (sw_out) AFHIP_CHECK(wide && a->N % 64 == 0 && !a->bias && !a->residual && !a->out_f32, "afhip_gemm_skinny: SWIGLU epilogue needs N >= 8192, N %% 64 == 0, M <= 32, no bias/residual");
    p.swiglu_out = sw_out ? 1 : 0;
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
    // narrow outputs (the 3584 / 4608-wide decoder projections): 16-row tiles give 224 or 288 workgroups on 256 CUs -- 12 % of the
    // chip idle, or a second round for 32 of them.  Cut N into one share of ceil(N / CUs) rows per CU instead (14 rows x 256 for
    // N = 3584, 2 x 9 rows x 256 for N = 4608): every CU streams the same bytes.  A row's K order does not change: same bits.
    int nt_narrow = 1;
    if (!wide && !sw_out) {
        const int rpw = cdiv(a->N, afhip_cu_count());
        nt_narrow = rpw <= 16 ? 1 : 2;
        p.tile_rows = rpw <= 16 ? rpw : (rpw <= 32 ? cdiv(rpw, 2) : 16);
        if (mt > 2 && nt_narrow == 2) { nt_narrow = 1; p.tile_rows = 16; }      // LDS of the K-slice combine: NT * MT <= 4 tiles
    }
    p.n_tiles = 1; p.am_iv = nullptr; p.am_n_iv = 0; p.am_val = nullptr; p.am_idx = nullptr;
    if (a->dtype == AFHIP_BF16 && mt == 1) {
        // the 7B decode step's form: one persistent workgroup per CU, the A operand through LDS only (gemm_stream.hip).
        // AFHIP_SKINNY_STREAM=0 keeps the round-3 kernels (A/B switch; afhip_set_option lets one process compare the forms bit for bit)
        if (afhip_opt(AFHIP_OPT_SKINNY_STREAM) != 0) {
            SkinnyP ps = p;
            ps.n_tiles = sw_out ? 2 : (wide ? (p.a_rows == 8 ? 4 : 2) : nt_narrow);
            if (afhip_gemm_stream_bf16(ps, amode, s)) { AFHIP_LAUNCH_CHECK(); return 0; }
        }
    }
    if (a->dtype == AFHIP_BF16 && sw_out && mt == 1 && amode != A_SWIGLU && a->K % 512 == 0) {
        // decode gate/up: the persistent pair form (one continuous weight stream per CU); AFHIP_SKINNY_PERSIST=0 keeps the plain form
        const int persist = afhip_opt(AFHIP_OPT_SKINNY_PERSIST) != 0;      // A/B switch (afhip_set_option flips it inside one process)
        const size_t lds = (size_t)(2 * 8 * 2 * 256 + 16) * sizeof(float) + (size_t)p.a_rows * a->K * 2;
        if (persist && lds <= 150 * 1024) {
            constexpr int PD = 4;      // K steps in flight per wave; 6 and 8 measured the same (3.47-3.52 ms per 7B step): not the limiter
            static unsigned long long attr_done = 0;
            if (afhip_first_use_on_device(&attr_done)) {
                (void)hipFuncSetAttribute((const void*)skinny_pair_persist_kernel<A_RMSNORM, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                (void)hipFuncSetAttribute((const void*)skinny_pair_persist_kernel<A_PLAIN, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            }
            const int units = cdiv(a->N / 2, p.tile_rows), cus = afhip_cu_count();
            const dim3 grid((unsigned)(units < cus ? units : cus)), block(512);
            if (amode == A_RMSNORM) hipLaunchKernelGGL((skinny_pair_persist_kernel<A_RMSNORM, PD>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((skinny_pair_persist_kernel<A_PLAIN, PD>), grid, block, lds, s, p);
            AFHIP_LAUNCH_CHECK();
            return 0;
        }
    }
    bool ok;
    if (a->dtype == AFHIP_BF16) {
        if (sw_out) ok = launch_mt<bf16, 2>(p, mt, amode, s);
        else if (wide) ok = launch_mt<bf16, 4>(p, mt, amode, s);
        else if (nt_narrow == 2) ok = launch_mt<bf16, 2>(p, mt, amode, s);
        else ok = launch_mt<bf16, 1>(p, mt, amode, s);
    } else {
        if (sw_out) ok = launch_mt<float, 2>(p, mt, amode, s);
        else if (wide) ok = launch_mt<float, 4>(p, mt, amode, s);
        else if (nt_narrow == 2) ok = launch_mt<float, 2>(p, mt, amode, s);
        else ok = launch_mt<float, 1>(p, mt, amode, s);
    }
    AFHIP_CHECK(ok, "afhip_gemm_skinny: internal: no kernel for %d row tiles at this width", mt);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
