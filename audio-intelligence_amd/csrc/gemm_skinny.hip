// Skinny GEMM for decode (M <= 64 rows): C[M,N] = A[M,K] . W[N,K]^T (+bias, +residual).  HBM-bound weight stream:
// W goes straight from global memory into MFMA B-operand registers (no LDS: every weight byte is used once),
// 32 contiguous bytes per lane per step so that the 4 lanes that share a weight row read one full 128-byte line;
// the K index inside a step is permuted identically on both operands, which MFMA does not care about.
//   bf16: step = 64 K = 2 x v_mfma_f32_16x16x32_bf16      f32: step = 32 K = 8 x v_mfma_f32_16x16x4_f32
// One workgroup = 8 waves = NT 16-row weight tiles; the waves interleave over K steps (wave w takes steps
// w, w+8, ...) and their partial sums are combined through LDS.  Activations (tiny, L2-resident) are re-read
// per step; NT = 4 amortises them for wide outputs (gate/up, lm_head), NT = 1 keeps the grid >= 224 workgroups
// for the 3584-wide projections.
#include "common.h"

namespace {

struct SkinnyP {
    const char* A;
    const char* W;
    const char* bias;
    const char* res;
    char* C;
    int M, N, K;
    long long lda, ldw, ldc, ldres;
    int out_f32;
};

template <typename T> struct Step;
template <> struct Step<bf16> { static constexpr int K = 64; };
template <> struct Step<float> { static constexpr int K = 32; };

template <typename T, int NT, int MT>
__global__ __launch_bounds__(512) void skinny_kernel(SkinnyP p) {
    constexpr int SZ = sizeof(T);
    constexpr int KS = Step<T>::K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [8 waves][NT][MT][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int n_base = blockIdx.x * (NT * 16);

    const char* wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int n = n_base + t * 16 + c16;
        n = n < p.N ? n : p.N - 1;
        wrow[t] = p.W + (long long)n * p.ldw * SZ + q * 32;
    }
    const char* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + c16;
        m = m < p.M ? m : p.M - 1;
        arow[t] = p.A + (long long)m * p.lda * SZ + q * 32;
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.K / KS;
    for (int s = wave; s < nsteps; s += 8) {
        const long long koff = (long long)s * KS * SZ;   // = s * 128 bytes
        u32x4 w0[NT], w1[NT], a0[MT], a1[MT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { w0[t] = ld16(wrow[t] + koff); w1[t] = ld16(wrow[t] + koff + 16); }
#pragma unroll
        for (int t = 0; t < MT; ++t) { a0[t] = ld16(arow[t] + koff); a1[t] = ld16(arow[t] + koff + 16); }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (SZ == 2) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0[mt]), __builtin_bit_cast(bf16x8, w0[nt]), acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[mt]), __builtin_bit_cast(bf16x8, w1[nt]), acc[nt][mt], 0, 0, 0);
                } else {
                    const f32x4 x0 = __builtin_bit_cast(f32x4, a0[mt]), x1 = __builtin_bit_cast(f32x4, a1[mt]);
                    const f32x4 y0 = __builtin_bit_cast(f32x4, w0[nt]), y1 = __builtin_bit_cast(f32x4, w1[nt]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], y0[j], acc[nt][mt], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], y1[j], acc[nt][mt], 0, 0, 0);
                }
            }
    }

    // ---- combine the 8 K-slices through LDS ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(red + ((((wave * NT + nt) * MT + mt) * 64 + lane) << 2)) = acc[nt][mt];
    __syncthreads();
    // C/D map of the 16x16 MFMA: col (n) = lane & 15, row (m) = 4 * (lane >> 4) + reg
    for (int o = tid; o < NT * MT * 256; o += 512) {
        const int reg = o & 3, ln = (o >> 2) & 63, tile = o >> 8;
        const int mt = tile % MT, nt = tile / MT;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[((((w * NT + nt) * MT + mt) * 64 + ln) << 2) + reg];
        const int n = n_base + nt * 16 + (ln & 15);
        const int m = mt * 16 + 4 * (ln >> 4) + reg;
        if (n < p.N && m < p.M) {
            if (p.bias) v += to_f32<T>(reinterpret_cast<const T*>(p.bias)[n]);
            if (p.res) v += to_f32<T>(reinterpret_cast<const T*>(p.res)[(long long)m * p.ldres + n]);
            if (p.out_f32) reinterpret_cast<float*>(p.C)[(long long)m * p.ldc + n] = v;
            else reinterpret_cast<T*>(p.C)[(long long)m * p.ldc + n] = from_f32<T>(v);
        }
    }
}

template <typename T, int NT>
void launch_mt(const SkinnyP& p, int mt, hipStream_t s) {
    const dim3 grid(cdiv(p.N, NT * 16)), block(512);
    const size_t lds = (size_t)8 * NT * mt * 64 * 4 * sizeof(float);
    switch (mt) {
        case 1: hipLaunchKernelGGL((skinny_kernel<T, NT, 1>), grid, block, lds, s, p); break;
        case 2: hipLaunchKernelGGL((skinny_kernel<T, NT, 2>), grid, block, lds, s, p); break;
        case 3: hipLaunchKernelGGL((skinny_kernel<T, NT, 3>), grid, block, lds, s, p); break;
        default: hipLaunchKernelGGL((skinny_kernel<T, NT, 4>), grid, block, lds, s, p); break;
    }
}

}  // namespace

extern "C" int afhip_gemm_skinny(const afhip_gemm_args* a, void* stream) {
    AFHIP_CHECK(a != nullptr, "afhip_gemm_skinny: null args");
    AFHIP_CHECK(a->dtype == AFHIP_F32 || a->dtype == AFHIP_BF16, "afhip_gemm_skinny: bad dtype %d", a->dtype);
    AFHIP_CHECK(a->M > 0 && a->M <= 64, "afhip_gemm_skinny: M=%d must be in [1,64] (use afhip_gemm)", a->M);
    AFHIP_CHECK(a->N > 0 && a->K > 0, "afhip_gemm_skinny: bad shape N=%d K=%d", a->N, a->K);
    const int ks = a->dtype == AFHIP_BF16 ? 64 : 32;
    AFHIP_CHECK(a->K % ks == 0, "afhip_gemm_skinny: K=%d must be a multiple of %d", a->K, ks);
    AFHIP_CHECK(a->A && a->W && a->C, "afhip_gemm_skinny: null operand");
    AFHIP_CHECK(a->act == AFHIP_ACT_NONE && a->conv_C == 0 && a->res_row_mod == 0, "afhip_gemm_skinny: act/conv/row-mod unsupported");
    const size_t sz = dtype_size(a->dtype);
    AFHIP_CHECK(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0 && (a->lda * sz) % 16 == 0 && (a->ldw * sz) % 16 == 0,
                "afhip_gemm_skinny: A/W rows must be 16-byte aligned");
    AFHIP_CHECK(a->lda >= a->K && a->ldw >= a->K && a->ldc >= a->N, "afhip_gemm_skinny: leading dimension too small");
    if (a->residual) AFHIP_CHECK(a->ldres >= a->N, "afhip_gemm_skinny: ldres < N");
    SkinnyP p;
    p.A = (const char*)a->A; p.W = (const char*)a->W; p.bias = (const char*)a->bias; p.res = (const char*)a->residual;
    p.C = (char*)a->C;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldres = a->ldres;
    p.out_f32 = a->out_f32;
    const int mt = cdiv(a->M, 16);
    hipStream_t s = (hipStream_t)stream;
    const bool wide = a->N >= 8192 && mt <= 2;   // NT=4 needs 8*NT*MT KiB of LDS for the K-slice combine
    if (a->dtype == AFHIP_BF16) {
        if (wide) launch_mt<bf16, 4>(p, mt, s); else launch_mt<bf16, 1>(p, mt, s);
    } else {
        if (wide) launch_mt<float, 4>(p, mt, s); else launch_mt<float, 1>(p, mt, s);
    }
    AFHIP_LAUNCH_CHECK();
    return 0;
}
