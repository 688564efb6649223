// MFMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] . W[N,K]^T), both operands K-contiguous ("NT").
// One kernel template serves bf16 (v_mfma_f32_32x32x16_bf16) and exact f32 (v_mfma_f32_32x32x2_f32) because
// the LDS image is defined in BYTES: a tile is 128 rows x 128 B (64 bf16 or 32 f32 of K), 16-B chunks
// XOR-swizzled by (row & 7), and every lane reads 8 consecutive K elements of its row (common.h mma16).
// 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile = 2x2 MFMA tiles, f32 accumulators.
// Register-prefetched double-buffered LDS, one barrier per K tile.  Optional implicit im2col on A (conv1d
// k=3 pad=1, channel-last input) and fused epilogues: bias, erf-GELU, SwiGLU pair, residual / position table.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;  // tile rows, bytes of K per row
constexpr int TILE_BYTES = BM * ROWB;           // 16 KiB per operand per stage

struct GemmP {
    const char* A;
    const char* W;
    const char* bias;
    const char* res;
    char* C;
    int M, N, K;
    long long lda, ldw, ldc, ldres;  // in elements
    int act, res_row_mod;
    int conv_Tin, conv_Tout, conv_stride, conv_C;
    int tiles_m, tiles_n;
    int out_f32;
};

// bijective XCD-aware remap of a linear workgroup id (blocks b and b+8 share an XCD under round-robin dispatch):
// each XCD walks a contiguous range of tiles so neighbouring tiles (same A panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7, s = id >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + s;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    constexpr int SZ = sizeof(T);
    constexpr int BK = ROWB / SZ;          // K elements per tile
    constexpr int KC = BK / 16;            // 16-wide MFMA steps per tile
    constexpr int CH_PER_STEP = 16 * SZ / 16;  // 16-B chunks per lane-half per step: 1 (bf16: 2 per step / 2 halves) handled below
    (void)CH_PER_STEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int tm = wg / p.tiles_n, tn = wg % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging coordinates: 4 chunks per operand ----
    const int cc = tid & 7;        // 16-B chunk inside the 128-B row
    const int r0 = tid >> 3;       // rows r0 + 32*i
    const char* a_ptr[4];
    int a_ts[4];                   // conv: first source time step (t*stride - 1); plain: unused
    const char* w_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m > p.M - 1) m = p.M - 1;
        if (p.conv_C > 0) {
            const int b = m / p.conv_Tout, t = m - b * p.conv_Tout;
            a_ts[i] = t * p.conv_stride - 1;
            a_ptr[i] = p.A + ((long long)b * p.conv_Tin) * p.conv_C * SZ + cc * 16;
        } else {
            a_ts[i] = 0;
            a_ptr[i] = p.A + (long long)m * p.lda * SZ + cc * 16;
        }
        int n = n0 + r0 + 32 * i;
        if (n > p.N - 1) n = p.N - 1;
        w_ptr[i] = p.W + (long long)n * p.ldw * SZ + cc * 16;
    }
    const int nk = p.K / BK;

    u32x4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
        const long long kbyte = (long long)kt * ROWB;
        if (p.conv_C > 0) {
            const int k0 = kt * BK;
            const int tap = k0 / p.conv_C, c0 = k0 - tap * p.conv_C;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = a_ts[i] + tap;
                if (ts >= 0 && ts < p.conv_Tin)
                    ra[i] = ld16(a_ptr[i] + ((long long)ts * p.conv_C + c0) * SZ);
                else
                    ra[i] = u32x4{0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = ld16(a_ptr[i] + kbyte);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) rw[i] = ld16(w_ptr[i] + kbyte);
    };
    auto store_tile = [&](int buf) {
        char* sa = smem + buf * 2 * TILE_BYTES;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = r0 + 32 * i;
            const int off = row * ROWB + ((cc ^ (row & 7)) << 4);
            st16(sa + off, ra[i]);
            st16(sw + off, rw[i]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    auto compute_tile = [&](int buf) {
        const char* sa = smem + buf * 2 * TILE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            typename Frag8<T>::type fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rowa = wm * 64 + i * 32 + fr;
                const int roww = wn * 64 + i * 32 + fr;
                if constexpr (SZ == 2) {
                    const int ch = kc * 2 + fh;
                    fa[i] = *reinterpret_cast<const bf16x8*>(sa + rowa * ROWB + ((ch ^ (rowa & 7)) << 4));
                    fb[i] = *reinterpret_cast<const bf16x8*>(sw + roww * ROWB + ((ch ^ (roww & 7)) << 4));
                } else {
                    const int ch = kc * 4 + fh * 2;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(sa + rowa * ROWB + ((ch ^ (rowa & 7)) << 4));
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(sa + rowa * ROWB + (((ch + 1) ^ (rowa & 7)) << 4));
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(sw + roww * ROWB + ((ch ^ (roww & 7)) << 4));
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(sw + roww * ROWB + (((ch + 1) ^ (roww & 7)) << 4));
                    fa[i] = f32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    fb[i] = f32x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma16(fa[i], fb[j], acc[i][j]);
        }
    };

    // ---- main loop: prefetch tile kt+1 into registers while computing tile kt from LDS ----
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        compute_tile(cur);
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* res = reinterpret_cast<const T*>(p.res);
    T* C = reinterpret_cast<T*>(p.C);
    if (p.act == AFHIP_ACT_SWIGLU) {
        // wave's 64 N-rows = 32 gate rows (j=0) then 32 up rows (j=1) of the same 32 outputs
        const int ncol = ((n0 + wn * 64) >> 1) + fr;
        const bool nok = (n0 + wn * 64 + fr) < p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + mfma32_row(e, lane);
                if (m < p.M && nok) {
                    const float g = acc[i][0][e], u = acc[i][1][e];
                    C[(long long)m * p.ldc + ncol] = from_f32<T>(silu(g) * u);
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.N) continue;
        const float bv = bias ? to_f32<T>(bias[n]) : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + mfma32_row(e, lane);
                if (m < p.M) {
                    float v = acc[i][j][e] + bv;
                    if (p.act == AFHIP_ACT_GELU) v = gelu_erf(v);
                    if (res) {
                        const int rm = p.res_row_mod > 0 ? (m % p.res_row_mod) : m;
                        v += to_f32<T>(res[(long long)rm * p.ldres + n]);
                    }
                    if (p.out_f32) reinterpret_cast<float*>(p.C)[(long long)m * p.ldc + n] = v;
                    else C[(long long)m * p.ldc + n] = from_f32<T>(v);
                }
            }
    }
}

}  // namespace

// ---- optional in-library timing of this kernel (bench.py roofline leg): HIP events on the launch stream ----
namespace {
struct GemmProf {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;      // 2 per launch
    double* flops = nullptr;
    int* dtype = nullptr;
} g_prof;
}  // namespace

extern "C" int afhip_prof_enable(int max_launches) {
    AFHIP_CHECK(max_launches > 0, "afhip_prof_enable: bad capacity");
    if (g_prof.cap < max_launches) {
        for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
        free(g_prof.ev); free(g_prof.flops); free(g_prof.dtype);
        g_prof.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * max_launches);
        g_prof.flops = (double*)malloc(sizeof(double) * max_launches);
        g_prof.dtype = (int*)malloc(sizeof(int) * max_launches);
        for (int i = 0; i < 2 * max_launches; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) { afhip_set_error("afhip_prof_enable: hipEventCreate failed"); return AFHIP_ERR_LAUNCH; }
        g_prof.cap = max_launches;
    }
    g_prof.n = 0;
    g_prof.on = true;
    return 0;
}

// Waits for the recorded events; returns launches / summed milliseconds / summed algorithmic FLOPs (2MNK) of
// the afhip_gemm launches of `dtype` since afhip_prof_enable, and switches recording off.
extern "C" int afhip_prof_collect(int dtype, int* n_launches, double* total_ms, double* total_flops) {
    AFHIP_CHECK(n_launches && total_ms && total_flops, "afhip_prof_collect: null pointer");
    g_prof.on = false;
    *n_launches = 0; *total_ms = 0.0; *total_flops = 0.0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (g_prof.dtype[i] != dtype) continue;
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) {
            afhip_set_error("afhip_prof_collect: event query failed");
            return AFHIP_ERR_LAUNCH;
        }
        *n_launches += 1; *total_ms += ms; *total_flops += g_prof.flops[i];
    }
    return 0;
}

extern "C" int afhip_gemm(const afhip_gemm_args* a, void* stream) {
    AFHIP_CHECK(a != nullptr, "afhip_gemm: null args");
    AFHIP_CHECK(a->dtype == AFHIP_F32 || a->dtype == AFHIP_BF16, "afhip_gemm: bad dtype %d", a->dtype);
    AFHIP_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "afhip_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    const int bk = a->dtype == AFHIP_BF16 ? 64 : 32;
    AFHIP_CHECK(a->K % bk == 0, "afhip_gemm: K=%d must be a multiple of %d", a->K, bk);
    AFHIP_CHECK(a->A && a->W && a->C, "afhip_gemm: null operand");
    AFHIP_CHECK(a->act >= 0 && a->act <= 2, "afhip_gemm: bad act %d", a->act);
    const size_t sz = dtype_size(a->dtype);
    AFHIP_CHECK(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, "afhip_gemm: A/W must be 16-byte aligned");
    AFHIP_CHECK((a->ldw * sz) % 16 == 0, "afhip_gemm: ldw*size must be a multiple of 16");
    if (a->conv_C > 0) {
        AFHIP_CHECK(a->conv_C % bk == 0, "afhip_gemm: conv_C=%d must be a multiple of %d", a->conv_C, bk);
        AFHIP_CHECK(a->K == 3 * a->conv_C, "afhip_gemm: conv K=%d != 3*C=%d", a->K, 3 * a->conv_C);
        AFHIP_CHECK(a->conv_Tin > 0 && a->conv_Tout > 0 && a->M % a->conv_Tout == 0, "afhip_gemm: conv M=%d not a multiple of Tout=%d", a->M, a->conv_Tout);
        AFHIP_CHECK((a->conv_Tout - 1) * a->conv_stride - 1 + 2 <= a->conv_Tin, "afhip_gemm: conv geometry Tin=%d Tout=%d stride=%d", a->conv_Tin, a->conv_Tout, a->conv_stride);
    } else {
        AFHIP_CHECK((a->lda * sz) % 16 == 0, "afhip_gemm: lda*size must be a multiple of 16");
        AFHIP_CHECK(a->lda >= a->K && a->ldw >= a->K, "afhip_gemm: lda/ldw < K");
    }
    if (a->act == AFHIP_ACT_SWIGLU) {
        AFHIP_CHECK(a->N % 64 == 0 && !a->bias && !a->residual && !a->out_f32, "afhip_gemm: SWIGLU needs N%%64==0, no bias/residual/out_f32");
        AFHIP_CHECK(a->ldc >= a->N / 2, "afhip_gemm: ldc < N/2");
    } else {
        AFHIP_CHECK(a->ldc >= a->N, "afhip_gemm: ldc < N");
    }
    if (a->residual) AFHIP_CHECK(a->ldres >= a->N, "afhip_gemm: ldres < N");

    GemmP p;
    p.A = (const char*)a->A; p.W = (const char*)a->W; p.bias = (const char*)a->bias; p.res = (const char*)a->residual;
    p.C = (char*)a->C;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc; p.ldres = a->ldres;
    p.act = a->act; p.res_row_mod = a->res_row_mod;
    p.conv_Tin = a->conv_Tin; p.conv_Tout = a->conv_Tout; p.conv_stride = a->conv_stride; p.conv_C = a->conv_C;
    p.tiles_m = cdiv(a->M, BM); p.tiles_n = cdiv(a->N, BN);
    p.out_f32 = a->out_f32;
    const long long nwg = (long long)p.tiles_m * p.tiles_n;
    AFHIP_CHECK(nwg < (1ll << 31), "afhip_gemm: grid too large");
    const size_t lds = 4 * TILE_BYTES;
    hipStream_t s = (hipStream_t)stream;
    const bool rec = g_prof.on && g_prof.n < g_prof.cap;
    const int slot = g_prof.n;
    if (rec) (void)hipEventRecord(g_prof.ev[2 * slot], s);
    if (a->dtype == AFHIP_BF16)
        hipLaunchKernelGGL(gemm_kernel<bf16>, dim3((unsigned)nwg), dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL(gemm_kernel<float>, dim3((unsigned)nwg), dim3(256), lds, s, p);
    if (rec) {
        (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
        g_prof.flops[slot] = 2.0 * (double)a->M * (double)a->N * (double)a->K;
        g_prof.dtype[slot] = a->dtype;
        g_prof.n = slot + 1;
    }
    AFHIP_LAUNCH_CHECK();
    return 0;
}
