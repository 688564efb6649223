// Do a wave's MFMAs overlap ANOTHER wave's VALU work on the same SIMD (gfx950)?  One 512-thread workgroup per CU: waves 0-3 run an
// MFMA loop (v_mfma_f32_32x32x16_bf16, 4 independent accumulators), waves 4-7 (their SIMD partners) a VALU loop (v_fma_f32 or
// v_exp_f32).  Times: MFMA alone, VALU alone, both.  Also one wave interleaving k VALU instructions after every MFMA.
// Build + run: hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k_split(float* out, int mfma_iters, int valu_iters, int valu_kind) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        b8v a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)1.0f; }
        for (int i = 0; i < mfma_iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        if (valu_kind == 0) {
            for (int i = 0; i < valu_iters; ++i)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else {
            for (int i = 0; i < valu_iters; ++i)
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
}

// one wave per SIMD: after every MFMA, K independent VALU instructions (compile-time K)
template <int K, int KIND>
__global__ __launch_bounds__(256) void k_inter(float* out, int iters) {
    f16v c0 = {0}, c1 = {0};
    b8v a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)1.0f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k & 7]));
            else asm volatile("v_exp_f32 %0, %0" : "+v"(v[k & 7]));
        }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[k & 7]));
            else asm volatile("v_exp_f32 %0, %0" : "+v"(v[k & 7]));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7];
}

static float timeit(void (*launch)(float*), float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch(out);            // warm the clocks
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); launch(out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
}
#define SPLIT(NAME, MI, VI, KIND) static void NAME(float* o) { k_split<<<256, 512>>>(o, MI, VI, KIND); }
SPLIT(s_m, 20000, 0, 0) SPLIT(s_f, 0, 40000, 0) SPLIT(s_mf, 20000, 40000, 0) SPLIT(s_e, 0, 20000, 1) SPLIT(s_me, 20000, 20000, 1)
#define INTER(NAME, K, KIND) static void NAME(float* o) { k_inter<K, KIND><<<256, 256>>>(o, 20000); }
INTER(i0, 0, 0) INTER(i2, 2, 0) INTER(i4, 4, 0) INTER(i6, 6, 0) INTER(i8, 8, 0) INTER(i12, 12, 0) INTER(e2, 2, 1) INTER(e4, 4, 1)

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const float m = timeit(s_m, out), f = timeit(s_f, out), mf = timeit(s_mf, out), e = timeit(s_e, out), me = timeit(s_me, out);
    printf("split waves: MFMA alone (80000 per wave) %.3f ms = %.2f ns/MFMA; fma alone (320000) %.3f ms = %.2f ns/instr; both %.3f ms (sum %.3f, max %.3f)\n", m, m * 1e6 / 80000, f, f * 1e6 / 320000, mf, m + f, m > f ? m : f);
    printf("split waves: exp alone (160000) %.3f ms = %.2f ns/instr; MFMA + exp %.3f ms (sum %.3f, max %.3f)\n", e, e * 1e6 / 160000, me, m + e, m > e ? m : e);
    const float t0 = timeit(i0, out);
    printf("one wave, MFMA + K fma per MFMA (40000 MFMAs): K=0 %.3f ms (%.2f ns/MFMA)", t0, t0 * 1e6 / 40000);
    printf("  K=2 %.3f  K=4 %.3f  K=6 %.3f  K=8 %.3f  K=12 %.3f ms\n", timeit(i2, out), timeit(i4, out), timeit(i6, out), timeit(i8, out), timeit(i12, out));
    printf("one wave, MFMA + K exp per MFMA: K=2 %.3f  K=4 %.3f ms\n", timeit(e2, out), timeit(e4, out));
    return 0;
}
