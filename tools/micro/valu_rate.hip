// Issue-rate microbenchmark for the VALU instructions of the attention softmax (gfx950): one wave per SIMD (or more), a loop of 64
// independent instructions of one kind, wall time by HIP events -> ns per wave-instruction, relative to v_fma_f32.
// Build + run:  hipcc -O2 --offload-arch=gfx950 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define KERNEL(NAME, ASM)                                                                            \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                             \
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < iters; ++i) {                                                            \
            REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                \
    }
// 8 instructions per asm block, 8 blocks per iteration = 64 instructions per iteration
KERNEL(k_fma, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7")
KERNEL(k_exp16, "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7")
KERNEL(k_max3, "v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1")
KERNEL(k_cvt, "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0")
KERNEL(k_add, "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %4\n v_add_f32 %4, %4, %5\n v_add_f32 %5, %5, %6\n v_add_f32 %6, %6, %7\n v_add_f32 %7, %7, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1\n v_ldexp_f32 %1, %1, %2\n v_ldexp_f32 %2, %2, %3\n v_ldexp_f32 %3, %3, %4\n v_ldexp_f32 %4, %4, %5\n v_ldexp_f32 %5, %5, %6\n v_ldexp_f32 %6, %6, %7\n v_ldexp_f32 %7, %7, %0")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 23, %1\n v_lshl_add_u32 %1, %1, 23, %2\n v_lshl_add_u32 %2, %2, 23, %3\n v_lshl_add_u32 %3, %3, 23, %4\n v_lshl_add_u32 %4, %4, 23, %5\n v_lshl_add_u32 %5, %5, 23, %6\n v_lshl_add_u32 %6, %6, 23, %7\n v_lshl_add_u32 %7, %7, 23, %0")
KERNEL(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7")
KERNEL(k_cvti, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7")
// packed f32: operands are register PAIRS; use 4 pairs
#define KERNEL2(NAME, ASM)                                                                           \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                             \
        typedef float f2 __attribute__((ext_vector_type(2)));                                        \
        f2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;              \
        for (int i = 0; i < iters; ++i) {                                                            \
            REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)                          \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[0] + a3[1];                          \
    }
KERNEL2(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %3, %0\n v_pk_fma_f32 %2, %2, %0, %1\n v_pk_fma_f32 %3, %3, %1, %2")
KERNEL2(k_pkadd, "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n v_pk_add_f32 %2, %2, %0\n v_pk_add_f32 %3, %3, %1")

template <typename K> double run(K k, const char* name, int waves_per_simd, float* out, double base) {
    const int iters = 4000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) k<<<grid, block>>>(out, iters);     // warm the clocks
    (void)hipDeviceSynchronize();
    float ms = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); k<<<grid, block>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1); ms = t < ms ? t : ms;
    }
    const double ns_per_instr = ms * 1e6 / ((double)iters * 64 * waves_per_simd);     // per wave-instruction per SIMD
    printf("%-22s waves/SIMD %d: %.3f ns per wave-instruction (x%.2f of v_fma_f32)\n", name, waves_per_simd, ns_per_instr, base > 0 ? ns_per_instr / base : 1.0);
    return ns_per_instr;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 3; ++w) {
        const double b = run(k_fma, "v_fma_f32", w, out, 0);
        run(k_exp, "v_exp_f32", w, out, b); run(k_exp16, "v_exp_f16", w, out, b); run(k_max3, "v_max3_f32", w, out, b);
        run(k_cvt, "v_cvt_pk_bf16_f32", w, out, b); run(k_add, "v_add_f32", w, out, b); run(k_ldexp, "v_ldexp_f32", w, out, b);
        run(k_lshladd, "v_lshl_add_u32", w, out, b); run(k_rndne, "v_rndne_f32", w, out, b); run(k_cvti, "v_cvt_i32_f32", w, out, b);
        run(k_pkfma, "v_pk_fma_f32", w, out, b); run(k_pkadd, "v_pk_add_f32", w, out, b);
    }
    return 0;
}
