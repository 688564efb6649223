// Does the SHAPE of a wave's weight loads bound the decode GEMMs (gfx950)?  The skinny kernels read the B operand of
// v_mfma_f32_16x16x32_bf16 straight from a row-major [N, K] weight: lane (n = lane & 15, q = lane >> 4) takes 32 B of row n, so one
// wave-instruction touches 16 different 128-B lines (16 B of each).  This probe streams the same bytes with the same waves, window
// depth and MFMA work in two layouts:
//   PAT 0  row-major weight, fragment-shaped loads (what gemm_skinny.hip did in rounds 1-3)
//   PAT 1  weight stored in fragment order [unit][K step][tile][half][q][row] x 16 B: a wave-instruction reads one contiguous run
// and with the activations (8 rows) taken from LDS (AL 0), from global memory row-shaped (AL 1), fragment-ordered (AL 2), or fragment-ordered
// with both fragments of a step in ONE 64-lane load (AL 3: the 8 spare MFMA rows carry the other fragment).
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/stream_shape.hip -o /tmp/stream_shape && /tmp/stream_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }

struct P { const char* W; const char* A; float* out; int K, TR, units; };

template <int PAT, int AL, int NT, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(P p) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
    const int TR = p.TR, cr = c16 < TR ? c16 : TR - 1;
    const int nsteps = p.K / 64, spw = nsteps / 8;
    const int my_units = (p.units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * spw;
    for (int i = tid; i < 4096; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    struct R { u32x4 w0[NT], w1[NT], a0, a1; };
    R r[DEPTH];
    auto issue = [&](int i, R& x) {
        const int u = (int)blockIdx.x + (i / spw) * (int)gridDim.x, s = wave + 8 * (i % spw);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (PAT == 0) {
                const char* row = p.W + ((long long)(u * NT + t) * TR + cr) * p.K * 2 + (long long)s * 128 + q * 32;
                x.w0[t] = ld16(row); x.w1[t] = ld16(row + 16);
            } else {
                const char* b = p.W + (long long)u * NT * TR * p.K * 2 + ((long long)(s * NT + t) * 2) * (64 * TR) + (q * TR + cr) * 16;
                x.w0[t] = ld16(b); x.w1[t] = ld16(b + 64 * TR);
            }
        }
        if (AL == 1) {
            const char* a = p.A + (long long)(c16 & 7) * p.K * 2 + (long long)s * 128 + q * 32;
            x.a0 = ld16(a); x.a1 = ld16(a + 16);
        } else if (AL == 2) {
            const char* a = p.A + (long long)s * 1024 + (q * 8 + (c16 & 7)) * 16;
            x.a0 = ld16(a); x.a1 = ld16(a + 512);
        } else if (AL == 3) {
            // 8 activation rows: ONE load per step -- lanes c16 < 8 take their row of fragment 0, lanes c16 >= 8 the row c16 - 8 of fragment 1
            // (the 1 KiB of a step, no duplicates); fragment 1 reaches lanes 0..7 by a row rotation (DPP) when it is consumed
            const char* a = p.A + (long long)s * 1024 + (c16 < 8 ? 0 : 512) + (q * 8 + (c16 & 7)) * 16;
            x.a0 = ld16(a);
        }
    };
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < total) issue(d, r[d]);
    for (int i0 = 0; i0 < total; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int i = i0 + d;
            if (i < total) {
                u32x4 a0, a1;
                if (AL == 0) {
                    const int s = wave + 8 * (i % spw);
                    a0 = *reinterpret_cast<const u32x4*>(lds + ((s & 7) * 1024 + (q * 8 + (c16 & 7)) * 16));
                    a1 = *reinterpret_cast<const u32x4*>(lds + ((s & 7) * 1024 + 512 + (q * 8 + (c16 & 7)) * 16));
                } else if (AL == 3) {
                    a0 = r[d].a0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a1[e] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[d].a0[e], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
                } else { a0 = r[d].a0; a1 = r[d].a1; }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, r[d].w0[t]), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, r[d].w1[t]), acc[t], 0, 0, 0);
                }
                if (i + DEPTH < total) issue(i + DEPTH, r[d]);
            }
        }
    }
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) v += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    p.out[blockIdx.x * 512 + tid] = v;
}


// The same stream with a TRUE rolling window: hipcc's own wait insertion gives up on this loop (vmcnt(0) at its head: the whole window drains
// before the first step of every round is consumed, then DEPTH steps are re-issued back to back).  Here the loads are inline asm (the
// compiler inserts no waits for them) and each step waits for exactly its own four loads: vmcnt(4 (DEPTH - 1)) while the window is full,
// counted down over the last DEPTH - 1 steps.  NT = 1, activations fragment-ordered (AL 2), weights in rows (PAT 0).
template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel_cw(P p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, q = lane >> 4;
    const int TR = p.TR, cr = c16 < TR ? c16 : TR - 1;
    const int nsteps = p.K / 64, spw = nsteps / 8;
    const int my_units = (p.units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * spw;
    struct R { u32x4 w0, w1, a0, a1; };
    R r[DEPTH];
    auto issue = [&](int i, R& x) {
        const int u = (int)blockIdx.x + (i / spw) * (int)gridDim.x, s = wave + 8 * (i % spw);
        const char* row = p.W + ((long long)u * TR + cr) * p.K * 2 + (long long)s * 128 + q * 32;
        const char* a = p.A + (long long)s * 1024 + (q * 8 + (c16 & 7)) * 16;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\tglobal_load_dwordx4 %2, %5, off\n\tglobal_load_dwordx4 %3, %5, off offset:512"
                     : "=&v"(x.w0), "=&v"(x.w1), "=&v"(x.a0), "=&v"(x.a1) : "v"(row), "v"(a) : "memory");
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < total) issue(d, r[d]);
    auto consume = [&](R& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x.a0), __builtin_bit_cast(bf16x8, x.w0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x.a1), __builtin_bit_cast(bf16x8, x.w1), acc, 0, 0, 0);
    };
#define CW_WAIT(N, x) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x.w0), "+v"(x.w1), "+v"(x.a0), "+v"(x.a1) : "n"(N))
    const int full = total - (DEPTH - 1) > 0 ? total - (DEPTH - 1) : 0;      // steps consumed while DEPTH - 1 younger steps are in flight
    int i0 = 0;
    for (; i0 + DEPTH <= full; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            CW_WAIT(4 * (DEPTH - 1), r[d]);
            consume(r[d]);
            if (i0 + d + DEPTH < total) issue(i0 + d + DEPTH, r[d]);        // always true here except in the last round
        }
    }
    // the rest (< 2 DEPTH steps): nothing younger is guaranteed, wait for everything once
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = i0; i < total; ++i) {
        const int d = i % DEPTH;
#pragma unroll
        for (int e = 0; e < DEPTH; ++e)
            if (e == d) { CW_WAIT(0, r[e]); consume(r[e]); if (i + DEPTH < total) { issue(i + DEPTH, r[e]); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } }
    }
    p.out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

static char* g_big; static size_t g_big_bytes; static char* g_a; static float* g_out;

template <int DEPTH>
void run_cw(const char* name, int N, int K, int TR, int grid) {
    const int units = N / TR;
    const size_t bytes = (size_t)units * TR * K * 2;
    const size_t stride = (bytes + (1 << 21)) & ~((size_t)(1 << 21) - 1);
    const int regions = (int)(g_big_bytes / stride);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&](int i) { P p{g_big + (size_t)(i % regions) * stride, g_a, g_out, K, TR, units}; stream_kernel_cw<DEPTH><<<grid, 512>>>(p); };
    for (int i = 0; i < 8; ++i) launch(i);
    (void)hipDeviceSynchronize();
    const int reps = 40;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(i + 8);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-10s N=%6d K=%5d NT=1 TR=%2d grid=%4d depth=%d pat=rows A=frag COUNTED WAITS : %7.2f us  %5.2f TB/s\n", name, N, K, TR, grid, DEPTH, us, bytes / us / 1e6);
    fflush(stdout);
}

template <int PAT, int AL, int NT, int DEPTH>
void run(const char* name, int N, int K, int TR, int grid) {
    const int units = N / (NT * TR);
    const size_t bytes = (size_t)units * NT * TR * K * 2;
    const size_t stride = (bytes + (1 << 21)) & ~((size_t)(1 << 21) - 1);
    const int regions = (int)(g_big_bytes / stride);
    if (grid <= 0 || grid > units) grid = units;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&](int i) {
        P p{g_big + (size_t)(i % regions) * stride, g_a, g_out, K, TR, units};
        stream_kernel<PAT, AL, NT, DEPTH><<<grid, 512>>>(p);
    };
    for (int i = 0; i < 8; ++i) launch(i);
    (void)hipDeviceSynchronize();
    const int reps = 40;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(i + 8);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-10s N=%6d K=%5d NT=%d TR=%2d grid=%4d depth=%d pat=%s A=%s : %7.2f us  %5.2f TB/s\n", name, N, K, NT, TR, grid, DEPTH,
           PAT ? "frag" : "rows", AL == 0 ? "lds " : (AL == 1 ? "rows" : (AL == 2 ? "frag" : "frag, ONE load per step")), us, bytes / us / 1e6);
    fflush(stdout);
}

int main() {
    g_big_bytes = (size_t)6 << 30;
    if (hipMalloc(&g_big, g_big_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&g_a, 8 * 18944 * 2 + 65536); (void)hipMalloc(&g_out, 4096 * 512 * 4);
    (void)hipMemset(g_big, 0x3c, g_big_bytes); (void)hipMemset(g_a, 0x3c, 8 * 18944 * 2 + 65536);
    // fill with non-trivial data (the clock under load depends on it): xorshift on the host for 64 MiB, replicated
    {
        const size_t n = (size_t)64 << 20;
        uint16_t* h = (uint16_t*)malloc(n);
        uint32_t s = 12345u;
        for (size_t i = 0; i < n / 2; ++i) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; h[i] = (uint16_t)(0x3c00u | (s & 0x83ffu)); }
        for (size_t off = 0; off < g_big_bytes; off += n) (void)hipMemcpy(g_big + off, h, n, hipMemcpyHostToDevice);
        (void)hipMemcpy(g_a, h, 8 * 18944 * 2, hipMemcpyHostToDevice);
        free(h);
    }
    for (int rep = 0; rep < 2; ++rep) {
        printf("---- pass %d\n", rep);
        run<0, 1, 1, 2>("down", 3584, 18944, 14, 256);
        run<0, 2, 1, 2>("down", 3584, 18944, 14, 256);
        run<0, 2, 1, 4>("down", 3584, 18944, 14, 256);
        run<0, 2, 1, 7>("down", 3584, 18944, 14, 256);
        run<0, 3, 1, 7>("down", 3584, 18944, 14, 256);
        run<0, 3, 1, 2>("down", 3584, 18944, 14, 256);
        run_cw<7>("down", 3584, 18944, 14, 256);
        run_cw<4>("down", 3584, 18944, 14, 256);
        run<0, 0, 1, 4>("down", 3584, 18944, 14, 256);
        run<0, 1, 1, 2>("o", 3584, 3584, 14, 256);
        run<0, 2, 1, 2>("o", 3584, 3584, 14, 256);
        run<0, 2, 1, 7>("o", 3584, 3584, 14, 256);
        run<0, 3, 1, 7>("o", 3584, 3584, 14, 256);
        run_cw<7>("o", 3584, 3584, 14, 256);
        run<0, 0, 1, 7>("o", 3584, 3584, 14, 256);
        run<0, 1, 2, 2>("qkv", 4608, 3584, 9, 256);
        run<0, 2, 2, 4>("qkv", 4608, 3584, 9, 256);
        run<0, 2, 2, 7>("qkv", 4608, 3584, 9, 256);
        run<0, 3, 2, 7>("qkv", 4608, 3584, 9, 256);
        run<0, 3, 2, 4>("qkv", 4608, 3584, 9, 256);
        run<0, 0, 2, 7>("qkv", 4608, 3584, 9, 256);
        run<0, 2, 2, 4>("gateup", 37890, 3584, 15, 256);
        run<0, 3, 2, 4>("gateup", 37890, 3584, 15, 256);
        run<0, 0, 2, 4>("gateup", 37890, 3584, 15, 256);
        run<0, 2, 4, 2>("lm_head", 160512, 3584, 16, 256);
        run<0, 0, 4, 2>("lm_head", 160512, 3584, 16, 256);
    }
    return 0;
}
