// Cost of a grid-wide barrier on MI355X (256 workgroups of 512 threads, one per CU, 8 XCDs): monotone counter in global memory,
// agent-scope release before the arrive, agent-scope acquire after the wait; every spin is bounded.  Variants: (a) fences only,
// (b) each workgroup also writes 2 KB before the barrier and reads another workgroup's 2 KB after it (a hand-off like the
// activations between two decode GEMMs).  Build + run: hipcc -O2 --offload-arch=gfx950 tools/micro/grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>

template <int FENCE, int SLEEP>
__global__ __launch_bounds__(512) void k_barrier(unsigned* cnt, float* buf, int iters, int handoff, unsigned* err) {
    const int tid = threadIdx.x, nb = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (handoff) buf[(size_t)blockIdx.x * 512 + tid] = (float)(it + blockIdx.x) + acc * 1e-9f;
        __syncthreads();
        if (tid == 0) {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * (unsigned)nb;
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (SLEEP) __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) { *err = 1; break; }          // bounded: never hang the GPU
            }
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (handoff) acc += buf[(size_t)((blockIdx.x + 37) % nb) * 512 + tid];
    }
    if (handoff) buf[(size_t)nb * 512 + (size_t)blockIdx.x * 512 + tid] = acc;
}

template <int FENCE, int SLEEP> void run(unsigned* cnt, unsigned* err, float* buf, int handoff) {
    float best = 1e9f; unsigned h_err = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(cnt, 0, 4); (void)hipMemset(err, 0, 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 2000;
        (void)hipEventRecord(e0);
        k_barrier<FENCE, SLEEP><<<256, 512>>>(cnt, buf, iters, handoff, err);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        (void)hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
    }
    printf("fences %d sleep %d handoff %d: %.2f us per grid barrier (256 workgroups x 512 threads), err %u\n", FENCE, SLEEP, handoff, best * 1e3 / 2000, h_err);
}
int main() {
    unsigned *cnt, *err; float* buf;
    (void)hipMalloc(&cnt, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&buf, 2 * 256 * 512 * 4);
    run<0, 0>(cnt, err, buf, 0); run<0, 1>(cnt, err, buf, 0); run<1, 0>(cnt, err, buf, 0); run<1, 1>(cnt, err, buf, 0);
    run<1, 0>(cnt, err, buf, 1); run<1, 1>(cnt, err, buf, 1);
    return 0;
}
