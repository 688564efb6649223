// Does kernarg preloading (user SGPRs filled by the command processor, gfx940+) shorten a dependent kernel boundary?
// The decode step has 171 launches whose first useful instruction (a weight load) needs a pointer from the kernel arguments.
// Build twice and run both:  hipcc -O3 --offload-arch=gfx950 -DTAG='"default"' tools/micro/kernarg_preload.hip -o /tmp/kp_default
//                            hipcc -O3 --offload-arch=gfx950 -DTAG='"preload"' -mllvm -amdgpu-kernarg-preload-count=16 tools/micro/kernarg_preload.hip -o /tmp/kp_preload
// Round 4, one MI355X: 1.61 us per dependent launch in a replayed graph without, 1.70 us WITH preloading (a struct passed by value is never preloaded: 1.60).
#include <hip/hip_runtime.h>
#include <cstdio>
struct S { const float* in; float* out; int n; int pad[13]; };
__global__ __launch_bounds__(512) void k_args(const float* in, float* out, int n) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
__global__ __launch_bounds__(512) void k_struct(S s) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    if (i < s.n) s.out[i] = s.in[i] + 1.0f;
}
template <typename F> static double run(F launch, int n) {
    hipStream_t st; (void)hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(st, i);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0, st); (void)hipGraphLaunch(ge, st); (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms * 1e3 / n < best) best = ms * 1e3 / n;
    }
    return best;
}
int main() {
    const int n = 256 * 512;
    float *a, *b; (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMemset(a, 0, n * 4); (void)hipMemset(b, 0, n * 4);
    for (int pass = 0; pass < 2; ++pass) {
        const double t1 = run([&](hipStream_t st, int i) { k_args<<<256, 512, 0, st>>>((i & 1) ? b : a, (i & 1) ? a : b, n); }, 2000);
        const double t2 = run([&](hipStream_t st, int i) { S s{}; s.in = (i & 1) ? b : a; s.out = (i & 1) ? a : b; s.n = n; k_struct<<<256, 512, 0, st>>>(s); }, 2000);
        printf("%s: scalar arguments %.3f us per dependent launch, one struct by value %.3f us\n", TAG, t1, t2);
    }
    return 0;
}
