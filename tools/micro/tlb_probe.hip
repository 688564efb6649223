// Is the ramp of a short weight-streaming launch (decode: 7-30 us kernels that each touch fresh pages) address translation?
// A "phase" kernel: 256 workgroups x 512 threads, each workgroup streams its own contiguous share of a region that has not been touched
// for > 1 GB of traffic (TLBs and caches cold).  Variants:
//   cold      nothing in front
//   touch     a tiny kernel in front reads ONE cache line per 64 KiB of the region (from every XCD): the translations are fresh, the data is not
//   warm      the same region streamed by the previous launch as well (translations AND the Infinity Cache warm): the floor
//   window    only the first 64 / 256 KiB of every workgroup's share read in front (by the same workgroup index), optionally with 64 MB of other
//             traffic between the two: what a tail prefetch by the previous phase could give
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/tlb_probe.hip -o /tmp/tlb_probe && /tmp/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void phase(const char* base, size_t share, float* out) {
    const char* p = base + (size_t)blockIdx.x * share;
    u32x4 acc = {0, 0, 0, 0};
    // 512 threads x 16 B = 8 KiB per step, 8 steps in flight
    for (size_t off = (size_t)threadIdx.x * 16; off < share; off += 8 * 8192) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (off + (size_t)i * 8192 < share) ? *reinterpret_cast<const u32x4*>(p + off + (size_t)i * 8192) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] + acc[1] + acc[2] + acc[3]);
}
// one line per `stride` bytes of [base, base + bytes), every workgroup walks the whole region (so every XCD's translation cache sees every page)
__global__ __launch_bounds__(256) void touch(const char* base, size_t bytes, size_t stride, float* out) {
    uint32_t a = 0;
    for (size_t off = (size_t)threadIdx.x * stride; off < bytes; off += 256 * stride) a += *reinterpret_cast<const uint32_t*>(base + off);
    if (a == 0x12345678u) out[blockIdx.x] = 1.f;
}
// workgroup i reads the first `win` bytes of share i (the bytes its first load window will ask for): same blockIdx -> same XCD as the phase kernel
__global__ __launch_bounds__(512) void touch_window(const char* base, size_t share, size_t win, float* out) {
    const char* p = base + (size_t)blockIdx.x * share;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t off = (size_t)threadIdx.x * 16; off < win; off += 8192) acc += *reinterpret_cast<const u32x4*>(p + off);
    if (acc[0] == 0x12345678u) out[blockIdx.x] = 1.f;
}
__global__ __launch_bounds__(512) void thrash(const char* base, size_t bytes, float* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t off = ((size_t)blockIdx.x * 512 + threadIdx.x) * 16; off < bytes; off += (size_t)gridDim.x * 8192) acc += *reinterpret_cast<const u32x4*>(base + off);
    out[blockIdx.x * 512 + threadIdx.x] = (float)acc[0];
}

int main() {
    const size_t total = (size_t)12 << 30;
    char* buf; float* out;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 1024 * 512 * 4);
    (void)hipMemset(buf, 1, total);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t thr_bytes = (size_t)4 << 30;                  // traffic between two measurements: evicts caches and translations
    struct Case { const char* name; size_t region; };
    const Case cases[] = {{"o / q|k|v-sized (26 MB)", (size_t)26 << 20}, {"down-sized (136 MB)", (size_t)136 << 20}, {"gate/up-sized (272 MB)", (size_t)272 << 20}};
    for (const Case& cs : cases) {
        const size_t share = (cs.region / 256) & ~(size_t)8191;
        for (int mode = 0; mode < 8; ++mode) {
            std::vector<float> us;
            for (int rep = 0; rep < 7; ++rep) {
                const char* region = buf + ((size_t)5 << 30) + (size_t)rep * ((size_t)512 << 20);
                thrash<<<1024, 512>>>(buf, thr_bytes, out);
                if (mode == 1) touch<<<64, 256>>>(region, share * 256, (size_t)65536, out);
                if (mode == 2) touch<<<64, 256>>>(region, share * 256, (size_t)2 << 20, out);
                if (mode == 3) phase<<<256, 512>>>(region, share, out);
                if (mode == 4 || mode == 5) touch_window<<<256, 512>>>(region, share, (size_t)65536, out);
                if (mode == 6 || mode == 7) touch_window<<<256, 512>>>(region, share, (size_t)262144 < share ? (size_t)262144 : share, out);
                if (mode == 5 || mode == 7) phase<<<256, 512>>>(buf + ((size_t)10 << 30), ((size_t)64 << 20) / 256, out);     // 64 MB of other traffic in between
                (void)hipEventRecord(e0);
                phase<<<256, 512>>>(region, share, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                us.push_back(ms * 1e3f);
            }
            std::sort(us.begin(), us.end());
            const char* mn[] = {"cold", "one line per 64 KiB touched in front", "one line per 2 MiB touched in front", "streamed by the previous launch too",
                                "first 64 KiB of every share read in front", "... then 64 MB of other traffic", "first 256 KiB of every share read in front", "... then 64 MB of other traffic"};
            printf("%-26s %-40s median %7.2f us  min %7.2f  (%.2f TB/s at the median)\n", cs.name, mn[mode], us[us.size() / 2], us[0], share * 256 / us[us.size() / 2] / 1e6);
            fflush(stdout);
        }
    }
    return 0;
}
