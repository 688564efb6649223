// What does the matrix pipe sustain at the board's power cap, and does the MFMA shape or the LDS operand traffic per FLOP move it?
// ("one energy-per-FLOP experiment, costed before it is built".)  Each case loops ~3 s on every CU with
// pseudo-random bf16 operands while a host thread samples the card's own hwmon files; a case "wins" only if its TFLOP/s AT THE CAP is
// higher, i.e. W per TFLOP/s is lower.
//   reg16      v_mfma_f32_16x16x32_bf16, 8 x 4 tiles per wave (128 x 64), operands held in registers: the pipe alone
//   reg32      v_mfma_f32_32x32x16_bf16, 4 x 2 tiles per wave (128 x 64), operands in registers (half the operand-register reads per FLOP)
//   lds16      reg16 + the fragment reads a 128 x 64 wave tile needs: 12 ds_read_b128 per 32 MFMAs (1/43 operand element per FLOP: gemm_pp.hip)
//   lds32      reg32 + the same reads: 12 ds_read_b128 per 16 MFMAs
//   lds32w     32x32x16, 4 x 4 tiles per wave (128 x 128, 4 waves per CU, accumulators in 256 AGPRs): 16 reads per 32 MFMAs (1/64 per FLOP)
//   lds16w     16x16x32, 8 x 8 tiles per wave (128 x 128, 4 waves per CU, 256 AGPRs), next step's 16 fragments read under this step's 64 MFMAs
//   dma16w     lds16w + the same operand traffic as dma16
//   dma16      lds16 + the global -> LDS operand traffic of gemm_pp.hip (64 KiB per 512 MFMAs of the workgroup, LDS-DMA) from an L2-resident window
// Build + run:  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_energy.hip -o /tmp/mfma_energy -lpthread && /tmp/mfma_energy
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
#include <limits.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct P { const uint32_t* src; float* out; int iters; };

__device__ __forceinline__ u32x4 rnd4(uint32_t& s) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        // two bf16 in [-2, 2) with random mantissas and signs: sign | exponent 0x3f / 0x3e | mantissa
        v[i] = (s & 0x80ff80ffu) | 0x3f003e00u;
    }
    return v;
}

// MODE 0 reg16, 1 reg32, 2 lds16, 3 lds32, 5 dma16 : 512 threads
template <int MODE>
__global__ __launch_bounds__(512) void k_a(P p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 512 + tid + 1);
    constexpr int LDSB = 65536;
    for (int i = tid; i < LDSB / 16; i += 512) reinterpret_cast<u32x4*>(lds)[i] = rnd4(s);
    __syncthreads();
    u32x4 fr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) fr[i] = rnd4(s);
    constexpr bool L = MODE == 2 || MODE == 3 || MODE == 5;
    if constexpr (MODE == 0 || MODE == 2 || MODE == 5) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < p.iters; ++it) {
            if constexpr (L) {
                const int base = ((it * 12) & 63) * 1024 + lane * 16;
#pragma unroll
                for (int i = 0; i < 12; ++i) fr[i] = *reinterpret_cast<const u32x4*>(lds + ((base + i * 1024 + wave * 2048) & (LDSB - 1)));
            }
            if constexpr (MODE == 5) {
                // gemm_pp.hip moves one 64-KiB operand stage (256 x 64 of A and of W) per 64 MFMAs of each of its 8 waves: 4 KiB per wave per
                // 32 MFMAs = four 1-KiB LDS-DMA loads per iteration, here from a 2-MiB window that stays in L2
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t* g = p.src + ((((size_t)blockIdx.x * 8 + wave) * 1024 + (size_t)(it & 63) * 8192 + i * 256 + lane * 4) & ((1u << 19) - 1));
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(lds + LDSB + wave * 4096 + i * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[i]), __builtin_bit_cast(bf16x8, fr[8 + j]), acc[i][j], 0, 0, 0);
        }
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        p.out[blockIdx.x * 512 + tid] = v;
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < p.iters; ++it) {
            // the same K = 32 per iteration as the 16x16x32 cases: two k-steps of 16, six fragments each
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (L) {
                    const int base = ((it * 12 + h * 6) & 63) * 1024 + lane * 16;
#pragma unroll
                    for (int i = 0; i < 6; ++i) fr[h * 6 + i] = *reinterpret_cast<const u32x4*>(lds + ((base + i * 1024 + wave * 2048) & (LDSB - 1)));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[h * 6 + i]), __builtin_bit_cast(bf16x8, fr[h * 6 + 4 + j]), acc[i][j], 0, 0, 0);
            }
        }
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) v += acc[i][j][e];
        p.out[blockIdx.x * 512 + tid] = v;
    }
}

// MODE 4 lds32w: 256 threads, one wave per SIMD, 128 x 128 per wave
__global__ __launch_bounds__(256) void k_w(P p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 256 + tid + 1);
    constexpr int LDSB = 65536;
    for (int i = tid; i < LDSB / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = rnd4(s);
    __syncthreads();
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < p.iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 fr[8];
            const int base = ((it * 16 + h * 8) & 63) * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) fr[i] = *reinterpret_cast<const u32x4*>(lds + ((base + i * 1024 + wave * 4096) & (LDSB - 1)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i]), __builtin_bit_cast(bf16x8, fr[4 + j]), acc[i][j], 0, 0, 0);
        }
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) v += acc[i][j][e];
    p.out[blockIdx.x * 256 + tid] = v;
}

// MODE 6 lds16w / 7 dma16w: 256 threads, one wave per SIMD, 128 x 128 per wave out of 8 x 8 tiles of 16x16x32 (accumulators in 256 AGPRs),
// 16 fragment reads per 64 MFMAs (1/64 operand element per FLOP), the next K step's fragments read while this one's MFMAs run
template <bool DMA>
__global__ __launch_bounds__(256) void k_w16(P p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 256 + tid + 1);
    constexpr int LDSB = 65536;
    for (int i = tid; i < LDSB / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = rnd4(s);
    __syncthreads();
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 cur[16], nxt[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cur[i] = rnd4(s);
    for (int it = 0; it < p.iters; ++it) {
        const int base = ((it * 16) & 63) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) nxt[i] = *reinterpret_cast<const u32x4*>(lds + ((base + i * 1024 + wave * 4096) & (LDSB - 1)));
        if constexpr (DMA) {
            // 64 KiB per 256 x 256 x 64 step of the workgroup = per wave (4 of them) 8 KiB per 64 MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t* g = p.src + ((((size_t)blockIdx.x * 4 + wave) * 2048 + (size_t)(it & 63) * 8192 + i * 256 + lane * 4) & ((1u << 19) - 1));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds + LDSB + wave * 8192 + i * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[i]), __builtin_bit_cast(bf16x8, cur[8 + j]), acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) v += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    p.out[blockIdx.x * 256 + tid] = v;
}

// ---- the card's own sensors ------------------------------------------------------------------------------------------------------
static std::string g_freq, g_power;
static bool readable(const std::string& f) { FILE* h = fopen(f.c_str(), "r"); if (!h) return false; char b[64]; bool ok = fgets(b, 64, h) != nullptr; fclose(h); return ok; }
static long long read_ll(const std::string& f) { FILE* h = fopen(f.c_str(), "r"); if (!h) return -1; long long v = -1; if (fscanf(h, "%lld", &v) != 1) v = -1; fclose(h); return v; }
static void find_sensors() {
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), 0) != hipSuccess) return;
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    DIR* d = opendir("/sys/class/drm");
    if (!d) return;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
        const std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
        char real[PATH_MAX];
        if (!realpath(dev.c_str(), real)) continue;
        const char* base = strrchr(real, '/');
        if (!base || strcmp(base + 1, bdf) != 0) continue;
        const std::string hw = dev + "/hwmon";
        DIR* h = opendir(hw.c_str());
        if (!h) continue;
        while (dirent* x = readdir(h)) {
            if (strncmp(x->d_name, "hwmon", 5) != 0) continue;
            const std::string b = hw + "/" + x->d_name + "/";
            if (readable(b + "freq1_input")) g_freq = b + "freq1_input";
            if (readable(b + "power1_average")) g_power = b + "power1_average";
            else if (readable(b + "power1_input")) g_power = b + "power1_input";
            if (readable(b + "power1_cap")) printf("power cap W: %.0f\n", read_ll(b + "power1_cap") / 1e6);
        }
        closedir(h);
        printf("cuda:0 is PCI %s -> %s\n", bdf, e->d_name);
    }
    closedir(d);
}

template <typename F>
static void run_case(const char* name, double flop_per_launch, F launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    std::atomic<bool> stop{false};
    std::vector<double> mhz, w;
    std::thread th([&] {
        while (!stop.load()) {
            if (!g_freq.empty()) { const long long v = read_ll(g_freq); if (v > 0) mhz.push_back(v / 1e6); }
            if (!g_power.empty()) { const long long v = read_ll(g_power); if (v > 0) w.push_back(v / 1e6); }
            usleep(20000);
        }
    });
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    (void)hipEventRecord(e0);
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) {
        for (int i = 0; i < 4; ++i) launch();
        (void)hipDeviceSynchronize();
        n += 4;
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    stop.store(true); th.join();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; v.erase(v.begin(), v.begin() + v.size() / 4); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double tf = flop_per_launch * n / (ms * 1e-3) / 1e12, m = med(mhz), pw = med(w);
    printf("%-8s %7.0f TFLOP/s | clock MHz %5.0f (peak at that clock %5.0f TF, frac %.3f) | power W %5.0f | W per TFLOP/s %.3f\n", name, tf, m,
           2500.0 * m / 2400.0, m > 0 ? tf / (2500.0 * m / 2400.0) : 0.0, pw, tf > 0 ? pw / tf : 0.0);
    fflush(stdout);
}

int main() {
    find_sensors();
    printf("sysfs: %s %s\n", g_freq.c_str(), g_power.c_str());
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    uint32_t* src; float* out;
    (void)hipMalloc(&src, 8u << 20); (void)hipMalloc(&out, (size_t)cus * 512 * 4);
    {
        std::vector<uint32_t> h((8u << 20) / 4);
        uint32_t s = 777u;
        for (auto& v : h) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; v = (s & 0x80ff80ffu) | 0x3f003e00u; }
        (void)hipMemcpy(src, h.data(), 8u << 20, hipMemcpyHostToDevice);
    }
    const int iters = 20000;
    P p{src, out, iters};
    (void)hipFuncSetAttribute((const void*)k_a<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 32768);
    const double f512 = 2.0 * 16 * 16 * 32 * 32 * 8.0 * cus * iters;        // 32 MFMAs of 16x16x32 per wave-iteration, 8 waves
    const double f256 = 2.0 * 32 * 32 * 16 * 32 * 4.0 * cus * iters;        // 32 MFMAs of 32x32x16 per wave-iteration, 4 waves
    const double f16w = 2.0 * 16 * 16 * 32 * 64 * 4.0 * cus * iters;        // 64 MFMAs of 16x16x32 per wave-iteration, 4 waves
    (void)hipFuncSetAttribute((const void*)k_w16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 32768);
    for (int pass = 0; pass < 2; ++pass) {
        printf("---- pass %d\n", pass);
        run_case("reg16", f512, [&] { k_a<0><<<cus, 512, 65536>>>(p); });
        run_case("reg32", f512, [&] { k_a<1><<<cus, 512, 65536>>>(p); });
        run_case("lds16", f512, [&] { k_a<2><<<cus, 512, 65536>>>(p); });
        run_case("lds32", f512, [&] { k_a<3><<<cus, 512, 65536>>>(p); });
        run_case("lds32w", f256, [&] { k_w<<<cus, 256, 65536>>>(p); });
        run_case("dma16", f512, [&] { k_a<5><<<cus, 512, 65536 + 32768>>>(p); });
        run_case("lds16w", f16w, [&] { k_w16<false><<<cus, 256, 65536>>>(p); });
        run_case("dma16w", f16w, [&] { k_w16<true><<<cus, 256, 65536 + 32768>>>(p); });
    }
    return 0;
}
