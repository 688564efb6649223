// Whisper log-mel front end on gfx950: wav [B, <=480000] f32 -> log-mel [B,128,3000] / [B,3000,128].
//
// Pass 1 (one workgroup = 64 frames of one clip, 4 waves):
//   * the 64*160+240 samples the frames touch are staged once in LDS (reflect padding at the clip edges,
//     zero beyond n_samples), one hop per LDS row padded to 161 floats so that "same sample offset, 32
//     consecutive frames" (the MFMA A-operand pattern) hits 32 different banks;
//   * the 400-point real DFT is folded on the Hann window's symmetry (w[n] == w[400-n], w[0] == 0):
//         Re X[k] =  sum_{n=1..200} w[n] (x[n] + [n<200] x[400-n]) cos(2 pi k n / 400)
//         Im X[k] = -sum_{n=1..199} w[n] (x[n] - x[400-n])         sin(2 pi k n / 400)
//     i.e. two [64 x 208] . [208 x 224] products in exact f32 on v_mfma_f32_32x32x2_f32; the folded operand
//     is built in registers from LDS, the cos / sin tables ([bin][n], 8 consecutive n per lane) stream from L2;
//   * power -> LDS -> banded mel filter bank (394 non-zeros) -> log10 -> f32 scratch [B,3000,128]
//     + per-clip running max (ordered-int atomicMax).
// Pass 2: max(x, clipmax - 8), (x + 4) / 4, cast / transpose into the caller's layout.
// Replaces transformers feature_extraction_whisper.py:135-170 (called from audio.py:1056-1069).
#include "common.h"
#include <math.h>
#include <string.h>

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NMEL = 128, NFRAMES = 3000, NSAMP = 480000;
constexpr int FT = 64;                    // frames per workgroup
constexpr int KPAD = 208;                 // folded K (200 / 199) padded to 13 steps of 16
constexpr int BINPAD = 224;               // 201 bins padded to 7 tiles of 32
constexpr int HOPROW = HOP + 1;           // LDS row pitch (floats)
constexpr int NHOPROWS = FT + 3;          // hops touched by 64 frames: 63 + ceil(400/160) = 66 (+1 slack)
constexpr int PROW = BINPAD + 1;          // LDS pitch of the power tile
constexpr int MAIN_FLOATS = (NHOPROWS * HOPROW > FT * PROW) ? NHOPROWS * HOPROW : FT * PROW;

// device-side constant tables (one buffer, built by afhip_log_mel_tables_host)
struct Tables {
    // offsets in floats
    static constexpr size_t COS = 0;                                  // [BINPAD][KPAD]
    static constexpr size_t SIN = COS + (size_t)BINPAD * KPAD;        // [BINPAD][KPAD] (already negated)
    static constexpr size_t WIN = SIN + (size_t)BINPAD * KPAD;        // [KPAD] window w[n], n = i+1 (0 beyond 200)
    static constexpr size_t FILT = WIN + KPAD;                        // [NBIN][NMEL] dense bank (kept for reference / tests)
    static constexpr size_t BAND = FILT + (size_t)NBIN * NMEL;        // [NMEL][2] int32 (first bin, count)
    static constexpr size_t COFF = BAND + 2 * NMEL;                   // [NMEL] int32 offset of the mel's weights in CW
    static constexpr size_t CW = COFF + NMEL;                         // [CWMAX] non-zero weights, mel-major
    static constexpr size_t CWMAX = 512;
    static constexpr size_t TOTAL = CW + CWMAX;
};

__device__ __forceinline__ int float_order_key(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float float_from_key(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void logmel_pass1(const float* __restrict__ wav, int n_samples, long long wav_stride,
                                                    const float* __restrict__ tab, float* __restrict__ scratch,
                                                    int* __restrict__ clipmax) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);          // [NHOPROWS][HOPROW] samples, later [FT][PROW] power
    float* cw = xs + MAIN_FLOATS;                            // [CWMAX] compact mel weights
    int* cband = reinterpret_cast<int*>(cw + Tables::CWMAX); // [NMEL][3] first bin, count, offset
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FT;
    const float* w = wav + (long long)b * wav_stride;

    for (int i = tid; i < (int)Tables::CWMAX; i += 256) cw[i] = tab[Tables::CW + i];
    for (int i = tid; i < NMEL; i += 256) {
        const int* band_g = reinterpret_cast<const int*>(tab + Tables::BAND);
        cband[3 * i] = band_g[2 * i];
        cband[3 * i + 1] = band_g[2 * i + 1];
        cband[3 * i + 2] = reinterpret_cast<const int*>(tab + Tables::COFF)[i];
    }
    // ---- stage samples: padded index p = t0*160 + s, original index i = p - 200 (reflect at both clip ends, zero
    //      beyond n_samples).  All loads of a thread are issued before the first LDS store (11 x 16 B in flight per lane);
    //      workgroups that touch neither clip end nor the zero-padded tail take aligned float4 loads. ----
    constexpr int NSTAGE = (FT - 1) * HOP + NFFT;          // 10480 samples = 2620 float4
    constexpr int NV4 = NSTAGE / 4, V4_PER_THREAD = (NV4 + 255) / 256;
    const int ibase = t0 * HOP - NFFT / 2;                  // multiple of 8: float4 loads stay 16-byte aligned
    const bool interior = ibase >= 0 && ibase + NSTAGE <= n_samples && (n_samples <= NSAMP) && ((wav_stride & 3) == 0) &&
                          ((reinterpret_cast<uintptr_t>(wav) & 15) == 0);
    f32x4 sv[V4_PER_THREAD];
#pragma unroll
    for (int u = 0; u < V4_PER_THREAD; ++u) {
        const int v = tid + 256 * u;
        sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (v < NV4) {
            if (interior) {
                sv[u] = *reinterpret_cast<const f32x4*>(w + ibase + 4 * v);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int i = ibase + 4 * v + e;
                    if (i < 0) i = -i;
                    if (i >= NSAMP) i = 2 * (NSAMP - 1) - i;
                    if (i >= 0 && i < n_samples) sv[u][e] = w[i];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < V4_PER_THREAD; ++u) {
        const int v = tid + 256 * u;
        if (v < NV4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sidx = 4 * v + e;
                xs[(sidx / HOP) * HOPROW + (sidx % HOP)] = sv[u][e];
            }
        }
    }
    __syncthreads();

    // ---- folded DFT on MFMA: this wave owns bin tiles j = wave and wave + 4 (7 tiles in all) ----
    const int nj = (wave + 4 < 7) ? 2 : 1;
    const float* cosT = tab + Tables::COS;
    const float* sinT = tab + Tables::SIN;
    const float* win = tab + Tables::WIN;
    f32x16 re[2][2], im[2][2];  // [m tile][bin tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) { re[a][c][e] = 0.f; im[a][c][e] = 0.f; }

    // twiddle fragments (B operand): row = bin, 8 consecutive n per lane.  They stream from L2 and the wave is alone on
    // its SIMD (the accumulators take most of the register file), so the fragments of step s+1 are fetched into a second
    // register set while the MFMAs of step s run.
    f32x8 fc[2], fs[2], nfc[2], nfs[2];
    auto load_tw = [&](int step, f32x8 (&c8)[2], f32x8 (&s8)[2]) {
        const int ii = step * 16 + fh * 8;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int bin = (wave + 4 * c) * 32 + fr;
            if (c < nj) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cosT + (size_t)bin * KPAD + ii);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(cosT + (size_t)bin * KPAD + ii + 4);
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sinT + (size_t)bin * KPAD + ii);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(sinT + (size_t)bin * KPAD + ii + 4);
                c8[c] = f32x8{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                s8[c] = f32x8{s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            } else {
                c8[c] = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
                s8[c] = c8[c];
            }
        }
    };
    load_tw(0, fc, fs);
#pragma unroll 1
    for (int step = 0; step < KPAD / 16; ++step) {
        const int i0 = step * 16 + fh * 8;             // first folded index of this lane's 8 elements
        if (step + 1 < KPAD / 16) load_tw(step + 1, nfc, nfs);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(win + i0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(win + i0 + 4);
        const float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int frow = mt * 32 + fr;               // frame inside the workgroup
            f32x8 fe, fo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = i0 + j + 1;                // 1..208
                const int nn = n <= 200 ? n : 200;       // clamp padded tail (window is 0 there)
                const int m = NFFT - nn;                 // mirrored sample
                const float xa = xs[(frow + nn / HOP) * HOPROW + nn % HOP];
                const float xb = xs[(frow + m / HOP) * HOPROW + m % HOP];
                const float xb_e = nn < 200 ? xb : 0.f;
                fe[j] = wv[j] * (xa + xb_e);
                fo[j] = nn < 200 ? wv[j] * (xa - xb) : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c < nj) {
                    re[mt][c] = mma16(fe, fc[c], re[mt][c]);
                    im[mt][c] = mma16(fo, fs[c], im[mt][c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) { fc[c] = nfc[c]; fs[c] = nfs[c]; }
    }
    __syncthreads();   // everyone is done with the samples: reuse LDS for the power tile

    float* pw = xs;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c < nj) {
                const int bin = (wave + 4 * c) * 32 + fr;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int f = mt * 32 + mfma32_row(e, lane);
                    const float r = re[mt][c][e], q = im[mt][c][e];
                    pw[f * PROW + bin] = r * r + q * q;
                }
            }
        }
    __syncthreads();

    // ---- banded mel + log10 (weights and band table in LDS: no dependent global loads); scratch is [B][3000][128] f32 ----
    float mx = -INFINITY;
    {
        const int m = tid & 127;
        const int k0 = cband[3 * m], kc = cband[3 * m + 1], wo = cband[3 * m + 2];
        for (int f = tid >> 7; f < FT; f += 2) {
            const int t = t0 + f;
            if (t >= NFRAMES) break;
            float acc = 0.f;
            for (int k = 0; k < kc; ++k) acc += cw[wo + k] * pw[f * PROW + k0 + k];
            const float v = log10f(fmaxf(acc, 1e-10f));
            scratch[((long long)b * NFRAMES + t) * NMEL + m] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = wave_max(mx);
    if (lane == 0 && mx > -INFINITY) atomicMax(clipmax + b, float_order_key(mx));
}

__global__ void logmel_init_max(int* clipmax, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) clipmax[i] = float_order_key(-INFINITY);
}

// Pass 2: one workgroup = 32 frames x 128 mels of one clip
template <typename TO, int LAYOUT>
__global__ __launch_bounds__(256) void logmel_pass2(const float* __restrict__ scratch, const int* __restrict__ clipmax,
                                                    TO* __restrict__ out) {
    __shared__ float tile[32][NMEL + 1];
    const int b = blockIdx.y, t0 = blockIdx.x * 32, tid = threadIdx.x;
    const float floorv = float_from_key(clipmax[b]) - 8.0f;
    for (int idx = tid; idx < 32 * NMEL; idx += 256) {
        const int f = idx >> 7, m = idx & 127;
        const int t = t0 + f;
        if (t < NFRAMES) {
            float v = scratch[((long long)b * NFRAMES + t) * NMEL + m];
            v = (fmaxf(v, floorv) + 4.0f) / 4.0f;
            if (LAYOUT == 1) out[((long long)b * NFRAMES + t) * NMEL + m] = from_f32<TO>(v);
            else tile[f][m] = v;
        }
    }
    if (LAYOUT == 0) {
        __syncthreads();
        for (int idx = tid; idx < 32 * NMEL; idx += 256) {
            const int m = idx >> 5, f = idx & 31;
            const int t = t0 + f;
            if (t < NFRAMES) out[((long long)b * NMEL + m) * NFRAMES + t] = from_f32<TO>(tile[f][m]);
        }
    }
}

}  // namespace

extern "C" size_t afhip_log_mel_tables_bytes(void) { return Tables::TOTAL * sizeof(float); }

// Fills a HOST buffer with the constant tables (computed in double, rounded to f32); the caller uploads it once.
// filters_host: [201,128] f32 mel filter bank (transformers/audio_utils.py:638-729).
extern "C" int afhip_log_mel_tables_host(void* host_buf, const float* filters_host) {
    AFHIP_CHECK(host_buf && filters_host, "afhip_log_mel_tables_host: null pointer");
    float* t = reinterpret_cast<float*>(host_buf);
    memset(t, 0, Tables::TOTAL * sizeof(float));
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < NBIN; ++k)
        for (int i = 0; i < 200; ++i) {
            const int n = i + 1;
            const long long kn = ((long long)k * n) % NFFT;   // exact angle reduction
            t[Tables::COS + (size_t)k * KPAD + i] = (float)cos(two_pi * (double)kn / NFFT);
            if (n < 200) t[Tables::SIN + (size_t)k * KPAD + i] = (float)(-sin(two_pi * (double)kn / NFFT));
        }
    for (int i = 0; i < 200; ++i) t[Tables::WIN + i] = (float)(0.5 - 0.5 * cos(two_pi * (double)(i + 1) / NFFT));
    memcpy(t + Tables::FILT, filters_host, sizeof(float) * NBIN * NMEL);
    int* band = reinterpret_cast<int*>(t + Tables::BAND);
    for (int m = 0; m < NMEL; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < NBIN; ++k)
            if (filters_host[k * NMEL + m] != 0.f) { if (lo < 0) lo = k; hi = k; }
        band[2 * m] = lo < 0 ? 0 : lo;
        band[2 * m + 1] = lo < 0 ? 0 : hi - lo + 1;
    }
    int* coff = reinterpret_cast<int*>(t + Tables::COFF);
    int off = 0;
    for (int m = 0; m < NMEL; ++m) {
        coff[m] = off;
        for (int k = 0; k < band[2 * m + 1]; ++k) {
            AFHIP_CHECK(off < (int)Tables::CWMAX, "afhip_log_mel_tables_host: filter bank has more than %d band entries", (int)Tables::CWMAX);
            t[Tables::CW + off++] = filters_host[(band[2 * m] + k) * NMEL + m];
        }
    }
    return 0;
}

extern "C" size_t afhip_log_mel_workspace_bytes(int B) {
    if (B <= 0) return 0;
    return (size_t)B * NFRAMES * NMEL * sizeof(float) + ((size_t)B * sizeof(int) + 255) / 256 * 256;
}

extern "C" int afhip_log_mel(const float* wav, int B, int n_samples, int wav_stride, void* mel_out, int layout, int out_dtype,
                             const float* tables, void* workspace, void* stream) {
    AFHIP_CHECK(wav && mel_out && tables && workspace, "afhip_log_mel: null pointer");
    AFHIP_CHECK(B > 0 && B <= 65535, "afhip_log_mel: bad batch %d", B);
    AFHIP_CHECK(n_samples > 0 && n_samples <= NSAMP, "afhip_log_mel: n_samples=%d must be in [1,%d] (truncate first, audio.py:1042-1044)", n_samples, NSAMP);
    AFHIP_CHECK(wav_stride >= n_samples, "afhip_log_mel: wav_stride < n_samples");
    AFHIP_CHECK(layout == 0 || layout == 1, "afhip_log_mel: bad layout %d", layout);
    AFHIP_CHECK(out_dtype == AFHIP_F32 || out_dtype == AFHIP_BF16, "afhip_log_mel: bad dtype %d", out_dtype);
    AFHIP_CHECK(((uintptr_t)tables % 16) == 0, "afhip_log_mel: tables must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int* clipmax = reinterpret_cast<int*>(workspace);
    float* scratch = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((size_t)B * sizeof(int) + 255) / 256 * 256);
    hipLaunchKernelGGL(logmel_init_max, dim3(cdiv(B, 256)), dim3(256), 0, s, clipmax, B);
    const size_t lds1 = sizeof(float) * (size_t)(MAIN_FLOATS + Tables::CWMAX + 3 * NMEL);
    hipLaunchKernelGGL(logmel_pass1, dim3(cdiv(NFRAMES, FT), B), dim3(256), lds1, s, wav, n_samples, (long long)wav_stride,
                       tables, scratch, clipmax);
    const dim3 g2(cdiv(NFRAMES, 32), B);
    if (out_dtype == AFHIP_F32) {
        if (layout == 0) hipLaunchKernelGGL((logmel_pass2<float, 0>), g2, dim3(256), 0, s, scratch, clipmax, (float*)mel_out);
        else hipLaunchKernelGGL((logmel_pass2<float, 1>), g2, dim3(256), 0, s, scratch, clipmax, (float*)mel_out);
    } else {
        if (layout == 0) hipLaunchKernelGGL((logmel_pass2<bf16, 0>), g2, dim3(256), 0, s, scratch, clipmax, (bf16*)mel_out);
        else hipLaunchKernelGGL((logmel_pass2<bf16, 1>), g2, dim3(256), 0, s, scratch, clipmax, (bf16*)mel_out);
    }
    AFHIP_LAUNCH_CHECK();
    return 0;
}
