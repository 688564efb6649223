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
//   * power -> LDS -> banded mel filter bank (394 non-zeros) -> log10 -> (x + 4) / 4 written ONCE, in the caller's layout and
//     dtype, + the (max, min) of the log10 values each workgroup wrote (no atomics, no init launch).
// Floor fix-up (in place): out = max(out, ((clipmax - 8) + 4) / 4).  Rounding is monotone, so this equals the reference's
//     (max(x, clipmax - 8) + 4) / 4 bit for bit in f32 and in bf16; the f32 [B,3000,128] intermediate of the old two-pass form
//     (49 MB written + 49 MB read per 32 clips) is gone: HBM traffic = wav read + mel write + one read of the mel.
// Replaces transformers feature_extraction_whisper.py:135-170 (called from audio.py:1056-1069).
#include "common.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

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
    static constexpr size_t TW200 = CW + CWMAX;                       // [25][8][2] e^{-2 pi i t q / 200} (FFT path)
    static constexpr size_t WINF = TW200 + 400;                       // [400] periodic Hann window (FFT path)
    static constexpr size_t TOTAL = WINF + 400;
};

__device__ __forceinline__ int float_order_key(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float float_from_key(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }


// Banded mel + log10 tail shared by both pass-1 forms: thread (m = tid & 127, half = tid >> 7) keeps its mel's <= KCMAX
// band weights in registers and walks every second frame of the tile; the KCMAX power reads of a frame are independent
// (fixed trip count, masked), so their LDS latency overlaps -- the data-dependent `for k < count` loop this replaces
// serialised one LDS round trip per band element.  Returns the running max of the values it wrote.
constexpr int KCMAX = 12;
// LAYOUT 1: out [B][3000][128] (time-major, what the encoder consumes): a frame's 128 mels are one coalesced row.
// LAYOUT 0: out [B][128][3000] (the reference's layout): the tile goes through the LDS tile `tr` [NFR][129] and leaves as runs of
//           NFR consecutive frames per mel.
// one 64-mel half (mels mbase .. mbase + 63, one per lane) over the frames w, w + 4, ... of the tile (w = wave), band padded to PAD
// elements.  Fully unrolled, no early exit: frames past the clip end still have their row in the tile and are simply not stored, so all
// power reads are independent and hipcc may issue them as far ahead as it has registers.  The band is read CONTIGUOUSLY (k0 .. k0 + PAD - 1,
// pairs merge into ds_read2_b32); past the band the weight is 0 and the value read -- the next bins, or for the last row whatever finite
// floats follow the tile -- adds exactly +0, so the sum keeps the reference's left-to-right order and its bits.
template <int NFR, int PITCH, typename TO, int LAYOUT, int PAD>
__device__ __forceinline__ float mel_tail_half(const float* __restrict__ pw, const float* __restrict__ cw, const int* __restrict__ cband,
                                               TO* __restrict__ out, float* __restrict__ tr, int b, int t0, int m, int w, float& mn) {
    const int k0 = cband[3 * m], kc = cband[3 * m + 1], wo = cband[3 * m + 2];
    float wk[PAD];
#pragma unroll
    for (int k = 0; k < PAD; ++k) wk[k] = k < kc ? cw[wo + k] : 0.f;
    float mx = -INFINITY;
#pragma unroll
    for (int fi = 0; fi < NFR / 4; ++fi) {
        const int f = w + 4 * fi;
        const int t = t0 + f;
        float pv[PAD];
#pragma unroll
        for (int k = 0; k < PAD; ++k) pv[k] = pw[f * PITCH + k0 + k];
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < PAD; ++k) acc = fmaf(wk[k], pv[k], acc);
        // log10 = log2 * log10(2) on v_log_f32 (1 ulp; the argument is >= 1e-10, far from the denormal range)
        const float v = __builtin_amdgcn_logf(fmaxf(acc, 1e-10f)) * 0.30102999566398120f;
        const float y = (v + 4.0f) / 4.0f;                 // the per-clip floor is applied in place by logmel_floor_kernel
        if (t < NFRAMES) {
            if constexpr (LAYOUT == 1) out[((long long)b * NFRAMES + t) * NMEL + m] = from_f32<TO>(y);
            else tr[f * (NMEL + 1) + m] = y;
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
        }
    }
    return mx;
}

template <int NFR, int PITCH, typename TO, int LAYOUT>
__device__ __forceinline__ float mel_tail(const float* __restrict__ pw, const float* __restrict__ cw, const int* __restrict__ cband,
                                          TO* __restrict__ out, float* __restrict__ tr, int b, int t0, int tid, float& mn) {
    // Round 3 (stamps: this tail was 37 % of a tile's time).  Whisper's 128 bands hold 1-2 bins below mel 64 and up to 9 above (394 in
    // all), so "one mel per thread, padded to 12" did 3.9x the band work and left the low-mel waves idle behind the high-mel ones.  Now
    // every wave takes a quarter of the frames and walks them twice: the 64 HIGH mels padded to the widest band among them, then the 64
    // LOW mels padded to theirs (wave-uniform, taken from the band table; 9 + 2 = 11 units per frame instead of 2 x 12).
    const int lane = tid & 63, w = tid >> 6;
    int ka = cband[3 * lane + 1], kb = cband[3 * (64 + lane) + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ka = max(ka, __shfl_xor(ka, o, 64)); kb = max(kb, __shfl_xor(kb, o, 64)); }
    float mx;
    if (kb <= 9) mx = mel_tail_half<NFR, PITCH, TO, LAYOUT, 9>(pw, cw, cband, out, tr, b, t0, 64 + lane, w, mn);
    else mx = mel_tail_half<NFR, PITCH, TO, LAYOUT, KCMAX>(pw, cw, cband, out, tr, b, t0, 64 + lane, w, mn);
    if (ka <= 2) mx = fmaxf(mx, mel_tail_half<NFR, PITCH, TO, LAYOUT, 2>(pw, cw, cband, out, tr, b, t0, lane, w, mn));
    else if (ka <= 4) mx = fmaxf(mx, mel_tail_half<NFR, PITCH, TO, LAYOUT, 4>(pw, cw, cband, out, tr, b, t0, lane, w, mn));
    else mx = fmaxf(mx, mel_tail_half<NFR, PITCH, TO, LAYOUT, KCMAX>(pw, cw, cband, out, tr, b, t0, lane, w, mn));
    if constexpr (LAYOUT == 0) {
        __syncthreads();
        for (int idx = tid; idx < NFR * NMEL; idx += 256) {
            const int mm = idx / NFR, f = idx % NFR;
            const int t = t0 + f;
            if (t < NFRAMES) out[((long long)b * NMEL + mm) * NFRAMES + t] = from_f32<TO>(tr[f * (NMEL + 1) + mm]);
        }
    }
    return mx;
}

// The per-clip maximum of the log10 values (feature_extraction_whisper.py:160-164) without atomics or an init launch: every pass-1
// workgroup leaves the (max, min) of the values it wrote in stats[clip][workgroup]; the floor launch reduces a clip's <= LM_STAT_SLOTS
// maxima itself and, from the minima, knows which workgroups' tiles hold NO value below the floor -- those it never reads.
constexpr int LM_STAT_SLOTS = 128;
__device__ __forceinline__ void lm_store_stats(float* __restrict__ stats, int b, float mx, float mn) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    mx = wave_max(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
    if (lane == 0) { red[wave] = mx; red[4 + wave] = mn; }
    __syncthreads();
    if (tid == 0) {
        stats[((long long)b * LM_STAT_SLOTS + blockIdx.x) * 2] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        stats[((long long)b * LM_STAT_SLOTS + blockIdx.x) * 2 + 1] = fminf(fminf(red[4], red[5]), fminf(red[6], red[7]));
    }
}

template <typename TO, int LAYOUT>
__global__ __launch_bounds__(256) void logmel_pass1(const float* __restrict__ wav, int n_samples, long long wav_stride,
                                                    const float* __restrict__ tab, TO* __restrict__ out,
                                                    float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);          // [NHOPROWS][HOPROW] samples, later [FT][PROW] power
    float* cw = xs + MAIN_FLOATS;                            // [CWMAX] compact mel weights
    int* cband = reinterpret_cast<int*>(cw + Tables::CWMAX); // [NMEL][3] first bin, count, offset
    float* tr = reinterpret_cast<float*>(cband + 3 * NMEL);  // LAYOUT 0 only: [FT][NMEL + 1] transposition tile
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

    // ---- banded mel + log10 (weights and band table in LDS: no dependent global loads) ----
    float mn = INFINITY;
    const float mx = mel_tail<FT, PROW, TO, LAYOUT>(pw, cw, cband, out, tr, b, t0, tid, mn);
    lm_store_stats(stats, b, mx, mn);
}


// ------------------------------------------------------------------------------------------------------------
// Pass 1, FFT form (default): the 400-point real DFT of a frame costs ~12 kFLOP as a mixed-radix FFT against 160 kFLOP
// for the folded DFT products above, which moves the kernel from the f32-MFMA roof towards the HBM roof.
//   * one workgroup = 32 frames of one clip; a frame is owned by 8 adjacent lanes;
//   * the windowed frame is packed as 200 complex points z[n] = x[2n] + i x[2n+1]; lane t takes the decimated
//     sequence z[8m + t], m = 0..24, and runs a 25-point DFT entirely in registers (5 x 5 Cooley-Tukey, radix-5
//     butterflies), multiplies by the twiddle e^{-2 pi i t q / 200}, and the 8 lanes finish with a radix-2
//     decimation-in-frequency DFT-8 ACROSS lanes (three lane^4 / lane^2 / lane^1 exchanges); lane t then holds
//     Z[q + 25 bitrev3(t)], q = 0..24;
//   * the real-input spectrum is unpacked from Z[k] and conj Z[200-k] (lane ^ 7, register 25 - q), power -> LDS,
//     then the same banded mel / log10 / running-max tail as the DFT form.
#ifndef LM_WG_PER_CU
#define LM_WG_PER_CU 3   /* workgroups per CU the FFT form is compiled for (register budget 512 / this per lane: 2 waves per SIMD -> 256, 3 -> 168).
                            3: no hoisted window, no register prefetch of the next tile's samples, 162 VGPRs: 70.4-71.8 us per 32 clips against
                            74.2 with 2 (same box); 4 would spill 43 registers */
#endif
constexpr int FFT_FT = 32;                 // frames per workgroup
constexpr int FFT_PITCH = 176;             // LDS floats per hop row: 176 f mod 64 = 0, 48, 32, 16 -> the 4 frames of a 32-lane group hit disjoint banks
constexpr int FFT_ROWS = FFT_FT + 2;       // hops touched: 31 + ceil(400 / 160)
constexpr int FFT_PROW = 216;              // LDS pitch of the power tile: 216 f + 25 k1 mod 32 is a bijection of (4 frames x 8 lanes) onto the 32 banks of a ds_write_b32 group (203 gave 2-way conflicts)
constexpr int FFT_MAIN = (FFT_ROWS * FFT_PITCH > FFT_FT * FFT_PROW) ? FFT_ROWS * FFT_PITCH : FFT_FT * FFT_PROW;

constexpr float W25R[17] = {1.000000000e+00f, 9.685831611e-01f, 8.763066800e-01f, 7.289686274e-01f, 5.358267950e-01f, 3.090169944e-01f, 6.279051953e-02f, -1.873813146e-01f, -4.257792916e-01f, -6.374239897e-01f, -8.090169944e-01f, -9.297764859e-01f, -9.921147013e-01f, -9.921147013e-01f, -9.297764859e-01f, -8.090169944e-01f, -6.374239897e-01f};
constexpr float W25I[17] = {-0.000000000e+00f, -2.486898872e-01f, -4.817536741e-01f, -6.845471059e-01f, -8.443279255e-01f, -9.510565163e-01f, -9.980267284e-01f, -9.822872507e-01f, -9.048270525e-01f, -7.705132428e-01f, -5.877852523e-01f, -3.681245527e-01f, -1.253332336e-01f, 1.253332336e-01f, 3.681245527e-01f, 5.877852523e-01f, 7.705132428e-01f};
constexpr float W400R[25] = {1.000000000e+00f, 9.998766325e-01f, 9.995065604e-01f, 9.988898750e-01f, 9.980267284e-01f, 9.969173337e-01f, 9.955619646e-01f, 9.939609555e-01f, 9.921147013e-01f, 9.900236577e-01f, 9.876883406e-01f, 9.851093262e-01f, 9.822872507e-01f, 9.792228106e-01f, 9.759167619e-01f, 9.723699204e-01f, 9.685831611e-01f, 9.645574185e-01f, 9.602936857e-01f, 9.557930148e-01f, 9.510565163e-01f, 9.460853588e-01f, 9.408807690e-01f, 9.354440308e-01f, 9.297764859e-01f};
constexpr float W400I[25] = {-0.000000000e+00f, -1.570731731e-02f, -3.141075908e-02f, -4.710645071e-02f, -6.279051953e-02f, -7.845909573e-02f, -9.410831332e-02f, -1.097343111e-01f, -1.253332336e-01f, -1.409012319e-01f, -1.564344650e-01f, -1.719291003e-01f, -1.873813146e-01f, -2.027872954e-01f, -2.181432414e-01f, -2.334453639e-01f, -2.486898872e-01f, -2.638730500e-01f, -2.789911060e-01f, -2.940403252e-01f, -3.090169944e-01f, -3.239174182e-01f, -3.387379202e-01f, -3.534748438e-01f, -3.681245527e-01f};
__device__ const float W16R[8] = {1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f, 0.f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f};
__device__ const float W16I[8] = {0.f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f, -1.000000000e+00f, -9.238795325e-01f, -7.071067812e-01f, -3.826834324e-01f};

// lane exchanges inside an 8-lane group as DPP moves (VALU, no LDS round trip): lane ^ 1, ^ 2 (quad_perm), ^ 4 (row_shl:4 on
// lanes 0-3 / 8-11 + row_shr:4 on lanes 4-7 / 12-15 of each 16-lane row), ^ 7 (row_half_mirror)
__device__ __forceinline__ float lane_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_xor4(float v) {
    const int x = __builtin_bit_cast(int, v);
    int a = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false);
    a = __builtin_amdgcn_update_dpp(a, x, 0x114, 0xF, 0xA, false);
    return __builtin_bit_cast(float, a);
}
__device__ __forceinline__ float lane_xor7(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }

// forward 5-point DFT in place: y_k = sum_n x_n e^{-2 pi i n k / 5}
__device__ __forceinline__ void dft5(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3,
                                     float& r4, float& i4) {
    constexpr float C1 = 3.090169944e-01f, C2 = -8.090169944e-01f, S1 = 9.510565163e-01f, S2 = 5.877852523e-01f;
    const float t1r = r1 + r4, t1i = i1 + i4, t2r = r2 + r3, t2i = i2 + i3;
    const float t3r = r1 - r4, t3i = i1 - i4, t4r = r2 - r3, t4i = i2 - i3;
    const float a1r = r0 + C1 * t1r + C2 * t2r, a1i = i0 + C1 * t1i + C2 * t2i;
    const float a2r = r0 + C2 * t1r + C1 * t2r, a2i = i0 + C2 * t1i + C1 * t2i;
    const float b1r = S1 * t3r + S2 * t4r, b1i = S1 * t3i + S2 * t4i;
    const float b2r = S2 * t3r - S1 * t4r, b2i = S2 * t3i - S1 * t4i;
    r0 = r0 + t1r + t2r; i0 = i0 + t1i + t2i;
    r1 = a1r + b1i; i1 = a1i - b1r;      // a1 - i b1
    r4 = a1r - b1i; i4 = a1i + b1r;      // a1 + i b1
    r2 = a2r + b2i; i2 = a2i - b2r;
    r3 = a2r - b2i; i3 = a2i + b2r;
}

template <typename TO, int LAYOUT>
__global__ __launch_bounds__(256, (LAYOUT == 0 && LM_WG_PER_CU > 2) ? 2 : LM_WG_PER_CU)      // [B, 128, 3000] output: its transposition tile costs the third workgroup (it spilled 8 registers at 3)
void logmel_pass1_fft(const float* __restrict__ wav, int n_samples, long long wav_stride, const float* __restrict__ tab, TO* __restrict__ out,
                      float* __restrict__ stats, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef AFHIP_LOGMEL_STAMPS   /* diagnostic build: -DAFHIP_LOGMEL_STAMPS, tools/mel_stamps.py */
#define LM_STAMP(k) do { if (dbg && blockIdx.x == 5 && blockIdx.y == 0 && threadIdx.x == 0) dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LM_STAMP(k) do { } while (0)
#endif
    LM_STAMP(0);
    float* xs = reinterpret_cast<float*>(smem_raw);          // [FFT_ROWS][FFT_PITCH] samples, later [FFT_FT][FFT_PROW] power
    float* tw = xs + FFT_MAIN;                               // [200][2] e^{-2 pi i j / 200}
    float* wn = tw + 400;                                    // [400] window
    float* cw = wn + 400;                                    // [CWMAX] compact mel weights
    int* cband = reinterpret_cast<int*>(cw + Tables::CWMAX); // [NMEL][3] first bin, count, offset
    float* trt = reinterpret_cast<float*>(cband + 3 * NMEL); // LAYOUT 0 only: [FFT_FT][NMEL + 1] transposition tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.y;
    constexpr int NTILE = (NFRAMES + FFT_FT - 1) / FFT_FT;
    const float* w = wav + (long long)b * wav_stride;

    // constant tables -> registers first (all global loads of the prologue are in flight together with the sample loads below;
    // the LDS stores follow the last load issue): twiddles + window (800 floats), compact mel weights (512), band table
    float tr[4], cr[2];
    int br[3] = {0, 0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        tr[2 * u] = i < 400 ? tab[Tables::TW200 + i] : 0.f;
        tr[2 * u + 1] = i < 400 ? tab[Tables::WINF + i] : 0.f;
        cr[u] = tab[Tables::CW + i];
    }
    if (tid < NMEL) {
        const int* band_g = reinterpret_cast<const int*>(tab + Tables::BAND);
        br[0] = band_g[2 * tid]; br[1] = band_g[2 * tid + 1]; br[2] = reinterpret_cast<const int*>(tab + Tables::COFF)[tid];
    }
    // ---- sample staging (same conventions as the DFT form): padded index p = t0*160 + s, original i = p - 200.  The workgroup
    //      is PERSISTENT over frame tiles of its clip: the global loads of tile n+1 are issued before the FFT of tile n and
    //      parked in registers (their HBM latency, ~3.5 us per tile when exposed, hides under the arithmetic) ----
    constexpr int NSTAGE = (FFT_FT - 1) * HOP + NFFT;       // 5360 samples = 1340 float4
    constexpr int NV4 = NSTAGE / 4, V4_PER_THREAD = (NV4 + 255) / 256;
    f32x4 sv[V4_PER_THREAD];
    const bool vec_ok = (n_samples <= NSAMP) && ((wav_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(wav) & 15) == 0);
    auto tile_interior = [&](int tile) __attribute__((always_inline)) {
        const int ibase = tile * FFT_FT * HOP - NFFT / 2;
        return vec_ok && ibase >= 0 && ibase + NSTAGE <= n_samples;
    };
    // interior tiles (neither clip edge nor zero-padded tail): aligned float4 loads into registers, stored to LDS later
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        const int ibase = tile * FFT_FT * HOP - NFFT / 2;
#pragma unroll
        for (int u = 0; u < V4_PER_THREAD; ++u) {
            const int v = tid + 256 * u;
            if (v < NV4) sv[u] = *reinterpret_cast<const f32x4*>(w + ibase + 4 * v);
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < V4_PER_THREAD; ++u) {
            const int v = tid + 256 * u;
            if (v < NV4) {
                const int sidx = 4 * v;                          // 160 % 4 == 0: the 4 samples share a hop row
                *reinterpret_cast<f32x4*>(xs + (sidx / HOP) * FFT_PITCH + (sidx % HOP)) = sv[u];
            }
        }
    };
    // edge tiles (2-3 of the 94 of a full clip): reflect padding / zero tail element by element, straight to LDS, not prefetched
    auto stage_edge_tile = [&](int tile) __attribute__((always_inline)) {
        const int ibase = tile * FFT_FT * HOP - NFFT / 2;
        for (int sidx = tid; sidx < NSTAGE; sidx += 256) {
            int i = ibase + sidx;
            if (i < 0) i = -i;
            if (i >= NSAMP) i = 2 * (NSAMP - 1) - i;
            xs[(sidx / HOP) * FFT_PITCH + (sidx % HOP)] = (i >= 0 && i < n_samples) ? w[i] : 0.f;
        }
    };
    const bool first_interior = tile_interior(blockIdx.x);
    if (first_interior) load_tile(blockIdx.x);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 256 * u;
        if (i < 400) { tw[i] = tr[2 * u]; wn[i] = tr[2 * u + 1]; }
        cw[i] = cr[u];
    }
    if (tid < NMEL) { cband[3 * tid] = br[0]; cband[3 * tid + 1] = br[1]; cband[3 * tid + 2] = br[2]; }
    if (first_interior) store_tile(); else stage_edge_tile(blockIdx.x);
    __syncthreads();

    LM_STAMP(1);
    // ---- per-lane constants ----
    const int f = tid >> 3, t = tid & 7;
    const int k1 = ((t & 1) << 2) | (t & 2) | ((t >> 2) & 1);            // bitrev3(t): this lane ends with Z[q + 25 k1]
    const float s4 = (t & 4) ? -1.f : 1.f, s2 = (t & 2) ? -1.f : 1.f, s1 = (t & 1) ? -1.f : 1.f;
    float wa_r = 1.f, wa_i = 0.f, wb_r = 1.f, wb_i = 0.f;                  // DIF twiddles of the lane^4 and lane^2 stages
    if (t & 4) { wa_r = W16R[2 * (t & 3)]; wa_i = W16I[2 * (t & 3)]; }     // e^{-2 pi i (t & 3) / 8}
    if (t & 2) { wb_r = W16R[4 * (t & 1)]; wb_i = W16I[4 * (t & 1)]; }     // e^{-2 pi i (t & 1) / 4}
    const float lr = W16R[k1], li = W16I[k1];                               // e^{-2 pi i 25 k1 / 400}
    const int k1p = (8 - k1) & 7;
    const int src0 = (lane & ~7) | (((k1p & 1) << 2) | (k1p & 2) | ((k1p >> 2) & 1));   // lane holding Z[200 - 25 k1]

    float mx = -INFINITY, mn = INFINITY;
    for (int tile = blockIdx.x; tile < NTILE; tile += gridDim.x) {
    const int t0 = tile * FFT_FT;
    const int next = tile + (int)gridDim.x;
    const bool next_interior = next < NTILE && tile_interior(next);
#if LM_WG_PER_CU < 3
    if (next_interior) load_tile(next);                      // in flight during this tile's arithmetic
#endif
    // the window, twiddle and W400 values are loop-invariant per lane: the 50 window values are left to be hoisted into
    // registers for the life of the workgroup (2 waves per SIMD have the room), the twiddle table base and the W400 lane factor
    // are hidden behind opaque moves so those 100 values are NOT hoisted too (256 VGPRs + scratch otherwise)
    const float* wn_t = wn; const float* tw_t = tw;
    asm volatile("" : "+v"(tw_t));
#if LM_WG_PER_CU >= 3
    asm volatile("" : "+v"(wn_t));                              // three workgroups per CU: no room for the hoisted window either
#endif
    float lr_t = lr, li_t = li;                                 // likewise the 25 per-lane W400 products of the unpack
    asm volatile("" : "+v"(lr_t), "+v"(li_t));
    // an opaque move also hides the ADDRESS SPACE: through wn_t / tw_t the 50 table reads of a frame were FLAT loads that the memory pipeline
    // routed to the LDS (vmcnt + lgkmcnt, TA issue) instead of ds_read_b64 -- the casts say "LDS" again
    typedef const __attribute__((address_space(3))) f32x2 lds_f32x2_t;
    typedef const __attribute__((address_space(3))) float lds_f32_t;
    lds_f32_t* const wn_l = (lds_f32_t*)wn_t;
    lds_f32_t* const tw_l = (lds_f32_t*)tw_t;
    // ---- load + window: z[m] = (x[16m + 2t], x[16m + 2t + 1]) . w ----
    float zr[25], zi[25];
#pragma unroll
    for (int m = 0; m < 25; ++m) {
        const f32x2 xv = *reinterpret_cast<const f32x2*>(xs + (f + m / 10) * FFT_PITCH + 16 * (m % 10) + 2 * t);
        const f32x2 wv = *(lds_f32x2_t*)(wn_l + 16 * m + 2 * t);
        zr[m] = xv[0] * wv[0];
        zi[m] = xv[1] * wv[1];
    }
    LM_STAMP(2);
    // ---- 25-point DFT in registers: m = 5 m1 + m2, output index k = ka + 5 kb ----
#pragma unroll
    for (int m2 = 0; m2 < 5; ++m2)
        dft5(zr[m2], zi[m2], zr[5 + m2], zi[5 + m2], zr[10 + m2], zi[10 + m2], zr[15 + m2], zi[15 + m2], zr[20 + m2], zi[20 + m2]);
    // now zr[5 ka + m2] = A[ka][m2]; twiddle by W25^{m2 ka}
#pragma unroll
    for (int ka = 1; ka < 5; ++ka)
#pragma unroll
        for (int m2 = 1; m2 < 5; ++m2) {
            const float wr = W25R[m2 * ka], wi = W25I[m2 * ka];
            const float ar = zr[5 * ka + m2], ai = zi[5 * ka + m2];
            zr[5 * ka + m2] = ar * wr - ai * wi;
            zi[5 * ka + m2] = ar * wi + ai * wr;
        }
#pragma unroll
    for (int ka = 0; ka < 5; ++ka)
        dft5(zr[5 * ka], zi[5 * ka], zr[5 * ka + 1], zi[5 * ka + 1], zr[5 * ka + 2], zi[5 * ka + 2], zr[5 * ka + 3], zi[5 * ka + 3],
             zr[5 * ka + 4], zi[5 * ka + 4]);
    // zr[5 ka + kb] = Y[ka + 5 kb]; gather into natural order yq[q], q = ka + 5 kb, with the inter-stage twiddle e^{-2 pi i t q / 200}
    float yr[25], yi[25];
#pragma unroll
    for (int ka = 0; ka < 5; ++ka)
#pragma unroll
        for (int kb = 0; kb < 5; ++kb) {
            const int q = ka + 5 * kb;
            const f32x2 tv = *(lds_f32x2_t*)(tw_l + 2 * (q * 8 + t));
            const float ar = zr[5 * ka + kb], ai = zi[5 * ka + kb];
            yr[q] = ar * tv[0] - ai * tv[1];
            yi[q] = ar * tv[1] + ai * tv[0];
        }
    LM_STAMP(3);
    // ---- DFT-8 across the 8 lanes of the frame (decimation in frequency) ----
#pragma unroll
    for (int q = 0; q < 25; ++q) {
        float orr = lane_xor4(yr[q]), oii = lane_xor4(yi[q]);
        float ur = orr + s4 * yr[q], ui = oii + s4 * yi[q];
        yr[q] = ur * wa_r - ui * wa_i; yi[q] = ur * wa_i + ui * wa_r;
        orr = lane_xor2(yr[q]); oii = lane_xor2(yi[q]);
        ur = orr + s2 * yr[q]; ui = oii + s2 * yi[q];
        yr[q] = ur * wb_r - ui * wb_i; yi[q] = ur * wb_i + ui * wb_r;
        orr = lane_xor1(yr[q]); oii = lane_xor1(yi[q]);
        yr[q] = orr + s1 * yr[q]; yi[q] = oii + s1 * yi[q];
    }
    LM_STAMP(4);
    __syncthreads();   // every lane of the workgroup is done with the samples: LDS becomes the power tile
    LM_STAMP(5);

    // ---- unpack the real-input spectrum: X[k] = E + W400^k O, E = (Z[k] + conj Z[200-k]) / 2, O = -i (Z[k] - conj Z[200-k]) / 2 ----
    float* pw = xs + f * FFT_PROW;
    {
        const float pr0 = __shfl(yr[0], src0, 64), pi0 = __shfl(yi[0], src0, 64);
#pragma unroll
        for (int q = 0; q < 25; ++q) {
            float pr, pi;
            if (q == 0) { pr = pr0; pi = pi0; }
            else { pr = lane_xor7(yr[25 - q]); pi = lane_xor7(yi[25 - q]); }
            const float er = 0.5f * (yr[q] + pr), ei = 0.5f * (yi[q] - pi);
            const float o_r = 0.5f * (yi[q] + pi), o_i = -0.5f * (yr[q] - pr);
            const float wr = W400R[q] * lr_t - W400I[q] * li_t, wi = W400R[q] * li_t + W400I[q] * lr_t;
            const float xr = er + wr * o_r - wi * o_i, xi = ei + wr * o_i + wi * o_r;
            pw[q + 25 * k1] = xr * xr + xi * xi;
        }
        if (t == 0) { const float x200 = yr[0] - yi[0]; pw[200] = x200 * x200; }
    }
    __syncthreads();

    // ---- banded mel + log10 + affine, straight into the caller's tensor ----
    LM_STAMP(6);
    mx = fmaxf(mx, mel_tail<FFT_FT, FFT_PROW, TO, LAYOUT>(xs, cw, cband, out, trt, b, t0, tid, mn));
    LM_STAMP(7);
    if (next < NTILE) {
        __syncthreads();                                     // the power tile has been consumed: LDS takes the next samples
#if LM_WG_PER_CU >= 3
        if (next_interior) load_tile(next);                  // no register prefetch: the other workgroups of the CU cover this latency
#endif
        if (next_interior) store_tile(); else stage_edge_tile(next);
        __syncthreads();
    }
    }
    lm_store_stats(stats, b, mx, mn);
}

// Floor fix-up, in place: out = max(out, TO(((clipmax - 8) + 4) / 4)).  Same grid as pass 1: workgroup g of clip b owns the frame tiles
// g, g + gridDim.x, ... (TF frames each) that pass-1 workgroup g wrote, and stats[b][g].min tells whether any of their values lies below
// the floor: if not (full-length audio: never; zero-padded clips: the padded tiles do) it returns after reading nwg (max, min) pairs.
// Rewrites only the 16-byte chunks it changes.  Rounding is monotone: min >= floor before rounding implies stored >= TO(floor).
template <typename TO, int LAYOUT>
__global__ __launch_bounds__(256) void logmel_floor_kernel(TO* __restrict__ out, const float* __restrict__ stats, int nwg, int TF) {
    constexpr int EPC = 16 / sizeof(TO);
    __shared__ float s_max[4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* st = stats + (long long)b * LM_STAT_SLOTS * 2;
    float m = tid < nwg ? st[2 * tid] : -INFINITY;
    m = wave_max(m);
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    const float cmax = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    const float fl = ((cmax - 8.0f) + 4.0f) / 4.0f;
    const float flr = to_f32<TO>(from_f32<TO>(fl));          // the floor as the output dtype holds it
    const float my_min = (st[2 * blockIdx.x + 1] + 4.0f) / 4.0f;
    if (!(my_min < flr)) return;                              // nothing this workgroup's tiles hold is below the floor (workgroup-uniform)
    const int ntile = (NFRAMES + TF - 1) / TF;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int t0 = tile * TF, nf = (t0 + TF <= NFRAMES) ? TF : NFRAMES - t0;
        if constexpr (LAYOUT == 1) {
            TO* base = out + ((long long)b * NFRAMES + t0) * NMEL;       // nf x 128 contiguous values
            for (int c = tid; c < nf * NMEL / EPC; c += 256) {
                u32x4 raw = ld16(base + (long long)c * EPC);
                TO* v = reinterpret_cast<TO*>(&raw);
                bool changed = false;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float x = to_f32<TO>(v[e]);
                    if (x < flr) { v[e] = from_f32<TO>(flr); changed = true; }
                }
                if (changed) st16(base + (long long)c * EPC, raw);
            }
        } else {
            for (int i = tid; i < nf * NMEL; i += 256) {                // [B, 128, 3000]: runs of nf frames per mel
                const int mm = i / nf, f = i - mm * nf;
                TO* pv = out + ((long long)b * NMEL + mm) * NFRAMES + t0 + f;
                if (to_f32<TO>(*pv) < flr) *pv = from_f32<TO>(flr);
            }
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
    // inter-stage twiddles of the FFT form, laid out [q = 0..24][t = 0..7] = e^{-2 pi i t q / 200}: the 8 lanes of a frame read 8 consecutive
    // 8-byte entries (16 banks), the 4 frames of a 32-lane group the same ones -- conflict-free for every q.  (The [j = t q] table this
    // replaces put lanes t on banks 2 t q mod 64: 2- to 4-way conflicts whenever q is a multiple of 4.)
    for (int q = 0; q < 25; ++q)
        for (int tt = 0; tt < 8; ++tt) {
            const int j = tt * q;
            t[Tables::TW200 + 2 * (q * 8 + tt)] = (float)cos(two_pi * (double)j / 200.0);
            t[Tables::TW200 + 2 * (q * 8 + tt) + 1] = (float)(-sin(two_pi * (double)j / 200.0));
        }
    for (int n = 0; n < NFFT; ++n) t[Tables::WINF + n] = (float)(0.5 - 0.5 * cos(two_pi * (double)n / NFFT));
    memcpy(t + Tables::FILT, filters_host, sizeof(float) * NBIN * NMEL);
    int* band = reinterpret_cast<int*>(t + Tables::BAND);
    for (int m = 0; m < NMEL; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < NBIN; ++k)
            if (filters_host[k * NMEL + m] != 0.f) { if (lo < 0) lo = k; hi = k; }
        band[2 * m] = lo < 0 ? 0 : lo;
        band[2 * m + 1] = lo < 0 ? 0 : hi - lo + 1;
        AFHIP_CHECK(band[2 * m + 1] <= KCMAX, "afhip_log_mel_tables_host: mel %d spans %d bins (> %d)", m, band[2 * m + 1], KCMAX);
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
    return (size_t)B * LM_STAT_SLOTS * 2 * sizeof(float);    // (max, min) per clip and pass-1 workgroup; the log-mel values are written once, in place
}

namespace {
template <typename TO, int LAYOUT>
void logmel_launch(const float* wav, int B, int n_samples, int wav_stride, TO* out, const float* tables, float* stats, hipStream_t s) {
    int nwg, tf;
    if (afhip_opt(AFHIP_OPT_LOGMEL_DFT) == 1) {   // A/B switch: the folded-DFT MFMA form
        const size_t lds1 = sizeof(float) * (size_t)(MAIN_FLOATS + Tables::CWMAX + 3 * NMEL + (LAYOUT == 0 ? FT * (NMEL + 1) : 0));
        static unsigned long long attr_done = 0;
        if (afhip_first_use_on_device(&attr_done))
            (void)hipFuncSetAttribute((const void*)logmel_pass1<TO, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        nwg = cdiv(NFRAMES, FT); tf = FT;
        hipLaunchKernelGGL((logmel_pass1<TO, LAYOUT>), dim3(nwg, B), dim3(256), lds1, s, wav, n_samples, (long long)wav_stride,
                           tables, out, stats);
    } else {
        const size_t lds1 = sizeof(float) * (size_t)(FFT_MAIN + 800 + Tables::CWMAX + 3 * NMEL + (LAYOUT == 0 ? FFT_FT * (NMEL + 1) : 0));
#ifdef AFHIP_LOGMEL_STAMPS   /* diagnostic build only (tools/mel_stamps.py): 8 x s_memtime stamps of one workgroup */
        const char* dp = getenv("AFHIP_LOGMEL_DBGPTR");
        unsigned long long* dbg = dp ? (unsigned long long*)strtoull(dp, nullptr, 0) : nullptr;
#else
        unsigned long long* dbg = nullptr;
#endif
        // persistent over frame tiles: LM_WG_PER_CU workgroups per CU in one generation, each walking ceil(94 / gx) tiles
        int gx = cdiv(256 * LM_WG_PER_CU, B);
        gx = gx < 1 ? 1 : (gx > cdiv(NFRAMES, FFT_FT) ? cdiv(NFRAMES, FFT_FT) : gx);
        nwg = gx; tf = FFT_FT;
        hipLaunchKernelGGL((logmel_pass1_fft<TO, LAYOUT>), dim3(gx, B), dim3(256), lds1, s, wav, n_samples, (long long)wav_stride,
                           tables, out, stats, dbg);
    }
    // floor fix-up on the grid of pass 1: a workgroup re-reads only the tiles its pass-1 twin reported a value below the floor for
    static_assert(NFRAMES / FFT_FT + 1 <= LM_STAT_SLOTS && NFRAMES / FT + 1 <= LM_STAT_SLOTS, "stats slots per clip");
    hipLaunchKernelGGL((logmel_floor_kernel<TO, LAYOUT>), dim3(nwg, B), dim3(256), 0, s, out, stats, nwg, tf);
}
}  // namespace

extern "C" int afhip_log_mel(const float* wav, int B, int n_samples, int wav_stride, void* mel_out, int layout, int out_dtype,
                             const float* tables, void* workspace, void* stream) {
    AFHIP_CHECK(wav && mel_out && tables && workspace, "afhip_log_mel: null pointer");
    AFHIP_CHECK(B > 0 && B <= 65535, "afhip_log_mel: bad batch %d", B);
    AFHIP_CHECK(n_samples > 0 && n_samples <= NSAMP, "afhip_log_mel: n_samples=%d must be in [1,%d] (truncate first, audio.py:1042-1044)", n_samples, NSAMP);
    AFHIP_CHECK(wav_stride >= n_samples, "afhip_log_mel: wav_stride < n_samples");
    AFHIP_CHECK(layout == 0 || layout == 1, "afhip_log_mel: bad layout %d", layout);
    AFHIP_CHECK(out_dtype == AFHIP_F32 || out_dtype == AFHIP_BF16, "afhip_log_mel: bad dtype %d", out_dtype);
    AFHIP_CHECK(((uintptr_t)tables % 16) == 0 && ((uintptr_t)mel_out % 16) == 0, "afhip_log_mel: tables and mel_out must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    AFHIP_CHECK(((uintptr_t)workspace % 8) == 0, "afhip_log_mel: workspace must be 8-byte aligned");
    float* clipmax = reinterpret_cast<float*>(workspace);      // [B][LM_STAT_SLOTS][2]: (max, min) per pass-1 workgroup, written before it is read
    if (out_dtype == AFHIP_F32) {
        if (layout == 0) logmel_launch<float, 0>(wav, B, n_samples, wav_stride, (float*)mel_out, tables, clipmax, s);
        else logmel_launch<float, 1>(wav, B, n_samples, wav_stride, (float*)mel_out, tables, clipmax, s);
    } else {
        if (layout == 0) logmel_launch<bf16, 0>(wav, B, n_samples, wav_stride, (bf16*)mel_out, tables, clipmax, s);
        else logmel_launch<bf16, 1>(wav, B, n_samples, wav_stride, (bf16*)mel_out, tables, clipmax, s);
    }
    AFHIP_LAUNCH_CHECK();
    return 0;
}
