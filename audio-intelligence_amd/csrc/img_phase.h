// Weight-streaming GEMM phase of the bf16 decode step whose activations arrive as a FRAGMENT-ORDER IMAGE in global memory
// (M <= 16 rows, K % 512 == 0): C[M,N] = A'[M,K] . W[N,K]^T.
//
// stream_phase.h takes a row-major activation and pays for it in every workgroup: the same [8, 3584] rows are re-read, multiplied by
// the RMSNorm gain and scattered into LDS 256 times per launch -- tools/stream_stamps.py: 5 of the 8.6 us of the q|k|v launch, before
// the first MFMA.  Here the PRODUCER of an activation (the embedding, the o / down projection's residual epilogue, the SwiGLU epilogue,
// the attention merge) writes it once in the order the MFMA A operand wants it,
//     image[k / 64][half = (k >> 3) & 1][q = (k >> 4) & 3][row < RM][k & 7]          (RM * 128 bytes per 64-deep K step),
// with the gain of the NEXT RMSNorm already applied (x stays available as plain rows for the residual adds), plus one partial sum of
// squares per row and producing workgroup.  A consuming wave then loads its two A fragments of a K step with two contiguous
// RM * 64-byte loads beside the weight loads (tools/micro/stream_shape.hip, A=frag: the rate of activations held in LDS to within a
// few per cent), there is no staging pass, no LDS image, no barrier in front of the K loop; the RMSNorm row scale is
// rsqrt(sum of the partials / K + eps) applied to the accumulator in the epilogue (Qwen2RMSNorm, modeling_qwen2.py:238-252).
// One workgroup per CU walks its units (NT tiles of <= 16 weight rows) in one continuous weight stream, eight waves split K and
// combine through LDS per unit -- as stream_phase.h.
#pragma once
#include "stream_phase.h"

namespace stream {

// byte offset of element (row m, column k) in a fragment-order image of RM rows
__device__ __forceinline__ long long img_off(int RM, int m, int k) {
    return (long long)(k >> 6) * (RM * 128) + ((((k >> 3) & 1) * (4 * RM) + ((k >> 4) & 3) * RM + m) << 4) + ((k & 7) << 1);
}

// The phase descriptor (kernel argument).  A = the activation image; ss_in = [RM][ss_n] partial sums of squares of its rows (RMS only).
struct ImgDesc {
    const char* A; const char* W; const char* bias; const char* res; char* C;      // res / C: plain [M, N] rows (bf16; C f32 if out_f32), C may be NULL
    const float* ss_in; int ss_n;
    const float* w_scale;                                                            // W8 phases: [N] f32 row scales of the e4m3 weight
    char* img_out; const char* img_gain; float* ss_out;                             // != NULL: the output also leaves as an image (x gain[n]) + partials [RM][grid]
    int M, N, K; long long ldw, ldc, ldres; int out_f32; float eps; int tile_rows;
    const int32_t* am_iv; int am_n_iv; float* am_val; int* am_idx;                  // greedy lm_head: argmax partials [RM][grid]
#ifdef AFHIP_STREAM_STAMPS
    unsigned long long* dbg;
#endif
};

// W8: the weights are OCP e4m3 bytes with one f32 scale per output row (W8A16, gemm_skinny_fp8.hip): a K step is 128 deep so that a lane
// still reads 32 contiguous bytes of its weight row; the 32 values are widened to bf16 in registers and meet the lane's four A fragments
// -- chunks of the SAME bf16 image (64-step 2 s + (q >> 1), column group 2 (q & 1) + (c >> 1), half c & 1) -- in four MFMAs.  K / 128 need not
// divide by the eight waves: wave w takes steps w, w + 8, ... of every unit and all of them meet at the unit's combine barrier.
template <int NT, bool PAIR, bool RMS, int RM, int DEPTH, bool W8 = false>
struct ImgPhase {
    static_assert(RM == 8 || RM == 16, "activation image rows");
    static_assert(!PAIR || NT == 2, "SwiGLU pairs: one gate tile + one up tile");
    static constexpr int NI = (NT * 256 + 511) / 512;               // epilogue items per thread
    static constexpr int NA = W8 ? 4 : 2, KSTEP = W8 ? 128 : 64, WB = W8 ? 1 : 2;       // A fragments per step, K per step, bytes per weight
    struct Regs { u32x4 w0[NT], w1[NT], a[NA]; };
    ImgDesc p;
    float* red;        // [2 unit parities][8 waves][NT tiles][64 lanes][4]
    float* red_ss;     // [16] row sums of squares
    float* am_v;       // [8 waves][4]
    int* am_i;
    int tid, lane, wave, c16, q, TR, gates, spw, my_units, total, cr;
    const char* wrow[NT];
    Regs r[DEPTH];
    long long aofs;
    int ig, iu, ij;
    // epilogue operands as LOADED (bf16 bits): widening them where they are fetched made the wave wait for them -- and, vmcnt being in order, for
    // the whole weight window issued just before -- at the top of the kernel instead of at the unit's end
    unsigned short ep_b[NI], ep_r[NI], ep_g[NI];
    float ep_s[NI];
    float ssp[2];      // this lane's share of the partial sums of squares of rows wave, wave + 8
    float ssv[RM / 8][4];  // ... as loaded (up to 256 partials per row: four per lane), summed when the K loop is about to start
    f32x4 acc[NT];
    int cu, cj;
    float am_best, ss_acc;
    int am_bi;
    int am_lo[4], am_hi[4];              // the first four allowed-id intervals of the greedy pick, fetched once (load_am)
    float pr_sg, pr_su;                  // e4m3 SwiGLU pair: the weight scales of this thread's gate / up row of the unit, fetched when the unit starts

    static constexpr size_t lds_bytes() { return (size_t)(2 * NW * NT * 256 + 16 + 64) * sizeof(float); }

    template <typename T> static __device__ __forceinline__ T* opq(T* v) { asm volatile("" : "+s"(v)); return v; }
    static __device__ __forceinline__ int opq(int v) { asm volatile("" : "+s"(v)); return v; }
    static __device__ __forceinline__ long long opq(long long v) { asm volatile("" : "+s"(v)); return v; }
    // field-by-field copy into scalar registers the compiler cannot re-derive from the kernel-argument segment (it re-read the
    // descriptors with s_load + s_waitcnt in front of every weight load of a chain launch)
    __device__ __forceinline__ ImgPhase(const ImgDesc& d, char* smem) {
        p.A = opq(d.A); p.W = opq(d.W); p.bias = opq(d.bias); p.res = opq(d.res); p.C = opq(d.C);
        p.ss_in = opq(d.ss_in); p.ss_n = opq(d.ss_n); p.w_scale = opq(d.w_scale);
        p.img_out = opq(d.img_out); p.img_gain = opq(d.img_gain); p.ss_out = opq(d.ss_out);
        p.M = opq(d.M); p.N = opq(d.N); p.K = opq(d.K); p.ldw = opq(d.ldw); p.ldc = opq(d.ldc); p.ldres = opq(d.ldres);
        p.out_f32 = opq(d.out_f32); p.eps = __int_as_float(opq(__float_as_int(d.eps))); p.tile_rows = opq(d.tile_rows);
        p.am_iv = opq(d.am_iv); p.am_n_iv = opq(d.am_n_iv); p.am_val = opq(d.am_val); p.am_idx = opq(d.am_idx);
#ifdef AFHIP_STREAM_STAMPS
        p.dbg = opq(d.dbg);
#endif
        red = reinterpret_cast<float*>(smem);
        red_ss = red + 2 * NW * NT * 256;
        am_v = red_ss + 16;
        am_i = reinterpret_cast<int*>(am_v + 32);
    }

    __device__ __forceinline__ void set_rows(int ui) {
        const int u = (int)blockIdx.x + ui * (int)gridDim.x;
        if constexpr (PAIR) {
            int g = u * TR + cr;
            g = g < gates ? g : gates - 1;
            const long long n = ((long long)(g >> 5) << 6) + (g & 31);   // gate g = W row 64 (g >> 5) + (g & 31), its up row 32 further
            wrow[0] = p.W + n * p.ldw * WB;
            wrow[1] = wrow[0] + 32 * p.ldw * WB;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int n = (u * NT + t) * TR + cr;
                n = n < p.N ? n : p.N - 1;
                wrow[t] = p.W + (long long)n * p.ldw * WB;
            }
        }
    }
    __device__ __forceinline__ void issue_a(Regs& x, int s) {
        if constexpr (W8) {
            // fragment c of lane q: image chunk (64-step 2 s + (q >> 1), column group 2 (q & 1) + (c >> 1), half c & 1)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                x.a[c] = ld16g(p.A, aofs + (long long)s * (RM * 256) + (c & 1) * (RM * 64) + (c >> 1) * (RM * 16));
        } else {
            x.a[0] = ld16g(p.A, aofs + (long long)s * (RM * 128));
            x.a[1] = ld16g(p.A, aofs + (long long)s * (RM * 128) + RM * 64);
        }
    }
    __device__ __forceinline__ void issue(Regs& x) {
        const int s = wave + NW * ij;
        const long long koff = (long long)s * 128 + q * 32;          // 32 bytes per lane and step in either weight format
#pragma unroll
        for (int t = 0; t < NT; ++t) { x.w0[t] = ld16g(wrow[t], koff); x.w1[t] = ld16g(wrow[t], koff + 16); }
        issue_a(x, s);
        ++ig;
        if (++ij == spw) { ij = 0; ++iu; if (iu < my_units) set_rows(iu); }
    }
    // e4m3 SwiGLU pair units: the two weight scales of this thread's output, fetched when the unit STARTS (read where they are used they were
    // loads younger than the next unit's window, and their wait drained it at every unit's end)
    __device__ __forceinline__ void fetch_pair(int ui) {
        if constexpr (PAIR && W8) {
            const int u = (int)blockIdx.x + ui * (int)gridDim.x;
            const int ln = (tid >> 2) & 63;
            int gi = u * TR + (ln & 15);
            gi = (tid < 256 && (ln & 15) < TR && gi < gates) ? gi : 0;
            const int ng = ((gi >> 5) << 6) + (gi & 31);
            pr_sg = ld_f32g(p.w_scale, (long long)ng * 4);
            pr_su = ld_f32g(p.w_scale, (long long)(ng + 32) * 4);
        }
    }
    // bias / residual / image gain of this thread's epilogue items (o = tid (+ 512): reg = o & 3, column = (o >> 2) & 15, row group =
    // (o >> 6) & 3, tile = o >> 8), fetched when a unit STARTS.
    __device__ __forceinline__ void fetch_epi(int ui) {
        if constexpr (!PAIR) {
            const int u = (int)blockIdx.x + ui * (int)gridDim.x;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int o = tid + 512 * i;
                const int reg = o & 3, ln = (o >> 2) & 63, nt = o >> 8;
                const int n = (u * NT + nt) * TR + (ln & 15), m = 4 * (ln >> 4) + reg;
                const bool ok = o < NT * 256 && (ln & 15) < TR && n < p.N && m < p.M;
                const int nc = ok ? n : 0, mc = ok ? m : 0;           // clamped: the loads are unconditional
                ep_b[i] = p.bias ? ld_u16g(p.bias, (long long)nc * 2) : (unsigned short)0;
                ep_g[i] = p.img_gain ? ld_u16g(p.img_gain, (long long)nc * 2) : (unsigned short)0x3f80;      // 1.0
                ep_r[i] = p.res ? ld_u16g(p.res, ((long long)mc * p.ldres + nc) * 2) : (unsigned short)0;
                ep_s[i] = W8 ? ld_f32g(p.w_scale, (long long)nc * 4) : 1.f;
            }
        }
    }

    // The allowed-id intervals of the greedy pick (lm_head), BEFORE the weight window like the partial sums of squares: read per output element in
    // the epilogue they were loads younger than the next unit's window, and vmcnt being in order their wait drained that window at every unit's end
    // (ten times per workgroup in the lm_head).
    __device__ __forceinline__ void load_am() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = p.am_val != nullptr && k < p.am_n_iv;
            am_lo[k] = on ? ld_i32g(p.am_iv, (long long)(2 * k) * 4) : 0;
            am_hi[k] = on ? ld_i32g(p.am_iv, (long long)(2 * k + 1) * 4) : 0;           // [0, 0): matches nothing
        }
    }
    __device__ __forceinline__ void load_ss() {
#pragma unroll
        for (int ri = 0; ri < RM / 8; ++ri) {
            const int m = wave + 8 * ri;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = lane + 64 * k;
                const int jc = j < p.ss_n ? j : p.ss_n - 1;                 // clamped: unconditional loads
                const float v = ld_f32g(p.ss_in, (long long)(m * p.ss_n + jc) * 4);
                ssv[ri][k] = j < p.ss_n ? v : 0.f;
            }
        }
    }
    // Indices, the partial sums of squares, the first weight window, the epilogue operands of the first unit.
    __device__ __forceinline__ void begin() {
        tid = threadIdx.x; lane = tid & 63; wave = tid >> 6;
        c16 = lane & 15; q = lane >> 4;
        TR = p.tile_rows;
        gates = p.N >> 1;
        const int n_units = PAIR ? (gates + TR - 1) / TR : (p.N + NT * TR - 1) / (NT * TR);
        // K steps per wave per unit: bf16 K % 512 == 0 (host), every wave the same count; e4m3: wave w takes steps w, w + 8, ... of K / 128
        spw = W8 ? (p.K / 128 - wave + 7) / 8 : p.K / (KS * NW);
        my_units = (n_units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
        if (my_units < 0) my_units = 0;
        total = my_units * spw;
        cr = c16 < TR ? c16 : TR - 1;                               // lanes past the share re-read its last row (same lines), their results are dropped
        aofs = W8 ? (long long)(q >> 1) * (RM * 128) + (long long)((((q & 1) * 2) * RM + (c16 & (RM - 1))) << 4)
                  : (long long)((q * RM + (c16 & (RM - 1))) << 4);
        ST_STAMP(0);
        ig = 0; iu = 0; ij = 0;
        // the partial sums of squares go out BEFORE the weight window: vmcnt retires in order, and behind the window their first use
        // drained all of it (gate/up: 5.8 us from launch to first MFMA, then a refill bubble)
        if constexpr (RMS) load_ss();
        load_am();
        if (my_units > 0) set_rows(0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (ig < total) issue(r[d]);
        ST_STAMP(1);
        if (my_units > 0) { fetch_epi(0); fetch_pair(0); }
    }

    __device__ __forceinline__ void finish_unit() {
        float* rp = red + (cu & 1) * (NW * NT * 256);              // two buffers: the next unit's barrier orders the reuse
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            *reinterpret_cast<f32x4*>(rp + (((wave * NT + t) * 64 + lane) << 2)) = acc[t];
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (RMS) {
            if (cu == 0) {                                          // the row sums of squares: this wave's two rows, once per phase
#pragma unroll
                for (int ri = 0; ri < RM / 8; ++ri) {
                    const float t = wave_sum(ssp[ri]);
                    if (lane == 0) red_ss[wave + 8 * ri] = t;
                }
            }
        }
        const int u = (int)blockIdx.x + cu * (int)gridDim.x;
        __syncthreads();
        if constexpr (PAIR) {
            if (tid < 256) {
                const int reg = tid & 3, ln = tid >> 2;
                float g = 0.f, uu = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    g += rp[(((w * 2 + 0) * 64 + ln) << 2) + reg];
                    uu += rp[(((w * 2 + 1) * 64 + ln) << 2) + reg];
                }
                const int mrow = 4 * (ln >> 4) + reg;
                const int gi = u * TR + (ln & 15);
                if ((ln & 15) < TR && gi < gates && mrow < p.M) {
                    if constexpr (RMS) {
                        const float rs = rsqrtf(red_ss[mrow] / (float)p.K + p.eps);
                        g *= rs; uu *= rs;
                    }
                    if constexpr (W8) { g *= pr_sg; uu *= pr_su; }
                    const float y = silu(g) * uu;
                    if (p.img_out) st_bf16g(p.img_out, img_off(RM, mrow, gi), y);
                    else st_bf16g(p.C, ((long long)mrow * p.ldc + gi) * 2, y);
                }
            }
            if (cu + 1 < my_units) fetch_pair(cu + 1);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int o = tid + 512 * i;
                if (o < NT * 256) {
                    const int reg = o & 3, ln = (o >> 2) & 63, nt = o >> 8;
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) v += rp[((((w * NT + nt) * 64) + ln) << 2) + reg];
                    const int nn = (u * NT + nt) * TR + (ln & 15), mm = 4 * (ln >> 4) + reg;
                    if ((ln & 15) < TR && nn < p.N && mm < p.M) {
                        if constexpr (RMS) v *= rsqrtf(red_ss[mm] / (float)p.K + p.eps);
                        if constexpr (W8) v *= ep_s[i];
                        if (p.bias) v += bf16_bits_f32(ep_b[i]);
                        if (p.res) v += bf16_bits_f32(ep_r[i]);
                        if (p.C) {
                            if (p.out_f32) st_f32g(p.C, ((long long)mm * p.ldc + nn) * 4, v);
                            else st_bf16g(p.C, ((long long)mm * p.ldc + nn) * 2, v);
                        }
                        if (p.img_out) {
                            // the stored (bf16) value is what the next RMSNorm sees: gain applied to it, its square summed
                            const float xb = (float)(bf16)v;
                            st_bf16g(p.img_out, img_off(RM, mm, nn), xb * bf16_bits_f32(ep_g[i]));
                            ss_acc += xb * xb;
                        }
                        if (p.am_val) {
                            bool ok = false;
                            #pragma unroll
                            for (int k = 0; k < 4; ++k) ok = ok || (nn >= am_lo[k] && nn < am_hi[k]);
                            for (int k = 4; k < p.am_n_iv; ++k) ok = ok || (nn >= ld_i32g(p.am_iv, (long long)(2 * k) * 4) && nn < ld_i32g(p.am_iv, (long long)(2 * k + 1) * 4));
                            const float vb = (float)(bf16)v;           // the reference takes argmax over model-dtype logits
                            if (ok && (vb > am_best || (vb == am_best && nn < am_bi))) { am_best = vb; am_bi = nn; }
                        }
                    }
                }
            }
            if (cu + 1 < my_units) fetch_epi(cu + 1);
        }
    }

    __device__ __forceinline__ void run() {
        if constexpr (RMS) {
            // partial sums of squares of rows wave (and wave + 8): lane-strided here, across the wave at the first unit's end
#pragma unroll
            for (int ri = 0; ri < RM / 8; ++ri) {
                float t = ((ssv[ri][0] + ssv[ri][1]) + ssv[ri][2]) + ssv[ri][3];
                const int m = wave + 8 * ri;
                for (int j = lane + 256; j < p.ss_n; j += 64) t += ld_f32g(p.ss_in, (long long)(m * p.ss_n + j) * 4);      // more than 256 producing workgroups: the rest, the slow way
                ssp[ri] = t;
            }
        }
        ST_STAMP(2);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        cu = 0; cj = 0;
        am_best = -INFINITY; am_bi = 0x7fffffff; ss_acc = 0.f;

        for (int g0 = 0; g0 < total; g0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (g0 + d < total) {                               // wave-uniform; every wave reaches each unit's combine barrier exactly once
                    if constexpr (W8) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            // widen this tile's 32 e4m3 weights: dword e of half h holds k = 16 h + 4 e .. + 4 -> bf16 chunk 2 h + (e >> 1)
                            u32x4 wb[4];
#pragma unroll
                            for (int h = 0; h < 2; ++h)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                                    const uint32_t v = h ? r[d].w1[t][e] : r[d].w0[t][e];
                                    wb[2 * h + (e >> 1)][2 * (e & 1)] = __builtin_bit_cast(uint32_t, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)v, 1.0f, false));
                                    wb[2 * h + (e >> 1)][2 * (e & 1) + 1] = __builtin_bit_cast(uint32_t, (bf16x2_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)v, 1.0f, true));
                                }
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r[d].a[c]), __builtin_bit_cast(bf16x8, wb[c]), acc[t], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r[d].a[0]), __builtin_bit_cast(bf16x8, r[d].w0[t]), acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, r[d].a[1]), __builtin_bit_cast(bf16x8, r[d].w1[t]), acc[t], 0, 0, 0);
                        }
                    }
                    if (g0 + d == 0) ST_STAMP(3);
                    if (ig < total) issue(r[d]);
                    if (g0 + d == total - 1) ST_STAMP(4);
                    if (++cj == spw) { finish_unit(); cj = 0; ++cu; }
                }
            }
        }
        ST_STAMP(5);
        if constexpr (!PAIR) {
            // this thread's epilogue items all belong to row 4 * ((tid >> 6) & 3) + (tid & 3): merge over the 16 columns (lane bits 2..5),
            // then over the two waves that share a row group
            if (p.ss_out) {
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) ss_acc += __shfl_xor(ss_acc, o, 64);
                __syncthreads();
                if (lane < 4) am_v[wave * 4 + lane] = ss_acc;
                __syncthreads();
                if (tid < RM) {
                    const int w0 = tid >> 2, rg = tid & 3;
                    float t = am_v[w0 * 4 + rg];
                    if (NT * 256 > 256) t += am_v[(w0 + 4) * 4 + rg];
                    st_f32g(reinterpret_cast<char*>(p.ss_out), ((long long)tid * (int)gridDim.x + (int)blockIdx.x) * 4, tid < p.M ? t : 0.f);
                }
            }
            if (p.am_val) {
#pragma unroll
                for (int o = 4; o < 64; o <<= 1) {
                    const float ov = __shfl_xor(am_best, o, 64);
                    const int oi = __shfl_xor(am_bi, o, 64);
                    if (ov > am_best || (ov == am_best && oi < am_bi)) { am_best = ov; am_bi = oi; }
                }
                __syncthreads();
                if (lane < 4) { am_v[wave * 4 + lane] = am_best; am_i[wave * 4 + lane] = am_bi; }
                __syncthreads();
                if (tid < 16 && tid < p.M) {
                    const int w0 = tid >> 2, rg = tid & 3;          // row tid = 4 * w0 + rg lives in waves w0 and w0 + 4
                    float b0 = am_v[w0 * 4 + rg];
                    int i0 = am_i[w0 * 4 + rg];
                    const float b1 = am_v[(w0 + 4) * 4 + rg];
                    const int i1 = am_i[(w0 + 4) * 4 + rg];
                    if (b1 > b0 || (b1 == b0 && i1 < i0)) { b0 = b1; i0 = i1; }
                    const long long slot = (long long)tid * (int)gridDim.x + (int)blockIdx.x;
                    st_f32g(reinterpret_cast<char*>(p.am_val), slot * 4, b0);
                    st_f32g(reinterpret_cast<char*>(p.am_idx), slot * 4, __int_as_float(i0));
                }
            }
        }
    }
};

}  // namespace stream
