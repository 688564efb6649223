// One weight-streaming GEMM of the bf16 decode shapes (M <= 16 rows, K % 512 == 0) over a ROW-MAJOR activation: C[M,N] = A'[M,K] . W[N,K]^T,
// as a device-side object (gemm_stream.hip wraps it in a kernel; img_phase.h is its sibling for activations that arrive as images).
//
// What bounds these GEMMs on MI355X is not the weight stream's shape but the ACTIVATION loads next to it
// (tools/micro/stream_shape.hip: the 136-MB down projection streams at 3.9 TB/s with the activation fragments read from global
// memory beside every weight fragment, 5.0 with those reads made contiguous, 5.5 with the activations in LDS; a weight stored in
// fragment order changes nothing).  So one workgroup per CU walks its units (NT tiles of <= 16 weight rows) in ONE continuous
// weight stream -- a rolling window of DEPTH K steps per wave that runs across unit boundaries, weights straight from HBM into
// v_mfma_f32_16x16x32_bf16 B registers, 32 B per lane -- and the A operand reaches the MFMAs through LDS only:
//   image form  (M * K fits): the whole activation [RM, K] is staged once per workgroup in MFMA A-operand order, RMSNorm gain
//               applied and the row sums of squares taken on the way; every load of the staging pass is in flight before the first
//               one is used;
//   slot form   (down projection, K = 18 944: 303 KB of activations): each wave fetches the RM x 128-B activation rows of ITS K step
//               with RM / 8 fully coalesced loads beside the weight loads and turns them into fragment order through a private LDS
//               slot (ds_write_b128, two ds_read_b128, no barrier).
// The eight waves of a workgroup split K (wave w takes steps w, w + 8, ...) and combine through LDS per unit; bias / residual are
// fetched when a unit starts.  K order per row and the order of the K-slice sum are those of skinny_kernel (gemm_skinny.hip).
#pragma once
#include "skinny.h"

namespace stream {

constexpr int NW = 8, KS = 64;

// Every pointer of a phase descriptor has been through an opaque scalar copy (opq in the phase structs), after which the compiler no longer knows that it
// points to global memory and would emit FLAT loads: those count on lgkmcnt as well, so every LDS wait of the per-unit combine would drain
// the weight window in flight.  These helpers put the address space back.
#define STREAM_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ u32x4 ld16g(const void* base, long long off) { return *(const STREAM_AS1 u32x4*)((const char*)base + off); }
__device__ __forceinline__ float ld_bf16g(const void* base, long long off) { return (float)*(const STREAM_AS1 bf16*)((const char*)base + off); }
__device__ __forceinline__ unsigned short ld_u16g(const void* base, long long off) { return *(const STREAM_AS1 unsigned short*)((const char*)base + off); }
__device__ __forceinline__ float bf16_bits_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ float ld_f32g(const void* base, long long off) { return *(const STREAM_AS1 float*)((const char*)base + off); }
__device__ __forceinline__ int ld_i32g(const void* base, long long off) { return *(const STREAM_AS1 int*)((const char*)base + off); }
__device__ __forceinline__ void st_bf16g(void* base, long long off, float v) { *(STREAM_AS1 bf16*)((char*)base + off) = (bf16)v; }
__device__ __forceinline__ void st_f32g(void* base, long long off, float v) { *(STREAM_AS1 float*)((char*)base + off) = v; }


#ifdef AFHIP_STREAM_STAMPS   /* diagnostic build (tools/stream_stamps.py): 100-MHz wall-clock stamps of wave 0 of every workgroup */
#define ST_STAMP(k) do { if (p.dbg && tid == 0) p.dbg[(long long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ST_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// 16-B chunk index inside a [2 halves][4 q][RM rows] fragment-order block of one K step, rotated so that the eight lanes that hold one
// row's eight chunks (a ds_write_b128 lane group) hit eight different 16-B bank slots
__device__ __forceinline__ int slot_chunk(int RM, int r, int piece) {
    return (piece & 1) * (4 * RM) + (piece >> 1) * RM + (r & 8) + (((r & 7) + piece) & 7);
}

template <int AMODE, int NT, bool PAIR, bool SLOT, int RM, int DEPTH>
struct StreamPhase {
    static_assert(RM == 8 || RM == 16, "activation image rows");
    static_assert(!PAIR || NT == 2, "SwiGLU pairs: one gate tile + one up tile");
    static_assert(!SLOT || AMODE == SKINNY_A_PLAIN, "slot form: plain activations");
    static constexpr int AH = SLOT ? RM / 8 : 1;                   // slot form: coalesced activation loads per step
    static constexpr int NI = (NT * 256 + 511) / 512;               // epilogue items per thread
    static constexpr int SB = 8, NROW = RM / 8;                     // image staging: 16-B chunks per lane per row and batch, rows per wave
    struct Regs { u32x4 w0[NT], w1[NT], a[AH]; };

    // the descriptor, copied field by field into (wave-uniform) registers the compiler cannot re-derive from the kernel-argument segment:
    // with the five descriptors of a chain launch it re-read them with s_load + s_waitcnt in front of every weight load (the window of
    // a phase took 4-5 us to issue)
    struct Desc {
        const char* A; const char* W; const char* bias; const char* res; const char* norm_w; char* C;
        int M, N, K; long long lda, ldw, ldc, ldres; int out_f32; float norm_eps; int tile_rows;
        const int32_t* am_iv; int am_n_iv; float* am_val; int* am_idx;
#ifdef AFHIP_STREAM_STAMPS
        unsigned long long* dbg;
#endif
    } p;
    float* red;        // [2 unit parities][8 waves][NT tiles][64 lanes][4]
    float* red_ss;     // [16] row sums of squares (A_RMSNORM)
    float* am_v;       // [8 waves][4] argmax partials (values, then indices)
    int* am_i;
    char* aimg;        // image: [K / 64] step blocks of RM * 128 B in slot_chunk order; slots: [8 waves][DEPTH] such blocks
    int tid, lane, wave, c16, q, TR, gates, spw, my_units, total, cr;
    const char* wrow[NT];
    Regs r[DEPTH];
    long long aoff[AH];
    int wofs[AH], rofs0, rofs1;
    int ig, iu, ij;
    u32x4 sv[NROW][SB], sg[SB];
    float ssq[NROW];
    float ep_b[NI], ep_r[NI];
    f32x4 acc[NT];
    int cu, cj;
    float am_best;
    int am_bi;
    int am_lo[4], am_hi[4];              // the first four allowed-id intervals of the greedy pick, fetched once in begin()

    static constexpr size_t fixed_lds() { return (size_t)(2 * NW * NT * 256 + 16 + 64) * sizeof(float); }
    static size_t lds_bytes(int K) { return fixed_lds() + (SLOT ? (size_t)NW * DEPTH * RM * 128 : (size_t)RM * K * 2); }

    // an empty asm statement that "rewrites" the scalar register: the value stays wave-uniform but is no longer a kernel-argument load
    // the compiler could rematerialise
    template <typename T> static __device__ __forceinline__ T* opq(T* v) { asm volatile("" : "+s"(v)); return v; }
    static __device__ __forceinline__ int opq(int v) { asm volatile("" : "+s"(v)); return v; }
    static __device__ __forceinline__ long long opq(long long v) { asm volatile("" : "+s"(v)); return v; }
    __device__ __forceinline__ StreamPhase(const SkinnyP& q_, char* smem) {
        p.A = opq(q_.A); p.W = opq(q_.W); p.bias = opq(q_.bias); p.res = opq(q_.res); p.norm_w = opq(q_.norm_w); p.C = opq(q_.C);
        p.M = opq(q_.M); p.N = opq(q_.N); p.K = opq(q_.K);
        p.lda = opq(q_.lda); p.ldw = opq(q_.ldw); p.ldc = opq(q_.ldc); p.ldres = opq(q_.ldres);
        p.out_f32 = opq(q_.out_f32); p.norm_eps = __int_as_float(opq(__float_as_int(q_.norm_eps))); p.tile_rows = opq(q_.tile_rows);
        p.am_iv = opq(q_.am_iv); p.am_n_iv = opq(q_.am_n_iv); p.am_val = opq(q_.am_val); p.am_idx = opq(q_.am_idx);
#ifdef AFHIP_STREAM_STAMPS
        p.dbg = opq(q_.dbg);
#endif
        red = reinterpret_cast<float*>(smem);
        red_ss = red + 2 * NW * NT * 256;
        am_v = red_ss + 16;
        am_i = reinterpret_cast<int*>(am_v + 32);
        aimg = reinterpret_cast<char*>(am_i + 32);
    }

    __device__ __forceinline__ void set_rows(int ui) {
        const int u = (int)blockIdx.x + ui * (int)gridDim.x;
        if constexpr (PAIR) {
            int g = u * TR + cr;
            g = g < gates ? g : gates - 1;
            const long long n = ((long long)(g >> 5) << 6) + (g & 31);   // gate g = W row 64 (g >> 5) + (g & 31), its up row 32 further
            wrow[0] = p.W + n * p.ldw * 2;
            wrow[1] = wrow[0] + 32 * p.ldw * 2;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                int n = (u * NT + t) * TR + cr;
                n = n < p.N ? n : p.N - 1;
                wrow[t] = p.W + (long long)n * p.ldw * 2;
            }
        }
    }
    __device__ __forceinline__ void issue_a(Regs& x, int s) {
        if constexpr (SLOT) {
#pragma unroll
            for (int h = 0; h < AH; ++h) x.a[h] = ld16g(p.A, aoff[h] + (long long)s * (KS * 2));
        }
    }
    __device__ __forceinline__ void issue(Regs& x) {
        const int s = wave + NW * ij;
        const long long koff = (long long)s * (KS * 2) + q * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) { x.w0[t] = ld16g(wrow[t], koff); x.w1[t] = ld16g(wrow[t], koff + 16); }
        issue_a(x, s);
        ++ig;
        if (++ij == spw) { ij = 0; ++iu; if (iu < my_units) set_rows(iu); }
    }
    __device__ __forceinline__ void stage_load(int c0) {
        const int nch = p.K >> 3;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            int cc = c0 + j * 64 + lane;
            cc = cc < nch ? cc : nch - 1;                            // clamped: unconditional loads
#pragma unroll
            for (int ri = 0; ri < NROW; ++ri) {
                const int m = wave + 8 * ri < p.M ? wave + 8 * ri : p.M - 1;
                sv[ri][j] = ld16g(p.A, (long long)m * p.lda * 2 + cc * 16);
            }
            if constexpr (AMODE == SKINNY_A_RMSNORM) sg[j] = ld16g(p.norm_w, cc * 16);
        }
    }
    __device__ __forceinline__ void stage_store(int c0) {
        const int nch = p.K >> 3;
#pragma unroll
        for (int ri = 0; ri < NROW; ++ri) {
            const int m = wave + 8 * ri;
            const bool real = m < p.M;
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int cc = c0 + j * 64 + lane;
                if (cc < nch) {
                    u32x4 o = sv[ri][j];
                    if constexpr (AMODE == SKINNY_A_RMSNORM) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x0 = bflo(sv[ri][j][e]), x1 = bfhi(sv[ri][j][e]);
                            ssq[ri] += x0 * x0;
                            ssq[ri] += x1 * x1;
                            o[e] = pack2(x0 * bflo(sg[j][e]), x1 * bfhi(sg[j][e]));
                        }
                    }
                    if (!real) o = u32x4{0u, 0u, 0u, 0u};
                    st16(aimg + (cc >> 3) * (RM * 128) + (slot_chunk(RM, m, cc & 7) << 4), o);
                }
            }
        }
    }
    // bias / residual of this thread's epilogue items (o = tid (+ 512): reg = o & 3, column = (o >> 2) & 15, row group = (o >> 6) & 3,
    // tile = o >> 8) are fetched when a unit STARTS: behind the combine barrier their L2 / HBM round trip would end every unit.
    __device__ __forceinline__ void fetch_epi(int ui) {
        if constexpr (!PAIR) {
            const int u = (int)blockIdx.x + ui * (int)gridDim.x;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int o = tid + 512 * i;
                const int reg = o & 3, ln = (o >> 2) & 63, nt = o >> 8;
                const int n = (u * NT + nt) * TR + (ln & 15), m = 4 * (ln >> 4) + reg;
                const bool ok = o < NT * 256 && (ln & 15) < TR && n < p.N && m < p.M;
                const int nc = ok ? n : 0, mc = ok ? m : 0;           // clamped: the loads are unconditional (no branch, no drain per element)
                ep_b[i] = p.bias ? ld_bf16g(p.bias, (long long)nc * 2) : 0.f;
                ep_r[i] = p.res ? ld_bf16g(p.res, ((long long)mc * p.ldres + nc) * 2) : 0.f;
            }
        }
    }

    // Indices, the first loads of the image (issued before the weight window: vmcnt retires in order), the window, the first epilogue operands.
    __device__ __forceinline__ void begin() {
        tid = threadIdx.x; lane = tid & 63; wave = tid >> 6;
        c16 = lane & 15; q = lane >> 4;
        TR = p.tile_rows;
        gates = p.N >> 1;
        const int n_units = PAIR ? (gates + TR - 1) / TR : (p.N + NT * TR - 1) / (NT * TR);
        spw = p.K / (KS * NW);                                      // K steps per wave per unit (host: K % 512 == 0)
        my_units = (n_units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
        if (my_units < 0) my_units = 0;
        total = my_units * spw;
        cr = c16 < TR ? c16 : TR - 1;                               // lanes past the share re-read its last row (same lines), their results are dropped
        ST_STAMP(0);
        if constexpr (SLOT) {
#pragma unroll
            for (int h = 0; h < AH; ++h) {
                const int row = (lane >> 3) + 8 * h, piece = lane & 7;
                const int rc = row < p.M ? row : p.M - 1;
                aoff[h] = (long long)rc * p.lda * 2 + piece * 16;
                wofs[h] = slot_chunk(RM, row, piece) << 4;
            }
        }
        {
            const int rr = c16 & (RM - 1);
            rofs0 = slot_chunk(RM, rr, 2 * q) << 4;
            rofs1 = slot_chunk(RM, rr, 2 * q + 1) << 4;
        }
#pragma unroll
        for (int ri = 0; ri < NROW; ++ri) ssq[ri] = 0.f;
        ig = 0; iu = 0; ij = 0;
        // the allowed-id intervals of the greedy pick, once and before the window (img_phase.h: load_am)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool on = p.am_val != nullptr && k < p.am_n_iv;
            am_lo[k] = on ? ld_i32g(p.am_iv, (long long)(2 * k) * 4) : 0;
            am_hi[k] = on ? ld_i32g(p.am_iv, (long long)(2 * k + 1) * 4) : 0;
        }
        // the staging loads go out BEFORE the weight window (vmcnt retires in order: behind the weights they would wait for HBM)
        if constexpr (!SLOT) stage_load(0);
        if (my_units > 0) set_rows(0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (ig < total) issue(r[d]);
        ST_STAMP(1);
        if (my_units > 0) fetch_epi(0);
    }

    __device__ __forceinline__ void finish_unit() {
        float* rp = red + (cu & 1) * (NW * NT * 256);              // two buffers: the next unit's barrier orders the reuse
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            *reinterpret_cast<f32x4*>(rp + (((wave * NT + t) * 64 + lane) << 2)) = acc[t];
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    if constexpr (AMODE == SKINNY_A_RMSNORM) {
                        const float rs = rsqrtf(red_ss[mrow] / (float)p.K + p.norm_eps);
                        g *= rs; uu *= rs;
                    }
                    st_bf16g(p.C, ((long long)mrow * p.ldc + gi) * 2, silu(g) * uu);
                }
            }
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
                        if constexpr (AMODE == SKINNY_A_RMSNORM) v *= rsqrtf(red_ss[mm] / (float)p.K + p.norm_eps);
                        if (p.bias) v += ep_b[i];
                        if (p.res) v += ep_r[i];
                        if (p.C) {
                            if (p.out_f32) st_f32g(p.C, ((long long)mm * p.ldc + nn) * 4, v);
                            else st_bf16g(p.C, ((long long)mm * p.ldc + nn) * 2, v);
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

    // The activations are there: image / slot loads, the K loop over all units, the epilogues.
    __device__ __forceinline__ void run() {
        if constexpr (!SLOT) {
            const int nch = p.K >> 3;
#ifdef AFHIP_STREAM_STAMPS
            { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH * 2 * NT) : "memory"); ST_STAMP(6); }   // the staging loads (issued first) have landed
#endif
            stage_store(0);
#ifdef AFHIP_STREAM_STAMPS
            { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ST_STAMP(7); }
#endif
            for (int c0 = 64 * SB; c0 < nch; c0 += 64 * SB) { stage_load(c0); stage_store(c0); }
            if constexpr (AMODE == SKINNY_A_RMSNORM) {
#pragma unroll
                for (int ri = 0; ri < NROW; ++ri) {
                    const float t = wave_sum(ssq[ri]);
                    if (lane == 0) red_ss[wave + 8 * ri] = (wave + 8 * ri < p.M) ? t : 0.f;
                }
            }
            __syncthreads();
        }
        ST_STAMP(2);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        cu = 0; cj = 0;
        am_best = -INFINITY; am_bi = 0x7fffffff;                   // greedy-decode lm_head: running first-index argmax of this thread's epilogue items

        for (int g0 = 0; g0 < total; g0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (g0 + d < total) {                               // workgroup-uniform: every wave has the same step count
                    u32x4 a0, a1;
                    if constexpr (SLOT) {
                        char* slot = aimg + (wave * DEPTH + d) * (RM * 128);
#pragma unroll
                        for (int h = 0; h < AH; ++h) st16(slot + wofs[h], r[d].a[h]);
                        a0 = *reinterpret_cast<const u32x4*>(slot + rofs0);
                        a1 = *reinterpret_cast<const u32x4*>(slot + rofs1);
                    } else {
                        const int s = wave + NW * cj;
                        a0 = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 128) + rofs0);
                        a1 = *reinterpret_cast<const u32x4*>(aimg + s * (RM * 128) + rofs1);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, r[d].w0[t]), acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, r[d].w1[t]), acc[t], 0, 0, 0);
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
            if (p.am_val) {
                // this thread's items all belong to row 4 * ((tid >> 6) & 3) + (tid & 3): merge over the 16 columns (lane bits 2..5), then
                // over the two waves that share a row group
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
                    st_f32g(p.am_val, (long long)slot * 4, b0);
                    st_f32g(p.am_idx, (long long)slot * 4, __int_as_float(i0));
                }
            }
        }
    }
};

}  // namespace stream
