// Software-pipelined flash attention for gfx950: bf16, head_dim 64, non-causal (the AF-Whisper encoder's 1500 x 1500
// self-attention, modeling_whisper.py:141-218 / 347-441).  afhip_attention dispatches here when attn_pp_eligible().
//
// One 512-thread workgroup = 256 queries of one (batch, head); every wave owns 32 queries for the whole kernel and the
// query lives on the MFMA lane (S^T = K . Q^T, O^T = V^T . P^T; fragment maps shared with attention.hip: key order
// permuted on the K side so a V^T fragment is 8 consecutive keys, V consumed through ds_read_b64_tr_b16).
//
// head_dim 64 attention is VALU-limited (8 MFMAs per 1024 scores), so the kernel is built around the VALU budget:
//   * VALU diet.  Q is pre-multiplied by scale * log2(e) once, so scores arrive in exp2 units.  The running max is
//     LAGGED: the S' MFMA chain starts from an accumulator tile holding -m (m = the max baked in at the last rescale),
//     so S' = S - m comes out of the matrix pipe and P = exp2(S') is ONE instruction per score; m only moves (O, l, S'
//     rescaled) when a tile's row max exceeds it by more than THR = 6 (P <= 64) or on the first tile -- softmax is
//     invariant to the offset.  Row max on v_max3 in four chains, halves merged by v_permlane32_swap.
//   * Each wave overlaps ITS OWN matrix and vector work (cross-wave MFMA || VALU on one SIMD does not pay: measured with
//     a two-group ping-pong build of this kernel, 0.49-0.54 ms per layer against 0.54 for attention.hip).  Iteration j
//     runs, with no dependence between the two columns,
//         MFMA:  O^T += V(j-1)^T . P(j-1)^T        VALU:  row max of S'(j), rare rescale
//         MFMA:  S'(j+1) = K(j+1) . Q^T - m        VALU:  P(j) = exp2(S'(j)), row sum, bf16 pack
//     so two waves per SIMD keep the matrix pipe fed while both issue vector work.
//   * K and V tiles (8 KiB each) arrive by LDS-DMA into two 4-slot rings, K(j+4) and V(j+3) issued in iteration j, with a
//     counted wait (vmcnt(4): two iterations of DMA stay in flight) and ONE barrier per iteration.
//   * masking (per-clip key length) is compiled into the last tile only.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int APP_QT = 256, APP_KT = 64, APP_HD = 64;
constexpr int APP_TILE = APP_KT * 128;        // 8 KiB: 64 keys x 128 B
constexpr int APP_NS = 4;
constexpr int APP_LDS = 2 * APP_NS * APP_TILE;   // K ring + V ring = 64 KiB
constexpr float APP_THR = 6.0f;

struct AttnPP {
    const char* q;
    const char* k;
    const char* v;
    char* o;
    const int32_t* key_len;
    int B, Tq, Tk, n_q, n_kv;
    long long ld_q, ld_kv, ld_o;       // elements
    long long q_bs, kv_bs, o_bs;
    long long q_hs, kv_hs, o_hs;
    float scale_log2;
    int n_xt;
    unsigned long long* dbg;   // DBG & 8: in-kernel s_memtime stamps of block 0 (diagnostic builds only)
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int app_swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

#define APP_BARRIER()                             \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

template <int DBG>
__global__ __launch_bounds__(512) void attn_pp_kernel(AttnPP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // (x-tile, head, batch) from a 1-D grid; workgroups id and id + 8 share an XCD, so the query tiles of one (batch, head)
    // run back to back on one XCD and its K/V comes from HBM once (same mapping as attention.hip)
    const int nh = p.n_q * p.B;
    int xt, hb;
    if ((nh & 7) == 0) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        xt = r % p.n_xt;
        hb = (r / p.n_xt) * 8 + xcd;
    } else {
        xt = blockIdx.x % p.n_xt;
        hb = blockIdx.x / p.n_xt;
    }
    const int b = hb / p.n_q, hq = hb % p.n_q;
    const int hkv = hq / (p.n_q / p.n_kv);
    const int q0 = xt * APP_QT;
    const int qrow = q0 + wave * 32 + fr;
    const int qrow_c = qrow < p.Tq ? qrow : p.Tq - 1;

    const char* kb = p.k + ((long long)b * p.kv_bs + (long long)hkv * p.kv_hs) * 2;
    const char* vb = p.v + ((long long)b * p.kv_bs + (long long)hkv * p.kv_hs) * 2;
    int klen = p.Tk;
    if (p.key_len) { const int kl = p.key_len[b]; klen = kl < klen ? kl : klen; }
    if (klen < 1) klen = 1;
    const int nt = (klen + APP_KT - 1) / APP_KT;
    const bool ragged = (klen & (APP_KT - 1)) != 0;

    // ---- LDS-DMA: this wave fills rows wave*8 .. +8 of a K tile (ring at 0) or a V tile (ring at APP_NS tiles) ----
    const int lrow = lane >> 3, lslot = lane & 7;
    const int drow = wave * 8 + lrow;
    const int kchunk = lslot ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7);   // slot ^ ((row >> 1) & 7)
    const int vchunk = lslot ^ (((lrow >> 1) & 1) << 2);                  // slot ^ (((row >> 1) & 1) << 2)
    const unsigned ldkv2 = (unsigned)(p.ld_kv * 2);
    const unsigned kv_bytes = (unsigned)(p.Tk - 1) * ldkv2 + 128u;
    char* const dma_dst = smem + wave * 1024;
    // row byte offset of key (t * 64 + drow), clamped to the last key: one multiply here, adds afterwards
    const int row_last = (p.Tk - 1) * (int)ldkv2, tile_step = APP_KT * (int)ldkv2;
    const int row0 = drow * (int)ldkv2;
    auto dma_k = [&](int t) __attribute__((always_inline)) {
        int off = row0 + t * tile_step;
        off = off < row_last ? off : row_last;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (int)kv_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dma_dst + (t & (APP_NS - 1)) * APP_TILE), 16, off + kchunk * 16, 0, 0, 0);
    };
    auto dma_v = [&](int t) __attribute__((always_inline)) {
        int off = row0 + t * tile_step;
        off = off < row_last ? off : row_last;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)kv_bytes, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dma_dst + (APP_NS + (t & (APP_NS - 1))) * APP_TILE), 16, off + vchunk * 16, 0, 0, 0);
    };

    // ---- fragment read maps (attention.hip) ----
    const int kx = (app_swap23(fr) >> 1) & 7;                     // swizzle of K tile row ks*32 + swap23(fr)
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, gsel = (lane >> 4) & 1;

    f32x16 ot[2], sta[2], stb[2], negm;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ot[0][e] = 0.f; ot[1][e] = 0.f; negm[e] = 0.f; }
    float m_run = 0.f, l_i = 0.f;
    bf16x8 qf[4], kf[2][4], vf[4][2], pfa[4], pfb[4];

    auto read_k = [&](int t) __attribute__((always_inline)) {
        const char* tile = smem + (t & (APP_NS - 1)) * APP_TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int krow = ks * 32 + app_swap23(fr);
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) kf[ks][dc] = *reinterpret_cast<const bf16x8*>(tile + krow * 128 + (((dc * 2 + fh) ^ kx) << 4));
        }
    };
    auto read_v = [&](int t) __attribute__((always_inline)) {
        const char* Vt = smem + (APP_NS + (t & (APP_NS - 1))) * APP_TILE;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int col = dt * 32 + gsel * 16 + tp * 4;
                const int key0 = s * 16 + fh * 8 + tq, key1 = key0 + 4;
                const char* a0 = Vt + key0 * 128 + (((col >> 3) ^ (((key0 >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
                const char* a1 = Vt + key1 * 128 + (((col >> 3) ^ (((key1 >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
                const s16x8 both = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                vf[s][dt] = __builtin_bit_cast(bf16x8, both);
            }
    };
    auto mfma_pv = [&](bf16x8 (&pf)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) ot[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[s][dt], pf[s], ot[dt], 0, 0, 0);
    };
    auto mfma_qk = [&](f32x16 (&st)[2], bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (first) {
#pragma unroll
                for (int e = 0; e < 16; ++e) st[ks][e] = 0.f;
            } else {
                st[ks] = negm;
            }
        }
#pragma unroll
        for (int dc = 0; dc < 4; ++dc)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) st[ks] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks][dc], qf[dc], st[ks], 0, 0, 0);
    };
    // row max of st (already minus m_run, exp2 units) and the rare move of the running max
    auto track_max = [&](f32x16 (&st)[2], int k0, bool first, bool edge) __attribute__((always_inline)) {
        if (edge) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = k0 + ks * 32 + app_swap23(mfma32_row(e, lane));
                    st[ks][e] = key < klen ? st[ks][e] : -INFINITY;
                }
        }
        float mxa = fmaxf(st[0][0], st[0][1]), mxb = fmaxf(st[1][0], st[1][1]);
        float mxc = fmaxf(st[0][8], st[0][9]), mxd = fmaxf(st[1][8], st[1][9]);
#pragma unroll
        for (int e = 2; e < 8; e += 2) {
            mxa = fmaxf(fmaxf(mxa, st[0][e]), st[0][e + 1]);
            mxb = fmaxf(fmaxf(mxb, st[1][e]), st[1][e + 1]);
            mxc = fmaxf(fmaxf(mxc, st[0][8 + e]), st[0][9 + e]);
            mxd = fmaxf(fmaxf(mxd, st[1][8 + e]), st[1][9 + e]);
        }
        float mx = fmaxf(fmaxf(mxa, mxb), fmaxf(mxc, mxd));
        {
            // lanes l and l ^ 32 hold the two halves of a query's keys: v_permlane32_swap (VALU, no LDS round trip) in inline asm,
            // because the builtin called with two copies of one value is folded away by the compiler
            float ma = mx, mb = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma), "+v"(mb));
            mx = fmaxf(ma, mb);
        }
        if (first || __any(mx > APP_THR)) {
            const float d = first ? mx : fmaxf(mx, 0.f);
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                l_i *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) { ot[0][e] *= alpha; ot[1][e] *= alpha; }
            }
            m_run += d;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[0][e] -= d; st[1][e] -= d; negm[e] = -m_run; }
        }
    };
    // P = exp2(S'), row sum on four accumulators, bf16 pack
    auto exp_pack = [&](f32x16 (&st)[2], bf16x8 (&pf)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 16; ++e) st[ks][e] = (DBG & 2) ? st[ks][e] : __builtin_amdgcn_exp2f(st[ks][e]);
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            ps[0] += st[0][e]; ps[1] += st[0][e + 1];
            ps[2] += st[1][e]; ps[3] += st[1][e + 1];
        }
        l_i += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[s][j] = (bf16)st[s >> 1][8 * (s & 1) + j];
    };

    // ---- prologue: K(0..3), V(0..2) in flight; Q fragments pre-multiplied by scale * log2(e) ----
    dma_k(0); dma_k(1); dma_v(0); dma_k(2); dma_v(1); dma_k(3); dma_v(2);
    {
        const bf16* qp = reinterpret_cast<const bf16*>(p.q) + (long long)b * p.q_bs + (long long)hq * p.q_hs + (long long)qrow_c * p.ld_q;
#pragma unroll
        for (int dc = 0; dc < 4; ++dc) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(qp + dc * 16 + fh * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[dc][e] = (bf16)((float)raw[e] * p.scale_log2);
        }
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // K(0), K(1), V(0) of this wave have landed
    APP_BARRIER();
    read_k(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    APP_BARRIER();                                       // every wave holds K(0) in registers before iteration 0 re-fills K slot 0
    mfma_qk(sta, true);

    // iteration j: st = S'(j) (in), stn = S'(j+1) (out), pfp = P(j-1) (in), pfc = P(j) (out)
    auto stamp = [&](int j, int k) __attribute__((always_inline)) {
        if constexpr (DBG & 8) {
            __builtin_amdgcn_sched_barrier(0);
            if (blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && j >= 8 && j < 12)
                p.dbg[((wave >> 2) * 4 + (j - 8)) * 8 + k] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto iteration = [&](int j, f32x16 (&st)[2], f32x16 (&stn)[2], bf16x8 (&pfp)[4], bf16x8 (&pfc)[4]) __attribute__((always_inline)) {
        stamp(j, 0);
        dma_k(j + 4);
        dma_v(j + 3);
        read_k(j + 1);
        if (j > 0) mfma_pv(pfp);
        stamp(j, 1);
        track_max(st, j * APP_KT, j == 0, ragged && j == nt - 1);
        stamp(j, 2);
        read_v(j);
        mfma_qk(stn, false);
        exp_pack(st, pfc);
        if constexpr (DBG == 1) {   // explicit MFMA : LDS : VALU interleave pattern -- measured 8-10 % SLOWER than the compiler's own order
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads (the 16 transposed V reads)
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // 5 VALU (an MFMA leaves 24 of its 32 cycles to the issue port)
            }
        }
        stamp(j, 3);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K(j+2), V(j+1) have landed (this wave's part); two iterations of DMA stay in flight
        stamp(j, 4);
        APP_BARRIER();
        stamp(j, 5);
    };
    for (int j = 0; j < nt; j += 2) {
        iteration(j, sta, stb, pfb, pfa);
        if (j + 1 < nt) iteration(j + 1, stb, sta, pfa, pfb);
    }
    // ---- the last P . V ----
    if ((nt - 1) & 1) mfma_pv(pfb); else mfma_pv(pfa);

    // ---- normalise and write O[query, d] ----
    const float l_tot = l_i + __shfl_xor(l_i, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (qrow < p.Tq) {
        bf16* op = reinterpret_cast<bf16*>(p.o) + (long long)b * p.o_bs + (long long)qrow * p.ld_o + (long long)hq * p.o_hs;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * fh;
                bf16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (bf16)(ot[dt][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(op + d) = w;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}

}  // namespace

bool attn_pp_eligible(const afhip_attn_args* a) {
    // OPT-IN (AFHIP_ATTN_PP=1).  Alone it runs 0.52 ms per encoder layer against 0.54 ms for attention.hip; inside the full
    // encoder step (same box, back to back) it is 0.5 ms per step SLOWER, so attention.hip stays the default.  With
    // q_prescaled (scale folded into the q projection, as the LayerNorm-folded encoder does) it costs no accuracy; on plain q
    // it adds a second bf16 rounding of q (max |err| 2.3e-3 vs 1.1e-3 on N(0,1) inputs).
    const char* e = getenv("AFHIP_ATTN_PP");
    if (!(e && e[0] == '1')) return false;
    if (a->dtype != AFHIP_BF16 || a->hd != 64 || a->causal || a->key_split > 0) return false;
    if (a->Tq < 64 || a->Tk < 64) return false;
    if ((long long)a->Tk * a->ld_kv * 2 >= (1ll << 31)) return false;
    return true;
}

int attn_pp_launch(const afhip_attn_args* a, hipStream_t s) {
    AttnPP p;
    p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.o = (char*)a->out;
    p.key_len = a->key_len;
    p.B = a->B; p.Tq = a->Tq; p.Tk = a->Tk; p.n_q = a->n_q; p.n_kv = a->n_kv;
    p.ld_q = a->ld_q; p.ld_kv = a->ld_kv; p.ld_o = a->ld_o;
    p.q_bs = a->q_batch_stride; p.kv_bs = a->kv_batch_stride; p.o_bs = a->o_batch_stride;
    p.q_hs = a->q_head_stride; p.kv_hs = a->kv_head_stride; p.o_hs = a->o_head_stride > 0 ? a->o_head_stride : a->hd;
    p.scale_log2 = a->q_prescaled ? 1.0f : a->scale * 1.4426950408889634f;
    p.n_xt = cdiv(a->Tq, APP_QT);
    AFHIP_CHECK((long long)p.n_xt * a->n_q * a->B < (1ll << 31), "afhip_attention: grid too large");
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done)) {
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, APP_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, APP_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, APP_LDS);
        (void)hipFuncSetAttribute((const void*)attn_pp_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, APP_LDS);
    }
    const char* de = getenv("AFHIP_ATTN_DBG");
    const int dbg = de ? atoi(de) : 0;
    const char* dp = getenv("AFHIP_ATTN_DBGPTR");
    p.dbg = dp ? (unsigned long long*)strtoull(dp, nullptr, 0) : nullptr;
    const dim3 grid((unsigned)(p.n_xt * a->n_q * a->B));
    if (dbg == 1) hipLaunchKernelGGL(attn_pp_kernel<1>, grid, dim3(512), APP_LDS, s, p);
    else if (dbg == 2) hipLaunchKernelGGL(attn_pp_kernel<2>, grid, dim3(512), APP_LDS, s, p);
    else if (dbg == 8 && p.dbg) hipLaunchKernelGGL(attn_pp_kernel<8>, grid, dim3(512), APP_LDS, s, p);
    else hipLaunchKernelGGL(attn_pp_kernel<0>, grid, dim3(512), APP_LDS, s, p);
    return 0;
}
