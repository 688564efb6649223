// Encoder self-attention at head_dim 64 (bf16, q prescaled to exp2 units, per-clip key length): the ONE-WAVE-PER-SIMD form.
// Same arithmetic as attn_kernel<bf16, 64, 2, LAG> (attention.hip): S^T = K . Q^T - m_lag with the query on the MFMA lane, P = exp2(S')
// straight out of the accumulators, O^T += V^T . P^T, lagged row maximum (raised only when a tile's partial row sum exceeds 2^16).
//
// Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"; MI355X_MICROARCH.md issue prices):
//   * workgroup = 4 waves = 256 queries of one (clip, head); a wave owns 64 queries as TWO 32-query blocks a, b and the whole
//     512-register file.  A K or V fragment read from LDS serves both blocks: 16 KB of LDS reads per 32 MFMAs instead of per 16.
//   * K / V tiles of 64 keys arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction) into a ring of 4 stages, three
//     tiles ahead of the arithmetic; ONE barrier per key tile, counted vmcnt (4 DMA instructions per wave per tile).
//   * per tile the wave runs two SLOTS of 16 MFMAs, each beside the softmax (32 x v_exp_f32, 32 x v_add_f32, 16 x v_cvt_pk) of ONE
//     query block, software-pipelined across tiles so that every softmax has 16 independent MFMAs to hide behind:
//         slot 1:  S_b(t) = K(t).Q_b^T     O_b += V(t-1).P_b(t-1)     ||  softmax of S_a(t)   -> P_a(t)
//         slot 2:  S_a(t+1) = K(t+1).Q_a^T  O_a += V(t).P_a(t)        ||  softmax of S_b(t)   -> P_b(t)
//     (issue budget per MFMA gap: 2 exp (16 cycles) + 2 add + 1 cvt (12) + the MFMA's own 8 = 36 of 32: the VALU, not the matrix pipe,
//     is the floor at head_dim 64 -- 0.89 of the MFMA roof if perfectly packed.)
//   * the rare "raise the lag" path sits in its own basic block at the END of a slot, so the slots stay straight-line code.
//   * O leaves through LDS as whole 128-byte rows (16 B per lane) instead of 8-byte row-strided stores.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int E_QT = 256, E_KT = 64, E_HD = 64;
constexpr int E_ROWB = E_HD * 2;                 // bytes per K / V row in LDS
constexpr int E_TILEB = E_KT * E_ROWB;           // 8 KiB
constexpr int E_STAGEB = 2 * E_TILEB;            // K tile + V tile
constexpr int E_NST = 4;
constexpr int E_LDS = E_NST * E_STAGEB;          // 64 KiB
constexpr float E_LAG_LIMIT = 65536.f;

struct EncAttnP {
    const char* q; const char* k; const char* v; char* o;
    const int32_t* key_len;     // [B] keys per clip (NULL: Tk)
    const int32_t* row_off;     // packed batches: first row of clip b (then key_len[b] = queries = keys); NULL: [B, T] batches
    int B, Tq, Tk, n_h;
    long long ld_q, ld_kv, ld_o;        // elements
    long long q_bs, kv_bs, o_bs;
    long long q_hs, kv_hs, o_hs;
    int n_qt;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int e_swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ uint32_t e_cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define E_BARRIER()                               \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// VAR: tuning / ablation switches (AFHIP_ENC64_VAR): bit 0 = softmax VALU software-pipelined by one MFMA gap (exps of pair i beside the
// adds / cvt of pair i-1, two partial sums); bit 1 = the tile's 4 DMA instructions spread over slot 2's gaps instead of a burst at the top;
// bit 2 (TIMING ONLY, wrong results) = no softmax; bit 3 (TIMING ONLY) = no DMA in the loop
template <int VAR>
__global__ __launch_bounds__(256, 1) void attn_enc64_kernel(EncAttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    // (query tile, head, clip) with the x-tiles of one (clip, head) back to back on ONE XCD (attention.hip)
    const int nh = p.n_h * p.B;
    int xt, hb;
    if ((nh & 7) == 0) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        xt = r % p.n_qt;
        hb = (r / p.n_qt) * 8 + xcd;
    } else {
        xt = blockIdx.x % p.n_qt;
        hb = blockIdx.x / p.n_qt;
    }
    const int b = hb / p.n_h, h = hb % p.n_h;
    const int q0 = xt * E_QT;
    long long q_off = (long long)b * p.q_bs, kv_off = (long long)b * p.kv_bs, o_off = (long long)b * p.o_bs;
    int Tq = p.Tq, klen = p.Tk;
    if (p.key_len) { const int kl = p.key_len[b]; klen = kl < klen ? kl : klen; }
    if (p.row_off) {
        const long long r = p.row_off[b];
        q_off = r * p.ld_q; kv_off = r * p.ld_kv; o_off = r * p.ld_o;
        Tq = p.key_len[b];
    }
    if (q0 >= Tq) return;                  // packed batches: query tile past this clip (workgroup-uniform, before any barrier)
    const char* qb_ = p.q + (q_off + (long long)h * p.q_hs) * 2;
    const char* kb_ = p.k + (kv_off + (long long)h * p.kv_hs) * 2;
    const char* vb_ = p.v + (kv_off + (long long)h * p.kv_hs) * 2;
    const int nt = klen > 0 ? (klen + E_KT - 1) / E_KT : 0;

    // ---- LDS-DMA (buffer_load_dwordx4 ... lds) as INLINE ASM: hipcc's waitcnt pass drains every LDS-DMA it knows about (vmcnt(0)) in
    //      front of the transposed LDS reads below; issued from asm the stream is invisible to it and stays in flight behind the counted
    //      vmcnt of the tile loop (cdna guide 5.7).  A wave-instruction fills 8 LDS rows of 128 B; lane (lrow, slot) writes slot `slot` of
    //      row r0 + lrow with SOURCE chunk slot ^ swz(row) (linear image, swizzle on the source address).  Wave w owns rows 16 w .. 16 w + 15
    //      of every K and V tile (two instructions each).  Keys past klen re-read the last live row (0 x garbage must stay 0). ----
    const int lrow = lane >> 3, lslot = lane & 7;
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    auto make_rsrc = [&](const char* base, int nrec) {
        const unsigned long long a = (unsigned long long)base;
        v4i_t r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
        r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));      // stride 0
        r[2] = __builtin_amdgcn_readfirstlane(nrec);                           // num_records (bytes): offsets at or past it read as 0
        r[3] = 0x00020000;
        return r;
    };
    const int ldkv2 = (int)(p.ld_kv * 2);
    // K rows past klen re-read the last live row (their scores equal the last live key's: finite, and removed from the row sums below);
    // V rows past klen are OUT OF RANGE of the V descriptor and arrive as zeros, so those keys add nothing to O
    const v4i_t krsrc = make_rsrc(kb_, 0x7fffffff), vrsrc = make_rsrc(vb_, (klen > 0 ? (klen - 1) * ldkv2 : 0) + E_ROWB);
    const int lds_base = (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring
    auto dma16 = [&](const v4i_t& rsrc, int voff, int lds_addr) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
    };
    // one of the 4 DMA instructions of tile t (j = 0..3: K rows 0-7, V rows 0-7, K rows 8-15, V rows 8-15 of this wave's 16 rows)
    auto dma_piece = [&](int t, const int j) {
        const int tt = t < nt ? t : nt - 1;
        const int base = __builtin_amdgcn_readfirstlane(lds_base + (t & (E_NST - 1)) * E_STAGEB + wave * 2048);
        const int u = j >> 1;
        const int row = wave * 16 + u * 8 + lrow;
        const int key = tt * E_KT + row;
        if ((j & 1) == 0) {
            const int keyc = key < klen ? key : klen - 1;
            dma16(krsrc, keyc * ldkv2 + (lslot ^ ((row >> 1) & 7)) * 16, base + u * 1024);
        } else {
            dma16(vrsrc, key * ldkv2 + (lslot ^ (((row >> 1) & 1) << 2)) * 16, base + E_TILEB + u * 1024);
        }
    };
    auto dma_tile = [&](int t, int stage) {
        const int ks = __builtin_amdgcn_readfirstlane(lds_base + stage * E_STAGEB + wave * 2048);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = wave * 16 + u * 8 + lrow;
            const int key = t * E_KT + row;
            const int keyc = key < klen ? key : klen - 1;
            const int kc = lslot ^ ((row >> 1) & 7);                 // rswz<128>
            const int vc = lslot ^ (((row >> 1) & 1) << 2);          // vtrswz<128>
            dma16(krsrc, keyc * ldkv2 + kc * 16, ks + u * 1024);
            dma16(vrsrc, key * ldkv2 + vc * 16, ks + E_TILEB + u * 1024);
        }
    };
    // tiles past the end re-fetch the last one: the vmcnt bookkeeping stays uniform and every stage always holds finite data
    auto dma_clamped = [&](int t) { dma_tile(t < nt ? t : nt - 1, t & (E_NST - 1)); };

    // ---- the accumulator file is ASM-OWNED (literal register names; hipcc allocates none of it: its own choice put S' in AGPRs, 64
    //      v_accvgpr_read per slot, and bounced fragments between the files).  Map:
    //        a[0:31]   O_a^T (d tiles 0, 1)      a[32:63]  O_b^T          a[64:79] Q_a fragments (4 x 4)   a[80:95] Q_b
    //        a[96:127] K fragment set A (8 x 4)  a[128:159] set B        a[160:191] V fragment set A      a[192:223] set B
    //      S', the chains' C input and the packed P stay in compiler-managed arch VGPRs (they are VALU operands). ----
    constexpr int A_OA = 0, A_OB = 32, A_QA = 64, A_QB = 80, A_KF = 96, A_VF = 160;
// (immediates above 64 print in hex: `a[0x50]` assembles, `a0x50` does not -- always the bracket form)
#define E_ACC_WRITE(IDX, VAL) asm volatile("v_accvgpr_write_b32 a[%0], %1" :: "n"(IDX), "v"(VAL) : "a255")
#define E_ACC_READ(DST, IDX) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(DST) : "n"(IDX))
    // ---- Q fragments (B operand of S^T), both query blocks, resident in a[64:95] ----
    {
        int ra = q0 + wave * 64 + fr, rb = ra + 32;
        ra = ra < Tq ? ra : Tq - 1;
        rb = rb < Tq ? rb : Tq - 1;
        const __bf16* pa = reinterpret_cast<const __bf16*>(qb_) + (long long)ra * p.ld_q;
        const __bf16* pb = reinterpret_cast<const __bf16*>(qb_) + (long long)rb * p.ld_q;
        u32x4 qa[4], qb[4];
#pragma unroll
        for (int dc = 0; dc < 4; ++dc) {
            qa[dc] = *reinterpret_cast<const u32x4*>(pa + dc * 16 + fh * 8);
            qb[dc] = *reinterpret_cast<const u32x4*>(pb + dc * 16 + fh * 8);
        }
        static_for<0, 16>([&](auto it) {
            constexpr int i = decltype(it)::value;
            const uint32_t xa = qa[i >> 2][i & 3], xb = qb[i >> 2][i & 3];    // (asm operands inside a generic lambda do not capture by themselves)
            E_ACC_WRITE(A_QA + i, xa);
            E_ACC_WRITE(A_QB + i, xb);
        });
    }
    if (nt == 0) {
        // no live key: zeros (l = 0), exactly what the plain kernel stores
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int qrow = q0 + wave * 64 + blk * 32 + fr;
            if (qrow < Tq) {
                __bf16* op = reinterpret_cast<__bf16*>(p.o) + o_off + (long long)qrow * p.ld_o + (long long)h * p.o_hs;
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(op + fh * 32 + c * 8) = u32x4{0, 0, 0, 0};
            }
        }
        return;
    }
    // O = 0, V fragment sets = 0 (the first slot multiplies V(-1) = 0 by P_b(-1) = 0)
    {
        const uint32_t zero = 0u;
        static_for<0, 64>([&](auto it) { const uint32_t z = zero; E_ACC_WRITE(A_OA + decltype(it)::value, z); });
        static_for<0, 64>([&](auto it) { const uint32_t z = zero; E_ACC_WRITE(A_VF + decltype(it)::value, z); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the Q loads are the only compiler-counted VMEM ops: retired before the
    __builtin_amdgcn_sched_barrier(0);                          // hand-counted DMA stream starts

    f32x16 sa[2], sb[2];                    // S'^T [key sub-tile]                            (VGPR)
    f32x16 ca, cb;                          // C input of each chain: -m_lag                  (VGPR)
    u32x4 pka[4], pkb[4];                   // P as packed bf16: the B operand of PV k-step s (VGPR)
    float ml_a = 0.f, ml_b = 0.f, l_a = 0.f, l_b = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ca[e] = 0.f; cb[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { pka[i] = u32x4{0, 0, 0, 0}; pkb[i] = u32x4{0, 0, 0, 0}; }

#define E_FENCE() __builtin_amdgcn_sched_barrier(0)
    // MFMA forms (KF / Q / VF / O = first register of the operand's tuple in the accumulator file)
#define E_MFMA_S_FIRST(S, KF, Q, C) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, a[%2:%3], a[%4:%5], %1" : "=&v"(S) : "v"(C), "n"(KF), "n"((KF) + 3), "n"(Q), "n"((Q) + 3))
#define E_MFMA_S_ACC(S, KF, Q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(S) : "n"(KF), "n"((KF) + 3), "n"(Q), "n"((Q) + 3))
#define E_MFMA_O(O, VF, PF) asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]" :: "v"(PF), "n"(O), "n"((O) + 15), "n"(VF), "n"((VF) + 3) : "memory")

    // fragment read addresses: lane part in a VGPR, (ring stage, sub-tile, k-step) part in the instruction's 16-bit offset
    const int krow0 = e_swap23(fr);
    int kaddr[4];
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) kaddr[dc] = lds_base + krow0 * E_ROWB + (((dc * 2 + fh) ^ ((krow0 >> 1) & 7)) << 4);   // sub-tile 1: + 32 rows = + 4096
    int vaddr[2];
    {
        const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, gsel = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + gsel * 16 + tp * 4;
            const int key0 = fh * 8 + tq;                       // k-step s: + 16 rows = + 2048; second half: + 4 rows = + 512 (same swizzle)
            vaddr[dt] = lds_base + E_TILEB + key0 * E_ROWB + (((col >> 3) ^ (((key0 >> 1) & 1) << 2)) << 4) + (col & 7) * 2;
        }
    }
    // fragment i of K set SET from ring stage STAGE: ds_read_b128 straight into a[..]
#define E_READ_K(SET, I, STAGE) asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(kaddr[(I) & 3]), "n"(A_KF + (SET) * 32 + (I) * 4), "n"(A_KF + (SET) * 32 + (I) * 4 + 3), \
                                             "n"((STAGE) * E_STAGEB + ((I) >> 2) * 4096) : "memory")
    // fragment i = (k-step i >> 1, d tile i & 1) of V set SET: two transposed 8-byte reads into the halves of a[..]
#define E_READ_V(SET, I, STAGE) asm volatile("ds_read_b64_tr_b16 a[%1:%2], %0 offset:%5\n\tds_read_b64_tr_b16 a[%3:%4], %0 offset:%6" :: "v"(vaddr[(I) & 1]), \
                                             "n"(A_VF + (SET) * 32 + (I) * 4), "n"(A_VF + (SET) * 32 + (I) * 4 + 1), "n"(A_VF + (SET) * 32 + (I) * 4 + 2), "n"(A_VF + (SET) * 32 + (I) * 4 + 3), \
                                             "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048), "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048 + 512) : "memory")

    // the two scores of MFMA gap i of a slot: P = exp2(S'), row-sum partial, packed bf16.  pk[s][j] holds registers 8 (s & 1) + 2 j, + 1
    // of sub-tile s >> 1, i.e. pk[s] is the 8 keys of PV k-step s.
    float pend0 = 0.f, pend1 = 0.f, psum2 = 0.f;     // VAR & 1: the pair exponentiated in the previous gap, second partial sum
    auto soft2 = [&](const int i, const f32x16 (&s)[2], u32x4 (&pk)[4], float& psum) __attribute__((always_inline)) {
        if constexpr (VAR & 4) return;
        const int ks = i >> 3, e = 2 * (i & 7);
        if constexpr (VAR & 1) {
            // gap i: exps of pair i; adds + cvt of pair i - 1 (pair 15 is finished by soft_tail)
            const float q0 = pend0, q1 = pend1;
            pend0 = __builtin_amdgcn_exp2f(s[ks][e]);
            pend1 = __builtin_amdgcn_exp2f(s[ks][e + 1]);
            if (i > 0) {
                const int j = i - 1;
                psum += q0;
                psum2 += q1;
                uint32_t w = e_cvt_pk(q0, q1);
                asm volatile("" : "+v"(w));
                pk[(j >> 3) * 2 + ((j & 7) >> 2)][j & 3] = w;
            }
        } else {
            const float p0 = __builtin_amdgcn_exp2f(s[ks][e]), p1 = __builtin_amdgcn_exp2f(s[ks][e + 1]);
            psum += p0;
            psum += p1;
            uint32_t w = e_cvt_pk(p0, p1);
            asm volatile("" : "+v"(w));        // pinned here: the rare path overwrites P, and hipcc would sink the slot's 16 v_cvt_pk below its branch
            pk[ks * 2 + ((i & 7) >> 2)][i & 3] = w;
        }
    };
    auto soft_tail = [&](u32x4 (&pk)[4], float& psum) __attribute__((always_inline)) {
        if constexpr ((VAR & 1) && !(VAR & 4)) {
            psum += pend0;
            psum2 += pend1;
            uint32_t w = e_cvt_pk(pend0, pend1);
            asm volatile("" : "+v"(w));
            pk[3][3] = w;
            psum += psum2;
            psum2 = 0.f;
        }
    };
    // the rare path: raise the lag of one query block (O at a[OBASE .. OBASE + 31]) to the true maximum of this tile and redo its P
    auto rebase = [&](auto obase_tag, const f32x16 (&s)[2], u32x4 (&pk)[4], f32x16& c, float& ml, float& l, float& d_out) -> float {
        constexpr int OBASE = decltype(obase_tag)::value;
        float mx = -INFINITY;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[ks][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float d = mx > 0.f ? mx : 0.f;
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l *= alpha;
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");        // the block's last O MFMA -> v_accvgpr_read (18 wait states)
        static_for<0, 32>([&](auto it) {
            constexpr int i = decltype(it)::value;
            float x;
            E_ACC_READ(x, OBASE + i);
            x *= alpha;
            E_ACC_WRITE(OBASE + i, x);
        });
        asm volatile("s_nop 1" ::: "memory");
        ml += d;
        d_out = d;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = -ml;
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ks = i >> 3, e = 2 * (i & 7);
            const float p0 = __builtin_amdgcn_exp2f(s[ks][e] - d), p1 = __builtin_amdgcn_exp2f(s[ks][e + 1] - d);
            psum += p0;
            psum += p1;
            pk[ks * 2 + ((i & 7) >> 2)][i & 3] = e_cvt_pk(p0, p1);
        }
        return psum;
    };
    // Last, partly filled tile: the keys past klen are copies of key klen - 1 (same K row -> bit-identical S'), with zero V rows.  They add
    // nothing to O; their equal terms exp2(S'_last - d) are taken out of this lane's row-sum partial here (each half of a lane pair sums 32
    // of the tile's 64 keys: count the copies this half holds).  Runs once per query block per workgroup, in a basic block of its own.
    auto dup_sum = [&](const f32x16 (&s)[2], float d) -> float {
        const int kk = (klen - 1) - (nt - 1) * E_KT;       // position of the last live key inside the last tile
        float v = 0.f;
        int mine = 0, cnt = 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = ks * 32 + e_swap23(mfma32_row(e, lane));
                v = key == kk ? s[ks][e] : v;
                mine |= key == kk ? 1 : 0;
                cnt += key > kk ? 1 : 0;
            }
        const float vo = __shfl_xor(v, 32, 64);
        const float sl = mine ? v : vo;                    // the last live key sits in exactly one half of the pair
        return (float)cnt * __builtin_amdgcn_exp2f(sl - d);
    };
    const bool has_edge = (klen % E_KT) != 0;

    // ---- prologue: tiles 0, 1, 2 on their way; the true row maxima of tile 0 become the lags; S'_a(0) ----
    dma_clamped(0);
    dma_clamped(1);
    dma_clamped(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // tile 0 of this wave has landed
    E_BARRIER();                                               // ... and everybody's
    static_for<0, 8>([&](auto it) { constexpr int i = decltype(it)::value; (void)&kaddr; E_READ_K(0, i, 0); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    E_FENCE();
    // S' chains of one block (KSET = fragment set, Q = first register of its Q fragments)
#define E_CHAINS(S, KSET, Q, C)                                                              \
    do {                                                                                      \
        E_MFMA_S_FIRST(S[0], A_KF + (KSET) * 32, Q, C);                                       \
        E_MFMA_S_ACC(S[0], A_KF + (KSET) * 32 + 4, (Q) + 4);                                  \
        E_MFMA_S_ACC(S[0], A_KF + (KSET) * 32 + 8, (Q) + 8);                                  \
        E_MFMA_S_ACC(S[0], A_KF + (KSET) * 32 + 12, (Q) + 12);                                \
        E_MFMA_S_FIRST(S[1], A_KF + (KSET) * 32 + 16, Q, C);                                  \
        E_MFMA_S_ACC(S[1], A_KF + (KSET) * 32 + 20, (Q) + 4);                                 \
        E_MFMA_S_ACC(S[1], A_KF + (KSET) * 32 + 24, (Q) + 8);                                 \
        E_MFMA_S_ACC(S[1], A_KF + (KSET) * 32 + 28, (Q) + 12);                                \
    } while (0)
    {
        E_CHAINS(sa, 0, A_QA, ca);
        E_CHAINS(sb, 0, A_QB, cb);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");        // MFMA results -> VALU readers (18 wait states for a 16-pass MFMA)
        E_FENCE();
        float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 16; ++e) { ma = fmaxf(ma, sa[ks][e]); mb = fmaxf(mb, sb[ks][e]); }
        ma = fmaxf(ma, __shfl_xor(ma, 32, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        ml_a = ma; ml_b = mb;                                  // finite: every key row of tile 0 is a live row (copies included)
#pragma unroll
        for (int e = 0; e < 16; ++e) { ca[e] = -ml_a; cb[e] = -ml_b; }
        E_FENCE();
        E_CHAINS(sa, 0, A_QA, ca);
    }

    // one MFMA of a slot: MFMAs 0..7 are the S' chains (sub-tile 0: 0..3, sub-tile 1: 4..7) of the block whose Q sits at QB, from K set
    // KSET; 8..15 are O (at OB) += V (set VSET) . P
#define E_SLOT_MFMA(I, S, KSET, QB, C, OB, VSET, PK)                                                           \
    do {                                                                                                        \
        if constexpr ((I) < 8) {                                                                                \
            if constexpr (((I) & 3) == 0) E_MFMA_S_FIRST(S[(I) >> 2], A_KF + (KSET) * 32 + (I) * 4, QB, C);     \
            else E_MFMA_S_ACC(S[(I) >> 2], A_KF + (KSET) * 32 + (I) * 4, (QB) + ((I) & 3) * 4);                \
        } else {                                                                                                \
            E_MFMA_O((OB) + (((I) - 8) & 1) * 16, A_VF + (VSET) * 32 + ((I) - 8) * 4, PK[((I) - 8) >> 1]);     \
        }                                                                                                       \
    } while (0)

    // one key tile; the loop is unrolled four times so that the ring stage (t & 3) is a compile-time constant and PAR = t & 1 names the
    // fragment sets: set PAR holds K(t) / V(t-1), set PAR ^ 1 receives K(t+1) / V(t).  Every MFMA gap is pinned by a scheduling fence:
    // [MFMA] [2 v_exp, 2 v_add, 1 v_cvt_pk] [<= 3 LDS reads].
    auto iter = [&](const int t, auto idx_tag) __attribute__((always_inline)) {
        constexpr int IDX = decltype(idx_tag)::value;
        constexpr int PAR = IDX & 1, NXT = PAR ^ 1;
        // tile t+1 must have landed before its K fragments are read in slot 1; tile t+2 may stay in flight (4 DMA instructions)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        E_BARRIER();
        if constexpr (!(VAR & 2) && !(VAR & 8)) dma_clamped(t + 3);   // stage (t+3)&3 = (t-1)&3: last read (V(t-1)) before the barrier above
        E_FENCE();
        // ---- slot 1: S'_b(t), O_b += V(t-1).P_b(t-1)  ||  softmax a(t)  ||  fragment reads K(t+1), V(t) into the other set ----
        float ps = 0.f, dsh = 0.f;
        static_for<0, 16>([&](auto it) {
            constexpr int i = decltype(it)::value;
            (void)&sb; (void)&cb; (void)&pkb; (void)&kaddr; (void)&vaddr;
            E_SLOT_MFMA(i, sb, PAR, A_QB, cb, A_OB, PAR, pkb);
            soft2(i, sa, pka, ps);
            if constexpr (i < 8) { E_READ_K(NXT, i, (IDX + 1) & (E_NST - 1)); E_READ_V(NXT, i, IDX); }
            E_FENCE();
        });
        soft_tail(pka, ps);
        if (__any(!(ps <= E_LAG_LIMIT))) ps = rebase(std::integral_constant<int, A_OA>{}, sa, pka, ca, ml_a, l_a, dsh);
        if (has_edge && t == nt - 1) { asm volatile("" ::: "memory"); ps -= dup_sum(sa, dsh); }
        l_a += ps;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the fragments read in slot 1 are in their registers
        E_FENCE();
        // ---- slot 2: S'_a(t+1), O_a += V(t).P_a(t)  ||  softmax b(t) ----
        ps = 0.f; dsh = 0.f;
        static_for<0, 16>([&](auto it) {
            constexpr int i = decltype(it)::value;
            (void)&sa; (void)&ca; (void)&pka;
            E_SLOT_MFMA(i, sa, NXT, A_QA, ca, A_OA, NXT, pka);
            soft2(i, sb, pkb, ps);
            if constexpr ((VAR & 2) && !(VAR & 8) && (i & 3) == 2) dma_piece(t + 3, i >> 2);   // gaps 2, 6, 10, 14
            E_FENCE();
        });
        soft_tail(pkb, ps);
        if (__any(!(ps <= E_LAG_LIMIT))) ps = rebase(std::integral_constant<int, A_OB>{}, sb, pkb, cb, ml_b, l_b, dsh);
        if (has_edge && t == nt - 1) { asm volatile("" ::: "memory"); ps -= dup_sum(sb, dsh); }
        l_b += ps;
        E_FENCE();
    };
    for (int t = 0; t < nt; t += 4) {
        iter(t, std::integral_constant<int, 0>{});
        if (t + 1 < nt) iter(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nt) iter(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < nt) iter(t + 3, std::integral_constant<int, 3>{});
    }
    // drain: O_b += V(nt-1) . P_b(nt-1); V(nt-1) sits in the fragment set the LAST iteration filled (set PAR ^ 1 of t = nt - 1)
    if (nt & 1) {
        static_for<8, 16>([&](auto it) { constexpr int i = decltype(it)::value; (void)&sb; (void)&cb; (void)&pkb; E_SLOT_MFMA(i, sb, 0, A_QB, cb, A_OB, 1, pkb); });
    } else {
        static_for<8, 16>([&](auto it) { constexpr int i = decltype(it)::value; (void)&sb; (void)&cb; (void)&pkb; E_SLOT_MFMA(i, sb, 0, A_QB, cb, A_OB, 0, pkb); });
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");   // no LDS-DMA may outlive the K / V ring (it becomes the O staging
    E_BARRIER();                                                                  // area); the last MFMA results are readable

    // ---- normalise, stage O through LDS, store whole rows.  O^T register e of d-tile dt is d = dt*32 + (e&3) + 8*(e>>2) + 4*fh of query
    //      (lane & 31): the lane writes 4 consecutive d (8 bytes) at [query][d]; rows are 128 B + 16 B pad (conflict-free 8-byte stores) ----
    const float la = l_a + __shfl_xor(l_a, 32, 64), lb = l_b + __shfl_xor(l_b, 32, 64);
    const float ia = la > 0.f ? 1.0f / la : 0.f, ib = lb > 0.f ? 1.0f / lb : 0.f;
    constexpr int OROW = 144;
    char* ost = smem + wave * (64 * OROW);
    static_for<0, 16>([&](auto it) {
        constexpr int i = decltype(it)::value;                   // (block, d tile, group of 4 registers)
        constexpr int blk = i >> 3, dt = (i >> 2) & 1, g = i & 3;
        float o0, o1, o2, o3;
        E_ACC_READ(o0, blk * 32 + dt * 16 + 4 * g);
        E_ACC_READ(o1, blk * 32 + dt * 16 + 4 * g + 1);
        E_ACC_READ(o2, blk * 32 + dt * 16 + 4 * g + 2);
        E_ACC_READ(o3, blk * 32 + dt * 16 + 4 * g + 3);
        const float inv = blk ? ib : ia;
        const uint32_t w0 = e_cvt_pk(o0 * inv, o1 * inv), w1 = e_cvt_pk(o2 * inv, o3 * inv);
        const int d = dt * 32 + 8 * g + 4 * fh;
        *reinterpret_cast<uint2*>(ost + (blk * 32 + fr) * OROW + d * 2) = uint2{w0, w1};
    });
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's own stores (its rows are private to it)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 8 + (lane >> 3), c = lane & 7;
        const int qrow = q0 + wave * 64 + r;
        const u32x4 val = *reinterpret_cast<const u32x4*>(ost + r * OROW + c * 16);
        if (qrow < Tq) {
            __bf16* op = reinterpret_cast<__bf16*>(p.o) + o_off + (long long)qrow * p.ld_o + (long long)h * p.o_hs;
            *reinterpret_cast<u32x4*>(op + c * 8) = val;
        }
    }
}

}  // namespace

// Called by afhip_attention (attention.hip) for the shape this form covers; returns false when it does not apply.
bool afhip_attention_enc64(const afhip_attn_args* a, hipStream_t s) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("AFHIP_ATTN_ENC64"); on = (e && e[0] == '0') ? 0 : 1; }   // A/B switch
    if (!on) return false;
    if (a->dtype != AFHIP_BF16 || a->hd != 64 || !a->q_prescaled || a->causal || a->key_split > 0 || a->n_q != a->n_kv || a->Tq != a->Tk) return false;
    if (a->new_k || a->seq_pos) return false;
    if ((long long)a->Tk * a->ld_kv * 2 >= (1ll << 31)) return false;      // 32-bit DMA offsets inside one (clip, head)
    EncAttnP p;
    p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.o = (char*)a->out;
    p.key_len = a->key_len; p.row_off = a->row_off;
    p.B = a->B; p.Tq = a->Tq; p.Tk = a->Tk; p.n_h = a->n_q;
    p.ld_q = a->ld_q; p.ld_kv = a->ld_kv; p.ld_o = a->ld_o;
    p.q_bs = a->q_batch_stride; p.kv_bs = a->kv_batch_stride; p.o_bs = a->o_batch_stride;
    p.q_hs = a->q_head_stride; p.kv_hs = a->kv_head_stride; p.o_hs = a->o_head_stride > 0 ? a->o_head_stride : a->hd;
    p.n_qt = cdiv(a->Tq, E_QT);
    if ((long long)p.n_qt * a->n_q * a->B >= (1ll << 31)) return false;
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done)) {
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
    }
    const dim3 grid((unsigned)(p.n_qt * a->n_q * a->B));
    const char* ev = getenv("AFHIP_ENC64_VAR");          // read per call: one process can A/B the variants
    const int var = ev ? atoi(ev) : 0;
    switch (var) {
        case 1: hipLaunchKernelGGL(attn_enc64_kernel<1>, grid, dim3(256), E_LDS, s, p); break;
        case 2: hipLaunchKernelGGL(attn_enc64_kernel<2>, grid, dim3(256), E_LDS, s, p); break;
        case 3: hipLaunchKernelGGL(attn_enc64_kernel<3>, grid, dim3(256), E_LDS, s, p); break;
        case 4: hipLaunchKernelGGL(attn_enc64_kernel<4>, grid, dim3(256), E_LDS, s, p); break;
        case 8: hipLaunchKernelGGL(attn_enc64_kernel<8>, grid, dim3(256), E_LDS, s, p); break;
        case 12: hipLaunchKernelGGL(attn_enc64_kernel<12>, grid, dim3(256), E_LDS, s, p); break;
        default: hipLaunchKernelGGL(attn_enc64_kernel<0>, grid, dim3(256), E_LDS, s, p); break;
    }
    return true;
}
