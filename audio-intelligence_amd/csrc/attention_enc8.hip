// Encoder self-attention at head_dim 64 (bf16, q prescaled to exp2 units, per-clip key length): the TWO-WAVES-PER-SIMD, PERSISTENT form.
// Same arithmetic and the same bits as attn_enc64_kernel (attention_enc.hip): S^T = K . Q^T - m_lag with the query on the MFMA lane,
// P = exp2(S') straight out of the accumulators, O^T += V^T . P^T, lagged row maximum raised only when a lane's partial row sum of a tile
// exceeds 2^16, four partial row sums in the same order, the same duplicate-key edge tile.
//
// Why a second form: in the one-wave-per-SIMD kernel the wave's own VALU stream (2 v_exp + 2 v_add + 1 v_cvt_pk per MFMA gap, 36 issue
// cycles against the MFMA's 32, + ~13 cycles per LDS fragment read) IS the critical path: s_memtime stamps put a key tile at ~2040 cycles
// against 1024 of matrix-pipe time (profiles/r03_attention_enc_stamps.txt), and re-spacing the same instructions only moved it the wrong
// way.  Here a wave owns ONE 32-query block and 256 registers, and runs its tile as two straight bursts -- softmax (VALU only), then 16
// MFMAs (S'(t+1) chains and O(t) products, nothing between them); two such waves share a SIMD and the hardware interleaves one wave's
// VALU burst with the other's MFMA burst (an MFMA holds the issue port 8 of its 32 cycles).
//   * workgroup = 8 waves = 256 queries of one (clip, head) -- the same block grid, ring, DMA images, swizzles and edge-tile trick as the
//     4-wave form; a wave DMAs 8 K rows + 8 V rows per tile (2 instructions) and its own 32 Q rows per block (4 instructions).
//   * per key tile and wave:  barrier | DMA tile t+3 | 8 K(t+1) + 16 V(t) fragment reads issued | softmax of S'(t) -> P(t), row sums |
//     rare lag raise | lgkmcnt(0) | MFMAs: S'(t+1) = K(t+1).Q^T (2 chains of 4) interleaved with O += V(t).P(t) (2 d tiles x 4 k-steps).
//     The fragment reads land under the wave's own softmax; single fragment set (a K / V fragment register is rewritten only by a read
//     issued after the MFMAs that consumed it, a whole softmax burst later).
//   * accumulator file asm-owned as in the 4-wave form: a[16:47] O^T, a[48:63] Q fragments, a[64:95] K fragments, a[96:127] V fragments.
//     The `a127` clobber makes the kernel a 128 + 128 register one; hipcc believes the file is free and parks arch-VGPR overflow in its
//     LOWEST registers, so a[0:15] are left to it and the Makefile rule rejects an object whose compiler-side code names a16 or above.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int E_QT = 256, E_KT = 64, E_HD = 64, E_NW = 8;
constexpr int E_ROWB = E_HD * 2;                 // bytes per K / V row in LDS
constexpr int E_TILEB = E_KT * E_ROWB;           // 8 KiB
constexpr int E_STAGEB = 2 * E_TILEB;            // K tile + V tile
constexpr int E_NST = 4;
constexpr int E_RING = E_NST * E_STAGEB;         // 64 KiB
constexpr int E_OROW = 144;                      // O staging row: 128 B + 16 B pad
constexpr int E_QOFF = E_RING + E_NW * 32 * E_OROW; // + 36 KiB of O staging (one private 32-row area per wave)
constexpr int E_LDS = E_QOFF + E_QT * E_ROWB;    // + 32 KiB: the block's Q rows (each wave DMAs and reads its own 32)
constexpr float E_LAG_LIMIT = 65536.f;

struct EncAttnP {
    const char* q; const char* k; const char* v; char* o;
    const int32_t* key_len;     // [B] keys per clip (NULL: Tk)
    const int32_t* row_off;     // packed batches: first row of clip b (then key_len[b] = queries = keys); NULL: [B, T] batches
    int B, Tq, Tk, n_h;
    long long ld_q, ld_kv, ld_o;        // elements
    long long q_bs, kv_bs, o_bs;
    long long q_hs, kv_hs, o_hs;
    int n_qt, n_blk;
};

typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int e_swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ uint32_t e_cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// The value x again, but behind a volatile asm: code that uses it is NOT hoisted out of the block loop.  hipcc has 128 arch VGPRs here; every
// lane constant it precomputes for the once-per-block paths (Q addresses, epilogue addresses, the 32 key positions of the edge tile) and
// cannot hold, it parks in the accumulator file -- which this kernel owns (Makefile guard).
__device__ __forceinline__ int e_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define E_FENCE() __builtin_amdgcn_sched_barrier(0)
#define E_BARRIER()                               \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// accumulator-file map (first register of each object); immediates above 64 print in hex, so always the bracket form a[..]
constexpr int A_O = 16, A_Q = 48, A_KF = 64, A_VF = 96;   // a[0:15] are hipcc's: the lowest-numbered ones are where it parks what 128 arch VGPRs cannot hold
#define E_ACC_WRITE(IDX, VAL) asm volatile("v_accvgpr_write_b32 a[%0], %1" :: "n"(IDX), "v"(VAL) : "a127")
#define E_ACC_READ(DST, IDX) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(DST) : "n"(IDX))
// S_FIRST opens a chain from the VGPR tuple C (written by VALU code shortly before: s_nop 1); S_ACC accumulates in place
#define E_MFMA_S_FIRST(S, KF, Q, C) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, a[%2:%3], a[%4:%5], %1" : "=&v"(S) : "v"(C), "n"(KF), "n"((KF) + 3), "n"(Q), "n"((Q) + 3))
#define E_MFMA_S_ACC(S, KF, Q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(S) : "n"(KF), "n"((KF) + 3), "n"(Q), "n"((Q) + 3))
#define E_MFMA_O(O, VF, PF) asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]" :: "v"(PF), "n"(O), "n"((O) + 15), "n"(VF), "n"((VF) + 3) : "memory")
// k-step KS of the S' chain of key sub-tile ST
#define E_S_STEP(S, ST, KS, C)                                                                       \
    do {                                                                                              \
        if constexpr ((KS) == 0) E_MFMA_S_FIRST(S[ST], A_KF + (ST) * 16, A_Q, C);                     \
        else E_MFMA_S_ACC(S[ST], A_KF + (ST) * 16 + (KS) * 4, A_Q + (KS) * 4);                        \
    } while (0)
// fragment i = (sub-tile i >> 2, k-step i & 3) of the K tile in ring stage STAGE: ds_read_b128 straight into a[..]
#define E_READ_K(I, STAGE) asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(kaddr[(I) & 3]), "n"(A_KF + (I) * 4), "n"(A_KF + (I) * 4 + 3), \
                                        "n"((STAGE) * E_STAGEB + ((I) >> 2) * 4096) : "memory")
// fragment i = (k-step i >> 1, d tile i & 1) of the V tile: two transposed 8-byte reads into the halves of a[..]
#define E_READ_V(I, STAGE) asm volatile("ds_read_b64_tr_b16 a[%1:%2], %0 offset:%5\n\tds_read_b64_tr_b16 a[%3:%4], %0 offset:%6" :: "v"(vaddr[(I) & 1]), \
                                        "n"(A_VF + (I) * 4), "n"(A_VF + (I) * 4 + 1), "n"(A_VF + (I) * 4 + 2), "n"(A_VF + (I) * 4 + 3),                     \
                                        "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048), "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048 + 512) : "memory")

// everything a block (query tile, head, clip) needs, wave-uniform
struct EncBlk {
    int q0, Tq, klen, nt, h;
    long long o_off;
    const char* qb;
    v4i_t krsrc, vrsrc;
};

__global__ __launch_bounds__(512, 1) void attn_enc64x8_kernel(EncAttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int lrow = lane >> 3, lslot = lane & 7;
    const int ldkv2 = (int)(p.ld_kv * 2);
    const int lds_base = (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring

    // block v -> (query tile, head, clip): same order as the 4-wave form (the query tiles of one (clip, head) run together on ONE XCD)
    const int nh = p.n_h * p.B;
    auto setup = [&](int v, EncBlk& k) -> bool {
        int xt, hb;
        if ((nh & 7) == 0) {
            const int xcd = v & 7, r = v >> 3;
            xt = r % p.n_qt;
            hb = (r / p.n_qt) * 8 + xcd;
        } else {
            xt = v % p.n_qt;
            hb = v / p.n_qt;
        }
        const int b = hb / p.n_h;
        k.h = hb % p.n_h;
        k.q0 = xt * E_QT;
        long long q_off = (long long)b * p.q_bs, kv_off = (long long)b * p.kv_bs;
        k.o_off = (long long)b * p.o_bs;
        k.Tq = p.Tq;
        k.klen = p.Tk;
        if (p.key_len) { const int kl = p.key_len[b]; k.klen = kl < k.klen ? kl : k.klen; }
        if (p.row_off) {
            const long long r = p.row_off[b];
            q_off = r * p.ld_q; kv_off = r * p.ld_kv; k.o_off = r * p.ld_o;
            k.Tq = p.key_len[b];
        }
        if (k.q0 >= k.Tq) return false;               // packed batches: query tile past this clip
        k.klen = k.klen > 0 ? k.klen : 0;
        k.nt = (k.klen + E_KT - 1) / E_KT;
        k.qb = p.q + (q_off + (long long)k.h * p.q_hs) * 2;
        const unsigned long long ka = (unsigned long long)(p.k + (kv_off + (long long)k.h * p.kv_hs) * 2);
        const unsigned long long va = (unsigned long long)(p.v + (kv_off + (long long)k.h * p.kv_hs) * 2);
        // K rows past klen re-read the last live row; V rows past klen are OUT OF RANGE of the V descriptor and arrive as zeros
        k.krsrc = v4i_t{__builtin_amdgcn_readfirstlane((int)(ka & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((ka >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
        k.vrsrc = v4i_t{__builtin_amdgcn_readfirstlane((int)(va & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((va >> 32) & 0xffffu)),
                        __builtin_amdgcn_readfirstlane((k.klen > 0 ? (k.klen - 1) * ldkv2 : 0) + E_ROWB), 0x00020000};
        return true;
    };
    auto next_valid = [&](int v, EncBlk& k) -> int {
        for (; v < p.n_blk; v += (int)gridDim.x)
            if (setup(v, k)) return v;
        return -1;
    };

    // LDS-DMA from inline asm (invisible to hipcc's waitcnt pass; M0 written and consumed inside the one statement -- see attention_enc.hip)
    auto dma16 = [&](const v4i_t& rsrc, int voff, int lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
    };
    // wave w owns rows 8 w .. 8 w + 7 of every K and V tile: one 1-KiB instruction each
    auto dma_tile = [&](const EncBlk& k, int t) {
        const int ks = __builtin_amdgcn_readfirstlane(lds_base + (t & (E_NST - 1)) * E_STAGEB + wave * 1024);
        const int row = wave * 8 + lrow;
        const int key = t * E_KT + row;
        const int keyc = key < k.klen ? key : k.klen - 1;
        dma16(k.krsrc, keyc * ldkv2 + (lslot ^ ((row >> 1) & 7)) * 16, ks);                    // rswz<128>
        dma16(k.vrsrc, key * ldkv2 + (lslot ^ (((row >> 1) & 1) << 2)) * 16, ks + E_TILEB);    // vtrswz<128>
    };
    // the wave's own 32 Q rows of a block (K's row swizzle), 4 instructions
    auto dma_q = [&](const EncBlk& k) {
        const unsigned long long qa_ = (unsigned long long)k.qb;
        const v4i_t qrsrc = {__builtin_amdgcn_readfirstlane((int)(qa_ & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((qa_ >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
        const int qs = __builtin_amdgcn_readfirstlane(lds_base + E_QOFF + wave * 32 * E_ROWB);
        const int ldq2 = (int)(p.ld_q * 2);
        const int lane_o = e_opaque(lane), lrow = lane_o >> 3, lslot = lane_o & 7;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wave * 32 + u * 8 + lrow;                  // row of the workgroup's Q tile
            int qrow = k.q0 + row;
            qrow = qrow < k.Tq ? qrow : k.Tq - 1;
            dma16(qrsrc, qrow * ldq2 + (lslot ^ ((row >> 1) & 7)) * 16, qs + u * 1024);
        }
    };
    auto prefetch_tiles = [&](const EncBlk& k) {
        if (k.nt > 0) dma_tile(k, 0);
        if (k.nt > 1) dma_tile(k, 1);
        if (k.nt > 2) dma_tile(k, 2);
    };

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

    EncBlk cur, nxt;
    int v = next_valid((int)blockIdx.x, cur);
    if (v < 0) return;                                           // workgroup-uniform, before any barrier
    dma_q(cur);
    prefetch_tiles(cur);

    for (;;) {
        const int nt = cur.nt, klen = cur.klen;
        {   // block start: O = 0
            const uint32_t zero = 0u;
            static_for<0, 32>([&](auto it) { const uint32_t z = zero; E_ACC_WRITE(A_O + decltype(it)::value, z); });
        }
        E_FENCE();

        f32x16 s[2];                            // S'^T [key sub-tile]                            (VGPR)
        f32x16 c;                               // C input of the chains: -m_lag                  (VGPR)
        u32x4 pk[4];                            // P as packed bf16: the B operand of PV k-step s (VGPR)
        float ml = 0.f, l = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = 0.f;

        // two scores: P = exp2(S'), row-sum partial (FOUR independent partial sums, same order as the 4-wave form), packed bf16.
        // pk[s][j] holds registers 8 (s & 1) + 2 j, + 1 of sub-tile s >> 1, i.e. pk[s] is the 8 keys of PV k-step s.
        auto soft2 = [&](const int i, float (&psum)[4]) __attribute__((always_inline)) {
            const int ks = i >> 3, e = 2 * (i & 7);
            const float p0 = __builtin_amdgcn_exp2f(s[ks][e]), p1 = __builtin_amdgcn_exp2f(s[ks][e + 1]);
            psum[(2 * i) & 3] += p0;
            psum[(2 * i + 1) & 3] += p1;
            uint32_t w = e_cvt_pk(p0, p1);
            asm volatile("" : "+v"(w));        // pinned here: the rare path overwrites P, and hipcc would sink the v_cvt_pk below its branch
            pk[ks * 2 + ((i & 7) >> 2)][i & 3] = w;
        };
        // the rare path: raise the lag to the true maximum of this tile and redo its P.  Only the queries whose OWN partial sum tripped the
        // limit (either half of the lane pair) move: a query's bits must not depend on which other queries share its wave.
        auto rebase = [&](float& d_out, float ps_own) -> float {
            const int own = !(ps_own <= E_LAG_LIMIT) ? 1 : 0;
            const bool trig = (own | __shfl_xor(own, 32, 64)) != 0;
            float mx = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[ks][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float d = (trig && mx > 0.f) ? mx : 0.f;
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l *= alpha;
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");        // the last O MFMA -> v_accvgpr_read (18 wait states)
            static_for<0, 32>([&](auto it) {
                constexpr int i = decltype(it)::value;
                float x;
                E_ACC_READ(x, A_O + i);
                x *= alpha;
                E_ACC_WRITE(A_O + i, x);
            });
            asm volatile("s_nop 1" ::: "memory");
            ml += d;
            d_out = d;
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = -ml;
            float ps4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ks = i >> 3, e = 2 * (i & 7);
                const float p0 = __builtin_amdgcn_exp2f(s[ks][e] - d), p1 = __builtin_amdgcn_exp2f(s[ks][e + 1] - d);
                ps4[(2 * i) & 3] += p0;
                ps4[(2 * i + 1) & 3] += p1;
                pk[ks * 2 + ((i & 7) >> 2)][i & 3] = e_cvt_pk(p0, p1);
            }
            return (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
        };
        // last, partly filled tile: the keys past klen are copies of key klen - 1 with zero V rows; their equal terms leave the row sum here
        auto dup_sum = [&](float d) -> float {
            const int kk = (klen - 1) - (nt - 1) * E_KT;       // position of the last live key inside the last tile
            const int lane_o = e_opaque(lane);
            float vv = 0.f;
            int mine = 0, cnt = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = ks * 32 + e_swap23(mfma32_row(e, lane_o));
                    vv = key == kk ? s[ks][e] : vv;
                    mine |= key == kk ? 1 : 0;
                    cnt += key > kk ? 1 : 0;
                }
            const float vo = __shfl_xor(vv, 32, 64);
            const float sl = mine ? vv : vo;                   // the last live key sits in exactly one half of the pair
            return (float)cnt * __builtin_amdgcn_exp2f(sl - d);
        };
        const bool has_edge = (klen % E_KT) != 0;

        if (nt > 0) {
            // Q and tile 0 have landed?  FIFO of this wave: [Q rows (4)][tiles 0..2 (2 each)][0..4 O stores of the previous block].  Only tiles
            // 1 and 2 are CERTAINLY younger than tile 0 (the stores are predicated per row)
            if (nt >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            E_BARRIER();                                               // ... and everybody's
            {
                const int lane_o = e_opaque(lane), fr_o = lane_o & 31, fh_o = lane_o >> 5;
                static_for<0, 4>([&](auto it) {
                    constexpr int i = decltype(it)::value;             // k-step
                    const int qaddr = lds_base + E_QOFF + (wave * 32 + fr_o) * E_ROWB + (((i * 2 + fh_o) ^ ((fr_o >> 1) & 7)) << 4);
                    asm volatile("ds_read_b128 a[%1:%2], %0" :: "v"(qaddr), "n"(A_Q + i * 4), "n"(A_Q + i * 4 + 3) : "memory");
                });
            }
            static_for<0, 8>([&](auto it) { constexpr int i = decltype(it)::value; (void)&kaddr; E_READ_K(i, 0); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            E_FENCE();
            // the true row maxima of tile 0 become the lag; then S'(0)
            static_for<0, 8>([&](auto it) { constexpr int i = decltype(it)::value; (void)&s; (void)&c; E_S_STEP(s, i & 1, i >> 1, c); });
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");        // MFMA results -> VALU readers (18 wait states for a 16-pass MFMA)
            E_FENCE();
            float m0 = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) m0 = fmaxf(m0, s[ks][e]);
            m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
            ml = m0;                                               // finite: every key row of tile 0 is a live row (copies included)
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = -ml;
            E_FENCE();
            static_for<0, 8>([&](auto it) { constexpr int i = decltype(it)::value; (void)&s; (void)&c; E_S_STEP(s, i & 1, i >> 1, c); });
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            E_FENCE();

            // softmax of S'(t) -> P(t), row sum; the rare lag raise; the duplicate keys of a partly filled last tile
            auto softmax_tile = [&](const int t) __attribute__((always_inline)) {
                float ps4[4] = {0.f, 0.f, 0.f, 0.f}, dsh = 0.f;
                static_for<0, 16>([&](auto it) { constexpr int i = decltype(it)::value; soft2(i, ps4); });
                E_FENCE();
                float ps = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
                if (__any(!(ps <= E_LAG_LIMIT))) ps = rebase(dsh, ps);
                if (has_edge && t == nt - 1) { asm volatile("" ::: "memory"); ps -= dup_sum(dsh); }
                l += ps;
                E_FENCE();
            };
            // 16 MFMAs: k-step by k-step, the two S'(t+1) chains and the two O d tiles (a dependent pair is 4 MFMAs apart)
            auto mfma_batch = [&]() __attribute__((always_inline)) {
                static_for<0, 4>([&](auto it) {
                    constexpr int ks = decltype(it)::value;
                    (void)&s; (void)&c; (void)&pk;
                    E_S_STEP(s, 0, ks, c);
                    E_S_STEP(s, 1, ks, c);
                    E_MFMA_O(A_O, A_VF + (ks * 2) * 4, pk[ks]);
                    E_MFMA_O(A_O + 16, A_VF + (ks * 2 + 1) * 4, pk[ks]);
                });
                E_FENCE();
            };
            // One key tile; unrolled four times so that the ring stage (t & 3) is a compile-time constant.  The two waves of a SIMD (w and
            // w + 4: a workgroup's waves go round the four SIMDs) meet at the SAME barrier every tile, so they are put half a tile apart by
            // construction: waves 0-3 run softmax(t) then the MFMAs of tile t, waves 4-7 (ROT) the MFMAs of tile t then softmax(t+1) -- one
            // wave's VALU burst beside the other's matrix burst.  Left alike, both bursts coincide (measured: 461 us against the 4-wave
            // form's 430).  Per query the sequence of operations is the same in both rotations, hence the same bits.
            const bool rot = wave >= 4;
            auto iter = [&](const int t, auto idx_tag) __attribute__((always_inline)) {
                constexpr int IDX = decltype(idx_tag)::value;
                // tile t+1 must have landed before its K fragments are read below; tile t+2 (2 DMA instructions) may stay in flight
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                E_BARRIER();
                if (t + 3 < nt) dma_tile(cur, t + 3);          // stage (t+3)&3 = (t-1)&3: last read (V(t-1)) before the barrier above
                E_FENCE();
                // fragment reads of this tile's MFMAs: K(t+1) (past the last tile: whatever the stage holds -- that S' is never used), V(t)
                static_for<0, 8>([&](auto it) {
                    constexpr int i = decltype(it)::value;
                    (void)&kaddr; (void)&vaddr;
                    E_READ_K(i, (IDX + 1) & (E_NST - 1));
                    E_READ_V(i, IDX);
                });
                E_FENCE();
                // softmax: the unrotated waves' tile t now; the rotated waves did it after the previous tile's MFMAs and do tile t+1 below
                const int ts = rot ? t + 1 : t;
                if (rot) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the fragments are in their registers
                    E_FENCE();
                    mfma_batch();
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // S'(t+1) is readable (its last MFMA is the 14th of the 16)
                    E_FENCE();
                }
                if (ts < nt) softmax_tile(ts);
                if (!rot) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    E_FENCE();
                    mfma_batch();
                }
            };
            if (rot) softmax_tile(0);
            for (int t = 0; t < nt; t += 4) {
                iter(t, std::integral_constant<int, 0>{});
                if (t + 1 < nt) iter(t + 1, std::integral_constant<int, 1>{});
                if (t + 2 < nt) iter(t + 2, std::integral_constant<int, 2>{});
                if (t + 3 < nt) iter(t + 3, std::integral_constant<int, 3>{});
            }
            E_BARRIER();                                              // every wave is done reading the ring: the next block's tiles may land
        }

        // the next block's Q rows and first tiles are requested BEFORE this block's epilogue (their latency hides behind it)
        const long long o_off = cur.o_off;
        const int q0 = cur.q0, Tq = cur.Tq, hh = cur.h;
        v = next_valid(v + (int)gridDim.x, nxt);
        if (v >= 0) {
            dma_q(nxt);
            prefetch_tiles(nxt);
        }
        E_FENCE();

        // epilogue: normalise, stage O through this wave's private LDS area, store whole rows.  O^T register e of d-tile dt is
        // d = dt*32 + (e&3) + 8*(e>>2) + 4*fh of query (lane & 31): the lane writes 4 consecutive d (8 bytes) at [query][d]
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");            // the last MFMA results are readable
        const float lt = l + __shfl_xor(l, 32, 64);
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;
        char* ost = smem + E_RING + wave * (32 * E_OROW);
        const int lane_e = e_opaque(lane), fr_e = lane_e & 31, fh_e = lane_e >> 5;
        static_for<0, 8>([&](auto it) {
            constexpr int i = decltype(it)::value;                   // (d tile, group of 4 registers)
            constexpr int dt = i >> 2, g = i & 3;
            float o0, o1, o2, o3;
            E_ACC_READ(o0, A_O + dt * 16 + 4 * g);
            E_ACC_READ(o1, A_O + dt * 16 + 4 * g + 1);
            E_ACC_READ(o2, A_O + dt * 16 + 4 * g + 2);
            E_ACC_READ(o3, A_O + dt * 16 + 4 * g + 3);
            const uint32_t w0 = e_cvt_pk(o0 * inv, o1 * inv), w1 = e_cvt_pk(o2 * inv, o3 * inv);
            const int d = dt * 32 + 8 * g + 4 * fh_e;
            *reinterpret_cast<uint2*>(ost + fr_e * E_OROW + d * 2) = uint2{w0, w1};
        });
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's own stores (its rows are private to it)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + (lane_e >> 3), cc = lane_e & 7;
            const int qrow = q0 + wave * 32 + r;
            const u32x4 val = *reinterpret_cast<const u32x4*>(ost + r * E_OROW + cc * 16);
            if (qrow < Tq) {
                __bf16* op = reinterpret_cast<__bf16*>(p.o) + o_off + (long long)qrow * p.ld_o + (long long)hh * p.o_hs;
                *reinterpret_cast<u32x4*>(op + cc * 8) = val;
            }
        }
        if (v < 0) break;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// Called by afhip_attention (attention.hip) before the 4-wave form; returns false when it does not apply or is switched off.
bool afhip_attention_enc64x8(const afhip_attn_args* a, hipStream_t s) {
    // OFF by default: bit-identical to the 4-wave form and 8 % slower (round 3, same box: 460 vs 425 us per B = 32 call, with and without
    // the half-tile rotation) -- the SIMD's issue port, not the overlap of VALU and matrix bursts, is what both forms run into.  AFHIP_ATTN_ENC8=1 selects it.
    { const char* e = getenv("AFHIP_ATTN_ENC8"); if (!(e && e[0] == '1')) return false; }   // read per call
    if (a->dtype != AFHIP_BF16 || a->hd != 64 || !a->q_prescaled || a->causal || a->key_split > 0 || a->n_q != a->n_kv || a->Tq != a->Tk) return false;
    if (a->new_k || a->seq_pos || a->out_fp8) return false;
    if ((long long)a->Tk * a->ld_kv * 2 >= (1ll << 31)) return false;      // 32-bit DMA offsets inside one (clip, head)
    EncAttnP p;
    p.q = (const char*)a->q; p.k = (const char*)a->k; p.v = (const char*)a->v; p.o = (char*)a->out;
    p.key_len = a->key_len; p.row_off = a->row_off;
    p.B = a->B; p.Tq = a->Tq; p.Tk = a->Tk; p.n_h = a->n_q;
    p.ld_q = a->ld_q; p.ld_kv = a->ld_kv; p.ld_o = a->ld_o;
    p.q_bs = a->q_batch_stride; p.kv_bs = a->kv_batch_stride; p.o_bs = a->o_batch_stride;
    p.q_hs = a->q_head_stride; p.kv_hs = a->kv_head_stride; p.o_hs = a->o_head_stride > 0 ? a->o_head_stride : a->hd;
    p.n_qt = cdiv(a->Tq, E_QT);
    const long long nblk = (long long)p.n_qt * a->n_q * a->B;
    if (nblk >= (1ll << 31)) return false;
    p.n_blk = (int)nblk;
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done))
        (void)hipFuncSetAttribute((const void*)attn_enc64x8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
    int ncu = afhip_cu_count();
    ncu = ncu >= 8 ? (ncu / 8) * 8 : ncu;
    const int grid = nblk < ncu ? (int)nblk : ncu;
    hipLaunchKernelGGL(attn_enc64x8_kernel, dim3((unsigned)grid), dim3(512), E_LDS, s, p);
    return true;
}
