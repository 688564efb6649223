// Encoder self-attention at head_dim 64 (bf16, q prescaled to exp2 units, per-clip key length): the ONE-WAVE-PER-SIMD, PERSISTENT form.
// Same arithmetic as attn_kernel<bf16, 64, 2, LAG> (attention.hip): S^T = K . Q^T - m_lag with the query on the MFMA lane, P = exp2(S')
// straight out of the accumulators, O^T += V^T . P^T, lagged row maximum (raised only when a tile's partial row sum exceeds 2^16).
//
// Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"; MI355X_MICROARCH.md issue prices):
//   * workgroup = 4 waves = 256 queries of one (clip, head); a wave owns 64 queries as TWO 32-query blocks a, b and the whole
//     512-register file.  A K or V fragment read from LDS serves both blocks: 16 KB of LDS reads per 32 MFMAs instead of per 16.
//   * one workgroup per CU walks the (query tile, head, clip) blocks; the NEXT block's Q rows and first three K / V tiles are requested
//     before the current block's epilogue runs (measured: with one workgroup per block, 5 us of launch ramp, Q / first-tile latency and
//     epilogue per block ran with nothing beside them -- a fifth of the kernel).
//   * K / V tiles of 64 keys arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, issued from inline asm) into a ring of
//     4 stages, three tiles ahead of the arithmetic; ONE barrier per key tile, counted vmcnt (4 DMA instructions per wave per tile).
//   * per tile the wave runs two SLOTS of 16 MFMAs, each beside the softmax (32 x v_exp_f32, 32 x v_add_f32, 16 x v_cvt_pk) of ONE
//     query block, software-pipelined across tiles so that every softmax has 16 independent MFMAs to hide behind:
//         slot 1:  S_b(t) = K(t).Q_b^T     O_b += V(t-1).P_b(t-1)     ||  softmax of S_a(t)   -> P_a(t)   ||  fragment reads K(t+1), V(t)
//         slot 2:  S_a(t+1) = K(t+1).Q_a^T  O_a += V(t).P_a(t)        ||  softmax of S_b(t)   -> P_b(t)
//     Every MFMA gap is pinned in the source by a scheduling fence: [MFMA] [2 v_exp, 2 v_add, 1 v_cvt_pk] [<= 3 LDS reads]  (issue
//     budget 8 + 16 + 12 = 36 cycles of the MFMA's 32: at head_dim 64 the VALU, not the matrix pipe, is the floor -- 0.89 of the roof).
//   * the accumulator file is ASM-OWNED: O, the Q / K / V fragments live at literal a[..] registers named in the asm text (hipcc's own
//     allocation put S' in AGPRs -- 64 v_accvgpr_read per slot -- and bounced fragments between the two files); S', the chains' C input
//     and the packed P are ordinary compiler-managed arch VGPRs, because the VALU reads and writes them.
//   * the rare "raise the lag" path sits in its own basic block at the END of a slot, so the slots stay straight-line code.
//   * keys past the clip's length in the last tile are copies of the last live key (clamped K rows) whose V rows arrive as zeros (out
//     of range of the V buffer descriptor): they add nothing to O and their equal terms are taken out of the row sums.
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
constexpr int E_RING = E_NST * E_STAGEB;         // 64 KiB
constexpr int E_OROW = 144;                      // O staging row: 128 B + 16 B pad
constexpr int E_QOFF = E_RING + 4 * 64 * E_OROW; // + 36 KiB of O staging (one private 64-row area per wave)
constexpr int E_LDS = E_QOFF + E_QT * E_ROWB;    // + 32 KiB: the block's Q rows (each wave DMAs and reads its own 64)
#ifdef AFHIP_ENC_STAMPS   /* diagnostic build: -DAFHIP_ENC_STAMPS, tools/attn_enc_stamps.py (never the product library) */
constexpr int E_LDS_TOTAL = E_LDS;
// A stamp is one s_memtime and one SCALAR store straight to the debug buffer (gfx9 still has s_store): no VGPR, no LDS, no vmcnt traffic,
// no state of its own -- the slot comes from (wave, t, K), the condition from values the loop keeps live anyway.  The kernel has no VGPR
// to spare: two more live values and hipcc parks one in the (asm-owned) accumulator file, which is a memory fault (Makefile guard).
#define E_STAMP(K) do { if (p.dbg && blockIdx.x == 0 && v == (int)(blockIdx.x + 2 * gridDim.x) && t >= 8 && t < 16 && (wave == 0 || wave == 3)) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                                               \
        const int off_ = ((wave ? 64 : 0) + (t - 8) * 5 + (K)) * 8;                                                                                \
        asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(t_), "s"(p.dbg), "s"(off_) : "memory"); } } while (0)
#else
constexpr int E_LDS_TOTAL = E_LDS;
#define E_STAMP(K) do { } while (0)
#endif
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
    int out_fp8;                // o is e4m3 bytes = sat(value * out_inv) (afhip_attn_args.out_fp8)
    float out_inv;
#ifdef AFHIP_ENC_STAMPS
    unsigned long long* dbg;    // AFHIP_ENC_DBGPTR: [2][64] stamps
#endif
};

typedef int v4i_t __attribute__((ext_vector_type(4)));

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

#define E_FENCE() __builtin_amdgcn_sched_barrier(0)
#define E_BARRIER()                               \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// accumulator-file map (first register of each object)
constexpr int A_OA = 0, A_OB = 32, A_QA = 64, A_QB = 80, A_KF = 96, A_VF = 160;
//   a[0:31] O_a^T (d tiles 0, 1)   a[32:63] O_b^T   a[64:79] Q_a fragments (4 x 4)   a[80:95] Q_b
//   a[96:127] K fragment set 0 (8 x 4)   a[128:159] set 1   a[160:191] V fragment set 0   a[192:223] set 1
// (immediates above 64 print in hex: `a[0x50]` assembles, `a0x50` does not -- always the bracket form)
#define E_ACC_WRITE(IDX, VAL) asm volatile("v_accvgpr_write_b32 a[%0], %1" :: "n"(IDX), "v"(VAL) : "a255")
#define E_ACC_READ(DST, IDX) asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(DST) : "n"(IDX))
// MFMA forms (KF / Q / VF / O = first register of the operand's tuple in the accumulator file).  S_FIRST opens a chain from the VGPR tuple
// C (written by VALU code shortly before: s_nop 1); the others accumulate in place.
#define E_REBASE_COND(PS) (__any(!((PS) <= E_LAG_LIMIT)))
#define E_MFMA_S_FIRST(S, KF, Q, C) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, a[%2:%3], a[%4:%5], %1" : "=&v"(S) : "v"(C), "n"(KF), "n"((KF) + 3), "n"(Q), "n"((Q) + 3))
#define E_MFMA_S_ACC(S, KF, Q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], a[%3:%4], %0" : "+v"(S) : "n"(KF), "n"((KF) + 3), "n"(Q), "n"((Q) + 3))
#define E_MFMA_O(O, VF, PF) asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], a[%3:%4], %0, a[%1:%2]" :: "v"(PF), "n"(O), "n"((O) + 15), "n"(VF), "n"((VF) + 3) : "memory")
// S' chains of one block from K set KSET (Q = first register of the block's Q fragments)
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
// one MFMA of a slot: MFMAs 0..7 are the S' chains (sub-tile 0: 0..3, sub-tile 1: 4..7) of the block whose Q sits at QB, from K set KSET;
// 8..15 are O (at OB) += V (set VSET) . P
#define E_SLOT_MFMA(I, S, KSET, QB, C, OB, VSET, PK)                                                           \
    do {                                                                                                        \
        if constexpr ((I) < 8) {                                                                                \
            if constexpr (((I) & 3) == 0) E_MFMA_S_FIRST(S[(I) >> 2], A_KF + (KSET) * 32 + (I) * 4, QB, C);     \
            else E_MFMA_S_ACC(S[(I) >> 2], A_KF + (KSET) * 32 + (I) * 4, (QB) + ((I) & 3) * 4);                \
        } else {                                                                                                \
            E_MFMA_O((OB) + (((I) - 8) & 1) * 16, A_VF + (VSET) * 32 + ((I) - 8) * 4, PK[((I) - 8) >> 1]);     \
        }                                                                                                       \
    } while (0)
// fragment i of K set SET from ring stage STAGE: ds_read_b128 straight into a[..] (kaddr = lane part, the rest in the 16-bit offset)
#define E_READ_K(SET, I, STAGE) asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(kaddr[(I) & 3]), "n"(A_KF + (SET) * 32 + (I) * 4), "n"(A_KF + (SET) * 32 + (I) * 4 + 3), \
                                             "n"((STAGE) * E_STAGEB + ((I) >> 2) * 4096) : "memory")
// fragment i = (k-step i >> 1, d tile i & 1) of V set SET: two transposed 8-byte reads into the halves of a[..]
#define E_READ_V(SET, I, STAGE) asm volatile("ds_read_b64_tr_b16 a[%1:%2], %0 offset:%5\n\tds_read_b64_tr_b16 a[%3:%4], %0 offset:%6" :: "v"(vaddr[(I) & 1]), \
                                             "n"(A_VF + (SET) * 32 + (I) * 4), "n"(A_VF + (SET) * 32 + (I) * 4 + 1), "n"(A_VF + (SET) * 32 + (I) * 4 + 2), "n"(A_VF + (SET) * 32 + (I) * 4 + 3), \
                                             "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048), "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048 + 512) : "memory")

// one half (H = 0, 1) of V fragment i
#define E_READ_VH(SET, I, H, STAGE) asm volatile("ds_read_b64_tr_b16 a[%1:%2], %0 offset:%3" :: "v"(vaddr[(I) & 1]), \
                                             "n"(A_VF + (SET) * 32 + (I) * 4 + 2 * (H)), "n"(A_VF + (SET) * 32 + (I) * 4 + 2 * (H) + 1), \
                                             "n"((STAGE) * E_STAGEB + ((I) >> 1) * 2048 + 512 * (H)) : "memory")
// a tile's 24 fragment reads sit in MFMA gaps 0-7 of slot 1, three each (K, V lo, V hi); two other placements (one V half in every gap; one read per
// gap over both slots) measured +1 ... +5 % and are gone (profiles/r03_attention_experiments.txt)

// everything a block (query tile, head, clip) needs, wave-uniform
struct EncBlk {
    int q0, Tq, klen, nt, h;
    long long o_off;
    const char* qb;
    v4i_t krsrc, vrsrc;
};

__global__ __launch_bounds__(256, 1) void attn_enc64_kernel(EncAttnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int lrow = lane >> 3, lslot = lane & 7;
    const int ldkv2 = (int)(p.ld_kv * 2);
    const int lds_base = (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the ring

    // block v -> (query tile, head, clip): blocks v and v + 8 share an XCD (round-robin dispatch; the grid is a multiple of 8, so a
    // persistent workgroup keeps its residue), and with v = xcd + 8 (xtile + n_qt * head_group), head = 8 head_group + xcd, the query tiles
    // of one (clip, head) run at the same time on ONE XCD: its K / V is fetched from HBM once.  Head-major order when heads * clips % 8 != 0.
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
        // K rows past klen re-read the last live row (their scores equal the last live key's: finite, and removed from the row sums);
        // V rows past klen are OUT OF RANGE of the V descriptor and arrive as zeros, so those keys add nothing to O
        k.krsrc = v4i_t{__builtin_amdgcn_readfirstlane((int)(ka & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((ka >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
        k.vrsrc = v4i_t{__builtin_amdgcn_readfirstlane((int)(va & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((va >> 32) & 0xffffu)),
                        __builtin_amdgcn_readfirstlane((k.klen > 0 ? (k.klen - 1) * ldkv2 : 0) + E_ROWB), 0x00020000};
        return true;
    };
    // next block of this workgroup that has work, or -1
    auto next_valid = [&](int v, EncBlk& k) -> int {
        for (; v < p.n_blk; v += (int)gridDim.x)
            if (setup(v, k)) return v;
        return -1;
    };

    // ---- LDS-DMA (buffer_load_dwordx4 ... lds) as INLINE ASM: hipcc's waitcnt pass drains every LDS-DMA it knows about (vmcnt(0)) in
    //      front of LDS reads; issued from asm the stream is invisible to it and stays in flight behind the counted vmcnt of the tile loop
    //      (cdna guide 5.7).  A wave-instruction fills 8 LDS rows of 128 B; lane (lrow, slot) writes slot `slot` of row r0 + lrow with
    //      SOURCE chunk slot ^ swz(row) (linear image, swizzle on the source address).  Wave w owns rows 16 w .. 16 w + 15 of every tile. ----
    auto dma16 = [&](const v4i_t& rsrc, int voff, int lds_addr) {
        // M0 is written and consumed inside the one statement and not restored: nothing hipcc emits in this kernel reads M0 (gfx950 LDS
        // instructions do not need it; there is no compiler-issued LDS-DMA, s_sendmsg or v_movrel here -- checked in the .s)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
    };
    auto dma_tile = [&](const EncBlk& k, int t) {
        const int ks = __builtin_amdgcn_readfirstlane(lds_base + (t & (E_NST - 1)) * E_STAGEB + wave * 2048);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = wave * 16 + u * 8 + lrow;
            const int key = t * E_KT + row;
            const int keyc = key < k.klen ? key : k.klen - 1;
            dma16(k.krsrc, keyc * ldkv2 + (lslot ^ ((row >> 1) & 7)) * 16, ks + u * 1024);                    // rswz<128>
            dma16(k.vrsrc, key * ldkv2 + (lslot ^ (((row >> 1) & 1) << 2)) * 16, ks + E_TILEB + u * 1024);    // vtrswz<128>
        }
    };
    // Q rows of a block: by LDS-DMA as well (8 instructions per wave, its own 64 rows; K's row swizzle).  A register-destination load
    // would be counted by hipcc, which -- blind to the asm DMA stream -- waits vmcnt(0) for it at the next block's start: every tile in
    // flight AND the previous block's O stores (measured: the persistent form was 6 % SLOWER than one workgroup per block that way).
    auto dma_q = [&](const EncBlk& k) {
        const unsigned long long qa_ = (unsigned long long)k.qb;
        const v4i_t qrsrc = {__builtin_amdgcn_readfirstlane((int)(qa_ & 0xffffffffu)), __builtin_amdgcn_readfirstlane((int)((qa_ >> 32) & 0xffffu)), 0x7fffffff, 0x00020000};
        const int qs = __builtin_amdgcn_readfirstlane(lds_base + E_QOFF + wave * 64 * E_ROWB);
        const int ldq2 = (int)(p.ld_q * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = wave * 64 + u * 8 + lrow;                  // row of the workgroup's Q tile
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
    int qaddr[4];
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) qaddr[dc] = lds_base + E_QOFF + (wave * 64 + fr) * E_ROWB + (((dc * 2 + fh) ^ ((fr >> 1) & 7)) << 4);   // block b: + 32 rows
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
#ifdef AFHIP_ENC_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();   // this workgroup's own duration (the counters of different XCDs are not aligned)
#endif
    dma_q(cur);
    prefetch_tiles(cur);
    {   // V fragment sets = 0 once: the first slot of every block multiplies "V(-1)" (whatever the sets hold: finite) by P_b(-1) = 0
        const uint32_t zero = 0u;
        static_for<0, 64>([&](auto it) { const uint32_t z = zero; E_ACC_WRITE(A_VF + decltype(it)::value, z); });
    }

    for (;;) {
        const int nt = cur.nt, klen = cur.klen;
        // ---- block start: O = 0 ----
        {
            const uint32_t zero = 0u;
            static_for<0, 64>([&](auto it) { const uint32_t z = zero; E_ACC_WRITE(A_OA + decltype(it)::value, z); });
        }
        E_FENCE();

        f32x16 sa[2], sb[2];                    // S'^T [key sub-tile]                            (VGPR)
        f32x16 ca, cb;                          // C input of each chain: -m_lag                  (VGPR)
        u32x4 pka[4], pkb[4];                   // P as packed bf16: the B operand of PV k-step s (VGPR)
        float ml_a = 0.f, ml_b = 0.f, l_a = 0.f, l_b = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { ca[e] = 0.f; cb[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { pka[i] = u32x4{0, 0, 0, 0}; pkb[i] = u32x4{0, 0, 0, 0}; }

        // the two scores of MFMA gap i of a slot: P = exp2(S'), row-sum partial, packed bf16.  pk[s][j] holds registers 8 (s & 1) + 2 j, + 1
        // of sub-tile s >> 1, i.e. pk[s] is the 8 keys of PV k-step s.
        // (the row sum runs on FOUR independent partial sums: one chain of 32 dependent v_add_f32 per slot stalled the wave's in-order issue --
        //  and with it the next MFMA -- behind every add's latency: 40 us of a 430-us launch)
        auto soft2 = [&](const int i, const f32x16 (&s)[2], u32x4 (&pk)[4], float (&psum)[4]) __attribute__((always_inline)) {
            const int ks = i >> 3, e = 2 * (i & 7);
            const float p0 = __builtin_amdgcn_exp2f(s[ks][e]), p1 = __builtin_amdgcn_exp2f(s[ks][e + 1]);
            psum[(2 * i) & 3] += p0;
            psum[(2 * i + 1) & 3] += p1;
            uint32_t w = e_cvt_pk(p0, p1);
            asm volatile("" : "+v"(w));        // pinned here: the rare path overwrites P, and hipcc would sink the slot's 16 v_cvt_pk below its branch
            pk[ks * 2 + ((i & 7) >> 2)][i & 3] = w;
        };
        // the rare path: raise the lag of one query block (O at a[OBASE .. OBASE + 31]) to the true maximum of this tile and redo its P
        // Only the queries whose OWN partial sum tripped the limit (either half of the lane pair) move their lag: a query's bits must not
        // depend on which other queries share its wave (packed and padded batches put different neighbours there).
        auto rebase = [&](auto obase_tag, const f32x16 (&s)[2], u32x4 (&pk)[4], f32x16& c, float& ml, float& l, float& d_out, float ps_own) -> float {
            constexpr int OBASE = decltype(obase_tag)::value;
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
            // same four partial sums, same order as the fast path: a lane whose own maximum did not move (d = 0) must get the bits it
            // would have got without its wave-mates' rebase (packed and padded batches put different queries side by side in a wave)
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
        // Last, partly filled tile: the keys past klen are copies of key klen - 1 (same K row -> bit-identical S'), with zero V rows.  They
        // add nothing to O; their equal terms exp2(S'_last - d) are taken out of this lane's row-sum partial here (each half of a lane pair
        // sums 32 of the tile's 64 keys: count the copies this half holds).  Once per query block per block, in a basic block of its own.
        auto dup_sum = [&](const f32x16 (&s)[2], float d) -> float {
            const int kk = (klen - 1) - (nt - 1) * E_KT;       // position of the last live key inside the last tile
            float vv = 0.f;
            int mine = 0, cnt = 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = ks * 32 + e_swap23(mfma32_row(e, lane));
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
            // ---- Q and tile 0 have landed?  FIFO of this wave: [Q rows (8)][tiles 0..2 of this block][0..8 O stores of the previous block].  The
            //      count may only assume ops that are CERTAINLY younger than tile 0: tiles 1 and 2 (the stores are predicated per row and
            //      may not exist; when they do, this merely waits for some of them as well) ----
            if (nt >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            E_BARRIER();                                               // ... and everybody's
            // Q fragments (B operand of S'^T): row = this lane's query, 16-byte chunk 2 dc + fh -> a[64:95]
            static_for<0, 8>([&](auto it) {
                constexpr int i = decltype(it)::value;                 // (block i >> 2, k-step i & 3)
                (void)&qaddr;
                asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(qaddr[i & 3]), "n"(A_QA + i * 4), "n"(A_QA + i * 4 + 3), "n"((i >> 2) * 32 * E_ROWB) : "memory");
            });
            static_for<0, 8>([&](auto it) { constexpr int i = decltype(it)::value; (void)&kaddr; E_READ_K(0, i, 0); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            E_FENCE();
            // the true row maxima of tile 0 become the lags; then S'_a(0)
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

            // one key tile; the loop is unrolled four times so that the ring stage (t & 3) is a compile-time constant and PAR = t & 1 names
            // the fragment sets: set PAR holds K(t) / V(t-1), set PAR ^ 1 receives K(t+1) / V(t)
            auto iter = [&](const int t, auto idx_tag) __attribute__((always_inline)) {
                constexpr int IDX = decltype(idx_tag)::value;
                constexpr int PAR = IDX & 1, NXT = PAR ^ 1;
                // tile t+1 must have landed before its K fragments are read in slot 1; tile t+2 (4 DMA instructions) may stay in flight
                E_STAMP(0);                                    // 0: tile starts
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                E_STAMP(1);                                    // 1: tile t+1 landed (this wave's part)
                E_BARRIER();
                E_STAMP(2);                                    // 2: barrier passed
                if (t + 3 < nt) dma_tile(cur, t + 3);          // stage (t+3)&3 = (t-1)&3: last read (V(t-1)) before the barrier above
                E_FENCE();
                // ---- slot 1 ----
                float ps4[4] = {0.f, 0.f, 0.f, 0.f}, dsh = 0.f;
                static_for<0, 16>([&](auto it) {
                    constexpr int i = decltype(it)::value;
                    (void)&sb; (void)&cb; (void)&pkb; (void)&kaddr; (void)&vaddr;
                    E_SLOT_MFMA(i, sb, PAR, A_QB, cb, A_OB, PAR, pkb);
                    soft2(i, sa, pka, ps4);
                    if constexpr (i < 8) { E_READ_K(NXT, i, (IDX + 1) & (E_NST - 1)); E_READ_V(NXT, i, IDX); }
                    E_FENCE();
                });
                float ps = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
                if (E_REBASE_COND(ps)) ps = rebase(std::integral_constant<int, A_OA>{}, sa, pka, ca, ml_a, l_a, dsh, ps);
                if (has_edge && t == nt - 1) { asm volatile("" ::: "memory"); ps -= dup_sum(sa, dsh); }
                l_a += ps;
                E_STAMP(3);                                    // 3: slot 1 issued
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the fragments read in slot 1 are in their registers
                E_FENCE();
                // ---- slot 2 ----
                ps4[0] = ps4[1] = ps4[2] = ps4[3] = 0.f; dsh = 0.f;
                static_for<0, 16>([&](auto it) {
                    constexpr int i = decltype(it)::value;
                    (void)&sa; (void)&ca; (void)&pka;
                    E_SLOT_MFMA(i, sa, NXT, A_QA, ca, A_OA, NXT, pka);
                    soft2(i, sb, pkb, ps4);
                    E_FENCE();
                });
                ps = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
                if (E_REBASE_COND(ps)) ps = rebase(std::integral_constant<int, A_OB>{}, sb, pkb, cb, ml_b, l_b, dsh, ps);
                if (has_edge && t == nt - 1) { asm volatile("" ::: "memory"); ps -= dup_sum(sb, dsh); }
                l_b += ps;
                E_STAMP(4);                                    // 4: slot 2 issued
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
            E_BARRIER();                                              // every wave is done reading the ring: the next block's tiles may land
        }

        // ---- the next block's Q rows and first tiles are requested BEFORE this block's epilogue (their latency hides behind it) ----
        const long long o_off = cur.o_off;
        const int q0 = cur.q0, Tq = cur.Tq, hh = cur.h;
        v = next_valid(v + (int)gridDim.x, nxt);
        if (v >= 0) {
            dma_q(nxt);
            prefetch_tiles(nxt);
        }
        E_FENCE();

        // ---- epilogue: normalise, stage O through this wave's private LDS area, store whole rows.  O^T register e of d-tile dt is
        //      d = dt*32 + (e&3) + 8*(e>>2) + 4*fh of query (lane & 31): the lane writes 4 consecutive d (8 bytes) at [query][d] ----
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");            // the last MFMA results are readable
        const float la = l_a + __shfl_xor(l_a, 32, 64), lb = l_b + __shfl_xor(l_b, 32, 64);
        const float ia = la > 0.f ? 1.0f / la : 0.f, ib = lb > 0.f ? 1.0f / lb : 0.f;
        char* ost = smem + E_RING + wave * (64 * E_OROW);
        if (p.out_fp8) {
            // statically quantised output (the out-projection's e4m3 A operand): 4 values -> one dword at [query][d], rows of 64 bytes
            static_for<0, 16>([&](auto it) {
                constexpr int i = decltype(it)::value;                   // (block, d tile, group of 4 registers)
                constexpr int blk = i >> 3, dt = (i >> 2) & 1, g = i & 3;
                float o0, o1, o2, o3;
                E_ACC_READ(o0, blk * 32 + dt * 16 + 4 * g);
                E_ACC_READ(o1, blk * 32 + dt * 16 + 4 * g + 1);
                E_ACC_READ(o2, blk * 32 + dt * 16 + 4 * g + 2);
                E_ACC_READ(o3, blk * 32 + dt * 16 + 4 * g + 3);
                const float sc = (blk ? ib : ia) * p.out_inv;
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(o0 * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(o1 * sc, -448.f, 448.f), w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(o2 * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(o3 * sc, -448.f, 448.f), w, true);
                const int d = dt * 32 + 8 * g + 4 * fh;
                *reinterpret_cast<int*>(ost + (blk * 32 + fr) * E_OROW + d) = w;
            });
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + (lane >> 2), c = lane & 3;
                const int qrow = q0 + wave * 64 + r;
                const u32x4 val = *reinterpret_cast<const u32x4*>(ost + r * E_OROW + c * 16);
                if (qrow < Tq) {
                    char* op = p.o + o_off + (long long)qrow * p.ld_o + (long long)hh * p.o_hs;      // one byte per element
                    *reinterpret_cast<u32x4*>(op + c * 16) = val;
                }
            }
        } else {
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
            *reinterpret_cast<uint2*>(ost + (blk * 32 + fr) * E_OROW + d * 2) = uint2{w0, w1};
        });
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's own stores (its rows are private to it)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 8 + (lane >> 3), c = lane & 7;
            const int qrow = q0 + wave * 64 + r;
            const u32x4 val = *reinterpret_cast<const u32x4*>(ost + r * E_OROW + c * 16);
            if (qrow < Tq) {
                __bf16* op = reinterpret_cast<__bf16*>(p.o) + o_off + (long long)qrow * p.ld_o + (long long)hh * p.o_hs;
                *reinterpret_cast<u32x4*>(op + c * 8) = val;
            }
        }
        }
        if (v < 0) break;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AFHIP_ENC_STAMPS
    if (p.dbg && wave == 0) {      // every workgroup's duration at [128 + blockIdx.x] (the load-balance question: tools/attn_enc_stamps.py)
        const unsigned long long t_ = __builtin_amdgcn_s_memtime() - t_start;
        const int off_ = (128 + (int)blockIdx.x) * 8;
        asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(t_), "s"(p.dbg), "s"(off_) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

}  // namespace

// Called by afhip_attention (attention.hip) for the shape this form covers; returns false when it does not apply.
bool afhip_attention_enc64(const afhip_attn_args* a, hipStream_t s) {
    if (!afhip_opt(AFHIP_OPT_ATTN_ENC64)) return false;   // A/B switch
    if (a->dtype != AFHIP_BF16 || a->hd != 64 || !a->q_prescaled || a->causal || a->key_split > 0 || a->n_q != a->n_kv || a->Tq != a->Tk) return false;
    if (a->new_k || a->seq_pos) return false;
    if ((long long)a->Tk * a->ld_kv * 2 >= (1ll << 31) || (long long)a->Tq * a->ld_q * 2 >= (1ll << 31)) return false;      // 32-bit DMA offsets inside one (clip, head), K / V and Q
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
    p.out_fp8 = a->out_fp8; p.out_inv = a->out_scale_inv;
#ifdef AFHIP_ENC_STAMPS
    { const char* e = getenv("AFHIP_ENC_DBGPTR"); p.dbg = e ? (unsigned long long*)strtoull(e, nullptr, 16) : nullptr; }
#endif
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done))
        (void)hipFuncSetAttribute((const void*)attn_enc64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS_TOTAL);
    // persistent: one workgroup per CU (the whole register file per wave), a multiple of 8 so that a workgroup keeps its XCD residue
    int ncu = afhip_cu_count();
    ncu = ncu >= 8 ? (ncu / 8) * 8 : ncu;
    int grid = nblk < ncu ? (int)nblk : ncu;
    if (afhip_opt(AFHIP_OPT_ENC64_ONE_BLOCK_PER_WG)) grid = (int)nblk;   // A/B switch
    hipLaunchKernelGGL(attn_enc64_kernel, dim3((unsigned)grid), dim3(256), E_LDS_TOTAL, s, p);
    return true;
}
