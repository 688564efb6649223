// Persistent "ping-pong" bf16 GEMM for gfx950: C[M,N] = epilogue(A[M,K] . W[N,K]^T), the throughput path of the
// encoder / prefill projections (afhip_gemm dispatches here when the shape qualifies, see gemm_pp_eligible).
//
// One 512-thread workgroup per CU walks a list of 256 x 256 output tiles.  Its 8 waves form two groups of four
// (group = wave >> 2; waves w and w + 4 share a SIMD).  A K tile (64 deep) is processed in four PHASES, one 64 x 32
// quadrant of the wave's 128 x 64 output per phase (16 x v_mfma_f32_16x16x32_bf16).  A phase is
//     [LOAD section: ds_read fragments, issue one half-tile of LDS-DMA, counted vmcnt]  s_barrier
//     [MFMA section: 16 MFMAs]                                                           s_barrier
// and group 1 runs one barrier behind group 0, so on every SIMD one wave's MFMA section overlaps its partner's LOAD
// section: the matrix pipe sees back-to-back MFMAs while LDS reads / DMA issue happen in the other wave.
//
// Operand staging: each K tile is four 16-KiB half-tiles (A0, A1, B0, B1: 128 rows x 128 B), two stages = 128 KiB of
// LDS, filled by buffer_load_dwordx4 ... lds (no VGPR staging).  The DMA stream runs ~2 K tiles ahead of the MFMAs and
// NEVER drains inside the loop: at most 4 half-tiles (8 DMA instructions per wave) stay in flight across barriers
// (s_waitcnt vmcnt(8)).  The stream is continuous across output tiles, so the next tile's first K tiles land while
// the current tile's epilogue runs, and the epilogue's global stores drain under the next tile's MFMAs.
//
// Schedule for K tile g in stage S = g & 1 (the quadrant order keeps B0 in registers from phase 0 to phase 3):
//   phase 0: read A0(g), B0(g)   issue B1(g+1) -> stage S^1   wait vmcnt(8) [retires B1(g)]     MFMA (A0, B0)
//   phase 1: read B1(g)          issue A1(g+1) -> stage S^1   wait vmcnt(8) [retires A1(g)]     MFMA (A0, B1)
//   phase 2: read A1(g)          issue A0(g+2) -> stage S                                      MFMA (A1, B1)
//   phase 3:                     issue B0(g+2) -> stage S     wait vmcnt(8) [A0, B0 of g+1]     MFMA (A1, B0)
// RAW: a buffer is read one phase (>= one barrier of every wave) after the wait that retires its DMA.
// WAR: a buffer is re-filled >= 2 phases after the phase that last read it (the staggered group's reads of phase p
//      are retired by its lgkmcnt(0) before the barrier that opens phase p + 2's LOAD sections).
//
// Row maps.  A half h holds, for reading group r, tile rows r*128 + h*64 + (0..63); W half h holds, for column-wave
// wn, the 32 output columns wn*64 + h*32 + (0..31), permuted (LDS row j*16 + 4q + k  <->  column q*8 + j*4 + k) so
// that the 8 accumulator values a lane owns for one output row are 8 CONSECUTIVE columns: the epilogue stores (and
// the residual loads) are 16 B per lane straight from registers, 64 contiguous bytes per row per instruction.
// LDS image: 128-B rows, 16-B chunk c of row r at slot c ^ ((r >> 1) & 7) (conflict-free ds_read_b128); the DMA
// writes linearly, so the swizzle is applied to its per-lane SOURCE chunk.
//
// Epilogues (template flags): + bias, erf-GELU, + residual (all 16 loads of a lane issued before the math), SwiGLU pair
// (32-row interleaved gate/up weights -> C [M, N/2]), and the two halves of the LayerNorm fold (afhip.h,
// afhip_gemm_args.ln_stats): LNFOLD applies rstd[m] (acc - mean[m] colsum[n]) + bias'[n] on a GEMM over the raw residual
// stream, STATS emits per-row (sum, sum of squares) partials of the rows it stores for the next LayerNorm.
#include "common.h"
#include <stdlib.h>
#include <type_traits>


namespace {

// The schedule below is the one that survived the round-2 / round-3 same-box A/Bs (profiles/r03_gemm_pp_sched_ab.txt): s_setprio(1) around
// each MFMA cluster; a phase's half-tile DMA issued in its LOAD section AFTER the fragment reads, no blanket lgkmcnt(0) in front of the
// MFMAs, fragment reads K-half-major; K half 0 of the next tile's A0 fragments read one phase early (8 / 4 / 8 / 4 fragment reads per LOAD
// section).  Measured and removed: the DMA inside the MFMA section (5-7 % slower), one more phase of DMA lookahead (0-7 % slower), two
// phases of 32 MFMAs per K tile (no gain), tile-major fragment reads.
constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64;
constexpr int PP_HALF = 16384;           // 128 rows x 128 B
constexpr int PP_STAGE = 4 * PP_HALF;    // A0 A1 B0 B1
constexpr int PP_OFF_A0 = 0, PP_OFF_A1 = PP_HALF, PP_OFF_B0 = 2 * PP_HALF, PP_OFF_B1 = 3 * PP_HALF;
constexpr int PP_LDS = 2 * PP_STAGE;     // 128 KiB
// LayerNorm-folded forms: the epilogue's operands of a tile -- colsum[256] | bias'[256] | (mean, rstd)[256] = 4 KiB -- are fetched by ONE
// LDS-DMA instruction per wave at the top of the tile's last K-tile pair and read back with ds_read in the epilogue.  Loaded where they
// are used (round 1-3) they were 16 global loads per lane issued BEHIND the run-ahead operand stream, and the epilogue stood still for
// one memory latency per tile: 6-7 % of the q | k | v and fc1 launches (timing build, round 4).  Two areas, by tile parity.
constexpr int PP_EPI = 4096;
constexpr int PP_LDS_TOTAL = PP_LDS + 2 * PP_EPI;

struct PPArgs {
    const char* A;
    const char* W;
    const bf16* bias;
    const bf16* res;
    bf16* C;
    int M, N, K;
    unsigned lda2, ldw2;      // row pitches in BYTES
    long long ldc, ldres;     // elements
    int act;
    int tiles_m, tiles_n, group_m;
    const float* ln_stats;    // LNFOLD: [M][2] row mean, rstd
    const float* ln_colsum;   // LNFOLD: [N] sum_k W'[n,k]
    const float* ln_bias;     // LNFOLD: [N] f32 bias (b + W beta)
    float* stats_out;         // STATS: [N/64][M][2] partial (sum, sum of squares) of the stored rows
    const float* a_scale;     // F8: [M] f32 scale of each e4m3 activation row; NULL: a_scale_const for every row
    float a_scale_const;
    int out_fp8;              // F8: C is e4m3 bytes [M, ldc] = sat(value * out_inv) (afhip_gemm_args.out_fp8)
    float out_inv;
    const float* w_scale;     // F8: [N] f32 scale of each e4m3 weight row (output channel)
    int esz;                  // operand element size in bytes: 2 (bf16) or 1 (e4m3)
#ifdef AFHIP_PP_STAMPS
    unsigned long long* dbg;  // AFHIP_PP_DBGPTR: [2][64] stamps (waves 0 and 4 of workgroup 0, second output tile)
#endif
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ int pp_xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7, s = id >> 3;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + s;
}

// DMA cursor: which (output tile, K tile) the stream is at.  All members are wave-uniform (SGPRs).
struct PPCur {
    const char* abase;   // A + m0 * lda2
    const char* wbase;   // W + n0 * ldw2
    unsigned anrec;      // valid bytes from abase (rows >= M read as out-of-range)
    unsigned wnrec;
    int koff;            // kt * 128 bytes
    int it;              // index into this workgroup's tile list
};

__device__ __forceinline__ void pp_tile_coords(const PPArgs& p, int v, int& m0, int& n0) {
    const int nwg = p.tiles_m * p.tiles_n;
    const int wg = pp_xcd_remap(v, nwg);
    const int width = p.group_m * p.tiles_n;
    const int grp = wg / width, first_m = grp * p.group_m;
    const int gsize = (p.tiles_m - first_m) < p.group_m ? (p.tiles_m - first_m) : p.group_m;
    const int tm = first_m + (wg % width) % gsize, tn = (wg % width) / gsize;
    m0 = tm * PP_BM;
    n0 = tn * PP_BN;
}

__device__ __forceinline__ void pp_cur_set_tile(const PPArgs& p, PPCur& c, int it) {
    int m0, n0;
    pp_tile_coords(p, (int)blockIdx.x + it * (int)gridDim.x, m0, n0);
    c.it = it;
    c.koff = 0;
    c.abase = p.A + (long long)m0 * p.lda2;
    c.wbase = p.W + (long long)n0 * p.ldw2;
    const int rows = (p.M - m0) < PP_BM ? (p.M - m0) : PP_BM;
    c.anrec = (unsigned)(rows - 1) * p.lda2 + (unsigned)(p.K * p.esz);
    c.wnrec = (unsigned)(PP_BN - 1) * p.ldw2 + (unsigned)(p.K * p.esz);
}

// next K tile of the stream; past the end of the tile list the cursor stays on the last K tile (the DMAs issued from
// it land in buffers nobody reads again and only keep the vmcnt bookkeeping uniform)
__device__ __forceinline__ void pp_cur_advance(const PPArgs& p, PPCur& c, int n_my) {
    const int kend = p.K * p.esz - 128;
    if (c.koff < kend) {
        c.koff += 128;
    } else if (c.it + 1 < n_my) {
        pp_cur_set_tile(p, c, c.it + 1);
    }
}

__device__ __forceinline__ void pp_dma_half(const char* base, unsigned nrec, int soff, int voff0, int voff1, char* lds_dst) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_dst, 16, voff0, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds_dst + 8192), 16, voff1, soff, 0, 0);
}

// counted wait of the LOAD sections of phases 3, 0, 1: the half-tile read in the NEXT phase has landed, younger ones stay in flight.
// DMA in the LOAD section: issued so far includes this phase's -> 4 half-tiles (8 instructions) younger; DMA in the MFMA section:
// this phase's is not issued yet -> 3 half-tiles (6 instructions) younger.
template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#define PP_WAIT_VM() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define PP_WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define PP_WAIT_VM10() asm volatile("s_waitcnt vmcnt(10)" ::: "memory")
#define PP_BARRIER()                              \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#ifdef AFHIP_PP_STAMPS
// one s_memtime and one SCALAR store straight to the debug buffer (gfx9 still has s_store): no VGPR, no LDS, and -- unlike a vector or
// FLAT store -- nothing on vmcnt, which the DMA stream's counted waits own
#define PP_STAMP() do { if (st_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); const int off_ = ((wave ? 64 : 0) + st_idx) * 8; \
        if (st_idx < 64) asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(t_), "s"(p.dbg), "s"(off_) : "memory"); ++st_idx; } } while (0)
#else
#define PP_STAMP() do { } while (0)
#endif

// F8: both operands are OCP e4m3 bytes (K tile = 128 elements = the same 128-byte LDS rows, so staging, swizzle, phases and
// vmcnt counts are untouched); a quadrant is 8 x v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 127) -- twice the
// bf16 MFMA rate -- and the per-row activation scale / per-channel weight scale multiply the f32 accumulator in the epilogue.
// A lane's 32 K bytes of a step are the two 16-byte chunks the bf16 form reads (q and 4 + q: conflict-free); that K order is not
// the natural one, but it is the same on both operands, which is all a dot product needs.
template <int ACT, bool HAS_BIAS, bool HAS_RES, bool LNFOLD, bool STATS, bool F8 = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(PPArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int c16 = lane & 15, q4 = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int n_my = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nk2 = (p.K * (F8 ? 1 : 2)) / (4 * PP_BK);          // pairs of 128-byte K tiles
#ifdef AFHIP_PP_STAMPS
    const bool st_wave = p.dbg && blockIdx.x == 0 && (wave == 0 || wave == 4);
    bool st_on = false;
    int st_idx = 0;
#endif

    // ---- DMA lane constants: instruction u of this wave fills LDS rows (u*8 + wave)*8 + lrow of a half-tile ----
    const int lrow = lane >> 3, lslot = lane & 7;
    const int src_chunk = lslot ^ (((wave & 1) << 2) + (lrow >> 1));
    int voffA[2][2], voffB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m_local = u * 128 + h * 64 + wave * 8 + lrow;
            voffA[h][u] = m_local * (int)p.lda2 + src_chunk * 16;
            const int rho = (u * 8 + wave) * 8 + lrow;
            const int pos = rho & 15;
            const int n_local = (rho >> 5) * 64 + h * 32 + (pos >> 2) * 8 + ((rho >> 4) & 1) * 4 + (pos & 3);
            voffB[h][u] = n_local * (int)p.ldw2 + src_chunk * 16;
        }
    char* const dma_dst = smem + wave * 1024;     // + stage + half offset

    // ---- fragment read offsets (bytes inside a half-tile): row*128 + ((kk*4 + q) ^ swz(row))*16, swz = (c >> 1) & 7 ----
    const int sw = (c16 >> 1) & 7;
    const int aoff0 = (grp * 64 + c16) * 128 + ((q4 ^ sw) << 4);
    const int aoff1 = (grp * 64 + c16) * 128 + (((4 + q4) ^ sw) << 4);
    const int boff0 = (wn * 32 + c16) * 128 + ((q4 ^ sw) << 4);
    const int boff1 = (wn * 32 + c16) * 128 + (((4 + q4) ^ sw) << 4);

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    bf16x8 fa0k0[4];                        // K half 0 of the NEXT tile's A0 fragments, read one phase early
    // ... except in the two forms whose epilogue (bias + residual + row statistics, or + GELU) already needs all 256 registers: the 16
    // fragment registers that then stay live across it spill 7-24 VGPRs, and fc2 (the statistics form) gained nothing from the schedule
    constexpr bool EARLY_A0 = !F8 && !(HAS_BIAS && HAS_RES && (STATS || ACT == AFHIP_ACT_GELU));
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef int v8i_t __attribute__((ext_vector_type(8)));
    const int unit_scale = 0x7f7f7f7f;      // E8M0 127 = 2^0 in every byte: block scales off
    v8i_t fa8[4], fb0_8[2], fb1_8[2];       // F8: a lane's 32 K bytes of a step as ONE 8-dword MFMA operand (loaded as two 16-B halves)

    // ---- prologue: A0 B0 B1 A1 of stream tile 0 (stage 0), A0 B0 of stream tile 1 (stage 1) ----
    PPCur c1, c2;
    pp_cur_set_tile(p, c1, 0);
    pp_dma_half(c1.abase, c1.anrec, c1.koff, voffA[0][0], voffA[0][1], dma_dst + PP_OFF_A0);
    pp_dma_half(c1.wbase, c1.wnrec, c1.koff, voffB[0][0], voffB[0][1], dma_dst + PP_OFF_B0);
    pp_dma_half(c1.wbase, c1.wnrec, c1.koff, voffB[1][0], voffB[1][1], dma_dst + PP_OFF_B1);
    pp_dma_half(c1.abase, c1.anrec, c1.koff, voffA[1][0], voffA[1][1], dma_dst + PP_OFF_A1);
    pp_cur_advance(p, c1, n_my);                  // stream tile 1
    pp_dma_half(c1.abase, c1.anrec, c1.koff, voffA[0][0], voffA[0][1], dma_dst + PP_STAGE + PP_OFF_A0);
    pp_dma_half(c1.wbase, c1.wnrec, c1.koff, voffB[0][0], voffB[0][1], dma_dst + PP_STAGE + PP_OFF_B0);
    c2 = c1;
    pp_cur_advance(p, c2, n_my);                  // stream tile 2
    PP_WAIT_VM8();                                // A0(0), B0(0) of this wave have landed
    PP_BARRIER();
    if constexpr (EARLY_A0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa0k0[i] = *reinterpret_cast<const bf16x8*>(smem + PP_OFF_A0 + aoff0 + i * 2048);
    }
    if (grp == 1) PP_BARRIER();                   // stagger: group 1 runs one barrier behind group 0

    // 16 MFMAs of one quadrant (with the DMA-in-MFMA variant, now removed, the phase's LDS-DMA half-tile was issued between the two K halves, in the
    // issue slots the wave's own MFMAs leave free (an MFMA holds the port 8 of its 16 cycles), instead of lengthening the LOAD
    // section its partner's matrix pipe waits on: 5-7 % slower)
    auto mfma_quad = [&](auto ha_tag, auto hb_tag, bf16x8 (&fbx)[2][2], auto&& dma) {
        constexpr int HA = decltype(ha_tag)::value, HB = decltype(hb_tag)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
            v8i_t (&fb8)[2] = (HB == 0) ? fb0_8 : fb1_8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    // inline asm with the accumulator TIED (dst == srcC): left to the builtin, hipcc allocates a fresh destination per
                    // MFMA of this VGPR-form instruction and spills ~150 registers in the K loop (3x slower than the bf16 form).
                    // Hazards: operands come from ds_read (the compiler waits on lgkmcnt for asm inputs); the accumulators are
                    // next read by VALU code a barrier and the whole tile-coordinate computation later.
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[HA][i][HB][j]) : "v"(fb8[j]), "v"(fa8[i]), "v"(unit_scale));
        }
#pragma unroll
        for (int kk = 0; kk < (F8 ? 0 : 2); ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[HA][i][HB][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbx[j][kk], (EARLY_A0 && HA == 0 && kk == 0) ? fa0k0[i] : fa[i][kk], acc[HA][i][HB][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // fragment reads in the order the MFMAs consume them (K half 0 of every tile, then K half 1): with per-fragment lgkmcnt waits
    // the first MFMAs start while the second half is still in flight
    auto read_a = [&](const char* half) {
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4i_t lo = *reinterpret_cast<const v4i_t*>(half + aoff0 + i * 2048), hi = *reinterpret_cast<const v4i_t*>(half + aoff1 + i * 2048);
                fa8[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i][0] = *reinterpret_cast<const bf16x8*>(half + aoff0 + i * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i][1] = *reinterpret_cast<const bf16x8*>(half + aoff1 + i * 2048);
    };
    auto read_b = [&](const char* half, bf16x8 (&fbx)[2][2]) {
        if constexpr (F8) {
            v8i_t (&fb8)[2] = (&fbx == &fb0) ? fb0_8 : fb1_8;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const v4i_t lo = *reinterpret_cast<const v4i_t*>(half + boff0 + j * 2048), hi = *reinterpret_cast<const v4i_t*>(half + boff1 + j * 2048);
                fb8[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) fbx[j][0] = *reinterpret_cast<const bf16x8*>(half + boff0 + j * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) fbx[j][1] = *reinterpret_cast<const bf16x8*>(half + boff1 + j * 2048);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // one K tile in stage S
    // XTRA: vector-memory instructions issued just before this K tile that are NOT part of the operand stream (the epilogue operands'
    // DMA): they sit in the in-order queue between the stream's older and newer half-tiles, so every counted wait of THIS K tile allows
    // that many more outstanding; from the next K tile on the awaited half-tiles are younger than they are and the counts are the usual ones
    auto ktile = [&](auto s_tag, auto x_tag) {
        constexpr int S = decltype(s_tag)::value;
        constexpr int XTRA = decltype(x_tag)::value;
        const char* st = smem + S * PP_STAGE;
        char* d_same = dma_dst + S * PP_STAGE;
        char* d_other = dma_dst + (S ^ 1) * PP_STAGE;
        auto dma0 = [&]() { pp_dma_half(c1.wbase, c1.wnrec, c1.koff, voffB[1][0], voffB[1][1], d_other + PP_OFF_B1); };
        auto dma1 = [&]() { pp_dma_half(c1.abase, c1.anrec, c1.koff, voffA[1][0], voffA[1][1], d_other + PP_OFF_A1); };
        auto dma2 = [&]() { pp_dma_half(c2.abase, c2.anrec, c2.koff, voffA[0][0], voffA[0][1], d_same + PP_OFF_A0); };
        auto dma3 = [&]() { pp_dma_half(c2.wbase, c2.wnrec, c2.koff, voffB[0][0], voffB[0][1], d_same + PP_OFF_B0); };
        {
        // phase 0
        read_b(st + PP_OFF_B0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (EARLY_A0) {     // K half 0 of these fragments was read in phase 3 of the previous K tile
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i][1] = *reinterpret_cast<const bf16x8*>(st + PP_OFF_A0 + aoff1 + i * 2048);
        } else {
            read_a(st + PP_OFF_A0);
        }
        dma0();
        pp_wait_vm<8 + XTRA>();
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        mfma_quad(I0{}, I0{}, fb0, dma0);
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        // phase 1
        read_b(st + PP_OFF_B1, fb1);
        dma1();
        pp_wait_vm<8 + XTRA>();
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        mfma_quad(I0{}, I1{}, fb1, dma1);
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        // phase 2
        read_a(st + PP_OFF_A1);
        dma2();
        if constexpr (EARLY_A0) pp_wait_vm<8 + XTRA>();   // A0 of the next tile has landed one phase early (it has the slack: issued 5 phases ago)
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        mfma_quad(I1{}, I1{}, fb1, dma2);
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        // phase 3
        if constexpr (EARLY_A0) {     // balance the LOAD sections (12 / 4 / 8 / 0 fragment reads -> 8 / 4 / 8 / 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa0k0[i] = *reinterpret_cast<const bf16x8*>(smem + (S ^ 1) * PP_STAGE + PP_OFF_A0 + aoff0 + i * 2048);
        }
        dma3();
        pp_wait_vm<8 + XTRA>();
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        mfma_quad(I1{}, I0{}, fb0, dma3);
        PP_STAMP();
        PP_BARRIER();
        PP_STAMP();
        }
        c1 = c2;
        pp_cur_advance(p, c2, n_my);
    };

    for (int it = 0; it < n_my; ++it) {
        for (int k2 = 0; k2 < nk2 - (LNFOLD ? 1 : 0); ++k2) {
#ifdef AFHIP_PP_STAMPS
            st_on = st_wave && it == 1 && k2 == 4;
#endif
            ktile(I0{}, I0{});
#ifdef AFHIP_PP_STAMPS
            st_on = false;
#endif
            ktile(I1{}, I0{});
        }
        if constexpr (LNFOLD) {
            // the tile's last K-tile pair, with the epilogue operands' DMA in front of it: wave w (and w + 4, same bytes) fetches piece w & 3
            // -- colsum, bias', (mean, rstd) of rows 0..127, of rows 128..255; rows past M are out of the descriptor's range and read 0
            int m0e, n0e;
            pp_tile_coords(p, (int)blockIdx.x + it * (int)gridDim.x, m0e, n0e);
            const int piece = wave & 3;
            const char* ebase;
            int enrec;
            if (piece == 0) { ebase = reinterpret_cast<const char*>(p.ln_colsum + n0e); enrec = 1024; }
            else if (piece == 1) { ebase = reinterpret_cast<const char*>(p.ln_bias + n0e); enrec = 1024; }
            else {
                const int r0 = m0e + (piece - 2) * 128;
                int rows = p.M - r0;
                rows = rows < 0 ? 0 : (rows > 128 ? 128 : rows);
                ebase = reinterpret_cast<const char*>(p.ln_stats + 2 * (long long)(rows > 0 ? r0 : 0));
                enrec = rows * 8;
            }
            __amdgpu_buffer_rsrc_t er = __builtin_amdgcn_make_buffer_rsrc((void*)ebase, 0, enrec, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(er, (lds_ptr_t)(smem + PP_LDS + (it & 1) * PP_EPI + piece * 1024), 16, lane * 16, 0, 0, 0);
            ktile(I0{}, I1{});
            ktile(I1{}, I0{});
        }
#ifdef AFHIP_PP_STAMPS
        st_on = st_wave && it == 1;
        if (st_on) st_idx = 40;
        PP_STAMP();                               // 40: epilogue starts
#endif
        // ---- epilogue straight from registers (no LDS, no barrier): lane (c, q) owns, for each of its 8 rows, the 8
        //      consecutive columns q*8 .. q*8+7 of both 32-column halves of the wave's 64 columns ----
        int m0, n0;
        pp_tile_coords(p, (int)blockIdx.x + it * (int)gridDim.x, m0, n0);
        const int ncol = n0 + wn * 64 + q4 * 8;
        if constexpr (ACT == AFHIP_ACT_SWIGLU) {
            // gate / up rows are interleaved in 32-row blocks (ParallelLLM.pack): this wave's first 32 columns are the gate,
            // its second 32 the up projection of the same 32 outputs; C is [M, N/2]
            const int ocol = ((n0 + wn * 64) >> 1) + q4 * 8;
            float sg[8], su[8];
            if constexpr (F8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sg[e] = p.w_scale[ncol + e]; su[e] = p.w_scale[ncol + 32 + e]; }
            }
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = m0 + grp * 128 + ha * 64 + i * 16 + c16;
                    float ra = 1.f;
                    if constexpr (F8) ra = p.a_scale ? p.a_scale[m < p.M ? m : p.M - 1] : p.a_scale_const;
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float g = acc[ha][i][0][e >> 2][e & 3], u = acc[ha][i][1][e >> 2][e & 3];
                        if constexpr (F8) { g *= ra * sg[e]; u *= ra * su[e]; }
                        o[e] = (bf16)(silu(g) * u);
                    }
                    if (m < p.M) *reinterpret_cast<bf16x8*>(p.C + (long long)m * p.ldc + ocol) = o;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        acc[ha][i][hb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc[ha][i][hb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            continue;
        }
        float bv[2][8], cv[2][8];
        float wsc[2][8];
        if constexpr (F8) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.w_scale + ncol + hb * 32), s1 = *reinterpret_cast<const f32x4*>(p.w_scale + ncol + hb * 32 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wsc[hb][e] = s0[e]; wsc[hb][4 + e] = s1[e]; }
            }
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if constexpr (LNFOLD) {
                // LayerNorm folded into this GEMM: A is the raw residual stream, W' = W diag(gamma), and
                //   LN(x) W^T + b = rstd[m] (x W'^T - mean[m] colsum[n]) + (b + W beta)[n]
                // from the tile's LDS area (fetched two K tiles ago; the last K tile's first wait and its barriers made it complete and visible)
                const float* ef = reinterpret_cast<const float*>(smem + PP_LDS + (it & 1) * PP_EPI) + wn * 64 + q4 * 8 + hb * 32;
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(ef), c1 = *reinterpret_cast<const f32x4*>(ef + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ef + 256), b1 = *reinterpret_cast<const f32x4*>(ef + 260);
#pragma unroll
                for (int e = 0; e < 4; ++e) { cv[hb][e] = c0[e]; cv[hb][4 + e] = c1[e]; bv[hb][e] = b0[e]; bv[hb][4 + e] = b1[e]; }
            } else if constexpr (HAS_BIAS) {
                const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(p.bias + ncol + hb * 32);
#pragma unroll
                for (int e = 0; e < 8; ++e) { bv[hb][e] = (float)b8[e]; cv[hb][e] = 0.f; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { bv[hb][e] = 0.f; cv[hb][e] = 0.f; }
            }
        }
        // residual: all 16 loads of the lane are issued before the first use (one exposed latency per tile instead of
        // eight); rows past M are clamped for the load and masked at the store
        bf16x8 r8[2][4][2];
        if constexpr (HAS_RES && !F8) {
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int m = m0 + grp * 128 + ha * 64 + i * 16 + c16;
                    m = m < p.M ? m : p.M - 1;
                    r8[ha][i][0] = *reinterpret_cast<const bf16x8*>(p.res + (long long)m * p.ldres + ncol);
                    r8[ha][i][1] = *reinterpret_cast<const bf16x8*>(p.res + (long long)m * p.ldres + ncol + 32);
                }
        }
        // LNFOLD: the 8 (mean, rstd) pairs of the lane's rows, issued together like the residual
        f32x2 st2[2][4];
        if constexpr (LNFOLD) {
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st2[ha][i] = *reinterpret_cast<const f32x2*>(smem + PP_LDS + (it & 1) * PP_EPI + 2048 + (grp * 128 + ha * 64 + i * 16 + c16) * 8);
                }
        }
        PP_STAMP();                               // 41: epilogue operand loads issued
        float rs[2][4], rss[2][4];
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + grp * 128 + ha * 64 + i * 16 + c16;
                const bool ok = m < p.M;
                if constexpr (HAS_RES && F8) {
                    // F8 keeps 16 scale registers live: the residual rows are loaded per row group instead of all 16 up front
                    const int mr = ok ? m : p.M - 1;
                    r8[ha][i][0] = *reinterpret_cast<const bf16x8*>(p.res + (long long)mr * p.ldres + ncol);
                    r8[ha][i][1] = *reinterpret_cast<const bf16x8*>(p.res + (long long)mr * p.ldres + ncol + 32);
                }
                float mean = 0.f, rstd = 1.f, asc = 1.f;
                if constexpr (LNFOLD) { mean = st2[ha][i][0]; rstd = st2[ha][i][1]; }
                if constexpr (F8) asc = p.a_scale ? p.a_scale[ok ? m : p.M - 1] : p.a_scale_const;
                rs[ha][i] = 0.f; rss[ha][i] = 0.f;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float v;
                        if constexpr (LNFOLD) v = fmaf(rstd, fmaf(-mean, cv[hb][e], acc[ha][i][hb][e >> 2][e & 3]), bv[hb][e]);
                        else if constexpr (F8) v = fmaf(acc[ha][i][hb][e >> 2][e & 3], asc * wsc[hb][e], bv[hb][e]);
                        else v = acc[ha][i][hb][e >> 2][e & 3] + bv[hb][e];
                        if constexpr (ACT == AFHIP_ACT_GELU) v = F8 ? gelu_tanh_fast(v) : gelu_act<bf16>(v);
                        if constexpr (HAS_RES) v += (float)r8[ha][i][hb][e];
                        o[e] = (bf16)v;
                        if constexpr (STATS) { const float r = (float)o[e]; rs[ha][i] += r; rss[ha][i] = fmaf(r, r, rss[ha][i]); }
                    }
                    if constexpr (F8 && !HAS_RES && !STATS) {
                        if (p.out_fp8) {
                            // statically quantised output: the consumer's e4m3 A operand, written here instead of bf16 + a quantisation pass.
                            // o[] (the bf16 rounding) is dead code on this path; the conversion takes the f32 values
                            float w8[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float v;
                                v = fmaf(acc[ha][i][hb][e >> 2][e & 3], asc * wsc[hb][e], bv[hb][e]);
                                if constexpr (ACT == AFHIP_ACT_GELU) v = F8 ? gelu_tanh_fast(v) : gelu_act<bf16>(v);
                                w8[e] = __builtin_amdgcn_fmed3f(v * p.out_inv, -448.f, 448.f);
                            }
                            int lo = 0, hi = 0;
                            lo = __builtin_amdgcn_cvt_pk_fp8_f32(w8[0], w8[1], lo, false);
                            lo = __builtin_amdgcn_cvt_pk_fp8_f32(w8[2], w8[3], lo, true);
                            hi = __builtin_amdgcn_cvt_pk_fp8_f32(w8[4], w8[5], hi, false);
                            hi = __builtin_amdgcn_cvt_pk_fp8_f32(w8[6], w8[7], hi, true);
                            if (ok) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.C) + (long long)m * p.ldc + ncol + hb * 32) = uint2{(unsigned)lo, (unsigned)hi};
                            acc[ha][i][hb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                            acc[ha][i][hb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                            continue;
                        }
                    }
                    if (ok) *reinterpret_cast<bf16x8*>(p.C + (long long)m * p.ldc + ncol + hb * 32) = o;
                    acc[ha][i][hb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[ha][i][hb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        if constexpr (STATS) {
            // row statistics of what was STORED (the next LayerNorm's input), summed over this wave's 64 columns: the 4 lanes
            // c16 + 16 q own a row.  All 32 exchanges are issued back to back; the [N/64][M] partials are merged by
            // afhip_ln_stats_finalize
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) { rs[ha][i] += __shfl_xor(rs[ha][i], 16, 64); rss[ha][i] += __shfl_xor(rss[ha][i], 16, 64); }
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) { rs[ha][i] += __shfl_xor(rs[ha][i], 32, 64); rss[ha][i] += __shfl_xor(rss[ha][i], 32, 64); }
            if (q4 == 0) {
#pragma unroll
                for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = m0 + grp * 128 + ha * 64 + i * 16 + c16;
                        if (m < p.M) *reinterpret_cast<f32x2*>(p.stats_out + 2 * ((long long)((n0 >> 6) + wn) * p.M + m)) = f32x2{rs[ha][i], rss[ha][i]};
                    }
            }
        }
        PP_STAMP();                               // 42: epilogue math done, stores issued
#ifdef AFHIP_PP_STAMPS
        st_on = false;
#endif
    }
    if (grp == 0) PP_BARRIER();                   // pairs with group 1's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
#ifdef AFHIP_PP_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

bool pp_enabled() {
    return afhip_opt(AFHIP_OPT_GEMM_PP) != 0;   // A/B switch for benchmarking (afhip_set_option flips it inside one process)
}

int pp_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int ACT, bool HB, bool HR, bool LF = false, bool ST = false, bool F8 = false>
void pp_launch_t(const PPArgs& p, int grid, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done)) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<ACT, HB, HR, LF, ST, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL);
    }
    hipLaunchKernelGGL((gemm_pp_kernel<ACT, HB, HR, LF, ST, F8>), dim3((unsigned)grid), dim3(512), PP_LDS_TOTAL, s, p);
}

}  // namespace

// false when the A/B switch AFHIP_GEMM_PP=0 is set: callers that would use the LayerNorm-folded forms must not (encoder.hip)
bool gemm_pp_available() { return pp_enabled(); }

// Shapes the ping-pong kernel takes: bf16 in / bf16 out, no implicit conv, whole 256-column tiles,
// an even number of 64-deep K tiles, 16-byte aligned rows everywhere, operands addressable with 32-bit byte offsets.
bool gemm_pp_eligible(const afhip_gemm_args* a) {
    if (a->a_fp8) {
        // e4m3 operands: this kernel is the only implementation; afhip_gemm reports what is missing when this says no
        if (a->dtype != AFHIP_BF16 || a->conv_C > 0 || a->out_f32 || a->res_row_mod > 0 || !(a->a_scale || a->a_scale_const > 0.f) || !a->w_scale) return false;
        if (a->out_fp8 && (a->residual || a->act == AFHIP_ACT_SWIGLU || !(a->out_scale_inv > 0.f) || (a->ldc % 16))) return false;
        if (a->ln_stats) return false;
        if (a->row_stats_out && (a->act != AFHIP_ACT_NONE || !a->bias || !a->residual || a->out_fp8)) return false;   // the fc2 form only
        if (a->act == AFHIP_ACT_SWIGLU && (a->bias || a->residual)) return false;
        if (a->M < 1 || (a->N % PP_BN) != 0 || (a->K % 256) != 0) return false;
        if ((a->lda % 16) || (a->ldw % 16) || (a->ldc % 8) || ((uintptr_t)a->A % 16) || ((uintptr_t)a->W % 16) || ((uintptr_t)a->C % 16)) return false;
        if (a->bias && ((uintptr_t)a->bias % 16)) return false;
        if (a->residual && ((a->ldres % 8) || ((uintptr_t)a->residual % 16))) return false;
        if (((uintptr_t)a->w_scale % 16) || (long long)a->lda * PP_BM >= (1ll << 31) || (long long)a->ldw * PP_BN >= (1ll << 31)) return false;
        return true;
    }
    if (!pp_enabled()) return false;
    if (a->dtype != AFHIP_BF16 || a->conv_C > 0 || a->out_f32 || a->res_row_mod > 0) return false;
    if (a->act == AFHIP_ACT_SWIGLU && (a->bias || a->residual)) return false;
    if (a->ln_stats && (a->act == AFHIP_ACT_SWIGLU || a->residual || a->row_stats_out || !a->ln_colsum || !a->ln_bias)) return false;
    if (a->row_stats_out && (a->act != AFHIP_ACT_NONE || !a->bias || !a->residual)) return false;
    // small M is a throughput heuristic only (the 128 x 128 kernel wastes less of a near-empty tile); the LayerNorm-folded forms
    // exist only here, so they take any M (a packed ragged batch can be a few hundred rows)
    if ((a->M < 512 && !a->ln_stats && !a->row_stats_out) || a->M < 1 || (a->N % PP_BN) != 0 || (a->K % (2 * PP_BK)) != 0) return false;
    if ((a->lda % 8) || (a->ldw % 8) || (a->ldc % 8) || ((uintptr_t)a->A % 16) || ((uintptr_t)a->W % 16) || ((uintptr_t)a->C % 16)) return false;
    if (a->bias && ((uintptr_t)a->bias % 16)) return false;
    if (a->residual && ((a->ldres % 8) || ((uintptr_t)a->residual % 16))) return false;
    if ((long long)a->lda * 2 * PP_BM >= (1ll << 31) || (long long)a->ldw * 2 * PP_BN >= (1ll << 31)) return false;
    return true;
}

int gemm_pp_launch(const afhip_gemm_args* a, int group_m, hipStream_t s) {
    PPArgs p;
    p.A = (const char*)a->A; p.W = (const char*)a->W;
    p.bias = (const bf16*)a->bias; p.res = (const bf16*)a->residual; p.C = (bf16*)a->C;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.esz = a->a_fp8 ? 1 : 2;
    p.lda2 = (unsigned)(a->lda * p.esz); p.ldw2 = (unsigned)(a->ldw * p.esz);
    p.a_scale = a->a_scale; p.w_scale = a->w_scale;
    p.a_scale_const = a->a_scale_const; p.out_fp8 = a->a_fp8 ? a->out_fp8 : 0; p.out_inv = a->out_scale_inv;
    p.ldc = a->ldc; p.ldres = a->ldres;
    p.act = a->act;
    p.tiles_m = cdiv(a->M, PP_BM); p.tiles_n = a->N / PP_BN; p.group_m = group_m;
#ifdef AFHIP_PP_STAMPS
    { const char* e = getenv("AFHIP_PP_DBGPTR"); p.dbg = e ? (unsigned long long*)strtoull(e, nullptr, 16) : nullptr; }
#endif
    const long long nwg = (long long)p.tiles_m * p.tiles_n;
    AFHIP_CHECK(nwg < (1ll << 30), "afhip_gemm: grid too large");
    const int ncu = pp_num_cus();
    const int grid = nwg < ncu ? (int)nwg : ncu;
    const bool hb = a->bias != nullptr, hr = a->residual != nullptr;
    p.ln_stats = a->ln_stats; p.ln_colsum = a->ln_colsum; p.ln_bias = a->ln_bias; p.stats_out = a->row_stats_out;
    if (a->a_fp8) {
        if (a->act == AFHIP_ACT_SWIGLU) pp_launch_t<AFHIP_ACT_SWIGLU, false, false, false, false, true>(p, grid, s);
        else if (a->act == AFHIP_ACT_GELU) {
            if (hb && hr) pp_launch_t<AFHIP_ACT_GELU, true, true, false, false, true>(p, grid, s);
            else if (hb) pp_launch_t<AFHIP_ACT_GELU, true, false, false, false, true>(p, grid, s);
            else if (hr) pp_launch_t<AFHIP_ACT_GELU, false, true, false, false, true>(p, grid, s);
            else pp_launch_t<AFHIP_ACT_GELU, false, false, false, false, true>(p, grid, s);
        } else {
            if (hb && hr && a->row_stats_out) pp_launch_t<AFHIP_ACT_NONE, true, true, false, true, true>(p, grid, s);
            else if (hb && hr) pp_launch_t<AFHIP_ACT_NONE, true, true, false, false, true>(p, grid, s);
            else if (hb) pp_launch_t<AFHIP_ACT_NONE, true, false, false, false, true>(p, grid, s);
            else if (hr) pp_launch_t<AFHIP_ACT_NONE, false, true, false, false, true>(p, grid, s);
            else pp_launch_t<AFHIP_ACT_NONE, false, false, false, false, true>(p, grid, s);
        }
        return 0;
    }
    if (a->ln_stats) {
        if (a->act == AFHIP_ACT_GELU) pp_launch_t<AFHIP_ACT_GELU, false, false, true, false>(p, grid, s);
        else pp_launch_t<AFHIP_ACT_NONE, false, false, true, false>(p, grid, s);
    } else if (a->row_stats_out) {
        pp_launch_t<AFHIP_ACT_NONE, true, true, false, true>(p, grid, s);
    } else if (a->act == AFHIP_ACT_SWIGLU) {
        pp_launch_t<AFHIP_ACT_SWIGLU, false, false>(p, grid, s);
    } else if (a->act == AFHIP_ACT_GELU) {
        if (hb && hr) pp_launch_t<AFHIP_ACT_GELU, true, true>(p, grid, s);
        else if (hb) pp_launch_t<AFHIP_ACT_GELU, true, false>(p, grid, s);
        else if (hr) pp_launch_t<AFHIP_ACT_GELU, false, true>(p, grid, s);
        else pp_launch_t<AFHIP_ACT_GELU, false, false>(p, grid, s);
    } else {
        if (hb && hr) pp_launch_t<AFHIP_ACT_NONE, true, true>(p, grid, s);
        else if (hb) pp_launch_t<AFHIP_ACT_NONE, true, false>(p, grid, s);
        else if (hr) pp_launch_t<AFHIP_ACT_NONE, false, true>(p, grid, s);
        else pp_launch_t<AFHIP_ACT_NONE, false, false>(p, grid, s);
    }
    return 0;
}
