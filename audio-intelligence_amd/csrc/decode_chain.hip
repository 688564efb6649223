// The dependent GEMMs of one greedy decode step as CHAINS inside one launch (bf16 weights, B <= 16 sequences).
//
// Per decoder layer the step of ParallelLLM._step (lm/parallel.py:570-597 over modeling_qwen2.py:258-299) is
//   q|k|v -> attention -> o (+x) -> gate/up (SwiGLU) -> down (+x)
// and every arrow is an all-to-all dependency on a [B, 3584 .. 18944] activation.  As separate launches each arrow costs a kernel
// boundary plus the ramp of the next weight stream -- tools/stream_stamps.py: the 33-MB q|k|v stream itself takes 4.5 us at 7 TB/s,
// the launch 10.8 us -- six times per layer, a fifth of the whole step.  Here one workgroup per CU stays resident and runs
//   [embed ->] q|k|v(0)                                            first launch of a step
//   o -> gate/up -> down -> q|k|v(l + 1)                            one launch per layer, between two attention launches
//   o -> gate/up -> down -> lm_head + argmax partials -> pick       last layer (greedy pick + stop bookkeeping of lm/parallel.py:494-513,599-601)
// with a grid barrier at every arrow.  A phase's first weight window is issued BEFORE the workgroup waits at the barrier in front of
// it, so HBM keeps streaming while the workgroups meet; the activations are handed over by sc1 (write-through) stores, one agent-scope
// counter add per workgroup after its stores have drained, an sc1 poll by one wave, a workgroup barrier, sc1 loads (cdna guide,
// Guideline 16, counter form: no L2 write-back, no invalidate, nothing that would drain the weight window).  Every spin is bounded: a
// barrier that is not met raises the abort word, all later waits return at once and the step's status word reports it.
#include "stream_phase.h"
#include "decode_chain.h"
#include <stdlib.h>

namespace {

using stream::StreamPhase;

struct PickArgs {
    const int32_t* iv; int n_iv;                // allowed id intervals (device)
    int64_t* prev_token; int64_t* out_tokens; int32_t* finished_at;
    int B, step, eos, eot;
    int32_t* seq_pos; int32_t* step_counter;
};

struct ChainArgs {
    SkinnyP o, gu, down, qkv, head;
    int has_embed, has_o, has_gu, has_down, has_qkv, has_head, has_pick;
    // embed phase: x[b, :] = sum_s table[id(b, s)], id(b, 0) = prev_token[b], id(b, s > 0) = 0 (lm/parallel.py:260,479,540-541)
    const int64_t* prev_token; const char* table; char* x; int B, S, H, vocab;
    PickArgs pick;
    unsigned* bar;       // [8 shards x 32 words] arrival counters + [256] abort word; zeroed once per decode step
    int bar0;            // barrier rounds completed by the earlier launches of this step
    int32_t* status;     // != NULL: set to 1 when a barrier timed out
};

#ifdef AFHIP_STREAM_STAMPS
#define CH_STAMP(k) do { if (p.dbg && threadIdx.x == 0) p.dbg[(long long)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CH_STAMP(k) do { } while (0)
#endif
constexpr int BAR_ABORT = 256;
constexpr int SPIN_LIMIT = 400000;

__device__ __forceinline__ void grid_arrive(unsigned* bar) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's (write-through) stores have left
    __syncthreads();                                           // ... and every wave's
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar + ((blockIdx.x & 7) << 5), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// round r (1-based over the whole step): every shard s has seen r x (workgroups with blockIdx & 7 == s) arrivals
__device__ __forceinline__ void grid_wait(unsigned* bar, int r, int32_t* status) {
    if (threadIdx.x < 64) {
        const int s = threadIdx.x & 7;
        const unsigned need = (unsigned)r * (((unsigned)gridDim.x - s + 7u) >> 3);
        int spins = 0;
        for (;;) {
            const unsigned v = __hip_atomic_load(bar + (s << 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned ab = __hip_atomic_load(bar + BAR_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= need) || ab != 0u) break;
            if (++spins > SPIN_LIMIT) {                        // bounded: never hang the GPU
                if (threadIdx.x == 0) {
                    __hip_atomic_store(bar + BAR_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (status) status[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}

template <typename PH> __device__ __forceinline__ void chained(const SkinnyP& p, char* smem, unsigned* bar, int& rounds, bool& fresh, int32_t* status) {
    PH ph(p, smem);
    if (fresh) {
        // first phase of the launch: its activations come from an earlier launch, the weight window only
        // has to wait for nothing
        ph.template begin<false>();
        ph.template run<false>();
    } else {
        ph.template begin<false>();                            // weight window in flight ...
        grid_wait(bar, rounds, status);                        // ... while the workgroups meet
        CH_STAMP(6);
        ph.template run<false>();
    }
    grid_arrive(bar);
    CH_STAMP(7);
    ++rounds;
    fresh = false;
}

template <int RM>
__global__ __launch_bounds__(512) void decode_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int rounds = a.bar0;
    bool fresh = true;

    if (a.has_embed) {
        if ((int)blockIdx.x < a.B) {
            const int b = blockIdx.x;
            long long id0 = a.prev_token[b];
            id0 = id0 < 0 ? 0 : (id0 >= a.vocab ? a.vocab - 1 : id0);
            for (int c = tid; c < (a.H >> 3); c += 512) {
                float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int s = 0; s < a.S; ++s) {
                    const u32x4 v = ld16(a.table + ((s == 0 ? id0 : 0ll) * a.H + c * 8) * 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[2 * e] += stream::bflo(v[e]); acc[2 * e + 1] += stream::bfhi(v[e]); }
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = stream::pack2(acc[2 * e], acc[2 * e + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(o, stream::rsrc_of(a.x), (int)(((long long)b * a.H + c * 8) * 2), 0, 16);
            }
        }
        grid_arrive(a.bar);
        ++rounds;
        fresh = false;
    }
    if (a.has_o) chained<StreamPhase<SKINNY_A_PLAIN, 1, false, false, RM, 7, true>>(a.o, smem, a.bar, rounds, fresh, a.status);
    if (a.has_gu) chained<StreamPhase<SKINNY_A_RMSNORM, 2, true, false, RM, 4, true>>(a.gu, smem, a.bar, rounds, fresh, a.status);
    if (a.has_down) chained<StreamPhase<SKINNY_A_PLAIN, 1, false, true, RM, 7, true>>(a.down, smem, a.bar, rounds, fresh, a.status);
    if (a.has_qkv) chained<StreamPhase<SKINNY_A_RMSNORM, 2, false, false, RM, 7, true>>(a.qkv, smem, a.bar, rounds, fresh, a.status);
    if (a.has_head) {
        constexpr int HNT = RM == 8 ? 4 : 2;
        chained<StreamPhase<SKINNY_A_RMSNORM, HNT, false, false, RM, RM == 8 ? 2 : 4, true>>(a.head, smem, a.bar, rounds, fresh, a.status);
    }
    if (a.has_pick && blockIdx.x == 0) {
        grid_wait(a.bar, rounds, a.status);
        const PickArgs& k = a.pick;
        const int lane = tid & 63, wave = tid >> 6;
        const int G = gridDim.x;
        const int step = k.step_counter ? k.step_counter[0] : k.step;
        __syncthreads();                                       // every thread has read the step before thread 0 bumps it
        for (int r = wave; r < k.B; r += 8) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int j = lane; j < G; j += 64) {
                const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(stream::rsrc_of(a.head.am_val), (r * G + j) * 4, 0, 16));
                const int i = (int)__builtin_amdgcn_raw_buffer_load_b32(stream::rsrc_of(a.head.am_idx), (r * G + j) * 4, 0, 16);
                if (v > best || (v == best && i < bi)) { best = v; bi = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) {
                // an all -inf / NaN row falls back to the first allowed id (masked_argmax_final_kernel, llm.hip)
                const int64_t t = (bi == 0x7fffffff) ? (k.n_iv > 0 ? k.iv[0] : 0) : bi;
                k.out_tokens[(long long)step * k.B + r] = t;
                k.prev_token[r] = t;
                if ((t == k.eos || t == k.eot) && k.finished_at[r] < 0) k.finished_at[r] = step;
                if (k.seq_pos) k.seq_pos[r] += 1;
            }
        }
        if (k.step_counter && tid == 0) k.step_counter[0] = step + 1;
    }
}

template <int RM> size_t chain_lds(int H, int I) {
    size_t m = 0;
    auto up = [&](size_t v) { m = v > m ? v : m; };
    up(StreamPhase<SKINNY_A_PLAIN, 1, false, false, RM, 7, true>::lds_bytes(H));
    up(StreamPhase<SKINNY_A_RMSNORM, 2, true, false, RM, 4, true>::lds_bytes(H));
    up(StreamPhase<SKINNY_A_PLAIN, 1, false, true, RM, 7, true>::lds_bytes(I));
    up(StreamPhase<SKINNY_A_RMSNORM, 2, false, false, RM, 7, true>::lds_bytes(H));
    up(StreamPhase<SKINNY_A_RMSNORM, RM == 8 ? 4 : 2, false, false, RM, RM == 8 ? 2 : 4, true>::lds_bytes(H));
    return m;
}

void fill(SkinnyP& p, const void* A, const void* W, const void* bias, const void* res, const void* norm_w, float eps, void* C, int M, int N, int K,
          long long lda, long long ldc, long long ldres, int tile_rows, int n_tiles, int swiglu) {
    p = SkinnyP{};
    p.A = (const char*)A; p.W = (const char*)W; p.bias = (const char*)bias; p.res = (const char*)res; p.norm_w = (const char*)norm_w;
    p.C = (char*)C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = K; p.ldc = ldc; p.ldres = ldres;
    p.out_f32 = 0; p.norm_eps = eps; p.swiglu_out = swiglu; p.a_rows = M <= 8 ? 8 : 16; p.tile_rows = tile_rows; p.n_tiles = n_tiles;
}

}  // namespace

// ---- host side (C++ linkage: called by llm.hip) ----------------------------------------------------------------------------
bool afhip_decode_chain_supported(const afhip_llm_weights* w, int B) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("AFHIP_DECODE_CHAIN"); off = (e && e[0] == '0') ? 1 : 0; }   // A/B switch (read once per process)
    if (off) return false;
    const int cus = afhip_cu_count();
    const int H = w->hidden, I = w->inter, qw = (w->n_q + 2 * w->n_kv) * w->hd, AO = w->n_q * w->hd;
    if (w->dtype != AFHIP_BF16 || B < 1 || B > 16 || w->n_stream < 1) return false;
    if (H % 512 != 0 || I % 512 != 0 || AO % 512 != 0 || AO > 2 * H) return false;
    if (cdiv(H, cus) > 16 || cdiv(qw, cus) > 32 || (2 * I) % 64 != 0 || 2 * I < 8192) return false;
    if ((size_t)B * H * 2 >= (1ull << 31) || (size_t)B * 2 * I * 2 >= (1ull << 31)) return false;
    const int rm = B <= 8 ? 8 : 16;
    if ((rm == 8 ? chain_lds<8>(H > AO ? H : AO, I) : chain_lds<16>(H > AO ? H : AO, I)) > 160 * 1024) return false;
    return true;
}

size_t afhip_decode_chain_scratch_bytes(int B) { (void)B; return 2048 + (size_t)16 * 1024 * 8; }

int afhip_decode_chain_launch(const afhip_chain_step& c, hipStream_t s, int* rounds_out) {
    const afhip_llm_weights* w = c.w;
    const int cus = afhip_cu_count();
    const int H = w->hidden, I = w->inter, qw = (w->n_q + 2 * w->n_kv) * w->hd, AO = w->n_q * w->hd, B = c.B;
    ChainArgs a = {};
    a.bar = (unsigned*)c.scratch;
    float* am_val = (float*)((char*)c.scratch + 2048);
    int* am_idx = (int*)(am_val + 16 * 1024);
    a.status = c.st->status ? c.st->status : (int32_t*)((unsigned*)c.scratch + 300);
    a.bar0 = c.bar0;
    int rounds = 0;
    const int rpw_h = cdiv(H, cus);                              // one equal share of output rows per CU (gemm_skinny.hip)
    const int rpw_q = cdiv(qw, cus);
    const int nt_q = rpw_q <= 16 ? 1 : 2, tr_q = rpw_q <= 16 ? rpw_q : cdiv(rpw_q, 2);
    auto fill_qkv = [&](int l) {
        fill(a.qkv, c.x, w->qkv_w[l], w->qkv_b[l], nullptr, w->ln1_w[l], w->rms_eps, c.qkv, B, qw, H, H, qw, 0, tr_q, nt_q, 0);
        // the chain's q|k|v phase is the two-tile form: a one-tile share is padded to two tiles of half the rows
        if (nt_q == 1) { a.qkv.tile_rows = cdiv(rpw_q, 2); a.qkv.n_tiles = 2; }
        a.has_qkv = 1; ++rounds;
    };
    if (c.layer < 0) {
        a.has_embed = 1; ++rounds;
        a.prev_token = c.st->prev_token; a.table = (const char*)w->embed; a.x = c.x; a.B = B; a.S = w->n_stream; a.H = H; a.vocab = w->vocab;
        fill_qkv(0);
    } else {
        const int l = c.layer;
        fill(a.o, c.att, w->o_w[l], nullptr, c.x, nullptr, 0.f, c.x, B, H, AO, AO, H, H, rpw_h, 1, 0);
        a.has_o = 1; ++rounds;
        // SwiGLU pairs: the TR <= 16 gate rows per unit whose ceil(units / CUs) * TR is smallest (gemm_skinny.hip)
        int best = 16, best_cost = cdiv(cdiv(I, 16), cus) * 16;
        for (int tr = 15; tr >= 12; --tr) {
            const int cost = cdiv(cdiv(I, tr), cus) * tr;
            if (cost < best_cost) { best_cost = cost; best = tr; }
        }
        fill(a.gu, c.x, w->gu_w[l], nullptr, nullptr, w->ln2_w[l], w->rms_eps, c.act, B, 2 * I, H, H, I, 0, best, 2, 1);
        a.has_gu = 1; ++rounds;
        fill(a.down, c.act, w->down_w[l], nullptr, c.x, nullptr, 0.f, c.x, B, H, I, I, H, H, rpw_h, 1, 0);
        a.has_down = 1; ++rounds;
        if (l + 1 < w->n_layers) {
            fill_qkv(l + 1);
        } else {
            const int rows = c.st->head_rows > 0 && c.st->head_rows < w->vocab ? c.st->head_rows : w->vocab;
            fill(a.head, c.x, w->lm_head, nullptr, nullptr, w->norm_w, w->rms_eps, nullptr, B, rows, H, H, 0, 0, 16, B <= 8 ? 4 : 2, 0);
            a.head.am_iv = c.st->allowed; a.head.am_n_iv = c.st->n_iv; a.head.am_val = am_val; a.head.am_idx = am_idx;
            a.has_head = 1; ++rounds;
            a.has_pick = 1;
            a.pick.iv = c.st->allowed; a.pick.n_iv = c.st->n_iv;
            a.pick.prev_token = c.st->prev_token; a.pick.out_tokens = c.st->out_tokens; a.pick.finished_at = c.st->finished_at;
            a.pick.B = B; a.pick.step = c.step; a.pick.eos = c.st->eos_id; a.pick.eot = c.st->eot_id;
            a.pick.seq_pos = c.st->seq_pos; a.pick.step_counter = c.st->step_counter;
        }
    }
    if (rounds_out) *rounds_out = rounds;
#ifdef AFHIP_STREAM_STAMPS
    {   // diagnostic build: AFHIP_STREAM_DBGPTR = [5 phases][workgroups][8] stamps of the launch of layer AFHIP_CHAIN_STAMP_LAYER
        const char* dp = getenv("AFHIP_STREAM_DBGPTR");
        const char* dl = getenv("AFHIP_CHAIN_STAMP_LAYER");
        unsigned long long* d = dp ? (unsigned long long*)strtoull(dp, nullptr, 0) : nullptr;
        if (d && dl && atoi(dl) == c.layer) {
            a.o.dbg = d; a.gu.dbg = d + 1 * cus * 8; a.down.dbg = d + 2 * cus * 8; a.qkv.dbg = d + 3 * cus * 8; a.head.dbg = d + 4 * cus * 8;
        }
    }
#endif
    const int rm = B <= 8 ? 8 : 16;
    const size_t lds = rm == 8 ? chain_lds<8>(H > AO ? H : AO, I) : chain_lds<16>(H > AO ? H : AO, I);
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done)) {
        (void)hipFuncSetAttribute((const void*)decode_chain_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)decode_chain_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (rm == 8) hipLaunchKernelGGL(decode_chain_kernel<8>, dim3((unsigned)cus), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(decode_chain_kernel<16>, dim3((unsigned)cus), dim3(512), lds, s, a);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
