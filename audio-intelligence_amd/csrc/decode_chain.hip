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
#include "img_phase.h"
#include "decode_chain.h"
#include <stdlib.h>

namespace {

using stream::ImgPhase;
using stream::ImgDesc;

struct PickArgs {
    const int32_t* iv; int n_iv;                // allowed id intervals (device)
    int64_t* prev_token; int64_t* out_tokens; int32_t* finished_at;
    int B, step, eos, eot;
    int32_t* seq_pos; int32_t* step_counter;
    const float* am_val; const int* am_idx; int am_n;
};

struct ChainArgs {
    ImgDesc o, gu, down, qkv, head;
    int phases;          // AFHIP_PH_* bits, executed in the order embed, o, gate/up, down, q|k|v, head, pick
    // embed phase: x[b, :] = sum_s table[id(b, s)], id(b, 0) = prev_token[b], id(b, s > 0) = 0 (lm/parallel.py:260,479,540-541);
    // leaves x as plain rows, as an image with the first layer's input gain applied, and its sums of squares
    const int64_t* prev_token; const char* table; char* x; char* ximg; const char* gain; float* ss; int ss_n; int B, S, H, vocab;
    PickArgs pick;
    unsigned* bar;       // [8 shards x 32 words] arrival counters + [256] abort word; zeroed once per decode step
    int bar0;            // barrier rounds completed by the earlier launches of this step
    int32_t* status;     // set to 1 when a barrier timed out
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

// one phase of the launch.  `fresh`: nothing of this launch precedes it -- its image was written by an earlier launch.  `last`: nothing
// of this launch follows -- the kernel boundary publishes its output.
template <typename PH> __device__ __forceinline__ void chained(const ImgDesc& d, char* smem, unsigned* bar, int& rounds, bool& fresh, bool last, int32_t* status) {
    PH ph(d, smem);
    const ImgDesc& p = d;
    (void)p;
    if (fresh) {
        ph.template begin<true>();
        ph.template run<true>();
    } else {
        ph.template begin<false>();                            // weight window in flight ...
        grid_wait(bar, rounds, status);                        // ... while the workgroups meet
        CH_STAMP(6);
        ph.template run<false>();
    }
    if (!last) {
        grid_arrive(bar);
        ++rounds;
    }
    CH_STAMP(7);
    fresh = false;
}

template <int RM> __device__ __forceinline__ void embed_phase(const ChainArgs& a, char* smem) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.B) {
        const int b = blockIdx.x;
        long long id0 = a.prev_token[b];
        id0 = id0 < 0 ? 0 : (id0 >= a.vocab ? a.vocab - 1 : id0);
        float sq = 0.f;
        for (int c = tid; c < (a.H >> 3); c += 512) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s = 0; s < a.S; ++s) {
                const u32x4 v = ld16(a.table + ((s == 0 ? id0 : 0ll) * a.H + c * 8) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[2 * e] += stream::bflo(v[e]); acc[2 * e + 1] += stream::bfhi(v[e]); }
            }
            const u32x4 gv = ld16(a.gain + c * 16);
            u32x4 o, oi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = stream::pack2(acc[2 * e], acc[2 * e + 1]);
                const float x0 = stream::bflo(o[e]), x1 = stream::bfhi(o[e]);
                sq += x0 * x0;
                sq += x1 * x1;
                oi[e] = stream::pack2(x0 * stream::bflo(gv[e]), x1 * stream::bfhi(gv[e]));
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, stream::rsrc_of(a.x), (int)(((long long)b * a.H + c * 8) * 2), 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(oi, stream::rsrc_of(a.ximg), (int)stream::img_off(RM, b, c * 8), 0, 16);
        }
        float* sh = reinterpret_cast<float*>(smem);
        sq = wave_sum(sq);
        if ((tid & 63) == 0) sh[tid >> 6] = sq;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += sh[w];
        for (int j = tid; j < a.ss_n; j += 512)          // one entry per workgroup of the CONSUMING phase's producers: the whole sum in entry 0
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(j == 0 ? tot : 0.f), stream::rsrc_of(a.ss), (b * a.ss_n + j) * 4, 0, 16);
    }
}

__device__ __forceinline__ void pick_phase(const PickArgs& k) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int G = k.am_n;
    const int step = k.step_counter ? k.step_counter[0] : k.step;
    __syncthreads();                                       // every thread has read the step before thread 0 bumps it
    for (int r = wave; r < k.B; r += 8) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = lane; j < G; j += 64) {
            const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(stream::rsrc_of(k.am_val), (r * G + j) * 4, 0, 16));
            const int i = (int)__builtin_amdgcn_raw_buffer_load_b32(stream::rsrc_of(k.am_idx), (r * G + j) * 4, 0, 16);
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

template <int RM>
__global__ __launch_bounds__(512) void decode_chain_kernel(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int rounds = a.bar0;
    bool fresh = true;
    const int ph = a.phases;
    auto is_last = [&](int bit) { return (ph & ~(bit | (bit - 1))) == 0; };      // no phase bit above `bit`

    if (ph & AFHIP_PH_EMBED) {
        embed_phase<RM>(a, smem);
        if (!is_last(AFHIP_PH_EMBED)) { grid_arrive(a.bar); ++rounds; }
        fresh = false;
    }
    if (ph & AFHIP_PH_O) chained<ImgPhase<1, false, false, RM, 7, true>>(a.o, smem, a.bar, rounds, fresh, is_last(AFHIP_PH_O), a.status);
    if (ph & AFHIP_PH_GU) chained<ImgPhase<2, true, true, RM, 4, true>>(a.gu, smem, a.bar, rounds, fresh, is_last(AFHIP_PH_GU), a.status);
    if (ph & AFHIP_PH_DOWN) chained<ImgPhase<1, false, false, RM, 7, true>>(a.down, smem, a.bar, rounds, fresh, is_last(AFHIP_PH_DOWN), a.status);
    if (ph & AFHIP_PH_QKV) chained<ImgPhase<2, false, true, RM, 7, true>>(a.qkv, smem, a.bar, rounds, fresh, is_last(AFHIP_PH_QKV), a.status);
    if (ph & AFHIP_PH_HEAD) chained<ImgPhase<4, false, true, RM, 2, true>>(a.head, smem, a.bar, rounds, fresh, is_last(AFHIP_PH_HEAD), a.status);
    if ((ph & AFHIP_PH_PICK) && blockIdx.x == 0) {
        if (!fresh) grid_wait(a.bar, rounds, a.status);
        pick_phase(a.pick);
    }
}

// ---- one phase per launch (AFHIP_DECODE_CHAIN=1): the same phase code as a lean kernel of its own -- one descriptor as the kernel
//      argument, plain (cached) accesses, the kernel boundary as the hand-off
template <typename PH> __global__ __launch_bounds__(512) void img_phase_kernel(ImgDesc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PH ph(d, smem);
    ph.template begin<true>();
    ph.template run<true>();
}
template <int RM> __global__ __launch_bounds__(512) void embed_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[64];
    embed_phase<RM>(a, smem);
}
__global__ __launch_bounds__(512) void pick_kernel(PickArgs k) { pick_phase(k); }

template <typename PH> void launch_phase(const ImgDesc& d, int grid, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done))
        (void)hipFuncSetAttribute((const void*)img_phase_kernel<PH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH::lds_bytes());
    hipLaunchKernelGGL((img_phase_kernel<PH>), dim3((unsigned)grid), dim3(512), PH::lds_bytes(), s, d);
}
template <int RM> void launch_single8(const ChainArgs& a, int cus, hipStream_t s) {      // e4m3 weights (W8A16)
    switch (a.phases) {
        case AFHIP_PH_O: launch_phase<ImgPhase<1, false, false, RM, 7, false, true>>(a.o, cus, s); break;
        case AFHIP_PH_GU: launch_phase<ImgPhase<2, true, true, RM, 4, false, true>>(a.gu, cus, s); break;
        case AFHIP_PH_DOWN: launch_phase<ImgPhase<1, false, false, RM, 7, false, true>>(a.down, cus, s); break;
        case AFHIP_PH_QKV: launch_phase<ImgPhase<2, false, true, RM, 4, false, true>>(a.qkv, cus, s); break;
        default: launch_phase<ImgPhase<4, false, true, RM, 2, false, true>>(a.head, cus, s); break;
    }
}
template <int RM> void launch_single(const ChainArgs& a, int cus, hipStream_t s) {
    switch (a.phases) {
        case AFHIP_PH_EMBED: hipLaunchKernelGGL(embed_kernel<RM>, dim3((unsigned)a.B), dim3(512), 0, s, a); break;
        case AFHIP_PH_O: launch_phase<ImgPhase<1, false, false, RM, 7, false>>(a.o, cus, s); break;
        case AFHIP_PH_GU: launch_phase<ImgPhase<2, true, true, RM, 4, false>>(a.gu, cus, s); break;
        case AFHIP_PH_DOWN: launch_phase<ImgPhase<1, false, false, RM, 7, false>>(a.down, cus, s); break;
        case AFHIP_PH_QKV: launch_phase<ImgPhase<2, false, true, RM, 7, false>>(a.qkv, cus, s); break;
        case AFHIP_PH_HEAD: launch_phase<ImgPhase<4, false, true, RM, 2, false>>(a.head, cus, s); break;
        default: hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(512), 0, s, a.pick); break;
    }
}

constexpr size_t chain_lds() { return ImgPhase<4, false, true, 8, 2, true>::lds_bytes(); }

void fill(ImgDesc& p, const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, long long ldc, long long ldres,
          int tile_rows, const float* ss_in, int ss_n, float eps) {
    p = ImgDesc{};
    p.A = (const char*)A; p.W = (const char*)W; p.bias = (const char*)bias; p.res = (const char*)res; p.C = (char*)C;
    p.M = M; p.N = N; p.K = K; p.ldw = K; p.ldc = ldc; p.ldres = ldres; p.tile_rows = tile_rows;
    p.ss_in = ss_in; p.ss_n = ss_n; p.eps = eps;
}

}  // namespace

// ---- host side (C++ linkage: called by llm.hip) ----------------------------------------------------------------------------
int afhip_decode_chain_mode() {
    return afhip_opt(AFHIP_OPT_DECODE_CHAIN);   // 0 = off, 1 = one phase per launch (default), 2 = chained phases
}

bool afhip_decode_chain_supported(const afhip_llm_weights* w, int B) {
    if (afhip_decode_chain_mode() == 0) return false;
    const int cus = afhip_cu_count();
    const int H = w->hidden, I = w->inter, qw = (w->n_q + 2 * w->n_kv) * w->hd, AO = w->n_q * w->hd;
    if (w->dtype != AFHIP_BF16 || B < 1 || B > 16 || w->n_stream < 1 || cus > 1024) return false;
    if (H % 512 != 0 || I % 512 != 0 || AO % 512 != 0) return false;
    if (cdiv(H, cus) > 16 || cdiv(qw, cus) > 32 || (2 * I) % 64 != 0 || 2 * I < 8192) return false;
    if ((size_t)16 * 2 * I * 2 >= (1ull << 31) || (size_t)B * qw * 2 >= (1ull << 31)) return false;
    return true;
}

// [0, 2048) barrier words | argmax partials 2 x [16][1024] | x sums of squares [16][1024] f32 | x image | attention image | SwiGLU image
static size_t off_am(int) { return 2048; }
static size_t off_ss() { return 2048 + (size_t)2 * 16 * 1024 * 4; }
static size_t off_ximg() { return off_ss() + (size_t)16 * 1024 * 4; }
size_t afhip_decode_chain_scratch_bytes(const afhip_llm_weights* w, int B) {
    const size_t rm = B <= 8 ? 8 : 16;
    return off_ximg() + rm * 2 * ((size_t)w->hidden + (size_t)w->n_q * w->hd + (size_t)w->inter) + 1024;
}

int afhip_decode_chain_launch(const afhip_chain_step& c, hipStream_t s, int* rounds_out) {
    const afhip_llm_weights* w = c.w;
    const int cus = afhip_cu_count();
    const int H = w->hidden, I = w->inter, qw = (w->n_q + 2 * w->n_kv) * w->hd, AO = w->n_q * w->hd, B = c.B;
    const int rm = B <= 8 ? 8 : 16;
    char* sc = (char*)c.scratch;
    ChainArgs a = {};
    a.bar = (unsigned*)sc;
    a.status = c.st->status ? c.st->status : (int32_t*)((unsigned*)sc + 300);
    float* am_val = (float*)(sc + off_am(0));
    int* am_idx = (int*)(am_val + 16 * 1024);
    float* xss = (float*)(sc + off_ss());
    char* ximg = sc + off_ximg();
    char* attimg = ximg + (size_t)rm * H * 2;
    char* actimg = attimg + (size_t)rm * AO * 2;
    a.bar0 = c.bar0;
    a.phases = c.phases;
    int rounds = 0, nph = 0;
    for (int b = 1; b <= AFHIP_PH_PICK; b <<= 1) nph += (c.phases & b) ? 1 : 0;
    rounds = nph > 0 ? nph - 1 : 0;                               // one barrier between consecutive phases of the launch
    const int rpw_h = cdiv(H, cus);                              // one equal share of output rows per CU (gemm_skinny.hip)
    const int rpw_q = cdiv(qw, cus);
    const bool f8 = w->qkv_w8 != nullptr;                        // W8A16 copies of the streamed weights (afhip.h): one-phase launches only
    if (c.phases & AFHIP_PH_EMBED) {
        a.prev_token = c.st->prev_token; a.table = (const char*)w->embed; a.x = c.x; a.ximg = ximg; a.gain = (const char*)w->ln1_w[0]; a.ss = xss; a.ss_n = cus;
        a.S = w->n_stream; a.H = H; a.vocab = w->vocab;
    }
    a.B = B;
    if (c.phases & AFHIP_PH_QKV) {
        const int l = c.qkv_layer;
        fill(a.qkv, ximg, f8 ? w->qkv_w8[l] : w->qkv_w[l], w->qkv_b[l], nullptr, c.qkv, B, qw, H, qw, 0, cdiv(rpw_q, 2), xss, cus, w->rms_eps);
        if (f8) a.qkv.w_scale = w->qkv_s[l];
    }
    if (c.phases & (AFHIP_PH_O | AFHIP_PH_GU | AFHIP_PH_DOWN)) {
        const int l = c.layer;
        // o: attention image -> x (+ residual), x image with the post-attention norm's gain, its sums of squares
        fill(a.o, attimg, f8 ? w->o_w8[l] : w->o_w[l], nullptr, c.x, c.x, B, H, AO, H, H, rpw_h, nullptr, 0, 0.f);
        if (f8) a.o.w_scale = w->o_s[l];
        a.o.img_out = ximg; a.o.img_gain = (const char*)w->ln2_w[l]; a.o.ss_out = xss;
        // SwiGLU pairs: the TR <= 16 gate rows per unit whose ceil(units / CUs) * TR is smallest (gemm_skinny.hip)
        int best = 16, best_cost = cdiv(cdiv(I, 16), cus) * 16;
        for (int tr = 15; tr >= 12; --tr) {
            const int cost = cdiv(cdiv(I, tr), cus) * tr;
            if (cost < best_cost) { best_cost = cost; best = tr; }
        }
        fill(a.gu, ximg, f8 ? w->gu_w8[l] : w->gu_w[l], nullptr, nullptr, nullptr, B, 2 * I, H, I, 0, best, xss, cus, w->rms_eps);
        if (f8) a.gu.w_scale = w->gu_s[l];
        a.gu.img_out = actimg;
        // down: SwiGLU image -> x (+ residual), x image with the NEXT norm's gain (next layer's input norm, or the final norm)
        fill(a.down, actimg, f8 ? w->down_w8[l] : w->down_w[l], nullptr, c.x, c.x, B, H, I, H, H, rpw_h, nullptr, 0, 0.f);
        if (f8) a.down.w_scale = w->down_s[l];
        a.down.img_out = ximg; a.down.img_gain = (const char*)(l + 1 < w->n_layers ? w->ln1_w[l + 1] : w->norm_w); a.down.ss_out = xss;
    }
    if (c.phases & AFHIP_PH_HEAD) {
        const int rows = c.st->head_rows > 0 && c.st->head_rows < w->vocab ? c.st->head_rows : w->vocab;
        const bool h8 = f8 && w->lm_head8 != nullptr;
        fill(a.head, ximg, h8 ? w->lm_head8 : w->lm_head, nullptr, nullptr, nullptr, B, rows, H, 0, 0, 16, xss, cus, w->rms_eps);
        if (h8) a.head.w_scale = w->lm_head_s;
        a.head.am_iv = c.st->allowed; a.head.am_n_iv = c.st->n_iv; a.head.am_val = am_val; a.head.am_idx = am_idx;
    }
    if (c.phases & AFHIP_PH_PICK) {
        a.pick.iv = c.st->allowed; a.pick.n_iv = c.st->n_iv;
        a.pick.prev_token = c.st->prev_token; a.pick.out_tokens = c.st->out_tokens; a.pick.finished_at = c.st->finished_at;
        a.pick.B = B; a.pick.step = c.step; a.pick.eos = c.st->eos_id; a.pick.eot = c.st->eot_id;
        a.pick.seq_pos = c.st->seq_pos; a.pick.step_counter = c.st->step_counter;
        a.pick.am_val = am_val; a.pick.am_idx = am_idx; a.pick.am_n = cus;
    }
    if (rounds_out) *rounds_out = rounds;
#ifdef AFHIP_STREAM_STAMPS
    {   // diagnostic build: AFHIP_STREAM_DBGPTR = [5 phases][workgroups][8] stamps of the launches of layer AFHIP_CHAIN_STAMP_LAYER
        const char* dp = getenv("AFHIP_STREAM_DBGPTR");
        const char* dl = getenv("AFHIP_CHAIN_STAMP_LAYER");
        unsigned long long* d = dp ? (unsigned long long*)strtoull(dp, nullptr, 0) : nullptr;
        if (d && dl && atoi(dl) == c.layer) {
            a.o.dbg = d; a.gu.dbg = d + 1 * cus * 8; a.down.dbg = d + 2 * cus * 8; a.qkv.dbg = d + 3 * cus * 8; a.head.dbg = d + 4 * cus * 8;
        }
    }
#endif
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done)) {
        (void)hipFuncSetAttribute((const void*)decode_chain_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds());
        (void)hipFuncSetAttribute((const void*)decode_chain_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds());
    }
    if (nph == 1) {
        const bool gemm = (c.phases & (AFHIP_PH_O | AFHIP_PH_GU | AFHIP_PH_DOWN | AFHIP_PH_QKV | AFHIP_PH_HEAD)) != 0;
        const bool w8 = gemm && f8 && !((c.phases & AFHIP_PH_HEAD) && w->lm_head8 == nullptr);
        if (w8) { if (rm == 8) launch_single8<8>(a, cus, s); else launch_single8<16>(a, cus, s); }
        else if (rm == 8) launch_single<8>(a, cus, s);
        else launch_single<16>(a, cus, s);
    } else if (rm == 8) hipLaunchKernelGGL(decode_chain_kernel<8>, dim3((unsigned)cus), dim3(512), chain_lds(), s, a);
    else hipLaunchKernelGGL(decode_chain_kernel<16>, dim3((unsigned)cus), dim3(512), chain_lds(), s, a);
    AFHIP_LAUNCH_CHECK();
    return 0;
}

// where the attention merge of the decode step leaves its output: the image the o phase reads (llm.hip)
void* afhip_decode_chain_att_image(const afhip_llm_weights* w, int B, void* scratch) {
    const int rm = B <= 8 ? 8 : 16;
    return (char*)scratch + off_ximg() + (size_t)rm * w->hidden * 2;
}
