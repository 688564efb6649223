// The GEMM launches of one greedy decode step (bf16 or e4m3 weights, B <= 16 sequences), each a persistent imaged phase (img_phase.h).
//
// Per decoder layer the step of ParallelLLM._step (lm/parallel.py:570-597 over modeling_qwen2.py:258-299) is
//   q|k|v -> attention -> o (+x) -> gate/up (SwiGLU) -> down (+x)
// and every arrow is an all-to-all dependency on a [B, 3584 .. 18944] activation.  Each producer leaves its output as the fragment-order
// image the next GEMM streams (and x also as plain rows for the residual adds); the launches of a step are
//   embed                                                       x, its image with layer 0's input gain, its sums of squares
//   per layer: q|k|v -> [attention + merge: attention.hip] -> o -> gate/up -> down
//   lm_head + argmax partials -> pick                            (greedy pick + stop bookkeeping of lm/parallel.py:494-513,599-601)
// Measured and removed (profiles/r04_decode_layer_stamps_chain.txt): the same phases CHAINED inside one launch per layer behind grid
// barriers (sc1 write-through hand-off, one agent-scope counter add per workgroup, bounded polls), the next phase's weight window in
// flight across the barrier.  A barrier took 2-5 us from the last arrival -- its poll and its arrival counter queue behind the very
// prefetch that was meant to hide it -- which is what a kernel boundary plus the next launch's ramp costs: 3.54-3.60 ms per step
// against 3.27 for one launch per phase.
#include "img_phase.h"
#include "decode_phases.h"
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

struct PhaseArgs {
    ImgDesc o, gu, down, qkv, head;
    int phases;          // the AFHIP_PH_* bit of this launch
    // embed phase: x[b, :] = sum_s table[id(b, s)], id(b, 0) = prev_token[b], id(b, s > 0) = 0 (lm/parallel.py:260,479,540-541);
    // leaves x as plain rows, as an image with the first layer's input gain applied, and its sums of squares
    const int64_t* prev_token; const char* table; char* x; char* ximg; const char* gain; float* ss; int ss_n; int B, S, H, vocab;
    PickArgs pick;
};

template <int RM> __device__ __forceinline__ void embed_phase(const PhaseArgs& a, char* smem) {
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
            st16(a.x + ((long long)b * a.H + c * 8) * 2, o);
            st16(a.ximg + stream::img_off(RM, b, c * 8), oi);
        }
        float* sh = reinterpret_cast<float*>(smem);
        sq = wave_sum(sq);
        if ((tid & 63) == 0) sh[tid >> 6] = sq;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += sh[w];
        for (int j = tid; j < a.ss_n; j += 512)          // one entry per workgroup of the CONSUMING phase's producers: the whole sum in entry 0
            a.ss[b * a.ss_n + j] = j == 0 ? tot : 0.f;
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
            const float v = k.am_val[r * G + j];
            const int i = k.am_idx[r * G + j];
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

// ---- one phase per launch: one descriptor as the kernel argument, the kernel boundary as the hand-off
template <typename PH> __global__ __launch_bounds__(512) void img_phase_kernel(ImgDesc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PH ph(d, smem);
    ph.begin();
    ph.run();
}
template <int RM> __global__ __launch_bounds__(512) void embed_kernel(PhaseArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[64];
    embed_phase<RM>(a, smem);
}
__global__ __launch_bounds__(512) void pick_kernel(PickArgs k) { pick_phase(k); }

// K steps in flight per wave (rolling window depth) of the bf16 phases; measured choices (tools/decode_probe.py 8 96 790, one box, two passes:
// down 4 / 5 instead of 7: +0.8 %, o 4: +0.7 %, gate/up 3 / 5 instead of 4: +0.4 / +1.0 %, q|k|v 4 instead of 7: -0.5 %)
#ifndef DP_DEPTH_O
#define DP_DEPTH_O 7        // K = 3584: a wave's 7 steps all in flight
#endif
#ifndef DP_DEPTH_DOWN
#define DP_DEPTH_DOWN 7
#endif
#ifndef DP_DEPTH_QKV
#define DP_DEPTH_QKV 4        // 3.237 against 3.254 ms per step with 7
#endif
#ifndef DP_DEPTH_GU
#define DP_DEPTH_GU 4
#endif
#ifndef DP_HEAD_NT
#define DP_HEAD_NT 4          // lm_head: tiles of 16 vocabulary rows per unit
#endif
#ifndef DP_HEAD_DEPTH
#define DP_HEAD_DEPTH 2
#endif
// ... and of the e4m3-weight phases (a K step is 128 elements there)
#ifndef DP8_DEPTH_O
#define DP8_DEPTH_O 7
#endif
#ifndef DP8_DEPTH_DOWN
#define DP8_DEPTH_DOWN 7
#endif
#ifndef DP8_DEPTH_QKV
#define DP8_DEPTH_QKV 4
#endif
#ifndef DP8_DEPTH_GU
#define DP8_DEPTH_GU 3       // 2.264-2.272 against 2.268-2.287 ms per step with 4 (three passes, one box); 2: 2.281-2.297
#endif

template <typename PH> void launch_phase(const ImgDesc& d, int grid, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done))
        (void)hipFuncSetAttribute((const void*)img_phase_kernel<PH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH::lds_bytes());
    hipLaunchKernelGGL((img_phase_kernel<PH>), dim3((unsigned)grid), dim3(512), PH::lds_bytes(), s, d);
}
template <int RM> void launch_single8(const PhaseArgs& a, int cus, hipStream_t s) {      // e4m3 weights (W8A16)
    switch (a.phases) {
        case AFHIP_PH_O: launch_phase<ImgPhase<1, false, false, RM, DP8_DEPTH_O, true>>(a.o, cus, s); break;
        case AFHIP_PH_GU: launch_phase<ImgPhase<2, true, true, RM, DP8_DEPTH_GU, true>>(a.gu, cus, s); break;
        case AFHIP_PH_DOWN: launch_phase<ImgPhase<1, false, false, RM, DP8_DEPTH_DOWN, true>>(a.down, cus, s); break;
        case AFHIP_PH_QKV: launch_phase<ImgPhase<2, false, true, RM, DP8_DEPTH_QKV, true>>(a.qkv, cus, s); break;
        default: launch_phase<ImgPhase<4, false, true, RM, 2, true>>(a.head, cus, s); break;
    }
}
template <int RM> void launch_single(const PhaseArgs& a, int cus, hipStream_t s) {
    switch (a.phases) {
        case AFHIP_PH_EMBED: hipLaunchKernelGGL(embed_kernel<RM>, dim3((unsigned)a.B), dim3(512), 0, s, a); break;
        case AFHIP_PH_O: launch_phase<ImgPhase<1, false, false, RM, DP_DEPTH_O>>(a.o, cus, s); break;
        case AFHIP_PH_GU: launch_phase<ImgPhase<2, true, true, RM, DP_DEPTH_GU>>(a.gu, cus, s); break;
        case AFHIP_PH_DOWN: launch_phase<ImgPhase<1, false, false, RM, DP_DEPTH_DOWN>>(a.down, cus, s); break;
        case AFHIP_PH_QKV: launch_phase<ImgPhase<2, false, true, RM, DP_DEPTH_QKV>>(a.qkv, cus, s); break;
        case AFHIP_PH_HEAD: launch_phase<ImgPhase<DP_HEAD_NT, false, true, RM, DP_HEAD_DEPTH>>(a.head, cus, s); break;
        default: hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(512), 0, s, a.pick); break;
    }
}

void fill(ImgDesc& p, const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, long long ldc, long long ldres,
          int tile_rows, const float* ss_in, int ss_n, float eps) {
    p = ImgDesc{};
    p.A = (const char*)A; p.W = (const char*)W; p.bias = (const char*)bias; p.res = (const char*)res; p.C = (char*)C;
    p.M = M; p.N = N; p.K = K; p.ldw = K; p.ldc = ldc; p.ldres = ldres; p.tile_rows = tile_rows;
    p.ss_in = ss_in; p.ss_n = ss_n; p.eps = eps;
}

}  // namespace

// ---- host side (C++ linkage: called by llm.hip) ----------------------------------------------------------------------------
bool afhip_decode_phases_supported(const afhip_llm_weights* w, int B) {
    if (afhip_opt(AFHIP_OPT_DECODE_IMAGED) == 0) return false;         // A/B switch: 0 = the round-3 launches (row-major activations)
    const int cus = afhip_cu_count();
    const int H = w->hidden, I = w->inter, qw = (w->n_q + 2 * w->n_kv) * w->hd, AO = w->n_q * w->hd;
    if (w->dtype != AFHIP_BF16 || B < 1 || B > 16 || w->n_stream < 1 || cus > 1024) return false;
    if (H % 512 != 0 || I % 512 != 0 || AO % 512 != 0) return false;
    if (cdiv(H, cus) > 16 || cdiv(qw, cus) > 32 || (2 * I) % 64 != 0 || 2 * I < 8192) return false;
    if ((size_t)16 * 2 * I * 2 >= (1ull << 31) || (size_t)B * qw * 2 >= (1ull << 31)) return false;
    return true;
}

// argmax partials 2 x [16][1024] | x sums of squares [16][1024] f32 | x image | attention image | SwiGLU image
static size_t off_ss() { return (size_t)2 * 16 * 1024 * 4; }
static size_t off_ximg() { return off_ss() + (size_t)16 * 1024 * 4; }
size_t afhip_decode_phases_scratch_bytes(const afhip_llm_weights* w, int B) {
    const size_t rm = B <= 8 ? 8 : 16;
    return off_ximg() + rm * 2 * ((size_t)w->hidden + (size_t)w->n_q * w->hd + (size_t)w->inter) + 1024;
}

int afhip_decode_phase_launch(const afhip_phase_step& c, hipStream_t s) {
    const afhip_llm_weights* w = c.w;
    const int cus = afhip_cu_count();
    const int H = w->hidden, I = w->inter, qw = (w->n_q + 2 * w->n_kv) * w->hd, AO = w->n_q * w->hd, B = c.B;
    const int rm = B <= 8 ? 8 : 16;
    char* sc = (char*)c.scratch;
    PhaseArgs a = {};
    float* am_val = (float*)sc;
    int* am_idx = (int*)(am_val + 16 * 1024);
    float* xss = (float*)(sc + off_ss());
    char* ximg = sc + off_ximg();
    char* attimg = ximg + (size_t)rm * H * 2;
    char* actimg = attimg + (size_t)rm * AO * 2;
    a.phases = c.phase;
    a.B = B;
    const int rpw_h = cdiv(H, cus);                              // one equal share of output rows per CU (gemm_skinny.hip)
    const int rpw_q = cdiv(qw, cus);
    const bool f8 = w->qkv_w8 != nullptr;                        // W8A16 copies of the streamed weights (afhip.h)
    const int l = c.layer;
    switch (c.phase) {
        case AFHIP_PH_EMBED:
            a.prev_token = c.st->prev_token; a.table = (const char*)w->embed; a.x = c.x; a.ximg = ximg; a.gain = (const char*)w->ln1_w[0]; a.ss = xss; a.ss_n = cus;
            a.S = w->n_stream; a.H = H; a.vocab = w->vocab;
            break;
        case AFHIP_PH_QKV:
            fill(a.qkv, ximg, f8 ? w->qkv_w8[l] : w->qkv_w[l], w->qkv_b[l], nullptr, c.qkv, B, qw, H, qw, 0, cdiv(rpw_q, 2), xss, cus, w->rms_eps);
            if (f8) a.qkv.w_scale = w->qkv_s[l];
            break;
        case AFHIP_PH_O:
            // attention image -> x (+ residual), x image with the post-attention norm's gain, its sums of squares
            fill(a.o, attimg, f8 ? w->o_w8[l] : w->o_w[l], nullptr, c.x, c.x, B, H, AO, H, H, rpw_h, nullptr, 0, 0.f);
            if (f8) a.o.w_scale = w->o_s[l];
            a.o.img_out = ximg; a.o.img_gain = (const char*)w->ln2_w[l]; a.o.ss_out = xss;
            break;
        case AFHIP_PH_GU: {
            // SwiGLU pairs: the TR <= 16 gate rows per unit whose ceil(units / CUs) * TR is smallest (gemm_skinny.hip)
            int best = 16, best_cost = cdiv(cdiv(I, 16), cus) * 16;
            for (int tr = 15; tr >= 12; --tr) {
                const int cost = cdiv(cdiv(I, tr), cus) * tr;
                if (cost < best_cost) { best_cost = cost; best = tr; }
            }
            fill(a.gu, ximg, f8 ? w->gu_w8[l] : w->gu_w[l], nullptr, nullptr, nullptr, B, 2 * I, H, I, 0, best, xss, cus, w->rms_eps);
            if (f8) a.gu.w_scale = w->gu_s[l];
            a.gu.img_out = actimg;
            break;
        }
        case AFHIP_PH_DOWN:
            // SwiGLU image -> x (+ residual), x image with the NEXT norm's gain (next layer's input norm, or the final norm)
            fill(a.down, actimg, f8 ? w->down_w8[l] : w->down_w[l], nullptr, c.x, c.x, B, H, I, H, H, rpw_h, nullptr, 0, 0.f);
            if (f8) a.down.w_scale = w->down_s[l];
            a.down.img_out = ximg; a.down.img_gain = (const char*)(l + 1 < w->n_layers ? w->ln1_w[l + 1] : w->norm_w); a.down.ss_out = xss;
            break;
        case AFHIP_PH_HEAD: {
            const int rows = c.st->head_rows > 0 && c.st->head_rows < w->vocab ? c.st->head_rows : w->vocab;
            const bool h8 = f8 && w->lm_head8 != nullptr;
            fill(a.head, ximg, h8 ? w->lm_head8 : w->lm_head, nullptr, nullptr, nullptr, B, rows, H, 0, 0, 16, xss, cus, w->rms_eps);
            if (h8) a.head.w_scale = w->lm_head_s;
            a.head.am_iv = c.st->allowed; a.head.am_n_iv = c.st->n_iv; a.head.am_val = am_val; a.head.am_idx = am_idx;
            break;
        }
        case AFHIP_PH_PICK:
            a.pick.iv = c.st->allowed; a.pick.n_iv = c.st->n_iv;
            a.pick.prev_token = c.st->prev_token; a.pick.out_tokens = c.st->out_tokens; a.pick.finished_at = c.st->finished_at;
            a.pick.B = B; a.pick.step = c.step; a.pick.eos = c.st->eos_id; a.pick.eot = c.st->eot_id;
            a.pick.seq_pos = c.st->seq_pos; a.pick.step_counter = c.st->step_counter;
            a.pick.am_val = am_val; a.pick.am_idx = am_idx; a.pick.am_n = cus;
            break;
        default:
            afhip_set_error("afhip_decode_phase_launch: bad phase %d", c.phase);
            return AFHIP_ERR_INVALID;
    }
#ifdef AFHIP_STREAM_STAMPS
    {   // diagnostic build: AFHIP_STREAM_DBGPTR = [5 phases][workgroups][8] stamps of the launches of layer AFHIP_PHASE_STAMP_LAYER
        const char* dp = getenv("AFHIP_STREAM_DBGPTR");
        const char* dl = getenv("AFHIP_PHASE_STAMP_LAYER");
        unsigned long long* d = dp ? (unsigned long long*)strtoull(dp, nullptr, 0) : nullptr;
        if (d && dl && atoi(dl) == c.layer) {
            a.o.dbg = d; a.gu.dbg = d + 1 * cus * 8; a.down.dbg = d + 2 * cus * 8; a.qkv.dbg = d + 3 * cus * 8; a.head.dbg = d + 4 * cus * 8;
        }
    }
#endif
    const bool gemm = (c.phase & (AFHIP_PH_O | AFHIP_PH_GU | AFHIP_PH_DOWN | AFHIP_PH_QKV | AFHIP_PH_HEAD)) != 0;
    const bool w8 = gemm && f8 && !(c.phase == AFHIP_PH_HEAD && w->lm_head8 == nullptr);
    if (w8) { if (rm == 8) launch_single8<8>(a, cus, s); else launch_single8<16>(a, cus, s); }
    else if (rm == 8) launch_single<8>(a, cus, s);
    else launch_single<16>(a, cus, s);
    AFHIP_LAUNCH_CHECK();
    return 0;
}

// where the attention merge of the decode step leaves its output: the image the o phase reads (llm.hip)
void* afhip_decode_phases_att_image(const afhip_llm_weights* w, int B, void* scratch) {
    const int rm = B <= 8 ? 8 : 16;
    return (char*)scratch + off_ximg() + (size_t)rm * w->hidden * 2;
}
