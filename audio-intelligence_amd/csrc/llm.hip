// Qwen2 decoder stack + UALM head on gfx950: host-side sequencing (one stream, no sync, no allocation) and the
// small head kernels.  Mirrors ParallelLLM._step (lm/parallel.py:570-597) over transformers Qwen2Model
// (modeling_qwen2.py:258-299), with a preallocated [layer][B][kv_head][cap][hd] KV cache instead of DynamicCache,
// and the greedy pick + stop bookkeeping of lm/parallel.py:494-513,599-601 kept on the device.
#include "common.h"
#include "decode_phases.h"

namespace {

size_t align256(size_t x) { return (x + 255) / 256 * 256; }
// merge of the decode attention's key-range partials: by the last-arriving workgroup inside the attention launch (1) or by a second
// launch (0, the default: the in-launch form measured slower).  Option DECODE_MERGE (common.h) selects it.
#define DECODE_IN_LAUNCH_MERGE (afhip_opt(AFHIP_OPT_DECODE_MERGE) == 1)
// keys per workgroup of the split-context decode attention (a multiple of 64).  DECODE_KEY_SPLIT_MIN sizes the partial workspace; the run-time
// choice is option DECODE_KEY_SPLIT (0 = DECODE_KEY_SPLIT_DEFAULT).
#define DECODE_KEY_SPLIT_MIN 64
#ifndef DECODE_KEY_SPLIT_DEFAULT
#define DECODE_KEY_SPLIT_DEFAULT 128
#endif
static int decode_key_split() {
    const int v = afhip_opt(AFHIP_OPT_DECODE_KEY_SPLIT);
    return (v >= DECODE_KEY_SPLIT_MIN && v % DECODE_KEY_SPLIT_MIN == 0) ? v : DECODE_KEY_SPLIT_DEFAULT;
}

struct LlmWs {
    char* x;     // [rows, H] residual stream
    char* nb;    // [rows, H]
    char* qkv;   // [rows, (nq + 2 nkv) hd]
    char* att;   // [rows, nq hd]
    char* act;   // [rows, 2*inter] (skinny path) / [rows, inter]
    char* act2;  // [rows, inter]
    char* part;  // decode attention partials (T == 1)
    size_t part_bytes;
    int* ticket; // [n_layers][rows * n_kv] arrival counters of the in-launch merge, zeroed once per decode step
    size_t ticket_bytes;
    size_t total;
};

LlmWs carve(const afhip_llm_weights* w, int rows, char* base, int max_ctx = 0) {
    const size_t sz = dtype_size(w->dtype);
    const size_t H = w->hidden, qw = (size_t)(w->n_q + 2 * w->n_kv) * w->hd;
    LlmWs ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    ws.x = take((size_t)rows * H * sz);
    ws.nb = take((size_t)rows * H * sz);
    ws.qkv = take((size_t)rows * qw * sz);
    ws.att = take((size_t)rows * w->n_q * w->hd * sz);
    ws.act = take((size_t)rows * 2 * w->inter * sz);
    ws.act2 = take((size_t)rows * w->inter * sz);
    // decode partials: ceil(ctx / DECODE_KEY_SPLIT_MIN) splits x rows(B) x n_kv x 32 x (hd+2) f32 (only used when T == 1)
    ws.part_bytes = max_ctx > 0 ? (size_t)((max_ctx + DECODE_KEY_SPLIT_MIN - 1) / DECODE_KEY_SPLIT_MIN) * rows * w->n_kv * 32 * (w->hd + 2) * sizeof(float) : 0;
    ws.part = take(ws.part_bytes);
    ws.ticket_bytes = max_ctx > 0 ? (size_t)w->n_layers * rows * w->n_kv * sizeof(int) : 0;
    ws.ticket = (int*)take(ws.ticket_bytes);
    ws.total = off;
    return ws;
}

// e4m3 x e4m3 projection of the prefill path (afhip_gemm_args.a_fp8)
int gemm8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* bias, const void* res, void* C,
          int M, int N, int K, int ldc, int ldres, int act, hipStream_t s) {
    afhip_gemm_args g = {};
    g.A = A8; g.W = W8; g.bias = bias; g.residual = res; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = ldc; g.ldres = ldres;
    g.dtype = AFHIP_BF16; g.act = act;
    g.a_fp8 = 1; g.a_scale = a_scale; g.w_scale = w_scale;
    return afhip_gemm(&g, s);
}

int gemm_any(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
             int ldc, int ldres, int dtype, int act, int out_f32, hipStream_t s, const void* norm_w = nullptr,
             float norm_eps = 0.f, int a_swiglu = 0, const float* w_scale = nullptr) {
    afhip_gemm_args g = {};
    g.A = A; g.W = W; g.bias = bias; g.residual = res; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldw = K; g.ldc = ldc; g.ldres = ldres;
    g.dtype = dtype; g.act = act; g.res_row_mod = 0;
    g.conv_Tin = g.conv_Tout = g.conv_stride = g.conv_C = 0;
    g.out_f32 = out_f32;
    g.a_norm_w = norm_w; g.a_norm_eps = norm_eps; g.a_swiglu = a_swiglu; g.w_scale = w_scale;
    if (M <= 64 && (act == AFHIP_ACT_NONE || (act == AFHIP_ACT_SWIGLU && N >= 8192 && M <= 32))) return afhip_gemm_skinny(&g, s);
    return afhip_gemm(&g, s);
}

// y[r, j] = silu(gu[r, 64*(j/32) + j%32]) * gu[r, 64*(j/32) + 32 + j%32]   (32-row interleaved gate/up)
template <typename T>
__global__ void swiglu_interleaved_kernel(const T* __restrict__ gu, T* __restrict__ y, int rows, int inter) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * inter) return;
    const int r = (int)(i / inter), j = (int)(i % inter);
    const T* row = gu + (long long)r * 2 * inter + 64 * (j >> 5) + (j & 31);
    y[i] = from_f32<T>(silu(to_f32<T>(row[0])) * to_f32<T>(row[32]));
}

// hs[(r*n_s + s), :] = hidden[r, :] + (s ? stream_emb[s, :] : 0)      (lm/parallel.py:588-591)
template <typename T>
__global__ void add_stream_emb_kernel(const T* __restrict__ hidden, const T* __restrict__ se, T* __restrict__ hs, int rows,
                                      int n_s, int H) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * n_s * H) return;
    const int c = (int)(i % H);
    const long long rs = i / H;
    const int s = (int)(rs % n_s);
    const long long r = rs / n_s;
    float v = to_f32<T>(hidden[r * H + c]);
    if (s > 0) v = to_f32<T>(from_f32<T>(v + to_f32<T>(se[(long long)s * H + c])));
    hs[i] = from_f32<T>(v);
}

// first-index argmax over the allowed intervals, two passes: AM_G workgroups per row scan interleaved 256-element
// slices and leave (value, index) partials in the CALLER's scratch (rows x AM_G x 8 bytes), one wave per row merges them
// (ties -> smaller index, like torch.argmax).  ROUND_BF16: every logit is rounded to bf16 before it is compared -- the
// reference takes argmax over model-dtype logits (lm/parallel.py:592-601: lm_head output in bf16), so equal bf16 values tie
// and the first index wins.
constexpr int AM_G = 64;

template <bool ROUND_BF16>
__global__ __launch_bounds__(256) void masked_argmax_part_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ iv,
                                                                 int n_iv, float* __restrict__ pval, int* __restrict__ pidx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int r = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (long long)r * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = 0; k < n_iv; ++k) {
        const int lo = iv[2 * k], hi = iv[2 * k + 1];
        for (int i = lo + g * 256 + tid; i < hi; i += AM_G * 256) {
            float v = row[i];
            if (ROUND_BF16) v = (float)(bf16)v;
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        pval[r * AM_G + g] = best;
        pidx[r * AM_G + g] = bi;
    }
}

__global__ __launch_bounds__(64) void masked_argmax_final_kernel(const int32_t* __restrict__ iv, int n_iv, const float* __restrict__ pval,
                                                                 const int* __restrict__ pidx, int64_t* __restrict__ token) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float best = pval[r * AM_G + lane];
    int bi = pidx[r * AM_G + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    // an all -inf / NaN row falls back to the first allowed id
    if (lane == 0) token[r] = (bi == 0x7fffffff) ? (n_iv > 0 ? iv[0] : 0) : bi;
}

// ids[b, 0] = prev[b]; ids[b, 1..S) = 0 (pad)      (prev_token layout of lm/parallel.py:479,540-541)
__global__ void build_ids_kernel(const int64_t* __restrict__ prev, int64_t* __restrict__ ids, int B, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * S) ids[i] = (i % S == 0) ? prev[i / S] : 0;
}

__global__ void decode_update_kernel(const int64_t* __restrict__ tok, int64_t* __restrict__ prev, int64_t* __restrict__ out_tokens,
                                     int32_t* __restrict__ finished_at, int B, int step, int eos, int eot,
                                     int32_t* __restrict__ seq_pos, int32_t* __restrict__ step_counter) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (step_counter) step = step_counter[0];          // every thread reads it before thread 0 of this (single) block bumps it
    if (b < B) {                                       // no early return: every thread of the block reaches the barrier below
        const int64_t t = tok[b];
        out_tokens[(long long)step * B + b] = t;
        prev[b] = t;
        if ((t == eos || t == eot) && finished_at[b] < 0) finished_at[b] = step;
        if (seq_pos) seq_pos[b] += 1;
    }
    __syncthreads();
    if (step_counter && b == 0) step_counter[0] = step + 1;
}

}  // namespace

extern "C" size_t afhip_llm_workspace_bytes(const afhip_llm_weights* w, int B, int T, int max_ctx) {
    if (!w || B <= 0 || T <= 0) return 0;
    // forward scratch + head scratch (n_stream rows of hidden per token row, f32 logits for one decode step, ids, token)
    const size_t rows = (size_t)B * T;
    size_t tot = carve(w, (int)rows, nullptr, T == 1 ? max_ctx : 0).total;
    tot += align256(rows * w->n_stream * w->hidden * dtype_size(w->dtype));
    tot += align256((size_t)B * w->vocab * sizeof(float));
    tot += align256((size_t)B * w->n_stream * sizeof(int64_t)) + align256((size_t)B * sizeof(int64_t));
    tot += align256((size_t)B * w->hidden * dtype_size(w->dtype)) * 2;
    tot += afhip_masked_argmax_workspace_bytes(B);
    if (T == 1) tot += align256(afhip_decode_phases_scratch_bytes(w, B));     // decode_phases.hip: argmax partials, sums of squares, activation images
    return tot;
}

// attention of layer l over the KV cache: causal GQA prefill, or (decode_attn) one new token per sequence -- the `rep` query heads that
// share a kv head are the query rows of one workgroup, RoPE + the cache append ride inside the launch, the context is split into
// key ranges over workgroups and merged (flash-decoding)
static int layer_attention(const afhip_llm_weights* w, const LlmWs& ws, int l, int B, int T, int pos0, afhip_kv_cache* cache, const int32_t* seq_pos,
                           bool decode_attn, hipStream_t s, int out_img_rows = 0) {
    const int dt = w->dtype, nq = w->n_q, nkv = w->n_kv, hd = w->hd;
    const size_t sz = dtype_size(dt);
    const int qw = (nq + 2 * nkv) * hd, rep = nq / nkv;
    const size_t layer_kv = (size_t)cache->B * nkv * cache->cap * hd * sz;
    char* kc = (char*)cache->k + (size_t)l * layer_kv;
    char* vc = (char*)cache->v + (size_t)l * layer_kv;
    afhip_attn_args a = {};
    a.q = ws.qkv; a.k = kc; a.v = vc; a.out = ws.att; a.key_len = nullptr;
    a.hd = hd; a.ld_kv = hd;
    a.kv_batch_stride = (long long)nkv * cache->cap * hd; a.kv_head_stride = (long long)cache->cap * hd;
    a.scale = 1.0f / sqrtf((float)hd); a.dtype = dt; a.q_prescaled = 0;
    a.B = B; a.Tk = pos0 + T;
    if (decode_attn) {
        // decode: the `rep` query heads that share a kv head are the query rows of one workgroup, so each K/V byte is
        // streamed once per group; the context is split into DECODE_KEY_SPLIT-key ranges over workgroups and merged (flash-decoding)
        a.Tq = rep; a.n_q = nkv; a.n_kv = nkv;
        a.ld_q = hd; a.q_head_stride = (long long)rep * hd; a.q_batch_stride = qw;
        a.ld_o = hd; a.o_head_stride = (long long)rep * hd; a.o_batch_stride = (long long)nq * hd;
        a.causal = 0; a.q_pos0 = 0;
        a.key_split = decode_key_split(); a.partial_ws = ws.part; a.partial_ws_bytes = ws.part_bytes;
        a.new_k = ws.qkv + (size_t)nq * hd * sz; a.new_v = ws.qkv + (size_t)(nq + nkv) * hd * sz; a.new_kv_batch_stride = qw;
        a.seq_pos = seq_pos;
        a.rope_cos = seq_pos ? w->rope_cos : w->rope_cos + (size_t)pos0 * (hd / 2);
        a.rope_sin = seq_pos ? w->rope_sin : w->rope_sin + (size_t)pos0 * (hd / 2);
        // the in-launch merge (a.split_ticket = ws.ticket + l * B * nkv) is correct and bit-identical but SLOWER here: 4.10 vs 3.76 ms
        // per 7B step -- 224 workgroups each paying an agent-scope release (L2 write-back) cost more than one 5-us combine launch
        a.split_ticket = (DECODE_IN_LAUNCH_MERGE && !out_img_rows) ? ws.ticket + (size_t)l * B * nkv : nullptr;
        a.out_img_rows = out_img_rows;
    } else {
        a.Tq = T; a.n_q = nq; a.n_kv = nkv;
        a.ld_q = qw; a.q_head_stride = hd; a.q_batch_stride = (long long)T * qw;
        a.ld_o = nq * hd; a.o_head_stride = 0; a.o_batch_stride = (long long)T * nq * hd;
        a.causal = 1; a.q_pos0 = pos0;
        a.key_split = 0; a.partial_ws = nullptr; a.partial_ws_bytes = 0;
    }
    return afhip_attention(&a, s);
}

static int llm_forward_impl(const afhip_llm_weights* w, const void* x, int B, int T, int pos0, afhip_kv_cache* cache,
                            void* hidden_out, void* workspace, size_t workspace_bytes, void* stream, const int32_t* seq_pos) {
    AFHIP_CHECK(w && x && cache && hidden_out && workspace, "afhip_llm_forward: null pointer");
    // seq_pos != NULL (T == 1 only): every sequence's position lives on the device; pos0 is then the largest position any of them
    // may hold during this call (it sizes the grid of the split-context attention and is range-checked here instead)
    AFHIP_CHECK(seq_pos == nullptr || (T == 1 && w->n_q / w->n_kv <= 32), "afhip_llm_forward_ragged: one token per sequence, <= 32 query heads per kv head");
    AFHIP_CHECK(B > 0 && T > 0 && pos0 >= 0, "afhip_llm_forward: bad B=%d T=%d pos0=%d", B, T, pos0);
    AFHIP_CHECK(w->dtype == AFHIP_F32 || w->dtype == AFHIP_BF16, "afhip_llm_forward: bad dtype");
    AFHIP_CHECK(w->hd == 64 || w->hd == 128, "afhip_llm_forward: head_dim %d unsupported", w->hd);
    AFHIP_CHECK(w->n_q % w->n_kv == 0 && w->inter % 32 == 0, "afhip_llm_forward: bad head / intermediate sizes");
    AFHIP_CHECK(cache->k && cache->v && cache->B >= B, "afhip_llm_forward: cache batch %d < %d", cache->B, B);
    AFHIP_CHECK(pos0 + T <= cache->cap, "afhip_llm_forward: positions [%d,%d) exceed KV capacity %d", pos0, pos0 + T, cache->cap);
    AFHIP_CHECK(pos0 + T <= w->rope_max_pos, "afhip_llm_forward: positions exceed rope table %d", w->rope_max_pos);
    const int rows = B * T;
    const LlmWs ws = carve(w, rows, (char*)workspace, T == 1 ? cache->cap : 0);
    if (workspace_bytes < ws.total) {
        afhip_set_error("afhip_llm_forward: workspace %zu < required %zu bytes", workspace_bytes, ws.total);
        return AFHIP_ERR_WORKSPACE;
    }
    AFHIP_CHECK(((uintptr_t)workspace % 256) == 0, "afhip_llm_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int dt = w->dtype;
    const size_t sz = dtype_size(dt);
    const int H = w->hidden, nq = w->n_q, nkv = w->n_kv, hd = w->hd, I = w->inter;
    const int qw = (nq + 2 * nkv) * hd;
    const size_t layer_kv = (size_t)cache->B * nkv * cache->cap * hd * sz;
    int rc;
    if (hipMemcpyAsync(ws.x, x, (size_t)rows * H * sz, hipMemcpyDeviceToDevice, s) != hipSuccess) { afhip_set_error("llm: input copy failed"); return AFHIP_ERR_LAUNCH; }
    if (DECODE_IN_LAUNCH_MERGE && T == 1 && ws.ticket_bytes > 0 && hipMemsetAsync(ws.ticket, 0, ws.ticket_bytes, s) != hipSuccess) { afhip_set_error("llm: ticket memset failed"); return AFHIP_ERR_LAUNCH; }

    for (int l = 0; l < w->n_layers; ++l) {
        char* kc = (char*)cache->k + (size_t)l * layer_kv;
        char* vc = (char*)cache->v + (size_t)l * layer_kv;
        const bool skinny = rows <= 64;       // decode: RMSNorm and SwiGLU are folded into the weight-streaming GEMMs
        const bool f8 = skinny && T == 1 && rows <= 32 && dt == AFHIP_BF16 && w->qkv_w8 != nullptr;   // W8A16 copies of the streamed weights: decode steps only, prefill keeps bf16
        // prefill on e4m3 operands (BASELINE config 5): the activation of every projection is quantised per row (RMSNorm fused into
        // that pass); q8 lives in the act2 buffer (rows x inter bytes fit its rows x inter x 2), the row scales in nb
        const bool p8 = !skinny && dt == AFHIP_BF16 && w->fp8_prefill && w->qkv_w8 && w->o_w8 && w->gu_w8 && w->down_w8 &&
                        H % 256 == 0 && I % 256 == 0 && (nq * hd) % 256 == 0 && qw % 256 == 0 && I <= 20480;
        char* q8 = ws.act2;
        float* sc8 = (float*)ws.nb;
        if (skinny) {
            if ((rc = gemm_any(ws.x, f8 ? w->qkv_w8[l] : w->qkv_w[l], w->qkv_b[l], nullptr, ws.qkv, rows, qw, H, H, qw, 0, dt, AFHIP_ACT_NONE, 0, s, w->ln1_w[l], w->rms_eps, 0, f8 ? w->qkv_s[l] : nullptr))) return rc;
        } else if (p8) {
            if ((rc = afhip_quant_rows(ws.x, H, w->ln1_w[l], nullptr, w->rms_eps, 2, q8, sc8, rows, H, s))) return rc;
            if ((rc = gemm8(q8, sc8, w->qkv_w8[l], w->qkv_s[l], w->qkv_b[l], nullptr, ws.qkv, rows, qw, H, qw, 0, AFHIP_ACT_NONE, s))) return rc;
        } else {
            if ((rc = afhip_rmsnorm(ws.x, w->ln1_w[l], ws.nb, rows, H, w->rms_eps, dt, s))) return rc;
            if ((rc = gemm_any(ws.nb, w->qkv_w[l], w->qkv_b[l], nullptr, ws.qkv, rows, qw, H, H, qw, 0, dt, AFHIP_ACT_NONE, 0, s))) return rc;
        }
        const int rep = nq / nkv;
        const bool decode_attn = T == 1 && rep <= 32;
        // decode: RoPE and the cache append ride inside the attention launch (afhip_attn_args.new_k); prefill: their own pass
        if (!decode_attn && (rc = afhip_rope_kv(ws.qkv, qw, w->rope_cos, w->rope_sin, pos0, kc, vc, B, T, nq, nkv, hd, cache->cap, w->rope_max_pos, dt, s))) return rc;
        if ((rc = layer_attention(w, ws, l, B, T, pos0, cache, seq_pos, decode_attn, s))) return rc;
        if (p8) {
            if ((rc = afhip_quant_rows(ws.att, nq * hd, nullptr, nullptr, 0.f, 0, q8, sc8, rows, nq * hd, s))) return rc;
            if ((rc = gemm8(q8, sc8, w->o_w8[l], w->o_s[l], nullptr, ws.x, ws.x, rows, H, nq * hd, H, H, AFHIP_ACT_NONE, s))) return rc;
            if ((rc = afhip_quant_rows(ws.x, H, w->ln2_w[l], nullptr, w->rms_eps, 2, q8, sc8, rows, H, s))) return rc;
            if ((rc = gemm8(q8, sc8, w->gu_w8[l], w->gu_s[l], nullptr, nullptr, ws.act, rows, 2 * I, H, I, 0, AFHIP_ACT_SWIGLU, s))) return rc;
            if ((rc = afhip_quant_rows(ws.act, I, nullptr, nullptr, 0.f, 0, q8, sc8, rows, I, s))) return rc;
            if ((rc = gemm8(q8, sc8, w->down_w8[l], w->down_s[l], nullptr, ws.x, ws.x, rows, H, I, H, H, AFHIP_ACT_NONE, s))) return rc;
            continue;
        }
        if ((rc = gemm_any(ws.att, f8 ? w->o_w8[l] : w->o_w[l], nullptr, ws.x, ws.x, rows, H, nq * hd, nq * hd, H, H, dt, AFHIP_ACT_NONE, 0, s, nullptr, 0.f, 0, f8 ? w->o_s[l] : nullptr))) return rc;
        if (skinny) {
            const char* mlp_in = ws.act;
            if (2 * I >= 8192 && rows <= 32) {
                // gate/up GEMM with RMSNorm on its A load and SwiGLU as its epilogue -> [rows, I]
                if ((rc = gemm_any(ws.x, f8 ? w->gu_w8[l] : w->gu_w[l], nullptr, nullptr, ws.act, rows, 2 * I, H, H, I, 0, dt, AFHIP_ACT_SWIGLU, 0, s, w->ln2_w[l], w->rms_eps, 0, f8 ? w->gu_s[l] : nullptr))) return rc;
            } else {
                if ((rc = gemm_any(ws.x, f8 ? w->gu_w8[l] : w->gu_w[l], nullptr, nullptr, ws.act, rows, 2 * I, H, H, 2 * I, 0, dt, AFHIP_ACT_NONE, 0, s, w->ln2_w[l], w->rms_eps, 0, f8 ? w->gu_s[l] : nullptr))) return rc;
                const long long n = (long long)rows * I;
                if (dt == AFHIP_BF16) hipLaunchKernelGGL(swiglu_interleaved_kernel<bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)ws.act, (bf16*)ws.act2, rows, I);
                else hipLaunchKernelGGL(swiglu_interleaved_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)ws.act, (float*)ws.act2, rows, I);
                AFHIP_LAUNCH_CHECK();
                mlp_in = ws.act2;
            }
            if ((rc = gemm_any(mlp_in, f8 ? w->down_w8[l] : w->down_w[l], nullptr, ws.x, ws.x, rows, H, I, I, H, H, dt, AFHIP_ACT_NONE, 0, s, nullptr, 0.f, 0, f8 ? w->down_s[l] : nullptr))) return rc;
        } else {
            if ((rc = afhip_rmsnorm(ws.x, w->ln2_w[l], ws.nb, rows, H, w->rms_eps, dt, s))) return rc;
            if ((rc = gemm_any(ws.nb, w->gu_w[l], nullptr, nullptr, ws.act, rows, 2 * I, H, H, I, 0, dt, AFHIP_ACT_SWIGLU, 0, s))) return rc;
            if ((rc = gemm_any(ws.act, w->down_w[l], nullptr, ws.x, ws.x, rows, H, I, I, H, H, dt, AFHIP_ACT_NONE, 0, s))) return rc;
        }
    }
    return afhip_rmsnorm(ws.x, w->norm_w, hidden_out, rows, H, w->rms_eps, dt, s);
}

extern "C" int afhip_llm_forward(const afhip_llm_weights* w, const void* x, int B, int T, int pos0, afhip_kv_cache* cache,
                                 void* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    return llm_forward_impl(w, x, B, T, pos0, cache, hidden_out, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int afhip_llm_forward_ragged(const afhip_llm_weights* w, const void* x, int B, const int32_t* seq_pos, int max_pos,
                                        afhip_kv_cache* cache, void* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    AFHIP_CHECK(seq_pos != nullptr && max_pos >= 0, "afhip_llm_forward_ragged: seq_pos / max_pos");
    return llm_forward_impl(w, x, B, 1, max_pos, cache, hidden_out, workspace, workspace_bytes, stream, seq_pos);
}

extern "C" int afhip_lm_head(const afhip_llm_weights* w, const void* hidden, int rows, int n_s, float* logits, void* workspace,
                             size_t workspace_bytes, void* stream) {
    AFHIP_CHECK(w && hidden && logits && workspace, "afhip_lm_head: null pointer");
    AFHIP_CHECK(rows > 0 && n_s >= 1 && n_s <= w->n_stream, "afhip_lm_head: bad rows=%d n_s=%d", rows, n_s);
    const int dt = w->dtype, H = w->hidden;
    const size_t need = align256((size_t)rows * n_s * H * dtype_size(dt));
    if (workspace_bytes < need) {
        afhip_set_error("afhip_lm_head: workspace %zu < required %zu bytes", workspace_bytes, need);
        return AFHIP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const void* a = hidden;
    if (n_s > 1) {
        AFHIP_CHECK(w->stream_emb != nullptr, "afhip_lm_head: stream_emb missing");
        const long long n = (long long)rows * n_s * H;
        if (dt == AFHIP_BF16) hipLaunchKernelGGL(add_stream_emb_kernel<bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)hidden, (const bf16*)w->stream_emb, (bf16*)workspace, rows, n_s, H);
        else hipLaunchKernelGGL(add_stream_emb_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)hidden, (const float*)w->stream_emb, (float*)workspace, rows, n_s, H);
        AFHIP_LAUNCH_CHECK();
        a = workspace;
    }
    const bool f8 = rows * n_s <= 32 && dt == AFHIP_BF16 && w->lm_head8 != nullptr;
    return gemm_any(a, f8 ? w->lm_head8 : w->lm_head, nullptr, nullptr, logits, rows * n_s, w->vocab, H, H, w->vocab, 0, dt, AFHIP_ACT_NONE, 1, s,
                    nullptr, 0.f, 0, f8 ? w->lm_head_s : nullptr);
}

extern "C" size_t afhip_masked_argmax_workspace_bytes(int rows) { return rows > 0 ? align256((size_t)rows * AM_G * 8) : 0; }

extern "C" int afhip_masked_argmax(const float* logits, int rows, int ld, const int32_t* allowed, int n_iv, int64_t* token,
                                   int logits_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    AFHIP_CHECK(logits && allowed && token && workspace && rows > 0 && n_iv > 0 && ld > 0, "afhip_masked_argmax: bad args");
    AFHIP_CHECK(logits_dtype == AFHIP_F32 || logits_dtype == AFHIP_BF16, "afhip_masked_argmax: bad logits_dtype");
    if (workspace_bytes < afhip_masked_argmax_workspace_bytes(rows)) {
        afhip_set_error("afhip_masked_argmax: workspace %zu < required %zu bytes", workspace_bytes, afhip_masked_argmax_workspace_bytes(rows));
        return AFHIP_ERR_WORKSPACE;
    }
    float* pval = (float*)workspace;
    int* pidx = (int*)((char*)workspace + (size_t)rows * AM_G * 4);
    if (logits_dtype == AFHIP_BF16)
        hipLaunchKernelGGL(masked_argmax_part_kernel<true>, dim3(AM_G, rows), dim3(256), 0, (hipStream_t)stream, logits, ld, allowed, n_iv, pval, pidx);
    else
        hipLaunchKernelGGL(masked_argmax_part_kernel<false>, dim3(AM_G, rows), dim3(256), 0, (hipStream_t)stream, logits, ld, allowed, n_iv, pval, pidx);
    hipLaunchKernelGGL(masked_argmax_final_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, allowed, n_iv, (const float*)pval, (const int*)pidx, token);
    AFHIP_LAUNCH_CHECK();
    return 0;
}

extern "C" int afhip_llm_decode_step(const afhip_llm_weights* w, afhip_kv_cache* cache, afhip_decode_state* st, int B, int pos,
                                     int step, void* workspace, size_t workspace_bytes, void* stream) {
    AFHIP_CHECK(w && cache && st && workspace, "afhip_llm_decode_step: null pointer");
    AFHIP_CHECK(st->prev_token && st->out_tokens && st->finished_at && st->allowed && st->n_iv > 0, "afhip_llm_decode_step: bad state");
    AFHIP_CHECK(B > 0 && step >= 0, "afhip_llm_decode_step: bad B/step");
    const size_t need = afhip_llm_workspace_bytes(w, B, 1, cache->cap);
    if (workspace_bytes < need) {
        afhip_set_error("afhip_llm_decode_step: workspace %zu < required %zu bytes", workspace_bytes, need);
        return AFHIP_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int dt = w->dtype, H = w->hidden, S = w->n_stream;
    const size_t sz = dtype_size(dt);
    char* base = (char*)workspace;
    const size_t fwd_bytes = carve(w, B, nullptr, cache->cap).total;
    size_t off = fwd_bytes;
    char* hs = base + off; off += align256((size_t)B * S * H * sz);
    float* logits = (float*)(base + off); off += align256((size_t)B * w->vocab * sizeof(float));
    int64_t* ids = (int64_t*)(base + off); off += align256((size_t)B * S * sizeof(int64_t));
    int64_t* tok = (int64_t*)(base + off); off += align256((size_t)B * sizeof(int64_t));
    char* emb = base + off; off += align256((size_t)B * H * sz);
    char* hid = base + off; off += align256((size_t)B * H * sz);
    char* am = base + off; off += afhip_masked_argmax_workspace_bytes(B);      // `off` = start of the decode_phases.hip scratch
    int rc;
    AFHIP_CHECK((st->seq_pos == nullptr) == (st->step_counter == nullptr), "afhip_llm_decode_step: seq_pos and step_counter go together");
    if (w->n_q / w->n_kv <= 32 && afhip_decode_phases_supported(w, B) &&
        (w->qkv_w8 == nullptr || (w->o_w8 && w->gu_w8 && w->down_w8 && w->qkv_s && w->o_s && w->gu_s && w->down_s))) {
        // bf16 model (bf16 or e4m3 weight copies), B <= 16: the GEMMs of the step are the persistent imaged phases of decode_phases.hip --
        // activations handed from producer to consumer as fragment-order images -- and the attention merge writes the image the o phase reads
        AFHIP_CHECK(pos >= 0 && pos + 1 <= cache->cap && pos + 1 <= w->rope_max_pos && cache->B >= B && cache->k && cache->v,
                    "afhip_llm_decode_step: position %d exceeds the KV capacity %d / rope table %d, or cache batch %d < %d", pos, cache->cap, w->rope_max_pos, cache->B, B);
        AFHIP_CHECK(w->hd == 64 || w->hd == 128, "afhip_llm_decode_step: head_dim %d unsupported", w->hd);
        LlmWs ws = carve(w, B, base, cache->cap);
        char* scratch = base + off;
        afhip_phase_step c = {};
        c.w = w; c.B = B; c.x = ws.x; c.qkv = ws.qkv; c.scratch = scratch; c.st = st; c.step = step;
        auto launch = [&](int phase, int layer) -> int { c.phase = phase; c.layer = layer; return afhip_decode_phase_launch(c, s); };
        ws.att = (char*)afhip_decode_phases_att_image(w, B, scratch);
        const int L = w->n_layers;
        if ((rc = launch(AFHIP_PH_EMBED, 0))) return rc;
        for (int l = 0; l < L; ++l) {
            if ((rc = launch(AFHIP_PH_QKV, l))) return rc;
            if ((rc = layer_attention(w, ws, l, B, 1, pos, cache, st->seq_pos, true, s, B <= 8 ? 8 : 16))) return rc;
            if ((rc = launch(AFHIP_PH_O, l)) || (rc = launch(AFHIP_PH_GU, l)) || (rc = launch(AFHIP_PH_DOWN, l))) return rc;
        }
        if ((rc = launch(AFHIP_PH_HEAD, L - 1)) || (rc = launch(AFHIP_PH_PICK, L - 1))) return rc;
        return 0;
    }
    hipLaunchKernelGGL(build_ids_kernel, dim3(cdiv(B * S, 256)), dim3(256), 0, s, (const int64_t*)st->prev_token, ids, B, S);
    AFHIP_LAUNCH_CHECK();
    if ((rc = afhip_embed_sum(ids, w->embed, emb, B, S, H, w->vocab, dt, s))) return rc;
    if ((rc = llm_forward_impl(w, emb, B, 1, pos, cache, hid, workspace, fwd_bytes, s, st->seq_pos))) return rc;
    if ((rc = afhip_lm_head(w, hid, B, 1, logits, hs, align256((size_t)B * S * H * sz), s))) return rc;
    if ((rc = afhip_masked_argmax(logits, B, w->vocab, st->allowed, st->n_iv, tok, dt, am, afhip_masked_argmax_workspace_bytes(B), s))) return rc;
    AFHIP_CHECK(B <= 1024, "afhip_llm_decode_step: B=%d > 1024", B);
    hipLaunchKernelGGL(decode_update_kernel, dim3(1), dim3((B + 63) / 64 * 64), 0, s, (const int64_t*)tok, st->prev_token, st->out_tokens,
                       st->finished_at, B, step, st->eos_id, st->eot_id, st->seq_pos, st->step_counter);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
