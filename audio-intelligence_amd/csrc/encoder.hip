// AF-Whisper encoder forward: host-side sequencing of the HIP kernels on one stream (no sync, no allocation).
// Mirrors AFWhisperEncoder.forward (modeling_whisper.py:640-756) + Qwen2AudioEncoderLayer.forward (:471-519);
// the additive [B,1,1500,1500] mask of audio.py:1147-1161 / sound_encoder.py:88-103 is the feat_len vector.
//   stem : conv1 (implicit GEMM, +bias, GELU) -> conv2 (stride 2, +bias, GELU, + position table)
//   layer: LN -> fused QKV GEMM (k bias = 0) -> flash attention -> out-proj GEMM (+bias, +residual in place)
//          LN -> fc1 GEMM (+bias, GELU) -> fc2 GEMM (+bias, +residual in place)
//          (bf16 throughput mode: both LNs folded into the GEMMs around them, see `fold` below)
//   tail : AvgPool1d(2,2) + LayerNorm fused
#include "common.h"
#include <stdlib.h>

bool gemm_pp_available();   // gemm_pp.hip

namespace {

struct EncWs {
    char* h;     // [B*Tp, d]
    char* big;   // max(B*2Tp*d, B*Tp*ffn)
    char* ln;    // [B*Tp, d]
    char* qkv;   // [B*Tp, 3d]
    char* att;   // [B*Tp, d]
    float* part; // [d/64][B*Tp][2] row (sum, sum of squares) partials of the stored residual stream (LayerNorm-folded mode)
    float* stats;// [B*Tp][2] row (mean, rstd)
    int32_t* roff; // [B + 1] first packed row of each clip (ragged forward)
    size_t total;
};

size_t align256(size_t x) { return (x + 255) / 256 * 256; }

EncWs carve(const afhip_encoder_weights* w, int B, char* base) {
    const size_t sz = dtype_size(w->dtype);
    const size_t Tp = w->max_pos, d = w->d_model, f = w->ffn_dim;
    const size_t rows = (size_t)B * Tp;
    EncWs ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    ws.h = take(rows * d * sz);
    const size_t big_elems = (2 * rows * d > rows * f) ? 2 * rows * d : rows * f;
    ws.big = take(big_elems * sz);
    ws.ln = take(rows * d * sz);
    ws.qkv = take(rows * 3 * d * sz);
    ws.att = take(rows * d * sz);
    ws.part = (float*)take((d / 64 + 1) * rows * 2 * sizeof(float));
    ws.stats = (float*)take(rows * 4 * sizeof(float));   // [rows][2] (mean, rstd); the e4m3 mode keeps its [rows] activation scales in the first half and the statistics behind them
    ws.roff = (int32_t*)take(((size_t)B + 1) * sizeof(int32_t));
    ws.total = off;
    return ws;
}

int gemm(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda, int ldc,
         int ldres, int dtype, int act, int row_mod, hipStream_t s, int cTin = 0, int cTout = 0, int cStride = 0, int cC = 0,
         const float* ln_stats = nullptr, const float* ln_colsum = nullptr, const float* ln_bias = nullptr, float* row_stats_out = nullptr) {
    afhip_gemm_args g = {};
    g.A = A; g.W = W; g.bias = bias; g.residual = res; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldw = K; g.ldc = ldc; g.ldres = ldres;
    g.dtype = dtype; g.act = act; g.res_row_mod = row_mod;
    g.conv_Tin = cTin; g.conv_Tout = cTout; g.conv_stride = cStride; g.conv_C = cC;
    g.out_f32 = 0;
    g.a_norm_w = nullptr; g.a_norm_eps = 0.f; g.a_swiglu = 0; g.w_scale = nullptr;
    g.ln_stats = ln_stats; g.ln_colsum = ln_colsum; g.ln_bias = ln_bias; g.row_stats_out = row_stats_out;
    return afhip_gemm(&g, s);
}

// e4m3 x e4m3 projection: A8 [M,K] bytes + row scales, W8 [N,K] bytes + channel scales -> bf16 C (+bias, act, +residual)
// a_scale == NULL: every row carries a_const (statically quantised A); out_inv > 0: C leaves as e4m3 bytes = sat(value * out_inv)
int gemm8(const void* A8, const float* a_scale, const void* W8, const float* w_scale, const void* bias, const void* res, void* C,
          int M, int N, int K, int ldc, int ldres, int act, hipStream_t s, float a_const = 0.f, float out_inv = 0.f, float* row_stats_out = nullptr) {
    afhip_gemm_args g = {};
    g.A = A8; g.W = W8; g.bias = bias; g.residual = res; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.lda = K; g.ldw = K; g.ldc = ldc; g.ldres = ldres;
    g.dtype = AFHIP_BF16; g.act = act;
    g.a_fp8 = 1; g.a_scale = a_scale; g.w_scale = w_scale;
    g.a_scale_const = a_const;
    g.out_fp8 = out_inv > 0.f ? 1 : 0; g.out_scale_inv = out_inv;
    g.row_stats_out = row_stats_out;
    return afhip_gemm(&g, s);
}

}  // namespace

extern "C" size_t afhip_encoder_workspace_bytes(const afhip_encoder_weights* w, int B) {
    if (!w || B <= 0) return 0;
    return carve(w, B, nullptr).total;
}

namespace {
// len_host != nullptr: packed (ragged) forward, see afhip.h afhip_encoder_forward_ragged
int encoder_forward_impl(const afhip_encoder_weights* w, const void* mel_btc, const int32_t* feat_len, const int32_t* len_host, int B,
                         void* out, void* hidden_out, int hidden_layer, void* workspace, size_t workspace_bytes,
                         void* stream) {
    AFHIP_CHECK(w && mel_btc && out && workspace, "afhip_encoder_forward: null pointer");
    AFHIP_CHECK(B > 0, "afhip_encoder_forward: bad batch %d", B);
    AFHIP_CHECK(w->dtype == AFHIP_F32 || w->dtype == AFHIP_BF16, "afhip_encoder_forward: bad dtype %d", w->dtype);
    AFHIP_CHECK(w->d_model > 0 && w->n_heads > 0 && w->d_model % w->n_heads == 0, "afhip_encoder_forward: bad d_model/heads %d/%d", w->d_model, w->n_heads);
    const int hd = w->d_model / w->n_heads;
    AFHIP_CHECK(hd == 64 || hd == 128, "afhip_encoder_forward: head_dim %d unsupported", hd);
    AFHIP_CHECK(w->max_pos > 0 && w->max_pos % 2 == 0 && w->n_layers >= 0 && w->n_mels > 0, "afhip_encoder_forward: bad config");
    AFHIP_CHECK(hidden_layer >= -1 && hidden_layer < w->n_layers, "afhip_encoder_forward: hidden_layer %d out of range", hidden_layer);
    const EncWs ws = carve(w, B, (char*)workspace);
    if (workspace_bytes < ws.total) {
        afhip_set_error("afhip_encoder_forward: workspace %zu < required %zu bytes", workspace_bytes, ws.total);
        return AFHIP_ERR_WORKSPACE;
    }
    AFHIP_CHECK(((uintptr_t)workspace % 256) == 0, "afhip_encoder_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int dt = w->dtype;
    const size_t sz = dtype_size(dt);
    const int d = w->d_model, f = w->ffn_dim, Tp = w->max_pos, Tm = 2 * w->max_pos, nm = w->n_mels;
    int rows = B * Tp;
    char* h = ws.h;
    char* att = ws.att;
    int rc;

    // conv stem (modeling_whisper.py:690-696)
    if ((rc = gemm(mel_btc, w->conv1_w, w->conv1_b, nullptr, ws.big, B * Tm, d, 3 * nm, nm, d, 0, dt, AFHIP_ACT_GELU, 0, s, Tm, Tm, 1, nm))) return rc;
    if ((rc = gemm(ws.big, w->conv2_w, w->conv2_b, w->pos_emb, ws.h, rows, d, 3 * d, d, d, d, dt, AFHIP_ACT_GELU, Tp, s, Tm, Tp, 2, d))) return rc;
    if (hidden_out && hidden_layer == -1) {
        if (hipMemcpyAsync(hidden_out, ws.h, (size_t)rows * d * sz, hipMemcpyDeviceToDevice, s) != hipSuccess) { afhip_set_error("encoder: hidden copy failed"); return AFHIP_ERR_LAUNCH; }
    }

    const bool ragged = len_host != nullptr;
    if (ragged) {
        // keep the valid positions only: every layer below runs on M = sum of lengths rows.  The stem ran on the full padded mel, so
        // the last valid position saw the frames behind it exactly as in the unpacked forward.
        long long tot = 0;
        for (int b = 0; b < B; ++b) {
            AFHIP_CHECK(len_host[b] >= 0 && len_host[b] <= Tp, "afhip_encoder_forward_ragged: feat_len[%d] = %d outside [0, %d]", b, len_host[b], Tp);
            tot += len_host[b];
        }
        if ((rc = afhip_ragged_row_offsets(feat_len, ws.roff, B, s))) return rc;
        if (tot == 0) {
            if (hipMemsetAsync(out, 0, (size_t)B * (Tp / 2) * d * sz, s) != hipSuccess) { afhip_set_error("encoder: memset failed"); return AFHIP_ERR_LAUNCH; }
            return 0;
        }
        if ((rc = afhip_ragged_pack_rows(ws.h, ws.att, ws.roff, feat_len, B, Tp, (int)(d * sz), s))) return rc;
        h = ws.att; att = ws.h;          // the packed stream lives in the (equally sized) attention buffer from here on
        rows = (int)tot;
    }

    // LayerNorm-folded mode (bf16, ping-pong-GEMM shapes): the two LayerNorms of a layer never write a normalised copy of the
    // stream.  out-proj / fc2 emit row (sum, sum of squares) partials of what they store, a 3-us finalize turns them into
    // (mean, rstd), and q/k/v / fc1 run on the RAW stream with gamma folded into their weights (afhip.h).
    const bool fold = dt == AFHIP_BF16 && w->qkv_wf && w->qkv_cs && w->qkv_bf && w->fc1_wf && w->fc1_cs && w->fc1_bf &&
                      B * Tp >= 512 && d % 256 == 0 && f % 256 == 0 && gemm_pp_available();      // decided on the PADDED row count: the packed
                                                                                               // forward must take the same arithmetic
    const int P = d / 64;
    // e4m3-operand mode (BASELINE config 5): explicit LayerNorm fused into the per-row quantisation pass, four fp8 GEMMs per layer
    const bool f8 = dt == AFHIP_BF16 && w->qkv_w8 && w->qkv_s8 && w->out_w8 && w->out_s8 && w->fc1_w8 && w->fc1_s8 && w->fc2_w8 && w->fc2_s8 &&
                    d % 256 == 0 && f % 256 == 0;
    for (int l = 0; l < w->n_layers; ++l) {
        afhip_attn_args a = {};
        a.q = ws.qkv; a.k = ws.qkv + (size_t)d * sz; a.v = ws.qkv + (size_t)2 * d * sz; a.out = att;
        a.key_len = feat_len;
        a.row_off = ragged ? ws.roff : nullptr;
        a.B = B; a.Tq = Tp; a.Tk = Tp; a.n_q = w->n_heads; a.n_kv = w->n_heads; a.hd = hd;
        a.ld_q = 3 * d; a.ld_kv = 3 * d; a.ld_o = d;
        a.q_batch_stride = (long long)Tp * 3 * d; a.kv_batch_stride = (long long)Tp * 3 * d; a.o_batch_stride = (long long)Tp * d;
        a.q_head_stride = hd; a.kv_head_stride = hd;
        a.o_head_stride = 0; a.key_split = 0; a.partial_ws = nullptr; a.partial_ws_bytes = 0;
        a.causal = 0; a.q_pos0 = 0; a.scale = 1.0f / sqrtf((float)hd); a.dtype = dt;
        a.q_prescaled = (fold && !f8 && w->q_prescaled) ? 1 : 0;
        if (f8) {
            float* sc = ws.stats;                           // [rows] row scales of the activation being multiplied
            // which projections take e4m3 operands: bit 0 qkv, 1 out, 2 fc1, 3 fc2 (AFHIP_FP8_MASK; the rest run the plain bf16 GEMM
            // behind an explicit LayerNorm).  Default 6 = out + fc1: q / k / v stay bf16 because the softmax amplifies e4m3 noise on
            // its logits more than anything downstream -- measured through the LLM (tests/test_gpu_config5.py, 10 clips x 32 steps):
            // mask 7 (qkv too) 284-289 of 320 tokens equal to the bf16 run, mask 6 296, mask 14 (+ fc2) 294, mask 15 285
            const int mask = afhip_opt(AFHIP_OPT_FP8_MASK) | (afhip_opt(AFHIP_OPT_FP8_FC2) ? 8 : 0);
            // q | k | v in bf16: the LayerNorm-folded GEMM over the raw stream when the folded weights exist (row statistics from the
            // conv stem / the previous layer's bf16 fc2 epilogue, exactly as in the bf16 mode; q leaves prescaled for the encoder
            // attention kernel), else an explicit LayerNorm + the plain GEMM
            const bool qkv_fold = !(mask & 1) && !(mask & 8) && fold;
            if (mask & 1) {
                if ((rc = afhip_quant_rows(h, d, w->ln1_w[l], w->ln1_b[l], 1e-5f, 1, ws.ln, sc, rows, d, s))) return rc;
                if ((rc = gemm8(ws.ln, sc, w->qkv_w8[l], w->qkv_s8[l], w->qkv_b[l], nullptr, ws.qkv, rows, 3 * d, d, 3 * d, 0, AFHIP_ACT_NONE, s))) return rc;
            } else if (qkv_fold) {
                float* st = ws.stats + (size_t)2 * rows;          // (mean, rstd) live behind the activation row scales (sc) in the stats buffer
                if (l == 0 && (rc = afhip_row_stats(h, rows, d, 1e-5f, dt, st, s))) return rc;
                if ((rc = gemm(h, w->qkv_wf[l], nullptr, nullptr, ws.qkv, rows, 3 * d, d, d, 3 * d, 0, dt, AFHIP_ACT_NONE, 0, s, 0, 0, 0, 0,
                               st, w->qkv_cs[l], w->qkv_bf[l], nullptr))) return rc;
                a.q_prescaled = w->q_prescaled ? 1 : 0;
            } else {
                if ((rc = afhip_layernorm(h, w->ln1_w[l], w->ln1_b[l], ws.ln, rows, d, 1e-5f, dt, s))) return rc;
                if ((rc = gemm(ws.ln, w->qkv_w[l], w->qkv_b[l], nullptr, ws.qkv, rows, 3 * d, d, d, 3 * d, 0, dt, AFHIP_ACT_NONE, 0, s))) return rc;
            }
            // attention output statically quantised (afhip_encoder_weights.att_out_scale): the attention kernel writes e4m3 bytes into `att`
            // and the out-projection reads them with one scale for every row -- no bf16 round trip, no quantisation launch.  Needs the
            // encoder attention form, i.e. the prescaled q of the LayerNorm-folded q | k | v GEMM
            const bool att_static = (mask & 2) && a.q_prescaled && hd == 64 && w->att_out_scale && w->att_out_scale[l] > 0.f;
            if (att_static) { a.out_fp8 = 1; a.out_scale_inv = 1.0f / w->att_out_scale[l]; }
            if ((rc = afhip_attention(&a, s))) return rc;
            if (w->calib_amax && !att_static && (rc = afhip_absmax_bf16(att, (long long)rows * d, w->calib_amax + w->n_layers + l, s))) return rc;
            if (att_static) {
                if ((rc = gemm8(att, nullptr, w->out_w8[l], w->out_s8[l], w->out_b[l], h, h, rows, d, d, d, d, AFHIP_ACT_NONE, s, w->att_out_scale[l]))) return rc;
            } else if (mask & 2) {
                if ((rc = afhip_quant_rows(att, d, nullptr, nullptr, 0.f, 0, ws.ln, sc, rows, d, s))) return rc;
                if ((rc = gemm8(ws.ln, sc, w->out_w8[l], w->out_s8[l], w->out_b[l], h, h, rows, d, d, d, d, AFHIP_ACT_NONE, s))) return rc;
            } else {
                if ((rc = gemm(att, w->out_w[l], w->out_b[l], h, h, rows, d, d, d, d, d, dt, AFHIP_ACT_NONE, 0, s))) return rc;
            }
            // fc2's input statically quantised (afhip_encoder_weights.fc2_in_scale): fc1's epilogue writes the GELU output as e4m3 bytes into
            // ws.big and fc2 reads them with one scale for every row -- no [rows, ffn] bf16 round trip and no quantisation launch
            const bool fc2_static = (mask & 4) && !(mask & 8) && w->fc2_in_scale && w->fc2_in_scale[l] > 0.f;
            if (mask & 4) {
                if ((rc = afhip_quant_rows(h, d, w->ln2_w[l], w->ln2_b[l], 1e-5f, 1, ws.ln, sc, rows, d, s))) return rc;
                if (fc2_static) {
                    if ((rc = gemm8(ws.ln, sc, w->fc1_w8[l], w->fc1_s8[l], w->fc1_b[l], nullptr, ws.big, rows, f, d, f, 0, AFHIP_ACT_GELU, s, 0.f, 1.0f / w->fc2_in_scale[l]))) return rc;
                } else {
                    if ((rc = gemm8(ws.ln, sc, w->fc1_w8[l], w->fc1_s8[l], w->fc1_b[l], nullptr, ws.big, rows, f, d, f, 0, AFHIP_ACT_GELU, s))) return rc;
                }
            } else {
                if ((rc = afhip_layernorm(h, w->ln2_w[l], w->ln2_b[l], ws.ln, rows, d, 1e-5f, dt, s))) return rc;
                if ((rc = gemm(ws.ln, w->fc1_w[l], w->fc1_b[l], nullptr, ws.big, rows, f, d, d, f, 0, dt, AFHIP_ACT_GELU, 0, s))) return rc;
            }
            // calibration run: max |GELU output| of this layer (only meaningful while that buffer is bf16)
            if (w->calib_amax && !fc2_static && (rc = afhip_absmax_bf16(ws.big, (long long)rows * f, w->calib_amax + l, s))) return rc;
            // fc2's input is the [rows, ffn] GELU output: a per-row quantisation pass over it (492 MB read + 246 MB written, 109 us
            // at B = 32) costs more than the e4m3 GEMM saves (273 -> 206 us) as long as that pass is a launch of its own
            if (fc2_static) {
                // ... with the row-statistics epilogue the next layer's LayerNorm-folded q | k | v reads (as the bf16 fc2 has)
                const bool want_stats = qkv_fold && l + 1 < w->n_layers;
                if ((rc = gemm8(ws.big, nullptr, w->fc2_w8[l], w->fc2_s8[l], w->fc2_b[l], h, h, rows, d, f, d, d, AFHIP_ACT_NONE, s, w->fc2_in_scale[l], 0.f,
                                want_stats ? ws.part : nullptr))) return rc;
                if (want_stats && (rc = afhip_ln_stats_finalize(ws.part, P, rows, d, 1e-5f, ws.stats + (size_t)2 * rows, s))) return rc;
            } else
            if (mask & 8) {
                if ((rc = afhip_quant_rows(ws.big, f, nullptr, nullptr, 0.f, 0, ws.qkv, sc, rows, f, s))) return rc;     // [rows, f] bytes fit the idle qkv buffer (3 d x 2 B)
                if ((rc = gemm8(ws.qkv, sc, w->fc2_w8[l], w->fc2_s8[l], w->fc2_b[l], h, h, rows, d, f, d, d, AFHIP_ACT_NONE, s))) return rc;
            } else if (qkv_fold) {
                // bf16 fc2 with the row-statistics epilogue: the next layer's LayerNorm-folded q | k | v reads them
                const bool last = l + 1 == w->n_layers;
                if ((rc = gemm(ws.big, w->fc2_w[l], w->fc2_b[l], h, h, rows, d, f, f, d, d, dt, AFHIP_ACT_NONE, 0, s, 0, 0, 0, 0,
                               nullptr, nullptr, nullptr, last ? nullptr : ws.part))) return rc;
                if (!last && (rc = afhip_ln_stats_finalize(ws.part, P, rows, d, 1e-5f, ws.stats + (size_t)2 * rows, s))) return rc;
            } else {
                if ((rc = gemm(ws.big, w->fc2_w[l], w->fc2_b[l], h, h, rows, d, f, f, d, d, dt, AFHIP_ACT_NONE, 0, s))) return rc;
            }
        } else if (fold) {
            if (l == 0 && (rc = afhip_row_stats(h, rows, d, 1e-5f, dt, ws.stats, s))) return rc;   // layer 0 reads the conv stem
            if ((rc = gemm(h, w->qkv_wf[l], nullptr, nullptr, ws.qkv, rows, 3 * d, d, d, 3 * d, 0, dt, AFHIP_ACT_NONE, 0, s, 0, 0, 0, 0,
                           ws.stats, w->qkv_cs[l], w->qkv_bf[l], nullptr))) return rc;
            if ((rc = afhip_attention(&a, s))) return rc;
            if ((rc = gemm(att, w->out_w[l], w->out_b[l], h, h, rows, d, d, d, d, d, dt, AFHIP_ACT_NONE, 0, s, 0, 0, 0, 0,
                           nullptr, nullptr, nullptr, ws.part))) return rc;
            if ((rc = afhip_ln_stats_finalize(ws.part, P, rows, d, 1e-5f, ws.stats, s))) return rc;
            if ((rc = gemm(h, w->fc1_wf[l], nullptr, nullptr, ws.big, rows, f, d, d, f, 0, dt, AFHIP_ACT_GELU, 0, s, 0, 0, 0, 0,
                           ws.stats, w->fc1_cs[l], w->fc1_bf[l], nullptr))) return rc;
            const bool last = l + 1 == w->n_layers;
            if ((rc = gemm(ws.big, w->fc2_w[l], w->fc2_b[l], h, h, rows, d, f, f, d, d, dt, AFHIP_ACT_NONE, 0, s, 0, 0, 0, 0,
                           nullptr, nullptr, nullptr, last ? nullptr : ws.part))) return rc;
            if (!last && (rc = afhip_ln_stats_finalize(ws.part, P, rows, d, 1e-5f, ws.stats, s))) return rc;
        } else {
            if ((rc = afhip_layernorm(h, w->ln1_w[l], w->ln1_b[l], ws.ln, rows, d, 1e-5f, dt, s))) return rc;
            if ((rc = gemm(ws.ln, w->qkv_w[l], w->qkv_b[l], nullptr, ws.qkv, rows, 3 * d, d, d, 3 * d, 0, dt, AFHIP_ACT_NONE, 0, s))) return rc;
            if ((rc = afhip_attention(&a, s))) return rc;
            if ((rc = gemm(att, w->out_w[l], w->out_b[l], h, h, rows, d, d, d, d, d, dt, AFHIP_ACT_NONE, 0, s))) return rc;
            if ((rc = afhip_layernorm(h, w->ln2_w[l], w->ln2_b[l], ws.ln, rows, d, 1e-5f, dt, s))) return rc;
            if ((rc = gemm(ws.ln, w->fc1_w[l], w->fc1_b[l], nullptr, ws.big, rows, f, d, d, f, 0, dt, AFHIP_ACT_GELU, 0, s))) return rc;
            if ((rc = gemm(ws.big, w->fc2_w[l], w->fc2_b[l], h, h, rows, d, f, f, d, d, dt, AFHIP_ACT_NONE, 0, s))) return rc;
        }
        if (hidden_out && hidden_layer == l) {
            if (hipMemcpyAsync(hidden_out, h, (size_t)rows * d * sz, hipMemcpyDeviceToDevice, s) != hipSuccess) { afhip_set_error("encoder: hidden copy failed"); return AFHIP_ERR_LAUNCH; }
        }
    }
    if (ragged) return afhip_ragged_avgpool_ln(h, ws.roff, feat_len, w->lnf_w, w->lnf_b, out, B, Tp / 2, d, 1e-5f, dt, s);
    return afhip_avgpool_ln(h, w->lnf_w, w->lnf_b, out, B, Tp / 2, d, 1e-5f, dt, s);
}
}  // namespace

extern "C" int afhip_encoder_forward(const afhip_encoder_weights* w, const void* mel_btc, const int32_t* feat_len, int B,
                                     void* out, void* hidden_out, int hidden_layer, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return encoder_forward_impl(w, mel_btc, feat_len, nullptr, B, out, hidden_out, hidden_layer, workspace, workspace_bytes, stream);
}

extern "C" int afhip_encoder_forward_ragged(const afhip_encoder_weights* w, const void* mel_btc, const int32_t* feat_len, const int32_t* feat_len_host,
                                            int B, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    AFHIP_CHECK(feat_len && feat_len_host, "afhip_encoder_forward_ragged: feat_len is needed on the device and on the host");
    return encoder_forward_impl(w, mel_btc, feat_len, feat_len_host, B, out, nullptr, -1, workspace, workspace_bytes, stream);
}
