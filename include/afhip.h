/*
 * afhip.h -- C ABI of the MI355X (gfx950) library behind the AF3 / UALM audio-understanding forward pass.
 *
 * The reference (NVIDIA/audio-intelligence, UALM/) is pure Python and has no FFI; its boundary for this
 * path is the Python plugin interface (AbsIO / ContinuousAudioIO / AFWhisperEncoder / ParallelLLM).  This
 * header is the native boundary a maintainer binds *under* those classes (ctypes stub: INTEGRATION.md).
 * Each entry point names the reference code whose arithmetic it replaces (paths relative to
 * UALM/models/ualm/ unless absolute).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is stream-ordered,
 *     nothing synchronises, allocates or frees: entry points are graph-capturable;
 *   - dtype: AFHIP_F32 (parity mode, exact-f32 MFMA) or AFHIP_BF16 (throughput mode, f32 accumulate);
 *   - return 0 on success, negative AFHIP_ERR_* otherwise; afhip_last_error() gives the message.
 *     Nothing aborts the process (the reference's callers catch Exception and continue,
 *     scripts/inference.py:271-279).
 */
#ifndef AFHIP_H
#define AFHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFHIP_F32 0
#define AFHIP_BF16 1

#define AFHIP_ERR_INVALID (-1)
#define AFHIP_ERR_LAUNCH (-2)
#define AFHIP_ERR_WORKSPACE (-3)

#define AFHIP_ACT_NONE 0
#define AFHIP_ACT_GELU 1   /* erf GELU (modeling_whisper.py:690-691,502) */
#define AFHIP_ACT_SWIGLU 2 /* silu(gate)*up on 32-row interleaved gate/up weights (modeling_qwen2.py:46-48) */

int afhip_version(void);
/* Tuning / A-B switches of the library live in one table (csrc/api.hip; names = the AFHIP_<NAME> environment variables, which are read
 * ONCE when the library is loaded).  This sets one by name afterwards -- the tests compare two forms of a kernel inside one process
 * with it.  Not a production control; returns an error for an unknown name. */
int afhip_set_option(const char* name, int value);
const char* afhip_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * log-mel front end.  Replaces transformers WhisperFeatureExtractor._torch_extract_fbank_features
 * (/usr/local/lib/python3.10/dist-packages/transformers/models/whisper/feature_extraction_whisper.py:135-170)
 * as called from multimodal_io/audio.py:1056-1069 (ContinuousAudioIO.preprocess).
 *   wav      [B, n_samples] f32, n_samples <= 480000 (shorter clips are zero-padded on the fly, :1056-1057)
 *   mel_out  layout 0: [B,128,3000] (extractor layout)   layout 1: [B,3000,128] (preprocess / encode_batch layout)
 *            out_dtype AFHIP_F32 or AFHIP_BF16 (the cast scripts/inference.py:272 applies); 16-byte aligned.
 *            Written once as (log10(mel) + 4) / 4; the per-clip floor max(x, clipmax - 8) of :160-162 is then applied IN PLACE
 *            (rounding is monotone, so max-after-rounding equals the reference's round-after-max bit for bit)
 *   tables   device copy of the constant block afhip_log_mel_tables_host() builds on the host from the
 *            [201,128] f32 mel filter bank (transformers/audio_utils.py:638-729): folded-DFT cos/sin tables,
 *            periodic Hann window, filter bands
 *   workspace >= afhip_log_mel_workspace_bytes(B): the per-clip running maxima (no intermediate copy of the features)
 */
size_t afhip_log_mel_tables_bytes(void);
int afhip_log_mel_tables_host(void* host_buf, const float* filters_host);
size_t afhip_log_mel_workspace_bytes(int B);
int afhip_log_mel(const float* wav, int B, int n_samples, int wav_stride, void* mel_out, int layout, int out_dtype,
                  const float* tables, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T): the arithmetic of every nn.Linear / nn.Conv1d on the path
 * (modeling_whisper.py:132-135,467-468,614-615; lm/parallel.py:146-149; modeling_qwen2.py:35-49,195-235).
 *   epilogue: + bias[N]  -> act -> + residual[m % res_row_mod or m, N]
 *   conv_C > 0: A is X[B,conv_Tin,conv_C] and row m=(b,t) of the implicit im2col matrix is
 *               [X[b, t*stride-1], X[b, t*stride], X[b, t*stride+1]] (kernel 3, pad 1); W is [N, 3*conv_C].
 *   K % 64 == 0 (bf16) / K % 32 == 0 (f32).  ACT_SWIGLU: N counts gate+up rows, C is [M, N/2].
 */
typedef struct {
    const void* A;
    const void* W;
    const void* bias;     /* may be NULL */
    const void* residual; /* may be NULL */
    void* C;
    int M, N, K;
    int lda, ldw, ldc, ldres;
    int dtype;
    int act;
    int res_row_mod;
    int conv_Tin, conv_Tout, conv_stride, conv_C;
    int out_f32; /* store C as f32 whatever `dtype` is (logits) */
    /* afhip_gemm_skinny only: producer ops folded into the A-operand load (decode path) */
    const void* a_norm_w; /* != NULL: A' = RMSNorm(A) with this gain [K] (modeling_qwen2.py:238-252) */
    float a_norm_eps;
    int a_swiglu;         /* != 0: A is the 32-row interleaved gate/up buffer [M, 2K], A' = silu(gate)*up */
    /* afhip_gemm_skinny only: W is OCP e4m3 bytes [N, ldw] with one f32 scale per output row (W8A16 decode, dtype BF16,
     * M <= 32, K % 128 == 0): C = A' . (w_scale[n] * Wq[n,:])^T.  ACT_SWIGLU is accepted here as the epilogue of a
     * 32-row interleaved gate/up weight (N >= 8192), for both weight formats. */
    const float* w_scale; /* != NULL selects the fp8-weight kernel */
    /* LayerNorm folded around the large bf16 GEMMs of the encoder layer (modeling_whisper.py:481-519): no normalised copy of
     * the residual stream is ever written.  Consumer (q/k/v, fc1): A is the RAW stream x, W = W0 diag(gamma) and
     *   C = rstd[m] * (x W^T - mean[m] * ln_colsum[n]) + ln_bias[n]   -> act,       ln_bias = b0 + W0 beta (f32)
     * with ln_stats [M][2] = (mean, rstd) per row.  Producer (out-proj, fc2: bias + residual): row_stats_out
     * [N/64][M][2] receives per-64-column partial (sum, sum of squares) of the rows it stores; afhip_ln_stats_finalize
     * turns them into ln_stats.  Both need the ping-pong kernel's shape rules (bf16, N % 256 == 0, K % 128 == 0,
     * M >= 512); afhip_gemm fails otherwise. */
    const float* ln_stats;
    const float* ln_colsum;
    const float* ln_bias;
    float* row_stats_out;
    /* afhip_gemm only: e4m3 x e4m3 operands (BASELINE config 5 "fp8 MFMA GEMMs"; no reference counterpart).  a_fp8 != 0: A is OCP
     * e4m3 bytes [M, lda] with one f32 scale per row (a_scale [M], from afhip_quant_rows), W is e4m3 bytes [N, ldw] with one f32
     * scale per output channel (w_scale [N]); C = epilogue(a_scale[m] * w_scale[n] * (Aq . Wq^T)) in bf16 (`dtype` = AFHIP_BF16
     * names the type of C / bias / residual).  f32 accumulation on the block-scaled MFMA with unit block scales (twice the bf16
     * MFMA rate).  Needs N % 256 == 0, K % 256 == 0, lda / ldw % 16 == 0; bias / GELU / residual / SWIGLU epilogues as for bf16. */
    int a_fp8;
    const float* a_scale;
    /* a_fp8 only.  a_scale == NULL: every row of A carries the ONE scale a_scale_const (> 0) -- activations that a producer's
     * epilogue quantised statically (out_fp8 below) instead of afhip_quant_rows's per-row pass. */
    float a_scale_const;
    /* a_fp8 only, act NONE or GELU, no residual / row statistics: out_fp8 != 0 writes C as OCP e4m3 BYTES [M, ldc] (ldc in bytes,
     * % 16 == 0) instead of bf16: each value is multiplied by out_scale_inv (= 1 / the scale its consumer will pass as
     * a_scale_const), clamped to +-448 and rounded to nearest even.  This is the fused form of "GEMM -> bf16 -> afhip_quant_rows". */
    int out_fp8;
    float out_scale_inv;
} afhip_gemm_args;
int afhip_gemm(const afhip_gemm_args* args, void* stream);

/* Measurement hook for bench.py's roofline leg: while enabled, every afhip_gemm launch is bracketed by HIP events on
 * its own stream; collect() waits for them and returns the launch count, summed milliseconds and summed
 * algorithmic FLOPs (2*M*N*K) of the launches of `dtype`, then disables recording. Not for production paths.
 * Launches served by the persistent ping-pong kernel (gemm_pp.hip) are recorded as `dtype | AFHIP_PROF_PINGPONG`, so the
 * roofline leg can quote that kernel alone (its rocprofv3 rows) and the remaining launches separately. */
#define AFHIP_PROF_PINGPONG 0x100
#define AFHIP_PROF_FP8 0x200       /* e4m3-operand launches are recorded as AFHIP_PROF_FP8 | AFHIP_PROF_PINGPONG */
#define AFHIP_PROF_ATTN 0x400      /* launches of the encoder attention kernel (attention_enc.hip); flops = 4 Tq Tk hd per head and clip */
int afhip_prof_enable(int max_launches);
int afhip_prof_collect(int dtype, int* n_launches, double* total_ms, double* total_flops);

/* Skinny GEMM for decode (M <= 64): weight-streaming, HBM-bound.  Same math as afhip_gemm, act NONE only. */
int afhip_gemm_skinny(const afhip_gemm_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row-wise normalisation and element-wise pieces.
 *   layernorm   : nn.LayerNorm eps 1e-5 (modeling_whisper.py:490,501,748)
 *   avgpool_ln  : AvgPool1d(2,2) over time then LayerNorm (modeling_whisper.py:744-748); x [B,2*Tout,D] -> y [B,Tout,D]
 *   rmsnorm     : Qwen2RMSNorm (modeling_qwen2.py:238-252)
 *   embed_sum   : embed_tokens(ids).sum(dim=2) (lm/parallel.py:260,580); ids [n_tok, S] int64
 *   rope_kv     : rotate-half RoPE on q,k of a fused qkv buffer [B*T, ld_qkv] (token (b,t) sits at position pos0+t)
 *                 and append of k,v to the cache [B,n_kv,cap,hd] (modeling_qwen2.py:105-136,219-222; replaces
 *                 DynamicCache.update's torch.cat); cos/sin tables [rope_max_pos, hd/2] f32
 *   transpose_cast: [B,R,C] -> [B,C,R] with dtype change (mel layout [B,128,3000] <-> [B,3000,128])
 */
int afhip_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int D, float eps, int dtype, void* stream);
int afhip_avgpool_ln(const void* x, const void* w, const void* b, void* y, int B, int Tout, int D, float eps, int dtype, void* stream);
int afhip_rmsnorm(const void* x, const void* w, void* y, int rows, int D, float eps, int dtype, void* stream);
/* LayerNorm statistics for the folded form above.  finalize: partials [P][rows][2] (sum, sum of squares over D/P columns
 * each) -> stats [rows][2] = (mean, rsqrt(var + eps)), var = E[x^2] - mean^2 in f32.  row_stats: the same from the rows
 * themselves (first layer, whose input comes from the conv stem). */
/* Dynamic per-row e4m3 quantisation of activations for the fp8-operand GEMM, with the normalisation in front of it fused:
 *   mode 0: y = x              mode 1: y = LayerNorm(x; w, b, eps) (modeling_whisper.py:490,501)
 *   mode 2: y = RMSNorm(x; w, eps) (modeling_qwen2.py:238-252; x*rstd rounded to bf16 before the gain, like the reference)
 * q[r, :] = e4m3(y[r, :] / s[r]), s[r] = max|y[r, :]| / 448 (1 for an all-zero row).  x [rows, ld_x] bf16, q [rows, D] bytes,
 * scale [rows] f32.  D % 8 == 0, D <= 20480. */
int afhip_quant_rows(const void* x, int ld_x, const void* w, const void* b, float eps, int mode, void* q, float* scale,
                     int rows, int D, void* stream);
/* Calibration helper for statically quantised activations (afhip_encoder_weights.fc2_in_scale): max |x| over n bf16 values
 * (n % 8 == 0, x 16-byte aligned), merged into *out with an atomic max on the float's bit pattern -- the caller zeroes *out.
 * An infinity or a NaN in x comes out as inf / NaN (their patterns order above every finite value): the caller's isfinite() check sees it. */
int afhip_absmax_bf16(const void* x, long long n, float* out, void* stream);
int afhip_ln_stats_finalize(const float* partials, int P, int rows, int D, float eps, float* stats, void* stream);
int afhip_row_stats(const void* x, int rows, int D, float eps, int dtype, float* stats, void* stream);
/* Row gather of the AF3 / Qwen2-Audio placeholder merge: replaces the three index_put / masked assignments of
 * Qwen2AudioForConditionalGeneration._merge_input_ids_with_audio_features (modeling_whisper.py:1056-1104).
 * plan [n_rows] int32 (device): >= 0 text row index into text_rows [n_text_rows, row_bytes]; <= -2 audio row -(v + 2) of
 * audio_rows [n_audio_rows, row_bytes]; -1 zero row (padding).  The host builds the plan (int64 index math of :1021-1096)
 * and range-checks it; n_text_rows / n_audio_rows are carried for the null-pointer checks only. */
int afhip_gather_rows(const void* text_rows, const void* audio_rows, const int32_t* plan, void* out, int n_rows,
                      int n_text_rows, int n_audio_rows, int row_bytes, void* stream);
int afhip_embed_sum(const int64_t* ids, const void* table, void* out, int n_tok, int S, int H, int vocab, int dtype, void* stream);
int afhip_rope_kv(void* qkv, int ld_qkv, const float* cos_table, const float* sin_table, int pos0,
                  void* k_cache, void* v_cache, int B, int T, int n_q, int n_kv, int hd, int cache_cap,
                  int rope_max_pos, int dtype, void* stream);
int afhip_transpose_cast(const void* x, void* y, int B, int R, int C, int in_dtype, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention (flash-style, never materialises [T,T] scores).
 *   softmax(q k^T * scale + mask) v, replaces modeling_whisper.py:186-203 / 418-425 (encoder; additive
 *   key-padding mask of audio.py:1147-1161 given as key_len[B]) and modeling_qwen2.py:150-173 (LLM, causal, GQA).
 *   q[b,t,h,:] = q + b*q_batch_stride + t*ld_q + h*q_head_stride (elements); k/v likewise with the kv strides
 *   (packed qkv rows and the [B,n_kv,cap,hd] cache are both expressible); out [B,Tq,n_q*hd], row stride ld_o.
 *   key_len[b] (may be NULL): keys >= key_len[b] are masked.  causal: key j visible to query i iff j <= i + q_pos0.
 *   Decode (one new token per sequence): pass the q heads of one kv group as the "query rows" (Tq = n_q/n_kv,
 *   ld_q = hd, q_head_stride = Tq*hd, n_q = n_kv) so a workgroup streams each K/V byte once for the whole group.
 */
typedef struct {
    const void* q;
    const void* k;
    const void* v;
    void* out;
    const int32_t* key_len;
    int B, Tq, Tk, n_q, n_kv, hd;
    int ld_q, ld_kv, ld_o;
    long long q_batch_stride, kv_batch_stride, o_batch_stride;
    long long q_head_stride, kv_head_stride; /* elements between heads (hd for packed rows, cap*hd for the KV cache) */
    int causal, q_pos0;
    float scale;
    int dtype;
    long long o_head_stride;   /* elements between output heads; 0 = hd */
    int key_split;             /* > 0: split the keys into ranges of this many (multiple of 64) over workgroups and merge
                                  the partial softmaxes in a second pass: for decode-size query counts (Tq <= 32, causal 0) */
    void* partial_ws;          /* f32 scratch, ceil(Tk/key_split) * B * n_q * 32 * (hd + 2) * 4 bytes */
    size_t partial_ws_bytes;
    int q_prescaled;           /* != 0: q already carries scale * log2(e) (folded into the q projection by the caller, one rounding);
                                * `scale` is ignored and exp2 is taken of the raw scores (one VALU op per score less). */
    /* Decode only (key_split > 0): rotate-half RoPE and the KV-cache append folded into this launch, so a decode step needs no
     * afhip_rope_kv pass.  new_k != NULL switches it on: q rows are rotated as they are loaded (cos/sin rows of the token's
     * position); the workgroup whose key range holds position Tk-1 reads the token's un-rotated k and its v from
     * new_k / new_v (+ b * new_kv_batch_stride + kv_head * hd elements), rotates k, writes both into k / v (the cache) at
     * position Tk-1 and uses them in place of the stale cache row.  Arithmetic identical to afhip_rope_kv
     * (modeling_qwen2.py:105-136: separate roundings of the two products and the sum, then the storage dtype). */
    const void* new_k;
    const void* new_v;
    long long new_kv_batch_stride;
    const float* rope_cos; /* [hd/2] f32 of the position being appended */
    const float* rope_sin;
    /* key_split > 0 only: merge the partial softmaxes INSIDE this launch instead of a second pass.  split_ticket: B * n_q device
     * ints, zero before the launch (the kernel leaves them zero again).  Every key-range workgroup publishes its partial
     * (stores -> s_waitcnt vmcnt(0) -> barrier -> agent-scope release -> relaxed agent atomic add); the workgroup that draws the
     * last ticket acquires and merges all ranges in the same order as the second pass would: bit-identical output. */
    int* split_ticket;
    /* key_split > 0 only: per-sequence context lengths read ON THE DEVICE.  seq_pos [B] int32 = position of the token being
     * generated for each sequence: sequence b attends to keys [0, seq_pos[b]] (its Tk is seq_pos[b] + 1), the fused RoPE uses row
     * seq_pos[b] of the FULL cos/sin tables (rope_cos / rope_sin then point at row 0) and the append lands at that slot.  `Tk` is
     * then only an upper bound that sizes the grid; key-range workgroups beyond a sequence's context exit.  This is what lets
     * one captured hipGraph of a decode step be replayed for every token (nothing position-dependent is baked into kernel
     * arguments) and what a ragged batch (AF3 generate() with left / right padded prompts) needs. */
    const int32_t* seq_pos;
    /* Packed (ragged) self-attention, key_split == 0 and causal == 0 only: row_off [B] int32 (device).  Sequence b's rows start at row
     * row_off[b] of q / k / v / out (row pitches ld_q / ld_kv / ld_o; the batch strides are ignored) and it has key_len[b] queries
     * and keys (key_len is then mandatory); Tq == Tk is the upper bound that sizes the grid, query tiles past a sequence's length
     * exit.  This is how a batch of clips of different lengths runs on M = sum of lengths rows instead of B * max length
     * (afhip_encoder_forward_ragged).  NULL = [B, T] batches. */
    const int32_t* row_off;
    /* encoder form only (bf16, head_dim 64, q_prescaled, non-causal; afhip_attention fails otherwise): out_fp8 != 0 writes `out` as OCP
     * e4m3 BYTES (same element strides, one byte per element) = sat(value * out_scale_inv) instead of bf16 -- the statically quantised
     * A operand of the out-projection (afhip_gemm_args.a_scale_const = 1 / out_scale_inv), without a quantisation pass. */
    int out_fp8;
    float out_scale_inv;
    /* split-context (decode) form, bf16 only: 8 or 16 (>= B) = `out` is not [B, n_q hd] rows but the fragment-order activation image
     * of that many rows the decode step's o projection streams (element (b, k) at (k / 64) * rows * 128 + (((k >> 3) & 1) * 4 * rows +
     * ((k >> 4) & 3) * rows + b) * 16 + (k & 7) * 2 bytes; csrc/img_phase.h).  0 = plain rows. */
    int out_img_rows;
} afhip_attn_args;
int afhip_attention(const afhip_attn_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AF-Whisper encoder forward.  Replaces AFWhisperEncoder.forward (modeling_whisper.py:640-756), the mask
 * construction of ContinuousAudioIO.encode_batch (audio.py:1129-1161) and SoundTower.forward
 * (sound_encoder.py:81-107): the key-padding mask is the per-clip feat_len vector, never a [B,1,1500,1500] tensor.
 * Weights are packed once by the host side (python: AFWhisperEncoder.pack()): conv weights as [d, 3*C]
 * (tap-major), q/k/v fused as [3d, d] with a zero k-bias section.
 */
typedef struct {
    int n_mels, d_model, n_heads, ffn_dim, n_layers, max_pos; /* 128,1280,20,5120,32,1500 */
    int dtype;
    const void* conv1_w; const void* conv1_b; /* [d, 3*n_mels], [d] */
    const void* conv2_w; const void* conv2_b; /* [d, 3*d], [d] */
    const void* pos_emb;                      /* [max_pos, d] */
    const void* const* ln1_w; const void* const* ln1_b; /* per layer */
    const void* const* qkv_w; const void* const* qkv_b; /* [3d, d], [3d] */
    const void* const* out_w; const void* const* out_b;
    const void* const* ln2_w; const void* const* ln2_b;
    const void* const* fc1_w; const void* const* fc1_b;
    const void* const* fc2_w; const void* const* fc2_b;
    const void* lnf_w; const void* lnf_b;
    /* optional, bf16 only (all six or none; NULL = every LayerNorm is its own pass): LayerNorm-folded copies of the two
     * projections that follow a LayerNorm, built once by the host (AFWhisperEncoder.pack()):
     *   qkv_wf [3d,d] = qkv_w diag(ln1_w) (bf16), qkv_cs [3d] = row sums of qkv_wf (f32), qkv_bf [3d] = qkv_b + qkv_w ln1_b (f32)
     *   fc1_wf [ffn,d], fc1_cs [ffn], fc1_bf [ffn] likewise with ln2.  See afhip_gemm_args.ln_stats. */
    const void* const* qkv_wf; const float* const* qkv_cs; const float* const* qkv_bf;
    const void* const* fc1_wf; const float* const* fc1_cs; const float* const* fc1_bf;
    int q_prescaled;  /* != 0: the q rows of qkv_wf / qkv_cs / qkv_bf also carry head_dim^-0.5 * log2(e) (softmax scale in exp2 units) */
    /* optional, bf16 models only (all eight or none): OCP e4m3 copies of the four projection weights of every layer + one f32
     * scale per output channel.  When present the layer runs its GEMMs on e4m3 operands (BASELINE config 5): each projection's
     * input is quantised per row by afhip_quant_rows (the two LayerNorms fused into that pass, no LayerNorm fold), f32 accumulate,
     * bf16 residual stream / attention unchanged.  No reference counterpart: tolerance = error vs the bf16 path (tests). */
    const void* const* qkv_w8; const float* const* qkv_s8;
    const void* const* out_w8; const float* const* out_s8;
    const void* const* fc1_w8; const float* const* fc1_s8;
    const void* const* fc2_w8; const float* const* fc2_s8;
    /* optional, e4m3 mode only.  fc2_in_scale (HOST array, [n_layers] f32, all > 0): static scale of fc2's input (the GELU output) per
     * layer; when present fc1's epilogue writes that activation as e4m3 directly (afhip_gemm_args.out_fp8) and fc2 runs on e4m3
     * operands with a_scale_const -- no [rows, ffn] bf16 round trip, no per-row quantisation pass.  Values come from calibration:
     * calib_amax (DEVICE array, [n_layers] f32, zeroed by the caller) makes a forward record max |GELU output| of every layer there
     * (python: AFWhisperEncoder.calibrate_fp8; [2 * n_layers], see att_out_scale).  Either may be NULL. */
    const float* fc2_in_scale;
    float* calib_amax;
    /* optional, e4m3 mode only: att_out_scale (HOST array, [n_layers] f32, all > 0): static scale of the attention output per layer;
     * when present the encoder attention kernel writes e4m3 directly (afhip_attn_args.out_fp8) and the out-projection reads it with
     * a_scale_const.  With calib_amax set, a forward also records max |attention output| of layer l at calib_amax[n_layers + l]
     * (calib_amax is then [2 * n_layers]). */
    const float* att_out_scale;
} afhip_encoder_weights;
size_t afhip_encoder_workspace_bytes(const afhip_encoder_weights* w, int B);
/* mel_btc [B, 2*max_pos, n_mels] (dtype of the weights); feat_len [B] int32 or NULL (no masking);
 * out [B, max_pos/2, d]; hidden_out (optional, may be NULL): [B, max_pos, d] pre-pool states of layer `hidden_layer`
 * (-1 = conv stem) for parity tests. */
int afhip_encoder_forward(const afhip_encoder_weights* w, const void* mel_btc, const int32_t* feat_len, int B,
                          void* out, void* hidden_out, int hidden_layer, void* workspace, size_t workspace_bytes,
                          void* stream);
/* The same forward for a batch of clips of DIFFERENT lengths, on packed rows: after the conv stem (which runs on the full padded
 * mel, so the last valid position still sees the frames behind it exactly as in the reference) only the feat_len[b] valid
 * positions of each clip are kept, and every layer -- GEMMs, LayerNorm statistics, attention -- runs on M = sum_b feat_len[b] rows
 * instead of B * max_pos.  Per valid position the arithmetic is the one afhip_encoder_forward performs (same kernels, same K
 * order, same key tiles): rows t < (feat_len[b] - 2) / 2 + 1 of out[b] are bit-identical to it; the rows behind them, which the
 * reference computes from padded positions and its callers trim (audio.py:1163-1187), are ZERO here.
 * feat_len [B] int32 on the device and feat_len_host, the same values on the host (the host sizes the launches from them);
 * 0 <= feat_len[b] <= max_pos.  Workspace as afhip_encoder_workspace_bytes(w, B). */
int afhip_encoder_forward_ragged(const afhip_encoder_weights* w, const void* mel_btc, const int32_t* feat_len, const int32_t* feat_len_host,
                                 int B, void* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LLM (Qwen2 decoder stack + UALM head).  Replaces ParallelLLM._step (lm/parallel.py:570-597) over
 * transformers Qwen2Model (modeling_qwen2.py:258-299) with a preallocated KV cache instead of DynamicCache.
 */
typedef struct {
    int hidden, n_layers, n_q, n_kv, hd, inter, vocab, n_stream;
    float rms_eps;
    int dtype;
    const void* embed;                          /* [vocab, hidden] */
    const void* const* ln1_w;                   /* input_layernorm */
    const void* const* qkv_w; const void* const* qkv_b; /* [(n_q+2 n_kv) hd, hidden] */
    const void* const* o_w;                     /* [hidden, n_q hd] */
    const void* const* ln2_w;                   /* post_attention_layernorm */
    const void* const* gu_w;                    /* [2 inter, hidden], gate/up interleaved in 32-row blocks */
    const void* const* down_w;                  /* [hidden, inter] */
    const void* norm_w;
    const void* lm_head;                        /* [vocab, hidden] */
    const void* stream_emb;                     /* [n_stream, hidden] */
    const float* rope_cos; const float* rope_sin; /* [max_pos, hd/2] f32 (modeling_qwen2.py:91-103) */
    int rope_max_pos;
    /* optional W8A16 copies of the weights the decode step streams (NULL = off): OCP e4m3 bytes in the same row order
     * as the bf16 tensors + one f32 scale per output row; used by decode steps only (T == 1, B <= 32, dtype BF16) */
    const void* const* qkv_w8; const float* const* qkv_s;
    const void* const* o_w8; const float* const* o_s;
    const void* const* gu_w8; const float* const* gu_s;
    const void* const* down_w8; const float* const* down_s;
    const void* lm_head8; const float* lm_head_s;
    /* != 0 (needs the e4m3 copies above): prefill (T > 1, more than 64 rows) also runs its four projections per layer on e4m3
     * operands -- activations quantised per row by afhip_quant_rows with the RMSNorm fused -- instead of bf16 */
    int fp8_prefill;
} afhip_llm_weights;

typedef struct {
    void* k; void* v;      /* [n_layers, B, n_kv, cap, hd] */
    int cap, B;
} afhip_kv_cache;

/* max_ctx: KV-cache capacity the T == 1 (decode) calls will run against (sizes the split-context attention partials) */
size_t afhip_llm_workspace_bytes(const afhip_llm_weights* w, int B, int T, int max_ctx);
/* Forward T tokens per sequence from embeddings x [B,T,hidden] starting at cache position pos0; appends K/V.
 * hidden_out [B,T,hidden] = final-normed hidden states (model.norm applied). */
int afhip_llm_forward(const afhip_llm_weights* w, const void* x, int B, int T, int pos0, afhip_kv_cache* cache,
                      void* hidden_out, void* workspace, size_t workspace_bytes, void* stream);

/* One new token per sequence with PER-SEQUENCE positions held on the device: seq_pos [B] int32 = cache slot / RoPE position of each
 * sequence's token (sequence b then attends to keys [0, seq_pos[b]]); max_pos >= every seq_pos[b] bounds them for the host-side
 * range checks and the grid.  Serves ragged batches (AF3 generate() over left / right padded prompts,
 * modeling_whisper.py:1250-1318) and graph replay (afhip_decode_state.seq_pos).  x [B, hidden], hidden_out [B, hidden]. */
int afhip_llm_forward_ragged(const afhip_llm_weights* w, const void* x, int B, const int32_t* seq_pos, int max_pos,
                             afhip_kv_cache* cache, void* hidden_out, void* workspace, size_t workspace_bytes, void* stream);

/* logits[r, s, :] = (hidden[r] + (s ? stream_emb[s] : 0)) . lm_head^T for s < n_s (lm/parallel.py:588-592); f32 out. */
int afhip_lm_head(const afhip_llm_weights* w, const void* hidden, int rows, int n_s, float* logits, void* workspace,
                  size_t workspace_bytes, void* stream);

/* Masked greedy pick on stream 0 (lm/parallel.py:594-601): argmax over ids inside `allowed` half-open intervals
 * [lo,hi) (n_iv of them, int32 pairs) with first-index tie-break; writes int64 token[r].
 * logits_dtype = the MODEL dtype: the reference takes argmax over lm_head's output in the model dtype, so with AFHIP_BF16 each
 * f32 logit is rounded to bf16 before it is compared (equal bf16 values tie, first index wins); AFHIP_F32 compares as is.
 * workspace >= afhip_masked_argmax_workspace_bytes(rows): partial (value, index) pairs; the library keeps no state of its own,
 * so calls on different streams may overlap. */
size_t afhip_masked_argmax_workspace_bytes(int rows);
int afhip_masked_argmax(const float* logits, int rows, int ld, const int32_t* allowed, int n_iv, int64_t* token,
                        int logits_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Sampling branch of the decode loop: top-k -> softmax(values / temperature) -> draw (lm/parallel.py:603-608), with the
 * classifier-free-guidance mix and re-mask of inference_segment folded in (lm/parallel.py:489-492:
 * logits * cfg + cfg_logits * (1 - cfg) as three separately rounded tensor ops, then the modality mask).  One row per
 * (sequence, stream).  The reference draws with torch.multinomial; here the draw is the inverse CDF at a caller-supplied uniform
 * u[r] (same distribution, no RNG stream to reproduce); leave u / token NULL to get only the top-k set and its probabilities.
 * Two stated deviations: (1) softmax(values / temperature) and the CDF are computed in f32 for every model dtype -- the reference
 * runs them on tensors of the model dtype (lm/parallel.py:603-608), so for bf16 models its probabilities carry bf16 rounding
 * (<= 2^-8 relative) that these do not; the top-k SET and the mixed logits are rounded exactly as the reference's tensors are.
 * (2) k <= 64 (the reference accepts any topk; conf/inference.yaml uses 20): larger k returns AFHIP_ERR_INVALID. */
typedef struct {
    const float* logits;      /* [rows, ld] f32: lm_head output of the conditional half */
    const float* cfg_logits;  /* [rows, ld] f32 of the unconditional (all-pad cache) half, or NULL: no guidance */
    float cfg;                /* guidance weight (conf/inference.yaml:7), used when cfg_logits != NULL */
    int rows, ld;
    const int32_t* allowed;   /* [rows][n_iv][2] half-open id intervals the row may emit (its stream's row of the modality mask;
                                 unused slots lo == hi) */
    int n_iv;
    int k;                    /* 1..64 (conf/inference.yaml:6: 20) */
    float temperature;        /* > 0 */
    int model_dtype;          /* AFHIP_BF16: logits and every intermediate of the mix are rounded to bf16, as the reference's tensors are */
    int32_t* topk_idx;        /* [rows, k] out (may be NULL): ids by descending value, ties by ascending id */
    float* topk_val;          /* [rows, k] out (may be NULL): the mixed, masked logits at those ids */
    float* topk_prob;         /* [rows, k] out (may be NULL): softmax(val / temperature) */
    const float* u;           /* [rows] uniforms in [0,1), or NULL */
    int64_t* token;           /* [rows] out, or NULL: topk_idx[first j with cdf_j > u] */
    float one_minus_cfg;      /* the weight of cfg_logits.  The reference computes `(1 - cfg)` in Python DOUBLE and the tensor op then
                                 rounds it to f32 (lm/parallel.py:489-492): pass (float)(1.0 - cfg_as_double).  0 = derive it here as
                                 (float)(1.0 - (double)cfg), which equals the reference only when cfg is exact in f32 (3.0 is, 1.3 is not) */
} afhip_sample_args;
int afhip_sample_topk(const afhip_sample_args* args, void* stream);

/* One greedy decode step for B sequences entirely on device: embed prev token (stream 0 = token, others pad),
 * forward 1 position, lm_head on stream 0, masked argmax, append to out_tokens[step], update finished flags
 * (eos/eot, lm/parallel.py:503-513).  No host sync. */
typedef struct {
    int64_t* prev_token;     /* [B] in/out */
    int64_t* out_tokens;     /* [max_step, B] */
    int32_t* finished_at;    /* [B], -1 while running */
    const int32_t* allowed;  /* [n_iv,2] */
    int n_iv;
    int eos_id, eot_id;
    /* Optional, both or neither (NULL = the host scalars `pos` / `step` of the call are used): device-resident loop state, so that
     * ONE captured hipGraph of afhip_llm_decode_step can be replayed for every token.  seq_pos [B]: position of the token each
     * sequence appends next (the step increments it); step_counter [1]: row of out_tokens to write (the step increments it).
     * `pos` is then the upper bound of seq_pos over the replays (checked against the cache capacity and the RoPE table). */
    int32_t* seq_pos;
    int32_t* step_counter;
    /* > 0: every id inside `allowed` is below head_rows, so the step may compute logits for lm_head rows [0, head_rows) only (text decode:
     * 256 + text vocab of the 160 520 rows, lm/parallel.py:557-568).  0 = all rows. */
    int head_rows;
} afhip_decode_state;
int afhip_llm_decode_step(const afhip_llm_weights* w, afhip_kv_cache* cache, afhip_decode_state* st, int B, int pos,
                          int step, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFHIP_H */
