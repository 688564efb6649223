"""`torch.library` registration of the stateless entry points of the C ABI (include/afhip.h) as PyTorch custom ops under the
`afhip::` namespace -- the boundary wording of BASELINE's north star ("behind PyTorch-ROCm custom ops").  The ops are thin: the
implementation of each is the ctypes call in ops.py (device pointers + the current HIP stream), so there is still exactly one native
library and no second code path; what the registration adds is the dispatcher entry, schema checking and fake-tensor (meta) kernels,
so the ops can be traced / shape-propagated (torch.compile, FakeTensorMode) without a GPU.

The entry points that carry the time -- `afhip_encoder_forward(_ragged)`, `afhip_llm_forward(_ragged)`, `afhip_lm_head`,
`afhip_llm_decode_step` -- are ops too, and the drop-in classes (AFWhisperEncoder.encode_btc, ParallelLLM._forward_hidden / _step /
the greedy device loop, Qwen2AudioForConditionalGeneration) call THROUGH them: `torch.ops.afhip.encoder_forward(...)`.  Their packed
weight table (`afhip_encoder_weights` / `afhip_llm_weights`: pointers to tensors the module keeps alive) travels as an OPAQUE uint8 CPU
tensor that aliases the ctypes struct (`pack().blob`); KV caches, workspaces and loop state are ordinary tensors the schema marks
as mutated.

    import audio_intelligence_amd.torch_ops          # registers torch.ops.afhip.*
    y = torch.ops.afhip.gemm(a, w, bias, 1, None)    # == ops.gemm(a, w, bias=bias, act=ACT_GELU)
"""
from typing import Optional

import torch

from . import _lib as L
from . import ops

_lib_def = torch.library.Library("afhip", "DEF")
_lib_def.define("gemm(Tensor a, Tensor w, Tensor? bias, int act, Tensor? residual) -> Tensor")
_lib_def.define("layernorm(Tensor x, Tensor w, Tensor b, float eps) -> Tensor")
_lib_def.define("rmsnorm(Tensor x, Tensor w, float eps) -> Tensor")
_lib_def.define("embed_sum(Tensor ids, Tensor table) -> Tensor")
_lib_def.define("attention_packed(Tensor qkv, int n_heads, Tensor? key_len, bool causal) -> Tensor")
_lib_def.define("log_mel(Tensor wav, bool btc, ScalarType dtype) -> Tensor")
_lib_def.define("quant_rows(Tensor x, int mode, Tensor? w, Tensor? b, float eps) -> (Tensor, Tensor)")
_lib_def.define("gemm_fp8(Tensor aq, Tensor a_scale, Tensor wq, Tensor w_scale, Tensor? bias, int act, Tensor? residual) -> Tensor")
# stateful entry points: `weights` = opaque uint8 CPU tensor aliasing the packed-weight struct
_lib_def.define("encoder_forward(Tensor weights, Tensor mel_btc, Tensor? feat_len, int hidden_layer, Tensor(a!) workspace) -> (Tensor, Tensor)")
_lib_def.define("encoder_forward_ragged(Tensor weights, Tensor mel_btc, Tensor feat_len, Tensor feat_len_host, Tensor(a!) workspace) -> Tensor")
_lib_def.define("llm_forward(Tensor weights, Tensor x, int pos0, Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor(c!) workspace) -> Tensor")
_lib_def.define("llm_forward_ragged(Tensor weights, Tensor x, Tensor seq_pos, int max_pos, Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor(c!) workspace) -> Tensor")
_lib_def.define("lm_head(Tensor weights, Tensor hidden_rows, int n_stream, Tensor(a!) workspace) -> Tensor")
_lib_def.define("llm_decode_step(Tensor weights, Tensor(a!) k_cache, Tensor(b!) v_cache, Tensor(c!) prev_token, Tensor(d!) out_tokens, Tensor(e!) finished_at, "
                "Tensor allowed, int eos_id, int eot_id, Tensor(f!) seq_pos, Tensor(g!) step_counter, int max_pos, Tensor(h!) workspace, "
                "int head_rows) -> ()")

_fe = None


def _extractor():
    global _fe
    if _fe is None:
        from .multimodal_io.feature_extraction import WhisperFeatureExtractorHIP
        _fe = WhisperFeatureExtractorHIP()
    return _fe


def _gemm(a, w, bias, act, residual):
    return ops.gemm(a, w, bias=bias, act=act, residual=residual)


def _gemm_meta(a, w, bias, act, residual):
    n = w.shape[0] // 2 if act == L.ACT_SWIGLU else w.shape[0]
    return a.new_empty((a.shape[0], n))


def _attention(qkv, n_heads, key_len, causal):
    return ops.attention_packed(qkv, n_heads, key_len=key_len, causal=causal)


def _log_mel(wav, btc, dtype):
    return _extractor().extract_device(wav, layout="btc" if btc else "bct", dtype=dtype)


def _log_mel_meta(wav, btc, dtype):
    B = wav.shape[0]
    return wav.new_empty((B, 3000, 128) if btc else (B, 128, 3000), dtype=dtype)


def _quant_rows(x, mode, w, b, eps):
    return ops.quant_rows(x, mode, w, b, eps)


def _quant_rows_meta(x, mode, w, b, eps):
    return x.new_empty(x.shape, dtype=torch.uint8), x.new_empty((x.shape[0],), dtype=torch.float32)


def _gemm_fp8(aq, a_scale, wq, w_scale, bias, act, residual):
    return ops.gemm_fp8(aq, a_scale, wq, w_scale, bias=bias, act=act, residual=residual)


def _gemm_fp8_meta(aq, a_scale, wq, w_scale, bias, act, residual):
    n = wq.shape[0] // 2 if act == L.ACT_SWIGLU else wq.shape[0]
    return aq.new_empty((aq.shape[0], n), dtype=torch.bfloat16)


# ---------------------------------------------------------------------------------------------------------------------------
# stateful entry points
import ctypes as C  # noqa: E402


def weights_blob(struct) -> torch.Tensor:
    """The opaque `weights` argument: a uint8 CPU tensor that ALIASES a packed-weight ctypes struct (no copy; the struct -- and the
    tensors / pointer arrays it points to -- must stay alive as long as the blob is used: the modules keep both in `pack()`)."""
    return torch.frombuffer(struct, dtype=torch.uint8)


def _struct(blob: torch.Tensor, typ):
    if blob.device.type != "cpu" or blob.dtype != torch.uint8 or blob.numel() != C.sizeof(typ):
        raise L.AfhipError(f"weights must be the uint8 CPU blob of a {typ.__name__} ({C.sizeof(typ)} bytes)")
    return C.cast(blob.data_ptr(), C.POINTER(typ))


def _kv(k_cache, v_cache):
    s = L.KvCache()
    s.k, s.v, s.cap, s.B = k_cache.data_ptr(), v_cache.data_ptr(), k_cache.shape[3], k_cache.shape[1]
    return s


def _enc_dims(blob):
    w = _struct(blob, L.EncoderWeights).contents
    return w.max_pos, w.d_model


def _encoder_forward(weights, mel_btc, feat_len, hidden_layer, workspace):
    lib = L.lib()
    w = _struct(weights, L.EncoderWeights)
    B = mel_btc.shape[0]
    Tp, d = w.contents.max_pos, w.contents.d_model
    out = torch.empty((B, Tp // 2, d), dtype=mel_btc.dtype, device=mel_btc.device)
    hid = torch.empty((B, Tp, d) if hidden_layer >= -1 else (0,), dtype=mel_btc.dtype, device=mel_btc.device)
    L.check(lib.afhip_encoder_forward(w, L.ptr(mel_btc), L.ptr(feat_len), B, L.ptr(out), L.ptr(hid) if hidden_layer >= -1 else None,
                                      hidden_layer if hidden_layer >= -1 else -1, L.ptr(workspace), workspace.numel(), L.stream_ptr()))
    return out, hid


def _encoder_forward_meta(weights, mel_btc, feat_len, hidden_layer, workspace):
    Tp, d = _enc_dims(weights)
    B = mel_btc.shape[0]
    return mel_btc.new_empty((B, Tp // 2, d)), mel_btc.new_empty((B, Tp, d) if hidden_layer >= -1 else (0,))


def _encoder_forward_ragged(weights, mel_btc, feat_len, feat_len_host, workspace):
    lib = L.lib()
    w = _struct(weights, L.EncoderWeights)
    B = mel_btc.shape[0]
    out = torch.empty((B, w.contents.max_pos // 2, w.contents.d_model), dtype=mel_btc.dtype, device=mel_btc.device)
    L.check(lib.afhip_encoder_forward_ragged(w, L.ptr(mel_btc), L.ptr(feat_len), feat_len_host.data_ptr(), B, L.ptr(out),
                                             L.ptr(workspace), workspace.numel(), L.stream_ptr()))
    return out


def _encoder_forward_ragged_meta(weights, mel_btc, feat_len, feat_len_host, workspace):
    Tp, d = _enc_dims(weights)
    return mel_btc.new_empty((mel_btc.shape[0], Tp // 2, d))


def _llm_forward(weights, x, pos0, k_cache, v_cache, workspace):
    lib = L.lib()
    B, T = x.shape[0], x.shape[1]
    hid = torch.empty_like(x)
    cs = _kv(k_cache, v_cache)
    L.check(lib.afhip_llm_forward(_struct(weights, L.LlmWeights), L.ptr(x), B, T, pos0, C.byref(cs), L.ptr(hid), L.ptr(workspace), workspace.numel(), L.stream_ptr()))
    return hid


def _llm_forward_ragged(weights, x, seq_pos, max_pos, k_cache, v_cache, workspace):
    lib = L.lib()
    hid = torch.empty_like(x)
    cs = _kv(k_cache, v_cache)
    L.check(lib.afhip_llm_forward_ragged(_struct(weights, L.LlmWeights), L.ptr(x), x.shape[0], L.ptr(seq_pos), max_pos, C.byref(cs), L.ptr(hid),
                                         L.ptr(workspace), workspace.numel(), L.stream_ptr()))
    return hid


def _lm_head(weights, hidden_rows, n_stream, workspace):
    lib = L.lib()
    w = _struct(weights, L.LlmWeights)
    n = hidden_rows.shape[0]
    logits = torch.empty((n, n_stream, w.contents.vocab), dtype=torch.float32, device=hidden_rows.device)
    L.check(lib.afhip_lm_head(w, L.ptr(hidden_rows), n, n_stream, L.ptr(logits), L.ptr(workspace), workspace.numel(), L.stream_ptr()))
    return logits


def _lm_head_meta(weights, hidden_rows, n_stream, workspace):
    return hidden_rows.new_empty((hidden_rows.shape[0], n_stream, _struct(weights, L.LlmWeights).contents.vocab), dtype=torch.float32)


def _llm_decode_step(weights, k_cache, v_cache, prev_token, out_tokens, finished_at, allowed, eos_id, eot_id, seq_pos, step_counter, max_pos, workspace,
                     head_rows):
    lib = L.lib()
    st = L.DecodeState()
    st.prev_token, st.out_tokens, st.finished_at = prev_token.data_ptr(), out_tokens.data_ptr(), finished_at.data_ptr()
    st.allowed, st.n_iv, st.eos_id, st.eot_id = allowed.data_ptr(), allowed.shape[0], eos_id, eot_id
    st.seq_pos, st.step_counter = seq_pos.data_ptr(), step_counter.data_ptr()
    st.head_rows = head_rows
    cs = _kv(k_cache, v_cache)
    L.check(lib.afhip_llm_decode_step(_struct(weights, L.LlmWeights), C.byref(cs), C.byref(st), prev_token.shape[0], max_pos, 0,
                                      L.ptr(workspace), workspace.numel(), L.stream_ptr()))


_impls = {
    "encoder_forward": (_encoder_forward, _encoder_forward_meta),
    "encoder_forward_ragged": (_encoder_forward_ragged, _encoder_forward_ragged_meta),
    "llm_forward": (_llm_forward, lambda weights, x, pos0, k_cache, v_cache, workspace: torch.empty_like(x)),
    "llm_forward_ragged": (_llm_forward_ragged, lambda weights, x, seq_pos, max_pos, k_cache, v_cache, workspace: torch.empty_like(x)),
    "lm_head": (_lm_head, _lm_head_meta),
    "llm_decode_step": (_llm_decode_step, lambda *a: None),
    "gemm": (_gemm, _gemm_meta),
    "layernorm": (lambda x, w, b, eps: ops.layernorm(x, w, b, eps), lambda x, w, b, eps: torch.empty_like(x)),
    "rmsnorm": (lambda x, w, eps: ops.rmsnorm(x, w, eps), lambda x, w, eps: torch.empty_like(x)),
    "embed_sum": (lambda ids, table: ops.embed_sum(ids, table), lambda ids, table: table.new_empty(tuple(ids.shape[:-1]) + (table.shape[1],))),
    "attention_packed": (_attention, lambda qkv, n_heads, key_len, causal: qkv.new_empty((qkv.shape[0], qkv.shape[1], qkv.shape[2] // 3))),
    "log_mel": (_log_mel, _log_mel_meta),
    "quant_rows": (_quant_rows, _quant_rows_meta),
    "gemm_fp8": (_gemm_fp8, _gemm_fp8_meta),
}
_lib_impl = torch.library.Library("afhip", "IMPL")
for _name, (_cuda, _meta) in _impls.items():
    _lib_impl.impl(_name, _cuda, "CUDA")          # "CUDA" is the HIP device key on ROCm builds; there is no CPU kernel (no fallback)
    _lib_impl.impl(_name, _meta, "Meta")
OP_NAMES = sorted(_impls)
