"""`torch.library` registration of the stateless entry points of the C ABI (include/afhip.h) as PyTorch custom ops under the
`afhip::` namespace -- the boundary wording of BASELINE's north star ("behind PyTorch-ROCm custom ops").  The ops are thin: the
implementation of each is the ctypes call in ops.py (device pointers + the current HIP stream), so there is still exactly one native
library and no second code path; what the registration adds is the dispatcher entry, schema checking and fake-tensor (meta) kernels,
so the ops can be traced / shape-propagated (torch.compile, FakeTensorMode) without a GPU.  Stateful pieces (the encoder and LLM
forwards, which take packed weight tables) stay methods of the drop-in classes.

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


_impls = {
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
