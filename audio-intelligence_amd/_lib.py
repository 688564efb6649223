"""ctypes binding of the C ABI in include/afhip.h (csrc/libafhip.so).

`torch` is imported first on purpose: the library's DT_NEEDED `libamdhip64.so.7` then resolves to the HIP
runtime PyTorch already loaded, so torch's device pointers and streams are valid inside the library.
There is no fallback of any kind: if the library is missing, cannot be loaded, or no GPU is visible,
`lib()` raises and every op built on it fails loudly.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libafhip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SWIGLU = 0, 1, 2

c_void_pp = C.POINTER(C.c_void_p)


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int), ("ldres", C.c_int),
                ("dtype", C.c_int), ("act", C.c_int), ("res_row_mod", C.c_int),
                ("conv_Tin", C.c_int), ("conv_Tout", C.c_int), ("conv_stride", C.c_int), ("conv_C", C.c_int),
                ("out_f32", C.c_int), ("a_norm_w", C.c_void_p), ("a_norm_eps", C.c_float), ("a_swiglu", C.c_int),
                ("w_scale", C.c_void_p),
                ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_bias", C.c_void_p), ("row_stats_out", C.c_void_p),
                ("a_fp8", C.c_int), ("a_scale", C.c_void_p),
                ("a_scale_const", C.c_float), ("out_fp8", C.c_int), ("out_scale_inv", C.c_float)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p), ("key_len", C.c_void_p),
                ("B", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("n_q", C.c_int), ("n_kv", C.c_int), ("hd", C.c_int),
                ("ld_q", C.c_int), ("ld_kv", C.c_int), ("ld_o", C.c_int),
                ("q_batch_stride", C.c_longlong), ("kv_batch_stride", C.c_longlong), ("o_batch_stride", C.c_longlong),
                ("q_head_stride", C.c_longlong), ("kv_head_stride", C.c_longlong),
                ("causal", C.c_int), ("q_pos0", C.c_int), ("scale", C.c_float), ("dtype", C.c_int),
                ("o_head_stride", C.c_longlong), ("key_split", C.c_int), ("partial_ws", C.c_void_p), ("partial_ws_bytes", C.c_size_t),
                ("q_prescaled", C.c_int), ("new_k", C.c_void_p), ("new_v", C.c_void_p), ("new_kv_batch_stride", C.c_longlong),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("split_ticket", C.c_void_p), ("seq_pos", C.c_void_p), ("row_off", C.c_void_p),
                ("out_fp8", C.c_int), ("out_scale_inv", C.c_float), ("out_img_rows", C.c_int)]


class EncoderWeights(C.Structure):
    _fields_ = [("n_mels", C.c_int), ("d_model", C.c_int), ("n_heads", C.c_int), ("ffn_dim", C.c_int),
                ("n_layers", C.c_int), ("max_pos", C.c_int), ("dtype", C.c_int),
                ("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
                ("pos_emb", C.c_void_p),
                ("ln1_w", c_void_pp), ("ln1_b", c_void_pp), ("qkv_w", c_void_pp), ("qkv_b", c_void_pp),
                ("out_w", c_void_pp), ("out_b", c_void_pp), ("ln2_w", c_void_pp), ("ln2_b", c_void_pp),
                ("fc1_w", c_void_pp), ("fc1_b", c_void_pp), ("fc2_w", c_void_pp), ("fc2_b", c_void_pp),
                ("lnf_w", C.c_void_p), ("lnf_b", C.c_void_p),
                ("qkv_wf", c_void_pp), ("qkv_cs", c_void_pp), ("qkv_bf", c_void_pp),
                ("fc1_wf", c_void_pp), ("fc1_cs", c_void_pp), ("fc1_bf", c_void_pp), ("q_prescaled", C.c_int),
                ("qkv_w8", c_void_pp), ("qkv_s8", c_void_pp), ("out_w8", c_void_pp), ("out_s8", c_void_pp),
                ("fc1_w8", c_void_pp), ("fc1_s8", c_void_pp), ("fc2_w8", c_void_pp), ("fc2_s8", c_void_pp),
                ("fc2_in_scale", C.c_void_p), ("calib_amax", C.c_void_p), ("att_out_scale", C.c_void_p)]


class LlmWeights(C.Structure):
    _fields_ = [("hidden", C.c_int), ("n_layers", C.c_int), ("n_q", C.c_int), ("n_kv", C.c_int), ("hd", C.c_int),
                ("inter", C.c_int), ("vocab", C.c_int), ("n_stream", C.c_int), ("rms_eps", C.c_float), ("dtype", C.c_int),
                ("embed", C.c_void_p),
                ("ln1_w", c_void_pp), ("qkv_w", c_void_pp), ("qkv_b", c_void_pp), ("o_w", c_void_pp),
                ("ln2_w", c_void_pp), ("gu_w", c_void_pp), ("down_w", c_void_pp),
                ("norm_w", C.c_void_p), ("lm_head", C.c_void_p), ("stream_emb", C.c_void_p),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p), ("rope_max_pos", C.c_int),
                ("qkv_w8", c_void_pp), ("qkv_s", c_void_pp), ("o_w8", c_void_pp), ("o_s", c_void_pp),
                ("gu_w8", c_void_pp), ("gu_s", c_void_pp), ("down_w8", c_void_pp), ("down_s", c_void_pp),
                ("lm_head8", C.c_void_p), ("lm_head_s", C.c_void_p), ("fp8_prefill", C.c_int)]


class KvCache(C.Structure):
    _fields_ = [("k", C.c_void_p), ("v", C.c_void_p), ("cap", C.c_int), ("B", C.c_int)]


class DecodeState(C.Structure):
    _fields_ = [("prev_token", C.c_void_p), ("out_tokens", C.c_void_p), ("finished_at", C.c_void_p),
                ("allowed", C.c_void_p), ("n_iv", C.c_int), ("eos_id", C.c_int), ("eot_id", C.c_int),
                ("seq_pos", C.c_void_p), ("step_counter", C.c_void_p), ("head_rows", C.c_int)]


class SampleArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("cfg_logits", C.c_void_p), ("cfg", C.c_float), ("rows", C.c_int), ("ld", C.c_int),
                ("allowed", C.c_void_p), ("n_iv", C.c_int), ("k", C.c_int), ("temperature", C.c_float), ("model_dtype", C.c_int),
                ("topk_idx", C.c_void_p), ("topk_val", C.c_void_p), ("topk_prob", C.c_void_p), ("u", C.c_void_p), ("token", C.c_void_p),
                ("one_minus_cfg", C.c_float)]


# name -> (restype, argtypes); mirrors include/afhip.h one to one (tests/test_cabi.py checks the header against this)
_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "afhip_version": (_I, []),
    "afhip_last_error": (C.c_char_p, []),
    "afhip_set_option": (_I, [C.c_char_p, _I]),
    "afhip_log_mel_tables_bytes": (_Z, []),
    "afhip_log_mel_tables_host": (_I, [_P, _P]),
    "afhip_log_mel_workspace_bytes": (_Z, [_I]),
    "afhip_log_mel": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "afhip_gemm": (_I, [C.POINTER(GemmArgs), _P]),
    "afhip_gemm_skinny": (_I, [C.POINTER(GemmArgs), _P]),
    "afhip_prof_enable": (_I, [_I]),
    "afhip_prof_collect": (_I, [_I, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "afhip_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "afhip_avgpool_ln": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "afhip_rmsnorm": (_I, [_P, _P, _P, _I, _I, _F, _I, _P]),
    "afhip_embed_sum": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "afhip_gather_rows": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "afhip_quant_rows": (_I, [_P, _I, _P, _P, C.c_float, _I, _P, _P, _I, _I, _P]),
    "afhip_absmax_bf16": (_I, [_P, C.c_longlong, _P, _P]),
    "afhip_ln_stats_finalize": (_I, [_P, _I, _I, _I, C.c_float, _P, _P]),
    "afhip_row_stats": (_I, [_P, _I, _I, C.c_float, _I, _P, _P]),
    "afhip_rope_kv": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "afhip_transpose_cast": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "afhip_attention": (_I, [C.POINTER(AttnArgs), _P]),
    "afhip_encoder_workspace_bytes": (_Z, [C.POINTER(EncoderWeights), _I]),
    "afhip_encoder_forward": (_I, [C.POINTER(EncoderWeights), _P, _P, _I, _P, _P, _I, _P, _Z, _P]),
    "afhip_encoder_forward_ragged": (_I, [C.POINTER(EncoderWeights), _P, _P, _P, _I, _P, _P, _Z, _P]),
    "afhip_llm_workspace_bytes": (_Z, [C.POINTER(LlmWeights), _I, _I, _I]),
    "afhip_llm_forward": (_I, [C.POINTER(LlmWeights), _P, _I, _I, _I, C.POINTER(KvCache), _P, _P, _Z, _P]),
    "afhip_llm_forward_ragged": (_I, [C.POINTER(LlmWeights), _P, _I, _P, _I, C.POINTER(KvCache), _P, _P, _Z, _P]),
    "afhip_lm_head": (_I, [C.POINTER(LlmWeights), _P, _I, _I, _P, _P, _Z, _P]),
    "afhip_masked_argmax_workspace_bytes": (_Z, [_I]),
    "afhip_masked_argmax": (_I, [_P, _I, _I, _P, _I, _P, _I, _P, _Z, _P]),
    "afhip_sample_topk": (_I, [C.POINTER(SampleArgs), _P]),
    "afhip_llm_decode_step": (_I, [C.POINTER(LlmWeights), C.POINTER(KvCache), C.POINTER(DecodeState), _I, _I, _I, _P, _Z, _P]),
}

_lib = None


class AfhipError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH):
    """dlopen the C-ABI library and attach signatures. Raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise AfhipError(f"HIP library not built: {path} (run `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"or `make -C {os.path.dirname(path)}`); this package has no CPU fallback")
    lib = C.CDLL(path)
    partial = os.environ.get("AFHIP_ALLOW_PARTIAL") == "1"  # bring-up only: tolerate not-yet-built symbols
    for name, (res, args) in SIGNATURES.items():
        if partial and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib():
    """The loaded library, for compute: additionally requires a visible GPU."""
    L = load_library()
    if not torch.cuda.is_available():
        raise AfhipError("no HIP device visible: the AF3/UALM forward pass runs on MI355X only (no CPU fallback)")
    return L


def check(rc: int):
    if rc != 0:
        msg = load_library().afhip_last_error().decode("utf-8", "replace")
        raise AfhipError(f"afhip error {rc}: {msg}")


def dtype_code(dt) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise AfhipError(f"unsupported dtype {dt} (float32 or bfloat16)")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr
