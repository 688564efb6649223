"""Tensor-level wrappers over the C ABI (device pointers + current HIP stream). PyTorch is plumbing here:
it owns device memory and streams; all arithmetic happens in csrc/."""
import ctypes as C
import math
from typing import Optional

import torch

from . import _lib as L


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise L.AfhipError(f"{name}: expected a CUDA/HIP tensor (no CPU fallback)")
    return t


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = L.ACT_NONE,
         residual: Optional[torch.Tensor] = None, res_row_mod: int = 0, out: Optional[torch.Tensor] = None,
         conv: Optional[tuple] = None, ln_fold: Optional[tuple] = None, row_stats_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(A @ W^T). a [M,K] (or x [B,Tin,C] with conv=(Tout, stride)), w [N,K].
    ln_fold = (stats [M,2] f32, colsum [N] f32, bias [N] f32): LayerNorm folded around the GEMM (afhip.h, afhip_gemm_args.ln_stats);
    row_stats_out [N/64, M, 2] f32 receives the (sum, sum of squares) partials of the stored rows."""
    lib = L.lib()
    _chk(a, "gemm.a"), _chk(w, "gemm.w")
    dt = L.dtype_code(a.dtype)
    N, K = w.shape
    args = L.GemmArgs()
    if conv is not None:
        Bc, Tin, Cc = a.shape
        Tout, stride = conv
        M = Bc * Tout
        assert a.is_contiguous() and K == 3 * Cc
        args.conv_Tin, args.conv_Tout, args.conv_stride, args.conv_C = Tin, Tout, stride, Cc
        args.lda = Cc
    else:
        assert a.dim() == 2 and a.stride(1) == 1 and a.shape[1] == K
        M = a.shape[0]
        args.lda = a.stride(0)
    assert w.stride(1) == 1 and w.dtype == a.dtype
    n_out = N // 2 if act == L.ACT_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=a.dtype, device=a.device)
    assert out.stride(1) == 1 and out.shape[0] == M and out.shape[1] == n_out
    args.A, args.W, args.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    args.bias = bias.data_ptr() if bias is not None else None
    args.residual = residual.data_ptr() if residual is not None else None
    args.M, args.N, args.K = M, N, K
    args.ldw, args.ldc = w.stride(0), out.stride(0)
    args.ldres = residual.stride(0) if residual is not None else 0
    args.dtype, args.act, args.res_row_mod = dt, act, res_row_mod
    if ln_fold is not None:
        st, cs, bf = ln_fold
        assert st.dtype == cs.dtype == bf.dtype == torch.float32 and st.shape == (M, 2) and cs.shape == (N,) and bf.shape == (N,)
        args.ln_stats, args.ln_colsum, args.ln_bias = st.data_ptr(), cs.data_ptr(), bf.data_ptr()
    if row_stats_out is not None:
        assert row_stats_out.dtype == torch.float32 and row_stats_out.is_contiguous() and row_stats_out.numel() >= (N // 64) * M * 2
        args.row_stats_out = row_stats_out.data_ptr()
    L.check(lib.afhip_gemm(C.byref(args), L.stream_ptr()))
    return out


def quant_rows(x: torch.Tensor, mode: int = 0, w: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None, eps: float = 1e-5):
    """Per-row e4m3 quantisation of bf16 rows [rows, D] (mode 0), of LayerNorm(x) (1) or RMSNorm(x) (2): (bytes [rows, D] uint8, scale [rows] f32)."""
    lib = L.lib()
    _chk(x, "quant_rows.x")
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.bfloat16
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    sc = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    L.check(lib.afhip_quant_rows(L.ptr(x), x.stride(0), L.ptr(w), L.ptr(b), eps, mode, L.ptr(q), L.ptr(sc), x.shape[0], x.shape[1], L.stream_ptr()))
    return q, sc


def gemm_fp8(aq: torch.Tensor, a_scale: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None,
             act: int = L.ACT_NONE, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C (bf16) = epilogue(a_scale[m] * w_scale[n] * (Aq . Wq^T)) on e4m3 bytes aq [M,K], wq [N,K] (uint8 views)."""
    lib = L.lib()
    _chk(aq, "gemm_fp8.aq")
    M, K = aq.shape
    N = wq.shape[0]
    n_out = N // 2 if act == L.ACT_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.bfloat16, device=aq.device)
    args = L.GemmArgs()
    args.A, args.W, args.C = aq.data_ptr(), wq.data_ptr(), out.data_ptr()
    args.bias = bias.data_ptr() if bias is not None else None
    args.residual = residual.data_ptr() if residual is not None else None
    args.M, args.N, args.K = M, N, K
    args.lda, args.ldw, args.ldc = aq.stride(0), wq.stride(0), out.stride(0)
    args.ldres = residual.stride(0) if residual is not None else 0
    args.dtype, args.act = L.BF16, act
    args.a_fp8, args.a_scale, args.w_scale = 1, a_scale.data_ptr(), w_scale.data_ptr()
    L.check(lib.afhip_gemm(C.byref(args), L.stream_ptr()))
    return out


def row_stats(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """[rows, D] -> [rows, 2] f32 (mean, rsqrt(var + eps)): the LayerNorm statistics of the GEMM-folded form."""
    lib = L.lib()
    _chk(x, "row_stats.x")
    assert x.dim() == 2 and x.is_contiguous()
    st = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    L.check(lib.afhip_row_stats(L.ptr(x), x.shape[0], x.shape[1], eps, L.dtype_code(x.dtype), L.ptr(st), L.stream_ptr()))
    return st


def ln_stats_finalize(partials: torch.Tensor, D: int, eps: float = 1e-5) -> torch.Tensor:
    """[P, rows, 2] f32 (sum, sum of squares) partials -> [rows, 2] (mean, rstd)."""
    lib = L.lib()
    assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.dim() == 3 and partials.shape[2] == 2
    st = torch.empty((partials.shape[1], 2), dtype=torch.float32, device=partials.device)
    L.check(lib.afhip_ln_stats_finalize(L.ptr(partials), partials.shape[0], partials.shape[1], D, eps, L.ptr(st), L.stream_ptr()))
    return st


def layernorm(x, w, b, eps: float = 1e-5):
    lib = L.lib()
    _chk(x, "layernorm.x")
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.is_contiguous()
    y = torch.empty_like(x2)
    L.check(lib.afhip_layernorm(L.ptr(x2), L.ptr(w), L.ptr(b), L.ptr(y), x2.shape[0], x2.shape[1], eps, L.dtype_code(x.dtype), L.stream_ptr()))
    return y.view(x.shape)


def avgpool_ln(x, w, b, eps: float = 1e-5):
    """x [B, 2*Tout, D] -> LayerNorm(mean of adjacent row pairs) [B, Tout, D]."""
    lib = L.lib()
    _chk(x, "avgpool_ln.x")
    B, T2, D = x.shape
    assert x.is_contiguous() and T2 % 2 == 0
    y = torch.empty((B, T2 // 2, D), dtype=x.dtype, device=x.device)
    L.check(lib.afhip_avgpool_ln(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, T2 // 2, D, eps, L.dtype_code(x.dtype), L.stream_ptr()))
    return y


def rmsnorm(x, w, eps: float = 1e-6):
    lib = L.lib()
    _chk(x, "rmsnorm.x")
    x2 = x.reshape(-1, x.shape[-1])
    assert x2.is_contiguous()
    y = torch.empty_like(x2)
    L.check(lib.afhip_rmsnorm(L.ptr(x2), L.ptr(w), L.ptr(y), x2.shape[0], x2.shape[1], eps, L.dtype_code(x.dtype), L.stream_ptr()))
    return y.view(x.shape)


def embed_sum(ids: torch.Tensor, table: torch.Tensor):
    """ids [..., S] int64 -> sum over streams of table rows [..., H]."""
    lib = L.lib()
    _chk(ids, "embed_sum.ids")
    S = ids.shape[-1]
    ids2 = ids.reshape(-1, S).contiguous()
    out = torch.empty((ids2.shape[0], table.shape[1]), dtype=table.dtype, device=table.device)
    L.check(lib.afhip_embed_sum(L.ptr(ids2), L.ptr(table), L.ptr(out), ids2.shape[0], S, table.shape[1], table.shape[0],
                                L.dtype_code(table.dtype), L.stream_ptr()))
    return out.view(*ids.shape[:-1], table.shape[1])


def gather_rows(text_rows: torch.Tensor, audio_rows: torch.Tensor, plan: torch.Tensor, n_rows: int):
    """out[r] = text_rows[plan[r]] (plan >= 0) | audio_rows[-(plan[r] + 2)] (plan <= -2) | 0 (plan == -1); rows are [*, H]."""
    lib = L.lib()
    _chk(text_rows, "gather_rows.text_rows")
    assert text_rows.is_contiguous() and audio_rows.is_contiguous() and plan.dtype == torch.int32 and plan.is_contiguous()
    assert text_rows.dtype == audio_rows.dtype and text_rows.shape[-1] == audio_rows.shape[-1]
    H = text_rows.shape[-1]
    out = torch.empty((n_rows, H), dtype=text_rows.dtype, device=text_rows.device)
    L.check(lib.afhip_gather_rows(L.ptr(text_rows), L.ptr(audio_rows), L.ptr(plan), L.ptr(out), n_rows, text_rows.numel() // H,
                                  audio_rows.numel() // H, H * text_rows.element_size(), L.stream_ptr()))
    return out


def transpose_cast(x: torch.Tensor, out_dtype=None):
    """[B,R,C] -> [B,C,R] (+ dtype change)."""
    lib = L.lib()
    _chk(x, "transpose_cast.x")
    B, R, Cc = x.shape
    assert x.is_contiguous()
    out_dtype = out_dtype or x.dtype
    y = torch.empty((B, Cc, R), dtype=out_dtype, device=x.device)
    L.check(lib.afhip_transpose_cast(L.ptr(x), L.ptr(y), B, R, Cc, L.dtype_code(x.dtype), L.dtype_code(out_dtype), L.stream_ptr()))
    return y


def attention_packed(qkv: torch.Tensor, n_heads: int, key_len: Optional[torch.Tensor] = None, causal: bool = False, q_prescaled: bool = False,
                     out_fp8_scale: Optional[float] = None):
    """Self-attention on fused rows qkv [B,T,3*d] (q|k|v, heads contiguous inside each) -> [B,T,d].
    q_prescaled: the q columns already carry head_dim^-0.5 * log2(e) (afhip_attn_args.q_prescaled).
    out_fp8_scale = s: the output leaves as OCP e4m3 bytes (uint8 tensor) = sat(value / s) (afhip_attn_args.out_fp8; encoder form only)."""
    lib = L.lib()
    _chk(qkv, "attention.qkv")
    B, T, D3 = qkv.shape
    d = D3 // 3
    hd = d // n_heads
    assert qkv.is_contiguous()
    out = torch.empty((B, T, d), dtype=torch.uint8 if out_fp8_scale else qkv.dtype, device=qkv.device)
    sz = qkv.element_size()
    a = L.AttnArgs()
    a.q, a.k, a.v = qkv.data_ptr(), qkv.data_ptr() + d * sz, qkv.data_ptr() + 2 * d * sz
    a.out = out.data_ptr()
    if out_fp8_scale:
        a.out_fp8, a.out_scale_inv = 1, 1.0 / float(out_fp8_scale)
    a.key_len = key_len.data_ptr() if key_len is not None else None
    a.B, a.Tq, a.Tk, a.n_q, a.n_kv, a.hd = B, T, T, n_heads, n_heads, hd
    a.ld_q = a.ld_kv = D3
    a.ld_o = d
    a.q_batch_stride = a.kv_batch_stride = T * D3
    a.o_batch_stride = T * d
    a.q_head_stride = a.kv_head_stride = hd
    a.causal, a.q_pos0, a.scale, a.dtype = int(causal), 0, 1.0 / math.sqrt(hd), L.dtype_code(qkv.dtype)
    a.q_prescaled = int(q_prescaled)
    L.check(lib.afhip_attention(C.byref(a), L.stream_ptr()))
    return out


def attention_ragged(qkv_rows: torch.Tensor, n_heads: int, lengths: torch.Tensor, max_len: int, q_prescaled: bool = False):
    """Self-attention on PACKED fused rows qkv [sum(lengths), 3*d]: sequence b owns rows [off_b, off_b + lengths[b]) with
    off = exclusive cumsum(lengths) (afhip_attn_args.row_off) -> [sum(lengths), d]."""
    lib = L.lib()
    _chk(qkv_rows, "attention.qkv")
    R, D3 = qkv_rows.shape
    d = D3 // 3
    hd = d // n_heads
    assert qkv_rows.is_contiguous() and int(lengths.sum()) == R and int(lengths.max()) <= max_len
    key_len = lengths.to(device=qkv_rows.device, dtype=torch.int32).contiguous()
    row_off = (torch.cumsum(key_len, 0, dtype=torch.int32) - key_len).contiguous()
    out = torch.empty((R, d), dtype=qkv_rows.dtype, device=qkv_rows.device)
    sz = qkv_rows.element_size()
    a = L.AttnArgs()
    a.q, a.k, a.v = qkv_rows.data_ptr(), qkv_rows.data_ptr() + d * sz, qkv_rows.data_ptr() + 2 * d * sz
    a.out = out.data_ptr()
    a.key_len, a.row_off = key_len.data_ptr(), row_off.data_ptr()
    a.B, a.Tq, a.Tk, a.n_q, a.n_kv, a.hd = key_len.numel(), max_len, max_len, n_heads, n_heads, hd
    a.ld_q = a.ld_kv = D3
    a.ld_o = d
    a.q_head_stride = a.kv_head_stride = hd
    a.causal, a.q_pos0, a.scale, a.dtype = 0, 0, 1.0 / math.sqrt(hd), L.dtype_code(qkv_rows.dtype)
    a.q_prescaled = int(q_prescaled)
    L.check(lib.afhip_attention(C.byref(a), L.stream_ptr()))
    return out


def attention_cache(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, n_q: int, n_kv: int, tk: int,
                    q_pos0: int, ld_q: Optional[int] = None):
    """Causal GQA attention of q rows [B,Tq,(>=)n_q*hd] against a KV cache [B,n_kv,cap,hd] holding tk valid keys."""
    lib = L.lib()
    _chk(q, "attention.q")
    B, Tq = q.shape[0], q.shape[1]
    cap, hd = k_cache.shape[2], k_cache.shape[3]
    out = torch.empty((B, Tq, n_q * hd), dtype=q.dtype, device=q.device)
    a = L.AttnArgs()
    a.q, a.k, a.v, a.out = q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr()
    a.key_len = None
    a.B, a.Tq, a.Tk, a.n_q, a.n_kv, a.hd = B, Tq, tk, n_q, n_kv, hd
    a.ld_q = ld_q if ld_q is not None else q.stride(1)
    a.ld_kv, a.ld_o = hd, n_q * hd
    a.q_batch_stride = q.stride(0)
    a.kv_batch_stride = n_kv * cap * hd
    a.o_batch_stride = Tq * n_q * hd
    a.q_head_stride, a.kv_head_stride = hd, cap * hd
    a.causal, a.q_pos0, a.scale, a.dtype = 1, q_pos0, 1.0 / math.sqrt(hd), L.dtype_code(q.dtype)
    L.check(lib.afhip_attention(C.byref(a), L.stream_ptr()))
    return out


def rope_kv(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, pos0: int, k_cache: torch.Tensor, v_cache: torch.Tensor,
            n_q: int, n_kv: int):
    """In-place RoPE on q of qkv [B,T,(n_q+2n_kv)*hd]; rotated k and v appended to caches [B,n_kv,cap,hd] at pos0.."""
    lib = L.lib()
    _chk(qkv, "rope_kv.qkv")
    B, T, ld = qkv.shape
    cap, hd = k_cache.shape[2], k_cache.shape[3]
    assert qkv.is_contiguous() and cos.dtype == torch.float32 and cos.shape[1] == hd // 2
    L.check(lib.afhip_rope_kv(L.ptr(qkv), ld, L.ptr(cos), L.ptr(sin), pos0, L.ptr(k_cache), L.ptr(v_cache), B, T, n_q, n_kv, hd,
                              cap, cos.shape[0], L.dtype_code(qkv.dtype), L.stream_ptr()))
    return qkv


def attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, n_q: int, n_kv: int, tk: int,
                     key_split: int = 256, ld_q: Optional[int] = None, fused_rope: Optional[tuple] = None, in_launch_merge: bool = False):
    """One new token per sequence: q [B, >= n_q*hd] (row stride ld_q) against the first `tk` positions of a KV cache
    [B,n_kv,cap,hd].  The q heads of a kv group become the query rows of one workgroup and the context is split into
    `key_split`-key ranges merged by a second pass (flash-decoding).
    fused_rope = (qkv [B, (n_q + 2 n_kv) * hd] un-rotated projection output (q is its first n_q*hd columns), cos [hd/2], sin [hd/2]
    f32 rows of position tk-1): RoPE on q and k and the append of k, v at cache position tk-1 happen inside the launch.
    in_launch_merge: the key-range partials are merged by the last-arriving workgroup instead of a second pass."""
    lib = L.lib()
    _chk(q, "attention_decode.q")
    B = q.shape[0]
    cap, hd = k_cache.shape[2], k_cache.shape[3]
    rep = n_q // n_kv
    out = torch.empty((B, n_q * hd), dtype=q.dtype, device=q.device)
    n_split = (tk + key_split - 1) // key_split
    part = torch.empty(n_split * B * n_kv * 32 * (hd + 2), dtype=torch.float32, device=q.device)
    a = L.AttnArgs()
    a.q, a.k, a.v, a.out = q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr()
    a.key_len = None
    a.B, a.Tq, a.Tk, a.n_q, a.n_kv, a.hd = B, rep, tk, n_kv, n_kv, hd
    a.ld_q, a.ld_kv, a.ld_o = hd, hd, hd
    a.q_batch_stride = ld_q if ld_q is not None else q.stride(0)
    a.kv_batch_stride = n_kv * cap * hd
    a.o_batch_stride = n_q * hd
    a.q_head_stride, a.kv_head_stride, a.o_head_stride = rep * hd, cap * hd, rep * hd
    a.causal, a.q_pos0, a.scale, a.dtype = 0, 0, 1.0 / math.sqrt(hd), L.dtype_code(q.dtype)
    a.key_split, a.partial_ws, a.partial_ws_bytes = key_split, part.data_ptr(), part.numel() * 4
    if in_launch_merge:
        ticket = torch.zeros(B * n_kv, dtype=torch.int32, device=q.device)
        a.split_ticket = ticket.data_ptr()
    if fused_rope is not None:
        qkv, cos, sin = fused_rope
        assert qkv.is_contiguous() and cos.dtype == sin.dtype == torch.float32 and cos.numel() == hd // 2
        sz = qkv.element_size()
        a.new_k, a.new_v = qkv.data_ptr() + n_q * hd * sz, qkv.data_ptr() + (n_q + n_kv) * hd * sz
        a.new_kv_batch_stride = qkv.stride(0)
        a.rope_cos, a.rope_sin = cos.data_ptr(), sin.data_ptr()
    L.check(lib.afhip_attention(C.byref(a), L.stream_ptr()))
    return out
