"""Qwen2 decoder stack with KV cache (oracle; test infrastructure only) -- PyTorch-CPU fp32.

Restates third-party arithmetic the reference resolves dynamically at lm/parallel.py:44-48 and calls at
lm/parallel.py:582-586 (no attention_mask, no position_ids => causal, positions = cache positions):
  transformers 5.15.0 (reference pins >=4.57.1) models/qwen2/modeling_qwen2.py
  * 35-49 MLP (SwiGLU)   * 91-136 RoPE (rotate-half, f32 cos/sin)   * 150-173 eager attention (f32 softmax)
  * 195-235 attention (q/k/v bias, o no bias, GQA)   * 238-252 RMSNorm   * 258-299 decoder layer
State-dict names are the HF ones under the ParallelLLM prefix `model.` (lm/parallel.py:127).
"""

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


def config_7b():
    return dict(hidden_size=3584, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                intermediate_size=18944, rope_theta=1e6, rms_norm_eps=1e-6, text_vocab=152064)


def config_tiny():
    return dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, num_key_value_heads=2,
                intermediate_size=3072, rope_theta=1e6, rms_norm_eps=1e-6, text_vocab=16384)


def rmsnorm(x, w, eps):
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


Cache = List[Tuple[torch.Tensor, torch.Tensor]]


@torch.no_grad()
def forward(embeds: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: dict,
            cache: Optional[Cache] = None, last_layer_rows: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Cache]:
    """embeds [B,T,H] -> (final-normed hidden [B,T,H], new cache). cache[l] = (k,v) each [B,kvh,ctx,hd].

    last_layer_rows (LongTensor of positions, optional): in the LAST layer only these query positions are computed (keys and
    values still cover every position) and the result is [B, len(rows), H] -- the same arithmetic, restricted to a row sample,
    which keeps a 15 000-token check affordable on the CPU."""
    B, T, H = embeds.shape
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = H // nh
    past = 0 if not cache else cache[0][0].shape[2]
    pos = torch.arange(past, past + T)
    cos, sin = rope_cos_sin(pos, hd, cfg["rope_theta"])
    cos, sin = cos[None, None], sin[None, None]
    # causal mask over [T, past+T]
    kpos = torch.arange(past + T)
    allowed = kpos[None, :] <= pos[:, None]
    amask = torch.zeros(T, past + T).masked_fill_(~allowed, float("-inf"))
    x = embeds
    new_cache: Cache = []
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        h = rmsnorm(x, sd[p + "input_layernorm.weight"], cfg["rms_norm_eps"])
        sub = last_layer_rows if (last_layer_rows is not None and l == cfg["num_hidden_layers"] - 1) else None
        Tq = T if sub is None else sub.numel()
        hq = h if sub is None else h[:, sub]
        q = F.linear(hq, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(B, Tq, nh, hd).transpose(1, 2)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(B, T, nkv, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(B, T, nkv, hd).transpose(1, 2)
        if sub is None:
            q = q * cos + _rot_half(q) * sin
        else:
            q = q * cos[:, :, sub] + _rot_half(q) * sin[:, :, sub]
        k = k * cos + _rot_half(k) * sin
        if cache:
            k = torch.cat([cache[l][0], k], dim=2)
            v = torch.cat([cache[l][1], v], dim=2)
        new_cache.append((k, v))
        rep = nh // nkv
        kk = k[:, :, None].expand(B, nkv, rep, past + T, hd).reshape(B, nh, past + T, hd)
        vv = v[:, :, None].expand(B, nkv, rep, past + T, hd).reshape(B, nh, past + T, hd)
        s = torch.matmul(q, kk.transpose(2, 3)) * (hd ** -0.5) + (amask if sub is None else amask[sub])
        pr = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
        o = torch.matmul(pr, vv).transpose(1, 2).reshape(B, Tq, H)
        x = (x if sub is None else x[:, sub]) + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        h = rmsnorm(x, sd[p + "post_attention_layernorm.weight"], cfg["rms_norm_eps"])
        g = F.linear(h, sd[p + "mlp.gate_proj.weight"])
        u = F.linear(h, sd[p + "mlp.up_proj.weight"])
        x = x + F.linear(F.silu(g) * u, sd[p + "mlp.down_proj.weight"])
    return rmsnorm(x, sd["model.norm.weight"], cfg["rms_norm_eps"]), new_cache


def params_per_layer(cfg):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    hd = H // cfg["num_attention_heads"]
    kv = cfg["num_key_value_heads"] * hd
    return H * H + H + 2 * (H * kv + kv) + H * H + 3 * H * I + 2 * H
