"""AF-Whisper encoder (oracle; test infrastructure only) -- PyTorch-CPU fp32 restatement.

Follows /root/reference/UALM/models/ualm/multimodal_io:
  * modeling_whisper.py:603-627  module layout / state-dict names
  * modeling_whisper.py:640-756  AFWhisperEncoder.forward
  * modeling_whisper.py:471-519  Qwen2AudioEncoderLayer.forward (pre-LN)
  * modeling_whisper.py:141-218 / 346-441  attention (q,v,out bias; k no bias; scale 1/sqrt(64))
  * audio.py:1103-1187           ContinuousAudioIO.encode_batch (mask from lengths, per-clip trim)
  * sound_encoder.py:81-107      SoundTower.forward (window stack, untrimmed)
Weights arrive as a plain dict of tensors keyed with the reference's state-dict names.
"""

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from .lengths import encode_batch_lengths, feat_extract_output_lengths


def default_config(**kw):
    cfg = dict(num_mel_bins=128, d_model=1280, encoder_attention_heads=20, encoder_ffn_dim=5120,
               encoder_layers=32, max_source_positions=1500)
    cfg.update(kw)
    return cfg


def tiny_config():
    return default_config(d_model=384, encoder_attention_heads=6, encoder_ffn_dim=1536, encoder_layers=2)


def _attention(x, sd, pre, n_heads, key_len, sdpa=True):
    """modeling_whisper.py:141-218 (eager) / 346-441 (sdpa). key_len: LongTensor[B] or None."""
    B, T, D = x.shape
    hd = D // n_heads
    q = F.linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"])
    k = F.linear(x, sd[pre + "k_proj.weight"])                      # no bias (:132)
    v = F.linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])
    q = q.view(B, T, n_heads, hd).transpose(1, 2)
    k = k.view(B, T, n_heads, hd).transpose(1, 2)
    v = v.view(B, T, n_heads, hd).transpose(1, 2)
    mask = None
    if key_len is not None:
        ar = torch.arange(T)
        mask = torch.zeros(B, 1, 1, T, dtype=x.dtype)
        mask.masked_fill_((ar[None, :] >= key_len[:, None])[:, None, None, :], float("-inf"))
        mask = mask.expand(B, 1, T, T)
    if sdpa:
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    else:
        s = torch.matmul(q * (hd ** -0.5), k.transpose(2, 3))
        if mask is not None:
            s = s + mask
        o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.transpose(1, 2).reshape(B, T, D)
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def conv_stem(mel_bct: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """modeling_whisper.py:690-696 -> [B,1500,d] (GELU erf, + learned positions)."""
    h = F.gelu(F.conv1d(mel_bct, sd["conv1.weight"], sd["conv1.bias"], padding=1))
    h = F.gelu(F.conv1d(h, sd["conv2.weight"], sd["conv2.bias"], stride=2, padding=1))
    return h.permute(0, 2, 1) + sd["embed_positions.weight"]


def encoder_layer(h, sd, i, n_heads, key_len, sdpa=True):
    """modeling_whisper.py:471-519."""
    p = f"layers.{i}."
    a = F.layer_norm(h, (h.shape[-1],), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
    h = h + _attention(a, sd, p + "self_attn.", n_heads, key_len, sdpa)
    f = F.layer_norm(h, (h.shape[-1],), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
    f = F.gelu(F.linear(f, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return h + F.linear(f, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


@torch.no_grad()
def encoder_forward(mel_bct: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: dict,
                    feat_len=None, sdpa: bool = True, return_states: bool = False):
    """AFWhisperEncoder.forward. mel_bct [B,128,3000]; feat_len LongTensor[B] (keys >= feat_len get
    -inf; None or >= 1500 means no masking). Returns [B,750,d] (and per-stage states)."""
    if mel_bct.shape[-1] != cfg["max_source_positions"] * 2:
        raise ValueError(f"expected mel length {cfg['max_source_positions'] * 2}, found {mel_bct.shape[-1]}")
    key_len = None
    if feat_len is not None:
        fl = torch.as_tensor(feat_len, dtype=torch.long)
        if bool((fl < cfg["max_source_positions"]).any()):
            key_len = fl
    h = conv_stem(mel_bct.float(), sd)
    states = [h]
    for i in range(cfg["encoder_layers"]):
        h = encoder_layer(h, sd, i, cfg["encoder_attention_heads"], key_len, sdpa)
        states.append(h)
    h = F.avg_pool1d(h.permute(0, 2, 1), 2, 2).permute(0, 2, 1)
    out = F.layer_norm(h, (h.shape[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    if return_states:
        return out, states
    return out


@torch.no_grad()
def encode_batch(feats_btc: torch.Tensor, length: torch.Tensor, sd, cfg, sdpa=True) -> List[torch.Tensor]:
    """ContinuousAudioIO.encode_batch (audio.py:1103-1187): [B,3000,128], length[B] -> list of [out_i,d]."""
    mel = feats_btc.transpose(1, 2)
    fl, ol = [], []
    for L in length.tolist():
        f, o = encode_batch_lengths(int(L))
        fl.append(f)
        ol.append(o)
    out = encoder_forward(mel, sd, cfg, feat_len=torch.tensor(fl), sdpa=sdpa)
    return [out[i, : ol[i]] for i in range(out.shape[0])]


@torch.no_grad()
def sound_tower(sounds: torch.Tensor, mask: torch.Tensor, sd, cfg, sdpa=True) -> torch.Tensor:
    """SoundTower.forward tensor branch (sound_encoder.py:81-107): [1,W,1,128,3000] + mask [1,W,1,3000]
    (or already squeezed [W,128,3000] / [W,1,3000]) -> [W,750,d], untrimmed."""
    if sounds.dim() == 5:
        sounds = sounds.squeeze(0).squeeze(1)
        mask = mask.squeeze(0)
    n = mask.sum(-1).reshape(-1)
    fl = [feat_extract_output_lengths(int(v))[0] for v in n.tolist()]
    return encoder_forward(sounds, sd, cfg, feat_len=torch.tensor(fl), sdpa=sdpa)


def flops_per_clip(cfg: dict) -> float:
    """Algorithmic FLOPs of one 30-s clip: stem + layers (QKVO + attention + FFN)."""
    d, f, L, T = cfg["d_model"], cfg["encoder_ffn_dim"], cfg["encoder_layers"], cfg["max_source_positions"]
    stem = 2 * 3000 * d * (3 * cfg["num_mel_bins"]) + 2 * T * d * (3 * d)
    per_layer = 2 * T * d * d * 4 + 4 * T * T * d + 2 * T * d * f * 2
    return float(stem + L * per_layer)
