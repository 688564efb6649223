"""Seeded random-init weights of the AF-Whisper / Qwen2 / UALM architectures.

There are no checkpoints offline, so benchmarks and parity tests run on random weights of the
true shapes.  Every tensor is generated from its own seed (hash of seed + tensor name), so a
single tensor can be regenerated anywhere -- in the build container when golden vectors are
captured from the reference, and on the GPU host -- without any weight file travelling.

Key names are the reference's state-dict names (modeling_whisper.py:132-135,463-469,614-621;
lm/parallel.py:93-96,127-128,138,146-149; HF Qwen2 names under `model.`).
"""

import hashlib
from typing import Dict, List, Tuple

import torch


def encoder_param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    d, f, m = cfg["d_model"], cfg["encoder_ffn_dim"], cfg["num_mel_bins"]
    out = [("conv1.weight", (d, m, 3)), ("conv1.bias", (d,)),
           ("conv2.weight", (d, d, 3)), ("conv2.bias", (d,)),
           ("embed_positions.weight", (cfg["max_source_positions"], d))]
    for i in range(cfg["encoder_layers"]):
        p = f"layers.{i}."
        out += [(p + "self_attn.k_proj.weight", (d, d)),
                (p + "self_attn.v_proj.weight", (d, d)), (p + "self_attn.v_proj.bias", (d,)),
                (p + "self_attn.q_proj.weight", (d, d)), (p + "self_attn.q_proj.bias", (d,)),
                (p + "self_attn.out_proj.weight", (d, d)), (p + "self_attn.out_proj.bias", (d,)),
                (p + "self_attn_layer_norm.weight", (d,)), (p + "self_attn_layer_norm.bias", (d,)),
                (p + "fc1.weight", (f, d)), (p + "fc1.bias", (f,)),
                (p + "fc2.weight", (d, f)), (p + "fc2.bias", (d,)),
                (p + "final_layer_norm.weight", (d,)), (p + "final_layer_norm.bias", (d,))]
    out += [("layer_norm.weight", (d,)), ("layer_norm.bias", (d,))]
    return out


def llm_param_shapes(cfg: dict, vocab_size: int, num_stream: int, enc_dim: int) -> List[Tuple[str, Tuple[int, ...]]]:
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    hd = H // cfg["num_attention_heads"]
    kv = cfg["num_key_value_heads"] * hd
    out = [("model.embed_tokens.weight", (vocab_size, H))]
    for l in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{l}."
        out += [(p + "self_attn.q_proj.weight", (H, H)), (p + "self_attn.q_proj.bias", (H,)),
                (p + "self_attn.k_proj.weight", (kv, H)), (p + "self_attn.k_proj.bias", (kv,)),
                (p + "self_attn.v_proj.weight", (kv, H)), (p + "self_attn.v_proj.bias", (kv,)),
                (p + "self_attn.o_proj.weight", (H, H)),
                (p + "mlp.gate_proj.weight", (I, H)), (p + "mlp.up_proj.weight", (I, H)),
                (p + "mlp.down_proj.weight", (H, I)),
                (p + "input_layernorm.weight", (H,)), (p + "post_attention_layernorm.weight", (H,))]
    out += [("model.norm.weight", (H,)), ("lm_head.weight", (vocab_size, H)),
            ("stream_emb.weight", (num_stream, H)),
            ("adaptor.continuous_audio.weight", (H, enc_dim)), ("adaptor.continuous_audio.bias", (H,))]
    return out


def _seed_for(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def _is_norm_gain(name: str) -> bool:
    return name.endswith("weight") and ("layer_norm" in name or "layernorm" in name or name.endswith("norm.weight"))


def synth_tensor(name: str, shape, seed: int, dtype=torch.float32, device="cpu") -> torch.Tensor:
    """One tensor: N(0,0.02) for matrices/biases (0.06 for q/k projections so attention is peaked
    enough to expose masking / position errors), N(0,1) for token and stream embeddings (the
    reference's own init, lm/parallel.py:93,138), N(0,1/fan_in) for adaptors, 1+N(0,0.1) for norm gains.

    Generated in fp32 on `device` from a generator seeded per (seed, name).  CPU and GPU generators
    give different streams: parity tests generate on CPU and copy; throughput runs may generate on GPU."""
    g = torch.Generator(device=device)
    g.manual_seed(_seed_for(seed, name))
    t = torch.randn(tuple(shape), generator=g, device=device, dtype=torch.float32)
    if _is_norm_gain(name):
        t = 1.0 + 0.1 * t
    elif name in ("model.embed_tokens.weight", "stream_emb.weight"):
        pass
    elif name.startswith("adaptor.") and name.endswith("weight"):
        t = t * (float(shape[-1]) ** -0.5)
    elif name.endswith("q_proj.weight") or name.endswith("k_proj.weight"):
        t = 0.06 * t
    else:
        t = 0.02 * t
    if name in ("model.embed_tokens.weight", "lm_head.weight"):
        t[0] = 0.0  # padding row (lm/parallel.py:93-96)
    return t.to(dtype)


def synth_state_dict(shapes, seed: int, dtype=torch.float32, device="cpu") -> Dict[str, torch.Tensor]:
    return {n: synth_tensor(n, s, seed, dtype, device) for n, s in shapes}


def make_wav(seed: int, n: int):
    """Synthetic 16 kHz mono clip: numpy default_rng(seed).standard_normal(n) * 0.1, float32."""
    import numpy as np
    return (np.random.default_rng(seed).standard_normal(n) * 0.1).astype(np.float32)


def make_prompt(text_vocab: int, n: int = 16, seed: int = 7):
    """Synthetic text ids in [1, text_vocab) (no tokenizer offline)."""
    import numpy as np
    return np.random.default_rng(seed).integers(1, text_vocab, size=n).tolist()


def make_offline_xcodec(seed: int = 1234):
    """The X-codec network of the audio-output side with seeded random weights: `transformers.XcodecModel(XcodecConfig())` -- the class
    the reference loads by tag (multimodal_io/audio.py:203-218), constructed offline from its default config (16 kHz, hop 320, 8 x 1024
    codebooks).  The same call in the fixture script (oracle/make_golden_codec.py) and in the tests gives the same parameters as long as
    torch / transformers are those of the image; `xcodec_fingerprint` detects a drift."""
    import transformers
    torch.manual_seed(seed)
    codec = transformers.XcodecModel(transformers.XcodecConfig()).eval()
    # the residual-VQ codebooks are BUFFERS the constructor leaves at zero (they are learnt by k-means / EMA): with them at zero every
    # frame encodes to entry 0 and the decoder ignores the codes.  Seeded entries make both directions depend on the codes.
    g = torch.Generator().manual_seed(seed + 1)
    for name, buf in codec.named_buffers():
        if name.endswith("codebook.embed") or name.endswith("codebook.embed_avg"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.5)
        elif name.endswith("codebook.inited"):
            buf.fill_(1)
        elif name.endswith("codebook.cluster_size"):
            buf.fill_(1.0)
    return codec


def xcodec_fingerprint(codec) -> float:
    """sum of |parameter| in float64: equal fingerprints <=> (practically) equal seeded weights"""
    return float(sum(p.detach().double().abs().sum() for p in codec.parameters()) + sum(b.detach().double().abs().sum() for b in codec.buffers()))
