"""ParallelLLM on MI355X: the UALM multi-stream LLM wrapper (Qwen2 backbone) over the HIP library.

Host-side mirror of `ParallelHFModel` / `ParallelLLM` (UALM/models/ualm/lm/parallel.py:17-646) for the
audio-understanding inference path: same factory arguments, same state-dict keys (HF Qwen2 names under `model.`,
`lm_head`, `stream_emb`, `adaptor.<io>`, `multimodal_io_dict.<io>.model.*`), same methods (`prepare_inference`,
`inference`, `inference_segment`, `_embed`, `_step`, `_logits_to_token`) and buffers.  All arithmetic runs in
csrc/ (llm.hip); the KV cache is a preallocated [layer,B,kv_head,cap,hd] buffer instead of DynamicCache, and
greedy decode keeps token selection and the eos/eot bookkeeping on the device (no per-token host sync).
Training (`forward` / `_loss`, lm/parallel.py:176-217,286-384) is out of scope.
"""
import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..multimodal_io.modeling_whisper import _Linear, _Norm, _Embedding


def ParallelHFModel(model_hf_tag, **kwargs):
    """Factory with the reference's signature (lm/parallel.py:17-28); `model_hf_tag` is a local directory."""
    return ParallelLLM.from_pretrained(model_hf_tag, **kwargs)


class KVCache:
    """Preallocated key/value store [n_layers, B, n_kv, cap, hd]; `length` = valid positions."""

    def __init__(self, n_layers, B, n_kv, cap, hd, dtype, device):
        self.k = torch.zeros((n_layers, B, n_kv, cap, hd), dtype=dtype, device=device)
        self.v = torch.zeros_like(self.k)
        self.length = 0

    @property
    def cap(self):
        return self.k.shape[3]

    @property
    def batch(self):
        return self.k.shape[1]

    def get_seq_length(self):
        return self.length

    def reserve(self, cap):
        if cap <= self.cap:
            return
        cap = (cap + 63) // 64 * 64
        for name in ("k", "v"):
            old = getattr(self, name)
            new = torch.zeros(old.shape[:3] + (cap, old.shape[4]), dtype=old.dtype, device=old.device)
            new[:, :, :, : self.length] = old[:, :, :, : self.length]
            setattr(self, name, new)

    def batch_select_indices(self, indices):
        self.k = self.k[:, indices].contiguous()
        self.v = self.v[:, indices].contiguous()

    def struct(self):
        s = L.KvCache()
        s.k, s.v, s.cap, s.B = self.k.data_ptr(), self.v.data_ptr(), self.cap, self.batch
        return s


class _SelfAttn(nn.Module):
    def __init__(self, H, nq, nkv, hd):
        super().__init__()
        self.q_proj = _Linear(H, nq * hd)
        self.k_proj = _Linear(H, nkv * hd)
        self.v_proj = _Linear(H, nkv * hd)
        self.o_proj = _Linear(nq * hd, H, bias=False)


class _MLP(nn.Module):
    def __init__(self, H, I):
        super().__init__()
        self.gate_proj = _Linear(H, I, bias=False)
        self.up_proj = _Linear(H, I, bias=False)
        self.down_proj = _Linear(I, H, bias=False)


class _DecoderLayer(nn.Module):
    def __init__(self, H, nq, nkv, hd, I):
        super().__init__()
        self.self_attn = _SelfAttn(H, nq, nkv, hd)
        self.mlp = _MLP(H, I)
        self.input_layernorm = _Norm(H, bias=False)
        self.post_attention_layernorm = _Norm(H, bias=False)


class _Backbone(nn.Module):
    def __init__(self, cfg, vocab_size):
        super().__init__()
        H = cfg["hidden_size"]
        nq, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
        hd = cfg.get("head_dim") or H // nq
        self.embed_tokens = _Embedding(vocab_size, H)
        self.layers = nn.ModuleList([_DecoderLayer(H, nq, nkv, hd, cfg["intermediate_size"]) for _ in range(cfg["num_hidden_layers"])])
        self.norm = _Norm(H, bias=False)


from ..utils.quant import quantize_rows_e4m3 as _quantize_rows_e4m3  # noqa: E402
from .. import torch_ops  # noqa: E402,F401  (registers torch.ops.afhip.*: the forwards below run through them)


def pack_qwen2_weights(backbone: "_Backbone", lm_head_w: torch.Tensor, stream_emb_w: Optional[torch.Tensor], cfg: dict, n_stream: int,
                       fp8: bool = False, max_positions: Optional[int] = None, fp8_prefill: bool = False):
    """Build `afhip_llm_weights` for a Qwen2 decoder stack held in HF parameter layout: q|k|v fused with biases, gate/up
    interleaved in 32-row blocks (SwiGLU pairs land in one wave's accumulators), RoPE tables exactly as transformers computes
    them (modeling_qwen2.py:91-121), optional e4m3 copies of the weights a decode step streams.  Keeps every tensor alive in
    the returned namespace.  Shared by ParallelLLM (UALM) and Qwen2AudioForConditionalGeneration (AF3)."""
    dev, dt = lm_head_w.device, lm_head_w.dtype
    if dev.type != "cuda":
        raise L.AfhipError("the LLM runs on the GPU only: call .to('cuda') first (no CPU fallback)")
    H, nq, nkv = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = cfg.get("head_dim") or H // nq
    I = cfg["intermediate_size"]
    max_pos = max(max_positions or 0, 4096)
    keep = []

    def P(t):
        t = t.detach().contiguous()
        keep.append(t)
        return t

    w = L.LlmWeights()
    w.hidden, w.n_layers, w.n_q, w.n_kv, w.hd, w.inter = H, cfg["num_hidden_layers"], nq, nkv, hd, I
    w.vocab, w.n_stream, w.rms_eps, w.dtype = lm_head_w.shape[0], n_stream, float(cfg.get("rms_norm_eps", 1e-6)), L.dtype_code(dt)
    w.embed = P(backbone.embed_tokens.weight).data_ptr()
    names = ["ln1_w", "qkv_w", "qkv_b", "o_w", "ln2_w", "gu_w", "down_w"]
    lists = {n: [] for n in names}
    for lyr in backbone.layers:
        a, m = lyr.self_attn, lyr.mlp
        lists["ln1_w"].append(P(lyr.input_layernorm.weight))
        lists["qkv_w"].append(P(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0)))
        lists["qkv_b"].append(P(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], dim=0)))
        lists["o_w"].append(P(a.o_proj.weight))
        lists["ln2_w"].append(P(lyr.post_attention_layernorm.weight))
        g, u = m.gate_proj.weight, m.up_proj.weight
        lists["gu_w"].append(P(torch.stack([g.view(I // 32, 32, H), u.view(I // 32, 32, H)], dim=1).reshape(2 * I, H)))
        lists["down_w"].append(P(m.down_proj.weight))
    arrays = {}
    for n in names:
        arrays[n] = L.ptr_array(lists[n])
        setattr(w, n, C.cast(arrays[n], L.c_void_pp))
    if fp8:
        for src, dst_w, dst_s in (("qkv_w", "qkv_w8", "qkv_s"), ("o_w", "o_w8", "o_s"), ("gu_w", "gu_w8", "gu_s"), ("down_w", "down_w8", "down_s")):
            qs, ss = [], []
            for t in lists[src]:
                q8, sc = _quantize_rows_e4m3(t)
                qs.append(P(q8))
                ss.append(P(sc))
            arrays[dst_w], arrays[dst_s] = L.ptr_array(qs), L.ptr_array(ss)
            setattr(w, dst_w, C.cast(arrays[dst_w], L.c_void_pp))
            setattr(w, dst_s, C.cast(arrays[dst_s], L.c_void_pp))
        q8, sc = _quantize_rows_e4m3(lm_head_w.detach())
        w.lm_head8, w.lm_head_s = P(q8).data_ptr(), P(sc).data_ptr()
        w.fp8_prefill = 1 if fp8_prefill else 0
    w.norm_w = P(backbone.norm.weight).data_ptr()
    w.lm_head = P(lm_head_w).data_ptr()
    w.stream_emb = P(stream_emb_w).data_ptr() if stream_emb_w is not None else None
    # rotary tables exactly as transformers computes them (modeling_qwen2.py:91-103,110-121): f32 on the host
    theta = float(cfg.get("rope_theta", 10000.0))
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    fr = torch.arange(max_pos, dtype=torch.float)[:, None] * inv[None, :]
    cos, sin = P(fr.cos().to(dev)), P(fr.sin().to(dev))
    w.rope_cos, w.rope_sin, w.rope_max_pos = cos.data_ptr(), sin.data_ptr(), max_pos
    from .. import torch_ops
    return SimpleNamespace(w=w, keep=keep, arrays=arrays, max_pos=max_pos, hd=hd, nq=nq, nkv=nkv, blob=torch_ops.weights_blob(w))


class ParallelLLM(nn.Module):
    """Parallel multimodal LLM supporting multi-stream token processing (inference path)."""

    def __init__(self, config: dict, multimodal_io: dict, vocab: List[str], vocab_intervals: Dict[str, list]):
        super().__init__()
        if config.get("architectures", ["Qwen2ForCausalLM"])[0] != "Qwen2ForCausalLM":
            raise NotImplementedError(f"architecture {config.get('architectures')} not supported (Qwen2ForCausalLM only)")
        self.config = SimpleNamespace(**config)
        self.cfg = config
        vocab_size = max(end for iv in vocab_intervals.values() for _, end in iv)      # lm/parallel.py:84-90
        H = config["hidden_size"]
        self.model = _Backbone(config, vocab_size)
        self.lm_head = _Linear(H, vocab_size, bias=False)
        streams = [io.num_stream() for io in multimodal_io.values() if io.is_discrete]
        if len(streams) == 0:
            raise ValueError("Cannot proceed with all IOs being continuous")
        self.num_stream = max(streams)
        self.stream_emb = _Embedding(self.num_stream, H)
        self.multimodal_io_dict = nn.ModuleDict(multimodal_io)
        self.adaptor = nn.ModuleDict()
        for io_name, io in self.multimodal_io_dict.items():
            if not io.is_discrete:
                self.adaptor[io_name] = _Linear(io.feature_dim(), H)
        self.vocab = vocab
        self.vocab_intervals = vocab_intervals
        self._packed = None
        self._ws = None
        self._allowed = {}
        self._allowed_hi = {}
        self._fp8_decode = False
        self._fp8_prefill = False

    _quantize_rows_e4m3 = staticmethod(_quantize_rows_e4m3)

    def enable_fp8_decode(self, on: bool = True):
        """W8A16 decode (BASELINE config 5): keep an OCP-e4m3 copy (+ one f32 scale per output row) of every weight the
        decode step streams -- q/k/v, o, gate/up, down, lm_head -- and let the skinny GEMMs read those (half the HBM bytes
        per token).  Prefill keeps the bf16 weights.  bf16 models only."""
        if on and self.dtype != torch.bfloat16:
            raise L.AfhipError("fp8 decode weights need a bfloat16 model")
        self._fp8_decode = bool(on)
        if not on:
            self._fp8_prefill = False
        self._packed = None
        return self

    def enable_fp8(self, on: bool = True):
        """BASELINE config 5 end to end on the LLM side: W8A16 decode (enable_fp8_decode) AND e4m3 x e4m3 MFMA GEMMs for the four
        projections of every layer in prefill (activations quantised per row, RMSNorm fused into that pass; f32 accumulate)."""
        self.enable_fp8_decode(on)
        self._fp8_prefill = bool(on)
        self._packed = None
        return self

    # ---------------------------------------------------------------- construction
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, multimodal_io, vocab, vocab_intervals,
                        max_loss_interval: int = 13192, **kwargs):
        """lm/parallel.py:51-174.  Reads config.json from a local directory; if base-LLM weights are present
        (model.safetensors with HF Qwen2 names) the text rows of the rebuilt embedding / lm_head tables are
        filled from them (:98-128); everything else is then expected from the UALM checkpoint
        (`load_state_dict(torch.load(...)["module"], strict=True)`, scripts/inference.py:150-152)."""
        with open(os.path.join(pretrained_model_name_or_path, "config.json")) as f:
            cfg = json.load(f)
        if "rope_theta" not in cfg and isinstance(cfg.get("rope_parameters"), dict):
            cfg["rope_theta"] = cfg["rope_parameters"].get("rope_theta", 10000.0)
        model = cls(cfg, multimodal_io, vocab, vocab_intervals)
        with torch.no_grad():   # defaults for anything a checkpoint does not overwrite (lm/parallel.py:93-96,138)
            for n, p in model.named_parameters():
                if n.startswith("multimodal_io_dict."):
                    continue
                if n.endswith("bias"):
                    p.zero_()
                elif p.dim() == 1:
                    p.fill_(1.0)
                elif n in ("model.embed_tokens.weight", "stream_emb.weight"):
                    p.normal_(0.0, 1.0)
                else:
                    p.normal_(0.0, 0.02)
            model.model.embed_tokens.weight[0] = 0.0
            model.lm_head.weight[0] = 0.0
        st = os.path.join(pretrained_model_name_or_path, "model.safetensors")
        if os.path.exists(st) and "text" in vocab_intervals:
            from safetensors.torch import load_file
            base = load_file(st)
            ts, te = vocab_intervals["text"][0]
            if te - ts != base["model.embed_tokens.weight"].shape[0]:
                raise ValueError(f"text_end - text_start ({te - ts}) must equal original vocab size ({base['model.embed_tokens.weight'].shape[0]})")
            own = model.state_dict()
            with torch.no_grad():
                for k, v in base.items():
                    if k == "model.embed_tokens.weight":
                        own[k][ts:te] = v
                    elif k == "lm_head.weight":
                        own[k][ts:te] = v
                    elif k in own:
                        own[k].copy_(v)
        dt = kwargs.get("dtype", kwargs.get("torch_dtype"))
        if dt is not None:
            model = model.to(dt)
        return model

    def _apply(self, fn, *a, **kw):
        self._packed = None
        self._ws = None
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed = None            # fires on every (recursive) load; nested encoders invalidate themselves the same way
        return super()._load_from_state_dict(*a, **kw)

    @property
    def dtype(self):
        return self.lm_head.weight.dtype

    @property
    def device(self):
        return self.lm_head.weight.device

    # ---------------------------------------------------------------- packing for the HIP library
    def pack(self, max_positions: Optional[int] = None):
        if self._packed is not None and (max_positions is None or max_positions <= self._packed.max_pos):
            return self._packed
        self._packed = pack_qwen2_weights(self.model, self.lm_head.weight, self.stream_emb.weight, self.cfg, self.num_stream,
                                          fp8=self._fp8_decode, max_positions=max_positions, fp8_prefill=self._fp8_prefill)
        return self._packed

    def _workspace(self, B, T, max_ctx=0):
        lib = L.lib()
        need = lib.afhip_llm_workspace_bytes(C.byref(self.pack().w), B, T, max_ctx)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def new_cache(self, B, cap):
        pk = self.pack(cap)
        return KVCache(self.cfg["num_hidden_layers"], B, pk.nkv, (cap + 63) // 64 * 64, pk.hd, self.dtype, self.device)

    # ---------------------------------------------------------------- inference logic (lm/parallel.py:386-644)
    def prepare_inference(self):
        """lm/parallel.py:535-568: special-token tensors, modality mask, per-IO restricted-decoding masks."""
        for token in ["assistant", "audio", "text", "eos", "eot"]:
            token_tensor = torch.zeros((1, 1, self.num_stream)).long()
            token_tensor[0, 0, 0] = self.vocab.index(f"<|{token}|>")
            self.register_buffer(f"{token}_token", token_tensor.to(self.device))
        mask = torch.ones(self.num_stream, len(self.vocab)).bool()
        for token in ["audio", "text", "image", "video", "toolcall"]:
            mask[0, self.vocab.index(f"<|{token}|>")] = False
        mask[1:, 0] = False
        self.register_buffer("modality_mask", mask[None, None].to(self.device))
        self.eot_token_id = self.vocab.index("<|eot|>")
        self.eos_token_id = self.vocab.index("<|eos|>")
        for io_name, intervals in self.vocab_intervals.items():
            mask = torch.ones(self.num_stream, len(self.vocab)).bool()
            for idx, (start, end) in enumerate(intervals):
                mask[idx, start:end] = False
            for idx in range(len(intervals), self.num_stream):
                mask[idx, 0] = False
            mask[0, self.eot_token_id] = False
            mask[0, self.eos_token_id] = False
            io_name = "audio" if io_name == "discrete_audio" else io_name
            self.register_buffer(f"{io_name}_mask", mask[None, None].to(self.device))
        self._allowed = {}
        self._allowed_hi = {}

    def _allowed_intervals(self, name: str):
        """Stream-0 allowed id runs of a mask buffer as a device int32 [n,2] tensor, plus whether every other
        stream may only emit pad (then the device-side greedy loop is exact)."""
        if name not in self._allowed:
            m = getattr(self, f"{name}_mask")[0, 0].cpu()
            ok = (~m[0]).to(torch.int8)
            d = torch.diff(torch.cat([torch.zeros(1, dtype=torch.int8), ok, torch.zeros(1, dtype=torch.int8)]))
            starts, ends = (d == 1).nonzero().flatten(), (d == -1).nonzero().flatten()
            iv = torch.stack([starts, ends], dim=1).to(torch.int32)
            pad_only = bool(((~m[1:]).sum(-1) == 1).all() and (~m[1:, 0]).all()) if self.num_stream > 1 else True
            self._allowed[name] = (iv.to(self.device).contiguous(), pad_only)
            self._allowed_hi[name] = int(ends.max()) if len(ends) else 0
        return self._allowed[name]

    def _stream_intervals(self, name: str):
        """Allowed id runs of EVERY stream of a mask buffer as a device int32 [S, n_iv, 2] table (unused slots lo == hi): the form
        afhip_sample_topk takes, one row of the modality mask (lm/parallel.py:557-568) per stream."""
        key = ("streams", name)
        if key not in self._allowed:
            m = getattr(self, f"{name}_mask")[0, 0].cpu()
            runs = []
            for srow in m:
                ok = (~srow).to(torch.int8)
                d = torch.diff(torch.cat([torch.zeros(1, dtype=torch.int8), ok, torch.zeros(1, dtype=torch.int8)]))
                runs.append(list(zip((d == 1).nonzero().flatten().tolist(), (d == -1).nonzero().flatten().tolist())))
            n_iv = max(1, max(len(r) for r in runs))
            t = torch.zeros((len(runs), n_iv, 2), dtype=torch.int32)
            for i, r in enumerate(runs):
                for j, (lo, hi) in enumerate(r):
                    t[i, j, 0], t[i, j, 1] = lo, hi
            self._allowed[key] = t.to(self.device).contiguous()
        return self._allowed[key]

    @torch.no_grad()
    def _embed(self, input_ids, kwargs):
        """lm/parallel.py:219-284: stream-summed token embeddings, continuous features spliced in through the adaptor."""
        # (1) discrete modalities handed over as raw features: tokenise on the fly and place the ids (lm/parallel.py:233-257).  The IO
        #     needs a tokeniser of its own (DiscreteAudioTokenIO.attach_codec); without one its encode_batch raises.
        for io_name, io in self.multimodal_io_dict.items():
            if not io.is_discrete:
                continue
            if f"{io_name}_indices" not in kwargs or f"{io_name}_feats" not in kwargs or f"{io_name}_lengths" not in kwargs:
                continue
            codes = io.encode_batch(kwargs[f"{io_name}_feats"], kwargs[f"{io_name}_lengths"])
            codes = codes + self.vocab_intervals[io_name][0][0]
            input_ids = input_ids.clone()
            for code, (bidx, start, length) in zip(codes, kwargs[f"{io_name}_indices"].tolist()):
                input_ids[bidx, start: start + length] = code[:length].to(input_ids.device)
        input_embeds = ops.embed_sum(input_ids.to(self.device), self.model.embed_tokens.weight)
        for io_name, io in self.multimodal_io_dict.items():
            if io.is_discrete:
                continue
            if f"{io_name}_indices" not in kwargs:
                continue
            if f"{io_name}_encoded" in kwargs:
                # long-audio path (long_audio.py): the encoder already ran, sharded over the GPUs of the node, and its tokens came
                # back by the all-gather -- [n, T_w, d] tensor or a list of [len_i, d]; entry i still lands at (b, start, len)
                feats = kwargs[f"{io_name}_encoded"]
            elif f"{io_name}_feats" in kwargs and f"{io_name}_lengths" in kwargs:
                feats = io.encode_batch(kwargs[f"{io_name}_feats"], kwargs[f"{io_name}_lengths"])
            else:
                continue
            ad = self.adaptor[io_name]
            entries = [(b_, s_, min(l_, f_.shape[0]), f_) for f_, (b_, s_, l_) in zip(feats, kwargs[f"{io_name}_indices"].tolist())]
            entries = [e for e in entries if e[2] > 0]
            if len(entries) == 1:
                bidx, start, n, feat = entries[0]
                ops.gemm(feat[:n], ad.weight, bias=ad.bias, out=input_embeds[bidx, start:start + n])
            elif entries:
                # several clips / windows: ONE adaptor GEMM over all their tokens and ONE row gather that drops every projected row
                # at its (b, start + j) -- not a GEMM launch per entry (the reference loops, lm/parallel.py:277-282)
                B_, T_, H_ = input_embeds.shape
                whole = isinstance(feats, torch.Tensor) and feats.dim() == 3 and len(entries) == feats.shape[0] \
                    and all(e[2] == feats.shape[1] for e in entries)
                rows = feats.reshape(-1, feats.shape[-1]) if whole else torch.cat([e[3][:e[2]] for e in entries])
                proj = ops.gemm(rows.contiguous(), ad.weight, bias=ad.bias)
                plan = np.arange(B_ * T_, dtype=np.int64)
                off = 0
                for bidx, start, n, _ in entries:
                    if not (0 <= bidx < B_ and 0 <= start and start + n <= T_):
                        raise IndexError(f"{io_name}_indices entry ({bidx}, {start}, {n}) outside input_ids [{B_}, {T_}]")
                    plan[bidx * T_ + start: bidx * T_ + start + n] = -(np.arange(off, off + n) + 2)
                    off += n
                plan_d = torch.from_numpy(plan.astype(np.int32)).to(self.device, non_blocking=True)
                input_embeds = ops.gather_rows(input_embeds.view(B_ * T_, H_), proj, plan_d, B_ * T_).view(B_, T_, H_)
        return input_embeds

    @torch.no_grad()
    def _forward_hidden(self, input_embeds, cache: Optional[KVCache]):
        lib = L.lib()
        B, T, H = input_embeds.shape
        if cache is None:
            cache = self.new_cache(B, T + 64)
        pos0 = cache.length
        cache.reserve(pos0 + T)
        pk = self.pack(cache.cap)
        x = input_embeds.to(self.dtype).contiguous()
        ws = self._workspace(B, T, cache.cap)
        hid = torch.ops.afhip.llm_forward(pk.blob, x, pos0, cache.k, cache.v, ws)      # custom op over the opaque packed-weight blob (torch_ops.py)
        cache.length = pos0 + T
        return hid, cache

    @torch.no_grad()
    def _step(self, input_ids=None, input_embeds=None, past_key_values=None, mask=None, last_only: bool = False):
        """lm/parallel.py:570-597 -> (logits [B,T,S,V] in the model dtype, cache).
        `last_only=True` (not in the reference): logits of the LAST position only, [B,1,S,V] -- all a decoding caller slices
        (lm/parallel.py:447).  The full form is produced in chunks of 64 rows through one f32 buffer, so a T = 790 prefill at the 7B
        vocabulary peaks at the [B,T,8,V] result itself (2 GB in bf16) instead of an f32 copy (4 GB) next to it."""
        assert (input_ids is None) != (input_embeds is None), "Either input_ids or input_embeds should be None"
        lib = L.lib()
        if input_ids is not None:
            assert input_ids.size(2) == self.num_stream
            input_embeds = ops.embed_sum(input_ids.to(self.device), self.model.embed_tokens.weight)
        hid, cache = self._forward_hidden(input_embeds, past_key_values)
        if last_only:
            hid = hid[:, -1:].contiguous()
        B, T, H = hid.shape
        S, V = self.num_stream, self.lm_head.weight.shape[0]
        rows = hid.reshape(B * T, H)
        logits = torch.empty((B * T, S, V), dtype=self.dtype, device=self.device)
        CH = 64
        ws = torch.empty(min(CH, B * T) * S * H * hid.element_size() + 256, dtype=torch.uint8, device=self.device)
        for r0 in range(0, B * T, CH):
            n = min(CH, B * T - r0)
            logits[r0:r0 + n].copy_(torch.ops.afhip.lm_head(self.pack().blob, rows[r0:r0 + n].contiguous(), S, ws))
        logits = logits.view(B, T, S, V)
        if mask is not None:
            logits.masked_fill_(mask, float("-inf"))
        return logits, cache

    @torch.no_grad()
    def _head_stream0(self, hidden: torch.Tensor) -> torch.Tensor:
        """Stream-0 logits [n, V] (f32) of final-normed hidden rows [n, H]: the only lm_head rows text decoding consumes
        (the reference computes all T x 8 rows and slices, lm/parallel.py:588-592,447)."""
        lib = L.lib()
        hidden = hidden.contiguous()
        n = hidden.shape[0]
        ws = torch.empty(n * hidden.shape[-1] * hidden.element_size() + 256, dtype=torch.uint8, device=hidden.device)
        return torch.ops.afhip.lm_head(self.pack().blob, hidden, 1, ws)[:, 0]

    @torch.no_grad()
    def _masked_pick(self, logits: torch.Tensor, iv: torch.Tensor) -> torch.Tensor:
        """First-index argmax over the allowed id intervals, on logits as the MODEL dtype sees them (lm/parallel.py:592-601)."""
        lib = L.lib()
        n = logits.shape[0]
        tok = torch.empty(n, dtype=torch.int64, device=logits.device)
        am = torch.empty(lib.afhip_masked_argmax_workspace_bytes(n), dtype=torch.uint8, device=logits.device)
        L.check(lib.afhip_masked_argmax(L.ptr(logits), n, logits.shape[1], L.ptr(iv), iv.shape[0], L.ptr(tok),
                                        L.dtype_code(self.dtype), L.ptr(am), am.numel(), L.stream_ptr()))
        return tok

    @torch.no_grad()
    def _topk_probs(self, logits_f32: torch.Tensor, allowed: torch.Tensor, topk: int, temperature: float,
                    cfg_logits: Optional[torch.Tensor] = None, cfg: float = 1.0, model_dtype=None, u: Optional[torch.Tensor] = None):
        """afhip_sample_topk over rows [n, V] (f32): (topk ids [n,k] int32, mixed logits [n,k], probabilities [n,k], drawn ids [n] | None).
        `allowed` [n, n_iv, 2] int32 id intervals per row; cfg_logits / cfg: the guidance mix of lm/parallel.py:489-492."""
        lib = L.lib()
        n, V = logits_f32.shape
        dev = logits_f32.device
        if not 1 <= int(topk) <= 64:
            raise ValueError(f"topk={topk}: afhip_sample_topk selects at most 64 candidates per row (the reference accepts any k; conf/inference.yaml uses 20)")
        idx = torch.empty((n, topk), dtype=torch.int32, device=dev)
        val = torch.empty((n, topk), dtype=torch.float32, device=dev)
        prob = torch.empty((n, topk), dtype=torch.float32, device=dev)
        tok = torch.empty(n, dtype=torch.int64, device=dev) if u is not None else None
        a = L.SampleArgs()
        a.logits, a.cfg_logits, a.cfg = logits_f32.data_ptr(), (cfg_logits.data_ptr() if cfg_logits is not None else None), float(cfg)
        a.rows, a.ld, a.allowed, a.n_iv, a.k = n, logits_f32.stride(0), allowed.data_ptr(), allowed.shape[1], int(topk)
        a.temperature = float(temperature)
        a.one_minus_cfg = float(1.0 - cfg)               # Python double, rounded to f32 once -- as the reference's `(1 - cfg)` operand is (:489-492)
        a.model_dtype = L.dtype_code(model_dtype if model_dtype is not None else self.dtype)
        a.topk_idx, a.topk_val, a.topk_prob = idx.data_ptr(), val.data_ptr(), prob.data_ptr()
        a.u, a.token = (u.data_ptr() if u is not None else None), (tok.data_ptr() if tok is not None else None)
        L.check(lib.afhip_sample_topk(C.byref(a), L.stream_ptr()))
        return idx, val, prob, tok

    def _logits_to_token(self, logits, temperature, topk):
        """lm/parallel.py:599-608 on masked logits [B,T,S,V] (any float dtype): greedy argmax, or top-k -> softmax(/T) -> draw.
        The draw is the inverse CDF at one torch uniform per row (torch.multinomial's distribution); `self._sampler`, when set
        (tests), receives (topk ids [B,T,S,k], probabilities) and returns the inner indices instead."""
        if temperature == 0:
            return logits.argmax(-1)
        shp = logits.shape[:-1]
        rows = logits.reshape(-1, logits.shape[-1]).float().contiguous()
        whole = torch.tensor([[[0, rows.shape[1]]]], dtype=torch.int32, device=rows.device).expand(rows.shape[0], 1, 2).contiguous()
        sampler = getattr(self, "_sampler", None)
        u = None if sampler is not None else torch.rand(rows.shape[0], device=rows.device)
        idx, val, prob, tok = self._topk_probs(rows, whole, topk, temperature, model_dtype=torch.float32, u=u)
        if sampler is not None:
            inner = sampler(idx.view(*shp, topk), prob.view(*shp, topk)).to(rows.device).long().view(-1, 1)
            tok = torch.gather(idx.long(), 1, inner).squeeze(1)
        return tok.view(*shp)

    @torch.no_grad()
    def _prepare_cfg_cache(self, cache: "KVCache") -> "KVCache":
        """lm/parallel.py:610-644: the unconditional half of classifier-free guidance -- a cache of the same length built from
        all-pad ids (zero embeddings) -- stacked under the conditional one along the batch axis."""
        B, length = cache.batch, cache.get_seq_length()
        zeros = torch.zeros((B, length, self.num_stream), dtype=torch.int64, device=self.device)
        _, cfg_cache = self._forward_hidden(ops.embed_sum(zeros, self.model.embed_tokens.weight), self.new_cache(B, cache.cap))
        both = self.new_cache(2 * B, cache.cap)
        for name in ("k", "v"):
            dst, a, b = getattr(both, name), getattr(cache, name), getattr(cfg_cache, name)
            dst[:, :B, :, :length] = a[:, :, :, :length]
            dst[:, B:, :, :length] = b[:, :, :, :length]
        both.length = length
        del cfg_cache                                          # the unconditional half now lives in `both`
        return both

    @torch.no_grad()
    def _greedy_device_loop(self, modality_token, cache: KVCache, modality: str, max_step: int, poll: int = 16):
        """Device-resident greedy loop: one `afhip_llm_decode_step` per token, eos/eot bookkeeping on the GPU, the
        host looks at the finished flags every `poll` steps only.  Exact w.r.t. lm/parallel.py:480-513 when every
        stream but 0 can only emit pad."""
        lib = L.lib()
        B = cache.batch
        iv, _ = self._allowed_intervals(modality)
        T0 = cache.length
        cache.reserve(T0 + max_step + 2)
        pk = self.pack(cache.cap)
        prev = modality_token[:, 0, 0].to(self.device).contiguous().clone()
        out_tokens = torch.zeros((max_step, B), dtype=torch.int64, device=self.device)
        finished = torch.full((B,), -1, dtype=torch.int32, device=self.device)
        # loop state on the device (afhip_decode_state.seq_pos / step_counter): nothing position-dependent is baked into kernel
        # arguments, so ONE captured hipGraph of a step serves every token
        seq_pos = torch.full((B,), T0, dtype=torch.int32, device=self.device)
        step_counter = torch.zeros(1, dtype=torch.int32, device=self.device)
        # every allowed id is below head_rows: the step need not stream the lm_head rows behind it (text decode: the 8 x 1025 audio-code
        # rows, lm/parallel.py:557-568)
        head_rows = self._allowed_hi.get(modality, 0)
        ws = self._workspace(B, 1, cache.cap)
        max_pos = T0 + max_step - 1                      # largest position this loop can append at

        def one_step():
            torch.ops.afhip.llm_decode_step(pk.blob, cache.k, cache.v, prev, out_tokens, finished, iv, self.eos_token_id, self.eot_token_id,
                                            seq_pos, step_counter, max_pos, ws, head_rows)

        # a capture costs a device synchronise + allocator housekeeping (~ms): only worth it for loops long enough to amortise it
        use_graph = os.environ.get("AFHIP_DECODE_GRAPH", "1") != "0" and max_step >= 16 and not torch.cuda.is_current_stream_capturing()
        graph = None
        n_done = max_step
        for step in range(max_step):
            captured_now = False
            if use_graph and step == 1:
                # step 0 ran eagerly (first-use set-up of the kernels happens outside the capture); capture step 1 once, replay it.
                # Capture does not execute, so when it fails (an outer capture, another thread issuing HIP calls under the global
                # capture mode, ...) nothing of this step has happened yet: fall back to eager launches for the rest of the loop.
                try:
                    g_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_):
                        one_step()
                    graph = g_
                except Exception as e:  # noqa: BLE001 -- any capture failure means "no graph", never "no decode"
                    graph, use_graph = None, False
                    self._graph_fallback = repr(e)
            if graph is not None:
                graph.replay()
            else:
                one_step()
            if (step + 1) % poll == 0 or step == max_step - 1:
                f = finished.cpu()
                if bool((f >= 0).all()):
                    n_done = int(f.max()) + 1      # the reference leaves the loop right after this step (:512-513)
                    break
        cache.length = T0 + n_done
        f = finished.cpu()
        finish_idx = torch.where(f < 0, torch.full_like(f, n_done - 1), f)
        hypos = torch.zeros((B, n_done, self.num_stream), dtype=torch.int64, device=self.device)
        hypos[:, :, 0] = out_tokens[:n_done].t()
        return hypos, finish_idx, cache

    @torch.no_grad()
    def inference_segment(self, config: dict, cache=None, enforce_modality: str = None, **kwargs):
        """lm/parallel.py:428-533."""
        input_ids = kwargs.get("seqs").to(self.device)
        B0 = input_ids.shape[0]
        input_ids = torch.cat([input_ids, self.assistant_token.expand(B0, -1, -1)], dim=1)
        device = input_ids.device
        input_embeds = self._embed(input_ids, kwargs)

        # (1) prefill; only the last position's stream-0 logits are consumed (:447-457)
        hid, cache = self._forward_hidden(input_embeds, cache)
        if enforce_modality is not None:
            modality_token = getattr(self, f"{enforce_modality}_token").expand(B0, -1, -1).clone()
        else:
            iv, _ = self._allowed_intervals("modality")
            tok = self._masked_pick(self._head_stream0(hid[:, -1]), iv)
            modality_token = torch.zeros((B0, 1, self.num_stream), dtype=torch.int64, device=device)
            modality_token[:, 0, 0] = tok

        # (2) modality and its mask
        modality = self.vocab[modality_token.flatten()[0].item()].replace("<|", "").replace("|>", "")
        if not hasattr(self, f"{modality}_mask"):
            raise ValueError(f"Try to predict {modality} modality But no decoding mask exists for it.")
        modality_mask = getattr(self, f"{modality}_mask")
        if modality not in config:
            raise ValueError(f"Try to predict {modality} modality But the corresponding inference config is missing.")
        this_config = config[modality]

        num_hypo = config.get("num_hypo", 1)
        if num_hypo > 1:
            indices = torch.zeros(num_hypo).long().to(device)
            cache.batch_select_indices(indices)
            modality_token = modality_token.tile(num_hypo, 1, 1)
        cfg = this_config.get("cfg", 1)
        _, pad_only = self._allowed_intervals(modality)

        if this_config["temperature"] == 0 and cfg <= 1 and pad_only:
            hypos, finish_idx, cache = self._greedy_device_loop(modality_token, cache, modality, this_config["max_step"])
            prev_token = hypos[:, -1:, :].clone()
        else:
            # general branch (lm/parallel.py:472-513): sampling, classifier-free guidance, or streams that carry real tokens.
            # Per step: embed -> one position through the stack -> lm_head on all S streams -> ONE kernel that applies the
            # guidance mix, the modality mask (as per-stream id intervals), top-k, softmax(/T) and the draw.
            lib = L.lib()
            nb = cache.batch                                   # num_hypo
            # capacity for the whole segment up front (as the greedy loop does): growing 64 positions at a time would re-copy the
            # cache O(n^2 / 64) times over an audio-output decode of thousands of steps
            cache.reserve(cache.length + this_config["max_step"] + 2)
            if cfg > 1:
                cache = self._prepare_cfg_cache(cache)
            S, V, H = self.num_stream, self.lm_head.weight.shape[0], self.cfg["hidden_size"]
            allowed = self._stream_intervals(modality)[None].expand(nb, -1, -1, -1).reshape(nb * S, -1, 2).contiguous()
            temperature, topk = this_config["temperature"], this_config["topk"]
            sampler = getattr(self, "_sampler", None)
            hyp_list = []
            finish_idx = torch.ones(nb).long().to(device) * -1
            prev_token = modality_token
            for step in range(this_config["max_step"]):
                if cfg > 1:
                    prev_token = prev_token.tile(2, 1, 1)
                nrow = prev_token.shape[0]
                hid, cache = self._forward_hidden(ops.embed_sum(prev_token, self.model.embed_tokens.weight), cache)
                ws = torch.empty(nrow * S * H * hid.element_size() + 256, dtype=torch.uint8, device=device)
                logits = torch.ops.afhip.lm_head(self.pack().blob, hid.reshape(nrow, H), S, ws).reshape(nrow * S, V)
                cond, uncond = (logits[: nb * S], logits[nb * S:]) if cfg > 1 else (logits, None)
                if temperature == 0:
                    idx, _, _, _ = self._topk_probs(cond, allowed, 1, 1.0, cfg_logits=uncond, cfg=cfg)
                    tok = idx[:, 0].long()
                else:
                    u = None if sampler is not None else torch.rand(nb * S, device=device)
                    idx, _, prob, tok = self._topk_probs(cond, allowed, topk, temperature, cfg_logits=uncond, cfg=cfg, u=u)
                    if sampler is not None:
                        inner = sampler(idx.view(nb, 1, S, topk), prob.view(nb, 1, S, topk)).to(device).long().view(-1, 1)
                        tok = torch.gather(idx.long(), 1, inner).squeeze(1)
                prev_token = tok.view(nb, 1, S)
                hyp_list.append(prev_token)
                finish_here = torch.logical_and(
                    torch.logical_or(prev_token[:, 0, 0] == self.eot_token_id, prev_token[:, 0, 0] == self.eos_token_id),
                    finish_idx == -1)
                finish_idx = torch.where(finish_here, step, finish_idx)
                if torch.all(finish_idx >= 0):
                    break
            finish_idx = torch.where(finish_idx == -1, step, finish_idx)
            hypos = torch.cat(hyp_list, dim=1)
            if cfg > 1:
                cache.batch_select_indices(torch.arange(nb, device=device))

        # (5) "prefill the last token" so that a following segment continues from it (:523-526)
        prev_token = prev_token.clone()
        prev_token[..., 1:] = 0
        emb = ops.embed_sum(prev_token, self.model.embed_tokens.weight)
        _, cache = self._forward_hidden(emb, cache)

        hypo_lst = []
        for idx, hypo in zip(finish_idx.tolist(), hypos):
            hypo_lst.append((hypo[: idx + 1], modality))
        return hypo_lst, cache

    @torch.no_grad()
    def inference(self, inference_config: dict, cache=None, **kwargs):
        """lm/parallel.py:387-426."""
        messages = []
        while True:
            decoded_sequences, cache = self.inference_segment(inference_config, cache=cache, enforce_modality=None, **kwargs)
            for seq, modality in decoded_sequences:
                if seq[-1, 0] == self.eos_token_id or seq[-1, 0] == self.eot_token_id:
                    seq = seq[:-1]
                io_name = "discrete_audio" if modality == "audio" else modality
                seq = seq.unsqueeze(0) - self.vocab_intervals[io_name][0][0]
                io = self.multimodal_io_dict[io_name]
                lengths = torch.Tensor([seq.size(1)]).long().to(seq.device)
                content = io.decode_batch(seq, lengths)
                messages.append(["assistant", modality, content])
            if len(decoded_sequences) > 1:
                break
            elif decoded_sequences[0][0][-1, 0] != self.eot_token_id:
                break
        return messages, cache
