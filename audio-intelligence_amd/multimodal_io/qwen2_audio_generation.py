"""AF3 / Qwen2-Audio style `generate()` surface over the HIP kernels.

Host-side mirror of `Qwen2AudioForConditionalGeneration` (UALM/models/ualm/multimodal_io/modeling_whisper.py:855-1369): same
sub-module names (`audio_tower`, `multi_modal_projector`, `language_model`), hence the same state-dict keys; same `forward`
arguments and return fields (`logits`, `past_key_values`, `attention_mask`); `prepare_inputs_for_generation` with the reference's
three input-slicing rules (:1250-1318); a greedy `generate()`.  `<|AUDIO|>` placeholders are widened by the merge that is already
pinned to the reference (`merge_input_ids_with_audio_features`, :913-1108).

Padded batches without a padding mask in the kernels: the merged batch is COMPACTED -- each sequence's valid tokens are moved to
cache slots [0, len_b) -- which is exactly what the reference's `attention_mask` + `position_ids = cumsum(mask) - 1` compute
(pad keys are never attended, valid tokens get positions 0 .. len_b - 1), for left- and right-padded batches alike.  Prefill then
runs as one causal batch over [B, max len] (rows past len_b are dead: nothing valid attends to them and decode overwrites their
cache slots), and each decode step appends at the per-sequence slot `seq_len[b]` through `afhip_llm_forward_ragged`.
"""
import ctypes as C
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from .modeling_whisper import (AFWhisperEncoder, AFWhisperEncoderConfig, Qwen2AudioMultiModalProjector, _Linear,
                               merge_input_ids_with_audio_features)


def _get(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


class _CausalLM(nn.Module):
    """`Qwen2ForCausalLM` parameter layout: `model.embed_tokens / layers / norm` + `lm_head` (no bias)."""

    def __init__(self, text_config: dict):
        super().__init__()
        from ..lm.parallel import _Backbone
        self.cfg = dict(text_config)
        self.model = _Backbone(self.cfg, self.cfg["vocab_size"])
        self.lm_head = _Linear(self.cfg["hidden_size"], self.cfg["vocab_size"], bias=False)
        self._packed = None

    def _apply(self, fn, *a, **kw):
        self._packed = None
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._packed = None
        return super()._load_from_state_dict(*a, **kw)

    def pack(self, max_positions=None):
        from ..lm.parallel import pack_qwen2_weights
        if self._packed is None or (max_positions is not None and max_positions > self._packed.max_pos):
            self._packed = pack_qwen2_weights(self.model, self.lm_head.weight, None, self.cfg, 1, max_positions=max_positions)
        return self._packed


class AF3Cache:
    """KV cache of a (possibly ragged) batch in compact layout + the bookkeeping `generate()` needs.  `get_seq_length()` is the
    length of the PADDED layout the caller sees (merged prompt width + generated tokens), like the reference's cache."""

    def __init__(self, kv, seq_len: torch.Tensor, padded_len: int, compact_len: int):
        self.kv, self.seq_len, self.padded_len = kv, seq_len, padded_len
        self.compact_len = compact_len            # longest sequence in cache slots (host-side bound for capacity checks)
        self.seen_tokens = padded_len

    def get_seq_length(self):
        return self.padded_len


class Qwen2AudioForConditionalGeneration(nn.Module):
    def __init__(self, config):
        """config: `audio_config` (fields of Qwen2AudioEncoderConfig), `text_config` (Qwen2 config dict: hidden_size,
        num_hidden_layers, num_attention_heads, num_key_value_heads, intermediate_size, vocab_size, rope_theta, rms_norm_eps),
        `audio_token_index`, optional `pad_token_id`, `ignore_index` (-100), `eos_token_id`."""
        super().__init__()
        self.config = config if not isinstance(config, dict) else SimpleNamespace(**config)
        acfg, tcfg = _get(config, "audio_config"), _get(config, "text_config")
        acfg = acfg if isinstance(acfg, dict) else vars(acfg)
        tcfg = tcfg if isinstance(tcfg, dict) else vars(tcfg)
        self.audio_tower = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(acfg))
        self.multi_modal_projector = Qwen2AudioMultiModalProjector(d_model=self.audio_tower.config.d_model, hidden_size=tcfg["hidden_size"])
        self.vocab_size = tcfg["vocab_size"]
        self.language_model = _CausalLM(tcfg)
        pad = _get(config, "pad_token_id")
        self.pad_token_id = pad if pad is not None else -1
        self.audio_token_index = _get(config, "audio_token_index")
        self.ignore_index = _get(config, "ignore_index", -100)
        self._padding_side = "left"
        self._ws = None

    # ---- modeling_whisper.py:866-875
    @property
    def padding_side(self):
        return self._padding_side

    @padding_side.setter
    def padding_side(self, padding_side: str):
        if padding_side not in ["left", "right"]:
            raise ValueError(f"{padding_side} is not `left` or `right`.")
        self._padding_side = padding_side

    @property
    def dtype(self):
        return self.language_model.lm_head.weight.dtype

    @property
    def device(self):
        return self.language_model.lm_head.weight.device

    def get_input_embeddings(self):
        table = self.language_model.model.embed_tokens.weight
        return lambda ids: ops.embed_sum(ids.to(table.device).unsqueeze(-1), table)

    def get_output_embeddings(self):
        return self.language_model.lm_head

    def _merge_input_ids_with_audio_features(self, audio_features, num_audio_tokens, inputs_embeds, input_ids, attention_mask, labels):
        return merge_input_ids_with_audio_features(audio_features, num_audio_tokens, inputs_embeds, input_ids, attention_mask, labels,
                                                   audio_token_index=self.audio_token_index, pad_token_id=self.pad_token_id,
                                                   ignore_index=self.ignore_index, padding_side=self.padding_side)

    # ---- device plumbing
    def _workspace(self, B, T, max_ctx):
        lib = L.lib()
        need = lib.afhip_llm_workspace_bytes(C.byref(self.language_model.pack().w), B, T, max_ctx)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _new_kv(self, B, cap):
        from ..lm.parallel import KVCache
        pk = self.language_model.pack(cap)
        return KVCache(self.language_model.cfg["num_hidden_layers"], B, pk.nkv, (cap + 63) // 64 * 64, pk.hd, self.dtype, self.device)

    def _head(self, hidden_rows: torch.Tensor) -> torch.Tensor:
        """[n, H] final-normed rows -> [n, V] f32 logits."""
        lib = L.lib()
        hidden_rows = hidden_rows.contiguous()
        n = hidden_rows.shape[0]
        ws = torch.empty(n * hidden_rows.shape[1] * hidden_rows.element_size() + 256, dtype=torch.uint8, device=hidden_rows.device)
        return torch.ops.afhip.lm_head(self.language_model.pack().blob, hidden_rows, 1, ws)[:, 0]

    def _encode_audio(self, input_features, feature_attention_mask):
        """modeling_whisper.py:1181-1207: per-clip key length from the feature mask, encoder, projector."""
        tower = self.audio_tower
        audio_feat_lengths, audio_output_lengths = tower._get_feat_extract_output_lengths(feature_attention_mask.sum(-1))
        x = input_features.to(self.device)
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        mel_btc = ops.transpose_cast(x.contiguous(), tower.dtype)
        feats = tower.encode_btc(mel_btc, feat_len=audio_feat_lengths.clamp(max=tower.config.max_source_positions))
        return self.multi_modal_projector(feats), audio_output_lengths

    # ---- forward (modeling_whisper.py:1112-1248)
    @torch.no_grad()
    def forward(self, input_ids=None, input_features=None, attention_mask=None, feature_attention_mask=None, position_ids=None,
                past_key_values: Optional[AF3Cache] = None, inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, logits_to_keep: int = 0, max_new_tokens: int = 64):
        """Returns a namespace with `logits` [B, L', V] (model dtype; rows of padding positions are zero -- the reference computes
        unspecified values there), `past_key_values`, `attention_mask` (the merged one, as the reference's output carries it).
        `logits_to_keep` = k > 0 restricts the head to the last k positions of the padded layout (0 = all, the reference's
        behaviour); `max_new_tokens` only sizes the cache it creates.  `labels` are carried through the merge; no loss (inference
        path).  `position_ids` is accepted for signature parity: positions are cumsum(mask) - 1, which is what the reference
        computes for generation (:1293-1299)."""
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not available from the fused kernels")
        lib = L.lib()
        dev = self.device
        if past_key_values is not None:
            # ---- one new token per sequence against the cache
            if inputs_embeds is None:
                if input_ids.shape[1] != 1:
                    raise ValueError("with past_key_values, pass exactly one new token per sequence")
                inputs_embeds = self.get_input_embeddings()(input_ids.to(dev))
            c = past_key_values
            B = inputs_embeds.shape[0]
            if c.compact_len + 1 > c.kv.cap:
                raise L.AfhipError(f"KV cache capacity {c.kv.cap} exhausted: create the cache with a larger max_new_tokens")
            max_pos = c.kv.cap - 1
            pk = self.language_model.pack(c.kv.cap)
            x = inputs_embeds.reshape(B, -1).to(self.dtype).contiguous()
            ws = self._workspace(B, 1, c.kv.cap)
            hid = torch.ops.afhip.llm_forward_ragged(pk.blob, x, c.seq_len, max_pos, c.kv.k, c.kv.v, ws)
            c.seq_len += 1
            c.padded_len += 1
            c.compact_len += 1
            c.seen_tokens = c.padded_len
            logits = self._head(hid).to(self.dtype).view(B, 1, -1)
            if attention_mask is not None:
                attention_mask = attention_mask.to(dev)
            return SimpleNamespace(loss=None, logits=logits, past_key_values=c, hidden_states=None, attentions=None, attention_mask=attention_mask)

        # ---- prompt
        hid, cache, mask, attention_mask, labels, position_ids = self._prefill(input_ids, input_features, attention_mask, feature_attention_mask,
                                                                               inputs_embeds, labels, max_new_tokens)
        B, Lp = mask.shape
        # logits back in the padded layout
        keep = Lp if logits_to_keep in (0, None) else min(int(logits_to_keep), Lp)
        cols = torch.arange(Lp - keep, Lp, device=dev)
        rank = torch.cumsum(mask.long(), dim=1) - 1                                       # compact slot of every valid padded position
        sel_valid = mask[:, cols]
        rows = hid[torch.arange(B, device=dev)[:, None].expand(B, keep)[sel_valid], rank[:, cols][sel_valid]]
        logits = torch.zeros((B, keep, self.vocab_size), dtype=self.dtype, device=dev)
        if rows.shape[0] > 0:
            logits[sel_valid] = self._head(rows).to(self.dtype)
        return SimpleNamespace(loss=None, logits=logits, past_key_values=cache if use_cache is not False else None, hidden_states=None,
                               attentions=None, attention_mask=attention_mask, labels=labels, position_ids=position_ids)

    def _prefill(self, input_ids, input_features, attention_mask, feature_attention_mask, inputs_embeds, labels, max_new_tokens):
        """embed -> (audio tower -> projector -> merge) -> compact -> causal prefill.  Returns the compact final-normed hidden
        states [B, max len, H], the cache, the boolean merged mask [B, L'], and the merge's other outputs."""
        lib = L.lib()
        dev = self.device
        position_ids = None
        if inputs_embeds is None:
            input_ids = input_ids.to(dev)
            inputs_embeds = self.get_input_embeddings()(input_ids)
            if attention_mask is None:
                attention_mask = torch.ones_like(input_ids)
            attention_mask = attention_mask.to(dev)
            if input_features is not None and input_ids.shape[1] != 1:
                audio_features, audio_output_lengths = self._encode_audio(input_features, feature_attention_mask.to(dev))
                inputs_embeds, attention_mask, labels, position_ids, _ = self._merge_input_ids_with_audio_features(
                    audio_features, audio_output_lengths, inputs_embeds, input_ids, attention_mask, labels)
        elif attention_mask is None:
            attention_mask = torch.ones(inputs_embeds.shape[:2], dtype=torch.long, device=dev)
        B, Lp, H = inputs_embeds.shape
        mask = attention_mask.to(dev).bool()
        lens = mask.sum(-1)
        Lmax = int(lens.max())
        # compact layout: valid tokens of sequence b -> slots [0, len_b) (stable order)
        order = torch.argsort((~mask).to(torch.int8), dim=1, stable=True)               # valid positions first, original order kept
        xc = torch.gather(inputs_embeds.to(self.dtype), 1, order[:, :Lmax, None].expand(B, Lmax, H)).contiguous()
        kv = self._new_kv(B, Lmax + max_new_tokens + 1)
        pk = self.language_model.pack(kv.cap)
        ws = self._workspace(B, Lmax, kv.cap)
        hid = torch.ops.afhip.llm_forward(pk.blob, xc, 0, kv.k, kv.v, ws)
        kv.length = Lmax
        return hid, AF3Cache(kv, lens.to(torch.int32).contiguous(), Lp, Lmax), mask, attention_mask, labels, position_ids

    __call__ = forward

    # ---- modeling_whisper.py:1250-1318
    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, input_features=None, attention_mask=None, **kwargs):
        if past_key_values is not None:
            cache_length = past_length = past_key_values.get_seq_length()
            if input_features is not None and kwargs.get("attention_mask") is not None:
                attention_mask = kwargs["attention_mask"]
                attention_mask = torch.cat([attention_mask, attention_mask.new_ones((attention_mask.shape[0], 1))], dim=-1)
            if attention_mask is not None and attention_mask.shape[1] > input_ids.shape[1]:
                input_ids = input_ids[:, -(attention_mask.shape[1] - past_length):]
            elif past_length < input_ids.shape[1]:
                input_ids = input_ids[:, past_length:]
            elif self.audio_token_index in input_ids:
                input_ids = input_ids[:, input_ids.shape[1] - 1:]
            if cache_length < past_length and attention_mask is not None:
                attention_mask = attention_mask[:, -(cache_length + input_ids.shape[1]):]
        position_ids = kwargs.get("position_ids", None)
        if attention_mask is not None and position_ids is None:
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values:
                position_ids = position_ids[:, -input_ids.shape[1]:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({"position_ids": position_ids, "past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                             "attention_mask": attention_mask, "input_features": input_features,
                             "feature_attention_mask": kwargs.get("feature_attention_mask", None)})
        return model_inputs

    # ---- greedy generation (GenerationMixin.generate, do_sample=False)
    @torch.no_grad()
    def generate(self, input_ids, input_features=None, attention_mask=None, feature_attention_mask=None, max_new_tokens: int = 20,
                 eos_token_id=None, pad_token_id=None):
        """Greedy decoding: returns [B, L + n_new] = the prompt ids followed by the generated ones; a sequence that has emitted
        `eos_token_id` is continued with `pad_token_id` (HF semantics), generation stops when every sequence has finished."""
        dev = self.device
        eos = eos_token_id if eos_token_id is not None else _get(self.config, "eos_token_id")
        eos = [] if eos is None else ([eos] if isinstance(eos, int) else list(eos))
        pad = pad_token_id if pad_token_id is not None else (self.pad_token_id if self.pad_token_id >= 0 else (eos[0] if eos else 0))
        input_ids = input_ids.to(dev)
        hid, cache, mask, _, _, _ = self._prefill(input_ids, input_features, attention_mask, feature_attention_mask, None, None, max_new_tokens)
        B = input_ids.shape[0]
        # the last VALID token of each prompt, wherever the padding sits (the reference's generate() reads column -1, i.e. needs
        # left padding; the compact layout does not care)
        nxt_logits = self._head(hid[torch.arange(B, device=dev), cache.seq_len.long() - 1]).to(self.dtype)
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        new = []
        for step in range(max_new_tokens):
            tok = nxt_logits.argmax(-1)
            tok = torch.where(unfinished, tok, torch.full_like(tok, pad))
            new.append(tok)
            for e in eos:
                unfinished = unfinished & (tok != e)
            if not bool(unfinished.any()) or step == max_new_tokens - 1:
                break
            o = self.forward(input_ids=tok[:, None], past_key_values=cache, use_cache=True)
            nxt_logits, cache = o.logits[:, -1], o.past_key_values
        return torch.cat([input_ids, torch.stack(new, dim=1)], dim=1)
