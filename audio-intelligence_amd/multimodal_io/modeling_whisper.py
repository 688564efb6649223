"""AF-Whisper encoder module backed by the HIP library.

Host-side mirror of `AFWhisperEncoder` (UALM/models/ualm/multimodal_io/modeling_whisper.py:589-765): same
constructor shape (a config object), same parameter names (so reference checkpoints load with strict=True,
modeling_whisper.py:132-135,463-469,614-621), same forward signature and the same length helper.  No arithmetic
happens in torch: forward() hands device pointers to `afhip_encoder_forward`.
"""
import ctypes as C
import json
import os
from dataclasses import dataclass, asdict
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops


@dataclass
class AFWhisperEncoderConfig:
    """Fields of Qwen2AudioEncoderConfig that the encoder reads (defaults = Whisper-large-v3 encoder)."""
    num_mel_bins: int = 128
    d_model: int = 1280
    encoder_attention_heads: int = 20
    encoder_ffn_dim: int = 5120
    encoder_layers: int = 32
    max_source_positions: int = 1500
    activation_function: str = "gelu"
    scale_embedding: bool = False
    pad_token_id: int = 0

    @classmethod
    def from_dict(cls, d: dict):
        keys = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in keys})


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's state-dict layout (weight [out,in], optional bias)."""

    def __init__(self, n_in, n_out, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(n_out), requires_grad=False) if bias else None


class _Norm(nn.Module):
    def __init__(self, d, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(d), requires_grad=False) if bias else None


class _Conv1d(nn.Module):
    def __init__(self, c_in, c_out, k, stride):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in, k), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(c_out), requires_grad=False)
        self.stride = (stride,)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = _Linear(d, d, bias=False)   # modeling_whisper.py:132
        self.v_proj = _Linear(d, d)
        self.q_proj = _Linear(d, d)
        self.out_proj = _Linear(d, d)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self_attn = _Attn(cfg.d_model)
        self.self_attn_layer_norm = _Norm(cfg.d_model)
        self.fc1 = _Linear(cfg.d_model, cfg.encoder_ffn_dim)
        self.fc2 = _Linear(cfg.encoder_ffn_dim, cfg.d_model)
        self.final_layer_norm = _Norm(cfg.d_model)


class _Embedding(nn.Module):
    def __init__(self, n, d):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d), requires_grad=False)


class AFWhisperEncoder(nn.Module):
    main_input_name = "input_features"

    def __init__(self, config: AFWhisperEncoderConfig):
        super().__init__()
        if config.activation_function != "gelu" or config.scale_embedding:
            raise ValueError("AF-Whisper uses erf GELU and no embedding scale (modeling_whisper.py:465,612)")
        self.config = config
        d = config.d_model
        self.num_mel_bins = config.num_mel_bins
        self.max_source_positions = config.max_source_positions
        self.conv1 = _Conv1d(config.num_mel_bins, d, 3, 1)
        self.conv2 = _Conv1d(d, d, 3, 2)
        self.embed_positions = _Embedding(config.max_source_positions, d)
        self.layers = nn.ModuleList([_Layer(config) for _ in range(config.encoder_layers)])
        self.layer_norm = _Norm(d)
        self._packed = None
        self._ws = None
        self._fp8 = False
        self._fc2_in_scale = None      # per-layer static scale of fc2's e4m3 input (calibrate_fp8)
        self._att_out_scale = None     # per-layer static scale of the attention output = the out-projection's e4m3 input
        self._calib = None             # device [n_layers] f32: where a calibration forward records max |GELU output|

    def enable_fp8(self, on: bool = True):
        """BASELINE config 5: run the four projections of every layer on OCP e4m3 operands (weights quantised once per output
        channel here, activations per row on the device with the LayerNorm fused; f32 accumulate, bf16 stream).  bf16 models only."""
        if on and self.dtype != torch.bfloat16:
            raise L.AfhipError("the fp8 encoder path needs a bfloat16 model")
        self._fp8 = bool(on)
        self._packed = None
        return self

    @torch.no_grad()
    def _collect_amax(self, batches) -> torch.Tensor:
        """max |GELU output| ([0, L)) and max |attention output| ([L, 2 L)) per layer over `batches` (one mel tensor, a (mel, feat_len) pair
        or a list of either), from forwards that keep those activations in bf16.  The device maxima accumulate across the forwards
        (afhip_absmax_bf16 merges with an atomic max); a non-finite activation comes back as a non-finite maximum."""
        if torch.is_tensor(batches) or (isinstance(batches, tuple) and len(batches) == 2 and torch.is_tensor(batches[0])):
            batches = [batches]
        nl = self.config.encoder_layers
        keep = (self._fc2_in_scale, self._att_out_scale)
        self._fc2_in_scale, self._att_out_scale = None, None
        self._calib = torch.zeros(2 * nl, dtype=torch.float32, device=self.device)
        self._packed = None
        try:
            for item in batches:
                mel, flen = item if isinstance(item, tuple) else (item, None)
                self.encode_btc(mel, flen)
            torch.cuda.synchronize(self.device)
            return self._calib.cpu()
        finally:
            self._calib = None
            self._fc2_in_scale, self._att_out_scale = keep
            self._packed = None

    def calibrate_fp8(self, mel_btc, feat_len: Optional[torch.Tensor] = None, margin: float = 2.0, attention_output: bool = True):
        """Static quantisation of fc2's input (afhip_encoder_weights.fc2_in_scale).  Runs the e4m3 forward on `mel_btc` (one batch, or a list
        of mel tensors / (mel, feat_len) pairs: the maxima accumulate over all of them) with the GELU output still in bf16 and records
        max |value| per layer on the device; scale_l = margin * amax_l / 448 (e4m3's relative precision does not depend on the scale, so
        the margin only costs range at the small end; larger values saturate at 448 -- `check_fp8_range` tells whether a batch would).
        From then on fc1 writes that activation as e4m3 bytes and fc2 reads them with scale_l for every row: the [rows, ffn] bf16 round
        trip and the per-row quantisation launch are gone.  `attention_output=True` does the same for the attention output ->
        out-projection (the encoder attention kernel writes e4m3).  Returns the fc2 scales.  `calibrate_fp8(None)` drops everything again.
        No reference counterpart: the evidence for this mode is a token-level contract on seeded synthetic weights, not a
        fixture of the reference."""
        if mel_btc is None:
            self._fc2_in_scale, self._att_out_scale, self._packed = None, None, None
            return None
        if not self._fp8:
            raise L.AfhipError("calibrate_fp8: call enable_fp8() first")
        self._fc2_in_scale, self._att_out_scale = None, None
        nl = self.config.encoder_layers
        amax = self._collect_amax((mel_btc, feat_len) if torch.is_tensor(mel_btc) else mel_btc)
        if not bool(torch.isfinite(amax).all()) or float(amax[:nl].min()) <= 0.0:
            raise L.AfhipError(f"calibrate_fp8: unusable activation maxima (non-finite activations, or a layer that is all zero) {amax.tolist()}")
        self._fc2_in_scale = (amax[:nl] * (margin / 448.0)).to(torch.float32).contiguous()
        if attention_output and float(amax[nl:].min()) > 0.0:
            self._att_out_scale = (amax[nl:] * (margin / 448.0)).to(torch.float32).contiguous()
        self._packed = None
        return self._fc2_in_scale.clone()

    def check_fp8_range(self, mel_btc, feat_len: Optional[torch.Tensor] = None, raise_on_saturation: bool = True) -> torch.Tensor:
        """Would `mel_btc` saturate the statically quantised activations?  Re-measures the per-layer maxima on it (bf16 activations, the
        calibration forward) and returns max |value| / (448 * scale) per quantised tensor ([0, L): fc2 input, [L, 2 L): attention output,
        NaN where that tensor is not statically quantised): above 1 the e4m3 epilogues clip at 448 silently.  Raises by default."""
        if self._fc2_in_scale is None:
            raise L.AfhipError("check_fp8_range: no static scales (calibrate_fp8 first)")
        nl = self.config.encoder_layers
        amax = self._collect_amax((mel_btc, feat_len) if torch.is_tensor(mel_btc) else mel_btc)
        lim = torch.full((2 * nl,), float("nan"))
        lim[:nl] = self._fc2_in_scale * 448.0
        if self._att_out_scale is not None:
            lim[nl:] = self._att_out_scale * 448.0
        ratio = amax / lim
        bad = (~torch.isfinite(amax)) | (ratio > 1.0)
        if raise_on_saturation and bool(bad.any()):
            raise L.AfhipError(f"check_fp8_range: activations exceed the calibrated e4m3 range (max / limit per tensor: {ratio.tolist()}): "
                               "re-calibrate on representative audio or use the per-row dynamic mode (calibrate_fp8(None))")
        return ratio

    # ---------------------------------------------------------------- checkpoints
    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, **kwargs):
        """Reads `config.json` + `model.safetensors` (or `pytorch_model.bin`, weights_only) from a local directory,
        as `AFWhisperEncoder.from_pretrained(encoder_local_path, torch_dtype=...)` does at audio.py:976-979."""
        with open(os.path.join(path, "config.json")) as f:
            cfg = AFWhisperEncoderConfig.from_dict(json.load(f))
        model = cls(cfg)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=True)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model

    def save_pretrained(self, path: str):
        from safetensors.torch import save_file
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(asdict(self.config), f, indent=1)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(path, "model.safetensors"))

    def _apply(self, fn, *a, **kw):   # .to() / .cuda() / .half(): packed weights must be rebuilt
        self._packed = None
        self._ws = None
        return super()._apply(fn, *a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        # nn.Module.load_state_dict recurses through _load_from_state_dict (never through a child's load_state_dict), so this
        # is the hook that also fires when a PARENT (ParallelLLM, inference.load_checkpoint) loads a checkpoint: the derived
        # copies (fused q|k|v, tap-major conv, LayerNorm-folded weights) must be rebuilt from the new parameters
        self._packed = None
        return super()._load_from_state_dict(*a, **kw)

    @property
    def dtype(self):
        return self.conv1.weight.dtype

    @property
    def device(self):
        return self.conv1.weight.device

    # ---------------------------------------------------------------- packing for the HIP library
    def pack(self):
        """Build the layouts `afhip_encoder_forward` wants: conv weights [d, 3*C] tap-major, q|k|v fused [3d, d]
        with a zero k-bias section.  Done once per device/dtype; keeps the tensors alive in `self._packed`."""
        if self._packed is not None:
            return self._packed
        cfg = self.config
        dev, dt = self.device, self.dtype
        if dev.type != "cuda":
            raise L.AfhipError("AFWhisperEncoder runs on the GPU only: call .to('cuda') first (no CPU fallback)")
        keep = []

        def P(t):
            t = t.detach().contiguous()
            keep.append(t)
            return t

        w = L.EncoderWeights()
        w.n_mels, w.d_model, w.n_heads = cfg.num_mel_bins, cfg.d_model, cfg.encoder_attention_heads
        w.ffn_dim, w.n_layers, w.max_pos, w.dtype = cfg.encoder_ffn_dim, cfg.encoder_layers, cfg.max_source_positions, L.dtype_code(dt)
        c1 = P(self.conv1.weight.permute(0, 2, 1).reshape(cfg.d_model, -1))
        c2 = P(self.conv2.weight.permute(0, 2, 1).reshape(cfg.d_model, -1))
        w.conv1_w, w.conv1_b = c1.data_ptr(), P(self.conv1.bias).data_ptr()
        w.conv2_w, w.conv2_b = c2.data_ptr(), P(self.conv2.bias).data_ptr()
        w.pos_emb = P(self.embed_positions.weight).data_ptr()
        names = ["ln1_w", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b"]
        lists = {n: [] for n in names}
        for lyr in self.layers:
            a = lyr.self_attn
            lists["ln1_w"].append(P(lyr.self_attn_layer_norm.weight))
            lists["ln1_b"].append(P(lyr.self_attn_layer_norm.bias))
            lists["qkv_w"].append(P(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], dim=0)))
            lists["qkv_b"].append(P(torch.cat([a.q_proj.bias, torch.zeros_like(a.q_proj.bias), a.v_proj.bias], dim=0)))
            lists["out_w"].append(P(a.out_proj.weight))
            lists["out_b"].append(P(a.out_proj.bias))
            lists["ln2_w"].append(P(lyr.final_layer_norm.weight))
            lists["ln2_b"].append(P(lyr.final_layer_norm.bias))
            lists["fc1_w"].append(P(lyr.fc1.weight))
            lists["fc1_b"].append(P(lyr.fc1.bias))
            lists["fc2_w"].append(P(lyr.fc2.weight))
            lists["fc2_b"].append(P(lyr.fc2.bias))
        # bf16 throughput mode: LayerNorm-folded copies of the two projections that follow a LayerNorm (afhip.h,
        # afhip_encoder_weights.qkv_wf ...): W' = W diag(gamma) rounded once to bf16, its f32 row sums (taken from the ROUNDED
        # W', so the mean term cancels exactly what the MFMAs accumulate) and the f32 bias b + W beta.  AFHIP_LN_FOLD=0 keeps
        # every LayerNorm as its own pass (A/B switch).
        fold_names = []
        if dt == torch.bfloat16 and os.environ.get("AFHIP_LN_FOLD", "1") != "0" and cfg.d_model % 256 == 0 and cfg.encoder_ffn_dim % 256 == 0:
            fold_names = ["qkv_wf", "qkv_cs", "qkv_bf", "fc1_wf", "fc1_cs", "fc1_bf"]
            for n in fold_names:
                lists[n] = []
            for i, lyr in enumerate(self.layers):
                for pre, wt, bt, ln in (("qkv", lists["qkv_w"][i], lists["qkv_b"][i], lyr.self_attn_layer_norm),
                                        ("fc1", lists["fc1_w"][i], lists["fc1_b"][i], lyr.final_layer_norm)):
                    w32, g32, b32 = wt.float(), ln.weight.detach().float(), ln.bias.detach().float()
                    rs = torch.ones(w32.shape[0], device=w32.device)
                    if pre == "qkv":
                        # the softmax scale in exp2 units rides on the q rows too (afhip_attn_args.q_prescaled): scores leave the
                        # q.k MFMA ready for exp2, and q is rounded to bf16 once, after the scale
                        rs[: cfg.d_model] = (cfg.d_model // cfg.encoder_attention_heads) ** -0.5 * 1.4426950408889634
                    wf = P((w32 * g32[None, :] * rs[:, None]).to(torch.bfloat16))
                    lists[pre + "_wf"].append(wf)
                    lists[pre + "_cs"].append(P(wf.float().sum(dim=1)))
                    lists[pre + "_bf"].append(P((bt.float() + w32 @ b32) * rs))
        w.q_prescaled = 1 if fold_names else 0
        if self._fp8 and dt == torch.bfloat16:
            from ..utils.quant import quantize_rows_e4m3
            f8_names = []
            for src in ("qkv", "out", "fc1", "fc2"):
                lists[src + "_w8"], lists[src + "_s8"] = [], []
                f8_names += [src + "_w8", src + "_s8"]
                for t in lists[src + "_w"]:
                    q8, sc = quantize_rows_e4m3(t)
                    lists[src + "_w8"].append(P(q8))
                    lists[src + "_s8"].append(P(sc))
            fold_names = fold_names + f8_names
        arrays = {}
        for n in names + fold_names:
            arrays[n] = L.ptr_array(lists[n])
            setattr(w, n, C.cast(arrays[n], L.c_void_pp))
        w.lnf_w, w.lnf_b = P(self.layer_norm.weight).data_ptr(), P(self.layer_norm.bias).data_ptr()
        if self._fp8 and dt == torch.bfloat16:
            if self._fc2_in_scale is not None:
                keep.append(self._fc2_in_scale)                    # HOST array: the library reads it while it builds the launches
                w.fc2_in_scale = self._fc2_in_scale.data_ptr()
            if self._att_out_scale is not None:
                keep.append(self._att_out_scale)                   # HOST array too
                w.att_out_scale = self._att_out_scale.data_ptr()
            if self._calib is not None:
                keep.append(self._calib)
                w.calib_amax = self._calib.data_ptr()
        from .. import torch_ops
        self._packed = SimpleNamespace(w=w, keep=keep, arrays=arrays, blob=torch_ops.weights_blob(w))
        return self._packed

    def _workspace(self, B: int) -> torch.Tensor:
        lib = L.lib()
        need = lib.afhip_encoder_workspace_bytes(C.byref(self.pack().w), B)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---------------------------------------------------------------- forward
    def _get_feat_extract_output_lengths(self, input_lengths):
        """modeling_whisper.py:759-765."""
        input_lengths = (input_lengths - 1) // 2 + 1
        output_lengths = (input_lengths - 2) // 2 + 1
        return input_lengths, output_lengths

    def encode_btc(self, mel_btc: torch.Tensor, feat_len: Optional[torch.Tensor] = None, hidden_layer: Optional[int] = None,
                   ragged: bool = False):
        """mel [B,3000,128] (model dtype, GPU) + optional per-clip key length [B] -> [B,750,d].
        `hidden_layer` (-1 = conv stem, l = output of layer l) additionally returns that [B,1500,d] state.
        `ragged=True` (needs feat_len): the layers run on the valid positions only, packed (afhip_encoder_forward_ragged); rows
        t < (feat_len - 2) // 2 + 1 of each clip are bit-identical to the padded forward, the rows behind them -- which callers
        trim (audio.py:1163-1187) -- are zero instead of values computed from padding."""
        lib = L.lib()
        pk = self.pack()
        cfg = self.config
        if mel_btc.dim() != 3 or mel_btc.shape[1] != 2 * cfg.max_source_positions or mel_btc.shape[2] != cfg.num_mel_bins:
            raise ValueError(f"Qwen2Audio expects the mel input features to be of length {2 * cfg.max_source_positions}, "
                             f"but found {mel_btc.shape[1]}. Make sure to pad the input mel features to {2 * cfg.max_source_positions}.")
        mel_btc = mel_btc.to(device=self.device, dtype=self.dtype).contiguous()
        B = mel_btc.shape[0]
        fl = None
        if feat_len is not None:
            fl = feat_len.to(device=self.device, dtype=torch.int32).contiguous()
        ws = self._workspace(B)
        # the forward itself is the custom op afhip::encoder_forward (torch_ops.py) over the opaque packed-weight blob
        if ragged:
            if feat_len is None or hidden_layer is not None:
                raise ValueError("ragged=True needs feat_len and cannot return hidden states")
            fl_host = feat_len.detach().to(device="cpu", dtype=torch.int32).contiguous()
            return torch.ops.afhip.encoder_forward_ragged(pk.blob, mel_btc, fl, fl_host, ws)
        out, hid = torch.ops.afhip.encoder_forward(pk.blob, mel_btc, fl, hidden_layer if hidden_layer is not None else -2, ws)
        return (out, hid) if hidden_layer is not None else out

    @torch.no_grad()
    def forward(self, input_features, attention_mask=None, head_mask=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, feat_len: Optional[torch.Tensor] = None):
        """Reference signature (modeling_whisper.py:640-648). input_features [B,128,3000]; attention_mask is the
        additive [B,1,1500,1500] key-padding mask callers build (audio.py:1147-1161): it is reduced to a key length
        per clip (its rows are identical by construction); pass `feat_len` instead to skip materialising it."""
        if head_mask is not None or output_attentions:
            raise NotImplementedError("head_mask / output_attentions are not supported by the fused attention kernel")
        exp = self.config.max_source_positions * 2
        if input_features.shape[-1] != exp:
            raise ValueError(f"Qwen2Audio expects the mel input features to be of length {exp}, but found "
                             f"{input_features.shape[-1]}. Make sure to pad the input mel features to {exp}.")
        x = input_features.to(device=self.device)
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        mel_btc = ops.transpose_cast(x.contiguous(), self.dtype)
        if feat_len is None and attention_mask is not None:
            feat_len = (attention_mask[:, 0, 0, :] == 0).sum(-1)
        states = None
        if output_hidden_states:
            # modeling_whisper.py:699-750: the input of every layer (entry 0 = conv stem + positions), then the pooled, normed output.
            # The fused forward keeps one intermediate state per call (`hidden_layer`), so this debugging feature costs one forward per
            # layer; the states come back in the model dtype, as the reference's do.
            hs = []
            for i in range(self.config.encoder_layers):
                out, h = self.encode_btc(mel_btc, feat_len, hidden_layer=i - 1)
                hs.append(h)
            states = tuple(hs) + (out,)
        else:
            out = self.encode_btc(mel_btc, feat_len)
        if return_dict is False:
            return tuple(v for v in (out, states) if v is not None)
        return SimpleNamespace(last_hidden_state=out, hidden_states=states, attentions=None)


class Qwen2AudioMultiModalProjector(nn.Module):
    """Drop-in for modeling_whisper.py:768-775: `linear` (d_model -> hidden_size, bias) over the encoder output, as one
    afhip_gemm with the bias in its epilogue.  Accepts either the reference's nested config (audio_config.d_model /
    text_config.hidden_size) or two integers."""

    def __init__(self, config=None, d_model: Optional[int] = None, hidden_size: Optional[int] = None):
        super().__init__()
        if config is not None:
            d_model, hidden_size = config.audio_config.d_model, config.text_config.hidden_size
        self.linear = _Linear(d_model, hidden_size, bias=True)

    def forward(self, audio_features: torch.Tensor) -> torch.Tensor:
        x = audio_features.contiguous()
        y = ops.gemm(x.reshape(-1, x.shape[-1]), self.linear.weight, bias=self.linear.bias)
        return y.view(*x.shape[:-1], y.shape[-1])


def merge_input_ids_with_audio_features(audio_features, num_audio_tokens, inputs_embeds, input_ids, attention_mask, labels=None, *,
                                        audio_token_index: int, pad_token_id: int = -1, ignore_index: int = -100,
                                        padding_side: str = "left"):
    """Drop-in for Qwen2AudioForConditionalGeneration._merge_input_ids_with_audio_features (modeling_whisper.py:913-1108):
    every <|AUDIO|> placeholder of `input_ids` widens to the `num_audio_tokens` rows of its audio, text embeddings keep
    their order, the batch is re-padded on the side the attention mask shows (or `padding_side` when it cannot tell).

    Returns the reference's 5-tuple (final_embedding [B, L', H], final_attention_mask, final_labels | None, position_ids,
    final_input_ids) on the device of `inputs_embeds`.  The int64 index plan (cumsum of token widths, :1021-1096) is a
    few hundred integers and is built on the host; the embedding rows move through ONE HIP row gather
    (afhip_gather_rows) instead of three index_put passes over a zero-filled tensor.  Raises the reference's ValueErrors
    (:1017-1019 both mask edges zero, :1098-1102 audio rows != placeholder slots)."""
    import numpy as np
    dev = inputs_embeds.device
    if dev.type != "cuda":
        raise L.AfhipError("merge_input_ids_with_audio_features runs on the GPU only (no CPU fallback)")
    n_audio, max_tok, H = audio_features.shape
    B, Lseq = input_ids.shape
    ids = input_ids.detach().cpu().numpy().astype(np.int64)
    am = attention_mask.detach().cpu().numpy().astype(np.int64)
    nat = num_audio_tokens.detach().cpu().numpy().astype(np.int64)
    lp, rp = bool((am[:, 0] == 0).any()), bool((am[:, -1] == 0).any())
    left_padding = True
    if B > 1:
        if lp and not rp:
            left_padding = True
        elif not lp and rp:
            left_padding = False
        elif not lp and not rp:
            left_padding = padding_side == "left"
        else:
            raise ValueError(f"both side of attention_mask has zero, invalid. {attention_mask}")
    special = ids == audio_token_index
    if int(special.sum()) != nat.shape[0]:
        raise ValueError(f"The input provided to the model are wrong. The number of audio tokens is {special.sum(-1)} while"
                         f" the number of audio given to the model is {n_audio}. This prevents correct indexing and breaks batch generation.")
    width = np.ones_like(ids)
    width[special] = nat                       # row-major order of appearance == stacking order of the audios
    new_pos = np.cumsum(width, -1) - 1
    M = int(width.sum(-1).max())
    if left_padding:
        new_pos = new_pos + (M - 1 - new_pos[:, -1])[:, None]
    tb, ts = np.nonzero((~special) & (am == 1))
    td = new_pos[tb, ts]
    plan = np.full((B, M), -1, np.int64)       # -1 = zero row
    audio_slot = np.ones((B, M), bool)
    audio_slot[tb, td] = False
    valid = width.sum(-1) - (am == 0).sum(-1)
    seq = np.arange(M)[None, :]
    audio_slot &= ((M - seq) <= valid[:, None]) if left_padding else (seq < valid[:, None])
    if int(audio_slot.sum()) != int(nat.sum()):
        raise ValueError(f"The input provided to the model are wrong. The number of audio tokens is {special.sum(-1)} while"
                         f" the number of audio given to the model is {n_audio}. This prevents correct indexing and breaks batch generation.")
    keep = np.arange(max_tok)[None, :] < nat[:, None]
    audio_rows = np.flatnonzero(keep.reshape(-1))                    # row index into audio_features.view(-1, H), stacking order
    plan[tb, td] = tb * Lseq + ts                                     # row index into inputs_embeds.view(-1, H)
    plan.reshape(-1)[np.flatnonzero(audio_slot.reshape(-1))] = -(audio_rows + 2)
    fmask = np.zeros((B, M), np.int64)
    fmask[tb, td] = am[tb, ts]
    fmask |= audio_slot
    fids = np.full((B, M), pad_token_id, np.int64)
    fids[tb, td] = ids[tb, ts]
    pos = np.cumsum(fmask, -1) - 1
    pos[fmask == 0] = 1
    final_labels = None
    if labels is not None:
        lab = labels.detach().cpu().numpy().astype(np.int64)
        fl = np.full((B, M), ignore_index, np.int64)
        fl[tb, td] = lab[tb, ts]
        final_labels = torch.from_numpy(fl).to(dev)
    feats = audio_features.to(device=dev, dtype=inputs_embeds.dtype).contiguous()
    plan_d = torch.from_numpy(plan.reshape(-1).astype(np.int32)).to(dev)
    emb = ops.gather_rows(inputs_embeds.contiguous().view(-1, H), feats.view(-1, H), plan_d, B * M).view(B, M, H)
    return (emb, torch.from_numpy(fmask).to(device=dev, dtype=attention_mask.dtype), final_labels,
            torch.from_numpy(pos).to(dev), torch.from_numpy(fids).to(device=dev, dtype=input_ids.dtype))


def __getattr__(name):
    # `Qwen2AudioForConditionalGeneration` lives in qwen2_audio_generation.py (it builds on the LLM side) but is importable from
    # here, where the reference defines it (modeling_whisper.py:855)
    if name == "Qwen2AudioForConditionalGeneration":
        from .qwen2_audio_generation import Qwen2AudioForConditionalGeneration
        return Qwen2AudioForConditionalGeneration
    raise AttributeError(name)
