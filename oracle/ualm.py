"""UALM wrapper: vocabulary, batch dict, splice, step, greedy loop (oracle; test infrastructure only).

Follows /root/reference/UALM/models/ualm:
  * ualm_job.py:71-110      _build_vocabulary (256 special slots, then each discrete IO)
  * ualm_job.py:311-418     UALMPreprocessor.preprocessing (bos, role, modality, content, eot/eos)
  * ualm_job.py:219-309     collate_fn (bucket) + utils/data.py:16-90 pad_list
  * lm/parallel.py:219-284  _embed (embedding sum over streams, adaptor, splice)
  * lm/parallel.py:535-568  prepare_inference (special tokens, modality / per-IO masks)
  * lm/parallel.py:570-597  _step (stack, stream_emb with stream 0 zeroed, lm_head, mask)
  * lm/parallel.py:428-533  inference_segment (greedy branch: temperature 0, cfg 1, num_hypo 1)
  * lm/parallel.py:599-601  _logits_to_token greedy
The text / discrete-audio IOs are described only by their vocabulary contract (stream intervals);
no tokenizer or codec exists offline.
"""

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import afwhisper, qwen2, logmel

SPECIAL = ["<|pad|>", "<|bos|>", "<|eos|>", "<|eot|>", "<|system|>", "<|user|>", "<|assistant|>",
           "<|text|>", "<|audio|>", "<|speech|>", "<|image|>", "<|video|>", "<|toolcall|>"]
NUM_SPECIAL = 256

TASKS = {  # task_conf_ualm.py:18-31
    "text_only": [("user", "text1"), ("assistant", "text2")],
    "caption_to_audio": [("user", "text1"), ("assistant", "audio1")],
    "audio_to_caption": [("user", "text1"), ("user", "audio1"), ("assistant", "text2")],
    "audio_to_conversation": [("user", "audio1"), ("user", "text1"), ("assistant", "text2")],
    "audio_only": [("user", "text1"), ("assistant", "audio1")],
    "transcription_to_speech": [("user", "text1"), ("assistant", "audio1")],
    "speech_to_transcription": [("user", "text1"), ("user", "audio1"), ("assistant", "text2")],
}


def build_vocabulary(text_vocab: int, audio_streams: int = 8, audio_codebook: int = 1025):
    """ualm_job.py:71-110 with IO order {text, discrete_audio, continuous_audio} (conf/train.yaml).

    Returns (vocab list, vocab_intervals dict). Token strings for the IO vocabularies are synthetic
    placeholders (only ids matter to the hot path)."""
    vocab = list(SPECIAL)
    while len(vocab) < NUM_SPECIAL:
        vocab.append(f"<|unused_{len(vocab)}|>")
    intervals = {"special_token": [(0, NUM_SPECIAL)]}
    start = NUM_SPECIAL
    vocab.extend(f"<text_{i}>" for i in range(text_vocab))
    intervals["text"] = [(start, start + text_vocab)]
    start = len(vocab)
    vocab.extend(f"<audio_{i}>" for i in range(audio_streams * audio_codebook))
    intervals["discrete_audio"] = [(start + s * audio_codebook, start + (s + 1) * audio_codebook)
                                   for s in range(audio_streams)]
    return vocab, intervals


def special_id(tok: str) -> int:
    return SPECIAL.index(tok)


def preprocessing(task: str, text_ids: Dict[str, np.ndarray], wav: Optional[np.ndarray],
                  intervals, num_stream: int = 8, is_train: bool = False):
    """UALMPreprocessor.preprocessing (ualm_job.py:311-418), inference branch.

    text_ids: {"text1": int array} -- already-tokenised text (the tokenizer is out of reach offline).
    Returns dict(sequence [T,S] int64, loss_mask [T,S] f32, conti_feats [(io,start,len,feat)])."""
    def sp(tok):
        r = np.zeros((1, num_stream), dtype=np.int64)
        r[0, 0] = special_id(tok)
        return r

    def smask(v):
        r = np.zeros((1, num_stream), dtype=np.float32)
        r[0, 0] = v
        return r

    msgs = []
    for role, entry in TASKS[task]:
        if role == "assistant" and not is_train:
            break
        if entry.startswith("audio"):
            msgs.append((role, "continuous_audio" if role in ("user", "system") else "discrete_audio", wav))
        else:
            msgs.append((role, "text", text_ids[entry]))
    seq, masks, conti = [sp("<|bos|>")], [smask(0.0)], []
    accum = 1
    eots = [a[0] == b[0] for a, b in zip(msgs[:-1], msgs[1:])] + [False]
    for apply_eot, (role, io, data) in zip(eots, msgs):
        apply_loss = float(role == "assistant")
        seq.append(sp(f"<|{role}|>"))
        masks.append(smask(apply_loss))
        if io == "text":
            seq.append(sp("<|text|>"))
        else:
            seq.append(sp("<|speech|>" if task in ("transcription_to_speech", "speech_to_transcription") else "<|audio|>"))
        masks.append(smask(apply_loss))
        accum += 2
        if io == "text":
            ids = np.asarray(data, dtype=np.int32).reshape(-1, 1)
            bias = intervals["text"][0][0]
            this = np.where(ids == 0, 0, ids + bias)           # ualm_job.py:370-373
            lm = np.ones_like(ids, dtype=np.float32)
            cf = None
        else:
            pads, cf, lm = logmel.preprocess(data)
            this = pads
            lm = lm.astype(np.float32)
        if this.shape[1] < num_stream:
            this = np.pad(this, ((0, 0), (0, num_stream - this.shape[1])))
            lm = np.pad(lm, ((0, 0), (0, num_stream - lm.shape[1])))
        seq.append(this.astype(np.int64))
        if cf is not None:
            conti.append((io, accum, cf[0], cf[1]))
        masks.append(lm * apply_loss)
        accum += this.shape[0]
        seq.append(sp("<|eot|>" if apply_eot else "<|eos|>"))
        masks.append(smask(apply_loss))
        accum += 1
    return {"sequence": np.concatenate(seq, 0), "loss_mask": np.concatenate(masks, 0), "conti_feats": conti}


def collate(samples: List[dict]) -> Dict[str, torch.Tensor]:
    """collate_fn bucket mode (ualm_job.py:219-309): right-pad, stack, gather continuous feats."""
    T = max(s["sequence"].shape[0] for s in samples)
    S = samples[0]["sequence"].shape[1]
    seqs = torch.zeros(len(samples), T, S, dtype=torch.int64)
    lms = torch.zeros(len(samples), T, S, dtype=torch.float32)
    idx, feats = [], []
    for b, s in enumerate(samples):
        n = s["sequence"].shape[0]
        seqs[b, :n] = torch.from_numpy(s["sequence"])
        lms[b, :n] = torch.from_numpy(s["loss_mask"])
        for io, start, length, feat in s["conti_feats"]:
            idx.append((b, start, length))
            feats.append(torch.from_numpy(feat))
    out = {"seqs": seqs, "loss_masks": lms}
    if feats:
        out["continuous_audio_indices"] = torch.tensor(idx, dtype=torch.long)
        out["continuous_audio_feats"] = torch.stack(feats)            # all [3000,128]
        out["continuous_audio_lengths"] = torch.tensor([f.shape[0] for f in feats], dtype=torch.long)
    return out


def masks(vocab_size: int, intervals, num_stream: int = 8):
    """prepare_inference (lm/parallel.py:535-568): returns dict of bool masks [S,V] (True = forbidden)."""
    out = {}
    m = torch.ones(num_stream, vocab_size, dtype=torch.bool)
    for tok in ["audio", "text", "image", "video", "toolcall"]:
        m[0, special_id(f"<|{tok}|>")] = False
    m[1:, 0] = False
    out["modality"] = m
    eot, eos = special_id("<|eot|>"), special_id("<|eos|>")
    for io, iv in intervals.items():
        m = torch.ones(num_stream, vocab_size, dtype=torch.bool)
        for i, (s, e) in enumerate(iv):
            m[i, s:e] = False
        for i in range(len(iv), num_stream):
            m[i, 0] = False
        m[0, eot] = False
        m[0, eos] = False
        out["audio" if io == "discrete_audio" else io] = m
    return out


@torch.no_grad()
def embed(input_ids: torch.Tensor, batch: dict, sd, enc_sd, enc_cfg, sdpa=True) -> torch.Tensor:
    """_embed (lm/parallel.py:219-284), continuous branch: ids [B,T,S] -> [B,T,H]."""
    emb = F.embedding(input_ids, sd["model.embed_tokens.weight"]).sum(dim=2)
    if "continuous_audio_feats" in batch:
        feats = afwhisper.encode_batch(batch["continuous_audio_feats"].float(), batch["continuous_audio_lengths"],
                                       enc_sd, enc_cfg, sdpa=sdpa)
        for feat, (b, start, length) in zip(feats, batch["continuous_audio_indices"].tolist()):
            f = F.linear(feat, sd["adaptor.continuous_audio.weight"], sd["adaptor.continuous_audio.bias"])
            emb[b, start:start + length] = f[:length]
    return emb


@torch.no_grad()
def step(sd, cfg, input_ids=None, input_embeds=None, cache=None, mask=None):
    """_step (lm/parallel.py:570-597) -> (logits [B,T,S,V], cache)."""
    assert (input_ids is None) != (input_embeds is None)
    if input_ids is not None:
        input_embeds = F.embedding(input_ids, sd["model.embed_tokens.weight"]).sum(dim=2)
    h, cache = qwen2.forward(input_embeds, sd, cfg, cache)
    se = sd["stream_emb.weight"].clone()
    se[0] = 0.0
    hs = h.unsqueeze(2) + se[None, None]
    logits = F.linear(hs, sd["lm_head.weight"])
    if mask is not None:
        logits.masked_fill_(mask[None, None], float("-inf"))
    return logits, cache


@torch.no_grad()
def inference_segment(batch: dict, sd, cfg, enc_sd, enc_cfg, intervals, max_step: int,
                      enforce_modality: Optional[str] = "text", num_stream: int = 8, sdpa=True,
                      stop_on_eos: bool = True, return_margins: bool = False, input_embeds: Optional[torch.Tensor] = None):
    """inference_segment greedy (lm/parallel.py:428-533), B==1, num_hypo 1, cfg 1.

    Returns (tokens [n,S] int64, modality str[, margins list]) -- tokens truncated at the first eos/eot
    (inclusive) like the reference (:529-531)."""
    V = sd["lm_head.weight"].shape[0]
    mk = masks(V, intervals, num_stream)
    ids = batch["seqs"]
    assert ids.shape[0] == 1
    a = torch.zeros(1, 1, num_stream, dtype=torch.long)
    a[0, 0, 0] = special_id("<|assistant|>")
    ids = torch.cat([ids, a], dim=1)
    # input_embeds: the caller already spliced the continuous features (long-audio path: one entry per 30-s window)
    emb = embed(ids, batch, sd, enc_sd, enc_cfg, sdpa) if input_embeds is None else input_embeds
    logits, cache = step(sd, cfg, input_embeds=emb, mask=mk["modality"])
    logits = logits[:, -1:]
    if enforce_modality is not None:
        tok = torch.zeros(1, 1, num_stream, dtype=torch.long)
        tok[0, 0, 0] = special_id(f"<|{enforce_modality}|>")
    else:
        tok = logits.argmax(3)
    modality = SPECIAL[int(tok.flatten()[0])].replace("<|", "").replace("|>", "")
    mmask = mk[modality]
    eos, eot = special_id("<|eos|>"), special_id("<|eot|>")
    hyp, margins = [], []
    finish = -1
    prev = tok
    for st in range(max_step):
        logits, cache = step(sd, cfg, input_ids=prev, cache=cache, mask=mmask)
        if return_margins:
            top2 = torch.topk(logits[0, 0, 0], 2).values
            margins.append(float(top2[0] - top2[1]))
        prev = logits.argmax(-1)
        hyp.append(prev)
        if stop_on_eos and finish < 0 and int(prev[0, 0, 0]) in (eos, eot):
            finish = st
            break
    if finish < 0:
        finish = st
    hyp = torch.cat(hyp, dim=1)[0][: finish + 1]
    if return_margins:
        return hyp, modality, margins
    return hyp, modality


# ------------------------------------------------------------------------------------------- sampling / guidance
def cfg_mix(logits: torch.Tensor, cfg_logits: torch.Tensor, cfg: float, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """lm/parallel.py:489-492: `logits * cfg + cfg_logits * (1 - cfg)` as separate tensor ops (each rounds to the tensors' dtype),
    then the modality mask again."""
    out = logits * cfg + cfg_logits * (1 - cfg)
    if mask is not None:
        out = out.masked_fill(mask, float("-inf"))
    return out


def topk_probs(logits: torch.Tensor, temperature: float, topk: int):
    """lm/parallel.py:603-604: (topk_values, topk_indices, softmax(topk_values / temperature))."""
    vals, idx = torch.topk(logits, topk)
    return vals, idx, torch.softmax(vals / temperature, dim=-1)


def inverse_cdf_pick(probs: torch.Tensor, idx: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """The draw the HIP path uses in place of torch.multinomial (same distribution): first j with cumsum(p)[j] > u."""
    cdf = torch.cumsum(probs.float(), dim=-1)
    j = (cdf <= u.unsqueeze(-1)).sum(-1).clamp(max=probs.shape[-1] - 1)
    return torch.gather(idx, -1, j.unsqueeze(-1)).squeeze(-1)
