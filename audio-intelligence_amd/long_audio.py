"""Long-audio path (BASELINE config 4): a clip longer than 30 s is cut into independent 30-s windows (the AF3 recipe the
vendored SoundTower serves, sound_encoder.py:81-107); the windows of a batch are sharded over the GPUs of one node, the
audio tokens come back with ONE all-gather over RCCL/xGMI, and each clip's owner GPU splices all its windows
into one prompt -- one `(b, start, len)` entry per window, exactly the `conti_feats` tuples `ParallelLLM._embed` already
consumes (lm/parallel.py:262-284) -- and runs the causal prefill (about 15 000 audio positions for 10 minutes) and the
greedy decode.  The gather moves encoder outputs (d_model wide, pre-adaptor): 2.8x fewer bytes than post-adaptor embeddings.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm); nothing here runs on the data path of short clips,
which shard as independent replicas with no collective at all.  The reference has no sequence / tensor parallelism to mirror
(SURVEY 2.3), so the LLM side of a long clip stays on one GPU.
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

WINDOW_SAMPLES = 480000


def split_windows(n_samples: int, window: int = WINDOW_SAMPLES) -> List[Tuple[int, int]]:
    """[start, end) sample ranges of the 30-s windows covering a clip (last one may be short; it is zero-padded
    by the log-mel kernel and masked through its sample count)."""
    if n_samples <= 0:
        return []
    return [(s, min(s + window, n_samples)) for s in range(0, n_samples, window)]


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [lo, hi); blocks differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def encode_windows_sharded(encode_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], windows: torch.Tensor,
                           n_valid: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                           out_spec: Optional[Tuple[int, int, torch.dtype]] = None, timing: Optional[dict] = None,
                           total_windows: Optional[int] = None) -> torch.Tensor:
    """windows [W, n<=480000] f32 waveform windows, n_valid [W] valid sample counts.
    Each rank encodes its contiguous block with `encode_fn(wav_block, n_valid_block) -> [w_local, T, d]` and ONE
    all_gather_into_tensor returns the full [W, T, d] on every rank (window order preserved); there is no other collective and
    no host synchronisation.
    `total_windows` = None: `windows` / `n_valid` hold ALL W windows on every rank (each rank slices its block out).
    `total_windows` = W: they hold ONLY this rank's block [lo, hi) = shard_range(W, rank, world) -- each rank uploads its own
    windows and nothing else (154 MB per rank stay on the host for config 4's 80 windows on 8 GPUs).
    `out_spec` = (T, d, dtype) of one window's tokens -- static facts of the encoder (750, d_model, model dtype) that a rank owning
    no window needs to size its send buffer; taken from `encode_fn.out_spec` when omitted (make_tower_encode_fn sets it).  Works
    with world size 1 and without an initialised process group.
    `timing`: optional dict that receives HIP events around the collective (`gather_ev`) and its byte count (`gather_bytes`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return encode_fn(windows, n_valid)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return encode_fn(windows, n_valid)
    spec = out_spec if out_spec is not None else getattr(encode_fn, "out_spec", None)
    if spec is None:
        raise ValueError("encode_windows_sharded needs out_spec=(tokens_per_window, d_model, dtype) when run on more than one rank")
    T, D, dtype = spec
    W = windows.shape[0] if total_windows is None else int(total_windows)
    lo, hi = shard_range(W, rank, world)
    if total_windows is None:
        windows, n_valid = windows[lo:hi], n_valid[lo:hi]
    elif windows.shape[0] != hi - lo or n_valid.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} of {world} owns windows [{lo}, {hi}) of {W} but was handed {windows.shape[0]}")
    per = (W + world - 1) // world                      # padded block size so the gather is a single fixed-size call
    out = torch.empty((world * per, T, D), dtype=dtype, device=windows.device)
    buf = torch.empty((per, T, D), dtype=dtype, device=windows.device)
    if hi > lo:
        local = encode_fn(windows, n_valid)
        if tuple(local.shape[1:]) != (T, D) or local.dtype != dtype:
            raise ValueError(f"encode_fn returned {tuple(local.shape)} {local.dtype}, out_spec says [*, {T}, {D}] {dtype}")
        buf[: hi - lo].copy_(local)
    if hi - lo < per:
        buf[hi - lo:].zero_()
    ev = None
    if timing is not None and windows.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.all_gather_into_tensor(out, buf, group=group)
    if ev is not None:
        ev[1].record()
        timing["gather_ev"] = ev
        timing["gather_bytes"] = buf.numel() * buf.element_size()
    if W == world * per:
        return out
    pieces = []
    for r in range(world):
        l, h = shard_range(W, r, world)
        pieces.append(out[r * per: r * per + (h - l)])
    return torch.cat(pieces, dim=0)


def make_tower_encode_fn(audio_io) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    """encode_fn for `encode_windows_sharded` backed by a ContinuousAudioIO: wav windows -> log-mel (HIP) ->
    AF-Whisper encoder (HIP) with the SoundTower length rule (mask.sum(-1) = valid mel frames, sound_encoder.py:86)."""
    def fn(wav: torch.Tensor, n_valid: torch.Tensor) -> torch.Tensor:
        mel = audio_io.processor.extract_device(wav, layout="btc", dtype=audio_io.dtype)
        frames = (n_valid.to(torch.long) // audio_io.hop_length).clamp(max=3000)
        feat_len, _ = audio_io.model._get_feat_extract_output_lengths(frames)
        return audio_io.model.encode_btc(mel, feat_len=feat_len)
    fn.out_spec = (audio_io.model.config.max_source_positions // 2, audio_io.d_model, audio_io.dtype)
    return fn


def window_token_counts(n_samples: int, hop: int = 160, window: int = WINDOW_SAMPLES) -> List[int]:
    """Audio tokens each window of a clip contributes to the prompt: the after-length formula of audio.py:1073,1094-1095
    applied to the window's valid samples (750 for a full window)."""
    out = []
    for a, b in split_windows(n_samples, window):
        before = (b - a) // hop
        after = (before - 1) // 2 + 1
        out.append(int((after - 2) // 2 + 1))
    return out


def build_long_prompt(prompt_ids: Sequence[int], n_samples: int, text_offset: int, num_stream: int = 8,
                      special: Optional[Dict[str, int]] = None):
    """The `audio_to_caption` layout of UALMPreprocessor.preprocessing (ualm_job.py:311-418) for ONE clip whose audio message is
    the concatenation of its 30-s windows: bos, user, text, prompt ids (+offset), eot, user, audio, sum(tokens) pad rows, eos.
    Returns (seq [T, S] int64, entries [(start, len)] one per window)."""
    sp = special or {"bos": 1, "eos": 2, "eot": 3, "user": 5, "text": 7, "audio": 8}
    counts = window_token_counts(n_samples)
    rows = [sp["bos"], sp["user"], sp["text"]] + [(0 if t == 0 else int(t) + text_offset) for t in prompt_ids] + [sp["eot"], sp["user"], sp["audio"]]
    start = len(rows)
    entries = []
    for c in counts:
        entries.append((start, c))
        start += c
    rows += [0] * sum(counts) + [sp["eos"]]
    seq = torch.zeros((len(rows), num_stream), dtype=torch.int64)
    seq[:, 0] = torch.tensor(rows, dtype=torch.int64)
    return seq, entries


@torch.no_grad()
def long_audio_inference(model, audio_io, clips: Sequence[np.ndarray], prompts: Sequence[Sequence[int]], inference_config: dict,
                         enforce_modality: Optional[str] = "text", group: Optional[dist.ProcessGroup] = None,
                         io_name: str = "continuous_audio", timing: Optional[dict] = None):
    """Config 4 end to end on one node.  `clips`: 16 kHz mono waveforms of any length, the same list on every rank (host memory; each rank uploads only its own windows); `prompts`: text
    ids per clip.  Steps: (1) every clip is cut into 30-s windows and ALL windows of the batch are sharded over the ranks
    (window w of the flat list -> the rank whose contiguous block holds it); (2) log-mel + encoder on the local block;
    (3) one all-gather returns every window's [750, d] tokens to every rank; (4) clip c is owned by rank c % world: the owner
    builds the prompt with one (b, start, len) entry per window, `_embed` projects + splices them (lm/parallel.py:277-282),
    then causal prefill and greedy decode exactly as `inference_segment` does for a short clip.
    Returns {clip_index: (token ids LongTensor [n, S], modality)} for the clips THIS rank owns (every clip when not distributed)."""
    dev = model.device
    dist_on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if dist_on else 0
    world = dist.get_world_size(group) if dist_on else 1
    spans, owner_of = [], []
    for c, wav in enumerate(clips):
        for a, b in split_windows(len(wav)):
            spans.append((c, a, b))
    W = len(spans)
    lo, hi = shard_range(W, rank, world)                  # this rank cuts, pads and uploads ONLY its own block of windows
    wins = torch.zeros((hi - lo, WINDOW_SAMPLES), dtype=torch.float32)
    for i, (c, a, b) in enumerate(spans[lo:hi]):
        wins[i, : b - a] = torch.from_numpy(np.ascontiguousarray(clips[c][a:b], dtype=np.float32))
    n_valid = torch.tensor([b - a for _, a, b in spans[lo:hi]], dtype=torch.long)
    tokens = encode_windows_sharded(make_tower_encode_fn(audio_io), wins.to(dev), n_valid.to(dev), group=group, timing=timing,
                                    total_windows=W)
    text_offset = model.vocab_intervals["text"][0][0]
    results = {}
    for c, wav in enumerate(clips):
        if c % world != rank:
            continue
        seq, entries = build_long_prompt(prompts[c], len(wav), text_offset, model.num_stream)
        idx = [i for i, (cc, _, _) in enumerate(spans) if cc == c]
        batch = {"seqs": seq[None].to(dev),
                 f"{io_name}_indices": torch.tensor([[0, s, n] for s, n in entries], dtype=torch.long),
                 f"{io_name}_encoded": tokens[idx[0]: idx[-1] + 1]}
        hyps, _ = model.inference_segment(inference_config, cache=None, enforce_modality=enforce_modality, **batch)
        results[c] = hyps[0]
    return results
