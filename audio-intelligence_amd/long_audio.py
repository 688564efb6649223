"""Long-audio path: a clip longer than 30 s is cut into independent 30-s windows (the AF3 recipe the vendored
SoundTower serves, sound_encoder.py:81-107); the windows of a batch are sharded over the GPUs of one node and the
audio tokens come back with ONE all-gather over RCCL/xGMI.  The gather moves encoder outputs
(d_model wide, pre-adaptor): 2.8x fewer bytes than post-adaptor embeddings.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm); nothing here runs on the data path of
short clips, which shard as independent replicas with no collective at all.
"""
from typing import Callable, List, Optional, Tuple

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
                           n_valid: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """windows [W, n<=480000] f32 waveform windows (identical on every rank), n_valid [W] valid sample counts.
    Each rank encodes its contiguous block with `encode_fn(wav_block, n_valid_block) -> [w_local, 750, d]` and one
    all_gather returns the full [W, 750, d] on every rank (window order preserved).  Works with world size 1 and
    without an initialised process group."""
    W = windows.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return encode_fn(windows, n_valid)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(W, rank, world)
    per = (W + world - 1) // world                      # padded block size so the gather is a single fixed-size call
    if hi > lo:
        local = encode_fn(windows[lo:hi], n_valid[lo:hi])
    else:
        local = None
    # shape of one window's tokens must be known on ranks that own nothing: broadcast it from rank 0's result
    shape = torch.zeros(3, dtype=torch.long, device=windows.device)
    dt_code = torch.zeros(1, dtype=torch.long, device=windows.device)
    if local is not None:
        shape = torch.tensor([1, local.shape[1], local.shape[2]], dtype=torch.long, device=windows.device)
        dt_code = torch.tensor([{torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[local.dtype]], device=windows.device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(dt_code, op=dist.ReduceOp.MAX, group=group)
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(dt_code.item())]
    T, D = int(shape[1]), int(shape[2])
    buf = torch.zeros((per, T, D), dtype=dtype, device=windows.device)
    if local is not None:
        buf[: hi - lo] = local
    out = torch.empty((world * per, T, D), dtype=dtype, device=windows.device)
    dist.all_gather_into_tensor(out, buf, group=group)
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
    return fn
