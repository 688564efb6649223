"""Thin inference driver and checkpoint ingestion.

Mirrors the call sequence of the reference's `scripts/inference.py` for the audio-understanding path, minus its CLI,
logging, dataset registry and multi-process launch (out of scope):

    load_checkpoint          scripts/inference.py:136-153     DeepSpeed `mp_rank_00_model_states.pt`["module"], strict
    to_device                utils/data.py:93-130             floats -> model dtype, ints untouched, lists / dicts recursed
    run_inference            scripts/inference.py:270-304     collate -> to_device -> model.inference, per-sample errors kept

Host-side only: every tensor op on the data path goes through the HIP library via the model classes.
"""
import os
from typing import Callable, Iterable, Optional

import numpy as np
import torch


def find_checkpoint_file(path: str) -> str:
    """Accepts the file itself, a DeepSpeed tag directory, or a DeepSpeed save directory holding a `latest` tag file
    (deepspeed.save_checkpoint layout: <dir>/<tag>/mp_rank_00_model_states.pt, <dir>/latest)."""
    if os.path.isfile(path):
        return path
    direct = os.path.join(path, "mp_rank_00_model_states.pt")
    if os.path.isfile(direct):
        return direct
    latest = os.path.join(path, "latest")
    if os.path.isfile(latest):
        with open(latest) as f:
            tag = f.read().strip()
        cand = os.path.join(path, tag, "mp_rank_00_model_states.pt")
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError(f"no mp_rank_00_model_states.pt under {path}")


def load_checkpoint(model, checkpoint_path: str, strict: bool = True):
    """scripts/inference.py:136-153.  `weights_only=True` (nothing from the file is executed); KeyError when the file has
    no "module" entry, RuntimeError from load_state_dict on a key / shape mismatch -- as the reference.  The packed device
    copies (fused q|k|v, LayerNorm-folded and fp8 weights) are rebuilt lazily at the next forward."""
    checkpoint = torch.load(find_checkpoint_file(checkpoint_path), map_location="cpu", weights_only=True)
    state_dict = checkpoint["module"]
    model.load_state_dict(state_dict, strict=strict)
    return model


def save_checkpoint(model, directory: str, tag: str = "global_step0") -> str:
    """Writes the layout load_checkpoint reads (for tests and for exporting synthetic weights): <dir>/<tag>/
    mp_rank_00_model_states.pt with {"module": state_dict} and <dir>/latest."""
    os.makedirs(os.path.join(directory, tag), exist_ok=True)
    path = os.path.join(directory, tag, "mp_rank_00_model_states.pt")
    torch.save({"module": {k: v.detach().cpu() for k, v in model.state_dict().items()}}, path)
    with open(os.path.join(directory, "latest"), "w") as f:
        f.write(tag)
    return path


def to_device(data, device=None, dtype=None, non_blocking: bool = False):
    """utils/data.py:93-130: tensors move to `device`; floating tensors are cast to `dtype`, integer tensors keep theirs;
    dicts / lists / tuples are walked; anything else is returned as is."""
    if isinstance(data, dict):
        return {k: to_device(v, device, dtype, non_blocking) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return type(data)(to_device(v, device, dtype, non_blocking) for v in data)
    if isinstance(data, np.ndarray):
        return to_device(torch.from_numpy(data), device, dtype, non_blocking)
    if isinstance(data, torch.Tensor):
        if dtype is not None and data.is_floating_point():
            return data.to(device=device, dtype=dtype, non_blocking=non_blocking)
        return data.to(device=device, non_blocking=non_blocking)
    return data


def _decode_batch(model, batches, inference_config, enforce_modality):
    """Several single-sample batch dicts with the SAME prompt length and audio placement -> one inference_segment call (B > 1).
    Per-sample results equal the B = 1 results (tests/test_gpu_llm.py, tests/test_gpu_bf16.py: batch invariance)."""
    merged = {}
    for k in batches[0]:
        vals = [b[k] for b in batches]
        if k.endswith("_indices"):
            merged[k] = torch.cat([torch.cat([torch.full_like(v[:, :1], i), v[:, 1:]], dim=1) for i, v in enumerate(vals)])
        else:
            merged[k] = torch.cat(vals)
    hyps, _ = model.inference_segment(inference_config, cache=None, enforce_modality=enforce_modality, **merged)
    return [[["assistant", modality, seq]] for seq, modality in hyps]


def run_inference(model, preprocessor, samples: Iterable, inference_config: dict, device: str = "cuda", dtype=None,
                  on_result: Optional[Callable] = None, enforce_modality: Optional[str] = None, batch_size: int = 1) -> dict:
    """scripts/inference.py:270-304 for one shard: `samples` yields (key, data_dict) with key = (task, data_name, example_id)
    as the reference's iterator does; each is collated alone (the reference's B = 1), moved to the device, decoded with
    `model.inference`, and its messages are stored under example_id.  A sample that raises is recorded as
    {"error": ...} and the loop continues (the reference logs and continues, :277-279).  Returns {example_id: messages}.
    `enforce_modality` (not in the reference driver) decodes one segment of that modality through `inference_segment` and
    returns raw token ids -- what a randomly initialised model needs, since its free choice of modality token can land on an
    IO that was never configured.
    `batch_size` > 1 (with `enforce_modality`): consecutive samples whose prompts have the same length and audio placement are
    decoded in ONE batched prefill / decode (the reference cannot: its assistant token is [1,1,S], lm/parallel.py:438); a sample
    that does not fit the open group, or a group that fails, falls back to one-by-one so error isolation is unchanged."""
    dtype = dtype if dtype is not None else next(model.parameters()).dtype
    results = {}
    order = []

    def emit(idx, example_id, value):
        results[example_id] = value
        if on_result is not None:
            on_result(idx, example_id, value)

    def to_lists(messages):
        out = []
        for role, modality, content in messages:
            if isinstance(content, torch.Tensor):
                content = content.detach().cpu().tolist()
            out.append([role, modality, content])
        return out

    def one(idx, example_id, batch):
        try:
            if enforce_modality is not None:
                hyps, _ = model.inference_segment(inference_config, cache=None, enforce_modality=enforce_modality, **batch)
                messages = [["assistant", modality, seq] for seq, modality in hyps]
            else:
                messages, _ = model.inference(inference_config, **batch)
            emit(idx, example_id, to_lists(messages))
        except Exception as e:  # noqa: BLE001  (per-sample isolation is the reference's behaviour)
            emit(idx, example_id, {"error": f"{type(e).__name__}: {e}"})

    group = []      # [(idx, example_id, batch)] with one shape signature

    def signature(batch):
        return tuple((k, tuple(v.shape), tuple(v[:, 1:].flatten().tolist()) if k.endswith("_indices") else None) for k, v in sorted(batch.items()))

    def flush():
        if not group:
            return
        if len(group) == 1:
            one(*group[0])
        else:
            try:
                outs = _decode_batch(model, [g[2] for g in group], inference_config, enforce_modality)
                for (idx, example_id, _), messages in zip(group, outs):
                    emit(idx, example_id, to_lists(messages))
            except Exception:  # noqa: BLE001  (isolate: redo the group one by one)
                for g in group:
                    one(*g)
        group.clear()

    for idx, (key, data) in enumerate(samples):
        example_id = key[2]
        try:
            batch = preprocessor.collate_fn([(key, data)])
            batch = to_device(batch, device, dtype=dtype)
            batch.pop("keys", None)
            batch.pop("loss_masks", None)
            batch = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
        except Exception as e:  # noqa: BLE001
            flush()
            emit(idx, example_id, {"error": f"{type(e).__name__}: {e}"})
            continue
        if batch_size <= 1 or enforce_modality is None:
            one(idx, example_id, batch)
            continue
        if group and (signature(group[0][2]) != signature(batch) or len(group) >= batch_size):
            flush()
        group.append((idx, example_id, batch))
    flush()
    return results
