#!/usr/bin/env python3
"""Capture golden vectors for the AF3 / Qwen2-Audio placeholder merge from the REAL reference function
(build container only; test infrastructure).

`Qwen2AudioForConditionalGeneration` cannot be constructed under transformers 5.x, so the function object
`_merge_input_ids_with_audio_features` (modeling_whisper.py:913-1108) is called with a stand-in `self` exposing
config.audio_token_index / config.ignore_index / pad_token_id / padding_side.  Cases: the four layouts of its docstring
(:939-997) plus seeded random batches.  Writes tests/golden/golden_merge.npz (inputs and expected outputs; data only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_merge.py
"""
import os
import sys
import types

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import transformers  # noqa: F401
import importlib.util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIO, PAD, IGN = 99, -1, -100


def load_fn():
    spec = importlib.util.spec_from_file_location("ref_modeling_whisper", "/root/reference/UALM/models/ualm/multimodal_io/modeling_whisper.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Qwen2AudioForConditionalGeneration._merge_input_ids_with_audio_features


def case_from_rows(rows, widths, H, rng, labels=True, side="left"):
    """rows: list of token lists with AUDIO placeholders and None for padding."""
    L = max(len(r) for r in rows)
    ids = np.array([[PAD if t is None else t for t in r] for r in rows], np.int64)
    am = np.array([[0 if t is None else 1 for t in r] for r in rows], np.int64)
    n = len(widths)
    T = max(widths)
    feats = rng.standard_normal((n, T, H)).astype(np.float32)
    emb = rng.standard_normal((len(rows), L, H)).astype(np.float32)
    lab = np.where(am == 1, ids, IGN) if labels else None
    return dict(audio_features=feats, num_audio_tokens=np.array(widths, np.int64), inputs_embeds=emb, input_ids=ids,
                attention_mask=am, labels=lab, padding_side=side)


def main():
    fn = load_fn()
    rng = np.random.default_rng(14)
    H = 8
    a = list(range(1, 30))
    X = AUDIO
    cases = {}
    # docstring example, right padding (:947-959) and left padding (:960-973): X 5 tokens, Y 3, Z 8
    r1 = a[0:6] + [X] + a[6:11] + [X] + a[11:13]
    r2 = a[13:17] + [X] + a[17:21]
    cases["doc_right"] = case_from_rows([r1, r2 + [None] * (len(r1) - len(r2))], [5, 3, 8], H, rng)
    cases["doc_left"] = case_from_rows([r1, [None] * (len(r1) - len(r2)) + r2], [5, 3, 8], H, rng)
    # edge case of the docstring (:974-997): equal token counts, different audio sizes -> padding_side decides
    e1 = a[0:4] + [X] + a[4:6]
    e2 = a[6:8] + [X] + a[8:12]
    cases["edge_left"] = case_from_rows([e1, e2], [3, 5], H, rng, side="left")
    cases["edge_right"] = case_from_rows([e1, e2], [3, 5], H, rng, side="right")
    cases["single_nolabels"] = case_from_rows([a[0:3] + [X] + a[3:5] + [X]], [4, 2], H, rng, labels=False)
    # seeded random batches
    for ci, (B, side) in enumerate([(3, "left"), (4, "right"), (2, "left")]):
        rows, widths = [], []
        for b in range(B):
            n_txt = int(rng.integers(3, 12))
            n_aud = int(rng.integers(0, 3))
            row = [int(t) for t in rng.integers(1, 90, n_txt)]
            for _ in range(n_aud):
                row.insert(int(rng.integers(0, len(row) + 1)), X)
                widths.append(int(rng.integers(1, 9)))
            rows.append(row)
        if not widths:
            rows[0].insert(1, X)
            widths.append(3)
        L = max(len(r) for r in rows)
        rows = [([None] * (L - len(r)) + r) if side == "left" else (r + [None] * (L - len(r))) for r in rows]
        cases[f"rand{ci}_{side}"] = case_from_rows(rows, widths, H, rng, side=side)
    out = {}
    for name, c in cases.items():
        stand_in = types.SimpleNamespace(config=types.SimpleNamespace(audio_token_index=AUDIO, ignore_index=IGN), pad_token_id=PAD,
                                         padding_side=c["padding_side"])
        t = lambda x: None if x is None else torch.from_numpy(x)
        emb, mask, lab, pos, ids = fn(stand_in, t(c["audio_features"]), t(c["num_audio_tokens"]), t(c["inputs_embeds"]), t(c["input_ids"]),
                                      t(c["attention_mask"]), t(c["labels"]))
        for k, v in c.items():
            if k == "padding_side":
                out[f"{name}/padding_side"] = np.array(v)
            elif v is not None:
                out[f"{name}/in/{k}"] = v
        out[f"{name}/out/final_embedding"] = emb.numpy()
        out[f"{name}/out/final_attention_mask"] = mask.numpy()
        out[f"{name}/out/position_ids"] = pos.numpy()
        out[f"{name}/out/final_input_ids"] = ids.numpy()
        if lab is not None:
            out[f"{name}/out/final_labels"] = lab.numpy()
        print(name, "ids", c["input_ids"].shape, "->", tuple(emb.shape))
    path = os.path.join(ROOT, "tests", "golden", "golden_merge.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
