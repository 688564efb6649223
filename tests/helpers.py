"""Shared test helpers: golden fixtures, seeded inputs/weights (test infrastructure)."""
import json
import os

import numpy as np
import torch

import oracle
from oracle import fixtures_common as fc
from audio_intelligence_amd.utils import synthetic as syn

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def golden():
    if "g" not in _cache:
        with open(os.path.join(GOLD_DIR, "golden.json")) as f:
            _cache["g"] = json.load(f)
        _cache["a"] = dict(np.load(os.path.join(GOLD_DIR, "golden_arrays.npz")))
    return _cache["g"], _cache["a"]


def tiny_enc():
    if "tiny_enc" not in _cache:
        cfg = oracle.afwhisper.tiny_config()
        _cache["tiny_enc"] = (cfg, syn.synth_state_dict(syn.encoder_param_shapes(cfg), fc.SEED_ENC_TINY))
    return _cache["tiny_enc"]


def tiny_llm():
    if "tiny_llm" not in _cache:
        cfg = oracle.qwen2.config_tiny()
        vocab, iv = oracle.ualm.build_vocabulary(cfg["text_vocab"])
        sd = syn.synth_state_dict(syn.llm_param_shapes(cfg, len(vocab), 8, 384), fc.SEED_LLM_TINY)
        _cache["tiny_llm"] = (cfg, sd, vocab, iv)
    return _cache["tiny_llm"]


def mel_of(seed, n):
    key = ("mel", seed, n)
    if key not in _cache:
        _cache[key] = oracle.logmel.log_mel(fc.make_wav(seed, n))
    return _cache[key]


def caption_batch(seed, n=160000, prompt=None):
    cfg, sd, vocab, iv = tiny_llm()
    if prompt is None:
        prompt = fc.make_prompt(cfg["text_vocab"])
    s = oracle.ualm.preprocessing("audio_to_caption", {"text1": np.array(prompt)}, fc.make_wav(seed, n), iv)
    return oracle.ualm.collate([s])
