#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (build container only; test infrastructure).

Imports /root/reference/UALM (read-only) with the shims of SURVEY Appendix C, overwrites every
reference parameter with the build-owned seeded generator (audio_intelligence_amd.utils.synthetic),
runs the reference's own classes on seeded synthetic inputs and writes small fixtures (data only:
inputs are regenerated from seeds, outputs are sampled values / ids / checksums) to tests/golden/.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--skip-full]

The reference never travels to the GPU box; these fixtures (and this script) are what pins the
CPU restatement in oracle/.
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time
import types

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
os.environ.setdefault("HF_TOKEN", "offline")
sys.dont_write_bytecode = True

import numpy as np
import torch
import transformers  # noqa: F401  (must be imported before the reference package)
from transformers import WhisperFeatureExtractor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from audio_intelligence_amd.utils import synthetic as syn  # noqa: E402
import oracle  # noqa: E402
from oracle import fixtures_common as fc  # noqa: E402


def import_reference():
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    sys.path.insert(0, "/root/reference/UALM")
    import models  # noqa: F401
    from models.ualm.multimodal_io import audio, modeling_whisper, sound_encoder, afwhisper_audio_encoder, abs_io
    from models.ualm.lm import parallel
    from models.ualm import ualm_job
    return types.SimpleNamespace(audio=audio, mw=modeling_whisper, se=sound_encoder,
                                 afse=afwhisper_audio_encoder, abs_io=abs_io, parallel=parallel, job=ualm_job)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_ref_encoder(ref, cfg, seed, tmp, attn="sdpa"):
    from transformers.models.qwen2_audio.configuration_qwen2_audio import Qwen2AudioEncoderConfig
    hcfg = Qwen2AudioEncoderConfig(num_mel_bins=cfg["num_mel_bins"], d_model=cfg["d_model"],
                                   encoder_attention_heads=cfg["encoder_attention_heads"],
                                   encoder_ffn_dim=cfg["encoder_ffn_dim"], encoder_layers=cfg["encoder_layers"],
                                   max_source_positions=cfg["max_source_positions"], pad_token_id=0,
                                   init_std=0.02, attn_implementation=attn)
    enc = ref.mw.AFWhisperEncoder(hcfg).eval()
    sd = syn.synth_state_dict(syn.encoder_param_shapes(cfg), seed)
    missing = enc.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    d = os.path.join(tmp, f"enc_{cfg['d_model']}_{attn}")
    enc.save_pretrained(d)
    return enc, sd, d


def g1_mel(ref, out):
    fe = WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, n_fft=400, padding_value=0.0)
    filt = fe.mel_filters
    out["filters"] = {"shape": list(filt.shape), "sha256_f64": sha(filt.astype(np.float64)),
                      "sample_idx": fc.FILTER_SAMPLE_IDX, "sample_val": [float(filt.reshape(-1)[i]) for i in fc.FILTER_SAMPLE_IDX],
                      "nnz": int((filt != 0).sum())}
    clips = {}
    t_all = []
    for name, seed, n in fc.mel_cases():
        wav = fc.make_wav(seed, n)
        w = wav
        if w.shape[0] < 480000:
            w = np.pad(w, (0, 480000 - w.shape[0]))
        t0 = time.perf_counter()
        feat = fe(w, sampling_rate=16000, return_tensors="np")["input_features"][0]
        t_all.append(time.perf_counter() - t0)
        assert feat.shape == (128, 3000) and feat.dtype == np.float32
        flat = feat.reshape(-1)
        clips[name] = {"seed": seed, "n": n, "sha256": sha(feat), "max": float(feat.max()), "mean": float(feat.astype(np.float64).mean()),
                       "sample_val": [float(flat[i]) for i in fc.MEL_SAMPLE_IDX]}
    out["mel_sample_idx"] = fc.MEL_SAMPLE_IDX
    out["clips"] = clips
    out["ref_seconds_per_clip_8core"] = float(np.median(t_all))


def g2_lengths(ref, io, out):
    rows = []
    for n in fc.LENGTH_CASES:
        wav = np.zeros(n, dtype=np.float32)
        fl = io.find_length((wav, 16000))
        _, (after, feat), _ = io.preprocess((wav, 16000))
        rows.append({"n": n, "find_length": int(fl), "after_length": int(after), "feat_shape": list(feat.shape)})
    out["length_table"] = rows
    out["find_length_8k"] = int(io.find_length((np.zeros(80000, dtype=np.float32), 8000)))


def g3_tiny_encoder(ref, out, tmp, arrays):
    cfg = oracle.afwhisper.tiny_config()
    enc, sd, d = build_ref_encoder(ref, cfg, fc.SEED_ENC_TINY, tmp, "sdpa")
    enc_e, _, _ = build_ref_encoder(ref, cfg, fc.SEED_ENC_TINY, tmp, "eager")
    io = ref.audio.ContinuousAudioIO(encoder_choice="AFWhisper", encoder_local_path=d, dtype="float32", device="cpu")
    res = {}
    for name, seed, n in [("s30", 2000, 480000), ("s10", 1000, 160000)]:
        wav = fc.make_wav(seed, n)
        _, (after, feat), _ = io.preprocess((wav, 16000))
        ft = torch.from_numpy(feat)[None]
        # stage states (unmasked)
        with torch.no_grad():
            o = enc(ft.transpose(1, 2), output_hidden_states=True)
        hs = o.hidden_states  # (stem, l0, ..., final)
        ent = {"after": int(after)}
        # hidden_states = (stem, input of layer 1 == output of layer 0, final) (modeling_whisper.py:709-750)
        for k, h in zip(["stem", "layer0", "final"], [hs[0], hs[1], o.last_hidden_state]):
            arrays[f"enc_tiny_{name}_{k}"] = fc.sample_rows(h[0].numpy())
            ent[k + "_sha256"] = sha(h[0].numpy())
        # both length conventions through encode_batch
        with torch.no_grad():
            pipe = io.encode_batch(ft, torch.tensor([3000]))[0]
            selft = io.encode_batch(ft, torch.tensor([after]))[0]
            eager = enc_e(ft.transpose(1, 2)).last_hidden_state[0]
        ent["pipeline_rows"] = int(pipe.shape[0])
        ent["selftest_rows"] = int(selft.shape[0])
        ent["pipeline_equals_unmasked_maxdiff"] = float((pipe - o.last_hidden_state[0]).abs().max())
        ent["eager_vs_sdpa_maxdiff"] = float((eager - o.last_hidden_state[0]).abs().max())
        arrays[f"enc_tiny_{name}_selftest"] = fc.sample_rows(selft.numpy())
        res[name] = ent
    # ragged batch, self-test convention (10/20/30 s)
    feats, lens = [], []
    for seed, n in [(1001, 160000), (1501, 320000), (2001, 480000)]:
        _, (after, feat), _ = io.preprocess((fc.make_wav(seed, n), 16000))
        feats.append(torch.from_numpy(feat))
        lens.append(after)
    with torch.no_grad():
        outs = io.encode_batch(torch.stack(feats), torch.tensor(lens))
    res["ragged"] = {"lens": [int(x) for x in lens], "rows": [int(o.shape[0]) for o in outs]}
    for i, o in enumerate(outs):
        arrays[f"enc_tiny_ragged_{i}"] = fc.sample_rows(o.numpy())
    # SoundTower window stack: 3 full windows + one 12-s window
    st = ref.afse.AFWhisperSoundTower(d, None)
    st.sound_tower = enc
    st.is_loaded = True
    sounds = torch.stack(feats + [feats[0]])[:, None].transpose(2, 3)[None]   # [1,4,1,128,3000]
    mask = torch.ones(1, 4, 1, 3000, dtype=torch.long)
    mask[0, 3, 0, 1200:] = 0
    with torch.no_grad():
        y = st(sounds, mask)
    res["sound_tower"] = {"shape": list(y.shape), "mask_sums": [3000, 3000, 3000, 1200]}
    for i in range(4):
        arrays[f"enc_tiny_tower_{i}"] = fc.sample_rows(y[i].numpy())
    out["tiny_encoder"] = res
    return enc, sd, d, io


def g4_full_encoder(ref, out, tmp, arrays):
    cfg = oracle.afwhisper.default_config()
    enc, sd, d = build_ref_encoder(ref, cfg, fc.SEED_ENC_FULL, tmp, "sdpa")
    fe = WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, n_fft=400, padding_value=0.0)
    wav = fc.make_wav(2000, 480000)
    feat = torch.from_numpy(fe(wav, sampling_rate=16000, return_tensors="np")["input_features"])
    with torch.no_grad():
        enc(feat)
        t0 = time.perf_counter()
        y = enc(feat).last_hidden_state[0]
        dt = time.perf_counter() - t0
    arrays["enc_full_s30_final"] = fc.sample_rows(y.numpy())
    out["full_encoder"] = {"sha256": sha(y.numpy()), "ref_seconds_per_clip_8core": dt,
                           "params": int(sum(p.numel() for p in enc.parameters()))}
    del enc, sd


def make_stub_ios(ref, text_vocab):
    AbsIO = ref.abs_io.AbsIO

    class StubText(AbsIO):
        def __init__(self):
            super().__init__(modality="text", is_discrete=True)
            self.vocab_size = text_vocab

        def preprocess(self, data):
            t = np.array(data, dtype=np.int32).reshape(-1, 1)
            return t, None, (t * 0 + 1).astype(np.float32)

        def find_length(self, data):
            return len(data)

        def copy_for_worker(self):
            return self

        def num_stream(self):
            return 1

        def get_vocabulary(self):
            return [f"<text_{i}>" for i in range(text_vocab)]

        def get_stream_interval(self):
            return [(0, text_vocab)]

        def decode_batch(self, tokens, lengths):
            return [t[:l, 0].tolist() for t, l in zip(tokens, lengths)]

    class StubAudio(AbsIO):
        def __init__(self):
            super().__init__(modality="audio", is_discrete=True)

        def copy_for_worker(self):
            return self

        def num_stream(self):
            return 8

        def get_vocabulary(self):
            return [f"<audio_{i}>" for i in range(8 * 1025)]

        def get_stream_interval(self):
            return [(s * 1025, (s + 1) * 1025) for s in range(8)]

    return StubText(), StubAudio()


def g5_g6_llm(ref, out, tmp, arrays, enc_dir):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    lcfg = oracle.qwen2.config_tiny()
    hf = Qwen2Config(vocab_size=lcfg["text_vocab"], hidden_size=lcfg["hidden_size"], num_hidden_layers=lcfg["num_hidden_layers"],
                     num_attention_heads=lcfg["num_attention_heads"], num_key_value_heads=lcfg["num_key_value_heads"],
                     intermediate_size=lcfg["intermediate_size"], rope_theta=lcfg["rope_theta"], rms_norm_eps=lcfg["rms_norm_eps"],
                     tie_word_embeddings=False, max_position_embeddings=4096)
    ldir = os.path.join(tmp, "llm_tiny")
    Qwen2ForCausalLM(hf).save_pretrained(ldir)
    text_io, audio_io = make_stub_ios(ref, lcfg["text_vocab"])
    cont_io = ref.audio.ContinuousAudioIO(encoder_choice="AFWhisper", encoder_local_path=enc_dir, dtype="float32", device="cpu")
    ios = {"text": text_io, "discrete_audio": audio_io, "continuous_audio": cont_io}
    # vocabulary exactly as UALMJobTemplate._build_vocabulary (ualm_job.py:71-110)
    job = ref.job.UALMJobTemplate.__new__(ref.job.UALMJobTemplate)
    job.multimodal_io = ios
    vocab, intervals = job._build_vocabulary()
    my_vocab, my_iv = oracle.ualm.build_vocabulary(lcfg["text_vocab"])
    assert len(vocab) == len(my_vocab) and {k: [tuple(x) for x in v] for k, v in intervals.items()} == my_iv
    model = ref.parallel.ParallelHFModel(ldir, multimodal_io=ios, vocab=vocab, vocab_intervals=intervals,
                                         attn_implementation="eager", dtype=torch.float32, local_files_only=True)
    enc_cfg = oracle.afwhisper.tiny_config()
    V = len(vocab)
    sd = syn.synth_state_dict(syn.llm_param_shapes(lcfg, V, 8, enc_cfg["d_model"]), fc.SEED_LLM_TINY)
    enc_sd = syn.synth_state_dict(syn.encoder_param_shapes(enc_cfg), fc.SEED_ENC_TINY)
    full = dict(sd)
    for k, v in enc_sd.items():
        full["multimodal_io_dict.continuous_audio.model." + k] = v
    res = model.load_state_dict(full, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.prepare_inference()
    model.eval()
    out["llm_tiny"] = {"vocab_size": V, "intervals": {k: [list(x) for x in v] for k, v in intervals.items()},
                       "state_dict_keys_sha256": hashlib.sha256("\n".join(sorted(full.keys())).encode()).hexdigest(),
                       "n_state_dict_keys": len(full),
                       "modality_mask_rowsum": model.modality_mask[0, 0].sum(-1).tolist(),
                       "text_mask_rowsum": model.text_mask[0, 0].sum(-1).tolist(),
                       "audio_mask_rowsum": model.audio_mask[0, 0].sum(-1).tolist()}
    pre = ref.job.UALMPreprocessor(False, {k: v.copy_for_worker() for k, v in ios.items()}, vocab, intervals)
    prompt = fc.make_prompt(lcfg["text_vocab"])
    # G5: collate dict with the survey's small sample (text ids [0,5,6,7], 10-s clip)
    data = {"audio": (fc.make_wav(1000, 160000)[None], 16000), "text": [["user", "text", [0, 5, 6, 7]]]}
    b = pre.collate_fn([(("audio_to_caption", "x", "y"), data)])
    out["collate_small"] = {"seqs_stream0": b["seqs"][0, :, 0].tolist(), "seqs_shape": list(b["seqs"].shape),
                            "other_streams_sum": int(b["seqs"][0, :, 1:].sum()),
                            "indices": b["continuous_audio_indices"].tolist(), "lengths": b["continuous_audio_lengths"].tolist(),
                            "feats_shape": list(b["continuous_audio_feats"].shape),
                            "loss_mask_sum": float(b["loss_masks"].sum()),
                            "find_length": int(pre.find_length(("audio_to_caption", "x", "y"), data))}
    # G6: greedy decode, 10 clips x 32 steps
    gaps_all, toks_all = [], []
    cfg = {"text": {"temperature": 0.0, "topk": 20, "cfg": 1, "max_step": fc.MAX_STEP}, "num_hypo": 1}
    orig = model._logits_to_token
    t_ref = []
    for i in range(10):
        data = {"audio": (fc.make_wav(1000 + i, 160000)[None], 16000), "text": [["user", "text", prompt]]}
        b = pre.collate_fn([(("audio_to_caption", "x", "y"), data)])
        gaps = []

        def rec(logits, temperature, topk, _g=gaps):
            t2 = torch.topk(logits[0, 0, 0], 2).values
            _g.append(float(t2[0] - t2[1]))
            return orig(logits, temperature=temperature, topk=topk)
        model._logits_to_token = rec
        kw = {k: v for k, v in b.items() if k not in ("keys", "loss_masks")}
        t0 = time.perf_counter()
        hyps, _ = model.inference_segment(cfg, cache=None, enforce_modality="text", **kw)
        t_ref.append(time.perf_counter() - t0)
        toks, modality = hyps[0]
        assert modality == "text"
        toks_all.append(toks[:, 0].tolist())
        assert int(toks[:, 1:].abs().sum()) == 0
        gaps_all.append(gaps)
        if i == 0:
            out["llm_tiny"]["seq_len_clip0"] = int(b["seqs"].shape[1])
            # prefill logits sample (last position, stream 0) for a tolerance check
            with torch.no_grad():
                emb = model._embed(torch.cat([b["seqs"], model.assistant_token], 1), kw)
                lg, _ = model._step(input_embeds=emb, mask=None)
            arrays["llm_tiny_prefill_logits_last_s0"] = lg[0, -1, 0, ::37].numpy().astype(np.float32)
            arrays["llm_tiny_embed_rows"] = fc.sample_rows(emb[0].numpy())
    out["llm_tiny"]["greedy_tokens"] = toks_all
    out["llm_tiny"]["greedy_gaps"] = gaps_all
    out["llm_tiny"]["ref_seconds_per_clip_8core"] = float(np.median(t_ref))
    out["llm_tiny"]["prompt"] = prompt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    ref = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    meta = {"generator": "oracle/make_golden.py", "torch": torch.__version__, "transformers": transformers.__version__,
            "cpu_count": os.cpu_count(), "reference": "NVIDIA/audio-intelligence @ /root/reference (2026-03-27)"}
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        g1_mel(ref, meta)
        enc, sd, enc_dir, io = g3_tiny_encoder(ref, meta, tmp, arrays)
        g2_lengths(ref, io, meta)
        g5_g6_llm(ref, meta, tmp, arrays, enc_dir)
        if not args.skip_full:
            g4_full_encoder(ref, meta, tmp, arrays)
    with open(os.path.join(GOLD, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "golden_arrays.npz"), **arrays)
    print("wrote", GOLD, {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
