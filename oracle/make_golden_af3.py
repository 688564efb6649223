#!/usr/bin/env python3
"""Capture golden vectors for the AF3 / Qwen2-Audio `generate()`-shaped entry from the reference (build container
only; test infrastructure).

`Qwen2AudioForConditionalGeneration` cannot be constructed under transformers 5.x, but its `forward` and
`prepare_inputs_for_generation` are plain functions: they are called UNBOUND on a stand-in `self` that carries the real parts --
the reference's `AFWhisperEncoder` (tiny config), its `Qwen2AudioMultiModalProjector`, its `_merge_input_ids_with_audio_features`
and a transformers `Qwen2ForCausalLM` -- with every parameter overwritten by the build-owned seeded generator.

Stored (tests/golden/golden_af3.json + golden_af3_arrays.npz):
  * a LEFT-padded batch of two prompts with one `<|AUDIO|>` placeholder each (10-s and 30-s clips): merged attention mask,
    prefill logits samples, 6 greedy steps (ids, top-2 gaps, logits samples) driven through `forward` with the cache;
  * the same prompts RIGHT-padded: logits at each row's last valid position;
  * `prepare_inputs_for_generation` known answers for its three input-slicing rules.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_af3.py
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import make_golden as mg  # noqa: E402
import oracle  # noqa: E402
from oracle import fixtures_common as fc  # noqa: E402
from audio_intelligence_amd.utils import synthetic as syn  # noqa: E402

SEED_ENC, SEED_LM, SEED_PROJ = 11, 12, 13
VOCAB, AUDIO_ID, PAD_ID = 3000, 2999, 0
N_STEPS = 6
LOGIT_STEP = 7


def text_cfg():
    return dict(hidden_size=768, num_hidden_layers=4, num_attention_heads=12, num_key_value_heads=2, intermediate_size=3072,
                rope_theta=1e6, rms_norm_eps=1e-6, vocab_size=VOCAB)


def lm_state(tcfg):
    shapes = [s for s in syn.llm_param_shapes({**tcfg, "text_vocab": VOCAB}, VOCAB, 1, 384) if not s[0].startswith(("stream_emb", "adaptor"))]
    return syn.synth_state_dict(shapes, SEED_LM)


def prompts():
    rng = np.random.default_rng(31)
    t0 = rng.integers(1, AUDIO_ID, size=5).tolist()
    t1 = rng.integers(1, AUDIO_ID, size=7).tolist()
    return [t0[:2] + [AUDIO_ID] + t0[2:], [AUDIO_ID] + t1]            # 6 and 8 tokens


def pad_batch(seqs, side):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), PAD_ID, dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for i, s in enumerate(seqs):
        if side == "left":
            ids[i, L - len(s):] = torch.tensor(s)
            mask[i, L - len(s):] = 1
        else:
            ids[i, : len(s)] = torch.tensor(s)
            mask[i, : len(s)] = 1
    return ids, mask


def main():
    torch.set_num_threads(os.cpu_count())
    ref = mg.import_reference()
    from transformers import Qwen2Config, Qwen2ForCausalLM, WhisperFeatureExtractor
    cls = ref.mw.Qwen2AudioForConditionalGeneration
    ecfg, tcfg = oracle.afwhisper.tiny_config(), text_cfg()
    out = {"generator": "oracle/make_golden_af3.py", "text_cfg": tcfg, "enc_cfg": ecfg, "seeds": [SEED_ENC, SEED_LM, SEED_PROJ],
           "audio_token_index": AUDIO_ID, "pad_token_id": PAD_ID, "prompts": prompts(), "clips": [[1000, 160000], [2000, 480000]],
           "n_steps": N_STEPS, "logit_step": LOGIT_STEP}
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        enc, _, _ = mg.build_ref_encoder(ref, ecfg, SEED_ENC, tmp, "sdpa")
    hf = Qwen2Config(vocab_size=VOCAB, hidden_size=tcfg["hidden_size"], num_hidden_layers=tcfg["num_hidden_layers"],
                     num_attention_heads=tcfg["num_attention_heads"], num_key_value_heads=tcfg["num_key_value_heads"],
                     intermediate_size=tcfg["intermediate_size"], rope_theta=tcfg["rope_theta"], rms_norm_eps=tcfg["rms_norm_eps"],
                     tie_word_embeddings=False, max_position_embeddings=4096, attn_implementation="eager")
    lm = Qwen2ForCausalLM(hf).eval()
    res = lm.load_state_dict(lm_state(tcfg), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    pcfg = types.SimpleNamespace(audio_config=types.SimpleNamespace(d_model=ecfg["d_model"]), text_config=types.SimpleNamespace(hidden_size=tcfg["hidden_size"]))
    proj = ref.mw.Qwen2AudioMultiModalProjector(pcfg).eval()
    with torch.no_grad():
        proj.linear.weight.copy_(syn.synth_tensor("multi_modal_projector.linear.weight", proj.linear.weight.shape, SEED_PROJ))
        proj.linear.bias.copy_(syn.synth_tensor("multi_modal_projector.linear.bias", proj.linear.bias.shape, SEED_PROJ))

    class Standin:
        pass
    s = Standin()
    s.audio_tower, s.multi_modal_projector, s.language_model = enc, proj, lm
    s.config = types.SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=True,
                                     audio_token_index=AUDIO_ID, ignore_index=-100)
    s.pad_token_id, s.padding_side = PAD_ID, "left"
    s.get_input_embeddings = lambda: lm.get_input_embeddings()
    s._merge_input_ids_with_audio_features = types.MethodType(cls._merge_input_ids_with_audio_features, s)

    fe = WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, n_fft=400, padding_value=0.0)
    feats, fmask = [], []
    for seed, n in out["clips"]:
        w = fc.make_wav(seed, n)
        w = np.pad(w, (0, 480000 - n)) if n < 480000 else w
        feats.append(torch.from_numpy(fe(w, sampling_rate=16000, return_tensors="np")["input_features"][0]))
        m = torch.zeros(3000, dtype=torch.long)
        m[: n // 160] = 1
        fmask.append(m)
    feats, fmask = torch.stack(feats), torch.stack(fmask)

    # ---- left-padded batch: prefill + greedy steps through forward()
    ids, mask = pad_batch(prompts(), "left")
    with torch.no_grad():
        o = cls.forward(s, input_ids=ids, input_features=feats, attention_mask=mask, feature_attention_mask=fmask, use_cache=True, return_dict=True)
    lg = o.logits.float()
    am = o.attention_mask
    out["left"] = {"input_ids": ids.tolist(), "attention_mask": mask.tolist(), "merged_len": int(lg.shape[1]), "merged_mask_sum": am.sum(-1).tolist(),
                   "merged_mask_first_valid": [int(r.nonzero()[0]) for r in am]}
    arrays["left_prefill_last_logits"] = lg[:, -1, ::LOGIT_STEP].numpy()
    mid = [int(r.nonzero()[0]) + 3 for r in am]                      # an early valid position of each row
    arrays["left_prefill_mid_logits"] = torch.stack([lg[b, mid[b], ::LOGIT_STEP] for b in range(2)]).numpy()
    out["left"]["mid_positions"] = mid
    steps, gaps, samples = [], [], []
    cache, cur_mask, nxt_logits = o.past_key_values, am, lg[:, -1]
    for st in range(N_STEPS):
        tok = nxt_logits.argmax(-1)
        t2 = torch.topk(nxt_logits, 2).values
        steps.append(tok.tolist())
        gaps.append((t2[:, 0] - t2[:, 1]).tolist())
        samples.append(nxt_logits[:, ::LOGIT_STEP].clone())
        cur_mask = torch.cat([cur_mask, cur_mask.new_ones((2, 1))], dim=-1)
        pos = (cur_mask.long().cumsum(-1) - 1)[:, -1:]
        with torch.no_grad():
            o2 = cls.forward(s, input_ids=tok[:, None], attention_mask=cur_mask, position_ids=pos, past_key_values=cache, use_cache=True, return_dict=True)
        cache, nxt_logits = o2.past_key_values, o2.logits[:, -1].float()
    out["left"]["greedy_ids"] = steps
    out["left"]["greedy_gaps"] = gaps
    arrays["left_step_logits"] = torch.stack(samples).numpy()            # [steps, 2, V/7]

    # ---- right-padded batch: prefill only
    s.padding_side = "right"
    ids_r, mask_r = pad_batch(prompts(), "right")
    with torch.no_grad():
        orr = cls.forward(s, input_ids=ids_r, input_features=feats, attention_mask=mask_r, feature_attention_mask=fmask, use_cache=True, return_dict=True)
    lr, amr = orr.logits.float(), orr.attention_mask
    last = [int(r.nonzero()[-1]) for r in amr]
    out["right"] = {"input_ids": ids_r.tolist(), "attention_mask": mask_r.tolist(), "merged_len": int(lr.shape[1]), "last_valid": last,
                    "merged_mask_sum": amr.sum(-1).tolist()}
    arrays["right_last_valid_logits"] = torch.stack([lr[b, last[b], ::LOGIT_STEP] for b in range(2)]).numpy()
    s.padding_side = "left"

    # ---- prepare_inputs_for_generation known answers (legacy tuple cache: past length = keys.shape[2])
    def pig(case):
        past = None if case["past"] is None else ((torch.zeros(2, 1, case["past"], 4), torch.zeros(2, 1, case["past"], 4)),)
        ii = torch.tensor(case["input_ids"])
        kw = {}
        if case.get("kw_mask") is not None:
            kw["attention_mask"] = torch.tensor(case["kw_mask"])
        r = cls.prepare_inputs_for_generation(s, ii, past_key_values=past, input_features=(feats if case["feat"] else None),
                                              attention_mask=(torch.tensor(case["mask"]) if case.get("mask") is not None else None), **kw)
        return {"input_ids": r["input_ids"].tolist(), "position_ids": None if r["position_ids"] is None else r["position_ids"].tolist(),
                "attention_mask": None if r["attention_mask"] is None else r["attention_mask"].tolist()}
    cases = [
        {"name": "first_call", "past": None, "input_ids": [[0, 0, 5, 6], [7, 8, 9, 10]], "mask": [[0, 0, 1, 1], [1, 1, 1, 1]], "feat": True},
        {"name": "rule1_mask_longer_than_ids", "past": 9, "input_ids": [[5, 6, 7, 8, 11], [7, 8, 9, 10, 12]],
         "mask": [[0, 0, 1, 1, 1, 1, 1, 1, 1, 1], [1] * 10], "feat": False},
        {"name": "rule2_past_shorter_than_ids", "past": 4, "input_ids": [[5, 6, 7, 8, 11], [7, 8, 9, 10, 12]], "mask": [[1] * 5, [1] * 5], "feat": False},
        {"name": "rule3_audio_token_present", "past": 9, "input_ids": [[5, AUDIO_ID, 7, 11], [7, 8, AUDIO_ID, 12]], "mask": None, "feat": False},
        # (the `kwargs.get("attention_mask")` branch of :1266-1271 is unreachable: attention_mask is a named parameter)
        {"name": "rule3_past_covers_ids_no_audio", "past": 9, "input_ids": [[5, 6, 7, 11], [7, 8, 9, 12]], "mask": None, "feat": True},
    ]
    for c in cases:
        c["expect"] = pig(c)
    out["prepare_inputs"] = cases

    with open(os.path.join(mg.GOLD, "golden_af3.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(mg.GOLD, "golden_af3_arrays.npz"), **arrays)
    print(json.dumps({"left": {k: v for k, v in out["left"].items() if k != "input_ids"}, "right": out["right"]["last_valid"],
                      "pig": [(c["name"], c["expect"]["input_ids"]) for c in cases]})[:2500])


if __name__ == "__main__":
    main()
