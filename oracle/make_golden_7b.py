#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference at the AF3-7B WIDTHS (BASELINE config 3's shape) with two decoder layers
(build container only; test infrastructure).

The reference's `ParallelHFModel` is built around a locally constructed Qwen2 config with the Qwen2.5-7B widths (hidden 3584,
28 query / 4 kv heads x 128, FFN 18 944, text vocab 152 064 => unified vocab 160 520) and 2 layers; the audio side is the real
`ContinuousAudioIO` over an `AFWhisperEncoder` of the full width (d 1280, 20 heads, FFN 5120) with ONE layer, so the adaptor is
the real 1280 -> 3584 projection.  Every parameter is overwritten with the build-owned seeded generator (regenerated on the GPU
host, nothing travels).  One `audio_to_caption` sample (30-s clip, 32 prompt ids => T = 789 + assistant) goes through the
reference's `collate_fn` -> `_embed` -> `_step` (prefill) -> 8 greedy `_step`s, once in float32 and once in bfloat16.

Stored (tests/golden/golden_7b.json + golden_7b_arrays.npz): fp32 final hidden states of sampled positions, the last position's
stream-0 logits (strided sample + top-16 ids / values), greedy ids with their top-2 gaps; and the reference's own bf16 loss on all
of them (max / mean |bf16 - fp32|, bf16 greedy ids under teacher forcing).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_7b.py
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import make_golden as mg  # noqa: E402
import oracle  # noqa: E402
from oracle import fixtures_common as fc  # noqa: E402
from audio_intelligence_amd.utils import synthetic as syn  # noqa: E402

SEED_LLM_WIDE = 3
SEED_ENC_WIDE = 4
N_DEC = 8
HID_STEP = 37          # sampled positions of the prefill hidden states
LOGIT_STEP = 61        # strided sample of the 160 520 logits


def wide_enc_cfg():
    cfg = dict(oracle.afwhisper.default_config())
    cfg["encoder_layers"] = 1
    return cfg


def wide_llm_cfg():
    cfg = dict(oracle.qwen2.config_7b())
    cfg["num_hidden_layers"] = 2
    return cfg


def main():
    torch.set_num_threads(os.cpu_count())
    ref = mg.import_reference()
    from transformers import Qwen2Config, Qwen2ForCausalLM
    lcfg, ecfg = wide_llm_cfg(), wide_enc_cfg()
    out = {"generator": "oracle/make_golden_7b.py", "llm_cfg": lcfg, "enc_cfg": ecfg, "seed_llm": SEED_LLM_WIDE, "seed_enc": SEED_ENC_WIDE,
           "n_dec": N_DEC, "hid_step": HID_STEP, "logit_step": LOGIT_STEP}
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        enc, enc_sd, enc_dir = mg.build_ref_encoder(ref, ecfg, SEED_ENC_WIDE, tmp, "sdpa")
        del enc
        hf = Qwen2Config(vocab_size=lcfg["text_vocab"], hidden_size=lcfg["hidden_size"], num_hidden_layers=2,
                         num_attention_heads=lcfg["num_attention_heads"], num_key_value_heads=lcfg["num_key_value_heads"],
                         intermediate_size=lcfg["intermediate_size"], rope_theta=lcfg["rope_theta"], rms_norm_eps=lcfg["rms_norm_eps"],
                         tie_word_embeddings=False, max_position_embeddings=4096)
        ldir = os.path.join(tmp, "llm_wide")
        Qwen2ForCausalLM(hf).save_pretrained(ldir)
        text_io, audio_io = mg.make_stub_ios(ref, lcfg["text_vocab"])
        cont_io = ref.audio.ContinuousAudioIO(encoder_choice="AFWhisper", encoder_local_path=enc_dir, dtype="float32", device="cpu")
        ios = {"text": text_io, "discrete_audio": audio_io, "continuous_audio": cont_io}
        job = ref.job.UALMJobTemplate.__new__(ref.job.UALMJobTemplate)
        job.multimodal_io = ios
        vocab, intervals = job._build_vocabulary()
        V = len(vocab)
        assert V == 256 + lcfg["text_vocab"] + 8 * 1025 == 160520
        model = ref.parallel.ParallelHFModel(ldir, multimodal_io=ios, vocab=vocab, vocab_intervals=intervals,
                                             attn_implementation="eager", dtype=torch.float32, local_files_only=True)
        full = {}
        for name, shape in syn.llm_param_shapes(lcfg, V, 8, ecfg["d_model"]):
            full[name] = syn.synth_tensor(name, shape, SEED_LLM_WIDE)
        for k, v in enc_sd.items():
            full["multimodal_io_dict.continuous_audio.model." + k] = v
        res = model.load_state_dict(full, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        del full
        model.prepare_inference()
        model.eval()
        pre = ref.job.UALMPreprocessor(False, {k: v.copy_for_worker() for k, v in ios.items()}, vocab, intervals)
        prompt = fc.make_prompt(lcfg["text_vocab"], n=32, seed=9)
        data = {"audio": (fc.make_wav(2005, 480000)[None], 16000), "text": [["user", "text", prompt]]}
        b = pre.collate_fn([(("audio_to_caption", "x", "y"), data)])
        kw = {k: v for k, v in b.items() if k not in ("keys", "loss_masks")}
        T = int(b["seqs"].shape[1]) + 1
        out.update({"prompt": prompt, "wav_seed": 2005, "seq_len": T, "indices": b["continuous_audio_indices"].tolist(), "vocab_size": V})
        allowed = ~model.text_mask[0, 0, 0]
        hid_rows = fc.sample_row_index(T, HID_STEP)

        def run(m, k, teacher=None):
            cap = {}
            hook = m.model.register_forward_hook(lambda mod, a, o: cap.__setitem__("h", o.last_hidden_state.detach().float().clone()))
            with torch.no_grad():
                emb = m._embed(torch.cat([k["seqs"], m.assistant_token], 1), k)
                lg, cache = m._step(input_embeds=emb, mask=None)
                r = {"emb": emb.float(), "hid": cap["h"][0], "last": lg[0, -1, 0].float().clone()}
                del lg
                tok = m.text_token.clone()
                steps = []
                for st in range(N_DEC):
                    lg, cache = m._step(input_ids=tok, past_key_values=cache, mask=m.text_mask)
                    row = lg[0, 0, 0].float().clone()
                    steps.append(row)
                    nxt = int(row.argmax()) if teacher is None else teacher[st]
                    tok = torch.zeros(1, 1, 8, dtype=torch.long)
                    tok[0, 0, 0] = nxt
                r["steps"] = steps
            hook.remove()
            return r

        r32 = run(model, kw)
        ids32 = [int(s.argmax()) for s in r32["steps"]]
        gaps32 = [float(torch.topk(s, 2).values[0] - torch.topk(s, 2).values[1]) for s in r32["steps"]]
        arrays["hid_rows_f32"] = r32["hid"][hid_rows].numpy()
        arrays["emb_rows_f32"] = r32["emb"][0][hid_rows].numpy()
        arrays["last_logits_sample_f32"] = r32["last"][::LOGIT_STEP].numpy()
        tk = torch.topk(r32["last"], 16)
        arrays["last_logits_top16_val"] = tk.values.numpy()
        arrays["last_logits_top16_idx"] = tk.indices.numpy().astype(np.int64)
        arrays["step_logits_sample_f32"] = torch.stack([s[::LOGIT_STEP] for s in r32["steps"]]).numpy()
        out["f32"] = {"greedy_ids": ids32, "greedy_gaps": gaps32, "hid_abs_mean": float(r32["hid"].abs().mean()),
                      "last_logit_abs_max": float(r32["last"].abs().max())}

        m16 = model.to(torch.bfloat16)
        kw16 = {k: (v.to(torch.bfloat16) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in kw.items()}
        r16 = run(m16, kw16, teacher=ids32)

        def es(a, b_):
            d = (a.float() - b_.float()).abs()
            return {"max": float(d.max()), "mean": float(d.mean())}
        step_err = [es(a[allowed], b_[allowed]) for a, b_ in zip(r16["steps"], r32["steps"])]
        out["bf16"] = {"emb_err": es(r16["emb"], r32["emb"]), "hid_err": es(r16["hid"], r32["hid"]), "last_logit_err": es(r16["last"], r32["last"]),
                       "step_logit_err_max": max(e["max"] for e in step_err), "step_logit_err_mean": float(np.mean([e["mean"] for e in step_err])),
                       "teacher_forced_ids": [int(s.argmax()) for s in r16["steps"]],
                       "regret_in_f32_logits": [float(s32.max() - s32[int(s16.argmax())]) for s16, s32 in zip(r16["steps"], r32["steps"])]}
    with open(os.path.join(mg.GOLD, "golden_7b.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(mg.GOLD, "golden_7b_arrays.npz"), **arrays)
    print(json.dumps({k: out[k] for k in ("f32", "bf16", "seq_len")}, indent=1))


if __name__ == "__main__":
    main()
