"""Pin the CPU oracle against golden vectors captured from the real reference (oracle/make_golden.py).

CPU-only (`-m "not gpu"`).  Tolerances: the oracle is fp32 PyTorch-CPU like the reference, so
activations agree to a few ulp-scale 1e-5; ids / lengths / index tables are exact."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import fixtures_common as fc
from tests import helpers as H


def test_mel_filter_bank():
    g, _ = H.golden()
    filt = oracle.logmel.mel_filter_bank()
    assert list(filt.shape) == g["filters"]["shape"]
    assert int((filt != 0).sum()) == g["filters"]["nnz"]
    got = filt.reshape(-1)[g["filters"]["sample_idx"]]
    np.testing.assert_allclose(got, g["filters"]["sample_val"], rtol=1e-12, atol=1e-15)


def test_log_mel_against_reference():
    g, _ = H.golden()
    idx = g["mel_sample_idx"]
    for name, c in g["clips"].items():
        wav = fc.make_wav(c["seed"], c["n"])
        for prec, tol in (("f32", 2e-6), ("f64", 1e-4)):
            m = oracle.logmel.log_mel(wav, precision=prec)
            assert m.shape == (128, 3000) and m.dtype == np.float32
            np.testing.assert_allclose(m.reshape(-1)[idx], c["sample_val"], rtol=0, atol=tol, err_msg=f"{name} {prec}")
            assert abs(float(m.max()) - c["max"]) <= tol
            assert abs(float(m.astype(np.float64).mean()) - c["mean"]) <= tol


def test_length_table():
    g, _ = H.golden()
    for row in g["length_table"]:
        assert oracle.lengths.find_length(row["n"]) == row["find_length"]
        assert oracle.lengths.after_length(min(row["n"], 480000)) == row["after_length"]
    assert oracle.lengths.find_length(80000, sr=8000) == g["find_length_8k"]
    # the two callers' conventions (SURVEY headline fact 5)
    assert oracle.lengths.encode_batch_lengths(3000) == (6000, 3000)
    assert oracle.lengths.encode_batch_lengths(250) == (500, 250)
    # reference self-test table (audio.py:1127-1128,1268-1271)
    for sec, exp in ((5, 125), (10, 250), (20, 500), (30, 750)):
        assert oracle.lengths.after_length(sec * 16000) == exp


def _rows(a):
    return a[fc.sample_row_index(a.shape[0])]


def test_tiny_encoder_stages_and_conventions():
    g, arr = H.golden()
    cfg, sd = H.tiny_enc()
    for name, seed, n in (("s30", 2000, 480000), ("s10", 1000, 160000)):
        mel = torch.from_numpy(H.mel_of(seed, n))[None]
        out, states = oracle.afwhisper.encoder_forward(mel, sd, cfg, return_states=True)
        np.testing.assert_allclose(_rows(states[0][0].numpy()), arr[f"enc_tiny_{name}_stem"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(_rows(states[1][0].numpy()), arr[f"enc_tiny_{name}_layer0"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(_rows(out[0].numpy()), arr[f"enc_tiny_{name}_final"], atol=5e-5, rtol=0)
        after = g["tiny_encoder"][name]["after"]
        feats = mel.transpose(1, 2)
        pipe = oracle.afwhisper.encode_batch(feats, torch.tensor([3000]), sd, cfg)[0]
        assert pipe.shape[0] == g["tiny_encoder"][name]["pipeline_rows"] == 750
        assert float((pipe - out[0]).abs().max()) == 0.0          # pipeline convention == unmasked
        st = oracle.afwhisper.encode_batch(feats, torch.tensor([after]), sd, cfg)[0]
        assert st.shape[0] == g["tiny_encoder"][name]["selftest_rows"]
        np.testing.assert_allclose(_rows(st.numpy()), arr[f"enc_tiny_{name}_selftest"], atol=5e-5, rtol=0)
        eager = oracle.afwhisper.encoder_forward(mel, sd, cfg, sdpa=False)
        assert float((eager - out).abs().max()) < 2e-5


def test_tiny_encoder_ragged_and_tower():
    g, arr = H.golden()
    cfg, sd = H.tiny_enc()
    feats, lens = [], []
    for seed, n in ((1001, 160000), (1501, 320000), (2001, 480000)):
        feats.append(torch.from_numpy(H.mel_of(seed, n)).T)
        lens.append(oracle.lengths.after_length(n))
    assert lens == g["tiny_encoder"]["ragged"]["lens"]
    outs = oracle.afwhisper.encode_batch(torch.stack(feats), torch.tensor(lens), sd, cfg)
    assert [o.shape[0] for o in outs] == g["tiny_encoder"]["ragged"]["rows"]
    for i, o in enumerate(outs):
        np.testing.assert_allclose(_rows(o.numpy()), arr[f"enc_tiny_ragged_{i}"], atol=5e-5, rtol=0)
    sounds = torch.stack(feats + [feats[0]])[:, None].transpose(2, 3)[None]
    mask = torch.ones(1, 4, 1, 3000, dtype=torch.long)
    mask[0, 3, 0, 1200:] = 0
    y = oracle.afwhisper.sound_tower(sounds, mask, sd, cfg)
    assert list(y.shape) == g["tiny_encoder"]["sound_tower"]["shape"]
    for i in range(4):
        np.testing.assert_allclose(_rows(y[i].numpy()), arr[f"enc_tiny_tower_{i}"], atol=5e-5, rtol=0)


def test_vocab_masks_and_collate():
    g, _ = H.golden()
    cfg, sd, vocab, iv = H.tiny_llm()
    L = g["llm_tiny"]
    assert len(vocab) == L["vocab_size"]
    assert {k: [list(x) for x in v] for k, v in iv.items()} == L["intervals"]
    mk = oracle.ualm.masks(len(vocab), iv)
    assert mk["modality"].sum(-1).tolist() == L["modality_mask_rowsum"]
    assert mk["text"].sum(-1).tolist() == L["text_mask_rowsum"]
    assert mk["audio"].sum(-1).tolist() == L["audio_mask_rowsum"]
    b = H.caption_batch(1000, prompt=[0, 5, 6, 7])
    c = g["collate_small"]
    assert list(b["seqs"].shape) == c["seqs_shape"]
    assert b["seqs"][0, :, 0].tolist() == c["seqs_stream0"]
    assert int(b["seqs"][0, :, 1:].sum()) == c["other_streams_sum"]
    assert b["continuous_audio_indices"].tolist() == c["indices"]
    assert b["continuous_audio_lengths"].tolist() == c["lengths"]
    assert list(b["continuous_audio_feats"].shape) == c["feats_shape"]
    assert float(b["loss_masks"].sum()) == c["loss_mask_sum"]
    # state-dict key set of the drop-in (names the reference checkpoint carries)
    import hashlib
    from audio_intelligence_amd.utils import synthetic as syn
    keys = [n for n, _ in syn.llm_param_shapes(cfg, len(vocab), 8, 384)]
    keys += ["multimodal_io_dict.continuous_audio.model." + n for n, _ in syn.encoder_param_shapes(oracle.afwhisper.tiny_config())]
    assert len(keys) == L["n_state_dict_keys"]
    assert hashlib.sha256("\n".join(sorted(keys)).encode()).hexdigest() == L["state_dict_keys_sha256"]


def test_llm_prefill_logits_and_greedy_tokens():
    g, arr = H.golden()
    cfg, sd, vocab, iv = H.tiny_llm()
    ecfg, esd = H.tiny_enc()
    L = g["llm_tiny"]
    b = H.caption_batch(1000)
    assert b["seqs"].shape[1] == L["seq_len_clip0"]
    a = torch.zeros(1, 1, 8, dtype=torch.long)
    a[0, 0, 0] = oracle.ualm.special_id("<|assistant|>")
    emb = oracle.ualm.embed(torch.cat([b["seqs"], a], 1), b, sd, esd, ecfg)
    np.testing.assert_allclose(_rows(emb[0].numpy()), arr["llm_tiny_embed_rows"], atol=5e-5, rtol=0)
    lg, _ = oracle.ualm.step(sd, cfg, input_embeds=emb)
    np.testing.assert_allclose(lg[0, -1, 0, ::37].numpy(), arr["llm_tiny_prefill_logits_last_s0"], atol=2e-4, rtol=0)
    for i in range(10):
        bi = H.caption_batch(1000 + i)
        toks, modality, gaps = oracle.ualm.inference_segment(bi, sd, cfg, esd, ecfg, iv, max_step=fc.MAX_STEP, return_margins=True)
        assert modality == "text"
        assert toks[:, 0].tolist() == L["greedy_tokens"][i], f"clip {i}"
        assert int(toks[:, 1:].abs().sum()) == 0
        np.testing.assert_allclose(gaps, L["greedy_gaps"][i][: len(gaps)], atol=2e-4, rtol=0)


def test_full_shape_encoder():
    g, arr = H.golden()
    cfg = oracle.afwhisper.default_config()
    from audio_intelligence_amd.utils import synthetic as syn
    sd = syn.synth_state_dict(syn.encoder_param_shapes(cfg), fc.SEED_ENC_FULL)
    assert sum(v.numel() for v in sd.values()) == g["full_encoder"]["params"]
    mel = torch.from_numpy(H.mel_of(2000, 480000))[None]
    out = oracle.afwhisper.encoder_forward(mel, sd, cfg)
    np.testing.assert_allclose(_rows(out[0].numpy()), arr["enc_full_s30_final"], atol=2e-4, rtol=0)


# ---- AF3 / Qwen2-Audio placeholder merge (modeling_whisper.py:913-1108) ----
def _merge_cases():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_merge.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    for n in names:
        ins = {k.split("/")[2]: z[k] for k in z.files if k.startswith(n + "/in/")}
        outs = {k.split("/")[2]: z[k] for k in z.files if k.startswith(n + "/out/")}
        yield n, ins, outs, str(z[n + "/padding_side"])


def test_merge_oracle_matches_reference_golden():
    from oracle import merge as om
    n_cases = 0
    for name, ins, outs, side in _merge_cases():
        emb, mask, lab, pos, ids = om.merge_input_ids_with_audio_features(
            ins["audio_features"], ins["num_audio_tokens"], ins["inputs_embeds"], ins["input_ids"], ins["attention_mask"],
            ins.get("labels"), audio_token_index=99, pad_token_id=-1, ignore_index=-100, padding_side=side)
        assert np.array_equal(emb, outs["final_embedding"]), name          # pure copies: bit-exact
        assert np.array_equal(mask, outs["final_attention_mask"]), name
        assert np.array_equal(pos, outs["position_ids"]), name
        assert np.array_equal(ids, outs["final_input_ids"]), name
        if "final_labels" in outs:
            assert np.array_equal(lab, outs["final_labels"]), name
        else:
            assert lab is None
        n_cases += 1
    assert n_cases == 8


def test_merge_oracle_rejects_what_the_reference_rejects():
    from oracle import merge as om
    ids = np.array([[1, 99, 2], [3, 4, 5]])
    feats = np.zeros((1, 4, 2), np.float32)
    emb = np.zeros((2, 3, 2), np.float32)
    with pytest.raises(ValueError):      # zeros on both edges of the mask (modeling_whisper.py:1017-1019)
        om.merge_plan([4], ids, np.array([[0, 1, 1], [1, 1, 0]]), 99)
    with pytest.raises(ValueError):      # audio rows offered != placeholder slots (modeling_whisper.py:1098-1102)
        om.merge_input_ids_with_audio_features(np.zeros((2, 4, 2), np.float32), [4, 2], emb, ids, np.ones((2, 3), np.int64), None, 99)
