"""GPU parity for SURVEY 8(a) row 14: the AF3 / Qwen2-Audio placeholder merge and projector
(modeling_whisper.py:768-775, 913-1108) through the C ABI (afhip_gather_rows, afhip_gemm).

Expected values come from tests/golden/golden_merge.npz, captured from the reference function itself
(oracle/make_golden_merge.py); the copies are bit-exact, so every comparison is array_equal."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

AUDIO, PAD, IGN = 99, -1, -100


def _cases():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_merge.npz"))
    for n in sorted({k.split("/")[0] for k in z.files}):
        ins = {k.split("/")[2]: z[k] for k in z.files if k.startswith(n + "/in/")}
        outs = {k.split("/")[2]: z[k] for k in z.files if k.startswith(n + "/out/")}
        yield n, ins, outs, str(z[n + "/padding_side"])


def _run(ins, side, dtype):
    from audio_intelligence_amd.multimodal_io.modeling_whisper import merge_input_ids_with_audio_features as merge
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(a).to(dev)
    lab = t(ins["labels"]) if "labels" in ins else None
    return merge(t(ins["audio_features"]).to(dtype), t(ins["num_audio_tokens"]), t(ins["inputs_embeds"]).to(dtype), t(ins["input_ids"]),
                 t(ins["attention_mask"]), lab, audio_token_index=AUDIO, pad_token_id=PAD, ignore_index=IGN, padding_side=side)


def test_merge_matches_reference_golden_f32():
    n = 0
    for name, ins, outs, side in _cases():
        emb, mask, lab, pos, ids = _run(ins, side, torch.float32)
        assert np.array_equal(emb.cpu().numpy(), outs["final_embedding"]), name
        assert np.array_equal(mask.cpu().numpy(), outs["final_attention_mask"]), name
        assert np.array_equal(pos.cpu().numpy(), outs["position_ids"]), name
        assert np.array_equal(ids.cpu().numpy(), outs["final_input_ids"]), name
        if "final_labels" in outs:
            assert np.array_equal(lab.cpu().numpy(), outs["final_labels"]), name
        else:
            assert lab is None
        n += 1
    assert n == 8


def test_merge_bf16_rows_are_exact_copies():
    from oracle import merge as om
    for name, ins, outs, side in _cases():
        emb, *_ = _run(ins, side, torch.bfloat16)
        r = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
        ref, *_ = om.merge_input_ids_with_audio_features(r(ins["audio_features"]), ins["num_audio_tokens"], r(ins["inputs_embeds"]), ins["input_ids"],
                                                         ins["attention_mask"], None, AUDIO, PAD, IGN, side)
        assert np.array_equal(emb.float().cpu().numpy(), ref), name


def test_merge_full_width_rows():
    """AF3-7B geometry: H = 3584 bf16 rows (7168 B), 750 audio tokens, left-padded batch of two."""
    from audio_intelligence_amd.multimodal_io.modeling_whisper import merge_input_ids_with_audio_features as merge
    from oracle import merge as om
    rng = np.random.default_rng(3)
    H, T = 3584, 750
    ids = np.array([[PAD] * 3 + [5, 6, AUDIO, 7, 8], [1, 2, AUDIO, 3, 4, AUDIO, 9, 9]], np.int64)
    am = (ids != PAD).astype(np.int64)
    nat = np.array([750, 125, 500], np.int64)
    feats = rng.standard_normal((3, T, H)).astype(np.float32)
    emb = rng.standard_normal((2, 8, H)).astype(np.float32)
    tb = lambda a: torch.from_numpy(a).to("cuda:0")
    out = merge(tb(feats).bfloat16(), tb(nat), tb(emb).bfloat16(), tb(ids), tb(am), None, audio_token_index=AUDIO, pad_token_id=PAD)
    r = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    ref = om.merge_input_ids_with_audio_features(r(feats), nat, r(emb), ids, am, None, AUDIO, PAD, IGN, "left")
    assert out[0].shape == (2, 5 + 125 + 500 + 6 - 2 + 0, H) or out[0].shape == ref[0].shape
    assert np.array_equal(out[0].float().cpu().numpy(), ref[0])
    for a, b in zip(out[1:], ref[1:]):
        assert (a is None and b is None) or np.array_equal(a.cpu().numpy(), b)


def test_merge_raises_like_the_reference():
    from audio_intelligence_amd.multimodal_io.modeling_whisper import merge_input_ids_with_audio_features as merge
    dev = "cuda:0"
    ids = torch.tensor([[1, AUDIO, 2], [3, 4, 5]], device=dev)
    emb = torch.zeros(2, 3, 8, device=dev)
    with pytest.raises(ValueError):   # zeros on both edges (modeling_whisper.py:1017-1019)
        merge(torch.zeros(1, 4, 8, device=dev), torch.tensor([4], device=dev), emb, ids, torch.tensor([[0, 1, 1], [1, 1, 0]], device=dev),
              audio_token_index=AUDIO)
    with pytest.raises(ValueError):   # two audios offered, one placeholder (modeling_whisper.py:1098-1102)
        merge(torch.zeros(2, 4, 8, device=dev), torch.tensor([4, 2], device=dev), emb, ids, torch.ones(2, 3, dtype=torch.long, device=dev),
              audio_token_index=AUDIO)


def test_projector_matches_linear():
    from audio_intelligence_amd.multimodal_io.modeling_whisper import Qwen2AudioMultiModalProjector
    torch.manual_seed(0)
    proj = Qwen2AudioMultiModalProjector(d_model=384, hidden_size=768).to("cuda:0")
    with torch.no_grad():
        proj.linear.weight.normal_(0, 0.05)
        proj.linear.bias.normal_(0, 0.1)
    x = torch.randn(2, 250, 384, device="cuda:0")
    y = proj(x)
    ref = torch.nn.functional.linear(x.double().cpu(), proj.linear.weight.double().cpu(), proj.linear.bias.double().cpu())
    assert y.shape == (2, 250, 768)
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-4      # f32 GEMM (exact-f32 MFMA), tolerance = accumulation order
