"""Host-side logic of the drop-in classes on CPU (no GPU): vocabulary / batch dict / masks / state-dict keys against
the reference golden, length maths, KV-cache bookkeeping, long-audio windowing -- and the product path's refusal to
run without the HIP device (no CPU fallback)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import fixtures_common as fc
from tests import helpers as H


class _FakeContinuous:
    """stand-in with the AbsIO surface ContinuousAudioIO shows to UALMPreprocessor (no kernels involved)"""
    modality, is_discrete = "audio", False

    def __init__(self, io):
        self.io = io

    def preprocess(self, data):
        wav, sr = data
        n = min(wav.shape[-1], 480000)
        after = self.io._after_length(n)
        pads = np.zeros((after, 1), np.int32)
        return pads, (after, np.zeros((3000, 128), np.float32)), pads

    def find_length(self, data):
        return self.io.find_length(data)


def _cpu_io():
    from audio_intelligence_amd.multimodal_io.audio import ContinuousAudioIO
    io = ContinuousAudioIO(encoder_choice="AFWhisper", dtype="float32", device="cpu", _skip_loading=True)
    io.sample_rate, io.hop_length, io.n_samples, io.d_model = 16000, 160, 480000, 384
    return io


def test_length_table_and_find_length():
    g, _ = H.golden()
    io = _cpu_io()
    for row in g["length_table"]:
        assert io.find_length((np.zeros(row["n"], np.float32), 16000)) == row["find_length"]
        assert io._after_length(min(row["n"], 480000)) == row["after_length"]
    assert io.find_length((np.zeros(80000, np.float32), 8000)) == g["find_length_8k"]
    assert io.feature_dim() == 384 and io.model is None


def test_mel_filter_bank_matches_reference():
    from audio_intelligence_amd.multimodal_io.feature_extraction import mel_filter_bank
    g, _ = H.golden()
    f = mel_filter_bank()
    assert list(f.shape) == g["filters"]["shape"] and int((f != 0).sum()) == g["filters"]["nnz"]
    np.testing.assert_allclose(f.reshape(-1)[g["filters"]["sample_idx"]], g["filters"]["sample_val"], rtol=1e-12, atol=1e-15)


def test_vocabulary_and_collate_against_reference_golden():
    from audio_intelligence_amd import ualm_job
    g, _ = H.golden()
    lcfg = H.tiny_llm()[0]
    text, audio = H.stub_ios(lcfg["text_vocab"])
    ios = {"text": text, "discrete_audio": audio, "continuous_audio": _FakeContinuous(_cpu_io())}
    vocab, iv = ualm_job.build_vocabulary(ios)
    assert len(vocab) == g["llm_tiny"]["vocab_size"]
    assert {k: [list(x) for x in v] for k, v in iv.items()} == g["llm_tiny"]["intervals"]
    pre = ualm_job.UALMPreprocessor(False, ios, vocab, iv)
    data = {"audio": (fc.make_wav(1000, 160000)[None], 16000), "text": [["user", "text", [0, 5, 6, 7]]]}
    b = pre.collate_fn([(("audio_to_caption", "x", "y"), data)])
    c = g["collate_small"]
    assert list(b["seqs"].shape) == c["seqs_shape"] and b["seqs"][0, :, 0].tolist() == c["seqs_stream0"]
    assert b["continuous_audio_indices"].tolist() == c["indices"] and b["continuous_audio_lengths"].tolist() == c["lengths"]
    assert list(b["continuous_audio_feats"].shape) == c["feats_shape"]
    assert pre.find_length(("audio_to_caption", "x", "y"), data) == c["find_length"]
    # ragged batch: right padding with pad id 0, per-sample feature indices (ualm_job.py:262-307)
    d2 = {"audio": (fc.make_wav(1, 80000)[None], 16000), "text": [["user", "text", [9, 9]]]}
    b2 = pre.collate_fn([(("audio_to_caption", "x", "y"), data), (("audio_to_caption", "x", "y"), d2)])
    assert list(b2["seqs"].shape) == [2, 261, 8] and int(b2["seqs"][1, 134:].abs().sum()) == 0
    assert b2["continuous_audio_indices"].tolist() == [[0, 10, 250], [1, 8, 125]]
    # a bad sample is dropped, not fatal (ualm_job.py:236-250); an all-bad batch raises
    b3 = pre.collate_fn([(("audio_to_caption", "x", "y"), data), (("no_such_task", "x", "y"), data)])
    assert b3["seqs"].shape[0] == 1
    with pytest.raises(ValueError):
        pre.collate_fn([(("no_such_task", "x", "y"), data)])


def test_parallel_llm_structure_masks_and_intervals_on_cpu():
    import json, os, tempfile
    from audio_intelligence_amd.lm.parallel import ParallelHFModel, KVCache
    g, _ = H.golden()
    lcfg, lsd, vocab, iv = H.tiny_llm()
    ecfg, esd = H.tiny_enc()
    from audio_intelligence_amd.multimodal_io.modeling_whisper import AFWhisperEncoder, AFWhisperEncoderConfig
    text, audio = H.stub_ios(lcfg["text_vocab"])
    cont = _cpu_io()
    cont.model = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(ecfg))
    with tempfile.TemporaryDirectory() as d:
        json.dump({"architectures": ["Qwen2ForCausalLM"], "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12,
                   "num_key_value_heads": 2, "intermediate_size": 3072, "rope_theta": 1e6, "rms_norm_eps": 1e-6}, open(os.path.join(d, "config.json"), "w"))
        model = ParallelHFModel(d, multimodal_io={"text": text, "discrete_audio": audio, "continuous_audio": cont}, vocab=vocab, vocab_intervals=iv)
    full = dict(lsd)
    full.update({"multimodal_io_dict.continuous_audio.model." + k: v for k, v in esd.items()})
    r = model.load_state_dict(full, strict=True)              # reference checkpoint key set, strict (scripts/inference.py:150-152)
    assert not r.missing_keys and not r.unexpected_keys
    keys = sorted(model.state_dict().keys())
    assert len(keys) == g["llm_tiny"]["n_state_dict_keys"]
    assert hashlib.sha256("\n".join(keys).encode()).hexdigest() == g["llm_tiny"]["state_dict_keys_sha256"]
    model.prepare_inference()
    L = g["llm_tiny"]
    assert model.modality_mask[0, 0].sum(-1).tolist() == L["modality_mask_rowsum"]
    assert model.text_mask[0, 0].sum(-1).tolist() == L["text_mask_rowsum"]
    assert model.audio_mask[0, 0].sum(-1).tolist() == L["audio_mask_rowsum"]
    ivt, pad_only = model._allowed_intervals("text")
    assert ivt.tolist() == [[2, 4], [256, 256 + 16384]] and pad_only
    ivm, _ = model._allowed_intervals("modality")
    assert ivm.tolist() == [[7, 9], [10, 13]]
    _, pad_only_audio = model._allowed_intervals("audio")
    assert not pad_only_audio
    assert model.num_stream == 8 and model.eos_token_id == 2 and model.eot_token_id == 3
    # no CPU fallback: compute on a CPU-resident model must raise, not silently run
    from audio_intelligence_amd._lib import AfhipError
    with pytest.raises(AfhipError):
        model._step(input_ids=torch.zeros(1, 1, 8, dtype=torch.long))
    c = KVCache(2, 1, 2, 64, 64, torch.float32, "cpu")
    c.k[:, :, :, :10] = 1.0
    c.length = 10
    c.reserve(100)
    assert c.cap == 128 and float(c.k[:, :, :, :10].min()) == 1.0 and float(c.k[:, :, :, 10:].abs().max()) == 0.0
    c.batch_select_indices(torch.zeros(3, dtype=torch.long))
    assert c.batch == 3 and c.get_seq_length() == 10


def test_product_path_refuses_cpu():
    from audio_intelligence_amd import ops
    from audio_intelligence_amd._lib import AfhipError
    from audio_intelligence_amd.multimodal_io.feature_extraction import WhisperFeatureExtractorHIP
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(AfhipError):
        ops.gemm(torch.zeros(128, 64), torch.zeros(128, 64))
    with pytest.raises(AfhipError):
        WhisperFeatureExtractorHIP()(np.zeros(16000, np.float32), sampling_rate=16000)


def test_long_audio_windowing():
    from audio_intelligence_amd.long_audio import split_windows, shard_range
    assert split_windows(0) == []
    assert split_windows(480000) == [(0, 480000)]
    w = split_windows(9600000)                       # 10 minutes
    assert len(w) == 20 and w[-1] == (9120000, 9600000)
    assert split_windows(500000) == [(0, 480000), (480000, 500000)]
    for n, world in ((80, 8), (20, 8), (3, 8), (7, 2)):
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        sizes = [h - l for l, h in spans]
        assert max(sizes) - min(sizes) <= 1


def test_merge_refuses_cpu():
    """the product merge has no CPU fallback (the oracle restatement lives in oracle/merge.py)."""
    import torch
    from audio_intelligence_amd import _lib as L
    from audio_intelligence_amd.multimodal_io.modeling_whisper import merge_input_ids_with_audio_features as merge
    with pytest.raises(L.AfhipError):
        merge(torch.zeros(1, 2, 8), torch.tensor([2]), torch.zeros(1, 3, 8), torch.tensor([[1, 99, 2]]), torch.ones(1, 3, dtype=torch.long),
              audio_token_index=99)


def test_checkpoint_roundtrip_and_errors(tmp_path):
    """DeepSpeed `mp_rank_00_model_states.pt`["module"] ingestion (scripts/inference.py:136-153)."""
    import torch
    from audio_intelligence_amd import inference as inf
    from audio_intelligence_amd.multimodal_io.modeling_whisper import AFWhisperEncoder, AFWhisperEncoderConfig
    ecfg, esd = H.tiny_enc()
    enc = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(ecfg))
    enc.load_state_dict(esd)
    path = inf.save_checkpoint(enc, str(tmp_path / "ckpt"), tag="global_step7")
    assert path.endswith("global_step7/mp_rank_00_model_states.pt")
    enc2 = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(ecfg))
    for p in enc2.parameters():
        torch.nn.init.zeros_(p)
    inf.load_checkpoint(enc2, str(tmp_path / "ckpt"))                      # via the `latest` tag file
    for k, v in enc.state_dict().items():
        assert torch.equal(v, enc2.state_dict()[k]), k
    inf.load_checkpoint(enc2, path)                                         # the file itself
    inf.load_checkpoint(enc2, str(tmp_path / "ckpt" / "global_step7"))      # the tag directory
    torch.save({"not_module": {}}, str(tmp_path / "bad.pt"))
    with pytest.raises(KeyError):
        inf.load_checkpoint(enc2, str(tmp_path / "bad.pt"))
    sd = {k: v for k, v in enc.state_dict().items() if not k.startswith("conv1")}
    torch.save({"module": sd}, str(tmp_path / "short.pt"))
    with pytest.raises(RuntimeError):                                        # strict=True, as the reference
        inf.load_checkpoint(enc2, str(tmp_path / "short.pt"))
    with pytest.raises(FileNotFoundError):
        inf.load_checkpoint(enc2, str(tmp_path / "nowhere"))


def test_to_device_casts_floats_only():
    """utils/data.py:93-130: floats take the model dtype, integer tensors keep theirs, containers are walked."""
    import torch
    from audio_intelligence_amd import inference as inf
    d = {"seqs": torch.zeros(1, 3, 8, dtype=torch.long), "feats": [torch.zeros(2, 2), np.zeros((2,), np.float32)], "keys": [("a", "b", "c")],
         "n": 3}
    o = inf.to_device(d, "cpu", dtype=torch.bfloat16)
    assert o["seqs"].dtype == torch.long and o["feats"][0].dtype == torch.bfloat16 and o["feats"][1].dtype == torch.bfloat16
    assert o["keys"] == [("a", "b", "c")] and o["n"] == 3


def test_af3_prepare_inputs_for_generation_known_answers():
    """The three input-slicing rules + on-the-fly position ids of Qwen2AudioForConditionalGeneration.prepare_inputs_for_generation
    (modeling_whisper.py:1250-1318) against answers captured from the reference function (oracle/make_golden_af3.py)."""
    import json
    import os
    import torch
    from audio_intelligence_amd.multimodal_io.modeling_whisper import Qwen2AudioForConditionalGeneration as Q
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_af3.json")) as f:
        g = json.load(f)

    class Past:
        def __init__(self, n):
            self.n = n

        def get_seq_length(self):
            return self.n

    fake = type("S", (), {"audio_token_index": g["audio_token_index"]})()
    for c in g["prepare_inputs"]:
        past = None if c["past"] is None else Past(c["past"])
        r = Q.prepare_inputs_for_generation(fake, torch.tensor(c["input_ids"]), past_key_values=past, input_features=(object() if c["feat"] else None),
                                            attention_mask=(torch.tensor(c["mask"]) if c.get("mask") is not None else None))
        e = c["expect"]
        assert r["input_ids"].tolist() == e["input_ids"], c["name"]
        assert (None if r["position_ids"] is None else r["position_ids"].tolist()) == e["position_ids"], c["name"]
        assert (None if r["attention_mask"] is None else r["attention_mask"].tolist()) == e["attention_mask"], c["name"]


def test_torch_library_ops_registered_with_meta_kernels():
    """The stateless C-ABI entry points are also `torch.ops.afhip.*` custom ops (BASELINE north star / SURVEY 8b): registered with
    schemas and meta kernels, so shapes propagate under FakeTensorMode with no GPU; there is no CPU kernel (no fallback)."""
    import torch
    import audio_intelligence_amd.torch_ops as T
    from torch._subclasses.fake_tensor import FakeTensorMode
    assert T.OP_NAMES == sorted(["gemm", "layernorm", "rmsnorm", "embed_sum", "attention_packed", "log_mel", "quant_rows", "gemm_fp8",
                                 "encoder_forward", "encoder_forward_ragged", "llm_forward", "llm_forward_ragged", "lm_head", "llm_decode_step"])
    for n in T.OP_NAMES:
        assert hasattr(torch.ops.afhip, n)
    with FakeTensorMode():
        a = torch.empty((300, 1280), dtype=torch.bfloat16, device="cuda")
        w = torch.empty((5120, 1280), dtype=torch.bfloat16, device="cuda")
        y = torch.ops.afhip.gemm(a, w, None, 1, None)
        assert tuple(y.shape) == (300, 5120) and y.dtype == torch.bfloat16
        g = torch.ops.afhip.gemm(a, w, None, 2, None)
        assert tuple(g.shape) == (300, 2560)
        q, s = torch.ops.afhip.quant_rows(a, 1, w[0], w[0], 1e-5)
        assert q.dtype == torch.uint8 and tuple(s.shape) == (300,)
        mel = torch.ops.afhip.log_mel(torch.empty((4, 480000), device="cuda"), True, torch.bfloat16)
        assert tuple(mel.shape) == (4, 3000, 128) and mel.dtype == torch.bfloat16
        att = torch.ops.afhip.attention_packed(torch.empty((2, 1500, 3840), dtype=torch.bfloat16, device="cuda"), 20, None, False)
        assert tuple(att.shape) == (2, 1500, 1280)
    # the stateful entry points (the ones that carry the time) take the packed-weight struct as an opaque uint8 CPU tensor that aliases it
    from audio_intelligence_amd import _lib as L
    ew = L.EncoderWeights()
    ew.max_pos, ew.d_model = 1500, 1280
    blob = T.weights_blob(ew)
    assert blob.dtype == torch.uint8 and blob.numel() == __import__("ctypes").sizeof(L.EncoderWeights)
    ew.d_model = 384                                                   # aliasing, not a copy
    out, hid = torch.ops.afhip.encoder_forward(blob, torch.empty((2, 3000, 128), device="meta"), None, -2, torch.empty(8, dtype=torch.uint8, device="meta"))
    assert tuple(out.shape) == (2, 750, 384) and hid.numel() == 0
    out, hid = torch.ops.afhip.encoder_forward(blob, torch.empty((2, 3000, 128), device="meta"), None, 0, torch.empty(8, dtype=torch.uint8, device="meta"))
    assert tuple(hid.shape) == (2, 1500, 384)
    lw = L.LlmWeights()
    lw.vocab = 24840
    lb = T.weights_blob(lw)
    lg = torch.ops.afhip.lm_head(lb, torch.empty((4, 768), device="meta"), 8, torch.empty(8, dtype=torch.uint8, device="meta"))
    assert tuple(lg.shape) == (4, 8, 24840) and lg.dtype == torch.float32
    x = torch.empty((2, 5, 768), dtype=torch.bfloat16, device="meta")
    kv = torch.empty((12, 2, 2, 64, 64), dtype=torch.bfloat16, device="meta")
    assert tuple(torch.ops.afhip.llm_forward(lb, x, 0, kv, kv.clone(), torch.empty(8, dtype=torch.uint8, device="meta")).shape) == (2, 5, 768)
    import pytest as _pt
    with _pt.raises(L.AfhipError):
        torch.ops.afhip.lm_head(blob, torch.empty((4, 768), device="meta"), 8, torch.empty(8, dtype=torch.uint8, device="meta"))   # wrong struct
    # real CPU tensors: no kernel for that backend -> an error, never a silent fallback
    import pytest
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.afhip.rmsnorm(torch.zeros(2, 8), torch.ones(8), 1e-6)


def test_discrete_audio_decode_batch_through_an_offline_xcodec():
    """`transformers.XcodecModel` DOES construct offline from its default config (16 kHz, hop 320, codebooks of
    1024; the reference fetches trained weights by tag, audio.py:203-218 -- here seeded random ones).  With it attached,
    DiscreteAudioTokenIO.decode_batch finishes the reference's decode_batch (audio.py:494-541 + _codec_decode_batch 573-588): delay
    de-interleave, vocabulary offsets removed, pad ids (-1) decoded as entry 0, [B, S, T] into codec.decode, lengths x hop.  The expected
    waveform is the same codec called directly on the codes.  CPU on purpose: the codec network is the caller's HF module, not HIP code."""
    import pytest
    transformers = pytest.importorskip("transformers")
    if not hasattr(transformers, "XcodecModel"):
        pytest.skip("transformers without XcodecModel")
    from audio_intelligence_amd.multimodal_io.audio import DiscreteAudioTokenIO, delay_interleave
    from audio_intelligence_amd.utils.synthetic import make_offline_xcodec
    codec = make_offline_xcodec(0)        # seeded weights AND seeded residual-VQ codebooks (the constructor leaves those at zero)
    assert codec.config.sample_rate == 16000 and codec.config.hop_length == 320 and codec.config.codebook_size == 1024
    io = DiscreteAudioTokenIO(n_stream=8, codebook_size=1024).attach_codec(codec)
    g = torch.Generator().manual_seed(1)
    B, T = 2, 12
    codes = torch.randint(0, 1024, (B, T, 8), generator=g)
    codes[1, -3:, :] = -1                                                   # a shorter clip: its tail holds pad entries
    rel = torch.stack([codes[..., s] + s * 1025 + 1 for s in range(8)], dim=-1)          # ids inside the IO's vocabulary (slot 0 = pad)
    inter = delay_interleave(rel, [s * 1025 for s in range(8)])
    with torch.no_grad():
        audio, alen, sr = io.decode_batch(inter, torch.tensor([T + 7, T + 7 - 3]))
        want = codec.decode(torch.where(codes < 0, torch.zeros_like(codes), codes).permute(0, 2, 1)).audio_values
    assert sr == 16000 and alen.tolist() == [T * 320, (T - 3) * 320]
    assert audio.shape == (B, 1, T * 320) and bool(torch.isfinite(audio).all())
    assert torch.equal(audio, want)
    # without a codec the method still stops at the codes (what the GPU generation tests pin against the reference's golden file)
    back, lens = DiscreteAudioTokenIO().decode_batch(inter, torch.tensor([T + 7, T + 7]))
    assert torch.equal(back, codes) and lens.tolist() == [T, T]


def test_discrete_audio_io_matches_the_reference_class_fixture():
    """SURVEY 8f-4, codec leg, PINNED: tests/golden/golden_codec.* were captured from the reference's own
    `DiscreteAudioIO(codec_choice="Xcodec", delay_interleave=True, _skip_loading=True)` (multimodal_io/audio.py:80-140) with the seeded
    offline X-codec attached (oracle/make_golden_codec.py): its find_length (:656-672), encode_batch (:416-492: whole-frame trim, first
    channel, bandwidth, offset + 1, replicate pad, delay interleave, a shorter clip in the batch) and decode_batch / _codec_decode_batch
    (:494-596: de-interleave, offsets removed, pad ids -> entry 0, [B, S, T], lengths x hop).  DiscreteAudioTokenIO must reproduce the ids
    exactly and the waveforms to float rounding."""
    import json
    import os
    import numpy as np
    import pytest
    transformers = pytest.importorskip("transformers")
    if not hasattr(transformers, "XcodecModel"):
        pytest.skip("transformers without XcodecModel")
    from audio_intelligence_amd.multimodal_io.audio import DiscreteAudioTokenIO
    from audio_intelligence_amd.utils import synthetic as syn
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    G = json.load(open(os.path.join(gold_dir, "golden_codec.json")))
    A = np.load(os.path.join(gold_dir, "golden_codec_arrays.npz"))
    codec = syn.make_offline_xcodec(G["codec_seed"])
    if abs(syn.xcodec_fingerprint(codec) - G["fingerprint"]) > 1e-6 * G["fingerprint"]:
        pytest.skip("the seeded offline X-codec differs from the one the fixture was captured with (torch / transformers version skew)")
    io = DiscreteAudioTokenIO(n_stream=8, codebook_size=1024).attach_codec(codec)
    assert io.sample_rate == G["sample_rate"] and io.frame_shift == G["frame_shift"] and io.codec_bandwidth == G["bandwidth"]
    assert io.num_stream() == G["num_stream"] and [list(iv) for iv in io.get_stream_interval()] == G["stream_intervals"]
    assert len(io.get_vocabulary()) == G["vocab_size"]
    for n, sr, want in G["find_length"]:
        assert io.find_length((np.zeros((1, n), np.float32), sr)) == want, (n, sr)
    g = torch.Generator().manual_seed(G["wav_seed"])
    n0, n1 = G["n_samples"]
    wav = torch.randn(2, n0, 1, generator=g) * 0.1
    step = G["sample_step"]
    with torch.no_grad():
        ids = io.encode_batch(wav, torch.tensor([n0, n1]))
        gold_ids = torch.from_numpy(A["encode_ids"]).long()
        assert ids.shape == gold_ids.shape and torch.equal(ids, gold_ids), f"{int((ids != gold_ids).sum())} ids differ from the reference's"
        T = gold_ids.shape[1]
        audio, alen, sr = io.decode_batch(gold_ids, torch.tensor([T, n1 // 320 + 7]))
        assert list(audio.shape) == G["decode_audio_shape"] and [int(x) for x in alen] == G["decode_lengths"] and sr == G["decode_sample_rate"]
        assert float((audio[:, 0, ::step] - torch.from_numpy(A["decode_audio_sample"])).abs().max()) <= 1e-5
        for b in range(2):
            assert abs(float(audio[b].double().sum()) - G["decode_audio_sum"][b]) <= 1e-3 * max(1.0, abs(G["decode_audio_sum"][b]))
        audio2, alen2, _ = io.decode_batch(torch.from_numpy(A["pad_ids"]).long(), torch.tensor([T, T]))
        assert [int(x) for x in alen2] == G["pad_lengths"]
        assert float((audio2[:, 0, ::step] - torch.from_numpy(A["pad_audio_sample"])).abs().max()) <= 1e-5
        assert float((audio2 - audio).abs().max()) > 1e-4          # the pad entries really changed the waveform (decoded as entry 0)


def test_discrete_audio_encode_batch_through_an_offline_xcodec():
    """audio.py:417-491 (codec-only configuration) with the offline X-codec: wav [B, samples, 1] -> ids in the IO's vocabulary, delay
    interleaved.  Expected = the codec called directly (audio.py:641-654: first channel, trimmed to whole frames, the target bandwidth
    closest to 8 quantizers) + stream offsets + the delay pattern; decode_batch (without a codec) takes the ids back to the codes."""
    import pytest
    transformers = pytest.importorskip("transformers")
    if not hasattr(transformers, "XcodecModel"):
        pytest.skip("transformers without XcodecModel")
    from audio_intelligence_amd.multimodal_io.audio import DiscreteAudioTokenIO, delay_interleave
    from audio_intelligence_amd.utils.synthetic import make_offline_xcodec
    codec = make_offline_xcodec(0)        # seeded weights AND seeded residual-VQ codebooks (the constructor leaves those at zero)
    io = DiscreteAudioTokenIO(n_stream=8, codebook_size=1024).attach_codec(codec)
    assert io.codec_bandwidth == 4 and io.frame_shift == 320
    g = torch.Generator().manual_seed(2)
    wav = torch.randn(2, 16000 + 123, 1, generator=g) * 0.1
    lengths = torch.tensor([16000 + 123, 9000])
    assert io.find_length((wav[0].T.numpy(), 16000)) == (16000 + 123) // 320 + 7
    assert io.find_length((wav[0].T.numpy(), 8000)) == int((16000 + 123) * 2 // 320) + 7
    with torch.no_grad():
        ids = io.encode_batch(wav, lengths)
        T = (16000 + 123) // 320
        direct = codec.encode(wav.transpose(1, 2)[:, :1, : T * 320], bandwidth=4, return_dict=False).permute(0, 2, 1)[:, :, :8]
    assert direct.shape[1] == T
    rel = torch.stack([direct[..., s].long() + s * 1025 + 1 for s in range(8)], dim=-1)
    assert ids.shape == (2, T + 7, 8) and torch.equal(ids, delay_interleave(rel, [s * 1025 for s in range(8)]))
    back, lens = DiscreteAudioTokenIO().decode_batch(ids, torch.tensor([T + 7, T + 7]))
    assert torch.equal(back, direct.long()) and lens.tolist() == [T, T]
    with pytest.raises(RuntimeError):
        DiscreteAudioTokenIO().encode_batch(wav, lengths)
    with pytest.raises(ValueError):
        io.encode_batch(wav[:, :, 0], lengths)
