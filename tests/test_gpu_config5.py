"""BASELINE configs 3 and 5 at the shapes the bench times: the e4m3 GEMM at the AF3-7B prefill shapes,
the 7B widths at B = 16 in bf16 and W8A16, one full-depth (28 layers) run, and the token-level contract of the fp8 encoder mode.
Tolerances are written next to each assertion; references are fp32 matmuls of the same fp8-rounded operands (operand level),
oracle/qwen2.py on the same rounded weights (model level) and the bf16 path itself (fp8 encoder: SURVEY 8d config 5 names
"token-match rate vs bf16 + logit RMSE" as the tolerance -- the reference has no fp8 path)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import fixtures_common as fc
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SLACK = 1.5


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")


def _q(t):
    from audio_intelligence_amd.utils.quant import quantize_rows_e4m3
    q, s = quantize_rows_e4m3(t)
    return q, s, q.view(torch.float8_e4m3fn).float() * s[:, None]


def _tile_rows(M, per_tile=4, tile=256, seed=0):
    """first / last / two random rows of every 256-row output tile: a wrong tile cannot hide between samples"""
    rng = np.random.default_rng(seed)
    rows = []
    for t0 in range(0, M, tile):
        t1 = min(M, t0 + tile)
        rows += [t0, t1 - 1] + rng.integers(t0, t1, size=per_tile - 2).tolist()
    return torch.tensor(sorted(set(rows)))


# T = 790 positions x B = 8 / 16 sequences (bench.py decode_leg): M = 6320 / 12640
@pytest.mark.parametrize("M", [6320, 12640])
@pytest.mark.parametrize("which", ["qkv", "gate_up_swiglu", "down_residual", "o_residual"])
def test_gemm_fp8_at_7b_prefill_shapes(M, which):
    """afhip_gemm a_fp8 at the AF3-7B prefill shapes bench.py times: q|k|v (N 4608, K 3584, bias), gate|up with the SwiGLU epilogue
    (N 37888 interleaved in 32-row blocks, K 3584), down (N 3584, K 18944, + residual), o (N 3584, K 3584, + residual).
    Reference: fp32 matmul of the SAME e4m3-rounded operands with the scales applied -- on the GPU over the whole output (torch
    fp32 matmul, independent of this library) and on the CPU over four rows of every 256-row tile.  What may differ: f32
    accumulation order and the bf16 rounding of the output: |err| <= 2e-2 + 1e-2 |ref| (test_gpu_fp8.py's operand-level bound)."""
    _need_gpu()
    from audio_intelligence_amd import ops, _lib as L
    N, K = {"qkv": (4608, 3584), "gate_up_swiglu": (37888, 3584), "down_residual": (3584, 18944), "o_residual": (3584, 3584)}[which]
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    a[5, 7] = 40.0                                              # an outlier sets that row's scale
    w = torch.randn(N, K, generator=g) * 0.03
    aq, sa, ad = _q(a)
    wq, sw, wd = _q(w)
    d = lambda t: t.to(DEV)
    rows = _tile_rows(M, seed=N)
    torch.set_num_threads(H.cpu_threads())
    if which == "qkv":
        bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
        out = ops.gemm_fp8(d(aq), d(sa), d(wq), d(sw), bias=d(bias))
        ref_gpu = d(ad) @ d(wd).T + d(bias).float()
        ref_cpu = ad[rows] @ wd.T + bias.float()
    elif which == "gate_up_swiglu":
        I = N // 2
        packed = torch.stack([wq[:I].view(I // 32, 32, K), wq[I:].view(I // 32, 32, K)], dim=1).reshape(N, K).contiguous()
        psc = torch.stack([sw[:I].view(I // 32, 32), sw[I:].view(I // 32, 32)], dim=1).reshape(N).contiguous()
        out = ops.gemm_fp8(d(aq), d(sa), d(packed), d(psc), act=L.ACT_SWIGLU)
        assert out.shape == (M, I)
        adg, wdg = d(ad), d(wd)
        ref_gpu = F.silu(adg @ wdg[:I].T) * (adg @ wdg[I:].T)
        ref_cpu = F.silu(ad[rows] @ wd[:I].T) * (ad[rows] @ wd[I:].T)
    else:
        res = torch.randn(M, N, generator=g).to(torch.bfloat16)
        out = ops.gemm_fp8(d(aq), d(sa), d(wq), d(sw), residual=d(res))
        ref_gpu = d(ad) @ d(wd).T + d(res).float()
        ref_cpu = ad[rows] @ wd.T + res[rows].float()
    assert bool(torch.isfinite(out).all())
    err = (out.float() - ref_gpu).abs()
    lim = 2e-2 + 1e-2 * ref_gpu.abs()
    bad = int((err > lim).sum())
    assert bad == 0, (which, M, bad, float(err.max()))
    err_c = (out[rows.to(DEV)].float().cpu() - ref_cpu).abs()
    assert bool((err_c <= 2e-2 + 1e-2 * ref_cpu.abs()).all()), (which, M, float(err_c.max()))


def _wide_inputs(vocab_text, B, n_prompt=32, seed=5):
    rng = np.random.default_rng(seed)
    S = 8
    seqs, feats = [], []
    for b in range(B):
        prompt = rng.integers(1, vocab_text, size=n_prompt).tolist()
        rows = [1, 5, 7] + [256 + t for t in prompt] + [3, 5, 8] + [0] * 750 + [2]
        s = torch.zeros((len(rows), S), dtype=torch.int64)
        s[:, 0] = torch.tensor(rows)
        seqs.append(s)
        g = torch.Generator().manual_seed(100 + b)
        feats.append((torch.randn(750, 1280, generator=g) * 0.8).to(torch.bfloat16))
    return torch.stack(seqs), torch.stack(feats), 3 + n_prompt + 3


def _gold7b():
    import json
    import os
    with open(os.path.join(H.GOLD_DIR, "golden_7b.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("mode", ["bf16", "w8a16"])
def test_llm_7b_widths_batch_of_16_invariance_and_oracle(mode):
    """BASELINE config 5 at its batch: 7B widths (2 layers), B = 16, T = 790 prefill + 8 greedy steps, bf16 weights and W8A16
    (e4m3 weights + per-row scale in the decode step, bf16 prefill) -- what bench.py's `decode_b16` leg runs.
    (a) a clip decoded alone gives the same BITS as inside the batch of 16 (batching is allowed only if per-clip ids
    equal the B = 1 results); (b) teacher-forced against oracle/qwen2.py in fp32 on the weights the device really multiplies with
    (bf16-rounded; for W8A16 the dequantised e4m3 rows): every pick is an allowed id within 1.5 x the reference's own bf16 regret
    (golden_7b.json) of the oracle's best logit."""
    _need_gpu()
    from audio_intelligence_amd.utils.quant import quantize_rows_e4m3
    B, n_dec = 16, 8
    ref16 = _gold7b()["bf16"]
    model, sd, cfg, vocab, iv = H.build_wide_llm(n_layers=2, dtype=torch.bfloat16, device=DEV)
    if mode == "w8a16":
        model.enable_fp8_decode(True)
    seqs, feats, start = _wide_inputs(cfg["text_vocab"], B)
    kw = {"seqs": seqs.to(DEV), "continuous_audio_feats": feats.to(DEV), "continuous_audio_lengths": torch.full((B,), 750, dtype=torch.long)}
    ids = torch.cat([kw["seqs"], model.assistant_token.expand(B, -1, -1)], dim=1)
    T = ids.shape[1]
    assert T == 790
    model.eos_token_id = model.eot_token_id = -1

    def run(sel):
        k = {"seqs": kw["seqs"][sel], "continuous_audio_feats": kw["continuous_audio_feats"][sel],
             "continuous_audio_lengths": kw["continuous_audio_lengths"][sel],
             "continuous_audio_indices": torch.tensor([[j, start, 750] for j in range(len(sel))])}
        emb = model._embed(ids[sel], k)
        hid, cache = model._forward_hidden(emb, model.new_cache(len(sel), T + n_dec + 8))
        logits = model._head_stream0(hid[:, -1])
        tok = model.text_token.expand(len(sel), -1, -1).clone()
        hyp, _, cache = model._greedy_device_loop(tok, cache, "text", n_dec, poll=10 ** 9)
        return emb, hid, logits, hyp[:, :, 0]

    emb, hid, logits, toks = run(list(range(B)))
    assert bool(torch.isfinite(hid).all()) and bool(torch.isfinite(logits).all())
    for b in (0, 9, 15):
        e1, h1, l1, t1 = run([b])
        assert torch.equal(e1[0], emb[b]) and torch.equal(h1[0], hid[b]), f"clip {b}: prefill differs between B=1 and B=16"
        assert torch.equal(l1[0], logits[b]), f"clip {b}: prefill logits differ between B=1 and B=16"
        assert t1[0].tolist() == toks[b].tolist(), f"{mode} clip {b}: B=1 ids {t1[0].tolist()} != batched ids {toks[b].tolist()}"

    torch.set_num_threads(H.cpu_threads())
    ref_emb = F.embedding(ids.cpu(), sd["model.embed_tokens.weight"]).sum(dim=2)
    ref_emb[:, start:start + 750] = F.linear(feats.float(), sd["adaptor.continuous_audio.weight"], sd["adaptor.continuous_audio.bias"])
    ref_hid, ref_cache = oracle.qwen2.forward(ref_emb, sd, cfg)
    h_err = (hid.float().cpu() - ref_hid).abs()
    print(f"7B widths, 2 layers, B=16 ({mode}): prefill hidden err max {float(h_err.max()):.4f} mean {float(h_err.mean()):.5f}")
    assert float(h_err.mean()) <= SLACK * ref16["hid_err"]["mean"] and float(h_err.max()) <= SLACK * ref16["hid_err"]["max"]
    assert float(h_err.mean()) <= 0.1                                   # the HIP path's own mark at B = 8 (measured 0.044)
    sd_dec = sd
    if mode == "w8a16":
        # the decode step multiplies with e4m3 rows x f32 scale: give the oracle exactly those values
        sd_dec = dict(sd)
        for k_, v in sd.items():
            if k_ == "lm_head.weight" or (k_.startswith("model.layers.") and k_.endswith("_proj.weight")):
                q8, sc = quantize_rows_e4m3(v.to(torch.bfloat16))
                sd_dec[k_] = q8.view(torch.float8_e4m3fn).float() * sc[:, None]
    allowed = torch.zeros(len(vocab), dtype=torch.bool)
    ts, te = iv["text"][0]
    allowed[ts:te] = True
    allowed[2] = allowed[3] = True
    eps_regret = SLACK * max(ref16["regret_in_f32_logits"])
    prev = torch.zeros(B, 1, 8, dtype=torch.long)
    prev[:, 0, 0] = oracle.ualm.special_id("<|text|>")
    exact, worst = 0, 0.0
    for st in range(n_dec):
        x = F.embedding(prev, sd["model.embed_tokens.weight"]).sum(dim=2)
        h, ref_cache = oracle.qwen2.forward(x, sd_dec, cfg, ref_cache)
        lg = F.linear(h[:, -1], sd_dec["lm_head.weight"]).masked_fill(~allowed[None], float("-inf"))
        for b in range(B):
            pick = int(toks[b, st])
            assert bool(allowed[pick]), f"step {st} clip {b}: id {pick} outside the text interval / eos / eot"
            regret = float(lg[b].max() - lg[b, pick])
            worst = max(worst, regret)
            assert regret <= eps_regret, f"{mode} step {st} clip {b}: pick {pick} is {regret:.4f} below the oracle's best (eps {eps_regret:.4f})"
            exact += int(pick == int(lg[b].argmax()))
        prev = torch.zeros(B, 1, 8, dtype=torch.long)
        prev[:, 0, 0] = toks[:, st].cpu()
    print(f"7B-width greedy B=16 ({mode}): {exact}/{B * n_dec} picks equal the fp32 oracle's argmax, worst regret {worst:.3f} (eps {eps_regret:.3f})")


def _build_7b_on_device(n_layers, seed=2):
    """AF3-7B shape with `n_layers` layers, weights generated ON the device (28 layers = 15 GB: no CPU copy, no oracle)."""
    from audio_intelligence_amd.lm.parallel import ParallelLLM
    from audio_intelligence_amd import ualm_job
    from audio_intelligence_amd.utils import synthetic as syn
    cfg = H.wide_llm_cfg(n_layers)
    text_io, audio_io = H.stub_ios(cfg["text_vocab"])
    ios = {"text": text_io, "discrete_audio": audio_io, "continuous_audio": H.stub_continuous(1280)}
    vocab, iv = ualm_job.build_vocabulary(ios)
    hf = {"architectures": ["Qwen2ForCausalLM"], "hidden_size": cfg["hidden_size"], "num_hidden_layers": n_layers,
          "num_attention_heads": cfg["num_attention_heads"], "num_key_value_heads": cfg["num_key_value_heads"],
          "intermediate_size": cfg["intermediate_size"], "rope_theta": cfg["rope_theta"], "rms_norm_eps": cfg["rms_norm_eps"],
          "vocab_size": cfg["text_vocab"]}
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(DEV):
            model = ParallelLLM(hf, ios, vocab, iv)
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(syn.synth_tensor(n, p.shape, seed, dtype=torch.bfloat16, device=DEV))
    model.prepare_inference()
    model.eval()
    return model, cfg, vocab, iv


@pytest.mark.parametrize("mode", ["bf16", "w8a16"])
def test_llm_7b_full_depth_properties(mode):
    """BASELINE config 3 at full depth: 28 layers, B = 8, T = 790 + 16 greedy steps -- the run bench.py times, checked through
    size-independent properties (no oracle at this size): everything finite, every id inside the text interval / eos / eot,
    a clip decoded alone gives the same bits as inside the batch, a second run reproduces the first bit for bit, and the KV cache
    positions behind the context stay untouched."""
    _need_gpu()
    B, n_dec = 8, 16
    model, cfg, vocab, iv = _build_7b_on_device(28)
    if mode == "w8a16":
        model.enable_fp8_decode(True)
    seqs, feats, start = _wide_inputs(cfg["text_vocab"], B)
    seqs, feats = seqs.to(DEV), feats.to(DEV)
    ids = torch.cat([seqs, model.assistant_token.expand(B, -1, -1)], dim=1)
    T = ids.shape[1]
    model.eos_token_id = model.eot_token_id = -1

    def run(sel):
        k = {"seqs": seqs[sel], "continuous_audio_feats": feats[sel], "continuous_audio_lengths": torch.full((len(sel),), 750, dtype=torch.long),
             "continuous_audio_indices": torch.tensor([[j, start, 750] for j in range(len(sel))])}
        emb = model._embed(ids[sel], k)
        cache = model.new_cache(len(sel), T + n_dec + 8)
        cache.k.fill_(7.0)
        cache.v.fill_(7.0)                                                # sentinel behind the context
        hid, cache = model._forward_hidden(emb, cache)
        logits = model._head_stream0(hid[:, -1])
        tok = model.text_token.expand(len(sel), -1, -1).clone()
        hyp, _, cache = model._greedy_device_loop(tok, cache, "text", n_dec, poll=10 ** 9)
        return hid, logits, hyp[:, :, 0], cache

    hid, logits, toks, cache = run(list(range(B)))
    assert bool(torch.isfinite(hid).all()) and bool(torch.isfinite(logits).all())
    ts, te = iv["text"][0]
    t = toks.cpu()
    assert bool((((t >= ts) & (t < te)) | (t == 2) | (t == 3)).all()), t
    used = cache.length
    assert used == T + n_dec, (used, T, n_dec)
    assert bool((cache.k[:, :, :, used:] == 7.0).all()) and bool((cache.v[:, :, :, used:] == 7.0).all()), "KV cache written behind the context"
    assert bool(torch.isfinite(cache.k[:, :, :, :used].float()).all())
    hid2, logits2, toks2, _ = run(list(range(B)))
    assert torch.equal(hid, hid2) and torch.equal(logits, logits2) and torch.equal(toks, toks2), "28-layer run is not reproducible"
    for b in (0, 6):
        h1, l1, t1, _ = run([b])
        assert torch.equal(h1[0], hid[b]) and torch.equal(l1[0], logits[b]), f"clip {b}: 28-layer prefill differs between B=1 and B=8"
        assert t1[0].tolist() == toks[b].tolist(), f"{mode} clip {b}: B=1 ids {t1[0].tolist()} != batched {toks[b].tolist()}"
    print(f"28 layers, B=8, T={T}+{n_dec} ({mode}): finite, ids in the allowed set, batch-invariant, reproducible; first ids {toks[0].tolist()[:8]}")


@pytest.mark.parametrize("mode", ["bf16", "w8a16"])
@pytest.mark.parametrize("B", [5, 11])
def test_llm_7b_decode_odd_batch_long_context(mode, B):
    """The imaged decode step (decode_phases.hip) away from the benchmark's shape: an ODD batch (5: 8-row activation images with three empty
    rows; 11: 16-row images) at a context of 2300 positions -- 18 key ranges of 128, the merge's general path (more than 16 ranges) writing the
    o-projection's image -- decoding across 2304 = 18 x 128, so the number of live ranges grows inside the replayed graph.  A sequence decoded
    alone gives the same bits as inside the batch, a second run reproduces the first, the cache behind the context stays untouched."""
    _need_gpu()
    T, n_dec = 2300, 8
    model, cfg, vocab, iv = _build_7b_on_device(2)
    if mode == "w8a16":
        model.enable_fp8_decode(True)
    model.eos_token_id = model.eot_token_id = -1
    g = torch.Generator(device=DEV).manual_seed(50 + B)
    x = (torch.randn((B, T, cfg["hidden_size"]), generator=g, device=DEV) * 0.5).to(torch.bfloat16)

    def run(sel):
        cache = model.new_cache(len(sel), T + n_dec + 8)
        cache.k.fill_(7.0)
        cache.v.fill_(7.0)
        hid, cache = model._forward_hidden(x[sel].contiguous(), cache)
        tok = model.text_token.expand(len(sel), -1, -1).clone()
        hyp, _, cache = model._greedy_device_loop(tok, cache, "text", n_dec, poll=10 ** 9)
        return hid, hyp[:, :, 0], cache

    hid, toks, cache = run(list(range(B)))
    assert bool(torch.isfinite(hid).all())
    ts, te = iv["text"][0]
    t = toks.cpu()
    assert bool((((t >= ts) & (t < te)) | (t == 2) | (t == 3)).all()), t
    used = cache.length
    assert used == T + n_dec
    assert bool((cache.k[:, :, :, used:] == 7.0).all()) and bool((cache.v[:, :, :, used:] == 7.0).all()), "KV cache written behind the context"
    assert bool(torch.isfinite(cache.k[:, :, :, :used].float()).all())
    hid2, toks2, _ = run(list(range(B)))
    assert torch.equal(hid, hid2) and torch.equal(toks, toks2), "not reproducible"
    for b in (0, B // 2, B - 1):
        h1, t1, _ = run([b])
        assert torch.equal(h1[0], hid[b]), f"sequence {b}: prefill differs between B=1 and B={B}"
        assert t1[0].tolist() == toks[b].tolist(), f"{mode} sequence {b}: B=1 ids {t1[0].tolist()} != batched {toks[b].tolist()}"


# ---------------------------------------------------------------------------------------------------------------------------
# fp8 encoder mode: the token-level contract
def _fp8_encoder_pipeline():
    """A UALM small enough for 10 clips x 32 steps whose encoder CAN run the e4m3 GEMMs (every K a multiple of 256):
    encoder d 512 / 8 heads x 64 / FFN 2048 / 4 layers over the tiny 12-layer LLM (adaptor 512 -> 768)."""
    from audio_intelligence_amd.multimodal_io.audio import ContinuousAudioIO
    from audio_intelligence_amd.multimodal_io.modeling_whisper import AFWhisperEncoder, AFWhisperEncoderConfig
    from audio_intelligence_amd.lm.parallel import ParallelLLM
    from audio_intelligence_amd import ualm_job
    from audio_intelligence_amd.utils import synthetic as syn
    ecfg = dict(oracle.afwhisper.tiny_config())
    ecfg.update(d_model=512, encoder_attention_heads=8, encoder_ffn_dim=2048, encoder_layers=4)
    lcfg = oracle.qwen2.config_tiny()
    enc = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(ecfg))
    enc.load_state_dict(syn.synth_state_dict(syn.encoder_param_shapes(ecfg), 31), strict=True)
    text_io, audio_io = H.stub_ios(lcfg["text_vocab"])
    cont = ContinuousAudioIO(encoder_choice="AFWhisper", dtype="bfloat16", device=DEV, encoder=enc)
    ios = {"text": text_io, "discrete_audio": audio_io, "continuous_audio": cont}
    vocab, iv = ualm_job.build_vocabulary(ios)
    hf = {"architectures": ["Qwen2ForCausalLM"], "hidden_size": lcfg["hidden_size"], "num_hidden_layers": lcfg["num_hidden_layers"],
          "num_attention_heads": lcfg["num_attention_heads"], "num_key_value_heads": lcfg["num_key_value_heads"],
          "intermediate_size": lcfg["intermediate_size"], "rope_theta": lcfg["rope_theta"], "rms_norm_eps": lcfg["rms_norm_eps"],
          "vocab_size": lcfg["text_vocab"]}
    model = ParallelLLM(hf, ios, vocab, iv)
    with torch.no_grad():
        params = dict(model.named_parameters())
        for name, shape in syn.llm_param_shapes(lcfg, len(vocab), 8, 512):
            params[name].copy_(syn.synth_tensor(name, shape, 32))
    model = model.to(DEV, torch.bfloat16)
    model.prepare_inference()
    model.eval()
    pre = ualm_job.UALMPreprocessor(False, {k: v.copy_for_worker() for k, v in ios.items()}, vocab, iv)
    return model, pre, lcfg, vocab, iv


def _teacher_forced_logits(model, kw, ids_forced, n_steps, allowed):
    emb = model._embed(torch.cat([kw["seqs"], model.assistant_token], dim=1), kw)
    hid, cache = model._forward_hidden(emb, None)
    tok = model.text_token.clone()
    out_logits, picks = [], []
    for st in range(n_steps):
        lg, cache = model._step(input_ids=tok, past_key_values=cache, mask=model.text_mask)
        l0 = lg[0, 0, 0].float()
        out_logits.append(l0[allowed])
        pick = int(l0.argmax())
        picks.append(pick)
        nxt = pick if ids_forced is None else ids_forced[st]
        tok = torch.zeros(1, 1, 8, dtype=torch.long, device=DEV)
        tok[0, 0, 0] = nxt
    return torch.stack(out_logits), picks


@pytest.mark.parametrize("mode", ["dynamic", "static_fc2", "static_fc2_attn"])
def test_fp8_encoder_token_level_contract(mode):
    """What e4m3 encoder activations do to TOKEN IDS.  10 clips x 32 greedy steps through encoder ->
    adaptor -> LLM in bf16 (run A, free-running), then the same with the encoder's projections on e4m3 operands (run B,
    teacher-forced with A's ids; the LLM stays bf16 so only the encoder differs).  Contract, stated before measuring B:
      * logit RMSE(B - A) over the allowed ids <= 0.10 x the standard deviation of A's logits (the decision variable moves by
        less than a tenth of its own spread);
      * B's argmax equals A's id on >= 90 % of the 320 steps;
      * every disagreement is a near-tie of A: A's top-2 gap there is <= 6 x that clip's logit RMSE (noise-consistent flips only);
      * the fp8 switch really changes the encoder output (the test is not vacuous).
    Measured (round 3): with q | k | v in e4m3 as well (AFHIP_FP8_MASK=7, round 2's default) the match was 284-289 of 320 -- on the
    wrong side of the 90 % line -- so the shipped mode keeps q | k | v in bf16 (mask 6: out-proj + fc1 in e4m3): 296 of 320, logit
    RMSE 0.038 x std, every flip within 2.2 x RMSE.  The budget was NOT moved; the mode was.
    mode "static_fc2": additionally fc2 on e4m3 operands, its input quantised in fc1's epilogue with a per-layer static scale
    (AFWhisperEncoder.calibrate_fp8 on a clip that is NOT one of the ten) -- the same contract, the same numbers.
    mode "static_fc2_attn": additionally the attention output leaves the attention kernel as e4m3 with a static scale (no quantisation
    pass in front of the out-projection)."""
    _need_gpu()
    model, pre, lcfg, vocab, iv = _fp8_encoder_pipeline()
    io = model.multimodal_io_dict["continuous_audio"]
    enc = io.model
    enc.calibrate_fp8(None)
    if mode != "dynamic":
        cal = pre.collate_fn([(("audio_to_caption", "x", "y"), {"audio": (fc.make_wav(999, 160000)[None], 16000), "text": [["user", "text", fc.make_prompt(lcfg["text_vocab"])]]})])
        enc.enable_fp8(True)
        scales = enc.calibrate_fp8(cal["continuous_audio_feats"].to(DEV, torch.bfloat16), attention_output=(mode == "static_fc2_attn"))
        assert scales is not None and bool((scales > 0).all())
        assert (enc._att_out_scale is not None) == (mode == "static_fc2_attn")
    text_mask = oracle.ualm.masks(len(vocab), iv)["text"]
    allowed = (~text_mask[0]).to(DEV)
    n_steps, total, match, worst_ratio = 32, 0, 0, 0.0
    se, var_sum, n_el = 0.0, 0.0, 0
    enc_rel = []
    for i in range(10):
        data = {"audio": (fc.make_wav(1000 + i, 160000)[None], 16000), "text": [["user", "text", fc.make_prompt(lcfg["text_vocab"])]]}
        b = pre.collate_fn([(("audio_to_caption", "x", "y"), data)])
        kw = {k: (v.to(DEV, torch.bfloat16) if v.is_floating_point() else v.to(DEV)) for k, v in b.items() if isinstance(v, torch.Tensor)}
        enc.enable_fp8(False)
        eA = io.encode_batch(kw["continuous_audio_feats"], kw["continuous_audio_lengths"])[0].float()
        lA, idsA = _teacher_forced_logits(model, kw, None, n_steps, allowed)
        enc.enable_fp8(True)
        eB = io.encode_batch(kw["continuous_audio_feats"], kw["continuous_audio_lengths"])[0].float()
        lB, idsB = _teacher_forced_logits(model, kw, idsA, n_steps, allowed)
        enc.enable_fp8(False)
        enc_rel.append(float((eB - eA).pow(2).mean().sqrt() / eA.pow(2).mean().sqrt()))
        d = lB - lA
        rmse_clip = float(d.pow(2).mean().sqrt())
        se += float(d.pow(2).sum())
        var_sum += float((lA - lA.mean(dim=1, keepdim=True)).pow(2).sum())
        n_el += d.numel()
        top2 = torch.topk(lA, 2, dim=1).values
        gaps = (top2[:, 0] - top2[:, 1]).cpu().tolist()
        for st in range(n_steps):
            total += 1
            if idsB[st] == idsA[st]:
                match += 1
            else:
                worst_ratio = max(worst_ratio, gaps[st] / max(rmse_clip, 1e-9))
    rmse, std = (se / n_el) ** 0.5, (var_sum / n_el) ** 0.5
    enc.calibrate_fp8(None)
    print(f"fp8 encoder ({mode}) vs bf16 encoder through the LLM: encoder output relative RMS diff {np.mean(enc_rel):.4f}; logit RMSE {rmse:.4f} "
          f"(logit std {std:.4f}, ratio {rmse / std:.4f}); token match {match}/{total}; largest bf16 gap among flips = {worst_ratio:.2f} x RMSE")
    assert min(enc_rel) > 1e-4, "fp8 switch did not change the encoder output"
    assert rmse <= 0.10 * std, (rmse, std)
    assert match >= 0.90 * total, (match, total)
    assert worst_ratio <= 6.0, worst_ratio


def test_fp8_encoder_on_the_7b_width_sample():
    """The sample oracle/make_golden_7b.py pushed through the reference (full-width one-layer encoder -> adaptor -> 7B-width
    2-layer LLM, T = 790): bf16 end to end against the same with the encoder in fp8 mode.  With these seeded weights attention is
    extremely peaked and the REFERENCE's own bf16 path moves the last-position logits by 0.45 on average (max 2.7) against its fp32
    run (golden_7b.json "bf16".last_logit_err) -- the yardstick every bf16 test of this shape uses.  Contract: what the e4m3 encoder
    projections add on top of bf16 stays BELOW that loss (mean and max, no slack factor), and the greedy ids agree with the bf16
    run wherever the reference's fp32 top-2 gap exceeds the reference's own worst bf16 regret."""
    _need_gpu()
    g = _gold7b()
    ref16 = g["bf16"]
    model, sd, cfg, vocab, iv = H.build_wide_llm(n_layers=2, dtype=torch.bfloat16, device=DEV, seed=g["seed_llm"], real_audio=True, enc_seed=g["seed_enc"])
    data = {"audio": (fc.make_wav(g["wav_seed"], 480000)[None], 16000), "text": [["user", "text", g["prompt"]]]}
    b = model._test_pre.collate_fn([(("audio_to_caption", "x", "y"), data)])
    kw = {k: (v.to(DEV, torch.bfloat16) if v.is_floating_point() else v.to(DEV)) for k, v in b.items() if isinstance(v, torch.Tensor)}
    ids = torch.cat([kw["seqs"], model.assistant_token], dim=1)
    enc = model.multimodal_io_dict["continuous_audio"].model
    model.eos_token_id = model.eot_token_id = -1
    ts, te = iv["text"][0]

    def run(fp8):
        enc.enable_fp8(fp8)
        emb = model._embed(ids, kw)
        hid, cache = model._forward_hidden(emb, model.new_cache(1, ids.shape[1] + g["n_dec"] + 8))
        last = model._head_stream0(hid[:, -1])[0].float()
        hyp, _, cache = model._greedy_device_loop(model.text_token.clone(), cache, "text", g["n_dec"], poll=10 ** 9)
        return emb[0].float(), last[ts:te], hyp[0, :, 0].cpu().tolist()

    eA, lA, idsA = run(False)
    eB, lB, idsB = run(True)
    enc.enable_fp8(False)
    d = (lB - lA).abs()
    emb_rel = float((eB - eA).pow(2).mean().sqrt() / eA.pow(2).mean().sqrt())
    print(f"7B-width sample, fp8 encoder vs bf16: spliced-embedding relative RMS diff {emb_rel:.4f}; last-position logit |diff| mean {float(d.mean()):.4f} max {float(d.max()):.4f} "
          f"(reference bf16 vs its fp32: {ref16['last_logit_err']}); ids bf16 {idsA} fp8 {idsB}; reference fp32 gaps {g['f32']['greedy_gaps']}")
    assert emb_rel > 1e-5, "fp8 switch did not change the embeddings"
    assert float(d.mean()) <= ref16["last_logit_err"]["mean"] and float(d.max()) <= ref16["last_logit_err"]["max"], (float(d.mean()), float(d.max()))
    worst_regret = max(ref16["regret_in_f32_logits"])
    for st, (a, b_) in enumerate(zip(idsA, idsB)):
        if a != b_:
            assert g["f32"]["greedy_gaps"][st] <= worst_regret, (st, a, b_, g["f32"]["greedy_gaps"][st], worst_regret)
            break                                                      # after the first flip the histories differ


def test_gemm_fp8_static_output_and_constant_row_scale():
    """The two GEMM forms behind the statically quantised fc2 input (afhip_gemm_args.out_fp8 / a_scale_const), at the encoder's fc1 / fc2
    shapes with M = 3000: (1) out_fp8: C as e4m3 bytes = sat(gelu(...) / s) -- dequantised, it is within one e4m3 step (2^-3 relative,
    2^-9 s absolute at the small end) of the bf16 output of the same GEMM, and values past 448 s saturate instead of becoming NaN;
    (2) a_scale_const = c gives the bits of a_scale filled with c."""
    _need_gpu()
    from audio_intelligence_amd import _lib as L
    from audio_intelligence_amd.utils.quant import quantize_rows_e4m3
    import ctypes as C
    g = torch.Generator().manual_seed(5)
    M, d, f = 3000, 1280, 5120
    x = torch.randn(M, d, generator=g).to(torch.bfloat16).to(DEV)
    w1 = (torch.randn(f, d, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    b1 = (torch.randn(f, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    x8, xs = quantize_rows_e4m3(x)
    w18, w1s = quantize_rows_e4m3(w1)

    def gemm8(A8, a_scale, W8, w_scale, bias, out, N, K, act, a_const=0.0, out_inv=0.0, res=None, stats=None):
        a = L.GemmArgs()
        a.A, a.W, a.C = A8.data_ptr(), W8.data_ptr(), out.data_ptr()
        a.bias = bias.data_ptr() if bias is not None else None
        a.residual = res.data_ptr() if res is not None else None
        a.M, a.N, a.K = A8.shape[0], N, K
        a.lda, a.ldw, a.ldc, a.ldres = K, K, N, N
        a.dtype, a.act = L.dtype_code(torch.bfloat16), act
        a.a_fp8 = 1
        a.a_scale = a_scale.data_ptr() if a_scale is not None else None
        a.w_scale = w_scale.data_ptr()
        a.a_scale_const, a.out_fp8, a.out_scale_inv = float(a_const), int(out_inv > 0), float(out_inv)
        a.row_stats_out = stats.data_ptr() if stats is not None else None
        L.check(L.lib().afhip_gemm(C.byref(a), L.stream_ptr()))

    ref = torch.empty(M, f, dtype=torch.bfloat16, device=DEV)
    gemm8(x8, xs, w18, w1s, b1, ref, f, d, L.ACT_GELU)
    amax = float(ref.float().abs().max())
    for s in (amax * 2.0 / 448.0, amax * 0.25 / 448.0):            # with headroom; and too small on purpose: the top saturates
        q = torch.zeros(M, f, dtype=torch.uint8, device=DEV)
        gemm8(x8, xs, w18, w1s, b1, q, f, d, L.ACT_GELU, out_inv=1.0 / s)
        torch.cuda.synchronize()
        deq = q.view(torch.float8_e4m3fn).float() * s
        assert bool(torch.isfinite(deq).all())
        want = ref.float().clamp(-448.0 * s, 448.0 * s)
        err = (deq - want).abs()
        tol = want.abs() * (2.0 ** -3) + s * (2.0 ** -9) + 1e-2 * want.abs()      # one e4m3 step + the bf16 rounding of `ref` itself
        assert bool((err <= tol).all()), (s, float((err - tol).max()))
        assert float(deq.abs().max()) <= 448.0 * s * (1 + 1e-6)
    # constant row scale == filled row-scale vector, bit for bit (fc2's shape, residual epilogue)
    s = amax * 2.0 / 448.0
    q = torch.zeros(M, f, dtype=torch.uint8, device=DEV)
    gemm8(x8, xs, w18, w1s, b1, q, f, d, L.ACT_GELU, out_inv=1.0 / s)
    w2 = (torch.randn(d, f, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    w28, w2s = quantize_rows_e4m3(w2)
    b2 = (torch.randn(d, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    res = torch.randn(M, d, generator=g).to(torch.bfloat16).to(DEV)
    o1 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    o2 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    gemm8(q, None, w28, w2s, b2, o1, d, f, L.ACT_NONE, a_const=s, res=res)
    gemm8(q, torch.full((M,), s, dtype=torch.float32, device=DEV), w28, w2s, b2, o2, d, f, L.ACT_NONE, res=res)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    # (3) the row-statistics epilogue of the e4m3 form (fc2 -> the next layer's LayerNorm-folded q | k | v): same output bits, and the
    # [N/64][M] partial (sum, sum of squares) add up to the statistics of the rows that were stored
    o3 = torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
    part = torch.zeros(d // 64, M, 2, dtype=torch.float32, device=DEV)
    gemm8(q, None, w28, w2s, b2, o3, d, f, L.ACT_NONE, a_const=s, res=res, stats=part)
    torch.cuda.synchronize()
    assert torch.equal(o3, o1)
    tot = part.sum(dim=0)
    assert torch.allclose(tot[:, 0], o1.float().sum(dim=1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(tot[:, 1], o1.float().pow(2).sum(dim=1), rtol=1e-4, atol=1e-2)
    # and against fp32 arithmetic on the dequantised operands
    want = (q.view(torch.float8_e4m3fn).float() * s) @ (w28.view(torch.float8_e4m3fn).float() * w2s[:, None]).t() + b2.float() + res.float()
    err = (o1.float() - want).abs()
    assert float(err.max()) <= 2e-2 + 1e-2 * float(want.abs().max()), float(err.max())
