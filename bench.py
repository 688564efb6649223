#!/usr/bin/env python3
"""Benchmark of the AF3 / UALM audio-understanding hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Started as ONE process with --gpus N > 1 (no RANK / WORLD_SIZE in the environment) it launches its N ranks itself: N fresh
child processes of this same script, one per GPU, started BEFORE anything touches the GPU (plain fork + exec of a process that
has made no HIP call; never a re-exec of an initialised one), rendezvous on 127.0.0.1.

Headline workload (BASELINE.json configs[1]): AF-Whisper encoder only -- log-mel + 32-layer encoder, batch = 32 clips
of 30 s @ 16 kHz per GPU, bf16 storage / f32 accumulate, synthetic audio and seeded random weights of the true shape.
One step = one pass of the hot path over one batch whose waveforms are already resident in HBM.  Clips are
independent (the reference shards them `samples[rank::world_size]` with no communication, dataloader/dataset.py:80),
so N GPUs = N replicas of the same per-GPU batch: weak scaling, no data-path collective.  For N > 1 the line additionally
carries a `long_audio` object: BASELINE configs[3]'s 80 windows sharded over the ranks with the ONE RCCL all-gather of audio
tokens the path has, the collective timed with HIP events against the xGMI figure.

Rank 0 prints ONE JSON line: the bench contract + `roofline` (dominant kernel = the bf16 MFMA GEMM, timed live with
HIP events inside the timed region), `cpu_baseline` (the CPU oracle timed on this host, N=1 only), plus the log-mel
kernel's HBM figure and an AF3-7B-shape greedy-decode leg (decode tokens/s, the second half of BASELINE's metric).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="30-s clips per GPU per step")
    ap.add_argument("--no-decode", action="store_true", help="skip the AF3-7B decode leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the fp8 / mixed-length encoder legs (profiles: only the headline kernels run)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the hipBLASLt comparator GEMMs")
    ap.add_argument("--no-long-audio", action="store_true", help="N > 1: skip the long-audio (RCCL all-gather) leg")
    ap.add_argument("--cpu-clips", type=int, default=8)
    ap.add_argument("--workload", choices=["encoder", "long_audio", "decode"], default="encoder",
                    help="encoder = BASELINE configs[1] (the headline); long_audio = configs[3]: 4 x 10-min clips, windows sharded over the ranks, "
                         "one RCCL all-gather; decode = ONLY the AF3-7B greedy decode loop (configs[2], B = 8, bf16), for kernel profiles")
    ap.add_argument("--decode-steps", type=int, default=128)
    ap.add_argument("--decode-batch", type=int, default=8)
    ap.add_argument("--decode-fp8", action="store_true", help="--workload decode: W8A16 weights instead of bf16")
    ap.add_argument("--selftest-rendezvous", action="store_true",
                    help="launcher self-test (CPU, gloo): ranks rendezvous, barrier, reduce a fake elapsed time, gather one small tensor; no GPU work")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pdeathsig():
    """preexec hook of a rank: SIGTERM when the launcher dies (Linux PR_SET_PDEATHSIG), so a launcher killed with SIGKILL -- which no
    handler can see -- still takes its ranks with it; a rank blocked in an RCCL barrier would otherwise keep its GPU"""
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)       # PR_SET_PDEATHSIG = 1
    except Exception:       # noqa: BLE001 -- best effort; the handlers below cover SIGTERM / SIGINT
        pass


def spawn_ranks(n, argv, extra_env=None, timeout=3600.0):
    """Start `n` fresh processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, as torch.distributed.run sets
    them), wait for all of them, return the worst exit code.  The caller has not touched the GPU: children are ordinary
    fork + exec of an uninitialised parent.  If a rank dies the others are ended by PID (they would wait at a barrier for ever).
    Like torch.distributed.run, the launcher never leaves ranks behind: SIGTERM / SIGINT to the launcher, an exception, or the
    timeout (exit code 124) terminate, then kill, the exact PIDs it started; each rank also asks the kernel for SIGTERM should the
    launcher die without running any of that."""
    port = _free_port()
    procs = []

    def reap(grace=5.0):
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.time() + grace
        for q in procs:
            try:
                q.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
        for q in procs:
            try:
                q.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pass

    class _Stop(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Stop(signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "AFHIP_BENCH_SPAWNED": "1"})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, preexec_fn=_pdeathsig))
        t0 = time.time()
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0:
                    rc = rc or (code if code > 0 else 128 - code)
                    for q in alive:                                      # exact PIDs we started
                        q.terminate()
            if alive and timeout is not None and time.time() - t0 > timeout:
                rc = rc or 124
                break
            time.sleep(0.05)
    except _Stop as e:
        rc = 128 + int(e.args[0])
    finally:
        reap()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


def _selftest_rendezvous(args):
    """What every multi-rank run does around its timed region, with no GPU: gloo rendezvous from the launcher's environment, barrier,
    MAX-reduce of a per-rank elapsed time, one all_gather_into_tensor in window order."""
    import torch  # noqa: E402
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    from audio_intelligence_amd.long_audio import encode_windows_sharded, shard_range
    W = 2 * world + 1
    lo, hi = shard_range(W, rank, world)
    blk = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1)
    out = encode_windows_sharded(lambda w, n: w.reshape(-1, 1, 1).expand(-1, 3, 2).contiguous(), blk, torch.ones(hi - lo), out_spec=(3, 2, torch.float32), total_windows=W)
    ok = bool(torch.equal(out[:, 0, 0], torch.arange(W, dtype=torch.float32)))
    if rank == 0:
        print(json.dumps({"selftest": "rendezvous", "n_gpus": world, "max_elapsed": float(t.item()), "gather_in_window_order": ok,
                          "spawned_by_bench": os.environ.get("AFHIP_BENCH_SPAWNED") == "1"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok and float(t.item()) == float(world) else 1


if __name__ == "__main__":
    _args = parse_args()
    if "WORLD_SIZE" not in os.environ and _args.gpus > 1:
        sys.exit(spawn_ranks(_args.gpus, sys.argv[1:]))              # before `import torch` / the HIP library: nothing GPU-side exists yet
    if _args.selftest_rendezvous:
        sys.exit(_selftest_rendezvous(_args))

import ctypes as C  # noqa: E402

import torch

from audio_intelligence_amd import _lib as L  # noqa: E402
from audio_intelligence_amd.utils import synthetic as syn  # noqa: E402
from audio_intelligence_amd.multimodal_io.modeling_whisper import AFWhisperEncoder, AFWhisperEncoderConfig  # noqa: E402
from audio_intelligence_amd.multimodal_io.feature_extraction import WhisperFeatureExtractorHIP  # noqa: E402

ENC_CFG = dict(num_mel_bins=128, d_model=1280, encoder_attention_heads=20, encoder_ffn_dim=5120, encoder_layers=32,
               max_source_positions=1500)
LLM_7B = dict(architectures=["Qwen2ForCausalLM"], hidden_size=3584, num_hidden_layers=28, num_attention_heads=28,
              num_key_value_heads=4, intermediate_size=18944, rope_theta=1e6, rms_norm_eps=1e-6, vocab_size=152064)
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0       # HBM3E spec (same table)
MEL_BYTES_PER_CLIP_SURVEY = 480000 * 4 + 128 * 3000 * 4      # read wav + write an f32 mel
MEL_BYTES_PER_CLIP = 480000 * 4 + 128 * 3000 * 2             # what the timed launch moves: f32 wav in, bf16 mel out (the encoder's dtype)


class BoardSampler:
    """Shader clock and board power of THIS process's GPU while the timed steps run, read from the card's own hwmon files (the card is
    found by the PCI address torch reports for the device; a box shows every card of its host).  The 2.5 PFLOP/s peak is priced at 2.4 GHz:
    a kernel that runs at the board's power cap is clocked lower by the firmware, and `peak_at_clock` is the matrix peak at the clock it
    actually got (profiles/r03_clock_power_probe.txt).  Returns None where the files are not readable."""

    def __init__(self, device_index):
        import glob
        self.freq = self.power = self.cap = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cards = [c for c in sorted(glob.glob("/sys/class/drm/card*")) if os.path.basename(os.path.realpath(c + "/device")) == bdf]
            if cards:
                hw = sorted(glob.glob(cards[0] + "/device/hwmon/hwmon*"))
                if hw:
                    self.freq = hw[0] + "/freq1_input"
                    self.power = hw[0] + "/power1_input" if os.path.exists(hw[0] + "/power1_input") else hw[0] + "/power1_average"
                    self.cap = hw[0] + "/power1_cap"
        except Exception:  # noqa: BLE001 - diagnostics only
            pass
        self.samples, self._stop, self._th = [], False, None

    def _run(self):
        while not self._stop:
            try:
                self.samples.append((int(open(self.freq).read()) / 1e6, int(open(self.power).read()) / 1e6))
            except (OSError, ValueError):
                return
            time.sleep(0.01)

    def start(self):
        if self.freq and self.power:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        if self._th is None:
            return None
        self._stop = True
        self._th.join()
        sm = self.samples[len(self.samples) // 5:]          # drop the ramp at the start of the timed region
        if len(sm) < 3:
            return None
        mhz = sorted(x[0] for x in sm)[len(sm) // 2]
        w = sorted(x[1] for x in sm)[len(sm) // 2]
        try:
            cap = int(open(self.cap).read()) / 1e6
        except (OSError, ValueError):
            cap = None
        return {"shader_clock_mhz_median": mhz, "board_power_w_median": w, "board_power_cap_w": cap, "samples": len(sm),
                "what": "hwmon freq1_input / power1_input of this GPU sampled every 10 ms over the timed steps (all kernels of the step, not the GEMM alone)"}


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (newest
    profiles/rNN_hbm_traffic_pmc.json; unit and gfx950 corrections applied as MI355X_MICROARCH.md prescribes).  PMC passes serialise
    every dispatch and need the profiler as the parent process, so they are collected offline on the same build and only READ
    here; (None, None, None) when no file is present."""
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for name, key in (("r04_hbm_traffic_pmc.json", "kernels_r04"), ("r03_hbm_traffic_pmc.json", "kernels_r03"), ("r02_hbm_traffic_pmc.json", "kernels_r02"), ("r01_hbm_traffic_pmc.json", "kernels_r01_e")):
        try:
            with open(os.path.join(base, name)) as f:
                k = json.load(f)[key]
            pp = [v for kn, v in k.items() if kn.startswith("gemm_pp_kernel")]
            n = sum(v["launches_sampled"] for v in pp)
            gemm = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in pp) / n     # launch-weighted mean over qkv / out / fc1 / fc2
            mel = sum(v["hbm_bytes_per_launch"] for kn, v in k.items() if kn.startswith("logmel"))
            return gemm, mel, name
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None, None


def enc_flops_per_clip(c):
    d, f, Lr, T = c["d_model"], c["encoder_ffn_dim"], c["encoder_layers"], c["max_source_positions"]
    stem = 2 * 3000 * d * 3 * c["num_mel_bins"] + 2 * T * d * 3 * d
    return float(stem + Lr * (8 * T * d * d + 4 * T * T * d + 4 * T * d * f))


def build_encoder(device, dtype, cpu_state=None):
    enc = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(ENC_CFG))
    if cpu_state is not None:
        enc.load_state_dict(cpu_state, strict=True)
        return enc.to(device, dtype)
    enc = enc.to(device, dtype)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            p.copy_(syn.synth_tensor(n, p.shape, 1, dtype=dtype, device=device))
    return enc


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_threads():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (the GPU box grants a share of the host)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cpu_state, n_clips, wav_cpu):
    """The CPU oracle (build-owned PyTorch-CPU fp32 restatement, pinned to the reference by tests/golden) on a bounded
    sample of the same workload: every core this process may use, one untimed warm-up clip, then `n_clips` clips."""
    import oracle
    threads = cpu_threads()
    torch.set_num_threads(threads)
    warm = oracle.logmel.log_mel(wav_cpu[:1].numpy())
    oracle.afwhisper.encoder_forward(torch.from_numpy(warm), cpu_state, ENC_CFG)
    t0 = time.perf_counter()
    mel = oracle.logmel.log_mel(wav_cpu[:n_clips].numpy())
    t_mel = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = oracle.afwhisper.encoder_forward(torch.from_numpy(mel), cpu_state, ENC_CFG)
    t_enc = time.perf_counter() - t0
    return {"value": n_clips * 30.0 / (t_mel + t_enc), "unit": "audio-s/s", "cores": threads, "cpu": cpu_model_name(), "host_cores_visible": len(os.sched_getaffinity(0)), "kind": "port",
            "sample": f"{n_clips} of the 32 clips after a 1-clip warm-up (log-mel {t_mel:.2f} s + encoder {t_enc:.2f} s, fp32, SDPA attention, {threads} threads)"}, out


def library_ceiling(device, B):
    """Measured ceiling BASELINE.md 3 asks for: the vendor library's plain GEMM (torch.matmul -> hipBLASLt, no epilogue) on this device,
    on the four encoder projection shapes with random data, launch-weighted like the bench's own figure.  A comparator only: the
    product never calls it."""
    M = 1500 * B
    shapes = [("qkv", 3840, 1280), ("out", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120)]
    tot_ms, tot_fl, per = 0.0, 0.0, {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, n, k in shapes:
        a = torch.randn(M, k, device=device, dtype=torch.bfloat16)
        w = (torch.randn(n, k, device=device, dtype=torch.bfloat16) * 0.03).t()
        out = torch.empty(M, n, device=device, dtype=torch.bfloat16)
        for _ in range(3):
            torch.matmul(a, w, out=out)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(8):
            torch.matmul(a, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        per[name] = 2.0 * M * n * k / ms / 1e9
        tot_ms += ms
        tot_fl += 2.0 * M * n * k
    return {"what": "hipBLASLt plain bf16 GEMM via torch.matmul, same device, random data, no epilogue; launch-weighted over qkv/out/fc1/fc2",
            "tflops": tot_fl / tot_ms / 1e9, "per_shape_tflops": per}


def build_llm_7b(device, enc):
    """AF3-7B shape (Qwen2.5-7B backbone, 8 streams, V = 160 520) with seeded random bf16 weights, over the given encoder."""
    from audio_intelligence_amd.lm.parallel import ParallelLLM
    from audio_intelligence_amd.multimodal_io.audio import ContinuousAudioIO
    from audio_intelligence_amd.multimodal_io.abs_io import AbsIO
    from audio_intelligence_amd import ualm_job

    class _Text(AbsIO):
        def __init__(self):
            super().__init__(modality="text", is_discrete=True)

        def num_stream(self):
            return 1

        def get_vocabulary(self):
            return [f"<text_{i}>" for i in range(LLM_7B["vocab_size"])]

        def get_stream_interval(self):
            return [(0, LLM_7B["vocab_size"])]

    class _Audio(AbsIO):
        def __init__(self):
            super().__init__(modality="audio", is_discrete=True)

        def num_stream(self):
            return 8

        def get_vocabulary(self):
            return [f"<audio_{i}>" for i in range(8 * 1025)]

        def get_stream_interval(self):
            return [(s * 1025, (s + 1) * 1025) for s in range(8)]

    cont = ContinuousAudioIO(encoder_choice="AFWhisper", dtype="bfloat16", device=str(device), encoder=enc)
    ios = {"text": _Text(), "discrete_audio": _Audio(), "continuous_audio": cont}
    vocab, iv = ualm_job.build_vocabulary(ios)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(device):
            model = ParallelLLM(LLM_7B, ios, vocab, iv)
    finally:
        torch.set_default_dtype(old)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if not n.startswith("multimodal_io_dict."):
                p.copy_(syn.synth_tensor(n, p.shape, 2, dtype=torch.bfloat16, device=device))
    model.prepare_inference()
    model.eos_token_id = model.eot_token_id = -1           # random weights: never stop early
    return model, len(vocab)


def decode_bytes_per_step(head_rows, B, ctx, wbytes):
    """layer weights + the lm_head rows the step streams (text decode: rows below 256 + text vocab, the 8 x 1025
    audio-code rows behind them are never allowed) + the K/V of the context, once per step."""
    H, I, nl = LLM_7B["hidden_size"], LLM_7B["intermediate_size"], LLM_7B["num_hidden_layers"]
    kvw = LLM_7B["num_key_value_heads"] * (H // LLM_7B["num_attention_heads"])
    w_elems = nl * (H * H * 2 + 2 * H * kvw + 3 * H * I) + head_rows * H
    return w_elems * wbytes + B * ctx * nl * 2 * kvw * 2


def decode_traffic(label):
    """(HBM bytes per decode step, file) from the committed rocprofv3 --pmc passes over the decode step of the newest round that has
    them (profiles/rNN_decode_traffic_pmc.json: FETCH_SIZE x 2 + WRITE_SIZE, separate passes), or (None, None)."""
    for name in ("r04_decode_traffic_pmc.json", "r02_decode_traffic_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)["per_step_bytes"][label], name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def decode_leg(device, model, n_vocab, fe, B, n_steps, warm, config_name):
    """B x 30-s clips + 32 prompt ids: mel -> encoder -> adaptor/splice -> prefill, then greedy decode with the device-resident
    loop (one captured hipGraph of a step, replayed per token).  Returns tokens/s and the HBM figure of the decode step, bf16
    weights and W8A16 (e4m3 weights)."""
    g = torch.Generator(device=device).manual_seed(99)
    wav = torch.randn((B, 480000), generator=g, device=device) * 0.1
    mel = fe.extract_device(wav, layout="btc", dtype=torch.bfloat16)
    prompt = syn.make_prompt(LLM_7B["vocab_size"], 32)
    S = 8
    rows = [[1], [5], [7]] + [[256 + t] for t in prompt] + [[3], [5], [8]] + [[0]] * 750 + [[2]]
    seq = torch.zeros((len(rows), S), dtype=torch.int64)
    seq[:, 0] = torch.tensor([r[0] for r in rows])
    start = 3 + len(prompt) + 3
    batch = {"seqs": seq[None].repeat(B, 1, 1).to(device), "continuous_audio_feats": mel,
             "continuous_audio_lengths": torch.full((B,), 3000, dtype=torch.long),
             "continuous_audio_indices": torch.tensor([[b, start, 750] for b in range(B)])}
    ids = torch.cat([batch["seqs"], model.assistant_token.expand(B, -1, -1)], dim=1)
    T = ids.shape[1]
    model.enable_fp8_decode(False)
    model.pack(T + 4 * (n_steps + warm) + 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = model._embed(ids, batch)
    hid, cache = model._forward_hidden(emb, model.new_cache(B, T + 4 * (n_steps + warm) + 16))
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t0
    tok = model.text_token.expand(B, -1, -1).clone()
    res = {"config": config_name, "model": "AF3-7B shape (Qwen2.5-7B backbone, 8 streams, V=%d), random bf16 weights" % n_vocab,
           "batch": B, "prompt_tokens": T, "decode_steps": n_steps, "prefill_s": t_prefill, "prefill_audio_s_per_s": B * 30.0 / t_prefill,
           "loop": "device-resident greedy loop, hipGraph replay per token" if os.environ.get("AFHIP_DECODE_GRAPH", "1") != "0" else "device-resident greedy loop, eager launches"}
    hyp = tok
    for label, wbytes, key in (("bf16", 2, None), ("e4m3 + per-row f32 scale (W8A16), bf16 activations / KV", 1, "fp8_weights")):
        model.enable_fp8_decode(key is not None)
        hyp, _, cache = model._greedy_device_loop(hyp[:, -1:, :], cache, "text", warm, poll=10 ** 9)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hyp, _, cache = model._greedy_device_loop(hyp[:, -1:, :], cache, "text", n_steps, poll=10 ** 9)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        model._allowed_intervals("text")
        head_rows = model._allowed_hi.get("text") or n_vocab
        step_bytes = decode_bytes_per_step(head_rows, B, cache.length - n_steps // 2, wbytes)
        gbs = step_bytes * n_steps / dt_s / 1e9
        traffic, traffic_file = decode_traffic(f"B{B}_{'fp8' if key else 'bf16'}")
        leg = {"weights": label, "tokens_per_s": B * n_steps / dt_s, "ms_per_step": dt_s / n_steps * 1e3,
               "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                            "traffic": traffic, "traffic_unit": f"HBM bytes per step, rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, profiles/{traffic_file}" if traffic_file else None,
                            "bytes_per_step": step_bytes, "lm_head_rows": head_rows}}
        if key is None:
            res.update(leg)
        else:
            res[key] = leg
    # fp8 prefill (e4m3 x e4m3 projections) of the same batch
    model.enable_fp8(True)
    model.pack(T + 64)
    emb = model._embed(ids, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hid8, c8 = model._forward_hidden(emb, model.new_cache(B, T + 16))
    torch.cuda.synchronize()
    res["fp8_weights"]["prefill_llm_only_s"] = time.perf_counter() - t0
    model.enable_fp8(False)
    model.pack(T + 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hid16, c16 = model._forward_hidden(emb, model.new_cache(B, T + 16))
    torch.cuda.synchronize()
    res["prefill_llm_only_s"] = time.perf_counter() - t0
    # (no accuracy figure here: 28 randomly initialised layers are chaotic under ANY perturbation -- the reference's own bf16 path
    #  loses 0.28 mean |err| after two of them, tests/golden/golden_7b.json; fp8 accuracy is budgeted in tests/test_gpu_fp8.py)
    del hid8, hid16, c8, c16, emb
    return res


def _rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:       # noqa: BLE001 -- a build without the binding: the record says so
        return None


def long_audio_workload(args, device, dist, rank, world, enc, fe, with_llm=True, steps=None, warmup=None):
    """BASELINE configs[3]: 4 x 10-minute clips = 80 windows of 30 s, sharded over the ranks in contiguous blocks; each rank
    generates / uploads ONLY its own block, runs log-mel + encoder on it, and ONE all_gather_into_tensor returns every window's
    [750, 1280] bf16 tokens to every rank (timed on its own with HIP events, against the xGMI figure of SURVEY 8d); then, with
    `with_llm`, clip c's owner rank (c % world) splices its 20 windows into one prompt, prefills ~15 000 positions at the AF3-7B
    shape and decodes 128 tokens."""
    from audio_intelligence_amd.long_audio import encode_windows_sharded, make_tower_encode_fn, build_long_prompt, shard_range
    from audio_intelligence_amd.multimodal_io.audio import ContinuousAudioIO
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    n_clips, win_per_clip = 4, 20
    W = n_clips * win_per_clip
    io = ContinuousAudioIO(encoder_choice="AFWhisper", dtype="bfloat16", device=str(device), encoder=enc)
    lo, hi = shard_range(W, rank, world)
    wins = torch.empty((hi - lo, 480000), dtype=torch.float32, device=device)
    for i, w in enumerate(range(lo, hi)):                     # window w has the same samples whatever the world size
        g = torch.Generator(device=device).manual_seed(3000 + w)
        wins[i] = torch.randn(480000, generator=g, device=device) * 0.1
    n_valid = torch.full((hi - lo,), 480000, dtype=torch.long, device=device)
    fn = make_tower_encode_fn(io)

    def step(timing=None):
        return encode_windows_sharded(fn, wins, n_valid, timing=timing, total_windows=W)

    for _ in range(warmup):
        tokens = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gathers = []
    t0 = time.perf_counter()
    for _ in range(steps):
        tm = {}
        tokens = step(tm)
        gathers.append(tm)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert tokens.shape[0] == W
    gather_ms = [tm["gather_ev"][0].elapsed_time(tm["gather_ev"][1]) for tm in gathers if "gather_ev" in tm]
    rehearsal = os.environ.get("AFHIP_BENCH_REHEARSAL") == "1"
    res = {"metric": "audio-seconds encoded/sec (long audio: log-mel + AF-Whisper encoder over window shards + token all-gather)",
           "value": steps * n_clips * 600.0 / elapsed, "unit": "audio-s/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
           "data": "synthetic",
           "config": {"workload": "AF3 long audio (BASELINE configs[3]): 4 x 10-min clips = 80 x 30-s windows, window-sharded over the ranks, one all-gather of [750,1280] bf16 tokens per window",
                      "windows": W, "windows_per_rank": (W + world - 1) // world, "windows_uploaded_by_this_rank": hi - lo,
                      "parallelism": f"window shards x{world} + 1 all_gather_into_tensor"}}
    if gather_ms:
        shard = gathers[0]["gather_bytes"]
        ms = sum(gather_ms) / len(gather_ms)
        sent = (world - 1) * shard
        res["collective"] = {"op": "all_gather_into_tensor (RCCL over xGMI)" if not rehearsal else "all_gather_into_tensor (gloo REHEARSAL on one GPU: not a measurement)",
                             "backend": dist.get_backend() if dist is not None else None, "ranks": world,
                             # what the process group itself reports, and the algorithm / protocol knobs RCCL ran under (unset = RCCL's
                             # own choice: the direct one-hop pattern vs ring question of SURVEY 5 is left to it and recorded here)
                             "ranks_seen_by_process_group": dist.get_world_size() if dist is not None else 1,
                             "rccl": {k: os.environ.get(k) for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS",
                                                                     "RCCL_MSCCL_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")},
                             "rccl_version": _rccl_version() if not rehearsal else None,
                             "shard_bytes": shard, "ms": ms, "ms_min": min(gather_ms), "bytes_sent_per_rank": sent,
                             "GBps_per_rank": sent / (ms * 1e-3) / 1e9, "xgmi_peak_GBps_per_rank": 7 * 153.0,
                             "frac_of_xgmi": sent / (ms * 1e-3) / 1e9 / (7 * 153.0),
                             "timed_with": "HIP events on the stream the collective is enqueued on, rank 0"}
    else:
        res["collective"] = None                     # one rank: nothing to gather
    if with_llm and not args.no_decode:
        model, n_vocab = build_llm_7b(device, enc)
        text_offset = model.vocab_intervals["text"][0][0]
        prompt = syn.make_prompt(LLM_7B["vocab_size"], 32)
        legs = []
        for c in range(n_clips):
            if c % world != rank:
                continue
            seq, entries = build_long_prompt(prompt, win_per_clip * 480000, text_offset, model.num_stream)
            batch = {"seqs": seq[None].to(device), "continuous_audio_indices": torch.tensor([[0, s0, n] for s0, n in entries]),
                     "continuous_audio_encoded": tokens[c * win_per_clip:(c + 1) * win_per_clip]}
            ids = torch.cat([batch["seqs"], model.assistant_token], dim=1)
            T = ids.shape[1]
            model.pack(T + args.decode_steps + 64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            emb = model._embed(ids, batch)
            hid, cache = model._forward_hidden(emb, model.new_cache(1, T + args.decode_steps + 16))
            torch.cuda.synchronize()
            t_pre = time.perf_counter() - t0
            t0 = time.perf_counter()
            hyp, _, cache = model._greedy_device_loop(model.text_token.clone(), cache, "text", args.decode_steps, poll=10 ** 9)
            torch.cuda.synchronize()
            t_dec = time.perf_counter() - t0
            H, nl = LLM_7B["hidden_size"], LLM_7B["num_hidden_layers"]
            lin = 2.0 * 6.526e9 * T
            att = 4.0 * T * T * H * nl / 2
            legs.append({"clip": c, "prompt_tokens": T, "prefill_s": t_pre, "prefill_tflops": (lin + att) / t_pre / 1e12,
                         "prefill_flop_linear": lin, "prefill_flop_attention_causal": att, "decode_tokens_per_s": args.decode_steps / t_dec,
                         "decode_ms_per_step": t_dec / args.decode_steps * 1e3})
        res["llm_per_owned_clip"] = legs
    return res


def decode_only_workload(args, device, enc, fe):
    """--workload decode: ONLY what a kernel profile of the decode step needs -- encoder + prefill of B clips once, `warmup` untimed
    steps, then `decode_steps` timed greedy steps (hipGraph replay per token), bf16 or W8A16 weights.  tools/check_profile.py
    sums the kernels whose call count is a multiple of (warmup + steps) in the rocprofv3 kernel trace of this command and holds
    the sum against `ms_per_step`."""
    model, n_vocab = build_llm_7b(device, enc)
    B, n_steps, warm = args.decode_batch, args.decode_steps, 8
    g = torch.Generator(device=device).manual_seed(99)
    wav = torch.randn((B, 480000), generator=g, device=device) * 0.1
    mel = fe.extract_device(wav, layout="btc", dtype=torch.bfloat16)
    prompt = syn.make_prompt(LLM_7B["vocab_size"], 32)
    rows = [1, 5, 7] + [256 + t for t in prompt] + [3, 5, 8] + [0] * 750 + [2]
    seq = torch.zeros((len(rows), 8), dtype=torch.int64)
    seq[:, 0] = torch.tensor(rows)
    start = 3 + len(prompt) + 3
    batch = {"seqs": seq[None].repeat(B, 1, 1).to(device), "continuous_audio_feats": mel,
             "continuous_audio_lengths": torch.full((B,), 3000, dtype=torch.long),
             "continuous_audio_indices": torch.tensor([[b, start, 750] for b in range(B)])}
    ids = torch.cat([batch["seqs"], model.assistant_token.expand(B, -1, -1)], dim=1)
    T = ids.shape[1]
    model.enable_fp8_decode(bool(args.decode_fp8))
    model.pack(T + n_steps + warm + 64)
    emb = model._embed(ids, batch)
    hid, cache = model._forward_hidden(emb, model.new_cache(B, T + n_steps + warm + 16))
    tok = model.text_token.expand(B, -1, -1).clone()
    hyp, _, cache = model._greedy_device_loop(tok, cache, "text", warm, poll=10 ** 9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hyp, _, cache = model._greedy_device_loop(hyp[:, -1:, :], cache, "text", n_steps, poll=10 ** 9)
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    wbytes = 1 if args.decode_fp8 else 2
    model._allowed_intervals("text")
    head_rows = model._allowed_hi.get("text") or n_vocab
    step_bytes = decode_bytes_per_step(head_rows, B, cache.length - n_steps // 2, wbytes)
    gbs = step_bytes * n_steps / dt_s / 1e9
    traffic, traffic_file = decode_traffic(f"B{B}_{'fp8' if args.decode_fp8 else 'bf16'}")
    return {"metric": "decode tokens/sec (AF3-7B shape greedy decode loop only)", "value": B * n_steps / dt_s, "unit": "tokens/s", "n_gpus": 1,
            "steps": n_steps, "warmup": warm, "ms_per_step": dt_s / n_steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.decode_fp8 else "bf16 activations, e4m3 weights", "data": "synthetic",
            "config": {"workload": f"AF3-7B greedy decode only, B={B}, ctx ~{cache.length - n_steps // 2}", "batch": B, "prompt_tokens": T},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "traffic": traffic, "traffic_unit": f"HBM bytes per step, rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, profiles/{traffic_file}" if traffic_file else None,
                         "bytes_per_step": step_bytes, "lm_head_rows": head_rows},
            "cpu_baseline": None}


def main():
    args = parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (plain `python bench.py --gpus N` does it itself)")
    # rehearsal switch (one-GPU box): AFHIP_BENCH_REHEARSAL=1 runs N ranks on ONE device over gloo, to exercise the multi-process
    # code path (launcher, rendezvous, barriers, max-over-ranks, the long-audio gather) where no second GPU exists.  Never a measurement.
    rehearsal = os.environ.get("AFHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)       # nccl == RCCL on ROCm
    lib = L.lib()

    B, dtype = args.batch, torch.bfloat16
    fe = WhisperFeatureExtractorHIP()
    if args.workload == "decode":
        if world != 1:
            raise SystemExit("--workload decode is a one-GPU profile target")
        print(json.dumps(decode_only_workload(args, device, build_encoder(device, dtype, None), fe)))
        return
    if args.workload == "long_audio":
        enc = build_encoder(device, dtype, None)
        res = long_audio_workload(args, device, dist, rank, world, enc, fe)
        if dist is not None:
            gathered = [None] * world
            dist.all_gather_object(gathered, res.get("llm_per_owned_clip"))
            res["llm_per_owned_clip"] = [x for g_ in gathered if g_ for x in g_]
        if rank == 0:
            res["cpu_baseline"] = None
            print(json.dumps(res))
        if dist is not None:
            dist.destroy_process_group()
        return
    do_cpu = (rank == 0 and world == 1 and not args.no_cpu)
    cpu_state = syn.synth_state_dict(syn.encoder_param_shapes(ENC_CFG), 1) if do_cpu else None
    enc = build_encoder(device, dtype, cpu_state)
    wav_cpu = None
    if do_cpu:
        wav_cpu = torch.stack([torch.from_numpy(syn.make_wav(2000 + i, 480000)) for i in range(B)])
        wav = wav_cpu.to(device)
    else:
        g = torch.Generator(device=device).manual_seed(2000 + rank)
        wav = torch.randn((B, 480000), generator=g, device=device) * 0.1

    mel_ws = torch.empty(lib.afhip_log_mel_workspace_bytes(B), dtype=torch.uint8, device=device)

    def step():
        mel = fe.extract_device(wav, layout="btc", dtype=dtype, workspace=mel_ws)
        return enc.encode_btc(mel)

    for _ in range(args.warmup):
        out = step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    L.check(lib.afhip_prof_enable(args.steps * (5 * ENC_CFG["encoder_layers"] + 2) + 8))       # 4 GEMMs + 1 attention per layer, 2 stem GEMMs
    board = BoardSampler(device.index if device.index is not None else 0)
    board.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    board_info = board.stop()
    fence()
    # the dominant kernel = gemm_pp_kernel (128 of the 130 GEMM launches of a forward, > 99 % of its FLOPs); the two
    # implicit-conv stem launches (gemm256_kernel) are collected separately
    n_l, ms, fl = C.c_int(), C.c_double(), C.c_double()
    n_o, ms_o, fl_o = C.c_int(), C.c_double(), C.c_double()
    L.check(lib.afhip_prof_collect(L.BF16, C.byref(n_o), C.byref(ms_o), C.byref(fl_o)))
    L.check(lib.afhip_prof_collect(L.BF16 | 0x100, C.byref(n_l), C.byref(ms), C.byref(fl)))
    # the encoder attention kernel AS IT IS PAID inside the timed steps (HIP events around each launch, afhip.h: AFHIP_PROF_ATTN)
    n_a, ms_a, fl_a = C.c_int(), C.c_double(), C.c_double()
    L.check(lib.afhip_prof_collect(0x400, C.byref(n_a), C.byref(ms_a), C.byref(fl_a)))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-stage figures on this rank (outside the timed region) ----
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    n_mel = 20
    ev0.record()
    for _ in range(n_mel):
        mel = fe.extract_device(wav, layout="btc", dtype=dtype, workspace=mel_ws)
    ev1.record()
    torch.cuda.synchronize()
    mel_ms = ev0.elapsed_time(ev1) / n_mel
    ev0.record()
    for _ in range(3):
        enc.encode_btc(mel)
    ev1.record()
    torch.cuda.synchronize()
    enc_ms = ev0.elapsed_time(ev1) / 3
    enc8, enc_mixed = None, None
    if not args.no_extra_legs:
        # BASELINE config 5 evidence (never the headline `value`): the same encoder batch with the four projections of every layer on
        # e4m3 operands (afhip_gemm a_fp8: block-scaled MFMA, 2x the bf16 rate; activations quantised per row, LayerNorm fused)
        ref_out = enc.encode_btc(mel)
        enc.enable_fp8(True)

        def fp8_leg():
            o = enc.encode_btc(mel)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(3):
                enc.encode_btc(mel)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 3
            dd = (o.float() - ref_out.float()).abs()
            return ms, float(dd.max()), float(dd.mean())

        dyn_ms, dyn_max, dyn_mean = fp8_leg()
        # fc2's input statically quantised in fc1's epilogue: calibrated on a DIFFERENT synthetic batch than the one timed
        gcal = torch.Generator(device=device).manual_seed(77)
        enc.calibrate_fp8((torch.randn((4, 3000, 128), generator=gcal, device=device) * 0.5).to(torch.bfloat16))
        enc8_ms, st_max, st_mean = fp8_leg()
        enc.calibrate_fp8(None)
        out8 = d8 = None
        enc8 = {"encoder_ms": enc8_ms, "encoder_audio_s_per_s": B * 30.0 / (enc8_ms * 1e-3), "speedup_vs_bf16": enc_ms / enc8_ms,
                "vs_bf16_output_max_abs": st_max, "vs_bf16_output_mean_abs": st_mean,
                "dynamic_scales_only": {"encoder_ms": dyn_ms, "speedup_vs_bf16": enc_ms / dyn_ms, "vs_bf16_output_max_abs": dyn_max, "vs_bf16_output_mean_abs": dyn_mean,
                                        "what": "out-proj + fc1 on e4m3 operands, every activation quantised per row by its own pass; fc2 bf16"},
                "what": "e4m3 x e4m3 MFMA GEMMs for out-proj, fc1 and fc2 (per-channel weight scales, f32 accumulate). fc1 input: per-row scales from the LayerNorm + quantise pass; "
                        "out-proj input: written as e4m3 by the encoder attention kernel, fc2 input: by fc1's GELU epilogue -- both with per-layer static scales from "
                        "AFWhisperEncoder.calibrate_fp8 (margin 2 x the calibration batch's max, saturating). q | k | v (LayerNorm-folded), the attention arithmetic and the "
                        "residual stream stay bf16: with q | k | v in e4m3 the token-level contract of tests/test_gpu_config5.py fails"}
        enc.enable_fp8(False)
        del ref_out, out8, d8
        # BASELINE config 2, second run: the same 32 clips with mixed 5-30 s lengths in the self-test convention
        # (length = after-conv length -> key-padding mask per clip, audio.py:1129-1161); audio-seconds = the real clip lengths
        gmix = torch.Generator().manual_seed(5)
        secs = torch.randint(5, 31, (B,), generator=gmix)
        secs[0] = 30
        feat_len = ((secs * 16000 // 160 - 1) // 2 + 1).to(torch.int32).to(device)        # positions after conv2 (stride 2), <= 1500
        enc.encode_btc(mel, feat_len=feat_len)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(3):
            enc.encode_btc(mel, feat_len=feat_len)
        ev1.record()
        torch.cuda.synchronize()
        mix_ms = ev0.elapsed_time(ev1) / 3
        # ... and on packed rows (afhip_encoder_forward_ragged): the layers run on M = sum of lengths rows; the kept rows are bit-identical
        enc.encode_btc(mel, feat_len=feat_len, ragged=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(3):
            enc.encode_btc(mel, feat_len=feat_len, ragged=True)
        ev1.record()
        torch.cuda.synchronize()
        rag_ms = ev0.elapsed_time(ev1) / 3
        enc_mixed = {"encoder_ms": rag_ms, "clip_seconds_total": int(secs.sum()), "encoder_audio_s_per_s": float(secs.sum()) / (rag_ms * 1e-3),
                     "padded_encoder_ms": mix_ms, "padded_encoder_audio_s_per_s": float(secs.sum()) / (mix_ms * 1e-3),
                     "what": "mixed 5-30 s clips, length = after-conv length (key-padding mask per clip). encoder_ms: layers on the packed valid positions "
                             "(M = sum of lengths, what ContinuousAudioIO.encode_batch runs for ragged batches); padded_*: all 1500 positions of every "
                             "clip as the reference computes them (dead key tiles skipped)"}

    res = None
    if rank == 0:
        audio_s = world * args.steps * B * 30.0
        gemm_tflops = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        gemm_traffic, mel_traffic, pmc_file = pmc_traffic()
        mel_gbs = MEL_BYTES_PER_CLIP * B / (mel_ms * 1e-3) / 1e9
        res = {
            "metric": "audio-seconds encoded/sec (log-mel + AF-Whisper encoder)", "value": audio_s / elapsed, "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "AF-Whisper encoder only (BASELINE configs[1]): log-mel + 32-layer d=1280 encoder, "
                                   "batch=32 x 30s@16kHz clips per GPU, wav resident in HBM",
                       "batch_per_gpu": B, "clip_seconds": 30, "parallelism": f"clip-level replicas x{world} (no data-path collective)",
                       "encoder_tflop_per_clip": enc_flops_per_clip(ENC_CFG) / 1e12},
            "roofline": {"bound": "mfma", "kernel": "gemm_pp_kernel (persistent ping-pong bf16 GEMM: the qkv / out / fc1 / fc2 projections, 128 launches per forward)",
                         "achieved": gemm_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tflops / PEAK_BF16_TFLOPS,
                         "traffic": gemm_traffic, "traffic_unit": f"L2-to-fabric bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/{pmc_file}; Infinity-Cache hits included; algorithmic A+W+C+residual bytes: 0.25-0.63 GB)",
                         "launches": n_l.value, "avg_launch_ms": ms.value / max(1, n_l.value),
                         "avg_launch_gflop": fl.value / max(1, n_l.value) / 1e9, "gemm_share_of_step": ms.value / (elapsed * 1e3) if world == 1 else None,
                         "other_gemm": {"kernel": "gemm256_kernel (implicit-conv stem)", "launches": n_o.value,
                                        "avg_launch_ms": ms_o.value / max(1, n_o.value),
                                        "tflops": fl_o.value / (ms_o.value * 1e-3) / 1e12 if ms_o.value > 0 else 0.0},
                         "board": board_info},
            "attn": {"kernel": "attn_enc64_kernel (encoder attention, one launch per layer), timed with HIP events around each launch INSIDE the timed steps",
                     "launches": n_a.value, "avg_launch_ms": ms_a.value / max(1, n_a.value),
                     "achieved": (fl_a.value / (ms_a.value * 1e-3) / 1e12) if ms_a.value > 0 else 0.0, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": (fl_a.value / (ms_a.value * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if ms_a.value > 0 else 0.0,
                     "share_of_step": ms_a.value / (elapsed * 1e3) if world == 1 else None},
            "stages": {"mel_ms": mel_ms, "mel_audio_s_per_s": B * 30.0 / (mel_ms * 1e-3),
                       "mel_roofline": {"bound": "hbm", "achieved": mel_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                        "frac": mel_gbs / PEAK_HBM_GBS, "traffic": mel_traffic, "bytes_per_clip": MEL_BYTES_PER_CLIP,
                                        "bytes": "wav f32 in + mel in the dtype the timed launch writes (bf16): 2 688 000 B per 30-s clip; "
                                                 "SURVEY 8(d)'s 3 456 000 B assumes an f32 mel",
                                        "traffic_over_bytes": (mel_traffic / (MEL_BYTES_PER_CLIP * B)) if mel_traffic else None},
                       "encoder_fp8": enc8,
                       "encoder_mixed_lengths": enc_mixed,
                       "encoder_ms": enc_ms, "encoder_audio_s_per_s": B * 30.0 / (enc_ms * 1e-3),
                       "encoder_tflops": enc_flops_per_clip(ENC_CFG) * B / (enc_ms * 1e-3) / 1e12},
        }
    if rank == 0 and res["roofline"].get("board"):
        bi = res["roofline"]["board"]
        bi["peak_at_clock_tflops"] = PEAK_BF16_TFLOPS * bi["shader_clock_mhz_median"] / 2400.0
        bi["frac_at_clock"] = res["roofline"]["achieved"] / bi["peak_at_clock_tflops"]
    if do_cpu:
        cb, ref_out = cpu_baseline(cpu_state, args.cpu_clips, wav_cpu)
        err = (out[: args.cpu_clips].float().cpu() - ref_out).abs()
        cb["gpu_bf16_vs_cpu_fp32_max_abs_err"] = float(err.max())
        cb["gpu_bf16_vs_cpu_fp32_mean_abs_err"] = float(err.mean())
        res["cpu_baseline"] = cb
    elif rank == 0:
        res["cpu_baseline"] = None
    if rank == 0 and not args.no_ceiling:
        res["roofline"]["measured_ceiling"] = library_ceiling(device, B)
        res["roofline"]["frac_of_measured_ceiling"] = res["roofline"]["achieved"] / res["roofline"]["measured_ceiling"]["tflops"]
    if world > 1 and not args.no_long_audio:
        # the one collective the path has: BASELINE configs[3]'s 80 windows sharded over these ranks, ONE RCCL
        # all-gather of audio tokens, timed with HIP events -- so a scaling run records that RCCL saw N ranks and what it cost
        la = long_audio_workload(args, device, dist, rank, world, enc, fe, with_llm=False, steps=max(3, min(args.steps, 10)), warmup=2)
        if rank == 0:
            res["long_audio"] = {k: la[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "scaling", "config", "collective")}
    if not args.no_decode:
        del out
        model, n_vocab = build_llm_7b(device, enc)
        d = decode_leg(device, model, n_vocab, fe, 8, args.decode_steps, 8, "BASELINE configs[2]: AF3-7B full pipeline, batch=8 x 30 s, greedy")
        d16 = decode_leg(device, model, n_vocab, fe, 16, args.decode_steps, 8, "BASELINE configs[4]: UALM 7B generate(), bf16 + fp8, batch=16 x 30 s")
        del model
        if dist is not None:
            for leg in (d, d16):
                t = torch.tensor([leg["tokens_per_s"]], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)        # replicas: whole-job tokens/s
                leg["tokens_per_s_all_gpus"] = float(t.item())
        if rank == 0:
            res["decode"] = d
            res["decode_b16"] = d16
            # the second half of BASELINE's metric at roofline level: the decode step is HBM-bound (weight + KV streaming)
            res["roofline"]["decode"] = dict(d["roofline"], kernel="one greedy decode step, AF3-7B shape, B=8, bf16 weights (skinny GEMMs + split-context attention + lm_head)",
                                             tokens_per_s=d["tokens_per_s"], ms_per_step=d["ms_per_step"])
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
