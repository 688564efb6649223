"""world_size-2 rehearsal of the multi-GPU paths on CPU (gloo): the long-audio window sharding + single all-gather
 and bench.py's max-over-ranks timing reduction.  The encoder itself needs a GPU, so a
deterministic stand-in produces each window's tokens; what is checked is partitioning, ordering and the gather."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _encode(wav, n_valid):
    # [w, 750, 4] tokens that identify the window they came from
    w = wav.shape[0]
    base = wav[:, :4].reshape(w, 1, 4) + n_valid.reshape(w, 1, 1).float()
    return base.expand(w, 750, 4).contiguous() + torch.arange(750).reshape(1, 750, 1)


def _worker(rank, world, port, n_windows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audio_intelligence_amd.long_audio import encode_windows_sharded
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(n_windows, 64, generator=g)
    n_valid = torch.arange(1, n_windows + 1) * 1000
    out = encode_windows_sharded(_encode, wav, n_valid, out_spec=(750, 4, torch.float32))
    ref = _encode(wav, n_valid)
    ok = bool(torch.equal(out, ref))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py's elapsed-time reduction
    # config 4 end to end with stand-ins for the two GPU pieces (encoder, LLM): window sharding, ONE gather, per-window
    # (b, start, len) entries and clip ownership must reproduce the single-process result
    res = _long_audio(rank, world)
    q.put((rank, ok, float(t.item()), res))
    dist.barrier()
    dist.destroy_process_group()


class _FakeEnc:
    class config:
        max_source_positions = 1500

    def _get_feat_extract_output_lengths(self, n):
        a = (n - 1) // 2 + 1
        return a, (a - 2) // 2 + 1

    def encode_btc(self, mel, feat_len=None):
        w = mel.shape[0]
        return (mel[:, :4].reshape(w, 1, 4) + feat_len.reshape(w, 1, 1).float()).expand(w, 750, 4).contiguous() + torch.arange(750).reshape(1, 750, 1)


class _FakeProc:
    def extract_device(self, wav, layout="btc", dtype=None):
        return wav[:, :16].to(dtype)                       # a few samples identify the window


class _FakeIO:
    model, processor, hop_length, d_model, dtype = _FakeEnc(), _FakeProc(), 160, 4, torch.float32


class _FakeLLM:
    device, num_stream, vocab_intervals = torch.device("cpu"), 8, {"text": [(256, 1256)]}

    def inference_segment(self, cfg, cache=None, enforce_modality=None, **b):
        enc, idx = b["continuous_audio_encoded"], b["continuous_audio_indices"]
        assert enc.shape[0] == idx.shape[0]
        digest = [int(b["seqs"].sum()), int(b["seqs"].shape[1])] + [int(x) for x in idx.flatten()] + [int(enc[i, : int(idx[i, 2])].sum()) for i in range(enc.shape[0])]
        return [(torch.tensor(digest)[:, None], enforce_modality)], None


def _long_audio(rank, world):
    import numpy as np
    from audio_intelligence_amd.long_audio import long_audio_inference
    rng = np.random.default_rng(3)
    clips = [rng.standard_normal(n).astype(np.float32) for n in (75 * 16000, 30 * 16000, 100 * 16000 + 77)]     # 3 + 1 + 4 windows
    prompts = [[5, 6, 7], [0, 9], [11] * 6]
    got = long_audio_inference(_FakeLLM(), _FakeIO(), clips, prompts, {"text": {}}, enforce_modality="text")
    return {c: t[:, 0].tolist() for c, (t, _) in got.items()}


def _run(n_windows, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_windows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), "gathered tokens differ from the single-process result"
    assert all(r[2] == 2.0 for r in res)
    # long-audio ownership: clip c belongs to rank c % 2, and every clip's digest equals the single-process one
    import torch.distributed as d2
    assert not d2.is_initialized()
    single = _long_audio(0, 1)
    merged = {}
    for r in res:
        assert set(r[3].keys()) == {c for c in single if c % 2 == r[0]}
        merged.update(r[3])
    assert merged == single
    assert single[0][1] == 3 + 3 + 3 + 750 + 750 + 375 + 1              # bos,user,text + prompt + eot,user,audio + tokens of 30+30+15 s + eos


def test_window_shard_all_gather_even():
    _run(8, 29731)


def test_window_shard_all_gather_ragged_and_tiny():
    _run(5, 29732)      # 3 + 2 windows
    _run(1, 29733)      # one rank owns nothing


def _bench(argv, extra_env=None, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_started_as_one_process_launches_its_ranks():
    """`python bench.py --gpus N` with no RANK / WORLD_SIZE in the environment must start its N ranks itself (VERDICT round 2,
    weak #8): the launcher's self-test makes the ranks rendezvous over gloo, reduce a per-rank elapsed time with MAX and run
    the long-audio gather on rank-local window blocks (`total_windows`), all without a GPU; rank 0 prints the one JSON line."""
    for n in (2, 3):
        rc, out, err = _bench(["--gpus", str(n), "--selftest-rendezvous"])
        assert rc == 0, err[-2000:]
        assert len(out) == 1, out
        assert out[0] == {"selftest": "rendezvous", "n_gpus": n, "max_elapsed": float(n), "gather_in_window_order": True, "spawned_by_bench": True}


def test_bench_launcher_reports_a_failed_rank():
    """a rank that dies must end the whole job with a non-zero code (the others would wait at a barrier for ever)"""
    import bench
    import sys
    code = "import os,sys,time; r=int(os.environ['RANK']); sys.exit(7) if r==1 else time.sleep(60)"
    real = bench.os.path.abspath
    import subprocess
    import time as _t
    t0 = _t.time()
    # spawn_ranks starts `python <this file> argv`: point it at a tiny script instead of bench.py
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(code)
        path = f.name
    try:
        bench.os.path.abspath = lambda p: path if p == bench.__file__ else real(p)
        rc = bench.spawn_ranks(2, [], timeout=50)
    finally:
        bench.os.path.abspath = real
        os.unlink(path)
    assert rc == 7 and _t.time() - t0 < 30


def test_bench_launcher_takes_its_ranks_with_it_when_it_is_signalled_or_times_out():
    """a launcher that is SIGTERMed (a driver timeout) or whose own timeout fires must not leave rank processes behind --
    a rank blocked in an RCCL barrier would keep its GPU.  A tiny launcher process calls bench.spawn_ranks on ranks that write their PID
    and sleep; after SIGTERM to the launcher (and, separately, after the launcher's timeout) none of those PIDs is alive."""
    import signal
    import subprocess
    import sys
    import tempfile
    import time as _t
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        rank_py = os.path.join(tmp, "rank.py")
        with open(rank_py, "w") as f:
            f.write("import os,time\nopen(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n" % tmp)
        launcher = ("import sys, os\nsys.path.insert(0, %r)\nimport bench\nreal = bench.os.path.abspath\n"
                    "bench.os.path.abspath = lambda p: %r if p == bench.__file__ else real(p)\n"
                    "sys.exit(bench.spawn_ranks(2, [], timeout=float(sys.argv[1])))\n" % (root, rank_py))

        def pids():
            out = []
            for r in range(2):
                fn = os.path.join(tmp, f"pid{r}")
                if os.path.exists(fn) and open(fn).read().strip():
                    out.append(int(open(fn).read()))
            return out

        def alive(pid):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                return False
            # a zombie still answers kill(0): look at its state
            try:
                return open(f"/proc/{pid}/stat").read().split()[2] != "Z"
            except FileNotFoundError:
                return False

        for mode in ("sigterm", "timeout"):
            for r in range(2):
                fn = os.path.join(tmp, f"pid{r}")
                if os.path.exists(fn):
                    os.unlink(fn)
            p = subprocess.Popen([sys.executable, "-c", launcher, "600" if mode == "sigterm" else "2"])
            t0 = _t.time()
            while len(pids()) < 2 and _t.time() - t0 < 60:
                _t.sleep(0.05)
            ranks = pids()
            assert len(ranks) == 2 and all(alive(x) for x in ranks)
            if mode == "sigterm":
                p.send_signal(signal.SIGTERM)
            rc = p.wait(timeout=60)
            assert rc == (128 + signal.SIGTERM if mode == "sigterm" else 124), (mode, rc)
            t1 = _t.time()
            while any(alive(x) for x in ranks) and _t.time() - t1 < 10:
                _t.sleep(0.05)
            assert not any(alive(x) for x in ranks), f"{mode}: rank processes survived the launcher"


def test_gather_with_rank_local_blocks_matches_full_upload():
    """encode_windows_sharded(total_windows=W) -- every rank hands in only its own block -- equals the form where every rank
    holds all windows (world size 2, ragged: 5 windows)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_local, args=(r, 2, 29741, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and all(r[2] for r in res), res


def _worker_local(rank, world, port, n_windows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audio_intelligence_amd.long_audio import encode_windows_sharded, shard_range
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(n_windows, 64, generator=g)
    n_valid = torch.arange(1, n_windows + 1) * 1000
    lo, hi = shard_range(n_windows, rank, world)
    full = encode_windows_sharded(_encode, wav, n_valid, out_spec=(750, 4, torch.float32))
    local = encode_windows_sharded(_encode, wav[lo:hi], n_valid[lo:hi], out_spec=(750, 4, torch.float32), total_windows=n_windows)
    bad = False
    try:
        encode_windows_sharded(_encode, wav, n_valid, out_spec=(750, 4, torch.float32), total_windows=n_windows)   # not a local block
    except ValueError:
        bad = True
    q.put((rank, bool(torch.equal(full, local)) and bool(torch.equal(full, _encode(wav, n_valid))), bad))
    dist.barrier()
    dist.destroy_process_group()
