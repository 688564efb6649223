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
    out = encode_windows_sharded(_encode, wav, n_valid)
    ref = _encode(wav, n_valid)
    ok = bool(torch.equal(out, ref))
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # bench.py's elapsed-time reduction
    q.put((rank, ok, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


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


def test_window_shard_all_gather_even():
    _run(8, 29731)


def test_window_shard_all_gather_ragged_and_tiny():
    _run(5, 29732)      # 3 + 2 windows
    _run(1, 29733)      # one rank owns nothing
