"""What do the HIP events that bench.py records around every GEMM / attention launch of the timed steps (the contract's live roofline
measurement) cost?  The headline step with and without them, same process.  Round 4, one MI355X: 67.7-68.1 ms per step without,
68.6-68.8 ms with the 324 events of a step (0.7-0.9 ms, 1.0-1.3 %): `value` is quoted WITH them.  usage: python tools/event_cost_probe.py"""
import os, sys, time, ctypes as C, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from audio_intelligence_amd import _lib as L
from audio_intelligence_amd.multimodal_io import feature_extraction as fe_mod
lib = L.lib()
dev = torch.device("cuda", 0)
enc = bench.build_encoder(dev, torch.bfloat16, None)
g = torch.Generator(device=dev).manual_seed(2000)
wav = torch.randn((32, 480000), generator=g, device=dev) * 0.1

from audio_intelligence_amd.multimodal_io.feature_extraction import WhisperFeatureExtractorHIP as FE
fe = FE()
ws = torch.empty(lib.afhip_log_mel_workspace_bytes(32), dtype=torch.uint8, device=dev)
def step():
    mel = fe.extract_device(wav, layout="btc", dtype=torch.bfloat16, workspace=ws)
    return enc.encode_btc(mel)
for _ in range(3): step()
torch.cuda.synchronize()
def timed(n, prof):
    if prof: L.check(lib.afhip_prof_enable(n * (5 * 32 + 2) + 8))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    if prof:
        a, b, c = C.c_int(), C.c_double(), C.c_double()
        for tag in (L.BF16, L.BF16 | 0x100, 0x400): L.check(lib.afhip_prof_collect(tag, C.byref(a), C.byref(b), C.byref(c)))
    return dt
for r in range(3):
    print(f"no events {timed(10, False):.3f} ms/step   events around 162 launches per step {timed(10, True):.3f} ms/step", flush=True)
