"""Diagnostic build of the log-mel kernel with s_memtime stamps (never the product library): compiles logmel.hip with
-DAFHIP_LOGMEL_STAMPS into /tmp, links it with the other objects, loads THAT library and prints where workgroup (5, 0) spends the
ticks of its first frame tile: staging, load + window, 25-point DFTs + twiddles, the cross-lane DFT-8, the barrier, unpack, mel tail."""
import glob, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
obj, lib = "/tmp/logmel_stamps.o", "/tmp/libafhip_melstamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-DAFHIP_LOGMEL_STAMPS",
                "-c", os.path.join(CSRC, "logmel.hip"), "-o", obj], check=True)
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("logmel.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
from audio_intelligence_amd import _lib as L
L.load_library(lib)
from audio_intelligence_amd.multimodal_io.feature_extraction import WhisperFeatureExtractorHIP
fe = WhisperFeatureExtractorHIP()
wav = torch.randn(32, 480000, device="cuda") * 0.1
for _ in range(3): fe.extract_device(wav, layout="btc", dtype=torch.bfloat16)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
os.environ["AFHIP_LOGMEL_DBGPTR"] = hex(buf.data_ptr())
fe.extract_device(wav, layout="btc", dtype=torch.bfloat16)
torch.cuda.synchronize()
t = buf.cpu().tolist()
names = ["start", "tables + first tile staged, sync", "loaded + windowed", "dft25 + twiddles", "cross-lane DFT-8", "sync", "unpack + sync", "mel tail"]
prev = t[0]
for k in range(1, 8):
    print(f"{names[k]:36s} +{t[k] - prev:6d}   (at {t[k] - t[0]})")
    prev = t[k]
