"""Diagnostic build of the persistent decode GEMM (gemm_stream.hip) with 100-MHz wall-clock stamps (s_memrealtime) of wave 0 of every
workgroup -- never the product library.  Per shape: when the workgroups start (skew of the launch), when the weight window has been
issued, when the activation image is staged, when the first K step has been consumed (first weights have arrived), when the last K
step has been consumed and when the workgroup ends; all relative to the first workgroup's start, in microseconds.
usage: python tools/stream_stamps.py [M]"""
import ctypes as C
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DAFHIP_STREAM_STAMPS"]
objs = []
for name in ("gemm_stream", "gemm_skinny"):
    obj = f"/tmp/{name}_stamps.o"
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(CSRC, name + ".hip"), "-o", obj], check=True)
    objs.append(obj)
lib = "/tmp/libafhip_streamstamps.so"
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if os.path.basename(o) not in ("gemm_stream.o", "gemm_skinny.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + others, check=True)
from audio_intelligence_amd import _lib as L  # noqa: E402
L.load_library(lib)
lb = L.lib()
dev = torch.device("cuda", 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, I, QW, V = 3584, 18944, 4608, 160520
SHAPES = {"qkv": (QW, H, True, True, False, False, False), "o": (H, H, False, False, True, False, False),
          "gateup": (2 * I, H, True, False, False, True, False), "down": (H, I, False, False, True, False, False),
          "lm_head": (V, H, True, False, False, False, True)}
g = torch.Generator(device=dev).manual_seed(3)
buf = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
os.environ["AFHIP_STREAM_DBGPTR"] = hex(buf.data_ptr())
for name, (N, K, rms, bias, res, swiglu, f32out) in SHAPES.items():
    copies = max(3, int(600e6 // (N * K * 2)) + 1)
    ws = [(torch.randn((N, K), generator=g, device=dev, dtype=torch.float32) * 0.03).to(torch.bfloat16) for _ in range(min(copies, 8))]
    a = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    gain = torch.ones(K, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    n_out = N // 2 if swiglu else N
    r = torch.zeros((M, n_out), device=dev, dtype=torch.bfloat16)
    out = torch.empty((M, n_out), device=dev, dtype=torch.float32 if f32out else torch.bfloat16)

    def call(i):
        ga = L.GemmArgs()
        ga.A, ga.C, ga.W = a.data_ptr(), out.data_ptr(), ws[i % len(ws)].data_ptr()
        ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldc = M, N, K, K, K, n_out
        ga.dtype, ga.out_f32 = L.BF16, 1 if f32out else 0
        if rms:
            ga.a_norm_w, ga.a_norm_eps = gain.data_ptr(), 1e-6
        if bias:
            ga.bias = b.data_ptr()
        if res:
            ga.residual, ga.ldres = r.data_ptr(), n_out
        if swiglu:
            ga.act = L.ACT_SWIGLU
        L.check(lb.afhip_gemm_skinny(C.byref(ga), L.stream_ptr()))

    # a graph of 12 back-to-back launches (as in the decode step); the stamps of the LAST one survive
    for i in range(4):
        call(i)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(12):
            call(i)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); graph.replay(); e1.record()
    torch.cuda.synchronize()
    t = buf.cpu().reshape(256, 8).double()
    t = t[t[:, 0] > 0]
    base = float(t[:, 0].min())
    us = (t - base) / 100.0
    names = ["start", "window issued", "image staged", "first step consumed", "last step consumed", "end", "staging loads landed (wave 0)", "image rows written (wave 0)"]
    print(f"{name}: M={M} N={N} K={K}: {e0.elapsed_time(e1) / 12 * 1e3:.2f} us per launch (stamped build), {t.shape[0]} workgroups")
    for k in (0, 1, 6, 7, 2, 3, 4, 5):
        nm = names[k]
        col = us[:, k].sort().values
        print(f"    {nm:20s} min {float(col[0]):6.2f}  median {float(col[len(col) // 2]):6.2f}  p90 {float(col[int(len(col) * 0.9)]):6.2f}  max {float(col[-1]):6.2f}")
    del ws
