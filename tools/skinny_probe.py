#!/usr/bin/env python3
"""Per-kernel time of the 7B decode GEMMs through the C ABI (afhip_gemm_skinny), each call on a DIFFERENT weight copy so that no weight
byte is served by the 256-MiB Infinity Cache (a decode step streams 14.6 GB between two uses of a weight).
usage: python tools/skinny_probe.py [M ...]        env PROBE_FP8=1 adds the e4m3-weight forms, PROBE_ONLY=qkv,o,gateup,down,lm_head"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import _lib as L  # noqa: E402
from audio_intelligence_amd.utils.quant import quantize_rows_e4m3  # noqa: E402

H, I, QW, V = 3584, 18944, 4608, 160520
SHAPES = {           # name: (N, K, rmsnorm on A, bias, residual, swiglu epilogue, f32 out)
    "qkv": (QW, H, True, True, False, False, False),
    "o": (H, H, False, False, True, False, False),
    "gateup": (2 * I, H, True, False, False, True, False),
    "down": (H, I, False, False, True, False, False),
    "lm_head": (V, H, False, False, False, False, True),
}


def main():
    lib = L.lib()
    dev = torch.device("cuda", 0)
    Ms = [int(a) for a in sys.argv[1:]] or [8, 16]
    only = [s for s in os.environ.get("PROBE_ONLY", "").split(",") if s] or list(SHAPES)
    fp8s = (False, True) if os.environ.get("PROBE_FP8") else (False,)
    g = torch.Generator(device=dev).manual_seed(3)
    for name in only:
        N, K, rms, bias, res, swiglu, f32out = SHAPES[name]
        wbytes = N * K * 2
        copies = max(3, int(600e6 // wbytes) + 1)
        ws = [(torch.randn((N, K), generator=g, device=dev, dtype=torch.float32) * 0.03).to(torch.bfloat16) for _ in range(min(copies, 24))]
        for fp8 in fp8s:
            w8 = [quantize_rows_e4m3(w) for w in ws] if fp8 else None
            for M in Ms:
                a = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
                gain = torch.ones(K, device=dev, dtype=torch.bfloat16)
                b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
                n_out = N // 2 if swiglu else N
                r = torch.zeros((M, n_out), device=dev, dtype=torch.bfloat16)
                out = torch.empty((M, n_out), device=dev, dtype=torch.float32 if f32out else torch.bfloat16)
                args = []
                for i in range(len(ws)):
                    ga = L.GemmArgs()
                    ga.A, ga.C = a.data_ptr(), out.data_ptr()
                    if fp8:
                        ga.W, ga.w_scale = w8[i][0].data_ptr(), w8[i][1].data_ptr()
                    else:
                        ga.W = ws[i].data_ptr()
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
                    args.append(ga)
                reps = 4 * len(args)
                for i in range(len(args)):
                    L.check(lib.afhip_gemm_skinny(C.byref(args[i]), L.stream_ptr()))
                torch.cuda.synchronize()
                # one captured graph of `reps` launches: a ctypes launch costs the host as much as the short kernels take
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for i in range(reps):
                        L.check(lib.afhip_gemm_skinny(C.byref(args[i % len(args)]), L.stream_ptr()))
                graph.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0.record()
                    graph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / reps * 1e3)
                byt = N * K * (1 if fp8 else 2)
                print(f"{name:8s} M={M:2d} {'fp8 ' if fp8 else 'bf16'} N={N:6d} K={K:5d}: {best:7.2f} us  {byt / best / 1e6:5.2f} TB/s", flush=True)
        del ws


if __name__ == "__main__":
    main()
