"""Diagnostic build of the ping-pong GEMM with s_memtime stamps (never the product library): compiles gemm_pp.hip with
-DAFHIP_PP_STAMPS [-DPP_SCHED=n] into /tmp, links it with the other objects, loads THAT library and prints, for waves 0 and 4
(one of each wave group) of workgroup 0 on its second output tile: the LOAD / barrier-wait / MFMA / barrier-wait cycles of the
four phases of one K tile, and the epilogue (operand loads issued, math + stores issued).  s_memtime counts at 100 MHz on
gfx950, so a stamp difference is in units of 10 ns.  Read the SHARES: every stamp drains lgkmcnt.
usage: python tools/gemm_stamps.py [sched]"""
import glob, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
sched = sys.argv[1] if len(sys.argv) > 1 else "0"
obj, lib = f"/tmp/gemm_pp_stamps{sched}.o", f"/tmp/libafhip_ppstamps{sched}.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DAFHIP_PP_STAMPS", f"-DPP_SCHED={sched}",
                "-c", os.path.join(CSRC, "gemm_pp.hip"), "-o", obj], check=True)
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("gemm_pp.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
from audio_intelligence_amd import _lib as L
L.load_library(lib)
from audio_intelligence_amd import ops

dev, dt = "cuda:0", torch.bfloat16
M = 48000
cases = [("fc1 plain (bias)", 5120, 1280, L.ACT_NONE, False), ("fc1 + gelu", 5120, 1280, L.ACT_GELU, False), ("out + residual", 1280, 1280, L.ACT_NONE, True),
         ("fc2 + residual", 1280, 5120, L.ACT_NONE, True)]
for name, n, k, act, res in cases:
    a = torch.randn(M, k, device=dev, dtype=dt)
    w = torch.randn(n, k, device=dev, dtype=dt) * 0.03
    bias = torch.randn(n, device=dev, dtype=dt)
    r = torch.randn(M, n, device=dev, dtype=dt) if res else None
    out = torch.empty(M, n, device=dev, dtype=dt)
    os.environ.pop("AFHIP_PP_DBGPTR", None)
    for _ in range(3):
        ops.gemm(a, w, bias=bias, act=act, residual=r, out=out)
    buf = torch.zeros(128, dtype=torch.int64, device=dev)
    os.environ["AFHIP_PP_DBGPTR"] = hex(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(a, w, bias=bias, act=act, residual=r, out=out)
    e1.record()
    torch.cuda.synchronize()
    t = buf.cpu().reshape(2, 64).tolist()
    print(f"{name}: M={M} N={n} K={k}  launch {e0.elapsed_time(e1) * 1e3:.0f} us (stamped build), sched {sched}")
    for g in range(2):
        s = t[g]
        # stamps 0..31: per phase (before barrier A, after A, before barrier B, after B); the first stamp of a phase follows the previous phase's B
        row = []
        for ph in range(4):
            b = s[4 * ph: 4 * ph + 4]
            prev = s[4 * ph - 1] if ph else None
            load = (b[0] - prev) if prev else None
            row.append(f"ph{ph}: load {load if load is not None else '-':>4} | wait {b[1] - b[0]:>3} | mfma {b[2] - b[1]:>3} | wait {b[3] - b[2]:>3}")
        ktile = s[15] - s[0]
        print(f"  wave {'0' if g == 0 else '4'}  K tile (first stamp to last): {ktile} x10ns   " + "   ".join(row))
        print(f"          epilogue: loads issued +{s[41] - s[40]}, math+stores issued +{s[42] - s[41]}  (x10 ns); whole tile K loop = {k // 64} K tiles")
