#!/usr/bin/env python3
"""Compiles every HIP source of the library to assembly (device side only) and lists kernels whose metadata reports spilled vector registers
(exit code 1) or a private segment without spills (a local array the kernel indexes at run time, or the emergency slot of spilled SCALAR
registers: reported, not an error).  Also fails on any FLAT memory instruction: the library has no pointer that may be LDS or global, so a
flat_load / flat_store means an address space got lost (an opaque `asm volatile("" : "+s"(ptr))` copy does that) -- round 4 found the decode
GEMMs streaming their weights through flat loads (lgkmcnt waits of the LDS combine then drained the weight window) and the log-mel FFT reading
its LDS tables through them.  usage: python tools/check_spills.py [file.hip ...]"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
extra = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "logmel.hip": ["-fno-slp-vectorize"],
         "attention_enc.hip": ["-Wno-inline-asm", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-fno-slp-vectorize"]}
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
bad = 0
for f in files:
    base = os.path.basename(f)
    out = f"/tmp/spills_{base}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S", f, "-o", out] + extra.get(base, []),
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    n = 0
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
        n += 1
        name, scratch, vg, sp = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
        if sp:
            bad += 1
            print(f"{base}: {name}: {sp} SPILLED VGPRs, {scratch} B scratch ({vg} VGPRs)")
        elif scratch:
            print(f"{base}: {name}: no spilled VGPRs; {scratch} B private segment ({vg} VGPRs)")
    flat = len(re.findall(r"^\s+flat_(?:load|store|atomic)", txt, flags=re.M))
    if flat:
        bad += 1
        print(f"{base}: {flat} FLAT memory instructions (an address space was lost)")
    print(f"{base}: {n} kernels checked")
sys.exit(1 if bad else 0)
