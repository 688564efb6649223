"""C-ABI contract (CPU-only): the library loads, exports every symbol include/afhip.h declares, the ctypes
binding mirrors the header one to one, argument validation fails with an error code + message (never aborts),
and the host-side table builder matches numpy.  No compute entry point is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "afhip.h")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(afhip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from audio_intelligence_amd import _lib as L
    lib = L.load_library()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/afhip.h but not exported by libafhip.so"
    assert sorted(L.SIGNATURES.keys()) == names, "ctypes SIGNATURES and include/afhip.h disagree"
    assert lib.afhip_version() >= 100


def test_struct_layouts_match_header():
    """field order / count of the ctypes structures against the header's struct definitions"""
    from audio_intelligence_amd import _lib as L
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for cname, cls in (("afhip_gemm_args", L.GemmArgs), ("afhip_attn_args", L.AttnArgs), ("afhip_encoder_weights", L.EncoderWeights),
                       ("afhip_llm_weights", L.LlmWeights), ("afhip_kv_cache", L.KvCache), ("afhip_decode_state", L.DecodeState), ("afhip_sample_args", L.SampleArgs)):
        body = re.search(r"typedef struct \{([^{}]*)\}\s*" + cname + ";", src).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
        assert fields == [f[0] for f in cls._fields_], cname


def test_struct_offsets_and_sizes_match_the_c_compiler(tmp_path):
    """Types, offsets and sizes, not just names: a C program compiled against include/afhip.h prints offsetof / sizeof of every
    field of every struct that crosses the boundary; the ctypes mirror must agree byte for byte."""
    import shutil
    import subprocess
    from audio_intelligence_amd import _lib as L
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler on this host")
    structs = (("afhip_gemm_args", L.GemmArgs), ("afhip_attn_args", L.AttnArgs), ("afhip_encoder_weights", L.EncoderWeights),
               ("afhip_llm_weights", L.LlmWeights), ("afhip_kv_cache", L.KvCache), ("afhip_decode_state", L.DecodeState),
               ("afhip_sample_args", L.SampleArgs))
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in structs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu %zu\\n", offsetof({cname}, {fname}), sizeof((({cname}*)0)->{fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-std=c11", "-o", str(exe), str(src)], check=True)
    got = dict()
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        parts = ln.split()
        got[parts[0]] = tuple(int(x) for x in parts[1:])
    for cname, cls in structs:
        assert got[cname] == (C.sizeof(cls),), f"sizeof({cname}): C {got[cname]} vs ctypes {C.sizeof(cls)}"
        for fname, ftype in cls._fields_:
            fd = getattr(cls, fname)
            assert got[f"{cname}.{fname}"] == (fd.offset, fd.size), f"{cname}.{fname}: C (offset, size) {got[f'{cname}.{fname}']} vs ctypes {(fd.offset, fd.size)}"


def test_argument_validation_returns_error_not_abort():
    from audio_intelligence_amd import _lib as L
    lib = L.load_library()
    assert lib.afhip_gemm(None, None) == -1
    assert b"null args" in lib.afhip_last_error()
    g = L.GemmArgs()
    g.dtype, g.M, g.N, g.K = 1, 128, 128, 100      # K not a multiple of 64
    g.A = g.W = g.C = 16
    assert lib.afhip_gemm(C.byref(g), None) == -1 and b"multiple of 64" in lib.afhip_last_error()
    a = L.AttnArgs()
    a.dtype, a.hd, a.B, a.Tq, a.Tk, a.n_q, a.n_kv = 0, 96, 1, 8, 8, 2, 2
    a.q = a.k = a.v = a.out = 16
    assert lib.afhip_attention(C.byref(a), None) == -1 and b"head_dim 96" in lib.afhip_last_error()
    assert lib.afhip_log_mel(None, 1, 480000, 480000, None, 0, 0, None, None, None) == -1
    assert lib.afhip_layernorm(16, 16, 16, 16, 4, 12, 1e-5, 0, None) == -1      # D % 8 != 0
    with pytest.raises(L.AfhipError):
        L.check(-1)


def test_log_mel_constant_tables_host():
    from audio_intelligence_amd import _lib as L
    from audio_intelligence_amd.multimodal_io.feature_extraction import mel_filter_bank
    lib = L.load_library()
    n = lib.afhip_log_mel_tables_bytes() // 4
    host = np.zeros(n, dtype=np.float32)
    filt = np.ascontiguousarray(mel_filter_bank().astype(np.float32))
    assert lib.afhip_log_mel_tables_host(host.ctypes.data_as(C.c_void_p), filt.ctypes.data_as(C.c_void_p)) == 0
    KP, BP = 208, 224
    cos = host[: BP * KP].reshape(BP, KP)
    sin = host[BP * KP: 2 * BP * KP].reshape(BP, KP)
    win = host[2 * BP * KP: 2 * BP * KP + KP]
    k = np.arange(201)[:, None].astype(np.float64)
    nn = np.arange(1, 201)[None, :].astype(np.float64)
    np.testing.assert_allclose(cos[:201, :200], np.cos(2 * np.pi * k * nn / 400), atol=1e-7)
    np.testing.assert_allclose(sin[:201, :199], -np.sin(2 * np.pi * k * nn[:, :199] / 400), atol=1e-7)
    assert not cos[201:].any() and not cos[:, 200:].any() and not sin[:, 199:].any()
    np.testing.assert_allclose(win[:200], 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1, 201) / 400), atol=1e-7)
    off = 2 * BP * KP + KP
    np.testing.assert_array_equal(host[off: off + 201 * 128].reshape(201, 128), filt)
    b0 = off + 201 * 128
    band = host[b0: b0 + 256].view(np.int32).reshape(128, 2)
    coff = host[b0 + 256: b0 + 384].view(np.int32)
    cw = host[b0 + 384: b0 + 384 + 512]
    for m in range(128):
        nz = np.nonzero(filt[:, m])[0]
        assert band[m, 0] == nz[0] and band[m, 1] == nz[-1] - nz[0] + 1
        np.testing.assert_array_equal(cw[coff[m]: coff[m] + band[m, 1]], filt[band[m, 0]: band[m, 0] + band[m, 1], m])
    # (max, min) per clip and pass-1 workgroup (128 slots): the features themselves are written once, in place
    assert lib.afhip_log_mel_workspace_bytes(32) == 32 * 128 * 2 * 4


def test_no_kernel_spills_and_no_flat_memory_instruction():
    """tools/check_spills.py over every HIP source (device assembly only, no GPU needed): no kernel of the library spills vector registers,
    and none contains a FLAT memory instruction -- the library has no pointer that may be LDS or global, so a flat_load means an address
    space got lost on the way (round 4: the decode GEMMs streamed their weights through flat loads after an opaque scalar copy of their
    arguments, and the log-mel FFT read its LDS tables through them)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_spills.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "kernels checked" in r.stdout
