"""Whisper log-mel feature extractor backed by the HIP kernel (csrc/logmel.hip).

Drop-in for the way the reference uses `transformers.WhisperFeatureExtractor`
(UALM/models/ualm/multimodal_io/audio.py:985-997 construction, :1060-1064 call, :1001-1002 attributes):
`processor(wav, sampling_rate=16000, return_tensors="np")["input_features"]` -> float32 [1,128,3000].
Only the constants (mel filter bank, window, DFT tables) are computed on the host; the STFT / filter bank /
log / per-clip floor run on the GPU.  `extract_device` is the batched device-to-device entry the fused
pipeline uses (no PCIe hop).
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from .. import _lib as L

N_FFT, HOP, N_SAMPLES, N_FRAMES = 400, 160, 480000, 3000


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    out = 3.0 * f / 200.0
    hi = f >= 1000.0
    out[hi] = 15.0 + np.log(f[hi] / 1000.0) * (27.0 / np.log(6.4))
    return out


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    out = 200.0 * m / 3.0
    hi = m >= 15.0
    out[hi] = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m[hi] - 15.0))
    return out


def mel_filter_bank(n_bins: int = 201, n_mels: int = 128, fmin: float = 0.0, fmax: float = 8000.0, sr: int = 16000):
    """[n_bins, n_mels] float64: Slaney mel scale, triangular filters, Slaney area normalisation
    (transformers/audio_utils.py:638-729 with norm="slaney", mel_scale="slaney")."""
    pts = _mel_to_hz(np.linspace(_hz_to_mel(np.array([fmin]))[0], _hz_to_mel(np.array([fmax]))[0], n_mels + 2))
    fft = np.linspace(0, sr // 2, n_bins)
    diff = np.diff(pts)
    sl = pts[None, :] - fft[:, None]
    bank = np.maximum(0.0, np.minimum(-sl[:, :-2] / diff[:-1], sl[:, 2:] / diff[1:]))
    return bank * (2.0 / (pts[2:n_mels + 2] - pts[:n_mels]))[None, :]


class WhisperFeatureExtractorHIP:
    """Same constructor / call surface as the subset of WhisperFeatureExtractor the reference touches."""

    model_input_names = ["input_features"]

    def __init__(self, feature_size: int = 128, sampling_rate: int = 16000, hop_length: int = 160, chunk_length: int = 30,
                 n_fft: int = 400, padding_value: float = 0.0, **kwargs):
        if (feature_size, sampling_rate, hop_length, n_fft, chunk_length) != (128, 16000, 160, 400, 30):
            raise ValueError("the HIP log-mel kernel is specialised for AF-Whisper: 128 mels, 16 kHz, hop 160, n_fft 400, 30 s")
        self.feature_size, self.sampling_rate, self.hop_length, self.n_fft = feature_size, sampling_rate, hop_length, n_fft
        self.chunk_length = chunk_length
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.padding_value = padding_value
        self.mel_filters = mel_filter_bank(1 + n_fft // 2, feature_size, 0.0, 8000.0, sampling_rate)
        self._tables = {}     # device index -> tensor

    # ------------------------------------------------------------------ device side
    def _device_tables(self, device: torch.device) -> torch.Tensor:
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in self._tables:
            lib = L.lib()
            nbytes = lib.afhip_log_mel_tables_bytes()
            host = np.zeros(nbytes // 4, dtype=np.float32)
            filt = np.ascontiguousarray(self.mel_filters.astype(np.float32))
            L.check(lib.afhip_log_mel_tables_host(host.ctypes.data_as(C.c_void_p), filt.ctypes.data_as(C.c_void_p)))
            self._tables[key] = torch.from_numpy(host).to(device)
        return self._tables[key]

    def extract_device(self, wav: torch.Tensor, layout: str = "btc", dtype: torch.dtype = torch.float32,
                       workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        """wav [B, n<=480000] float32 on the GPU -> log-mel on the GPU.
        layout "bct": [B,128,3000] (extractor layout); "btc": [B,3000,128] (encode_batch layout)."""
        lib = L.lib()
        if not wav.is_cuda or wav.dtype != torch.float32 or wav.dim() != 2:
            raise L.AfhipError("extract_device expects a float32 CUDA/HIP tensor [B, n_samples]")
        if wav.shape[1] > N_SAMPLES:
            wav = wav[:, :N_SAMPLES]          # audio.py:1042-1044 truncation
        if wav.stride(1) != 1:
            wav = wav.contiguous()
        B, n = wav.shape
        out = torch.empty((B, 128, N_FRAMES) if layout == "bct" else (B, N_FRAMES, 128), dtype=dtype, device=wav.device)
        need = lib.afhip_log_mel_workspace_bytes(B)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need, dtype=torch.uint8, device=wav.device)
        row_stride = wav.stride(0) if B > 1 else n   # torch reports arbitrary strides for size-1 dims
        L.check(lib.afhip_log_mel(L.ptr(wav), B, n, row_stride, L.ptr(out), 0 if layout == "bct" else 1, L.dtype_code(dtype),
                                  L.ptr(self._device_tables(wav.device)), L.ptr(workspace), L.stream_ptr()))
        return out

    # ------------------------------------------------------------------ reference call surface
    def __call__(self, raw_speech, sampling_rate: Optional[int] = None, return_tensors: Optional[str] = "np", **kwargs):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"sampling_rate {sampling_rate} != {self.sampling_rate}")
        arr = np.asarray(raw_speech, dtype=np.float32)
        if arr.ndim == 1:
            arr = arr[None]
        if arr.shape[-1] > self.n_samples:
            arr = arr[..., : self.n_samples]
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if dev is None:
            L.lib()  # raises: no CPU fallback
        mel = self.extract_device(torch.from_numpy(np.ascontiguousarray(arr)).to(dev), layout="bct")
        if return_tensors == "pt":
            return {"input_features": mel.cpu()}
        return {"input_features": mel.cpu().numpy()}
