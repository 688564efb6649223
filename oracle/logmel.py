"""Whisper log-mel front end (oracle; test infrastructure only).

Restates third-party arithmetic the reference calls at audio.py:1056-1069:
  transformers 5.15.0 (reference pins >=4.57.1, UALM/requirement.txt:3)
  * models/whisper/feature_extraction_whisper.py:135-170  _torch_extract_fbank_features
  * audio_utils.py:448-482 (hertz_to_mel), 541-560 (triangular bank), 638-729 (mel_filter_bank)
Algorithm: zero-pad/truncate to 480000 samples -> STFT(n_fft=400, hop=160, periodic Hann,
center=True reflect pad 200) -> drop last frame -> |X|^2 -> filters^T @ P -> log10(clamp 1e-10)
-> max(x, clipmax-8) -> (x+4)/4.
"""

import numpy as np

N_FFT = 400
HOP = 160
N_MELS = 128
SR = 16000
N_SAMPLES = 480000
N_FRAMES = 3000
N_BINS = 201


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    log_region = f >= 1000.0
    out = mels.copy()
    out[log_region] = 15.0 + np.log(f[log_region] / 1000.0) * logstep
    return out


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    f = 200.0 * m / 3.0
    log_region = m >= 15.0
    out = f.copy()
    out[log_region] = 1000.0 * np.exp(logstep * (m[log_region] - 15.0))
    return out


def mel_filter_bank(n_bins: int = N_BINS, n_mels: int = N_MELS, fmin: float = 0.0,
                    fmax: float = 8000.0, sr: int = SR) -> np.ndarray:
    """[n_bins, n_mels] float64, Slaney scale + Slaney area norm (audio_utils.py:638-729)."""
    mel_pts = np.linspace(_hz_to_mel_slaney(np.array([fmin]))[0],
                          _hz_to_mel_slaney(np.array([fmax]))[0], n_mels + 2)
    filt_hz = _mel_to_hz_slaney(mel_pts)
    fft_hz = np.linspace(0, sr // 2, n_bins)
    diff = np.diff(filt_hz)
    slopes = filt_hz[None, :] - fft_hz[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    bank = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filt_hz[2:n_mels + 2] - filt_hz[:n_mels])
    return bank * enorm[None, :]


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """torch.hann_window(n) (periodic) in float64."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def pad_or_trim(wav: np.ndarray, n: int = N_SAMPLES) -> np.ndarray:
    """audio.py:1042-1044,1056-1057."""
    wav = np.asarray(wav)
    if wav.shape[-1] > n:
        wav = wav[..., :n]
    if wav.shape[-1] < n:
        pad = [(0, 0)] * (wav.ndim - 1) + [(0, n - wav.shape[-1])]
        wav = np.pad(wav, pad)
    return wav


def log_mel(wav: np.ndarray, precision: str = "f32") -> np.ndarray:
    """wav [N<=480000] or [B,N] float32 -> [B,128,3000] float32 (per-clip max floor).

    precision="f32": torch.stft in float32, the primitive the reference itself calls
    (feature_extraction_whisper.py:150), so this leg follows its rounding.
    precision="f64": explicit framing + float64 rFFT; the mathematically tighter value used
    to judge kernel accuracy independent of f32 FFT rounding.
    """
    w = np.asarray(wav, dtype=np.float32)
    squeeze = w.ndim == 1
    if squeeze:
        w = w[None]
    w = pad_or_trim(w)
    filt = mel_filter_bank()
    if precision == "f32":
        import torch
        t = torch.from_numpy(np.ascontiguousarray(w))
        win = torch.hann_window(N_FFT)
        st = torch.stft(t, N_FFT, HOP, window=win, return_complex=True)
        mag = (st[..., :-1].abs() ** 2).contiguous()
        mel = torch.from_numpy(filt.astype(np.float32)).T @ mag
        ls = torch.clamp(mel, min=1e-10).log10()
        mx = ls.amax(dim=(1, 2), keepdim=True)
        ls = torch.maximum(ls, mx - 8.0)
        ls = (ls + 4.0) / 4.0
        out = ls.numpy()
    elif precision == "f64":
        x = w.astype(np.float64)
        xp = np.pad(x, ((0, 0), (N_FFT // 2, N_FFT // 2)), mode="reflect")
        idx = np.arange(N_FRAMES)[:, None] * HOP + np.arange(N_FFT)[None, :]
        win = hann_periodic()
        out = np.empty((w.shape[0], N_MELS, N_FRAMES), dtype=np.float32)
        for b in range(w.shape[0]):
            fr = xp[b][idx] * win[None, :]
            spec = np.fft.rfft(fr, axis=1)
            p = spec.real ** 2 + spec.imag ** 2            # [3000,201]
            mel = filt.T @ p.T                              # [128,3000]
            ls = np.log10(np.maximum(mel, 1e-10))
            ls = np.maximum(ls, ls.max() - 8.0)
            out[b] = ((ls + 4.0) / 4.0).astype(np.float32)
    else:
        raise ValueError(precision)
    return out[0] if squeeze else out


def preprocess(wav: np.ndarray, sr: int = SR):
    """ContinuousAudioIO.preprocess (audio.py:1013-1101), AFWhisper branch, sr==16000 only.

    Returns (paddings[after,1] int32, (after, feat[3000,128] f32), paddings).
    """
    from .lengths import after_length
    if sr != SR:
        raise ValueError("oracle covers 16 kHz input only (reference resamples with librosa, audio.py:1034)")
    w = np.asarray(wav)
    if w.ndim > 1:
        w = w[0]
    if w.shape[0] > N_SAMPLES:
        w = w[:N_SAMPLES]
    n_orig = w.shape[0]
    feat = log_mel(w.astype(np.float32)).T.copy()
    after = after_length(n_orig)
    pads = np.zeros((after, 1), dtype=np.int32)
    return pads, (after, feat), pads
