"""ContinuousAudioIO on MI355X: the audio-input plugin of UALM (log-mel + AF-Whisper encoder) over HIP.

Mirrors `ContinuousAudioIO` (UALM/models/ualm/multimodal_io/audio.py:873-1262): same constructor arguments,
`preprocess` / `encode_batch` / `find_length` / `copy_for_worker` / `feature_dim`, same attributes
(`model`, `processor`, `d_model`, `sample_rate`, `hop_length`, `n_samples`).  The key-padding mask of
encode_batch is handed to the kernels as the per-clip length vector the reference derives it from
(audio.py:1135-1142); the [B,1,1500,1500] tensor is never built.
"""
from typing import List, Tuple

import numpy as np
import torch

from .abs_io import AbsIO
from .feature_extraction import WhisperFeatureExtractorHIP
from .modeling_whisper import AFWhisperEncoder


class ContinuousAudioIO(AbsIO):
    def __init__(self, encoder_choice: str = "AFWhisper", encoder_hf_model_tag: str = "Qwen/Qwen2.5-Omni-7B",
                 encoder_local_path: str = None, attn_implementation: str = None, dtype: str = "bfloat16",
                 device: str = "cuda", _skip_loading: bool = False, encoder: AFWhisperEncoder = None):
        super().__init__(modality="audio", is_discrete=False)
        self.device = device
        self.encoder_choice = encoder_choice
        self.encoder_hf_model_tag = encoder_hf_model_tag
        self.encoder_local_path = encoder_local_path
        self.attn_implementation = attn_implementation   # accepted for signature parity; attention is always the fused kernel
        self.dtype_str = dtype
        self.dtype = getattr(torch, dtype)
        if _skip_loading:
            self.model = None
            self.processor = None
        else:
            self._init_encoder(encoder)

    def _init_encoder(self, encoder=None):
        if self.encoder_choice != "AFWhisper":
            # audio.py:929-959: the "huggingface" choice fetches Qwen/Qwen2.5-Omni-7B from the network
            raise NotImplementedError(f"Encoder choice {self.encoder_choice} not implemented (AF-Whisper by local path only)")
        if encoder is None:
            if self.encoder_local_path is None:
                raise ValueError("encoder_local_path must be provided for AFWhisper encoder choice")
            encoder = AFWhisperEncoder.from_pretrained(self.encoder_local_path, torch_dtype=self.dtype)
        self.model = encoder.to(device=self.device, dtype=self.dtype)
        self.model.eval()
        self.processor = WhisperFeatureExtractorHIP(feature_size=self.model.config.num_mel_bins, sampling_rate=16000,
                                                    hop_length=160, n_fft=400, padding_value=0.0)
        self.d_model = self.model.config.d_model
        self.sample_rate = self.processor.sampling_rate
        self.hop_length = self.processor.hop_length
        self.n_samples = self.model.config.max_source_positions * 2 * self.hop_length

    # ------------------------------------------------------------------ CPU-facing API (numpy in / numpy out)
    def _after_length(self, n_samples: int) -> int:
        before = n_samples // self.hop_length
        after = (before - 1) // 2 + 1
        return int((after - 2) // 2 + 1)

    def preprocess(self, data: Tuple[np.ndarray, int]):
        """audio.py:1013-1101: (wav, sr) -> (zeros[after,1] int32, (after, mel[3000,128] f32), zeros[after,1])."""
        wav, fs = data
        if fs != self.sample_rate:
            raise ValueError(f"sampling rate {fs} != {self.sample_rate}: resample first (the reference calls librosa.resample, audio.py:1034)")
        if len(wav.shape) > 1:
            wav = wav[0]
        if wav.shape[0] > self.n_samples:
            print(f"Warning: Input audio is too long to process, will truncate to {self.n_samples} samples")
            wav = wav[: self.n_samples]
        if not isinstance(wav, np.ndarray):
            wav = np.array(wav)
        original_samples = wav.shape[0]
        feat = self.processor(wav, sampling_rate=self.sample_rate, return_tensors="np")["input_features"][0].T
        after_length = self._after_length(original_samples)
        paddings = np.zeros((after_length, 1)).astype(np.int32)
        return paddings, (after_length, np.ascontiguousarray(feat)), paddings

    def find_length(self, data: Tuple[np.ndarray, int]) -> int:
        """audio.py:1189-1214."""
        wav, sr = data
        num_samples = int(wav.shape[-1] * self.sample_rate / sr) if sr != self.sample_rate else wav.shape[-1]
        return self._after_length(num_samples)

    # ------------------------------------------------------------------ GPU API
    @torch.no_grad()
    def encode_batch(self, batch_data: torch.Tensor, length: torch.Tensor) -> List[torch.Tensor]:
        """audio.py:1103-1187: mel [B,3000,128] + length[B] -> list of [out_len_i, d_model] tensors.

        `length` follows the caller's convention (SURVEY headline fact 5): the pipeline passes 3000 (no key is
        masked, all 750 rows returned); the self-test passes after_length (keys >= feat_len masked, rows trimmed)."""
        if self.model is None:
            raise RuntimeError("worker copy has no encoder (copy_for_worker)")
        input_mel_lengths = length.to(torch.long).cpu() * 4 - 1
        feat_len, out_len = self.model._get_feat_extract_output_lengths(input_mel_lengths)
        out = self.model.encode_btc(batch_data, feat_len=feat_len.clamp(max=self.model.config.max_source_positions))
        return [out[i, : int(out_len[i])] for i in range(out.shape[0])]

    @torch.no_grad()
    def encode_wav_batch(self, wav: torch.Tensor, length: torch.Tensor = None) -> List[torch.Tensor]:
        """Fused device path: raw 16 kHz clips [B, n<=480000] f32 on the GPU -> encoder outputs, no host hop.
        `length` as in encode_batch (default: pipeline convention, 3000)."""
        mel = self.processor.extract_device(wav, layout="btc", dtype=self.dtype)
        if length is None:
            length = torch.full((wav.shape[0],), 3000, dtype=torch.long)
        return self.encode_batch(mel, length)

    def copy_for_worker(self) -> "ContinuousAudioIO":
        """audio.py:1216-1254: model-less copy that keeps processor + metadata."""
        c = self.__class__(encoder_choice=self.encoder_choice, encoder_hf_model_tag=self.encoder_hf_model_tag,
                           encoder_local_path=self.encoder_local_path, attn_implementation=self.attn_implementation,
                           dtype=self.dtype_str, device="cpu", _skip_loading=True)
        c.processor = self.processor
        c.d_model, c.sample_rate, c.hop_length, c.n_samples = self.d_model, self.sample_rate, self.hop_length, self.n_samples
        return c

    def feature_dim(self) -> int:
        return self.d_model
