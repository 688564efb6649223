"""AF3 / VILA `SoundTower` window-stack wrapper over the HIP encoder (long-audio entry).

Mirrors UALM/models/ualm/multimodal_io/sound_encoder.py:35-132 and afwhisper_audio_encoder.py:31-35: a stack of
30-s windows [1,W,1,128,3000] + sample mask [1,W,1,3000] -> [W,750,d], untrimmed.  Windows are independent, which
is what `encode_windows_sharded` exploits to shard a long clip over the GPUs of a node (one all-gather)."""
import torch
import torch.nn as nn

from .modeling_whisper import AFWhisperEncoder


class SoundTower(nn.Module):
    def __init__(self, sound_tower, args=None, delay_load=False):
        super().__init__()
        self.is_loaded = False
        self.sound_tower_name = sound_tower
        self.cfg_only = None

    def _get_feat_extract_output_lengths(self, input_lengths):
        input_lengths = (input_lengths - 1) // 2 + 1
        output_lengths = (input_lengths - 2) // 2 + 1
        return input_lengths, output_lengths

    @torch.no_grad()
    def forward(self, sounds, mask=None):
        if type(sounds) is list:
            # sound_encoder.py:54-80: every element is a window batch [w,128,3000]; the mask is shared
            feats = []
            for sound in sounds:
                fl, _ = self._get_feat_extract_output_lengths(mask.sum(-1).reshape(-1))
                feats.append(self.sound_tower(sound, feat_len=fl).last_hidden_state.to(sound.dtype))
            return feats
        if sounds.dim() == 5:
            sounds = sounds.squeeze(0).squeeze(1)
            mask = mask.squeeze(0)
        fl, _ = self._get_feat_extract_output_lengths(mask.sum(-1).reshape(-1))
        return self.sound_tower(sounds, feat_len=fl).last_hidden_state.to(sounds.dtype)

    @property
    def dtype(self):
        return self.sound_tower.dtype

    @property
    def device(self):
        return self.sound_tower.device

    @property
    def config(self):
        return self.sound_tower.config if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.d_model


class AFWhisperSoundTower(SoundTower):
    def __init__(self, model_name_or_path: str, config=None, encoder: AFWhisperEncoder = None):
        super().__init__(model_name_or_path, config)
        self.sound_tower = encoder if encoder is not None else AFWhisperEncoder.from_pretrained(model_name_or_path)
        self.is_loaded = True
