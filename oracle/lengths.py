"""Length arithmetic of the audio path (oracle; test infrastructure only).

Follows (reference, relative to /root/reference/UALM/models/ualm/multimodal_io):
  * audio.py:1073,1094-1095   before = n_samples // hop ; after = ((before-1)//2+1 - 2)//2 + 1
  * audio.py:1189-1214        find_length (same formula, with sample-rate scaling)
  * audio.py:1135-1142        encode_batch: L_mel = 4*length - 1 -> (feat_len, out_len)
  * modeling_whisper.py:759-765  _get_feat_extract_output_lengths
"""

HOP = 160
N_SAMPLES = 480000
N_FRAMES = 3000
MAX_SRC = 1500


def after_length(n_samples: int, hop: int = HOP) -> int:
    """audio.py:1073,1094-1095 (n_samples is the *untruncated-to-30s-capped* sample count)."""
    before = n_samples // hop
    a = (before - 1) // 2 + 1
    a = (a - 2) // 2 + 1
    return int(a)


def find_length(num_samples: int, sr: int = 16000, target_sr: int = 16000, hop: int = HOP) -> int:
    """audio.py:1189-1214. No 30 s cap here (the reference does not apply one)."""
    if sr != target_sr:
        num_samples = int(num_samples * target_sr / sr)
    return after_length(num_samples, hop)


def feat_extract_output_lengths(input_length: int):
    """modeling_whisper.py:759-765 -> (feat_len after conv2, out_len after avg-pool)."""
    feat = (input_length - 1) // 2 + 1
    out = (feat - 2) // 2 + 1
    return int(feat), int(out)


def encode_batch_lengths(length: int):
    """audio.py:1135-1142: `length` as handed to encode_batch -> (feat_len, out_len).

    Pipeline convention passes 3000 (=> feat_len 6000 >= 1500: no key masked, out_len 3000
    slices the whole [750] output); self-test convention passes after_length.
    """
    return feat_extract_output_lengths(4 * int(length) - 1)
