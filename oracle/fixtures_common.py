"""Shared definitions for golden-vector capture and checking (test infrastructure only)."""
import numpy as np

from audio_intelligence_amd.utils.synthetic import make_wav, make_prompt  # noqa: F401

SEED_ENC_TINY = 0
SEED_LLM_TINY = 0
SEED_ENC_FULL = 1
MAX_STEP = 32

LENGTH_CASES = [1600, 16000, 79999, 80000, 160000, 320000, 479999, 480000, 480001, 600000]

_rng = np.random.default_rng(12345)
MEL_SAMPLE_IDX = sorted(int(i) for i in _rng.choice(128 * 3000, size=64, replace=False))
FILTER_SAMPLE_IDX = sorted(int(i) for i in _rng.choice(201 * 128, size=32, replace=False))
# make sure the filter sample hits non-zero entries too (the bank is ~98 % zeros)
FILTER_SAMPLE_IDX += [1 * 128 + 0, 2 * 128 + 1, 50 * 128 + 46, 100 * 128 + 78, 150 * 128 + 103, 199 * 128 + 127]


def mel_cases():
    """(name, seed, n_samples): 10 x 10 s, 4 x 30 s, one 5 s, one >30 s (truncated by the caller)."""
    out = [(f"s10_{i}", 1000 + i, 160000) for i in range(10)]
    out += [(f"s30_{i}", 2000 + i, 480000) for i in range(4)]
    out += [("s5_0", 500, 80000)]
    return out


def sample_rows(a: np.ndarray, step: int = 13) -> np.ndarray:
    """Strided row sample of a [T,d] activation (keeps fixtures small): rows 0, step, 2*step, ... + last."""
    idx = list(range(0, a.shape[0], step))
    if idx[-1] != a.shape[0] - 1:
        idx.append(a.shape[0] - 1)
    return np.ascontiguousarray(a[idx]).astype(np.float32)


def sample_row_index(n_rows: int, step: int = 13):
    idx = list(range(0, n_rows, step))
    if idx[-1] != n_rows - 1:
        idx.append(n_rows - 1)
    return idx
