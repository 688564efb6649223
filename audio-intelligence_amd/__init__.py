"""MI355X-native AF3 / UALM audio-understanding forward pass (log-mel -> AF-Whisper -> LLM decode).

Host-side mirror of the reference's plugin interface (AbsIO / ContinuousAudioIO / AFWhisperEncoder /
SoundTower / ParallelLLM) over a C-ABI HIP library (`csrc/`, `include/afhip.h`).  The HIP library is
loaded on first use; every compute entry point raises if it is missing -- there is no CPU fallback.
"""

__version__ = "0.1.0"
