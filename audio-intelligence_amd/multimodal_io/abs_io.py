"""Plugin contract for multimodal I/O -- same surface as the reference's AbsIO
(UALM/models/ualm/multimodal_io/abs_io.py:21-216): an abstract nn.Module whose optional methods raise
NotImplementedError until a modality implements them."""
from abc import ABC
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
from torch.nn import Module


class AbsIO(ABC, Module):
    def __init__(self, modality: str, is_discrete: bool):
        super().__init__()
        self.modality = modality
        self.is_discrete = is_discrete

    # data path
    def preprocess(self, data: Any) -> Tuple[np.ndarray, Optional[Tuple[int, np.ndarray]], np.ndarray]:
        raise NotImplementedError

    def encode_batch(self, batch_data: List[Any]) -> Dict[str, Any]:
        raise NotImplementedError

    def decode_batch(self, batch_encoded: Dict[str, Any]) -> List[Any]:
        raise NotImplementedError

    # utilities
    def find_length(self, data: Any) -> int:
        raise NotImplementedError

    def copy_for_worker(self) -> "AbsIO":
        raise NotImplementedError

    # modality properties
    def feature_dim(self) -> Optional[int]:
        raise NotImplementedError

    def num_stream(self) -> Optional[int]:
        raise NotImplementedError

    def get_vocabulary(self) -> Optional[List[str]]:
        raise NotImplementedError

    def get_stream_interval(self) -> Optional[List[Tuple[int, int]]]:
        raise NotImplementedError

    def get_stream_weight(self) -> Optional[List[float]]:
        raise NotImplementedError
