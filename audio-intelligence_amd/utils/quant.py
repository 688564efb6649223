"""Host-side e4m3 packing of weights for the fp8 paths (BASELINE config 5): one f32 scale per output row."""
import torch


def quantize_rows_e4m3(w: torch.Tensor):
    """[N,K] -> (OCP e4m3 bytes [N,K] as uint8, f32 scale [N]) with w ~= scale[n] * q[n,k]; 448 = e4m3 max."""
    amax = w.float().abs().amax(dim=1).clamp_min(1e-12)
    scale = (amax / 448.0).contiguous()
    q = (w.float() / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), scale
