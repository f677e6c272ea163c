"""GlobalCMVN (reference: wenet/transformer/cmvn.py:20-47): (x - mean) * istd with (80,) buffers."""
import torch


class GlobalCMVN(torch.nn.Module):
    def __init__(self, mean: torch.Tensor, istd: torch.Tensor, norm_var: bool = True):
        super().__init__()
        assert mean.shape == istd.shape
        self.norm_var = norm_var
        self.register_buffer("mean", mean)
        self.register_buffer("istd", istd)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x - self.mean
        if self.norm_var:
            x = x * self.istd
        return x
