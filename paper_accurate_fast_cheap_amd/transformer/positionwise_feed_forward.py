"""PositionwiseFeedForward (reference: wenet/transformer/positionwise_feed_forward.py:20-55):
w_2(dropout(act(w_1 x))), 512 -> 2048 -> 512, activation SiLU in the paper's configs."""
import torch


class PositionwiseFeedForward(torch.nn.Module):
    def __init__(self, idim: int, hidden_units: int, dropout_rate: float,
                 activation: torch.nn.Module = torch.nn.ReLU(), bias: bool = True):
        super().__init__()
        self.w_1 = torch.nn.Linear(idim, hidden_units, bias=bias)
        self.activation = activation
        self.dropout = torch.nn.Dropout(dropout_rate)
        self.w_2 = torch.nn.Linear(hidden_units, idim, bias=bias)

    def forward(self, xs: torch.Tensor) -> torch.Tensor:
        if xs.is_cuda and torch.is_grad_enabled():      # training on the GPU: weight gradients through gemm_tn
            from ..hip_ops import linear, silu_dropout
            h = linear(xs, self.w_1.weight, self.w_1.bias)
            if isinstance(self.activation, torch.nn.SiLU):     # activation + dropout: one pass forward, one backward
                h = silu_dropout(h, self.dropout.p, self.training)
            else:
                h = self.dropout(self.activation(h))
            return linear(h, self.w_2.weight, self.w_2.bias)
        return self.w_2(self.dropout(self.activation(self.w_1(xs))))
