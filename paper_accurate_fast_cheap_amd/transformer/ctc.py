"""CTC head (reference: wenet/transformer/ctc.py:20-124): Linear C -> V, log-softmax; CTC loss for training."""
from typing import Tuple

import torch
import torch.nn.functional as F


class CTC(torch.nn.Module):
    def __init__(self, odim: int, encoder_output_size: int, dropout_rate: float = 0.0, reduce: bool = True,
                 blank_id: int = 0):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.ctc_lo = torch.nn.Linear(encoder_output_size, odim)
        self.ctc_loss = torch.nn.CTCLoss(blank=blank_id, reduction="sum" if reduce else "none", zero_infinity=True)
        # long fp32 inputs: the head's product as split operands on the bf16 matrix cores (~2^-16 relative); init_model clears
        # this for a pure-fp32 model (encoder.fp32_split_operands), which keeps exact fp32 products
        self.fp32_split_operands = True

    def forward(self, hs_pad: torch.Tensor, hlens: torch.Tensor, ys_pad: torch.Tensor, ys_lens: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """ctc.py:61-104 (non-focal branch): loss summed over the batch then divided by B; also returns log-probs."""
        ys_hat = self.ctc_lo(F.dropout(hs_pad, p=self.dropout_rate))
        ys_hat = ys_hat.transpose(0, 1).log_softmax(2)
        loss = self.ctc_loss(ys_hat, ys_pad, hlens, ys_lens) / ys_hat.size(1)
        return loss, ys_hat.transpose(0, 1)

    def log_softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        """ctc.py:106-114.  Inference on the GPU: the (B, T', V) logits -- the largest activation of the pass -- are
        normalised in place by one kernel that reads them once; with autograd (training) the framework op is used."""
        if hs_pad.is_cuda and not torch.is_grad_enabled() and hs_pad.dtype in (torch.float32, torch.bfloat16) \
                and hs_pad.dtype == self.ctc_lo.weight.dtype:
            from ..hip_ops import linear_fused, log_softmax_rows
            logits = linear_fused(hs_pad.contiguous(), self.ctc_lo.weight, self.ctc_lo.bias, "none",
                                  split_ok=self.fp32_split_operands)
            return log_softmax_rows(logits, inplace=True)
        return F.log_softmax(self.ctc_lo(hs_pad), dim=2)

    def argmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.ctc_lo(hs_pad), dim=2)
