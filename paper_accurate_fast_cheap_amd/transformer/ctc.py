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

    def loss(self, hs_pad: torch.Tensor, hlens: torch.Tensor, ys_pad: torch.Tensor, ys_lens: torch.Tensor) -> torch.Tensor:
        """forward()[0] for callers that drop the log-probabilities (the training objectives): in the GPU training step under
        bf16 autocast the head, the loss and their gradients run on the hand-written kernels (hip_ops.ctc_head_loss), which never
        form the (B, L, V) log-probabilities; otherwise forward()."""
        if self.dropout_rate == 0.0 and self.ctc_loss.reduction == "sum" and self.ctc_loss.zero_infinity:
            from ..hip_ops import ctc_head_loss, ctc_head_loss_eligible
            if ctc_head_loss_eligible(hs_pad, self.ctc_lo.weight, ys_pad):
                return ctc_head_loss(hs_pad, self.ctc_lo.weight, self.ctc_lo.bias, hlens, ys_pad, ys_lens, self.ctc_loss.blank)
        return self.forward(hs_pad, hlens, ys_pad, ys_lens)[0]

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
