"""Conformer ConvolutionModule (reference: wenet/transformer/convolution.py:23-144).

pointwise 512->1024, GLU, depthwise k=31, LayerNorm, SiLU, pointwise 512->512, with the padded frames zeroed
before and after.  Parameters keep the reference's Conv1d names and shapes ((2C,C,1), (C,1,K), (C,C,1)) so
checkpoints load unchanged; the computation stays in (B, T, C) layout: the 1x1 convolutions are GEMMs over
the channel axis and the depthwise convolution is a channels-last HIP kernel, so none of the reference's four
transposes is materialised."""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from .layer_norm import LayerNorm


class ConvolutionModule(nn.Module):
    def __init__(self, channels: int, kernel_size: int = 15, activation: nn.Module = nn.ReLU(),
                 norm: str = "batch_norm", causal: bool = False, bias: bool = True):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, kernel_size=1, stride=1, padding=0, bias=bias)
        if causal:
            padding = 0
            self.lorder = kernel_size - 1
        else:
            assert (kernel_size - 1) % 2 == 0
            padding = (kernel_size - 1) // 2
            self.lorder = 0
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, stride=1, padding=padding,
                                        groups=channels, bias=bias)
        assert norm in ["batch_norm", "layer_norm"]
        if norm == "batch_norm":
            self.use_layer_norm = False
            self.norm = nn.BatchNorm1d(channels)
        else:
            self.use_layer_norm = True
            self.norm = LayerNorm(channels)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, kernel_size=1, stride=1, padding=0, bias=bias)
        self.activation = activation
        self.kernel_size = kernel_size

    def forward(self, x: torch.Tensor, mask_pad: torch.Tensor = torch.ones((0, 0, 0), dtype=torch.bool),
                cache: torch.Tensor = torch.zeros((0, 0, 0))) -> Tuple[torch.Tensor, torch.Tensor]:
        """x (B, T, C), mask_pad (B, 1, T) or (0,0,0), cache (B, C, lorder) for causal -> (B, T, C), new_cache."""
        from ..hip_ops import depthwise_conv1d_cl, depthwise_conv1d_cl_autograd, linear
        keep = mask_pad.transpose(1, 2) if mask_pad.size(2) > 0 else None  # (B, T, 1)
        if keep is not None:
            # masked_fill(~keep, 0) as ONE kernel each way: masked_fill = clone + fill, its mask a bitwise_not, and its autograd the
            # same again (10 launches per layer and training step for the two fills of this module; where: 4)
            x = torch.where(keep, x, 0.0)
        if self.lorder > 0:
            if cache.size(2) == 0:
                x = F.pad(x, (0, 0, self.lorder, 0), "constant", 0.0)
            else:
                x = torch.cat((cache.transpose(1, 2), x), dim=1)
            new_cache = x[:, -self.lorder:, :].transpose(1, 2)
        else:
            new_cache = torch.zeros((0, 0, 0), dtype=x.dtype, device=x.device)
        x = linear(x, self.pointwise_conv1.weight.squeeze(-1), self.pointwise_conv1.bias)
        x = F.glu(x, dim=-1)
        lp = 0 if self.lorder > 0 else (self.kernel_size - 1) // 2
        out_len = x.size(1) - self.lorder if self.lorder > 0 else x.size(1)
        if x.dtype not in (torch.float32, torch.bfloat16):
            # fp16 autocast (`--use_amp`): the kernels compute in fp32 / bf16 only -- run them in fp32 (exact for fp16
            # inputs) and round the result back, rather than sending a grouped convolution to the library
            xd = x.dtype
            if torch.is_grad_enabled() and (x.requires_grad or self.depthwise_conv.weight.requires_grad):
                x = depthwise_conv1d_cl_autograd(x.float(), self.depthwise_conv.weight, self.depthwise_conv.bias, lp, out_len)
            else:
                x = depthwise_conv1d_cl(x.float(), self.depthwise_conv.weight.float(),
                                        None if self.depthwise_conv.bias is None else self.depthwise_conv.bias.float(),
                                        left_pad=lp, out_len=out_len)
            x = x.to(xd)
        elif torch.is_grad_enabled() and (x.requires_grad or self.depthwise_conv.weight.requires_grad):
            x = depthwise_conv1d_cl_autograd(x, self.depthwise_conv.weight, self.depthwise_conv.bias, lp, out_len)
        else:
            x = depthwise_conv1d_cl(x, self.depthwise_conv.weight, self.depthwise_conv.bias, left_pad=lp, out_len=out_len)
        if self.use_layer_norm:
            from ..hip_ops import ln_silu_train, ln_silu_train_eligible
            if isinstance(self.activation, torch.nn.SiLU) and isinstance(self.norm, torch.nn.LayerNorm) and self.norm.elementwise_affine \
                    and ln_silu_train_eligible(x, self.norm.weight, self.norm.bias):
                x = ln_silu_train(x, self.norm.weight, self.norm.bias, self.norm.eps)       # GPU training step: one kernel each way
            else:
                x = self.activation(self.norm(x))
        else:
            x = self.activation(self.norm(x.transpose(1, 2)).transpose(1, 2))
        x = linear(x, self.pointwise_conv2.weight.squeeze(-1), self.pointwise_conv2.bias)
        if keep is not None:
            x = torch.where(keep, x, 0.0)
        return x, new_cache
