"""Conv2dSubsampling4 (reference: wenet/transformer/subsampling.py:172-226): Conv2d(1->C,3,s2)+ReLU,
Conv2d(C->C,3,s2)+ReLU, flatten (C x F') per frame, Linear -> C, then the positional-encoding scale.
Parameter names match the reference (embed.conv.{0,2}.*, embed.out.0.*)."""
from typing import Tuple, Union

import os

import torch


class BaseSubsampling(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.right_context = 0
        self.subsampling_rate = 1

    def position_encoding(self, offset: Union[int, torch.Tensor], size: int) -> torch.Tensor:
        return self.pos_enc.position_encoding(offset, size)

    def subsampled_length(self, n: int) -> int:
        """Valid output frames for n valid input frames, as the module's mask slicing counts them."""
        return n


class Conv2dSubsampling4(BaseSubsampling):
    def __init__(self, idim: int, odim: int, dropout_rate: float, pos_enc_class: torch.nn.Module):
        super().__init__()
        self.conv = torch.nn.Sequential(
            torch.nn.Conv2d(1, odim, 3, 2),
            torch.nn.ReLU(),
            torch.nn.Conv2d(odim, odim, 3, 2),
            torch.nn.ReLU(),
        )
        self.out = torch.nn.Sequential(torch.nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim))
        self.pos_enc = pos_enc_class
        self.subsampling_rate = 4
        self.right_context = 6  # (3-1)*1 + (3-1)*2, subsampling.py:197-199

    def subsampled_length(self, n: int) -> int:
        """x_mask[:, :, 2::2][:, :, 2::2] keeps input positions 6, 10, 14, ...: (n - 7) // 4 + 1 of the first n (0 below 7)."""
        return (n - 7) // 4 + 1 if n >= 7 else 0

    # derived weight copies of the inference schedule (_forward_nhwc): rebuilt on demand, never copied or pickled with the module
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if not (k.startswith("_w_") or k.startswith("_nhwc_"))}

    def _forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """Inference schedule of the same arithmetic, kept channels-last end to end (measured on MI355X at the
        30-minute shape: 8.1 ms vs 18.0 ms for the NCHW module chain, whose conv2 is wrapped in two layout
        transposes by the library).  conv1 (1 input channel, 9 taps) is a K=9 GEMM over unfolded 3x3 patches that
        writes (B, T1, F1, C) directly; conv2 then runs on that memory as a channels_last tensor; its (B, T', F', C)
        output feeds `out` through a weight whose columns are permuted from (c, f) to (f, c) order."""
        import torch.nn.functional as F
        c1, c2, lin = self.conv[0], self.conv[2], self.out[0]
        B, T, Fd = x.shape
        C = c1.out_channels
        split_ok = getattr(self, "fp32_split_operands", True)     # the encoder clears it for a pure-fp32 model
        T1, F1 = (T - 3) // 2 + 1, (Fd - 3) // 2 + 1
        from ..hip_ops import param_epoch
        stamp = (lin.weight.data_ptr(), lin.weight._version, c2.weight.data_ptr(), c2.weight._version, x.dtype, param_epoch())
        if getattr(self, "_nhwc_stamp", None) != stamp:
            Fo = lin.in_features // C
            self._w_lin = lin.weight.detach().view(-1, C, Fo).permute(0, 2, 1).reshape(-1, Fo * C).contiguous()
            self._w_c2 = c2.weight.detach().contiguous(memory_format=torch.channels_last)
            self._w_c2_taps = c2.weight.detach().permute(2, 3, 0, 1).reshape(9, C, C).contiguous()   # (tap, co, ci)
            self._w_c2_split = self._w_c2_3 = self._w_lin_3 = None
            if x.dtype == torch.float32 and c2.weight.dtype == torch.float32:
                from ..hip_ops import split_bf16, split_planes
                self._w_c2_split = split_bf16(self._w_c2_taps)
                if C % 128 == 0:
                    self._w_c2_3 = split_planes(self._w_c2_taps, triple=True)          # (9, C, 3C) = [hi | hi | lo] per tap
                    if lin.weight.dtype == torch.float32 and self._w_lin.shape[1] % 128 == 0:
                        # the planes of the DERIVED tensor live and die with it, here, under the same stamp (a cache keyed on
                        # the object would see a new `_w_lin` with `_version` 0 after every weight update)
                        self._w_lin_3 = split_planes(self._w_lin, triple=True)
            self._nhwc_stamp = stamp
            from ..hip_ops import DerivedFill
            self._nhwc_fill = DerivedFill(x.device)      # batches in flight on other streams wait for this fill first
        else:
            self._nhwc_fill.use(x.device)
        if x.dtype == torch.bfloat16 and C % 128 == 0 and 256 % (C // 8) == 0:
            # conv1 + ReLU: write-bound direct kernel (its output is the largest tensor of the whole pass);
            # conv2 + ReLU: hand-written implicit GEMM on the matrix cores; NHWC in and out
            from ..hip_ops import conv3x3s2_c1_nhwc, conv3x3s2_nhwc
            y = conv3x3s2_c1_nhwc(x.contiguous(), c1.weight, c1.bias, relu=True)            # (B, T1, F1, C)
            y = conv3x3s2_nhwc(y, self._w_c2_taps, c2.bias, relu=True)                       # (B, T', F', C)
            b, t, f, c = y.shape
            from ..hip_ops import linear_fused
            return linear_fused(y.view(b, t, f * c), self._w_lin, lin.bias, "none")
        if x.dtype == torch.float32 and self._w_c2_3 is not None and 256 % (C // 8) == 0 and lin.weight.dtype == torch.float32:
            # long fp32 inputs: conv2 as the split-operand implicit GEMM of the phase-pipelined kernel, its output left as the
            # bf16 planes Linear(F' C, odim) reads as they lie (hip_ops.conv_sub_f32split_planes)
            from .. import hip_ops
            T2 = ((T - 3) // 2 + 1 - 3) // 2 + 1
            F2 = ((Fd - 3) // 2 + 1 - 3) // 2 + 1
            if (B * T2 >= hip_ops._SPLIT_GEMM_MIN_ROWS and split_ok and lin.out_features >= 256 and lin.out_features % 8 == 0
                    and self._w_lin_3 is not None):
                y = hip_ops.conv_sub_f32split_planes(x.contiguous(), c1.weight, c1.bias, self._w_c2_3, c2.bias)
                return hip_ops.gemm_ph_ex(y.view(B * T2, F2 * 2 * C), self._w_lin_3, lin.bias, a_split=True,
                                          out_kind="f32", a_plane_block=C).view(B, T2, lin.out_features)
        if x.dtype == torch.float32 and self._w_c2_split is not None and C % 128 == 0 and 256 % (C // 8) == 0:
            # fp32 model: both convolutions on the bf16 matrix cores with hi + lo split operands (fp32 accumulation,
            # ~1e-5 relative to the fp32 convolution) instead of the fp32 MFMA path (57 ms -> 13 ms per 30-minute file)
            from ..hip_ops import conv_sub_f32split
            y = conv_sub_f32split(x.contiguous(), c1.weight, c1.bias, self._w_c2_split[0], self._w_c2_split[1], c2.bias)
            b, t, f, c = y.shape
            from ..hip_ops import linear_fused
            return linear_fused(y.view(b, t, f * c), self._w_lin, lin.bias, "none", split_ok=split_ok)   # (long inputs of a model with the bf16 slot: split operands too)
        p = x.unsqueeze(1).unfold(2, 3, 2).unfold(3, 3, 2).reshape(B, T1 * F1, 9)
        # relu(bias + p W^T) in one GEMM epilogue
        y = torch._addmm_activation(c1.bias, p.view(B * T1 * F1, 9), c1.weight.view(C, 9).t(), use_gelu=False)
        y = y.view(B, T1, F1, C).permute(0, 3, 1, 2)                            # NCHW view of NHWC memory
        y = F.relu(F.conv2d(y, self._w_c2, c2.bias, stride=2))
        b, c, t, f = y.shape
        return F.linear(y.permute(0, 2, 3, 1).reshape(b, t, f * c), self._w_lin, lin.bias)

    def forward(self, x: torch.Tensor, x_mask: torch.Tensor, offset: Union[int, torch.Tensor] = 0
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(B, T, idim), (B, 1, T) -> (B, T', odim), pos_emb, (B, 1, T') with T' = ((T-1)//2-1)//2."""
        if x.is_cuda and not torch.is_grad_enabled() and x.size(1) >= 7:
            x, pos_emb = self.pos_enc(self._forward_nhwc(x), offset)
            return x, pos_emb, x_mask[:, :, 2::2][:, :, 2::2]
        if self._train_kernels_eligible(x):
            # GPU training step under bf16 autocast: both convolutions on the hand-written kernels, NHWC end to end; the
            # (c, f) -> (f, c) column permutation of `out`'s weight is part of the graph, so its gradient lands in place
            from .. import hip_ops
            c1, c2, lin = self.conv[0], self.conv[2], self.out[0]
            y = hip_ops.conv_sub_train(x, c1.weight, c1.bias, c2.weight, c2.bias)
            b, t, f, c = y.shape
            w_lin = lin.weight.view(-1, c, f).permute(0, 2, 1).reshape(-1, f * c)
            if os.environ.get("PAFC_TRAIN_SUB_LINEAR", "1") != "0" and hip_ops.train_gemms_own() and y.dtype == torch.bfloat16:
                x = hip_ops.linear_train(y.view(b, t, f * c), w_lin, lin.bias)       # round 6: the last library GEMMs of the step
            else:
                x = torch.nn.functional.linear(y.view(b, t, f * c), w_lin, lin.bias)
            x, pos_emb = self.pos_enc(x, offset)
            return x, pos_emb, x_mask[:, :, 2::2][:, :, 2::2]
        x = x.unsqueeze(1)  # (B, 1, T, F)
        x = self.conv(x)
        b, c, t, f = x.size()
        x = self.out(x.transpose(1, 2).contiguous().view(b, t, c * f))
        x, pos_emb = self.pos_enc(x, offset)
        return x, pos_emb, x_mask[:, :, 2::2][:, :, 2::2]

    def _train_kernels_eligible(self, x: torch.Tensor) -> bool:
        if not (x.is_cuda and torch.is_grad_enabled() and x.size(1) >= 7 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            return False
        from ..hip_ops import train_kernels_enabled
        C = self.conv[0].out_channels
        return train_kernels_enabled() and C % 128 == 0 and 256 % (C // 8) == 0 and self.conv[2].in_channels % 64 == 0


class LinearNoSubsampling(BaseSubsampling):
    """input_layer: linear (subsampling.py:66-113)."""

    def __init__(self, idim: int, odim: int, dropout_rate: float, pos_enc_class: torch.nn.Module):
        super().__init__()
        self.out = torch.nn.Sequential(
            torch.nn.Linear(idim, odim),
            torch.nn.LayerNorm(odim, eps=1e-5),
            torch.nn.Dropout(dropout_rate),
        )
        self.pos_enc = pos_enc_class

    def forward(self, x, x_mask, offset=0):
        x = self.out(x)
        x, pos_emb = self.pos_enc(x, offset)
        return x, pos_emb, x_mask
