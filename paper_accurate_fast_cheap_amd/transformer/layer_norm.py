"""nn.LayerNorm with the reference's parameters and state_dict keys (`weight`, `bias`; encoder_layer.py:61-73,
convolution.py:84, src/model.py:268), whose GPU training step runs on the hand-written kernels: forward = the inference
LayerNorm kernel, backward = pafc_layernorm_bwd.  Everywhere else (CPU, no autograd) it is nn.LayerNorm itself."""
import torch
from torch import nn


class LayerNorm(nn.LayerNorm):
    # True when the only consumer is a projection that autocast would feed bf16 anyway (pre-norm branches): under bf16
    # autocast the fp32 norm then writes bf16 directly -- the values the consumer's cast would have produced
    consumer_casts: bool = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and torch.is_grad_enabled() and self.elementwise_affine and self.bias is not None \
                and len(self.normalized_shape) == 1:
            from ..hip_ops import layer_norm
            return layer_norm(x, self.weight, self.bias, self.eps, bf16_out=self.consumer_casts)
        return super().forward(x)

    def forward_skip(self, x: torch.Tensor):
        """(norm(x), x') for a pre-norm residual branch x + f(norm(x)): add the branch to x' and, in the GPU training step, the
        gradient of that residual path is added inside this norm's backward kernel (hip_ops.layer_norm_with_skip).  Elsewhere
        x' is x."""
        if x.is_cuda and torch.is_grad_enabled() and self.elementwise_affine and self.bias is not None \
                and len(self.normalized_shape) == 1:
            from ..hip_ops import layer_norm_with_skip
            out = layer_norm_with_skip(x, self.weight, self.bias, self.eps, bf16_out=self.consumer_casts)
            if out is not None:
                return out
        return self.forward(x), x
