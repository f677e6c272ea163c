"""ConformerEncoderLayer (reference: wenet/transformer/encoder_layer.py:112-261).

x += 1/2 FFN_macaron(LN(x)); x += slot(LN(x)); x += conv(LN(x)); x += 1/2 FFN(LN(x)); x = LN_final(x), pre-norm
(normalize_before) or post-norm, dropout identity in eval.  Sub-module and LayerNorm names are the reference's
(feed_forward{,_macaron}, self_attn, conv_module, norm_{ff,mha,ff_macaron,conv,final})."""
from typing import Optional, Tuple

import torch
from torch import nn

from .layer_norm import LayerNorm


class ConformerEncoderLayer(nn.Module):
    def __init__(self, size: int, self_attn: nn.Module, feed_forward: Optional[nn.Module] = None,
                 feed_forward_macaron: Optional[nn.Module] = None, conv_module: Optional[nn.Module] = None,
                 dropout_rate: float = 0.1, normalize_before: bool = True):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.feed_forward_macaron = feed_forward_macaron
        self.conv_module = conv_module
        self.norm_ff = LayerNorm(size, eps=1e-5)
        self.norm_mha = LayerNorm(size, eps=1e-5)
        if feed_forward_macaron is not None:
            self.norm_ff_macaron = LayerNorm(size, eps=1e-5)
            self.ff_scale = 0.5
        else:
            self.ff_scale = 1.0
        if self.conv_module is not None:
            self.norm_conv = LayerNorm(size, eps=1e-5)
            self.norm_final = LayerNorm(size, eps=1e-5)
        # pre-norm branches whose first operation is a projection (FFN w_1, pointwise_conv1) or the slot's own cast
        # to bf16: the consumer takes bf16 under autocast (layer_norm.py).  A post-norm layer's norms feed the residual
        # stream itself, which stays in the stream's dtype.
        if normalize_before:
            self.norm_ff.consumer_casts = True
            if feed_forward_macaron is not None:
                self.norm_ff_macaron.consumer_casts = True
            if self.conv_module is not None:
                self.norm_conv.consumer_casts = True
            self.norm_mha.consumer_casts = bool(getattr(self_attn, "do_bfloat16", False))
        self.dropout = nn.Dropout(dropout_rate)
        self.size = size
        self.normalize_before = normalize_before

    def forward(self, x: torch.Tensor, mask: torch.Tensor, pos_emb: torch.Tensor,
                mask_pad: torch.Tensor = torch.ones((0, 0, 0), dtype=torch.bool),
                att_cache: torch.Tensor = torch.zeros((0, 0, 0, 0)),
                cnn_cache: torch.Tensor = torch.zeros((0, 0, 0, 0)),
                cat_embs: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (x, mask, new_att_cache, new_cnn_cache), the tuple of the reference's layer (encoder_layer.py:165-261).
        A chain of pre-norm residual branches -- half-step macaron FFN, slot, conv module, half-step FFN -- closed by
        norm_final when there is a conv module.  The recurrent slot hands `att_cache` back untouched."""
        drop, pre = self.dropout, self.normalize_before
        side = {}

        def branch(x, norm, fn, scale=1.0):
            # pre-norm: x + scale * fn(norm(x));  post-norm: norm(x + scale * fn(x))   (encoder_layer.py:201-256)
            y = fn(norm(x) if pre else x)
            if isinstance(y, tuple):
                y, side["cache"] = y
            x = x + (drop(y) if scale == 1.0 else scale * drop(y))
            return x if pre else norm(x)

        if self.feed_forward_macaron is not None:
            x = branch(x, self.norm_ff_macaron, self.feed_forward_macaron, self.ff_scale)
        x = branch(x, self.norm_mha, lambda h: self.self_attn(h, h, h, mask, pos_emb, att_cache))
        new_att_cache = side.pop("cache")
        new_cnn_cache = x.new_zeros((0, 0, 0))
        if self.conv_module is not None:
            if not pre and mask_pad.size(2) > 0:
                # the reference's conv module zeroes padded frames IN PLACE on a view of its input (convolution.py:105-109);
                # post-norm hands it the residual stream itself, which therefore loses them too
                x = x.masked_fill(~mask_pad.transpose(1, 2), 0.0)
            x = branch(x, self.norm_conv, lambda h: self.conv_module(h, mask_pad, cnn_cache))
            new_cnn_cache = side.pop("cache")
        x = branch(x, self.norm_ff, self.feed_forward, self.ff_scale)
        if self.conv_module is not None:
            x = self.norm_final(x)
        return x, mask, new_att_cache, new_cnn_cache

    def forward_carry(self, x: torch.Tensor, carry: Optional[dict]) -> Tuple[torch.Tensor, dict]:
        """One chunk WITH recurrent-state carry (uni-directional slot only): what the reference's forward_chunk
        lacks (its wrappers return `cache` untouched, rwkv_wrapper.py:81).  carry = {"shift": (B,1,C) last
        normalised frame of the previous chunk, "wkv": float32 (B,H,N,N) scan state, "cnn": (B,C,lorder) causal-conv
        left context} or None at the start of a stream.  With a causal conv module, chunked == full sequence."""
        from ..rwkv_v6.rwkv_wrapper import RWKV_TmixWrapper
        slot = self.self_attn
        if type(slot) is not RWKV_TmixWrapper:
            raise NotImplementedError("state carry is defined for the uni-directional slot (rwkv_tmix60)")
        if not self.normalize_before:
            raise NotImplementedError("state carry is defined for pre-norm layers")
        carry = carry or {}
        if self.feed_forward_macaron is not None:
            x = x + self.ff_scale * self.feed_forward_macaron(self.norm_ff_macaron(x))
        h = self.norm_mha(x)
        qd = h.dtype
        if slot.do_bfloat16:
            h = h.to(torch.bfloat16)
        att, shift, wkv = slot.tmix_block.forward_state(h, carry.get("shift"), carry.get("wkv"))
        x = x + att.to(qd)
        new = {"shift": shift, "wkv": wkv}
        if self.conv_module is not None:
            empty_mask = torch.ones((0, 0, 0), dtype=torch.bool, device=x.device)
            cnn = carry.get("cnn", torch.zeros((0, 0, 0), dtype=x.dtype, device=x.device))
            c, new_cnn = self.conv_module(self.norm_conv(x), empty_mask, cnn)
            x = x + c
            new["cnn"] = new_cnn
        x = x + self.ff_scale * self.feed_forward(self.norm_ff(x))
        if self.conv_module is not None:
            x = self.norm_final(x)
        return x, new
