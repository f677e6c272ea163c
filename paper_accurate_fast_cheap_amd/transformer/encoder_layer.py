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
            if pre and hasattr(norm, "forward_skip"):
                h, x = norm.forward_skip(x)     # (GPU training step: the residual path's gradient is added inside the norm's backward)
                y = fn(h)
            else:
                y = fn(norm(x) if pre else x)
            if isinstance(y, tuple):
                y, side["cache"] = y
            if x.is_cuda and torch.is_grad_enabled():
                # GPU training step: residual + scale * dropout(branch) as ONE kernel forward and one backward
                # (hip_ops.residual_dropout; falls back to the operators below when the kernels do not take the call)
                from ..hip_ops import residual_dropout
                x = residual_dropout(x, y, scale, drop.p, self.training)
            else:
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

    def forward_lookahead(self, x: torch.Tensor, carry: Optional[dict], final: bool = False) -> Tuple[torch.Tensor, dict]:
        """One chunk WITH state carry for the model the reference ships for the uni-directional slot: a NON-causal conv module
        (conf/rwkv/giga.rwkv_uni_ds4k31nc_12le.trans-longutts.yaml:14-16: cnn_module_kernel 31, `# causal: true` commented
        out), whose depthwise convolution reads (k - 1) / 2 = 15 frames either side of a frame (convolution.py:56-60,128).
        Streaming it exactly means each layer EMITS 15 frames behind what it has received: the macaron FFN and the slot run on
        the new frames at once (they look back only), the conv branch finalises the frames whose right context has arrived,
        the second FFN and norm_final follow those.  x: (B, m, C) new frames (m may be 0); carry: {"shift", "wkv"} as
        forward_carry plus "cu" -- the depthwise convolution's input (GLU output) of the last <= k - 1 frames, starting as
        (k - 1) / 2 zero frames = the module's own left padding -- and "x2" -- the residual stream behind the slot for the
        frames not yet emitted.  final: the stream ends: (k - 1) / 2 zero frames of right padding drain the layer.
        Returns (the frames finalised by this call (B, v, C), new carry); over a whole stream the concatenated outputs equal
        the layer's whole-sequence forward."""
        from ..hip_ops import depthwise_conv1d_cl, linear
        from ..rwkv_v6.rwkv_wrapper import RWKV_TmixWrapper
        slot, cm = self.self_attn, self.conv_module
        if type(slot) is not RWKV_TmixWrapper or not self.normalize_before or cm is None or cm.lorder > 0 or not cm.use_layer_norm:
            raise NotImplementedError("look-ahead carry: uni-directional slot, pre-norm, non-causal conv module with layer_norm")
        carry = dict(carry or {})
        B, m, C = x.shape
        half = (cm.kernel_size - 1) // 2
        if "cu" not in carry:
            carry["cu"] = x.new_zeros(B, half, C)                       # the non-causal module's left zero padding
            carry["x2"] = x.new_zeros(B, 0, C)
        if m > 0:
            if self.feed_forward_macaron is not None:
                x = x + self.ff_scale * self.feed_forward_macaron(self.norm_ff_macaron(x))
            h = self.norm_mha(x)
            qd = h.dtype
            if slot.do_bfloat16:
                h = h.to(torch.bfloat16)
            att, carry["shift"], carry["wkv"] = slot.tmix_block.forward_state(h.contiguous(), carry.get("shift"), carry.get("wkv"))
            x2 = x + att.to(qd)
            u = torch.nn.functional.glu(linear(self.norm_conv(x2), cm.pointwise_conv1.weight.squeeze(-1), cm.pointwise_conv1.bias), dim=-1)
            U = torch.cat([carry["cu"], u], dim=1)
            X2 = torch.cat([carry["x2"], x2], dim=1)
        else:
            U, X2 = carry["cu"], carry["x2"]
        if final:
            U = torch.cat([U, U.new_zeros(B, half, C)], dim=1)          # right zero padding
        v = max(0, U.size(1) - 2 * half)                                # frames whose whole window has arrived
        assert v <= X2.size(1)
        if v > 0:
            dw = depthwise_conv1d_cl(U.contiguous(), cm.depthwise_conv.weight, cm.depthwise_conv.bias, left_pad=0, out_len=v)
            c = linear(cm.activation(cm.norm(dw)), cm.pointwise_conv2.weight.squeeze(-1), cm.pointwise_conv2.bias)
            y = X2[:, :v] + c
            y = y + self.ff_scale * self.feed_forward(self.norm_ff(y))
            y = self.norm_final(y)
        else:
            y = x.new_zeros(B, 0, C)
        keep = min(U.size(1), 2 * half)
        carry["cu"] = U[:, U.size(1) - keep:].contiguous()
        carry["x2"] = X2[:, v:].contiguous()
        return y, carry
