"""Inference executor for a ConformerEncoderLayer with the recurrent slot: same modules, same parameters, same
arithmetic as ConformerEncoderLayer.forward / RWKV_Tmix_x060c.forward (reference: wenet/transformer/encoder_layer.py:
165-261, wenet/rwkv_v6/src/model.py:271-325, wenet/transformer/convolution.py:89-144), re-scheduled for HBM:

  * every residual-add + pre-norm pair is ONE pass (pafc_add_layernorm), including norm_final + the next layer's
    first pre-norm, the conv module's masked_fill / LayerNorm / SiLU, and the two ln_x of the slot, which land side
    by side in one (M, 2C) buffer so that both output projections and the (a + b) / 2 become one GEMM with K = 2C;
  * token shift and the five lerps are two passes for BOTH directions (pafc_tmix_shift_mix, pafc_tmix_mix4)
    instead of ~17 element-wise kernels per direction, and nothing is flipped;
  * r/k/v projections of both directions are one batched GEMM, the decay LoRA one batched GEMM pair;
  * GLU is fused into the channels-last depthwise convolution.

Every bf16 projection of the layer runs on the hand-written GEMM (csrc/gemm_ph.hip for the long-form shapes,
csrc/gemm_bf16.hip otherwise) with its bias / activation / GLU / residual as the epilogue; fp32 projections run as split
operands on the same kernels (long inputs of a model with the bf16 slot) or on the hand-written fp32 GEMM (csrc/gemm_f32.hip,
through hip_ops.linear_bias_act / gemm_f32) -- no library GEMM on any inference path.  Used only under torch.no_grad() on GPU tensors; training goes through the plain
module path (autograd)."""
from typing import List, Optional, Tuple

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip_ops
from ..rwkv_v6.rwkv_wrapper import RWKV_TmixWrapper
from ..rwkv_v6.rwkv_wrapper_bidirectional import RWKV_TmixWrapper_bidirectional
from ..rwkv_v6.rwkv_wrapper_bidirectional_direction_dropout import (RWKV_TmixWrapper_bidirectional_direction_dropout,
                                                                     RWKV_TmixWrapper_bidirectional_direction_dropout_both)
from ..rwkv_v6.wkv6_op import wkv6_forward, wkv6_forward_bidir


_DIR_DROP_SLOTS = (RWKV_TmixWrapper_bidirectional_direction_dropout, RWKV_TmixWrapper_bidirectional_direction_dropout_both)
_RWKV_SLOTS = (RWKV_TmixWrapper, RWKV_TmixWrapper_bidirectional) + _DIR_DROP_SLOTS


class LayerPlan:
    """Stacked / pre-transposed views of one layer's parameters for the batched GEMMs (rebuilt when stale)."""

    def __init__(self, layer: nn.Module):
        self.layer = layer
        slot = layer.self_attn
        # the slot is either the RWKV time-mix (fused here too) or any other registry slot with the MHA-shaped forward
        # (Mamba-2): then only the rest of the layer is re-scheduled and the slot module is called as is
        self.rwkv = type(slot) in _RWKV_SLOTS
        self.reverse0 = False                  # direction of blocks[0]: left-to-right unless an eval-time variant says otherwise
        if type(slot) in _DIR_DROP_SLOTS:
            # the EVAL branches of the direction-dropout wrappers (rwkv_wrapper_bidirectional_direction_dropout{,_both}.py:70-92;
            # the executor only runs in eval mode): both directions averaged in the layers RWKV_BIDIRECTIONAL_LAYERS keeps
            # bidirectional, else one direction -- right-to-left in odd layers under RWKV_ALT_DECODING, left-to-right otherwise.
            # Both are fixed at construction, as in the reference (:25-33).
            fwd, bwd = slot.rwkv_wrapper_forward, slot.rwkv_wrapper_backward
            if slot.bi_active:
                self.blocks = [fwd.tmix_block, bwd.tmix_block]
            elif slot.alt_decoding and slot.layer_id % 2 == 1:
                self.blocks, self.reverse0 = [bwd.tmix_block], True
            else:
                self.blocks = [fwd.tmix_block]
            self.slot_bf16 = bool(fwd.do_bfloat16)
        elif isinstance(slot, RWKV_TmixWrapper_bidirectional):
            self.blocks = [slot.rwkv_wrapper_forward.tmix_block, slot.rwkv_wrapper_backward.tmix_block]
        elif self.rwkv:
            self.blocks = [slot.tmix_block]
        else:
            self.blocks = []
        self.ndir = len(self.blocks)
        if type(slot) not in _DIR_DROP_SLOTS:
            self.slot_bf16 = bool(getattr(slot, "do_bfloat16", False))
        self._stamp = None
        self.refresh()

    def _current_stamp(self):
        # (storage, in-place version, dtype) of every parameter the plan derives something from, and the parameter epoch
        # (hip_ops.param_epoch: fused optimizers move parameters without touching Tensor._version).  The parameter OBJECTS are
        # looked up once per epoch -- walking the modules on every forward cost 90 us per layer of a launch-bound decode batch.
        # Contract: code that REPLACES a parameter object (`module.weight = nn.Parameter(...)`, pruning / re-parametrisation, weight
        # tying) calls hip_ops.bump_param_epoch(); .to() / .cuda() / load_state_dict / train() / eval() / train_step do it themselves.
        ep = hip_ops.param_epoch()
        if getattr(self, "_stamp_params", None) is None or self._stamp_epoch != ep:
            L = self.layer
            ps = [p for b in self.blocks for p in b.parameters()] + [L.feed_forward_macaron.w_2.bias, L.feed_forward.w_2.bias,
                                                                        L.conv_module.pointwise_conv1.weight,
                                                                        L.conv_module.pointwise_conv1.bias,
                                                                        L.feed_forward_macaron.w_1.weight, L.feed_forward_macaron.w_2.weight,
                                                                        L.feed_forward.w_1.weight, L.feed_forward.w_2.weight,
                                                                        L.conv_module.pointwise_conv2.weight,
                                                                        L.norm_ff_macaron.weight, L.norm_ff_macaron.bias, L.norm_ff.weight,
                                                                        L.norm_ff.bias, L.norm_conv.weight, L.norm_conv.bias,
                                                                        L.feed_forward_macaron.w_1.bias, L.feed_forward.w_1.bias]
            self._stamp_params = [p for p in ps if p is not None]
            self._stamp_epoch = ep
        return (ep,) + tuple((p.data_ptr(), p._version, p.dtype) for p in self._stamp_params)

    def refresh(self):
        """Rebuild the derived tensors when a parameter they come from has changed; either way, make the CURRENT stream wait for
        the stream that filled them (hip_ops.DerivedFill: decode batches in flight on side streams share one plan)."""
        stamp = self._current_stamp()
        if stamp == self._stamp:
            self._fill.use()
            return
        bl = self.blocks
        with torch.no_grad():
            if self.rwkv:
                self._refresh_rwkv(bl)
            # biases of the two FFN output projections, pre-multiplied by ff_scale: the residual-fused GEMM computes
            # x + ff_scale * (h W2^T) + (ff_scale * b2)
            L = self.layer
            self.b2_macaron = (L.feed_forward_macaron.w_2.bias * L.ff_scale).contiguous()
            self.b2 = (L.feed_forward.w_2.bias * L.ff_scale).contiguous()
            # pointwise_conv1 with its rows interleaved (64 values, 64 gates, ...) so that F.glu is a GEMM epilogue
            pw1 = L.conv_module.pointwise_conv1
            self.pw1_glu = None                # {block half: (weight, bias)} in the two row orders the GEMM kernels want
            rows2 = pw1.weight.shape[0]                      # 2C: h = 64 blocks need 2C % 128 == 0, h = 32 blocks 2C % 256 == 0
            if pw1.weight.dtype == torch.bfloat16 and pw1.weight.is_cuda and rows2 % 128 == 0 \
                    and pw1.weight.shape[1] % 64 == 0:
                self.pw1_glu = {h: (hip_ops.glu_interleave(pw1.weight.squeeze(-1), h),
                                    hip_ops.glu_interleave(pw1.bias, h) if pw1.bias is not None else None)
                                for h in ((64, 32) if rows2 % 256 == 0 else (64,))}
            self._refresh_split()
            self._refresh_lnfold()
        self._stamp = stamp
        self._fill = hip_ops.DerivedFill()

    def _refresh_lnfold(self):
        """bf16 layer, C = 512: the three pre-norm LayerNorms whose consumer is a projection (norm_ff_macaron -> w_1,
        norm_conv -> pointwise_conv1, norm_ff -> w_1) folded into that projection (csrc/gemm_ph.hip, LNF):
        W' = bf16(gamma * W), b' = b + W beta, csum[n] = sum_k W'[n][k] (fp32, of the ROUNDED W': what the MFMA multiplies)."""
        L = self.layer
        self.lnf = None
        w1 = L.feed_forward.w_1.weight
        cm = L.conv_module
        if not (self.rwkv and w1.dtype == torch.bfloat16 and w1.is_cuda and L.size == 512 and L.feed_forward_macaron is not None
                and cm is not None and cm.pointwise_conv1.weight.shape[0] % 256 == 0 and cm.lorder == 0
                and os.environ.get("PAFC_LN_FOLD", "1") != "0"):
            return

        def fold(w, b, norm, interleave=False):
            wf, g, be = w.float(), norm.weight.float(), norm.bias.float()
            bf = (b.float() if b is not None else 0) + wf @ be
            wp = (wf * g).to(torch.bfloat16)
            if interleave:
                wp, bf = hip_ops.glu_interleave(wp, 32), hip_ops.glu_interleave(bf, 32)
            return wp.contiguous(), bf.to(torch.bfloat16).contiguous(), wp.float().sum(-1).contiguous()

        pw1 = cm.pointwise_conv1
        self.lnf = dict(ffm=fold(L.feed_forward_macaron.w_1.weight, L.feed_forward_macaron.w_1.bias, L.norm_ff_macaron),
                        ff=fold(L.feed_forward.w_1.weight, L.feed_forward.w_1.bias, L.norm_ff),
                        pw1=fold(pw1.weight.squeeze(-1), pw1.bias, L.norm_conv, interleave=True))

    def carry_folds(self):
        """The chunk step's two LayerNorms whose consumer is a few-rows projection -- ln_x -> output, norm_ff -> w_1 -- folded
        into that projection (csrc/gemm_skinny.hip, statistics formed in the launch): W' = bf16(gamma * W), b' = b + W beta,
        csum = row sums of the ROUNDED W'.  Built on first use, rebuilt when the parameters change (refresh)."""
        if getattr(self, "_carry_folds", None) is not None and self._carry_folds[0] == self._stamp:
            self._carry_folds[2].use()
            return self._carry_folds[1]
        L = self.layer

        def fold(w, b, norm):
            wf, g, be = w.float(), norm.weight.float(), norm.bias.float()
            bf = (b.float() if b is not None else 0) + wf @ be
            wp = (wf * g).to(torch.bfloat16).contiguous()
            return wp, bf.to(torch.bfloat16).contiguous(), wp.float().sum(-1).contiguous(), norm.eps

        folds = dict(out=fold(self.Wo, None, self.blocks[0].ln_x), ff=fold(L.feed_forward.w_1.weight, L.feed_forward.w_1.bias, L.norm_ff))
        self._carry_folds = (self._stamp, folds, hip_ops.DerivedFill())
        return folds

    def _refresh_split(self):
        """fp32 layer around a bf16 slot (the reference's default precision): every fp32 projection as the split-operand
        weight [hi | hi | lo] of csrc/gemm_ph.hip (three bf16 products per fp32 product, ~2^-16 relative), biases fp32."""
        L = self.layer
        self.split = None
        w1 = L.feed_forward.w_1.weight
        cm = L.conv_module
        if not (self.rwkv and self.slot_bf16 and w1.dtype == torch.float32 and w1.is_cuda and L.size % 128 == 0
                and w1.shape[0] % 128 == 0 and cm is not None and cm.pointwise_conv1.weight.shape[0] % 256 == 0):
            return
        sp = hip_ops.split_planes
        pw1 = cm.pointwise_conv1
        self.split = dict(
            ffm_w1=sp(L.feed_forward_macaron.w_1.weight.contiguous(), True), ffm_w2=sp(L.feed_forward_macaron.w_2.weight.contiguous(), True),
            ff_w1=sp(L.feed_forward.w_1.weight.contiguous(), True), ff_w2=sp(L.feed_forward.w_2.weight.contiguous(), True),
            pw1=sp(hip_ops.glu_interleave(pw1.weight.squeeze(-1), 32).contiguous(), True),
            pw1_b=hip_ops.glu_interleave(pw1.bias, 32).contiguous() if pw1.bias is not None else None,
            pw1_plain=sp(pw1.weight.squeeze(-1).contiguous(), True),          # rows as they lie: the few-rows form (GLU in the dwconv)
            pw2=sp(cm.pointwise_conv2.weight.squeeze(-1).contiguous(), True))

    def _refresh_rwkv(self, bl):
        self.maa_x = [b.time_maa_x.reshape(-1).contiguous() for b in bl]
        self.maa_x_n = torch.stack(self.maa_x).contiguous()                            # (nd, C)
        self.W1 = torch.stack([b.time_maa_rkvw_w1 for b in bl]).contiguous()           # (nd, C, 128)
        self.W1n = torch.stack([b.time_maa_rkvw_w1.t() for b in bl]).contiguous()      # (nd, 128, C): Linear layout
        self.D1n = torch.stack([b.time_decay_w1.t() for b in bl]).contiguous()         # (nd, 64, C)
        self.W2 = [b.time_maa_rkvw_w2.contiguous() for b in bl]                        # (4, 32, C) each
        self.W2t = torch.stack([b.time_maa_rkvw_w2.transpose(1, 2) for b in bl]).contiguous()  # (nd, 4, C, 32)
        self.maa4 = torch.stack([torch.stack([b.time_maa_r.reshape(-1), b.time_maa_k.reshape(-1),
                                              b.time_maa_v.reshape(-1), b.time_maa_w.reshape(-1)])
                                 for b in bl]).contiguous()                             # (nd, 4, C)
        # q-major, direction-minor: [r_0, r_1, k_0, k_1, v_0, v_1], each W^T (C_in, C_out)
        self.Wrkv = torch.stack([getattr(b, n).weight.t() for n in ("receptance", "key", "value") for b in bl]
                                ).contiguous()
        self.Wrkv_n = torch.stack([getattr(b, n).weight for n in ("receptance", "key", "value") for b in bl]
                                  ).contiguous()                                        # the same in nn.Linear layout (N, K)
        self.D2n = torch.stack([b.time_decay_w2.t() for b in bl]).contiguous()         # (nd, C, 64)
        self.D1 = torch.stack([b.time_decay_w1 for b in bl]).contiguous()              # (nd, C, 64)
        self.D2 = torch.stack([b.time_decay_w2 for b in bl]).contiguous()              # (nd, 64, C)
        self.time_decay = torch.stack([b.time_decay.reshape(1, -1) for b in bl]).contiguous()  # (nd, 1, C)
        self.u = [b.time_faaaa.contiguous() for b in bl]
        wo = torch.cat([b.output.weight for b in bl], dim=1)                           # (C, nd*C)
        self.Wo = (wo * 0.5 if self.ndir == 2 else wo).contiguous()                    # /2 is exact in bf16


def eligible(layer: nn.Module) -> bool:
    from .mamba2 import MambaAttWrapper
    slot = layer.self_attn
    if type(slot) not in _RWKV_SLOTS + (MambaAttWrapper,):
        return False
    cm = layer.conv_module
    return (layer.normalize_before and layer.feed_forward_macaron is not None and cm is not None
            and cm.use_layer_norm and isinstance(cm.activation, nn.SiLU)
            and isinstance(layer.feed_forward.activation, nn.SiLU)
            and isinstance(layer.feed_forward_macaron.activation, nn.SiLU) and cm.kernel_size in (3, 7, 15, 31)
            and layer.size % 8 == 0 and layer.size <= 1024)


# bf16 projections with at least this many rows run on the hand-written tiled GEMMs (hip_ops.DISPATCH: since the small kernel has
# 128 x 64 / 64 x 64 tile variants that is every row count that is not a few-rows launch of a chunk step); unmasked inputs of
# at least _LN_FOLD_MIN_ROWS rows take the folded-LayerNorm schedule.
_OWN_GEMM_MIN_ROWS = hip_ops.DISPATCH["own_gemm_min_rows"]
_LN_FOLD_MIN_ROWS = hip_ops.DISPATCH["ln_fold_min_rows"]


def _own_gemm(x: torch.Tensor, w: torch.Tensor) -> bool:
    """bf16 projections run on the hand-written GEMMs (csrc/gemm_ph.hip / gemm_bf16.hip) -- except the few rows of a
    streaming chunk step (csrc/gemm_skinny.hip, asked first); fp32 ones on the fp32 GEMM or as split operands."""
    rows = x.numel() // w.shape[-1]
    return (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.shape[-1] % 64 == 0 and w.shape[-2] % 8 == 0
            and rows >= _OWN_GEMM_MIN_ROWS and not hip_ops.skinny_ok(rows, w.shape[-2], w.shape[-1]))


def _skinny(x: torch.Tensor, w: torch.Tensor, glu: bool = False) -> bool:
    """bf16 projections of a handful of rows (a streaming chunk of one or two streams): csrc/gemm_skinny.hip."""
    return (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_cuda and w.dim() == 2 and w.is_contiguous()
            and hip_ops.skinny_ok(x.numel() // w.shape[-1], w.shape[0], w.shape[1], glu))


def proj(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: str = "none", alpha: float = 1.0,
         residual: Optional[torch.Tensor] = None, inplace: bool = False) -> torch.Tensor:
    """act(alpha * x w^T + bias + residual) as ONE GEMM with a fused epilogue; w in nn.Linear layout (N, K)."""
    if _own_gemm(x, w):
        N, K = w.shape
        x2 = x.reshape(-1, K)
        r2 = residual.reshape(-1, N) if residual is not None else None
        out = hip_ops.gemm_bf16(x2, w, bias, act, alpha=alpha, residual=r2, out=r2 if (inplace and r2 is not None) else None)
        return out.view(x.shape[:-1] + (N,))
    if _skinny(x, w):      # a streaming chunk: few rows, launch-bound
        N, K = w.shape
        x2 = x.reshape(-1, K)
        r2 = residual.reshape(-1, N) if residual is not None else None
        out = hip_ops.gemm_skinny(x2, w, bias, act, alpha=alpha, residual=r2, out=r2 if (inplace and r2 is not None) else None)
        return out.view(x.shape[:-1] + (N,))
    return hip_ops.linear_bias_act(x, w, bias, act, alpha=alpha, residual=residual, inplace=inplace)


def _ffn_residual(ff: nn.Module, h: torch.Tensor, x: torch.Tensor, scale: float, b2_scaled: torch.Tensor,
                  inplace: bool) -> torch.Tensor:
    """x + scale * ff(h): bias + SiLU ride on the w_1 GEMM, the residual add on the w_2 GEMM, so the hidden tensor is
    written once and the pre-norm that follows reads ONE tensor."""
    hid = proj(h, ff.w_1.weight, ff.w_1.bias, "silu")
    return proj(hid, ff.w_2.weight, b2_scaled, "none", alpha=scale, residual=x, inplace=inplace)


def _pw1_glu(plan: "LayerPlan", h2: torch.Tensor) -> torch.Tensor:
    """pointwise_conv1 + F.glu as one GEMM: the weight rows in the block order the kernel picked for this row count wants."""
    w64 = plan.pw1_glu[64][0]
    half = hip_ops.gemm_glu_half(h2.shape[0], w64.shape[0], w64.shape[1])   # 32 only when 2C % 256 == 0 (ph_tile_m)
    w, b = plan.pw1_glu[half]
    return hip_ops.gemm_bf16(h2, w, b, act="glu")


def slot_forward(plan: LayerPlan, h: torch.Tensor, residual: Optional[torch.Tensor] = None,
                 stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h: (B, T, C) in the slot dtype -> slot output (B, T, C) in the slot dtype (both directions averaged).
    With ``residual`` (same dtype) the result is residual + slot output, written over ``residual``; ``stats`` (rows, 8, 2)
    then also receives the new rows' statistics (the LayerNorm that follows is folded into its consumer)."""
    B, T, C = h.shape
    M, nd = B * T, plan.ndir
    rev = plan.reverse0                       # (two directions: always left-to-right first)
    own_gemm = h.dtype == torch.bfloat16 and C % 64 == 0
    f32_own = h.dtype == torch.float32 and C % 4 == 0         # (anything else: a bf16 slot with C % 64 != 0 -- the framework's bmm)
    if own_gemm:   # token shift, first lerp, down-projection and tanh in one pass (one rounding of the product, then tanh)
        t = hip_ops.tmix_lora_down(h, plan.maa_x_n, plan.W1n, reverse0=rev)                  # (nd, M, 128)
    else:
        xxx = hip_ops.tmix_shift_mix(h, plan.maa_x[0], plan.maa_x[1] if nd == 2 else None, reverse0=rev)
        if f32_own:     # fp32 slot (rwkv_do_bfloat16: False): exact fp32 products on the hand-written fp32 GEMM, tanh in its epilogue
            t = hip_ops.gemm_f32(xxx.view(nd, M, C), plan.W1n, None, "tanh")
        else:
            t = torch.tanh(torch.bmm(xxx.view(nd, M, C), plan.W1))
    if h.dtype == torch.bfloat16 and t.shape[-1] == 128 and C % 64 == 0:
        z = hip_ops.tmix_lora_mix4(h, t, plan.W2t, plan.maa4, reverse0=rev)   # LoRA up-projection on MFMA inside the lerp pass
    else:
        m = torch.empty((nd, 4, M, C), dtype=h.dtype, device=h.device)
        for d in range(nd):
            if f32_own:     # four (M, 32) x (32, C) products per direction: one batched launch over strided views of t
                hip_ops.gemm_f32(t[d].view(M, 4, -1).transpose(0, 1), plan.W2t[d], out=m[d])
            else:
                torch.bmm(t[d].view(M, 4, -1).transpose(0, 1), plan.W2[d], out=m[d])
        z = hip_ops.tmix_mix4(h, m, plan.maa4, reverse0=rev)                                # (4, nd, M, C)
    if own_gemm and hip_ops.skinny_ok(M, C, C):  # a chunk step: the few-rows kernel
        rkv = hip_ops.gemm_skinny(z[:3].view(3 * nd, M, C), plan.Wrkv_n)
    elif own_gemm and M >= _OWN_GEMM_MIN_ROWS:
        rkv = hip_ops.gemm_bf16(z[:3].view(3 * nd, M, C), plan.Wrkv_n)                      # (3nd, M, C), one launch
    elif f32_own:
        rkv = hip_ops.gemm_f32(z[:3].view(3 * nd, M, C), plan.Wrkv_n)                       # (3nd, M, C), one launch
    else:
        rkv = torch.bmm(z[:3].view(3 * nd, M, C), plan.Wrkv)
    if own_gemm:
        # decay LoRA in one pass (the 64-wide hidden tensor stays on chip); there time_decay rides along, rounded where the
        # reference's `self.time_decay + lora` rounds (src/model.py:289), and the scan runs its variant without a bias, which
        # issues neither the add nor the rounding per step (in both of its passes).  Below the one-pass size a bias would be
        # an extra pass over w: the bidirectional scan adds it itself.
        bias_in_lora = nd == 1 or hip_ops.decay_lora_one_pass(M, C, plan.D1n.shape[1])
        w = hip_ops.decay_lora(z[3], plan.D1n, plan.D2n, plan.time_decay.view(nd, C) if bias_in_lora else None)
    elif f32_own:
        w = hip_ops.gemm_f32(hip_ops.gemm_f32(z[3], plan.D1n, None, "tanh"), plan.D2n)
    else:
        w = torch.bmm(torch.tanh(torch.bmm(z[3], plan.D1)), plan.D2)
    if nd == 1 and not own_gemm:
        w = w + plan.time_decay            # uni: one extra pass; bi: time_decay is added inside the scan kernel
    ycat = torch.empty((M, nd * C), dtype=h.dtype, device=h.device)
    if nd == 2:
        ys = wkv6_forward_bidir(
            (rkv[0].view(B, T, C), rkv[2].view(B, T, C), rkv[4].view(B, T, C), w[0].view(B, T, C), plan.u[0]),
            (rkv[1].view(B, T, C), rkv[3].view(B, T, C), rkv[5].view(B, T, C), w[1].view(B, T, C), plan.u[1]),
            w_bias=None if own_gemm and bias_in_lora else (plan.time_decay[0].view(-1), plan.time_decay[1].view(-1)))
    else:
        ys = (wkv6_forward(rkv[0].view(B, T, C), rkv[1].view(B, T, C), rkv[2].view(B, T, C), w[0].view(B, T, C),
                           plan.u[0], reverse=rev),)
    for d, y in enumerate(ys):
        ln = plan.blocks[d].ln_x
        hip_ops.add_layernorm(y.view(M, C), None, 1.0, ln.weight, ln.bias, out1=ycat[:, d * C:(d + 1) * C], eps=ln.eps)
    if residual is not None and residual.dtype == torch.float32 and ycat.dtype == torch.bfloat16:
        # bf16 slot inside an fp32 stream: the output projection adds straight into the fp32 residual (rwkv_wrapper_
        # bidirectional.py:55-56 `.float()` + encoder_layer.py:232), one rounding less than cast + add
        r2 = residual.view(M, C)
        return hip_ops.gemm_ph_ex(ycat, plan.Wo, None, residual=r2, out=r2, out_kind="f32").view(B, T, C)
    if residual is not None and stats is not None:
        r2 = residual.view(M, C)
        return hip_ops.gemm_bf16_ln(ycat, plan.Wo, None, stats, residual=r2, out=r2).view(B, T, C)
    if residual is not None:
        return proj(ycat, plan.Wo, None, "none", residual=residual.view(M, C), inplace=True).view(B, T, C)
    return proj(ycat, plan.Wo, None).view(B, T, C)


def _dwconv_norm_silu(cm, p: torch.Tensor, left_pad: int, T: int) -> torch.Tensor:
    """activation(norm(depthwise_conv(p))) of the conv module (convolution.py:131-138): one pass when the convolution kernel
    can normalise its own rows (bf16, 512 channels, LayerNorm as the module's norm, SiLU), else convolution + LayerNorm pass."""
    if hip_ops.dwconv_ln_silu_ok(p, cm.depthwise_conv.weight):
        return hip_ops.depthwise_conv1d_cl_ln_silu(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, left_pad, T,
                                                   cm.norm.weight, cm.norm.bias, cm.norm.eps)
    dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, left_pad, T)
    return hip_ops.add_layernorm(dw, None, 1.0, cm.norm.weight, cm.norm.bias, silu=True, eps=cm.norm.eps)[1]


def layer_forward(plan: LayerPlan, x: torch.Tensor, h: torch.Tensor, lens: Optional[torch.Tensor],
                  next_norm: Optional[nn.LayerNorm]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x: residual stream (B, T, C); h = norm_ff_macaron(x), already computed by the previous step.
    Returns (layer output = norm_final(...), next_norm(layer output) or None)."""
    L = plan.layer
    B, T, C = x.shape
    slot_dtype = torch.bfloat16 if plan.slot_bf16 else x.dtype
    masked = lens is not None
    cm = L.conv_module
    # Residual adds ride on the GEMM that produces the branch output whenever branch and stream share a dtype.  The
    # first add of a layer writes a fresh tensor (the incoming stream may be a caller-visible layer output), the
    # later ones update the layer-private stream in place.
    x = _ffn_residual(L.feed_forward_macaron, h, x, L.ff_scale, plan.b2_macaron, inplace=False)
    _, h, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_mha.weight, L.norm_mha.bias, out_dtype=slot_dtype, want_x=False, eps=L.norm_mha.eps)
    if not plan.rwkv:
        # another slot from the registry (Mamba-2): the module as is; the rest of the layer stays re-scheduled
        att = L.self_attn(h, h, h)[0]
        x, h, _ = hip_ops.add_layernorm(x, att.to(x.dtype).contiguous(), 1.0, L.norm_conv.weight, L.norm_conv.bias,
                                        zero_rows=masked, lens=lens, T=T, eps=L.norm_conv.eps)
    elif slot_dtype == x.dtype:
        x = slot_forward(plan, h, residual=x)
        _, h, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_conv.weight, L.norm_conv.bias, zero_rows=masked, lens=lens,
                                        T=T, want_x=False, eps=L.norm_conv.eps)
    else:
        att = slot_forward(plan, h).to(x.dtype)
        x, h, _ = hip_ops.add_layernorm(x, att, 1.0, L.norm_conv.weight, L.norm_conv.bias, zero_rows=masked, lens=lens, T=T, eps=L.norm_conv.eps)
    left_pad, Tc = (cm.kernel_size - 1) // 2, T
    if cm.lorder > 0:
        # causal module: the reference zero-pads lorder frames BEFORE pointwise_conv1 (convolution.py:112-118), so the
        # left context the depthwise convolution sees is GLU(bias), not zero -- prepend the zero frames here as well
        h = torch.cat([h.new_zeros(B, cm.lorder, C), h], dim=1)
        left_pad, Tc = 0, T + cm.lorder
    if plan.pw1_glu is not None and h.dtype == torch.bfloat16:
        # F.glu rides on pointwise_conv1 (half the write, and the depthwise kernel no longer recomputes sigmoids)
        p = _pw1_glu(plan, h.view(B * Tc, C)).view(B, Tc, C)
        g = _dwconv_norm_silu(cm, p, left_pad, T)
    else:
        p = hip_ops.linear_bias_act(h, cm.pointwise_conv1.weight.squeeze(-1), cm.pointwise_conv1.bias, "none")
        dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, left_pad, T, glu=True)
        _, g, _ = hip_ops.add_layernorm(dw, None, 1.0, cm.norm.weight, cm.norm.bias, silu=True, eps=cm.norm.eps)
    if not masked:
        x = proj(g, cm.pointwise_conv2.weight.squeeze(-1), cm.pointwise_conv2.bias, "none", residual=x, inplace=True)
        _, h, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_ff.weight, L.norm_ff.bias, want_x=False, eps=L.norm_ff.eps)
    else:   # padded frames of the conv branch count as zero (convolution.py:140-141): the add stays in the norm pass
        c = proj(g, cm.pointwise_conv2.weight.squeeze(-1), cm.pointwise_conv2.bias)
        x, h, _ = hip_ops.add_layernorm(x, c, 1.0, L.norm_ff.weight, L.norm_ff.bias, lens=lens, T=T, mask_y=True, eps=L.norm_ff.eps)
    x = _ffn_residual(L.feed_forward, h, x, L.ff_scale, plan.b2, inplace=True)
    if next_norm is not None and next_norm.eps != L.norm_final.eps:      # the one-pass pair shares one epsilon
        _, out, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False,
                                          eps=L.norm_final.eps)
        _, hn, _ = hip_ops.add_layernorm(out, None, 1.0, next_norm.weight, next_norm.bias, want_x=False, eps=next_norm.eps)
        return out, hn
    _, out, hn = hip_ops.add_layernorm(x, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False,
                                       gamma2=next_norm.weight if next_norm is not None else None,
                                       beta2=next_norm.bias if next_norm is not None else None, eps=L.norm_final.eps)
    return out, hn


def lnfold_eligible(plan: LayerPlan, x: torch.Tensor, lens: Optional[torch.Tensor]) -> bool:
    """Long unmasked bf16 inputs (the 30-minute file, equal-length windows): enough rows for the 256-wide tiles to fill
    the chip; a ragged batch keeps the LayerNorm passes (they also apply the padding masks)."""
    return (plan.lnf is not None and lens is None and x.dtype == torch.bfloat16 and x.is_cuda
            and x.numel() // x.shape[-1] >= _LN_FOLD_MIN_ROWS)


def layer_forward_lnfold(plan: LayerPlan, x: torch.Tensor, st: torch.Tensor, next_norm: Optional[nn.LayerNorm], next_fold: bool):
    """layer_forward for the long unmasked bf16 case with the three projection-feeding pre-norms folded into their GEMMs:
    the residual GEMM that produces a row also writes its (sum, sum of squares), the projection that follows reads the
    UN-normalised row and normalises in its epilogue -- norm_ff_macaron(x), norm_conv(x), norm_ff(x) never exist in memory
    (two of the layer's seven LayerNorm passes, and norm_final's second output, are gone).
    x: residual stream, st: its row statistics (rows, 8, 2).  Returns (layer output, its statistics or next_norm(output))."""
    L, F = plan.layer, plan.lnf
    B, T, C = x.shape
    M = B * T
    cm = L.conv_module
    G = hip_ops.gemm_bf16_ln
    new_stats = lambda: torch.empty((M, 8, 2), dtype=torch.float32, device=x.device)
    x2 = x.view(M, C)
    w, b, cs = F["ffm"]
    rb = _FFN_ROW_BLOCK
    if rb > 0:
        # experiment (PAFC_FFN_ROW_BLOCK=<rows>): w_1 -> w_2 per row block, so that the block's hidden slice (rows x 2048 bf16) is
        # still in the 256 MiB Infinity Cache when w_2 reads it back.  Measured and NOT the default: DESIGN section 4, round 5
        xn = torch.empty_like(x2)
        for r0 in range(0, M, rb):
            r1 = min(M, r0 + rb)
            hb = G(x2[r0:r1], w, b, st[r0:r1], act="silu", csum=cs, eps=L.norm_ff_macaron.eps)
            hip_ops.gemm_bf16(hb, L.feed_forward_macaron.w_2.weight, plan.b2_macaron, "none", alpha=L.ff_scale, residual=x2[r0:r1], out=xn[r0:r1])
        x2 = xn
    else:
        hid = G(x2, w, b, st, act="silu", csum=cs, eps=L.norm_ff_macaron.eps)
        x2 = hip_ops.gemm_bf16(hid, L.feed_forward_macaron.w_2.weight, plan.b2_macaron, "none", alpha=L.ff_scale, residual=x2)
    _, h, _ = hip_ops.add_layernorm(x2.view(B, T, C), None, 1.0, L.norm_mha.weight, L.norm_mha.bias, want_x=False, eps=L.norm_mha.eps)
    st2 = new_stats()
    slot_forward(plan, h, residual=x2.view(B, T, C), stats=st2)
    w, b, cs = F["pw1"]
    p = G(x2, w, b, st2, act="glu", csum=cs, eps=L.norm_conv.eps).view(B, T, C)
    g = _dwconv_norm_silu(cm, p, (cm.kernel_size - 1) // 2, T)
    st3 = new_stats()
    G(g.view(M, C), cm.pointwise_conv2.weight.squeeze(-1), cm.pointwise_conv2.bias, st3, residual=x2, out=x2)
    w, b, cs = F["ff"]
    if rb > 0:
        for r0 in range(0, M, rb):
            r1 = min(M, r0 + rb)
            hb = G(x2[r0:r1], w, b, st3[r0:r1], act="silu", csum=cs, eps=L.norm_ff.eps)
            hip_ops.gemm_bf16(hb, L.feed_forward.w_2.weight, plan.b2, "none", alpha=L.ff_scale, residual=x2[r0:r1], out=x2[r0:r1])
    else:
        hid = G(x2, w, b, st3, act="silu", csum=cs, eps=L.norm_ff.eps)
        hip_ops.gemm_bf16(hid, L.feed_forward.w_2.weight, plan.b2, "none", alpha=L.ff_scale, residual=x2, out=x2)
    xo = x2.view(B, T, C)
    if next_fold:       # the next layer's first pre-norm is folded too: it wants this layer's output and its statistics
        stn = new_stats()
        _, out, _ = hip_ops.add_layernorm(xo, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, eps=L.norm_final.eps,
                                          stats_out1=stn)
        return out, stn
    if next_norm is not None and next_norm.eps != L.norm_final.eps:
        _, out, _ = hip_ops.add_layernorm(xo, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, eps=L.norm_final.eps)
        _, hn, _ = hip_ops.add_layernorm(out, None, 1.0, next_norm.weight, next_norm.bias, want_x=False, eps=next_norm.eps)
        return out, hn
    _, out, hn = hip_ops.add_layernorm(xo, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False,
                                       gamma2=next_norm.weight if next_norm is not None else None,
                                       beta2=next_norm.bias if next_norm is not None else None, eps=L.norm_final.eps)
    return out, hn


_FFN_ROW_BLOCK = int(os.environ.get("PAFC_FFN_ROW_BLOCK", "0"))


# fp32 streams of a model with the bf16 slot shorter than this take exact fp32 products (csrc/gemm_f32.hip); from here on the
# split-operand schedule (small tiles up to hip_ops.DISPATCH["split_small_max_rows"] rows, the 256-wide kernel beyond)
_SPLIT_GEMM_MIN_ROWS = hip_ops.DISPATCH["split_layers_min_rows"]


def split_eligible(plan: LayerPlan, x: torch.Tensor) -> bool:
    return (plan.split is not None and x.dtype == torch.float32 and x.is_cuda
            and x.numel() // x.shape[-1] >= _SPLIT_GEMM_MIN_ROWS)


def layer_forward_split(plan: LayerPlan, x: torch.Tensor, hp: torch.Tensor, lens: Optional[torch.Tensor],
                        next_norm: Optional[nn.LayerNorm], next_split: bool):
    """layer_forward for an fp32 layer around the bf16 slot, on the bf16 matrix cores: every fp32 activation that feeds a
    projection travels as bf16 planes [hi | lo] (written by the kernel that produces it: LayerNorm passes and the w_1 GEMM),
    every projection is the split-operand GEMM with its bias / SiLU / GLU / residual epilogue, the residual stream stays fp32.
    hp = planes of norm_ff_macaron(x).  Returns (layer output fp32, next_norm(output) as planes / fp32 / None)."""
    L, S = plan.layer, plan.split
    B, T, C = x.shape
    M = B * T
    masked = lens is not None
    cm = L.conv_module
    G = hip_ops.gemm_ph_ex

    def ffn(ff, w1, w2, b2_scaled, hpl, xr, inplace):
        hid = G(hpl.view(M, 2 * C), w1, ff.w_1.bias, "silu", a_split=True, out_kind="planes")
        out = xr.view(M, C) if inplace else torch.empty((M, C), dtype=torch.float32, device=xr.device)
        return G(hid, w2, b2_scaled, alpha=L.ff_scale, residual=xr.view(M, C), out=out, a_split=True, out_kind="f32").view(B, T, C)

    x = ffn(L.feed_forward_macaron, S["ffm_w1"], S["ffm_w2"], plan.b2_macaron, hp, x, False)
    _, h, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_mha.weight, L.norm_mha.bias, out_dtype=torch.bfloat16, want_x=False,
                                    eps=L.norm_mha.eps)
    x = slot_forward(plan, h, residual=x)
    _, hc, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_conv.weight, L.norm_conv.bias, zero_rows=masked, lens=lens, T=T,
                                     want_x=False, eps=L.norm_conv.eps, split1=True)
    left_pad, Tc = (cm.kernel_size - 1) // 2, T
    if cm.lorder > 0:    # causal module: lorder zero frames in front of pointwise_conv1 (convolution.py:112-118)
        hc = torch.cat([hc.new_zeros(B, cm.lorder, 2 * C), hc], dim=1)
        left_pad, Tc = 0, T + cm.lorder
    if B * Tc <= hip_ops._SPLIT_SMALL_MAX_ROWS and B * Tc * 2 * C <= (1 << 22):
        # few rows (the small-tile kernel has no GLU epilogue): the plain projection, F.glu inside the depthwise convolution
        p = G(hc.view(B * Tc, 2 * C), S["pw1_plain"], cm.pointwise_conv1.bias, a_split=True, out_kind="f32").view(B, Tc, 2 * C)
        dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, left_pad, T, glu=True)
    else:
        p = G(hc.view(B * Tc, 2 * C), S["pw1"], S["pw1_b"], "glu", a_split=True, out_kind="f32").view(B, Tc, C)
        dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, left_pad, T)
    _, g, _ = hip_ops.add_layernorm(dw, None, 1.0, cm.norm.weight, cm.norm.bias, silu=True, eps=cm.norm.eps, split1=True)
    if not masked:
        x2 = x.view(M, C)
        G(g.view(M, 2 * C), S["pw2"], cm.pointwise_conv2.bias, residual=x2, out=x2, a_split=True, out_kind="f32")
        _, h2, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_ff.weight, L.norm_ff.bias, want_x=False, eps=L.norm_ff.eps, split1=True)
    else:   # padded frames of the conv branch count as zero (convolution.py:140-141): the add stays in the norm pass
        c = G(g.view(M, 2 * C), S["pw2"], cm.pointwise_conv2.bias, a_split=True, out_kind="f32").view(B, T, C)
        x, h2, _ = hip_ops.add_layernorm(x, c, 1.0, L.norm_ff.weight, L.norm_ff.bias, lens=lens, T=T, mask_y=True,
                                         eps=L.norm_ff.eps, split1=True)
    x = ffn(L.feed_forward, S["ff_w1"], S["ff_w2"], plan.b2, h2, x, True)
    if next_norm is not None and next_norm.eps != L.norm_final.eps:      # the one-pass pair shares one epsilon
        _, out, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, eps=L.norm_final.eps)
        _, hn, _ = hip_ops.add_layernorm(out, None, 1.0, next_norm.weight, next_norm.bias, want_x=False, eps=next_norm.eps,
                                         split1=next_split)
        return out, hn
    _, out, hn = hip_ops.add_layernorm(x, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False,
                                       gamma2=next_norm.weight if next_norm is not None else None,
                                       beta2=next_norm.bias if next_norm is not None else None, eps=L.norm_final.eps,
                                       split2=next_split and next_norm is not None)
    return out, hn


def carry_eligible(layer: nn.Module, x: torch.Tensor) -> bool:
    """State-carrying chunk step through the fused kernels: uni-directional bf16 slot, causal conv; B concurrent streams
    of equal chunk length (B = 1 is the reference's forward_chunk contract, B > 1 is B independent streams per step)."""
    cm = layer.conv_module
    return (cm is not None and type(layer.self_attn) is RWKV_TmixWrapper and layer.normalize_before
            and layer.feed_forward_macaron is not None and cm.use_layer_norm and cm.lorder > 0
            and isinstance(cm.activation, nn.SiLU) and isinstance(layer.feed_forward.activation, nn.SiLU)
            and isinstance(layer.feed_forward_macaron.activation, nn.SiLU) and cm.kernel_size <= 31
            and layer.size % 64 == 0 and layer.size <= 1024 and x.is_cuda and x.dtype == torch.bfloat16
            and bool(layer.self_attn.do_bfloat16))


def layer_forward_carry(plan: LayerPlan, x: torch.Tensor, carry: Optional[dict], h0: Optional[torch.Tensor] = None,
                        next_norm: Optional[nn.LayerNorm] = None, in_place: bool = False, pending: Optional[list] = None):
    """ConformerEncoderLayer.forward_carry (one chunk with recurrent-state carry) on the fused kernels, bf16, B streams:
    the same carries -- "shift" (B, 1, C) last normalised frame, "wkv" float32 (B, H, N, N), "cnn" (B, C, lorder) --
    and the same arithmetic as the module path.  A chunk step is launch-bound (a 64-frame chunk is ~5 us of work per kernel),
    so the step is built to be FEW launches: the frame before the chunk reaches the token-shift kernels through their `prev`
    pointer (no concatenation, no extra row to drop again), the scan starts from the carried state and -- `in_place`, the
    captured step of `stream_chunks`, whose carries are fixed buffers -- writes the new state over it, "shift" and "cnn" are
    refreshed by one small copy each, and `norm_final` + the next layer's first pre-norm are one pass (h0 = that pre-norm of x
    when the previous layer already produced it; next_norm = the norm to apply to this layer's output for the next one).
    `pending` (with in_place): instead of copying, the (destination, source) pairs of the small carry refreshes are appended
    to it and the caller performs them all in ONE multi-tensor copy after the last layer.  A carry "cx" (B, lorder + T, C) --
    set up by stream_chunks for one stream -- is the conv module's input buffer kept across steps: norm_conv writes the chunk
    behind the cached rows (no concatenation) and the refresh moves the tail to the front.
    Returns (layer output, carries, next_norm(output) or None)."""
    L = plan.layer
    carry = carry if carry is not None else {}
    B, T, C = x.shape
    M = B * T
    x = x.contiguous()
    if h0 is None:
        _, h0, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_ff_macaron.weight, L.norm_ff_macaron.bias, want_x=False, eps=L.norm_ff_macaron.eps)
    x = _ffn_residual(L.feed_forward_macaron, h0, x, L.ff_scale, plan.b2_macaron, inplace=False)
    _, h, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_mha.weight, L.norm_mha.bias, want_x=False, eps=L.norm_mha.eps)
    shift = carry.get("shift")
    if shift is None:
        shift = h.new_zeros(B, 1, C)
    elif shift.dtype != h.dtype or not shift.is_contiguous():
        shift = shift.to(h.dtype).contiguous()
    few = True                                           # the chunk step is launch-bound: one kernel each for the two LoRA chains
    sk = h.dtype == torch.bfloat16 and hip_ops.skinny_ok(M, C, C)     # a handful of rows: the few-rows GEMM and its fusions
    if sk:   # token shift + lerp as the operand producer of the down-projection (one ~5 us launch)
        t = hip_ops.gemm_skinny(h.view(M, C), plan.W1n[0], None, "tanh", mix_maa=plan.maa_x_n[0], mix_prev=shift, mix_T=T).view(1, M, -1)
    else:
        t = hip_ops.tmix_lora_down(h, plan.maa_x_n, plan.W1n, prev=shift, one_pass=few)
    z = hip_ops.tmix_lora_mix4(h, t, plan.W2t, plan.maa4, prev=shift)                         # (4, 1, M, C)
    if hip_ops.skinny_ok(M, C, C):
        rkv = hip_ops.gemm_skinny(z[:3].view(3, M, C), plan.Wrkv_n).view(3, B, T, C)
    elif M >= _OWN_GEMM_MIN_ROWS:
        rkv = hip_ops.gemm_bf16(z[:3].view(3, M, C), plan.Wrkv_n).view(3, B, T, C)
    else:
        rkv = torch.bmm(z[:3].view(3, M, C), plan.Wrkv).view(3, B, T, C)
    # (the decay chain and the r / k / v projections are independent, but as two branches of the captured graph -- a second HIP
    # stream forked and joined by events -- the step got 25 % SLOWER, 1.33 -> 1.66 ms: cross-queue dependencies cost more than
    # the 5 us they hide; one stream)
    if sk and plan.D1n.shape[1] == 64:   # one short launch: bf16(tanh(z_w D1)) in LDS, then bf16(. D2) + time_decay, rounded
        w = hip_ops.decay_lora_skinny(z[3].view(M, C), plan.D1n[0], plan.D2n[0], plan.time_decay.view(C)).view(B, T, C)   # as the op chain rounds
    elif sk:
        td = hip_ops.gemm_skinny(z[3].view(M, C), plan.D1n[0], None, "tanh")
        w = hip_ops.gemm_skinny(td, plan.D2n[0], plan.time_decay.view(C), round_first=True).view(B, T, C)
    else:
        w = hip_ops.decay_lora(z[3].view(1, M, C), plan.D1n, plan.D2n, plan.time_decay.view(1, C), one_pass=few).view(B, T, C)
    s_in = carry.get("wkv")
    new = carry if in_place else {}
    if in_place and s_in is not None and hip_ops.wkv6_single_chunk(B, T, C, plan.u[0].shape[0]):
        y, _ = wkv6_forward(rkv[0], rkv[1], rkv[2], w, plan.u[0], s_in=s_in, s_out=s_in)      # the state is updated where it lies
    else:
        y, s_out = wkv6_forward(rkv[0], rkv[1], rkv[2], w, plan.u[0], s_in=s_in, want_state=True)
        if in_place and s_in is not None:
            s_in.copy_(s_out)
        else:
            new["wkv"] = s_out
    if in_place and carry.get("shift") is not None and carry["shift"].dtype == h.dtype:
        if pending is not None:
            pending.append((carry["shift"], h[:, -1:]))
        else:
            carry["shift"].copy_(h[:, -1:])           # (after the two passes that read the old one)
    else:
        new["shift"] = h[:, -1:].contiguous()
    ln = plan.blocks[0].ln_x
    if sk:   # ln_x folded into the output projection
        wo, bo, cso, epo = plan.carry_folds()["out"]
        x2 = x.view(M, C)
        x = hip_ops.gemm_skinny(y.view(M, C), wo, bo, residual=x2, out=x2, ln_self=True, ln_csum=cso, ln_eps=epo).view(B, T, C)
    else:
        _, yn, _ = hip_ops.add_layernorm(y.view(M, C), None, 1.0, ln.weight, ln.bias, eps=ln.eps, want_x=False)
        x = proj(yn, plan.Wo, None, "none", residual=x.view(M, C), inplace=True).view(B, T, C)
    cm = L.conv_module
    cnn = carry.get("cnn")
    cxb = carry.get("cx") if in_place else None
    if cxb is not None and B == 1 and cxb.shape == (1, cm.lorder + T, C) and cxb.dtype == x.dtype and T >= cm.lorder:
        cx = cxb                                      # rows [0, lorder) = the cache, the chunk's rows go behind them
        hip_ops.add_layernorm(x, None, 1.0, L.norm_conv.weight, L.norm_conv.bias, want_x=False, eps=L.norm_conv.eps,
                              out1=cx[0, cm.lorder:])
        cnn = None                                    # ("cnn" is rebuilt from cx by stream_chunks when the captured steps end)
    else:
        _, hc, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_conv.weight, L.norm_conv.bias, want_x=False, eps=L.norm_conv.eps)
        left = cnn.transpose(1, 2).to(hc.dtype) if cnn is not None and cnn.numel() > 0 else hc.new_zeros(B, cm.lorder, C)
        cx = torch.cat([left, hc], dim=1)                                                     # (B, lorder + T, C)
    pw1 = cm.pointwise_conv1
    if _skinny(cx, pw1.weight.view(2 * C, C), glu=True):
        p = hip_ops.gemm_skinny(cx.view(-1, C), pw1.weight.view(2 * C, C), pw1.bias, "glu").view(B, -1, C)
        dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, 0, T)
    elif plan.pw1_glu is not None:
        p = _pw1_glu(plan, cx.view(-1, C)).view(B, -1, C)
        dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, 0, T)
    else:
        p = F.linear(cx, cm.pointwise_conv1.weight.squeeze(-1), cm.pointwise_conv1.bias)
        dw = hip_ops.depthwise_conv1d_cl(p, cm.depthwise_conv.weight, cm.depthwise_conv.bias, 0, T, glu=True)
    if cx is cxb:                                     # (after pointwise_conv1 has read the old cache rows)
        if pending is not None:
            pending.append((cx[:, :cm.lorder], cx[:, T:]))
        else:
            cx[:, :cm.lorder].copy_(cx[:, T:])
    else:
        new_cnn = cx[:, -cm.lorder:, :].transpose(1, 2)
        if in_place and cnn is not None and cnn.shape == new_cnn.shape and cnn.dtype == new_cnn.dtype:
            cnn.copy_(new_cnn)
        else:
            new["cnn"] = new_cnn
    pw2 = cm.pointwise_conv2
    if sk and dw.dtype == torch.bfloat16 and cm.norm.weight.dtype == torch.bfloat16 and pw2.weight.is_contiguous():
        # the conv module's LayerNorm + SiLU as the operand producer of pointwise_conv2 (one launch instead of two)
        x2 = x.view(M, C)
        x = hip_ops.gemm_skinny(dw.view(M, C), pw2.weight.view(C, C), pw2.bias, residual=x2, out=x2,
                                norm_silu=(cm.norm.weight, cm.norm.bias, cm.norm.eps)).view(B, T, C)
    else:
        _, g, _ = hip_ops.add_layernorm(dw, None, 1.0, cm.norm.weight, cm.norm.bias, silu=True, eps=cm.norm.eps)
        x = proj(g, pw2.weight.squeeze(-1), pw2.bias, "none", residual=x, inplace=True)
    if sk and hip_ops.skinny_ok(M, L.feed_forward.w_1.weight.shape[0], C):     # norm_ff folded into w_1
        w1, b1, cs1, ep1 = plan.carry_folds()["ff"]
        hid = hip_ops.gemm_skinny(x.view(M, C), w1, b1, "silu", ln_self=True, ln_csum=cs1, ln_eps=ep1)
        x = proj(hid, L.feed_forward.w_2.weight, plan.b2, "none", alpha=L.ff_scale, residual=x.view(M, C), inplace=True).view(B, T, C)
    else:
        _, h2, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_ff.weight, L.norm_ff.bias, want_x=False, eps=L.norm_ff.eps)
        x = _ffn_residual(L.feed_forward, h2, x, L.ff_scale, plan.b2, inplace=True)
    if next_norm is not None and next_norm.eps == L.norm_final.eps:        # the one-pass pair shares one epsilon
        _, out, hn = hip_ops.add_layernorm(x, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, gamma2=next_norm.weight,
                                           beta2=next_norm.bias, eps=L.norm_final.eps)
        return out, new, hn
    _, out, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, eps=L.norm_final.eps)
    hn = None
    if next_norm is not None:
        _, hn, _ = hip_ops.add_layernorm(out, None, 1.0, next_norm.weight, next_norm.bias, want_x=False, eps=next_norm.eps)
    return out, new, hn


def lookahead_eligible(layer: nn.Module, x: torch.Tensor) -> bool:
    """The steady-state look-ahead chunk step (non-causal conv module, the shipped uni YAML) on the fused kernels: one bf16
    stream, uni-directional bf16 slot, pre-norm, layer_norm conv module with an odd kernel <= 31."""
    cm = layer.conv_module
    return (cm is not None and type(layer.self_attn) is RWKV_TmixWrapper and layer.normalize_before
            and layer.feed_forward_macaron is not None and cm.use_layer_norm and cm.lorder == 0 and cm.kernel_size % 2 == 1
            and isinstance(cm.activation, nn.SiLU) and isinstance(layer.feed_forward.activation, nn.SiLU)
            and isinstance(layer.feed_forward_macaron.activation, nn.SiLU) and cm.kernel_size <= 31
            and layer.size % 64 == 0 and layer.size <= 1024 and x.is_cuda and x.dtype == torch.bfloat16 and x.size(0) == 1
            and bool(layer.self_attn.do_bfloat16))


def layer_forward_lookahead(plan: LayerPlan, x: torch.Tensor, carry: dict, h0: Optional[torch.Tensor], next_norm: Optional[nn.LayerNorm],
                            pending: list):
    """ConformerEncoderLayer.forward_lookahead in its STEADY state -- T new frames in, T finalised frames out, 15 frames
    behind -- on the fused chunk-step kernels, one stream, every carry a fixed buffer updated where it lies (the form a captured
    hipGraph step needs):  carry["U"] (1, 2 half + T, C) = the depthwise convolution's input, the last 2 half frames of the
    previous steps in front and this step's GLU output written behind them; carry["X2"] (1, half + T, C) = the residual stream
    behind the slot, the `half` frames not yet emitted in front; "shift" (1, 1, C), "wkv" (1, H, N, N).  The small refreshes
    (tails to the front, last frame to "shift") are appended to `pending` as (destination, source) pairs: ONE multi-tensor
    copy after the last layer performs them (T >= 2 half, so no pair overlaps itself).
    Returns (layer output for the T emitted frames, next_norm(output) or None)."""
    L = plan.layer
    cm = L.conv_module
    B, T, C = x.shape
    half = (cm.kernel_size - 1) // 2
    U, X2, shift, s_in = carry["U"], carry["X2"], carry["shift"], carry["wkv"]
    assert B == 1 and U.shape == (1, 2 * half + T, C) and X2.shape == (1, half + T, C) and T >= 2 * half
    M = T
    x = x.contiguous()
    if h0 is None:
        _, h0, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_ff_macaron.weight, L.norm_ff_macaron.bias, want_x=False, eps=L.norm_ff_macaron.eps)
    x = _ffn_residual(L.feed_forward_macaron, h0, x, L.ff_scale, plan.b2_macaron, inplace=False)
    _, h, _ = hip_ops.add_layernorm(x, None, 1.0, L.norm_mha.weight, L.norm_mha.bias, want_x=False, eps=L.norm_mha.eps)
    # slot: token shift + lerp as the down-projection's operand, LoRA-up + lerps, r / k / v, decay LoRA, scan from / into the state
    t = hip_ops.gemm_skinny(h.view(M, C), plan.W1n[0], None, "tanh", mix_maa=plan.maa_x_n[0], mix_prev=shift, mix_T=T).view(1, M, -1)
    z = hip_ops.tmix_lora_mix4(h, t, plan.W2t, plan.maa4, prev=shift)
    rkv = hip_ops.gemm_skinny(z[:3].view(3, M, C), plan.Wrkv_n).view(3, B, T, C)
    if plan.D1n.shape[1] == 64:
        w = hip_ops.decay_lora_skinny(z[3].view(M, C), plan.D1n[0], plan.D2n[0], plan.time_decay.view(C)).view(B, T, C)
    else:
        td = hip_ops.gemm_skinny(z[3].view(M, C), plan.D1n[0], None, "tanh")
        w = hip_ops.gemm_skinny(td, plan.D2n[0], plan.time_decay.view(C), round_first=True).view(B, T, C)
    if hip_ops.wkv6_single_chunk(B, T, C, plan.u[0].shape[0]):
        y, _ = wkv6_forward(rkv[0], rkv[1], rkv[2], w, plan.u[0], s_in=s_in, s_out=s_in)
    else:
        y, s_out = wkv6_forward(rkv[0], rkv[1], rkv[2], w, plan.u[0], s_in=s_in, want_state=True)
        s_in.copy_(s_out)
    pending.append((shift, h[:, -1:]))
    # ln_x folded into the output projection; the new rows of the residual stream go BEHIND the frames still waiting in X2
    wo, bo, cso, epo = plan.carry_folds()["out"]
    x2new = X2[0, half:]
    hip_ops.gemm_skinny(y.view(M, C), wo, bo, residual=x.view(M, C), out=x2new, ln_self=True, ln_csum=cso, ln_eps=epo)
    # conv branch, front half: norm_conv -> pointwise_conv1 + GLU, written behind the cached rows of U
    _, hc, _ = hip_ops.add_layernorm(x2new.view(1, T, C), None, 1.0, L.norm_conv.weight, L.norm_conv.bias, want_x=False, eps=L.norm_conv.eps)
    pw1 = cm.pointwise_conv1
    hip_ops.gemm_skinny(hc.view(M, C), pw1.weight.view(2 * C, C), pw1.bias, "glu", out=U[0, 2 * half:])
    # back half, on the T frames whose window is now complete (centres = rows [half, half + T) of U = rows [0, T) of X2)
    dw = hip_ops.depthwise_conv1d_cl(U, cm.depthwise_conv.weight, cm.depthwise_conv.bias, 0, T)
    pw2 = cm.pointwise_conv2
    x3 = hip_ops.gemm_skinny(dw.view(M, C), pw2.weight.view(C, C), pw2.bias, residual=X2[0, :T],
                             norm_silu=(cm.norm.weight, cm.norm.bias, cm.norm.eps))
    pending.append((U[:, :2 * half], U[:, T:T + 2 * half]))
    pending.append((X2[:, :half], X2[:, T:T + half]))
    w1, b1, cs1, ep1 = plan.carry_folds()["ff"]
    hid = hip_ops.gemm_skinny(x3, w1, b1, "silu", ln_self=True, ln_csum=cs1, ln_eps=ep1)
    x4 = proj(hid, L.feed_forward.w_2.weight, plan.b2, "none", alpha=L.ff_scale, residual=x3, inplace=True).view(B, T, C)
    if next_norm is not None and next_norm.eps == L.norm_final.eps:
        _, out, hn = hip_ops.add_layernorm(x4, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, gamma2=next_norm.weight,
                                           beta2=next_norm.bias, eps=L.norm_final.eps)
        return out, hn
    _, out, _ = hip_ops.add_layernorm(x4, None, 1.0, L.norm_final.weight, L.norm_final.bias, want_x=False, eps=L.norm_final.eps)
    hn = None
    if next_norm is not None:
        _, hn, _ = hip_ops.add_layernorm(out, None, 1.0, next_norm.weight, next_norm.bias, want_x=False, eps=next_norm.eps)
    return out, hn


class EncoderPlan:
    def __init__(self, encoders: nn.ModuleList):
        self.layers = [LayerPlan(l) for l in encoders]

    def refresh(self):
        for p in self.layers:
            p.refresh()


def unmasked_schedule_possible(plan: EncoderPlan, xs: torch.Tensor) -> bool:
    """Could this (B, T', C) input take the schedule without padding masks if every row turned out to be full length?  (Long
    inputs only: the decision costs one host read of the lengths.)"""
    return (xs.numel() // xs.shape[-1] >= _LN_FOLD_MIN_ROWS
            and (all(lnfold_eligible(lp, xs, None) for lp in plan.layers) or all(split_eligible(lp, xs) for lp in plan.layers)))


def encoder_layers_forward(plan: EncoderPlan, xs: torch.Tensor, masks: torch.Tensor, after_norm: Optional[nn.LayerNorm],
                           want_layers: bool = False, all_full: Optional[bool] = None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """The `for layer in self.encoders` loop of BaseEncoder.forward (+ after_norm), encoder.py:141-146.
    masks: (B, 1, T') prefix masks from make_pad_mask, or a (0,0,0) fake mask (forward_chunk).
    all_full: the caller already knows whether every row is full length (the hipGraph cache reads the lengths BEFORE it captures
    and keys the graph on the answer, so that a replay runs the schedule the eager pass of the same batch runs); None = find out
    here, with one host read, when the unmasked schedule is possible at all (never while capturing)."""
    plan.refresh()
    lens = masks.squeeze(1).sum(1).to(torch.int32) if masks.numel() > 0 else None
    xs = xs.contiguous()
    first = plan.layers[0].layer.norm_ff_macaron
    outs: List[torch.Tensor] = []
    n = len(plan.layers)
    if lens is not None and unmasked_schedule_possible(plan, xs) and (
            all_full if all_full is not None else
            (not torch.cuda.is_current_stream_capturing() and int(lens.min()) == xs.shape[1])):
        # a long input whose rows are all full length (the 30-minute file, B = 1): the padding masks are no-ops -- worth one
        # host read of the lengths per pass to take the unmasked schedule (bf16: the folded LayerNorms; fp32 around the bf16
        # slot: pointwise_conv2's residual add rides on the GEMM instead of a second operand of the LayerNorm pass)
        lens = None
    if all(lnfold_eligible(lp, xs, lens) for lp in plan.layers):
        st = torch.empty((xs.numel() // xs.shape[-1], 8, 2), dtype=torch.float32, device=xs.device)
        hip_ops.add_layernorm(xs, None, 1.0, first.weight, first.bias, want_ln=False, stats_x=st)    # statistics only
        h = st
        for i, lp in enumerate(plan.layers):
            nxt = plan.layers[i + 1].layer.norm_ff_macaron if i + 1 < n else after_norm
            xs, h = layer_forward_lnfold(lp, xs, h, nxt, i + 1 < n)
            if want_layers:
                outs.append(xs)
        if after_norm is not None:
            xs = h
        return xs, outs
    split = [split_eligible(lp, xs) for lp in plan.layers] + [False]   # (after_norm's output goes to the caller: fp32)
    _, h, _ = hip_ops.add_layernorm(xs, None, 1.0, first.weight, first.bias, eps=first.eps, split1=split[0])
    for i, lp in enumerate(plan.layers):
        nxt = plan.layers[i + 1].layer.norm_ff_macaron if i + 1 < n else after_norm
        if split[i]:
            xs, h = layer_forward_split(lp, xs, h, lens, nxt, split[i + 1])
        else:
            xs, h = layer_forward(lp, xs, h, lens, nxt)
            if split[i + 1]:
                h = hip_ops.split_planes(h.contiguous())
        if want_layers:
            outs.append(xs)
    if after_norm is not None:
        xs = h
    return xs, outs
