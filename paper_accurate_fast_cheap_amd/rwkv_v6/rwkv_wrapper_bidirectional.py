"""Bidirectional attention-slot adapter: the paper's headline layer.

Reference: wenet/rwkv_v6/rwkv_wrapper_bidirectional.py:7-64 (registry key rwkv_tmix60_bidirectional) and
rwkv_wrapper_bidirectional2.py:95-150 (key rwkv_tmix60_bidirectional2: same arithmetic with a persistent
flip buffer).  out = (Tmix_f(x) + flip(Tmix_b(flip(x)))) / 2, the flip being over the whole PADDED length, so
for a short utterance the right-to-left state is warmed by its padding; cast to bf16 on the way in and to
fp32 on the way out when do_bfloat16; `cache` returned untouched.

Here nothing is flipped: the right-to-left block shifts tokens the other way and its scan walks t downward,
and both directions' scans are ONE kernel launch over one grid."""
from typing import Optional, Tuple

import torch

from .rwkv_wrapper import RWKV_TmixWrapper, _EMPTY_CACHE
from .wkv6_op import wkv6, wkv6_forward_bidir


class RWKV_TmixWrapper_bidirectional(torch.nn.Module):
    def __init__(self, head_size: int, dim_att: int, num_blocks: int, rnn_att_version: str, rnn_att_direction: str,
                 ctx_len: int = 2048, do_bfloat16: bool = True, layer_id: int = 1):
        super().__init__()
        self.do_bfloat16 = do_bfloat16
        self.layer_id = layer_id
        self.rwkv_wrapper_forward = RWKV_TmixWrapper(head_size, dim_att, num_blocks, rnn_att_version,
                                                     rnn_att_direction, ctx_len, do_bfloat16, layer_id)
        self.rwkv_wrapper_backward = RWKV_TmixWrapper(head_size, dim_att, num_blocks, rnn_att_version,
                                                      rnn_att_direction, ctx_len, do_bfloat16, layer_id)
        # the casts happen once, here (rwkv_wrapper_bidirectional.py:24-28); the parameters stay bf16
        self.rwkv_wrapper_forward.do_bfloat16 = False
        self.rwkv_wrapper_backward.do_bfloat16 = False

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, pos_emb: Optional[torch.Tensor] = None,
                cache: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        query_dtype = query.dtype
        x = query.to(dtype=torch.bfloat16) if self.do_bfloat16 else query
        blk_f = self.rwkv_wrapper_forward.tmix_block
        blk_b = self.rwkv_wrapper_backward.tmix_block
        pf = blk_f.mix_project(x, reverse=False)
        pb = blk_b.mix_project(x, reverse=True)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            yf = wkv6(*pf, blk_f.time_faaaa, False)
            yb = wkv6(*pb, blk_b.time_faaaa, True)
        else:
            yf, yb = wkv6_forward_bidir((*pf, blk_f.time_faaaa.contiguous()), (*pb, blk_b.time_faaaa.contiguous()))
        out = (blk_f.finish(yf) + blk_b.finish(yb)) / 2
        if self.do_bfloat16:
            # reference: out.float() (rwkv_wrapper_bidirectional.py:55-56).  Identical for an fp32 residual stream;
            # for a whole-model-bf16 encoder (`encoder-rtf.py --bf16`) the reference then crashes in the next
            # LayerNorm, so returning the query dtype is the only runnable reading (DESIGN.md "precision modes").
            out = out.to(query_dtype)
        return out, (cache if cache is not None else _EMPTY_CACHE.to(query.device))
