"""Direction-dropout variants of the bidirectional slot.

Reference: wenet/rwkv_v6/rwkv_wrapper_bidirectional_direction_dropout.py:7-97 (key rwkv_tmix60_dir_layer_drop)
and ..._direction_dropout_both.py (key rwkv_tmix60_dir_layer_drop_both).  Train time: with p = 0.2 one Bernoulli
draw per call drops the right-to-left branch ("both": drops one of the two, chosen by a second draw).  Eval
time: RWKV_BIDIRECTIONAL_LAYERS (comma list of layer ids that stay bidirectional) and RWKV_ALT_DECODING
(other layers alternate left-only / right-only by layer parity) are read from the environment AT CONSTRUCTION,
as in the reference (:25-33).  Unlike the plain bidirectional wrapper the two inner wrappers keep their own
bf16 cast, so the average is taken in the query dtype."""
import os
from typing import Optional, Tuple

import torch

from .rwkv_wrapper import RWKV_TmixWrapper, _EMPTY_CACHE


class RWKV_TmixWrapper_bidirectional_direction_dropout(torch.nn.Module):
    both = False
    p_drop = 0.2

    def __init__(self, head_size: int, dim_att: int, num_blocks: int, rnn_att_version: str, rnn_att_direction: str,
                 ctx_len: int = 2048, do_bfloat16: bool = True, layer_id: int = 1):
        super().__init__()
        self.layer_id = layer_id
        self.num_blocks = num_blocks
        self.bi_layers_actives = []
        self.bi_active = True
        self.alt_decoding = os.environ.get("RWKV_ALT_DECODING", "0") == "1"
        if os.environ.get("RWKV_BIDIRECTIONAL_LAYERS"):
            self.bi_layers_actives = [int(s) for s in os.environ["RWKV_BIDIRECTIONAL_LAYERS"].split(",")]
            self.bi_active = layer_id in self.bi_layers_actives
        self.rwkv_wrapper_forward = RWKV_TmixWrapper(head_size, dim_att, num_blocks, rnn_att_version,
                                                     rnn_att_direction, ctx_len, do_bfloat16, layer_id)
        self.rwkv_wrapper_backward = RWKV_TmixWrapper(head_size, dim_att, num_blocks, rnn_att_version,
                                                      rnn_att_direction, ctx_len, do_bfloat16, layer_id)

    def _l2r(self, x):
        return self.rwkv_wrapper_forward(x, reverse=False)[0]

    def _r2l(self, x):
        return self.rwkv_wrapper_backward(x, reverse=True)[0]

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, pos_emb: Optional[torch.Tensor] = None,
                cache: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        x = query
        if self.training:
            bern = torch.distributions.bernoulli.Bernoulli(1 - self.p_drop)
            keep = bern.sample((1,))
            if keep == 1:
                out = (self._l2r(x) + self._r2l(x)) / 2
            elif not self.both:
                out = self._l2r(x)
            else:
                out = self._l2r(x) if bern.sample((1,)) <= 0.5 else self._r2l(x)
        elif self.bi_active:
            out = (self._l2r(x) + self._r2l(x)) / 2
        elif self.alt_decoding and self.layer_id % 2 == 1:
            out = self._r2l(x)
        else:
            out = self._l2r(x)
        return out, (cache if cache is not None else _EMPTY_CACHE.to(query.device))


class RWKV_TmixWrapper_bidirectional_direction_dropout_both(RWKV_TmixWrapper_bidirectional_direction_dropout):
    both = True
