"""Uni-directional attention-slot adapter (reference: wenet/rwkv_v6/rwkv_wrapper.py:5-83).

Same positional constructor the encoder builds (encoder.py:553-561 + layer_id at :592-593) and the same
MHA-shaped forward(query, key, value, mask, pos_emb, cache) -> (x_att, new_att_cache).  key / value / mask /
pos_emb are accepted and ignored, exactly like the reference; `cache` is returned untouched."""
from typing import Optional, Tuple

import torch

from .tmix import RWKV_Tmix_x060c

_EMPTY_CACHE = torch.zeros((0, 0, 0, 0))


class RWKV_TmixWrapper(torch.nn.Module):
    def __init__(self, head_size: int, dim_att: int, num_blocks: int, rnn_att_version: Optional[str] = None,
                 rnn_att_direction: Optional[str] = None, ctx_len: int = 2048, do_bfloat16: bool = True,
                 layer_id: int = 1):
        super().__init__()
        self.head_size = head_size
        self.dim_att = dim_att
        self.num_blocks = num_blocks
        self.rnn_att_version = rnn_att_version
        self.rnn_att_direction = rnn_att_direction
        self.ctx_len = ctx_len  # kept for config compatibility; the HIP kernels have no T limit
        self.do_bfloat16 = do_bfloat16
        self.layer_id = layer_id
        self.n_head = dim_att // head_size
        self.n_embd = dim_att
        self.tmix_block = RWKV_Tmix_x060c(head_size=head_size, n_layers=num_blocks, n_embd=dim_att,
                                          dim_att=dim_att, layer_id=layer_id)
        if do_bfloat16:  # rwkv_wrapper.py:53-54: parameters are STORED in bf16
            self.tmix_block = self.tmix_block.to(dtype=torch.bfloat16)

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                mask: Optional[torch.Tensor] = None, pos_emb: Optional[torch.Tensor] = None,
                cache: Optional[torch.Tensor] = None, reverse: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        query_dtype = query.dtype
        if self.do_bfloat16:
            query = query.to(dtype=torch.bfloat16)
        x_att = self.tmix_block(query, reverse=reverse)
        if self.do_bfloat16:
            x_att = x_att.to(dtype=query_dtype)
        return x_att, (cache if cache is not None else _EMPTY_CACHE.to(query.device))
