"""Positional encodings of the encoder front end (reference: wenet/transformer/embedding.py:25-147).

The recurrent slot ignores pos_emb (rwkv_wrapper.py:57-83), so the only arithmetic that reaches the output
is the x * sqrt(d_model) scale of RelPositionalEncoding.forward (embedding.py:144).  The sinusoid table is
still provided, lazily and on the input's device, for callers that ask for it (forward_chunk builds one).
The reference materialises a 100000 x d table per module at construction (204 MB at d = 512) and moves it
at every call (embedding.py:39,47-56,143); here a row range is generated on demand."""
import math
from typing import Tuple, Union

import torch


class RelPositionalEncoding(torch.nn.Module):
    def __init__(self, d_model: int, dropout_rate: float, max_len: int = 100000):
        super().__init__()
        self.d_model = d_model
        self.xscale = math.sqrt(d_model)
        self.dropout = torch.nn.Dropout(p=dropout_rate)
        self.max_len = max_len

    def position_encoding(self, offset: Union[int, torch.Tensor], size: int, apply_dropout: bool = True,
                          device=None) -> torch.Tensor:
        """Rows [offset, offset+size) of the sin/cos table (embedding.py:47-56,78-116), shape (1, size, d)."""
        if isinstance(offset, torch.Tensor):
            offset = int(offset.item())
        assert offset + size <= self.max_len
        pos = torch.arange(offset, offset + size, dtype=torch.float32, device=device).unsqueeze(1)
        div = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float32, device=device)
                        * -(math.log(10000.0) / self.d_model))
        pe = torch.zeros(size, self.d_model, device=device)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        pe = pe.unsqueeze(0)
        return self.dropout(pe) if apply_dropout else pe

    def forward(self, x: torch.Tensor, offset: Union[int, torch.Tensor] = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """(x * sqrt(d), pos_emb) -- embedding.py:133-147.  pos_emb is an empty placeholder: nothing on the
        recurrent path reads it; use position_encoding() to get the table."""
        x = x * self.xscale
        return self.dropout(x), x.new_zeros((1, 0, self.d_model))


class PositionalEncoding(RelPositionalEncoding):
    """abs_pos (embedding.py:25-77): x * sqrt(d) + pe."""

    def forward(self, x, offset=0):
        pe = self.position_encoding(offset, x.size(1), False, device=x.device).to(x.dtype)
        x = x * self.xscale + pe
        return self.dropout(x), self.dropout(pe)


class NoPositionalEncoding(torch.nn.Module):
    def __init__(self, d_model: int, dropout_rate: float):
        super().__init__()
        self.d_model = d_model
        self.dropout = torch.nn.Dropout(p=dropout_rate)

    def position_encoding(self, offset, size, apply_dropout=True, device=None):
        return torch.zeros(1, size, self.d_model, device=device)

    def forward(self, x, offset=0):
        return self.dropout(x), x.new_zeros((1, 0, self.d_model))
