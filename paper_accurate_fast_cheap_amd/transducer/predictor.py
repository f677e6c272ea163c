"""RNN-T prediction network (reference: wenet/transducer/predictor.py:60-206): embedding -> 2-layer LSTM ->
projection, with the per-step cache interface the beam search uses."""
from typing import List, Optional, Tuple

import torch
from torch import nn


def ApplyPadding(input, padding, pad_value) -> torch.Tensor:
    """predictor.py:8-14: padding * pad_value + input * (1 - padding)."""
    return padding * pad_value + input * (1 - padding)


class RNNPredictor(nn.Module):
    def __init__(self, voca_size: int, embed_size: int, output_size: int, embed_dropout: float, hidden_size: int,
                 num_layers: int, bias: bool = True, rnn_type: str = "lstm", dropout: float = 0.1) -> None:
        super().__init__()
        if rnn_type != "lstm":
            raise NotImplementedError("the paper's configs use rnn_type: lstm")
        self.n_layers = num_layers
        self.hidden_size = hidden_size
        self._output_size = output_size
        self.embed = nn.Embedding(voca_size, embed_size)
        self.dropout = nn.Dropout(embed_dropout)
        self.rnn = nn.LSTM(input_size=embed_size, hidden_size=hidden_size, num_layers=num_layers, bias=bias,
                           batch_first=True, dropout=dropout)
        self.projection = nn.Linear(hidden_size, output_size)

    def output_size(self):
        return self._output_size

    def init_state(self, batch_size: int, device: torch.device, method: str = "zero") -> List[torch.Tensor]:
        assert batch_size > 0
        return [torch.zeros(self.n_layers, batch_size, self.hidden_size, device=device),
                torch.zeros(self.n_layers, batch_size, self.hidden_size, device=device)]

    def forward(self, input: torch.Tensor, cache: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        embed = self.dropout(self.embed(input))
        if cache is None:
            state = self.init_state(batch_size=input.size(0), device=input.device)
            states = (state[0].to(embed.dtype), state[1].to(embed.dtype))
        else:
            assert len(cache) == 2
            states = (cache[0], cache[1])
        out, _ = self.rnn(embed, states)
        return self.projection(out)

    def batch_to_cache(self, cache: List[torch.Tensor]) -> List[List[torch.Tensor]]:
        assert len(cache) == 2 and cache[0].size(1) == cache[1].size(1)
        return [[m, c] for m, c in zip(torch.split(cache[0], 1, dim=1), torch.split(cache[1], 1, dim=1))]

    def cache_to_batch(self, cache: List[List[torch.Tensor]]) -> List[torch.Tensor]:
        return [torch.cat([s[0] for s in cache], dim=1), torch.cat([s[1] for s in cache], dim=1)]

    def forward_step(self, input: torch.Tensor, padding: torch.Tensor, cache: List[torch.Tensor]
                     ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """input (N, 1) token ids, padding (N, 1) (1 = keep the old state), cache [m, c] each (layers, N, H)."""
        assert len(cache) == 2
        state_m, state_c = cache[0], cache[1]
        embed = self.dropout(self.embed(input))
        out, (m, c) = self.rnn(embed, (state_m, state_c))
        out = self.projection(out)
        m = ApplyPadding(m, padding.unsqueeze(0), state_m)
        c = ApplyPadding(c, padding.unsqueeze(0), state_c)
        return out, [m, c]
