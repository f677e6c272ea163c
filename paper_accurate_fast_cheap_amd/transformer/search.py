"""CTC greedy search (reference: wenet/transformer/search.py:106-121, wenet/utils/ctc_utils.py:22-32).
Token ids are the bit-exact parity bar."""
from typing import List

import torch

from ..utils.mask import make_pad_mask


class DecodeResult:
    def __init__(self, tokens: List[int], score: float = 0.0, confidence: float = 0.0):
        self.tokens = tokens
        self.score = score
        self.confidence = confidence


def remove_duplicates_and_blank(hyp: List[int], blank_id: int = 0) -> List[int]:
    new_hyp: List[int] = []
    cur = 0
    while cur < len(hyp):
        if hyp[cur] != blank_id:
            new_hyp.append(hyp[cur])
        prev = cur
        while cur < len(hyp) and hyp[cur] == hyp[prev]:
            cur += 1
    return new_hyp


def ctc_greedy_search(ctc_probs: torch.Tensor, ctc_lens: torch.Tensor, blank_id: int = 0) -> List[DecodeResult]:
    batch_size, maxlen = ctc_probs.shape[:2]
    topk_index = ctc_probs.argmax(dim=2)  # == topk(1): ties resolve to the lowest index in both
    mask = make_pad_mask(ctc_lens, maxlen)
    topk_index = topk_index.masked_fill(mask, blank_id)
    hyps = topk_index.tolist()  # ONE device->host copy, then the reference's collapse per utterance
    return [DecodeResult(remove_duplicates_and_blank(h, blank_id)) for h in hyps]
