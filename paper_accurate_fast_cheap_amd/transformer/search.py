"""CTC greedy and prefix beam search (reference: wenet/transformer/search.py:106-248,
wenet/utils/ctc_utils.py:22-32, wenet/utils/common.py:355-363).  Token ids are the bit-exact parity bar."""
import math
from collections import defaultdict
from typing import List, Optional

import torch

from ..utils.mask import make_pad_mask


class DecodeResult:
    def __init__(self, tokens: List[int], score: float = 0.0, confidence: float = 0.0, tokens_confidence=None,
                 times=None, nbest: Optional[List[List[int]]] = None, nbest_scores: Optional[List[float]] = None,
                 nbest_times=None):
        self.tokens = tokens
        self.score = score
        self.confidence = confidence
        self.tokens_confidence = tokens_confidence
        self.times = times
        self.nbest = nbest
        self.nbest_scores = nbest_scores
        self.nbest_times = nbest_times


def log_add(args: List[float]) -> float:
    """Stable log-sum-exp over python floats (common.py:355-363)."""
    if all(a == -float("inf") for a in args):
        return -float("inf")
    a_max = max(args)
    return a_max + math.log(sum(math.exp(a - a_max) for a in args))


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


def ctc_greedy_search(ctc_probs: torch.Tensor, ctc_lens: torch.Tensor, blank_id: int = 0, defer: bool = False):
    """search.py:106-121.  On the GPU the argmax, the padding rule and the collapse run in two kernels
    (``pafc_ctc_greedy``) and only the collapsed ids come back -- two small copies for the whole batch instead of a
    (B, T) copy and a Python loop per frame.  Host tensors take the reference's own steps below.
    defer (GPU only): return a zero-argument function that fetches the List[DecodeResult] -- the device work is queued
    now, the host waits for it only when the function is called, so several batches can be in flight."""
    if ctc_probs.is_cuda:
        from ..hip_ops import ctc_greedy
        tokens, ntok = ctc_greedy(ctc_probs.contiguous(), ctc_lens.to(ctc_probs.device), blank_id)
        if defer:       # nothing here may wait for the device (a boolean-mask gather would): whole rows come back later
            def fetch() -> List[DecodeResult]:
                rows, counts = tokens.tolist(), ntok.tolist()
                return [DecodeResult(r[:n]) for r, n in zip(rows, counts)]
            return fetch
        keep = torch.arange(tokens.shape[1], device=tokens.device)[None, :] < ntok[:, None]
        flat = tokens[keep].tolist()          # all utterances back to back
        counts = ntok.tolist()
        out, pos = [], 0
        for n in counts:
            out.append(DecodeResult(flat[pos:pos + n]))
            pos += n
        return out
    batch_size, maxlen = ctc_probs.shape[:2]
    topk_index = ctc_probs.argmax(dim=2)  # == topk(1): ties resolve to the lowest index in both
    mask = make_pad_mask(ctc_lens, maxlen)
    topk_index = topk_index.masked_fill(mask, blank_id)
    hyps = topk_index.tolist()  # ONE device->host copy, then the reference's collapse per utterance
    return [DecodeResult(remove_duplicates_and_blank(h, blank_id)) for h in hyps]


class _PrefixScore:
    """Blank-ending / non-blank-ending log-probabilities of a prefix (search.py:59-103, without the viterbi
    time stamps and the context graph, which the parity bar -- tokens and scores -- does not involve)."""
    __slots__ = ("s", "ns")

    def __init__(self, s: float = -float("inf"), ns: float = -float("inf")):
        self.s = s
        self.ns = ns

    def score(self) -> float:
        return log_add([self.s, self.ns])


def ctc_prefix_beam_search(ctc_probs: torch.Tensor, ctc_lens: torch.Tensor, beam_size: int, context_graph=None,
                           blank_id: int = 0) -> List[DecodeResult]:
    """search.py:124-248.  The per-frame top-`beam` tokens of the WHOLE batch are taken in one device op and
    copied to the host once (the reference calls .item() per candidate); the prefix bookkeeping then follows the
    reference's loop order exactly, because that order decides ties in its stable sort."""
    if context_graph is not None:
        raise NotImplementedError("context biasing is outside the accelerated path")
    B = ctc_probs.shape[0]
    k = min(beam_size, ctc_probs.shape[-1])
    top_p, top_i = ctc_probs.float().topk(k, dim=-1)          # (B, T, k)
    if ctc_probs.is_cuda and beam_size <= 16:
        # GPU-resident: one wave per utterance walks the frames (pafc_ctc_prefix_beam_search); only the n-best lists
        # come back.  Same candidates, merges and tie order as the loop below; float64 scores from the device's
        # exp / log agree with the host's to a few ulps.
        from ..hip_ops import ctc_prefix_beam
        toks, lens_n, scores = ctc_prefix_beam(top_p.contiguous(), top_i.contiguous(), ctc_lens.to(ctc_probs.device),
                                               beam_size, blank_id)
        lens_h, scores_h = lens_n.tolist(), scores.tolist()
        maxlen = max(1, int(lens_n.max()))
        toks_h = toks[:, :, :maxlen].tolist()
        results = []
        for b in range(B):
            nbest = [tuple(toks_h[b][n][:lens_h[b][n]]) for n in range(beam_size) if lens_h[b][n] >= 0]
            nsc = [scores_h[b][n] for n in range(beam_size) if lens_h[b][n] >= 0]
            results.append(DecodeResult(tokens=nbest[0], score=nsc[0], nbest=nbest, nbest_scores=nsc))
        return results
    top_p, top_i, lens = top_p.cpu().tolist(), top_i.cpu().tolist(), [int(v) for v in ctc_lens.tolist()]
    results = []
    for b in range(B):
        cur = [(tuple(), _PrefixScore(s=0.0, ns=-float("inf")))]
        for t in range(lens[b]):
            nxt = defaultdict(_PrefixScore)
            for prob, u in zip(top_p[b][t], top_i[b][t]):
                for prefix, ps in cur:
                    last = prefix[-1] if len(prefix) > 0 else None
                    if u == blank_id:
                        n = nxt[prefix]
                        n.s = log_add([n.s, ps.score() + prob])
                    elif u == last:
                        n1 = nxt[prefix]                       # *uu -> *u
                        n1.ns = log_add([n1.ns, ps.ns + prob])
                        n2 = nxt[prefix + (u,)]                # *u-u -> *uu
                        n2.ns = log_add([n2.ns, ps.s + prob])
                    else:
                        n = nxt[prefix + (u,)]
                        n.ns = log_add([n.ns, ps.score() + prob])
            cur = sorted(nxt.items(), key=lambda x: x[1].score(), reverse=True)[:beam_size]
        nbest = [list(y[0]) for y in cur]
        nbest_scores = [y[1].score() for y in cur]
        results.append(DecodeResult(tokens=tuple(nbest[0]), score=nbest_scores[0], nbest=[tuple(n) for n in nbest],
                                    nbest_scores=nbest_scores))
    return results
