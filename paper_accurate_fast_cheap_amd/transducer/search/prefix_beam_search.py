"""CTC-fused RNN-T prefix beam search (reference: wenet/transducer/search/prefix_beam_search.py:428-574, the
`prefix_beam_search_decode_batch` that `Transducer.beam_search_decode` reaches through :219-220).

Per frame t and utterance: one predictor step + joint for every live beam, log_softmax, shallow fusion
log(w_rnnt e^{rnnt} + w_ctc e^{ctc}), top-`beam` tokens per beam, candidates visited in descending score order,
equal hypotheses merged with log_add, stop once `beam` distinct hypotheses are collected, keep the best `beam`;
at most one symbol per frame.  Token ids are the bit-exact parity bar, so the candidate walk below follows the
reference statement by statement -- including its rounding points (beam scores go through float32 every frame)
and its early stop (later duplicates of an already-collected hypothesis are NOT merged once the beam is full).

What changed for the GPU: the reference reads every candidate with `.item()` (beam^2 x 3 host syncs per utterance
and frame) and concatenates per-beam LSTM states with torch.cat every frame.  Here the LSTM states of all beams
of all utterances stay in two batched device tensors that are re-indexed once per frame, and the host receives
ONE packed (scores, indices) block per frame for the whole batch."""
from typing import List, Optional

import torch

from ...transformer.search import DecodeResult, log_add


class Sequence:
    __slots__ = ("hyp", "score", "cache")

    def __init__(self, hyp: List[int], score: float, cache: int):
        self.hyp = hyp
        self.score = score
        self.cache = cache   # column of the batched state tensors that holds this beam's LSTM state


class PrefixBeamSearch:
    def __init__(self, encoder, predictor, joint, ctc, blank):
        self.encoder = encoder
        self.predictor = predictor
        self.joint = joint
        self.ctc = ctc
        self.blank = blank
        self.device_resident = True   # GPU tensors: keep the beams on the device (False: host bookkeeping, one copy per frame)
        self.use_graph = True         # device-resident path: replay the frame body from a captured hipGraph

    def forward_decoder_one_step(self, encoder_x: torch.Tensor, pre_t: torch.Tensor, cache: List[torch.Tensor]):
        padding = torch.zeros(pre_t.size(0), 1, device=encoder_x.device, dtype=cache[0].dtype)
        pre_t, new_cache = self.predictor.forward_step(pre_t.unsqueeze(-1), padding, cache)
        x = self.joint(encoder_x, pre_t)
        return x.log_softmax(dim=-1), new_cache

    @torch.no_grad()
    def prefix_beam_search_decode(self, encoder_outs, encoder_lens, ctc_probs, decoding_chunk_size: int = -1,
                                  beam_size: int = 5, num_decoding_left_chunks: int = -1,
                                  simulate_streaming: bool = False, ctc_weight: float = 0.3,
                                  transducer_weight: float = 0.7, cat_embs: Optional[torch.Tensor] = None):
        assert encoder_outs.shape[0] == encoder_lens.shape[0] == ctc_probs.shape[0]
        return self.prefix_beam_search_decode_batch(encoder_outs, encoder_lens, ctc_probs, decoding_chunk_size,
                                                    beam_size, num_decoding_left_chunks, simulate_streaming,
                                                    ctc_weight, transducer_weight, cat_embs)

    @torch.no_grad()
    def prefix_beam_search_decode_batch(self, encoder_outs, encoder_lens, ctc_probs, decoding_chunk_size: int = -1,
                                        beam_size: int = 5, num_decoding_left_chunks: int = -1,
                                        simulate_streaming: bool = False, ctc_weight: float = 0.3,
                                        transducer_weight: float = 0.7, cat_embs: Optional[torch.Tensor] = None):
        device = encoder_outs.device
        B = encoder_outs.shape[0]
        if self.device_resident and encoder_outs.is_cuda and beam_size <= 16 and B > 0 and encoder_outs.shape[1] > 0:
            return self._decode_batch_resident(encoder_outs, encoder_lens, ctc_probs, beam_size, ctc_weight,
                                               transducer_weight)
        lens = [int(v) for v in encoder_lens.tolist()]
        max_len = max(lens) if lens else 0
        state = self.predictor.init_state(B, method="zero", device=device)
        state = [s.to(encoder_outs.dtype) for s in state]        # column b = utterance b's single start beam
        beams = [[Sequence([self.blank], 0.0, b)] for b in range(B)]

        for t in range(max_len):
            active = [i for i in range(B) if t < lens[i]]
            if not active:
                break
            rows, toks, cols, scores = [], [], [], []
            for i in active:
                for s in beams[i]:
                    rows.append(i)
                    toks.append(s.hyp[-1])
                    cols.append(s.cache)
                    scores.append(s.score)
            n = len(rows)
            rows_t = torch.tensor(rows, device=device)
            cols_t = torch.tensor(cols, device=device)
            cache = [state[0].index_select(1, cols_t), state[1].index_select(1, cols_t)]
            enc = encoder_outs[rows_t, t, :].unsqueeze(1)                                  # (n, 1, D)
            logp, new_cache = self.forward_decoder_one_step(enc, torch.tensor(toks, device=device), cache)
            logp = logp.squeeze(1).squeeze(1)                                              # (n, V)
            logp = torch.log(torch.add(transducer_weight * torch.exp(logp),
                                       ctc_weight * torch.exp(ctc_probs[rows_t, t, :])))
            top_k_logp, top_k_index = logp.topk(beam_size)                                 # (n, beam)
            cand = torch.tensor(scores, device=device).unsqueeze(1) + top_k_logp          # float32, as the reference
            packed = torch.cat([cand.float(), top_k_index.float()], dim=1).cpu()           # ONE device->host copy
            cand_h = packed[:, :beam_size]
            idx_h = packed[:, beam_size:].to(torch.int64)
            # next frame's state pool: old states (kept by blank extensions) then new states
            state = [torch.cat([cache[0], new_cache[0]], dim=1), torch.cat([cache[1], new_cache[1]], dim=1)]

            cur = 0
            for i in active:
                nb = len(beams[i])
                flat = cand_h[cur:cur + nb].reshape(-1)
                toks_flat = idx_h[cur:cur + nb].reshape(-1).tolist()
                vals = flat.tolist()
                order = torch.argsort(flat, descending=True).tolist()                     # reference: :524
                beam_A: List[Sequence] = []
                seen = set()
                for k in order:
                    b_idx, tok, score = k // beam_size, toks_flat[k], vals[k]
                    base = beams[i][b_idx]
                    new_hyp = list(base.hyp) if tok == self.blank else base.hyp + [tok]
                    key = tuple(new_hyp)
                    if key in seen:
                        for ex in beam_A:
                            if ex.hyp == new_hyp:
                                ex.score = log_add([ex.score, score])
                                break
                    else:
                        seen.add(key)
                        beam_A.append(Sequence(new_hyp, score, (cur + b_idx) if tok == self.blank else (n + cur + b_idx)))
                        if len(beam_A) >= beam_size:
                            break
                beam_A.sort(key=lambda s: s.score, reverse=True)
                beams[i] = beam_A[:beam_size]
                cur += nb

        results = []
        for bs in beams:
            nbest = [b.hyp[1:] for b in bs]
            nbest_scores = [b.score for b in bs]
            results.append(DecodeResult(tokens=nbest[0], score=nbest_scores[0], nbest=nbest, nbest_scores=nbest_scores))
        return results

    @torch.no_grad()
    def _decode_batch_resident(self, encoder_outs, encoder_lens, ctc_probs, beam_size: int, ctc_weight: float,
                               transducer_weight: float):
        """The same search with the beams on the device (pafc_rnnt_beam_*): B x beam fixed slots, predictor step,
        joint, fusion and top-k as batched ops over all slots, the candidate walk in a kernel, LSTM states re-indexed
        by the kernel's output -- no host synchronisation until the n-best lists are read back."""
        from ...hip_ops import RnntBeamState
        device = encoder_outs.device
        B, T, _ = encoder_outs.shape
        n = B * beam_size
        lens64 = encoder_lens.to(device=device, dtype=torch.int64).contiguous()
        st = RnntBeamState(B, T, beam_size, self.blank, device)
        state = self.predictor.init_state(n, method="zero", device=device)
        cache = [s.to(encoder_outs.dtype).contiguous() for s in state]        # static buffers, updated in place
        t_dev = torch.zeros(1, dtype=torch.int64, device=device)
        ctc_probs = ctc_probs.contiguous()

        def frame():
            # one frame for all B x beam slots; every tensor it touches has a fixed address and shape, and the frame
            # index lives on the device (t_dev), so the body can be captured once and replayed
            enc = encoder_outs.index_select(1, t_dev.clamp(max=T - 1)).squeeze(1)
            enc = enc.repeat_interleave(beam_size, dim=0).unsqueeze(1)                              # (n, 1, D)
            logp, new_cache = self.forward_decoder_one_step(enc, st.last_tok, cache)
            logp = logp.squeeze(1).squeeze(1)                                                        # (n, V)
            ctc_t = ctc_probs.index_select(1, t_dev.clamp(max=T - 1)).squeeze(1).repeat_interleave(beam_size, dim=0)
            logp = torch.log(torch.add(transducer_weight * torch.exp(logp), ctc_weight * torch.exp(ctc_t)))
            top_val, top_idx = logp.topk(beam_size)
            st.step(0, lens64, top_val.float().contiguous(), top_idx.contiguous(), t_dev=t_dev)
            cache[0].copy_(torch.cat([cache[0], new_cache[0]], dim=1).index_select(1, st.next_idx))
            cache[1].copy_(torch.cat([cache[1], new_cache[1]], dim=1).index_select(1, st.next_idx))
            t_dev.add_(1)

        done = 0
        if self.use_graph and T >= 8:
            # launch-bound loop (~25 small kernels per frame): two eager frames warm every library handle, then the
            # body is captured into a hipGraph and replayed for the remaining frames.  MIOpen's RNN call is not
            # capturable (it sizes its workspace inside the call), so the LSTM runs through the framework's own cell.
            with torch.backends.cudnn.flags(enabled=False):
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    frame(); frame()
                torch.cuda.current_stream(device).wait_stream(side)
                done = 2
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph):
                        frame()
                except RuntimeError as e:
                    # only a REFUSED capture (an operation the stream capture does not permit in this build) finishes eagerly --
                    # nothing ran during the failed capture, so t_dev still stands behind the two warm-up frames; a failing
                    # launch or a PafcError inside the frame is a real error and surfaces
                    from ..._lib import PafcError
                    torch.cuda.synchronize(device)
                    if isinstance(e, PafcError) or "captur" not in str(e).lower():
                        raise
                    graph = None
                if graph is not None:          # errors of the replays are genuine kernel / launch errors: not swallowed
                    for _ in range(T - done):
                        graph.replay()
                    done = T
        for _ in range(T - done):
            frame()
        toks, lens_n, scores = st.finish()
        lens_h, scores_h = lens_n.tolist(), scores.tolist()
        maxlen = max(1, int(lens_n.max()))
        toks_h = toks[:, :, :maxlen].tolist()
        results = []
        for b in range(B):
            nbest = [toks_h[b][k][:lens_h[b][k]] for k in range(beam_size) if lens_h[b][k] >= 0]
            nsc = [scores_h[b][k] for k in range(beam_size) if lens_h[b][k] >= 0]
            results.append(DecodeResult(tokens=nbest[0], score=nsc[0], nbest=nbest, nbest_scores=nsc))
        return results
