"""Encoder + CTC container exposing what the hot path's callers use on the reference's ASRModel / Transducer:
`.encoder`, `.ctc`, `_forward_encoder` (asr_model.py:294-321), `ctc_logprobs` (:324-335) and `decode` for the
CTC modes (:337-440).  The attention decoder, RNN-T predictor/joint and their losses are containers around the
path, not the path; transducer decoding joins in paper_accurate_fast_cheap_amd/transducer."""
from typing import Dict, List, Optional, Tuple

import torch

from .ctc import CTC
from .search import DecodeResult, ctc_greedy_search


class ASRModel(torch.nn.Module):
    def __init__(self, vocab_size: int, encoder: torch.nn.Module, ctc: CTC, ctc_weight: float = 1.0,
                 special_tokens: Optional[dict] = None, **_unused_model_conf):
        super().__init__()
        self.vocab_size = vocab_size
        self.encoder = encoder
        self.ctc = ctc
        self.ctc_weight = ctc_weight
        self.special_tokens = special_tokens
        self.sos = (vocab_size - 1) if special_tokens is None else special_tokens.get("<sos>", vocab_size - 1)
        self.eos = (vocab_size - 1) if special_tokens is None else special_tokens.get("<eos>", vocab_size - 1)

    def forward(self, batch: dict, device: torch.device) -> Dict[str, Optional[torch.Tensor]]:
        """CTC-only training objective over the accelerated encoder (the hybrid losses live outside the path)."""
        speech = batch["feats"].to(device)
        speech_lengths = batch["feats_lengths"].to(device)
        text = batch["target"].to(device)
        text_lengths = batch["target_lengths"].to(device)
        encoder_out, encoder_mask = self.encoder(speech, speech_lengths)
        encoder_out_lens = encoder_mask.squeeze(1).sum(1)
        loss_ctc = self.ctc.loss(encoder_out.float(), encoder_out_lens, text, text_lengths)
        return {"loss": loss_ctc, "loss_ctc": loss_ctc}

    def _forward_encoder(self, speech: torch.Tensor, speech_lengths: torch.Tensor, decoding_chunk_size: int = -1,
                         num_decoding_left_chunks: int = -1, simulate_streaming: bool = False,
                         cat_embs: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if simulate_streaming and decoding_chunk_size > 0:
            return self.encoder.forward_chunk_by_chunk(speech, decoding_chunk_size=decoding_chunk_size,
                                                       num_decoding_left_chunks=num_decoding_left_chunks)
        return self.encoder(speech, speech_lengths, decoding_chunk_size=decoding_chunk_size,
                            num_decoding_left_chunks=num_decoding_left_chunks, cat_embs=cat_embs)

    def ctc_logprobs(self, encoder_out: torch.Tensor, blank_penalty: float = 0.0, blank_id: int = 0) -> torch.Tensor:
        if blank_penalty > 0.0:
            logits = self.ctc.ctc_lo(encoder_out)
            logits[:, :, blank_id] -= blank_penalty
            return logits.log_softmax(dim=2)
        return self.ctc.log_softmax(encoder_out)

    @torch.no_grad()
    def decode(self, methods: List[str], speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int = 10,
               decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1, ctc_weight: float = 0.0,
               simulate_streaming: bool = False, reverse_weight: float = 0.0, blank_id: int = 0,
               blank_penalty: float = 0.0, cat_embs: Optional[torch.Tensor] = None, **_ignored
               ) -> Dict[str, List[DecodeResult]]:
        assert speech.shape[0] == speech_lengths.shape[0]
        encoder_out, encoder_mask = self._forward_encoder(speech, speech_lengths, decoding_chunk_size,
                                                          num_decoding_left_chunks, simulate_streaming, cat_embs)
        encoder_lens = encoder_mask.squeeze(1).sum(1)
        ctc_probs = self.ctc_logprobs(encoder_out, blank_penalty, blank_id)
        results = {}
        for m in methods:
            if m == "ctc_greedy_search":
                results[m] = ctc_greedy_search(ctc_probs, encoder_lens, blank_id)
            elif m == "ctc_prefix_beam_search":
                from .search import ctc_prefix_beam_search
                results[m] = ctc_prefix_beam_search(ctc_probs, encoder_lens, beam_size, blank_id=blank_id)
            else:
                raise NotImplementedError(f"decode mode {m!r} is outside the accelerated path")
        return results
