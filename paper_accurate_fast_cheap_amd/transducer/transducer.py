"""Transducer container around the accelerated encoder (reference: wenet/transducer/transducer.py): encoder + CTC +
RNN predictor + joint, `decode(methods=[...,'rnnt_beam_search'])` (:695-813) and `beam_search_decode` (:644-693).

Training objective (forward, :105-175): transducer_weight * RNN-T loss + ctc_weight * CTC loss over the accelerated
encoder.  The reference's RNN-T loss is the third-party `optimized_transducer.transducer_loss` (transducer.py:506-523;
Rev fork, unpinned): restated from the published definition in `loss.py` -- PARITY UNPINNED.  Its attention decoder was
not released (decoder.py is swallowed by .gitignore:44), so the attention branch does not exist here."""
from typing import Dict, List, Optional

import torch

from ..transformer.asr_model import ASRModel
from ..transformer.search import DecodeResult
from .loss import transducer_loss
from .search.prefix_beam_search import PrefixBeamSearch

IGNORE_ID = -1


def add_blank(ys_pad: torch.Tensor, blank: int, ignore_id: int) -> torch.Tensor:
    """wenet/utils/common.py:78-107: prepend <blank>, padding (ignore_id) -> blank: (B, L) -> (B, L + 1)."""
    bs = ys_pad.size(0)
    _blank = torch.full((bs, 1), blank, dtype=ys_pad.dtype, device=ys_pad.device)
    out = torch.cat([_blank, ys_pad], dim=1)
    return torch.where(out == ignore_id, blank, out)


class Transducer(ASRModel):
    def __init__(self, vocab_size: int, blank: int, encoder: torch.nn.Module, predictor: torch.nn.Module,
                 joint: torch.nn.Module, ctc=None, special_tokens: Optional[dict] = None, attention_decoder=None,
                 ctc_weight: float = 0.0, transducer_weight: float = 1.0, attention_weight: float = 0.0,
                 **_unused_model_conf):
        super().__init__(vocab_size, encoder, ctc, ctc_weight, special_tokens)
        self.blank = blank
        self.predictor = predictor
        self.joint = joint
        self.transducer_weight = transducer_weight
        self.attention_decoder_weight = attention_weight
        self.bs: Optional[PrefixBeamSearch] = None

    def forward(self, batch: dict, device: torch.device) -> Dict[str, Optional[torch.Tensor]]:
        """transducer.py:105-175 without the attention decoder: encoder -> RNN-T loss (+ ctc_weight * CTC)."""
        speech = batch["feats"].to(device)
        speech_lengths = batch["feats_lengths"].to(device)
        text = batch["target"].to(device)
        text_lengths = batch["target_lengths"].to(device)
        assert speech.shape[0] == speech_lengths.shape[0] == text.shape[0] == text_lengths.shape[0]
        encoder_out, encoder_mask = self.encoder(speech, speech_lengths)
        encoder_out_lens = encoder_mask.squeeze(1).sum(1)
        loss_rnnt = self._compute_loss(encoder_out, encoder_out_lens, text, text_lengths)
        loss = self.transducer_weight * loss_rnnt
        loss_ctc = None
        if self.ctc_weight != 0.0 and self.ctc is not None:
            loss_ctc = self.ctc.loss(encoder_out.float(), encoder_out_lens, text, text_lengths)
            loss = loss + self.ctc_weight * loss_ctc.sum()
        return {"loss": loss, "loss_att": None, "loss_ctc": loss_ctc, "loss_rnnt": loss_rnnt, "th_accuracy": -1.0}

    def _compute_loss(self, encoder_out, encoder_out_lens, text, text_lengths) -> torch.Tensor:
        """transducer.py:525-561 (optimized_transducer branch): predictor over blank-prepended targets, joint on the
        valid lattices only, loss with reduction "mean"."""
        ys_in_pad = add_blank(text, self.blank, IGNORE_ID)
        predictor_out = self.predictor(ys_in_pad)
        rnnt_text = torch.where(text == IGNORE_ID, 0, text.to(torch.int64)).to(torch.int32)
        joint_out = self.joint.forward_optimized(encoder_out.to(predictor_out.dtype), predictor_out,
                                                 encoder_out_lens.to(torch.int32), text_lengths.to(torch.int32))
        return transducer_loss(joint_out, rnnt_text, encoder_out_lens, text_lengths, self.blank, reduction="mean",
                               from_log_softmax=False)

    def init_bs(self):
        if self.bs is None:
            self.bs = PrefixBeamSearch(self.encoder, self.predictor, self.joint, self.ctc, self.blank)

    def beam_search_decode(self, encoder_outs, encoder_lens, ctc_probs, decoding_chunk_size: int = -1,
                           beam_size: int = 5, num_decoding_left_chunks: int = -1, simulate_streaming: bool = False,
                           ctc_weight: float = 0.3, transducer_weight: float = 0.7, cat_embs=None) -> List[DecodeResult]:
        self.init_bs()
        return self.bs.prefix_beam_search_decode(encoder_outs, encoder_lens, ctc_probs, decoding_chunk_size, beam_size,
                                                 num_decoding_left_chunks, simulate_streaming, ctc_weight,
                                                 transducer_weight, cat_embs)

    @torch.no_grad()
    def decode(self, methods: List[str], speech: torch.Tensor, speech_lengths: torch.Tensor, beam_size: int = 10,
               decoding_chunk_size: int = -1, num_decoding_left_chunks: int = -1, ctc_weight: float = 0.0,
               transducer_weight: float = 0.0, simulate_streaming: bool = False, reverse_weight: float = 0.0,
               context_graph=None, blank_id: int = 0, blank_penalty: float = 0.0, cat_embs=None, **_ignored
               ) -> Dict[str, List[DecodeResult]]:
        rest = [m for m in methods if m != "rnnt_beam_search"]
        encoder_out, encoder_mask = self._forward_encoder(speech, speech_lengths, decoding_chunk_size,
                                                          num_decoding_left_chunks, simulate_streaming, cat_embs)
        encoder_lens = encoder_mask.squeeze(1).sum(1)
        ctc_probs = self.ctc_logprobs(encoder_out, blank_penalty, blank_id)
        results = {}
        for m in rest:
            if m == "ctc_greedy_search":
                from ..transformer.search import ctc_greedy_search
                results[m] = ctc_greedy_search(ctc_probs, encoder_lens, blank_id)
            elif m == "ctc_prefix_beam_search":
                from ..transformer.search import ctc_prefix_beam_search
                results[m] = ctc_prefix_beam_search(ctc_probs, encoder_lens, beam_size, context_graph, blank_id)
            else:
                raise NotImplementedError(f"decode mode {m!r} is outside the accelerated path")
        if "rnnt_beam_search" in methods:
            results["rnnt_beam_search"] = self.beam_search_decode(
                encoder_outs=encoder_out, encoder_lens=encoder_lens, ctc_probs=ctc_probs, beam_size=beam_size,
                ctc_weight=ctc_weight, transducer_weight=transducer_weight, cat_embs=cat_embs)
        return results
