"""Long-form decoding by fixed-size windows (reference: wenet/bin/recognize_wav2.py:323-351 `feats_batcher`, the decode
loop :431-470, and the window time shift of `hyps_to_ctm` :523-548): a long file is cut into windows of `chunk_size`
input frames, `batch_size` windows form one batch, the last window is zero-padded and its length shortened; the
hypotheses of the windows are concatenated in order and every window starts `chunk_size * frame_shift` later than the
previous one.  This is how the bidirectional encoder runs on audio of arbitrary length without any state carry (the
uni-directional one can stream instead: encoder.forward_chunk_carry / stream_chunks).

The word-level CTM of the reference additionally needs `wenet/bin/ctc_align.py` (`ctc_align`,
`adjust_model_time_offset`, recognize_wav2.py:41), which is not part of the released tree; what can be restated is the
token-level stitching below: token ids with the start time of their first frame."""
import contextlib
import math
from typing import Dict, Iterator, List, Optional, Tuple

import torch
import torch.nn.functional as F


def feats_batcher(infeats: torch.Tensor, chunk_size: int, batch_size: int, device=None
                  ) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
    """(1, T, F) features -> batches ((n_windows, chunk_size, F), int32 lengths (n_windows,)); n_windows == batch_size
    except possibly in the last batch; only the very last window can be shorter than chunk_size (zero-padded)."""
    assert infeats.dim() == 3 and infeats.shape[0] == 1 and chunk_size > 0 and batch_size > 0
    T, nf = infeats.shape[1], infeats.shape[2]
    per = chunk_size * batch_size
    for b in range(math.ceil(T / per)):
        fb = infeats[:, b * per:(b + 1) * per, :]
        nw = math.ceil(fb.shape[1] / chunk_size)
        lens = torch.full((nw,), chunk_size, dtype=torch.int32)
        pad = nw * chunk_size - fb.shape[1]
        if pad > 0:
            lens[-1] -= pad
            fb = F.pad(fb, (0, 0, 0, pad, 0, 0), mode="constant", value=0)
        yield fb.reshape(nw, chunk_size, nf), (lens.to(device) if device is not None else lens.to(infeats.device))


def window_offsets_ms(n_windows: int, chunk_size: int, input_frame_ms: float = 10.0) -> List[float]:
    """Start time of every window (hyps_to_ctm: `time_shift_ms += chunk_size * input_frame_length` per window)."""
    return [i * chunk_size * input_frame_ms for i in range(n_windows)]


def merged_batch_size(chunk_size: int, batch_size: int, merge_frames: int) -> int:
    """Windows per launch when consecutive batches are merged: the largest whole multiple of batch_size whose windows hold at
    most merge_frames input frames, never fewer than batch_size (merge_frames <= 0: batch_size, one forward per batch)."""
    if merge_frames <= 0:
        return batch_size
    return max(1, merge_frames // (chunk_size * batch_size)) * batch_size


_SIDE = {}


def _multi_stream_safe(model) -> bool:
    """Several batches in flight on side streams only for models whose whole pass runs on this package's kernels
    (BaseEncoder.multi_stream_safe): the framework's fp32 library GEMM deadlocks under concurrent streams."""
    enc = getattr(model, "encoder", None)
    fn = getattr(enc, "multi_stream_safe", None)
    return bool(fn()) if callable(fn) else False



def _side_streams(device: torch.device, n: int):
    """The same n side streams for every call on a device: the encoder's graph cache is keyed by (shape, stream), so fresh
    streams per file would mean fresh captures per file."""
    have = _SIDE.setdefault(torch.device(device).index or 0, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


@torch.no_grad()
def greedy_decode_batches(model, batches, streams: int = 2, blank_id: int = 0, want_tokens: bool = True):
    """One pass over decode batches [(feats (B, T, F), lengths (B,)), ...] already resident on the GPU: encoder + CTC
    log-softmax (+ greedy search, search.py:106-121) per batch, `streams` batches in flight on HIP streams of their own
    (the launch-bound stretches of one batch overlap the other's), the token lists fetched once after the last batch is queued
    (`ctc_greedy_search(defer=True)`: nothing in the loop waits for the device).  The decode loop of recognize.py around
    `model.decode` (wenet/bin/recognize.py, `--batch_size`), without its per-batch host round trip.
    Returns (token lists per batch -- List[List[DecodeResult]] -- or None, log-probabilities of the last batch).
    Give the batches longest first (utils.sharding.decode_batches): see there."""
    from ..transformer.search import ctc_greedy_search
    if not batches:
        return ([] if want_tokens else None), None
    device = batches[0][0].device
    n_side = max(1, min(int(streams), len(batches)))
    if n_side > 1 and not _multi_stream_safe(model):
        n_side = 1                   # a pass that calls the framework's library GEMMs must not overlap another one (see there)
    main = torch.cuda.current_stream(device)
    side = _side_streams(device, n_side) if n_side > 1 else []
    for s_ in side:
        s_.wait_stream(main)
    pending, logp = [], None
    for i, (fb, lens) in enumerate(batches):
        with (torch.cuda.stream(side[i % n_side]) if side else contextlib.nullcontext()):
            enc, mask = model._forward_encoder(fb, lens)
            logp = model.ctc_logprobs(enc)
            if want_tokens:
                pending.append(ctc_greedy_search(logp, mask.squeeze(1).sum(1), blank_id, defer=True))
    for s_ in side:
        main.wait_stream(s_)
    return ([f() for f in pending] if want_tokens else None), logp


@torch.no_grad()
def decode_windows(model, feats: torch.Tensor, chunk_size: int, batch_size: int, mode: str = "ctc_greedy_search",
                   beam_size: int = 10, input_frame_ms: float = 10.0, output_frame_ms: float = 40.0, streams: int = 3,
                   graph_cache: bool = True, merge_frames: int = 0, **decode_kw) -> Dict[str, object]:
    """Decode a long file window by window with `model.decode` and stitch the token sequences.

    Returns {"tokens": all token ids in order, "windows": per-window token lists, "window_start_ms": start time of each
    window, "token_start_ms": start time of every token when the search reports frame indices (GPU greedy search),
    else None}.  Shards naturally: give each rank a contiguous range of batches (utils/sharding.py).

    GPU greedy search: `streams` window batches are in flight on HIP streams of their own, each replayed from the encoder's
    hipGraph of its (shape, stream) once the shape has been seen twice on that stream (`graph_cache`) -- a batch of 2 000-frame windows is a string of ~300 kernels of 10-20 us, which one
    stream cannot keep the chip busy with (one box, 8 x 2 000 frames per batch: 1 / 2 / 3 / 4 in flight = 38 000 / 53 300 /
    59 900 / 50 400 audio-sec/sec) -- and the token lists come back once, after the last batch is queued."""
    windows: List[List[int]] = []
    frames: List[Optional[List[int]]] = []
    if mode == "ctc_greedy_search" and feats.is_cuda:
        # same search, plus the first-frame index of every token (the kernel reports it for free)
        from ..hip_ops import ctc_greedy
        batch_size = merged_batch_size(chunk_size, batch_size, merge_frames)
        batches = list(feats_batcher(feats, chunk_size, batch_size, feats.device))
        n_side = max(1, min(int(streams), len(batches)))
        if n_side > 1 and not _multi_stream_safe(model):
            n_side = 1               # a pass that calls the framework's library GEMMs must not overlap another one
        main = torch.cuda.current_stream(feats.device)
        side = _side_streams(feats.device, n_side) if n_side > 1 else []
        for s_ in side:
            s_.wait_stream(main)                   # the batches above were cut on the caller's stream
        enc_mod = getattr(model, "encoder", None)
        if graph_cache and getattr(enc_mod, "graph_cache_size", None) is not None and len(batches) > 1:
            # the window shape recurs batch after batch and file after file: the encoder keeps a graph per (shape, stream).  The
            # setting is left in place for the next file (the graphs pin their activations: encoder.graph_cache_size = 0 and
            # encoder._graphs.clear() release them)
            enc_mod.graph_cache_size = max(enc_mod.graph_cache_size, 2 * n_side)    # (full batches, the last one) x streams
        pending = []
        for i, (fb, lens) in enumerate(batches):
            with (torch.cuda.stream(side[i % n_side]) if side else contextlib.nullcontext()):
                enc, mask = model._forward_encoder(fb, lens)
                logp = model.ctc_logprobs(enc)
                pending.append(ctc_greedy(logp.contiguous(), mask.squeeze(1).sum(1), decode_kw.get("blank_id", 0),
                                          want_frames=True))
        for s_ in side:
            main.wait_stream(s_)
        for tk, nt, fr in pending:
            tk, nt, fr = tk.cpu(), nt.cpu(), fr.cpu()
            for i in range(tk.shape[0]):
                n = int(nt[i])
                windows.append(tk[i, :n].tolist())
                frames.append(fr[i, :n].tolist())
    else:
        for fb, lens in feats_batcher(feats, chunk_size, batch_size, feats.device):
            res = model.decode([mode], fb, lens, beam_size=beam_size, **decode_kw)[mode]
            for r in res:
                windows.append(list(r.tokens))
                frames.append(None)
    starts = window_offsets_ms(len(windows), chunk_size, input_frame_ms)
    tokens = [t for w in windows for t in w]
    tstart = None
    if all(f is not None for f in frames):
        tstart = [s + f * output_frame_ms for s, fr in zip(starts, frames) for f in fr]
    return {"tokens": tokens, "windows": windows, "window_start_ms": starts, "token_start_ms": tstart}
